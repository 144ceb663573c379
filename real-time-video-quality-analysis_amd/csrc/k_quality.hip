// k_quality.hip — PSNR's squared-error reduction and SSIM for gfx950.
//
// Reference function replaced: run_ffmpeg_metrics, video_processing.py:270-297
// (FFmpeg `psnr` and `ssim` filters on each 8-bit plane of the two inputs).
//
//   k_ssim_gauss : SSE + Gaussian-windowed SSIM (11x11, sigma 1.5, K1 .01, K2 .03,
//                  valid region) — the window BASELINE.json's north_star decrees.
//   k_ssim_ffmpeg_fast / k_ssim_ffmpeg: SSE + FFmpeg vf_ssim's integer 8x8-window / stride-4 SSIM — what
//                  the reference's subprocess actually computes (:276).  _fast (dword loads + v_dot4) serves
//                  4-byte-aligned planar planes and packed BGR24; the byte kernel is the general path for
//                  every other layout (odd offsets / strides, other pixel steps) — not an A/B variant.
//
// k_ssim_gauss design (the shipped kernel is k_ssim_gauss_p2<256, 8>, below).  The separable window needs 2 x 11
// taps on 4 moment maps (E[x], E[y], E[x^2+y^2], E[xy]) = 88 FMA per pixel, ~115 VALU ops per pixel in total
// against 2 bytes of input: ~57 op/B where the chip balances at 78.6 T lane-op/s / 8 TB/s ~ 10 op/B.  The kernel
// is VALU-bound by construction (at 100 % of the fp32 vector peak it would reach 0.22 of the HBM roof); the design
// removes everything else:
//   * a workgroup of 256 threads owns 246 output columns of a strip of rows; every thread owns ONE input column
//     and marches DOWN the strip.  The vertical pass is a rolling file of accumulators in registers - a ring of
//     12 slots (11 live output rows + 1 spare so that the 8 rows of a step divide it), unrolled so slot indices
//     are static: each input pixel is loaded (buffer_load_ubyte, scalar row offset), converted and multiplied
//     once, there is no vertical halo recompute inside a strip;
//   * 8 vertically filtered rows (4 maps, one float4 per column) go through LDS per round trip - a single buffer
//     of 8 rows x 257 float4 (33 KB, 4 waves per SIMD) with two barriers per step (before the stores: the previous
//     step's readers are done; before the reads).  For the horizontal pass a GROUP OF 8 LANES shares 8 adjacent
//     columns: lane 8 j + i filters columns 8 j .. 8 j + 7 of the step's i-th row and reads the 18 columns
//     8 j .. 8 j + 17 once (2.25 ds_read_b128 per output pixel instead of 11: LDS reads were 19 % of the launch's
//     energy, and the kernel runs at the board's power limit - DESIGN.md section 5);
//   * rows of the LDS buffer are 257 float4 apart, so the lanes of a group sit in different bank groups and every
//     ds_read_b128 / ds_write_b128 is conflict-free; the next row's pixels are loaded before the current row's
//     arithmetic;
//   * samples are centred (x - 128) so the variance terms lose 4x less to fp32 cancellation; SSE is exact integer
//     arithmetic (flushed from fp32 partial sums every <= 240 rows, while they are still exact).
// The one-row-per-barrier kernel of rounds 1-3 (k_ssim_gauss, two adjacent columns per thread) survives only in
// the lab build (VQA_SSIM_VARIANT), for re-measurement.
// Algorithmic HBM bytes: 2P per plane pair.
#include "vqa_dev.hpp"
#include "vqa_kernels.hpp"
#include "vqa_math.hpp"

#include <cstdlib>

namespace vqa {

// Row strips per plane.  Long strips amortise the 10-row vertical warm-up; but the chip holds ~1536
// of these 256-thread workgroups at once (6 waves/SIMD), so a launch wants >= ~4 residency rounds of
// workgroups or its tail runs on a part-empty chip.  strips = what the height needs (<= ~1100 rows each)
// or what the workgroup count needs, whichever is larger, with strips never shorter than 64 rows.
static inline int ssim_max_strips(int h) { const int s = (h - 10) / 64; return s < 1 ? 1 : s; }
static inline int ssim_strips(int h, long long groups /* workgroups per strip row: n * planes * column blocks */)
{
    int ns = (h - 10 + 1099) / 1100;
    const long long want = (6144 + groups - 1) / (groups > 0 ? groups : 1);
    if (want > ns) ns = (int)(want > 4096 ? 4096 : want);
    const int cap = ssim_max_strips(h);
    return ns > cap ? cap : ns;
}
#ifndef SSIM_FIXED_SUM
#define SSIM_FIXED_SUM 1 // 1 (shipped): the SSIM map is summed in 2^-27 fixed point - integer sums are associative, so a plane's mean
                         // does not depend on how the launch cut it into strips (= on the batch size); 0: measurement build, the
                         // float / double sums of rounds 1-5 (<= 1e-8 relative apart between batch geometries)
#endif
#ifndef SSIM_ROWS
#define SSIM_ROWS 8   // rows per LDS round trip of k_ssim_gauss_p2 (8 = shipped; 2, 4 = measurement builds)
#endif
// rows a strip owns: a multiple of 6 * SSIM_ROWS (k_ssim_gauss_p2 consumes rows in groups of that size and owns whole
// groups); in the lab build also of 11 (its one-row kernel owns groups of 11)
static inline int ssim_strip_rows(int h, int ns)
{
#ifdef VQA_AB_VARIANTS
    const int m = 66 * (SSIM_ROWS == 8 ? 4 : SSIM_ROWS);
#else
    const int m = (SSIM_ROWS == 8 ? 3 : 6) * SSIM_ROWS;
#endif
    return ((h - 10 + ns - 1) / ns + m - 1) / m * m;
}

typedef float f2 __attribute__((ext_vector_type(2)));

// 11-tap Gaussian, sigma 1.5, normalised in double and rounded to float (oracle: vqo_gauss11)
__host__ __device__ constexpr float gw(int k)
{
    return (k < 0 || k > 10) ? 0.f
           : (k == 0 || k == 10) ? 1.028380124e-03f
           : (k == 1 || k == 9) ? 7.598758209e-03f
           : (k == 2 || k == 8) ? 3.600077331e-02f
           : (k == 3 || k == 7) ? 1.093606874e-01f
           : (k == 4 || k == 6) ? 2.130055428e-01f
                                : 2.660117149e-01f;
}

// o0 = (mx, my): means of (x-128), (y-128); o1 = (sq, xy) = (E[(x-128)^2 + (y-128)^2], E[(x-128)(y-128)]).
// Written on register PAIRS so the whole formula is 13 vector instructions: the (den, num) factors of the
// luminance and the contrast/structure terms ride in the two halves of packed operations.
// The uncentred means u = m + 128 ARE formed (exactly: |m| <= 128): round 4 tried ux uy = mx my + 128 (mx + my) + 128^2
// to save three instructions and lost the 1e-4 bar on full-white against full-black (2.5e-4: the luminance numerator
// 2 ux uy + C1 = 6.5 came out of terms of size 32768) - caught by tests/golden/skimage_pins.json, and the kernel is
// power-bound, not issue-bound, so the instructions bought nothing (LAB_NOTES.md).
__device__ __forceinline__ float ssim_centered(f2 o0, f2 o1)
{
    const float C1 = 6.5025f, C2 = 58.5225f;
    const f2 mm = o0 * o0;                                    // (mx^2, my^2)
    const f2 t = f2{mm.x + mm.y, o0.x * o0.y};                // (mx^2 + my^2, mx my)
    const f2 vc = o1 - t;                                     // (var_x + var_y, cov)
    const f2 u = o0 + f2{128.f, 128.f};                       // (ux, uy)
    const f2 uu = u * u;
    const f2 q = f2{uu.x + uu.y, u.x * u.y};                  // (ux^2 + uy^2, ux uy)
    const f2 A = __builtin_elementwise_fma(q, f2{1.f, 2.f}, f2{C1, C1});   // (den, num) of the luminance term
    const f2 B = __builtin_elementwise_fma(vc, f2{1.f, 2.f}, f2{C2, C2});  // (den, num) of the contrast/structure term
    const f2 nd = A * B;
    return nd.y * __builtin_amdgcn_rcpf(nd.x);
}

// Up to 4 planes of identical geometry (e.g. the B, G, R channels of packed BGR24) are
// handled by ONE launch.  Workgroups are dealt round-robin over the 8 XCDs (each with its own L2),
// so the blocks b, b+8, b+16 .. share an XCD: the plane index is therefore taken from (b / 8) % count,
// which puts the workgroups that read the same cache lines (the channels of one tile) on the SAME
// XCD, back to back (placement is a speed matter only).
struct plane_group {
    int64_t offset[4];
    int plane_index[4];
    int count;
};

#ifdef VQA_AB_VARIANTS // round 3's kernel (one row per barrier): lab build only
// One column per thread; the four moment maps ride in two float2 registers
// (E[x],E[y]) and (E[x^2+y^2],E[xy]), so every tap is two v_pk_fma_f32.  The rolling
// accumulator file is 11 slots x 2 float2 = 44 VGPRs (80 VGPRs in all: 6 waves/SIMD).
// PF = how many rows ahead the pixel loads run.
// grid = (ncb * nstrips * group.count, n_frames)
#ifndef SSIM_PROBE
#define SSIM_PROBE 0   // measurement builds only (scripts/build_probes.sh SSIM_PROBE 1 2 3 ...): see LAB_NOTES.md
#endif
#if SSIM_PROBE == 3
#define SSIM_WAVES __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define SSIM_WAVES
#endif
template <int QT, int PF, int HPF>
__global__ __launch_bounds__(QT) SSIM_WAVES void k_ssim_gauss(const uint8_t *__restrict__ ref, const uint8_t *__restrict__ dist,
                                                   int64_t ref_fs, int64_t dist_fs, plane_group g, int64_t row_stride,
                                                   int step, int w, int h, int ncb, int nstrips, int QS,
                                                   double *__restrict__ partials, int64_t partial_plane_stride,
                                                   int n_planes, vqa_plane_metrics *__restrict__ res)
{
    constexpr int QOUT = QT - 10; // output columns per workgroup
    __shared__ float4 vb[2][QT];  // [row buffer][column] = (E[x], E[y], E[x^2+y^2], E[xy]) after the vertical pass
    __shared__ double red[4];
    __shared__ unsigned long long redu[4];
    const int f = blockIdx.y;
    const int t = threadIdx.x;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int ch = seq % g.count, tile = (seq / g.count) * 8 + xcd;
    const int bpp = ncb * nstrips;
    if (tile >= bpp) return; // grid is padded to a multiple of 8 tiles
    const int cb = tile % ncb, sb = tile / ncb;
    const int xs = cb * QOUT, ys = sb * QS;
    const int xin = xs + t;
    const int ow = w - 10;
    const int nrows = min(QS + 10, h - ys);
    const bool last_cb = cb == ncb - 1, last_sb = sb == nstrips - 1;
    const bool in_c = xin < w;
    const bool own_c = in_c && (last_cb || t < QOUT);
    const bool out_c = (t < QOUT) && (xin < ow);
    // wave-uniform row base (scalar registers) + a per-thread 32-bit column offset that never changes: the
    // row-to-row address arithmetic stays on the scalar unit, the vector ALUs only see the loads
    const int64_t base = g.offset[ch] + (int64_t)ys * row_stride;
    const uint8_t *rbase = ref + (int64_t)f * ref_fs + base;
    const uint8_t *dbase = dist + (int64_t)f * dist_fs + base;
    const uint32_t coff = (uint32_t)((in_c ? xin : 0) * step);

    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 acc[11]; // .xy = (E[x], E[y]) column sums, .zw = (E[x^2+y^2], E[xy]): one ds_write_b128 per row, no repacking
#pragma unroll
    for (int s = 0; s < 11; s++) acc[s] = f4{0.f, 0.f, 0.f, 0.f};
    float ssim_acc = 0.f;
    uint32_t sse_acc = 0;
    // SSE = sum (x - y)^2 = sum (x^2 + y^2) - 2 sum x y, from the centred moments already in registers: two FMAs per
    // pixel, exact in fp32 (integer sums below 2^24: flushed into sse_acc every 242 rows)
    float sse_sq = 0.f, sse_xy = 0.f;
    float own_f = own_c ? 1.f : 0.f;   // 0: a halo column (it belongs to the right neighbour block) or, below, a halo row

    // Pixel loads run one row ahead of the arithmetic (PF; depths 1..4 measured the same).  They are BUFFER loads: a
    // scalar resource per plane, the lane's constant column offset as the vector offset and the row offset in a scalar
    // register, so the vector ALUs spend nothing on addresses (the global_load form cost one v_lshl_add_u64 per load).
    static_assert(PF == 1 || PF == 2, "prefetch depth");
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void *)rbase, (short)0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t dres = __builtin_amdgcn_make_buffer_rsrc((void *)dbase, (short)0, -1, 0x00020000);
    auto ld = [&](const __amdgpu_buffer_rsrc_t &rs, int rr) -> uint32_t {
        return (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rs, coff, (int)((int64_t)rr * row_stride), 0);
    };
    uint32_t nr = ld(rres, 0), nd = ld(dres, 0);   // row 0; row r + 1 is requested while row r is being consumed

    // One input row: p = r mod 11 (a compile-time constant after unrolling: static accumulator slots), emit = the row
    // completes an output row (r >= 10; wave-uniform).
    auto row = [&](const int p, const int r, const bool emit) {
        const uint32_t cr = nr, cd = nd;
        {
            const int rr = min(r + 1, nrows - 1);
            nr = ld(rres, rr);
            nd = ld(dres, rr);
        }
        // u8 -> centred float without a conversion instruction: 0x4B000000 | v is the float 2^23 + v, and
        // subtracting 2^23 + 128 is exact (v_or_b32 + v_sub_f32 issue at full rate, v_cvt_f32_ubyte at half)
        const f2 xy = f2{__uint_as_float(0x4B000000u | cr), __uint_as_float(0x4B000000u | cd)} - f2{8388736.f, 8388736.f};
        f2 v1 = xy.xx * xy;              // (x^2, x y)
        v1.x = fmaf(xy.y, xy.y, v1.x);   // (x^2 + y^2, x y): one scalar FMA, no operand repacking
        sse_sq = fmaf(own_f, v1.x, sse_sq);
        sse_xy = fmaf(own_f, v1.y, sse_xy);
        // vertical pass: input row r is tap k of output row r-k, kept in slot (r-k) mod 11
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const int s = (p - k + 11) % 11;
            if (k == 0) {
                acc[s].xy = gw(0) * xy;
                acc[s].zw = gw(0) * v1;
            } else {
                acc[s].xy = __builtin_elementwise_fma(f2{gw(k), gw(k)}, xy, acc[s].xy);
                acc[s].zw = __builtin_elementwise_fma(f2{gw(k), gw(k)}, v1, acc[s].zw);
            }
        }
        if (emit) {
            const int s = (p + 1) % 11; // slot of output row o = r - 10, now complete
            const int buf = (r - 10) & 1;
            *(f4 *)&vb[buf][t] = acc[s];
            __syncthreads();
            if (out_c) {
                f2 o0 = {0.f, 0.f}, o1 = {0.f, 0.f};
                if (HPF == 0) {
#pragma unroll
                    for (int k = 0; k < 11; k++) {
#if SSIM_PROBE == 1   // no LDS reads: the lane's own vertical sums stand in for its neighbours' (wrong numbers)
                        const float4 q = make_float4(acc[s].x, acc[s].y, acc[s].z, acc[s].w);
#elif SSIM_PROBE == 2 // 6 LDS reads instead of 11 (what a column-pair kernel would issue), same FMAs (wrong numbers)
                        const float4 q = vb[buf][t + (k >> 1)];
#else
                        const float4 q = vb[buf][t + k];
#endif
                        o0 = __builtin_elementwise_fma(f2{gw(k), gw(k)}, f2{q.x, q.y}, o0);
                        o1 = __builtin_elementwise_fma(f2{gw(k), gw(k)}, f2{q.z, q.w}, o1);
                    }
                } else {
                    // (lab build) issue the LDS reads in groups of HPF before any arithmetic on them
                    float4 q[11];
#pragma unroll
                    for (int k0 = 0; k0 < 11; k0 += HPF) {
#pragma unroll
                        for (int k = k0; k < k0 + HPF && k < 11; k++) q[k] = vb[buf][t + k];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int k = k0; k < k0 + HPF && k < 11; k++) {
                            o0 = __builtin_elementwise_fma(f2{gw(k), gw(k)}, f2{q[k].x, q[k].y}, o0);
                            o1 = __builtin_elementwise_fma(f2{gw(k), gw(k)}, f2{q[k].z, q[k].w}, o1);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                ssim_acc += ssim_centered(o0, o1);
            }
        }
    };

    // Rows are consumed in groups of 11 (one turn of the accumulator ring).  The host picks QS as a multiple of 11, so the
    // QS rows a strip OWNS (for the squared error) are exactly its full groups; what remains - the 10 halo rows shared
    // with the next strip, or the last strip's leftover - goes through one guarded copy of the group below the loop.
    // The main loop carries no per-row guards (round 3's single 22-row body tested r < nrows in every row, and the
    // compiler paid for the merged control flow with four register moves per row).
    const int nfull = nrows / 11;
    int flush = 0;
    for (int g = 0; g < nfull; g++) {
        const int r0 = g * 11;
        if (++flush == 22) { sse_acc += (uint32_t)(sse_sq - 2.f * sse_xy); sse_sq = sse_xy = 0.f; flush = 0; } // 242 rows
#pragma unroll
        for (int pp = 0; pp < 11; pp++) row(pp, r0 + pp, pp == 10 || g > 0);
    }
    sse_acc += (uint32_t)(sse_sq - 2.f * sse_xy);
    sse_sq = sse_xy = 0.f;
    if (!last_sb) own_f = 0.f; // the halo rows belong to the next strip
    {
        const int r0 = nfull * 11;
#pragma unroll
        for (int pp = 0; pp < 10; pp++)
            if (r0 + pp < nrows) row(pp, r0 + pp, r0 + pp >= 10);
    }
    sse_acc += (uint32_t)(sse_sq - 2.f * sse_xy);
    const double bs = block_sum((double)ssim_acc, red);
    const unsigned long long be = block_sum_u64((unsigned long long)sse_acc, redu);
    if (t == 0) {
        const int pidx = g.plane_index[ch];
        partials[(int64_t)pidx * partial_plane_stride + (int64_t)f * bpp + tile] = bs;
        if (be) atomicAdd((unsigned long long *)&res[(int64_t)f * n_planes + pidx].sse, be);
    }
}

#endif // VQA_AB_VARIANTS

// ---------------------------------------------------------------------------
// k_ssim_gauss_p2<QT, R>: the same arithmetic with a FIFTH of the LDS read traffic.  Round 4 found the kernel
// POWER-bound: the chip sits at its ~1.27 kW cap and sets the clock to fit (1.9 GHz under round 3's kernel); removing
// 7 % of the instructions changed neither time nor energy, while the 11 ds_read_b128 per output pixel were worth 19 %
// of a launch's energy (LAB_NOTES.md L4, profiles/round4_ssim_clock.json).  So the horizontal pass is shared:
// R rows go through LDS per round trip and a GROUP OF R LANES shares R adjacent columns - lane R j + i filters the
// columns R j .. R j + R - 1 of the step's i-th row.  Each lane reads the 10 + R columns R j .. R j + R + 9 once
// ((10 + R) / R reads per output pixel: 6 for R = 2, 3.5 for R = 4, 2.25 for R = 8, against 11) and runs the same
// 22 R packed FMAs and R SSIM formulas a lane of the one-row kernel runs for its R rows.  The vertical pass is
// unchanged (one column per thread, rolling accumulators) except that the ring has 12 slots (11 live + 1 spare) so
// that R divides it and 3 or 6 steps make a loop iteration.  LDS rows are QT + 1 float4 apart: the lanes of a group
// sit in different bank groups and every ds_read_b128 stays conflict-free.
// Measured per 256 x 1080p x 3-plane launch (same box, interleaved): one row 4.51-4.66 ms, R = 2 4.33-4.39,
// R = 4 (double-buffered) 4.08-4.25, R = 8 (single buffer, two barriers per step: 33 KB LDS, 4 waves/SIMD) 4.14
// against 4.23-4.25 for R = 4 on that box; R = 8 double-buffered (66 KB, 2 waves/SIMD) 4.54 at the full 2.38 GHz:
// no longer power-bound but latency-bound.
// ---------------------------------------------------------------------------
template <int QT, int R>
__global__ __launch_bounds__(QT) void k_ssim_gauss_p2(const uint8_t *__restrict__ ref, const uint8_t *__restrict__ dist,
                                                      int64_t ref_fs, int64_t dist_fs, plane_group g, int64_t row_stride,
                                                      int step, int w, int h, int ncb, int nstrips, int QS,
                                                      double *__restrict__ partials, int64_t partial_plane_stride,
                                                      int n_planes, vqa_plane_metrics *__restrict__ res)
{
    static_assert(R == 2 || R == 4 || R == 8, "rows per barrier = lanes per group = adjacent columns per lane");
    constexpr int QOUT = QT - 10;
    constexpr int NS = 12;                  // accumulator ring: 11 live output rows + 1 spare slot, so that R divides the ring
    constexpr int ST = R == 8 ? 3 : 6;      // steps per loop iteration
    constexpr int GR = ST * R;              // rows per loop iteration: a whole number of ring turns
    constexpr int NB = R == 8 ? 1 : 2;      // LDS row buffers: R = 8 (shipped) has ONE and two barriers per step (33 KB, 4 waves
                                            // per SIMD; double-buffered it needs 66 KB, drops to 2 waves and turns latency-bound);
                                            // the R = 2 / 4 measurement builds double-buffer with one barrier per step
    __shared__ float4 vb[NB][R][QT + 1];    // [step parity][row of the step][column]
    __shared__ double red[4];
    __shared__ unsigned long long redu[4];
    __shared__ unsigned long long redf[4];
    const int f = blockIdx.y;
    const int t = threadIdx.x;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int ch = seq % g.count, tile = (seq / g.count) * 8 + xcd;
    const int bpp = ncb * nstrips;
    if (tile >= bpp) return;
    const int cb = tile % ncb, sb = tile / ncb;
    const int xs = cb * QOUT, ys = sb * QS;
    const int xin = xs + t;
    const int ow = w - 10;
    const int nrows = min(QS + 10, h - ys);
    const bool last_cb = cb == ncb - 1, last_sb = sb == nstrips - 1;
    const bool in_c = xin < w;
    const bool own_c = in_c && (last_cb || t < QOUT);
    // horizontal role: the R adjacent columns hc .. hc + R - 1 of row (t % R) of the step
    const int hc = t & ~(R - 1), hrow = t & (R - 1);
    bool out_j[R];
#pragma unroll
    for (int j = 0; j < R; j++) out_j[j] = hc + j < QOUT && xs + hc + j < ow;
    const int64_t base = g.offset[ch] + (int64_t)ys * row_stride;
    const uint8_t *rbase = ref + (int64_t)f * ref_fs + base;
    const uint8_t *dbase = dist + (int64_t)f * dist_fs + base;
    const uint32_t coff = (uint32_t)((in_c ? xin : 0) * step);

    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 acc[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) acc[s] = f4{0.f, 0.f, 0.f, 0.f};
    // The SSIM map's sum.  Every value of the map is the same float whichever strip computes it (fixed tap order down and
    // across); what used to follow the strip geometry - and so the batch size: ssim_strips() - was only the ORDER the values
    // were added in (a float chain per lane, doubles per workgroup).  A lane's step yields the float sum of <= R adjacent
    // values of one row, in column order (|sum| <= R); that is rounded ONCE to a multiple of 2^-27 and added as an integer:
    // integer addition is associative, so lanes, workgroups and strips can be cut anywhere and k_ssim_finalize gets the
    // same total, bit for bit.  The rounding is <= 2^-28 per R values: <= 5e-10 on the mean (the bar is 1e-4), and the
    // integer sum carries none of the float chain's own ~1e-7 drift.
    long long ssim_fx = 0;
    float ssim_acc = 0.f;
    uint32_t sse_acc = 0;
    float sse_sq = 0.f, sse_xy = 0.f;
    float own_f = own_c ? 1.f : 0.f;
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void *)rbase, (short)0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t dres = __builtin_amdgcn_make_buffer_rsrc((void *)dbase, (short)0, -1, 0x00020000);
    auto ld = [&](const __amdgpu_buffer_rsrc_t &rs, int rr) -> uint32_t {
        return (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rs, coff, (int)((int64_t)rr * row_stride), 0);
    };
    uint32_t nr = ld(rres, 0), nd = ld(dres, 0);

    // LDS row buffers.  NB = 1 (shipped): a barrier before the step's stores (the previous step's readers are done) and one
    // before its reads.  NB = 2: step st stores into buffer st & 1 - the one the step BEFORE LAST read from - and one
    // barrier per step orders that, PROVIDED the parity carries across loop iterations: an iteration must have an even
    // number of steps, or flip gpar.  (Round 4's first kernel had 11 steps per iteration and no flip: its last step and
    // the next iteration's first shared a buffer with no barrier between read and overwrite - a race that every parity
    // test survived and that the bench's serial-vs-timed comparison caught; test_results_are_bit_identical_run_to_run
    // is its permanent test.)
    int gpar = 0;
    // vertical pass of one input row; p = r mod NS (static after unrolling).  Input row r is tap k of output row r - k,
    // kept in ring slot (r - k) mod NS.  emit: store the completed output row r - 10 into vb[sp][slot][t]
    auto vrow = [&](const int p, const int r, const bool emit, const int sp, const int slot) {
        const uint32_t cr = nr, cd = nd;
        {
            const int rr = min(r + 1, nrows - 1);
            nr = ld(rres, rr);
            nd = ld(dres, rr);
        }
        const f2 xy = f2{__uint_as_float(0x4B000000u | cr), __uint_as_float(0x4B000000u | cd)} - f2{8388736.f, 8388736.f};
        f2 v1 = xy.xx * xy;
        v1.x = fmaf(xy.y, xy.y, v1.x);
        sse_sq = fmaf(own_f, v1.x, sse_sq);
        sse_xy = fmaf(own_f, v1.y, sse_xy);
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const int s = (p - k + NS) % NS;
            if (k == 0) {
                acc[s].xy = gw(0) * xy;
                acc[s].zw = gw(0) * v1;
            } else {
                acc[s].xy = __builtin_elementwise_fma(f2{gw(k), gw(k)}, xy, acc[s].xy);
                acc[s].zw = __builtin_elementwise_fma(f2{gw(k), gw(k)}, v1, acc[s].zw);
            }
        }
        if (emit) *(f4 *)&vb[sp][slot][t] = acc[(p + NS - 10) % NS];
    };
    // horizontal pass + SSIM of the step's output rows lo .. hi - 1 (wave-uniform bounds): this lane's row is hrow
    auto hpass = [&](const int sp, const int lo, const int hi) {
        __syncthreads();
        if (out_j[0] && hrow >= lo && hrow < hi) {
            const float4 *src = &vb[sp][hrow][hc];
            f2 o0[R], o1[R];
#pragma unroll
            for (int j = 0; j < R; j++) o0[j] = o1[j] = f2{0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 10 + R; i++) {
                const float4 q = src[i];
#pragma unroll
                for (int j = 0; j < R; j++)
                    if (i - j >= 0 && i - j < 11) {
                        o0[j] = __builtin_elementwise_fma(f2{gw(i - j), gw(i - j)}, f2{q.x, q.y}, o0[j]);
                        o1[j] = __builtin_elementwise_fma(f2{gw(i - j), gw(i - j)}, f2{q.z, q.w}, o1[j]);
                    }
            }
            float sum = ssim_centered(o0[0], o1[0]);
#pragma unroll
            for (int j = 1; j < R; j++) {
                const float v = ssim_centered(o0[j], o1[j]);
                sum += out_j[j] ? v : 0.f;
            }
            if (SSIM_FIXED_SUM) ssim_fx += (long long)__float2int_rn(sum * 134217728.f); // 2^27: |sum| <= 8 fits 31 bits
            else ssim_acc += sum;
        }
    };

    // GR rows per iteration; QS is a multiple of GR, so a strip's owned rows are its full iterations and the rest (10
    // halo rows, or the last strip's leftover) goes through one guarded copy below.
    const int nfull = nrows / GR;
    int flush = 0;
    for (int gI = 0; gI < nfull; gI++) {
        const int r0 = gI * GR;
        if (++flush == 240 / GR) { sse_acc += (uint32_t)(sse_sq - 2.f * sse_xy); sse_sq = sse_xy = 0.f; flush = 0; } // <= 240 rows
#pragma unroll
        for (int st = 0; st < ST; st++) {
            const int sp = NB == 1 ? 0 : ((ST & 1) ? ((st & 1) ^ gpar) : (st & 1));
            if (NB == 1) __syncthreads(); // single buffer: the previous step's readers are done
#pragma unroll
            for (int j = 0; j < R; j++) vrow((R * st + j) % NS, r0 + R * st + j, gI > 0 || R * st + j >= 10, sp, j);
            const int lo0 = 10 - R * st; // rows 0..9 of a strip complete no output row
            const int lo = gI > 0 ? 0 : (lo0 > 0 ? lo0 : 0);
            if (lo < R) hpass(sp, lo, R);
        }
        if (ST & 1) gpar ^= 1;
    }
    sse_acc += (uint32_t)(sse_sq - 2.f * sse_xy);
    sse_sq = sse_xy = 0.f;
    if (!last_sb) own_f = 0.f; // the halo rows belong to the next strip
    {
        const int r0 = nfull * GR;
#pragma unroll
        for (int st = 0; st < ST; st++) {
            const int ra = r0 + R * st;
            if (ra < nrows) {
                const int sp = NB == 1 ? 0 : ((ST & 1) ? ((st & 1) ^ gpar) : (st & 1));
                if (NB == 1) __syncthreads();
                const int hi = min(R, nrows - ra);
#pragma unroll
                for (int j = 0; j < R; j++)
                    if (j < hi) vrow((R * st + j) % NS, ra + j, ra + j >= 10, sp, j);
                const int lo = ra >= 10 ? 0 : 10 - ra;
                if (lo < hi) hpass(sp, lo, hi);
            }
        }
    }
    sse_acc += (uint32_t)(sse_sq - 2.f * sse_xy);
    static_assert(R <= 8, "2^27 x R must fit an int");
    const double bs = SSIM_FIXED_SUM ? 0.0 : block_sum((double)ssim_acc, red);
    const unsigned long long bf = SSIM_FIXED_SUM ? block_sum_u64((unsigned long long)ssim_fx, redf) : 0ull; // (two's complement: signed sums wrap right)
    const unsigned long long be = block_sum_u64((unsigned long long)sse_acc, redu);
    if (t == 0) {
        const int pidx = g.plane_index[ch];
        double *slot = &partials[(int64_t)pidx * partial_plane_stride + (int64_t)f * bpp + tile];
        if (SSIM_FIXED_SUM) *reinterpret_cast<unsigned long long *>(slot) = bf;
        else *slot = bs;
        if (be) atomicAdd((unsigned long long *)&res[(int64_t)f * n_planes + pidx].sse, be);
    }
}

// FIXED: the partials are 2^-27 fixed-point integer sums (k_ssim_gauss_p2), else doubles (vf_ssim kernels, lab kernels)
template <bool FIXED>
__global__ void k_ssim_finalize(const double *__restrict__ partials, int bpp, int n, double inv_count, int plane_index,
                                int n_planes, vqa_plane_metrics *__restrict__ res)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    if (FIXED) {
        long long s = 0;
        const long long *p = reinterpret_cast<const long long *>(partials);
        for (int i = 0; i < bpp; i++) s += p[(int64_t)f * bpp + i];
        res[(int64_t)f * n_planes + plane_index].ssim = (double)s * (1.0 / 134217728.0) * inv_count; // (|s| < 2^53: exact)
        return;
    }
    double s = 0;
    for (int i = 0; i < bpp; i++) s += partials[(int64_t)f * bpp + i];
    res[(int64_t)f * n_planes + plane_index].ssim = s * inv_count;
}

// Shipped: k_ssim_gauss_p2<256>.  Lab build only (-DVQA_AB_VARIANTS, VQA_SSIM_VARIANT): the one-row-per-barrier kernel
// k_ssim_gauss<threads, -, LDS reads in flight>: 1 = {256,11}, 2 = {256,6}, 3 = {256,4}, 4 = {128,compiler's},
// 5 = {256,compiler's} (round 3's shipped configuration)
static int ssim_variant()
{
    static const int v = [] { const int e = ab_knob("VQA_SSIM_VARIANT", 0); return (e < 0 || e > 5) ? 0 : e; }();
    return v;
}
static int ssim_qout() { return (ssim_variant() == 4 ? 128 : 256) - 10; }

int ssim_gauss_blocks(int h, int w)
{
    if (h < 11 || w < 11) return 0;
    const int qout = ssim_qout();
    const int ncb = (w - 10 + qout - 1) / qout;
    return ncb * ssim_max_strips(h); // upper bound: sizes the per-plane partials segment
}

// planes[idx[0..count)] share width, height, row_stride and pixel_step
void launch_quality_gauss(hipStream_t st, const uint8_t *ref, const uint8_t *dist, int n, int64_t ref_frame_stride,
                          int64_t dist_frame_stride, const vqa_plane_desc *planes, const int *idx, int count,
                          int n_planes, double *partials, int64_t partial_plane_stride, vqa_plane_metrics *res)
{
    if (n <= 0 || count <= 0) return;
    const vqa_plane_desc &pd = planes[idx[0]];
    const int w = pd.width, h = pd.height;
    const int qout = ssim_qout();
    const int ncb = (w - 10 + qout - 1) / qout;
    const int ns0 = ssim_strips(h, (long long)n * count * ncb), QS = ssim_strip_rows(h, ns0);
    const int ns = (h - 10 + QS - 1) / QS; // strips actually needed at this strip height
    const int bpp = ncb * ns;
    plane_group g;
    g.count = count;
    for (int i = 0; i < 4; i++) { g.offset[i] = planes[idx[i < count ? i : 0]].offset; g.plane_index[i] = idx[i < count ? i : 0]; }
#ifdef VQA_AB_VARIANTS
#define LAUNCH_SSIM(NT, PF, HPF)                                                                                      \
    hipLaunchKernelGGL((k_ssim_gauss<NT, PF, HPF>), dim3((bpp + 7) / 8 * 8 * count, n), dim3(NT), 0, st, ref, dist, ref_frame_stride,\
                       dist_frame_stride, g, pd.row_stride, pd.pixel_step, w, h, ncb, ns, QS, partials,               \
                       partial_plane_stride, n_planes, res)
#endif
    switch (ssim_variant()) {
#ifdef VQA_AB_VARIANTS
    case 1: LAUNCH_SSIM(256, 2, 11); break;
    case 2: LAUNCH_SSIM(256, 2, 6); break;
    case 3: LAUNCH_SSIM(256, 2, 4); break;
    case 4: LAUNCH_SSIM(128, 2, 0); break;
    case 5: LAUNCH_SSIM(256, 2, 0); break; // round 3's shipped kernel (one row per barrier), with round 4's loop structure
#endif
    default: // the shipped kernel: SSIM_ROWS = 8 rows per LDS round trip, groups of 8 lanes share the horizontal pass
        hipLaunchKernelGGL((k_ssim_gauss_p2<256, SSIM_ROWS>), dim3((bpp + 7) / 8 * 8 * count, n), dim3(256), 0, st, ref, dist, ref_frame_stride,
                           dist_frame_stride, g, pd.row_stride, pd.pixel_step, w, h, ncb, ns, QS, partials,
                           partial_plane_stride, n_planes, res);
        break;
    }
#undef LAUNCH_SSIM
    const bool fixed = SSIM_FIXED_SUM && ssim_variant() == 0; // (the lab build's one-row kernels leave doubles)
    for (int i = 0; i < count; i++) {
        if (fixed)
            hipLaunchKernelGGL(k_ssim_finalize<true>, dim3((n + 63) / 64), dim3(64), 0, st,
                               partials + (int64_t)idx[i] * partial_plane_stride, bpp, n,
                               1.0 / ((double)(w - 10) * (double)(h - 10)), idx[i], n_planes, res);
        else
            hipLaunchKernelGGL(k_ssim_finalize<false>, dim3((n + 63) / 64), dim3(64), 0, st,
                               partials + (int64_t)idx[i] * partial_plane_stride, bpp, n,
                               1.0 / ((double)(w - 10) * (double)(h - 10)), idx[i], n_planes, res);
    }
}

// ---------------------------------------------------------------------------
// FFmpeg vf_ssim 8-bit: 4x4 block sums (s1, s2, ss, s12); a sample pools a 2x2
// group of blocks.  Thread <-> one column of 4x4 blocks, marching down block
// rows; the pair sum of the previous block row stays in registers, the right
// neighbour's pair sum comes from the next lane.  A wave of 64 block columns
// yields 63 samples, so waves overlap by one block column.
// grid = (ncw * nstrips, n_frames); block = 64 (one wave)
// ---------------------------------------------------------------------------
constexpr int FS_ROWS = 32; // sample rows per strip

__global__ __launch_bounds__(64) void k_ssim_ffmpeg(const uint8_t *__restrict__ ref, const uint8_t *__restrict__ dist,
                                                    int64_t ref_fs, int64_t dist_fs, int64_t offset,
                                                    int64_t row_stride, int step, int w, int h, int ncw, int nstrips,
                                                    double *__restrict__ partials, int plane_index, int n_planes,
                                                    vqa_plane_metrics *__restrict__ res)
{
    const int f = blockIdx.y;
    const int lane = threadIdx.x;
    const int cw = blockIdx.x % ncw, sb = blockIdx.x / ncw;
    const int bw = w >> 2, bh = h >> 2;
    const int bx = cw * 63 + lane; // block column
    const int by0 = sb * FS_ROWS;  // first block row of this strip (sample rows by0 .. by0+FS_ROWS-1)
    const int by_end = min(by0 + FS_ROWS + 1, bh);
    const bool last_cw = cw == ncw - 1, last_sb = sb == nstrips - 1;
    const uint8_t *rb = ref + (int64_t)f * ref_fs + offset;
    const uint8_t *db = dist + (int64_t)f * dist_fs + offset;
    int p1 = 0, p2 = 0, pss = 0, p12 = 0;
    double ssim_acc = 0;
    unsigned long long sse_acc = 0;
    for (int by = by0; by < by_end; by++) {
        int s1 = 0, s2 = 0, ss = 0, s12 = 0;
        uint32_t se = 0;
        if (bx < bw) {
#pragma unroll
            for (int y = 0; y < 4; y++) {
                const uint8_t *rr = rb + (int64_t)(by * 4 + y) * row_stride + (int64_t)(bx * 4) * step;
                const uint8_t *dr = db + (int64_t)(by * 4 + y) * row_stride + (int64_t)(bx * 4) * step;
#pragma unroll
                for (int x = 0; x < 4; x++) {
                    const int a = rr[(int64_t)x * step], b = dr[(int64_t)x * step];
                    s1 += a; s2 += b; ss += a * a + b * b; s12 += a * b;
                    se += (uint32_t)((a - b) * (a - b));
                }
            }
        }
        // SSE ownership: block column owned unless it is the overlap column of a non-last wave;
        // block row owned unless it is the overlap row of a non-last strip.
        const bool own = (bx < bw) && (last_cw || lane < 63) && (last_sb || by < by0 + FS_ROWS);
        if (own) sse_acc += se;
        if (by > by0) {
            // vertical pair (by-1, by), then horizontal pair with lane+1
            const int v1 = p1 + s1, v2 = p2 + s2, vss = pss + ss, v12 = p12 + s12;
            const int n1 = __shfl_down(v1, 1, 64), n2 = __shfl_down(v2, 1, 64);
            const int nss = __shfl_down(vss, 1, 64), n12 = __shfl_down(v12, 1, 64);
            if (lane < 63 && bx + 1 < bw)
                ssim_acc += (double)ssim_ffmpeg_end1(v1 + n1, v2 + n2, vss + nss, v12 + n12);
        }
        p1 = s1; p2 = s2; pss = ss; p12 = s12;
    }
    ssim_acc = wave_sum(ssim_acc);
    sse_acc = wave_sum(sse_acc);
    if (lane == 0) {
        partials[(int64_t)f * gridDim.x + blockIdx.x] = ssim_acc;
        if (sse_acc) atomicAdd((unsigned long long *)&res[(int64_t)f * n_planes + plane_index].sse, sse_acc);
    }
}

// Fast path of the same filter: dword loads and v_dot4_u32_u8.  One load of 4 pixels feeds five dot
// products (s1 = <a,1>, s2 = <b,1>, ss = <a,a> + <b,b>, s12 = <a,b>), the squared error falls out as
// ss - 2 s12, and for packed BGR24 the three channels are de-interleaved in registers (2 v_perm_b32
// each) so ONE pass over the bytes serves all three planes: 6P bytes per frame pair, HBM-bound.
// NCH = 1: planar plane, pixel_step 1.  NCH = 3: planes at byte offsets +0,+1,+2 with pixel_step 3.
template <int NCH>
__global__ __launch_bounds__(64) void k_ssim_ffmpeg_fast(const uint8_t *__restrict__ ref,
                                                         const uint8_t *__restrict__ dist, int64_t ref_fs,
                                                         int64_t dist_fs, int64_t offset, int64_t row_stride, int w,
                                                         int h, int ncw, int nstrips, double *__restrict__ partials,
                                                         int64_t partial_plane_stride, int plane_index0, int n_planes,
                                                         vqa_plane_metrics *__restrict__ res)
{
    const int f = blockIdx.y;
    const int lane = threadIdx.x;
    const int cw = blockIdx.x % ncw, sb = blockIdx.x / ncw;
    const int bw = w >> 2, bh = h >> 2;
    const int bx = cw * 63 + lane;
    const int by0 = sb * FS_ROWS;
    const int by_end = min(by0 + FS_ROWS + 1, bh);
    const bool last_cw = cw == ncw - 1, last_sb = sb == nstrips - 1;
    const bool have = bx < bw;
    const uint8_t *rb = ref + (int64_t)f * ref_fs + offset + (int64_t)(have ? bx : 0) * 4 * NCH;
    const uint8_t *db = dist + (int64_t)f * dist_fs + offset + (int64_t)(have ? bx : 0) * 4 * NCH;
    int p1[NCH], p2[NCH], pss[NCH], p12[NCH];
    double ssim_acc[NCH];
    unsigned long long sse_acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) { p1[c] = p2[c] = pss[c] = p12[c] = 0; ssim_acc[c] = 0; sse_acc[c] = 0; }
    for (int by = by0; by < by_end; by++) {
        uint32_t s1[NCH], s2[NCH], ss[NCH], s12[NCH];
#pragma unroll
        for (int c = 0; c < NCH; c++) s1[c] = s2[c] = ss[c] = s12[c] = 0;
        struct __attribute__((aligned(4))) wvec { uint32_t v[NCH]; }; // NCH = 3 -> one global_load_dwordx3
        uint32_t ra[4][NCH], da[4][NCH];
#pragma unroll
        for (int y = 0; y < 4; y++) {
            wvec rv, dv;
#pragma unroll
            for (int c = 0; c < NCH; c++) rv.v[c] = dv.v[c] = 0u;
            if (have) {
                rv = *(const wvec *)(rb + (int64_t)(by * 4 + y) * row_stride);
                dv = *(const wvec *)(db + (int64_t)(by * 4 + y) * row_stride);
            }
#pragma unroll
            for (int c = 0; c < NCH; c++) { ra[y][c] = rv.v[c]; da[y][c] = dv.v[c]; }
        }
#pragma unroll
        for (int y = 0; y < 4; y++) {
            uint32_t a[NCH], b[NCH];
            if (NCH == 3) {
                // 12 bytes B0 G0 R0 B1 .. R3 -> (B0 B1 B2 B3), (G0 G1 G2 G3), (R0 R1 R2 R3)
                const uint32_t *wa = ra[y], *wb = da[y];
                a[0] = __builtin_amdgcn_perm(wa[2], __builtin_amdgcn_perm(wa[1], wa[0], 0x0c060300u), 0x05020100u);
                a[1 % NCH] = __builtin_amdgcn_perm(wa[2], __builtin_amdgcn_perm(wa[1], wa[0], 0x0c070401u), 0x06020100u);
                a[2 % NCH] = __builtin_amdgcn_perm(wa[2], __builtin_amdgcn_perm(wa[1], wa[0], 0x0c0c0502u), 0x07040100u);
                b[0] = __builtin_amdgcn_perm(wb[2], __builtin_amdgcn_perm(wb[1], wb[0], 0x0c060300u), 0x05020100u);
                b[1 % NCH] = __builtin_amdgcn_perm(wb[2], __builtin_amdgcn_perm(wb[1], wb[0], 0x0c070401u), 0x06020100u);
                b[2 % NCH] = __builtin_amdgcn_perm(wb[2], __builtin_amdgcn_perm(wb[1], wb[0], 0x0c0c0502u), 0x07040100u);
            } else {
                a[0] = ra[y][0];
                b[0] = da[y][0];
            }
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                s1[c] = __builtin_amdgcn_udot4(a[c], 0x01010101u, s1[c], false);
                s2[c] = __builtin_amdgcn_udot4(b[c], 0x01010101u, s2[c], false);
                ss[c] = __builtin_amdgcn_udot4(a[c], a[c], ss[c], false);
                ss[c] = __builtin_amdgcn_udot4(b[c], b[c], ss[c], false);
                s12[c] = __builtin_amdgcn_udot4(a[c], b[c], s12[c], false);
            }
        }
        const bool own = have && (last_cw || lane < 63) && (last_sb || by < by0 + FS_ROWS);
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            if (own) sse_acc[c] += (unsigned long long)(ss[c] - 2u * s12[c]);
            if (by > by0) {
                const int v1 = p1[c] + (int)s1[c], v2 = p2[c] + (int)s2[c], vss = pss[c] + (int)ss[c],
                          v12 = p12[c] + (int)s12[c];
                const int n1 = __shfl_down(v1, 1, 64), n2 = __shfl_down(v2, 1, 64);
                const int nss = __shfl_down(vss, 1, 64), n12 = __shfl_down(v12, 1, 64);
                if (lane < 63 && bx + 1 < bw)
                    ssim_acc[c] += (double)ssim_ffmpeg_end1(v1 + n1, v2 + n2, vss + nss, v12 + n12);
            }
            p1[c] = (int)s1[c]; p2[c] = (int)s2[c]; pss[c] = (int)ss[c]; p12[c] = (int)s12[c];
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const double sa = wave_sum(ssim_acc[c]);
        const unsigned long long se = wave_sum(sse_acc[c]);
        if (lane == 0) {
            partials[(int64_t)(plane_index0 + c) * partial_plane_stride + (int64_t)f * gridDim.x + blockIdx.x] = sa;
            if (se) atomicAdd((unsigned long long *)&res[(int64_t)f * n_planes + plane_index0 + c].sse, se);
        }
    }
}

// SSE of the rows/columns that no 4x4 block covers (ragged right/bottom edges);
// one thread per frame is plenty: at most 3 rows + 3 columns.
__global__ void k_sse_ragged(const uint8_t *__restrict__ ref, const uint8_t *__restrict__ dist, int64_t ref_fs,
                             int64_t dist_fs, int64_t offset, int64_t row_stride, int step, int w, int h, int n,
                             int plane_index, int n_planes, vqa_plane_metrics *__restrict__ res)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    const uint8_t *rb = ref + (int64_t)f * ref_fs + offset;
    const uint8_t *db = dist + (int64_t)f * dist_fs + offset;
    const int bw4 = (w >> 2) << 2, bh4 = (h >> 2) << 2;
    unsigned long long s = 0;
    for (int y = 0; y < h; y++) {
        const int xb = y < bh4 ? bw4 : 0;
        for (int x = xb; x < w; x++) {
            const int a = rb[(int64_t)y * row_stride + (int64_t)x * step], b = db[(int64_t)y * row_stride + (int64_t)x * step];
            s += (unsigned long long)((a - b) * (a - b));
        }
    }
    if (s) atomicAdd((unsigned long long *)&res[(int64_t)f * n_planes + plane_index].sse, s);
}

int ssim_ffmpeg_blocks(int h, int w)
{
    const int bw = w >> 2, bh = h >> 2;
    if (bw < 2 || bh < 2) return 0;
    const int ncw = (bw - 1 + 62) / 63, ns = (bh - 1 + FS_ROWS - 1) / FS_ROWS;
    return ncw * ns;
}

static bool aligned4(const void *p, int64_t a, int64_t b, int64_t c, int64_t d)
{
    return (((uintptr_t)p | (uint64_t)a | (uint64_t)b | (uint64_t)c | (uint64_t)d) & 3u) == 0;
}

// planes[idx[0..count)] share geometry.  count == 3 with pixel_step 3 and byte offsets o, o+1, o+2 (packed
// BGR24) takes the fused three-channel path; pixel_step 1 takes the planar fast path; anything else the
// generic byte kernel.  partials: [n_planes][partial_plane_stride].
void launch_quality_ffmpeg(hipStream_t st, const uint8_t *ref, const uint8_t *dist, int n, int64_t ref_frame_stride,
                           int64_t dist_frame_stride, const vqa_plane_desc *planes, const int *idx, int count,
                           int n_planes, double *partials, int64_t partial_plane_stride, vqa_plane_metrics *res)
{
    if (n <= 0 || count <= 0) return;
    const vqa_plane_desc &pd = planes[idx[0]];
    const int w = pd.width, h = pd.height;
    const int bw = w >> 2, bh = h >> 2;
    const int ncw = (bw - 1 + 62) / 63, ns = (bh - 1 + FS_ROWS - 1) / FS_ROWS;
    const int bpp = ncw * ns;
    const bool al = aligned4(ref, (int64_t)(uintptr_t)dist, ref_frame_stride, dist_frame_stride, pd.row_stride);
    const bool fused3 = al && count == 3 && pd.pixel_step == 3 && (pd.offset & 3) == 0 &&
                        planes[idx[1]].offset == pd.offset + 1 && planes[idx[2]].offset == pd.offset + 2 &&
                        idx[1] == idx[0] + 1 && idx[2] == idx[0] + 2;
    if (fused3) {
        hipLaunchKernelGGL(k_ssim_ffmpeg_fast<3>, dim3(bpp, n), dim3(64), 0, st, ref, dist, ref_frame_stride,
                           dist_frame_stride, pd.offset, pd.row_stride, w, h, ncw, ns, partials, partial_plane_stride,
                           idx[0], n_planes, res);
    } else {
        for (int i = 0; i < count; i++) {
            const vqa_plane_desc &q = planes[idx[i]];
            if (al && q.pixel_step == 1 && (q.offset & 3) == 0)
                hipLaunchKernelGGL(k_ssim_ffmpeg_fast<1>, dim3(bpp, n), dim3(64), 0, st, ref, dist, ref_frame_stride,
                                   dist_frame_stride, q.offset, q.row_stride, w, h, ncw, ns, partials,
                                   partial_plane_stride, idx[i], n_planes, res);
            else
                hipLaunchKernelGGL(k_ssim_ffmpeg, dim3(bpp, n), dim3(64), 0, st, ref, dist, ref_frame_stride,
                                   dist_frame_stride, q.offset, q.row_stride, q.pixel_step, w, h, ncw, ns,
                                   partials + (int64_t)idx[i] * partial_plane_stride, idx[i], n_planes, res);
        }
    }
    for (int i = 0; i < count; i++) {
        const vqa_plane_desc &q = planes[idx[i]];
        if ((w & 3) || (h & 3))
            hipLaunchKernelGGL(k_sse_ragged, dim3((n + 63) / 64), dim3(64), 0, st, ref, dist, ref_frame_stride,
                               dist_frame_stride, q.offset, q.row_stride, q.pixel_step, w, h, n, idx[i], n_planes, res);
        hipLaunchKernelGGL(k_ssim_finalize<false>, dim3((n + 63) / 64), dim3(64), 0, st,
                           partials + (int64_t)idx[i] * partial_plane_stride, bpp, n,
                           1.0 / ((double)(bh - 1) * (double)(bw - 1)), idx[i], n_planes, res);
    }
}

} // namespace vqa
