// k_canny.hip — cv2.Canny(gray, low, high) edge-pixel count for gfx950.
//
// Reference function replaced: process_edge_frame, complexity_metrics.py:477-504
//   cv2.Canny(gray, 100, 200) (aperture 3, L1 gradient) -> np.sum(edges > 0)
//
// Stage 1 — gradient, non-maximum suppression, double threshold.  Output: two BIT-PLANES (edge,
//   weak candidate), one u64 per 64 pixels, 2 x P/8 bytes instead of a P-byte map.
//   k_canny_nms3 (default): lane = column; a vector compare of 64 columns IS a bit-plane word and every NMS decision is
//     mask logic on the scalar unit; one unaligned dword load per pixel; v_sad_u32 magnitudes; DPP ring for the
//     horizontal neighbours; lane r captures row r, so a strip leaves as four 512-byte tile stores (details below).
//   k_canny_nms  (frames narrower than 4 pixels, which the dword window cannot serve): 64x32 tile per workgroup through
//     LDS, words built with __ballot (round 1's kernel).
//   k_canny_nms2 (lab build only, VQA_NMS_VARIANT=2: round 2's kernel): 4 pixels per lane, packed 16-bit Sobel.
//   All: Sobel 3x3 on replicated borders, L1 magnitude = 0 outside the image (OpenCV's zero-bordered
//   buffer), TG22 fixed-point sector test.
// Stage 2 (k_canny_hyst_*): 8-connected hysteresis as an iterate-to-fixpoint on
//   64x64-pixel tiles held ENTIRELY IN REGISTERS: lane r owns row r as a u64.
//   One step = vertical neighbours by whole-wave DPP shifts, 3x3 dilation, then a flood along the row done by the
//   integer adder (carry propagation through runs; bit-reversed for the other direction), so a horizontal chain of
//   any length is absorbed in ONE step.  Promotion is monotone (weak -> edge only), so the fixpoint is unique and
//   independent of scheduling: the count is bit-exact with OpenCV's sequential stack walk.
//   A tile re-enqueues a neighbour only when one of the neighbour's candidates touches a pixel it
//   promoted (dedup flag + one append-counter atomic per tile into one of 16 list segments per frame).  Round 0 =
//   every tile, six rounds = the lists on a wide grid, then one persistent workgroup per frame runs its frame's
//   remaining rounds with an agent-scope release/acquire between rounds: the host never waits inside the fixpoint.
// Stage 3 (k_canny_count): np.sum(edges > 0) = the set bits of the final edge plane, one pass.
//
// Roofline: HBM, P + P/4 (+ P/4 per hysteresis pass over live tiles) bytes per frame; in practice stage 1 is bound by
// the CU's scalar unit and the vector ALUs, stage 2 by latency.
#include "vqa_dev.hpp"
#include "vqa_kernels.hpp"
#include "vqa_math.hpp"

#include <cstdlib>

namespace vqa {

constexpr int TW = 64, TH = 32;
constexpr int GP = 72;      // LDS gray row pitch (bytes): [0,4) left halo slot, [4,68) interior, [68,72) right
constexpr int GR = TH + 4;  // gray rows
constexpr int MW = TW + 2, MH = TH + 2;

canny_geom canny_tiles(int h, int w) { return canny_geom{(w + TW - 1) / TW, (h + TH - 1) / TH}; }

// strong / weak: bit-planes in TILE-MAJOR order, [n][tiles_y][ww][64] u64 (ww = ceil(w/64), tiles_y =
// ceil(h/64)): word r of tile (ty, tx) holds pixels (64 ty + r, 64 tx .. 64 tx + 63), bit k <-> column
// 64 tx + k.  A hysteresis wave therefore reads its tile as 512 contiguous bytes.
__host__ __device__ __forceinline__ int64_t bp_index(int f, int y, int tx, int ww, int tiles_y)
{
    return ((((int64_t)f * tiles_y + (y >> 6)) * ww + tx) << 6) + (y & 63);
}

__global__ __launch_bounds__(256) void k_canny_nms(const uint8_t *__restrict__ gray, int pitch, int64_t plane_stride,
                                                   int h, int w, int low, int high,
                                                   unsigned long long *__restrict__ strong,
                                                   unsigned long long *__restrict__ weak, int ww,
                                                   vqa_frame_metrics *__restrict__ res)
{
    __shared__ uint8_t sg[GR * GP];
    __shared__ uint16_t smag[MH * MW];
    __shared__ int sgxy[MH * MW]; // (gy << 16) | (gx & 0xffff)
    __shared__ unsigned s_cnt[2];
    const int f = blockIdx.z;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const uint8_t *g = gray + (int64_t)f * plane_stride;
    const int tid = threadIdx.x;
    if (tid < 2) s_cnt[tid] = 0;
    // ---- stage gray rows y0-2 .. y0+TH+1, columns x0-2 .. x0+TW+1 (replicated borders)
    for (int i = tid; i < GR * 18; i += 256) {
        const int ry = i / 18, u = i - ry * 18; // u: 0 = left halo, 1..16 = interior dwords, 17 = right halo
        const int y = min(max(y0 - 2 + ry, 0), h - 1);
        const uint8_t *row = g + (int64_t)y * pitch;
        uint8_t *d = sg + ry * GP;
        if (u == 0) {
            d[2] = row[min(max(x0 - 2, 0), w - 1)];
            d[3] = row[min(max(x0 - 1, 0), w - 1)];
        } else if (u == 17) {
            d[68] = row[min(x0 + TW, w - 1)];
            d[69] = row[min(x0 + TW + 1, w - 1)];
        } else {
            const int x = x0 + (u - 1) * 4;
            uint32_t v;
            if (x + 4 <= w) {
                v = *(const uint32_t *)(row + x); // pitch % 4 == 0, x % 4 == 0
            } else {
                v = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) v |= (uint32_t)row[min(x + k, w - 1)] << (8 * k);
            }
            *(uint32_t *)(d + 4 + (u - 1) * 4) = v;
        }
    }
    __syncthreads();
    // ---- Sobel + L1 magnitude on the (TH+2) x (TW+2) window; position (ly,lx) <-> image (y0-1+ly, x0-1+lx)
    for (int i = tid; i < MH * MW; i += 256) {
        const int ly = i / MW, lx = i - ly * MW;
        const int y = y0 - 1 + ly, x = x0 - 1 + lx;
        int gx = 0, gy = 0, m = 0;
        if (y >= 0 && y < h && x >= 0 && x < w) {
            // gray sample (yy, xx) lives at sg[(yy - y0 + 2) * GP + (xx - x0 + 4)]
            const uint8_t *c = sg + (ly + 1) * GP + (lx + 3);
            const int p00 = c[-GP - 1], p01 = c[-GP], p02 = c[-GP + 1];
            const int p10 = c[-1], p12 = c[1];
            const int p20 = c[GP - 1], p21 = c[GP], p22 = c[GP + 1];
            gx = (p02 + 2 * p12 + p22) - (p00 + 2 * p10 + p20);
            gy = (p20 + 2 * p21 + p22) - (p00 + 2 * p01 + p02);
            m = abs(gx) + abs(gy);
        }
        smag[i] = (uint16_t)m;
        sgxy[i] = (int)(((uint32_t)gy << 16) | ((uint32_t)gx & 0xffffu));
    }
    __syncthreads();
    // ---- NMS + double threshold on the TH x TW interior; a wave owns one 64-pixel row per step,
    //      so __ballot hands us that row's strong / weak words directly
    unsigned n_strong = 0, n_weak = 0;
    const int tiles_y = (h + 63) >> 6;
    for (int i = tid; i < TH * TW; i += 256) {
        const int ly = i / TW, lx = i - ly * TW;
        const int y = y0 + ly, x = x0 + lx;
        int s = 0;
        if (y < h && x < w) {
            const uint16_t *c = smag + (ly + 1) * MW + (lx + 1);
            const int m = c[0];
            const int nb[8] = {c[-MW - 1], c[-MW], c[-MW + 1], c[-1], c[1], c[MW - 1], c[MW], c[MW + 1]};
            const int pk = sgxy[(ly + 1) * MW + (lx + 1)];
            const int gx = (int)(int16_t)(pk & 0xffff), gy = pk >> 16;
            s = canny_classify(m, gx, gy, nb, low, high);
        }
        const unsigned long long sm = __ballot(s == 2), wm = __ballot(s == 1);
        if (lane_id() == 0 && y < h) {
            const int64_t j = bp_index(f, y, blockIdx.x, ww, tiles_y);
            strong[j] = sm;
            weak[j] = wm;
            n_strong += __popcll(sm);
            n_weak += __popcll(wm);
        }
    }
    if (lane_id() == 0) {
        if (n_strong) atomicAdd(&s_cnt[0], n_strong);
        if (n_weak) atomicAdd(&s_cnt[1], n_weak);
    }
    __syncthreads();
    if (tid == 0) {
        if (s_cnt[0]) atomicAdd(&res[f].edge_strong, s_cnt[0]);
        if (s_cnt[1]) atomicAdd(&res[f].edge_weak, s_cnt[1]);
    }
}

#ifdef VQA_AB_VARIANTS // round 2's kernel, lab build only (VQA_NMS_VARIANT=2)
// ---------------------------------------------------------------------------
// Stage 1, register-rolling form (k_canny_nms2): no LDS, no barriers, no cross-lane traffic until the
// bit-plane words are assembled.  A wave covers 256 columns and marches down a 64-row strip; every lane
// owns FOUR consecutive pixels.  Per row it loads the 8 gray bytes x-2 .. x+5 (clamped = replicated
// border), forms the horizontal Sobel partial sums for pixels x-1 .. x+4 as packed 16-bit pairs
// (v_pk_add/sub/mad_u16: two pixels per instruction), and keeps three rows of them in registers; the
// vertical combination gives gx, gy and |gx|+|gy| for the SAME six pixels, so the magnitudes of the
// lane's left and right neighbours (needed by NMS) are computed locally instead of fetched.  Three rows
// of magnitudes roll in registers; NMS + double threshold runs on the lane's four pixels; the four
// class bits per lane are OR-reduced over each 16-lane DPP row into one u64 bit-plane word per tile row.
// ---------------------------------------------------------------------------
typedef short s2 __attribute__((ext_vector_type(2)));

struct nms_rows {
    s2 h1[3][3], h2[3][3]; // [row slot][pixel pair (-1,0) (1,2) (3,4)]
    s2 mag[3][3];
    s2 gx[3][3], gy[3][3];
};

__device__ __forceinline__ uint32_t dpp_or_row(uint32_t v)
{
    // inclusive prefix OR over the 16 lanes of a DPP row: lane 15 ends up with the OR of all 16
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); // row_shr:1
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true); // row_shr:2
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true); // row_shr:4
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true); // row_shr:8
    return v;
}

template <int PH>
__device__ __forceinline__ void nms2_step(nms_rows &S, int j, const uint8_t *__restrict__ g, int pitch, int h, int w,
                                          int y0, int xm2, const s2 (&cmask)[3], int low, int high, int f, int tx,
                                          int ww, int tiles_y, unsigned long long *__restrict__ strong,
                                          unsigned long long *__restrict__ weak, unsigned &n_strong, unsigned &n_weak)
{
    constexpr int A = (PH + 1) % 3, B = (PH + 2) % 3, Cc = PH; // rows R-2, R-1, R of the h window
    const int lane = lane_id();
    const int R = y0 - 2 + j;
    // ---- horizontal partial sums of gray row R (vertical border: replicate)
    {
        // wave-uniform row base in scalar registers + 32-bit clamped column offsets (loop-invariant): the loads
        // take the saddr form and cost no vector ALU work for addressing
        const gptr_u8 row = uniform_ptr(g + (int64_t)min(max(R, 0), h - 1) * pitch);
        int p[8];
#pragma unroll
        for (int k = 0; k < 8; k++) p[k] = row[(uint32_t)min(max(xm2 + k, 0), w - 1)];
        const s2 P0 = {(short)p[0], (short)p[1]}, Q0 = {(short)p[1], (short)p[2]}, P1 = {(short)p[2], (short)p[3]};
        const s2 Q1 = {(short)p[3], (short)p[4]}, P2 = {(short)p[4], (short)p[5]}, Q2 = {(short)p[5], (short)p[6]};
        const s2 P3 = {(short)p[6], (short)p[7]};
        S.h1[Cc][0] = P1 - P0; S.h1[Cc][1] = P2 - P1; S.h1[Cc][2] = P3 - P2;
        S.h2[Cc][0] = P0 + Q0 + Q0 + P1; S.h2[Cc][1] = P1 + Q1 + Q1 + P2; S.h2[Cc][2] = P2 + Q2 + Q2 + P3;
    }
    // ---- gradient and magnitude of row R-1 (0 outside the image)
    {
        const bool row_in = (R - 1 >= 0) && (R - 1 < h);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const s2 gx = S.h1[A][i] + S.h1[B][i] + S.h1[B][i] + S.h1[Cc][i];
            const s2 gy = S.h2[Cc][i] - S.h2[A][i];
            const s2 m = (__builtin_elementwise_abs(gx) + __builtin_elementwise_abs(gy)) & cmask[i];
            S.gx[B][i] = gx;
            S.gy[B][i] = gy;
            S.mag[B][i] = row_in ? m : s2{0, 0};
        }
    }
    // ---- NMS + double threshold of row R-2: magnitudes of rows R-3 (slot Cc), R-2 (A), R-1 (B)
    const int y = R - 2;
    if (j >= 4 && y < h) { // wave-uniform
        // six magnitudes per row: index 0..5 <-> pixels -1..4
        int up[6], md[6], dn[6];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            up[2 * i] = S.mag[Cc][i].x; up[2 * i + 1] = S.mag[Cc][i].y;
            md[2 * i] = S.mag[A][i].x;  md[2 * i + 1] = S.mag[A][i].y;
            dn[2 * i] = S.mag[B][i].x;  dn[2 * i + 1] = S.mag[B][i].y;
        }
        const int gxs[4] = {S.gx[A][0].y, S.gx[A][1].x, S.gx[A][1].y, S.gx[A][2].x};
        const int gys[4] = {S.gy[A][0].y, S.gy[A][1].x, S.gy[A][1].y, S.gy[A][2].x};
        uint32_t nib_s = 0, nib_w = 0;
        // a row of 256 pixels with no gradient above `low` (flat regions) skips NMS and the word assembly
        const bool any_cand = __any(max(max(md[1], md[2]), max(md[3], md[4])) > low);
        if (!any_cand) {
            if ((lane & 15) == 15 && tx < ww) {
                const int64_t o = bp_index(f, y, tx, ww, tiles_y);
                strong[o] = 0ull;
                weak[o] = 0ull;
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int nb[8] = {up[k], up[k + 1], up[k + 2], md[k], md[k + 2], dn[k], dn[k + 1], dn[k + 2]};
            const int c = canny_classify(md[k + 1], gxs[k], gys[k], nb, low, high);
            nib_s |= (uint32_t)(c == 2) << k;
            nib_w |= (uint32_t)(c == 1) << k;
        }
        // lane (16 q + i) holds columns 4 i .. 4 i + 3 of tile-row word q: bits 4 i .. 4 i + 3
        const int i16 = lane & 15;
        const uint32_t sh = (uint32_t)(i16 & 7) * 4;
        uint32_t s_lo = i16 < 8 ? nib_s << sh : 0u, s_hi = i16 < 8 ? 0u : nib_s << sh;
        uint32_t w_lo = i16 < 8 ? nib_w << sh : 0u, w_hi = i16 < 8 ? 0u : nib_w << sh;
        s_lo = dpp_or_row(s_lo); s_hi = dpp_or_row(s_hi);
        w_lo = dpp_or_row(w_lo); w_hi = dpp_or_row(w_hi);
        if (i16 == 15 && tx < ww) {
            const int64_t o = bp_index(f, y, tx, ww, tiles_y);
            strong[o] = (unsigned long long)s_lo | ((unsigned long long)s_hi << 32);
            weak[o] = (unsigned long long)w_lo | ((unsigned long long)w_hi << 32);
            n_strong += __popc(s_lo) + __popc(s_hi);
            n_weak += __popc(w_lo) + __popc(w_hi);
        }
    }
}

// grid = (ceil(w / 256), ceil(h / 64), n_frames), block = 64 (one wave)
__global__ __launch_bounds__(64) void k_canny_nms2(const uint8_t *__restrict__ gray, int pitch, int64_t plane_stride,
                                                   int h, int w, int low, int high,
                                                   unsigned long long *__restrict__ strong,
                                                   unsigned long long *__restrict__ weak, int ww,
                                                   vqa_frame_metrics *__restrict__ res)
{
    const int f = blockIdx.z;
    const int lane = lane_id();
    const int x0 = blockIdx.x * 256, y0 = blockIdx.y * 64;
    const int x = x0 + 4 * lane; // first of the lane's four pixels
    const uint8_t *g = gray + (int64_t)f * plane_stride;
    const int tiles_y = (h + 63) >> 6;
    const int tx = blockIdx.x * 4 + (lane >> 4);
    // pixels -1..4 of this lane that lie inside the image keep their magnitude, the others read 0
    s2 cmask[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int xa = x - 1 + 2 * i, xb = xa + 1;
        cmask[i] = s2{(short)((xa >= 0 && xa < w) ? -1 : 0), (short)((xb >= 0 && xb < w) ? -1 : 0)};
    }
    nms_rows S;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int i = 0; i < 3; i++) {
            S.h1[r][i] = S.h2[r][i] = S.mag[r][i] = S.gx[r][i] = S.gy[r][i] = s2{0, 0};
        }
    unsigned n_strong = 0, n_weak = 0;
    const int rows_out = min(64, h - y0);
    const int steps = rows_out + 4;
    for (int j0 = 0; j0 < steps; j0 += 3) {
        nms2_step<0>(S, j0, g, pitch, h, w, y0, x - 2, cmask, low, high, f, tx, ww, tiles_y, strong, weak, n_strong, n_weak);
        if (j0 + 1 < steps)
            nms2_step<1>(S, j0 + 1, g, pitch, h, w, y0, x - 2, cmask, low, high, f, tx, ww, tiles_y, strong, weak, n_strong, n_weak);
        if (j0 + 2 < steps)
            nms2_step<2>(S, j0 + 2, g, pitch, h, w, y0, x - 2, cmask, low, high, f, tx, ww, tiles_y, strong, weak, n_strong, n_weak);
    }
    n_strong = wave_sum(n_strong);
    n_weak = wave_sum(n_weak);
    if (lane == 0) {
        if (n_strong) atomicAdd(&res[f].edge_strong, n_strong);
        if (n_weak) atomicAdd(&res[f].edge_weak, n_weak);
    }
}

#endif // VQA_AB_VARIANTS

// ---------------------------------------------------------------------------
// Stage 1, lane-per-column form (k_canny_nms3, the default): a wave still covers 256 columns x a 64-row strip
// (= 4 hysteresis tiles), but lane L owns the FOUR COLUMNS x0 + 64 k + L (k = 0..3), so the 64 lanes of group k
// are the 64 columns of tile k and a vector compare of group k IS that tile row's bit-plane word (one
// v_cmp writing an SGPR pair): every NMS decision is taken on 64-pixel masks by the scalar unit, there is no
// per-pixel branch, no nibble assembly and no unpacking.
//   * gray: ONE unaligned buffer_load_dword per pixel (the bytes x-1 .. x+2; the lane offset is a loop-invariant
//     VGPR, the row offset the instruction's scalar offset, so addressing costs no vector instruction), requested one
//     row ahead; a wave that touches the image border realigns the replicated-border bytes with one v_perm_b32;
//   * gradient, 7 vector instructions per pixel: h1 = p[x+1] - p[x-1] (v_sub_u32_sdwa), h2 = p[x-1] + 2 p[x] + p[x+1]
//     (v_dot4_u32_u8); rows are combined through running sums (gx + 1024 = s[R-1] + s[R], s[R] = h1[R-1] + h1[R] + 512)
//     and the L1 magnitude is two v_sad_u32: ax = |gx + 1024 - 1024|, m = |h2[R] - h2[R-2]| + ax (gy is never formed);
//   * horizontal neighbours: the five column groups (4 tiles + a halo group whose lane 63 / lane 0 are the
//     columns left / right of the wave's span) form a ring; wave_ror:1 / wave_rol:1 DPP moves plus one select
//     for the seam lane give every row's m[x-1] and m[x+1] once, when the row is new;
//   * sector test on integers (TG22 fixed point, OpenCV's own), as two unsigned compares of m << 15 against
//     46341 |gx| and 111877 |gx| (|gy| = m - |gx|); the diagonal's sign test (gx ^ gy) < 0 is two compares;
//   * all compares between two consecutive rows are made once, when the lower row is new, and serve both rows
//     (as "below" of the upper one, as "above" of the lower one): only TWO rows of magnitudes live in registers,
//     the upper row's partial decisions wait in scalar registers as masks;
//   * lane r captures row r's words (v_writelane), so after the strip every lane holds its row of the 4 tiles -
//     the layout the hysteresis uses - and the bit-planes leave as 512-byte contiguous tile stores.
// ---------------------------------------------------------------------------
// |a - b| + c: the compiler selects v_sad_u32 for this shape.  NOT inline asm: the hazard recognizer does not see
// an asm statement as a vector-ALU reader, so a v_sad_u32 written in asm right behind the v_dot4_u32_u8 that
// produces its operand misses the wait states gfx950 needs between a DOT result and its first VALU use and reads a
// stale register (measured: wrong magnitudes whose pattern changed with unrelated code edits).
__device__ __forceinline__ uint32_t sad_u32(uint32_t a, uint32_t b, uint32_t c) { return (max(a, b) - min(a, b)) + c; }
// (mov_dpp: every lane has a source lane in a rotation, so there is no "old" value to pre-load into the destination)
__device__ __forceinline__ uint32_t wave_ror1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x13C, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t wave_rol1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x134, 0xf, 0xf, false); }

// lane `lane` of (a, b) <- the wave-uniform words (lo, hi).  The lane select goes through M0 (a VOP3 on gfx9 reads at most
// one SGPR besides it).  M0 is written and consumed INSIDE one asm statement that declares the clobber: the compiler can
// neither schedule anything that touches M0 between the write and its readers nor assume M0 survives the statement
// (round 3 set M0 in one asm and read it in others, which only held as long as the compiler emitted nothing
// M0-related in between).
__device__ __forceinline__ void writelane2(uint32_t &a, uint32_t &b, int lane, uint32_t lo, uint32_t hi)
{
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0"
                 : "+v"(a), "+v"(b)
                 : "s"(lane), "s"(lo), "s"(hi)
                 : "m0");
}

typedef unsigned long long u64;

struct nms3_state {
    // gradient pipeline, 5 column groups (0..3 = tiles, 4 = halo): previous row's h1, previous running sum,
    // h2 of the two previous rows (slot = row parity)
    uint32_t h1[2][5], s[2][5], h2[2][5]; // slot = row parity: nothing rotates through register copies
    // magnitudes of the previous mag row (slot PH^1) and the new one (slot PH), with their left / right shifted copies
    uint32_t m[2][4], mL[2][4], mR[2][4];
    // decisions of the previous mag row that wait for the row below it (wave-uniform masks)
    u64 HH[4], VV[4], DO[4], DS[4];
    // captured words: lane r <-> row r of the strip
    uint32_t cs[4][2], cw[4][2];
    // gray bytes of the NEXT step's row, loaded one step ahead (the loop is otherwise a chain of dependent loads:
    // with 4 waves per SIMD nothing else hides a row's memory latency)
    uint32_t pn[2][5];
};

// one UNALIGNED dword per column group: the bytes at columns x-1 .. x+2 of the lane's pixel.  Buffer loads: the lane
// offset is a loop-invariant VGPR and the row offset the instruction's scalar offset, so addressing costs no vector
// instruction (byte-unaligned dword addresses are honoured: scripts/probes/unaligned_dword.hip)
__device__ __forceinline__ void nms3_load(uint32_t (&p)[5], const __amdgpu_buffer_rsrc_t rs, const int (&off)[5], int R, int h, int pitch)
{
    const int soff = min(max(R, 0), h - 1) * pitch; // vertical border: replicate
#pragma unroll
    for (int g = 0; g < 5; g++) p[g] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, off[g], soff, 0);
}

template <int PH, bool EDGE>
__device__ __forceinline__ void nms3_step(nms3_state &S, int j, const __amdgpu_buffer_rsrc_t rs, int pitch, int h, int y0,
                                          const int (&off)[5], const uint32_t (&sel)[5], const uint32_t (&cmask)[5],
                                          const u64 (&colmask)[4], int low, int high, bool lane0, bool lane63)
{
    const int R = y0 - 2 + j;            // gray row loaded by this step
    const int yy = R - 1;                // magnitude row completed by this step
    // ---- this row's bytes were requested by the previous step; request the next row's now
    // (the fence keeps the scheduler from sinking the new loads below the arithmetic, which would turn the
    // prefetch back into a load-use chain)
    nms3_load(S.pn[PH ^ 1], rs, off, R + 1, h, pitch);
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t (&p)[5] = S.pn[PH];
    // ---- gradient + L1 magnitude of row yy
    const bool row_in = yy >= 0 && yy < h;
    uint32_t zero = 0;
    asm("" : "+v"(zero));
    uint32_t mnew[5], ax[4];
    u64 opp[4];
#pragma unroll
    for (int g = 0; g < 5; g++) {
        // bytes 0..2 = p[x-1], p[x], p[x+1].  A wave that touches the image's left or right border loaded from a
        // clamped offset and puts its bytes in place with one v_perm (replicated border); interior waves use
        // the dword as it is (byte 3 = p[x+2] has weight 0 below)
        const uint32_t v = EDGE ? __builtin_amdgcn_perm(p[g], p[g], sel[g]) : p[g];
        const uint32_t h1 = ((v >> 16) & 0xffu) - (v & 0xffu);                 // v_sub_u32_sdwa
        const uint32_t h2 = __builtin_amdgcn_udot4(v, 0x00010201u, 0u, false);   // p[x-1] + 2 p[x] + p[x+1]
        const uint32_t s = S.h1[PH ^ 1][g] + h1 + 512u;
        const uint32_t gxb = S.s[PH ^ 1][g] + s;          // gx + 1024, always positive
        const uint32_t a = sad_u32(gxb, 1024u, zero);     // |gx| (zero: an opaque 0, the + 0 would be folded and the pattern lost)
        const uint32_t h2a = S.h2[PH][g];                 // row R-2 (same parity as R)
        // |gx| + |gy|, 0 outside the image's columns (an interior wave has no such column: its halo group's unused
        // lanes repeat a valid column and are never read)
        mnew[g] = EDGE ? (sad_u32(h2, h2a, a) & cmask[g]) : sad_u32(h2, h2a, a);
        __builtin_assume(a <= 1020u);
        if (g < 4) {
            ax[g] = a;
            opp[g] = __ballot(gxb < 1024u) ^ __ballot(h2 < h2a); // (gx ^ gy) < 0: gx < 0 differs from gy < 0
        }
        S.h1[PH][g] = h1; S.s[PH][g] = s; S.h2[PH][g] = h2;
    }
#ifndef NMS3_PROBE
#define NMS3_PROBE 0   // measurement builds only (scripts/build_probes.sh): 1 no capture, 2 no decisions, 3 no neighbours either
#endif
    // ---- left / right neighbours of the new row through the ring H,0,1,2,3,H
    if (NMS3_PROBE < 3) {
        // (named scalars, not arrays: the optimiser turns "lane0 ? a[i] : a[j]" into a select of the INDEX followed by
        // a four-deep select chain over the array)
        const uint32_t r0 = wave_ror1(mnew[0]), r1 = wave_ror1(mnew[1]), r2 = wave_ror1(mnew[2]), r3 = wave_ror1(mnew[3]);
        const uint32_t rH = wave_ror1(mnew[4]);
        const uint32_t l0 = wave_rol1(mnew[0]), l1 = wave_rol1(mnew[1]), l2 = wave_rol1(mnew[2]), l3 = wave_rol1(mnew[3]);
        const uint32_t lH = wave_rol1(mnew[4]);
        S.m[PH][0] = mnew[0]; S.m[PH][1] = mnew[1]; S.m[PH][2] = mnew[2]; S.m[PH][3] = mnew[3];
        S.mL[PH][0] = lane0 ? rH : r0; S.mL[PH][1] = lane0 ? r0 : r1; S.mL[PH][2] = lane0 ? r1 : r2; S.mL[PH][3] = lane0 ? r2 : r3;
        S.mR[PH][0] = lane63 ? l1 : l0; S.mR[PH][1] = lane63 ? l2 : l1; S.mR[PH][2] = lane63 ? l3 : l2; S.mR[PH][3] = lane63 ? lH : l3;
        if (!row_in) { // rows -1 and h have magnitude 0.  A real (wave-uniform) branch taken twice per frame column: the asm
                       // keeps the optimiser from turning it into twelve selects on every row ("+v": no copies either)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                asm volatile("v_mov_b32 %0, 0" : "+v"(S.m[PH][k]));
                asm volatile("v_mov_b32 %0, 0" : "+v"(S.mL[PH][k]));
                asm volatile("v_mov_b32 %0, 0" : "+v"(S.mR[PH][k]));
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) { S.m[PH][k] = mnew[k] + mnew[4]; S.cs[k][0] += S.m[PH][k] + ax[k] + (uint32_t)opp[k]; }
    }
    if (NMS3_PROBE >= 2) {
#pragma unroll
        for (int k = 0; k < 4; k++) S.cs[k][1] += S.m[PH][k] + S.mL[PH][k] + S.mR[PH][k] + ax[k] + (uint32_t)opp[k];
        return;
    }
    // ---- per tile: finish the row above (it now has its lower neighbours), prepare the new row.  The scalar unit
    // (one per CU, shared by the four SIMDs) is this kernel's scarcest resource: no per-class branches
    const int yout = yy - 1 - y0; // strip row finished by this step (wave-uniform); lane yout captures its row
    u64 ST_new[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t m1 = S.m[PH][k], m1L = S.mL[PH][k], m1R = S.mR[PH][k];
        const uint32_t m2 = S.m[PH ^ 1][k], m2L = S.mL[PH ^ 1][k], m2R = S.mR[PH ^ 1][k];
        const u64 gt = __ballot(m1 > m2); // serves both rows: "m2 >= m1" of the row above, "m1 > m2" of the new row
        // row above: vertical needs m2 >= m1, same-sign diagonal m2 > m1[x+1], opposite-sign diagonal m2 > m1[x-1]
        const u64 keep = S.HH[k] | (S.VV[k] & ~gt) | (S.DS[k] & __ballot(m2 > m1R)) | (S.DO[k] & __ballot(m2 > m1L));
        if (NMS3_PROBE == 1) { S.cs[k][0] += (uint32_t)keep; S.cw[k][0] += (uint32_t)(keep >> 32); }
        else if (yout >= 0) { // lane yout captures the row's kept pixels (its above-high word was captured a step ago)
            writelane2(S.cs[k][0], S.cs[k][1], yout, (uint32_t)keep, (uint32_t)(keep >> 32));
        }
        // new row as the centre: direction sectors, same-row and upward compares
        u64 cand = __ballot((int)m1 > low);
        if (EDGE) cand &= colmask[k]; // (m is 0 outside the image; this only matters for a negative threshold)
        u64 HH = 0, VV = 0, DO = 0, DS = 0, ST = 0;
        if (cand) { // (also keeps the masks of one tile from being live across the next tile's compares: without it the
                    // scheduler hoists every compare and the kernel spills scalar registers)
            // OpenCV: y = |gy| << 15, tg22x = |gx| * 13573, tg67x = tg22x + (|gx| << 16); horizontal iff y < tg22x,
            // vertical iff y > tg67x.  With |gy| = m - |gx| both become compares of m << 15 (all values < 2^28)
            const uint32_t u = m1 << 15;
            const uint32_t t1 = __umul24(ax[k], 46341u);       // tg22x + (|gx| << 15)
            const uint32_t t2 = t1 + (ax[k] << 16);            // tg67x + (|gx| << 15)
            const u64 ch = __ballot(u < t1), cv = __ballot(u > t2);
            const u64 nh = cand & ~ch, ds = nh & ~cv;
            HH = cand & ch & __ballot(m1 > m1L) & __ballot(m1 >= m1R);
            VV = nh & cv & gt;
            DS = ds & ~opp[k] & __ballot(m1 > m2L);
            DO = ds & opp[k] & __ballot(m1 > m2R);
            ST = __ballot((int)m1 > high);
        }
        S.HH[k] = HH; S.VV[k] = VV; S.DO[k] = DO; S.DS[k] = DS; ST_new[k] = ST;
    }
    // the new row's above-high words go to lane yout + 1 now (8 scalar registers fewer to carry into the next step)
    if (NMS3_PROBE != 1 && yout + 1 >= 0 && yout + 1 < 64) {
#pragma unroll
        for (int k = 0; k < 4; k++)
            writelane2(S.cw[k][0], S.cw[k][1], yout + 1, (uint32_t)ST_new[k], (uint32_t)(ST_new[k] >> 32));
    }
}

// grid = 8 * ceil(strips / 8) workgroups of one wave, strips = ceil(w / 256) * ceil(h / 64) * n_frames.
// XCD-aware placement: workgroup ids go round-robin over the 8 XCDs (id % 8), and at 1080p a strip row has exactly 8
// strips - with the plain (x, y, frame) grid every strip's left and right neighbours sat on OTHER XCDs, so the two
// 128-byte lines that hold a strip's halo columns were fetched once per L2 (measured: 1.10 GB fetched for a 0.56 GB
// footprint).  Here XCD c takes the contiguous strip range [c * per, (c + 1) * per): neighbours share an L2 and are
// dispatched back to back.
__global__ __launch_bounds__(64) void k_canny_nms3(const uint8_t *__restrict__ gray, int pitch, int64_t plane_stride,
                                                   int h, int w, int low, int high,
                                                   unsigned long long *__restrict__ strong,
                                                   unsigned long long *__restrict__ weak, int ww,
                                                   vqa_frame_metrics *__restrict__ res, int strips_x, int strips_y,
                                                   int n_strips)
{
    const int per = (n_strips + 7) >> 3;
    const int t = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || t >= n_strips) return; // (uniform: the whole wave leaves)
    const int bx = t % strips_x, by = (t / strips_x) % strips_y;
    const int f = t / (strips_x * strips_y);
    const int lane = lane_id();
    const int x0 = bx * 256, y0 = by * 64;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(gray + (int64_t)f * plane_stride), (short)0, -1, 0x00020000);
    const int tiles_y = (h + 63) >> 6;
    // column of (group, lane); the halo group holds x0 - 1 in lane 63 and x0 + 256 in lane 0.  Every lane loads
    // the dword at clamp(x - 1, 0, w - 4); sel[] maps the replicated-border columns x-1, x, x+1 into that dword
    int off[5];
    uint32_t sel[5], cmask[5];
    u64 colmask[4];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int x = k < 4 ? x0 + 64 * k + lane : (lane == 63 ? x0 - 1 : x0 + 256);
        const bool in = x >= 0 && x < w && (k < 4 || lane == 0 || lane == 63);
        cmask[k] = in ? ~0u : 0u;
        if (k < 4) colmask[k] = __ballot(in);
        const int xc = min(max(x, -1), w); // columns beyond the border behave like the first one outside (magnitude 0 anyway)
        off[k] = min(max(xc - 1, 0), w - 4);
        uint32_t sl = 0x0c000000u;
#pragma unroll
        for (int t = 0; t < 3; t++) sl |= (uint32_t)(min(max(xc - 1 + t, 0), w - 1) - off[k]) << (8 * t);
        sel[k] = sl;
    }
    const bool edge = x0 == 0 || x0 + 259 > w; // wave-uniform: some lane's dword window (x-1 .. x+2) leaves the image
    nms3_state S;
#pragma unroll
    for (int k = 0; k < 5; k++) { S.h1[0][k] = S.h1[1][k] = S.s[0][k] = S.s[1][k] = S.h2[0][k] = S.h2[1][k] = 0; }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        S.m[0][k] = S.m[1][k] = S.mL[0][k] = S.mL[1][k] = S.mR[0][k] = S.mR[1][k] = 0;
        S.HH[k] = S.VV[k] = S.DO[k] = S.DS[k] = 0;
        S.cs[k][0] = S.cs[k][1] = S.cw[k][0] = S.cw[k][1] = 0;
    }
    const bool lane0 = lane == 0, lane63 = lane == 63;
    const int rows_out = min(64, h - y0);
    const int steps = rows_out + 4;
    nms3_load(S.pn[0], rs, off, y0 - 2, h, pitch);
    if (edge) {
        for (int j0 = 0; j0 < steps; j0 += 2) {
            nms3_step<0, true>(S, j0, rs, pitch, h, y0, off, sel, cmask, colmask, low, high, lane0, lane63);
            if (j0 + 1 < steps) nms3_step<1, true>(S, j0 + 1, rs, pitch, h, y0, off, sel, cmask, colmask, low, high, lane0, lane63);
        }
    } else {
        for (int j0 = 0; j0 < steps; j0 += 2) {
            nms3_step<0, false>(S, j0, rs, pitch, h, y0, off, sel, cmask, colmask, low, high, lane0, lane63);
            if (j0 + 1 < steps) nms3_step<1, false>(S, j0 + 1, rs, pitch, h, y0, off, sel, cmask, colmask, low, high, lane0, lane63);
        }
    }
    // ---- lane r holds row y0 + r of the four tiles: contiguous tile stores, strong / weak totals
    unsigned n_strong = 0, n_weak = 0;
    if (lane < rows_out) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int tx = bx * 4 + k;
            if (tx < ww) {
                const int64_t o = bp_index(f, y0 + lane, tx, ww, tiles_y);
                const uint32_t s_lo = S.cs[k][0] & S.cw[k][0], s_hi = S.cs[k][1] & S.cw[k][1];
                const uint32_t w_lo = S.cs[k][0] & ~S.cw[k][0], w_hi = S.cs[k][1] & ~S.cw[k][1];
                strong[o] = (u64)s_lo | ((u64)s_hi << 32);
                weak[o] = (u64)w_lo | ((u64)w_hi << 32);
                n_strong += __popc(s_lo) + __popc(s_hi);
                n_weak += __popc(w_lo) + __popc(w_hi);
            }
        }
    }
    n_strong = wave_sum(n_strong);
    n_weak = wave_sum(n_weak);
    if (lane == 0) {
        if (n_strong) atomicAdd(&res[f].edge_strong, n_strong);
        if (n_weak) atomicAdd(&res[f].edge_weak, n_weak);
    }
}

// ---------------------------------------------------------------------------
// Hysteresis on bit-planes.  Tile = 64 columns x 64 rows = one wave; lane r <-> row r.
// ---------------------------------------------------------------------------

__device__ __forceinline__ u64 shfl_up64(u64 v) { return __shfl_up(v, 1, 64); }
__device__ __forceinline__ u64 shfl_dn64(u64 v) { return __shfl_down(v, 1, 64); }

// 64-bit rows as two 32-bit halves: gfx950 issues 64-bit shifts at a fraction of the 32-bit rate, so the
// shift/and/or ladder is written on halves with v_alignbit_b32 carrying bits across the middle.
struct row64 {
    uint32_t lo, hi;
};
__device__ __forceinline__ row64 to_row(u64 v) { return row64{(uint32_t)v, (uint32_t)(v >> 32)}; }
__device__ __forceinline__ u64 to_u64(row64 r) { return (u64)r.lo | ((u64)r.hi << 32); }
template <int K>
__device__ __forceinline__ row64 shl(row64 a)
{
    if (K == 32) return row64{0u, a.lo};
    return row64{a.lo << K, __builtin_amdgcn_alignbit(a.hi, a.lo, 32 - K)}; // (hi:lo) >> (32-K), low word
}
template <int K>
__device__ __forceinline__ row64 shr(row64 a)
{
    if (K == 32) return row64{a.hi, 0u};
    return row64{__builtin_amdgcn_alignbit(a.hi, a.lo, K), a.hi >> K};
}
__device__ __forceinline__ row64 operator&(row64 a, row64 b) { return row64{a.lo & b.lo, a.hi & b.hi}; }
__device__ __forceinline__ row64 operator|(row64 a, row64 b) { return row64{a.lo | b.lo, a.hi | b.hi}; }
// a | (m & t): one v_and_or_b32 per half
__device__ __forceinline__ row64 and_or(row64 m, row64 t, row64 a) { return row64{(m.lo & t.lo) | a.lo, (m.hi & t.hi) | a.hi}; }

#ifdef VQA_AB_VARIANTS
// (round 2's ladder) occluded fill along a row: every run of `pro` cells touching a `gen` cell becomes gen
__device__ __forceinline__ row64 flood_row(row64 gen, row64 pro)
{
    row64 a = gen, m = pro;
    a = and_or(m, shl<1>(a), a);  m = m & shl<1>(m);
    a = and_or(m, shl<2>(a), a);  m = m & shl<2>(m);
    a = and_or(m, shl<4>(a), a);  m = m & shl<4>(m);
    a = and_or(m, shl<8>(a), a);  m = m & shl<8>(m);
    a = and_or(m, shl<16>(a), a); m = m & shl<16>(m);
    a = and_or(m, shl<32>(a), a);
    row64 b = gen;
    m = pro;
    b = and_or(m, shr<1>(b), b);  m = m & shr<1>(m);
    b = and_or(m, shr<2>(b), b);  m = m & shr<2>(m);
    b = and_or(m, shr<4>(b), b);  m = m & shr<4>(m);
    b = and_or(m, shr<8>(b), b);  m = m & shr<8>(m);
    b = and_or(m, shr<16>(b), b); m = m & shr<16>(m);
    b = and_or(m, shr<32>(b), b);
    return a | b;
}

// 3-wide horizontal dilation of a row plus the tile's left (bit 0) and right (bit 63) halo bits
__device__ __forceinline__ row64 dil3(row64 s, uint32_t l, uint32_t r)
{
    const row64 sl = shl<1>(s), sr = shr<1>(s);
    return row64{s.lo | sl.lo | sr.lo | l, s.hi | sl.hi | sr.hi | (r << 31)};
}
#endif

// lane i <- lane i-1 (lane 0 keeps its own) / lane i <- lane i+1 (lane 63 keeps its own): whole-wave DPP shifts
// (wave_shr:1 / wave_shl:1, one vector-ALU instruction) instead of ds_bpermute round trips through the LDS pipe -
// the relaxation loop is a dependent chain, so the shuffle latency is on its critical path
__device__ __forceinline__ uint32_t lane_up32(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t lane_dn32(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ row64 shfl_up_row(row64 v) { return row64{lane_up32(v.lo), lane_up32(v.hi)}; }
__device__ __forceinline__ row64 shfl_dn_row(row64 v) { return row64{lane_dn32(v.lo), lane_dn32(v.hi)}; }

// Work lists: per frame HSEG segments of capacity tiles-per-frame each, a tile appends to segment (its id % HSEG).
// One counter per frame put ~2000 same-address atomics in a row at 2160p (64 frames x 2040 tiles): L2 serialises
// them, and round 0 took 0.73 ms there against 0.39 ms at 1080p for the same 130 560 tiles and the same 16-17
// relaxation steps per tile.  16 counters per frame cut the queue behind each address 16-fold.
constexpr int HSEG = 16;
__host__ __device__ __forceinline__ size_t hyst_list_index(unsigned f, unsigned seg, unsigned tpf, unsigned i)
{
    return ((size_t)f * HSEG + seg) * tpf + i;
}

// item g of the concatenation of a frame's HSEG segments -> (segment, index inside it); cnt[] = the segment sizes
__device__ __forceinline__ void hyst_locate(const unsigned (&cnt)[HSEG], unsigned g, unsigned &seg, unsigned &idx)
{
    seg = 0; idx = g;
#pragma unroll
    for (int k = 0; k < HSEG - 1; k++)
        if (seg == (unsigned)k && idx >= cnt[k]) { idx -= cnt[k]; seg = k + 1; }
}

struct hyst_args {
    u64 *strong;
    const u64 *weak;
    int h, ww, tiles_y;       // tiles_x == ww
    unsigned *queued;         // per tile: already in the list being BUILT (flags of the list being consumed
                              // live in a second array, so "queued for this round" never hides a re-enqueue)
    unsigned *out_list;       // tile ids for the next round, one segment of tiles_y*ww entries PER FRAME
    unsigned *out_count;      // HSEG append counters per frame, chosen by the ENQUEUING tile (see HSEG)
    vqa_frame_metrics *res;
    int sub;                  // cheap vertical sub-steps per horizontal flood (see relax_tile)
    int stats;                // accumulate the diagnostic hyst_steps (one more atomic per tile visit)
};

// Relax one tile to its local fixpoint.  tile id = (f * tiles_y + ty) * ww + tx.  Whole wave calls this.
__device__ __forceinline__ void relax_tile(const hyst_args &A, unsigned tile)
{
    const int lane = lane_id();
    const int tx = tile % A.ww;
    const int ty = (tile / A.ww) % A.tiles_y;
    const int f = tile / (A.ww * A.tiles_y);
    const int y = ty * 64 + lane;
    const bool in_img = y < A.h;
    const int64_t idx = bp_index(f, in_img ? y : ty * 64, tx, A.ww, A.tiles_y);
    // all of the tile's loads (own rows, left/right neighbours, the rows above and below) are issued BEFORE the
    // "nothing to promote" test: one memory round trip per tile instead of two
    const u64 s0 = in_img ? A.strong[idx] : 0ull;
    const u64 w0 = in_img ? A.weak[idx] : 0ull;
    // constant halo: the columns left / right of the tile and the rows above / below it.  Besides the
    // halo's edge bits (h*, e*) we keep its CANDIDATE bits (weak and not yet edge: c*, ec*): a neighbour is
    // only worth re-visiting if one of its candidates touches a pixel this tile promotes.
    u64 sL = 0, wL = 0, sR = 0, wR = 0;
    if (in_img && tx > 0) { sL = A.strong[idx - 64]; wL = A.weak[idx - 64]; }
    if (in_img && tx + 1 < A.ww) { sR = A.strong[idx + 64]; wR = A.weak[idx + 64]; }
    u64 es = 0, ew = 0, tl = 0, twl = 0, tr = 0, twr = 0; // lane 0: row above the tile; lane 63: row below it
    {
        const int yy = lane == 0 ? ty * 64 - 1 : (lane == 63 ? ty * 64 + 64 : -1);
        if (yy >= 0 && yy < A.h) {
            const int64_t j = bp_index(f, yy, tx, A.ww, A.tiles_y);
            es = A.strong[j];
            ew = A.weak[j];
            if (tx > 0) { tl = A.strong[j - 64]; twl = A.weak[j - 64]; }
            if (tx + 1 < A.ww) { tr = A.strong[j + 64]; twr = A.weak[j + 64]; }
        }
    }
    const u64 w = w0 & ~s0;
    if (!__any(w != 0)) return; // nothing here can be promoted
    const u64 hl = sL >> 63, cl = (wL & ~sL) >> 63, hr = sR & 1ull, cr = (wR & ~sR) & 1ull;
    const u64 ecs = ew & ~es;
    const u64 el = tl >> 63, ecl = (twl & ~tl) >> 63, er = tr & 1ull, ecr = (twr & ~tr) & 1ull;
    const uint32_t hl32 = (uint32_t)hl, hr32 = (uint32_t)hr;
    uint32_t up_l = lane_up32(hl32), up_r = lane_up32(hr32);
    uint32_t dn_l = lane_dn32(hl32), dn_r = lane_dn32(hr32);
    if (lane == 0) { up_l = (uint32_t)el; up_r = (uint32_t)er; }
    if (lane == 63) { dn_l = (uint32_t)el; dn_r = (uint32_t)er; }
    [[maybe_unused]] const row64 wr = to_row(w), F = to_row(s0 | w); // (the lab build's ladder)
    const row64 esr = to_row(es);
    row64 sr = to_row(s0);
    unsigned steps = 0;
#ifdef VQA_AB_VARIANTS
    if (A.sub == 0)
#endif
    {
        // The schedule.  One iteration = ONE 3x3 dilation step (rows above/below by whole-wave DPP shifts whose
        // `old` operand supplies the halo row to lane 0 / lane 63, then a 3-wide horizontal dilation of the OR of the
        // three rows) followed by a flood along the rows done by the integer adder: for seeds S inside a mask F of
        // runs, F + S carries from every seed to the end of its run, so (F & ~(F + S)) | S is the run filled from the
        // seed towards bit 63; the other direction is the same on bit-reversed words.  A horizontal chain of any
        // length closes in one iteration for ~20 instructions (the shift/and/or ladder it replaces took ~90).
        const u64 F64 = s0 | w, rF64 = __builtin_bitreverse64(F64);
        const uint32_t hl_any = up_l | hl32 | dn_l, hr_any = (up_r | hr32 | dn_r) << 31; // halo columns, three rows
        u64 cur = s0;
        for (;;) {
            steps++;
            const uint32_t lo = (uint32_t)cur, hi = (uint32_t)(cur >> 32);
            const uint32_t up_lo = (uint32_t)__builtin_amdgcn_update_dpp((int)esr.lo, (int)lo, 0x138, 0xf, 0xf, false);
            const uint32_t up_hi = (uint32_t)__builtin_amdgcn_update_dpp((int)esr.hi, (int)hi, 0x138, 0xf, 0xf, false);
            const uint32_t dn_lo = (uint32_t)__builtin_amdgcn_update_dpp((int)esr.lo, (int)lo, 0x130, 0xf, 0xf, false);
            const uint32_t dn_hi = (uint32_t)__builtin_amdgcn_update_dpp((int)esr.hi, (int)hi, 0x130, 0xf, 0xf, false);
            const row64 v = row64{up_lo | lo | dn_lo, up_hi | hi | dn_hi};
            const row64 vl = shl<1>(v), vr = shr<1>(v);
            const uint32_t d_lo = v.lo | vl.lo | vr.lo | hl_any, d_hi = v.hi | vl.hi | vr.hi | hr_any;
            const u64 cand = w & ~cur & ((u64)d_lo | ((u64)d_hi << 32));
            if (!__any(cand != 0)) break; // nothing anywhere in the tile => fixpoint
            const u64 seed = cur | cand;
            const u64 rs = __builtin_bitreverse64(seed);
            cur = (F64 & ~(F64 + seed)) | __builtin_bitreverse64(rF64 & ~(rF64 + rs)) | seed;
        }
        sr = to_row(cur);
    }
#ifdef VQA_AB_VARIANTS
    else {
        // (round 2's schedule) One iteration = A.sub cheap sub-steps (each: rows above/below by lane shift, 3-wide dilation, promote the weak
        // cells that touch an edge cell: a chain advances one row per sub-step, diagonals included) followed by ONE
        // Kogge-Stone flood along the rows.  Promotion is monotone, so any schedule reaches the same fixpoint; a chain
        // that runs down the tile costs ~45 instructions per row instead of a whole flood (~130) per row.
        for (;;) {
            steps++;
            uint32_t any_c = 0;
            for (int k = 0; k < A.sub; k++) {
                row64 up = shfl_up_row(sr), dn = shfl_dn_row(sr);
                if (lane == 0) up = esr;
                if (lane == 63) dn = esr;
                const row64 d = dil3(up, up_l, up_r) | dil3(sr, hl32, hr32) | dil3(dn, dn_l, dn_r);
                // weak cells touching an edge cell that are not edges yet
                const row64 cand = row64{wr.lo & d.lo & ~sr.lo, wr.hi & d.hi & ~sr.hi};
                sr = sr | cand;
                any_c |= cand.lo | cand.hi;
            }
            if (!__any(any_c != 0u)) break; // none anywhere in the tile during a whole iteration => fixpoint
            sr = flood_row(sr, F);
        }
    }
#endif
    const u64 s = to_u64(sr);
    const u64 promoted = s & ~s0;
    if (A.stats && lane == 0) atomicAdd(&A.res[f].hyst_steps, steps); // diagnostic, VQA_OPT_HYST_STATS only
    if (!__any(promoted != 0)) return; // nothing changed: nothing to store, nobody to wake
    if (promoted) A.strong[idx] = s;   // (the edge pixels are counted once, from the final plane: k_canny_count)
    // Which neighbours must be re-visited: those holding a candidate pixel 8-adjacent to a promoted one.
    // (Candidate bits may be stale by one promotion: stale edge bits only ever make this test MORE
    // inclusive, never less, because edges only grow.)
    const u64 pb0 = promoted & 1ull, pb63 = promoted >> 63;
    // (DPP moves read their source lanes under the CURRENT exec mask: shift first, with every lane active, select after)
    const uint32_t p0 = (uint32_t)pb0, p63 = (uint32_t)pb63;
    const uint32_t p0u = lane_up32(p0), p0d = lane_dn32(p0), p63u = lane_up32(p63), p63d = lane_dn32(p63);
    const uint32_t m_up = lane > 0 ? ~0u : 0u, m_dn = lane < 63 ? ~0u : 0u;
    const u64 n0 = pb0 | (u64)(p0u & m_up) | (u64)(p0d & m_dn);
    const u64 n63 = pb63 | (u64)(p63u & m_up) | (u64)(p63d & m_dn);
    const bool L = __any((cl & n0) != 0), R = __any((cr & n63) != 0);
    const u64 pd = promoted | (promoted << 1) | (promoted >> 1); // promoted row, dilated along x
    const bool Uf = __any(lane == 0 && (ecs & pd) != 0), Df = __any(lane == 63 && (ecs & pd) != 0);
    const bool ULf = __any(lane == 0 && (ecl & pb0) != 0), URf = __any(lane == 0 && (ecr & pb63) != 0);
    const bool DLf = __any(lane == 63 && (ecl & pb0) != 0), DRf = __any(lane == 63 && (ecr & pb63) != 0);
    // Wake the neighbours: lanes 0..7 take one neighbour each (their dedup atomics overlap), then ONE append-counter
    // atomic per tile reserves the list slots.  The per-frame counter is a single address: at 2160p (64 frames, 2040
    // tiles each) one atomicAdd per enqueued neighbour serialised in L2 and made round 0 twice as slow as at 1080p for
    // the same tile count and the same 16-17 relaxation steps per tile.
    const unsigned nbmask = (ULf ? 1u : 0u) | (Uf ? 2u : 0u) | (URf ? 4u : 0u) | (L ? 8u : 0u) | (R ? 16u : 0u) |
                            (DLf ? 32u : 0u) | (Df ? 64u : 0u) | (DRf ? 128u : 0u);
    if (nbmask) { // wave-uniform
        const int b = lane & 7;
        const int ty2 = ty + (int)((0xA940u >> (2 * b)) & 3u) - 1, tx2 = tx + (int)((0x9224u >> (2 * b)) & 3u) - 1;
        const bool want = lane < 8 && ((nbmask >> b) & 1u) && ty2 >= 0 && ty2 < A.tiles_y && tx2 >= 0 && tx2 < A.ww;
        const unsigned t2 = (unsigned)((f * A.tiles_y + ty2) * A.ww + tx2);
        const bool got = want && atomicExch(&A.queued[t2], 1u) == 0u;
        const u64 gm = __ballot(got);
        if (gm) {
            unsigned base = 0;
            const unsigned seg = tile % HSEG;
            if (lane == 0) base = atomicAdd(&A.out_count[(unsigned)f * HSEG + seg], (unsigned)__popcll(gm));
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            if (got) A.out_list[hyst_list_index((unsigned)f, seg, (unsigned)(A.tiles_y * A.ww), base + (unsigned)__popcll(gm & ((1ull << lane) - 1ull)))] = t2;
        }
    }
}

// round 0: every tile.  grid = ceil(n_tiles / 4), block = 256 (4 independent waves)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_canny_hyst_all(hyst_args A, unsigned n_tiles)
{
    // XCD-aware: XCD c (= workgroup id % 8) walks the contiguous tile range [c * per, (c + 1) * per) * 4, so a tile and
    // the neighbours whose rows it reads as its halo sit behind one L2
    const unsigned per = (gridDim.x + 7u) >> 3;
    const unsigned wg = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    const unsigned tile = wg * 4 + wave_id();
    if ((blockIdx.x >> 3) < per && tile < n_tiles) relax_tile(A, tile);
}

// later rounds: the tiles some neighbour enqueued in the previous round
__global__ __launch_bounds__(256) void k_canny_hyst_list(hyst_args A, const unsigned *__restrict__ in_list,
                                                         const unsigned *__restrict__ in_count,
                                                         unsigned *__restrict__ in_queued,
                                                         unsigned *__restrict__ zero_count)
{
    const unsigned f = blockIdx.y, tpf = (unsigned)(A.tiles_y * A.ww);
    // the counters the NEXT round appends to (nobody reads or writes them during this round: vqa_capi.hip)
    if (blockIdx.x == 0 && threadIdx.x < HSEG) zero_count[f * HSEG + threadIdx.x] = 0;
    unsigned cnt[HSEG], n = 0;
#pragma unroll
    for (int k = 0; k < HSEG; k++) { cnt[k] = in_count[f * HSEG + k]; n += cnt[k]; } // wave-uniform: scalar loads
    for (unsigned g = blockIdx.x * 4 + wave_id(); g < n; g += gridDim.x * 4) {
        unsigned seg, idx;
        hyst_locate(cnt, g, seg, idx);
        const unsigned tile = in_list[hyst_list_index(f, seg, tpf, idx)];
        if (lane_id() == 0) in_queued[tile] = 0; // this list is consumed; its flags are reused two rounds on
        relax_tile(A, tile);
    }
}

// Tail of the fixpoint without the host: ONE workgroup per frame keeps relaxing that frame's own work
// list, round after round, until a round enqueues nothing.  Rounds are separated by an agent-scope
// fence + workgroup barrier: the promotions a round stored are released to L2 and this CU's L1 is
// invalidated before the next round loads them (every producer and consumer of a frame's tiles is in
// this one workgroup, so nothing outside it needs to see the flag traffic).  The loop is bounded.
//
// rescue = 0: the tail proper.  It stops at max_rounds (CANNY_HYST_MAX_ROUNDS; a drain bound, far above what frames need)
//   and then SAYS so: res[f].hyst_overflow = 1 - the edge plane is a subset of the fixpoint, the count a lower bound.
// rescue = 1: launched right after the tail, same stream.  A frame whose flag is clear leaves at once (one load per
//   workgroup).  A flagged frame is finished here: the aborted run's lists are forgotten, round 0 relaxes EVERY tile of
//   the frame (relaxation is monotone and idempotent, so starting over from the current planes is exact), then the list
//   rounds run to the fixpoint under the bound the algorithm itself gives: a list is non-empty only because the round
//   before it promoted a weak pixel, and a pixel is promoted once, so there are at most edge_weak + 1 non-empty rounds.
//   On success the flag becomes 2 ("completed by the rescue pass": edge_count is exact).  A flag still 1 at
//   vqa_complexity_wait means the bound of the proof was hit - a defect, not a frame - and the wait fails
//   (VQA_ERR_INCOMPLETE): a lower bound is never returned as a count.
__device__ __forceinline__ void hyst_round_handoff()
{
    // (guide G16) every storing wave drains its stores, the workgroup meets, ONE lane releases to L2 and then
    // invalidates this CU's L1, the workgroup meets again
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void k_canny_hyst_tail(hyst_args A, unsigned *__restrict__ list0,
                                                         unsigned *__restrict__ cnt0, unsigned *__restrict__ q0,
                                                         unsigned *__restrict__ list1, unsigned *__restrict__ cnt1,
                                                         unsigned *__restrict__ q1, int first_in, int max_rounds, int rescue)
{
    __shared__ unsigned s_cnt[HSEG];
    __shared__ unsigned s_flag, s_weak;
    const unsigned f = blockIdx.x, tpf = (unsigned)(A.tiles_y * A.ww);
    unsigned *lists[2] = {list0, list1}, *cnts[2] = {cnt0, cnt1}, *qs[2] = {q0, q1};
    int in = first_in;
    if (rescue) {
        if (threadIdx.x == 0) { s_flag = A.res[f].hyst_overflow; s_weak = A.res[f].edge_weak; }
        __syncthreads();
        if (s_flag != 1u) return; // (workgroup-uniform) the tail reached this frame's fixpoint: nothing to do
        // forget the aborted run: both dedup flag sets of the frame's tiles and both append counters
        for (unsigned t = threadIdx.x; t < tpf; t += blockDim.x) { q0[f * tpf + t] = 0; q1[f * tpf + t] = 0; }
        if (threadIdx.x < HSEG) { cnt0[f * HSEG + threadIdx.x] = 0; cnt1[f * HSEG + threadIdx.x] = 0; }
        hyst_round_handoff();
        // round 0: every tile of the frame, wakes into list 1
        A.queued = qs[1];
        A.out_list = lists[1];
        A.out_count = cnts[1];
        for (unsigned g = wave_id(); g < tpf; g += blockDim.x / 64) relax_tile(A, f * tpf + g);
        hyst_round_handoff();
        in = 1;
        if (max_rounds <= 0) { // the proof's bound (see above), clamped to an int
            const unsigned wk = s_weak;
            max_rounds = wk > 0x7ffffff0u ? 0x7ffffff2 : (int)wk + 2;
        }
    }
    bool reached_fixpoint = false;
    for (int round = 0; round < max_rounds; round++) {
        if (threadIdx.x < HSEG) {
            s_cnt[threadIdx.x] = cnts[in][f * HSEG + threadIdx.x];
            cnts[in ^ 1][f * HSEG + threadIdx.x] = 0; // the lists this round builds ...
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // ... acknowledged by L2 before any wave's atomicAdd on them
        }
        __syncthreads();
        unsigned cnt[HSEG], n = 0;
#pragma unroll
        for (int k = 0; k < HSEG; k++) { cnt[k] = s_cnt[k]; n += cnt[k]; }
        if (n == 0) { reached_fixpoint = true; break; }
        A.queued = qs[in ^ 1];
        A.out_list = lists[in ^ 1];
        A.out_count = cnts[in ^ 1];
        for (unsigned g = wave_id(); g < n; g += blockDim.x / 64) {
            unsigned seg, idx;
            hyst_locate(cnt, g, seg, idx);
            const unsigned tile = lists[in][hyst_list_index(f, seg, tpf, idx)];
            if (lane_id() == 0) qs[in][tile] = 0;
            relax_tile(A, tile);
        }
        hyst_round_handoff(); // hand-off to the next round
        in ^= 1;
    }
    if (rescue) {
        // (a bound that ends the loop exactly after the last non-empty round leaves an empty list behind: that IS the fixpoint)
        if (!reached_fixpoint) {
            __syncthreads(); // (s_cnt is rewritten: every wave has left the loop's reads)
            if (threadIdx.x < HSEG) s_cnt[threadIdx.x] = cnts[in][f * HSEG + threadIdx.x];
            __syncthreads();
            unsigned n = 0;
#pragma unroll
            for (int k = 0; k < HSEG; k++) n += s_cnt[k];
            reached_fixpoint = n == 0;
        }
        if (reached_fixpoint && threadIdx.x == 0) A.res[f].hyst_overflow = 2u; // completed here: the count is exact
        return;
    }
    // the bound exists so the grid always drains; hitting it leaves an UNDER-count, which the record must say
    if (!reached_fixpoint && threadIdx.x < HSEG && cnts[in][f * HSEG + threadIdx.x] != 0) A.res[f].hyst_overflow = 1u;
}

// np.sum(edges > 0) (complexity_metrics.py:504): the set bits of the final edge plane.  One pass over P/8 bytes per
// frame replaces a popcount + wave reduction + atomic in every tile visit of the fixpoint.
// grid = (ceil(tiles per frame / 32), n), block = 256: a wave walks 8 tiles, lane r <-> row r.
__global__ __launch_bounds__(256) void k_canny_count(const unsigned long long *__restrict__ strong, int h, int ww,
                                                     int tiles_y, vqa_frame_metrics *__restrict__ res)
{
    const int f = blockIdx.y, lane = lane_id();
    const int tpf = tiles_y * ww;
    unsigned cnt = 0;
#pragma unroll 4
    for (int i = 0; i < 8; i++) {
        const int t = (blockIdx.x * 4 + wave_id()) * 8 + i;
        if (t >= tpf) break;
        const int ty = t / ww;
        if (ty * 64 + lane < h) cnt += (unsigned)__popcll(strong[((int64_t)f * tpf + t) * 64 + lane]);
    }
    cnt = wave_sum(cnt);
    if (lane == 0 && cnt) atomicAdd(&res[f].edge_count, cnt);
}

void launch_canny_nms(hipStream_t st, const uint8_t *gray, int pitch, int64_t plane_stride, int n, int h, int w,
                      int low, int high, unsigned long long *strong, unsigned long long *weak,
                      vqa_frame_metrics *res)
{
    if (n <= 0) return;
    // w < 4: the dword window of k_canny_nms3 needs 4 columns; such slivers take the LDS-tile kernel.
    // Lab build (VQA_NMS_VARIANT): 3 = lane-per-column kernel (shipped), 2 = round 2's register-rolling kernel, 1 = LDS tiles
    static const int variant = ab_knob("VQA_NMS_VARIANT", 3);
    if (variant == 3 && w >= 4) {
        const int sx = (w + 255) / 256, sy = (h + 63) / 64;
        const long long ns = (long long)sx * sy * n; // (n <= 32768 per launch: fits an int)
        hipLaunchKernelGGL(k_canny_nms3, dim3((unsigned)(8 * ((ns + 7) / 8))), dim3(64), 0, st, gray, pitch,
                           plane_stride, h, w, low, high, strong, weak, (w + 63) / 64, res, sx, sy, (int)ns);
#ifdef VQA_AB_VARIANTS
    } else if (variant == 2) {
        hipLaunchKernelGGL(k_canny_nms2, dim3((w + 255) / 256, (h + 63) / 64, n), dim3(64), 0, st, gray, pitch,
                           plane_stride, h, w, low, high, strong, weak, (w + 63) / 64, res);
#endif
    } else {
        const canny_geom g = canny_tiles(h, w);
        hipLaunchKernelGGL(k_canny_nms, dim3(g.tiles_x, g.tiles_y, n), dim3(256), 0, st, gray, pitch, plane_stride, h, w,
                           low, high, strong, weak, (w + 63) / 64, res);
    }
}

static hyst_args make_hyst_args(unsigned long long *strong, const unsigned long long *weak, int h, int w,
                                unsigned *queued, unsigned *out_list, unsigned *out_count, vqa_frame_metrics *res, int stats)
{
    hyst_args A;
    A.strong = strong; A.weak = weak; A.h = h; A.ww = (w + 63) / 64; A.tiles_y = (h + 63) / 64;
    A.queued = queued; A.out_list = out_list; A.out_count = out_count; A.res = res;
    // 0 = dilation + carry flood (shipped); lab build, VQA_HYST_SUB=1..8: round 2's shift/and/or ladder with that many sub-steps
    static const int sub = [] { const int e = ab_knob("VQA_HYST_SUB", 0); return (e >= 1 && e <= 8) ? e : 0; }();
    A.sub = sub;
    A.stats = stats;
    return A;
}

unsigned canny_hyst_tiles(int n, int h, int w) { return (unsigned)n * ((h + 63) / 64) * ((w + 63) / 64); }

void launch_canny_hyst_all(hipStream_t st, unsigned long long *strong, const unsigned long long *weak, int n, int h,
                           int w, unsigned *queued, unsigned *out_list, unsigned *out_count, vqa_frame_metrics *res,
                           int stats)
{
    if (n <= 0) return;
    const unsigned nt = canny_hyst_tiles(n, h, w);
    hipLaunchKernelGGL(k_canny_hyst_all, dim3(8 * (((nt + 3) / 4 + 7) / 8)), dim3(256), 0, st,
                       make_hyst_args(strong, weak, h, w, queued, out_list, out_count, res, stats), nt);
}

void launch_canny_hyst_list(hipStream_t st, unsigned long long *strong, const unsigned long long *weak, int n, int h,
                            int w, unsigned *in_queued, const unsigned *in_list, const unsigned *in_count,
                            unsigned *out_queued, unsigned *out_list, unsigned *out_count, unsigned *zero_count,
                            vqa_frame_metrics *res, int stats)
{
    if (n <= 0) return;
    // workgroups per frame: ~64 tiles of the frame per 4-wave workgroup (1080p: 8, 2160p: 32)
    const canny_geom g = canny_tiles(h, w);
    int gx = (g.tiles_x * g.tiles_y + 63) / 64;
    gx = gx < 8 ? 8 : (gx > 32 ? 32 : gx);
    hipLaunchKernelGGL(k_canny_hyst_list, dim3(gx, n), dim3(256), 0, st,
                       make_hyst_args(strong, weak, h, w, out_queued, out_list, out_count, res, stats), in_list, in_count,
                       in_queued, zero_count);
}

// rounds 2.. to convergence, one workgroup per frame, no host involvement.  lists/counts/queued: the two
// per-frame work lists; first_in = index of the list the first tail round consumes.  Two launches: the tail, then the
// rescue pass (k_canny_hyst_tail, rescue = 1), which costs one load per frame unless the tail stopped at its bound.
// max_rounds / rescue_max_rounds: 0 = the shipped bounds (lab seams pass small ones to force either outcome).
void launch_canny_hyst_tail(hipStream_t st, unsigned long long *strong, const unsigned long long *weak, int n, int h,
                            int w, unsigned *list0, unsigned *cnt0, unsigned *q0, unsigned *list1, unsigned *cnt1,
                            unsigned *q1, int first_in, vqa_frame_metrics *res, int stats, int max_rounds, int rescue_max_rounds)
{
    if (n <= 0) return;
    if (max_rounds <= 0) max_rounds = CANNY_HYST_MAX_ROUNDS;
    const hyst_args A = make_hyst_args(strong, weak, h, w, nullptr, nullptr, nullptr, res, stats);
    hipLaunchKernelGGL(k_canny_hyst_tail, dim3(n), dim3(1024), 0, st, A, list0, cnt0, q0, list1, cnt1, q1, first_in, max_rounds, 0);
    hipLaunchKernelGGL(k_canny_hyst_tail, dim3(n), dim3(1024), 0, st, A, list0, cnt0, q0, list1, cnt1, q1, first_in,
                       rescue_max_rounds > 0 ? rescue_max_rounds : 0, 1);
}

void launch_canny_finish(hipStream_t st, const unsigned long long *strong, int n, int h, int w, vqa_frame_metrics *res)
{
    if (n <= 0) return;
    const int ww = (w + 63) / 64, tiles_y = (h + 63) / 64;
    hipLaunchKernelGGL(k_canny_count, dim3((tiles_y * ww + 31) / 32, n), dim3(256), 0, st, strong, h, ww, tiles_y, res);
}

} // namespace vqa
