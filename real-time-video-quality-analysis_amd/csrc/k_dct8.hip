// k_dct8.hip — 8x8 block DCT energy and temporal 8x8-DCT difference for gfx950.
//
// Reference functions replaced (complexity_metrics.py):
//   :363-364  np.sum(cv2.dct(np.float32(gray)) ** 2)            -> dct_energy
//   :574-579  np.sum(np.abs(cv2.dct(prev) - cv2.dct(curr)))     -> temporal_dct_l1
// in the 8x8-BLOCK form BASELINE.json's north_star decrees.  The energy equals
// the reference's full-frame value (Parseval, both transforms orthonormal,
// partial blocks zero-padded); the block L1 is a different metric from the
// full-frame L1 and has the CPU restatement as its oracle.
// By linearity dct(prev) - dct(curr) = dct(prev - curr): one transform of the
// exact integer difference instead of two transforms.
//
// Mapping: ONE 8x8 BLOCK PER LANE.  A wave covers 64 consecutive blocks of the
// frame's block raster, so each of the 8 row loads is 64 lanes x 8 B = 512
// contiguous bytes; both 1-D passes run entirely in the lane's registers (64
// floats): no LDS, no cross-lane traffic.  Roofline: HBM, 2P bytes per frame
// (current + previous gray plane), ~22 VALU ops per pixel.
#include "vqa_dev.hpp"
#include "vqa_kernels.hpp"
#include "vqa_math.hpp"

#include <cstdlib>

namespace vqa {

#ifdef VQA_AB_VARIANTS // round 1's kernels (two transforms per pair), lab build only: VQA_DCT_VARIANT=1..4
__device__ __forceinline__ void load_block_u8(const uint8_t *__restrict__ plane, int pitch, int h, int w, int by,
                                              int bx, uint32_t lo[8], uint32_t hi[8])
{
    const int x0 = bx * 8;
    // columns >= w are zero-padded (keeps Parseval exact on ragged widths)
    const int nvalid = min(8, w - x0);
    const uint64_t mask = nvalid >= 8 ? ~0ull : ((1ull << (8 * nvalid)) - 1ull);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int y = by * 8 + r;
        uint64_t v = 0;
        if (y < h) v = *(const uint64_t *)(plane + (int64_t)y * pitch + x0) & mask; // pitch % 8 == 0
        lo[r] = (uint32_t)v;
        hi[r] = (uint32_t)(v >> 32);
    }
}

__device__ __forceinline__ float ub(uint32_t v, int k) { return (float)((v >> (8 * k)) & 0xffu); }

// grid = (blocks_per_frame, n_frames), block = 256 (4 independent waves).
// planes: slot 0 = frame preceding the batch, slot i+1 = batch frame i.
// partials[(f * gridDim.x + blockIdx.x) * 2 + {0,1}] = {energy, temporal L1} of this block's share.
template <bool ENERGY, bool TEMPORAL, int MINW>
__global__ __launch_bounds__(256, MINW) void k_dct8(const uint8_t *__restrict__ planes, int pitch, int64_t plane_stride,
                                              int h, int w, int first_has_prev, double *__restrict__ partials)
{
    __shared__ double red[4];
    const int f = blockIdx.y;
    const uint8_t *curr = planes + (int64_t)(f + 1) * plane_stride;
    const uint8_t *prev = planes + (int64_t)f * plane_stride;
    const bool temporal = TEMPORAL && (f > 0 || first_has_prev);
    const int nbx = (w + 7) >> 3, nby = (h + 7) >> 3;
    const int nblk = nbx * nby;
    double acc_e = 0, acc_t = 0;
    for (int b = blockIdx.x * 256 + threadIdx.x; b < nblk; b += gridDim.x * 256) {
        const int by = b / nbx, bx = b - by * nbx;
        uint32_t clo[8], chi[8];
        load_block_u8(curr, pitch, h, w, by, bx, clo, chi);
        float v[64];
        if (ENERGY) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
#pragma unroll
                for (int k = 0; k < 4; k++) { v[8 * r + k] = ub(clo[r], k); v[8 * r + 4 + k] = ub(chi[r], k); }
            }
            dct8x8(v);
            float e = 0;
#pragma unroll
            for (int i = 0; i < 64; i++) e = fmaf(v[i], v[i], e);
            acc_e += (double)e;
        }
        // the two transforms reuse the same 64 registers: do not let the scheduler interleave them
        if (MINW > 1) __builtin_amdgcn_sched_barrier(0);
        if (temporal) {
            uint32_t plo[8], phi[8];
            load_block_u8(prev, pitch, h, w, by, bx, plo, phi);
#pragma unroll
            for (int r = 0; r < 8; r++) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    v[8 * r + k] = ub(plo[r], k) - ub(clo[r], k);
                    v[8 * r + 4 + k] = ub(phi[r], k) - ub(chi[r], k);
                }
            }
            dct8x8(v);
            float t = 0;
#pragma unroll
            for (int i = 0; i < 64; i++) t += fabsf(v[i]);
            acc_t += (double)t;
        }
    }
    const double be = block_sum(acc_e, red);
    const double bt = block_sum(acc_t, red);
    if (threadIdx.x == 0) {
        double *o = partials + ((int64_t)f * gridDim.x + blockIdx.x) * 2;
        o[0] = be;
        o[1] = bt;
    }
}

// Paired variant: the energy transform (of curr) and the temporal transform (of prev - curr) run as the
// two halves of ONE float2 transform, so every butterfly is a packed instruction.  128 VGPRs of
// accumulator + inputs => 2 waves/SIMD; the next block's bytes are fetched before this block's arithmetic.
__global__ __launch_bounds__(256, 2) void k_dct8_pair(const uint8_t *__restrict__ planes, int pitch,
                                                      int64_t plane_stride, int h, int w, int first_has_prev,
                                                      double *__restrict__ partials)
{
    __shared__ double red[4];
    const int f = blockIdx.y;
    const uint8_t *curr = planes + (int64_t)(f + 1) * plane_stride;
    const uint8_t *prev = planes + (int64_t)f * plane_stride;
    const bool temporal = (f > 0 || first_has_prev);
    const int nbx = (w + 7) >> 3, nby = (h + 7) >> 3;
    const int nblk = nbx * nby;
    double acc_e = 0, acc_t = 0;
    int b = blockIdx.x * 256 + threadIdx.x;
    uint32_t clo[8], chi[8], plo[8], phi[8];
    if (b < nblk) {
        load_block_u8(curr, pitch, h, w, b / nbx, b % nbx, clo, chi);
        load_block_u8(temporal ? prev : curr, pitch, h, w, b / nbx, b % nbx, plo, phi);
    }
    while (b < nblk) {
        vqa_f2 v[64];
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int c0 = (clo[r] >> (8 * k)) & 0xff, p0 = (plo[r] >> (8 * k)) & 0xff;
                const int c1 = (chi[r] >> (8 * k)) & 0xff, p1 = (phi[r] >> (8 * k)) & 0xff;
                v[8 * r + k] = vqa_f2{(float)c0, (float)(p0 - c0)};
                v[8 * r + 4 + k] = vqa_f2{(float)c1, (float)(p1 - c1)};
            }
        }
        // prefetch the next block of this lane while the transform runs
        const int bn = b + gridDim.x * 256;
        if (bn < nblk) {
            load_block_u8(curr, pitch, h, w, bn / nbx, bn % nbx, clo, chi);
            load_block_u8(temporal ? prev : curr, pitch, h, w, bn / nbx, bn % nbx, plo, phi);
        }
        dct8x8_x2(v);
        float e = 0, t = 0;
#pragma unroll
        for (int i = 0; i < 64; i++) { e = fmaf(v[i].x, v[i].x, e); t += fabsf(v[i].y); }
        acc_e += (double)e;
        acc_t += (double)t;
        b = bn;
    }
    const double be = block_sum(acc_e, red);
    const double bt = block_sum(acc_t, red);
    if (threadIdx.x == 0) {
        double *o = partials + ((int64_t)f * gridDim.x + blockIdx.x) * 2;
        o[0] = be;
        o[1] = temporal ? bt : 0.0;
    }
}

#endif // VQA_AB_VARIANTS

// ---------------------------------------------------------------------------
// Frame-marching form (the shipped kernel).  The reference's temporal metric is dct(prev) - dct(curr)
// (complexity_metrics.py:574-578): consecutive pairs share a transform, DCT(curr_t) IS DCT(prev_{t+1}).
// So a lane keeps ONE 8x8 block position and marches through a chunk of consecutive frames, carrying
// the 64 coefficients of the previous frame in registers: one 2-D transform per frame instead of two,
// every gray plane read once (+ one halo frame per chunk).  Per frame a lane does
//     64 v_cvt_f32_ubyte  +  16 x 1-D DCT (plain fp32 fma/mul/add)  +  64 fma (energy)  +  128 sub/|add| (L1).
// A workgroup is ONE wave (64 block positions); the two sums of a frame are reduced over the wave in a fixed
// DPP order and written as one float2 partial per (frame, wave) - k_dct_finalize2 adds them in double, in
// index order, so results are bit-reproducible.
//   grid = (waves per frame / 4, chunks), block = 256 (four independent waves);  planes slot 0 = the frame preceding the batch
// ---------------------------------------------------------------------------
typedef const __attribute__((address_space(1))) uint64_t *gptr_u64;

template <int ctrl, int row_mask>
__device__ __forceinline__ float dpp_add(float v)
{
    // v + (v moved by the DPP pattern; lanes the pattern does not write contribute 0)
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, row_mask, 0xf, false);
    return v + __int_as_float(moved);
}
// sum over the 64 lanes in a fixed order; the total lands in lane 63
__device__ __forceinline__ float wave_sum_dpp(float v)
{
    v = dpp_add<0xB1, 0xf>(v);  // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);  // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v); // row_half_mirror
    v = dpp_add<0x140, 0xf>(v); // row_mirror: every lane of a 16-lane row holds the row's sum
    v = dpp_add<0x142, 0xa>(v); // row_bcast:15 -> rows 1 and 3
    v = dpp_add<0x143, 0xc>(v); // row_bcast:31 -> rows 2 and 3
    return v;
}

struct dct_bytes {
    uint32_t lo[8], hi[8];
};

// the 8 rows of one 8x8 block of `plane`: rows are 8-byte loads at a wave-uniform row base + a per-lane offset
// 8-point orthonormal DCT-II for the marching kernel: the butterflies of vqa_math.hpp's dct8 with every constant an
// instruction LITERAL.  Measured (profiles/round2_valu_calib.json): v_fma/v_mul/v_add_f32 issue at ~2.4-2.9 cycles per
// wave with VGPR, literal or inline-constant sources but at 4.2-4.5 with an SGPR source, and the compiler keeps the
// cosine constants of its v_fma_f32 (VOP3: no literals on gfx9) in SGPRs.  v_mul_f32 / v_fmac_f32 are VOP2 and take one.
#define VQA_FBITS(x) __builtin_bit_cast(int, (float)(x))
template <int KBITS>
__device__ __forceinline__ float lit_mul(float x)
{
    float d;
    asm("v_mul_f32 %0, %1, %2" : "=v"(d) : "i"(KBITS), "v"(x));
    return d;
}
template <int KBITS>
__device__ __forceinline__ float lit_fmac(float acc, float x) // acc + K * x
{
    asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "i"(KBITS), "v"(x));
    return acc;
}
template <int STRIDE>
__device__ __forceinline__ void dct8_lit(float *v)
{
    constexpr float r8 = 0.35355339059327373f, c1 = 0.49039264020161522f, c2 = 0.46193976625564337f,
                    c3 = 0.41573480615127262f, c5 = 0.27778511650980114f, c6 = 0.19134171618254492f,
                    c7 = 0.09754516100806417f;
    const float x0 = v[0 * STRIDE], x1 = v[1 * STRIDE], x2 = v[2 * STRIDE], x3 = v[3 * STRIDE];
    const float x4 = v[4 * STRIDE], x5 = v[5 * STRIDE], x6 = v[6 * STRIDE], x7 = v[7 * STRIDE];
    const float s0 = x0 + x7, s1 = x1 + x6, s2 = x2 + x5, s3 = x3 + x4;
    const float d0 = x0 - x7, d1 = x1 - x6, d2 = x2 - x5, d3 = x3 - x4;
    const float t0 = s0 + s3, t1 = s1 + s2, t2 = s0 - s3, t3 = s1 - s2;
    v[0 * STRIDE] = lit_mul<VQA_FBITS(r8)>(t0 + t1);
    v[4 * STRIDE] = lit_mul<VQA_FBITS(r8)>(t0 - t1);
    v[2 * STRIDE] = lit_fmac<VQA_FBITS(c2)>(lit_mul<VQA_FBITS(c6)>(t3), t2);
    v[6 * STRIDE] = lit_fmac<VQA_FBITS(c6)>(lit_mul<VQA_FBITS(-c2)>(t3), t2);
    v[1 * STRIDE] = lit_fmac<VQA_FBITS(c1)>(lit_fmac<VQA_FBITS(c3)>(lit_fmac<VQA_FBITS(c5)>(lit_mul<VQA_FBITS(c7)>(d3), d2), d1), d0);
    v[3 * STRIDE] = lit_fmac<VQA_FBITS(c3)>(lit_fmac<VQA_FBITS(-c7)>(lit_fmac<VQA_FBITS(-c1)>(lit_mul<VQA_FBITS(-c5)>(d3), d2), d1), d0);
    v[5 * STRIDE] = lit_fmac<VQA_FBITS(c5)>(lit_fmac<VQA_FBITS(-c1)>(lit_fmac<VQA_FBITS(c7)>(lit_mul<VQA_FBITS(c3)>(d3), d2), d1), d0);
    v[7 * STRIDE] = lit_fmac<VQA_FBITS(c7)>(lit_fmac<VQA_FBITS(-c5)>(lit_fmac<VQA_FBITS(c3)>(lit_mul<VQA_FBITS(-c1)>(d3), d2), d1), d0);
}
// 2-D transform with the instruction scheduler fenced every two 1-D transforms: left alone it interleaves all
// eight rows for ILP and needs ~60 temporaries on top of the 64 coefficients (3 waves/SIMD supply the parallelism)
__device__ __forceinline__ void dct8x8_fenced(float *v)
{
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
        dct8_lit<1>(v + 8 * r);
        dct8_lit<1>(v + 8 * r + 8);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int c = 0; c < 8; c += 2) {
        dct8_lit<8>(v + c);
        dct8_lit<8>(v + c + 1);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <bool FULL>
__device__ __forceinline__ void march_load(dct_bytes &q, const uint8_t *plane, int pitch, uint32_t voff, int by8, int bx8,
                                           int rows_valid, uint64_t colmask, int h)
{
#pragma unroll
    for (int r = 0; r < 8; r++) {
        uint64_t v;
        if (FULL) {
            v = *(gptr_u64)(uniform_ptr(plane + (int64_t)r * pitch) + voff);
        } else { // ragged bottom / right edge: clamp the row, then zero what lies outside (Parseval stays exact)
            const int y = min(by8 + r, h - 1);
            v = *(const uint64_t *)(plane + (int64_t)y * pitch + bx8);
            v = (r < rows_valid) ? (v & colmask) : 0ull;
        }
        q.lo[r] = (uint32_t)v;
        q.hi[r] = (uint32_t)(v >> 32);
    }
}

// byte k of a dword as a float: ONE v_cvt_f32_ubyteN (asm so that the compiler cannot turn the first butterfly stage
// into SDWA byte adds + shifts + v_cvt_f32_i32, which costs 232 instructions per block against 64 + 64 here)
template <int K>
__device__ __forceinline__ float ubf(uint32_t v)
{
    float f;
    if (K == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(v));
    else if (K == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(v));
    else if (K == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(v));
    else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(f) : "v"(v));
    return f;
}

// acc + |d| as ONE full-rate v_add_f32 with the abs source modifier (left to the compiler the sums are SLP-packed into
// v_pk_add_f32, which has no abs modifier and needs a v_and_b32 per value on top of the half-rate packed add)
__device__ __forceinline__ float abs_acc(float acc, float d)
{
    asm("v_add_f32_e64 %0, |%1|, %0" : "+v"(acc) : "v"(d));
    return acc;
}

__device__ __forceinline__ void march_unpack(const dct_bytes &q, float *v)
{
#pragma unroll
    for (int r = 0; r < 8; r++) {
        v[8 * r + 0] = ubf<0>(q.lo[r]); v[8 * r + 1] = ubf<1>(q.lo[r]); v[8 * r + 2] = ubf<2>(q.lo[r]); v[8 * r + 3] = ubf<3>(q.lo[r]);
        v[8 * r + 4] = ubf<0>(q.hi[r]); v[8 * r + 5] = ubf<1>(q.hi[r]); v[8 * r + 6] = ubf<2>(q.hi[r]); v[8 * r + 7] = ubf<3>(q.hi[r]);
    }
}

// RAGGED = the plane has a partial last block row/column (w or h not a multiple of 8): per-lane clamped rows and
// masks cost address registers, so that variant is compiled for 2 waves/SIMD; 1080p/2160p take the other one.
template <bool ENERGY, bool TEMPORAL, bool RAGGED, bool LOAD_EARLY>
__global__ __launch_bounds__(256, RAGGED ? 2 : 3) void k_dct8_march(const uint8_t *__restrict__ planes, int pitch, int64_t plane_stride,
                                                      int h, int w, int n, int fch, int nw, int first_has_prev,
                                                      float2 *__restrict__ partials)
{
    // a workgroup is four INDEPENDENT waves (no barrier, no LDS): 256 threads only so that the dispatcher spreads
    // them one per SIMD (single-wave workgroups measured 2.4x slower: they do not spread evenly over the SIMDs)
    const int lane = threadIdx.x & 63;
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6); // wave index in the frame's block raster
    const int f0 = blockIdx.y * fch, f1 = min(n, f0 + fch);
    const int nbx = (w + 7) >> 3, nby = (h + 7) >> 3;
    const int nblk = nbx * nby;
    if (wv >= nw) return;
    int b = wv * 64 + lane;
    const bool valid = b < nblk;
    if (!valid) b = nblk - 1;
    const int by = b / nbx, bx = b - by * nbx;
    const int by8 = by * 8, bx8 = bx * 8;
    const int rows_valid = min(8, h - by8), cols_valid = min(8, w - bx8);
    const uint64_t colmask = cols_valid >= 8 ? ~0ull : ((1ull << (8 * cols_valid)) - 1ull);
    const uint32_t voff = (uint32_t)by8 * (uint32_t)pitch + (uint32_t)bx8;
    const float keep = valid ? 1.f : 0.f;

    float A[64], B[64];
    dct_bytes q;
    bool have_prev = false;
    auto load = [&](int slot) {
        const uint8_t *pl = planes + (int64_t)slot * plane_stride;
        march_load<!RAGGED>(q, pl, pitch, voff, by8, bx8, rows_valid, colmask, h);
    };
    // one frame: bytes in q -> CUR = DCT(frame f); sums against PRV = DCT(frame f-1); next frame's bytes fetched
    // while the transform runs
    // The next frame's bytes are requested only after the transform (the byte registers are dead during it: 64 carried +
    // 64 working coefficients leave no room for them at 3 waves/SIMD); the sums and the other waves cover the latency.
    auto step = [&](int f, float *CUR, const float *PRV) {
        march_unpack(q, CUR);
        __builtin_amdgcn_sched_barrier(0);
        dct8x8_fenced(CUR);
        if (LOAD_EARLY) load(min(f + 2, n)); // the chunk's last step re-reads a valid slot instead of branching
        float e0 = 0.f, e1 = 0.f, t0 = 0.f, t1 = 0.f;
        if (ENERGY) {
#pragma unroll
            for (int i = 0; i < 64; i += 2) { e0 = fmaf(CUR[i], CUR[i], e0); e1 = fmaf(CUR[i + 1], CUR[i + 1], e1); }
        }
        if (TEMPORAL && have_prev) {
#pragma unroll
            for (int i = 0; i < 64; i += 2) { t0 = abs_acc(t0, PRV[i] - CUR[i]); t1 = abs_acc(t1, PRV[i + 1] - CUR[i + 1]); }
        }
        if (!LOAD_EARLY) { __builtin_amdgcn_sched_barrier(0); load(min(f + 2, n)); }
        const float e = wave_sum_dpp((e0 + e1) * keep);
        const float t = wave_sum_dpp((t0 + t1) * keep);
        if (lane == 63) partials[(int64_t)f * nw + wv] = make_float2(e, t);
        __builtin_amdgcn_sched_barrier(0);
    };

    if (TEMPORAL && (f0 > 0 || first_has_prev)) { // halo: the transform of the frame before the chunk
        load(f0);
        march_unpack(q, B);
        __builtin_amdgcn_sched_barrier(0);
        dct8x8_fenced(B);
        have_prev = true;
    }
    load(f0 + 1);
    for (int f = f0; f < f1; f += 2) {
        step(f, A, B);
        have_prev = TEMPORAL;
        if (f + 1 < f1) step(f + 1, B, A);
    }
}

// one wave per frame: lane i adds partials i, i+64, ... in double, then a fixed-order shuffle tree (bit-reproducible)
__global__ __launch_bounds__(64) void k_dct_finalize2(const float2 *__restrict__ partials, int nw, int n,
                                                      vqa_frame_metrics *__restrict__ res, int write_energy,
                                                      int write_temporal, int first_has_prev)
{
    const int f = blockIdx.x;
    if (f >= n) return;
    double e = 0, t = 0;
    for (int i = threadIdx.x; i < nw; i += 64) { const float2 p = partials[(int64_t)f * nw + i]; e += (double)p.x; t += (double)p.y; }
    e = wave_sum(e);
    t = wave_sum(t);
    if (threadIdx.x == 0) {
        if (write_energy) res[f].dct_energy = e;
        if (write_temporal) res[f].temporal_dct_l1 = (f > 0 || first_has_prev) ? t : 0.0;
    }
}

#ifdef VQA_AB_VARIANTS
// Deterministic second stage: one thread per frame adds the per-block partials
// in a fixed order and stores them into the result records.
__global__ void k_dct_finalize(const double *__restrict__ partials, int pb, int n, vqa_frame_metrics *__restrict__ res,
                               int write_energy, int write_temporal, int first_has_prev)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    double e = 0, t = 0;
    for (int i = 0; i < pb; i++) { e += partials[((int64_t)f * pb + i) * 2]; t += partials[((int64_t)f * pb + i) * 2 + 1]; }
    if (write_energy) res[f].dct_energy = e;
    if (write_temporal) res[f].temporal_dct_l1 = (f > 0 || first_has_prev) ? t : 0.0;
}

static int dct8_blocks_legacy(int h, int w)
{
    const int nblk = ((w + 7) / 8) * ((h + 7) / 8);
    int pb = (nblk + 256 * 4 - 1) / (256 * 4); // ~4 blocks of pixels per lane
    return pb < 1 ? 1 : (pb > 64 ? 64 : pb);
}

#endif // VQA_AB_VARIANTS

// 16-byte partial slots per frame the DCT launch needs (sizes the scratch buffer): the marching kernel writes one
// float2 per wave of 64 block positions (the lab build's legacy kernels: two doubles per workgroup)
int dct8_blocks_per_frame(int h, int w)
{
    const int nblk = ((w + 7) / 8) * ((h + 7) / 8);
    const int nw = (nblk + 63) / 64;
#ifdef VQA_AB_VARIANTS
    const int legacy = dct8_blocks_legacy(h, w);
    return (nw + 1) / 2 > legacy ? (nw + 1) / 2 : legacy;
#else
    return (nw + 1) / 2;
#endif
}

#ifdef VQA_AB_VARIANTS
static int dct_variant()
{
    static const int v = [] { const int e = ab_knob("VQA_DCT_VARIANT", 0); return (e < 0 || e > 4) ? 0 : e; }();
    return v;
}

template <int MINW>
static void launch_dct8_v(hipStream_t st, dim3 grid, const uint8_t *planes, int pitch, int64_t plane_stride, int h, int w,
                          bool energy, bool temporal, bool first_has_prev, double *partials)
{
    dim3 block(256);
    if (energy && temporal)
        hipLaunchKernelGGL((k_dct8<true, true, MINW>), grid, block, 0, st, planes, pitch, plane_stride, h, w,
                           (int)first_has_prev, partials);
    else if (energy)
        hipLaunchKernelGGL((k_dct8<true, false, MINW>), grid, block, 0, st, planes, pitch, plane_stride, h, w,
                           (int)first_has_prev, partials);
    else
        hipLaunchKernelGGL((k_dct8<false, true, MINW>), grid, block, 0, st, planes, pitch, plane_stride, h, w,
                           (int)first_has_prev, partials);
}

#endif // VQA_AB_VARIANTS

// frames per chunk of the marching kernel: every chunk pays one halo transform (1/chunk of extra work); more, smaller
// chunks give the dispatcher finer grains at the tail of the grid.  Measured on 256 x 1080p and 64 x 2160p: 12..24
// frames per chunk are within 2 % of each other and 3-5 % better than one grid-filling round of 43; so: 16, shrunk
// until the grid has at least two waves per wave slot of the chip.
static int dct_march_chunk(int n, int nw, int slots)
{
    { static const int v = ab_knob("VQA_DCT_FCH", 0); if (v > 0) return v < n ? v : n; } // lab build: tuning knob (a constant 0 in the shipped build)
    int fch = n < 16 ? n : 16;
    while (fch > 1 && (long)((n + fch - 1) / fch) * nw < 2L * slots) fch = (fch + 1) / 2;
    return fch;
}

int dct8_wave_slots()
{
    int dev = 0, cus = 256, per_cu = 3;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_dct8_march<true, true, false, false>, 256, 0) != hipSuccess || per_cu < 1)
        per_cu = 3;
    (void)hipGetLastError();
    return cus * per_cu * 4; // waves
}

void launch_dct8(hipStream_t st, const uint8_t *planes, int pitch, int64_t plane_stride, int n, int h, int w,
                 bool energy, bool temporal, bool first_has_prev, double *partials, vqa_frame_metrics *res, int wave_slots)
{
    if (n <= 0 || (!energy && !temporal)) return;
#ifdef VQA_AB_VARIANTS
    if (dct_variant() == 0)
#endif
    { // frame-marching kernel: one transform per frame
        const int nblk = ((w + 7) / 8) * ((h + 7) / 8), nw = (nblk + 63) / 64;
        const int fch = dct_march_chunk(n, nw, wave_slots > 0 ? wave_slots : 3072);
        dim3 grid((nw + 3) / 4, (n + fch - 1) / fch);
        float2 *p2 = (float2 *)partials;
        const bool ragged = ((w | h) & 7) != 0;
#ifdef VQA_AB_VARIANTS
        static const bool early = ab_knob("VQA_DCT_LOAD_EARLY", 0) != 0;
#define LAUNCH_MARCH(E, T, R)                                                                                             \
    do {                                                                                                                  \
        if (early)                                                                                                        \
            hipLaunchKernelGGL((k_dct8_march<E, T, R, true>), grid, dim3(256), 0, st, planes, pitch, plane_stride, h, w, n,\
                               fch, nw, (int)first_has_prev, p2);                                                           \
        else                                                                                                              \
            hipLaunchKernelGGL((k_dct8_march<E, T, R, false>), grid, dim3(256), 0, st, planes, pitch, plane_stride, h, w, \
                               n, fch, nw, (int)first_has_prev, p2);                                                        \
    } while (0)
#else
#define LAUNCH_MARCH(E, T, R)                                                                                             \
    hipLaunchKernelGGL((k_dct8_march<E, T, R, false>), grid, dim3(256), 0, st, planes, pitch, plane_stride, h, w, n, fch,  \
                       nw, (int)first_has_prev, p2)
#endif
        if (energy && temporal) { if (ragged) LAUNCH_MARCH(true, true, true); else LAUNCH_MARCH(true, true, false); }
        else if (energy) { if (ragged) LAUNCH_MARCH(true, false, true); else LAUNCH_MARCH(true, false, false); }
        else { if (ragged) LAUNCH_MARCH(false, true, true); else LAUNCH_MARCH(false, true, false); }
#undef LAUNCH_MARCH
        hipLaunchKernelGGL(k_dct_finalize2, dim3(n), dim3(64), 0, st, p2, nw, n, res, (int)energy, (int)temporal,
                           (int)first_has_prev);
        return;
    }
#ifdef VQA_AB_VARIANTS
    // legacy kernels (VQA_DCT_VARIANT 1, 2, 4 = two transforms per pair at min 1 / 4 / 3 waves per SIMD; 3 = float2-paired)
    const int pb = dct8_blocks_legacy(h, w);
    for (int a = 0; a < n; a += 65535) { // frames ride in gridDim.y
        const int m = n - a < 65535 ? n - a : 65535;
        dim3 grid(pb, m);
        const uint8_t *pl = planes + (int64_t)a * plane_stride;
        double *pp = partials + (int64_t)a * pb * 2;
        const bool fhp = a > 0 || first_has_prev;
        if (dct_variant() == 3 && energy && temporal) {
            hipLaunchKernelGGL(k_dct8_pair, grid, dim3(256), 0, st, pl, pitch, plane_stride, h, w, (int)fhp, pp);
            hipLaunchKernelGGL(k_dct_finalize, dim3((m + 63) / 64), dim3(64), 0, st, pp, pb, m, res + a, 1, 1, (int)fhp);
            continue;
        }
        switch (dct_variant()) {
        case 1: launch_dct8_v<1>(st, grid, pl, pitch, plane_stride, h, w, energy, temporal, fhp, pp); break;
        case 2: launch_dct8_v<4>(st, grid, pl, pitch, plane_stride, h, w, energy, temporal, fhp, pp); break;
        default: launch_dct8_v<3>(st, grid, pl, pitch, plane_stride, h, w, energy, temporal, fhp, pp); break;
        }
        hipLaunchKernelGGL(k_dct_finalize, dim3((m + 63) / 64), dim3(64), 0, st, pp, pb, m, res + a, (int)energy,
                           (int)temporal, (int)fhp);
    }
#endif // VQA_AB_VARIANTS
}

} // namespace vqa
