// k_sad.hip — 16x16 block-SAD full-search motion for gfx950.
//
// Stands in for process_frame_complexity (complexity_metrics.py:313-343), whose
// Farneback optical flow BASELINE.json's north_star replaces with block-SAD
// motion.  There is no numeric parity with Farneback; the spec is the CPU
// restatement (oracle/vqa_oracle.c, vqo_block_sad):
//   gray planes at full resolution (the reference never resizes for motion),
//   16x16 blocks, only whole blocks; dy,dx in [-R,R], R <= 7; candidate valid iff
//   the displaced block is inside the frame; winner = min SAD, then min dx^2+dy^2,
//   then raster order; outputs: sum of winning SADs and a histogram of dx^2+dy^2.
//
// Mapping: one wave handles 4 horizontally adjacent blocks.  lane = blk*16 + dyi:
// each lane owns one vertical displacement dy = dyi-8 of one block and evaluates
// all 16 horizontal displacements dx = -8..7 with V_QSAD_PK_U16_U8, which yields
// four packed 16-bit SADs (4 consecutive dx) per instruction: 16 absolute
// differences per lane-op.  The 16-bit accumulators cannot overflow
// (16*16*255 = 65280).  The (sad, d2, raster) key is min-reduced over the 16
// lanes of a block with xor-shuffles.  Current block rows and the previous
// frame's 31 x 80 search window are staged in LDS per wave.
//
// Roofline: HBM 2P bytes per frame pair; ~16 QSADs per pixel of VALU work.
#include "vqa_dev.hpp"
#include "vqa_kernels.hpp"

#include <cstdlib>

namespace vqa {

constexpr int PW_DW = 20; // prev window row: 80 bytes
constexpr int PROWS = 31; // rows y0-8 .. y0+22

struct __align__(16) sad_lds {
    uint32_t cur[16][16];        // 16 rows x 64 bytes
    uint32_t prv[PROWS][PW_DW];  // 31 rows x 80 bytes, col 0 <-> x0-8
};

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_min(uint32_t v)
{
    return min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false));
}

__global__ __launch_bounds__(256) void k_block_sad(const uint8_t *__restrict__ planes, int pitch,
                                                   int64_t plane_stride, int h, int w, int range, int first_has_prev,
                                                   vqa_frame_metrics *__restrict__ res, int bpf, int n_wg)
{
    __shared__ sad_lds lds[4];
    __shared__ unsigned hist[129];
    __shared__ unsigned long long red[4];
    // XCD-aware placement (workgroup ids go round-robin over the 8 XCDs): XCD c takes the contiguous range
    // [c * per, (c + 1) * per) of the (frame, block) space, so all tiles of a frame - whose 31 x 80 search windows
    // overlap 2.4-fold - meet in ONE L2 instead of eight (PMC traffic 1.71x -> see profiles/)
    const int per = (n_wg + 7) >> 3;
    const int wg = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || wg >= n_wg) return;
    const int f = wg / bpf, bx = wg - f * bpf;
    if (f == 0 && !first_has_prev) return;
    const uint8_t *curr = planes + (int64_t)(f + 1) * plane_stride;
    const uint8_t *prev = planes + (int64_t)f * plane_stride;
    const int nby = h >> 4, nbx = w >> 4;
    const int ngx = (nbx + 3) >> 2;
    const int tasks = nby * ngx;
    const int lane = lane_id();
    const int wv = __builtin_amdgcn_readfirstlane((int)wave_id()); // wave-uniform: tile geometry stays on the scalar unit
    const int blk = lane >> 4, dyi = lane & 15, dy = dyi - 8;
    sad_lds &L = lds[wv];
    for (int i = threadIdx.x; i < 129; i += 256) hist[i] = 0;
    unsigned long long sad_total = 0;

    // ---- lane constants of the staging: a lane owns 2 of the tile's 128 current-row units and 5 of its 310 window
    // units (8 bytes each).  Their offsets from the tile origin never change, so an INTERIOR tile (window inside
    // the plane) loads with a scalar base + these 32-bit offsets: no vector ALU work for addressing or clamping.
    int coff[2], poff[5], pr_[5], pc_[5];
#pragma unroll
    for (int i = 0; i < 2; i++) { const int u = lane + 64 * i; coff[i] = (u >> 3) * pitch + (u & 7) * 8; }
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const int u = min(lane + 64 * i, PROWS * 10 - 1); // units 310..319 do not exist: they alias the last one
        pr_[i] = u / 10; pc_[i] = u - pr_[i] * 10;
        poff[i] = (pr_[i] - 8) * pitch + pc_[i] * 8 - 8;
    }
    // ---- lane constants of the epilogue: key = (sad << 16) | kc[j]; an invalid candidate's kc is all ones.
    // For an interior tile validity is the range test alone (lane- and j-constant).
    const uint32_t lanepart = ((uint32_t)(dy * dy) << 8) | ((uint32_t)dyi << 4);
    uint32_t kci[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int dx = j - 8;
        const bool v = (dy >= -range) && (dy <= range) && (dx >= -range) && (dx <= range);
        kci[j] = v ? lanepart + (((uint32_t)(dx * dx) << 8) | (uint32_t)j) : 0xffffffffu;
    }

    uint64_t pcur[2], pprv[5]; // the NEXT tile's bytes, in flight while this tile's QSADs run
    auto fetch = [&](int t) {
        if (t >= tasks) return; // wave-uniform
        const int by = t / ngx, gx = t - by * ngx;
        const int y0 = by * 16, x0 = gx * 64;
        const bool interior = (y0 >= 8) && (y0 + 23 <= h) && (x0 >= 8) && (x0 + 72 <= pitch);
        if (interior) {
            const gptr_u8 cb = uniform_ptr(curr + (int64_t)y0 * pitch + x0), pb = uniform_ptr(prev + (int64_t)y0 * pitch + x0);
#pragma unroll
            for (int i = 0; i < 2; i++) pcur[i] = *(const __attribute__((address_space(1))) uint64_t *)(cb + coff[i]);
#pragma unroll
            for (int i = 0; i < 5; i++) pprv[i] = *(const __attribute__((address_space(1))) uint64_t *)(pb + poff[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int u = lane + 64 * i, r = u >> 3, c = u & 7;
                pcur[i] = 0;
                if (x0 + c * 8 + 8 <= pitch) pcur[i] = *(const uint64_t *)(curr + (int64_t)(y0 + r) * pitch + x0 + c * 8);
            }
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const int y = min(max(y0 - 8 + pr_[i], 0), h - 1), x = x0 - 8 + pc_[i] * 8;
                pprv[i] = 0;
                if (x >= 0 && x + 8 <= pitch) pprv[i] = *(const uint64_t *)(prev + (int64_t)y * pitch + x);
            }
        }
    };
    fetch(bx * 4 + wv);
    for (int t0 = bx * 4; t0 < tasks; t0 += bpf * 4) {
        const int t = t0 + wv;
        const bool active = t < tasks; // wave-uniform
        const int by = active ? t / ngx : 0;
        const int gx = active ? t - by * ngx : 0;
        const int y0 = by * 16, x0 = gx * 64;
        __syncthreads(); // previous iteration's LDS reads are done (also orders the hist clear)
        if (active) {
#pragma unroll
            for (int i = 0; i < 2; i++) { const int u = lane + 64 * i; *(uint64_t *)&L.cur[u >> 3][2 * (u & 7)] = pcur[i]; }
#pragma unroll
            for (int i = 0; i < 5; i++) *(uint64_t *)&L.prv[pr_[i]][2 * pc_[i]] = pprv[i]; // (aliased units rewrite the same bytes)
        }
        fetch(t + bpf * 4);
        __syncthreads();
        if (active) {
            uint64_t acc[4] = {0, 0, 0, 0};
#pragma unroll 4
            for (int r = 0; r < 16; r++) {
                const uint4 c4 = *(const uint4 *)&L.cur[r][blk * 4];
                const uint4 pa = *(const uint4 *)&L.prv[r + dyi][blk * 4];
                const uint4 pb = *(const uint4 *)&L.prv[r + dyi][blk * 4 + 4];
                const uint32_t D[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
                const uint32_t C[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                for (int g = 0; g < 4; g++) {
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const uint64_t s0 = (uint64_t)D[c + g] | ((uint64_t)D[c + g + 1] << 32);
                        acc[g] = __builtin_amdgcn_qsad_pk_u16_u8(s0, C[c], acc[g]);
                    }
                }
            }
            // winner of the lane's 16 candidates: key = (sad << 16) | kc[j]
            auto lane_best = [&](const uint32_t (&kc)[16]) -> uint32_t {
                uint32_t best = 0xffffffffu;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const uint32_t lo = (uint32_t)acc[g], hi = (uint32_t)(acc[g] >> 32);
                    const uint32_t k0 = (lo << 16) | kc[4 * g], k1 = (lo & 0xffff0000u) | kc[4 * g + 1];
                    const uint32_t k2 = (hi << 16) | kc[4 * g + 2], k3 = (hi & 0xffff0000u) | kc[4 * g + 3];
                    best = min(min(best, k0), min(k1, min(k2, k3)));
                }
                return best;
            };
            const int bx = gx * 4 + blk;
            const bool interior = (y0 >= 8) && (y0 + 23 <= h) && (x0 >= 8) && (x0 + 72 <= w) && (gx * 4 + 3 < nbx);
            uint32_t best;
            if (interior) { // wave-uniform
                best = lane_best(kci);
            } else {
                const bool vy = (dy >= -range) && (dy <= range) && (y0 + dy >= 0) && (y0 + 16 + dy <= h) && (bx < nbx);
                uint32_t kc[16];
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int dx = j - 8;
                    const bool v = vy && (dx >= -range) && (dx <= range) && (bx * 16 + dx >= 0) && (bx * 16 + 16 + dx <= w);
                    kc[j] = v ? lanepart + (((uint32_t)(dx * dx) << 8) | (uint32_t)j) : 0xffffffffu;
                }
                best = lane_best(kc);
            }
            // min over the block's 16 dy lanes (one DPP row); every lane ends with the row's minimum
            best = dpp_min<0xB1>(best);  // quad_perm [1,0,3,2]
            best = dpp_min<0x4E>(best);  // quad_perm [2,3,0,1]
            best = dpp_min<0x141>(best); // row_half_mirror
            best = dpp_min<0x140>(best); // row_mirror
            if (dyi == 0 && best != 0xffffffffu) {
                sad_total += best >> 16;
                atomicAdd(&hist[(best >> 8) & 0xffu], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 129; i += 256)
        if (hist[i]) atomicAdd(&res[f].mv_d2_hist[i], hist[i]);
    const unsigned long long tot = block_sum_u64(sad_total, red);
    if (threadIdx.x == 0) {
        if (tot) atomicAdd((unsigned long long *)&res[f].sad_sum, tot);
        if (bx == 0) res[f].sad_blocks = (uint32_t)(nby * nbx);
    }
}

#ifdef VQA_AB_VARIANTS // lab build only
// ---------------------------------------------------------------------------
// Pruned form (VQA_SAD_VARIANT=2; NOT the default): the same winner, found without evaluating most candidates.
// Measured on MI355X, 256 x 1080p (round 2): S-natural 1.59-1.65 ms, S-noise 3.5 ms, against 1.29 ms for the exhaustive
// kernel on either content.  v_qsad_pk_u16_u8 makes the exhaustive search cost ~4.6 issue cycles per candidate; the
// bound below costs ~2.3 (box sums 0.26 ms + bound 0.16 ms per batch), and the ~20 surviving groups of a tile still
// occupy one full 64-QSAD pass of the wave with scattered LDS reads (0.49 ms).  Kept for re-measurement and as the
// reference implementation of the exactness argument; bit-identical results (tests/test_gpu_parity.py).
//
// Successive elimination on 4x16 sub-block sums.  For any displacement d the triangle inequality gives
//     SAD(d) >= LB(d) = sum_{t<4} | Rc4[t] - Rp4[8+dy+4t][16 blk + 8+dx] |
// with Rc4[t] = sum of rows 4t..4t+3 of the current block and Rp4[y][x] = sum of the 4x16 box of the previous
// frame's search window whose corner is (y, x).  A candidate whose LB exceeds the SAD of ANY evaluated candidate
// cannot win and cannot tie (the key orders by SAD first), so skipping it leaves the (SAD, d^2, raster) winner
// bit-identical to the exhaustive search.  Per 64x16 tile (4 blocks, one wave, wave-private LDS, no barriers):
//   1. stage the current rows and the 31x80 window in LDS (as the exhaustive kernel)
//   2. Rp4: v_qsad_pk_u16_u8 against zero gives four sliding 4-pixel sums per instruction; each lane slides a
//      4-row window down its column group, then four neighbours are added (packed u16, plain 32-bit adds)
//   3. LB for all 16x16 (dy,dx) of a block: lane = (block, dy), 2 v_perm_b32 + 2 v_sad_u16 per candidate
//   4. seed: the 4-candidate group with the smallest LB is evaluated by the block's 16 lanes, one row each
//   5. every group whose LB <= the seed's best SAD is compacted into a list (ballot + mbcnt); one lane evaluates one
//      group (64 QSADs, the exhaustive kernel's inner loop) and ds_min's its key into the block's slot
// The work is content-dependent: natural frames leave 3-10 of 60 groups per block, uncorrelated noise leaves all
// of them.
// ---------------------------------------------------------------------------
constexpr int R4_STRIDE = 72; // u16 per Rp4 row: 64 + 8 pad (144 B: 16 consecutive rows fall into 16 distinct bank groups)

struct __align__(16) sea_lds {
    uint32_t cur[16][16];             // 16 rows x 64 bytes
    uint32_t prv[PROWS][PW_DW];       // 31 rows x 80 bytes, col 0 <-> x0-8
    uint2 T[28][20];                  // vertical 4-row sums of the 4-pixel sums: T[y][j] = 4 x u16 for x = 4j..4j+3
    uint16_t r4[28][R4_STRIDE];       // Rp4[y][x], x = 0..63
    uint16_t rc4[4][4];               // Rc4[blk][t]
    uint32_t best[4];                 // per block: min key of the evaluated candidates
    uint8_t list[256];                // surviving (lane << 2 | group) items
};

__device__ __forceinline__ void wave_lds_sync()
{
    // LDS traffic between lanes of ONE wave: the LDS queue is in order per wave, only the compiler must not reorder
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint2 add2(uint2 a, uint2 b) { return make_uint2(a.x + b.x, a.y + b.y); }

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
// sum over the 16 lanes of a DPP row, result in every lane
__device__ __forceinline__ uint32_t row_sum16(uint32_t v)
{
    v += dpp_mov<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v); // row_half_mirror
    v += dpp_mov<0x140>(v); // row_mirror
    return v;
}

// key of candidate (dyi, j) with the given SAD: the exhaustive kernel's (sad, d^2, raster)
__device__ __forceinline__ uint32_t sad_key(uint32_t sad, int dyi, int j)
{
    const int dy = dyi - 8, dx = j - 8;
    return (sad << 16) | ((uint32_t)(dy * dy + dx * dx) << 8) | (uint32_t)(dyi * 16 + j);
}

__global__ __launch_bounds__(256) void k_block_sad_sea(const uint8_t *__restrict__ planes, int pitch,
                                                       int64_t plane_stride, int h, int w, int range, int first_has_prev,
                                                       vqa_frame_metrics *__restrict__ res)
{
    __shared__ sea_lds lds[4];
    __shared__ unsigned hist[129];
    __shared__ unsigned long long red[4];
    const int f = blockIdx.y;
    if (f == 0 && !first_has_prev) return;
    const uint8_t *curr = planes + (int64_t)(f + 1) * plane_stride;
    const uint8_t *prev = planes + (int64_t)f * plane_stride;
    const int nby = h >> 4, nbx = w >> 4;
    const int ngx = (nbx + 3) >> 2;
    const int tasks = nby * ngx;
    const int lane = lane_id(), wv = wave_id();
    const int blk = lane >> 4, dyi = lane & 15, dy = dyi - 8;
    sea_lds &L = lds[wv];
    for (int i = threadIdx.x; i < 129; i += 256) hist[i] = 0;
    __syncthreads();
    unsigned long long sad_total = 0;
    for (int t = blockIdx.x * 4 + wv; t < tasks; t += gridDim.x * 4) {
        const int by = t / ngx, gx = t - by * ngx;
        const int y0 = by * 16, x0 = gx * 64;
        wave_lds_sync(); // the previous tile's reads are done
        // ---- 1. stage (identical to the exhaustive kernel)
        for (int u = lane; u < 16 * 8; u += 64) {
            const int r = u >> 3, c = u & 7;
            uint64_t v = 0;
            if (x0 + c * 8 + 8 <= pitch) v = *(const uint64_t *)(curr + (int64_t)(y0 + r) * pitch + x0 + c * 8);
            L.cur[r][2 * c] = (uint32_t)v;
            L.cur[r][2 * c + 1] = (uint32_t)(v >> 32);
        }
        for (int u = lane; u < PROWS * 10; u += 64) {
            const int r = u / 10, c = u - r * 10;
            const int y = min(max(y0 - 8 + r, 0), h - 1);
            const int x = x0 - 8 + c * 8;
            uint64_t v = 0;
            if (x >= 0 && x + 8 <= pitch) v = *(const uint64_t *)(prev + (int64_t)y * pitch + x);
            L.prv[r][2 * c] = (uint32_t)v;
            L.prv[r][2 * c + 1] = (uint32_t)(v >> 32);
        }
        if (lane < 4) L.best[lane] = 0xffffffffu;
        wave_lds_sync();
        // ---- 2a. Rc4: lane (blk, r) sums its row, rows r..r+3 meet over DPP row_shl
        {
            const uint4 c4 = *(const uint4 *)&L.cur[dyi][blk * 4];
            uint32_t s = __builtin_amdgcn_sad_u8(c4.x, 0u, 0u);
            s = __builtin_amdgcn_sad_u8(c4.y, 0u, s);
            s = __builtin_amdgcn_sad_u8(c4.z, 0u, s);
            s = __builtin_amdgcn_sad_u8(c4.w, 0u, s);
            const uint32_t s4 = s + dpp_mov<0x101>(s) + dpp_mov<0x102>(s) + dpp_mov<0x103>(s); // row_shl:1,2,3
            if ((dyi & 3) == 0) L.rc4[blk][dyi >> 2] = (uint16_t)s4;
        }
        // ---- 2b. T[y][j] = sum_{i<4} Q[y+i][4j..4j+3], Q = sliding 4-pixel sums (QSAD against zero)
        if (lane < 57) {
            const int seg = lane / 19, j = lane - seg * 19;
            const int ys = seg * 10, ye = seg == 2 ? 28 : ys + 10;
            auto qv = [&](int y) -> uint2 {
                const uint64_t s0 = (uint64_t)L.prv[y][j] | ((uint64_t)L.prv[y][j + 1] << 32);
                const uint64_t q = __builtin_amdgcn_qsad_pk_u16_u8(s0, 0u, 0ull);
                return make_uint2((uint32_t)q, (uint32_t)(q >> 32));
            };
            uint2 a0 = qv(ys), a1 = qv(ys + 1), a2 = qv(ys + 2);
            for (int y = ys; y < ye; y++) {
                const uint2 a3 = qv(y + 3);
                L.T[y][j] = add2(add2(a0, a1), add2(a2, a3)); // u16 fields <= 4 * 1020: no carry between them
                a0 = a1; a1 = a2; a2 = a3;
            }
        }
        wave_lds_sync();
        // ---- 2c. Rp4[y][4j..4j+3] = T[y][j] + T[y][j+1] + T[y][j+2] + T[y][j+3]   (fields <= 16320)
        for (int id = lane; id < 28 * 16; id += 64) {
            const int y = id >> 4, j = id & 15;
            const uint2 v = add2(add2(L.T[y][j], L.T[y][j + 1]), add2(L.T[y][j + 2], L.T[y][j + 3]));
            *(uint2 *)&L.r4[y][4 * j] = v;
        }
        wave_lds_sync();
        // ---- 3. lower bounds of the lane's 16 candidates (dy fixed, j = dx + 8 = 0..15)
        const int bx = gx * 4 + blk;
        const bool vy = (dy >= -range) && (dy <= range) && (y0 + dy >= 0) && (y0 + 16 + dy <= h) && (bx < nbx);
        uint32_t vxmask = 0; // bit j: dx = j - 8 is inside the range and the frame
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int dx = j - 8;
            if ((dx >= -range) && (dx <= range) && (bx * 16 + dx >= 0) && (bx * 16 + 16 + dx <= w)) vxmask |= 1u << j;
        }
        uint32_t lbg[4];
        {
            const uint32_t c01 = *(const uint32_t *)&L.rc4[blk][0], c23 = *(const uint32_t *)&L.rc4[blk][2];
            uint32_t lb[16];
#pragma unroll
            for (int half = 0; half < 2; half++) { // terms (t, t+1) = (0,1) then (2,3)
                const uint32_t cc = half ? c23 : c01;
                const uint4 a0 = *(const uint4 *)&L.r4[dyi + 8 * half][16 * blk], a1 = *(const uint4 *)&L.r4[dyi + 8 * half][16 * blk + 8];
                const uint4 b0 = *(const uint4 *)&L.r4[dyi + 8 * half + 4][16 * blk], b1 = *(const uint4 *)&L.r4[dyi + 8 * half + 4][16 * blk + 8];
                const uint32_t A[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                const uint32_t B[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int p = 0; p < 8; p++) {
                    const uint32_t e = __builtin_amdgcn_perm(B[p], A[p], 0x05040100u); // (A.lo16, B.lo16): candidate 2p
                    const uint32_t o = __builtin_amdgcn_perm(B[p], A[p], 0x07060302u); // (A.hi16, B.hi16): candidate 2p+1
                    lb[2 * p] = __builtin_amdgcn_sad_u16(e, cc, half ? lb[2 * p] : 0u);
                    lb[2 * p + 1] = __builtin_amdgcn_sad_u16(o, cc, half ? lb[2 * p + 1] : 0u);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; g++) {
                uint32_t m = 0xffffffu;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int j = 4 * g + i;
                    m = min(m, (vy && ((vxmask >> j) & 1u)) ? lb[j] : 0xffffffu);
                }
                lbg[g] = m;
            }
        }
        // ---- 4. seed: the block's group with the smallest bound, one row per lane
        uint32_t skey = 0xffffffffu;
#pragma unroll
        for (int g = 0; g < 4; g++) skey = min(skey, (lbg[g] << 8) | (uint32_t)(dyi << 2) | (uint32_t)g);
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) skey = min(skey, (uint32_t)__shfl_xor((int)skey, m, 16));
        const bool blk_live = (skey >> 8) != 0xffffffu; // at least one valid candidate
        const int s_dyi = (skey >> 2) & 15, s_g = skey & 3;
        uint32_t best_key = 0xffffffffu;
        {
            const int r = dyi; // this lane's row
            const uint4 c4 = *(const uint4 *)&L.cur[r][blk * 4];
            const uint32_t *pr = &L.prv[r + s_dyi][blk * 4 + s_g];
            const uint32_t d0 = pr[0], d1 = pr[1], d2 = pr[2], d3 = pr[3], d4 = pr[4];
            uint64_t acc = __builtin_amdgcn_qsad_pk_u16_u8((uint64_t)d0 | ((uint64_t)d1 << 32), c4.x, 0ull);
            acc = __builtin_amdgcn_qsad_pk_u16_u8((uint64_t)d1 | ((uint64_t)d2 << 32), c4.y, acc);
            acc = __builtin_amdgcn_qsad_pk_u16_u8((uint64_t)d2 | ((uint64_t)d3 << 32), c4.z, acc);
            acc = __builtin_amdgcn_qsad_pk_u16_u8((uint64_t)d3 | ((uint64_t)d4 << 32), c4.w, acc);
            const uint32_t lo = row_sum16((uint32_t)acc), hi = row_sum16((uint32_t)(acc >> 32)); // 16 rows: <= 65280 per field
            const uint32_t sads[4] = {lo & 0xffffu, lo >> 16, hi & 0xffffu, hi >> 16};
            const int sdy = s_dyi - 8;
            const bool svy = (sdy >= -range) && (sdy <= range) && (y0 + sdy >= 0) && (y0 + 16 + sdy <= h) && (bx < nbx);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int j = 4 * s_g + i;
                if (svy && ((vxmask >> j) & 1u)) best_key = min(best_key, sad_key(sads[i], s_dyi, j));
            }
        }
        if (!blk_live) best_key = 0xffffffffu;
        const uint32_t best_sad = best_key >> 16; // 0xffff when the block has no valid candidate
        // ---- 5. survivors: groups that could still hold a SAD <= the seed's; compact them, one lane per group
        int n_items = 0;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const bool sv = blk_live && (lbg[g] <= best_sad) && !(dyi == s_dyi && g == s_g);
            const uint64_t m = __ballot(sv);
            if (sv) {
                const int idx = n_items + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                L.list[idx] = (uint8_t)((lane << 2) | g);
            }
            n_items += __popcll(m);
        }
        if (dyi == 0 && blk_live) L.best[blk] = best_key;
        wave_lds_sync();
        for (int it = lane; it < n_items; it += 64) {
            const uint32_t d = L.list[it];
            const int ib = d >> 6, idyi = (d >> 2) & 15, ig = d & 3;
            uint64_t acc = 0;
#pragma unroll 4
            for (int r = 0; r < 16; r++) {
                const uint4 c4 = *(const uint4 *)&L.cur[r][ib * 4];
                const uint32_t *pr = &L.prv[r + idyi][ib * 4 + ig];
                const uint32_t d0 = pr[0], d1 = pr[1], d2 = pr[2], d3 = pr[3], d4 = pr[4];
                acc = __builtin_amdgcn_qsad_pk_u16_u8((uint64_t)d0 | ((uint64_t)d1 << 32), c4.x, acc);
                acc = __builtin_amdgcn_qsad_pk_u16_u8((uint64_t)d1 | ((uint64_t)d2 << 32), c4.y, acc);
                acc = __builtin_amdgcn_qsad_pk_u16_u8((uint64_t)d2 | ((uint64_t)d3 << 32), c4.z, acc);
                acc = __builtin_amdgcn_qsad_pk_u16_u8((uint64_t)d3 | ((uint64_t)d4 << 32), c4.w, acc);
            }
            const int ibx = gx * 4 + ib, idy = idyi - 8;
            const bool ivy = (idy >= -range) && (idy <= range) && (y0 + idy >= 0) && (y0 + 16 + idy <= h) && (ibx < nbx);
            uint32_t key = 0xffffffffu;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int j = 4 * ig + i, dx = j - 8;
                const bool v = ivy && (dx >= -range) && (dx <= range) && (ibx * 16 + dx >= 0) && (ibx * 16 + 16 + dx <= w);
                const uint32_t sad = (uint32_t)(acc >> (16 * i)) & 0xffffu;
                if (v) key = min(key, sad_key(sad, idyi, j));
            }
            if (key != 0xffffffffu) atomicMin(&L.best[ib], key);
        }
        wave_lds_sync();
        if (dyi == 0) {
            const uint32_t bk = L.best[blk];
            if (bk != 0xffffffffu) {
                sad_total += bk >> 16;
                atomicAdd(&hist[(bk >> 8) & 0xffu], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 129; i += 256)
        if (hist[i]) atomicAdd(&res[f].mv_d2_hist[i], hist[i]);
    const unsigned long long tot = block_sum_u64(sad_total, red);
    if (threadIdx.x == 0) {
        if (tot) atomicAdd((unsigned long long *)&res[f].sad_sum, tot);
        if (blockIdx.x == 0) res[f].sad_blocks = (uint32_t)(nby * nbx);
    }
}

#endif // VQA_AB_VARIANTS

void launch_block_sad(hipStream_t st, const uint8_t *planes, int pitch, int64_t plane_stride, int n, int h, int w,
                      int range, bool first_has_prev, vqa_frame_metrics *res)
{
    if (n <= 0) return;
    const int nby = h / 16, nbx = w / 16;
    const int tasks = nby * ((nbx + 3) / 4);
    int bpf = (tasks + 4 * 4 - 1) / (4 * 4); // ~4 tasks per wave
    bpf = bpf < 1 ? 1 : (bpf > 256 ? 256 : bpf);
#ifdef VQA_AB_VARIANTS
    static const int variant = ab_knob("VQA_SAD_VARIANT", 0) == 2 ? 2 : 0; // 0 = exhaustive search (shipped), 2 = pruned search
    if (variant == 2) {
        hipLaunchKernelGGL(k_block_sad_sea, dim3(bpf, n), dim3(256), 0, st, planes, pitch, plane_stride, h, w, range,
                           (int)first_has_prev, res);
        return;
    }
#endif
    hipLaunchKernelGGL(k_block_sad, dim3((unsigned)(8 * (((long long)bpf * n + 7) / 8))), dim3(256), 0, st, planes, pitch,
                       plane_stride, h, w, range, (int)first_has_prev, res, bpf, bpf * n);
}

} // namespace vqa
