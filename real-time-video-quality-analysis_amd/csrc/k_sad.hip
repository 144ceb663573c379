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

namespace vqa {

constexpr int PW_DW = 20; // prev window row: 80 bytes
constexpr int PROWS = 31; // rows y0-8 .. y0+22

struct __align__(16) sad_lds {
    uint32_t cur[16][16];        // 16 rows x 64 bytes
    uint32_t prv[PROWS][PW_DW];  // 31 rows x 80 bytes, col 0 <-> x0-8
};

__global__ __launch_bounds__(256) void k_block_sad(const uint8_t *__restrict__ planes, int pitch,
                                                   int64_t plane_stride, int h, int w, int range, int first_has_prev,
                                                   vqa_frame_metrics *__restrict__ res)
{
    __shared__ sad_lds lds[4];
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
    sad_lds &L = lds[wv];
    for (int i = threadIdx.x; i < 129; i += 256) hist[i] = 0;
    unsigned long long sad_total = 0;
    for (int t0 = blockIdx.x * 4; t0 < tasks; t0 += gridDim.x * 4) {
        const int t = t0 + wv;
        const bool active = t < tasks;
        const int by = active ? t / ngx : 0;
        const int gx = active ? t - by * ngx : 0;
        const int y0 = by * 16, x0 = gx * 64;
        __syncthreads(); // previous iteration's LDS reads are done (also orders the hist clear)
        if (active) {
            // current rows: 16 x 8 units of 8 bytes
            for (int u = lane; u < 16 * 8; u += 64) {
                const int r = u >> 3, c = u & 7;
                uint64_t v = 0;
                if (x0 + c * 8 + 8 <= pitch) v = *(const uint64_t *)(curr + (int64_t)(y0 + r) * pitch + x0 + c * 8);
                L.cur[r][2 * c] = (uint32_t)v;
                L.cur[r][2 * c + 1] = (uint32_t)(v >> 32);
            }
            // previous window: 31 rows x 10 units; rows clamped, columns outside the pitch read as 0
            for (int u = lane; u < PROWS * 10; u += 64) {
                const int r = u / 10, c = u - r * 10;
                const int y = min(max(y0 - 8 + r, 0), h - 1);
                const int x = x0 - 8 + c * 8;
                uint64_t v = 0;
                if (x >= 0 && x + 8 <= pitch) v = *(const uint64_t *)(prev + (int64_t)y * pitch + x);
                L.prv[r][2 * c] = (uint32_t)v;
                L.prv[r][2 * c + 1] = (uint32_t)(v >> 32);
            }
        }
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
            const int bx = gx * 4 + blk;
            const bool vy = (dy >= -range) && (dy <= range) && (y0 + dy >= 0) && (y0 + 16 + dy <= h) && (bx < nbx);
            uint32_t best = 0xffffffffu;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int dx = j - 8;
                const uint32_t sad = (uint32_t)(acc[j >> 2] >> (16 * (j & 3))) & 0xffffu;
                const bool v = vy && (dx >= -range) && (dx <= range) && (bx * 16 + dx >= 0) && (bx * 16 + 16 + dx <= w);
                const uint32_t key = (sad << 16) | ((uint32_t)(dy * dy + dx * dx) << 8) | (uint32_t)(dyi * 16 + j);
                best = min(best, v ? key : 0xffffffffu);
            }
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) best = min(best, (uint32_t)__shfl_xor((int)best, m, 16));
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
        if (blockIdx.x == 0) res[f].sad_blocks = (uint32_t)(nby * nbx);
    }
}

void launch_block_sad(hipStream_t st, const uint8_t *planes, int pitch, int64_t plane_stride, int n, int h, int w,
                      int range, bool first_has_prev, vqa_frame_metrics *res)
{
    if (n <= 0) return;
    const int nby = h / 16, nbx = w / 16;
    const int tasks = nby * ((nbx + 3) / 4);
    int bpf = (tasks + 4 * 4 - 1) / (4 * 4); // ~4 tasks per wave
    bpf = bpf < 1 ? 1 : (bpf > 256 ? 256 : bpf);
    hipLaunchKernelGGL(k_block_sad, dim3(bpf, n), dim3(256), 0, st, planes, pitch, plane_stride, h, w, range,
                       (int)first_has_prev, res);
}

} // namespace vqa
