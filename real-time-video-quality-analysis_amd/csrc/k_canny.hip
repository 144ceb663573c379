// k_canny.hip — cv2.Canny(gray, low, high) edge-pixel count for gfx950.
//
// Reference function replaced: process_edge_frame, complexity_metrics.py:477-504
//   cv2.Canny(gray, 100, 200) (aperture 3, L1 gradient) -> np.sum(edges > 0)
//
// Stage 1 (k_canny_nms): 64x32 tile per workgroup.  Gray tile + 2-pixel
//   replicated halo staged in LDS, Sobel 3x3 + L1 magnitude for the tile + 1
//   halo (magnitude outside the image = 0, as OpenCV's zero-bordered buffer),
//   NMS with the TG22 fixed-point sector test, double threshold.  Writes a
//   1 B/px state map: 0 none, 1 weak candidate, 2 edge.
// Stage 2 (k_canny_hyst): 8-connected hysteresis as an iterate-to-fixpoint over
//   the same tiles: each workgroup relaxes its tile in LDS until nothing
//   changes, then marks the neighbour tiles whose halo it changed.  Promotion
//   is monotone (1 -> 2 only), so the fixpoint is unique and independent of
//   scheduling: the count is bit-exact with the sequential stack walk.
//
// Roofline: HBM, 3P bytes per frame (read gray, write state, read state).
#include "vqa_dev.hpp"
#include "vqa_kernels.hpp"
#include "vqa_math.hpp"

namespace vqa {

constexpr int TW = 64, TH = 32;
constexpr int GP = 72;      // LDS gray row pitch (bytes): [0,4) left halo slot, [4,68) interior, [68,72) right
constexpr int GR = TH + 4;  // gray rows
constexpr int MW = TW + 2, MH = TH + 2;

canny_geom canny_tiles(int h, int w) { return canny_geom{(w + TW - 1) / TW, (h + TH - 1) / TH}; }

// tile_flags bit 0: tile holds weak pixels.
__global__ __launch_bounds__(256) void k_canny_nms(const uint8_t *__restrict__ gray, int pitch, int64_t plane_stride,
                                                   int h, int w, int low, int high, uint8_t *__restrict__ state,
                                                   uint32_t *__restrict__ tile_flags,
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
    // ---- NMS + double threshold on the TH x TW interior
    unsigned n_strong = 0, n_weak = 0;
    uint8_t *st = state + (int64_t)f * plane_stride;
    for (int i = tid; i < TH * TW; i += 256) {
        const int ly = i / TW, lx = i - ly * TW;
        const int y = y0 + ly, x = x0 + lx;
        if (y < h && x < w) {
            const uint16_t *c = smag + (ly + 1) * MW + (lx + 1);
            const int m = c[0];
            const int nb[8] = {c[-MW - 1], c[-MW], c[-MW + 1], c[-1], c[1], c[MW - 1], c[MW], c[MW + 1]};
            const int pk = sgxy[(ly + 1) * MW + (lx + 1)];
            const int gx = (int)(int16_t)(pk & 0xffff), gy = pk >> 16;
            const int s = canny_classify(m, gx, gy, nb, low, high);
            n_strong += (s == 2);
            n_weak += (s == 1);
            st[(int64_t)y * pitch + x] = (uint8_t)s;
        }
    }
    n_strong = wave_sum(n_strong);
    n_weak = wave_sum(n_weak);
    if (lane_id() == 0) {
        if (n_strong) atomicAdd(&s_cnt[0], n_strong);
        if (n_weak) atomicAdd(&s_cnt[1], n_weak);
    }
    __syncthreads();
    if (tid == 0) {
        if (s_cnt[0]) atomicAdd(&res[f].edge_strong, s_cnt[0]);
        if (s_cnt[1]) atomicAdd(&res[f].edge_weak, s_cnt[1]);
        tile_flags[((int64_t)f * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s_cnt[1] ? 1u : 0u;
    }
}

// dirty bit layout for the 8 neighbours (dy, dx):
//   bit0 (-1,-1) bit1 (-1,0) bit2 (-1,+1) bit3 (0,-1) bit4 (0,+1) bit5 (+1,-1) bit6 (+1,0) bit7 (+1,+1)
constexpr int SW = TW + 2, SH = TH + 2;

__global__ __launch_bounds__(256) void k_canny_hyst(uint8_t *__restrict__ state, int pitch, int64_t plane_stride,
                                                    int h, int w, int round, const uint32_t *__restrict__ tile_flags,
                                                    uint32_t *__restrict__ dirty_in, uint32_t *__restrict__ dirty_out,
                                                    uint32_t *__restrict__ again, vqa_frame_metrics *__restrict__ res)
{
    __shared__ uint8_t ss[SH * SW];
    __shared__ unsigned s_flags[2]; // [0] neighbour-dirty mask, [1] promoted count
    const int f = blockIdx.z;
    const int64_t tile = ((int64_t)f * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    // wave-uniform activity test
    bool active;
    if (round == 0) {
        active = tile_flags[tile] != 0;
    } else {
        active = dirty_in[tile] != 0;
    }
    __syncthreads(); // every thread has read dirty_in before it is cleared
    if (round != 0 && threadIdx.x == 0 && active) dirty_in[tile] = 0;
    if (!active) return;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    uint8_t *st = state + (int64_t)f * plane_stride;
    const int tid = threadIdx.x;
    if (tid < 2) s_flags[tid] = 0;
    for (int i = tid; i < SH * SW; i += 256) {
        const int ly = i / SW, lx = i - ly * SW;
        const int y = y0 - 1 + ly, x = x0 - 1 + lx;
        ss[i] = (y >= 0 && y < h && x >= 0 && x < w) ? st[(int64_t)y * pitch + x] : (uint8_t)0;
    }
    __syncthreads();
    // thread owns 8 pixels: row ly = tid / 8 ... (TH*TW/256 = 8): column lx = tid % 64, rows (tid / 64) * 8 + k
    const int lx = tid & 63, lyb = (tid >> 6) * 8;
    uint32_t promoted_mask = 0;
    for (;;) {
        int changed = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint8_t *c = ss + (lyb + k + 1) * SW + (lx + 1);
            if (c[0] == 1) {
                const int any2 = (c[-SW - 1] == 2) | (c[-SW] == 2) | (c[-SW + 1] == 2) | (c[-1] == 2) | (c[1] == 2) |
                                 (c[SW - 1] == 2) | (c[SW] == 2) | (c[SW + 1] == 2);
                if (any2) { c[0] = 2; changed = 1; promoted_mask |= 1u << k; }
            }
        }
        if (!__syncthreads_or(changed)) break;
    }
    // write back promotions, count them, and find which neighbours saw their halo change
    unsigned cnt = 0, nbr = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (promoted_mask & (1u << k)) {
            const int ly = lyb + k;
            st[(int64_t)(y0 + ly) * pitch + (x0 + lx)] = 2;
            cnt++;
            const bool top = ly == 0, bot = (ly == TH - 1), lef = lx == 0, rig = (lx == TW - 1);
            if (top) nbr |= 1u << 1;
            if (bot) nbr |= 1u << 6;
            if (lef) nbr |= 1u << 3;
            if (rig) nbr |= 1u << 4;
            if (top && lef) nbr |= 1u << 0;
            if (top && rig) nbr |= 1u << 2;
            if (bot && lef) nbr |= 1u << 5;
            if (bot && rig) nbr |= 1u << 7;
        }
    }
    cnt = wave_sum(cnt);
    if (lane_id() == 0 && cnt) atomicAdd(&s_flags[1], cnt);
    if (nbr) atomicOr(&s_flags[0], nbr);
    __syncthreads();
    if (tid == 0) {
        if (s_flags[1]) atomicAdd(&res[f].edge_count, s_flags[1]);
        const unsigned m = s_flags[0];
        if (m) {
            const int dys[8] = {-1, -1, -1, 0, 0, 1, 1, 1}, dxs[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
            bool any = false;
            for (int b = 0; b < 8; b++) {
                if (!(m & (1u << b))) continue;
                const int ty = (int)blockIdx.y + dys[b], tx = (int)blockIdx.x + dxs[b];
                if (ty < 0 || ty >= (int)gridDim.y || tx < 0 || tx >= (int)gridDim.x) continue;
                const int64_t t2 = ((int64_t)f * gridDim.y + ty) * gridDim.x + tx;
                if (tile_flags[t2]) { dirty_out[t2] = 1; any = true; } // tiles without weak pixels cannot change
            }
            if (any) *again = 1;
        }
    }
}

// edge_count so far holds the promotions; add the strong pixels.
__global__ void k_canny_finish(int n, vqa_frame_metrics *__restrict__ res)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n) res[f].edge_count += res[f].edge_strong;
}

void launch_canny_nms(hipStream_t st, const uint8_t *gray, int pitch, int64_t plane_stride, int n, int h, int w,
                      int low, int high, uint8_t *state, uint32_t *tile_flags, vqa_frame_metrics *res)
{
    if (n <= 0) return;
    const canny_geom g = canny_tiles(h, w);
    hipLaunchKernelGGL(k_canny_nms, dim3(g.tiles_x, g.tiles_y, n), dim3(256), 0, st, gray, pitch, plane_stride, h, w,
                       low, high, state, tile_flags, res);
}

void launch_canny_hyst(hipStream_t st, uint8_t *state, int pitch, int64_t plane_stride, int n, int h, int w,
                       int round, uint32_t *tile_flags, uint32_t *dirty_in, uint32_t *dirty_out, uint32_t *again,
                       vqa_frame_metrics *res)
{
    if (n <= 0) return;
    const canny_geom g = canny_tiles(h, w);
    hipLaunchKernelGGL(k_canny_hyst, dim3(g.tiles_x, g.tiles_y, n), dim3(256), 0, st, state, pitch, plane_stride, h, w,
                       round, tile_flags, dirty_in, dirty_out, again, res);
}

void launch_canny_finish(hipStream_t st, int n, vqa_frame_metrics *res)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_canny_finish, dim3((n + 63) / 64), dim3(64), 0, st, n, res);
}

} // namespace vqa
