// FETCH_SIZE calibration on known byte counts, per access width (MI355X_MICROARCH.md, HBM section: the x2 correction
// is established for 16 B/lane streaming reads only; "calibrate on a known byte count in your own access pattern").
// Each kernel streams the same 1 GiB buffer once (larger than the 256 MiB Infinity Cache) and writes one dword per wave.
//   rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib      ->  FETCH_SIZE (KiB) per kernel vs 1 048 576 KiB read
// (measurement tool, not product code)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32_any __attribute__((aligned(1)));
__global__ void read_b8(const uint8_t *p, size_t n, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void read_b32(const uint32_t *p, size_t n, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / 4; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
// the Canny NMS pattern: one UNALIGNED dword per lane at byte offset lane + 1 within a 64-byte group per wave-load
// (consecutive lanes one byte apart: a wave-load touches 67 contiguous bytes); every byte is fetched by ~4 lanes
__global__ void read_b32_step1(const uint8_t *p, size_t n, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + 4 < n; i += (size_t)gridDim.x * blockDim.x)
        acc += *(const u32_any *)(p + i);
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void read_b128(const uint4 *p, size_t n, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / 16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main()
{
    const size_t n = (size_t)1 << 30;
    uint8_t *d; uint32_t *o;
    if (hipMalloc(&d, n) != hipSuccess || hipMalloc(&o, 64) != hipSuccess) return 1;
    hipMemset(d, 1, n);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(read_b8, dim3(8192), dim3(256), 0, 0, d, n, o);
        hipLaunchKernelGGL(read_b32, dim3(8192), dim3(256), 0, 0, (const uint32_t *)d, n, o);
        hipLaunchKernelGGL(read_b32_step1, dim3(8192), dim3(256), 0, 0, d, n, o);
        hipLaunchKernelGGL(read_b128, dim3(8192), dim3(256), 0, 0, (const uint4 *)d, n, o);
    }
    hipDeviceSynchronize();
    printf("each kernel read %zu KiB\n", n / 1024);
    return 0;
}
