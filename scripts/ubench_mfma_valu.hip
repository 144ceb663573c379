// ubench_mfma_valu.hip — does v_mfma_f32_4x4x1_16b_f32 issue concurrently with v_pk_fma_f32 on gfx950?
// (sizing probe for moving the vertical SSIM taps to the matrix pipe; not product code)
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_mfma_valu ubench_mfma_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// same question for the 16x16x4 tile (8 passes): 4 MFMAs (1024 MACs each) against 16 packed FMAs
template <int MODE>
__global__ __launch_bounds__(256) void k_probe16(float *out, const float *in, int iters)
{
    f4 acc[4];
    f2 v[8];
    const float a = in[threadIdx.x & 7], b = in[(threadIdx.x >> 3) & 7];
    const f2 w = {in[3], in[4]};
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = f4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = f2{a + i, b - i};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (MODE & 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            if (MODE & 2) {
#pragma unroll
                for (int q = 0; q < 4; q++)
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v[(2 * i + q) & 7]) : "v"(w), "v"(w));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
#pragma unroll
    for (int i = 0; i < 8; i++) s += v[i].x + v[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> // 1 = mfma only, 2 = pk_fma only, 3 = both interleaved
__global__ __launch_bounds__(256) void k_probe(float *out, const float *in, int iters)
{
    f4 acc[8];
    f2 v[8];
    const float a = in[threadIdx.x & 7], b = in[(threadIdx.x >> 3) & 7];
    const f2 w = {in[3], in[4]};
#pragma unroll
    for (int i = 0; i < 8; i++) { acc[i] = f4{0, 0, 0, 0}; v[i] = f2{a + i, b - i}; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE & 1) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
            if (MODE & 2) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(w), "v"(w));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v[(i + 4) & 7]) : "v"(w), "v"(w));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w + v[i].x + v[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    float *out, *in; float hin[16];
    for (int i = 0; i < 16; i++) hin[i] = 1.0f + i * 1e-3f;
    CHK(hipMalloc(&out, 256 * 4096 * 4)); CHK(hipMalloc(&in, 64)); CHK(hipMemcpy(in, hin, 64, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 4000;
    const char *names[] = {"", "mfma_4x4x1 x8", "pk_fma x16", "mfma x8 + pk_fma x16"};
    for (int wps = 1; wps <= 8; wps *= 2) {
        for (int mode = 1; mode <= 3; mode++) {
            dim3 grid(256 * wps);
            for (int rep = 0; rep < 2; rep++) {
                CHK(hipEventRecord(e0));
                if (mode == 1) hipLaunchKernelGGL(k_probe<1>, grid, dim3(256), 0, 0, out, in, iters);
                if (mode == 2) hipLaunchKernelGGL(k_probe<2>, grid, dim3(256), 0, 0, out, in, iters);
                if (mode == 3) hipLaunchKernelGGL(k_probe<3>, grid, dim3(256), 0, 0, out, in, iters);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            }
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            // cycles per loop iteration per wave, on a SIMD shared by `wps` waves, at an assumed 2.4 GHz
            double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * wps);
            printf("wps=%d %-24s %8.3f ms  %7.1f cyc/iter/wave(@2.4GHz)\n", wps, names[mode], ms, cyc);
        }
    }
    const char *names16[] = {"", "mfma_16x16x4 x4", "pk_fma x16", "mfma16 x4 + pk_fma x16"};
    for (int wps = 1; wps <= 8; wps *= 2) {
        for (int mode = 1; mode <= 3; mode++) {
            dim3 grid(256 * wps);
            for (int rep = 0; rep < 2; rep++) {
                CHK(hipEventRecord(e0));
                if (mode == 1) hipLaunchKernelGGL(k_probe16<1>, grid, dim3(256), 0, 0, out, in, iters);
                if (mode == 2) hipLaunchKernelGGL(k_probe16<2>, grid, dim3(256), 0, 0, out, in, iters);
                if (mode == 3) hipLaunchKernelGGL(k_probe16<3>, grid, dim3(256), 0, 0, out, in, iters);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            }
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * wps);
            printf("wps=%d %-24s %8.3f ms  %7.1f cyc/iter/wave(@2.4GHz)\n", wps, names16[mode], ms, cyc);
        }
    }
    return 0;
}
