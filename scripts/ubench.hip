// ubench.hip — instruction-throughput probes for gfx950 used to size the kernels (not product code).
// build: hipcc --offload-arch=gfx950 -O3 -o ubench ubench.hip ; run: ./ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(float *out, const float *in, int iters)
{
    float a[8], b = in[threadIdx.x & 7], c = in[8];
    uint64_t q[4] = {0, 0, 0, 0};
    uint32_t qs = (uint32_t)threadIdx.x * 2654435761u;
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = in[i] + threadIdx.x;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { // scalar fma, 8 independent chains
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (MODE == 1) { // packed fma on pairs
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 x = {a[i], a[i + 1]}, y = {b, b}, z = {c, c};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
                a[i] = x.x; a[i + 1] = x.y;
            }
        } else if (MODE == 2) { // qsad
#pragma unroll
            for (int i = 0; i < 4; i++) {
                q[i] = __builtin_amdgcn_qsad_pk_u16_u8(q[i] ^ 0x0102030405060708ull, qs, q[i]);
                q[i] = __builtin_amdgcn_qsad_pk_u16_u8(q[i] ^ 0x0102030405060708ull, qs, q[i]);
            }
        } else if (MODE == 3) { // sad_u8
            uint32_t s[8];
#pragma unroll
            for (int i = 0; i < 8; i++) s[i] = (uint32_t)q[i & 3];
#pragma unroll
            for (int i = 0; i < 8; i++) s[i] = __builtin_amdgcn_sad_u8(s[i], qs, s[i]);
#pragma unroll
            for (int i = 0; i < 4; i++) q[i] = s[i] + s[i + 4];
        } else if (MODE == 4) { // cvt ubyte
#pragma unroll
            for (int i = 0; i < 8; i++) { uint32_t u = __float_as_uint(a[i]); float r; asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(r) : "v"(u)); a[i] = r; }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)(q[0] + q[1] + q[2] + q[3]);
}

__global__ __launch_bounds__(256) void k_lds(float *out, int iters)
{
    __shared__ float4 buf[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) buf[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    float4 acc = make_float4(0, 0, 0, 0);
    int idx = threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++) { float4 v = buf[(idx + k * 37) & 1023]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        idx = (idx + 1) & 1023;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main()
{
    float *out, *in; float hin[16];
    for (int i = 0; i < 16; i++) hin[i] = 1.0f + i * 1e-3f;
    CHK(hipMalloc(&out, 256 * 4096 * 4)); CHK(hipMalloc(&in, 64)); CHK(hipMemcpy(in, hin, 64, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 4000;
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32(x2 flops)", "v_qsad_pk_u16_u8", "v_sad_u8", "v_cvt_f32_ubyte0"};
    for (int wps = 1; wps <= 8; wps *= 2) { // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
        for (int mode = 0; mode < 5; mode++) {
            dim3 grid(256 * wps);
            for (int rep = 0; rep < 2; rep++) {
                CHK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k_probe<0>, grid, dim3(256), 0, 0, out, in, iters);
                if (mode == 1) hipLaunchKernelGGL(k_probe<1>, grid, dim3(256), 0, 0, out, in, iters);
                if (mode == 2) hipLaunchKernelGGL(k_probe<2>, grid, dim3(256), 0, 0, out, in, iters);
                if (mode == 3) hipLaunchKernelGGL(k_probe<3>, grid, dim3(256), 0, 0, out, in, iters);
                if (mode == 4) hipLaunchKernelGGL(k_probe<4>, grid, dim3(256), 0, 0, out, in, iters);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            }
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            double instr = (double)grid.x * 4 /*waves*/ * iters * 8.0; // wave-instructions
            double per_simd_cycles = ms * 1e-3 * 2.4e9 / (instr / (256.0 * 4)); // cycles per wave-instr per SIMD @2.4GHz
            printf("wps=%d %-24s %8.3f ms  %6.2f cyc/wave-instr/SIMD (@2.4GHz)  %8.2f T lane-ops/s\n", wps, names[mode], ms,
                   per_simd_cycles, instr * 64 / (ms * 1e-3) / 1e12);
        }
    }
    for (int wps = 1; wps <= 8; wps *= 2) {
        dim3 grid(256 * wps);
        for (int rep = 0; rep < 2; rep++) { CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_lds, grid, dim3(256), 0, 0, out, 2000); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); }
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        double bytes = (double)grid.x * 256 * 2000 * 8 * 16;
        printf("wps=%d ds_read_b128: %8.3f ms  %7.1f TB/s chip  %6.1f B/clk/CU (@2.4GHz)\n", wps, ms, bytes / (ms * 1e-3) / 1e12,
               bytes / 256 / (ms * 1e-3 * 2.4e9));
    }
    return 0;
}
