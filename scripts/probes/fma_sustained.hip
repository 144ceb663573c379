// Sustained fp32 vector FMA rate of the whole chip (not the issue rate of a short loop: scripts/valu_calib.hip measures
// that): every SIMD runs 8 waves of back-to-back independent v_pk_fma_f32 (or v_fma_f32) for seconds, so the number
// includes whatever clock the power management grants such a load.  Prints TFLOP/s; sample rocm-smi beside it.
//   ./fma_sustained [seconds] [mode: 0 = v_pk_fma_f32, 1 = v_fma_f32]
// (measurement tool, not product code)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k_fma(float *out, int iters, float w)
{
    f2 a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = f2{(float)threadIdx.x * 1e-3f + i, (float)i};
    const f2 x = f2{1.0000001f, 0.9999999f}, ww = f2{w, w};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) a[i] = __builtin_elementwise_fma(a[i], x, ww);
                else { a[i].x = fmaf(a[i].x, x.x, w); a[i].y = fmaf(a[i].y, x.y, w); }
            }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
    if (s == 12345.678f) out[0] = s;
}

int main(int argc, char **argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 3.0;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    float *o;
    if (hipMalloc(&o, 64) != hipSuccess) return 1;
    const int blocks = 256 * 8, iters = 20000; // 8 waves per SIMD on 256 CUs
    const double flop_per_launch = (double)blocks * 256 * iters * 64 * 2 * 2; // 64 pk_fma per iteration, 2 FMAs each, 2 flop
    auto run = [&]() {
        if (mode == 0) hipLaunchKernelGGL(k_fma<0>, dim3(blocks), dim3(256), 0, 0, o, iters, 1e-9f);
        else hipLaunchKernelGGL(k_fma<1>, dim3(blocks), dim3(256), 0, 0, o, iters, 1e-9f);
    };
    run(); hipDeviceSynchronize();
    int launches = 0;
    const auto t0 = std::chrono::steady_clock::now();
    double el = 0;
    while (el < secs) {
        for (int i = 0; i < 4; i++) run();
        hipDeviceSynchronize();
        launches += 4;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    printf("mode %s: %.1f TFLOP/s sustained over %.1f s (%d launches)\n", mode == 0 ? "v_pk_fma_f32" : "v_fma_f32",
           flop_per_launch * launches / el / 1e12, el, launches);
    return 0;
}
