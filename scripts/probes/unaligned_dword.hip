// Probe: do global / buffer dword loads honour byte-unaligned addresses on gfx950?  (measurement tool, not product code)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32_any __attribute__((aligned(1)));
__global__ void k(const uint8_t *src, uint32_t *outg, uint32_t *outb, uint32_t *outd, const uint32_t *selv, uint32_t *outp)
{
    const int lane = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, (short)0, -1, 0x00020000);
    const uint32_t g = *(const u32_any *)(src + lane + 1);
    const uint32_t b = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, lane + 1, 0, 0);
    outg[lane] = g;
    outb[lane] = b;
    outp[lane] = __builtin_amdgcn_perm(g, g, selv[lane]);
    outd[lane] = __builtin_amdgcn_udot4(g, 0x00010201u, 0u, false) | ((((g >> 16) & 0xffu) - (g & 0xffu)) << 16);
}
int main()
{
    uint8_t h[512];
    for (int i = 0; i < 512; i++) h[i] = (uint8_t)(i * 7 + 3);
    uint8_t *d; uint32_t *og, *ob, *od, *sv, *op; uint32_t hs[64], pp[64];
    for (int i = 0; i < 64; i++) hs[i] = i % 3 == 0 ? 0x0c020100u : (i % 3 == 1 ? 0x0c010000u : 0x0c030302u);
    hipMalloc(&sv, 256); hipMalloc(&op, 256); hipMemcpy(sv, hs, 256, hipMemcpyHostToDevice);
    hipMalloc(&d, 512); hipMalloc(&og, 256); hipMalloc(&ob, 256); hipMalloc(&od, 256);
    hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, og, ob, od, sv, op);
    uint32_t g[64], b[64], dd[64];
    hipMemcpy(g, og, 256, hipMemcpyDeviceToHost); hipMemcpy(b, ob, 256, hipMemcpyDeviceToHost); hipMemcpy(dd, od, 256, hipMemcpyDeviceToHost);
    hipMemcpy(pp, op, 256, hipMemcpyDeviceToHost);
    int badg = 0, badb = 0, badd = 0, badp = 0;
    for (int l = 0; l < 64; l++) {
        uint32_t e = h[l + 1] | (h[l + 2] << 8) | (h[l + 3] << 16) | ((uint32_t)h[l + 4] << 24);
        uint32_t ed = (uint32_t)(h[l + 1] + 2 * h[l + 2] + h[l + 3]) | ((uint32_t)(h[l + 3] - h[l + 1]) << 16);
        badg += g[l] != e; badb += b[l] != e; badd += dd[l] != ed;
        { uint32_t ep = 0; for (int t = 0; t < 3; t++) ep |= ((e >> (8 * ((hs[l] >> (8 * t)) & 3))) & 0xffu) << (8 * t); badp += pp[l] != ep; if (l < 4) printf("  perm sel %08x in %08x out %08x expect %08x\n", hs[l], e, pp[l], ep); }
        if (l < 4) printf("lane %d expect %08x global %08x buffer %08x  dot/sub %08x expect %08x\n", l, e, g[l], b[l], dd[l], ed);
    }
    printf("mismatches: global %d buffer %d dot4/sdwa %d perm %d\n", badg, badb, badd, badp);
    return 0;
}
