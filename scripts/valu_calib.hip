// valu_calib.hip — vector-ALU issue-rate calibration for gfx950 (measurement tool, not product code).
//
// For every opcode the kernels of this repository lean on, W = 1,2,3,4,6,8 waves per SIMD each run a loop of
// 16 INDEPENDENT instructions (own destination register each) and stamp s_memtime (shader cycles) around it.
// Waves also record HW_ID / XCC_ID, so the host groups them by physical SIMD and reports, for the SIMDs that
// really held W waves:   cycles per wave-instruction per SIMD = (max end - min start) / instructions issued there.
// lane-ops/clk/SIMD = 64 / that (x2 for packed opcodes).  The clock is read from s_memrealtime (100 MHz).
//
// build: hipcc --offload-arch=gfx950 -O3 -o valu_calib valu_calib.hip ; run: ./valu_calib > valu_calib.json
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

typedef float f2 __attribute__((ext_vector_type(2)));
struct Stamp { uint64_t t0, t1, r0, r1; uint32_t hwid, xcc; };

enum Op { FMA, FMAC, ADDF, MULF, FMA_SGPR, PKFMA, PKFMA_SGPR, PKADDF, PKMULF, PKADDU16, PKSUBI16, PKMAXI16, CVTUB0, CVTUB3, QSAD, SADU8, DOT4, PERM,
          ALIGNBIT, MADU24, MULLO, ADDU32, AND, LSHL, CNDMASK, MOV, MOVDPP, ADD3, LSHLADD, MAD64, BFE, MAX3F, SUBREVF, MINU32, CMPGT, NOPS, FMA_DEP, PKFMA_DEP, CND_SGPR, CND_VCC_FRESH, CMP_CND_PAIR, LSHR, OR, XOR, SUBU32, MAXF, MULU24, BFI, ANDOR, OR3, LSHLOR, CVTF32U32, CVTU32F32, RCP, FMA64, ADD64, MED3I, ADDCO, ASHR, PKMULLO16, PKMAD16, MAXI32, ABSDIFF, MUL_LIT, MUL_SGPR, FMAC_SGPR, FMAC_LIT, FMAMK, FMAAK, ADD_INL, FMA_NEG, FMA_INL, ADDU_SDWA, ADDF_SDWA, CVTI, ADD_SGPR, ADDU_SGPR, ADDU_LIT, MUL_ABS, S_ADD, S_AND64, S_SAVEEXEC, S_MUL, VCMP_SGPR, N_OPS };
static const char *op_names[N_OPS] = {"v_fma_f32", "v_fmac_f32", "v_add_f32", "v_mul_f32", "v_fma_f32(sgpr src)", "v_pk_fma_f32", "v_pk_fma_f32(sgpr src)", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_add_u16", "v_pk_sub_i16", "v_pk_max_i16", "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte3", "v_qsad_pk_u16_u8", "v_sad_u8", "v_dot4_u32_u8", "v_perm_b32",
                                      "v_alignbit_b32", "v_mad_u32_u24", "v_mul_lo_u32", "v_add_u32", "v_and_b32", "v_lshlrev_b32", "v_cndmask_b32", "v_mov_b32", "v_mov_b32 dpp row_shr:1", "v_add3_u32", "v_lshl_add_u32", "v_mad_u64_u32", "v_bfe_u32", "v_max3_f32", "v_subrev_f32", "v_min_u32", "v_cmp_gt_u32", "s_nop 0", "v_fma_f32 (1 dependent chain)", "v_pk_fma_f32 (1 dependent chain)", "v_cndmask_b32 (sgpr-pair mask)", "v_cndmask_b32 (vcc written by s_mov each 16)", "v_cmp_gt_u32 + v_cndmask_b32 (pair = 2 instr)", "v_lshrrev_b32", "v_or_b32", "v_xor_b32", "v_sub_u32", "v_max_f32", "v_mul_u32_u24", "v_bfi_b32", "v_and_or_b32", "v_or3_b32", "v_lshl_or_b32", "v_cvt_f32_u32", "v_cvt_u32_f32", "v_rcp_f32", "v_fma_f64", "v_add_f64", "v_med3_i32", "v_add_co_u32", "v_ashrrev_i32", "v_pk_mul_lo_u16", "v_pk_mad_u16", "v_max_i32", "v_sad_u32", "v_mul_f32 (literal src)", "v_mul_f32 (sgpr src)", "v_fmac_f32 (sgpr src)", "v_fmac_f32 (literal src)", "v_fmamk_f32", "v_fmaak_f32", "v_add_f32 (inline const)", "v_fma_f32 (neg modifier)", "v_fma_f32 (inline const src)", "v_add_u32_sdwa (byte selects)", "v_add_f32_sdwa", "v_cvt_f32_i32", "v_add_f32 (sgpr src)", "v_add_u32 (sgpr src)", "v_add_u32 (literal src)", "v_add_f32 (abs modifier, e64)", "s_add_u32", "s_and_b64", "s_and_saveexec_b64 + s_mov exec", "s_mul_i32", "v_cmp_gt_u32 (sgpr-pair dst, e64)"};
static const int op_lanes_mul[N_OPS] = {1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};

#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(1024) void k_calib(Stamp *st, const float *in, int iters)
{
    extern __shared__ char lds_pad[]; // only sizes the block's LDS footprint => blocks per CU
    float a[16]; f2 p[16]; uint32_t u[16]; uint64_t q[16];
    const float b = in[threadIdx.x & 7], c = in[8 + (threadIdx.x & 3)];
    const f2 b2 = {b, c}, c2 = {c, b};
    // wave-uniform operands pinned to scalar registers
    const uint32_t sbu = __builtin_amdgcn_readfirstlane(__float_as_uint(in[3])), scu = __builtin_amdgcn_readfirstlane(__float_as_uint(in[5]));
    const uint64_t sb2 = ((uint64_t)scu << 32) | sbu;
    const uint64_t q64 = 0x3ff0000000000001ull + threadIdx.x;
    uint32_t su[4] = {sbu, scu, sbu + 1, scu + 1};
    uint64_t sq[4] = {~0ull, ~0ull, ~0ull, ~0ull};
    const uint32_t ub = __float_as_uint(b) | 0x01020304u, uc = threadIdx.x * 2654435761u;
#pragma unroll
    for (int i = 0; i < 16; i++) { a[i] = in[i & 7] + threadIdx.x + i; p[i] = f2{a[i], a[i] + 1.f}; u[i] = uc + i * 977u; q[i] = ((uint64_t)u[i] << 32) | ub; }
    if (lds_pad[0] == 77 && iters < 0) a[0] += 1.f; // keep the LDS symbol alive
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < iters; it++) {
#define X_FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
#define X_FMAC(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define X_ADDF(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define X_MULF(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define X_FMAS(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(sbu), "v"(c));
#define X_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(b2), "v"(c2));
#define X_PKFMAS(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(b2), "s"(sb2));
#define X_PKADDF(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(b2));
#define X_PKMULF(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(b2));
#define X_PKADDU16(i) asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(u[i]) : "v"(ub));
#define X_PKSUBI16(i) asm volatile("v_pk_sub_i16 %0, %1, %0" : "+v"(u[i]) : "v"(ub));
#define X_PKMAXI16(i) asm volatile("v_pk_max_i16 %0, %1, %0" : "+v"(u[i]) : "v"(ub));
#define X_CVT0(i) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a[i]) : "v"(u[i]));
#define X_CVT3(i) asm volatile("v_cvt_f32_ubyte3 %0, %1" : "=v"(a[i]) : "v"(u[i]));
#define X_QSAD(i) asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(q[i]) : "v"(q[(i + 1) & 15]), "v"(ub));
#define X_SAD(i) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(u[i]) : "v"(ub), "v"(uc));
#define X_DOT4(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(u[i]) : "v"(ub), "v"(uc));
#define X_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ub), "v"(uc));
#define X_ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %1, 8" : "+v"(u[i]) : "v"(ub));
#define X_MADU24(i) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(u[i]) : "v"(ub), "v"(uc));
#define X_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
#define X_ADDU(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[i]) : "v"(ub));
#define X_AND(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[i]) : "v"(ub));
#define X_LSHL(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));
#define X_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(ub) : );
#define X_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(ub));
#define X_MOVDPP(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(ub));
#define X_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ub), "v"(uc));
#define X_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(ub));
#define X_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(ub), "v"(uc) : "vcc");
#define X_BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(u[i]));
#define X_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define X_SUBREV(i) asm volatile("v_subrev_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define X_MINU(i) asm volatile("v_min_u32 %0, %1, %0" : "+v"(u[i]) : "v"(ub));
#define X_CMP(i) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(u[i]), "v"(ub) : "vcc");
#define X_NOP(i) asm volatile("s_nop 0");
#define X_CNDS(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ub), "s"(sb2));
#define X_CNDF(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(ub) : );
#define X_CMPCND(i) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(ub) : "vcc");
#define X_LSHR(i) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(u[i]));
#define X_OR(i) asm volatile("v_or_b32 %0, %1, %0" : "+v"(u[i]) : "v"(ub));
#define X_XOR(i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(u[i]) : "v"(ub));
#define X_SUBU(i) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(u[i]) : "v"(ub));
#define X_MAXF(i) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define X_MULU24(i) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(u[i]) : "v"(ub));
#define X_BFI(i) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(u[i]) : "v"(ub), "v"(uc));
#define X_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ub), "v"(uc));
#define X_OR3(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ub), "v"(uc));
#define X_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(ub));
#define X_CVTFU(i) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(u[i]));
#define X_CVTUF(i) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(u[i]) : "v"(a[i]));
#define X_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define X_FMA64(i) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(q[i]) : "v"(q64));
#define X_ADD64(i) asm volatile("v_add_f64 %0, %1, %0" : "+v"(q[i]) : "v"(q64));
#define X_MED3(i) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ub), "v"(uc));
#define X_ADDCO(i) asm volatile("v_add_co_u32 %0, vcc, %1, %0" : "+v"(u[i]) : "v"(ub) : "vcc");
#define X_ASHR(i) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(u[i]));
#define X_PKMULLO(i) asm volatile("v_pk_mul_lo_u16 %0, %1, %0" : "+v"(u[i]) : "v"(ub));
#define X_PKMAD(i) asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(u[i]) : "v"(ub), "v"(uc));
#define X_MAXI(i) asm volatile("v_max_i32 %0, %1, %0" : "+v"(u[i]) : "v"(ub));
#define X_SADU32(i) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(u[i]) : "v"(ub), "v"(uc));
#define X_MULLIT(i) asm volatile("v_mul_f32 %0, 0x3dc7c5c2, %0" : "+v"(a[i]));
#define X_MULS(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sbu));
#define X_FMACS(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(sbu), "v"(c));
#define X_FMACL(i) asm volatile("v_fmac_f32 %0, 0x3dc7c5c2, %1" : "+v"(a[i]) : "v"(c));
#define X_FMAMK(i) asm volatile("v_fmamk_f32 %0, %1, 0x3dc7c5c2, %0" : "+v"(a[i]) : "v"(c));
#define X_FMAAK(i) asm volatile("v_fmaak_f32 %0, %1, %0, 0x3dc7c5c2" : "+v"(a[i]) : "v"(c));
#define X_ADDINL(i) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(a[i]));
#define X_FMANEG(i) asm volatile("v_fma_f32 %0, %1, %2, -%0" : "+v"(a[i]) : "v"(b), "v"(c));
#define X_FMAINL(i) asm volatile("v_fma_f32 %0, %1, 2.0, %0" : "+v"(a[i]) : "v"(b));
#define X_ADDUSDWA(i) asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_3" : "+v"(u[i]) : "v"(ub));
#define X_ADDFSDWA(i) asm volatile("v_add_f32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(a[i]) : "v"(b));
#define X_CVTI(i) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[i]) : "v"(u[i]));
#define X_ADDS(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sbu));
#define X_ADDUS(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[i]) : "s"(sbu));
#define X_ADDUL(i) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(u[i]));
#define X_MULABS(i) asm volatile("v_add_f32_e64 %0, |%1|, %0" : "+v"(a[i]) : "v"(b));
#define X_SADD(i) asm volatile("s_add_u32 %0, %0, 3" : "+s"(su[i & 3]) : : "scc");
#define X_SAND(i) asm volatile("s_and_b64 %0, %0, %1" : "+s"(sq[i & 3]) : "s"(sb2) : "scc");
#define X_SSAVE(i) asm volatile("s_and_saveexec_b64 %0, %1\n\ts_mov_b64 exec, %0" : "=&s"(sq[i & 3]) : "s"(sb2) : "scc", "exec");
#define X_SMUL(i) asm volatile("s_mul_i32 %0, %0, 3" : "+s"(su[i & 3]));
#define X_VCMPS(i) asm volatile("v_cmp_gt_u32_e64 %0, %1, %2" : "=s"(sq[i & 3]) : "v"(u[i]), "v"(ub));
#define X_FMADEP(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[0]) : "v"(b), "v"(c));
#define X_PKFMADEP(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[0]) : "v"(b2), "v"(c2));
        if (OP == FMA) { R16(X_FMA) } else if (OP == FMAC) { R16(X_FMAC) } else if (OP == ADDF) { R16(X_ADDF) }
        else if (OP == MULF) { R16(X_MULF) } else if (OP == FMA_SGPR) { R16(X_FMAS) } else if (OP == PKFMA) { R16(X_PKFMA) }
        else if (OP == PKFMA_SGPR) { R16(X_PKFMAS) } else if (OP == PKADDF) { R16(X_PKADDF) } else if (OP == PKMULF) { R16(X_PKMULF) }
        else if (OP == PKADDU16) { R16(X_PKADDU16) } else if (OP == PKSUBI16) { R16(X_PKSUBI16) } else if (OP == PKMAXI16) { R16(X_PKMAXI16) }
        else if (OP == CVTUB0) { R16(X_CVT0) } else if (OP == CVTUB3) { R16(X_CVT3) } else if (OP == QSAD) { R16(X_QSAD) }
        else if (OP == SADU8) { R16(X_SAD) } else if (OP == DOT4) { R16(X_DOT4) } else if (OP == PERM) { R16(X_PERM) }
        else if (OP == ALIGNBIT) { R16(X_ALIGN) } else if (OP == MADU24) { R16(X_MADU24) } else if (OP == MULLO) { R16(X_MULLO) }
        else if (OP == ADDU32) { R16(X_ADDU) } else if (OP == AND) { R16(X_AND) } else if (OP == LSHL) { R16(X_LSHL) }
        else if (OP == CNDMASK) { R16(X_CND) } else if (OP == MOV) { R16(X_MOV) } else if (OP == MOVDPP) { R16(X_MOVDPP) }
        else if (OP == ADD3) { R16(X_ADD3) } else if (OP == LSHLADD) { R16(X_LSHLADD) } else if (OP == MAD64) { R16(X_MAD64) }
        else if (OP == BFE) { R16(X_BFE) } else if (OP == MAX3F) { R16(X_MAX3) } else if (OP == SUBREVF) { R16(X_SUBREV) }
        else if (OP == MINU32) { R16(X_MINU) } else if (OP == CMPGT) { R16(X_CMP) } else if (OP == NOPS) { R16(X_NOP) }
        else if (OP == FMA_DEP) { R16(X_FMADEP) } else if (OP == PKFMA_DEP) { R16(X_PKFMADEP) }
        else if (OP == CND_SGPR) { R16(X_CNDS) } else if (OP == CND_VCC_FRESH) { asm volatile("s_mov_b64 vcc, %0" : : "s"(sb2) : "vcc"); R16(X_CNDF) }
        else if (OP == CMP_CND_PAIR) { R16(X_CMPCND) } else if (OP == LSHR) { R16(X_LSHR) } else if (OP == OR) { R16(X_OR) }
        else if (OP == XOR) { R16(X_XOR) } else if (OP == SUBU32) { R16(X_SUBU) } else if (OP == MAXF) { R16(X_MAXF) }
        else if (OP == MULU24) { R16(X_MULU24) } else if (OP == BFI) { R16(X_BFI) } else if (OP == ANDOR) { R16(X_ANDOR) }
        else if (OP == OR3) { R16(X_OR3) } else if (OP == LSHLOR) { R16(X_LSHLOR) } else if (OP == CVTF32U32) { R16(X_CVTFU) }
        else if (OP == CVTU32F32) { R16(X_CVTUF) } else if (OP == RCP) { R16(X_RCP) } else if (OP == FMA64) { R16(X_FMA64) }
        else if (OP == ADD64) { R16(X_ADD64) } else if (OP == MED3I) { R16(X_MED3) } else if (OP == ADDCO) { R16(X_ADDCO) }
        else if (OP == ASHR) { R16(X_ASHR) } else if (OP == PKMULLO16) { R16(X_PKMULLO) } else if (OP == PKMAD16) { R16(X_PKMAD) }
        else if (OP == MAXI32) { R16(X_MAXI) } else if (OP == ABSDIFF) { R16(X_SADU32) }
        else if (OP == MUL_LIT) { R16(X_MULLIT) } else if (OP == MUL_SGPR) { R16(X_MULS) } else if (OP == FMAC_SGPR) { R16(X_FMACS) }
        else if (OP == FMAC_LIT) { R16(X_FMACL) } else if (OP == FMAMK) { R16(X_FMAMK) } else if (OP == FMAAK) { R16(X_FMAAK) }
        else if (OP == ADD_INL) { R16(X_ADDINL) } else if (OP == FMA_NEG) { R16(X_FMANEG) } else if (OP == FMA_INL) { R16(X_FMAINL) }
        else if (OP == ADDU_SDWA) { R16(X_ADDUSDWA) } else if (OP == ADDF_SDWA) { R16(X_ADDFSDWA) } else if (OP == CVTI) { R16(X_CVTI) }
        else if (OP == ADD_SGPR) { R16(X_ADDS) } else if (OP == ADDU_SGPR) { R16(X_ADDUS) } else if (OP == ADDU_LIT) { R16(X_ADDUL) }
        else if (OP == MUL_ABS) { R16(X_MULABS) }
        else if (OP == S_ADD) { R16(X_SADD) } else if (OP == S_AND64) { R16(X_SAND) } else if (OP == S_SAVEEXEC) { R16(X_SSAVE) }
        else if (OP == S_MUL) { R16(X_SMUL) } else if (OP == VCMP_SGPR) { R16(X_VCMPS) }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = 0; uint32_t us = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { // only the register file the opcode works on stays live (<= 64 VGPRs => 8 waves/SIMD fit)
        constexpr bool is_pk = OP == PKFMA || OP == PKFMA_SGPR || OP == PKADDF || OP == PKMULF || OP == PKFMA_DEP;
        constexpr bool is_q = OP == QSAD || OP == MAD64 || OP == FMA64 || OP == ADD64;
        constexpr bool is_f = OP == FMA || OP == FMAC || OP == ADDF || OP == MULF || OP == FMA_SGPR || OP == MAX3F || OP == SUBREVF || OP == FMA_DEP || OP == MAXF || OP == RCP || OP == MUL_LIT || OP == MUL_SGPR || OP == FMAC_SGPR || OP == FMAC_LIT || OP == FMAMK || OP == FMAAK || OP == ADD_INL || OP == FMA_NEG || OP == FMA_INL || OP == ADDF_SDWA || OP == ADD_SGPR || OP == MUL_ABS;
        constexpr bool is_cvt = OP == CVTUB0 || OP == CVTUB3 || OP == CVTF32U32 || OP == CVTU32F32 || OP == CVTI;
        if (is_pk) s += p[i].x + p[i].y;
        else if (is_q) us += (uint32_t)q[i] + (uint32_t)(q[i] >> 32);
        else if (is_f) s += a[i];
        else if (is_cvt) { s += a[i]; us += u[i]; }
        else us += u[i];
    }
    uint32_t hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) {
        Stamp &o = st[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)];
        o.t0 = t0; o.t1 = t1; o.r0 = r0; o.r1 = r1; o.hwid = hwid; o.xcc = xcc;
    }
    if (s == 1.2345f && us == 77 + su[0] + su[1] + su[2] + su[3] + (uint32_t)(sq[0] ^ sq[1] ^ sq[2] ^ sq[3])) st[0].t0 = 0; // keep results live
}

typedef void (*kfn)(Stamp *, const float *, int);
template <int OP> struct Tab { static void fill(kfn *t) { t[OP] = k_calib<OP>; Tab<OP + 1>::fill(t); } };
template <> struct Tab<N_OPS> { static void fill(kfn *) {} };

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    kfn tab[N_OPS]; Tab<0>::fill(tab);
    float *in; float hin[16];
    for (int i = 0; i < 16; i++) hin[i] = 1.0f + i * 1e-3f;
    Stamp *st; const int max_blocks = 256 * 8; // in units of 4 waves
    CHK(hipMalloc(&st, sizeof(Stamp) * max_blocks * 4)); CHK(hipMalloc(&in, 64)); CHK(hipMemcpy(in, hin, 64, hipMemcpyHostToDevice));
    std::vector<Stamp> h(max_blocks * 4);
    const int wlist[] = {1, 2, 3, 4, 6, 8};
    printf("{\"tool\": \"scripts/valu_calib.hip\", \"iters\": %d, \"instr_per_iter\": 16,\n \"method\": \"s_memtime around a loop of 16 independent instructions; waves grouped by physical SIMD (HW_ID, XCC_ID); "
           "cycles per wave-instruction per SIMD = (max end - min start)/(instructions issued on that SIMD), median over SIMDs that held exactly W waves\",\n \"ops\": {\n", iters);
    for (int op = 0; op < N_OPS; op++) {
        printf("  \"%s\": {", op_names[op]);
        for (int wi = 0; wi < 6; wi++) {
            const int W = wlist[wi];
            const int bpc = W > 4 ? 2 : 1, threads = 256 * W / bpc; // blocks per CU, threads per block
            const size_t lds = (160 * 1024 / bpc) & ~1023;           // LDS footprint pins bpc workgroups on a CU
            CHK(hipFuncSetAttribute((const void *)tab[op], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int grid = 256 * bpc, nw = threads / 64;
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(tab[op], dim3(grid), dim3(threads), lds, 0, st, in, iters);
                CHK(hipDeviceSynchronize());
            }
            CHK(hipMemcpy(h.data(), st, sizeof(Stamp) * grid * nw, hipMemcpyDeviceToHost));
            struct Acc { uint64_t t0 = ~0ull, t1 = 0; int n = 0; };
            std::map<uint32_t, Acc> simd;
            double clk_sum = 0;
            std::vector<double> own;
            for (int i = 0; i < grid * nw; i++) {
                const Stamp &s = h[i];
                // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
                uint32_t key = ((s.xcc & 0xf) << 16) | (s.hwid & 0xff30);
                Acc &a = simd[key]; a.t0 = std::min(a.t0, s.t0); a.t1 = std::max(a.t1, s.t1); a.n++;
                clk_sum += (double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 100.0; // MHz
                own.push_back((double)(s.t1 - s.t0) / ((double)iters * 16));
            }
            std::vector<double> cyc; int other = 0;
            for (auto &kv : simd) {
                if (kv.second.n != W) { other++; continue; }
                cyc.push_back((double)(kv.second.t1 - kv.second.t0) / ((double)W * iters * 16));
            }
            double med = -1;
            if (!cyc.empty()) { std::sort(cyc.begin(), cyc.end()); med = cyc[cyc.size() / 2]; }
            std::sort(own.begin(), own.end());
            printf("%s\"%d\": {\"wave_own_cyc_per_instr\": %.3f, \"cyc_per_wave_instr\": %.3f, \"lane_ops_per_clk_simd\": %.2f, \"simds_with_W\": %d, \"simds_other\": %d, \"clock_MHz\": %.0f}",
                   wi ? ", " : "", W, own[own.size() / 2], med, med > 0 ? 64.0 * op_lanes_mul[op] / med : 0.0, (int)cyc.size(), other, clk_sum / (grid * nw));
        }
        printf("}%s\n", op + 1 < N_OPS ? "," : "");
    }
    printf(" }\n}\n");
    return 0;
}
