// valu_rate.hip — what does one vector instruction cost on gfx950 when several
// waves share a SIMD?  Each kernel runs a long stream of one instruction kind
// (32 per loop trip, independent unless the name says "dep"), on W waves per
// SIMD on every CU; the table gives SIMD cycles per wave instruction at the
// measured shader clock, i.e. the reciprocal issue rate of the whole SIMD.
// Diagnostic only; not part of the library.
//
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o raytracer_amd/lib/valu_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                             \
  do {                                                                    \
    hipError_t e_ = (x);                                                  \
    if (e_ != hipSuccess) {                                               \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                            \
    }                                                                     \
  } while (0)

// sixteen destinations v[10..41] in pairs; sources v[2:7] hold ordinary numbers
#define D16(OP)                                                                                               \
  OP("10", "11") OP("12", "13") OP("14", "15") OP("16", "17") OP("18", "19") OP("20", "21") OP("22", "23")    \
      OP("24", "25") OP("26", "27") OP("28", "29") OP("30", "31") OP("32", "33") OP("34", "35") OP("36", "37") \
          OP("38", "39") OP("40", "41")

#define CLOB                                                                                                      \
  "v2", "v3", "v4", "v5", "v6", "v7", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", \
      "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35",    \
      "v36", "v37", "v38", "v39", "v40", "v41", "s20", "s21", "s22", "s23", "vcc", "scc"

#define SETUP                                                                                 \
  asm volatile(                                                                               \
      "s_mov_b64 vcc, exec\n"                                                                \
      "v_mov_b32 v2, 1.0\n v_mov_b32 v3, 0.5\n v_mov_b32 v4, 2.0\n v_mov_b32 v5, 4.0\n"       \
      "v_mov_b32 v6, 0.5\n v_mov_b32 v7, 1.0\n s_mov_b64 s[20:21], exec\n" ::                 \
          : CLOB)

// two instructions per OP so that every kind issues 32 per trip
#define OP_FMA(a, b) "v_fma_f32 v" a ", v2, v3, v4\n v_fma_f32 v" b ", v5, v6, v7\n"
#define OP_MUL(a, b) "v_mul_f32 v" a ", v2, v3\n v_mul_f32 v" b ", v5, v6\n"
#define OP_MOV(a, b) "v_mov_b32 v" a ", v2\n v_mov_b32 v" b ", v5\n"
#define OP_PKFMA(a, b) "v_pk_fma_f32 v[" a ":" b "], v[2:3], v[4:5], v[6:7]\n v_pk_fma_f32 v[" a ":" b "], v[4:5], v[6:7], v[2:3]\n"
#define OP_PKFMA_SEL(a, b)                                                                  \
  "v_pk_fma_f32 v[" a ":" b "], v[2:3], v[4:5], v[6:7] op_sel:[1,0,0] op_sel_hi:[0,1,1]\n" \
  "v_pk_fma_f32 v[" a ":" b "], v[4:5], v[6:7], v[2:3] op_sel:[1,0,0] op_sel_hi:[0,1,1]\n"
#define OP_PKMUL(a, b) "v_pk_mul_f32 v[" a ":" b "], v[2:3], v[4:5]\n v_pk_mul_f32 v[" a ":" b "], v[4:5], v[6:7]\n"
#define OP_PKADD(a, b) "v_pk_add_f32 v[" a ":" b "], v[2:3], v[4:5]\n v_pk_add_f32 v[" a ":" b "], v[4:5], v[6:7]\n"
#define OP_CNDMASK(a, b) "v_cndmask_b32_e64 v" a ", v2, v3, s[20:21]\n v_cndmask_b32_e64 v" b ", v5, v6, s[20:21]\n"
#define OP_MAX3(a, b) "v_max3_f32 v" a ", v2, v3, v4\n v_max3_f32 v" b ", v5, v6, v7\n"
#define OP_CMP(a, b) "v_cmp_le_f32_e64 s[22:23], v2, v3\n v_cmp_le_f32_e64 s[22:23], v5, v6\n"
#define OP_RCP(a, b) "v_rcp_f32 v" a ", v2\n v_rcp_f32 v" b ", v5\n"
#define OP_INT(a, b) "v_lshl_add_u32 v" a ", v2, 5, v3\n v_and_b32 v" b ", v5, v6\n"
#define OP_MAD24(a, b) "v_mad_u32_u24 v" a ", v2, 48, v3\n v_mad_u32_u24 v" b ", v5, 48, v6\n"
#define OP_ADD(a, b) "v_add_f32 v" a ", v2, v3\n v_sub_f32 v" b ", v5, v6\n"
#define OP_MINMAX(a, b) "v_max_f32 v" a ", v2, v3\n v_min_f32 v" b ", v5, v6\n"
#define OP_FMAC(a, b) "v_fmac_f32 v" a ", v2, v3\n v_fmac_f32 v" b ", v5, v6\n"
#define OP_FMA_NEG(a, b) "v_fma_f32 v" a ", -v2, v3, v4\n v_fma_f32 v" b ", v5, v6, 1.0\n"
#define OP_FMA_SGPR(a, b) "v_fma_f32 v" a ", s20, v3, v4\n v_mul_f32 v" b ", s21, v6\n"
#define OP_CND32(a, b) "v_cndmask_b32_e32 v" a ", v2, v3, vcc\n v_cndmask_b32_e32 v" b ", v5, v6, vcc\n"
#define OP_CND_CONST(a, b) "v_cndmask_b32_e64 v" a ", 0, 1, s[20:21]\n v_cndmask_b32_e64 v" b ", 0, 16, s[20:21]\n"
#define OP_CMP32(a, b) "v_cmp_le_f32_e32 vcc, v2, v3\n v_cmp_ne_u32_e32 vcc, v5, v6\n"
#define OP_CMPU64(a, b) "v_cmp_eq_u32_e64 s[22:23], 0, v3\n v_cmp_lt_u32_e64 s[22:23], s20, v6\n"
#define OP_AND_OR(a, b) "v_and_b32 v" a ", v2, v3\n v_or_b32 v" b ", v5, v6\n"
#define OP_ADDU(a, b) "v_add_u32 v" a ", v2, v3\n v_add_u32 v" b ", -1, v6\n"
#define OP_SHIFT(a, b) "v_lshlrev_b32 v" a ", 5, v3\n v_lshrrev_b32 v" b ", 24, v6\n"
#define OP_LSHLADD(a, b) "v_lshl_add_u32 v" a ", v2, 5, v3\n v_lshl_add_u32 v" b ", v5, 5, v6\n"
#define OP_OR3(a, b) "v_or3_b32 v" a ", v2, v3, v4\n v_and_or_b32 v" b ", v5, v6, v7\n"
#define OP_BFE(a, b) "v_bfe_u32 v" a ", v2, 24, 8\n v_bfe_u32 v" b ", v5, 24, 8\n"
#define OP_MED3(a, b) "v_med3_f32 v" a ", v2, v3, v4\n v_med3_f32 v" b ", v5, v6, v7\n"
#define OP_READLANE(a, b) "v_readlane_b32 s22, v2, 3\n v_readlane_b32 s23, v5, 7\n"
#define OP_DPP(a, b) "v_mov_b32_dpp v" a ", v2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v" b ", v5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define OP_BPERM(a, b) "ds_bpermute_b32 v" a ", v2, v3\n ds_bpermute_b32 v" b ", v5, v6\n"
#define OP_MBCNT(a, b) "v_mbcnt_lo_u32_b32 v" a ", s20, 0\n v_mbcnt_hi_u32_b32 v" b ", s21, v6\n"
#define OP_MIX_FAST_SLOW(a, b) "v_fma_f32 v" a ", v2, v3, v4\n v_max3_f32 v" b ", v5, v6, v7\n"
#define OP_SAVEEXEC(a, b) "s_and_saveexec_b64 s[22:23], s[20:21]\n s_or_b64 exec, exec, s[22:23]\n"
#define OP_FMA_SAVEEXEC(a, b) "v_fma_f32 v" a ", v2, v3, v4\n s_and_saveexec_b64 s[22:23], s[20:21]\n v_fma_f32 v" b ", v5, v6, v7\n s_or_b64 exec, exec, s[22:23]\n"
#define OP_CND64VCC(a, b) "v_cndmask_b32_e64 v" a ", v2, v3, vcc\n v_cndmask_b32_e64 v" b ", v5, v6, vcc\n"
#define OP_CMP_CND(a, b) "v_cmp_lt_u32_e32 vcc, v2, v3\n v_cndmask_b32_e32 v" a ", v5, v6, vcc\n"
#define OP_MINDPP(a, b) "v_min_u32_dpp v" a ", v2, v3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp v" b ", v5, v6 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define OP_CMPLTU64(a, b) "v_cmp_lt_u64_e32 vcc, v[2:3], v[4:5]\n v_cmp_lt_u64_e32 vcc, v[4:5], v[6:7]\n"
#define OP_MINU(a, b) "v_min_u32 v" a ", v2, v3\n v_max_u32 v" b ", v5, v6\n"
#define OP_SWIZZLE(a, b) "ds_swizzle_b32 v" a ", v2 offset:swizzle(SWAP,1)\n ds_swizzle_b32 v" b ", v5 offset:swizzle(SWAP,16)\n"
#define OP_PERMLANE32(a, b) "v_permlane32_swap_b32 v" a ", v" b "\n v_permlane16_swap_b32 v" a ", v" b "\n"
#define OP_READFIRST(a, b) "v_readfirstlane_b32 s22, v2\n v_readfirstlane_b32 s23, v5\n"
#define OP_MIX_FSS(a, b) "v_fma_f32 v" a ", v2, v3, v4\n v_max3_f32 v" b ", v5, v6, v7\n v_cndmask_b32_e64 v" a ", v5, v6, s[20:21]\n"
#define OP_MIX_FSSS(a, b) "v_fma_f32 v" a ", v2, v3, v4\n v_max3_f32 v" b ", v5, v6, v7\n v_cndmask_b32_e64 v" a ", v5, v6, s[20:21]\n v_lshlrev_b32 v" b ", 5, v3\n"
#define OP_MIX_FS_SALU(a, b) "v_fma_f32 v" a ", v2, v3, v4\n v_max3_f32 v" b ", v5, v6, v7\n s_and_b64 s[22:23], s[20:21], exec\n"
#define OP_MIX_FFSS_SALU(a, b) "v_fma_f32 v" a ", v2, v3, v4\n v_max3_f32 v" b ", v5, v6, v7\n s_and_b64 s[22:23], s[20:21], exec\n v_mul_f32 v" a ", v2, v3\n v_cmp_le_f32_e64 s[22:23], v2, v3\n"
#define OP_FMA_DEP(a, b) "v_fma_f32 v10, v10, v3, v4\n v_fma_f32 v10, v10, v6, v7\n"
#define OP_PKFMA_DEP(a, b) "v_pk_fma_f32 v[10:11], v[10:11], v[4:5], v[6:7]\n v_pk_fma_f32 v[10:11], v[10:11], v[6:7], v[2:3]\n"
#define OP_FMA_SALU(a, b) "v_fma_f32 v" a ", v2, v3, v4\n s_and_b64 s[22:23], s[20:21], exec\n"
#define OP_FMA_2SALU(a, b) "v_fma_f32 v" a ", v2, v3, v4\n s_and_b64 s[22:23], s[20:21], exec\n s_or_b64 s[22:23], s[20:21], exec\n"
#define OP_SALU(a, b) "s_and_b64 s[22:23], s[20:21], exec\n s_or_b64 s[22:23], s[20:21], exec\n"
#define OP_DIV(a, b)                                                        \
  "v_div_scale_f32 v" a ", s[22:23], v2, v2, 1.0\n v_rcp_f32 v" b ", v" a "\n" \
  "v_div_fmas_f32 v" a ", v2, v3, v4\n v_div_fixup_f32 v" b ", v2, v3, 1.0\n"

#define KERNEL(NAME, OP)                                                              \
  __global__ __launch_bounds__(256) void NAME(uint32_t trips, uint32_t *out) {        \
    SETUP;                                                                            \
    for (uint32_t i = 0; i < trips; ++i) asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 vcc, exec\n" D16(OP)::: CLOB);               \
    if (trips == 0xFFFFFFFFu) out[threadIdx.x] = 1u;                                  \
  }

KERNEL(k_fma, OP_FMA)
KERNEL(k_mul, OP_MUL)
KERNEL(k_mov, OP_MOV)
KERNEL(k_pkfma, OP_PKFMA)
KERNEL(k_pkfma_sel, OP_PKFMA_SEL)
KERNEL(k_pkmul, OP_PKMUL)
KERNEL(k_pkadd, OP_PKADD)
KERNEL(k_cndmask, OP_CNDMASK)
KERNEL(k_max3, OP_MAX3)
KERNEL(k_cmp, OP_CMP)
KERNEL(k_rcp, OP_RCP)
KERNEL(k_int, OP_INT)
KERNEL(k_mad24, OP_MAD24)
KERNEL(k_fma_dep, OP_FMA_DEP)
KERNEL(k_pkfma_dep, OP_PKFMA_DEP)
KERNEL(k_fma_salu, OP_FMA_SALU)
KERNEL(k_fma_2salu, OP_FMA_2SALU)
KERNEL(k_salu, OP_SALU)
KERNEL(k_add, OP_ADD)
KERNEL(k_minmax, OP_MINMAX)
KERNEL(k_fmac, OP_FMAC)
KERNEL(k_fma_neg, OP_FMA_NEG)
KERNEL(k_fma_sgpr, OP_FMA_SGPR)
KERNEL(k_cnd32, OP_CND32)
KERNEL(k_cnd_const, OP_CND_CONST)
KERNEL(k_cmp32, OP_CMP32)
KERNEL(k_cmpu64, OP_CMPU64)
KERNEL(k_and_or, OP_AND_OR)
KERNEL(k_addu, OP_ADDU)
KERNEL(k_shift, OP_SHIFT)
KERNEL(k_lshladd, OP_LSHLADD)
KERNEL(k_or3, OP_OR3)
KERNEL(k_bfe, OP_BFE)
KERNEL(k_med3, OP_MED3)
KERNEL(k_readlane, OP_READLANE)
KERNEL(k_dpp, OP_DPP)
KERNEL(k_bperm, OP_BPERM)
KERNEL(k_mbcnt, OP_MBCNT)
KERNEL(k_mix_fast_slow, OP_MIX_FAST_SLOW)
KERNEL(k_saveexec, OP_SAVEEXEC)
KERNEL(k_fma_saveexec, OP_FMA_SAVEEXEC)
KERNEL(k_cnd64vcc, OP_CND64VCC)
KERNEL(k_cmp_cnd, OP_CMP_CND)
KERNEL(k_mindpp, OP_MINDPP)
KERNEL(k_cmpltu64, OP_CMPLTU64)
KERNEL(k_minu, OP_MINU)
KERNEL(k_swizzle, OP_SWIZZLE)
KERNEL(k_permlane32, OP_PERMLANE32)
KERNEL(k_readfirst, OP_READFIRST)
KERNEL(k_mix_fss, OP_MIX_FSS)
KERNEL(k_mix_fsss, OP_MIX_FSSS)
KERNEL(k_mix_fs_salu, OP_MIX_FS_SALU)
KERNEL(k_mix_ffss_salu, OP_MIX_FFSS_SALU)
KERNEL(k_div, OP_DIV)

// the cycle counter against the 100 MHz wall clock
__global__ void k_clock(unsigned long long *out) {
  const unsigned long long r0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  unsigned long long w1 = w0;
  while (w1 - w0 < 1000000ull) w1 = wall_clock64();  // 10 ms at 100 MHz
  const unsigned long long r1 = __builtin_readcyclecounter();
  out[0] = r1 - r0;
  out[1] = w1 - w0;
}

struct Kind {
  const char *name;
  void (*fn)(uint32_t, uint32_t *);
  int per_trip;
};

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  uint32_t *out;
  unsigned long long *clk;
  CK(hipMalloc(&out, 4096));
  CK(hipMalloc(&clk, 16));
  k_clock<<<1, 1>>>(clk);
  unsigned long long hclk[2];
  CK(hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost));
  const double ghz = (double)hclk[0] / ((double)hclk[1] / 100e6) / 1e9;
  printf("# %s, %d CUs, cycle counter %.3f GHz (clockRate %.3f GHz)\n", prop.name, cus, ghz, prop.clockRate / 1e6);
  const double use_ghz = prop.clockRate / 1e6;
  const Kind kinds[] = {{"v_fma_f32", k_fma, 32},          {"v_mul_f32", k_mul, 32},
                        {"v_mov_b32", k_mov, 32},          {"v_pk_fma_f32", k_pkfma, 32},
                        {"v_pk_fma_f32 op_sel", k_pkfma_sel, 32}, {"v_pk_mul_f32", k_pkmul, 32},
                        {"v_pk_add_f32", k_pkadd, 32},     {"v_cndmask_b32 (sgpr mask)", k_cndmask, 32},
                        {"v_max3_f32", k_max3, 32},        {"v_cmp_f32 -> sgpr", k_cmp, 32},
                        {"v_rcp_f32", k_rcp, 32},          {"v_lshl_add / v_and", k_int, 32},
                        {"v_mad_u32_u24", k_mad24, 32},    {"v_fma_f32 dep chain", k_fma_dep, 32},
                        {"v_pk_fma_f32 dep chain", k_pkfma_dep, 32},
                        {"v_fma + s_and (per pair)", k_fma_salu, 16},
                        {"v_fma + 2 salu (per triple)", k_fma_2salu, 16},
                        {"s_and/s_or", k_salu, 32},
                        {"v_add_f32 / v_sub_f32", k_add, 32},
                        {"v_max_f32 / v_min_f32", k_minmax, 32},
                        {"v_fmac_f32 (VOP2)", k_fmac, 32},
                        {"v_fma_f32 neg / const operand", k_fma_neg, 32},
                        {"v_fma/v_mul with an SGPR operand", k_fma_sgpr, 32},
                        {"v_cndmask_b32_e32 (vcc)", k_cnd32, 32},
                        {"v_cndmask_b32_e64 0,const,sgpr", k_cnd_const, 32},
                        {"v_cmp_*_e32 -> vcc", k_cmp32, 32},
                        {"v_cmp_*_u32_e64 -> sgpr", k_cmpu64, 32},
                        {"v_and_b32 / v_or_b32", k_and_or, 32},
                        {"v_add_u32", k_addu, 32},
                        {"v_lshlrev / v_lshrrev", k_shift, 32},
                        {"v_lshl_add_u32", k_lshladd, 32},
                        {"v_or3_b32 / v_and_or_b32", k_or3, 32},
                        {"v_bfe_u32", k_bfe, 32},
                        {"v_med3_f32", k_med3, 32},
                        {"v_readlane_b32", k_readlane, 32},
                        {"v_mov_b32 dpp", k_dpp, 32},
                        {"ds_bpermute_b32", k_bperm, 32},
                        {"v_mbcnt_lo/hi", k_mbcnt, 32},
                        {"v_fma + v_max3 alternating", k_mix_fast_slow, 32},
                        {"s_and_saveexec + s_or exec (pair)", k_saveexec, 16},
                        {"2 v_fma + saveexec/restore (per 4)", k_fma_saveexec, 16},
                        {"v_cndmask_b32_e64 with vcc", k_cnd64vcc, 32},
                        {"v_cmp_e32 + v_cndmask_e32 (per pair)", k_cmp_cnd, 16},
                        {"v_min_u32_dpp row_shr / row_bcast", k_mindpp, 32},
                        {"v_cmp_lt_u64", k_cmpltu64, 32},
                        {"v_min_u32 / v_max_u32", k_minu, 32},
                        {"ds_swizzle_b32", k_swizzle, 32},
                        {"v_permlane32_swap / 16_swap", k_permlane32, 32},
                        {"v_readfirstlane_b32", k_readfirst, 32},
                        {"fast+2 slow (per 3)", k_mix_fss, 16},
                        {"fast+3 slow (per 4)", k_mix_fsss, 16},
                        {"fast+slow+salu (per 3)", k_mix_fs_salu, 16},
                        {"2 fast+2 slow+salu (per 5)", k_mix_ffss_salu, 16},
                        {"div_scale+rcp+fmas+fixup (per 4)", k_div, 16}};
  const int waves[] = {1, 2, 3, 4, 5, 8};
  printf("%-36s", "SIMD cycles per wave instruction, W =");
  for (int w : waves) printf("%8d", w);
  printf("\n");
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (const Kind &k : kinds) {
    printf("%-36s", k.name);
    for (int w : waves) {
      const uint32_t trips = 20000;
      k.fn<<<cus * w, 256>>>(100, out);  // warm
      CK(hipEventRecord(e0));
      k.fn<<<cus * w, 256>>>(trips, out);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double per = ms * 1e-3 * use_ghz * 1e9 / ((double)trips * k.per_trip * w);
      printf("%8.2f", per);
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
