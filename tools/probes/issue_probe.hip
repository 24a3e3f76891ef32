// Probe: what bounds a SIMD's vector issue on gfx950 when a wave's stream is NOT a bare run of one opcode.
// Each test is a straight-line body of 256 vector instructions on explicit registers (v64..v127), looped `reps`
// times, run with 1 / 2 / 4 waves per SIMD on all 256 CUs; printed: cycles per vector instruction per wave and
// per SIMD.  Tests: operand banks (register index mod 4), VOP2 vs VOP3 encodings (instruction bytes), scalar
// instructions interleaved at the ratios of the fused STFT loop (SALU, s_nop, s_waitcnt, untaken branches),
// cross-lane forms, LDS table reads beside the arithmetic.
// Build: hipcc -O3 --offload-arch=gfx950 -o issue_probe issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>

#define CLOB                                                                                                          \
  "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",     \
      "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", \
      "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109",     \
      "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", \
      "v124", "v125", "v126", "v127", "s40", "s41", "s42", "s43", "s44", "s45", "vcc", "scc"

// 16 instructions: I(d, a, b, c) with d = 64 + i, sources rotated so that the operand banks differ (DB) or coincide (SB)
#define R16_DB(I)                                                                                                      \
  I(64, 81, 98, 115) I(65, 82, 99, 116) I(66, 83, 100, 117) I(67, 84, 101, 118) I(68, 85, 102, 119) I(69, 86, 103, 120) \
      I(70, 87, 104, 121) I(71, 88, 105, 122) I(72, 89, 106, 123) I(73, 90, 107, 124) I(74, 91, 108, 125)              \
          I(75, 92, 109, 126) I(76, 93, 110, 127) I(77, 94, 111, 112) I(78, 95, 96, 113) I(79, 80, 97, 114)
#define R16_SB(I)                                                                                                      \
  I(64, 80, 96, 112) I(65, 81, 97, 113) I(66, 82, 98, 114) I(67, 83, 99, 115) I(68, 84, 100, 116) I(69, 85, 101, 117)  \
      I(70, 86, 102, 118) I(71, 87, 103, 119) I(72, 88, 104, 120) I(73, 89, 105, 121) I(74, 90, 106, 122)              \
          I(75, 91, 107, 123) I(76, 92, 108, 124) I(77, 93, 109, 125) I(78, 94, 110, 126) I(79, 95, 111, 127)

#define S(x) #x
#define ADD32(d, a, b, c) "v_add_f32_e32 v" S(d) ", v" S(a) ", v" S(b) "\n\t"
#define ADD64(d, a, b, c) "v_add_f32_e64 v" S(d) ", v" S(a) ", v" S(b) "\n\t"
#define FMA(d, a, b, c) "v_fma_f32 v" S(d) ", v" S(a) ", v" S(b) ", v" S(c) "\n\t"
#define FMAD(d, a, b, c) "v_fma_f32 v" S(d) ", v" S(a) ", v" S(b) ", v" S(d) "\n\t"
#define FMAC(d, a, b, c) "v_fmac_f32_e32 v" S(d) ", v" S(a) ", v" S(b) "\n\t"
#define MULK(d, a, b, c) "v_mul_f32_e32 v" S(d) ", 0.5, v" S(a) "\n\t"
#define MULS(d, a, b, c) "v_mul_f32_e32 v" S(d) ", s41, v" S(a) "\n\t"
#define PKFMA(d, a, b, c) ""
#define DPPROR(d, a, b, c) "v_mov_b32_dpp v" S(d) ", v" S(a) " row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
#define DPPQP(d, a, b, c) "v_mov_b32_dpp v" S(d) ", v" S(a) " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define FMACDPP(d, a, b, c) "v_fmac_f32_dpp v" S(d) ", v" S(a) ", v" S(b) " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define ADDDPP(d, a, b, c) "v_add_f32_dpp v" S(d) ", v" S(a) ", v" S(b) " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define SWAP32(d, a, b, c) "v_permlane32_swap_b32 v" S(d) ", v" S(a) "\n\t"
#define SWAP16(d, a, b, c) "v_permlane16_swap_b32 v" S(d) ", v" S(a) "\n\t"
#define CNDM(d, a, b, c) "v_cndmask_b32_e32 v" S(d) ", v" S(a) ", v" S(b) ", vcc\n\t"
#define CNDM64(d, a, b, c) "v_cndmask_b32_e64 v" S(d) ", v" S(a) ", v" S(b) ", s[44:45]\n\t"
#define CNDM64N(d, a, b, c) "v_cndmask_b32_e64 v" S(d) ", v" S(a) ", -v" S(b) ", s[44:45]\n\t"
#define CNDM0(d, a, b, c) "v_cndmask_b32_e32 v" S(d) ", v" S(a) ", v" S(b) ", vcc\n\t"
#define MOV(d, a, b, c) "v_mov_b32_e32 v" S(d) ", v" S(a) "\n\t"
#define DEP(d, a, b, c) "v_add_f32_e32 v64, v64, v" S(a) "\n\t"
#define DEP2(d, a, b, c) "v_add_f32_e32 v64, v64, v" S(a) "\n\tv_add_f32_e32 v65, v65, v" S(b) "\n\t"
// scalar companions
#define ADD_SALU1(d, a, b, c) ADD32(d, a, b, c) "s_add_u32 s40, s40, 1\n\t"
#define ADD_SMUL(d, a, b, c) ADD32(d, a, b, c) "s_mul_i32 s40, s40, 3\n\t"
#define ADD_SMULHI(d, a, b, c) ADD32(d, a, b, c) "s_mul_hi_u32 s40, s40, s42\n\t"
#define ADD_NOP(d, a, b, c) ADD32(d, a, b, c) "s_nop 0\n\t"
#define ADD_NOP1(d, a, b, c) ADD32(d, a, b, c) "s_nop 1\n\t"
#define ADD_WAIT(d, a, b, c) ADD32(d, a, b, c) "s_waitcnt lgkmcnt(0)\n\t"
#define ADD_2SALU(d, a, b, c) ADD32(d, a, b, c) "s_add_u32 s40, s40, 1\n\ts_addc_u32 s43, s43, 0\n\t"
#define ADD_MOV(d, a, b, c) ADD32(d, a, b, c) MOV(c, b, a, d)

#define X16(B) B B B B B B B B B B B B B B B B
#define X4(B) B B B B

enum {
  T_ADD32_DB, T_ADD32_SB, T_ADD64_DB, T_FMA_DB, T_FMA_SB, T_FMAD_DB, T_FMAC_DB, T_FMAC_SB, T_MULK, T_MULS, T_MIX_ADD_FMA,
  T_DEP, T_DEP2, T_ADD_SALU1, T_ADD_2SALU, T_ADD_SMUL, T_ADD_SMULHI, T_ADD_NOP, T_ADD_NOP1, T_ADD_WAIT, T_ADD_NOP_Q, T_ADD_WAIT_Q, T_ADD_BR_Q,
  T_DPPROR, T_DPPQP, T_FMACDPP, T_ADDDPP, T_SWAP32, T_SWAP16, T_CNDM, T_CNDM64, T_CNDM64N, T_CNDM_FULL, T_MOV, T_ADD_MOV, T_LDS12, T_LDS4, T_KMIX, T_COUNT
};

// quarter-rate companions: one scalar instruction per 4 vector instructions
#define Q4_NOP ADD32(64, 81, 98, 0) ADD32(65, 82, 99, 0) ADD32(66, 83, 100, 0) ADD32(67, 84, 101, 0) "s_nop 0\n\t"
#define Q4_WAIT ADD32(64, 81, 98, 0) ADD32(65, 82, 99, 0) ADD32(66, 83, 100, 0) ADD32(67, 84, 101, 0) "s_waitcnt lgkmcnt(0)\n\t"
#define Q4_BR ADD32(64, 81, 98, 0) ADD32(65, 82, 99, 0) ADD32(66, 83, 100, 0) ADD32(67, 84, 101, 0) "s_cmp_eq_u32 s40, s42\n\ts_cbranch_scc1 1f\n\t1:\n\t"
// LDS table reads beside the arithmetic: one ds_read_b64 per 12 (4) vector instructions, waited for at the end of the body
#define L12 ADD32(64, 81, 98, 0) ADD32(65, 82, 99, 0) ADD32(66, 83, 100, 0) ADD32(67, 84, 101, 0) ADD32(68, 85, 102, 0) ADD32(69, 86, 103, 0) \
  ADD32(70, 87, 104, 0) ADD32(71, 88, 105, 0) ADD32(72, 89, 106, 0) ADD32(73, 90, 107, 0) ADD32(74, 91, 108, 0) ADD32(75, 92, 109, 0) "ds_read_b64 v[126:127], v63\n\t"
#define L4 ADD32(64, 81, 98, 0) ADD32(65, 82, 99, 0) ADD32(66, 83, 100, 0) ADD32(67, 84, 101, 0) "ds_read_b64 v[126:127], v63\n\t"
// the fused STFT loop's mix per 16 vector instructions (main loop of stft2048_power_kernel: 850 vector, 97 s_waitcnt,
// 48 s_nop, ~400 SALU, ~60 branches per frame): 16 vector (5 add, 5 sub->add, 3 mul, 2 fmac, 1 fma) + 8 SALU + 2 waitcnt + 1 nop + 1 branch
#define KMIX                                                                                                           \
  ADD32(64, 81, 98, 0) "s_add_u32 s40, s40, 1\n\t" ADD32(65, 82, 99, 0) ADD32(66, 83, 100, 0) "s_addc_u32 s43, s43, 0\n\t" ADD32(67, 84, 101, 0) \
  ADD32(68, 85, 102, 0) "s_mul_i32 s40, s40, 3\n\t" "s_waitcnt lgkmcnt(0)\n\t" ADD32(69, 86, 103, 0) ADD32(70, 87, 104, 0) "s_add_u32 s40, s40, 1\n\t" \
  ADD32(71, 88, 105, 0) ADD32(72, 89, 106, 0) "s_nop 1\n\t" ADD32(73, 90, 107, 0) "s_mul_hi_u32 s40, s40, s42\n\t" MULK(74, 91, 0, 0) MULK(75, 92, 0, 0) \
  "s_add_u32 s40, s40, 1\n\t" MULK(76, 93, 0, 0) "s_waitcnt lgkmcnt(0)\n\t" FMAC(77, 94, 111, 0) "s_cmp_eq_u32 s40, s42\n\ts_cbranch_scc1 1f\n\t1:\n\t" \
  FMAC(78, 95, 96, 0) "s_addc_u32 s43, s43, 0\n\t" FMA(79, 80, 97, 114) "s_add_u32 s40, s40, 1\n\t"

template <int T>
__global__ void __launch_bounds__(1024) k(unsigned long long *cyc, int reps, float *sink) {
  __shared__ float lds[2048];
  lds[threadIdx.x & 2047] = 1.0f;
  lds[(threadIdx.x + 1024) & 2047] = 1.0f;
  // finite values everywhere, v63 = an LDS byte address (lane-contiguous 8-byte reads)
  asm volatile(
      "v_and_b32 v63, 63, v0\n\tv_lshlrev_b32 v63, 3, v63\n\t"
      "s_mov_b32 s40, 0\n\ts_mov_b32 s41, 0x3f7fff00\n\ts_mov_b32 s42, 0x7fffffff\n\ts_mov_b32 s43, 0\n\t"
      "v_cmp_gt_u32 vcc, 32, v63\n\ts_mov_b64 s[44:45], vcc\n\t" ::: "v63", CLOB);
  if constexpr (T == T_CNDM_FULL) asm volatile("s_mov_b64 vcc, -1" ::: "vcc");
#define INIT(d, a, b, c) "v_mov_b32 v" S(d) ", 1.0\n\tv_mov_b32 v" S(a) ", 0.5\n\tv_mov_b32 v" S(b) ", 0.25\n\tv_mov_b32 v" S(c) ", 0.125\n\t"
  asm volatile(R16_SB(INIT)::: CLOB);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    if constexpr (T == T_ADD32_DB) asm volatile(X16(R16_DB(ADD32))::: CLOB);
    if constexpr (T == T_ADD32_SB) asm volatile(X16(R16_SB(ADD32))::: CLOB);
    if constexpr (T == T_ADD64_DB) asm volatile(X16(R16_DB(ADD64))::: CLOB);
    if constexpr (T == T_FMA_DB) asm volatile(X16(R16_DB(FMA))::: CLOB);
    if constexpr (T == T_FMA_SB) asm volatile(X16(R16_SB(FMA))::: CLOB);
    if constexpr (T == T_FMAD_DB) asm volatile(X16(R16_DB(FMAD))::: CLOB);
    if constexpr (T == T_FMAC_DB) asm volatile(X16(R16_DB(FMAC))::: CLOB);
    if constexpr (T == T_FMAC_SB) asm volatile(X16(R16_SB(FMAC))::: CLOB);
    if constexpr (T == T_MULK) asm volatile(X16(R16_DB(MULK))::: CLOB);
    if constexpr (T == T_MULS) asm volatile(X16(R16_DB(MULS))::: CLOB);
    if constexpr (T == T_MIX_ADD_FMA) asm volatile(X4(R16_DB(ADD32) R16_DB(FMA) R16_DB(ADD32) R16_DB(FMA))::: CLOB);
    if constexpr (T == T_DEP) asm volatile(X16(R16_DB(DEP))::: CLOB);
    if constexpr (T == T_DEP2) asm volatile(X4(R16_DB(DEP2) R16_DB(DEP2))::: CLOB);
    if constexpr (T == T_ADD_SALU1) asm volatile(X16(R16_DB(ADD_SALU1))::: CLOB);
    if constexpr (T == T_ADD_2SALU) asm volatile(X16(R16_DB(ADD_2SALU))::: CLOB);
    if constexpr (T == T_ADD_SMUL) asm volatile(X16(R16_DB(ADD_SMUL))::: CLOB);
    if constexpr (T == T_ADD_SMULHI) asm volatile(X16(R16_DB(ADD_SMULHI))::: CLOB);
    if constexpr (T == T_ADD_NOP) asm volatile(X16(R16_DB(ADD_NOP))::: CLOB);
    if constexpr (T == T_ADD_NOP1) asm volatile(X16(R16_DB(ADD_NOP1))::: CLOB);
    if constexpr (T == T_ADD_WAIT) asm volatile(X16(R16_DB(ADD_WAIT))::: CLOB);
    if constexpr (T == T_ADD_NOP_Q) asm volatile(X16(X4(Q4_NOP))::: CLOB);
    if constexpr (T == T_ADD_WAIT_Q) asm volatile(X16(X4(Q4_WAIT))::: CLOB);
    if constexpr (T == T_ADD_BR_Q) asm volatile(X16(X4(Q4_BR))::: CLOB);
    if constexpr (T == T_DPPROR) asm volatile(X16(R16_DB(DPPROR))::: CLOB);
    if constexpr (T == T_DPPQP) asm volatile(X16(R16_DB(DPPQP))::: CLOB);
    if constexpr (T == T_FMACDPP) asm volatile(X16(R16_DB(FMACDPP))::: CLOB);
    if constexpr (T == T_ADDDPP) asm volatile(X16(R16_DB(ADDDPP))::: CLOB);
    if constexpr (T == T_SWAP32) asm volatile(X16(R16_DB(SWAP32))::: CLOB);
    if constexpr (T == T_SWAP16) asm volatile(X16(R16_DB(SWAP16))::: CLOB);
    if constexpr (T == T_CNDM) asm volatile(X16(R16_DB(CNDM))::: CLOB);
    if constexpr (T == T_CNDM64) asm volatile(X16(R16_DB(CNDM64))::: CLOB);
    if constexpr (T == T_CNDM64N) asm volatile(X16(R16_DB(CNDM64N))::: CLOB);
    if constexpr (T == T_CNDM_FULL) asm volatile(X16(R16_DB(CNDM0))::: CLOB);
    if constexpr (T == T_MOV) asm volatile(X16(R16_DB(MOV))::: CLOB);
    if constexpr (T == T_ADD_MOV) asm volatile(X4(R16_DB(ADD_MOV) R16_DB(ADD_MOV))::: CLOB);   // 256 vector: 128 add + 128 mov
    if constexpr (T == T_LDS12) asm volatile(X16(L12 ADD32(76, 93, 110, 0) ADD32(77, 94, 111, 0) ADD32(78, 95, 96, 0) ADD32(79, 80, 97, 0)) "s_waitcnt lgkmcnt(0)\n\t" ::: CLOB, "memory");
    if constexpr (T == T_LDS4) asm volatile(X16(X4(L4)) "s_waitcnt lgkmcnt(0)\n\t" ::: CLOB, "memory");
    if constexpr (T == T_KMIX) asm volatile(X16(KMIX)::: CLOB);
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (reps < 0) sink[threadIdx.x] = lds[threadIdx.x];
}

template <int T>
void run(const char *name, int vec_per_body) {
  unsigned long long *cyc;
  float *sink;
  hipMalloc(&cyc, 256 * 8);
  hipMalloc(&sink, 4096);
  const int reps = 200;
  printf("%-44s", name);
  for (int threads : {256, 512, 768, 1024}) {
    hipLaunchKernelGGL(k<T>, dim3(256), dim3(threads), 0, 0, cyc, reps, sink);
    hipLaunchKernelGGL(k<T>, dim3(256), dim3(threads), 0, 0, cyc, reps, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto c : h) mean += c;
    mean /= 256;
    const double per_wave = mean / ((double)reps * vec_per_body);
    printf("  %dw: %5.2f /wave %5.2f /SIMD", threads / 256, per_wave, per_wave / (threads / 256));
  }
  printf("\n");
  hipFree(cyc);
  hipFree(sink);
}

int main() {
  printf("cycles per VECTOR instruction (s_memtime; per wave, and per SIMD = per wave / waves per SIMD)\n");
  run<T_ADD32_DB>("v_add_f32 e32, operand banks differ", 256);
  run<T_ADD32_SB>("v_add_f32 e32, operands in one bank", 256);
  run<T_ADD64_DB>("v_add_f32 e64 (8-byte encoding)", 256);
  run<T_FMA_DB>("v_fma_f32 d,a,b,c banks differ", 256);
  run<T_FMA_SB>("v_fma_f32 d,a,b,c one bank", 256);
  run<T_FMAD_DB>("v_fma_f32 d,a,b,d", 256);
  run<T_FMAC_DB>("v_fmac_f32 e32 banks differ", 256);
  run<T_FMAC_SB>("v_fmac_f32 e32 one bank", 256);
  run<T_MULK>("v_mul_f32 const, v", 256);
  run<T_MULS>("v_mul_f32 sgpr, v", 256);
  run<T_MIX_ADD_FMA>("add / fma alternating groups of 16", 256);
  run<T_DEP>("dependent chain v_add v64,v64,x", 256);
  run<T_DEP2>("two interleaved dependent chains", 256);
  run<T_ADD_SALU1>("add + 1 s_add_u32 each", 256);
  run<T_ADD_2SALU>("add + s_add_u32 + s_addc_u32 each", 256);
  run<T_ADD_SMUL>("add + s_mul_i32 each", 256);
  run<T_ADD_SMULHI>("add + s_mul_hi_u32 each", 256);
  run<T_ADD_NOP>("add + s_nop 0 each", 256);
  run<T_ADD_NOP1>("add + s_nop 1 each", 256);
  run<T_ADD_WAIT>("add + s_waitcnt lgkmcnt(0) each", 256);
  run<T_ADD_NOP_Q>("4 add + s_nop 0", 256);
  run<T_ADD_WAIT_Q>("4 add + s_waitcnt", 256);
  run<T_ADD_BR_Q>("4 add + s_cmp + untaken s_cbranch", 256);
  run<T_DPPROR>("v_mov_b32_dpp row_ror:8", 256);
  run<T_DPPQP>("v_mov_b32_dpp quad_perm", 256);
  run<T_FMACDPP>("v_fmac_f32_dpp quad_perm", 256);
  run<T_ADDDPP>("v_add_f32_dpp quad_perm", 256);
  run<T_SWAP32>("v_permlane32_swap_b32", 256);
  run<T_SWAP16>("v_permlane16_swap_b32", 256);
  run<T_CNDM>("v_cndmask_b32 e32 (vcc)", 256);
  run<T_CNDM64>("v_cndmask_b32 e64 (sgpr pair)", 256);
  run<T_CNDM64N>("v_cndmask_b32 e64 (sgpr pair, neg)", 256);
  run<T_CNDM_FULL>("v_cndmask_b32 e32, vcc all ones", 256);
  run<T_MOV>("v_mov_b32", 256);
  run<T_ADD_MOV>("add + mov alternating", 256);
  run<T_LDS12>("16 add + 1 ds_read_b64 (x16), wait at end", 256);
  run<T_LDS4>("4 add + 1 ds_read_b64 (x64), wait at end", 256);
  run<T_KMIX>("fused-loop mix: 16 vec + 8 SALU + 2 wait + nop + br", 256);
  return 0;
}
