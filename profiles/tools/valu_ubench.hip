// Issue cost of the instructions the carve kernel is made of, measured on this GPU.
//   hipcc --offload-arch=gfx950 -O2 -o valu_ubench valu_ubench.hip && ./valu_ubench
// The chip is saturated with waves of the same loop (16 workgroups of 4 waves per CU, twice what fits):
// an unrolled block of 32 independent instructions of ONE kind, repeated.  The launch is timed with HIP
// events; every wave also brackets its loop with s_memtime, whose global span / wall time gives the
// shader clock of that launch.  Reported: SIMD cycles per wave-instruction =
//   wall * clock * (4 * CUs) / (waves * instructions per wave),
// and, next to it, the same figure with ONE wave per SIMD (latency-bound issue rate of a lone wave).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

constexpr int kIters = 4000;

#define UB_KERNEL(NAME, BODY16)                                                                  \
  __global__ __launch_bounds__(256) void NAME(unsigned long long* out, float seed) {             \
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f,   \
          a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;                                           \
    float b = seed * 0.5f + 1.0f, c = seed * 0.25f + 0.5f;                                       \
    int lds_addr = (threadIdx.x & 63) * 16;                                                      \
    (void)lds_addr;                                                                              \
    unsigned long long t0, t1;                                                                   \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)); \
    for (int it = 0; it < kIters; ++it) {                                                        \
      asm volatile(BODY16 BODY16                                                                 \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                   : "v"(b), "v"(c), "v"(lds_addr), "s"(seed)                                    \
                   : "vcc", "memory");                                                           \
    }                                                                                            \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)); \
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                             \
    if (s == 12345.678f) out[8] = 1;                                                       \
    if ((threadIdx.x & 63) == 0) { atomicMin(&out[0], t0); atomicMax(&out[1], t1); atomicMax(&out[2], t1 - t0); } \
  }

// 16 instructions, two per accumulator register (independent across registers)
#define X8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define X16(OP) X8(OP) X8(OP)

#define I_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n\t"
#define I_FMAC(k) "v_fmac_f32 %" #k ", %8, %9\n\t"
#define I_ADD(k) "v_add_f32 %" #k ", %" #k ", %8\n\t"
#define I_ADDS(k) "v_add_f32 %" #k ", %11, %" #k "\n\t"
#define I_MUL(k) "v_mul_f32 %" #k ", %" #k ", %8\n\t"
#define I_SUBLIT(k) "v_sub_f32 %" #k ", 0x3f8ccccd, %" #k "\n\t"
#define I_SUB1(k) "v_sub_f32 %" #k ", 1.0, %" #k "\n\t"
#define I_RCP(k) "v_rcp_f32 %" #k ", %" #k "\n\t"
#define I_FLOOR(k) "v_floor_f32 %" #k ", %" #k "\n\t"
#define I_FRACT(k) "v_fract_f32 %" #k ", %" #k "\n\t"
#define I_CVTI(k) "v_cvt_i32_f32 %" #k ", %" #k "\n\t"
#define I_CVTF(k) "v_cvt_f32_i32 %" #k ", %" #k "\n\t"
#define I_MAX(k) "v_max_f32 %" #k ", %" #k ", %8\n\t"
#define I_CMP(k) "v_cmp_gt_f32 vcc, %" #k ", %8\n\t"
#define I_CMPS(k) "v_cmp_gt_f32 s[20:21], %" #k ", %8\n\t"
#define I_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n\t"
#define I_ADDC(k) "v_addc_co_u32 %" #k ", vcc, 0, %" #k ", vcc\n\t"
#define I_MOV(k) "v_mov_b32 %" #k ", %8\n\t"
#define I_ADDU(k) "v_add_u32 %" #k ", %" #k ", %8\n\t"
#define I_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 4, %8\n\t"
#define I_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n\t"
#define I_MAD24(k) "v_mad_u32_u24 %" #k ", %" #k ", %8, %9\n\t"
#define I_MUL24(k) "v_mul_u32_u24 %" #k ", %" #k ", %8\n\t"
#define I_AND(k) "v_and_b32 %" #k ", %" #k ", %8\n\t"
#define I_MED3(k) "v_med3_f32 %" #k ", %" #k ", %8, %9\n\t"
#define I_DPP(k) "v_min_f32_dpp %" #k ", %" #k ", %" #k " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define I_MOVDPP(k) "v_mov_b32_dpp %" #k ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_READFL(k) "v_readfirstlane_b32 s20, %" #k "\n\t"
#define I_MIN(k) "v_min_f32 %" #k ", %" #k ", %8\n\t"
#define I_SUBV(k) "v_sub_f32 %" #k ", %" #k ", %8\n\t"
#define I_MULS(k) "v_mul_f32 %" #k ", %11, %" #k "\n\t"
#define I_FMACS(k) "v_fmac_f32 %" #k ", %11, %9\n\t"
#define I_FMAS(k) "v_fma_f32 %" #k ", %11, %" #k ", %9\n\t"
#define I_OR(k) "v_or_b32 %" #k ", %" #k ", %8\n\t"
#define I_XOR(k) "v_xor_b32 %" #k ", %" #k ", %8\n\t"
#define I_LSHL(k) "v_lshlrev_b32 %" #k ", 4, %" #k "\n\t"
#define I_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 4, 8\n\t"
#define I_TRUNC(k) "v_trunc_f32 %" #k ", %" #k "\n\t"
#define I_RNDNE(k) "v_rndne_f32 %" #k ", %" #k "\n\t"
#define I_CVTU(k) "v_cvt_u32_f32 %" #k ", %" #k "\n\t"
#define I_SUBU(k) "v_sub_u32 %" #k ", %" #k ", %8\n\t"
#define I_ADD3(k) "v_add3_u32 %" #k ", %" #k ", %8, %9\n\t"
#define I_CMPLT(k) "v_cmp_lt_f32 vcc, %" #k ", %8\n\t"
#define I_CMPI(k) "v_cmp_gt_i32 vcc, %" #k ", %8\n\t"
#define I_CNDS(k) "v_cndmask_b32 %" #k ", %" #k ", %8, s[20:21]\n\t"
#define I_ADDCO(k) "v_add_co_u32 %" #k ", vcc, %" #k ", %8\n\t"
#define I_MOVS(k) "v_mov_b32 %" #k ", %11\n\t"
#define I_MADF(k) "v_mad_f32 %" #k ", %" #k ", %8, %9\n\t"
#define I_SMUL(k) "s_mul_i32 s20, s20, 3\n\t"
#define I_SNOP(k) "s_nop 0\n\t"
#define I_SADD(k) "s_add_u32 s20, s20, 1\n\t"

UB_KERNEL(k_fma, X16(I_FMA))
UB_KERNEL(k_fmac, X16(I_FMAC))
UB_KERNEL(k_add, X16(I_ADD))
UB_KERNEL(k_add_sgpr, X16(I_ADDS))
UB_KERNEL(k_mul, X16(I_MUL))
UB_KERNEL(k_sub_literal, X16(I_SUBLIT))
UB_KERNEL(k_sub_inline1, X16(I_SUB1))
UB_KERNEL(k_rcp, X16(I_RCP))
UB_KERNEL(k_floor, X16(I_FLOOR))
UB_KERNEL(k_fract, X16(I_FRACT))
UB_KERNEL(k_cvt_i32_f32, X16(I_CVTI))
UB_KERNEL(k_cvt_f32_i32, X16(I_CVTF))
UB_KERNEL(k_max, X16(I_MAX))
UB_KERNEL(k_cmp_vcc, X16(I_CMP))
UB_KERNEL(k_cmp_sgpr, X16(I_CMPS))
UB_KERNEL(k_cndmask, X16(I_CNDMASK))
UB_KERNEL(k_addc, X16(I_ADDC))
UB_KERNEL(k_mov, X16(I_MOV))
UB_KERNEL(k_add_u32, X16(I_ADDU))
UB_KERNEL(k_lshl_add, X16(I_LSHLADD))
UB_KERNEL(k_mul_lo_u32, X16(I_MULLO))
UB_KERNEL(k_mad_u32_u24, X16(I_MAD24))
UB_KERNEL(k_mul_u32_u24, X16(I_MUL24))
UB_KERNEL(k_and, X16(I_AND))
UB_KERNEL(k_med3, X16(I_MED3))
UB_KERNEL(k_min_dpp, X16(I_DPP))
UB_KERNEL(k_mov_dpp, X16(I_MOVDPP))
UB_KERNEL(k_readfirstlane, X16(I_READFL))
UB_KERNEL(k_min, X16(I_MIN))
UB_KERNEL(k_subv, X16(I_SUBV))
UB_KERNEL(k_mul_sgpr, X16(I_MULS))
UB_KERNEL(k_fmac_sgpr, X16(I_FMACS))
UB_KERNEL(k_fma_sgpr, X16(I_FMAS))
UB_KERNEL(k_or, X16(I_OR))
UB_KERNEL(k_xor, X16(I_XOR))
UB_KERNEL(k_lshl, X16(I_LSHL))
UB_KERNEL(k_bfe, X16(I_BFE))
UB_KERNEL(k_trunc, X16(I_TRUNC))
UB_KERNEL(k_rndne, X16(I_RNDNE))
UB_KERNEL(k_cvtu, X16(I_CVTU))
UB_KERNEL(k_subu, X16(I_SUBU))
UB_KERNEL(k_add3, X16(I_ADD3))
UB_KERNEL(k_cmplt, X16(I_CMPLT))
UB_KERNEL(k_cmpi, X16(I_CMPI))
UB_KERNEL(k_cnds, X16(I_CNDS))
UB_KERNEL(k_addco, X16(I_ADDCO))
UB_KERNEL(k_movs, X16(I_MOVS))
UB_KERNEL(k_smul, X16(I_SMUL))
#define I_CMPCND(k) "v_cmp_gt_f32 vcc, %" #k ", %8\n\tv_cndmask_b32 %" #k ", %" #k ", %9, vcc\n\t"
#define I_CMPCNDS(k) "v_cmp_gt_f32 s[20:21], %" #k ", %8\n\tv_cndmask_b32 %" #k ", %" #k ", %9, s[20:21]\n\t"
#define I_CMPADDC(k) "v_cmp_gt_f32 vcc, %" #k ", %8\n\tv_addc_co_u32 %" #k ", vcc, 0, %" #k ", vcc\n\t"
UB_KERNEL(k_cmpcnd, X8(I_CMPCND))
UB_KERNEL(k_cmpcnds, X8(I_CMPCNDS))
UB_KERNEL(k_cmpaddc, X8(I_CMPADDC))
UB_KERNEL(k_s_nop, X16(I_SNOP))
UB_KERNEL(k_s_add, X16(I_SADD))

// packed fp32: register pairs
#define PK_KERNEL(NAME, OPSTR)                                                                   \
  __global__ __launch_bounds__(256) void NAME(unsigned long long* out, float seed) {             \
    typedef float f2 __attribute__((ext_vector_type(2)));                                        \
    f2 a0 = {seed + threadIdx.x, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f,              \
       a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;                               \
    f2 b = {seed * 0.5f + 1.0f, seed}, c = {seed * 0.25f + 0.5f, seed};                          \
    unsigned long long t0, t1;                                                                   \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)); \
    for (int it = 0; it < kIters; ++it) {                                                        \
      asm volatile(REP4(OPSTR(0) OPSTR(1) OPSTR(2) OPSTR(3) OPSTR(4) OPSTR(5) OPSTR(6) OPSTR(7))  \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                   : "v"(b), "v"(c));                                                            \
    }                                                                                            \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)); \
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                \
    if (s.x + s.y == 12345.678f) out[8] = 1;                                               \
    if ((threadIdx.x & 63) == 0) { atomicMin(&out[0], t0); atomicMax(&out[1], t1); atomicMax(&out[2], t1 - t0); } \
  }
#define P_FMA(k) "v_pk_fma_f32 %" #k ", %" #k ", %8, %9\n\t"
#define P_MUL(k) "v_pk_mul_f32 %" #k ", %" #k ", %8\n\t"
#define P_ADD(k) "v_pk_add_f32 %" #k ", %" #k ", %8\n\t"
PK_KERNEL(k_pk_fma, P_FMA)
PK_KERNEL(k_pk_mul, P_MUL)
PK_KERNEL(k_pk_add, P_ADD)

// LDS reads: every lane its own 16 bytes (conflict-free b128), or a gathered pattern
typedef float lds_f4 __attribute__((ext_vector_type(4)));
typedef float lds_f2 __attribute__((ext_vector_type(2)));
__device__ inline float first(lds_f4 v) { return v[0]; }
__device__ inline float first(lds_f2 v) { return v[0]; }
__device__ inline float first(float v) { return v; }
#define LDS_KERNEL(NAME, OPSTR, FV, STRIDE)                                                     \
  __global__ __launch_bounds__(256) void NAME(unsigned long long* out, float seed) {             \
    __shared__ float4 lds[2048];                                                                 \
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = make_float4(seed, 1.f, 2.f, 3.f);      \
    __syncthreads();                                                                             \
    FV r0, r1, r2, r3, r4, r5, r6, r7;                                                           \
    int addr = (((threadIdx.x & 63) * STRIDE) & 511) * 16 + (threadIdx.x >> 6) * 8192;            \
    unsigned long long t0, t1;                                                                   \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)); \
    for (int it = 0; it < kIters; ++it) {                                                        \
      asm volatile(REP4(OPSTR(0) OPSTR(1) OPSTR(2) OPSTR(3) OPSTR(4) OPSTR(5) OPSTR(6) OPSTR(7)) "s_waitcnt lgkmcnt(0)\n\t" \
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) \
                   : "v"(addr) : "memory");                                                      \
    }                                                                                            \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)); \
    float s = first(r0) + first(r1) + first(r2) + first(r3) + first(r4) + first(r5) + first(r6) + first(r7); \
    if (s == 12345.678f) out[8] = 1;                                                       \
    if ((threadIdx.x & 63) == 0) { atomicMin(&out[0], t0); atomicMax(&out[1], t1); atomicMax(&out[2], t1 - t0); } \
  }
#define L_B128(k) "ds_read_b128 %" #k ", %8 offset:0x" #k "0\n\t"
#define L_B64(k) "ds_read_b64 %" #k ", %8 offset:0x" #k "0\n\t"
#define L_B32(k) "ds_read_b32 %" #k ", %8 offset:0x" #k "0\n\t"
LDS_KERNEL(k_ds_read_b128, L_B128, lds_f4, 1)
LDS_KERNEL(k_ds_read_b128_gather, L_B128, lds_f4, 37)
LDS_KERNEL(k_ds_read_b64, L_B64, lds_f2, 1)
LDS_KERNEL(k_ds_read_b32, L_B32, float, 1)

typedef void (*kern_t)(unsigned long long*, float);
struct Case { const char* name; kern_t k; int per_iter; };

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  unsigned long long* d = nullptr;
  (void)hipMalloc(&d, ((1 << 20) + 16) * sizeof(unsigned long long));
  std::vector<Case> cases = {
      {"v_fma_f32", k_fma, 32}, {"v_fmac_f32", k_fmac, 32}, {"v_add_f32", k_add, 32},
      {"v_add_f32 (sgpr src)", k_add_sgpr, 32}, {"v_mul_f32", k_mul, 32},
      {"v_sub_f32 (32-bit literal)", k_sub_literal, 32}, {"v_sub_f32 (inline 1.0)", k_sub_inline1, 32},
      {"v_pk_fma_f32", k_pk_fma, 32}, {"v_pk_mul_f32", k_pk_mul, 32}, {"v_pk_add_f32", k_pk_add, 32},
      {"v_rcp_f32", k_rcp, 32}, {"v_floor_f32", k_floor, 32}, {"v_fract_f32", k_fract, 32},
      {"v_cvt_i32_f32", k_cvt_i32_f32, 32}, {"v_cvt_f32_i32", k_cvt_f32_i32, 32}, {"v_max_f32", k_max, 32},
      {"v_cmp_gt_f32 vcc", k_cmp_vcc, 32}, {"v_cmp_gt_f32 sgpr", k_cmp_sgpr, 32},
      {"v_cndmask_b32", k_cndmask, 32}, {"v_addc_co_u32", k_addc, 32}, {"v_mov_b32", k_mov, 32},
      {"v_add_u32", k_add_u32, 32}, {"v_lshl_add_u32", k_lshl_add, 32}, {"v_mul_lo_u32", k_mul_lo_u32, 32},
      {"v_mad_u32_u24", k_mad_u32_u24, 32}, {"v_mul_u32_u24", k_mul_u32_u24, 32}, {"v_and_b32", k_and, 32},
      {"v_med3_f32", k_med3, 32}, {"v_min_f32 dpp", k_min_dpp, 32}, {"v_mov_b32 dpp", k_mov_dpp, 32},
      {"v_readfirstlane_b32", k_readfirstlane, 32}, {"s_nop 0", k_s_nop, 32}, {"s_add_u32", k_s_add, 32},
      {"s_mul_i32", k_smul, 32},
      {"v_cmp vcc + v_cndmask vcc (pair)", k_cmpcnd, 32}, {"v_cmp sgpr + v_cndmask sgpr (pair)", k_cmpcnds, 32},
      {"v_cmp vcc + v_addc vcc (pair)", k_cmpaddc, 32},
      {"v_min_f32", k_min, 32}, {"v_sub_f32", k_subv, 32}, {"v_mul_f32 (sgpr src)", k_mul_sgpr, 32},
      {"v_fmac_f32 (sgpr src)", k_fmac_sgpr, 32}, {"v_fma_f32 (sgpr src)", k_fma_sgpr, 32},
      {"v_or_b32", k_or, 32}, {"v_xor_b32", k_xor, 32}, {"v_lshlrev_b32", k_lshl, 32}, {"v_bfe_u32", k_bfe, 32},
      {"v_trunc_f32", k_trunc, 32}, {"v_rndne_f32", k_rndne, 32}, {"v_cvt_u32_f32", k_cvtu, 32},
      {"v_sub_u32", k_subu, 32}, {"v_add3_u32", k_add3, 32}, {"v_cmp_lt_f32 vcc", k_cmplt, 32},
      {"v_cmp_gt_i32 vcc", k_cmpi, 32}, {"v_cndmask_b32 (sgpr mask)", k_cnds, 32},
      {"v_add_co_u32", k_addco, 32}, {"v_mov_b32 (sgpr src)", k_movs, 32},
      {"ds_read_b128 (linear)", k_ds_read_b128, 32}, {"ds_read_b128 (stride 37 quads)", k_ds_read_b128_gather, 32},
      {"ds_read_b64", k_ds_read_b64, 32}, {"ds_read_b32", k_ds_read_b32, 32},
  };
  std::printf("device %s, %d CUs; SIMD cycles per wave-instruction\n", p.gcnArchName, ncu);
  std::printf("%-34s %10s %10s %10s\n", "instruction", "saturated", "lone wave", "clock GHz");
  for (const Case& c : cases) {
    double lone = 0, ghz = 0, sat = 0;
    for (int mode = 1; mode >= 0; --mode) {
      const int blocks = mode == 0 ? ncu * 16 : ncu;  // saturated / one wave per SIMD
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0);
      (void)hipEventCreate(&e1);
      hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, d, 1.0f);  // warm-up
      (void)hipDeviceSynchronize();
      unsigned long long init[3] = {~0ull, 0ull, 0ull};
      (void)hipMemcpy(d, init, sizeof(init), hipMemcpyHostToDevice);
      (void)hipEventRecord(e0, 0);
      hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[3];
      (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      const double insts = (double)kIters * c.per_iter;
      if (mode == 1) {
        lone = (double)h[2] / insts;          // slowest lone wave, its own s_memtime ticks
        ghz = (double)h[2] / (ms * 1e6);      // ticks of that wave / wall time of the launch (slightly low)
      } else {
        // wall time in shader cycles * SIMDs / wave-instructions
        sat = ms * 1e6 * ghz * (4.0 * ncu) / ((double)blocks * 4 * insts);
      }
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
    }
    std::printf("%-34s %10.2f %10.2f %10.2f\n", c.name, sat, lone, ghz);
    std::fflush(stdout);
  }
  (void)hipFree(d);
  return 0;
}
