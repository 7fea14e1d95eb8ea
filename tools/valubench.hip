// Development tool: issue cost of the vector instructions the noise generator is made of, on gfx950, in the regime the
// sweep kernels run in (8 wavefronts per SIMD on every CU).  One kernel per instruction: a loop of 256 copies of the
// instruction on four independent register chains; time per wavefront-instruction per SIMD from HIP events, printed in ns
// and relative to v_mov_b32.  Build: hipcc -O3 --offload-arch=gfx950 tools/valubench.hip -o tools/valubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

#define KERNEL(name, decl, asmtext, ...)                                                                            \
  __global__ __launch_bounds__(256) void name(int iters, double *sink)                                             \
  {                                                                                                                \
    decl;                                                                                                          \
    for (int it = 0; it < iters; ++it) { REP64(asm volatile(asmtext : __VA_ARGS__);) }                            \
    if (iters < 0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3;                           \
  }

#define U32DECL uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3
#define F64DECL double a0 = threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3
#define U64DECL uint64_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3
#define C4 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
// 64-bit chains a0..a3 with 32-bit side operands b0..b3 (%4..%7)
#define MIXDECL(T) T a0 = (T)threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3; uint32_t b0 = threadIdx.x | 1u, b1 = b0 + 2, b2 = b0 + 4, b3 = b0 + 6
#define C8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)

KERNEL(k_mov, U32DECL, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0", C4)
KERNEL(k_xor, U32DECL, "v_xor_b32 %0, %1, %0\n v_xor_b32 %1, %2, %1\n v_xor_b32 %2, %3, %2\n v_xor_b32 %3, %0, %3", C4)
KERNEL(k_add, U32DECL, "v_add_u32 %0, %1, %0\n v_add_u32 %1, %2, %1\n v_add_u32 %2, %3, %2\n v_add_u32 %3, %0, %3", C4)
KERNEL(k_cndmask, U32DECL, "v_cndmask_b32 %0, %1, %0, vcc\n v_cndmask_b32 %1, %2, %1, vcc\n v_cndmask_b32 %2, %3, %2, vcc\n v_cndmask_b32 %3, %0, %3, vcc", C4)
// the same select with the mask in an SGPR pair (what the compiler emits for a wave-varying bool) and with VCC written by a
// v_cmp inside the block
#define SMASKDECL uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3; uint64_t m = 0x5555555555555555ull
KERNEL(k_cndmask_sgpr, SMASKDECL, "v_cndmask_b32_e64 %0, %1, %0, %4\n v_cndmask_b32_e64 %1, %2, %1, %4\n v_cndmask_b32_e64 %2, %3, %2, %4\n v_cndmask_b32_e64 %3, %0, %3, %4", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(m))
KERNEL(k_cmp_cndmask, U32DECL, "v_cmp_gt_u32 vcc, %1, %0\n v_cndmask_b32 %1, %2, %1, vcc\n v_cmp_gt_u32 vcc, %3, %2\n v_cndmask_b32 %3, %0, %3, vcc", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc")
KERNEL(k_cmp, U32DECL, "v_cmp_gt_u32 vcc, %1, %0\n v_cmp_gt_u32 vcc, %2, %1\n v_cmp_gt_u32 vcc, %3, %2\n v_cmp_gt_u32 vcc, %0, %3", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc")
KERNEL(k_cndmask_mix, U32DECL, "v_cndmask_b32 %0, %1, %0, vcc\n v_xor_b32 %1, %2, %1\n v_xor_b32 %2, %3, %2\n v_xor_b32 %3, %0, %3", C4)
KERNEL(k_bfi, U32DECL, "v_bfi_b32 %0, %1, %0, %2\n v_bfi_b32 %1, %2, %1, %3\n v_bfi_b32 %2, %3, %2, %0\n v_bfi_b32 %3, %0, %3, %1", C4)
KERNEL(k_and_or, U32DECL, "v_and_or_b32 %0, %1, %0, %2\n v_and_or_b32 %1, %2, %1, %3\n v_and_or_b32 %2, %3, %2, %0\n v_and_or_b32 %3, %0, %3, %1", C4)
KERNEL(k_lshl_add, U32DECL, "v_lshl_add_u32 %0, %1, 3, %0\n v_lshl_add_u32 %1, %2, 3, %1\n v_lshl_add_u32 %2, %3, 3, %2\n v_lshl_add_u32 %3, %0, 3, %3", C4)
KERNEL(k_lshlrev, U32DECL, "v_lshlrev_b32 %0, 3, %1\n v_lshlrev_b32 %1, 3, %2\n v_lshlrev_b32 %2, 3, %3\n v_lshlrev_b32 %3, 3, %0", C4)
KERNEL(k_fmac_f64, F64DECL, "v_fmac_f64 %0, %1, %2\n v_fmac_f64 %1, %2, %3\n v_fmac_f64 %2, %3, %0\n v_fmac_f64 %3, %0, %1", C4)
KERNEL(k_max_f64, F64DECL, "v_max_f64 %0, %1, %0\n v_max_f64 %1, %2, %1\n v_max_f64 %2, %3, %2\n v_max_f64 %3, %0, %3", C4)
KERNEL(k_mul_f64_mix, MIXDECL(double), "v_mul_f64 %0, %1, %0\n v_xor_b32 %4, %5, %4\n v_mul_f64 %2, %3, %2\n v_xor_b32 %6, %7, %6", C8)
// scalar unit: one per CU, shared by the four SIMDs
#define SDECL uint32_t a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3; uint32_t s0 = iters, s1 = iters + 1, s2 = iters + 2, s3 = iters + 3
#define CS "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
KERNEL(k_s_mov, SDECL, "s_mov_b32 %0, %1\n s_mov_b32 %1, %2\n s_mov_b32 %2, %3\n s_mov_b32 %3, %0", CS)
KERNEL(k_s_movlit, SDECL, "s_mov_b32 %0, 0x12345678\n s_mov_b32 %1, 0x23456789\n s_mov_b32 %2, 0x3456789a\n s_mov_b32 %3, 0x456789ab", CS)
KERNEL(k_s_add, SDECL, "s_add_i32 %0, %1, %0\n s_add_i32 %1, %2, %1\n s_add_i32 %2, %3, %2\n s_add_i32 %3, %0, %3", CS)
KERNEL(k_s_mul, SDECL, "s_mul_i32 %0, %1, %0\n s_mul_i32 %1, %2, %1\n s_mul_i32 %2, %3, %2\n s_mul_i32 %3, %0, %3", CS)
KERNEL(k_sv_mix, SDECL, "s_mov_b32 %0, 0x12345678\n v_xor_b32 %4, %5, %4\n s_mov_b32 %2, 0x3456789a\n v_xor_b32 %6, %7, %6", CS)
KERNEL(k_sv_mix31, SDECL, "s_mov_b32 %0, 0x12345678\n v_xor_b32 %4, %5, %4\n v_xor_b32 %5, %6, %5\n v_xor_b32 %6, %7, %6", CS)
KERNEL(k_alignbit, U32DECL, "v_alignbit_b32 %0, %1, %0, 11\n v_alignbit_b32 %1, %2, %1, 11\n v_alignbit_b32 %2, %3, %2, 11\n v_alignbit_b32 %3, %0, %3, 11", C4)
KERNEL(k_dpp, U32DECL, "v_mov_b32_dpp %0, %1 wave_shr:1\n v_mov_b32_dpp %1, %2 wave_shr:1\n v_mov_b32_dpp %2, %3 wave_shr:1\n v_mov_b32_dpp %3, %0 wave_shr:1", C4)
KERNEL(k_mul_lo, U32DECL, "v_mul_lo_u32 %0, %1, %0\n v_mul_lo_u32 %1, %2, %1\n v_mul_lo_u32 %2, %3, %2\n v_mul_lo_u32 %3, %0, %3", C4)
KERNEL(k_mul_hi, U32DECL, "v_mul_hi_u32 %0, %1, %0\n v_mul_hi_u32 %1, %2, %1\n v_mul_hi_u32 %2, %3, %2\n v_mul_hi_u32 %3, %0, %3", C4)
KERNEL(k_mul_u24, U32DECL, "v_mul_u32_u24 %0, %1, %0\n v_mul_u32_u24 %1, %2, %1\n v_mul_u32_u24 %2, %3, %2\n v_mul_u32_u24 %3, %0, %3", C4)
KERNEL(k_fma_f32, U32DECL, "v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %1, %2, %1, %1\n v_fma_f32 %2, %3, %2, %2\n v_fma_f32 %3, %0, %3, %3", C4)
KERNEL(k_mad_u64, MIXDECL(uint64_t), "v_mad_u64_u32 %0, vcc, %4, %5, 0\n v_mad_u64_u32 %1, vcc, %5, %6, 0\n v_mad_u64_u32 %2, vcc, %6, %7, 0\n v_mad_u64_u32 %3, vcc, %7, %4, 0", C8)
KERNEL(k_fma_f64, F64DECL, "v_fma_f64 %0, %1, %0, %0\n v_fma_f64 %1, %2, %1, %1\n v_fma_f64 %2, %3, %2, %2\n v_fma_f64 %3, %0, %3, %3", C4)
KERNEL(k_mul_f64, F64DECL, "v_mul_f64 %0, %1, %0\n v_mul_f64 %1, %2, %1\n v_mul_f64 %2, %3, %2\n v_mul_f64 %3, %0, %3", C4)
KERNEL(k_add_f64, F64DECL, "v_add_f64 %0, %1, %0\n v_add_f64 %1, %2, %1\n v_add_f64 %2, %3, %2\n v_add_f64 %3, %0, %3", C4)
KERNEL(k_rsq_f64, F64DECL, "v_rsq_f64 %0, %1\n v_rsq_f64 %1, %2\n v_rsq_f64 %2, %3\n v_rsq_f64 %3, %0", C4)
KERNEL(k_rcp_f64, F64DECL, "v_rcp_f64 %0, %1\n v_rcp_f64 %1, %2\n v_rcp_f64 %2, %3\n v_rcp_f64 %3, %0", C4)
KERNEL(k_ldexp_f64, F64DECL, "v_ldexp_f64 %0, %1, 3\n v_ldexp_f64 %1, %2, 3\n v_ldexp_f64 %2, %3, 3\n v_ldexp_f64 %3, %0, 3", C4)
KERNEL(k_cvt_f64_u32, MIXDECL(double), "v_cvt_f64_u32 %0, %4\n v_cvt_f64_u32 %1, %5\n v_cvt_f64_u32 %2, %6\n v_cvt_f64_u32 %3, %7", C8)
KERNEL(k_cvt_i32_f64, MIXDECL(double), "v_cvt_i32_f64 %4, %0\n v_cvt_i32_f64 %5, %1\n v_cvt_i32_f64 %6, %2\n v_cvt_i32_f64 %7, %3", C8)
KERNEL(k_cvt_f32_f64, MIXDECL(double), "v_cvt_f32_f64 %4, %0\n v_cvt_f32_f64 %5, %1\n v_cvt_f32_f64 %6, %2\n v_cvt_f32_f64 %7, %3", C8)
KERNEL(k_log_f32, U32DECL, "v_log_f32 %0, %1\n v_log_f32 %1, %2\n v_log_f32 %2, %3\n v_log_f32 %3, %0", C4)
KERNEL(k_sin_f32, U32DECL, "v_sin_f32 %0, %1\n v_sin_f32 %1, %2\n v_sin_f32 %2, %3\n v_sin_f32 %3, %0", C4)
KERNEL(k_pk_fma_f32, U64DECL, "v_pk_fma_f32 %0, %1, %0, %0\n v_pk_fma_f32 %1, %2, %1, %1\n v_pk_fma_f32 %2, %3, %2, %2\n v_pk_fma_f32 %3, %0, %3, %3", C4)

struct Entry {
  const char *name;
  void (*fn)(int, double *);
};

int main(int argc, char **argv)
{
  const int iters  = argc > 1 ? atoi(argv[1]) : 200;
  const int wpsimd = argc > 2 ? atoi(argv[2]) : 8; // wavefronts per SIMD
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  double   *sink;
  (void)hipMalloc(&sink, 4096);
  const Entry tab[] = {{"v_mov_b32", k_mov}, {"v_xor_b32", k_xor}, {"v_add_u32", k_add}, {"v_cndmask_b32", k_cndmask}, {"v_cndmask(sgpr)", k_cndmask_sgpr}, {"cmp+cndmask /2", k_cmp_cndmask}, {"v_cmp_gt_u32", k_cmp}, {"1cnd+3xor", k_cndmask_mix}, {"v_bfi_b32", k_bfi}, {"v_and_or_b32", k_and_or}, {"v_lshl_add_u32", k_lshl_add}, {"v_lshlrev_b32", k_lshlrev}, {"s_mov_b32", k_s_mov}, {"s_mov_b32 lit", k_s_movlit}, {"s_add_i32", k_s_add}, {"s_mul_i32", k_s_mul}, {"s_mov+v_xor", k_sv_mix}, {"1 s_mov+3 v_xor", k_sv_mix31}, {"v_alignbit_b32", k_alignbit}, {"v_mov_b32_dpp", k_dpp}, {"v_fma_f32", k_fma_f32}, {"v_pk_fma_f32", k_pk_fma_f32}, {"v_mul_u32_u24", k_mul_u24}, {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi}, {"v_mad_u64_u32", k_mad_u64}, {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}, {"v_fmac_f64", k_fmac_f64}, {"v_max_f64", k_max_f64}, {"mul_f64+xor", k_mul_f64_mix}, {"v_ldexp_f64", k_ldexp_f64}, {"v_cvt_f64_u32", k_cvt_f64_u32}, {"v_cvt_i32_f64", k_cvt_i32_f64}, {"v_cvt_f32_f64", k_cvt_f32_f64}, {"v_rsq_f64", k_rsq_f64}, {"v_rcp_f64", k_rcp_f64}, {"v_log_f32", k_log_f32}, {"v_sin_f32", k_sin_f32}};
  hipEvent_t  e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  // blocks of 256 threads = one wavefront per SIMD of a CU; wpsimd blocks per CU
  const dim3 grid(cus * wpsimd), block(256);
  double     base = 0;
  printf("# %s, %d CUs, %d wavefronts per SIMD, %d x 256 instructions per wavefront\n", prop.gcnArchName, cus, wpsimd, iters);
  printf("%-16s %10s %8s\n", "instruction", "ns/instr", "rel");
  for (const Entry &e : tab) {
    hipLaunchKernelGGL(e.fn, grid, block, 0, 0, iters, sink); // warm-up
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(e.fn, grid, block, 0, 0, iters, sink);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    // one SIMD issues wpsimd wavefronts x iters x 256 instructions during the launch
    const double ns = best * 1e6 / ((double)wpsimd * iters * 256.0);
    if (base == 0) base = ns;
    printf("%-16s %10.3f %8.2f\n", e.name, ns, ns / base);
  }
  return 0;
}
