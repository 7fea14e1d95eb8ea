// Low-rank (Woodbury) pieces of the samplers on MATLRC operators A + B S B^T, B dense N x k (gfx950).
//
// Replace the dense PETSc products of the reference: MatMultTranspose(B, y, w) / MatMult(Bb, w, z) / VecAXPY in
// MCSORPostSOR_LRC (src/mc_sor.c:101-112) and PCSORGibbsSample (src/pc_sorgibbs.c:97-101), MatMultAdd(B, wk, w, w)
// in PrepareRHS_LRC (src/pc_mcgibbs.c:130-140, src/pc_sorgibbs.c:86-90), and MatMatMult(C, Sb) of
// MCSORBuildLRCCorrection (src/mc_sor.c:535).  k is small (3..17): every kernel streams the N x k matrix once,
// column-major with leading dimension ld, rows in the sampler's storage layout -- 8 N k bytes, HBM bound.
#include <hip/hip_runtime.h>
#include "pmg_kernels.h"

namespace {

// partial[block*k + c] = sum over the block's rows of M[r + ld*c] * y[r]
__global__ __launch_bounds__(256) void lrc_btx_partial_kernel(int64_t n, int k, const double *__restrict__ M, int64_t ld, const double *__restrict__ y, double *__restrict__ partial)
{
  __shared__ double red[4];
  const int64_t r0 = (int64_t)blockIdx.x * 4096;
  for (int c = 0; c < k; ++c) {
    double s = 0.0;
    for (int64_t r = r0 + threadIdx.x; r < r0 + 4096 && r < n; r += 256) s = fma(M[r + ld * c], y[r], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * k + c] = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
  }
}

// out[c] = scale[c] * sum_b partial[b*k + c]  (fixed order: bit-reproducible); scale may be null
__global__ void lrc_reduce_kernel(int nblocks, int k, const double *__restrict__ partial, const double *__restrict__ scale, double *__restrict__ out)
{
  const int c = threadIdx.x;
  if (c >= k) return;
  double s = 0.0;
  for (int b = 0; b < nblocks; ++b) s += partial[(int64_t)b * k + c];
  out[c] = scale ? scale[c] * s : s;
}

// out[r] = in[r] + sign * sum_c M[r + ld*c] * coef[c]
__global__ __launch_bounds__(256) void lrc_axpy_cols_kernel(int64_t n, int k, const double *__restrict__ M, int64_t ld, const double *__restrict__ coef, double sign, const double *__restrict__ in, double *__restrict__ out)
{
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  double s = 0.0;
  for (int c = 0; c < k; ++c) s = fma(M[r + ld * c], coef[c], s);
  out[r] = in[r] + sign * s;
}

// Bb[r + ld*c] = sum_j C[r + ld*j] * Sb[j + k*c]
__global__ __launch_bounds__(256) void lrc_gemm_small_kernel(int64_t n, int k, const double *__restrict__ Cm, int64_t ld, const double *__restrict__ Sb, double *__restrict__ Bb)
{
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  for (int c = 0; c < k; ++c) {
    double s = 0.0;
    for (int j = 0; j < k; ++j) s = fma(Cm[r + ld * j], Sb[j + k * c], s);
    Bb[r + ld * c] = s;
  }
}

__global__ void lrc_mul_kernel(int k, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ out)
{
  const int c = threadIdx.x;
  if (c < k) out[c] = a[c] * b[c];
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? 0 : 1; }

} // namespace

extern "C" int pmgk_lrc_nblocks(int64_t n) { return (int)((n + 4095) / 4096); }

extern "C" int pmgk_lrc_btx(int64_t n, int k, const double *M, int64_t ld, const double *y, double *partial, const double *scale, double *out, void *stream)
{
  if (n <= 0 || k <= 0) return 0;
  const int nb = pmgk_lrc_nblocks(n);
  hipLaunchKernelGGL(lrc_btx_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, n, k, M, ld, y, partial);
  hipLaunchKernelGGL(lrc_reduce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, nb, k, partial, scale, out);
  return launch_status();
}

extern "C" int pmgk_lrc_axpy_cols(int64_t n, int k, const double *M, int64_t ld, const double *coef, double sign, const double *in, double *out, void *stream)
{
  if (n <= 0) return 0;
  hipLaunchKernelGGL(lrc_axpy_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, k, M, ld, coef, sign, in, out);
  return launch_status();
}

extern "C" int pmgk_lrc_gemm_small(int64_t n, int k, const double *Cm, int64_t ld, const double *Sb, double *Bb, void *stream)
{
  if (n <= 0) return 0;
  hipLaunchKernelGGL(lrc_gemm_small_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, k, Cm, ld, Sb, Bb);
  return launch_status();
}

extern "C" int pmgk_lrc_mul(int k, const double *a, const double *b, double *out, void *stream)
{
  hipLaunchKernelGGL(lrc_mul_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, k, a, b, out);
  return launch_status();
}
