// Multicolour Gibbs/SOR sweep on a colour-partitioned sliced-ELL copy of an AIJ matrix (gfx950).
//
// Replaces MCSORApply_SEQAIJ (reference src/mc_sor.c:241-296), the row kernel of PCPARSOR
// (SORLocalForwardSweepIS, reference src/pc_parsor.c:666-701) and PETSc's MatSOR forward sweep as called at
// reference src/pc_sorgibbs.c:94, for any matrix and any valid colouring, with the noisy right-hand side of
// PrepareRHS_Default (reference src/pc_mcgibbs.c:119-128) formed in registers.
//
// Layout: rows are renumbered so that the rows of one colour are contiguous and start on a 64-row slice
// boundary; one wavefront owns one slice and streams its off-diagonal entries column-major, so that the 64
// lanes read 64 consecutive doubles (vals) and 64 consecutive ints (cols) per step -- fully coalesced -- and
// only the gather y[col] is irregular.  Entries keep the CSR storage order of their row (lower part, then
// upper part) so the sum is bit-identical to the reference loop; pad entries have value 0 and point at the
// row itself.  Vectors handled here are in the permuted numbering.
#include <hip/hip_runtime.h>
#include <cstring>
#include "pmg_kernels.h"
#define PMG_RNG_LITERALS // the transform's constants as literals here: scalar loads in the middle of these kernels' sums cost more than they save (st27 phase +9 % by GRBM_GUI_ACTIVE)
#define PMG_RNG_TU csr
#include "pmg_rng.hpp"

namespace {

// Off-diagonal part of one row: sum -= a_j y[c_j] in storage order.  The entries are fetched in batches of SELL_BATCH:
// all value / column loads of a batch are issued together, then all gathers y[c_j], and only then the dependent
// subtractions run.  A plain `for j` loop compiles to load c -> wait -> load y[c] -> wait per entry, i.e. 2 w serial
// memory latencies per row -- a 63 000-row colour of the refined L-shape mesh took 7 us, all of it latency.
// The slice width w is wave-uniform, so the guards are scalar branches; a batch reads the slice's last column again
// instead of running past it (a valid address whose value is not used).  Same terms, same order: same bits.
constexpr int SELL_BATCH = 8;
__device__ __forceinline__ double sell_row_sum(double sum, int w, const double *__restrict__ v, const int32_t *__restrict__ c, const double *y)
{
  for (int j0 = 0; j0 < w; j0 += SELL_BATCH) {
    double  a[SELL_BATCH], yv[SELL_BATCH];
    int32_t cj[SELL_BATCH];
#pragma unroll
    for (int q = 0; q < SELL_BATCH; ++q) {
      const int64_t jj = (int64_t)min(j0 + q, w - 1) * 64;
      a[q]             = v[jj];
      cj[q]            = c[jj];
    }
#pragma unroll
    for (int q = 0; q < SELL_BATCH; ++q) yv[q] = y[cj[q]];
#pragma unroll
    for (int q = 0; q < SELL_BATCH; ++q)
      if (j0 + q < w) sum = sum - a[q] * yv[q];
  }
  return sum;
}

template <bool NOISY>
__global__ __launch_bounds__(256) void sell_color_sweep_kernel(pmgk_sell S, int slice0, int nsl, double omega, double one_minus_omega, uint32_t key0, uint32_t key1, uint64_t sweep, const double *__restrict__ b, double *y)
{
  __shared__ pmg::LogTabEntry s_logtab[NOISY ? 4 * PMG_LOGTAB_SIZE : 1];
  const int                   lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int                   sl   = blockIdx.x * (blockDim.x >> 6) + wv;
  if (sl >= nsl) return;
  const pmg::LogTabEntry *tab = s_logtab + (NOISY ? wv * PMG_LOGTAB_SIZE : 0);
  if (NOISY) pmg::load_log_table_wave(s_logtab + wv * PMG_LOGTAB_SIZE, lane); // per wavefront: no block barrier in front of the row loads
  const int     s   = slice0 + sl;
  const int     row = s * 64 + lane;
  const int64_t off = S.soff[s];
  const int     w   = __builtin_amdgcn_readfirstlane(S.swidth[s]);
  const int     org = S.orig[row];
  double        sum = b[row];
  const double  yold = y[row], idg = S.idiag[row]; // issued with the first batch, not behind the dependent chain
  if (NOISY) {
    // row stream: entries (2q, 2q+1) of the ORIGINAL numbering share one Box-Muller pair
    const uint32_t uorg = org < 0 ? 0u : (uint32_t)(S.noise_row0 + org); // global row of a row block
    double         z0, z1;
    pmg::normal_pair(uorg >> 1, 0u, (uint32_t)sweep, (uint32_t)(sweep >> 32), key0, key1, tab, z0, z1);
    const double xi = (uorg & 1u) ? z1 : z0;
    sum             = xi * S.sqrtdiag[row] + sum;
  }
  sum = sell_row_sum(sum, w, S.vals + off + lane, S.cols + off + lane, y);
  if (org >= 0) y[row] = one_minus_omega * yold + idg * sum;
  (void)omega;
}

// r = b - A y in the permuted numbering; the diagonal term is added last (the reference's residual goes
// through PETSc MatMult whose summation order is storage order; differences are O(eps)).
__global__ __launch_bounds__(256) void sell_residual_kernel(pmgk_sell S, const double *__restrict__ b, const double *__restrict__ y, double *__restrict__ r)
{
  const int lane = threadIdx.x & 63;
  const int s    = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (s >= S.nslices) return;
  const int      row = s * 64 + lane;
  const int64_t  off = S.soff[s];
  const int      w   = __builtin_amdgcn_readfirstlane(S.swidth[s]);
  const double  *v   = S.vals + off + lane;
  const int32_t *c   = S.cols + off + lane;
  double         sum = 0.0;
  for (int j0 = 0; j0 < w; j0 += SELL_BATCH) { // batched like sell_row_sum: loads first, the dependent sum afterwards
    double  a[SELL_BATCH], yv[SELL_BATCH];
    int32_t cj[SELL_BATCH];
#pragma unroll
    for (int q = 0; q < SELL_BATCH; ++q) {
      const int64_t jj = (int64_t)min(j0 + q, w - 1) * 64;
      a[q]             = v[jj];
      cj[q]            = c[jj];
    }
#pragma unroll
    for (int q = 0; q < SELL_BATCH; ++q) yv[q] = y[cj[q]];
#pragma unroll
    for (int q = 0; q < SELL_BATCH; ++q)
      if (j0 + q < w) sum = sum + a[q] * yv[q];
  }
  sum    = sum + S.diag[row] * y[row];
  r[row] = S.orig[row] >= 0 ? b[row] - sum : 0.0;
}

__global__ void permute_in_kernel(int32_t ld, const int32_t *__restrict__ orig, const double *__restrict__ nat, double *__restrict__ perm)
{
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= ld) return;
  const int o = orig[r];
  perm[r]     = o >= 0 ? nat[o] : 0.0;
}

__global__ void permute_out_kernel(int32_t ld, const int32_t *__restrict__ orig, const double *__restrict__ perm, double *__restrict__ nat)
{
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= ld) return;
  const int o = orig[r];
  if (o >= 0) nat[o] = perm[r];
}

// One wavefront per row group: plain CSR product for the (short-row) transfer operators.
__global__ __launch_bounds__(256) void csr_spmv_kernel(int32_t nrows, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const double *__restrict__ vals, double alpha, const double *__restrict__ x, double beta, double *__restrict__ y)
{
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  double sum = 0.0;
  for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) sum = sum + vals[k] * x[colidx[k]];
  y[r] = beta == 0.0 ? alpha * sum : alpha * sum + beta * y[r];
}

template <bool ACC>
__global__ __launch_bounds__(256) void csr_spmv_rows_kernel(int32_t nrows, const int32_t *__restrict__ rowpos, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const double *__restrict__ vals, const double *__restrict__ x, double *__restrict__ y, double *__restrict__ zero)
{
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  // entries in batches of SELL_BATCH like sell_row_sum: all value / column loads of a batch, then all gathers, then the sum in
  // storage order (same terms, same order, same bits).  The plain loop ran load -> gather -> add per entry: two dependent
  // fetches per ENTRY of a row of P^T (a dozen entries on an aggregation hierarchy), in a kernel that lives for 5 us
  const int k0 = rowptr[r], k1 = rowptr[r + 1];
  const int o  = rowpos[r];
  double    yo = 0.0;
  if (ACC) yo = y[o];
  double sum = 0.0;
  for (int kb = k0; kb < k1; kb += SELL_BATCH) {
    double  a[SELL_BATCH], xv[SELL_BATCH];
    int32_t cj[SELL_BATCH];
#pragma unroll
    for (int q = 0; q < SELL_BATCH; ++q) {
      const int kk = min(kb + q, k1 - 1);
      a[q]         = vals[kk];
      cj[q]        = colidx[kk];
    }
#pragma unroll
    for (int q = 0; q < SELL_BATCH; ++q) xv[q] = x[cj[q]];
#pragma unroll
    for (int q = 0; q < SELL_BATCH; ++q)
      if (kb + q < k1) sum = sum + a[q] * xv[q];
  }
  y[o] = ACC ? yo + sum : sum;
  if (zero) zero[o] = 0.0; // the restriction of a V-cycle also sets the coarse level's zero guess (a fill kernel less per level)
}

__global__ void axpy_kernel(int64_t n, double alpha, const double *__restrict__ x, double *__restrict__ y)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = y[i] + alpha * x[i];
}

/* SCALED: xi = scale o z in the same launch (VecSetRandomStandardNormal + VecPointwiseMult of the low-rank noise terms,
   src/pc_mcgibbs.c:130-134): the product a separate kernel would form, one dependent launch fewer */
template <bool SCALED>
__global__ void fill_normal_rows_kernel(int64_t n, uint32_t key0, uint32_t key1, uint64_t sweep, const double *__restrict__ scale, double *__restrict__ xi)
{
  __shared__ pmg::LogTabEntry s_logtab[PMG_LOGTAB_SIZE];
  pmg::load_log_table(s_logtab);
  __syncthreads();
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * q >= n) return;
  double s0 = 1., s1 = 1.;
  if (SCALED) {
    s0 = scale[2 * q];
    if (2 * q + 1 < n) s1 = scale[2 * q + 1];
  }
  double z0, z1;
  pmg::normal_pair((uint32_t)q, (uint32_t)((uint64_t)q >> 32), (uint32_t)sweep, (uint32_t)(sweep >> 32), key0, key1, s_logtab, z0, z1);
  if (SCALED) {
    z0 *= s0;
    z1 *= s1;
  }
  xi[2 * q] = z0;
  if (2 * q + 1 < n) xi[2 * q + 1] = z1;
}

// Several row streams in ONE launch: block s draws the n normals of the stream (seed[s], sweep[s]) exactly as
// fill_normal_rows_kernel<true> does -- same counters, same pairing, same product with `scale` -- into xi + s * stride.  The
// low-rank noise terms of a whole V-cycle sample (one draw of k numbers per directional sweep and level, src/pc_mcgibbs.c:
// 130-134) are known when the cycle starts; drawn here together, they cost one dependent launch instead of one per sweep.
struct normal_batch {
  uint64_t seed[PMGK_NORMAL_BATCH_MAX], sweep[PMGK_NORMAL_BATCH_MAX];
};
__global__ void fill_normal_batch_kernel(normal_batch B, int64_t n, const double *__restrict__ scale, double *__restrict__ xi, int64_t stride)
{
  __shared__ pmg::LogTabEntry s_logtab[PMG_LOGTAB_SIZE];
  pmg::load_log_table(s_logtab);
  __syncthreads();
  const uint64_t seed = B.seed[blockIdx.x], sweep = B.sweep[blockIdx.x];
  double        *out  = xi + (int64_t)blockIdx.x * stride;
  for (int64_t q = threadIdx.x; 2 * q < n; q += blockDim.x) {
    const double s0 = scale[2 * q], s1 = 2 * q + 1 < n ? scale[2 * q + 1] : 1.;
    double       z0, z1;
    pmg::normal_pair((uint32_t)q, (uint32_t)((uint64_t)q >> 32), (uint32_t)sweep, (uint32_t)(sweep >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), s_logtab, z0, z1);
    out[2 * q] = z0 * s0;
    if (2 * q + 1 < n) out[2 * q + 1] = z1 * s1;
  }
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? 0 : 1; }

} // namespace

// xi[s * stride + i] = scale[i] * z_i(seed[s], sweep[s]), i < n, for s < nstreams <= PMGK_NORMAL_BATCH_MAX (host arrays of seeds and sweeps)
extern "C" int pmgk_fill_normal_batch(int nstreams, int64_t n, const uint64_t *seed, const uint64_t *sweep, const double *scale, double *xi, int64_t stride, void *stream)
{
  if (nstreams <= 0 || n <= 0) return 0;
  if (nstreams > PMGK_NORMAL_BATCH_MAX) return 1;
  normal_batch B;
  memset(&B, 0, sizeof B);
  for (int s = 0; s < nstreams; ++s) {
    B.seed[s]  = seed[s];
    B.sweep[s] = sweep[s];
  }
  hipLaunchKernelGGL(fill_normal_batch_kernel, dim3(nstreams), dim3(64), 0, (hipStream_t)stream, B, n, scale, xi, stride);
  return launch_status();
}

extern "C" int pmgk_sell_color_sweep(const pmgk_sell *S, int slice0, int nsl, double omega, int noisy, uint64_t seed, uint64_t sweep, const double *b, double *y, void *stream)
{
  if (nsl <= 0) return 0;
  const dim3 block(256), grid((nsl + 3) / 4);
  const double om1 = 1. - omega;
  if (noisy) hipLaunchKernelGGL((sell_color_sweep_kernel<true>), grid, block, 0, (hipStream_t)stream, *S, slice0, nsl, omega, om1, (uint32_t)seed, (uint32_t)(seed >> 32), sweep, b, y);
  else hipLaunchKernelGGL((sell_color_sweep_kernel<false>), grid, block, 0, (hipStream_t)stream, *S, slice0, nsl, omega, om1, (uint32_t)seed, (uint32_t)(seed >> 32), sweep, b, y);
  return launch_status();
}

extern "C" int pmgk_sell_residual(const pmgk_sell *S, const double *b, const double *y, double *r, void *stream)
{
  if (S->nslices <= 0) return 0;
  hipLaunchKernelGGL(sell_residual_kernel, dim3((S->nslices + 3) / 4), dim3(256), 0, (hipStream_t)stream, *S, b, y, r);
  return launch_status();
}

extern "C" int pmgk_permute_in(int32_t ld, const int32_t *orig, const double *nat, double *perm, void *stream)
{
  if (ld <= 0) return 0;
  hipLaunchKernelGGL(permute_in_kernel, dim3((ld + 255) / 256), dim3(256), 0, (hipStream_t)stream, ld, orig, nat, perm);
  return launch_status();
}

extern "C" int pmgk_permute_out(int32_t ld, const int32_t *orig, const double *perm, double *nat, void *stream)
{
  if (ld <= 0) return 0;
  hipLaunchKernelGGL(permute_out_kernel, dim3((ld + 255) / 256), dim3(256), 0, (hipStream_t)stream, ld, orig, perm, nat);
  return launch_status();
}

extern "C" int pmgk_csr_spmv(int32_t nrows, const int32_t *rowptr, const int32_t *colidx, const double *vals, double alpha, const double *x, double beta, double *y, void *stream)
{
  if (nrows <= 0) return 0;
  hipLaunchKernelGGL(csr_spmv_kernel, dim3((nrows + 255) / 256), dim3(256), 0, (hipStream_t)stream, nrows, rowptr, colidx, vals, alpha, x, beta, y);
  return launch_status();
}

extern "C" int pmgk_csr_spmv_rows(int32_t nrows, const int32_t *rowpos, const int32_t *rowptr, const int32_t *colidx, const double *vals, const double *x, double *y, int accumulate, double *zero, void *stream)
{
  if (nrows <= 0) return 0;
  const dim3 grid((nrows + 255) / 256), block(256);
  if (accumulate) hipLaunchKernelGGL((csr_spmv_rows_kernel<true>), grid, block, 0, (hipStream_t)stream, nrows, rowpos, rowptr, colidx, vals, x, y, zero);
  else hipLaunchKernelGGL((csr_spmv_rows_kernel<false>), grid, block, 0, (hipStream_t)stream, nrows, rowpos, rowptr, colidx, vals, x, y, zero);
  return launch_status();
}

extern "C" int pmgk_axpy(int64_t n, double alpha, const double *x, double *y, void *stream)
{
  if (n <= 0) return 0;
  hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, alpha, x, y);
  return launch_status();
}

extern "C" int pmgk_fill_normal_rows(int64_t n, uint64_t seed, uint64_t sweep, double *xi, void *stream)
{
  if (n <= 0) return 0;
  const int64_t pairs = (n + 1) / 2;
  hipLaunchKernelGGL((fill_normal_rows_kernel<false>), dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, (uint32_t)seed, (uint32_t)(seed >> 32), sweep, (const double *)nullptr, xi);
  return launch_status();
}

extern "C" int pmgk_fill_normal_rows_scaled(int64_t n, uint64_t seed, uint64_t sweep, const double *scale, double *xi, void *stream)
{
  if (n <= 0) return 0;
  const int64_t pairs = (n + 1) / 2;
  hipLaunchKernelGGL((fill_normal_rows_kernel<true>), dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, (uint32_t)seed, (uint32_t)(seed >> 32), sweep, scale, xi);
  return launch_status();
}
