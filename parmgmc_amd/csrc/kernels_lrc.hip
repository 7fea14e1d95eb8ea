// Low-rank (Woodbury) pieces of the samplers on MATLRC operators A + B S B^T, B dense N x k (gfx950).
//
// Replace the dense PETSc products of the reference: MatMultTranspose(B, y, w) / MatMult(Bb, w, z) / VecAXPY in
// MCSORPostSOR_LRC (src/mc_sor.c:101-112) and PCSORGibbsSample (src/pc_sorgibbs.c:97-101), MatMultAdd(B, wk, w, w)
// in PrepareRHS_LRC (src/pc_mcgibbs.c:130-140, src/pc_sorgibbs.c:86-90), and MatMatMult(C, Sb) of
// MCSORBuildLRCCorrection (src/mc_sor.c:535).  k is small (3..17): every kernel streams the N x k matrix once,
// column-major with leading dimension ld, rows in the sampler's storage layout -- 8 N k bytes, HBM bound.
#include <hip/hip_runtime.h>
#include "pmg_kernels.h"
#define PMG_RNG_LITERALS // the transform's constants as literals here: scalar loads in the middle of these kernels' sums cost more than they save (st27 phase +9 % by GRBM_GUI_ACTIVE)
#define PMG_RNG_TU lrc
#include "pmg_rng.hpp"

// rows per thread of the row-compact B^T y kernels: a block of 256 threads sums 256 * PMG_LRC_RPT support rows.  Round 3 used 16
// (100 blocks for the 407 k support rows of the 257^3 level: 100 of 256 CUs, 16 dependent gathers per thread); 4 gives four
// times the blocks and shorter chains: 257^3 k = 3 sample 0.823 -> 0.798 ms, k = 17 1.358 -> 1.287 ms (same box).  The
// grouping of the partial sums changes the rounding of B^T y, not its order within a block.
#ifndef PMG_LRC_RPT
#define PMG_LRC_RPT 4
#endif

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

// out[c] = scale[c] * sum_b partial[b*k + c]; scale may be null.  One wavefront per column: lane l adds the blocks
// l, l + 64, ... in that order, then the lanes are combined by a fixed shuffle tree -- a fixed order for a given block
// count, so the result is reproducible (and identical on every rank of a replicated level), and 4000 partial sums of a
// dense 257^3 column no longer take one thread 146 us.
__global__ __launch_bounds__(64) void lrc_reduce_kernel(int nblocks, int k, const double *__restrict__ partial, const double *__restrict__ scale, double *__restrict__ out)
{
  const int c = blockIdx.x, lane = threadIdx.x;
  if (c >= k) return;
  double s = 0.0;
  for (int b = lane; b < nblocks; b += 64) s += partial[(int64_t)b * k + c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) out[c] = scale ? scale[c] * s : s;
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

// ---- row-compact form: the observation vectors are supported on small balls (reference src/obs.c:39-50), and one
// sweep from a zero guess spreads that support by one layer per colour only, so B, Bb are stored for the ns << N
// support rows: Mc[q + ns*c] = M[rows[q] + ld*c].  8 ns k bytes per pass instead of 8 N k.
__global__ __launch_bounds__(256) void lrc_mark_rows_kernel(int64_t n, int k, const double *__restrict__ A0, const double *__restrict__ A1, const double *__restrict__ A2, int64_t ld, unsigned char *__restrict__ mask)
{
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  bool nz = false;
  for (int c = 0; c < k; ++c) nz = nz || A0[r + ld * c] != 0.0 || A1[r + ld * c] != 0.0 || A2[r + ld * c] != 0.0;
  mask[r] = nz ? 1 : 0;
}

__global__ __launch_bounds__(256) void lrc_gather_rows_kernel(int64_t ns, int k, const double *__restrict__ M, int64_t ld, const int64_t *__restrict__ rows, double *__restrict__ Mc)
{
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= ns) return;
  const int64_t r = rows[q];
  for (int c = 0; c < k; ++c) Mc[q + ns * c] = M[r + ld * c];
}

// partial[block*k + c] = sum over the block's 256 * PMG_LRC_RPT compact rows q of Mc[q + ns*c] * y[rows[q]].  A thread owns PMG_LRC_RPT rows
// (q0 + tid + 256 i): their positions and y values are fetched ONCE, all 32 loads in flight, and reused for every column;
// per column the 16 factor loads are independent too.  (Round 1 walked the rows once per column with a load -> gather ->
// fma chain per row: 100 us per call at 257^3 for k = 3, 0.4 of the 1.16 ms low-rank V-cycle sample.)  The order of
// every sum is unchanged (rows ascending per thread, the same wave and block reductions): same bits.
// RESTORE: w[rows[q]] = save[q] in the same pass (lrc_scatter_rows_kernel: the right-hand side entries under the noise term go
// back behind the sweep) -- a load and a store per row beside the sums, on nobody's dependency chain.
template <bool RESTORE>
__global__ __launch_bounds__(256) void lrc_btx_rows_partial_kernel(int64_t ns, int k, const double *__restrict__ Mc, const int64_t *__restrict__ rows, const double *__restrict__ y, double *__restrict__ partial, const double *__restrict__ save, double *__restrict__ w)
{
  __shared__ double red[64][4];
  const int64_t q0 = (int64_t)blockIdx.x * (256 * PMG_LRC_RPT) + threadIdx.x;
  // three dependent fetches (positions -> y at the positions; the factors) from cold memory are what this kernel takes its
  // time for: the first column's factors are requested together with the positions, in front of the gathers that wait for
  // them, and every further column in front of the sum of the one before (round 4)
  int64_t rr[PMG_LRC_RPT];
  double  yv[PMG_LRC_RPT], m[PMG_LRC_RPT];
#pragma unroll
  for (int i = 0; i < PMG_LRC_RPT; ++i) {
    const int64_t q = q0 + 256 * i;
    rr[i]           = q < ns ? rows[q] : -1;
  }
#pragma unroll
  for (int i = 0; i < PMG_LRC_RPT; ++i) {
    const int64_t q = q0 + 256 * i;
    m[i]            = q < ns ? Mc[q] : 0.0;
  }
  double sv[PMG_LRC_RPT];
  if (RESTORE) {
#pragma unroll
    for (int i = 0; i < PMG_LRC_RPT; ++i) {
      const int64_t q = q0 + 256 * i;
      sv[i]           = q < ns ? save[q] : 0.0;
    }
  }
#pragma unroll
  for (int i = 0; i < PMG_LRC_RPT; ++i) yv[i] = rr[i] >= 0 ? y[rr[i]] : 0.0;
  if (RESTORE) {
#pragma unroll
    for (int i = 0; i < PMG_LRC_RPT; ++i)
      if (rr[i] >= 0) w[rr[i]] = sv[i];
  }
  for (int c0 = 0; c0 < k; c0 += 64) { // k <= 64 in practice: one round
    const int kc = min(64, k - c0);
    for (int c = 0; c < kc; ++c) {
      double mn[PMG_LRC_RPT];
      if (c0 + c + 1 < k) {
        const double *col = Mc + ns * (int64_t)(c0 + c + 1);
#pragma unroll
        for (int i = 0; i < PMG_LRC_RPT; ++i) {
          const int64_t q = q0 + 256 * i;
          mn[i]           = q < ns ? col[q] : 0.0;
        }
      }
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < PMG_LRC_RPT; ++i)
        if (q0 + 256 * i < ns) s = fma(m[i], yv[i], s);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
      if ((threadIdx.x & 63) == 0) red[c][threadIdx.x >> 6] = s;
      if (c0 + c + 1 < k) {
#pragma unroll
        for (int i = 0; i < PMG_LRC_RPT; ++i) m[i] = mn[i];
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < kc) partial[(int64_t)blockIdx.x * k + c0 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    __syncthreads();
  }
}

// v[rows[q]] = v[rows[q]] + sign * sum_c Mc[q + ns*c] * coef[c]; save != null keeps the old value (to undo it exactly)
__global__ __launch_bounds__(256) void lrc_axpy_rows_kernel(int64_t ns, int k, const double *__restrict__ Mc, const int64_t *__restrict__ rows, const double *__restrict__ coef, double sign, double *__restrict__ v, double *__restrict__ save)
{
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= ns) return;
  double s = 0.0;
  for (int c = 0; c < k; ++c) s = fma(Mc[q + ns * c], coef[c], s);
  const int64_t r = rows[q];
  const double  o = v[r];
  if (save) save[q] = o;
  v[r] = o + sign * s;
}

// lrc_reduce_kernel + lrc_axpy_rows_kernel in one launch: EVERY block adds the nb partial sums of the k columns itself, in
// lrc_reduce_kernel's order (one wavefront per column: lane l adds the blocks l, l + 64, ..., then the shuffle tree; scale
// applied to the sum) -- the same bits -- while its own row positions, old values and first factors are already on their way.
// The partial sums are a few KB in L2; the launch this removes is one of three dependent ones per repair.  k <= 64.
__global__ __launch_bounds__(256) void lrc_reduce_axpy_rows_kernel(int64_t ns, int k, const double *__restrict__ Mc, const int64_t *__restrict__ rows, int nb, const double *__restrict__ partial, const double *__restrict__ scale, double sign, double *__restrict__ v, double *__restrict__ save)
{
  __shared__ double s_coef[64];
  const int64_t     q    = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool        live = q < ns;
  const int64_t     r    = live ? rows[q] : 0;
  const double      m0   = live ? Mc[q] : 0.0;
  const int         wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int c = wv; c < k; c += 4) {
    double s = 0.0;
    for (int b = lane; b < nb; b += 64) s += partial[(int64_t)b * k + c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) s_coef[c] = scale ? scale[c] * s : s;
  }
  const double o = live ? v[r] : 0.0;
  __syncthreads();
  if (!live) return;
  double s = fma(m0, s_coef[0], 0.0);
  for (int c = 1; c < k; ++c) s = fma(Mc[q + ns * c], s_coef[c], s);
  if (save) save[q] = o;
  v[r] = o + sign * s;
}

// The repair of a sweep that is followed by a residual (the down leg of the V-cycle): lrc_reduce_axpy_rows_kernel's update
// v[rows] += sign * Mb coef AND, with the new values still in registers, the partial sums of Mc^T v that the residual's low-rank
// term starts with (lrc_btx_rows_partial_kernel on the same rows) -- one pass less over the support rows and one launch less per
// level.  Blocks of 256 * PMG_LRC_RPT rows, a thread owns the rows q0 + 256 i, the sums in lrc_btx_rows_partial_kernel's order:
// partial_out holds the same bits as that kernel's output on the updated vector.  k <= 8.
__global__ __launch_bounds__(256) void lrc_reduce_axpy_btx_rows_kernel(int64_t ns, int k, const double *__restrict__ Mb, const int64_t *__restrict__ rows, int nb, const double *__restrict__ partial_in, double sign, double *__restrict__ v, const double *__restrict__ Mc, double *__restrict__ partial_out)
{
  __shared__ double s_coef[8];
  __shared__ double red[8][4];
  const int64_t     q0 = (int64_t)blockIdx.x * (256 * PMG_LRC_RPT) + threadIdx.x;
  int64_t           rr[PMG_LRC_RPT];
  double            o[PMG_LRC_RPT], m[PMG_LRC_RPT], yv[PMG_LRC_RPT];
#pragma unroll
  for (int i = 0; i < PMG_LRC_RPT; ++i) {
    const int64_t q = q0 + 256 * i;
    rr[i]           = q < ns ? rows[q] : -1;
  }
#pragma unroll
  for (int i = 0; i < PMG_LRC_RPT; ++i) {
    const int64_t q = q0 + 256 * i;
    m[i]            = q < ns ? Mb[q] : 0.0;
  }
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int c = wv; c < k; c += 4) {
    double s = 0.0;
    for (int b = lane; b < nb; b += 64) s += partial_in[(int64_t)b * k + c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) s_coef[c] = s;
  }
#pragma unroll
  for (int i = 0; i < PMG_LRC_RPT; ++i) o[i] = rr[i] >= 0 ? v[rr[i]] : 0.0;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PMG_LRC_RPT; ++i) {
    const int64_t q = q0 + 256 * i;
    double        a = fma(m[i], s_coef[0], 0.0);
    if (q < ns)
      for (int c = 1; c < k; ++c) a = fma(Mb[q + ns * c], s_coef[c], a);
    yv[i] = o[i] + sign * a;
    if (q < ns) v[rr[i]] = yv[i];
  }
  for (int c = 0; c < k; ++c) {
    const double *col = Mc + ns * (int64_t)c;
    double        s   = 0.0;
#pragma unroll
    for (int i = 0; i < PMG_LRC_RPT; ++i) {
      const int64_t q = q0 + 256 * i;
      if (q < ns) s = fma(col[q], yv[i], s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) red[c][wv] = s;
  }
  __syncthreads();
  if ((int)threadIdx.x < k) partial_out[(int64_t)blockIdx.x * k + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

__global__ __launch_bounds__(256) void lrc_scatter_rows_kernel(int64_t ns, const int64_t *__restrict__ rows, const double *__restrict__ save, double *__restrict__ v)
{
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q < ns) v[rows[q]] = save[q];
}

// ---- round 4: the per-sweep chain of the row-compact form in four launches instead of seven -------------------------------
// (profiles/r03_cycles_summary.json: at 257^3, k = 3 the low-rank V-cycle sample was 101 launches, 69 of them these small
// kernels at their launch floor -- +57 % time for three columns on 2 % of the rows).  What was also built and REMOVED: the
// partial sums and their reduction in one launch, the last block to finish adding them up behind a device-wide ticket
// (__threadfence + atomic, as the halo hand-shake of kernels_grid.hip): bit-identical, but 12-22 us per launch against
// 9.5 + 2.5 for the two kernels -- on this chip the fence that publishes a block's sums to the other XCDs' L2s costs more than
// the launch boundary it saves (gpurun_out/r4_lrc_trace.log; the V-cycle sample 0.900 ms fused against 0.879).

// Block-wide sum of the k column sums of one block of 256 * PMG_LRC_RPT compact rows, exactly as lrc_btx_rows_partial_kernel forms them:
// a thread owns the rows q0 + 256 i, i < PMG_LRC_RPT, in ascending order; wavefront shuffle tree; (red0 + red1) + (red2 + red3).
// Thread c < kc holds the block's sum of column c0 + c on return (others: unspecified).  red: [64][4] doubles of LDS.
__device__ __forceinline__ double lrc_block_colsum(int64_t ns, int kc, int64_t col0, const double *__restrict__ Mc, const double (&yv)[PMG_LRC_RPT], int64_t q0, double (*red)[4])
{
  for (int c = 0; c < kc; ++c) {
    const double *col = Mc + ns * (col0 + c);
    double        m[PMG_LRC_RPT];
#pragma unroll
    for (int i = 0; i < PMG_LRC_RPT; ++i) {
      const int64_t q = q0 + 256 * i;
      m[i]            = q < ns ? col[q] : 0.0;
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < PMG_LRC_RPT; ++i)
      if (q0 + 256 * i < ns) s = fma(m[i], yv[i], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[c][threadIdx.x >> 6] = s;
  }
  __syncthreads();
  double r = 0.0;
  if ((int)threadIdx.x < kc) r = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
  __syncthreads();
  return r;
}

// v[rows[q]] += sign * sum_c Mc[q + ns c] coef[c] (lrc_axpy_rows_kernel) and, in the same pass over the support rows,
// w[rows[q]] = save[q]: the right-hand side entries under the noise term go back behind the sweep (lrc_scatter_rows_kernel)
__global__ __launch_bounds__(256) void lrc_axpy_restore_rows_kernel(int64_t ns, int k, const double *__restrict__ Mc, const int64_t *__restrict__ rows, const double *__restrict__ coef, double sign, double *__restrict__ v, const double *__restrict__ save, double *__restrict__ w)
{
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= ns) return;
  double s = 0.0;
  for (int c = 0; c < k; ++c) s = fma(Mc[q + ns * c], coef[c], s);
  const int64_t r = rows[q];
  v[r]            = v[r] + sign * s;
  w[r]            = save[q];
}

// fill_normal_rows_kernel<true> (draw and scale by sqrt(S)) + lrc_axpy_rows_kernel(save) in one launch: EVERY block draws the k normals of
// the row stream (seed, sweep) itself -- counter-based: the same numbers in every block -- scales them by sqrt(S) and adds
// B eta to the support rows of b, keeping the old entries.  Same draws (pair c >> 1, branch c & 1), same roundings.
__global__ __launch_bounds__(256) void lrc_rhs_rows_kernel(int64_t ns, int k, const double *__restrict__ Mc, const int64_t *__restrict__ rows, const double *__restrict__ sqrtS, uint32_t key0, uint32_t key1, uint64_t sweep, double *__restrict__ b, double *__restrict__ save)
{
  __shared__ pmg::LogTabEntry s_logtab[PMG_LOGTAB_SIZE];
  __shared__ double           s_eta[64];
  pmg::load_log_table(s_logtab);
  __syncthreads();
  if (2 * (int)threadIdx.x < k) {
    const uint32_t p = threadIdx.x;
    double         z0, z1;
    pmg::normal_pair(p, 0u, (uint32_t)sweep, (uint32_t)(sweep >> 32), key0, key1, s_logtab, z0, z1);
    s_eta[2 * p] = z0 * sqrtS[2 * p];
    if (2 * (int)p + 1 < k) s_eta[2 * p + 1] = z1 * sqrtS[2 * p + 1];
  }
  __syncthreads();
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= ns) return;
  double s = 0.0;
  for (int c = 0; c < k; ++c) s = fma(Mc[q + ns * c], s_eta[c], s);
  const int64_t r = rows[q];
  const double  o = b[r];
  save[q]         = o;
  b[r]            = o + 1.0 * s;
}

// Small supports (one block of 256 * PMG_LRC_RPT rows): B^T y AND the update that consumes it in ONE launch of one workgroup --
//   wk = scale o (M1^T y[rows1]);  v[rows2] += sign * M2 wk;  optionally w[rows2] = save[q]
// (rows1 / M1 = the support of B on the level whose vector is read, rows2 / M2 = the block that is applied: Bb of the same
// level for the post-correction, B of the next coarser level for the restricted residual term).  Same sums in the same
// order as the separate kernels.
__global__ __launch_bounds__(256) void lrc_btx_axpy_small_kernel(int64_t ns1, int k, const double *__restrict__ M1, const int64_t *__restrict__ rows1, const double *y, const double *__restrict__ scale, double *__restrict__ wk, int64_t ns2, const double *__restrict__ M2, const int64_t *__restrict__ rows2, double sign, double *v, const double *__restrict__ save, double *w)
{
  __shared__ double red[64][4];
  __shared__ double s_wk[64];
  const int64_t     q0 = threadIdx.x;
  double            yv[PMG_LRC_RPT];
#pragma unroll
  for (int i = 0; i < PMG_LRC_RPT; ++i) {
    const int64_t q = q0 + 256 * i;
    yv[i]           = q < ns1 ? y[rows1[q]] : 0.0;
  }
  // k <= 64: one round (the host checks)
  const double r = lrc_block_colsum(ns1, k, 0, M1, yv, q0, red);
  if ((int)threadIdx.x < k) {
    // lrc_reduce_kernel with one block: lane 0 holds the block's sum, the others add zeros
    const double t = scale ? scale[threadIdx.x] * r : r;
    s_wk[threadIdx.x] = t;
    wk[threadIdx.x]   = t;
  }
  __syncthreads();
  for (int64_t q = threadIdx.x; q < ns2; q += 256) {
    double s = 0.0;
    for (int c = 0; c < k; ++c) s = fma(M2[q + ns2 * c], s_wk[c], s);
    const int64_t rr = rows2[q];
    v[rr]            = v[rr] + sign * s;
    if (save) w[rr] = save[q];
  }
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? 0 : 1; }

} // namespace

extern "C" int pmgk_lrc_axpy_restore_rows(int64_t ns, int k, const double *Mc, const int64_t *rows, const double *coef, double sign, double *v, const double *save, double *w, void *stream)
{
  if (ns <= 0) return 0;
  hipLaunchKernelGGL(lrc_axpy_restore_rows_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ns, k, Mc, rows, coef, sign, v, save, w);
  return launch_status();
}

extern "C" int pmgk_lrc_rhs_rows(int64_t ns, int k, const double *Mc, const int64_t *rows, const double *sqrtS, uint64_t seed, uint64_t sweep, double *b, double *save, void *stream)
{
  if (ns <= 0 || k <= 0 || k > 64) return k > 64;
  hipLaunchKernelGGL(lrc_rhs_rows_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ns, k, Mc, rows, sqrtS, (uint32_t)seed, (uint32_t)(seed >> 32), sweep, b, save);
  return launch_status();
}

/* one workgroup: ns1 <= pmgk_lrc_rows_per_block() (one block of lrc_btx_rows_partial_kernel), k <= 64; ns2 is looped over */
extern "C" int pmgk_lrc_btx_axpy_small(int64_t ns1, int k, const double *M1, const int64_t *rows1, const double *y, const double *scale, double *wk, int64_t ns2, const double *M2, const int64_t *rows2, double sign, double *v, const double *save, double *w, void *stream)
{
  if (ns1 <= 0 || ns1 > 256 * PMG_LRC_RPT || k <= 0 || k > 64) return 1;
  hipLaunchKernelGGL(lrc_btx_axpy_small_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ns1, k, M1, rows1, y, scale, wk, ns2, M2, rows2, sign, v, save, w);
  return launch_status();
}

extern "C" int pmgk_lrc_nblocks(int64_t n) { return (int)((n + 4095) / 4096); } /* dense form */
extern "C" int pmgk_lrc_rows_per_block(void) { return 256 * PMG_LRC_RPT; }
extern "C" int pmgk_lrc_rows_nblocks(int64_t ns) { return (int)((ns + 256 * PMG_LRC_RPT - 1) / (256 * PMG_LRC_RPT)); } /* row-compact form */

extern "C" int pmgk_lrc_btx(int64_t n, int k, const double *M, int64_t ld, const double *y, double *partial, const double *scale, double *out, void *stream)
{
  if (n <= 0 || k <= 0) return 0;
  const int nb = pmgk_lrc_nblocks(n);
  hipLaunchKernelGGL(lrc_btx_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, n, k, M, ld, y, partial);
  hipLaunchKernelGGL(lrc_reduce_kernel, dim3(k), dim3(64), 0, (hipStream_t)stream, nb, k, partial, scale, out);
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

extern "C" int pmgk_lrc_mark_rows(int64_t n, int k, const double *A0, const double *A1, const double *A2, int64_t ld, unsigned char *mask, void *stream)
{
  if (n <= 0) return 0;
  hipLaunchKernelGGL(lrc_mark_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, k, A0, A1, A2, ld, mask);
  return launch_status();
}

extern "C" int pmgk_lrc_gather_rows(int64_t ns, int k, const double *M, int64_t ld, const int64_t *rows, double *Mc, void *stream)
{
  if (ns <= 0) return 0;
  hipLaunchKernelGGL(lrc_gather_rows_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ns, k, M, ld, rows, Mc);
  return launch_status();
}

extern "C" int pmgk_lrc_btx_rows(int64_t ns, int k, const double *Mc, const int64_t *rows, const double *y, double *partial, const double *scale, double *out, const double *save, double *w, void *stream)
{
  if (ns <= 0 || k <= 0) return 0;
  const int nb = pmgk_lrc_rows_nblocks(ns);
  if (save) hipLaunchKernelGGL((lrc_btx_rows_partial_kernel<true>), dim3(nb), dim3(256), 0, (hipStream_t)stream, ns, k, Mc, rows, y, partial, save, w);
  else hipLaunchKernelGGL((lrc_btx_rows_partial_kernel<false>), dim3(nb), dim3(256), 0, (hipStream_t)stream, ns, k, Mc, rows, y, partial, save, w);
  if (out) hipLaunchKernelGGL(lrc_reduce_kernel, dim3(k), dim3(64), 0, (hipStream_t)stream, nb, k, partial, scale, out); /* else: pmgk_lrc_reduce_axpy_rows adds them */
  return launch_status();
}

/* v[rows] += sign * Mc (scale o sum of the nb partial sums pmgk_lrc_btx_rows(..., out = NULL) left); 1 <= k <= 64 */
extern "C" int pmgk_lrc_reduce_axpy_rows(int64_t ns, int k, const double *Mc, const int64_t *rows, int nb, const double *partial, const double *scale, double sign, double *v, double *save, void *stream)
{
  if (ns <= 0) return 0;
  if (k < 1 || k > 64 || nb < 1) return 1;
  hipLaunchKernelGGL(lrc_reduce_axpy_rows_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ns, k, Mc, rows, nb, partial, scale, sign, v, save);
  return launch_status();
}

/* pmgk_lrc_reduce_axpy_rows(Mb, partial_in, no scale) and, on the updated v, the partial sums pmgk_lrc_btx_rows(Mc, ..., out = NULL)
   would leave in partial_out (nb = pmgk_lrc_rows_nblocks(ns) blocks each); 1 <= k <= 8 */
extern "C" int pmgk_lrc_reduce_axpy_btx_rows(int64_t ns, int k, const double *Mb, const int64_t *rows, const double *partial_in, double sign, double *v, const double *Mc, double *partial_out, void *stream)
{
  if (ns <= 0) return 0;
  if (k < 1 || k > 8) return 1;
  const int nb = pmgk_lrc_rows_nblocks(ns);
  hipLaunchKernelGGL(lrc_reduce_axpy_btx_rows_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, ns, k, Mb, rows, nb, partial_in, sign, v, Mc, partial_out);
  return launch_status();
}

extern "C" int pmgk_lrc_axpy_rows(int64_t ns, int k, const double *Mc, const int64_t *rows, const double *coef, double sign, double *v, double *save, void *stream)
{
  if (ns <= 0) return 0;
  hipLaunchKernelGGL(lrc_axpy_rows_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ns, k, Mc, rows, coef, sign, v, save);
  return launch_status();
}

extern "C" int pmgk_lrc_scatter_rows(int64_t ns, const int64_t *rows, const double *save, double *v, void *stream)
{
  if (ns <= 0) return 0;
  hipLaunchKernelGGL(lrc_scatter_rows_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ns, rows, save, v);
  return launch_status();
}
