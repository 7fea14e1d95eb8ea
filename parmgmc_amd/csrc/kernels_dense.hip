// Coarse-grid exact sampler kernels (gfx950).
//
// Replaces PCCHOLSAMPLER's dense path (reference src/pc_chols.c): LAPACKpotrf_("L") at set-up (:174-194) and the
// two BLAS trsv calls per sample, y = L^-T (L^-1 b + xi) (:220-260, :284-288).
//
// Set-up (O(N^3) flops, the one place of the sampler where the matrix cores pay): blocked right-looking
// Cholesky with 32x32 tiles -- diagonal tile factored (and inverted) by one workgroup in LDS, panel = tile *
// inverse-diagonal^T, trailing update C_ik -= L_ij L_kj^T on v_mfma_f64_16x16x4_f64 (one wavefront per 32x32
// tile, 4 accumulators x 8 k-steps) -- followed by the blocked inverse W = L^-1, computed by recursive doubling over pairs of
// inverted diagonal blocks with the same MFMA tile product.
// Per sample: a triangular solve is a chain of N dependent steps, so the sample uses W: two triangular
// matrix-vector products, one wavefront per row, reading W (stored twice: row-major lower and row-major upper
// = W^T) exactly once -- HBM/L2 bound, 2 * N(N+1)/2 * 8 bytes.
//
// f64 MFMA operand maps (cdna_hip_programming.md section 3): A[row = lane&15][k = lane>>4], B[k = lane>>4][col =
// lane&15], C/D col = lane&15, row = (lane>>4) + 4*reg.
#include <hip/hip_runtime.h>
#include "pmg_kernels.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int NB = 32;

// acc[2][2] (16x16 sub-tiles of a 32x32 tile) += A(32 x K) * B(K x 32).
// A(r,k) = a[r + lda*k].  B(k,c) = bt ? b[c + ldb*k] : b[k + ldb*c].
__device__ __forceinline__ void mfma_tile32(v4d acc[2][2], const double *__restrict__ a, int64_t lda, const double *__restrict__ b, int64_t ldb, bool bt, int K)
{
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  for (int k0 = 0; k0 < K; k0 += 4) {
    const int    k  = k0 + lk;
    const double a0 = a[lr + lda * k], a1 = a[16 + lr + lda * k];
    const double b0 = bt ? b[lr + ldb * k] : b[k + ldb * lr];
    const double b1 = bt ? b[16 + lr + ldb * k] : b[k + ldb * (16 + lr)];
    acc[0][0]       = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1]       = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0]       = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1]       = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
  }
}

// element (row, col) of the 32x32 tile held by this lane in acc[ti][tj][reg]
__device__ __forceinline__ int tile_row(int ti, int reg) { return 16 * ti + ((threadIdx.x & 63) >> 4) + 4 * reg; }
__device__ __forceinline__ int tile_col(int tj) { return 16 * tj + (threadIdx.x & 15); }

// Diagonal tile j: factor in LDS, write L_jj (upper part zeroed) and its inverse Dinv[j] (32x32, column-major).
// info: 0 or the 1-based order of the first non-positive pivot.
__global__ __launch_bounds__(1024) void potrf_diag_kernel(double *__restrict__ A, int64_t ld, int j, double *__restrict__ Dinv, int *__restrict__ info)
{
  __shared__ double S[NB][NB + 1], W[NB][NB + 1];
  const int r = threadIdx.x, c = threadIdx.y;
  double   *T = A + (int64_t)j * NB * (ld + 1);
  S[r][c]     = T[r + ld * c];
  __syncthreads();
  for (int k = 0; k < NB; ++k) {
    if (r == k && c == k) {
      const double d = S[k][k];
      if (!(d > 0.0)) {
        atomicCAS(info, 0, j * NB + k + 1);
        S[k][k] = 1.0;
      } else {
        S[k][k] = sqrt(d);
      }
    }
    __syncthreads();
    if (c == k && r > k) S[r][k] = S[r][k] / S[k][k];
    __syncthreads();
    if (c > k && r >= c) S[r][c] = S[r][c] - S[r][k] * S[c][k];
    __syncthreads();
  }
  // inverse of the lower-triangular tile by forward substitution, one thread per column
  if (c == 0) {
    const int col = r;
    for (int i = 0; i < NB; ++i) {
      double s = (i == col) ? 1.0 : 0.0;
      for (int k = col; k < i; ++k) s -= S[i][k] * W[k][col];
      W[i][col] = i >= col ? s / S[i][i] : 0.0;
    }
  }
  __syncthreads();
  T[r + ld * c]                             = r >= c ? S[r][c] : 0.0;
  Dinv[(int64_t)j * NB * NB + r + NB * c] = W[r][c];
}

// Panel below diagonal tile j: L_ij = A_ij * Dinv_j^T for every tile row i > j (one workgroup per tile)
__global__ __launch_bounds__(1024) void potrf_panel_kernel(double *__restrict__ A, int64_t ld, int j, const double *__restrict__ Dinv)
{
  __shared__ double At[NB][NB + 1], Dt[NB][NB + 1];
  const int r = threadIdx.x, c = threadIdx.y;
  const int i = j + 1 + blockIdx.x;
  double   *T = A + (int64_t)i * NB + (int64_t)j * NB * ld;
  At[r][c]    = T[r + ld * c];
  Dt[r][c]    = Dinv[(int64_t)j * NB * NB + r + NB * c];
  __syncthreads();
  double s = 0.0;
  for (int k = 0; k <= c; ++k) s = fma(At[r][k], Dt[c][k], s); // (Dinv^T)(k,c) = Dinv(c,k), lower triangular
  T[r + ld * c] = s;
}

// Trailing update after panel j: C_ik -= L_ij L_kj^T for all tiles i >= k > j; one wavefront per tile.
__global__ __launch_bounds__(256) void potrf_update_kernel(double *__restrict__ A, int64_t ld, int j, int nt)
{
  const int m    = nt - j - 1; // trailing tile rows
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= m * (m + 1) / 2) return;
  // tile -> (ii >= kk) in the lower triangle, row-major enumeration
  int ii = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
  while ((ii + 1) * (ii + 2) / 2 <= tile) ++ii;
  while (ii * (ii + 1) / 2 > tile) --ii;
  const int kk = tile - ii * (ii + 1) / 2;
  const int i = j + 1 + ii, k = j + 1 + kk;
  v4d       acc[2][2] = {};
  mfma_tile32(acc, A + (int64_t)i * NB + (int64_t)j * NB * ld, ld, A + (int64_t)k * NB + (int64_t)j * NB * ld, ld, true, NB);
  double *C = A + (int64_t)i * NB + (int64_t)k * NB * ld;
  for (int ti = 0; ti < 2; ++ti)
    for (int tj = 0; tj < 2; ++tj)
      for (int reg = 0; reg < 4; ++reg) {
        const int64_t o = tile_row(ti, reg) + ld * tile_col(tj);
        C[o]            = C[o] - acc[ti][tj][reg];
      }
}

// Blocked inverse W = L^-1 by recursive doubling.  With the diagonal blocks A (tiles [a0, a0+m)) and B (tiles [a0+m, b1))
// of a pair already inverted, the block below the diagonal is  W_BA = -W_BB (L_BA W_AA):  two tile products per level,
//   phase 0:  T_{ib,ja} =  sum_{k in A, k >= ja} L_{ib,k} W_{k,ja}        (W_AA is lower triangular in tiles)
//   phase 1:  W_{ib,ja} = -sum_{k in B, k <= ib} W_{ib,k} T_{k,ja}
// one wavefront per 32x32 output tile on v_mfma_f64_16x16x4_f64, all pairs of a level in one launch: 2 log2(nt) launches
// with thousands of independent tiles each, instead of nt - 1 anti-diagonal launches in which a handful of wavefronts
// each ran a chain of up to nt tile products (round 1: 153 launches of ~300 us for 17^3 = 4913 rows, the largest entry of
// the bench's kernel table although it is set-up).  grid = (m, m, pairs), tiles outside a ragged last block return.
__global__ __launch_bounds__(64) void inv_pair_kernel(const double *__restrict__ L, double *__restrict__ W, double *__restrict__ T, int64_t ld, int m, int nt, int phase)
{
  const int a0 = 2 * m * (int)blockIdx.z, b0 = a0 + m, b1 = min(b0 + m, nt);
  const int ja = a0 + (int)blockIdx.x, ib = b0 + (int)blockIdx.y;
  if (ib >= b1) return;
  v4d acc[2][2] = {};
  if (phase == 0) mfma_tile32(acc, L + (int64_t)ib * NB + (int64_t)ja * NB * ld, ld, W + (int64_t)ja * NB + (int64_t)ja * NB * ld, ld, false, (b0 - ja) * NB);
  else mfma_tile32(acc, W + (int64_t)ib * NB + (int64_t)b0 * NB * ld, ld, T + (int64_t)b0 * NB + (int64_t)ja * NB * ld, ld, false, (ib - b0 + 1) * NB);
  double *O = (phase == 0 ? T : W) + (int64_t)ib * NB + (int64_t)ja * NB * ld;
  for (int ti = 0; ti < 2; ++ti)
    for (int tj = 0; tj < 2; ++tj)
      for (int reg = 0; reg < 4; ++reg) O[tile_row(ti, reg) + ld * tile_col(tj)] = phase == 0 ? acc[ti][tj][reg] : -acc[ti][tj][reg];
}

__global__ void inv_diag_kernel(double *__restrict__ W, int64_t ld, const double *__restrict__ Dinv)
{
  const int j = blockIdx.x;
  for (int e = threadIdx.x; e < NB * NB; e += blockDim.x) {
    const int r = e & (NB - 1), c = e >> 5;
    W[(int64_t)j * NB * (ld + 1) + r + ld * c] = Dinv[(int64_t)j * NB * NB + e];
  }
}

// out(row-major n x n) = in(column-major, leading dimension ld)^T or plain copy: the two layouts the sampler reads
__global__ void pack_rowmajor_kernel(int32_t n, const double *__restrict__ in, int64_t ld, int transpose, double *__restrict__ out)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c >= n) return;
  // out[r*n + c] = M(r,c) with M = in (transpose = 0) or in^T (transpose = 1)
  out[(int64_t)r * n + c] = transpose ? in[c + ld * r] : in[r + ld * c];
}

// out[i] = sum_{k in [lo_i, hi_i)} M[i*n + k] * x[k] (+ add[i]); lower: [0, i+1), upper: [i, n).
// One wavefront per row; a lane takes two adjacent entries at a time (16-byte loads, rows of odd length are only 8-byte
// aligned) at two places 128 entries apart, i.e. four independent accumulators per lane and 4 KB of the row in flight
// per wavefront-iteration: the product is a pure HBM stream (the triangle of W is read once per sample), and a single
// dependent fma chain per lane left it at 2.3 TB/s.  The order of the sum is fixed (a function of n and i only), so the
// result is reproducible and identical on every rank of a replicated coarse level.
typedef double d2g __attribute__((ext_vector_type(2), aligned(8)));
template <bool UPPER>
__global__ __launch_bounds__(256) void tri_gemv_kernel(int32_t n, const double *__restrict__ M, const double *__restrict__ x, const double *__restrict__ add, double *__restrict__ out)
{
  const int lane = threadIdx.x & 63;
  const int i    = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= n) return;
  const int     k0  = UPPER ? i : 0, k1 = UPPER ? n : i + 1;
  const double *row = M + (int64_t)i * n;
  double        s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int kb = k0; kb < k1; kb += 256) {
    const int ka = kb + 2 * lane, kc = ka + 128;
    if (kc + 1 < k1) { // both pairs inside the row part (wave-divergent only in the last iteration)
      const d2g ra = *reinterpret_cast<const d2g *>(row + ka), xa = *reinterpret_cast<const d2g *>(x + ka);
      const d2g rc = *reinterpret_cast<const d2g *>(row + kc), xc = *reinterpret_cast<const d2g *>(x + kc);
      s0 = fma(ra.x, xa.x, s0);
      s1 = fma(ra.y, xa.y, s1);
      s2 = fma(rc.x, xc.x, s2);
      s3 = fma(rc.y, xc.y, s3);
    } else {
      if (ka < k1) s0 = fma(row[ka], x[ka], s0);
      if (ka + 1 < k1) s1 = fma(row[ka + 1], x[ka + 1], s1);
      if (kc < k1) s2 = fma(row[kc], x[kc], s2);
      if (kc + 1 < k1) s3 = fma(row[kc + 1], x[kc + 1], s3);
    }
  }
  double s = (s0 + s1) + (s2 + s3);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) out[i] = add ? s + add[i] : s;
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? 0 : 1; }

} // namespace

// In-place lower Cholesky of the column-major npad x npad matrix A (npad a multiple of 32, padding = identity),
// then W = L^-1 (column-major, same shape; zero on entry).  T: scratch of the same shape, Dinv: npad/32 tiles of 32x32
// scratch.  info: device int, 0 on entry.
extern "C" int pmgk_potrf_inverse(int32_t npad, double *A, double *W, double *T, double *Dinv, int *info, void *stream)
{
  hipStream_t s  = (hipStream_t)stream;
  const int   nt = npad / NB;
  for (int j = 0; j < nt; ++j) {
    hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(NB, NB), 0, s, A, (int64_t)npad, j, Dinv, info);
    const int m = nt - j - 1;
    if (m > 0) {
      hipLaunchKernelGGL(potrf_panel_kernel, dim3(m), dim3(NB, NB), 0, s, A, (int64_t)npad, j, Dinv);
      const int tiles = m * (m + 1) / 2;
      hipLaunchKernelGGL(potrf_update_kernel, dim3((tiles + 3) / 4), dim3(256), 0, s, A, (int64_t)npad, j, nt);
    }
  }
  hipLaunchKernelGGL(inv_diag_kernel, dim3(nt), dim3(256), 0, s, W, (int64_t)npad, Dinv);
  for (int m = 1; m < nt; m *= 2) { // blocks of m tiles are inverted: pair them
    const int pairs = (nt + 2 * m - 1) / (2 * m);
    for (int phase = 0; phase < 2; ++phase) hipLaunchKernelGGL(inv_pair_kernel, dim3(m, m, pairs), dim3(64), 0, s, A, W, T, (int64_t)npad, m, nt, phase);
  }
  return launch_status();
}

extern "C" int pmgk_pack_rowmajor(int32_t n, const double *in, int64_t ld, int transpose, double *out, void *stream)
{
  if (n <= 0) return 0;
  hipLaunchKernelGGL(pack_rowmajor_kernel, dim3((n + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, n, in, ld, transpose, out);
  return launch_status();
}

extern "C" int pmgk_tri_gemv(int32_t n, int upper, const double *M, const double *x, const double *add, double *out, void *stream)
{
  if (n <= 0) return 0;
  const dim3 block(256), grid((n + 3) / 4);
  if (upper) hipLaunchKernelGGL((tri_gemv_kernel<true>), grid, block, 0, (hipStream_t)stream, n, M, x, add, out);
  else hipLaunchKernelGGL((tri_gemv_kernel<false>), grid, block, 0, (hipStream_t)stream, n, M, x, add, out);
  return launch_status();
}
