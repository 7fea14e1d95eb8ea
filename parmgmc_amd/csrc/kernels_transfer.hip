// Matrix-free Q1 transfers between a red-black grid level (colour-partitioned cvec) and the next coarser level
// (any layout, given by a position map) -- gfx950.
//
// Replace the two sparse products PCMG issues per level and cycle, MatRestrict (b_c = P^T r) and
// MatInterpolateAdd (x += P e_c) (PETSc, entered from reference src/pc_gamgmc.c:246,255), for DMDA's Q1
// interpolation: fine point 2I coincides with coarse I (weight 1), fine 2I+1 lies midway between coarse I and I+1
// (1/2 each), tensor product over the refined directions.  No matrix is read: 8 B per fine unknown + 8 B per
// coarse unknown instead of 12 B per stored entry (3.4 entries per fine row for P, 27 per coarse row for P^T).
// Sums run over the same entries in the same (ascending fine / coarse index) order as the CSR product of the
// stored transposed / plain interpolation, so results are bit-identical to the assembled operators.
#include <hip/hip_runtime.h>
#include "pmg_kernels.h"

namespace {

__device__ __forceinline__ int64_t cvec_pos(const pmgk_grid_layout &L, int i, int j, int k)
{
  const int c = (i + j + k + L.kz0) & 1;
  return (int64_t)c * L.cs + (int64_t)(k + 1) * L.sp + (int64_t)j * L.sx + (i >> 1);
}

__device__ __forceinline__ int64_t coarse_pos(const pmgk_st27_dims &C, const int32_t *__restrict__ cpos, int I, int J, int K)
{
  return cpos ? (int64_t)cpos[I + C.nx * (J + C.ny * K)] : I + (int64_t)C.nx * (J + (int64_t)C.ny * (K - C.kz0 + 1));
}

// b_c(I) = sum over the <= 27 fine points 2I+d, d in {-1,0,1}^3 (only refined directions), of w(d) r(2I+d); K global
__global__ __launch_bounds__(256) void q1_restrict_kernel(pmgk_grid_layout L, pmgk_st27_dims C, int rx, int ry, int rz, const int32_t *__restrict__ cpos, const double *__restrict__ r, double *__restrict__ bc)
{
  const int flat = blockIdx.x * blockDim.x + threadIdx.x, J = flat / C.nx, I = flat - J * C.nx, K = C.kz0 + blockIdx.z; // lines packed into the wavefronts
  if (J >= C.ny) return;
  const int fi = rx ? 2 * I : I, fj = ry ? 2 * J : J, fk = rz ? 2 * K : K;
  double    s  = 0.0;
  for (int dz = rz ? -1 : 0; dz <= (rz ? 1 : 0); ++dz) {
    const int kg = fk + dz;
    if (kg < 0 || kg >= L.nzg) continue;
    for (int dy = ry ? -1 : 0; dy <= (ry ? 1 : 0); ++dy) {
      const int j = fj + dy;
      if (j < 0 || j >= L.ny) continue;
      for (int dx = rx ? -1 : 0; dx <= (rx ? 1 : 0); ++dx) {
        const int i = fi + dx;
        if (i < 0 || i >= L.nx) continue;
        const double w = (dx ? 0.5 : 1.0) * (dy ? 0.5 : 1.0) * (dz ? 0.5 : 1.0);
        s              = s + w * r[cvec_pos(L, i, j, kg - L.kz0)];
      }
    }
  }
  bc[coarse_pos(C, cpos, I, J, K)] = s;
}

// x += P e_c on the colour-partitioned fine vector: thread = two consecutive same-colour points (one 16-byte
// read-modify-write), blocks of 64 lanes x 4 lines like the sweep; up to 8 coarse reads per point (L2-resident),
// summed in ascending coarse index.
__device__ __forceinline__ double q1_interp_point(int i, int j, int k, const pmgk_st27_dims &C, int rx, int ry, int rz, const int32_t *__restrict__ cpos, const double *__restrict__ ec)
{ // k: global fine plane
  const int    I0 = rx ? i >> 1 : i, J0 = ry ? j >> 1 : j, K0 = rz ? k >> 1 : k;
  const int    mx = (rx && (i & 1)) ? 2 : 1, my = (ry && (j & 1)) ? 2 : 1, mz = (rz && (k & 1)) ? 2 : 1;
  const double w  = (mx == 2 ? 0.5 : 1.0) * (my == 2 ? 0.5 : 1.0) * (mz == 2 ? 0.5 : 1.0);
  double       s  = 0.0;
  for (int c = 0; c < mz; ++c)
    for (int bq = 0; bq < my; ++bq)
      for (int a = 0; a < mx; ++a) s = s + w * ec[coarse_pos(C, cpos, I0 + a, J0 + bq, K0 + c)];
  return s;
}

typedef double d2t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void q1_prolong_add_kernel(pmgk_grid_layout L, pmgk_st27_dims C, int rx, int ry, int rz, int kbegin, int tplE, const int32_t *__restrict__ cpos, const double *__restrict__ ec, double *__restrict__ x)
{
  // tplE threads per line own a point; the lines of a plane are packed into the wavefronts without gaps
  const int flat = blockIdx.x * 256 + threadIdx.y * 64 + threadIdx.x, j = flat / tplE, t = flat - j * tplE;
  const int k = kbegin + (int)(blockIdx.z >> 1), c = blockIdx.z & 1; // k: local plane, -1 / nz = ghosts
  if (j >= L.ny || 2 * t >= L.sx) return;
  const int p  = (c + j + k + L.kz0) & 1;
  const int i0 = 4 * t + p, i1 = i0 + 2;
  if (i0 >= L.nx) return;
  double       *px = x + (int64_t)c * L.cs + (int64_t)(k + 1) * L.sp + (int64_t)j * L.sx + 2 * t;
  d2t           v  = *reinterpret_cast<d2t *>(px);
  v.x              = v.x + q1_interp_point(i0, j, k + L.kz0, C, rx, ry, rz, cpos, ec);
  if (i1 < L.nx) v.y = v.y + q1_interp_point(i1, j, k + L.kz0, C, rx, ry, rz, cpos, ec);
  *reinterpret_cast<d2t *>(px) = v;
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? 0 : 1; }

} // namespace

extern "C" int pmgk_q1_restrict(const pmgk_grid_layout *L, const pmgk_st27_dims *C, const int32_t *cpos, const double *r_cvec, double *bc, void *stream)
{
  if (C->nz <= 0) return 0;
  const int  rx = C->nx != L->nx, ry = C->ny != L->ny, rz = C->nzg != L->nzg;
  const dim3 block(256), grid((unsigned)(((int64_t)C->nx * C->ny + 255) / 256), 1, C->nz);
  hipLaunchKernelGGL(q1_restrict_kernel, grid, block, 0, (hipStream_t)stream, *L, *C, rx, ry, rz, cpos, r_cvec, bc);
  return launch_status();
}

extern "C" int pmgk_q1_prolong_add(const pmgk_grid_layout *L, const pmgk_st27_dims *C, const int32_t *cpos, int kbegin, int kcount, const double *ec, double *x_cvec, void *stream)
{
  if (kcount <= 0) return 0;
  const int  rx = C->nx != L->nx, ry = C->ny != L->ny, rz = C->nzg != L->nzg;
  const int  tplE = ((L->nx + 1) / 2 + 1) / 2;
  const dim3 block(64, 4), grid((unsigned)(((int64_t)L->ny * tplE + 255) / 256), 1, 2 * kcount);
  hipLaunchKernelGGL(q1_prolong_add_kernel, grid, block, 0, (hipStream_t)stream, *L, *C, rx, ry, rz, kbegin, tplE, cpos, ec, x_cvec);
  return launch_status();
}
