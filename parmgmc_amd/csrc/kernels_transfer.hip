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
#include <stdlib.h>
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

__device__ __forceinline__ const double *at_bytes(const double *base, uint32_t byte_off) { return reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + byte_off); }

// the common case (all three directions refined, coarse level in plane-padded natural storage, fine cvec < 4 GiB):
// no branch per fine point -- a point outside the domain is read at the centre with weight 0 (s + 0*r = s, so the sum
// is the same bits), 32-bit offsets from the scalar vector base, all 27 loads in flight at once
__global__ __launch_bounds__(256) void q1_restrict_full_kernel(pmgk_grid_layout L, pmgk_st27_dims C, const double *__restrict__ r, double *__restrict__ bc)
{
  const int flat = blockIdx.x * blockDim.x + threadIdx.x, J = flat / C.nx, I = flat - J * C.nx, K = C.kz0 + blockIdx.z;
  if (J >= C.ny) return;
  const int32_t sx = (int32_t)L.sx, sp = (int32_t)L.sp, cs = (int32_t)L.cs;
  const int     fi = 2 * I, fj = 2 * J, fk = 2 * K;
  double        s  = 0.0;
#pragma unroll
  for (int dz = -1; dz <= 1; ++dz) {
    const int     kg   = fk + dz;
    const bool    okz  = (unsigned)kg < (unsigned)L.nzg;
    const int     kk   = okz ? kg : fk;
    const int32_t zoff = (kk - L.kz0 + 1) * sp;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int     j    = fj + dy;
      const bool    oky  = (unsigned)j < (unsigned)L.ny;
      const int     jj   = oky ? j : fj;
      const int32_t yoff = zoff + jj * sx;
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int     i   = fi + dx;
        const bool    okx = (unsigned)i < (unsigned)L.nx;
        const int     ii  = okx ? i : fi;
        const int32_t off = yoff + (ii >> 1) + (((ii + jj + kk) & 1) ? cs : 0);
        const double  w0  = (dx ? 0.5 : 1.0) * (dy ? 0.5 : 1.0) * (dz ? 0.5 : 1.0);
        const double  w   = (okx && oky && okz) ? w0 : 0.0;
        s                 = s + w * *at_bytes(r, 8u * (uint32_t)off);
      }
    }
  }
  bc[I + (int64_t)C.nx * (J + (int64_t)C.ny * (K - C.kz0 + 1))] = s;
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

__global__ __launch_bounds__(256) void q1_prolong_add_kernel(pmgk_grid_layout L, pmgk_st27_dims C, int rx, int ry, int rz, int kbegin, int tplE, int csel, const int32_t *__restrict__ cpos, const double *__restrict__ ec, double *__restrict__ x)
{
  // tplE threads per line own a point; the lines of a plane are packed into the wavefronts without gaps
  const int flat = blockIdx.x * 256 + threadIdx.y * 64 + threadIdx.x, j = flat / tplE, t = flat - j * tplE;
  const int k = kbegin + (int)(csel >= 0 ? blockIdx.z : blockIdx.z >> 1), c = csel >= 0 ? csel : (int)(blockIdx.z & 1); // k: local plane, -1 / nz = ghosts
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

// same common case for the prolongation: always 8 coarse reads per point, the ones a non-midpoint direction does not
// use carry weight 0 (and address the used one, so no extra traffic); the two points of a thread share their parities
__global__ __launch_bounds__(256) void q1_prolong_add_full_kernel(pmgk_grid_layout L, pmgk_st27_dims C, int kbegin, int tplE, int csel, const double *__restrict__ ec, double *__restrict__ x)
{
  const int flat = blockIdx.x * 256 + threadIdx.y * 64 + threadIdx.x, j = flat / tplE, t = flat - j * tplE;
  const int k = kbegin + (int)(csel >= 0 ? blockIdx.z : blockIdx.z >> 1), c = csel >= 0 ? csel : (int)(blockIdx.z & 1); // k: local plane, -1 / nz = ghosts
  if (j >= L.ny || 2 * t >= L.sx) return;
  const int kg = k + L.kz0, p = (c + j + kg) & 1;
  const int i0 = 4 * t + p, i1 = i0 + 2;
  if (i0 >= L.nx) return;
  double       *px = x + (int64_t)c * L.cs + (int64_t)(k + 1) * L.sp + (int64_t)j * L.sx + 2 * t;
  d2t           v  = *reinterpret_cast<d2t *>(px);
  const int32_t cnx = C.nx, cnxy = C.nx * C.ny;
  const int     oddx = p, oddy = j & 1, oddz = kg & 1;
  const int32_t base = ((kg >> 1) - C.kz0 + 1) * cnxy + (j >> 1) * cnx + (i0 >> 1);
  const int32_t second = i1 < L.nx ? 1 : 0; // the second point's coarse neighbours lie one to the right
  double        s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int cz = 0; cz < 2; ++cz) {
    const double wz = oddz ? 0.5 : (cz ? 0.0 : 1.0);
#pragma unroll
    for (int by = 0; by < 2; ++by) {
      const double wy = oddy ? 0.5 : (by ? 0.0 : 1.0);
#pragma unroll
      for (int ax = 0; ax < 2; ++ax) {
        const double  wx  = oddx ? 0.5 : (ax ? 0.0 : 1.0);
        const double  w   = wx * wy * wz;
        const int32_t off = base + (cz & oddz) * cnxy + (by & oddy) * cnx + (ax & oddx);
        s0                = s0 + w * *at_bytes(ec, 8u * (uint32_t)off);
        s1                = s1 + w * *at_bytes(ec, 8u * (uint32_t)(off + second));
      }
    }
  }
  v.x = v.x + s0;
  if (i1 < L.nx) v.y = v.y + s1;
  *reinterpret_cast<d2t *>(px) = v;
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? 0 : 1; }

// all directions refined, natural coarse storage, and both vectors addressable with 32-bit byte offsets
inline bool transfer_full_case(const pmgk_grid_layout *L, const pmgk_st27_dims *C, const int32_t *cpos)
{
  static const int off = getenv("PMG_TRANSFER_GENERIC") != nullptr;
  if (off || cpos) return false;
  if (C->nx == L->nx || C->ny == L->ny || C->nzg == L->nzg) return false;
  if (2 * (int64_t)L->cs * 8 >= ((int64_t)1 << 31)) return false;
  return (int64_t)C->nx * C->ny * (C->nz + 2) * 8 < ((int64_t)1 << 31);
}

} // namespace

extern "C" int pmgk_q1_restrict(const pmgk_grid_layout *L, const pmgk_st27_dims *C, const int32_t *cpos, const double *r_cvec, double *bc, void *stream)
{
  if (C->nz <= 0) return 0;
  const int  rx = C->nx != L->nx, ry = C->ny != L->ny, rz = C->nzg != L->nzg;
  const dim3 block(256), grid((unsigned)(((int64_t)C->nx * C->ny + 255) / 256), 1, C->nz);
  if (transfer_full_case(L, C, cpos)) {
    hipLaunchKernelGGL(q1_restrict_full_kernel, grid, block, 0, (hipStream_t)stream, *L, *C, r_cvec, bc);
    return launch_status();
  }
  hipLaunchKernelGGL(q1_restrict_kernel, grid, block, 0, (hipStream_t)stream, *L, *C, rx, ry, rz, cpos, r_cvec, bc);
  return launch_status();
}

// only_color >= 0: correct that colour only (the caller knows the other one is overwritten before it is read)
extern "C" int pmgk_q1_prolong_add(const pmgk_grid_layout *L, const pmgk_st27_dims *C, const int32_t *cpos, int kbegin, int kcount, int only_color, const double *ec, double *x_cvec, void *stream)
{
  if (kcount <= 0) return 0;
  const int  rx = C->nx != L->nx, ry = C->ny != L->ny, rz = C->nzg != L->nzg;
  const int  tplE = ((L->nx + 1) / 2 + 1) / 2;
  const dim3 block(64, 4), grid((unsigned)(((int64_t)L->ny * tplE + 255) / 256), 1, only_color >= 0 ? kcount : 2 * kcount);
  if (transfer_full_case(L, C, cpos)) {
    hipLaunchKernelGGL(q1_prolong_add_full_kernel, grid, block, 0, (hipStream_t)stream, *L, *C, kbegin, tplE, only_color, ec, x_cvec);
    return launch_status();
  }
  hipLaunchKernelGGL(q1_prolong_add_kernel, grid, block, 0, (hipStream_t)stream, *L, *C, rx, ry, rz, kbegin, tplE, only_color, cpos, ec, x_cvec);
  return launch_status();
}
