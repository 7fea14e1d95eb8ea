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

// ---- paired forms of the two common-case kernels above (round 2) ---------------------------------------------------------
// The per-point kernels are bound by the number of load INSTRUCTIONS (27 per coarse point, 16 per two fine points, many
// of them lanes striding over every other element): `SQ_WAIT_INST_ANY` 72-75 % of the wave cycles.  Here a thread takes
// TWO neighbouring points and fetches every line it needs as one 16-byte load; the value next to the pair comes from the
// neighbouring lane (DPP full-wave shift) where that lane sits on the same line, from one extra load where it does not.
// Same terms, same order as the kernels above, so the same bits.
typedef double d2a __attribute__((ext_vector_type(2), aligned(8)));

__device__ __forceinline__ double lane_prev(double v) // lane i <- lane i-1 (lane 0: 0)
{
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ double lane_next(double v) // lane i <- lane i+1 (lane 63: 0)
{
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, false));
}

// restriction: thread = coarse points I = 2q, 2q+1 of line J.  A fine line (jj, kk) holds x = 2I in the colour array
// c0 = (jj + kk) & 1 at m = I and x = 2I-1, 2I+1 in the other one at m = I-1, I: two 16-byte loads per fine line give
// x = 2I, 2I+2 and x = 2I+1, 2I+3; x = 2I-1 is the previous thread's x = 2I+3.
__global__ __launch_bounds__(256) void q1_restrict_pair_kernel(pmgk_grid_layout L, pmgk_st27_dims C, int tplC, const double *__restrict__ r, double *__restrict__ bc)
{
  const int lane = threadIdx.x & 63;
  const int flat = blockIdx.x * 256 + threadIdx.x, J = flat / tplC, q = flat - J * tplC, K = C.kz0 + blockIdx.z;
  const bool live = J < C.ny; // lanes behind the last line keep running (DPP sources), on clamped addresses
  const int  Jc = live ? J : C.ny - 1, I0 = 2 * q;
  const bool has1 = I0 + 1 < C.nx;
  const int32_t sx = (int32_t)L.sx, sp = (int32_t)L.sp, cs = (int32_t)L.cs;
  const int     fj = 2 * Jc, fk = 2 * K;
  const bool    own_left = lane == 0 || q == 0; // the lane before me is on another line (or there is none): load x = 2 I0 - 1 myself
  double        s0 = 0.0, s1 = 0.0;
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
      const int     c0   = (jj + kk) & 1;                       // colour of the even fine points of this line
      const double *ev   = r + (c0 ? cs : 0) + yoff + I0;       // x = 2 I0, 2 I0 + 2
      const double *od   = r + (c0 ? 0 : cs) + yoff + I0;       // x = 2 I0 + 1, 2 I0 + 3
      const d2a     E = *reinterpret_cast<const d2a *>(ev), O = *reinterpret_cast<const d2a *>(od); // pad slots are zero and inside the line
      double        Lf = lane_prev(O.y);                        // x = 2 I0 - 1
      if (own_left) Lf = I0 > 0 ? od[-1] : E.x;                // absent (x = -1): the centre, with weight 0 below
      const double wyz = ((dy ? 0.5 : 1.0) * (dz ? 0.5 : 1.0));
      const bool   ok  = oky && okz;
      // the per-point kernel forms w0 = (wx * wy) * wz with wx first: (0.5 * wy) * wz and (1.0 * wy) * wz -- the same
      // numbers as products of powers of two, so any association gives the same bits
      const double wl0 = (ok && I0 > 0) ? 0.5 * wyz : 0.0, wm = ok ? wyz : 0.0;
      const double wr0 = (ok && 2 * I0 + 1 < L.nx) ? 0.5 * wyz : 0.0;
      s0 = s0 + wl0 * Lf;
      s0 = s0 + wm * E.x;
      s0 = s0 + wr0 * O.x;
      const double wr1 = (ok && 2 * I0 + 3 < L.nx) ? 0.5 * wyz : 0.0;
      s1 = s1 + (ok ? 0.5 * wyz : 0.0) * O.x; // x = 2 I1 - 1 = 2 I0 + 1 exists whenever I1 does
      s1 = s1 + wm * E.y;
      s1 = s1 + wr1 * O.y;
    }
  }
  if (!live || I0 >= C.nx) return;
  double *o = bc + I0 + (int64_t)C.nx * (Jc + (int64_t)C.ny * (K - C.kz0 + 1));
  o[0]      = s0;
  if (has1) o[1] = s1;
}

// prolongation, single device: thread = the colour-c points i0, i0+2 of the FOUR fine lines (2J, 2J+1) x (2K, 2K+1), which
// share the coarse lines (J, J+1) x (K, K+1): four coarse loads for four read-modify-writes instead of four per line.
// (The per-line kernel fetched 1.42 GB for 0.68 GB of operands at 513^3, FETCH_SIZE: the coarse rows came back from
// beyond the L2 for almost every fine line that uses them.)  Same sums in the same order for every point.
__device__ __forceinline__ void prolong_line_pair(int oddx, int oddy, int oddz, bool act1, const double (&A)[2][2], const double (&B)[2][2], const double (&Cn)[2][2], double &s0, double &s1)
{
  s0 = 0.0;
  s1 = 0.0;
#pragma unroll
  for (int cz = 0; cz < 2; ++cz) {
    const double wz = oddz ? 0.5 : (cz ? 0.0 : 1.0);
#pragma unroll
    for (int by = 0; by < 2; ++by) {
      const double wy = oddy ? 0.5 : (by ? 0.0 : 1.0);
      const int    a = cz & oddz, bq = by & oddy;
#pragma unroll
      for (int ax = 0; ax < 2; ++ax) {
        const double wx = oddx ? 0.5 : (ax ? 0.0 : 1.0);
        const double w  = wx * wy * wz;
        const double v0 = (ax & oddx) ? B[a][bq] : A[a][bq];
        const double v1 = act1 ? ((ax & oddx) ? Cn[a][bq] : B[a][bq]) : v0;
        s0              = s0 + w * v0;
        s1              = s1 + w * v1;
      }
    }
  }
}

__global__ __launch_bounds__(256) void q1_prolong_add_quad_kernel(pmgk_grid_layout L, pmgk_st27_dims C, int tplE, int csel, int xcd_runs, int gbeg, int gend, const double *__restrict__ ec, double *__restrict__ x)
{
  // xcd_runs: gridDim.x is a multiple of 8 and XCD x (= blockIdx.x % 8 under round-robin dispatch; speed only) takes a contiguous run
  // of the line pairs of EVERY plane pair: the coarse rows two neighbouring line pairs or plane pairs share meet in one L2
  const int  lane = threadIdx.x & 63, npair = (L.ny + 1) / 2;
  const int  vb = xcd_runs ? ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const int  flat = vb * 256 + threadIdx.x, Jp = flat / tplE, t = flat - Jp * tplE;
  // global fine planes gbeg .. gend-1 (a z-slab: its planes and in-domain ghost planes); plane pair K = planes 2K, 2K+1
  const int  K = (gbeg >> 1) + (int)(csel >= 0 ? blockIdx.z : blockIdx.z >> 1), c = csel >= 0 ? csel : (int)(blockIdx.z & 1);
  const bool live = Jp < npair;
  const int  J = live ? Jp : npair - 1;
  const int32_t cnx = C.nx, cnxy = C.nx * C.ny;
  const int     I0 = min(2 * t, cnx - 1);
  const bool    own_right = lane == 63 || t == tplE - 1;
  const bool    line1 = 2 * J + 1 < L.ny;                      // the odd line of the quad exists
  const bool    plane0 = 2 * K >= gbeg && 2 * K < gend, plane1 = 2 * K + 1 >= gbeg && 2 * K + 1 < gend; // its planes are in the range
  const int32_t base = (K - C.kz0 + 1) * cnxy + J * cnx;
  // fine values first: one round trip for everything
  d2t     v[2][2];
  double *px[2][2];
#pragma unroll
  for (int dz = 0; dz < 2; ++dz)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const int k = min(max(2 * K + dz, gbeg), gend - 1) - L.kz0, j = min(2 * J + dy, L.ny - 1); // k: plane inside the slab (-1, nz: ghosts)
      px[dz][dy]  = x + (int64_t)c * L.cs + (int64_t)(k + 1) * L.sp + (int64_t)j * L.sx + 2 * min(t, tplE - 1);
      v[dz][dy]   = *reinterpret_cast<d2t *>(px[dz][dy]);
    }
  double A[2][2], B[2][2], Cn[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int bq = 0; bq < 2; ++bq) {
      const double *row = ec + base + (a && plane1 ? cnxy : 0) + (bq && line1 ? cnx : 0);
      if (I0 + 1 < cnx) {
        const d2a w = *reinterpret_cast<const d2a *>(row + I0);
        A[a][bq]    = w.x;
        B[a][bq]    = w.y;
      } else {
        A[a][bq] = B[a][bq] = row[I0];
      }
      Cn[a][bq] = lane_next(A[a][bq]);
    }
  if (own_right) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int bq = 0; bq < 2; ++bq) {
        const double *row = ec + base + (a && plane1 ? cnxy : 0) + (bq && line1 ? cnx : 0);
        Cn[a][bq]         = I0 + 2 < cnx ? row[I0 + 2] : B[a][bq];
      }
  }
  if (!live || 2 * t >= L.sx) return;
#pragma unroll
  for (int dz = 0; dz < 2; ++dz)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      if ((dz ? !plane1 : !plane0) || (dy && !line1)) continue;
      const int  p = (c + dy + dz) & 1; // (c + j + k) & 1 with j = 2J + dy, k = 2K + dz (global plane)
      const int  i0 = 4 * t + p;
      const bool act0 = i0 < L.nx, act1 = act0 && i0 + 2 < L.nx;
      if (!act0) continue;
      double s0, s1;
      prolong_line_pair(p, dy, dz, act1, A, B, Cn, s0, s1);
      d2t w = v[dz][dy];
      w.x   = w.x + s0;
      if (act1) w.y = w.y + s1;
      *reinterpret_cast<d2t *>(px[dz][dy]) = w;
    }
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
    const int tplC = (C->nx + 1) / 2;
    hipLaunchKernelGGL(q1_restrict_pair_kernel, dim3((unsigned)(((int64_t)C->ny * tplC + 255) / 256), 1, C->nz), block, 0, (hipStream_t)stream, *L, *C, tplC, r_cvec, bc);
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
  if (transfer_full_case(L, C, cpos) && C->ny == (L->ny + 1) / 2 && C->nzg == (L->nzg + 1) / 2) { // otherwise (semicoarsened, permuted or even extents): the generic kernel
    const int  gbeg = L->kz0 + kbegin, gend = gbeg + kcount; // global planes
    const int  npair = (L->ny + 1) / 2, nkp = ((gend - 1) >> 1) - (gbeg >> 1) + 1;
    const int64_t nb = ((int64_t)npair * tplE + 255) / 256;
    const int     runs = nb >= 64; // below that the padding to a multiple of 8 costs more than the shared rows bring (257^3: 35 vs 39 us)
    const dim3    qgrid((unsigned)(runs ? (nb + 7) / 8 * 8 : nb), 1, only_color >= 0 ? nkp : 2 * nkp);
    hipLaunchKernelGGL(q1_prolong_add_quad_kernel, qgrid, dim3(256), 0, (hipStream_t)stream, *L, *C, tplE, only_color, runs, gbeg, gend, ec, x_cvec);
    return launch_status();
  }
  hipLaunchKernelGGL(q1_prolong_add_kernel, grid, block, 0, (hipStream_t)stream, *L, *C, rx, ry, rz, kbegin, tplE, only_color, cpos, ec, x_cvec);
  return launch_status();
}
