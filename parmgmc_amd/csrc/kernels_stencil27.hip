// Gibbs/SOR sweep, residual and Q1 transfers for the COARSE levels of a DMDA hierarchy (gfx950).
//
// A Galerkin coarse operator P^T A P of a constant-coefficient fine operator is a 27-point (9-point in 2-D) stencil
// whose coefficients depend only on the position class of a point -- first / interior / last in each direction,
// 27 classes -- so it is stored as a 27 x 27 table (5.8 KB, in LDS) instead of 27 values + 27 column indices per
// row (324 B/row in the sliced-ELL form): a sweep moves ~24 B per unknown like the fine level.  Replaces, for these
// levels, MCSORApply_SEQAIJ (reference src/mc_sor.c:241-296) + PrepareRHS (src/pc_mcgibbs.c:119-128) and PCMG's
// residual / MatRestrict / MatInterpolateAdd (PETSc, entered at reference src/pc_gamgmc.c:246,255).
// Vectors hold the owned planes kz0 .. kz0+nz-1 of the nzg global planes in natural order (i fastest) between one
// ghost plane below and one above (element (i, j, k) at i + nx (j + ny (k - kz0 + 1))); on a single device kz0 = 0,
// nz = nzg and the ghost planes are never read, on a z-slab they hold the neighbouring device's boundary planes.
// Position classes, parities and noise counters use GLOBAL indices, so a slab run is bit-identical to the
// single-device run.  Colours are the 8 parities (i&1, j&1, k&1) -- red-black is not a valid
// colouring of a 27-point stencil -- swept in the order of their compressed index, one launch per colour.  The sum
// runs over the neighbours in ascending natural index with the diagonal skipped, i.e. in CSR storage order, absent
// neighbours contributing an exact +0, and the noise is the row stream of the natural index, so results are
// bit-identical to the sliced-ELL kernel on the assembled matrix.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "pmg_kernels.h"
#define PMG_RNG_LITERALS // the transform's constants as literals here: scalar loads in the middle of these kernels' sums cost more than they save (st27 phase +9 % by GRBM_GUI_ACTIVE)
#define PMG_RNG_TU st27
#include "pmg_rng.hpp"

namespace {

__device__ __forceinline__ int pos_class(int i, int n) { return i == 0 ? 0 : (i == n - 1 ? 2 : 1); }

// "uniform base pointer + unsigned 32-bit byte offset": one global access with a scalar base, no 64-bit address
// arithmetic on the vector unit (a level below the grid level has far fewer than 2^28 entries)
__device__ __forceinline__ const double *at_bytes(const double *base, uint32_t byte_off) { return reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + byte_off); }

// sum over the 26 neighbours of cf[e] * y[neighbour], branch-free: an absent neighbour is read at the centre (a valid
// address) and enters with its table coefficient, which is an exact zero for out-of-domain offsets.  All 26 loads are
// in flight together; with a branch per neighbour every load was waited for before the next one was issued.
// ADD = false subtracts the products from `acc` one by one (the sweep's order), true adds them (the residual's).
template <bool ADD>
__device__ __forceinline__ double st27_neighbour_sum(const pmgk_st27 &S, int i, int j, int k, int32_t centre, const double *cf, const double *__restrict__ y, double acc)
{
  const int32_t sx = S.nx, sxy = S.nx * S.ny;
  const bool    okx[3] = {i > 0, true, i < S.nx - 1}, oky[3] = {j > 0, true, j < S.ny - 1}, okz[3] = {k > 0, true, k < S.nzg - 1};
  int           e      = 0;
#pragma unroll
  for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx, ++e) {
        if (e == 13) continue; // the diagonal
        const bool    ok  = okz[dz + 1] && oky[dy + 1] && okx[dx + 1];
        const int32_t idx = ok ? centre + dx + sx * dy + sxy * dz : centre;
        const double  v   = *at_bytes(y, 8u * (uint32_t)idx);
        acc               = ADD ? acc + cf[e] * v : acc - cf[e] * v;
      }
  return acc;
}

// update of one point (i, j, global plane k): shared by the one-colour kernel and the plane kernel below
template <bool NOISY>
__device__ __forceinline__ void st27_update_point(const pmgk_st27 &S, int i, int j, int k, const double *s_coef, const double *s_idiag, const double *s_sqrtd, const pmg::LogTabEntry *s_logtab, double one_minus_omega, uint32_t key0, uint32_t key1, uint64_t sweep, const double *__restrict__ b, double *y)
{
  const int64_t row  = i + (int64_t)S.nx * (j + (int64_t)S.ny * k); // global natural index: the noise counter
  const int64_t lrow = i + (int64_t)S.nx * (j + (int64_t)S.ny * (k - S.kz0 + 1));
  const int     cls  = pos_class(i, S.nx) + 3 * pos_class(j, S.ny) + 9 * pos_class(k, S.nzg);
  const double *cf   = s_coef + 27 * cls;
  double        sum  = b[lrow];
  if (NOISY) {
    double z0, z1;
    pmg::normal_pair((uint32_t)((uint64_t)row >> 1), 0u, (uint32_t)sweep, (uint32_t)(sweep >> 32), key0, key1, s_logtab, z0, z1);
    sum = ((row & 1) ? z1 : z0) * s_sqrtd[cls] + sum;
  }
  sum = st27_neighbour_sum<false>(S, i, j, k, (int32_t)lrow, cf, y, sum); // CSR rows hold in-domain entries only
  y[lrow] = one_minus_omega * y[lrow] + s_idiag[cls] * sum;
}

template <bool NOISY>
__global__ __launch_bounds__(256) void st27_color_sweep_kernel(pmgk_st27 S, int px, int py, int kfirst, double one_minus_omega, uint32_t key0, uint32_t key1, uint64_t sweep, const double *__restrict__ b, double *y)
{
  __shared__ double           s_coef[27 * 27], s_idiag[27], s_sqrtd[27];
  __shared__ pmg::LogTabEntry s_logtab[NOISY ? PMG_LOGTAB_SIZE : 1];
  const int                   tid = threadIdx.x;
  for (int q = tid; q < 27 * 27; q += 256) s_coef[q] = S.coef[q];
  if (tid < 27) {
    s_idiag[tid] = S.idiag[tid];
    s_sqrtd[tid] = S.sqrtdiag[tid];
  }
  if (NOISY) pmg::load_log_table(s_logtab);
  __syncthreads();
  // the points of this colour in one plane, numbered line after line and dealt to the threads without gaps (a line
  // of a 2^k+1 grid has 2^(k-1)+1 points of a colour: one line per block would leave most lanes idle)
  const int cx   = (S.nx - px + 1) / 2;
  const int flat = blockIdx.x * 256 + tid;
  const int jj   = flat / cx, ii = flat - jj * cx;
  const int i = 2 * ii + px, j = 2 * jj + py, k = 2 * (int)blockIdx.z + kfirst; // k: global plane
  if (j >= S.ny || k >= S.kz0 + S.nz) return;
  st27_update_point<NOISY>(S, i, j, k, s_coef, s_idiag, s_sqrtd, s_logtab, one_minus_omega, key0, key1, sweep, b, y);
}

// The four colours of one phase (one z-parity) in ONE launch: a workgroup owns a whole plane and sweeps its four (px, py)
// colours one after the other with a workgroup barrier in between.  That is legal because the 26 neighbours of a point
// in the planes above and below have the OTHER z-parity and do not change during the phase: all dependencies between
// the colours of a phase lie inside the plane.  Against four launches: the planes k-1, k, k+1 are fetched once instead
// of four times, and the small levels pay one launch latency instead of four.  Same arithmetic per point.
template <bool NOISY>
__global__ __launch_bounds__(1024) void st27_phase_kernel(pmgk_st27 S, int backward, int phase, int kfirst, double one_minus_omega, uint32_t key0, uint32_t key1, uint64_t sweep, const double *__restrict__ b, double *y)
{
  __shared__ double           s_coef[27 * 27], s_idiag[27], s_sqrtd[27];
  __shared__ pmg::LogTabEntry s_logtab[NOISY ? PMG_LOGTAB_SIZE : 1];
  const int                   tid = threadIdx.x;
  if (tid < 27 * 27) s_coef[tid] = S.coef[tid];
  if (tid < 27) {
    s_idiag[tid] = S.idiag[tid];
    s_sqrtd[tid] = S.sqrtdiag[tid];
  }
  if (NOISY) pmg::load_log_table(s_logtab);
  __syncthreads();
  const int k = kfirst + 2 * (int)blockIdx.x; // global plane; the grid has exactly the owned planes of this parity
  for (int q = 4 * phase; q < 4 * phase + 4; ++q) {
    const int col = backward ? 7 - q : q, px = col & 1, py = (col >> 1) & 1;
    const int cx = (S.nx - px + 1) / 2, cy = (S.ny - py + 1) / 2;
    for (int f = tid; f < cx * cy; f += 1024) {
      const int jj = f / cx, ii = f - jj * cx;
      st27_update_point<NOISY>(S, 2 * ii + px, 2 * jj + py, k, s_coef, s_idiag, s_sqrtd, s_logtab, one_minus_omega, key0, key1, sweep, b, y);
    }
    __syncthreads(); // the next colour reads what this one stored (same workgroup: one L1, no cache maintenance)
  }
}

// r = b - A y; the diagonal term is added last, like sell_residual_kernel
__global__ __launch_bounds__(256) void st27_residual_kernel(pmgk_st27 S, const double *__restrict__ b, const double *__restrict__ y, double *__restrict__ r)
{
  __shared__ double s_coef[27 * 27];
  for (int q = threadIdx.x; q < 27 * 27; q += 256) s_coef[q] = S.coef[q];
  __syncthreads();
  const int flat = blockIdx.x * 256 + threadIdx.x, j = flat / S.nx, i = flat - j * S.nx, k = S.kz0 + blockIdx.z; // k: global plane
  if (j >= S.ny) return;
  const int64_t row = i + (int64_t)S.nx * (j + (int64_t)S.ny * (k - S.kz0 + 1));
  const double *cf  = s_coef + 27 * (pos_class(i, S.nx) + 3 * pos_class(j, S.ny) + 9 * pos_class(k, S.nzg));
  double sum = st27_neighbour_sum<true>(S, i, j, k, (int32_t)row, cf, y, 0.0);
  sum    = sum + cf[13] * y[row];
  r[row] = b[row] - sum;
}

// Q1 restriction b_c = P^T r and prolongation x += P e_c between two such levels.  F / C give the fine and coarse
// extents: n* global sizes, kz0 / nz the owned planes; a coarse plane K is owned by the device that owns fine plane
// 2K, its restriction reads the fine planes 2K-1 .. 2K+1 (the outer ones may be ghost planes of r).
__global__ __launch_bounds__(256) void st27_restrict_kernel(pmgk_st27_dims F, pmgk_st27_dims C, const double *__restrict__ r, double *__restrict__ bc)
{
  const int flat = blockIdx.x * 256 + threadIdx.x, J = flat / C.nx, I = flat - J * C.nx, K = C.kz0 + blockIdx.z;
  if (J >= C.ny) return;
  const int rx = F.nx != C.nx, ry = F.ny != C.ny, rz = F.nzg != C.nzg;
  const int fi = rx ? 2 * I : I, fj = ry ? 2 * J : J, fk = rz ? 2 * K : K;
  double    s  = 0.0;
  for (int dz = rz ? -1 : 0; dz <= (rz ? 1 : 0); ++dz) {
    const int k = fk + dz;
    if (k < 0 || k >= F.nzg) continue;
    for (int dy = ry ? -1 : 0; dy <= (ry ? 1 : 0); ++dy) {
      const int j = fj + dy;
      if (j < 0 || j >= F.ny) continue;
      for (int dx = rx ? -1 : 0; dx <= (rx ? 1 : 0); ++dx) {
        const int i = fi + dx;
        if (i < 0 || i >= F.nx) continue;
        const double w = (dx ? 0.5 : 1.0) * (dy ? 0.5 : 1.0) * (dz ? 0.5 : 1.0);
        s              = s + w * r[i + (int64_t)F.nx * (j + (int64_t)F.ny * (k - F.kz0 + 1))];
      }
    }
  }
  bc[I + (int64_t)C.nx * (J + (int64_t)C.ny * (K - C.kz0 + 1))] = s;
}

// fine planes kbegin .. kbegin+gridDim.z-1 (global): the owned ones and, on a slab, the in-domain ghost planes, whose
// interpolated values every device can form from its own coarse planes + coarse ghost planes
__global__ __launch_bounds__(256) void st27_prolong_add_kernel(pmgk_st27_dims F, pmgk_st27_dims C, int kbegin, const double *__restrict__ ec, double *__restrict__ x)
{
  const int flat = blockIdx.x * 256 + threadIdx.x, j = flat / F.nx, i = flat - j * F.nx, k = kbegin + blockIdx.z;
  if (j >= F.ny) return;
  const int    rx = F.nx != C.nx, ry = F.ny != C.ny, rz = F.nzg != C.nzg;
  const int    I0 = rx ? i >> 1 : i, J0 = ry ? j >> 1 : j, K0 = rz ? k >> 1 : k;
  const int    mx = (rx && (i & 1)) ? 2 : 1, my = (ry && (j & 1)) ? 2 : 1, mz = (rz && (k & 1)) ? 2 : 1;
  const double w  = (mx == 2 ? 0.5 : 1.0) * (my == 2 ? 0.5 : 1.0) * (mz == 2 ? 0.5 : 1.0);
  double       s  = 0.0;
  for (int c = 0; c < mz; ++c)
    for (int bq = 0; bq < my; ++bq)
      for (int a = 0; a < mx; ++a) s = s + w * ec[(I0 + a) + (int64_t)C.nx * ((J0 + bq) + (int64_t)C.ny * (K0 + c - C.kz0 + 1))];
  const int64_t p = i + (int64_t)F.nx * (j + (int64_t)F.ny * (k - F.kz0 + 1));
  x[p]            = x[p] + s;
}

// the usual case (all three directions refined, < 2 GiB per vector): branch-free forms of the two kernels above -- a
// fine point outside the domain is read at the centre with weight 0, an unused coarse neighbour with weight 0 at the
// used one's address; same terms in the same order, so the same bits
__global__ __launch_bounds__(256) void st27_restrict_full_kernel(pmgk_st27_dims F, pmgk_st27_dims C, const double *__restrict__ r, double *__restrict__ bc)
{
  const int flat = blockIdx.x * 256 + threadIdx.x, J = flat / C.nx, I = flat - J * C.nx, K = C.kz0 + blockIdx.z;
  if (J >= C.ny) return;
  const int32_t fnx = F.nx, fnxy = F.nx * F.ny;
  const int     fi = 2 * I, fj = 2 * J, fk = 2 * K;
  double        s  = 0.0;
#pragma unroll
  for (int dz = -1; dz <= 1; ++dz) {
    const int     k    = fk + dz;
    const bool    okz  = (unsigned)k < (unsigned)F.nzg;
    const int32_t zoff = ((okz ? k : fk) - F.kz0 + 1) * fnxy;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int     j    = fj + dy;
      const bool    oky  = (unsigned)j < (unsigned)F.ny;
      const int32_t yoff = zoff + (oky ? j : fj) * fnx;
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int     i   = fi + dx;
        const bool    okx = (unsigned)i < (unsigned)F.nx;
        const int32_t off = yoff + (okx ? i : fi);
        const double  w0  = (dx ? 0.5 : 1.0) * (dy ? 0.5 : 1.0) * (dz ? 0.5 : 1.0);
        const double  w   = (okx && oky && okz) ? w0 : 0.0;
        s                 = s + w * *at_bytes(r, 8u * (uint32_t)off);
      }
    }
  }
  bc[I + (int64_t)C.nx * (J + (int64_t)C.ny * (K - C.kz0 + 1))] = s;
}

// prolongation on a single device: thread = the coarse cell (I, J, K), i.e. the eight fine points (2I + dx, 2J + dy, 2K + dz)
// that interpolate from its eight corners: four 16-byte coarse loads and four 16-byte read-modify-writes instead of eight
// 8-byte loads and one 8-byte read-modify-write per fine point; the same sum in the same order for every point
typedef double d2u8 __attribute__((ext_vector_type(2), aligned(8)));

__global__ __launch_bounds__(256) void st27_prolong_add_cell_kernel(pmgk_st27_dims F, pmgk_st27_dims C, int gbeg, int gend, const double *__restrict__ ec, double *__restrict__ x)
{
  // global fine planes gbeg .. gend-1 (a z-slab: its planes and in-domain ghost planes); cell K = the planes 2K, 2K+1
  const int flat = blockIdx.x * 256 + threadIdx.x, J = flat / C.nx, I = flat - J * C.nx, K = (gbeg >> 1) + (int)blockIdx.z;
  if (J >= C.ny) return;
  const int32_t cnx = C.nx, cnxy = C.nx * C.ny, fnx = F.nx, fnxy = F.nx * F.ny;
  const bool    x1 = 2 * I + 1 < F.nx, y1 = 2 * J + 1 < F.ny; // the odd point / line of the cell exists
  const bool    z0 = 2 * K >= gbeg && 2 * K < gend, z1 = 2 * K + 1 >= gbeg && 2 * K + 1 < gend; // its planes are in the range
  const int32_t cbase = (K - C.kz0 + 1) * cnxy + J * cnx + I, fbase = (2 * K - F.kz0 + 1) * fnxy + 2 * J * fnx + 2 * I;
  // fine values first
  double f[2][2][2];
#pragma unroll
  for (int dz = 0; dz < 2; ++dz)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const int32_t off = fbase + ((dz ? z1 : !z0) ? fnxy : 0) + (dy && y1 ? fnx : 0); // a plane outside the range: the other one's address
      if (x1) {
        const d2u8 v = *reinterpret_cast<const d2u8 *>(x + off);
        f[dz][dy][0] = v.x;
        f[dz][dy][1] = v.y;
      } else f[dz][dy][0] = f[dz][dy][1] = x[off];
    }
  double e[2][2][2]; // e[a][b][c] = coarse (K + a, J + b, I + c), clamped where the cell has no odd side (never used there)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int bq = 0; bq < 2; ++bq) {
      const int32_t off = cbase + (a && z1 ? cnxy : 0) + (bq && y1 ? cnx : 0);
      if (x1) {
        const d2u8 v = *reinterpret_cast<const d2u8 *>(ec + off);
        e[a][bq][0]  = v.x;
        e[a][bq][1]  = v.y;
      } else e[a][bq][0] = e[a][bq][1] = ec[off];
    }
#pragma unroll
  for (int dz = 0; dz < 2; ++dz)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      if ((dz ? !z1 : !z0) || (dy && !y1)) continue;
      double out[2];
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        double sum = 0.0;
#pragma unroll
        for (int cz = 0; cz < 2; ++cz) {
          const double wz = dz ? 0.5 : (cz ? 0.0 : 1.0);
#pragma unroll
          for (int by = 0; by < 2; ++by) {
            const double wy = dy ? 0.5 : (by ? 0.0 : 1.0);
#pragma unroll
            for (int ax = 0; ax < 2; ++ax) {
              const double wx = dx ? 0.5 : (ax ? 0.0 : 1.0);
              sum             = sum + (wx * wy * wz) * e[cz & dz][by & dy][ax & dx];
            }
          }
        }
        out[dx] = f[dz][dy][dx] + sum;
      }
      const int32_t off = fbase + (dz ? fnxy : 0) + (dy ? fnx : 0);
      if (x1) *reinterpret_cast<d2u8 *>(x + off) = d2u8{out[0], out[1]};
      else x[off] = out[0];
    }
}

inline bool st27_transfer_full_case(const pmgk_st27_dims *F, const pmgk_st27_dims *C)
{
  static const int off = getenv("PMG_TRANSFER_GENERIC") != nullptr;
  if (off || F->nx == C->nx || F->ny == C->ny || F->nzg == C->nzg) return false;
  return (int64_t)F->nx * F->ny * (F->nz + 2) * 8 < ((int64_t)1 << 31);
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? 0 : 1; }

} // namespace

// half of one directional sweep: phase 0 = the first four colours of the sweep order, phase 1 = the last four.  A
// phase touches planes of ONE z-parity (forward: even planes first), so on a z-slab the boundary planes are exchanged
// once per phase.
extern "C" int pmgk_st27_sweep_phase(const pmgk_st27 *S, int backward, int phase, double omega, int noisy, uint64_t seed, uint64_t sweep, const double *b, double *y, void *stream)
{
  const double om1 = 1. - omega;
  // one workgroup per plane pays only while the launch latency of four kernels is what a phase costs: measured on
  // MI355X, planes of 33 x 33 points gain (257^3 V-cycle 0.912 -> 0.893 ms), 65 x 65 break even, 129 x 129 and larger
  // lose (one CU per plane is latency-bound: 1.05 ms)
  static int plane_limit = -1;
  if (plane_limit < 0) {
    const char *e = getenv("PMG_ST27_PHASE_MAX_PLANE");
    plane_limit   = e ? atoi(e) : 1100; /* points per plane up to which the plane kernel is used */
  }
  if ((int64_t)S->nx * S->ny <= plane_limit) {
    const int pz = backward ? 1 - phase : phase;
    const int kfirst = S->kz0 + ((pz - S->kz0) & 1), cz = (S->kz0 + S->nz - kfirst + 1) / 2;
    if (cz <= 0) return 0;
    if (noisy) hipLaunchKernelGGL((st27_phase_kernel<true>), dim3(cz), dim3(1024), 0, (hipStream_t)stream, *S, backward, phase, kfirst, om1, (uint32_t)seed, (uint32_t)(seed >> 32), sweep, b, y);
    else hipLaunchKernelGGL((st27_phase_kernel<false>), dim3(cz), dim3(1024), 0, (hipStream_t)stream, *S, backward, phase, kfirst, om1, (uint32_t)seed, (uint32_t)(seed >> 32), sweep, b, y);
    return launch_status();
  }
  for (int q = 4 * phase; q < 4 * phase + 4; ++q) {
    const int col = backward ? 7 - q : q;
    const int px = col & 1, py = (col >> 1) & 1, pz = (col >> 2) & 1;
    const int kfirst = S->kz0 + ((pz - S->kz0) & 1); // first owned plane of parity pz
    const int cx = (S->nx - px + 1) / 2, cy = (S->ny - py + 1) / 2, cz = (S->kz0 + S->nz - kfirst + 1) / 2; // points of this parity
    if (cx <= 0 || cy <= 0 || cz <= 0) continue;
    const dim3 grid((unsigned)(((int64_t)cx * cy + 255) / 256), 1, cz), block(256);
    if (noisy) hipLaunchKernelGGL((st27_color_sweep_kernel<true>), grid, block, 0, (hipStream_t)stream, *S, px, py, kfirst, om1, (uint32_t)seed, (uint32_t)(seed >> 32), sweep, b, y);
    else hipLaunchKernelGGL((st27_color_sweep_kernel<false>), grid, block, 0, (hipStream_t)stream, *S, px, py, kfirst, om1, (uint32_t)seed, (uint32_t)(seed >> 32), sweep, b, y);
  }
  return launch_status();
}

// one directional sweep: the 8 (4 in 2-D) parity colours in ascending / descending compressed-colour order
extern "C" int pmgk_st27_sweep(const pmgk_st27 *S, int backward, double omega, int noisy, uint64_t seed, uint64_t sweep, const double *b, double *y, void *stream)
{
  if (pmgk_st27_sweep_phase(S, backward, 0, omega, noisy, seed, sweep, b, y, stream)) return 1;
  return pmgk_st27_sweep_phase(S, backward, 1, omega, noisy, seed, sweep, b, y, stream);
}

extern "C" int pmgk_st27_residual(const pmgk_st27 *S, const double *b, const double *y, double *r, void *stream)
{
  if (S->nz <= 0) return 0;
  hipLaunchKernelGGL(st27_residual_kernel, dim3((unsigned)(((int64_t)S->nx * S->ny + 255) / 256), 1, S->nz), dim3(256), 0, (hipStream_t)stream, *S, b, y, r);
  return launch_status();
}

extern "C" int pmgk_st27_restrict(const pmgk_st27_dims *F, const pmgk_st27_dims *C, const double *r, double *bc, void *stream)
{
  if (C->nz <= 0) return 0;
  if (st27_transfer_full_case(F, C)) {
    hipLaunchKernelGGL(st27_restrict_full_kernel, dim3((unsigned)(((int64_t)C->nx * C->ny + 255) / 256), 1, C->nz), dim3(256), 0, (hipStream_t)stream, *F, *C, r, bc);
    return launch_status();
  }
  hipLaunchKernelGGL(st27_restrict_kernel, dim3((unsigned)(((int64_t)C->nx * C->ny + 255) / 256), 1, C->nz), dim3(256), 0, (hipStream_t)stream, *F, *C, r, bc);
  return launch_status();
}

extern "C" int pmgk_st27_prolong_add(const pmgk_st27_dims *F, const pmgk_st27_dims *C, int kbegin, int kcount, const double *ec, double *x, void *stream)
{
  if (kcount <= 0) return 0;
  if (st27_transfer_full_case(F, C) && C->nx == (F->nx + 1) / 2 && C->ny == (F->ny + 1) / 2 && C->nzg == (F->nzg + 1) / 2) {
    const int gend = kbegin + kcount, ncell = ((gend - 1) >> 1) - (kbegin >> 1) + 1; /* kbegin: global plane */
    hipLaunchKernelGGL(st27_prolong_add_cell_kernel, dim3((unsigned)(((int64_t)C->nx * C->ny + 255) / 256), 1, ncell), dim3(256), 0, (hipStream_t)stream, *F, *C, kbegin, gend, ec, x);
    return launch_status();
  }
  hipLaunchKernelGGL(st27_prolong_add_kernel, dim3((unsigned)(((int64_t)F->nx * F->ny + 255) / 256), 1, kcount), dim3(256), 0, (hipStream_t)stream, *F, *C, kbegin, ec, x);
  return launch_status();
}
