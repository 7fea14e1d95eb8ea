// Matrix-free red-black Gibbs/SOR sweep on colour-partitioned DMDA vectors (gfx950).
//
// Replaces, for the operator of MatAssembleShiftedLaplaceFD (reference src/problems.c:14-75, 3-D analogue),
// the CPU loops
//   MCSORApply_SEQAIJ / _MPIAIJ      reference src/mc_sor.c:241-296 / :298-381   (row update)
//   PrepareRHS_Default                reference src/pc_mcgibbs.c:119-128          (w = xi*sqrtdiag + b)
//   VecSetRandomStandardNormal        reference src/parmgmc.c:70-116              (xi)
// fused into one pass per colour: the noisy right-hand side is formed in registers and never stored.
//
// Arithmetic is kept in the reference's order so that the deterministic sweep is bit-identical to the CSR
// loop: sum starts at w, the off-diagonal terms are subtracted in CSR storage order
// (k-1)(j-1)(i-1)(i+1)(j+1)(k+1), each as its own rounded product, then y = (1-omega)*y + idiag*sum.  With
// a = -h2 the reference's "sum -= a*y" equals "sum += h2*y" bit for bit.  The file is compiled with
// -ffp-contract=off (no FMA contraction), like the CPU oracle.
#include <hip/hip_runtime.h>
#include "pmg_kernels.h"
#include "pmg_rng.hpp"

namespace {

struct d2 {
  double x, y;
};

__device__ __forceinline__ d2 ld2(const double *p) { return *reinterpret_cast<const d2 *>(p); }

// Sweep of one colour.  Thread = two consecutive points (m = 2t, 2t+1) of colour `c` on one grid line.
template <bool NOISY, bool OMEGA1>
__global__ __launch_bounds__(256) void grid_color_sweep_kernel(pmgk_grid_layout L, pmgk_grid_op op, int c, const double *__restrict__ b, double *__restrict__ y)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (j >= L.ny || 2 * t >= L.sx) return;
  const int kg = k + L.kz0;
  const int p  = (c + j + kg) & 1;
  const int i0 = 4 * t + p, i1 = i0 + 2; // grid columns of the two points
  if (i0 >= L.nx) return;
  const bool v1 = i1 < L.nx;

  const int64_t line = (int64_t)(k + 1) * L.sp + (int64_t)j * L.sx;
  const double *yo   = y + (int64_t)(1 - c) * L.cs + line; // other colour, same line
  double       *ys   = y + (int64_t)c * L.cs + line;       // this colour
  const double *bs   = b + (int64_t)c * L.cs + line;

  const bool hasD = kg > 0, hasU = kg < L.nzg - 1, hasS = j > 0, hasN = j < L.ny - 1;
  const d2   zero = {0.0, 0.0};
  // other-colour values: same line, m' = 2t-1+p .. 2t+1+p
  const d2     oc  = ld2(yo + 2 * t);
  // p=0: left(0)=m' 2t-1, right(0)=2t,   left(1)=2t,   right(1)=2t+1
  // p=1: left(0)=m' 2t,   right(0)=2t+1, left(1)=2t+1, right(1)=2t+2
  const double L0 = p ? oc.x : (t > 0 ? yo[2 * t - 1] : 0.0);
  const double R0 = p ? oc.y : oc.x;
  const double L1 = p ? oc.y : oc.x;
  const double R1 = p ? ((2 * t + 2 < L.sx) ? yo[2 * t + 2] : 0.0) : oc.y;
  const d2 oS = hasS ? ld2(yo - L.sx + 2 * t) : zero;
  const d2 oN = hasN ? ld2(yo + L.sx + 2 * t) : zero;
  const d2 oD = hasD ? ld2(yo - L.sp + 2 * t) : zero;
  const d2 oU = hasU ? ld2(yo + L.sp + 2 * t) : zero;
  const d2 bb = ld2(bs + 2 * t);

  const bool hasW0 = i0 > 0, hasE0 = i0 < L.nx - 1;
  const bool hasW1 = true, hasE1 = i1 < L.nx - 1; // i1 >= 2 always has a west neighbour
  const int  nyz = (int)hasD + (int)hasU + (int)hasS + (int)hasN;
  const int  nn0 = nyz + (int)hasW0 + (int)hasE0;
  const int  nn1 = nyz + (int)hasW1 + (int)hasE1;

  double w0 = bb.x, w1 = bb.y;
  if (NOISY) {
    double z0, z1;
    pmg::normal_pair((uint32_t)t, (uint32_t)(j + (int64_t)L.ny * kg), (uint32_t)op.sweep, ((uint32_t)(op.sweep >> 32) & 0x7fffffffu) | ((uint32_t)c << 31), op.key0, op.key1, z0, z1);
    w0 = z0 * op.sqrtdiag[nn0] + bb.x;
    w1 = z1 * op.sqrtdiag[nn1] + bb.y;
  }
  const double h2 = op.h2;
  double       s0 = w0, s1 = w1;
  // CSR storage order: (k-1) (j-1) (i-1) | (i+1) (j+1) (k+1); absent neighbours contribute an exact +0
  s0 = s0 + h2 * oD.x;
  s1 = s1 + h2 * oD.y;
  s0 = s0 + h2 * oS.x;
  s1 = s1 + h2 * oS.y;
  s0 = s0 + h2 * (hasW0 ? L0 : 0.0);
  s1 = s1 + h2 * L1;
  s0 = s0 + h2 * (hasE0 ? R0 : 0.0);
  s1 = s1 + h2 * (hasE1 ? R1 : 0.0);
  s0 = s0 + h2 * oN.x;
  s1 = s1 + h2 * oN.y;
  s0 = s0 + h2 * oU.x;
  s1 = s1 + h2 * oU.y;

  double r0, r1;
  if (OMEGA1) {
    // (1-omega)*y == 0 exactly; the reference still adds it (src/mc_sor.c:267), which can only change the
    // sign of an exact zero -- numerically equal
    r0 = op.idiag[nn0] * s0;
    r1 = op.idiag[nn1] * s1;
  } else {
    const d2 yo2 = ld2(ys + 2 * t);
    r0           = op.one_minus_omega * yo2.x + op.idiag[nn0] * s0;
    r1           = op.one_minus_omega * yo2.y + op.idiag[nn1] * s1;
  }
  if (v1) {
    d2 out = {r0, r1};
    *reinterpret_cast<d2 *>(ys + 2 * t) = out;
  } else {
    ys[2 * t] = r0;
  }
}

// natural (DMDA, i fastest) <-> colour-partitioned storage
__global__ void grid_to_cvec_kernel(pmgk_grid_layout L, const double *__restrict__ nat, double *__restrict__ cv)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= L.nx || j >= L.ny) return;
  const int c = (i + j + k + L.kz0) & 1;
  cv[(int64_t)c * L.cs + (int64_t)(k + 1) * L.sp + (int64_t)j * L.sx + (i >> 1)] = nat[i + (int64_t)L.nx * (j + (int64_t)L.ny * k)];
}

__global__ void grid_from_cvec_kernel(pmgk_grid_layout L, const double *__restrict__ cv, double *__restrict__ nat)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= L.nx || j >= L.ny) return;
  const int c = (i + j + k + L.kz0) & 1;
  nat[i + (int64_t)L.nx * (j + (int64_t)L.ny * k)] = cv[(int64_t)c * L.cs + (int64_t)(k + 1) * L.sp + (int64_t)j * L.sx + (i >> 1)];
}

// r = b - A y for both colours (one thread per point of colour c = blockIdx-parity free variant: thread handles
// the point pair like the sweep).  Row sum in CSR order INCLUDING the diagonal at its place, as MatMult does,
// then r = b - s  (VecAYPX(w,-1,b), reference src/pc_gamgmc.c:253-254).
__global__ __launch_bounds__(256) void grid_residual_kernel(pmgk_grid_layout L, pmgk_grid_op op, int c, const double *__restrict__ b, const double *__restrict__ y, double *__restrict__ r)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (j >= L.ny || 2 * t >= L.sx) return;
  const int kg = k + L.kz0;
  const int p  = (c + j + kg) & 1;
  const int i0 = 4 * t + p, i1 = i0 + 2;
  if (i0 >= L.nx) return;
  const bool    v1   = i1 < L.nx;
  const int64_t line = (int64_t)(k + 1) * L.sp + (int64_t)j * L.sx;
  const double *yo   = y + (int64_t)(1 - c) * L.cs + line;
  const double *ys   = y + (int64_t)c * L.cs + line;
  const double *bs   = b + (int64_t)c * L.cs + line;
  double       *rs   = r + (int64_t)c * L.cs + line;
  const bool    hasD = kg > 0, hasU = kg < L.nzg - 1, hasS = j > 0, hasN = j < L.ny - 1;
  const d2      zero = {0.0, 0.0};
  const d2      oc   = ld2(yo + 2 * t);
  const double  L0   = p ? oc.x : (t > 0 ? yo[2 * t - 1] : 0.0);
  const double  R0   = p ? oc.y : oc.x;
  const double  L1   = p ? oc.y : oc.x;
  const double  R1   = p ? ((2 * t + 2 < L.sx) ? yo[2 * t + 2] : 0.0) : oc.y;
  const d2      oS   = hasS ? ld2(yo - L.sx + 2 * t) : zero;
  const d2      oN   = hasN ? ld2(yo + L.sx + 2 * t) : zero;
  const d2      oD   = hasD ? ld2(yo - L.sp + 2 * t) : zero;
  const d2      oU   = hasU ? ld2(yo + L.sp + 2 * t) : zero;
  const d2      bb   = ld2(bs + 2 * t);
  const d2      yy   = ld2(ys + 2 * t);
  const bool    hasW0 = i0 > 0, hasE0 = i0 < L.nx - 1, hasE1 = i1 < L.nx - 1;
  const int     nyz = (int)hasD + (int)hasU + (int)hasS + (int)hasN;
  const int     nn0 = nyz + (int)hasW0 + (int)hasE0, nn1 = nyz + 1 + (int)hasE1;
  const double  a   = -op.h2;
  double        s0 = 0.0, s1 = 0.0;
  s0 = s0 + a * oD.x;
  s1 = s1 + a * oD.y;
  s0 = s0 + a * oS.x;
  s1 = s1 + a * oS.y;
  s0 = s0 + a * (hasW0 ? L0 : 0.0);
  s1 = s1 + a * L1;
  s0 = s0 + op.diag[nn0] * yy.x;
  s1 = s1 + op.diag[nn1] * yy.y;
  s0 = s0 + a * (hasE0 ? R0 : 0.0);
  s1 = s1 + a * (hasE1 ? R1 : 0.0);
  s0 = s0 + a * oN.x;
  s1 = s1 + a * oN.y;
  s0 = s0 + a * oU.x;
  s1 = s1 + a * oU.y;
  const double r0 = bb.x - s0, r1 = bb.y - s1;
  if (v1) {
    d2 out = {r0, r1};
    *reinterpret_cast<d2 *>(rs + 2 * t) = out;
  } else {
    rs[2 * t] = r0;
  }
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? 0 : 1; }

} // namespace

extern "C" int pmgk_grid_color_sweep(const pmgk_grid_layout *L, const pmgk_grid_op *op, int color, const double *b, double *y, void *stream)
{
  const int  tpl = L->sx / 2; // threads per line
  const dim3 block(64, 4, 1);
  const dim3 grid((tpl + 63) / 64, (L->ny + 3) / 4, L->nz);
  hipStream_t s = (hipStream_t)stream;
  if (op->noisy) {
    if (op->omega_is_one) hipLaunchKernelGGL((grid_color_sweep_kernel<true, true>), grid, block, 0, s, *L, *op, color, b, y);
    else hipLaunchKernelGGL((grid_color_sweep_kernel<true, false>), grid, block, 0, s, *L, *op, color, b, y);
  } else {
    if (op->omega_is_one) hipLaunchKernelGGL((grid_color_sweep_kernel<false, true>), grid, block, 0, s, *L, *op, color, b, y);
    else hipLaunchKernelGGL((grid_color_sweep_kernel<false, false>), grid, block, 0, s, *L, *op, color, b, y);
  }
  return launch_status();
}

extern "C" int pmgk_grid_residual(const pmgk_grid_layout *L, const pmgk_grid_op *op, const double *b, const double *y, double *r, void *stream)
{
  const int  tpl = L->sx / 2;
  const dim3 block(64, 4, 1);
  const dim3 grid((tpl + 63) / 64, (L->ny + 3) / 4, L->nz);
  for (int c = 0; c < 2; ++c) hipLaunchKernelGGL(grid_residual_kernel, grid, block, 0, (hipStream_t)stream, *L, *op, c, b, y, r);
  return launch_status();
}

extern "C" int pmgk_grid_to_cvec(const pmgk_grid_layout *L, const double *nat, double *cvec, void *stream)
{
  const dim3 block(64, 4, 1);
  const dim3 grid((L->nx + 63) / 64, (L->ny + 3) / 4, L->nz);
  hipLaunchKernelGGL(grid_to_cvec_kernel, grid, block, 0, (hipStream_t)stream, *L, nat, cvec);
  return launch_status();
}

extern "C" int pmgk_grid_from_cvec(const pmgk_grid_layout *L, const double *cvec, double *nat, void *stream)
{
  const dim3 block(64, 4, 1);
  const dim3 grid((L->nx + 63) / 64, (L->ny + 3) / 4, L->nz);
  hipLaunchKernelGGL(grid_from_cvec_kernel, grid, block, 0, (hipStream_t)stream, *L, cvec, nat);
  return launch_status();
}
