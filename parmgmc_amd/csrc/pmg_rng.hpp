// Counter-based Gaussian noise for the Gibbs sweeps (device side, gfx950).
//
// Replaces VecSetRandomStandardNormal (reference src/parmgmc.c:70-116): same Box-Muller transform
// (radius = sqrt(-2 ln u1), theta = 2 pi u2, cos branch for the even entry of a pair and sin branch for the
// odd one, :99-110) but on a counter-based uniform source, Philox4x32-10 (Salmon et al., SC'11), so that a
// normal is a pure function of (seed, sweep number, global index) and a chain is identical on 1/2/4/8 GPUs.
//
// The transcendental part is hand-written for the fp64 VALU (the sweep is VALU-bound with the stock libm
// calls): both uniforms stay 53-bit INTEGERS as long as possible --
//   ln u1 : exponent/mantissa split on the integer, z in [0.6875,1.375) so that u -> 1 needs no cancelling
//           k*ln2 term, 89-entry {1/c, ln c} table (c = i/128) in LDS, r = z/c - 1 by one fma, degree-7 log1p;
//   sqrt  : v_rsq_f64 + one Goldschmidt step + two residual corrections;
//   sin/cos(2 pi u2): quadrant from the top bits of the integer, remainder |r| <= 1/8 turn converted exactly,
//           Taylor series in turns (no multiplication by pi, no range-reduction error).
// Every piece is accurate to ~1 ulp; tests compare with glibc log/sin/cos at 1e-13.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pmg_rng_tables.inc"

namespace pmg {

struct Philox4 {
  uint32_t r0, r1, r2, r3;
};

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0; // one v_mad_u64_u32 gives hi and lo
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c0 = n0;
    c1 = (uint32_t)p1;
    c2 = n2;
    c3 = (uint32_t)p0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}

// fp64 FMAs with one operand in a SCALAR register pair.  A VALU fp64 instruction takes no 64-bit literal, and left to
// itself the compiler turns every Horner step fma(s, w, C) into two v_mov_b32 (C into the accumulator) + v_fmac_f64:
// three vector instructions in a kernel that is bound by vector issue.  With the constant in SGPRs (two s_mov_b32 on
// the scalar unit, which has slots to spare) the step is ONE v_fma_f64.  `c` MUST be a compile-time constant (a
// wave-uniform value): the "s" constraint would silently broadcast lane 0's value otherwise.  Same operation, same
// rounding as fma().
__device__ __forceinline__ double fma_vvs(double a, double b, double c) // a * b + c
{
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
  return r;
}
__device__ __forceinline__ double fma_vsv(double a, double c, double b) // a * c + b
{
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(c), "v"(b));
  return r;
}

struct LogTabEntry {
  double invc, logc;
};
__device__ const LogTabEntry g_logtab[PMG_LOGTAB_SIZE] = {PMG_LOGTAB_ENTRIES};

// Copies the log table into LDS; every thread of the block must call it, followed by __syncthreads().
__device__ __forceinline__ void load_log_table(LogTabEntry *lds_tab)
{
  for (int i = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z); i < PMG_LOGTAB_SIZE; i += blockDim.x * blockDim.y * blockDim.z) lds_tab[i] = g_logtab[i];
}

// Per-wavefront copy (no block barrier: short-lived blocks must not serialise a table fetch in front of their
// streaming loads).  `wave_tab` is this wave's private PMG_LOGTAB_SIZE-entry LDS region; a wavefront's LDS
// operations execute in order, the wave barrier only stops the compiler from reordering them.
__device__ __forceinline__ void load_log_table_wave(LogTabEntry *wave_tab, int lane)
{
  wave_tab[lane] = g_logtab[lane];
  if (lane < PMG_LOGTAB_SIZE - 64) wave_tab[64 + lane] = g_logtab[64 + lane];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// X = (x >> 11) + 1 for the 64-bit word x = hi:lo, as (Xhi:Xlo); X in [1, 2^53].  u = X * 2^-53 in (0,1]:
// never 0, so ln u is finite (the reference's PetscRandom may return 0 and then yields inf,
// src/parmgmc.c:103-106).
__device__ __forceinline__ void u53_int(uint32_t lo, uint32_t hi, uint32_t &xlo, uint32_t &xhi)
{
  const uint32_t l = (lo >> 11) | (hi << 21);
  xlo              = l + 1u;
  xhi              = (hi >> 11) + (xlo == 0u ? 1u : 0u);
}

// s = -2 ln(X * 2^-53), X = xhi:xlo in [1, 2^53]
__device__ __forceinline__ double minus2_log_u(uint32_t xlo, uint32_t xhi, const LogTabEntry *tab)
{
  const double d = fma_vsv((double)xhi, 4294967296.0, (double)xlo); // exact
  // d = 2^k' * z with z in [0.6875, 1.375): subtract the bit pattern of 0.6875 from the high word
  const uint32_t dh  = (uint32_t)__double2hiint(d);
  const uint32_t tmp = dh - 0x3fe60000u;
  const int      k   = (int)tmp >> 20; // exponent of d relative to z
  const double   z   = __hiloint2double((int)(dh - ((uint32_t)k << 20)), __double2loint(d));
  const int      i   = (int)fma(z, 128.0, 0.5); // round(z*128) in [88,176]
  const LogTabEntry e = tab[i - PMG_LOGTAB_FIRST];
  const double   r  = fma(z, e.invc, -1.0); // z/c - 1, |r| <= 1/176
  const double   w  = r * r;
  // log1p(r) = r - w/2 + r w (1/3 - r/4 + w (1/5 - r/6 + w/7))
  const double A  = fma(r, -0.25, 1.0 / 3.0);
  const double B  = fma(r, -1.0 / 6.0, 0.2);
  const double q  = fma(w, fma_vsv(w, 1.0 / 7.0, B), A);
  const double lp = fma(r * w, q, fma(w, -0.5, r));
  const double kd = (double)(k - 53); // ln u = (k-53) ln2 + ln c + log1p(r)
  const double hi = fma_vsv(kd, PMG_LN2_HI, e.logc);
  const double ln = hi + fma_vsv(kd, PMG_LN2_LO, lp);
  return -2.0 * ln;
}

// sqrt(s) for 0 <= s < 128 (s = 0 returns ~1e-150, numerically zero)
__device__ __forceinline__ double sqrt_pos(double s)
{
  s              = fmax(s, 0x1.0p-1000);
  const double y = __builtin_amdgcn_rsq(s);
  double       g = s * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g              = fma(g, r, g);
  h              = fma(h, r, h);
  double dd      = fma(-g, g, s);
  g              = fma(dd, h, g);
  dd             = fma(-g, g, s);
  g              = fma(dd, h, g);
  return g;
}

// sin and cos of 2 pi Y 2^-53 for the integer Y = yhi:ylo in [1, 2^53]
__device__ __forceinline__ void sincos_turns(uint32_t ylo, uint32_t yhi, double &sn, double &cs)
{
  const uint32_t q  = (yhi + (1u << 18)) >> 19;        // nearest quarter turn, 0..4
  const int32_t  rh = (int32_t)(yhi - (q << 19));      // remainder Y - q 2^51 in [-2^50, 2^50], high word
  const double   r  = fma_vsv((double)rh, 4294967296.0, (double)ylo) * 0x1.0p-53; // turns, |r| <= 1/8, exact
  const double   w  = r * r;
  double s = PMG_SIN_C8, c = PMG_COS_C9;
  s = fma_vvs(s, w, PMG_SIN_C7);
  c = fma_vvs(c, w, PMG_COS_C8);
  s = fma_vvs(s, w, PMG_SIN_C6);
  c = fma_vvs(c, w, PMG_COS_C7);
  s = fma_vvs(s, w, PMG_SIN_C5);
  c = fma_vvs(c, w, PMG_COS_C6);
  s = fma_vvs(s, w, PMG_SIN_C4);
  c = fma_vvs(c, w, PMG_COS_C5);
  s = fma_vvs(s, w, PMG_SIN_C3);
  c = fma_vvs(c, w, PMG_COS_C4);
  s = fma_vvs(s, w, PMG_SIN_C2);
  c = fma_vvs(c, w, PMG_COS_C3);
  s = fma_vvs(s, w, PMG_SIN_C1);
  c = fma_vvs(c, w, PMG_COS_C2);
  s = fma_vvs(s, w, PMG_SIN_C0);
  c = fma_vvs(c, w, PMG_COS_C1);
  s = s * r;
  c = fma(c, w, 1.0);
  // rotate by q quarter turns: q odd swaps, sin negated for q = 2,3, cos negated for q = 1,2
  const bool   odd = q & 1u;
  const double s0  = odd ? c : s;
  const double c0  = odd ? s : c;
  sn               = (q & 2u) ? -s0 : s0;
  cs               = ((q + 1u) & 2u) ? -c0 : c0;
}

// One Box-Muller pair from one Philox block.  z0 = r cos(2 pi u2), z1 = r sin(2 pi u2).
__device__ __forceinline__ void normal_pair(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, const LogTabEntry *tab, double &z0, double &z1)
{
  const Philox4 p = philox4x32_10(c0, c1, c2, c3, k0, k1);
  uint32_t      xlo, xhi, ylo, yhi;
  u53_int(p.r0, p.r1, xlo, xhi);
  u53_int(p.r2, p.r3, ylo, yhi);
  const double radius = sqrt_pos(minus2_log_u(xlo, xhi, tab));
  double       s, c;
  sincos_turns(ylo, yhi, s, c);
  z0 = radius * c;
  z1 = radius * s;
}

} // namespace pmg
