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
//           k*ln2 term, 64-entry {1/c, ln c} table (c = i/92: one entry per lane of a wavefront, so ONE 16-byte load
//           per lane fills a wavefront's copy) in LDS, r = z/c - 1 by one fma, degree-7 log1p;
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

// The fp64 constants of the transform, fetched with SCALAR loads (s_load_dwordx8/x16 from constant memory) instead of being
// materialised by two s_mov_b32 each: the sweep kernels issue as many scalar as vector instructions, the CU's one scalar
// unit serves four SIMDs, and 24 constants were 48 of ~230 scalar instructions per wavefront (tools/valubench.hip: an
// s_mov_b32 costs a SIMD's issue port what a v_fma_f64 does).  Not `const`: the optimiser would fold the values back into
// literals.  For the same reason the array has EXTERNAL linkage (a `static` one that nothing in the translation unit
// writes is folded just the same), so every kernel file names its copy: #define PMG_RNG_TU <tag> before this header.
#ifndef PMG_RNG_TU
#error "define PMG_RNG_TU (a tag unique to the including .hip file) before including pmg_rng.hpp"
#endif
#define PMG_RNG_CAT2(a, b) a##b
#define PMG_RNG_CAT(a, b) PMG_RNG_CAT2(a, b)
#define g_rngc PMG_RNG_CAT(g_pmg_rngc_, PMG_RNG_TU)
enum {
  RC_SIN0 = 0, // .. RC_SIN0 + 8
  RC_COS1 = 9, // .. RC_COS1 + 8  (COS_C1..C9)
  RC_LOG0  = 18,
  RC_THIRD = 18,
  RC_MSIXTH,
  RC_FIFTH,
  RC_SEVENTH,
  RC_LN2_HI,
  RC_LN2_LO,
  RC_COUNT
};
#define PMG_RNGC_VALUES PMG_SIN_C0, PMG_SIN_C1, PMG_SIN_C2, PMG_SIN_C3, PMG_SIN_C4, PMG_SIN_C5, PMG_SIN_C6, PMG_SIN_C7, PMG_SIN_C8, PMG_COS_C1, PMG_COS_C2, PMG_COS_C3, PMG_COS_C4, PMG_COS_C5, PMG_COS_C6, PMG_COS_C7, PMG_COS_C8, PMG_COS_C9, 1.0 / 3.0, -1.0 / 6.0, 0.2, 1.0 / 7.0, PMG_LN2_HI, PMG_LN2_LO
__constant__ double g_rngc[RC_COUNT] = {PMG_RNGC_VALUES};

// ONE base address in a scalar register pair (opaque to the optimiser, which would otherwise address every element
// pc-relative by itself) and constant-address-space loads at immediate offsets from it, which the backend merges into
// s_load_dwordx4/x8/x16
typedef const double __attribute__((address_space(4))) *rng_consts_t;
__device__ __forceinline__ rng_consts_t rng_consts()
{
  rng_consts_t p = (rng_consts_t)(unsigned long long)&g_rngc[0];
  asm("" : "+s"(p));
  return p;
}

// in scalar registers, in two groups so that a wavefront stays within the 96 SGPRs that eight wavefronts per SIMD allow:
// a kernel calls load_sincos_consts() where it wants those loads ISSUED (in front of the Philox rounds: they are waited
// for only at the first use); the six of the logarithm are fetched behind the sin/cos polynomials, whose registers they
// take over, in the shadow of the table exchange through LDS
struct RngConsts {
  double c[RC_LOG0];
};
struct RngLogConsts {
  double c[RC_COUNT - RC_LOG0];
};
__device__ __forceinline__ RngConsts load_sincos_consts()
{
#ifndef PMG_RNG_LITERALS
  const rng_consts_t p = rng_consts();
#else
  constexpr double p[RC_COUNT] = {PMG_RNGC_VALUES};
#endif
  RngConsts K;
#pragma unroll
  for (int i = 0; i < RC_LOG0; ++i) K.c[i] = p[i];
  return K;
}
__device__ __forceinline__ RngLogConsts load_log_consts()
{
#ifndef PMG_RNG_LITERALS
  const rng_consts_t p = rng_consts();
#else
  constexpr double p[RC_COUNT] = {PMG_RNGC_VALUES};
#endif
  RngLogConsts K;
#pragma unroll
  for (int i = 0; i < RC_COUNT - RC_LOG0; ++i) K.c[i] = p[RC_LOG0 + i];
  return K;
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
  static_assert(PMG_LOGTAB_SIZE == 64, "one table entry per lane");
  wave_tab[lane] = g_logtab[lane];
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
__device__ __forceinline__ double minus2_log_u(uint32_t xlo, uint32_t xhi, const LogTabEntry *tab, const RngLogConsts &K)
{
  const double *rc = K.c - RC_LOG0;
  const double d = fma_vsv((double)xhi, 4294967296.0, (double)xlo); // exact
  // d = 2^k' * z with z in [0.6875, 1.375): subtract the bit pattern of 0.6875 from the high word
  const uint32_t dh  = (uint32_t)__double2hiint(d);
  const uint32_t tmp = dh - 0x3fe60000u;
  const int      k   = (int)tmp >> 20; // exponent of d relative to z
  const double   z   = __hiloint2double((int)(dh - ((uint32_t)k << 20)), __double2loint(d));
  const int      i   = (int)fma(z, PMG_LOGTAB_SCALE, 0.5); // round(z*92) in [63,126]
  const LogTabEntry e = tab[i - PMG_LOGTAB_FIRST];
  const double   r  = fma(z, e.invc, -1.0); // z/c - 1, |r| <= 1/126
  const double   w  = r * r;
  // log1p(r) = r - w/2 + r w (1/3 - r/4 + r^2/5 - r^3/6 + r^4/7), Horner in r: one constant per step, each from a scalar
  // register pair
  double q = r * rc[RC_SEVENTH];
  q        = q + rc[RC_MSIXTH];
  q        = fma_vvs(q, r, rc[RC_FIFTH]);
  q        = fma(q, r, -0.25);
  q        = fma_vvs(q, r, rc[RC_THIRD]);
  const double lp = fma(r * w, q, fma(w, -0.5, r));
  const double kd = (double)(k - 53); // ln u = (k-53) ln2 + ln c + log1p(r)
  const double hi = fma_vsv(kd, rc[RC_LN2_HI], e.logc);
  const double ln = hi + fma_vsv(kd, rc[RC_LN2_LO], lp);
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
__device__ __forceinline__ void sincos_turns(uint32_t ylo, uint32_t yhi, const RngConsts &K, double &sn, double &cs)
{
  const double *rc = K.c;
  const uint32_t q  = (yhi + (1u << 18)) >> 19;        // nearest quarter turn, 0..4
  const int32_t  rh = (int32_t)(yhi - (q << 19));      // remainder Y - q 2^51 in [-2^50, 2^50], high word
  const double   r  = fma_vsv((double)rh, 4294967296.0, (double)ylo) * 0x1.0p-53; // turns, |r| <= 1/8, exact
  const double   w  = r * r;
  double s = rc[RC_SIN0 + 8], c = rc[RC_COS1 + 8];
  s = fma_vvs(s, w, rc[RC_SIN0 + 7]);
  c = fma_vvs(c, w, rc[RC_COS1 + 7]);
  s = fma_vvs(s, w, rc[RC_SIN0 + 6]);
  c = fma_vvs(c, w, rc[RC_COS1 + 6]);
  s = fma_vvs(s, w, rc[RC_SIN0 + 5]);
  c = fma_vvs(c, w, rc[RC_COS1 + 5]);
  s = fma_vvs(s, w, rc[RC_SIN0 + 4]);
  c = fma_vvs(c, w, rc[RC_COS1 + 4]);
  s = fma_vvs(s, w, rc[RC_SIN0 + 3]);
  c = fma_vvs(c, w, rc[RC_COS1 + 3]);
  s = fma_vvs(s, w, rc[RC_SIN0 + 2]);
  c = fma_vvs(c, w, rc[RC_COS1 + 2]);
  s = fma_vvs(s, w, rc[RC_SIN0 + 1]);
  c = fma_vvs(c, w, rc[RC_COS1 + 1]);
  s = fma_vvs(s, w, rc[RC_SIN0 + 0]);
  c = fma_vvs(c, w, rc[RC_COS1 + 0]);
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
// DEFERRED table fill: the caller requested its lane's table entry (`entry` = g_logtab[lane], a global load that may still be
// in flight) before it issued its own streaming loads; the entry goes into the wavefront's LDS copy only here, behind the
// Philox rounds, so that no wavefront sits on a table fetch before its streaming loads are out.  EVERY lane of the
// wavefront must get here (lane i provides entry i).
__device__ __forceinline__ void normal_pair_fill(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, LogTabEntry entry, LogTabEntry *wave_tab, int lane, const RngConsts &K, double &z0, double &z1)
{
  const Philox4 p = philox4x32_10(c0, c1, c2, c3, k0, k1);
  uint32_t      xlo, xhi, ylo, yhi;
  u53_int(p.r0, p.r1, xlo, xhi);
  u53_int(p.r2, p.r3, ylo, yhi);
  double s, c;
  sincos_turns(ylo, yhi, K, s, c);
  const RngLogConsts KL = load_log_consts();
  wave_tab[lane]        = entry;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const double radius = sqrt_pos(minus2_log_u(xlo, xhi, wave_tab, KL));
  z0 = radius * c;
  z1 = radius * s;
}

__device__ __forceinline__ void normal_pair(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, const LogTabEntry *tab, double &z0, double &z1)
{
  const Philox4 p = philox4x32_10(c0, c1, c2, c3, k0, k1);
  uint32_t      xlo, xhi, ylo, yhi;
  u53_int(p.r0, p.r1, xlo, xhi);
  u53_int(p.r2, p.r3, ylo, yhi);
  const RngConsts K      = load_sincos_consts();
  const double    radius = sqrt_pos(minus2_log_u(xlo, xhi, tab, load_log_consts()));
  double          s, c;
  sincos_turns(ylo, yhi, K, s, c);
  z0 = radius * c;
  z1 = radius * s;
}

} // namespace pmg
