// Counter-based Gaussian noise for the Gibbs sweeps (device side, gfx950).
//
// Replaces VecSetRandomStandardNormal (reference src/parmgmc.c:70-116): same Box-Muller transform
// (radius = sqrt(-2 ln u1), theta = 2 pi u2, cos branch for the even entry of a pair and sin branch for the
// odd one, :99-110) but on a counter-based uniform source, Philox4x32-10 (Salmon et al., SC'11), so that a
// normal is a pure function of (seed, sweep number, global index) and a chain is identical on 1/2/4/8 GPUs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pmg {

struct Philox4 {
  uint32_t r0, r1, r2, r3;
};

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0;
    c1 = lo1;
    c2 = n2;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}

// 53-bit uniform in (0,1]: ((x >> 11) + 1) * 2^-53, x = hi:lo.  Never 0, so ln(u) is finite (the reference's
// PetscRandom may return 0 and then yields inf, src/parmgmc.c:103-106).
__device__ __forceinline__ double u53(uint32_t lo, uint32_t hi)
{
  const uint64_t x = (((uint64_t)hi << 32) | lo) >> 11;
  return (double)(x + 1) * 0x1.0p-53;
}

// One Box-Muller pair from one Philox block.  z0 = r cos(2 pi u2), z1 = r sin(2 pi u2).
__device__ __forceinline__ void normal_pair(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, double &z0, double &z1)
{
  const Philox4 p  = philox4x32_10(c0, c1, c2, c3, k0, k1);
  const double  u1 = u53(p.r0, p.r1), u2 = u53(p.r2, p.r3);
  const double  radius = sqrt(-2.0 * log(u1));
  double        s, c;
  sincospi(2.0 * u2, &s, &c); // = sin/cos(2 pi u2) without rounding 2 pi u2 first
  z0 = radius * c;
  z1 = radius * s;
}

} // namespace pmg
