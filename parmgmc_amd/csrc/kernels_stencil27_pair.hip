// Phase-fused, paired class-stencil Gibbs/SOR sweep for the coarse levels of a DMDA hierarchy (gfx950).
//
// Same operation and the same bits as st27_color_sweep_kernel (kernels_stencil27.hip): MCSORApply_SEQAIJ (reference
// src/mc_sor.c:241-296) + PrepareRHS (src/pc_mcgibbs.c:119-128) on a 27-point Galerkin operator under the 8-colour
// parity colouring, vectors in natural order.  What changes is how the work is cut:
//
//   * OUT OF PLACE.  A sweep reads the vector `y_in` and writes `y_out` (the caller swaps the two afterwards).  With
//     old and new values in different buffers there is no read-after-write hazard BETWEEN workgroups, which is what
//     forces the in-place kernel to one launch per colour.
//   * ONE LAUNCH PER z-PARITY PHASE instead of four: the planes of one z-parity are independent of each other during
//     a phase (their 18 out-of-plane neighbours have the other parity).  Inside a plane the four colours (px, py)
//     depend on each other; a workgroup resolves that for a tile of lines by itself:
//       - a thread owns the PAIR of points x0 = 2p, x1 = 2p + 1 of a line, i.e. one point of each x-parity.  The second
//         of them needs the first one's new value at x0 / x0 + 2 (forward) -- its own register and its neighbour
//         lane's, one wave shuffle;
//       - stage A sweeps the lines of the first y-parity of the tile (T + 1 of them, one per wavefront), leaves the new
//         values in LDS, and stage B, behind ONE workgroup barrier, sweeps the T lines in between, reading the new
//         values of the lines above and below from LDS;
//       - whatever a tile needs from beyond its edges it recomputes: three pairs of the 64 of a wavefront and one of
//         the T + 1 first-stage lines are redundant.  Counter-based noise and a fixed arithmetic order make the
//         recomputed values identical to the ones the owning tile stores, so only owners store.
//   * DENSE LOADS.  The 27 neighbours of the two points lie on 9 lines; each is fetched as one 16-byte load (x0, x1)
//     plus the two values beside it, consecutive lanes reading consecutive addresses.  The per-colour kernel issues 54
//     stride-2 loads for the same two points and uses half of every cache line it touches.
//
// Arithmetic order per point is unchanged (sum starts at w, neighbours subtracted in ascending natural index, absent
// ones entering with a zero table coefficient at a valid address), the file is compiled with -ffp-contract=off, so
// results are bit-identical to the per-colour kernel and to the sliced-ELL sweep of the assembled matrix.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "pmg_kernels.h"
#define PMG_RNG_LITERALS // the transform's constants as literals here: scalar loads in the middle of these kernels' sums cost more than they save (st27 phase +9 % by GRBM_GUI_ACTIVE)
#define PMG_RNG_TU st27pair
#include "pmg_rng.hpp"

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d2u __attribute__((ext_vector_type(2), aligned(8))); // natural-order lines of odd length start 8-byte aligned

constexpr int PT    = 4;  // line pairs per tile: T + 1 = 5 wavefronts per workgroup
constexpr int VALID = 60; // pairs per wavefront whose results are final (2 + 2 halo pairs)
constexpr int HL    = 2;  // halo pairs on the left

__device__ __forceinline__ int pos_class(int i, int n) { return i == 0 ? 0 : (i == n - 1 ? 2 : 1); }

// Lane i receives lane i-1's (from_prev) / lane i+1's (from_next) value: DPP full-wavefront shifts of GFX9
// (wave_shr:1 = 0x138, wave_shl:1 = 0x130), two v_mov_b32 per double, no LDS crossbar.  Lane 0 of from_prev and lane 63 of
// from_next receive 0: only the outermost halo lanes use them and nothing final depends on those lanes.
__device__ __forceinline__ double from_prev(double v)
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_next(double v)
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// XCD-aware block order.  Workgroups are dealt round-robin to the 8 XCDs (block id mod 8), each with its own L2.  With
// the plain (x, y, z) order every L2 ends up fetching more than half of every plane (a tile reads the lines around it
// in three planes, and the tiles of a plane are spread over all XCDs: 4.5 x the vector in total, rocprofv3 FETCH_SIZE).
// Here the (line tile, plane) items are numbered plane after plane and XCD x takes the x-th eighth of that list, walking
// through it in order: an L2 fetches its own planes once (plus one seam plane per neighbour), the planes k-1 / k+1 of one
// tile are the L2-resident neighbours of the next, and the shares differ by at most one tile (bands of line tiles of a
// 2^k+1 grid cannot be balanced: 33 tiles over 8 XCDs leave one XCD with 5 and the others with 4).
// Placement only: no result depends on it.  Returns false for the padding blocks.
__device__ __forceinline__ bool xcd_block(int nbx, int nby, int nbz, int &bx, int &by, int &bz, int bid = (int)blockIdx.x)
{
  const int     xcd = bid & 7, q = bid >> 3; // q: index inside this XCD's share
  const int64_t T   = (int64_t)nby * nbz;
  const int     lo = (int)(xcd * T / 8), hi = (int)((xcd + 1) * T / 8);
  bx               = q % nbx;
  const int t      = lo + q / nbx;
  by               = t % nby;
  bz               = t / nby;
  return t < hi;
}
static inline unsigned xcd_grid(int nbx, int nby, int nbz)
{
  const int64_t T = (int64_t)nby * nbz;
  return (unsigned)(8 * nbx * ((T + 7) / 8));
}

struct pair_ctx {
  int      nx, ny, nzg;
  int      x0, j, k;
  int      xu; // 2 p without the clamp of the halo lanes outside the domain: the noise counter of a halo pair must be the
               // one of ITS position, because its sin branch is handed to the next lane
  bool     act0, act1;
  int64_t  line; // natural index of x = 0 on this line: nx (j + ny k)
  uint32_t key0, key1;
  uint64_t sweep;
  double   om1;
  const double *gcoef;
};

// How the pairs of a line are dealt to wavefronts (round 4).  A wavefront holds `valid` pairs of one line plus `halo` lanes
// on either side that only supply their neighbours; a line of 2^k+1 points has 2^(k-1)+1 pairs -- ONE more than a power of
// two -- so its last x segment is almost empty (129 pairs = 60 + 60 + 9, 65 = 60 + 5) and cost a whole wavefront per line:
// a third of the 257^3 level's wavefronts, half of the 129^3 level's (tools/st27bench.py with that segment dropped:
// residual 174 -> 129 us, noisy sweep 233 -> 178 us at 257^3; 24 -> 15 and 49 -> 32 us at 129^3).  The REMAINDER segment
// is therefore swept by a second launch whose wavefronts hold the remainders of `nseg` different lines side by side, each
// with its own halo lanes (segment width segw = rem + 2 halo); everything that was wave-uniform per line is per lane
// there.  Same arithmetic per point: same bits.
struct pair_split {
  int nbx_main; // x segments of the one-line-per-wavefront launch
  int p0;       // first pair of the remainder segment (packed launch), -1: none
  int segw, nseg;
};
static inline pair_split split_pairs(int npairs, int valid, int halo)
{
  pair_split P;
  const int  nbx = (npairs + valid - 1) / valid, rem = npairs - valid * (nbx - 1), segw = rem + 2 * halo;
  static int env = -1;
  if (env < 0) {
    const char *e = getenv("PMG_ST27_PACK_REMAINDER");
    env           = e ? atoi(e) : 1;
  }
  if (env && 64 / segw >= 2) { // two or more remainders fit one wavefront
    P.nbx_main = nbx - 1;
    P.p0       = valid * (nbx - 1);
    P.segw     = segw;
    P.nseg     = 64 / segw;
  } else {
    P.nbx_main = nbx;
    P.p0       = -1;
    P.segw = P.nseg = 0;
  }
  return P;
}

// New values of the two points of a pair.  FIRST1: the point x1 is swept before x0 (backward order).
// get_row(row, r) delivers row `row` = 3 (dz+1) + (dy+1) of the neighbourhood as r = y at x0-1, x0, x1, x1+1, holding the
// values the FIRST point sees.  The rows are consumed as a stream, in the order of the sum: the first point takes all
// of them; the second point's terms of the rows before its own line (0..3) are accumulated in the same pass, its own
// line (row 4) needs the first colour's new values on both sides -- own register and the neighbouring lane's, one
// wave shuffle -- and the rows behind it (5..8) are fetched again at that point (round 3; round 2 kept their 12 values in
// registers: 78 VGPRs).  Keeping all 36 values of the neighbourhood costs 130 VGPRs and two fifths of the occupancy.
template <bool NOISY, bool FIRST1, class RowFn, class ZeroFn>
__device__ __forceinline__ d2 sweep_pair(const pair_ctx &C, RowFn &&get_row, ZeroFn &&row_is_zero, const double *s_coef, const double *s_idiag, const double *s_sqrtd, const pmg::LogTabEntry *s_logtab, d2 bb)
{
  constexpr int F = FIRST1 ? 1 : 0, S = 1 - F;
  const int     cyz   = 3 * pos_class(C.j, C.ny) + 9 * pos_class(C.k, C.nzg);
  const int     cls0  = pos_class(C.x0, C.nx) + cyz, cls1 = pos_class(C.x0 + 1, C.nx) + cyz;
  const int     clsF  = F ? cls1 : cls0, clsS = S ? cls1 : cls0;
  double        w0 = bb.x, w1 = bb.y;
  if (NOISY) {
    // noise = row stream of the natural index r: counter r >> 1, branch r & 1.  x0 is even, so whether the pair shares
    // one Box-Muller draw depends on the line alone (wave-uniform)
    const int64_t r0 = C.line + C.xu;
    double        z0, z1;
    if (__builtin_amdgcn_readfirstlane((int)(C.line & 1)) == 0) {
      pmg::normal_pair((uint32_t)((uint64_t)r0 >> 1), 0u, (uint32_t)C.sweep, (uint32_t)(C.sweep >> 32), C.key0, C.key1, s_logtab, z0, z1);
    } else {
      // r0 is odd: the draw of counter (r0 + 1) >> 1 serves x1 (cos branch) and the NEXT pair's x0 (sin branch), so
      // every thread still makes one draw and receives its x0 value from the lane before it
      double zs;
      pmg::normal_pair((uint32_t)((uint64_t)(r0 + 1) >> 1), 0u, (uint32_t)C.sweep, (uint32_t)(C.sweep >> 32), C.key0, C.key1, s_logtab, z1, zs);
      z0 = from_prev(zs);
    }
    w0 = z0 * s_sqrtd[cls0] + w0;
    w1 = z1 * s_sqrtd[cls1] + w1;
  }
#ifdef PMG_ST27_PROBE_INTERIOR
  // TIMING PROBE ONLY (wrong at the boundaries): every point takes the interior class, coefficients from scalar loads
  typedef const double __attribute__((address_space(4))) *cptr;
  const cptr cfF = (cptr)(unsigned long long)C.gcoef + 27 * 13, cfS = cfF;
#else
  const double *cfF = s_coef + 27 * clsF, *cfS = s_coef + 27 * clsS;
#endif
  double        accF = F ? w1 : w0, accS = S ? w1 : w0;
  double        oldF = 0.0, oldS = 0.0;
#pragma unroll
  for (int row = 0; row < 9; ++row) {
    // a row known to hold zeros (first sweep from a zero guess): its terms are skipped altogether -- they would subtract
    // +-0 -- and nothing is loaded; after unrolling the test is a compile-time constant
    if (row_is_zero(row)) continue;
    double r[4];
    get_row(row, r);
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int e = 3 * row + dx + 1;
      if (e != 13) accF = accF - cfF[e] * r[1 + F + dx];
    }
    if (row < 4) {
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) accS = accS - cfS[3 * row + dx + 1] * r[1 + S + dx];
    } else if (row == 4) {
      oldF = r[1 + F];
      oldS = r[1 + S];
    }
  }
  double nF = C.om1 * oldF + s_idiag[clsF] * accF;
  nF        = (F ? C.act1 : C.act0) ? nF : 0.0;
  // the second point's neighbours on its own line: FIRST1 = false: x0 (own) and the next pair's x0; true: the previous
  // pair's x1 and x1 (own)
  const double nb = FIRST1 ? from_prev(nF) : from_next(nF);
  accS            = accS - cfS[12] * (FIRST1 ? nb : nF);
  accS            = accS - cfS[14] * (FIRST1 ? nF : nb);
#pragma unroll
  for (int row = 5; row < 9; ++row) {
    if (row_is_zero(row)) continue;
    double r[4]; // fetched a SECOND time (L1 hit in stage A, register / LDS copy in stage B) instead of twelve values carried
    get_row(row, r); // over the first point's sum: 78 -> 72 VGPRs, 6 -> 7 waves per SIMD, 257^3 level sweep -3 %
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) accS = accS - cfS[3 * row + dx + 1] * r[S + dx + 1];
  }
  double nS = C.om1 * oldS + s_idiag[clsS] * accS;
  nS        = (S ? C.act1 : C.act0) ? nS : 0.0;
  const d2 out = {F ? nS : nF, F ? nF : nS};
  return out;
}

// one row of the neighbourhood from a vector in natural order behind one ghost plane; an absent line / plane is
// replaced by the centre line (finite values that meet a zero coefficient).  kk: plane index inside the slab (global
// plane - kz0; -1 and nz are the ghost planes of a z-slab)
__device__ __forceinline__ void load_row(double (&r)[4], const double *vec, int nx, int ny, int xc0, int jj, int kk)
{
  const double *row = vec + (int64_t)nx * (jj + (int64_t)ny * (kk + 1));
  const d2u     v   = *reinterpret_cast<const d2u *>(row + xc0);
  r[0]              = from_prev(v.y); // y at x0 - 1 = the previous pair's x1: one 16-byte load per row and lane, the two
  r[1]              = v.x;            // values beside the pair come from the neighbouring lanes
  r[2]              = v.y;
  r[3]              = from_next(v.x);
}

// `kfirst`, `kfirst + 2`, ...: the planes of this phase.  y_other: the vector that holds the CURRENT values of the
// planes of the other z-parity (y_in in the first phase of a sweep, y_out in the second).
#ifndef PMG_ST27_PAIR_WAVES
#define PMG_ST27_PAIR_WAVES 5
#endif
// ZIN: y_in is all zeros and is not read (the pre-smoothing sweep of a V-cycle level starts from a zero guess: no memset of
// the level's iterate, no loads of zeros); ZOTHER: the planes of the other z-parity are zero too (first phase of such a sweep)
// PACK: the remainder x segment (see split_pairs below): the wavefront w of a workgroup sweeps its line of `nseg` different
// line tiles side by side -- lane = segment * segw + slot, pair p0 - HL + slot, line tile nseg * by + segment -- so that what
// depends on the line is per lane; the lanes of a segment whose line tile lies beyond the plane work on a valid line and
// store nothing (no divergent exit: neighbouring lanes exchange values by DPP).  Line tiles are 2 PT = 8 lines apart: the
// parity of a line, which decides how the Box-Muller pairs are shared, is the same in all segments of a wavefront.
template <bool NOISY, bool BACKWARD, bool ZIN, bool ZOTHER, bool PACK>
__device__ __forceinline__ void st27_pair_phase_body(const pmgk_st27 &S, int bid, int nbx, int nby, int nbz, int kfirst, double one_minus_omega, uint32_t key0, uint32_t key1, uint64_t sweep, int p0, int segw, int nseg, const double *__restrict__ b, const double *y_in, double *y_out, const double *y_other, double *s_coef, double *s_idiag, double *s_sqrtd, pmg::LogTabEntry *s_logtab, d2 (*s_new)[64])
{
  const int                   lane = threadIdx.x, w = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int                   tid  = lane + 64 * w;
  for (int q = tid; q < 27 * 27; q += 64 * (PT + 1)) s_coef[q] = S.coef[q]; // (padding blocks of the last band leave below, before the barrier)
  if (tid < 27) {
    s_idiag[tid] = S.idiag[tid];
    s_sqrtd[tid] = S.sqrtdiag[tid];
  }
  int bx, by, bz;
  if (!xcd_block(nbx, nby, nbz, bx, by, bz, bid)) return; // whole workgroup: before any barrier
  if (NOISY) pmg::load_log_table(s_logtab);
  __syncthreads();

  const int nx = S.nx, ny = S.ny;
  const int npairs = (nx + 1) / 2;
  int  p, byl = by;
  bool seg_ok = true, final_lane;
  if (PACK) {
    const int sidx = lane / segw, slot = lane - sidx * segw;
    p          = p0 - HL + slot;
    byl        = by * nseg + sidx;
    seg_ok     = sidx < nseg;
    final_lane = seg_ok && slot >= HL && slot < segw - HL;
  } else {
    p          = VALID * bx - HL + lane;
    final_lane = lane >= HL && lane < HL + VALID; // lanes whose results are complete at the end of each stage
  }
  const int pc     = min(max(p, 0), npairs - 1); // inactive lanes work on a valid pair and publish zeros
  pair_ctx  C;
  C.nx   = nx;
  C.ny   = ny;
  C.nzg  = S.nzg;
  C.x0   = 2 * pc;
  C.xu   = 2 * p;
  C.k    = kfirst + 2 * bz;
  const bool pact0 = p >= 0 && p < npairs && seg_ok, pact1 = pact0 && C.x0 + 1 < nx;
  C.act0 = pact0;
  C.act1 = pact1;
  C.key0 = key0;
  C.key1 = key1;
  C.sweep = sweep;
  C.om1   = one_minus_omega;
  C.gcoef = S.coef;
  const int  xc0 = C.x0;
  const int  jt = 2 * PT * byl;
  const bool hasD = C.k > 0, hasU = C.k < S.nzg - 1;
  const int  kl = C.k - S.kz0; // plane inside the slab: addresses; C.k (global) keys the noise and the boundary classes

  // ---- stage A: the lines of the y-parity that is swept first --------------------------------------------------------
  {
    const int  jAu = BACKWARD ? jt - 1 + 2 * w : jt + 2 * w;
    const bool okA = jAu >= 0 && jAu < ny;
    const int  jA  = PACK ? min(max(jAu, 0), ny - 1) : jAu;
    d2         nw  = {0.0, 0.0};
    if (PACK) { // per lane: a lane without a line sweeps a valid one as an inactive lane (zeros, no store)
      C.act0 = pact0 && okA;
      C.act1 = pact1 && okA;
    }
    if (PACK || okA) {
      C.j    = jA;
      // PACK: from the UNCLAMPED line -- its parity (the wave-uniform branch of the noise in sweep_pair) is that of the valid
      // segments' lines, which a clamped line's need not be (backward, first tile: line -1 clamps to line 0)
      C.line = (int64_t)nx * ((PACK ? jAu : jA) + (int64_t)ny * C.k);
      const int jS = jA > 0 ? jA - 1 : jA, jN = jA < ny - 1 ? jA + 1 : jA;
      const int64_t lrow = (int64_t)nx * (jA + (int64_t)ny * (kl + 1)) + xc0;
      const d2u     bv   = *reinterpret_cast<const d2u *>(b + lrow);
      const d2      bb   = {bv.x, bv.y};
      auto          rows = [&](int row, double (&r)[4]) {
        const int dz = row / 3 - 1, dy = row % 3 - 1;
        const bool okz = dz < 0 ? hasD : (dz > 0 ? hasU : true);
        load_row(r, (dz != 0 && (okz || ZIN)) ? y_other : y_in, nx, ny, xc0, dy < 0 ? jS : (dy > 0 ? jN : jA), okz ? kl + dz : kl); // ZIN: y_in does not exist; an absent plane is replaced by y_other's centre plane (finite, zero coefficients)
      };
      // stage A reads only old values: in-plane rows from y_in, the planes above / below from y_other
      auto zero = [&](int row) { return (row / 3 == 1) ? ZIN : ZOTHER; };
      nw = sweep_pair<NOISY, BACKWARD>(C, rows, zero, s_coef, s_idiag, s_sqrtd, s_logtab, bb);
      const bool owned   = BACKWARD ? w >= 1 : w < PT; // the other first-stage line belongs to the neighbouring tile
      if (owned && final_lane && C.act0) {
        if (C.act1) *reinterpret_cast<d2u *>(y_out + lrow) = d2u{nw.x, nw.y};
        else y_out[lrow] = nw.x;
      }
    }
    s_new[w][lane] = nw;
  }
  // stage B's seven rows that come from memory (its own line and the six lines of the planes above / below) are requested
  // BEFORE the barrier: their latency runs under the wait for stage A's slowest wavefront (257^3 phase 135 -> 131 us)
  d2u pre[9];
  {
    const int jBu = BACKWARD ? jt + 2 * w : jt + 2 * w + 1, jB = PACK ? min(jBu, ny - 1) : jBu;
    if (w < PT && (PACK || jB < ny)) {
      const int jS = jB > 0 ? jB - 1 : jB, jN = jB < ny - 1 ? jB + 1 : jB;
#pragma unroll
      for (int row = 0; row < 9; ++row) {
        const int dz = row / 3 - 1, dy = row % 3 - 1;
        if (dz == 0 && dy != 0) continue;
        const bool zero = (row / 3 == 1) ? (row == 4 && ZIN) : ZOTHER;
        if (zero) continue;
        const bool    okz = dz < 0 ? hasD : (dz > 0 ? hasU : true);
        const double *vec = (dz != 0 && (okz || ZIN)) ? y_other : y_in;
        const int     jj = dy < 0 ? jS : (dy > 0 ? jN : jB), kk = okz ? kl + dz : kl;
        pre[row]         = *reinterpret_cast<const d2u *>(vec + (int64_t)nx * (jj + (int64_t)ny * (kk + 1)) + xc0);
      }
    }
  }
  __syncthreads();
  // ---- stage B: the lines in between, whose in-plane neighbours above and below are the new values in LDS ------------
  if (w < PT) {
    const int jBu = BACKWARD ? jt + 2 * w : jt + 2 * w + 1, jB = PACK ? min(jBu, ny - 1) : jBu;
    if (PACK) {
      C.act0 = pact0 && jBu < ny;
      C.act1 = pact1 && jBu < ny;
    }
    if (PACK || jB < ny) {
      C.j    = jB;
      C.line = (int64_t)nx * ((PACK ? jBu : jB) + (int64_t)ny * C.k);
      const int64_t lrow = (int64_t)nx * (jB + (int64_t)ny * (kl + 1)) + xc0;
      const d2u     bv   = *reinterpret_cast<const d2u *>(b + lrow);
      const d2      bb   = {bv.x, bv.y};
      auto          rows = [&](int row, double (&r)[4]) {
        const int dz = row / 3 - 1, dy = row % 3 - 1;
        if (dz == 0 && dy != 0) { // the lines above / below in this plane: new values of stage A
          const int q = dy < 0 ? w : w + 1;
          const d2  c = s_new[q][lane];
          r[0]        = from_prev(c.y);
          r[1]        = c.x;
          r[2]        = c.y;
          r[3]        = from_next(c.x);
          return;
        }
        const d2u v = pre[row]; // requested in front of the barrier
        r[0]        = from_prev(v.y);
        r[1]        = v.x;
        r[2]        = v.y;
        r[3]        = from_next(v.x);
      };
      // stage B: the own line is old (y_in), the lines above / below in this plane are stage A's new values (LDS)
      auto zero = [&](int row) { return (row / 3 == 1) ? (row == 4 && ZIN) : ZOTHER; };
      const d2 nw = sweep_pair<NOISY, BACKWARD>(C, rows, zero, s_coef, s_idiag, s_sqrtd, s_logtab, bb);
      if (final_lane && C.act0) {
        if (C.act1) *reinterpret_cast<d2u *>(y_out + lrow) = d2u{nw.x, nw.y};
        else y_out[lrow] = nw.x;
      }
    }
  }
}

// The full x segments (PACK = false) and the packed remainder segments (PACK = true) are two launches: as ONE kernel with a
// workgroup-uniform branch between the two bodies the register allocation no longer fits 72 VGPRs (84, the noise-free
// instantiations 96 with spills) and costs a wavefront per SIMD everywhere.
template <bool NOISY, bool BACKWARD, bool ZIN, bool ZOTHER, bool PACK>
__global__ __launch_bounds__(64 * (PT + 1)) __attribute__((amdgpu_waves_per_eu(PMG_ST27_PAIR_WAVES, 8))) void st27_pair_phase_kernel(pmgk_st27 S, int nbx, int nby, int nbz, int kfirst, double one_minus_omega, uint32_t key0, uint32_t key1, uint64_t sweep, int p0, int segw, int nseg, const double *__restrict__ b, const double *y_in, double *y_out, const double *y_other)
{
  __shared__ double           s_coef[27 * 27], s_idiag[27], s_sqrtd[27];
  __shared__ pmg::LogTabEntry s_logtab[NOISY ? PMG_LOGTAB_SIZE : 1];
  __shared__ d2               s_new[PT + 1][64];
  st27_pair_phase_body<NOISY, BACKWARD, ZIN, ZOTHER, PACK>(S, (int)blockIdx.x, nbx, nby, nbz, kfirst, one_minus_omega, key0, key1, sweep, p0, segw, nseg, b, y_in, y_out, y_other, s_coef, s_idiag, s_sqrtd, s_logtab, s_new);
}

// r = b - A y with the pair mapping (dense loads); the diagonal term is added last, like st27_residual_kernel.
// PACK: the remainder segments of nseg lines per wavefront (pair p0 + slot - 1 of line (4 by + wy) nseg + lane / segw)
template <bool PACK>
__global__ __launch_bounds__(256) void st27_pair_residual_kernel(pmgk_st27 S, int nbx, int nby, int nbz, int p0, int segw, int nseg, const double *__restrict__ b, const double *__restrict__ y, double *__restrict__ r)
{
  __shared__ double s_coef[27 * 27];
  const int         tid = threadIdx.x + 64 * threadIdx.y;
  int bx, by, bz;
  if (!xcd_block(nbx, nby, nbz, bx, by, bz)) return;
  for (int q = tid; q < 27 * 27; q += 256) s_coef[q] = S.coef[q];
  __syncthreads();
  const int nx = S.nx, ny = S.ny;
  // lanes 0 and 63 (PACK: the first and last lane of a segment) only supply their neighbours with the values beside the
  // pairs: 62 pairs per wavefront
  const int npairs = (nx + 1) / 2;
  const int lane = (int)threadIdx.x, wy = __builtin_amdgcn_readfirstlane(threadIdx.y), k = S.kz0 + bz; // global plane
  int       p, j;
  bool      inner;
  if (PACK) {
    const int sidx = lane / segw, slot = lane - sidx * segw;
    p     = p0 - 1 + slot;
    j     = (4 * by + wy) * nseg + sidx;
    inner = sidx < nseg && slot >= 1 && slot <= segw - 2 && j < ny;
    j     = min(j, ny - 1); // a lane without a line works on a valid one and stores nothing (its neighbours use DPP)
  } else {
    p     = 62 * bx - 1 + lane;
    j     = 4 * by + wy;
    inner = lane >= 1 && lane <= 62;
    if (j >= ny) return; // wave-uniform
  }
  const int  x0   = 2 * min(max(p, 0), npairs - 1);
  const bool act0 = p >= 0 && p < npairs && inner, act1 = act0 && x0 + 1 < nx;
  const int  jS = j > 0 ? j - 1 : j, jN = j < ny - 1 ? j + 1 : j, kD = k > 0 ? k - 1 : k, kU = k < S.nzg - 1 ? k + 1 : k;
  const int     cyz  = 3 * pos_class(j, ny) + 9 * pos_class(k, S.nzg);
  const double *cf0 = s_coef + 27 * (pos_class(x0, nx) + cyz), *cf1 = s_coef + 27 * (pos_class(x0 + 1, nx) + cyz);
  // the per-colour kernel ADDS the products one by one starting from 0 and the diagonal term last: same order here
  double s0 = 0.0, s1 = 0.0, c0 = 0.0, c1 = 0.0;
#pragma unroll
  for (int row = 0; row < 9; ++row) {
    const int dz = row / 3 - 1, dy = row % 3 - 1;
    double    r4[4];
    load_row(r4, y, nx, ny, x0, dy < 0 ? jS : (dy > 0 ? jN : j), (dz < 0 ? kD : (dz > 0 ? kU : k)) - S.kz0);
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int e = 3 * row + dx + 1;
      if (e == 13) continue;
      s0 = s0 + cf0[e] * r4[1 + dx];
      s1 = s1 + cf1[e] * r4[2 + dx];
    }
    if (row == 4) {
      c0 = r4[1];
      c1 = r4[2];
    }
  }
  s0 = s0 + cf0[13] * c0;
  s1 = s1 + cf1[13] * c1;
  const int64_t lrow = (int64_t)nx * (j + (int64_t)ny * (k - S.kz0 + 1)) + x0;
  const d2u     bv   = *reinterpret_cast<const d2u *>(b + lrow);
  if (act1) *reinterpret_cast<d2u *>(r + lrow) = d2u{bv.x - s0, bv.y - s1};
  else if (act0) r[lrow] = bv.x - s0;
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? 0 : 1; }

template <bool NOISY, bool BACKWARD, bool ZIN = false, bool ZOTHER = false>
void launch_phase(const pmgk_st27 &S, int pz, double om1, uint64_t seed, uint64_t sweep, const double *b, const double *y_in, double *y_out, const double *y_other, hipStream_t s)
{
  const int kfirst = S.kz0 + ((pz - S.kz0) & 1), cz = (S.kz0 + S.nz - kfirst + 1) / 2; // this slab's planes of the (global) parity pz
  if (cz <= 0) return;
  const int        npairs = (S.nx + 1) / 2;
  const pair_split P      = split_pairs(npairs, VALID, HL);
  const int        nby    = (S.ny + 2 * PT - 1) / (2 * PT);
  const dim3       block(64, PT + 1);
  // the two launches of a phase are independent of each other (disjoint pairs of the same lines; what one reads of the
  // other's pairs are OLD values from y_in / y_other, or first-colour values it recomputes in its halo lanes)
  if (P.nbx_main > 0) hipLaunchKernelGGL((st27_pair_phase_kernel<NOISY, BACKWARD, ZIN, ZOTHER, false>), dim3(xcd_grid(P.nbx_main, nby, cz)), block, 0, s, S, P.nbx_main, nby, cz, kfirst, om1, (uint32_t)seed, (uint32_t)(seed >> 32), sweep, 0, 0, 0, b, y_in, y_out, y_other);
  if (P.p0 >= 0) {
    const int nbyp = (nby + P.nseg - 1) / P.nseg;
    hipLaunchKernelGGL((st27_pair_phase_kernel<NOISY, BACKWARD, ZIN, ZOTHER, true>), dim3(xcd_grid(1, nbyp, cz)), block, 0, s, S, 1, nbyp, cz, kfirst, om1, (uint32_t)seed, (uint32_t)(seed >> 32), sweep, P.p0, P.segw, P.nseg, b, y_in, y_out, y_other);
  }
}

} // namespace

// One z-parity phase of a directional sweep OUT OF PLACE (phase 0 reads the other parity's planes from y_in, phase 1 from
// y_out, where phase 0 and -- on a z-slab -- the halo exchange behind it have put the new values); y_in is left
// untouched, the two vectors must not overlap.  y_in == NULL: the sweep starts from a ZERO vector, which is neither stored
// nor read (omega-independent: the old value enters as (1 - omega) * 0).
extern "C" int pmgk_st27_sweep_pp_phase(const pmgk_st27 *S, int backward, int phase, double omega, int noisy, uint64_t seed, uint64_t sweep, const double *b, const double *y_in, double *y_out, void *stream)
{
  if (y_in == y_out || phase < 0 || phase > 1) return 1;
  const double  om1 = 1. - omega;
  hipStream_t   s   = (hipStream_t)stream;
  const bool    zin = y_in == nullptr;
  const int     pz    = backward ? 1 - phase : phase;
  const double *other = phase == 0 ? y_in : y_out;
#define PMG_PHASE(N, B) \
  do { \
    if (!zin) launch_phase<N, B, false, false>(*S, pz, om1, seed, sweep, b, y_in, y_out, other, s); \
    else if (phase == 0) launch_phase<N, B, true, true>(*S, pz, om1, seed, sweep, b, y_out, y_out, y_out, s); /* pointers unused: any valid address */ \
    else launch_phase<N, B, true, false>(*S, pz, om1, seed, sweep, b, y_out, y_out, other, s); \
  } while (0)
  if (noisy) {
    if (backward) PMG_PHASE(true, true);
    else PMG_PHASE(true, false);
  } else {
    if (backward) PMG_PHASE(false, true);
    else PMG_PHASE(false, false);
  }
#undef PMG_PHASE
  return launch_status();
}

// One directional sweep OUT OF PLACE on a level that lives on one device (all planes owned: kz0 = 0, nz = nzg): both phases
extern "C" int pmgk_st27_sweep_pp(const pmgk_st27 *S, int backward, double omega, int noisy, uint64_t seed, uint64_t sweep, const double *b, const double *y_in, double *y_out, void *stream)
{
  if (S->kz0 != 0 || S->nz != S->nzg) return 1;
  for (int phase = 0; phase < 2; ++phase)
    if (pmgk_st27_sweep_pp_phase(S, backward, phase, omega, noisy, seed, sweep, b, y_in, y_out, stream)) return 1;
  return 0;
}

extern "C" int pmgk_st27_residual_pair(const pmgk_st27 *S, const double *b, const double *y, double *r, void *stream)
{
  if (S->nz <= 0) return 0; /* z-slabs too: the slab's planes, neighbours in the ghost planes */
  const int        npairs = (S->nx + 1) / 2;
  const pair_split P      = split_pairs(npairs, 62, 1);
  const int        nby    = (S->ny + 3) / 4;
  if (P.nbx_main > 0) hipLaunchKernelGGL((st27_pair_residual_kernel<false>), dim3(xcd_grid(P.nbx_main, nby, S->nz)), dim3(64, 4), 0, (hipStream_t)stream, *S, P.nbx_main, nby, S->nz, 0, 0, 0, b, y, r);
  if (P.p0 >= 0) { /* the remainder segments, nseg lines per wavefront */
    const int nbyp = (S->ny + 4 * P.nseg - 1) / (4 * P.nseg);
    hipLaunchKernelGGL((st27_pair_residual_kernel<true>), dim3(xcd_grid(1, nbyp, S->nz)), dim3(64, 4), 0, (hipStream_t)stream, *S, 1, nbyp, S->nz, P.p0, P.segw, P.nseg, b, y, r);
  }
  return launch_status();
}
