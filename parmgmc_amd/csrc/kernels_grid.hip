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
#include <cstring>
#include <stdlib.h>
#include "pmg_kernels.h"
#define PMG_RNG_TU grid
#include "pmg_rng.hpp"

namespace {

// first-class 16-byte vector: one global_load/store_dwordx4 that the optimiser cannot split per component
typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ d2 ld2(const double *p) { return *reinterpret_cast<const d2 *>(p); }

// 16-byte access to PEER-SHARED memory (the halo receive blocks) at system scope: two relaxed 64-bit atomics, which the
// compiler emits as write-through / cache-bypassing accesses (sc0 sc1) -- the data is performed at the system's point
// of coherence when the wavefront's s_waitcnt returns, which ordinary stores do not promise
__device__ __forceinline__ d2 ld2_sys(const double *p)
{
  const unsigned long long a = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned long long b = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  d2                       v;
  v.x = __longlong_as_double((long long)a);
  v.y = __longlong_as_double((long long)b);
  return v;
}
__device__ __forceinline__ void st2_sys(double *p, d2 v)
{
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p) + 1, (unsigned long long)__double_as_longlong(v.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// "uniform base pointer + unsigned 32-bit BYTE offset": the form the backend turns into one global access with a scalar
// base (saddr) and a 32-bit lane offset, without 64-bit address arithmetic on the vector unit
__device__ __forceinline__ const double *at_bytes(const double *base, uint32_t byte_off) { return reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + byte_off); }
__device__ __forceinline__ double       *at_bytes(double *base, uint32_t byte_off) { return reinterpret_cast<double *>(reinterpret_cast<char *>(base) + byte_off); }

// wave-uniform double -> SGPR pair
__device__ __forceinline__ double uniform(double v)
{
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// Sweep of one colour.
//
// Block = 64 lanes along x (one wavefront = 128 consecutive points of this colour on one grid line = 256 grid
// columns) x 4 grid lines of ONE plane; each thread owns two consecutive points (m = 2t, 2t+1: one 16-byte
// load/store per array).  Measured on MI355X (tools/streambench.hip): a one-tile-per-block grid dispatched in
// memory order streams at 6.1 TB/s for this read-2/write-1 mix, a grid-stride/marching loop only 4.4-5.1 TB/s,
// so blocks are kept short and the dispatch order is what creates locality:
//   * blockIdx % 8 selects the XCD under round-robin dispatch (speed heuristic only, nothing depends on it);
//     XCD x gets the band of grid lines [x*band, (x+1)*band) of EVERY plane, in plane order, so the 6 reads a
//     point makes of the other colour (same line, lines j-1/j+1, planes k-1/k+1) hit lines that the SAME XCD's
//     L2 fetched a few hundred blocks earlier, and all 8 XCDs advance through the planes together, which keeps
//     the set of open DRAM pages compact.
// Boundary handling is branch-free: an absent neighbour is read from a clamped (valid, finite) address and
// enters with coefficient 0 instead of h2, which adds an exact +0.
// HALO (multi-GPU face planes, transport "ipc"): the other colour's ghost plane below / above the slab is read from
// `halo.glo` / `halo.ghi` (one plane each, this rank's receive block) instead of the vector's own ghost planes, and
// the results of plane 0 / nz-1 are ALSO stored into `halo.plo` / `halo.phi`, which point into the z-neighbours'
// receive blocks (peer memory over xGMI): the halo exchange costs no extra launch and no copy.
// entry nyz+1 / nyz+2 of an 8-entry operator table: wave-uniform (one line per wave) -> SGPR pair; PACKED (lanes of a
// wave may sit on different lines) -> per-lane select chain
template <bool PACKED>
__device__ __forceinline__ double table_at(const double (&a)[8], int idx)
{
  if (!PACKED) return uniform(a[idx]);
  double v = a[1];
  v        = idx == 2 ? a[2] : v;
  v        = idx == 3 ? a[3] : v;
  v        = idx == 4 ? a[4] : v;
  v        = idx == 5 ? a[5] : v;
  v        = idx == 6 ? a[6] : v;
  return v;
}

// PACKED: the threads of a plane are numbered line after line with tplE = ceil(ceil(nx/2)/2) threads per line and
// dealt to the wavefronts without gaps.  With one line per wavefront (the default) a line of a 2^k+1 grid -- the
// multigrid sizes -- needs one wavefront more than its power-of-two neighbour and leaves it almost empty (257: 65
// threads in 128 lanes); packed, every wavefront is full.  Same arithmetic per point, so the results do not change.
// one wavefront waits (lane 0 polls, the others follow) until *f >= v; gives up after ~90 s (2^28 polls: four ranks SHARING one GPU have been seen to keep a neighbour's kernel off the device for a minute; the shorter limit of round 1 turned that into an error).  The flag words and the
// planes they announce live in FINE-GRAINED memory (never cached in L2), so relaxed polls and a plain ordering fence
// are enough -- a system-scope acquire would invalidate the L2 under the interior sweep on every poll.
__device__ __forceinline__ void wave_wait_flag(const uint64_t *f, uint64_t v, unsigned *err, unsigned long long *spin_total)
{
  if (threadIdx.x == 0) {
    unsigned long long spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < v) {
      __builtin_amdgcn_s_sleep(8);
      ++spins;
      if ((spins & 0xFFFFu) == 0 && err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break; // somebody already gave up: do not stack timeouts
      if (spins > (1ull << 28)) {
        if (err) { // what was waited for: [1] = wait site, [2] = expected, [3] = seen (low words)
          err[1] = 0x10u;
          err[2] = (unsigned)v;
          err[3] = (unsigned)__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        break;
      }
    }
    if (spins && spin_total) atomicAdd(spin_total, spins); /* only a wavefront that actually waited pays for the report */
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <bool NOISY, bool OMEGA1, bool HALO, bool PACKED>
__device__ __forceinline__ void grid_color_sweep_body(const pmgk_grid_layout &L, const pmgk_grid_op &op, int c, int t, int j, int k, pmg::LogTabEntry *tab, pmg::LogTabEntry tab_entry, const pmgk_grid_halo &halo, const double *__restrict__ b_own, const double *__restrict__ y_other, double *__restrict__ y_own)
{
  const int kg = k + L.kz0;
  // NOISY: lane i of the wavefront carries entry i of the log table (requested by the caller, still in flight) into the
  // wavefront's LDS copy inside normal_pair_fill, so every lane must stay until then: a lane without a point moves to
  // (t, j) = (0, 0) -- valid addresses -- and only its stores are suppressed
  bool live = j < L.ny && 2 * t < L.sx && 4 * t + ((c + j + kg) & 1) < L.nx;
  if (NOISY) {
    if (!__builtin_amdgcn_ballot_w64(live)) return;
    t = live ? t : 0;
    if (PACKED) j = live ? j : 0;
  } else if (!live)
    return;
  const int p  = (c + j + kg) & 1;
  const int i0 = 4 * t + p, i1 = i0 + 2; // grid columns of the two points
  const bool v1 = i1 < L.nx;

  const bool    hasS = j > 0, hasN = j < L.ny - 1, hasD = kg > 0, hasU = kg < L.nzg - 1;
  const double  h2 = op.h2;
  // element (plane k, line j, m = 2t).  The row part of the address is wave-uniform (one line per wavefront unless
  // PACKED): row pointers live in SGPRs and every access is "scalar base + unsigned 32-bit lane offset" -- no 64-bit
  // address arithmetic on the vector unit, which this kernel is bound by
  const int64_t  rowoff = (int64_t)(k + 1) * L.sp + (int64_t)j * L.sx;
  const uint32_t lo     = 16u * (uint32_t)t; // byte offset of m = 2t inside the row (a row is at most 2^28 points long)
  const double  *yo_row = y_other + rowoff;
  // p=0: left(0)=m' 2t-1 (ed), right(0)=2t,   left(1)=2t,   right(1)=2t+1
  // p=1: left(0)=m' 2t,        right(0)=2t+1, left(1)=2t+1, right(1)=2t+2 (ed)
  const int    eo = p ? (2 * t + 2 < L.sx ? 2 : 1) : (t > 0 ? -1 : 0); // clamped lane-neighbour offset
  const bool   hasW0 = i0 > 0, hasE0 = i0 < L.nx - 1, hasE1 = i1 < L.nx - 1;
  // diagonal-dependent constants: the point has nyz in-domain y/z neighbours (wave-uniform) plus 1 or 2 in x
  // min(distance to the face, 1): integer arithmetic keeps the wave-uniform count on the scalar unit (summing the four
  // booleans goes through v_cndmask 0/1).  Fetched here, in front of the streaming loads (scalar loads of the operator
  // table: behind the loads they would be waited for a second time)
  const int    nyz = PACKED ? (int)hasS + (int)hasN + (int)hasD + (int)hasU : min(j, 1) + min(L.ny - 1 - j, 1) + min(kg, 1) + min(L.nzg - 1 - kg, 1);
  const bool   two0 = hasW0 && hasE0, two1 = hasE1;
  const double idA = table_at<PACKED>(op.idiag, nyz + 1), idB = table_at<PACKED>(op.idiag, nyz + 2);
  const double sqA = NOISY ? table_at<PACKED>(op.sqrtdiag, nyz + 1) : 0.0, sqB = NOISY ? table_at<PACKED>(op.sqrtdiag, nyz + 2) : 0.0;
  double      *out_row = y_own + rowoff;
  d2     Vc = ld2(at_bytes(yo_row, lo));
  double ed = *at_bytes(yo_row, lo + 8u * (uint32_t)eo);
  d2     oS = ld2(at_bytes(yo_row - (hasS ? L.sx : 0), lo));
  d2     oN = ld2(at_bytes(yo_row + (hasN ? L.sx : 0), lo));
  const int64_t inplane = (int64_t)j * L.sx + 2 * t; // offset inside one plane
  d2      oD = (HALO && k == 0 && halo.glo) ? ld2_sys(halo.glo + inplane) : ld2(at_bytes(yo_row - (hasD ? L.sp : 0), lo));
  d2      oU = (HALO && k == L.nz - 1 && halo.ghi) ? ld2_sys(halo.ghi + inplane) : ld2(at_bytes(yo_row + (hasU ? L.sp : 0), lo));
#ifndef PMG_GRID_NO_NT
  // b is read once and y_own written once per colour pass: non-temporal accesses keep them from displacing the other
  // colour's lines, which five neighbouring rows re-read (measured: 612 -> 597 us per 512^3 sweep)
  d2     bb = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(at_bytes(b_own + rowoff, lo)));
#else
  d2     bb = ld2(at_bytes(b_own + rowoff, lo));
#endif
  pmg::RngConsts K;
  if (NOISY) {
    __builtin_amdgcn_sched_barrier(0); // all eight requests go out before the generator starts
    K = pmg::load_sincos_consts();     // scalar loads of the polynomial coefficients: issued here (the row pointers'
    __builtin_amdgcn_sched_barrier(0); // registers are free again), waited for behind the Philox rounds
  }

  // the noise first, while the eight loads above are in flight: nothing below may touch a loaded value before z0, z1 exist
  // (left alone the scheduler starts with the selects on Vc / ed and parks the wavefront on its first load before the
  // ~150 instructions of the generator, which need no memory at all)
  double z0 = 0.0, z1 = 0.0;
  if (NOISY) {
    pmg::normal_pair_fill((uint32_t)t, (uint32_t)(j + (int64_t)L.ny * kg), (uint32_t)op.sweep, ((uint32_t)(op.sweep >> 32) & 0x7fffffffu) | ((uint32_t)c << 31), op.key0, op.key1, tab_entry, tab, (int)threadIdx.x, K, z0, z1);
    asm volatile("" : "+v"(Vc.x), "+v"(Vc.y), "+v"(ed), "+v"(oS.x), "+v"(oS.y), "+v"(oN.x), "+v"(oN.y), "+v"(oD.x), "+v"(oD.y), "+v"(oU.x), "+v"(oU.y), "+v"(bb.x), "+v"(bb.y) : "v"(z0), "v"(z1));
  }
  const double L0 = p ? Vc.x : ed, R0 = p ? Vc.y : Vc.x, L1 = R0, R1 = p ? ed : Vc.y;
  const double idg0 = two0 ? idB : idA, idg1 = two1 ? idB : idA;
  const double hS = hasS ? h2 : 0.0, hN = hasN ? h2 : 0.0, hD = hasD ? h2 : 0.0, hU = hasU ? h2 : 0.0;

  double w0 = bb.x, w1 = bb.y;
  if (NOISY) {
    const double sq0 = two0 ? sqB : sqA, sq1 = two1 ? sqB : sqA;
    w0 = z0 * sq0 + bb.x;
    w1 = z1 * sq1 + bb.y;
  }
  // CSR storage order: (k-1) (j-1) (i-1) | (i+1) (j+1) (k+1); absent neighbours contribute an exact +0
  double s0 = w0, s1 = w1;
  s0 = s0 + hD * oD.x;
  s1 = s1 + hD * oD.y;
  s0 = s0 + hS * oS.x;
  s1 = s1 + hS * oS.y;
  s0 = s0 + (hasW0 ? h2 : 0.0) * L0;
  s1 = s1 + h2 * L1;
  s0 = s0 + (hasE0 ? h2 : 0.0) * R0;
  s1 = s1 + (hasE1 ? h2 : 0.0) * R1;
  s0 = s0 + hN * oN.x;
  s1 = s1 + hN * oN.y;
  s0 = s0 + hU * oU.x;
  s1 = s1 + hU * oU.y;

  double r0, r1;
  if (OMEGA1) {
    // (1-omega)*y == 0 exactly; the reference still adds it (src/mc_sor.c:267), which can only change the
    // sign of an exact zero -- numerically equal
    r0 = idg0 * s0;
    r1 = idg1 * s1;
  } else {
    const d2 yo2 = ld2(at_bytes(out_row, lo));
    r0           = op.one_minus_omega * yo2.x + idg0 * s0;
    r1           = op.one_minus_omega * yo2.y + idg1 * s1;
  }
  // the slot of a non-existent second point (odd nx) is a pad slot of this line: keep it zero
  const d2 out = {r0, v1 ? r1 : 0.0};
  if (NOISY) {
    // the whole computation stays in front of this exit (the optimiser would otherwise sink the streaming loads into the
    // `live` branch, behind the noise)
    asm volatile("" ::"v"(out.x), "v"(out.y), "s"(out_row));
    if (!live) return;
  }
#ifndef PMG_GRID_NO_NT
  __builtin_nontemporal_store(out, reinterpret_cast<d2 *>(at_bytes(out_row, lo)));
#else
  *reinterpret_cast<d2 *>(at_bytes(out_row, lo)) = out;
#endif
  if (HALO) {
    if (k == 0 && halo.plo) st2_sys(halo.plo + inplane, out);
    if (k == L.nz - 1 && halo.phi) st2_sys(halo.phi + inplane, out);
  }
}

// thread -> (t, line j) of its plane.  PACKED: the threads of a plane are numbered line after line with nbx = tplE
// threads per line and dealt to the wavefronts without gaps; otherwise one line per wavefront, grid = (8*nbx, band, nz)
// in XCD-banded order [the linear block id is blockIdx.x mod 8, so blockIdx.x & 7 is the XCD] or (nbx, nby, nz) plain
template <bool PACKED>
__device__ __forceinline__ void grid_thread_position(int nbx, int bandw, int ty, int &t, int &j)
{
  if (PACKED) {
    const int flat = ((int)blockIdx.x * 4 + ty) * 64 + (int)threadIdx.x;
    j              = flat / nbx;
    t              = flat - j * nbx;
  } else {
    // banded: XCD x owns the lines [x*bandw, (x+1)*bandw) -- bands counted in LINES, so that 8 bands of ceil(ny/8)
    // lines differ by less than one line tile (513 lines: 65 per XCD instead of 17 tiles = 68)
    const int bx = bandw > 0 ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int jl = (int)blockIdx.y * 4 + ty;
    t            = bx * 64 + threadIdx.x;
    j            = bandw > 0 ? (jl < bandw ? (int)(blockIdx.x & 7u) * bandw + jl : 0x40000000) : jl;
  }
}

// TAIL (lines of 64 m + a few threads -- the multigrid sizes 2^k+1 have tplE = 64 m + 1): the first `tmain` threads of
// every line run in the one-line-per-wavefront mapping with all lanes busy, and the `tailw` threads left over per line
// are collected, 256 per block, in extra blocks behind the last plane (blockIdx.z >= kcount) instead of occupying one
// nearly empty wavefront per line.  Same arithmetic per point and noise addressed by grid position, so the mapping does
// not change the results.
template <bool NOISY, bool OMEGA1, bool HALO, bool PACKED, bool TAIL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(96))) void grid_color_sweep_kernel(pmgk_grid_layout L, pmgk_grid_op op, int c, int nbx, int zmain, int bandw, int kbegin, int kstride, int kcount, int tmain, int tailw, pmgk_grid_halo halo, const double *__restrict__ b_own, const double *__restrict__ y_other, double *__restrict__ y_own)
{
  // blockDim.x == 64: a wavefront is one grid line, so everything that depends on (line, plane) only is
  // wave-uniform; readfirstlane tells the compiler, which then keeps the boundary logic on the scalar unit
  const int ty = __builtin_amdgcn_readfirstlane(threadIdx.y);
  __shared__ pmg::LogTabEntry s_logtab[NOISY ? 4 * PMG_LOGTAB_SIZE : 1];
  pmg::LogTabEntry           *tab = s_logtab + (NOISY ? ty * PMG_LOGTAB_SIZE : 0);
  // this lane's entry of the log table: requested FIRST, used last (normal_pair_fill) -- the fetch overlaps the
  // wavefront's streaming loads instead of standing in front of them (round 4; round 3 filled the LDS copy here and
  // waited: two dependent L2 round trips before a wavefront had a single load in flight)
  pmg::LogTabEntry            tab_entry = {0.0, 0.0};
  if (NOISY) {
    const d2 e = ld2(reinterpret_cast<const double *>(pmg::g_logtab + threadIdx.x));
    tab_entry  = {e.x, e.y};
  }
  int                         t, j;
  const int                   zsplit = zmain > 0 ? zmain : kcount; // z layers of main blocks; the tail blocks follow
  if (TAIL && (int)blockIdx.z >= zsplit) {
    const int tb  = (int)blockIdx.x + (int)gridDim.x * ((int)blockIdx.y + (int)gridDim.y * ((int)blockIdx.z - zsplit));
    // On the multigrid sizes nx = 4 tmain + 1 the one tail thread of a line owns the point i = nx - 1 alone, and that point
    // has this colour only on the lines with (c + j + k) even: the tails of those lines are collected (every lane of a tail
    // block busy) instead of one thread per line, half of which would leave at once.  Measured on one box against the
    // one-thread-per-line form (tools/alignbench.py, A/B builds): 257^3 5.74 -> 5.56 ns per 1000 points, but 513^3 5.64 -> 5.78
    // -- so only where the lines are packed tightly (the Infinity-Cache regime of pmg_grid_line_stride), whose tails share
    // cache lines with the next line's head
    const bool compact = tailw == 1 && L.nx == 4 * tmain + 1 && (L.sx & 15) != 0;
    const int  slots   = compact ? (L.ny + 1) / 2 : L.ny * tailw;
    const int  per     = (slots + 255) / 256; // tail blocks per plane
    const int  kz = tb / per, part = tb - kz * per;
    if (kz >= kcount) return;
    const int f = (part * 4 + ty) * 64 + (int)threadIdx.x, kt = kbegin + kz * kstride;
    if (compact) {
      j = 2 * f + ((c + kt + L.kz0) & 1);
      t = tmain;
    } else {
      j = f / tailw;
      t = tmain + f - j * tailw;
    }
    grid_color_sweep_body<NOISY, OMEGA1, false, true>(L, op, c, t, j, kt, tab, tab_entry, halo, b_own, y_other, y_own);
    return;
  }
  grid_thread_position<PACKED>(nbx, bandw, ty, t, j);

  int k = kbegin + (int)blockIdx.z * kstride;
  if (!HALO && !PACKED && zmain > 0) {
    // FLAT (round 4): the wavefronts of an XCD walk the (plane, line) pairs of its band without gaps instead of a rectangle of
    // line tiles per plane.  513 lines are eight bands of 65 = sixteen workgroups of four lines and a seventeenth with ONE: three
    // wavefronts per band and plane that leave at once but had to be launched (4.4 % of the launches, 8.3 % at 257 lines) --
    // measured 2.7 % of the sweep at ny = 520, more at 257 (tools/oddbench.py).  A wavefront's plane and line are wave-uniform as
    // before; the four wavefronts of a workgroup share nothing, so a workgroup may straddle two planes.
    // bandw = floor(ny / 8) lines per band here, the ny - 8 bandw < 8 lines left over go one each to the LAST bands: the first
    // bands start on whole multiples of bandw lines (measured: -1 % against eight bands of ceil(ny / 8) with a short last one)
    const int xcd = (int)(blockIdx.x & 7u), extra = max(xcd - (8 - (L.ny - 8 * bandw)), 0);
    const int first = xcd * bandw + extra, nl = bandw + (xcd >= 8 - (L.ny - 8 * bandw) ? 1 : 0);
    const int w     = ((int)blockIdx.y + (int)gridDim.y * (int)blockIdx.z) * 4 + ty;
    if (nl <= 0) return;
    const int kz = w / nl;
    if (kz >= kcount) return; // whole wavefront (the last z layers of the shorter bands)
    t = (int)(blockIdx.x >> 3) * 64 + (int)threadIdx.x;
    j = first + (w - kz * nl);
    k = kbegin + kz * kstride;
  }
  if (HALO && halo.full) { // face planes first: they carry the halo traffic
    const int z = (int)blockIdx.z;
    k           = z == 0 ? 0 : (z == 1 ? L.nz - 1 : z - 1);
    const bool face_lo = k == 0, face_hi = k == L.nz - 1;
    if (face_lo && halo.wlo) wave_wait_flag(halo.wlo, halo.wval, halo.err, halo.spins);
    if (face_hi && halo.whi) wave_wait_flag(halo.whi, halo.wval, halo.err, halo.spins);
    grid_color_sweep_body<NOISY, OMEGA1, HALO, PACKED>(L, op, c, t, j, k, tab, tab_entry, halo, b_own, y_other, y_own);
    if (face_lo || face_hi) { // every block of a face plane reports; the last one tells the neighbours
      // the peer stores are system-scope write-through stores (st2_sys): waiting for their completion is all a
      // wavefront has to do -- a system-scope release FENCE here would also write the whole L2 back, once per
      // wavefront, while the interior blocks are filling it (measured: 2.5x slower).  The wait must be EXPLICIT and
      // per wavefront: a workgroup-scope fence + barrier only drains lgkmcnt on gfx950, so without this s_waitcnt
      // three of the four wavefronts of a face block could still have plane stores in flight over xGMI when the
      // block's reporting thread bumps the counter (checked on the ISA by tests/test_isa_halo_wait.py)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads(); // one report per block: thousands of wavefronts on one counter would queue up at the L2
      if (threadIdx.x == 0 && threadIdx.y == 0) {
        const unsigned nface  = L.nz > 1 ? 2u : 1u;
        const unsigned expect = gridDim.x * gridDim.y * nface;
        if (atomicAdd(halo.counter, 1u) == expect - 1u) {
          *halo.counter = 0;
          // every face wavefront waited for the completion of its peer stores before it reported, so the planes are
          // in the neighbours' memory: a relaxed store of the flag word is ordered behind them
          if (halo.slo) __hip_atomic_store(halo.slo, halo.sval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (halo.shi) __hip_atomic_store(halo.shi, halo.sval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    return;
  }
  grid_color_sweep_body<NOISY, OMEGA1, HALO, PACKED>(L, op, c, t, j, k, tab, tab_entry, halo, b_own, y_other, y_own);
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

// r = b - A y on cvecs, BOTH colours in one launch (y is read from HBM once), same tiling and XCD-banded order as
// the sweep: a thread owns the four consecutive grid points 4t .. 4t+3 of its line -- two of each colour, i.e. the 16-byte
// slot m = 2t, 2t+1 of both colour arrays.  Row sum in CSR storage order INCLUDING the diagonal at its place (what
// PETSc MatMult does), then r = b - s (VecAYPX(w,-1,b), reference src/pc_gamgmc.c:253-254).
struct residual_consts {
  double a, dA, dB, aS, aN, aD, aU;
};

__device__ __forceinline__ d2 residual_pair(const residual_consts &K, bool hasW0, bool hasE0, bool hasE1, bool v1, d2 oD, d2 oS, d2 oN, d2 oU, double L0, double R0, double L1, double R1, d2 yy, d2 bb)
{
  const double dg0 = (hasW0 && hasE0) ? K.dB : K.dA, dg1 = hasE1 ? K.dB : K.dA;
  double       s0 = 0.0, s1 = 0.0;
  s0 = s0 + K.aD * oD.x;
  s1 = s1 + K.aD * oD.y;
  s0 = s0 + K.aS * oS.x;
  s1 = s1 + K.aS * oS.y;
  s0 = s0 + (hasW0 ? K.a : 0.0) * L0;
  s1 = s1 + K.a * L1;
  s0 = s0 + dg0 * yy.x;
  s1 = s1 + dg1 * yy.y;
  s0 = s0 + (hasE0 ? K.a : 0.0) * R0;
  s1 = s1 + (hasE1 ? K.a : 0.0) * R1;
  s0 = s0 + K.aN * oN.x;
  s1 = s1 + K.aN * oN.y;
  s0 = s0 + K.aU * oU.x;
  s1 = s1 + K.aU * oU.y;
  const d2 out = {bb.x - s0, v1 ? bb.y - s1 : 0.0};
  return out;
}

// the residuals of the four points 4t .. 4t+3 of line (j, k): outA = points 4t, 4t+2 (colour (j + k) & 1, the even x),
// outB = points 4t+1, 4t+3 (zeros behind the line end)
template <bool PACKED>
__device__ __forceinline__ void grid_residual_values(const pmgk_grid_layout &L, const pmgk_grid_op &op, int t, int j, int k, const double *__restrict__ b, const double *__restrict__ y, int64_t &offa, int64_t &offb, d2 &outA, d2 &outB)
{
  const int      kg   = k + L.kz0;
  const int      ca   = (j + kg) & 1; // colour of the points 4t, 4t+2 (p = 0); the other colour owns 4t+1, 4t+3 (p = 1)
  const bool     hasS = j > 0, hasN = j < L.ny - 1, hasD = kg > 0, hasU = kg < L.nzg - 1;
  const int64_t  rowoff = (int64_t)(k + 1) * L.sp + (int64_t)j * L.sx; // scalar row bases + 32-bit lane byte offsets, as in the sweep
  const uint32_t lo     = 16u * (uint32_t)t;
  offa = (int64_t)ca * L.cs + rowoff;
  offb = (int64_t)(1 - ca) * L.cs + rowoff;
  const double  *ya = y + offa, *yb = y + offb;
  const int64_t  dS = hasS ? L.sx : 0, dN = hasN ? L.sx : 0, dD = hasD ? L.sp : 0, dU = hasU ? L.sp : 0;
  const d2       Ya = ld2(at_bytes(ya, lo)), Yb = ld2(at_bytes(yb, lo));
  const double   edA = *at_bytes(yb, lo - (t > 0 ? 8u : 0u));              // west neighbour of point 4t: m = 2t-1 of the other colour
  const double   edB = *at_bytes(ya, lo + (2 * t + 2 < L.sx ? 16u : 8u));  // east neighbour of point 4t+3: m = 2t+2
  const int      nyz = (int)hasS + (int)hasN + (int)hasD + (int)hasU;
  residual_consts K;
  K.a  = -op.h2;
  K.dA = table_at<PACKED>(op.diag, nyz + 1);
  K.dB = table_at<PACKED>(op.diag, nyz + 2);
  K.aS = hasS ? K.a : 0.0;
  K.aN = hasN ? K.a : 0.0;
  K.aD = hasD ? K.a : 0.0;
  K.aU = hasU ? K.a : 0.0;
  const int i0 = 4 * t;
  // colour ca: points i0, i0+2, neighbours in the other colour's array
  outA = residual_pair(K, i0 > 0, i0 < L.nx - 1, i0 + 2 < L.nx - 1, i0 + 2 < L.nx, ld2(at_bytes(yb - dD, lo)), ld2(at_bytes(yb - dS, lo)), ld2(at_bytes(yb + dN, lo)), ld2(at_bytes(yb + dU, lo)), edA, Yb.x, Yb.x, Yb.y, Ya, ld2(at_bytes(b + offa, lo)));
  outB = d2{0.0, 0.0};
  if (i0 + 1 < L.nx) // the other colour: points i0+1, i0+3
    outB = residual_pair(K, true, i0 + 1 < L.nx - 1, i0 + 3 < L.nx - 1, i0 + 3 < L.nx, ld2(at_bytes(ya - dD, lo)), ld2(at_bytes(ya - dS, lo)), ld2(at_bytes(ya + dN, lo)), ld2(at_bytes(ya + dU, lo)), Ya.x, Ya.y, Ya.y, edB, Yb, ld2(at_bytes(b + offb, lo)));
}

template <bool PACKED>
__device__ __forceinline__ void grid_residual_body(const pmgk_grid_layout &L, const pmgk_grid_op &op, int t, int j, int k, const double *__restrict__ b, const double *__restrict__ y, double *__restrict__ r)
{
  if (j >= L.ny || 2 * t >= L.sx || 4 * t >= L.nx) return;
  int64_t offa, offb;
  d2      outA, outB;
  grid_residual_values<PACKED>(L, op, t, j, k, b, y, offa, offb, outA, outB);
  const uint32_t lo = 16u * (uint32_t)t;
  *reinterpret_cast<d2 *>(at_bytes(r + offa, lo)) = outA;
  if (4 * t + 1 < L.nx) *reinterpret_cast<d2 *>(at_bytes(r + offb, lo)) = outB;
}

// ---- residual + Q1 restriction in one pass (single-device grid level of the V-cycle) --------------------------------
// b_coarse = P^T (b - A y) without writing r: saves one write and one read of a fine vector per cycle.  Thread = the
// coarse points I = 2t, 2t+1 of ONE coarse line J, i.e. the fine points 4t-1 .. 4t+3 of the fine lines 2J-1, 2J, 2J+1,
// marching through the fine planes of a chunk of coarse planes.  Values by x parity: ev = y at x = 4t, 4t+2, od = y at
// x = 4t+1, 4t+3; x = 4t-1 and 4t+4 are the neighbouring lanes' values (DPP), so lanes 0 and 63 of a wavefront only
// supply their neighbours (62 of 64 store).  The residuals are formed by the expression of the residual kernel and
// summed in the order of q1_restrict_pair_kernel (planes, then lines, then x), so the result has the SAME BITS as
// residual-then-restrict.  The threads of a coarse plane are numbered line after line and dealt to the wavefronts without
// gaps; XCD x takes a contiguous run of the (chunk, wavefront) list.
// The z neighbours stay in registers: the thread carries its three fine lines of the planes k-1 and k through the march
// and loads plane k+1 once (it is U now, C in the next step, D in the one after); only the lines beside its three (S of
// the first, N of the last) are fetched per plane: 16 loads per plane step, and every y plane crosses the L2 once per
// thread instead of three times -- between two uses lie the steps of all resident wavefronts, 10+ MB per XCD, beyond
// its L2.  Measured (tools/rrbench.py): 513^3 578 us against 1064 us for the two kernels, 257^3 88 against 152 us; the
// variant that re-loads the z neighbours per plane (rolling window over lines instead) took 833 / 122 us.
__device__ __forceinline__ double lane_prev_d(double v) // lane i <- lane i-1
{
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ double lane_next_d(double v) // lane i <- lane i+1
{
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, false));
}

struct rr_acc {
  double s0, s1;
};
struct rr_thread {
  int      t, J0;
  uint32_t lo;
  bool     left, right0, right1;                 // the fine points 4t-1, 4t+1, 4t+3 exist
  bool     eW0, eE0, eE1, eV1;                   // even points 4t, 4t+2: neighbours / existence as residual_pair wants them
  bool     oE0, oE1, oV1, ovalid;                // odd points 4t+1, 4t+3
};

__device__ __forceinline__ void rr_add(rr_acc &a, double wyz, const rr_thread &T, double Lf, const d2 &E, const d2 &O)
{
  const double h = 0.5 * wyz;
  a.s0 = a.s0 + (T.left ? h : 0.0) * Lf;
  a.s0 = a.s0 + wyz * E.x;
  a.s0 = a.s0 + (T.right0 ? h : 0.0) * O.x;
  a.s1 = a.s1 + h * O.x;
  a.s1 = a.s1 + wyz * E.y;
  a.s1 = a.s1 + (T.right1 ? h : 0.0) * O.y;
}

struct rr_line {
  d2 ev, od;
};

// where the fine plane g (GLOBAL index) of a vector lives: in the cvec itself (the slab's planes and its two ghost planes),
// or -- z-slabs only -- in one of the two extra planes lo2 / hi2 (the planes kz0 - 2 and kz0 + nz + 1: colour 0, then colour 1,
// sp doubles each), which the residual of a ghost plane reads.  Wave-uniform.
struct rr_planes {
  const double *v, *lo2, *hi2;
  int           kz0, nz;
  uint32_t      sp8, cs8;
};
struct rr_plane {
  const double *p;
  uint32_t      base, cst;
};
__device__ __forceinline__ rr_plane rr_plane_of(const rr_planes &P, int g)
{
  const int kl = g - P.kz0; // plane inside the slab
  rr_plane  r;
  if (kl >= -1 && kl <= P.nz) {
    r.p    = P.v;
    r.base = (uint32_t)(kl + 1) * P.sp8;
    r.cst  = P.cs8;
  } else {
    r.p    = kl < 0 ? P.lo2 : P.hi2;
    r.base = 0;
    r.cst  = P.sp8;
  }
  return r;
}

__device__ __forceinline__ rr_line rr_load_line(const rr_plane &pl, uint32_t line, int parity)
{
  const uint32_t c = parity ? pl.cst : 0u;
  rr_line        r;
  r.ev = ld2(at_bytes(pl.p, pl.base + line + c));
  r.od = ld2(at_bytes(pl.p, pl.base + line + (pl.cst - c)));
  return r;
}

// L, C: fine layout and coarse extents of this rank (a single device: kz0 = 0, nz = nzg, ylo2 = yhi2 = NULL).  A z-slab
// restricts into the coarse planes it owns (K with fine plane 2K on this rank): it needs the residual on its two ghost
// planes, i.e. y two planes deep (ylo2, yhi2) and b on the ghost planes.
template <bool SYNC>
__global__ __launch_bounds__(256) void grid_residual_restrict_kernel(pmgk_grid_layout L, pmgk_grid_op op, pmgk_st27_dims C, int tplE, int wpp, int kc, int nchunks, const double *__restrict__ b, const double *__restrict__ y, const double *__restrict__ ylo2, const double *__restrict__ yhi2, double *__restrict__ bc)
{
  const int xcd = (int)blockIdx.x & 7, q = (int)blockIdx.x >> 3, per = (int)gridDim.x >> 3;
  const int gw  = __builtin_amdgcn_readfirstlane((xcd * per + q) * 4 + (int)(threadIdx.x >> 6));
  // SYNC: the wavefronts of a chunk are counted in whole workgroups (wppb = wpp rounded up to 4), so that the four wavefronts
  // of a workgroup -- neighbours in the (coarse line, x) list, which read each other's fine lines -- belong to ONE chunk, make
  // the same number of plane steps and can meet at a barrier per step (the spare wavefronts of a chunk repeat its last slots
  // without storing)
  const int wppb = SYNC ? (wpp + 3) / 4 * 4 : wpp;
  const int zc = gw / wppb, wv = gw - zc * wppb;
  if (zc >= nchunks) return; // whole wavefront (SYNC: whole workgroup)
  const int  lane = threadIdx.x & 63, nslots = C.ny * tplE;
  const int  fu = 62 * wv + lane - 1, f = min(max(fu, 0), nslots - 1);
  const bool store = lane >= 1 && lane <= 62 && fu < nslots;
  rr_thread  T;
  const int  Jc = f / tplE;
  T.t      = f - Jc * tplE;
  T.J0     = Jc;
  T.lo     = 16u * (uint32_t)T.t;
  const int i0 = 4 * T.t, I0 = 2 * T.t;
  T.left   = I0 > 0;
  T.right0 = i0 + 1 < L.nx;
  T.right1 = i0 + 3 < L.nx;
  T.eW0    = i0 > 0;
  T.eE0    = i0 < L.nx - 1;
  T.eE1    = i0 + 2 < L.nx - 1;
  T.eV1    = i0 + 2 < L.nx;
  T.oE0    = i0 + 1 < L.nx - 1;
  T.oE1    = i0 + 3 < L.nx - 1;
  T.oV1    = i0 + 3 < L.nx;
  T.ovalid = i0 + 1 < L.nx;
  const int      K0 = C.kz0 + zc * kc, K1 = min(K0 + kc, C.kz0 + C.nz); // coarse planes of this chunk (global)
  const int      kfirst = max(2 * K0 - 1, 0), klast = min(2 * K1 - 1, L.nzg - 1), jmax = L.ny - 1; // fine planes (global)
  const uint32_t sx8 = 8u * (uint32_t)L.sx;
  const rr_planes PY = {y, ylo2, yhi2, L.kz0, L.nz, 8u * (uint32_t)L.sp, 8u * (uint32_t)L.cs};
  const rr_planes PB = {b, b, b, L.kz0, L.nz, 8u * (uint32_t)L.sp, 8u * (uint32_t)L.cs}; // b is read on residual planes only: never beyond the ghost planes
  const double   ma = -op.h2;
  rr_acc         a = {0.0, 0.0}, n = {0.0, 0.0};
  double        *out = bc + I0 + (int64_t)C.nx * (Jc + (int64_t)C.ny * (K0 - C.kz0 + 1));
  const int64_t  cplane = (int64_t)C.nx * C.ny;
  rr_line        Dv[3], Cv[3], Uv[3], hS, hN; // hS, hN: the lines 2J-2 and 2J+2 of the plane k (S of the first line, N of the last)
  {
    const int      kd = max(kfirst - 1, 0);
    const rr_plane pd = rr_plane_of(PY, kd), pc = rr_plane_of(PY, kfirst);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int      jc = min(max(2 * Jc - 1 + i, 0), jmax);
      const uint32_t ln = (uint32_t)jc * sx8 + T.lo;
      Dv[i] = rr_load_line(pd, ln, (jc + kd) & 1);
      Cv[i] = rr_load_line(pc, ln, (jc + kfirst) & 1);
    }
    const int jS = min(max(2 * Jc - 2, 0), jmax), jN = min(max(2 * Jc + 2, 0), jmax);
    hS = rr_load_line(pc, (uint32_t)jS * sx8 + T.lo, (jS + kfirst) & 1);
    hN = rr_load_line(pc, (uint32_t)jN * sx8 + T.lo, (jN + kfirst) & 1);
  }
  for (int k = kfirst; k <= klast; ++k) {
    // one barrier per plane step keeps the four wavefronts of a workgroup on the same plane: the two lines beside a
    // wavefront's three and the line it shares with its neighbour are then requested by both within one step and the second
    // request finds them in the CU's L1 / the XCD's L2 instead of fetching them again over the fabric (all resident
    // wavefronts of an XCD together touch more than its L2 holds per step, so without the barrier a drifting neighbour's
    // line is gone before it is needed)
    if (SYNC) __syncthreads();
    int J = Jc; // per plane: what depends on the lines alone would otherwise be kept in registers across the march
    asm volatile("" : "+v"(J));
    const bool     hasD = k > 0, hasU = k < L.nzg - 1, odd = k & 1; // wave-uniform
    const int      nzc = (int)hasD + (int)hasU, ku = hasU ? k + 1 : k;
    const double   dg2 = uniform(op.diag[nzc + 2]), dg3 = uniform(op.diag[nzc + 3]), dg4 = uniform(op.diag[nzc + 4]);
    const double   wz = odd ? 0.5 : 1.0;
    const rr_plane pu = rr_plane_of(PY, ku), pb = rr_plane_of(PB, k);
    // the lines beside the thread's three are fetched a plane ahead like its own: in the step in which their owners load
    // them, so that one of the two requests finds the line in the L2
    const int      jS = min(max(2 * J - 2, 0), jmax), jN = min(max(2 * J + 2, 0), jmax);
#ifdef PMG_RR_PROBE_IDEAL /* timing only: every wavefront fetches the two fine lines it would OWN in a tiled form, nothing else */
    (void)jS;
    (void)jN;
#pragma unroll
    for (int i = 1; i < 3; ++i) {
      const int      jc = min(max(2 * J - 1 + i, 0), jmax);
      Uv[i] = rr_load_line(pu, (uint32_t)jc * sx8 + T.lo, (jc + ku) & 1);
    }
    Uv[0] = Uv[2];
    const rr_line hSu = Uv[1], hNu = Uv[1];
#else
    const rr_line  hSu = rr_load_line(pu, (uint32_t)jS * sx8 + T.lo, (jS + ku) & 1);
    const rr_line  hNu = rr_load_line(pu, (uint32_t)jN * sx8 + T.lo, (jN + ku) & 1);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int      jc = min(max(2 * J - 1 + i, 0), jmax);
      Uv[i] = rr_load_line(pu, (uint32_t)jc * sx8 + T.lo, (jc + ku) & 1);
    }
#endif
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int      j = 2 * J - 1 + i, jc = min(max(j, 0), jmax);
#ifdef PMG_RR_PROBE_IDEAL
      const int      jb = min(max(2 * J - 1 + (i == 0 ? 2 : i), 0), jmax);
      const rr_line  bb = rr_load_line(pb, (uint32_t)jb * sx8 + T.lo, (jb + k) & 1);
#else
      const rr_line  bb = rr_load_line(pb, (uint32_t)jc * sx8 + T.lo, (jc + k) & 1);
#endif
      const rr_line &S = i == 0 ? hS : Cv[i > 0 ? i - 1 : 0], &N = i == 2 ? hN : Cv[i < 2 ? i + 1 : 2];
      const bool     hasS = jc > 0, hasN = jc < jmax, inner = hasS && hasN;
      residual_consts Kc;
      Kc.a  = ma;
      Kc.dA = inner ? dg3 : dg2;
      Kc.dB = inner ? dg4 : dg3;
      Kc.aS = hasS ? ma : 0.0;
      Kc.aN = hasN ? ma : 0.0;
      Kc.aD = hasD ? ma : 0.0;
      Kc.aU = hasU ? ma : 0.0;
      const double edA = lane_prev_d(Cv[i].od.y), edB = lane_next_d(Cv[i].ev.x);
      const d2     E = residual_pair(Kc, T.eW0, T.eE0, T.eE1, T.eV1, Dv[i].ev, S.ev, N.ev, Uv[i].ev, edA, Cv[i].od.x, Cv[i].od.x, Cv[i].od.y, Cv[i].ev, bb.ev);
      d2           O = residual_pair(Kc, true, T.oE0, T.oE1, T.oV1, Dv[i].od, S.od, N.od, Uv[i].od, Cv[i].ev.x, Cv[i].ev.y, Cv[i].ev.y, edB, Cv[i].od, bb.od);
      if (!T.ovalid) O = d2{0.0, 0.0};
      const double Lf = lane_prev_d(O.y);
      const double w  = ((unsigned)j < (unsigned)L.ny) ? (i == 1 ? wz : 0.5 * wz) : 0.0;
      rr_add(a, w, T, Lf, E, O);
      if (odd) rr_add(n, w, T, Lf, E, O);
    }
    if (odd) { // plane 2K+1 closes coarse plane K (unless it only opened the chunk's first one) and opens K+1
      if (k > 2 * K0 - 1) {
        if (store) {
          out[0] = a.s0;
          if (I0 + 1 < C.nx) out[1] = a.s1;
        }
        out += cplane;
      }
      a = n;
      n = rr_acc{0.0, 0.0};
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      Dv[i] = Cv[i];
      Cv[i] = Uv[i];
    }
    hS = hSu;
    hN = hNu;
  }
  if (2 * K1 - 1 > klast && store) { // the top coarse plane has no plane above it
    out[0] = a.s0;
    if (I0 + 1 < C.nx) out[1] = a.s1;
  }
}

template <bool PACKED, bool TAIL>
__global__ __launch_bounds__(256) void grid_residual_kernel(pmgk_grid_layout L, pmgk_grid_op op, int bandw, int tplE, int tmain, int tailw, const double *__restrict__ b, const double *__restrict__ y, double *__restrict__ r)
{
  const int ty = __builtin_amdgcn_readfirstlane(threadIdx.y);
  if (TAIL && (int)blockIdx.z >= L.nz) { // the threads left over per line, 256 per block (see the sweep kernel)
    const int tb  = (int)blockIdx.x + (int)gridDim.x * ((int)blockIdx.y + (int)gridDim.y * ((int)blockIdx.z - L.nz));
    const int per = (L.ny * tailw + 255) / 256;
    const int kz = tb / per, part = tb - kz * per;
    if (kz >= L.nz) return;
    const int f = (part * 4 + ty) * 64 + (int)threadIdx.x, j = f / tailw;
    grid_residual_body<true>(L, op, tmain + f - j * tailw, j, kz, b, y, r);
    return;
  }
  int t, j;
  grid_thread_position<PACKED>(tplE, bandw, ty, t, j);
  grid_residual_body<PACKED>(L, op, t, j, (int)blockIdx.z, b, y, r);
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? 0 : 1; }

} // namespace

// threads per line that own a point, and whether packing the lines into wavefronts pays: one line per wavefront
// fills at most two thirds of its lanes (257^3: 165 -> 129 us per sweep; at 513^3, 67 % full, the XCD-banded order of
// the unpacked mapping is worth more: 857 vs 883 us)
static inline int grid_threads_per_line(const pmgk_grid_layout *L) { return ((L->nx + 1) / 2 + 1) / 2; }
static inline bool grid_use_packed(const pmgk_grid_layout *L)
{
  static int env = -1;
  if (env < 0) {
    const char *e = getenv("PMG_GRID_PACKED");
    env           = e ? atoi(e) : 2; // 0 = never, 1 = always, 2 = by lane efficiency
  }
  if (env != 2) return env == 1;
  const int tplE = grid_threads_per_line(L), lanes = (L->sx / 2 + 63) / 64 * 64;
  return 2 * lanes >= 3 * tplE;
}

// how one plane's threads are dealt to the wavefronts
struct grid_mapping {
  bool packed, tail;
  int  nbx, nby, bandw, tmain, tailw, ztail, zmain;
  dim3 grid;
};

static grid_mapping grid_choose_mapping(const pmgk_grid_layout *L, int kcount, bool allow_tail, bool allow_flat = false)
{
  static int banded_env = -1, tail_env = -1, flat_env = -1;
  if (banded_env < 0) {
    const char *e = getenv("PMG_GRID_BANDED");
    banded_env    = e ? atoi(e) : 1;
    e             = getenv("PMG_GRID_TAIL");
    tail_env      = e ? atoi(e) : 1;
    e             = getenv("PMG_GRID_FLAT");
    flat_env      = e ? atoi(e) : 1;
  }
  grid_mapping M;
  const int    tpl = L->sx / 2, tplE = grid_threads_per_line(L);
  M.nby   = (L->ny + 3) / 4;
  M.tmain = tplE / 64 * 64;
  M.tailw = tplE - M.tmain;
  // a few threads more than full wavefronts per line: full wavefronts + collected tails
  M.tail   = allow_tail && tail_env && M.tmain > 0 && M.tailw > 0 && M.tailw <= 8;
  M.packed = !M.tail && grid_use_packed(L);
  M.nbx    = M.packed ? tplE : (M.tail ? M.tmain / 64 : (tpl + 63) / 64);
  // XCD-banded dispatch order needs enough line tiles to give every XCD a band
  M.bandw = (!M.packed && banded_env && M.nby >= 16) ? (L->ny + 7) / 8 : 0; // lines per XCD band
  M.ztail = 0;
  M.zmain = 0;
  if (M.packed) M.grid = dim3((unsigned)(((int64_t)L->ny * tplE + 255) / 256), 1, kcount);
  else {
    const unsigned gx = M.bandw > 0 ? 8 * M.nbx : M.nbx, gy = M.bandw > 0 ? (M.bandw + 3) / 4 : M.nby;
    // bands that are not whole line tiles, or a short last band: the (plane, line) pairs of a band in one run (see the kernel)
    if (allow_flat && flat_env && M.bandw > 0 && ((M.bandw & 3) || 8 * M.bandw != L->ny)) {
      M.bandw           = L->ny / 8; // floor: the lines left over go one each to the last bands (see the kernel)
      const int maxband = M.bandw + (L->ny > 8 * M.bandw ? 1 : 0);
      M.zmain           = (int)(((int64_t)maxband * kcount + 4 * gy - 1) / (4 * gy));
    }
    const int zlayers = M.zmain > 0 ? M.zmain : kcount;
    if (M.tail) {
      const int64_t nblocks = (int64_t)kcount * (((int64_t)L->ny * M.tailw + 255) / 256);
      M.ztail               = (int)((nblocks + (int64_t)gx * gy - 1) / ((int64_t)gx * gy));
    }
    M.grid = dim3(gx, gy, zlayers + M.ztail);
  }
  if (!M.tail) M.tmain = M.tailw = 0;
  return M;
}

template <bool NOISY, bool OMEGA1, bool HALO>
static void launch_sweep(const grid_mapping &M, dim3 block, hipStream_t s, const pmgk_grid_layout &L, const pmgk_grid_op &op, int color, int kbegin, int kstride, int kcount, const pmgk_grid_halo &h, const double *bo, const double *yo, double *ys)
{
  if (M.packed) hipLaunchKernelGGL((grid_color_sweep_kernel<NOISY, OMEGA1, HALO, true, false>), M.grid, block, 0, s, L, op, color, M.nbx, 0, M.bandw, kbegin, kstride, kcount, 0, 0, h, bo, yo, ys);
  else if (!HALO && M.tail) hipLaunchKernelGGL((grid_color_sweep_kernel<NOISY, OMEGA1, false, false, true>), M.grid, block, 0, s, L, op, color, M.nbx, M.zmain, M.bandw, kbegin, kstride, kcount, M.tmain, M.tailw, h, bo, yo, ys);
  else hipLaunchKernelGGL((grid_color_sweep_kernel<NOISY, OMEGA1, HALO, false, false>), M.grid, block, 0, s, L, op, color, M.nbx, HALO ? 0 : M.zmain, M.bandw, kbegin, kstride, kcount, 0, 0, h, bo, yo, ys);
}

extern "C" int pmgk_grid_color_sweep(const pmgk_grid_layout *L, const pmgk_grid_op *op, int color, int kbegin, int kcount, int kstride, const pmgk_grid_halo *halo, const double *b, double *y, void *stream)
{
  if (kcount <= 0) return 0;
  const grid_mapping M = grid_choose_mapping(L, kcount, !halo, !halo);
  const dim3         block(64, 4, 1);
  hipStream_t   s  = (hipStream_t)stream;
  const double *bo = b + (int64_t)color * L->cs, *yo = y + (int64_t)(1 - color) * L->cs;
  double       *ys = y + (int64_t)color * L->cs;
  pmgk_grid_halo h0;
  memset(&h0, 0, sizeof h0);
  if (halo) {
    if (op->noisy) {
      if (op->omega_is_one) launch_sweep<true, true, true>(M, block, s, *L, *op, color, kbegin, kstride, kcount, *halo, bo, yo, ys);
      else launch_sweep<true, false, true>(M, block, s, *L, *op, color, kbegin, kstride, kcount, *halo, bo, yo, ys);
    } else {
      if (op->omega_is_one) launch_sweep<false, true, true>(M, block, s, *L, *op, color, kbegin, kstride, kcount, *halo, bo, yo, ys);
      else launch_sweep<false, false, true>(M, block, s, *L, *op, color, kbegin, kstride, kcount, *halo, bo, yo, ys);
    }
  } else if (op->noisy) {
    if (op->omega_is_one) launch_sweep<true, true, false>(M, block, s, *L, *op, color, kbegin, kstride, kcount, h0, bo, yo, ys);
    else launch_sweep<true, false, false>(M, block, s, *L, *op, color, kbegin, kstride, kcount, h0, bo, yo, ys);
  } else {
    if (op->omega_is_one) launch_sweep<false, true, false>(M, block, s, *L, *op, color, kbegin, kstride, kcount, h0, bo, yo, ys);
    else launch_sweep<false, false, false>(M, block, s, *L, *op, color, kbegin, kstride, kcount, h0, bo, yo, ys);
  }
  return launch_status();
}

extern "C" int pmgk_grid_residual(const pmgk_grid_layout *L, const pmgk_grid_op *op, const double *b, const double *y, double *r, void *stream)
{
  if (L->nz <= 0) return 0;
  const grid_mapping M = grid_choose_mapping(L, L->nz, true);
  const dim3         block(64, 4, 1);
  const int          tplE = grid_threads_per_line(L);
  if (M.packed) hipLaunchKernelGGL((grid_residual_kernel<true, false>), M.grid, block, 0, (hipStream_t)stream, *L, *op, M.bandw, tplE, 0, 0, b, y, r);
  else if (M.tail) hipLaunchKernelGGL((grid_residual_kernel<false, true>), M.grid, block, 0, (hipStream_t)stream, *L, *op, M.bandw, tplE, M.tmain, M.tailw, b, y, r);
  else hipLaunchKernelGGL((grid_residual_kernel<false, false>), M.grid, block, 0, (hipStream_t)stream, *L, *op, M.bandw, tplE, 0, 0, b, y, r);
  return launch_status();
}

// b_coarse = P^T (b - A y) in one launch; returns -1 (nothing launched) where the fused kernel does not apply: semicoarsened
// or permuted coarse levels, even extents.  A z-slab (L->kz0, L->nz; C: the coarse planes it owns) passes y's planes
// kz0 - 2 and kz0 + nz + 1 in ylo2 / yhi2 (colour 0 plane, then colour 1 plane; NULL at a domain face) and needs b and y
// current on the ghost planes; a single device passes NULL.
// 1 where the fused residual + restriction applies to this (slab of a) grid level and its coarse level C; a pure function of
// the two layouts and of which of the planes kz0 - 2 / kz0 + nz + 1 the caller can supply -- the set-up of a distributed
// hierarchy evaluates it on every rank and agrees on the result before any rank relies on it
extern "C" int pmgk_grid_residual_restrict_applies(const pmgk_grid_layout *L, const pmgk_st27_dims *C, int have_lo2, int have_hi2)
{
  static const int off  = getenv("PMG_GRID_FUSED_RR") ? !atoi(getenv("PMG_GRID_FUSED_RR")) : 0;
  const int        tplE = grid_threads_per_line(L);
  const bool       slab = L->kz0 != 0 || L->nz != L->nzg;
  if (off || C->nz <= 0) return 0;
  if (!slab && (C->kz0 != 0 || C->nz != C->nzg)) return 0;
  if (slab && ((L->kz0 > 0 && !have_lo2) || (L->kz0 + L->nz < L->nzg && !have_hi2) || L->nz < 2)) return 0;
  if (L->nx < 3 || L->ny < 3 || L->nzg < 3 || !(L->nx & 1) || !(L->ny & 1) || !(L->nzg & 1)) return 0;
  if (C->nx != (L->nx + 1) / 2 || C->ny != (L->ny + 1) / 2 || C->nzg != (L->nzg + 1) / 2) return 0;
  if (tplE != (C->nx + 1) / 2 || (int64_t)C->ny * tplE >= ((int64_t)1 << 30)) return 0;
  if (2 * (int64_t)L->cs * 8 >= ((int64_t)1 << 32)) return 0; // 32-bit byte offsets inside a vector
  if (slab && (2 * C->kz0 < L->kz0 || 2 * (C->kz0 + C->nz - 1) >= L->kz0 + L->nz)) return 0; // a coarse plane belongs to the owner of its fine plane
  return 1;
}

extern "C" int pmgk_grid_residual_restrict(const pmgk_grid_layout *L, const pmgk_grid_op *op, const pmgk_st27_dims *C, const double *b, const double *y, const double *ylo2, const double *yhi2, double *bc, void *stream)
{
  static const int kc_env = getenv("PMG_GRID_RR_CHUNK") ? atoi(getenv("PMG_GRID_RR_CHUNK")) : 0;
  if (!pmgk_grid_residual_restrict_applies(L, C, ylo2 != nullptr, yhi2 != nullptr)) return -1;
  const int     tplE = grid_threads_per_line(L);
  // one barrier per plane step (see the kernel): -3 % time and fabric reads at 513^3 and 257^3, +5 % on grids of a few
  // microseconds, where the wavefronts barely drift (tools/rrbench.py, tools/pmc_one.sh; PMG_GRID_RR_SYNC = 0 | 1 forces)
  static const int sync_knob = getenv("PMG_GRID_RR_SYNC") ? atoi(getenv("PMG_GRID_RR_SYNC")) : -1;
  const int        sync_env  = sync_knob >= 0 ? sync_knob : ((int64_t)L->nx * L->ny * L->nz >= 8000000);
  const int     wpp = (int)(((int64_t)C->ny * tplE + 61) / 62);
  int           kc  = kc_env > 0 ? kc_env : 8; // coarse planes per chunk: each chunk re-reads one fine plane
  while (kc_env <= 0 && kc > 2 && (int64_t)wpp * ((C->nz + kc - 1) / kc) < 2048) kc >>= 1;
  const int     nchunks = (C->nz + kc - 1) / kc;
  if (sync_env) {
    const int64_t nblocks = (int64_t)((wpp + 3) / 4) * nchunks;
    hipLaunchKernelGGL(grid_residual_restrict_kernel<true>, dim3((unsigned)((nblocks + 7) / 8 * 8)), dim3(256), 0, (hipStream_t)stream, *L, *op, *C, tplE, wpp, kc, nchunks, b, y, ylo2, yhi2, bc);
  } else {
    const int64_t nblocks = ((int64_t)wpp * nchunks + 3) / 4;
    hipLaunchKernelGGL(grid_residual_restrict_kernel<false>, dim3((unsigned)((nblocks + 7) / 8 * 8)), dim3(256), 0, (hipStream_t)stream, *L, *op, *C, tplE, wpp, kc, nchunks, b, y, ylo2, yhi2, bc);
  }
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
