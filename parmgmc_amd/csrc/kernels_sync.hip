// Message kernels of the "ipc" halo transport (gfx950): a producer copies planes into the consumer's receive block
// (peer memory over xGMI) and then raises 64-bit flag words there to the round number; the consumer waits for the words
// and copies the planes out.  Replaces the VecScatterBegin/End rendezvous of MCSORApply_MPIAIJ (reference
// src/mc_sor.c:317-340) outside the sweep kernels, which carry their own hand-shake (kernels_grid.hip).
#include <hip/hip_runtime.h>
#include "pmg_kernels.h"

namespace {

// ---- generic neighbour exchange in two launches -----------------------------------------------------------------
// push : copy up to PMGK_XCH_MAXSEG segments into the neighbours' message slots (peer stores), then the LAST block to
//        finish raises the neighbours' flag words (every thread fences its stores at system scope first);
// pull : every block waits for my flag words, then the segments are copied out of my slots.  The slots live in
//        fine-grained memory, so the loads after the acquire see the peer's stores without a kernel boundary.
// SRC_SHARED / DST_SHARED: that side lives in a peer-shared receive block and is accessed with system-scope relaxed
// atomics (write-through / cache-bypassing), so that "my s_waitcnt returned" means "performed for the peer"
template <bool SRC_SHARED, bool DST_SHARED>
__device__ __forceinline__ void copy_segments(const pmgk_xch_args &a)
{
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  for (int q = 0; q < a.nseg; ++q) {
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(a.src[q]);
    unsigned long long       *dst = reinterpret_cast<unsigned long long *>(a.dst[q]);
    for (int64_t i = tid; i < a.n[q]; i += nth) {
      const unsigned long long v = SRC_SHARED ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : src[i];
      if (DST_SHARED) __hip_atomic_store(dst + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      else dst[i] = v;
    }
  }
}

__global__ __launch_bounds__(256) void xch_push_kernel(pmgk_xch_args a, unsigned *counter, unsigned *dbg)
{
  copy_segments<false, true>(a);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned done = atomicAdd(counter, 1u);
    if (done == gridDim.x - 1) {
      *counter = 0;
      __threadfence_system();
      for (int q = 0; q < 4; ++q)
        if (a.flag[q]) __hip_atomic_store(a.flag[q], a.value[q], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      if (dbg) { // diagnostics: number of pushes that raised their flags, and the last sequence number raised
        dbg[0] += 1;
        dbg[1] = (unsigned)a.value[0] | (unsigned)a.value[1] << 16;
      }
    } else if (dbg && done >= gridDim.x) dbg[2] = done; // the counter did not start at zero
  }
}

__global__ __launch_bounds__(256) void xch_pull_kernel(pmgk_xch_args a, unsigned *err)
{
  __shared__ int ok;
  if (threadIdx.x == 0) {
    ok = 1;
    for (int q = 0; q < 4 && ok; ++q) {
      if (!a.flag[q]) continue;
      unsigned long long spins = 0;
      while (__hip_atomic_load(a.flag[q], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < a.value[q]) {
        __builtin_amdgcn_s_sleep(32);
        ++spins;
        const bool gave_up = (spins & 0xFFFu) == 0 && err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (gave_up || spins > (1ull << 26)) {
          if (err && !gave_up) { // what was waited for: [1] = wait site, [2] = expected, [3] = seen (low words)
            err[1] = 0x30u + (unsigned)q;
            err[2] = (unsigned)a.value[q];
            err[3] = (unsigned)__hip_atomic_load(a.flag[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
          if (err) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          ok = 0;
          break;
        }
      }
    }
  }
  __syncthreads();
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  copy_segments<true, false>(a);
}

// ---- all-gather over all-peer mappings ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void allgather_push_kernel(int nranks, int me, const double *__restrict__ src, int64_t n, double *const *__restrict__ dst, int64_t dst_off, uint64_t *const *__restrict__ flag, uint64_t value, unsigned *counter)
{
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  const unsigned long long *s = reinterpret_cast<const unsigned long long *>(src);
  for (int p = 0; p < nranks; ++p) {
    if (p == me || !dst[p]) continue;
    unsigned long long *d = reinterpret_cast<unsigned long long *>(dst[p] + dst_off);
    for (int64_t i = tid; i < n; i += nth) __hip_atomic_store(d + i, s[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(counter, 1u) == gridDim.x - 1) {
    *counter = 0;
    __threadfence_system();
    for (int p = 0; p < nranks; ++p)
      if (p != me && flag[p]) __hip_atomic_store(flag[p], value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ void allgather_wait_kernel(int nranks, int me, const uint64_t *myflags, uint64_t value, unsigned *err)
{
  const int p = threadIdx.x;
  if (p >= nranks || p == me) return;
  unsigned long long spins = 0;
  while (__hip_atomic_load(myflags + p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < value) {
    __builtin_amdgcn_s_sleep(32);
    ++spins;
    const bool gave_up = (spins & 0xFFFu) == 0 && err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (gave_up || spins > (1ull << 26)) {
      if (err && !gave_up) {
        err[1] = 0x40u + (unsigned)p;
        err[2] = (unsigned)value;
        err[3] = (unsigned)__hip_atomic_load(myflags + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if (err) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
  }
}

__global__ void sum_rows_kernel(int nrows, int count, const double *__restrict__ in, double *__restrict__ out)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= count) return;
  double s = 0.0;
  for (int r = 0; r < nrows; ++r) s = s + in[(int64_t)r * count + c];
  out[c] = s;
}

// zero fill at streaming rate: the runtime's fill kernel runs 256 workgroups whatever the size (136 MB: 182 us; this
// one: one 16-byte store per thread, ~30 us)
typedef double d2z __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void fill_zero_kernel(double *__restrict__ p, int64_t n2, int64_t n)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n2) reinterpret_cast<d2z *>(p)[i] = d2z{0.0, 0.0};
  if (i == 0 && (n & 1)) p[n - 1] = 0.0;
}

// c = a + 0.5 b with the sweep kernel's access shape: 16 bytes per lane, one tile of 256 lanes per block, blocks in memory
// order -- the access mix of one colour pass (read two streams, write one), nothing else.  What this runs at is the ceiling a
// kernel of that mix can reach on the device it is measured on ("a measured device-copy/triad ceiling").
__global__ __launch_bounds__(256) void stream_triad_kernel(const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ c, int64_t n2)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n2) return;
  const d2z x = reinterpret_cast<const d2z *>(a)[i], y = reinterpret_cast<const d2z *>(b)[i];
  reinterpret_cast<d2z *>(c)[i] = d2z{x.x + 0.5 * y.x, x.y + 0.5 * y.y};
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? 0 : 1; }

} // namespace

// n even, pointers 16-byte aligned
extern "C" int pmgk_stream_triad(int64_t n, const double *a, const double *b, double *c, void *stream)
{
  if (n <= 0) return 0;
  const int64_t n2 = n / 2;
  hipLaunchKernelGGL(stream_triad_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, c, n2);
  return launch_status();
}

// p must be 16-byte aligned (device allocations are)
extern "C" int pmgk_fill_zero(double *p, int64_t n, void *stream)
{
  if (n <= 0) return 0;
  const int64_t n2 = n / 2, nb = (n2 + 255) / 256;
  hipLaunchKernelGGL(fill_zero_kernel, dim3((unsigned)(nb > 0 ? nb : 1)), dim3(256), 0, (hipStream_t)stream, p, n2, n);
  return launch_status();
}

extern "C" int pmgk_sum_rows(int nrows, int count, const double *in, double *out, void *stream)
{
  if (count <= 0) return 0;
  hipLaunchKernelGGL(sum_rows_kernel, dim3((count + 63) / 64), dim3(64), 0, (hipStream_t)stream, nrows, count, in, out);
  return launch_status();
}

extern "C" int pmgk_allgather_push(int nranks, int me, const double *src, int64_t n, double *const *dst_dev, int64_t dst_off, uint64_t *const *flag_dev, uint64_t value, unsigned *counter, void *stream)
{
  const int nb = n > 0 ? (int)((n + 4095) / 4096 < 64 ? (n + 4095) / 4096 : 64) : 1;
  hipLaunchKernelGGL(allgather_push_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, nranks, me, src, n, dst_dev, dst_off, flag_dev, value, counter);
  return launch_status();
}

extern "C" int pmgk_allgather_wait(int nranks, int me, const uint64_t *myflags, uint64_t value, unsigned *err, void *stream)
{
  hipLaunchKernelGGL(allgather_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, nranks, me, myflags, value, err);
  return launch_status();
}

extern "C" int pmgk_xch_push_dbg(const pmgk_xch_args *a, unsigned *counter, unsigned *dbg, void *stream)
{
  int64_t total = 0;
  for (int q = 0; q < a->nseg; ++q) total += a->n[q];
  const int nb = total > 0 ? (int)((total + 4095) / 4096 < 64 ? (total + 4095) / 4096 : 64) : 1;
  hipLaunchKernelGGL(xch_push_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, *a, counter, dbg);
  return launch_status();
}

extern "C" int pmgk_xch_push(const pmgk_xch_args *a, unsigned *counter, void *stream) { return pmgk_xch_push_dbg(a, counter, nullptr, stream); }

extern "C" int pmgk_xch_pull(const pmgk_xch_args *a, unsigned *err, void *stream)
{
  int64_t total = 0;
  for (int q = 0; q < a->nseg; ++q) total += a->n[q];
  const int nb = total > 0 ? (int)((total + 4095) / 4096 < 64 ? (total + 4095) / 4096 : 64) : 1;
  hipLaunchKernelGGL(xch_pull_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, *a, err);
  return launch_status();
}

// ---- ghost updates of the row-block distributed sampler (pmg_distmcsor.c; reference VecScatter of src/mc_sor.c:318-319) ----
namespace {
__global__ __launch_bounds__(256) void gather_idx_kernel(int64_t n, const int32_t *__restrict__ idx, const double *__restrict__ src, double *__restrict__ dst)
{
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q < n) dst[q] = src[idx[q]];
}
__global__ __launch_bounds__(256) void scatter_idx_kernel(int64_t n, const int32_t *__restrict__ src_idx, const int32_t *__restrict__ dst_idx, const double *__restrict__ src, double *__restrict__ dst)
{
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q < n) dst[dst_idx[q]] = src[src_idx[q]];
}
} // namespace

extern "C" int pmgk_gather_idx(int64_t n, const int32_t *idx, const double *src, double *dst, void *stream)
{
  if (n <= 0) return 0;
  hipLaunchKernelGGL(gather_idx_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, idx, src, dst);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

extern "C" int pmgk_scatter_idx(int64_t n, const int32_t *src_idx, const int32_t *dst_idx, const double *src, double *dst, void *stream)
{
  if (n <= 0) return 0;
  hipLaunchKernelGGL(scatter_idx_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, src_idx, dst_idx, src, dst);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
