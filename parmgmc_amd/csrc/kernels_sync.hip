// Flag words for the "ipc" halo transport (gfx950): a producer raises a 64-bit word in the consumer's memory (peer
// memory over xGMI) to the round number AFTER its kernel has stored a boundary plane there; the consumer's stream runs
// a one-wave kernel that waits for the word before the kernel that reads the plane starts.  Both are separate launches
// on purpose: the kernel boundaries before the signal and after the wait are what makes the plane's ordinary stores
// visible to the ordinary loads of the consumer.  Replaces the per-colour VecScatterBegin/End rendezvous of
// MCSORApply_MPIAIJ (reference src/mc_sor.c:317-340).
#include <hip/hip_runtime.h>
#include "pmg_kernels.h"

namespace {

__global__ void flag_signal_kernel(uint64_t *p0, uint64_t *p1, uint64_t v0, uint64_t v1)
{
  if (threadIdx.x != 0) return;
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  if (p0) __hip_atomic_store(p0, v0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  if (p1) __hip_atomic_store(p1, v1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// waits until *f >= v for every non-null f; gives up after ~10 s (sets *err) so that a lost peer cannot hang the device
__global__ void flag_wait_kernel(const uint64_t *f0, const uint64_t *f1, uint64_t v0, uint64_t v1, unsigned *err)
{
  if (threadIdx.x != 0) return;
  const uint64_t *f[2] = {f0, f1};
  const uint64_t  v[2] = {v0, v1};
  for (int q = 0; q < 2; ++q) {
    if (!f[q]) continue;
    unsigned long long spins = 0;
    while (__hip_atomic_load(f[q], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v[q]) {
      __builtin_amdgcn_s_sleep(32);
      ++spins;
      if ((spins & 0xFFFu) == 0 && err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return; // somebody already gave up
      if (spins > (1ull << 24)) {
        if (err) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
      }
    }
  }
}

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

__global__ __launch_bounds__(256) void xch_push_kernel(pmgk_xch_args a, unsigned *counter)
{
  copy_segments<false, true>(a);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned done = atomicAdd(counter, 1u);
    if (done == gridDim.x - 1) {
      *counter = 0;
      __threadfence_system();
      if (a.flag[0]) __hip_atomic_store(a.flag[0], a.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      if (a.flag[1]) __hip_atomic_store(a.flag[1], a.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ __launch_bounds__(256) void xch_pull_kernel(pmgk_xch_args a, unsigned *err)
{
  __shared__ int ok;
  if (threadIdx.x == 0) {
    ok = 1;
    for (int q = 0; q < 2 && ok; ++q) {
      if (!a.flag[q]) continue;
      unsigned long long spins = 0;
      while (__hip_atomic_load(a.flag[q], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < a.value) {
        __builtin_amdgcn_s_sleep(32);
        ++spins;
        const bool gave_up = (spins & 0xFFFu) == 0 && err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (gave_up || spins > (1ull << 24)) {
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

inline int launch_status() { return hipGetLastError() == hipSuccess ? 0 : 1; }

} // namespace

extern "C" int pmgk_xch_push(const pmgk_xch_args *a, unsigned *counter, void *stream)
{
  int64_t total = 0;
  for (int q = 0; q < a->nseg; ++q) total += a->n[q];
  const int nb = total > 0 ? (int)((total + 4095) / 4096 < 64 ? (total + 4095) / 4096 : 64) : 1;
  hipLaunchKernelGGL(xch_push_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, *a, counter);
  return launch_status();
}

extern "C" int pmgk_xch_pull(const pmgk_xch_args *a, unsigned *err, void *stream)
{
  int64_t total = 0;
  for (int q = 0; q < a->nseg; ++q) total += a->n[q];
  const int nb = total > 0 ? (int)((total + 4095) / 4096 < 64 ? (total + 4095) / 4096 : 64) : 1;
  hipLaunchKernelGGL(xch_pull_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, *a, err);
  return launch_status();
}

extern "C" int pmgk_flag_signal(uint64_t *p0, uint64_t v0, uint64_t *p1, uint64_t v1, void *stream)
{
  if (!p0 && !p1) return 0;
  hipLaunchKernelGGL(flag_signal_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, p0, p1, v0, v1);
  return launch_status();
}

extern "C" int pmgk_flag_wait(const uint64_t *f0, uint64_t v0, const uint64_t *f1, uint64_t v1, unsigned *err, void *stream)
{
  if (!f0 && !f1) return 0;
  hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, f0, f1, v0, v1, err);
  return launch_status();
}
