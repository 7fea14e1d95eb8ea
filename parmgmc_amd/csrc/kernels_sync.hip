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
      if (++spins > (1ull << 24)) {
        if (err) atomicExch(err, 1u);
        return;
      }
    }
  }
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? 0 : 1; }

} // namespace

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
