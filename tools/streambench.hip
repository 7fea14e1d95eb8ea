// Streaming-bandwidth calibration for the sweep's access mix (development tool): read 2 x N/2 doubles, write
// N/2 doubles, N = n^3 -- the algorithmic traffic of one colour pass -- with 16-byte accesses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct d2 { double x, y; };
__global__ __launch_bounds__(256) void triad(const d2 *__restrict__ a, const d2 *__restrict__ b, d2 *__restrict__ c, size_t n2)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
    d2 x = a[i], y = b[i];
    c[i] = d2{x.x + 0.5 * y.x, x.y + 0.5 * y.y};
  }
}
__global__ __launch_bounds__(256) void copyk(const d2 *__restrict__ a, d2 *__restrict__ c, size_t n2)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) c[i] = a[i];
}
__global__ __launch_bounds__(256) void readk(const d2 *__restrict__ a, double *out, size_t n2)
{
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) { d2 x = a[i]; s += x.x + x.y; }
  if (s == 1.2345) out[0] = s;
}
__global__ __launch_bounds__(256) void writek(d2 *__restrict__ c, size_t n2)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) c[i] = d2{1.0, 2.0};
}
int main(int argc, char **argv)
{
  int n = argc > 1 ? atoi(argv[1]) : 512;
  size_t N = (size_t)n * n * n, n2 = N / 4; // N/2 doubles per array = N/4 d2
  d2 *a, *b, *c; double *o;
  hipMalloc(&a, n2 * 16); hipMalloc(&b, n2 * 16); hipMalloc(&c, n2 * 16); hipMalloc(&o, 8);
  hipMemset(a, 0, n2 * 16); hipMemset(b, 0, n2 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int grids[] = {2048, 4096, 8192, 16384, (int)((n2 + 255) / 256)};
  for (int g : grids) {
    float ms;
    auto run = [&](const char *name, auto f, double bytes) {
      f(); hipDeviceSynchronize();
      hipEventRecord(e0); for (int r = 0; r < 20; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1); ms /= 20;
      printf("n=%d grid=%6d %-6s %8.1f us %8.1f GB/s\n", n, g, name, ms * 1e3, bytes / ms / 1e6);
    };
    run("triad", [&] { hipLaunchKernelGGL(triad, dim3(g), dim3(256), 0, 0, a, b, c, n2); }, 3.0 * n2 * 16);
    run("copy", [&] { hipLaunchKernelGGL(copyk, dim3(g), dim3(256), 0, 0, a, c, n2); }, 2.0 * n2 * 16);
    run("read", [&] { hipLaunchKernelGGL(readk, dim3(g), dim3(256), 0, 0, a, o, n2); }, 1.0 * n2 * 16);
    run("write", [&] { hipLaunchKernelGGL(writek, dim3(g), dim3(256), 0, 0, c, n2); }, 1.0 * n2 * 16);
  }
  return 0;
}
