/* Error reporting, version strings and device-memory helpers of libparmgmc_hip. */
#include "pmg_internal.h"
#include <pthread.h>
#include <dlfcn.h>
#include <stdarg.h>

static _Thread_local char pmg_errbuf[512] = "";

pmg_status pmg_set_error(pmg_status code, const char *file, int line, const char *fmt, ...)
{
  va_list ap;
  int     n = snprintf(pmg_errbuf, sizeof pmg_errbuf, "[pmg error %d] %s:%d: ", code, file, line);
  if (n < 0) n = 0;
  if ((size_t)n < sizeof pmg_errbuf) {
    va_start(ap, fmt);
    vsnprintf(pmg_errbuf + n, sizeof pmg_errbuf - (size_t)n, fmt, ap);
    va_end(ap);
  }
  return code;
}

const char *pmg_last_error_string(void) { return pmg_errbuf; }
const char *pmg_version(void) { return PMG_VERSION_STRING; }
const char *pmg_gpu_arch(void) { return "gfx950"; }

pmg_status pmg_dev_alloc(void **p, size_t bytes)
{
  *p = NULL;
  if (bytes == 0) bytes = 8;
  PMG_HIP(hipMalloc(p, bytes));
  PMG_HIP(hipMemset(*p, 0, bytes));
  return PMG_SUCCESS;
}

pmg_status pmg_dev_upload(void **p, const void *host, size_t bytes)
{
  PMG_CALL(pmg_dev_alloc(p, bytes));
  if (bytes) PMG_HIP(hipMemcpy(*p, host, bytes, hipMemcpyHostToDevice));
  return PMG_SUCCESS;
}

void pmg_dev_free(void *p)
{
  if (p) (void)hipFree(p);
}

pmg_status pmg_vec_set_random_standard_normal(int64_t n, double *x_dev, uint64_t seed, uint64_t counter, void *stream)
{
  PMG_CHECK(n >= 0, PMG_ERR_ARG_OUTOFRANGE, "negative length %lld", (long long)n);
  PMG_CHECK(x_dev || n == 0, PMG_ERR_ARG_NULL, "null vector");
  pmg_trace_begin(PMG_EVENT_VEC_SET_RANDOM_NORMAL);
  const int rc = pmgk_fill_normal_rows(n, seed, counter, x_dev, stream);
  pmg_trace_end();
  PMG_KERNEL(rc);
  return PMG_SUCCESS;
}

/* measurement aid: one launch of c = a + 0.5 b over n doubles with the colour sweep's access mix (include/parmgmc_hip.h) */
pmg_status pmg_stream_triad(int64_t n, const double *a_dev, const double *b_dev, double *c_dev, void *stream)
{
  PMG_CHECK(n >= 0 && (n & 1) == 0, PMG_ERR_ARG_OUTOFRANGE, "an even, non-negative length expected, got %lld", (long long)n);
  PMG_CHECK(n < ((int64_t)1 << 37), PMG_ERR_ARG_OUTOFRANGE, "length %lld too large for one launch", (long long)n); /* n / 512 workgroups */
  PMG_CHECK((a_dev && b_dev && c_dev) || n == 0, PMG_ERR_ARG_NULL, "null vector");
  PMG_CHECK(((uintptr_t)a_dev | (uintptr_t)b_dev | (uintptr_t)c_dev) % 16 == 0, PMG_ERR_ARG_WRONG, "16-byte aligned device vectors expected");
  PMG_KERNEL(pmgk_stream_triad(n, a_dev, b_dev, c_dev, stream));
  return PMG_SUCCESS;
}

/* Index arrays of the caller's PetscInt width (reference include/parmgmc/parmgmc.h:18-24 builds against 32- and 64-bit
   PetscInt; the arrays come from MatSeqAIJGetCSRAndMemType, src/mc_sor.c:250).  The library's own index type is 32-bit
   (an MI355X holds far fewer than 2^31 rows of an AIJ matrix per device): width 32 borrows the arrays, width 64 makes
   checked 32-bit copies (*rp_own / *ci_own, malloc'ed, the caller frees them) and fails with PETSC_ERR_ARG_OUTOFRANGE
   if a row pointer or a column index does not fit. */
pmg_status pmg_narrow_csr(int64_t nrows, int64_t ncols, const void *rowptr, const void *colidx, int idx_width, const int32_t **rp, const int32_t **ci, int32_t **rp_own, int32_t **ci_own)
{
  *rp_own = *ci_own = NULL;
  PMG_CHECK(idx_width == 32 || idx_width == 64, PMG_ERR_ARG_OUTOFRANGE, "idx_width = %d (32 or 64: sizeof(PetscInt) * 8)", idx_width);
  PMG_CHECK(nrows >= 0 && nrows < 2147483647 && ncols < 2147483647, PMG_ERR_ARG_OUTOFRANGE, "%lld x %lld exceeds the 32-bit local sizes of the library", (long long)nrows, (long long)ncols);
  PMG_CHECK(rowptr, PMG_ERR_ARG_NULL, "null row pointer array");
  if (idx_width == 32) {
    *rp = (const int32_t *)rowptr;
    *ci = (const int32_t *)colidx;
    return PMG_SUCCESS;
  }
  const int64_t *rp64 = (const int64_t *)rowptr, *ci64 = (const int64_t *)colidx;
  const int64_t  nnz  = rp64[nrows];
  PMG_CHECK(nnz >= 0 && nnz < 2147483647, PMG_ERR_ARG_OUTOFRANGE, "%lld stored entries exceed the 32-bit row pointers of the library", (long long)nnz);
  PMG_CHECK(ci64 || nnz == 0, PMG_ERR_ARG_NULL, "null column index array");
  int32_t *a = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nrows + 1)), *b = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
  if (!a || !b) {
    free(a);
    free(b);
    PMG_FAIL(PMG_ERR_MEM, "out of host memory");
  }
  for (int64_t r = 0; r <= nrows; ++r) {
    if (rp64[r] < 0 || rp64[r] > nnz || (r > 0 && rp64[r] < rp64[r - 1])) {
      free(a);
      free(b);
      PMG_FAIL(PMG_ERR_ARG_WRONG, "row pointer %lld = %lld is not monotone within [0, %lld]", (long long)r, (long long)rp64[r], (long long)nnz);
    }
    a[r] = (int32_t)rp64[r];
  }
  for (int64_t k = 0; k < nnz; ++k) {
    if (ci64[k] < 0 || ci64[k] >= ncols) {
      free(a);
      free(b);
      PMG_FAIL(PMG_ERR_ARG_OUTOFRANGE, "column index %lld at entry %lld outside [0, %lld)", (long long)ci64[k], (long long)k, (long long)ncols);
    }
    b[k] = (int32_t)ci64[k];
  }
  *rp = *rp_own = a;
  *ci = *ci_own = b;
  return PMG_SUCCESS;
}

/* ---- trace ranges ------------------------------------------------------------------------------------------------
   The reference brackets its two hot functions with PETSc log events "MulticolSOR" (src/mc_sor.c:221,237) and
   "VecSetRandN" (src/parmgmc.c:75,114), registered in ParMGMCInitialize (src/parmgmc.c:118-127).  Here the same
   names are ROCTx ranges, so `rocprofv3 --marker-trace` shows them on the host time line next to the kernels.  The
   marker library is used only when it is already in the process (the profiler loaded it) or PMG_TRACE=1 asks for it;
   otherwise a range costs one predictable branch. */
typedef int (*pmg_roctx_push_fn)(const char *);
typedef int (*pmg_roctx_pop_fn)(void);
static pmg_roctx_push_fn pmg_roctx_push;
static pmg_roctx_pop_fn  pmg_roctx_pop;
static int               pmg_trace_on; /* written once, inside pthread_once, after both function pointers */
static pthread_once_t    pmg_trace_once = PTHREAD_ONCE_INIT;

/* Runs exactly once per process, however many host threads enter the C-ABI together (the error buffer is thread-local, so
   they may): pthread_once publishes the pointers and the flag to every thread that returns from it.  A range that one
   thread pushes is popped by the same thread, and both see the same answer -- the round-2 code flipped a plain int through
   "off" while it was still probing, so a second thread could skip a push and then pop. */
static void pmg_trace_probe(void)
{
  static const char *names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
  const char        *env     = getenv("PMG_TRACE");
  const int          force   = env && env[0] == '1';
  if (env && env[0] == '0') return;
  for (unsigned q = 0; q < sizeof names / sizeof names[0]; ++q) {
    void *h = dlopen(names[q], RTLD_LAZY | (force ? 0 : RTLD_NOLOAD));
    if (!h) continue;
    pmg_roctx_push_fn push = (pmg_roctx_push_fn)dlsym(h, "roctxRangePushA");
    pmg_roctx_pop_fn  pop  = (pmg_roctx_pop_fn)dlsym(h, "roctxRangePop");
    if (push && pop) {
      pmg_roctx_push = push;
      pmg_roctx_pop  = pop;
      pmg_trace_on   = 1;
      return;
    }
  }
}

void pmg_trace_begin(const char *name)
{
  pthread_once(&pmg_trace_once, pmg_trace_probe);
  if (pmg_trace_on) (void)pmg_roctx_push(name);
}

void pmg_trace_end(void)
{
  pthread_once(&pmg_trace_once, pmg_trace_probe);
  if (pmg_trace_on) (void)pmg_roctx_pop();
}

/* 1 if ranges are being emitted (diagnostic for the tests) */
int pmg_trace_enabled(void)
{
  pthread_once(&pmg_trace_once, pmg_trace_probe);
  return pmg_trace_on;
}
