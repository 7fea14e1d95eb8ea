/* Error reporting, version strings and device-memory helpers of libparmgmc_hip. */
#include "pmg_internal.h"
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
  PMG_KERNEL(pmgk_fill_normal_rows(n, seed, counter, x_dev, stream));
  return PMG_SUCCESS;
}
