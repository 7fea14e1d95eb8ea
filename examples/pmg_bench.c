/* Plain-C driver on the C-ABI (no Python, no PETSc): the measurement loop of the reference's benchmark program
 * (examples/benchmark/main.cc:105-149 SamplerCreate / Burnin / Sample, :261-309 "Measure sampling time" and
 * "Measure IACT") on the DMDA problem of its PETSc problem class -- MatAssembleShiftedLaplaceFD, b = 1, x0 = 0
 * (examples/benchmark/problem_petsc.hh:155-156, examples/ex1.c:88,109).
 *
 *   pmg_bench [-dim 2|3] [-n <points per direction>] [-kappa <k>] [-n_burnin N] [-n_samples N]
 *             [-measure_sampling_time] [-measure_iact] [-view_sampler]
 *             [any option of the samplers, e.g. -pc_type mcgibbs|sorgibbs|gamgmc|cholsampler
 *              -pc_mcgibbs_omega 1.2 -pc_mcgibbs_symmetric -gamgmc_pc_mg_levels 4 -gamgmc_mg_levels_pc_type mcgibbs ...]
 *
 * Options are handed to the library's options database exactly as PETSc's command line would be (a flag followed by
 * another flag or by nothing is a boolean).  The quantity of interest of the IACT measurement is the value at the
 * centre of the grid (the reference integrates against a measurement vector; one entry is the same kind of linear
 * functional and needs no extra kernel here).
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <parmgmc_hip.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define CHK(expr) \
  do { \
    const int s_ = (expr); \
    if (s_) { \
      fprintf(stderr, "%s failed: %s\n", #expr, pmg_last_error_string()); \
      return 1; \
    } \
  } while (0)
#define HIPCHK(expr) \
  do { \
    const hipError_t e_ = (expr); \
    if (e_ != hipSuccess) { \
      fprintf(stderr, "%s failed: %s\n", #expr, hipGetErrorString(e_)); \
      return 1; \
    } \
  } while (0)

static double now(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct {
  double *qois;
  int64_t index; /* entry of the sample that is recorded */
  int     failed;
} sample_ctx;

/* PCSetSampleCallback callback (SaveSample, examples/benchmark/main.cc:151-175): y is the device vector */
static int save_sample(int32_t it, const double *y_dev, int32_t n, void *ctx)
{
  sample_ctx *c = (sample_ctx *)ctx;
  (void)n;
  if (hipMemcpy(&c->qois[it], y_dev + c->index, sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) c->failed = 1;
  return 0;
}

int main(int argc, char **argv)
{
  int    dim = 3, n = 65, n_burnin = 20, n_samples = 100, t_sampling = 0, t_iact = 0, view = 0;
  double kappa = 10.0;
  CHK(pmg_initialize());
  CHK(pmg_options_set_value("-pc_type", "mcgibbs"));
  for (int a = 1; a < argc; ++a) {
    const char *val = (a + 1 < argc && !(argv[a + 1][0] == '-' && (argv[a + 1][1] < '0' || argv[a + 1][1] > '9') && argv[a + 1][1] != '.')) ? argv[a + 1] : NULL;
    if (!strcmp(argv[a], "-dim") && val) dim = atoi(val);
    else if (!strcmp(argv[a], "-n") && val) n = atoi(val);
    else if (!strcmp(argv[a], "-kappa") && val) kappa = atof(val);
    else if (!strcmp(argv[a], "-n_burnin") && val) n_burnin = atoi(val);
    else if (!strcmp(argv[a], "-n_samples") && val) n_samples = atoi(val);
    else if (!strcmp(argv[a], "-measure_sampling_time")) t_sampling = 1;
    else if (!strcmp(argv[a], "-measure_iact")) t_iact = 1;
    else if (!strcmp(argv[a], "-view_sampler")) view = 1;
    else CHK(pmg_options_set_value(argv[a], val ? val : ""));
    if (val && strcmp(argv[a], "-measure_sampling_time") && strcmp(argv[a], "-measure_iact") && strcmp(argv[a], "-view_sampler")) ++a;
  }
  if (!t_sampling && !t_iact) t_sampling = 1;
  const int32_t nz = dim == 3 ? n : 1;
  const int64_t N  = (int64_t)n * n * nz;

  pmg_mat A  = NULL;
  pmg_pc  pc = NULL;
  CHK(pmg_mat_create_dmda(n, n, nz, kappa, &A));
  double t0 = now();
  CHK(pmg_pc_create(&pc));
  CHK(pmg_pc_set_operators(pc, A));
  CHK(pmg_pc_set_from_options(pc)); /* KSPSetFromOptions, main.cc:113 */
  CHK(pmg_pc_setup(pc));            /* KSPSetUp, main.cc:125 */
  HIPCHK(hipDeviceSynchronize());
  printf("Setup sampler: %.6f s\n", now() - t0);

  double *b = NULL, *x = NULL, *ones = (double *)malloc(sizeof(double) * (size_t)N);
  if (!ones) return 1;
  for (int64_t i = 0; i < N; ++i) ones[i] = 1.0;
  HIPCHK(hipMalloc((void **)&b, sizeof(double) * (size_t)N));
  HIPCHK(hipMalloc((void **)&x, sizeof(double) * (size_t)N));
  HIPCHK(hipMemcpy(b, ones, sizeof(double) * (size_t)N, hipMemcpyHostToDevice));
  free(ones);

  if (t_sampling) {
    printf("################################################################################\n");
    printf("                              Measure sampling time\n");
    printf("################################################################################\n");
    HIPCHK(hipMemset(x, 0, sizeof(double) * (size_t)N));
    t0 = now();
    CHK(pmg_ksp_richardson_solve(pc, b, x, n_burnin, 0, NULL)); /* Burnin, main.cc:132-142 */
    HIPCHK(hipDeviceSynchronize());
    printf("Burn-in: %.6f s\n", now() - t0);
    t0 = now();
    CHK(pmg_ksp_richardson_solve(pc, b, x, n_samples, 1, NULL)); /* Sample, main.cc:144-149 */
    HIPCHK(hipDeviceSynchronize());
    const double t = now() - t0;
    printf("Sampling: %.6f s\n", t);
    printf("Time per sample [ms]: %.6f\n\n", t / n_samples * 1000);
  }
  if (t_iact) {
    printf("################################################################################\n");
    printf("                                  Measure IACT\n");
    printf("################################################################################\n");
    sample_ctx ctx = {(double *)calloc((size_t)n_samples + 1, sizeof(double)), (n / 2) + (int64_t)n * ((n / 2) + (int64_t)n * (nz / 2)), 0};
    if (!ctx.qois) return 1;
    HIPCHK(hipMemset(x, 0, sizeof(double) * (size_t)N));
    CHK(pmg_ksp_richardson_solve(pc, b, x, n_burnin, 0, NULL));
    CHK(pmg_pc_set_sample_callback(pc, save_sample, &ctx, NULL));
    t0 = now();
    CHK(pmg_ksp_richardson_solve(pc, b, x, n_samples, 1, NULL));
    HIPCHK(hipDeviceSynchronize());
    const double t = now() - t0;
    if (ctx.failed) {
      fprintf(stderr, "reading a sample back failed\n");
      return 1;
    }
    double tau   = 0;
    int    valid = 0;
    CHK(pmg_iact(n_samples, ctx.qois, &tau, NULL, &valid)); /* main.cc:285 */
    if (!valid) printf("WARNING: Chain is too short to give reliable IACT estimate (need at least %d)\n", (int)ceil(500 * tau));
    printf("IACT: %.5f\n", tau);
    printf("Time per independent sample [ms]: %.6f\n\n", (tau > 1 ? tau : 1) * t / n_samples * 1000);
    double mean = 0;
    for (int i = 0; i < n_samples; ++i) mean += ctx.qois[i] / n_samples;
    printf("Mean of the quantity of interest: %.6f\n", mean);
    free(ctx.qois);
  }
  printf("Problem size (degrees of freedom): %lld\n", (long long)N);
  if (view) {
    char buf[1024];
    CHK(pmg_pc_view(pc, buf, (int32_t)sizeof buf));
    printf("%s\n", buf);
  }
  CHK(pmg_pc_destroy(&pc));
  CHK(pmg_mat_destroy(&A));
  (void)hipFree(b);
  (void)hipFree(x);
  CHK(pmg_finalize());
  return 0;
}
