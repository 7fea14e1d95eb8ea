/* Plain-C driver on the C-ABI (no Python, no PETSc): the measurement loop of the reference's benchmark program
 * (examples/benchmark/main.cc:105-149 SamplerCreate / Burnin / Sample, :261-309 "Measure sampling time" and
 * "Measure IACT") on the DMDA problem of its PETSc problem class -- MatAssembleShiftedLaplaceFD, b = 1, x0 = 0
 * (examples/benchmark/problem_petsc.hh:155-156, examples/ex1.c:88,109).
 *
 *   pmg_bench [-dim 2|3] [-n <points per direction>] [-kappa <k>] [-n_burnin N] [-n_samples N]
 *             [-measure_sampling_time] [-measure_iact] [-view_sampler]
 *             [-ranks N [-share_device] [-dist_levels L] [-dump <file prefix>]]      (see "more than one rank" below)
 *             [any option of the samplers, e.g. -pc_type mcgibbs|sorgibbs|gamgmc|cholsampler
 *              -pc_mcgibbs_omega 1.2 -pc_mcgibbs_symmetric -gamgmc_pc_mg_levels 4 -gamgmc_mg_levels_pc_type mcgibbs ...]
 *
 * Options are handed to the library's options database exactly as PETSc's command line would be (a flag followed by
 * another flag or by nothing is a boolean).  The quantity of interest of the IACT measurement is the value at the
 * centre of the grid (the reference integrates against a measurement vector; one entry is the same kind of linear
 * functional and needs no extra kernel here).
 *
 * More than one rank (-ranks N, no MPI, no torch): the process forks N children BEFORE anything touches the GPU and stays
 * behind as a relay -- it never initialises HIP.  A child is one rank on device (rank mod #devices), or all on device 0 with
 * -share_device (rehearsal on a one-GPU box).  The ranks bootstrap the "ipc" halo transport with pmg_dist_create_comm over a
 * byte all-gather made of pipes through the relay (the pmg_host_comm callback a PETSc adapter fills with MPI_Allgather),
 * split the n^3 DMDA in z-slabs and run the reference's distributed chains through the C-ABI alone:
 *   -pc_type mcgibbs | sorgibbs   pmg_dist_sample_cvec       (MCSORApply_MPIAIJ's role on a DMDA, reference src/mc_sor.c:298-381)
 *   -pc_type gamgmc               pmg_mgmc_create_dmda_slab  (PCGAMGMC over MPI ranks, reference src/pc_gamgmc.c), -dist_levels L
 * -dump P writes every rank's slab of the final sample to P.<rank> (natural order): the chains are keyed on global indices,
 * so the concatenation must be the SAME BYTES for every N (tests/test_gpu_pc_layer.py checks N = 1 against N = 2).
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <parmgmc_hip.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

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

/* ---- more than one rank: fork, pipes, a relay --------------------------------------------------------------------- */
typedef struct {
  int to_relay, from_relay;
} pipe_comm;

static int read_all(int fd, void *buf, size_t n)
{
  char *p = (char *)buf;
  while (n) {
    const ssize_t r = read(fd, p, n);
    if (r <= 0) return 1;
    p += r, n -= (size_t)r;
  }
  return 0;
}
static int write_all(int fd, const void *buf, size_t n)
{
  const char *p = (const char *)buf;
  while (n) {
    const ssize_t r = write(fd, p, n);
    if (r <= 0) return 1;
    p += r, n -= (size_t)r;
  }
  return 0;
}

/* pmg_allgather_fn over the relay: my block up, everybody's blocks down */
static int pipe_allgather(void *ctx, const void *send, int64_t nbytes, void *recv)
{
  const pipe_comm *pc = (const pipe_comm *)ctx;
  if (write_all(pc->to_relay, &nbytes, sizeof nbytes) || (nbytes && write_all(pc->to_relay, send, (size_t)nbytes))) return 1;
  int64_t total = 0;
  if (read_all(pc->from_relay, &total, sizeof total) || total < 0) return 1;
  return total ? read_all(pc->from_relay, recv, (size_t)total) : 0;
}

/* the parent: gathers one block from every child, hands all of them to every child, until the children hang up.
   Never calls into HIP or the library's device code. */
static int relay(int nranks, const int *up, const int *down, const pid_t *pids)
{
  int    failed = 0;
  char  *buf = NULL;
  size_t cap = 0;
  for (;;) {
    int64_t nb = -1, total = 0;
    int     eof = 0;
    for (int r = 0; r < nranks && !eof; ++r) {
      int64_t mine;
      if (read_all(up[r], &mine, sizeof mine)) {
        eof = 1;
        break;
      }
      if (nb < 0) nb = mine;
      if (mine != nb || mine < 0) { /* ranks disagree on the block size: tell everybody */
        failed = 1;
        nb = -1;
        eof = 1;
        break;
      }
      if ((size_t)(nb * nranks) > cap) {
        cap = (size_t)(nb * nranks);
        buf = (char *)realloc(buf, cap ? cap : 1);
        if (!buf) return 1;
      }
      if (nb && read_all(up[r], buf + (size_t)nb * (size_t)r, (size_t)nb)) {
        eof = 1;
        failed = 1;
      }
    }
    if (eof) break;
    total = nb * nranks;
    for (int r = 0; r < nranks; ++r)
      if (write_all(down[r], &total, sizeof total) || (total && write_all(down[r], buf, (size_t)total))) failed = 1;
  }
  free(buf);
  for (int r = 0; r < nranks; ++r) {
    close(down[r]); /* a rank still waiting for blocks sees end-of-file and fails instead of hanging */
    int st = 0;
    if (waitpid(pids[r], &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) failed = 1;
  }
  return failed;
}

static void barrier(const pmg_host_comm *hc)
{
  char c = 0, all[64];
  if (hc->nranks > 1) hc->allgather(hc->ctx, &c, 1, all);
}

/* one rank of the distributed run */
static int run_rank(const pmg_host_comm *hc, int share, int n, double kappa, const char *pc_type, int levels, int n_burnin, int n_samples, const char *dump)
{
  const int np = hc->nranks, me = hc->rank;
  int       ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev < 1) {
    fprintf(stderr, "no GPU\n");
    return 1;
  }
  HIPCHK(hipSetDevice(share ? 0 : me % ndev));
  int32_t cuts[65];
  for (int r = 0; r <= np; ++r) cuts[r] = (int32_t)(((int64_t)n * r) / np); /* PETSc's ownership rule up to the remainder's place */
  const int32_t kz0 = cuts[me], nzl = cuts[me + 1] - cuts[me];
  const int64_t Nl  = (int64_t)n * n * nzl;
  pmg_grid g = NULL;
  pmg_dist d = NULL;
  pmg_mgmc mg = NULL;
  CHK(pmg_grid_create(n, n, n, kz0, nzl, kappa, &g));
  CHK(pmg_dist_create_comm(hc, "ipc", g, NULL, &d));
  const int is_mg = !strcmp(pc_type, "gamgmc"), scaled = strcmp(pc_type, "sorgibbs") != 0;
  double   *b = NULL, *y = NULL, *bc = NULL, *yc = NULL, *ones = (double *)malloc(sizeof(double) * (size_t)Nl);
  if (!ones) return 1;
  for (int64_t i = 0; i < Nl; ++i) ones[i] = 1.0;
  HIPCHK(hipMalloc((void **)&b, sizeof(double) * (size_t)Nl));
  HIPCHK(hipMalloc((void **)&y, sizeof(double) * (size_t)Nl));
  HIPCHK(hipMemcpy(b, ones, sizeof(double) * (size_t)Nl, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(y, 0, sizeof(double) * (size_t)Nl));
  uint64_t ctr = 0;
  double   t0 = now();
  if (is_mg) {
    CHK(pmg_mgmc_create_dmda_slab(n, n, n, kappa, levels, g, d, cuts, &mg));
    CHK(pmg_mgmc_setup(mg));
  } else {
    int64_t len = 0;
    CHK(pmg_grid_cvec_len(g, &len));
    HIPCHK(hipMalloc((void **)&bc, sizeof(double) * (size_t)len));
    HIPCHK(hipMalloc((void **)&yc, sizeof(double) * (size_t)len));
    HIPCHK(hipMemset(bc, 0, sizeof(double) * (size_t)len));
    HIPCHK(hipMemset(yc, 0, sizeof(double) * (size_t)len));
    CHK(pmg_grid_to_cvec(g, b, bc, NULL));
  }
  HIPCHK(hipDeviceSynchronize());
  barrier(hc);
  if (me == 0) printf("Setup sampler: %.6f s (%d ranks, ipc transport)\n", now() - t0, np);
  for (int phase = 0; phase < 2; ++phase) { /* burn-in, then the timed samples */
    const int its = phase ? n_samples : n_burnin;
    barrier(hc);
    t0 = now();
    if (is_mg) CHK(pmg_mgmc_sample(mg, b, y, its, 0, 0xCAFE, ctr, &ctr, NULL, NULL, NULL));
    else CHK(pmg_dist_sample_cvec(d, bc, yc, its, scaled, PMG_SOR_FORWARD_SWEEP, 0xCAFE, ctr, &ctr, NULL));
    HIPCHK(hipDeviceSynchronize());
    CHK(pmg_dist_check(d));
    barrier(hc);
    const double t = now() - t0;
    if (me == 0 && phase) printf("Sampling: %.6f s\nTime per sample [ms]: %.6f\n", t, t / n_samples * 1000);
    if (phase) { /* the self-describing record of a multi-rank run, one line per rank in rank order (the same fields as
                    bench.py's "ranks"): where the rank ran, whether its device reaches its z-neighbours' devices, its own time */
      pmg_dist_description ds;
      const int            mydev = share ? 0 : me % ndev;
      CHK(pmg_dist_describe(d, me > 0 ? (share ? 0 : (me - 1) % ndev) : -1, me < np - 1 ? (share ? 0 : (me + 1) % ndev) : -1, &ds));
      for (int r = 0; r < np; ++r) {
        barrier(hc);
        if (r == me) {
          printf("rank %d/%d: device %d pci_bus_id %s transport %s neighbour_ranks [%d, %d] peer_access_lo_hi [%d, %d] rccl_comm_count %d halo_wait_polls %llu timed_region_s %.6f\n", ds.rank, ds.nranks, mydev, ds.pci_bus_id, ds.transport, ds.neighbour[0], ds.neighbour[1], ds.peer_access[0], ds.peer_access[1], ds.rccl_comm_count, (unsigned long long)ds.halo_wait_polls, t);
          fflush(stdout);
        }
      }
      barrier(hc);
    }
  }
  if (!is_mg) {
    CHK(pmg_grid_from_cvec(g, yc, y, NULL));
    HIPCHK(hipDeviceSynchronize());
  }
  if (dump) {
    char path[1024];
    snprintf(path, sizeof path, "%s.%d", dump, me);
    HIPCHK(hipMemcpy(ones, y, sizeof(double) * (size_t)Nl, hipMemcpyDeviceToHost));
    FILE *f = fopen(path, "wb");
    if (!f || fwrite(ones, sizeof(double), (size_t)Nl, f) != (size_t)Nl) {
      fprintf(stderr, "cannot write %s\n", path);
      return 1;
    }
    fclose(f);
  }
  if (me == 0) printf("Problem size (degrees of freedom): %lld\n", (long long)n * n * n);
  free(ones);
  CHK(pmg_mgmc_destroy(&mg)); /* before the transport and the slab it borrows */
  CHK(pmg_dist_destroy_comm(hc, &d));
  CHK(pmg_grid_destroy(&g));
  (void)hipFree(b), (void)hipFree(y), (void)hipFree(bc), (void)hipFree(yc);
  return 0;
}

int main(int argc, char **argv)
{
  int         dim = 3, n = 65, n_burnin = 20, n_samples = 100, t_sampling = 0, t_iact = 0, view = 0, ranks = 0, share = 0, dist_levels = 3;
  double      kappa = 10.0;
  const char *dump = NULL, *pc_type = "mcgibbs";
  CHK(pmg_initialize()); /* registers the PC types: host tables only, no GPU call */
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
    else if (!strcmp(argv[a], "-ranks") && val) ranks = atoi(val);
    else if (!strcmp(argv[a], "-share_device")) share = 1;
    else if (!strcmp(argv[a], "-dist_levels") && val) dist_levels = atoi(val);
    else if (!strcmp(argv[a], "-dump") && val) dump = val;
    else {
      if (!strcmp(argv[a], "-pc_type") && val) pc_type = val;
      CHK(pmg_options_set_value(argv[a], val ? val : ""));
    }
    if (val && strcmp(argv[a], "-measure_sampling_time") && strcmp(argv[a], "-measure_iact") && strcmp(argv[a], "-view_sampler") && strcmp(argv[a], "-share_device")) ++a;
  }
  if (ranks >= 1) { /* the distributed run: fork BEFORE the first HIP call; this process becomes the relay */
    if (dim != 3 || ranks > 64 || ranks > n) {
      fprintf(stderr, "-ranks: a 3-D grid with at least one plane per rank, at most 64 ranks\n");
      return 1;
    }
    int   up[64], down[64];
    pid_t pids[64];
    fflush(stdout);
    for (int r = 0; r < ranks; ++r) {
      int u[2], dn[2];
      if (pipe(u) || pipe(dn)) return 1;
      pids[r] = fork();
      if (pids[r] < 0) return 1;
      if (pids[r] == 0) { /* child = rank r */
        close(u[0]), close(dn[1]);
        for (int q = 0; q < r; ++q) close(up[q]), close(down[q]);
        pipe_comm     pcm = {u[1], dn[0]};
        pmg_host_comm hc  = {r, ranks, pipe_allgather, &pcm};
        const int     rc  = run_rank(&hc, share, n, kappa, pc_type, dist_levels, n_burnin, n_samples, dump);
        fflush(stdout);
        close(u[1]), close(dn[0]);
        _exit(rc);
      }
      close(u[1]), close(dn[0]);
      up[r] = u[0], down[r] = dn[1];
    }
    return relay(ranks, up, down, pids);
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
