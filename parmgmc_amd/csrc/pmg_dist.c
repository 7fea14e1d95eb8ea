/* Multi-GPU Gibbs sampling on a z-slab decomposition -- host side (C11): one process per GPU, the per-colour halo
 * exchange over RCCL (ncclSend / ncclRecv across xGMI), overlapped with the interior sweep.
 *
 * Replaces MCSORApply_MPIAIJ's per-colour ghost update (reference src/mc_sor.c:317-340: VecScatterBegin/End, then
 * the rows of the colour; plan built in MatCreateScatters :152-214).  In the colour-partitioned layout the boundary
 * plane of one colour is one contiguous block, so an exchange is ONE send and ONE receive per z-neighbour.
 * Schedule per colour c (the overlap idea of PCPARSOR, which starts `botsct` before its INT1 rows,
 * src/pc_parsor.c:739-745):
 *     compute stream:  wait[ghost(1-c) landed] -> sweep boundary planes 0, nz-1 of colour c -> event B_c
 *                      -> sweep interior planes 1..nz-2 of colour c
 *     comm stream   :  wait[B_c] -> ncclGroup{send own planes of colour c, recv ghost planes of colour c} -> event X_c
 * Both streams are in-order, so X_(1-c) of a later exchange implies the earlier exchange of c has completed, which
 * is what makes re-use of the send and ghost planes safe without further flags.  Noise depends on global indices
 * only: the chain is bit-identical for every number of devices.
 *
 * RCCL is loaded at run time (dlopen of the path the caller names -- the copy PyTorch bundles when used beside
 * torch, so that the process keeps one RCCL and one HIP runtime); no link-time dependency.
 */
#define _GNU_SOURCE
#include "pmg_internal.h"
#include <dlfcn.h>

typedef struct {
  char internal[128];
} pmg_nccl_uid; /* ncclUniqueId, NCCL_UNIQUE_ID_BYTES = 128 */
typedef void *pmg_nccl_comm;
#define PMG_NCCL_DOUBLE 8 /* ncclFloat64 */

typedef struct {
  void *handle;
  int (*GetUniqueId)(pmg_nccl_uid *);
  int (*CommInitRank)(pmg_nccl_comm *, int, pmg_nccl_uid, int);
  int (*CommDestroy)(pmg_nccl_comm);
  int (*Send)(const void *, size_t, int, int, pmg_nccl_comm, hipStream_t);
  int (*Recv)(void *, size_t, int, int, pmg_nccl_comm, hipStream_t);
  int (*GroupStart)(void);
  int (*GroupEnd)(void);
  const char *(*GetErrorString)(int);
} pmg_rccl_api;

static pmg_status rccl_load(const char *path, pmg_rccl_api *api)
{
  memset(api, 0, sizeof *api);
  api->handle = dlopen(path && path[0] ? path : "librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  PMG_CHECK(api->handle, PMG_ERR_LIB, "cannot load RCCL (%s): %s", path ? path : "librccl.so.1", dlerror());
#define PMG_SYM(field, name) \
  do { \
    *(void **)(&api->field) = dlsym(api->handle, name); \
    PMG_CHECK(api->field, PMG_ERR_LIB, "RCCL symbol %s missing", name); \
  } while (0)
  PMG_SYM(GetUniqueId, "ncclGetUniqueId");
  PMG_SYM(CommInitRank, "ncclCommInitRank");
  PMG_SYM(CommDestroy, "ncclCommDestroy");
  PMG_SYM(Send, "ncclSend");
  PMG_SYM(Recv, "ncclRecv");
  PMG_SYM(GroupStart, "ncclGroupStart");
  PMG_SYM(GroupEnd, "ncclGroupEnd");
  PMG_SYM(GetErrorString, "ncclGetErrorString");
#undef PMG_SYM
  return PMG_SUCCESS;
}

#define PMG_NCCL(d, expr) \
  do { \
    int pmg_r_ = (expr); \
    if (pmg_r_ != 0) return pmg_set_error(PMG_ERR_LIB, __FILE__, __LINE__, "%s: %s", #expr, (d)->api.GetErrorString(pmg_r_)); \
  } while (0)

struct pmg_dist_s {
  pmg_grid      g;
  int           rank, nranks, lo, hi; /* z-neighbours (-1 = physical boundary); lo == hi == rank in loopback mode */
  int           loopback;
  pmg_rccl_api  api;
  pmg_nccl_comm comm;
  hipStream_t   cs;       /* communication stream */
  hipEvent_t    evB[2];   /* boundary planes of colour c swept */
  hipEvent_t    evX[2];   /* exchange of colour c complete     */
  hipEvent_t    evS;      /* caller's stream reached the call  */
  int32_t       nz;
};

pmg_status pmg_dist_get_unique_id(const char *rccl_path, void *id128)
{
  PMG_CHECK(id128, PMG_ERR_ARG_NULL, "null id buffer");
  pmg_rccl_api api;
  PMG_CALL(rccl_load(rccl_path, &api));
  pmg_nccl_uid uid;
  const int    r = api.GetUniqueId(&uid);
  PMG_CHECK(r == 0, PMG_ERR_LIB, "ncclGetUniqueId: %s", api.GetErrorString(r));
  memcpy(id128, &uid, sizeof uid);
  return PMG_SUCCESS; /* the library handle stays open for the life of the process */
}

/* `g` must own the planes of rank `rank` out of `nranks` z-slabs (pmg_grid_create with kz0/nz).  loopback != 0
   (nranks must be 1) makes the single rank its own neighbour: the domain becomes periodic in z for the halo only --
   used to exercise the RCCL calls on one GPU. */
pmg_status pmg_dist_create(pmg_grid g, int32_t rank, int32_t nranks, const void *id128, const char *rccl_path, int loopback, pmg_dist *out)
{
  PMG_CHECK(out && g, PMG_ERR_ARG_NULL, "null argument");
  *out = NULL;
  PMG_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, PMG_ERR_ARG_OUTOFRANGE, "rank %d of %d", rank, nranks);
  PMG_CHECK(!loopback || nranks == 1, PMG_ERR_ARG_WRONG, "loopback needs a single rank");
  pmg_dist d = (pmg_dist)calloc(1, sizeof *d);
  PMG_CHECK(d, PMG_ERR_MEM, "out of host memory");
  d->g        = g;
  d->rank     = rank;
  d->nranks   = nranks;
  d->loopback = loopback;
  d->lo       = loopback ? rank : (rank > 0 ? rank - 1 : -1);
  d->hi       = loopback ? rank : (rank < nranks - 1 ? rank + 1 : -1);
  pmgk_grid_layout L;
  pmg_status       st = pmg_grid_get_kernel_layout(g, &L);
  d->nz               = L.nz;
  if (!st && (nranks > 1 || loopback)) {
    PMG_CHECK(id128, PMG_ERR_ARG_NULL, "null RCCL unique id");
    st = rccl_load(rccl_path, &d->api);
    if (!st) {
      pmg_nccl_uid uid;
      memcpy(&uid, id128, sizeof uid);
      const int r = d->api.CommInitRank(&d->comm, nranks, uid, rank);
      if (r != 0) st = pmg_set_error(PMG_ERR_LIB, __FILE__, __LINE__, "ncclCommInitRank: %s", d->api.GetErrorString(r));
    }
  }
  if (!st && hipStreamCreateWithFlags(&d->cs, hipStreamNonBlocking) != hipSuccess) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "stream creation failed");
  for (int c = 0; c < 2 && !st; ++c) {
    if (hipEventCreateWithFlags(&d->evB[c], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&d->evX[c], hipEventDisableTiming) != hipSuccess) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "event creation failed");
  }
  if (!st && hipEventCreateWithFlags(&d->evS, hipEventDisableTiming) != hipSuccess) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "event creation failed");
  if (st) {
    pmg_dist_destroy(&d);
    return st;
  }
  *out = d;
  return PMG_SUCCESS;
}

pmg_status pmg_dist_destroy(pmg_dist *dp)
{
  if (!dp || !*dp) return PMG_SUCCESS;
  pmg_dist d = *dp;
  if (d->comm && d->api.CommDestroy) d->api.CommDestroy(d->comm);
  for (int c = 0; c < 2; ++c) {
    if (d->evB[c]) (void)hipEventDestroy(d->evB[c]);
    if (d->evX[c]) (void)hipEventDestroy(d->evX[c]);
  }
  if (d->evS) (void)hipEventDestroy(d->evS);
  if (d->cs) (void)hipStreamDestroy(d->cs);
  free(d);
  *dp = NULL;
  return PMG_SUCCESS;
}

/* enqueue the exchange of colour c on the comm stream (after event `after`), record evX[c] */
static pmg_status dist_exchange(pmg_dist d, int c, double *y, hipEvent_t after)
{
  PMG_HIP(hipStreamWaitEvent(d->cs, after, 0));
  if (d->lo >= 0 || d->hi >= 0) {
    int64_t own0, ghost0, own1, ghost1, n;
    PMG_CALL(pmg_grid_halo_plane(d->g, c, 0, &own0, &ghost0, &n));
    PMG_CALL(pmg_grid_halo_plane(d->g, c, 1, &own1, &ghost1, &n));
    PMG_NCCL(d, d->api.GroupStart());
    if (d->lo >= 0) { /* my low plane -> neighbour's high ghost; neighbour's high plane -> my low ghost */
      PMG_NCCL(d, d->api.Send(y + own0, (size_t)n, PMG_NCCL_DOUBLE, d->lo, d->comm, d->cs));
      PMG_NCCL(d, d->api.Recv(y + ghost0, (size_t)n, PMG_NCCL_DOUBLE, d->lo, d->comm, d->cs));
    }
    if (d->hi >= 0) {
      PMG_NCCL(d, d->api.Send(y + own1, (size_t)n, PMG_NCCL_DOUBLE, d->hi, d->comm, d->cs));
      PMG_NCCL(d, d->api.Recv(y + ghost1, (size_t)n, PMG_NCCL_DOUBLE, d->hi, d->comm, d->cs));
    }
    PMG_NCCL(d, d->api.GroupEnd());
  }
  PMG_HIP(hipEventRecord(d->evX[c], d->cs));
  return PMG_SUCCESS;
}

/* The distributed sample loop: `its` samples of the sorgibbs/mcgibbs chain on this rank's slab (cvec vectors). */
pmg_status pmg_dist_sample_cvec(pmg_dist d, const double *b, double *y, int32_t its, int scaled, int sweep_type, uint64_t seed, uint64_t counter0, uint64_t *counter_out, void *stream)
{
  PMG_CHECK(d && b && y, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(its >= 0, PMG_ERR_ARG_OUTOFRANGE, "its = %d", its);
  PMG_CHECK(pmg_sweep_type_ok(sweep_type), PMG_ERR_SUP, "Only forward, backward and symmetric sweep supported");
  hipStream_t   s  = (hipStream_t)stream;
  const int32_t nz = d->nz;
  /* the caller's y has no ghost values yet: exchange both colours once the caller's prior work is done */
  PMG_HIP(hipEventRecord(d->evS, s));
  PMG_CALL(dist_exchange(d, 0, y, d->evS));
  PMG_CALL(dist_exchange(d, 1, y, d->evS));
  uint64_t ctr = counter0;
  for (int32_t it = 0; it < its; ++it) {
    const int ndir = sweep_type == PMG_SOR_SYMMETRIC_SWEEP ? 2 : 1;
    for (int q = 0; q < ndir; ++q) {
      const int dir = sweep_type == PMG_SOR_SYMMETRIC_SWEEP ? (q == 0 ? PMG_SOR_FORWARD_SWEEP : PMG_SOR_BACKWARD_SWEEP) : sweep_type;
      for (int cc = 0; cc < 2; ++cc) {
        const int c = dir == PMG_SOR_FORWARD_SWEEP ? cc : 1 - cc;
        PMG_HIP(hipStreamWaitEvent(s, d->evX[1 - c], 0)); /* colour c reads colour 1-c across the slab faces */
        PMG_CALL(pmg_grid_sweep_color_planes_cvec(d->g, c, 0, 1, 1, scaled, seed, ctr, b, y, s));
        if (nz > 1) PMG_CALL(pmg_grid_sweep_color_planes_cvec(d->g, c, nz - 1, 1, 1, scaled, seed, ctr, b, y, s));
        PMG_HIP(hipEventRecord(d->evB[c], s));
        PMG_CALL(dist_exchange(d, c, y, d->evB[c]));
        if (nz > 2) PMG_CALL(pmg_grid_sweep_color_planes_cvec(d->g, c, 1, nz - 2, 1, scaled, seed, ctr, b, y, s));
      }
      ++ctr;
    }
  }
  PMG_HIP(hipStreamWaitEvent(s, d->evX[0], 0));
  PMG_HIP(hipStreamWaitEvent(s, d->evX[1], 0));
  if (counter_out) *counter_out = ctr;
  return PMG_SUCCESS;
}
