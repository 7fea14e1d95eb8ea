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
 * Second transport, "ipc" (for the latency-bound strong-scaling regime where a 1 MB RCCL send/recv kernel costs
 * ~40 us): every rank owns one receive block in fine-grained device memory (flag words, one plane per colour and
 * side, a generic message area) exported with hipIpcGetMemHandle.  The sweep kernel of a colour takes the two face
 * planes first: their wavefronts wait, inside the kernel, for the flag words that announce the neighbours' planes of
 * the other colour, read those planes from the block, store the new planes into y AND into the neighbours' blocks over
 * xGMI, and the last face wavefront raises the neighbours' flag words; the other blocks of the same launch sweep the
 * interior.  One launch per colour, one stream, no events, no host rendezvous (interprocess events are capped at 32
 * records by the runtime and serviced by a host thread); the host only throttles itself to a few rounds ahead of its
 * device.
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
  int (*CommCount)(pmg_nccl_comm, int *);
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
  PMG_SYM(CommCount, "ncclCommCount");
  PMG_SYM(GetErrorString, "ncclGetErrorString");
#undef PMG_SYM
  return PMG_SUCCESS;
}

#define PMG_NCCL(d, expr) \
  do { \
    int pmg_r_ = (expr); \
    if (pmg_r_ != 0) return pmg_set_error(PMG_ERR_LIB, __FILE__, __LINE__, "%s: %s", #expr, (d)->api.GetErrorString(pmg_r_)); \
  } while (0)

#define PMG_IPC_MAXRANKS 64
#define PMG_IPC_WINDOW 8
#define PMG_IPC_HDR 512 /* doubles reserved at the start of a receive block for the flag words */
/* flag words (uint64) at the start of a receive block: [c*2 + side] = round of the latest colour-c plane pushed into
   slot (c, side); [4 + side] = round of the latest generic message pushed into side `side` */
typedef struct {
  hipIpcMemHandle_t mem;   /* the rank's receive block */
  int64_t           plane; /* doubles per plane (must agree between neighbours) */
  int64_t           gcap;  /* doubles per generic message slot */
  int64_t           pooled; /* the block lives as long as its process: a peer may keep its mapping (ipc_open_cached) */
} pmg_ipc_blob;

struct pmg_dist_s {
  int           transport; /* 0 = RCCL, 1 = IPC peer stores + flag words */
  /* ipc: block = [PMG_IPC_HDR flag doubles][recv: 4 planes][grecv: 2 parities x 2 sides x gcap] */
  double       *block, *peer_block[2]; /* own / neighbours' blocks (side 0 = lo, 1 = hi) */
  int           block_pooled, peer_cached[2], all_cached[PMG_IPC_MAXRANKS]; /* process-wide pool / table of open handles: see ipc_pool_get */
  double       *recv, *peer_recv[2];
  double       *grecv, *peer_grecv[2];
  uint64_t      round[2];              /* pushes of colour c issued so far */
  uint64_t      ground;                /* generic exchanges issued so far  */
  int64_t       plane, gcap;
  hipEvent_t    evT[PMG_IPC_WINDOW];   /* ring: host run-ahead throttle */
  uint64_t      nthrottle;
  unsigned     *err_dev;               /* pinned host word (device-visible), set by a flag wait that gave up */
  unsigned     *xch_counter;           /* device: blocks of the push kernel that have finished */
  unsigned long long *spins_dev;       /* device: polls of halo flag words that were not yet raised (pmg_dist_describe) */
  double       *red_buf;               /* device: nranks x 4096 partial sums of pmg_dist_allreduce_sum */
  /* all-peer mappings (optional, pmg_dist_ipc_connect_all): single-step all-gather.  Every block has, behind the
     generic slots, two gather areas (parity) of gcap doubles; flag word 8 + src announces rank src's block */
  double       *all_block[PMG_IPC_MAXRANKS];
  double      **ag_dst_dev[2];         /* device arrays [nranks]: peers' gather areas of each parity */
  uint64_t    **ag_flag_dev;           /* device array [nranks]: peers' flag word 8 + my rank */
  uint64_t      aground;
  int           all_connected;
  pmg_grid      g;
  int           rank, nranks, lo, hi; /* z-neighbours (-1 = physical boundary); lo == hi == rank in loopback mode */
  int           loopback;
  pmg_rccl_api  api;
  pmg_nccl_comm comm;
  hipStream_t   cs;       /* communication stream */
  hipEvent_t    evB[2];   /* boundary planes of colour c swept */
  hipEvent_t    evX[2];   /* exchange of colour c complete     */
  hipEvent_t    evS;      /* caller's stream reached the call  */
  hipEvent_t    evG, evGx; /* generic exchange edges (RCCL)    */
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
  PMG_CHECK(out, PMG_ERR_ARG_NULL, "null argument"); /* g == NULL: a transport without a slab (pmg_dist_exchange / _allgather only) */
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
  memset(&L, 0, sizeof L);
  pmg_status st = g ? pmg_grid_get_kernel_layout(g, &L) : PMG_SUCCESS;
  d->nz         = L.nz;
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
  if (!st && (hipEventCreateWithFlags(&d->evG, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&d->evGx, hipEventDisableTiming) != hipSuccess)) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "event creation failed");
  d->gcap = (int64_t)1 << 40; /* RCCL moves data between the caller's buffers: no staging limit */
  if (st) {
    pmg_dist_destroy(&d);
    return st;
  }
  *out = d;
  return PMG_SUCCESS;
}

/* ---- IPC transport --------------------------------------------------------------------------------------- */

/* Receive blocks and the peers' mappings of them live as long as the process: a chain of objects in one process
   (bench.py: headline sampler, then three V-cycle lines) would otherwise free, allocate, export and re-open a block of
   the same size four times per rank, and freeing a block that a peer had mapped a moment ago has failed on this runtime
   before (pmg_dist_ipc_disconnect).  A block goes back to a pool and is handed out again -- the same handle, found in
   the peer's table of open handles; nothing is closed, nothing is freed; a few blocks of tens of MB per process.  Blocks
   beyond the pool are ordinary allocations, and peers do not keep their mappings (the blob says which kind it is).
   Not thread-safe, like the rest of the object's life cycle. */
#define PMG_IPC_POOL 16
#define PMG_IPC_MAPS 128
static struct {
  double *ptr;
  size_t  bytes;
  int     busy;
} g_ipc_pool[PMG_IPC_POOL];
static struct {
  hipIpcMemHandle_t h;
  void             *ptr;
} g_ipc_maps[PMG_IPC_MAPS];
static int g_ipc_nmaps;

static double *ipc_pool_get(size_t bytes, int *pooled)
{
  *pooled = 1;
  for (int q = 0; q < PMG_IPC_POOL; ++q)
    if (g_ipc_pool[q].ptr && !g_ipc_pool[q].busy && g_ipc_pool[q].bytes == bytes) {
      g_ipc_pool[q].busy = 1;
      return g_ipc_pool[q].ptr;
    }
  double *p = NULL;
  if (hipExtMallocWithFlags((void **)&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) return NULL;
  for (int q = 0; q < PMG_IPC_POOL; ++q)
    if (!g_ipc_pool[q].ptr) {
      g_ipc_pool[q].ptr   = p;
      g_ipc_pool[q].bytes = bytes;
      g_ipc_pool[q].busy  = 1;
      return p;
    }
  *pooled = 0; /* pool full: an ordinary allocation, freed with the object */
  return p;
}

static void ipc_pool_put(double *p, int pooled)
{
  if (!p) return;
  if (!pooled) {
    (void)hipFree(p);
    return;
  }
  for (int q = 0; q < PMG_IPC_POOL; ++q)
    if (g_ipc_pool[q].ptr == p) g_ipc_pool[q].busy = 0;
}

/* the mapping of a peer's block: opened once per handle and kept */
static hipError_t ipc_open_cached(void **ptr, hipIpcMemHandle_t h, int pooled, int *cached)
{
  for (int q = 0; q < g_ipc_nmaps && pooled; ++q)
    if (!memcmp(&g_ipc_maps[q].h, &h, sizeof h)) {
      *ptr    = g_ipc_maps[q].ptr;
      *cached = 1;
      return hipSuccess;
    }
  const hipError_t e = hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
  *cached            = 0;
  if (e == hipSuccess && pooled && g_ipc_nmaps < PMG_IPC_MAPS) { /* a block that its owner may free is never kept: its handle could come back for other memory */
    g_ipc_maps[g_ipc_nmaps].h   = h;
    g_ipc_maps[g_ipc_nmaps].ptr = *ptr;
    ++g_ipc_nmaps;
    *cached = 1;
  }
  return e;
}

pmg_status pmg_dist_create_ipc(pmg_grid g, int32_t rank, int32_t nranks, pmg_dist *out)
{
  PMG_CHECK(out, PMG_ERR_ARG_NULL, "null argument"); /* g == NULL: a transport without a slab */
  *out = NULL;
  PMG_CHECK(nranks >= 1 && nranks <= PMG_IPC_MAXRANKS && rank >= 0 && rank < nranks, PMG_ERR_ARG_OUTOFRANGE, "rank %d of %d", rank, nranks);
  pmg_dist d = (pmg_dist)calloc(1, sizeof *d);
  PMG_CHECK(d, PMG_ERR_MEM, "out of host memory");
  d->transport = 1;
  d->g         = g;
  d->rank      = rank;
  d->nranks    = nranks;
  d->lo        = rank > 0 ? rank - 1 : -1;
  d->hi        = rank < nranks - 1 ? rank + 1 : -1;
  pmgk_grid_layout L;
  memset(&L, 0, sizeof L);
  pmg_status st = g ? pmg_grid_get_kernel_layout(g, &L) : PMG_SUCCESS;
  d->nz         = L.nz;
  d->plane      = L.sp;
  d->gcap             = 2 * d->plane > ((int64_t)1 << 19) ? 2 * d->plane : ((int64_t)1 << 19); /* two colour planes of the fine level, or 4 MB */
  const size_t bytes  = sizeof(double) * (size_t)(PMG_IPC_HDR + 4 * d->plane + 6 * d->gcap); /* flags, colour planes, 4 generic slots, 2 gather areas */
  /* fine-grained device memory: flag words and planes are written by a PEER device while this device's kernels poll
     and read them IN THE SAME KERNEL, so the block must be coherent at system scope without cache maintenance (what
     RCCL uses for its peer-written buffers) */
  if (!st && !(d->block = ipc_pool_get(bytes, &d->block_pooled))) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "hipExtMallocWithFlags(fine-grained) of the halo receive block failed");
  /* the flag words are sequence numbers that only grow: a block handed out again after an earlier object of this
     process still holds that object's (large) numbers, and every wait of the new object would pass before its data has
     arrived -- or a neighbour's small number would replace a large one under a waiting wavefront.  Zero the block before
     anybody can have its handle (export comes later, behind the launcher's barrier). */
  if (!st && (hipMemset(d->block, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "clearing the halo receive block failed");
  d->recv  = d->block ? d->block + PMG_IPC_HDR : NULL;
  d->grecv = d->block ? d->recv + 4 * d->plane : NULL;
  if (!st && hipStreamCreateWithFlags(&d->cs, hipStreamNonBlocking) != hipSuccess) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "stream creation failed");
  for (int c = 0; c < 2 && !st; ++c) {
    if (hipEventCreateWithFlags(&d->evB[c], hipEventDisableTiming) != hipSuccess) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "event creation failed");
  }
  if (!st && hipEventCreateWithFlags(&d->evS, hipEventDisableTiming) != hipSuccess) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "event creation failed");
  for (int q = 0; q < PMG_IPC_WINDOW && !st; ++q)
    if (hipEventCreateWithFlags(&d->evT[q], hipEventDisableTiming) != hipSuccess) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "event creation failed");
  if (!st && hipHostMalloc((void **)&d->err_dev, 8 * sizeof(unsigned), hipHostMallocMapped) != hipSuccess) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "hipHostMalloc failed");
  if (!st) memset(d->err_dev, 0, 8 * sizeof(unsigned));
  if (!st) st = pmg_dev_alloc((void **)&d->xch_counter, 2 * sizeof(unsigned)); /* [0] generic exchange, [1] face wavefronts */
  if (!st && hipMemset(d->xch_counter, 0, 2 * sizeof(unsigned)) != hipSuccess) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "memset failed");
  if (!st) st = pmg_dev_alloc((void **)&d->spins_dev, sizeof(unsigned long long));
  if (!st && hipMemset(d->spins_dev, 0, sizeof(unsigned long long)) != hipSuccess) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "memset failed");
  if (st) {
    pmg_dist_destroy(&d);
    return st;
  }
  *out = d;
  return PMG_SUCCESS;
}

pmg_status pmg_dist_ipc_blob_bytes(int32_t *bytes)
{
  PMG_CHECK(bytes, PMG_ERR_ARG_NULL, "null argument");
  *bytes = (int32_t)sizeof(pmg_ipc_blob);
  return PMG_SUCCESS;
}

pmg_status pmg_dist_ipc_export(pmg_dist d, void *blob)
{
  PMG_CHECK(d && blob && d->transport == 1, PMG_ERR_ARG_WRONG, "not an IPC dist object");
  pmg_ipc_blob bl;
  memset(&bl, 0, sizeof bl);
  PMG_HIP(hipIpcGetMemHandle(&bl.mem, d->block));
  bl.plane  = d->plane;
  bl.gcap   = d->gcap;
  bl.pooled = d->block_pooled;
  memcpy(blob, &bl, sizeof bl);
  return PMG_SUCCESS;
}

/* blobs of the z-neighbours (NULL where there is none); all ranks must have exported before anyone connects */
pmg_status pmg_dist_ipc_connect(pmg_dist d, const void *blob_lo, const void *blob_hi)
{
  PMG_CHECK(d && d->transport == 1, PMG_ERR_ARG_WRONG, "not an IPC dist object");
  const void *blobs[2] = {blob_lo, blob_hi};
  const int   nb[2]    = {d->lo, d->hi};
  for (int side = 0; side < 2; ++side) {
    if (nb[side] < 0) continue;
    PMG_CHECK(blobs[side], PMG_ERR_ARG_NULL, "missing blob of neighbour rank %d", nb[side]);
    pmg_ipc_blob bl;
    memcpy(&bl, blobs[side], sizeof bl);
    PMG_CHECK(bl.plane == d->plane && bl.gcap == d->gcap, PMG_ERR_ARG_SIZ, "neighbour receive block (%lld, %lld) != (%lld, %lld)", (long long)bl.plane, (long long)bl.gcap, (long long)d->plane, (long long)d->gcap);
    PMG_HIP(ipc_open_cached((void **)&d->peer_block[side], bl.mem, bl.pooled != 0, &d->peer_cached[side]));
    d->peer_recv[side]  = d->peer_block[side] + PMG_IPC_HDR;
    d->peer_grecv[side] = d->peer_recv[side] + 4 * d->plane;
  }
  return PMG_SUCCESS;
}

/* optional: map EVERY rank's block (blobs[r], r = 0..nranks-1; own entry ignored) so that pmg_dist_allgather is one
   push + one wait instead of nranks - 1 rounds along the chain.  Call after pmg_dist_ipc_connect. */
pmg_status pmg_dist_ipc_connect_all(pmg_dist d, const void *const *blobs)
{
  PMG_CHECK(d && d->transport == 1 && blobs, PMG_ERR_ARG_WRONG, "not an IPC dist object");
  if (d->nranks == 1 || d->loopback) return PMG_SUCCESS;
  double   *dst[2][PMG_IPC_MAXRANKS];
  uint64_t *flg[PMG_IPC_MAXRANKS];
  for (int r = 0; r < d->nranks; ++r) {
    dst[0][r] = dst[1][r] = NULL;
    flg[r]                = NULL;
    if (r == d->rank) continue;
    if (r == d->lo) d->all_block[r] = d->peer_block[0];
    else if (r == d->hi) d->all_block[r] = d->peer_block[1];
    else {
      PMG_CHECK(blobs[r], PMG_ERR_ARG_NULL, "missing blob of rank %d", r);
      pmg_ipc_blob bl;
      memcpy(&bl, blobs[r], sizeof bl);
      PMG_CHECK(bl.plane == d->plane && bl.gcap == d->gcap, PMG_ERR_ARG_SIZ, "rank %d has a different receive block", r);
      PMG_HIP(ipc_open_cached((void **)&d->all_block[r], bl.mem, bl.pooled != 0, &d->all_cached[r]));
    }
    double *gather = d->all_block[r] + PMG_IPC_HDR + 4 * d->plane + 4 * d->gcap;
    dst[0][r]      = gather;
    dst[1][r]      = gather + d->gcap;
    flg[r]         = (uint64_t *)d->all_block[r] + 8 + d->rank;
  }
  PMG_CHECK(8 + d->nranks <= PMG_IPC_HDR, PMG_ERR_ARG_OUTOFRANGE, "too many ranks for the flag header");
  for (int q = 0; q < 2; ++q) PMG_CALL(pmg_dev_upload((void **)&d->ag_dst_dev[q], dst[q], sizeof(double *) * (size_t)d->nranks));
  PMG_CALL(pmg_dev_upload((void **)&d->ag_flag_dev, flg, sizeof(uint64_t *) * (size_t)d->nranks));
  d->all_connected = 1;
  return PMG_SUCCESS;
}

/* single rank as its own z-neighbour for the halo only (timing / smoke test on one GPU, like the RCCL loopback):
   the "peer" receive block is the rank's own, no IPC handle is opened */
pmg_status pmg_dist_ipc_connect_loopback(pmg_dist d)
{
  PMG_CHECK(d && d->transport == 1 && d->nranks == 1, PMG_ERR_ARG_WRONG, "loopback needs a single-rank IPC dist object");
  d->lo = d->hi = 0;
  d->loopback   = 1;
  for (int side = 0; side < 2; ++side) {
    d->peer_block[side] = d->block;
    d->peer_recv[side]  = d->recv;
    d->peer_grecv[side] = d->grecv;
  }
  return PMG_SUCCESS;
}

/* call once per round on the stream the round's work was queued on: blocks the host until the round issued
   PMG_IPC_WINDOW rounds ago has finished on the device (bounded queue depth, nothing more) */
static pmg_status ipc_throttle(pmg_dist d, hipStream_t s)
{
  const int slot = (int)(d->nthrottle % PMG_IPC_WINDOW);
  if (d->nthrottle >= PMG_IPC_WINDOW) PMG_HIP(hipEventSynchronize(d->evT[slot]));
  PMG_HIP(hipEventRecord(d->evT[slot], s));
  d->nthrottle += 1;
  return PMG_SUCCESS;
}

/* flag words: mine = (uint64_t *)block, the neighbours' = (uint64_t *)peer_block[side].  Index c*2 + slot for the
   colour planes (slot = the side of the RECEIVER the plane comes from), 4 + slot for generic messages. */
static uint64_t *flag_mine(pmg_dist d, int idx) { return (uint64_t *)d->block + idx; }
static uint64_t *flag_peer(pmg_dist d, int side, int idx) { return (side == 0 ? d->lo : d->hi) >= 0 ? (uint64_t *)d->peer_block[side] + idx : NULL; }

/* a flag wait of an EARLIER round gave up (lost neighbour): fail the next call instead of computing on */
static pmg_status ipc_check(pmg_dist d)
{
  volatile unsigned *e = (volatile unsigned *)d->err_dev;
  if (e[0] != 0 && getenv("PMG_IPC_DEBUG")) { /* my flag words as they are now: [0..3] colour planes, [4..5] generic, [8+r] all-gather */
    uint64_t w[16];
    if (hipMemcpy(w, d->block, sizeof w, hipMemcpyDeviceToHost) == hipSuccess)
      fprintf(stderr, "[pmg ipc debug] rank %d block %p flags: planes %llu %llu %llu %llu generic %llu %llu gather %llu %llu %llu %llu; rounds %llu %llu ground %llu\n", d->rank, (void *)d->block, (unsigned long long)w[0], (unsigned long long)w[1], (unsigned long long)w[2], (unsigned long long)w[3], (unsigned long long)w[4], (unsigned long long)w[5], (unsigned long long)w[8], (unsigned long long)w[9], (unsigned long long)w[10], (unsigned long long)w[11], (unsigned long long)d->round[0], (unsigned long long)d->round[1], (unsigned long long)d->ground);
  } /* [1..3]: wait site (0x10 face plane, 0x30+q generic message, 0x40+r all-gather block of rank r), sequence number expected / seen; all 0 if another wait of this rank gave up first */
  PMG_CHECK(e[0] == 0, PMG_ERR_LIB, "rank %d: a halo flag never arrived (neighbour lost?): wait site 0x%x, expected %u, seen %u, exchanges %llu, all-gathers %llu; my pushes that raised flags %u (last numbers 0x%x), counter fault %u", d->rank, e[1], e[2], e[3], (unsigned long long)d->ground, (unsigned long long)d->aground, e[4], e[5], e[6]);
  return PMG_SUCCESS;
}

/* One launch per colour on the caller's stream.  The sweep kernel takes the face planes first: their wavefronts wait
   (in the kernel) until the flag words say that the neighbours' planes of the other colour have landed in my block,
   read them from there, store the new planes into y and straight into the neighbours' blocks, and the last face
   wavefront raises the neighbours' flag words; meanwhile the remaining blocks sweep the interior planes.  The
   neighbours' planes a colour needs were pushed one whole colour pass earlier, so nothing waits in steady state; no
   second stream, no events, no host rendezvous.  Safe re-use of a receive slot: my faces of colour c overwrite the
   neighbour's slot c only after its push of colour 1-c has arrived, which it issued after reading slot c. */
static pmg_status ipc_sample(pmg_dist d, const double *b, double *y, int32_t its, int noisy, int scaled, int sweep_type, uint64_t seed, uint64_t counter0, uint64_t *counter_out, hipStream_t s)
{
  int64_t own, ghost, n;
  PMG_CALL(ipc_check(d));
  { /* the receive blocks hold nothing of this y yet: my boundary planes of both colours go over in one launch */
    pmgk_xch_args push;
    memset(&push, 0, sizeof push);
    for (int c = 0; c < 2; ++c) {
      d->round[c] += 1;
      for (int side = 0; side < 2; ++side) {
        if ((side == 0 ? d->lo : d->hi) < 0) continue;
        PMG_CALL(pmg_grid_halo_plane(d->g, c, side, &own, &ghost, &n));
        push.src[push.nseg]    = y + own;
        push.dst[push.nseg]    = d->peer_recv[side] + (int64_t)(c * 2 + (1 - side)) * d->plane;
        push.n[push.nseg++]    = n;
        push.flag[c * 2 + side]  = flag_peer(d, side, c * 2 + (1 - side));
        push.value[c * 2 + side] = d->round[c];
      }
    }
    PMG_KERNEL(pmgk_xch_push_dbg(&push, d->xch_counter, d->err_dev + 4, s));
  }
  uint64_t ctr = counter0;
  for (int32_t it = 0; it < its; ++it) {
    const int ndir = sweep_type == PMG_SOR_SYMMETRIC_SWEEP ? 2 : 1;
    for (int q = 0; q < ndir; ++q) {
      const int dir = sweep_type == PMG_SOR_SYMMETRIC_SWEEP ? (q == 0 ? PMG_SOR_FORWARD_SWEEP : PMG_SOR_BACKWARD_SWEEP) : sweep_type;
      for (int cc = 0; cc < 2; ++cc) {
        const int      c = dir == PMG_SOR_FORWARD_SWEEP ? cc : 1 - cc;
        pmgk_grid_halo h;
        memset(&h, 0, sizeof h);
        h.full = 1;
        h.glo  = d->lo >= 0 ? d->recv + (int64_t)((1 - c) * 2 + 0) * d->plane : NULL;
        h.ghi  = d->hi >= 0 ? d->recv + (int64_t)((1 - c) * 2 + 1) * d->plane : NULL;
        h.plo  = d->lo >= 0 ? d->peer_recv[0] + (int64_t)(c * 2 + 1) * d->plane : NULL;
        h.phi  = d->hi >= 0 ? d->peer_recv[1] + (int64_t)(c * 2 + 0) * d->plane : NULL;
        h.wlo  = d->lo >= 0 ? flag_mine(d, (1 - c) * 2 + 0) : NULL; /* colour c reads colour 1-c across the slab faces */
        h.whi  = d->hi >= 0 ? flag_mine(d, (1 - c) * 2 + 1) : NULL;
        h.wval = d->round[1 - c];
        d->round[c] += 1;
        h.slo     = flag_peer(d, 0, c * 2 + 1);
        h.shi     = flag_peer(d, 1, c * 2 + 0);
        h.sval    = d->round[c];
        h.counter = d->xch_counter + 1;
        h.err     = d->err_dev;
        h.spins   = d->spins_dev;
        PMG_CALL(pmg_grid_sweep_color_halo_cvec(d->g, c, noisy, scaled, seed, ctr, &h, b, y, s));
      }
      if ((ctr & 1) == 1) PMG_CALL(ipc_throttle(d, s)); /* an event every other sweep: each one costs a marker on the stream */
      ++ctr;
    }
  }
  { /* leave y self-contained: the latest neighbour planes go into its own ghost planes, one launch */
    pmgk_xch_args pull;
    memset(&pull, 0, sizeof pull);
    for (int c = 0; c < 2; ++c)
      for (int side = 0; side < 2; ++side) {
        if ((side == 0 ? d->lo : d->hi) < 0) continue;
        PMG_CALL(pmg_grid_halo_plane(d->g, c, side, &own, &ghost, &n));
        pull.src[pull.nseg]      = d->recv + (int64_t)(c * 2 + side) * d->plane;
        pull.dst[pull.nseg]      = y + ghost;
        pull.n[pull.nseg++]      = n;
        pull.flag[c * 2 + side]  = flag_mine(d, c * 2 + side);
        pull.value[c * 2 + side] = d->round[c];
      }
    PMG_KERNEL(pmgk_xch_pull(&pull, d->err_dev, s));
  }
  if (counter_out) *counter_out = ctr;
  return PMG_SUCCESS;
}

/* Unmap the peers' receive blocks.  Tear-down of the ipc transport is a three-step protocol when more objects are to be
   created afterwards: every rank disconnects, the caller runs a barrier, then every rank destroys (which frees its own
   block).  Freeing a block that a peer still has mapped and exporting a fresh one right away made hipIpcGetMemHandle fail
   with "invalid argument" on this runtime (seen with 4 ranks; pmg_dist_destroy alone still unmaps, without the barrier). */
pmg_status pmg_dist_ipc_disconnect(pmg_dist d)
{
  PMG_CHECK(d, PMG_ERR_ARG_NULL, "null dist object");
  if (d->transport != 1) return PMG_SUCCESS;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < d->nranks && d->all_connected; ++r) {
    if (d->all_block[r] && r != d->lo && r != d->hi && r != d->rank && !d->all_cached[r]) (void)hipIpcCloseMemHandle(d->all_block[r]);
    d->all_block[r] = NULL;
  }
  d->all_connected = 0;
  for (int side = 0; side < 2 && !d->loopback; ++side) {
    if (d->peer_block[side] && !d->peer_cached[side]) (void)hipIpcCloseMemHandle(d->peer_block[side]);
    d->peer_block[side] = NULL;
    d->peer_recv[side] = d->peer_grecv[side] = NULL;
  }
  return PMG_SUCCESS;
}

pmg_status pmg_dist_destroy(pmg_dist *dp)
{
  if (!dp || !*dp) return PMG_SUCCESS;
  pmg_dist d = *dp;
  if (d->transport == 1) {
    (void)hipDeviceSynchronize();
    for (int side = 0; side < 2 && !d->loopback; ++side)
      if (d->peer_block[side] && !d->peer_cached[side]) (void)hipIpcCloseMemHandle(d->peer_block[side]);
    for (int q = 0; q < PMG_IPC_WINDOW; ++q)
      if (d->evT[q]) (void)hipEventDestroy(d->evT[q]);
    for (int r = 0; r < d->nranks && d->all_connected; ++r)
      if (d->all_block[r] && r != d->lo && r != d->hi && r != d->rank && !d->all_cached[r]) (void)hipIpcCloseMemHandle(d->all_block[r]);
    pmg_dev_free(d->ag_dst_dev[0]);
    pmg_dev_free(d->ag_dst_dev[1]);
    pmg_dev_free(d->ag_flag_dev);
    if (d->err_dev) (void)hipHostFree(d->err_dev);
    pmg_dev_free(d->xch_counter);
    pmg_dev_free(d->spins_dev);
    ipc_pool_put(d->block, d->block_pooled);
  }
  pmg_dev_free(d->red_buf);
  if (d->comm && d->api.CommDestroy) d->api.CommDestroy(d->comm);
  for (int c = 0; c < 2; ++c) {
    if (d->evB[c]) (void)hipEventDestroy(d->evB[c]);
    if (d->evX[c]) (void)hipEventDestroy(d->evX[c]);
  }
  if (d->evS) (void)hipEventDestroy(d->evS);
  if (d->evG) (void)hipEventDestroy(d->evG);
  if (d->evGx) (void)hipEventDestroy(d->evGx);
  if (d->cs) (void)hipStreamDestroy(d->cs);
  free(d);
  *dp = NULL;
  return PMG_SUCCESS;
}

/* ---- generic neighbour exchange ---------------------------------------------------------------------------------
   One round trip with both z-neighbours on the caller's stream: up to PMG_XCH_MAXSEG contiguous segments per side.
   What I send to my low neighbour arrives in ITS high receive buffers and vice versa; the segment sizes of a pair
   must agree (sender's n_send = receiver's n_recv), sides without a neighbour are skipped, zero-length segments are
   allowed.  Every rank must make the same sequence of calls (the IPC transport counts rounds).
   RCCL: one grouped ncclSend/ncclRecv on the communication stream, between two event edges.
   IPC : the segments are copied straight into the neighbour's generic message slots (two parities, so that round
         r + 2 may be pushed while the neighbour still holds round r: I push r + 2 only after its round r + 1 arrived,
         which it sent after copying round r out -- its stream is in order), its flag word is raised to the round
         number, my stream waits for my flag word and copies the message out. */
pmg_status pmg_dist_exchange(pmg_dist d, int nseg, const double *const *send_lo, const int64_t *nsend_lo, double *const *recv_lo, const int64_t *nrecv_lo, const double *const *send_hi, const int64_t *nsend_hi, double *const *recv_hi, const int64_t *nrecv_hi, void *stream)
{
  PMG_CHECK(d, PMG_ERR_ARG_NULL, "null dist object");
  PMG_CHECK(nseg >= 0 && nseg <= PMG_XCH_MAXSEG, PMG_ERR_ARG_OUTOFRANGE, "nseg = %d", nseg);
  hipStream_t s = (hipStream_t)stream;
  if (d->lo < 0 && d->hi < 0) return PMG_SUCCESS;
  const double *const *snd[2] = {send_lo, send_hi};
  double *const       *rcv[2] = {recv_lo, recv_hi};
  const int64_t       *ns[2] = {nsend_lo, nsend_hi}, *nr[2] = {nrecv_lo, nrecv_hi};
  const int            nb[2] = {d->lo, d->hi};
  if (d->transport == 0) {
    PMG_HIP(hipEventRecord(d->evG, s));
    PMG_HIP(hipStreamWaitEvent(d->cs, d->evG, 0));
    PMG_NCCL(d, d->api.GroupStart());
    for (int side = 0; side < 2; ++side) {
      if (nb[side] < 0) continue;
      for (int q = 0; q < nseg; ++q) {
        if (ns[side][q] > 0) PMG_NCCL(d, d->api.Send(snd[side][q], (size_t)ns[side][q], PMG_NCCL_DOUBLE, nb[side], d->comm, d->cs));
        if (nr[side][q] > 0) PMG_NCCL(d, d->api.Recv(rcv[side][q], (size_t)nr[side][q], PMG_NCCL_DOUBLE, nb[side], d->comm, d->cs));
      }
    }
    PMG_NCCL(d, d->api.GroupEnd());
    PMG_HIP(hipEventRecord(d->evGx, d->cs));
    PMG_HIP(hipStreamWaitEvent(s, d->evGx, 0));
    return PMG_SUCCESS;
  }
  /* IPC: two launches on the caller's stream -- push my segments into the neighbours' slots of parity p and raise
     their flags; wait for mine and copy the messages out */
  PMG_CALL(ipc_check(d));
  const int p = (int)(d->ground & 1);
  d->ground += 1;
  pmgk_xch_args push, pull;
  memset(&push, 0, sizeof push);
  memset(&pull, 0, sizeof pull);
  for (int side = 0; side < 2; ++side) {
    if (nb[side] < 0) continue;
    int64_t off = 0, roff = 0;
    for (int q = 0; q < nseg; ++q) {
      PMG_CHECK(off + ns[side][q] <= d->gcap && roff + nr[side][q] <= d->gcap, PMG_ERR_ARG_SIZ, "exchange of %lld doubles exceeds the receive block (%lld)", (long long)(off + ns[side][q]), (long long)d->gcap);
      if (ns[side][q] > 0) {
        push.src[push.nseg] = snd[side][q];
        push.dst[push.nseg] = d->peer_grecv[side] + (int64_t)(p * 2 + (1 - side)) * d->gcap + off;
        push.n[push.nseg++] = ns[side][q];
      }
      if (nr[side][q] > 0) {
        pull.src[pull.nseg] = d->grecv + (int64_t)(p * 2 + side) * d->gcap + roff;
        pull.dst[pull.nseg] = rcv[side][q];
        pull.n[pull.nseg++] = nr[side][q];
      }
      off += ns[side][q];
      roff += nr[side][q];
    }
    push.flag[side]  = flag_peer(d, side, 4 + (1 - side));
    pull.flag[side]  = flag_mine(d, 4 + side);
    push.value[side] = pull.value[side] = d->ground;
  }
  PMG_KERNEL(pmgk_xch_push_dbg(&push, d->xch_counter, d->err_dev + 4, s));
  PMG_KERNEL(pmgk_xch_pull(&pull, d->err_dev, s));
  return ipc_throttle(d, s);
}

/* every rank ends up with all blocks: block r = counts[r] doubles at buf + offsets[r], rank r owns block r.
   RCCL: one group of sends to / receives from every other rank.  IPC: with all-peer mappings
   (pmg_dist_ipc_connect_all) one push of my block into every rank's gather area + one wait + one local copy;
   with neighbour links only, nranks - 1 rounds of passing blocks along the chain in both directions. */
pmg_status pmg_dist_allgather(pmg_dist d, double *buf, const int64_t *offsets, const int64_t *counts, void *stream)
{
  PMG_CHECK(d && buf && offsets && counts, PMG_ERR_ARG_NULL, "null argument");
  hipStream_t s = (hipStream_t)stream;
  if (d->nranks == 1) return PMG_SUCCESS;
  if (d->transport == 0) {
    PMG_HIP(hipEventRecord(d->evG, s));
    PMG_HIP(hipStreamWaitEvent(d->cs, d->evG, 0));
    PMG_NCCL(d, d->api.GroupStart());
    for (int r = 0; r < d->nranks; ++r) {
      if (r == d->rank) continue;
      if (counts[d->rank] > 0) PMG_NCCL(d, d->api.Send(buf + offsets[d->rank], (size_t)counts[d->rank], PMG_NCCL_DOUBLE, r, d->comm, d->cs));
      if (counts[r] > 0) PMG_NCCL(d, d->api.Recv(buf + offsets[r], (size_t)counts[r], PMG_NCCL_DOUBLE, r, d->comm, d->cs));
    }
    PMG_NCCL(d, d->api.GroupEnd());
    PMG_HIP(hipEventRecord(d->evGx, d->cs));
    PMG_HIP(hipStreamWaitEvent(s, d->evGx, 0));
    return PMG_SUCCESS;
  }
  if (d->all_connected) { /* one push into every rank's gather area, one wait, one local copy out */
    int64_t base = offsets[0], end = offsets[0] + counts[0];
    int     tiled = 1;
    for (int r = 1; r < d->nranks; ++r) {
      if (offsets[r] != end) tiled = 0; /* the blocks must tile one contiguous range in rank order (plane ranges do) */
      end = offsets[r] + counts[r];
    }
    if (tiled && end - base <= d->gcap) {
      PMG_CALL(ipc_check(d));
      const int p = (int)(d->aground & 1);
      d->aground += 1;
      const double *mine = d->block + PMG_IPC_HDR + 4 * d->plane + 4 * d->gcap + (int64_t)p * d->gcap; /* my gather area */
      PMG_KERNEL(pmgk_allgather_push(d->nranks, d->rank, buf + offsets[d->rank], counts[d->rank], d->ag_dst_dev[p], offsets[d->rank] - base, d->ag_flag_dev, d->aground, d->xch_counter, s));
      PMG_KERNEL(pmgk_allgather_wait(d->nranks, d->rank, (const uint64_t *)d->block + 8, d->aground, d->err_dev, s));
      pmgk_xch_args pull;
      memset(&pull, 0, sizeof pull);
      const int64_t lo_n = offsets[d->rank] - base, hi_0 = offsets[d->rank] + counts[d->rank];
      if (lo_n > 0) {
        pull.src[pull.nseg] = mine;
        pull.dst[pull.nseg] = buf + base;
        pull.n[pull.nseg++] = lo_n;
      }
      if (end - hi_0 > 0) {
        pull.src[pull.nseg] = mine + (hi_0 - base);
        pull.dst[pull.nseg] = buf + hi_0;
        pull.n[pull.nseg++] = end - hi_0;
      }
      PMG_KERNEL(pmgk_xch_pull(&pull, d->err_dev, s));
      return ipc_throttle(d, s);
    }
  }
  for (int t = 1; t < d->nranks; ++t) {
    /* upward: I pass block (rank - t + 1) to hi and receive block (rank - t) from lo; downward mirrored */
    const int     up_s = d->rank - t + 1, up_r = d->rank - t, dn_s = d->rank + t - 1, dn_r = d->rank + t;
    const double *shi = NULL, *slo = NULL;
    double       *rlo = NULL, *rhi = NULL;
    int64_t       nshi = 0, nslo = 0, nrlo = 0, nrhi = 0;
    if (d->hi >= 0 && up_s >= 0) { shi = buf + offsets[up_s]; nshi = counts[up_s]; }
    if (d->lo >= 0 && up_r >= 0) { rlo = buf + offsets[up_r]; nrlo = counts[up_r]; }
    if (d->lo >= 0 && dn_s < d->nranks) { slo = buf + offsets[dn_s]; nslo = counts[dn_s]; }
    if (d->hi >= 0 && dn_r < d->nranks) { rhi = buf + offsets[dn_r]; nrhi = counts[dn_r]; }
    PMG_CALL(pmg_dist_exchange(d, 1, &slo, &nslo, &rlo, &nrlo, &shi, &nshi, &rhi, &nrhi, stream));
  }
  return PMG_SUCCESS;
}

/* after the stream has been synchronised: did every halo flag arrive?  (a wait that gives up marks the object; all
   later calls fail) */
pmg_status pmg_dist_check(pmg_dist d)
{
  PMG_CHECK(d, PMG_ERR_ARG_NULL, "null dist object");
  return d->transport == 1 ? ipc_check(d) : PMG_SUCCESS;
}

/* What a multi-GPU run should say about itself (bench.py --gpus N and examples/pmg_bench -ranks N print it per rank, so that
   the first run on more than one physical GPU can be read whatever it shows): the device this rank computes on, its PCI bus
   id, whether the device can access the devices `lo_device` / `hi_device` of its z-neighbours as peers (-1: no such
   neighbour; the caller knows the neighbours' device indices -- the local ranks under torchrun), the transport, the size of
   the RCCL communicator as RCCL itself counts it (0 for the ipc transport), and the polls of halo flag words that found
   them not yet raised since the object was created (ipc; synchronises the device). */
pmg_status pmg_dist_describe(pmg_dist d, int32_t lo_device, int32_t hi_device, pmg_dist_description *out)
{
  PMG_CHECK(d && out, PMG_ERR_ARG_NULL, "null argument");
  memset(out, 0, sizeof *out);
  int dev = -1;
  PMG_HIP(hipGetDevice(&dev));
  out->device = dev;
  if (hipDeviceGetPCIBusId(out->pci_bus_id, (int)sizeof out->pci_bus_id, dev) != hipSuccess) snprintf(out->pci_bus_id, sizeof out->pci_bus_id, "unknown");
  const int32_t peer[2] = {lo_device, hi_device};
  for (int s = 0; s < 2; ++s) {
    int can = -1;
    if (peer[s] >= 0) {
      if (peer[s] == dev) can = 1; /* ranks sharing a device (rehearsal) */
      else if (hipDeviceCanAccessPeer(&can, dev, peer[s]) != hipSuccess) can = -1;
    }
    out->peer_access[s] = can;
  }
  out->rank      = d->rank;
  out->nranks    = d->nranks;
  out->neighbour[0] = d->lo, out->neighbour[1] = d->hi;
  snprintf(out->transport, sizeof out->transport, "%s", d->transport == 1 ? "ipc" : "rccl");
  if (d->transport == 0 && d->comm && d->api.CommCount) {
    int n = 0;
    if (d->api.CommCount(d->comm, &n) == 0) out->rccl_comm_count = n;
  }
  if (d->spins_dev) {
    unsigned long long v = 0;
    PMG_HIP(hipDeviceSynchronize());
    PMG_HIP(hipMemcpy(&v, d->spins_dev, sizeof v, hipMemcpyDeviceToHost));
    out->halo_wait_polls = (uint64_t)v;
  }
  return PMG_SUCCESS;
}

pmg_status pmg_dist_get_info(pmg_dist d, int32_t *rank, int32_t *nranks, int64_t *capacity)
{
  PMG_CHECK(d, PMG_ERR_ARG_NULL, "null dist object");
  if (rank) *rank = d->rank;
  if (nranks) *nranks = d->nranks;
  if (capacity) *capacity = d->gcap;
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

/* `its` sweeps on this rank's slab (cvec vectors), noisy (the sample loop) or deterministic (MCSORApply) */
static pmg_status dist_sweeps(pmg_dist d, const double *b, double *y, int32_t its, int noisy, int scaled, int sweep_type, uint64_t seed, uint64_t counter0, uint64_t *counter_out, void *stream)
{
  PMG_CHECK(d && b && y, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(d->g, PMG_ERR_ARG_WRONGSTATE, "this transport was created without a grid slab");
  PMG_CHECK(its >= 0, PMG_ERR_ARG_OUTOFRANGE, "its = %d", its);
  PMG_CHECK(pmg_sweep_type_ok(sweep_type), PMG_ERR_SUP, "Only forward, backward and symmetric sweep supported");
  hipStream_t   s  = (hipStream_t)stream;
  const int32_t nz = d->nz;
  if (d->transport == 1) return ipc_sample(d, b, y, its, noisy, scaled, sweep_type, seed, counter0, counter_out, s);
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
        PMG_CALL(pmg_grid_sweep_color_faces_cvec(d->g, c, noisy, scaled, seed, ctr, NULL, b, y, s)); /* planes 0 and nz-1, one launch */
        PMG_HIP(hipEventRecord(d->evB[c], s));
        PMG_CALL(dist_exchange(d, c, y, d->evB[c]));
        if (nz > 2) PMG_CALL(pmg_grid_sweep_color_planes_cvec(d->g, c, 1, nz - 2, noisy, scaled, seed, ctr, b, y, s));
      }
      ++ctr;
    }
  }
  PMG_HIP(hipStreamWaitEvent(s, d->evX[0], 0));
  PMG_HIP(hipStreamWaitEvent(s, d->evX[1], 0));
  if (counter_out) *counter_out = ctr;
  return PMG_SUCCESS;
}

/* The distributed sample loop: `its` samples of the sorgibbs/mcgibbs chain on this rank's slab (cvec vectors). */
pmg_status pmg_dist_sample_cvec(pmg_dist d, const double *b, double *y, int32_t its, int scaled, int sweep_type, uint64_t seed, uint64_t counter0, uint64_t *counter_out, void *stream)
{
  return dist_sweeps(d, b, y, its, 1, scaled, sweep_type, seed, counter0, counter_out, stream);
}

/* MCSORApply on the slabs: one deterministic sweep of the given type */
pmg_status pmg_dist_apply_cvec(pmg_dist d, const double *b, double *y, int sweep_type, void *stream)
{
  return dist_sweeps(d, b, y, 1, 0, 0, sweep_type, 0, 0, NULL, stream);
}

/* vals[0..count) (device) <- their sum over all ranks, formed in rank order on every rank (identical bits
   everywhere); count <= 4096.  One all-gather of the partial sums + a small kernel. */
pmg_status pmg_dist_allreduce_sum(pmg_dist d, double *vals_dev, int32_t count, void *stream)
{
  PMG_CHECK(d && vals_dev, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(count >= 0 && count <= 4096, PMG_ERR_ARG_OUTOFRANGE, "count = %d", count);
  if (d->nranks == 1 || count == 0) return PMG_SUCCESS;
  if (!d->red_buf) PMG_CALL(pmg_dev_alloc((void **)&d->red_buf, sizeof(double) * 4096 * (size_t)d->nranks));
  int64_t offs[PMG_IPC_MAXRANKS], cnts[PMG_IPC_MAXRANKS];
  PMG_CHECK(d->nranks <= PMG_IPC_MAXRANKS, PMG_ERR_ARG_OUTOFRANGE, "too many ranks");
  for (int r = 0; r < d->nranks; ++r) {
    offs[r] = (int64_t)r * count;
    cnts[r] = count;
  }
  PMG_HIP(hipMemcpyAsync(d->red_buf + offs[d->rank], vals_dev, sizeof(double) * (size_t)count, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  PMG_CALL(pmg_dist_allgather(d, d->red_buf, offs, cnts, stream));
  PMG_KERNEL(pmgk_sum_rows(d->nranks, count, d->red_buf, vals_dev, stream));
  return PMG_SUCCESS;
}
