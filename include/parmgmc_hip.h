/*
 * parmgmc_hip.h -- C-ABI of the MI355X (gfx950) implementation of ParMGMC's Gibbs/SOR hot path.
 *
 * Drop-in boundary.  Every entry point replaces one interface of the reference (nilsfriess/ParMGMC; all
 * file:line citations are relative to that repository) on RAW arrays instead of PETSc objects: plain
 * pointers and sizes, `int` status codes, no PETSc, no torch, no C++ types.  A PETSc build binds these
 * through the adapter shown in INTEGRATION.md (MatSeqAIJGetCSRAndMemType / VecGetArray... hand over exactly
 * the arrays named here, reference src/mc_sor.c:250-255).
 *
 * Conventions
 *   - status: 0 = success (PETSC_SUCCESS); non-zero values use PETSc's PetscErrorCode numbers so that the
 *     adapter can return them unchanged (values from petscerror.h, PETSc source not present here).
 *     Nothing throws, nothing aborts, nothing is printed; pmg_last_error_string() gives the message.
 *   - "host" pointers are ordinary memory and are only read during the call (borrowed, never kept, like
 *     the reference's borrowed Mat arrays, src/pc_sorgibbs.c:35-38); "dev" pointers are HBM addresses of
 *     the current HIP device (hipMalloc / torch tensor.data_ptr()).
 *   - indices: the reference builds against either PetscInt width (include/parmgmc/parmgmc.h:18-24); entry points
 *     ending in _idx take `idx_width` = sizeof(PetscInt)*8 (32 or 64) with the index arrays as `const void *`, the
 *     others are their 32-bit forms.  Inside, rows and stored entries per device are 32-bit; a 64-bit matrix whose
 *     local sizes do not fit fails with PETSC_ERR_ARG_OUTOFRANGE.  Scalars are double (PetscScalar = PetscReal).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  All device work is enqueued on
 *     it and NOT synchronised: calls return as soon as the work is queued unless stated otherwise.
 *   - objects are not thread safe (the reference is single threaded per rank, src/parmgmc.c:38-42).
 */
#ifndef PARMGMC_HIP_H
#define PARMGMC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------------ */
/* status codes (PetscErrorCode numbering)                                                                 */
/* ------------------------------------------------------------------------------------------------------ */
typedef int pmg_status;
#define PMG_SUCCESS 0
#define PMG_ERR_MEM 55            /* PETSC_ERR_MEM            */
#define PMG_ERR_SUP 56            /* PETSC_ERR_SUP            */
#define PMG_ERR_ORDER 58          /* PETSC_ERR_ORDER          */
#define PMG_ERR_ARG_SIZ 60        /* PETSC_ERR_ARG_SIZ        */
#define PMG_ERR_ARG_WRONG 62      /* PETSC_ERR_ARG_WRONG      */
#define PMG_ERR_ARG_OUTOFRANGE 63 /* PETSC_ERR_ARG_OUTOFRANGE */
#define PMG_ERR_ARG_WRONGSTATE 73 /* PETSC_ERR_ARG_WRONGSTATE */
#define PMG_ERR_LIB 76            /* PETSC_ERR_LIB            */
#define PMG_ERR_PLIB 77           /* PETSC_ERR_PLIB           */
#define PMG_ERR_MAT_CH_ZRPVT 81   /* PETSC_ERR_MAT_CH_ZRPVT   */
#define PMG_ERR_ARG_NULL 85       /* PETSC_ERR_ARG_NULL       */
#define PMG_ERR_ARG_UNKNOWN_TYPE 86 /* PETSC_ERR_ARG_UNKNOWN_TYPE */
#define PMG_ERR_GPU 97            /* PETSC_ERR_GPU            */

/* Message of the last failing call on this thread ("" if none). */
const char *pmg_last_error_string(void);
/* 1 if the library is emitting ROCTx ranges named like the reference's log events "MulticolSOR" / "VecSetRandN"
   (src/parmgmc.c:118-127): the marker library is in the process (rocprofv3 --marker-trace) or PMG_TRACE=1 */
int pmg_trace_enabled(void);
/* Library version "major.minor.patch" and the GPU architecture it was compiled for ("gfx950"). */
const char *pmg_version(void);
const char *pmg_gpu_arch(void);

/* MatSORType values accepted by the reference (src/mc_sor.c:427): PETSc's SOR_FORWARD_SWEEP = 1,
   SOR_BACKWARD_SWEEP = 2, SOR_SYMMETRIC_SWEEP = 3.  Anything else -> PMG_ERR_SUP. */
#define PMG_SOR_FORWARD_SWEEP 1
#define PMG_SOR_BACKWARD_SWEEP 2
#define PMG_SOR_SYMMETRIC_SWEEP 3

/* Colouring rules (reference: one colour in serial, src/mc_sor.c:397-410; PETSc JP in parallel, :383-395).
   A GPU sweep needs a VALID distance-1 colouring, so the serial "one colour" rule is replaced by LEXLEVELS,
   which reproduces the same lexicographic Gauss-Seidel result update for update. */
#define PMG_COLORING_GREEDY 0    /* first-fit in natural row order (deterministic; replaces randomised JP)   */
#define PMG_COLORING_LEXLEVELS 1 /* dependency levels of the natural-order sweep == serial reference / MatSOR */
#define PMG_COLORING_USER 2      /* caller-supplied ISColoringValue array, validated                          */
#define PMG_COLORING_ITERATED 3  /* first-fit, then first-fit once more with the classes visited last class first (one round of
                                    iterated greedy): never more classes than first-fit, 5 instead of 6 on the P1 matrices of
                                    lshape.msh and their Galerkin levels, i.e. one dependent launch fewer per sweep            */

/* ------------------------------------------------------------------------------------------------------ */
/* MCSOR on an assembled AIJ matrix: replaces include/parmgmc/mc_sor.h:17-30                               */
/* ------------------------------------------------------------------------------------------------------ */
typedef struct pmg_mcsor_s *pmg_mcsor; /* opaque, like `struct _MCSOR { void *ctx; }` (mc_sor.h:17-19) */

/* MCSORCreate(Mat, MCSOR*) (src/mc_sor.c:618-642) for a MATSEQAIJ given as host CSR (the arrays of
   MatSeqAIJGetCSRAndMemType, src/mc_sor.c:250).  Copies nothing yet; the arrays must stay valid until
   pmg_mcsor_setup returns.  omega = 1, sweep = forward, colouring = GREEDY. */
pmg_status pmg_mcsor_create_csr(int32_t n, const int32_t *rowptr_host, const int32_t *colidx_host, const double *vals_host, pmg_mcsor *mc);
/* the same for either PetscInt width (idx_width = 32 | 64); 64-bit arrays are narrowed to checked 32-bit copies at once,
   so they need not outlive the call */
pmg_status pmg_mcsor_create_csr_idx(int64_t n, const void *rowptr_host, const void *colidx_host, const double *vals_host, int idx_width, pmg_mcsor *mc);
/* Choose the colouring before setup.  user_colors_host (n entries, colours 0..ncolors-1) only for USER. */
pmg_status pmg_mcsor_set_coloring(pmg_mcsor mc, int rule, const int32_t *user_colors_host);
/* MCSORSetUp (src/mc_sor.c:553-605): diagonal pointers (:126-150), colouring (:441-454), idiag (:114-124);
   builds the colour-partitioned sliced-ELL copy and uploads it.  Synchronous. */
pmg_status pmg_mcsor_setup(pmg_mcsor mc);
/* MCSORSetOmega (src/mc_sor.c:412-420); takes effect at the next apply (omega_changed, :222). */
pmg_status pmg_mcsor_set_omega(pmg_mcsor mc, double omega);
/* MCSORSetSweepType / MCSORGetSweepType (src/mc_sor.c:422-439). */
pmg_status pmg_mcsor_set_sweep_type(pmg_mcsor mc, int type);
pmg_status pmg_mcsor_get_sweep_type(pmg_mcsor mc, int *type);
/* MCSORGetNumColors (src/mc_sor.c:607-616) and MCSORGetISColoring (:92-99; writes n colour values). */
pmg_status pmg_mcsor_get_num_colors(pmg_mcsor mc, int32_t *ncolors);
pmg_status pmg_mcsor_get_coloring(pmg_mcsor mc, int32_t *colors_host);
/* MCSORApply(mc, b, y) (src/mc_sor.c:216-239): one deterministic sweep of the current type, y in/out,
   vectors in the matrix's own (natural) row numbering on the device. */
pmg_status pmg_mcsor_apply(pmg_mcsor mc, const double *b_dev, double *y_dev, void *stream);
/* The sample loop of PCApplyRichardson_MulticolorGibbs (src/pc_mcgibbs.c:155-188) without the callback:
   `its` times { w = xi*sqrtdiag + b ; sweep }, symmetric = forward + backward with two fresh draws
   (:172-181).  scaled != 0: sqrtdiag carries sqrt((2-omega)/omega) (mcgibbs, :142-153); scaled == 0 is
   PCSORGibbsSample (src/pc_sorgibbs.c:76-103, requires omega == 1).  Draw d of the call uses noise
   counter `counter0 + d`; returns the next free counter in *counter_out (may be NULL). */
pmg_status pmg_mcsor_sample(pmg_mcsor mc, const double *b_dev, double *y_dev, int32_t its, int scaled, uint64_t seed, uint64_t counter0, uint64_t *counter_out, void *stream);
/* r = b - A y (PCMGResidualDefault / src/pc_gamgmc.c:253-254), natural numbering. */
pmg_status pmg_mcsor_residual(pmg_mcsor mc, const double *b_dev, const double *y_dev, double *r_dev, void *stream);
/* The same operations on vectors that already live in the library's colour-partitioned numbering ("layout":
   length pmg_mcsor_layout_len >= n, pad entries zero), for callers that keep their vectors on the device between
   calls (the V-cycle does): no permutation pass per call.  pmg_mcsor_get_layout writes, for every matrix row, its
   position in the layout. */
pmg_status pmg_mcsor_layout_len(pmg_mcsor mc, int32_t *ld);
pmg_status pmg_mcsor_get_size(pmg_mcsor mc, int32_t *n); /* rows of the operator */
pmg_status pmg_mcsor_get_layout(pmg_mcsor mc, int32_t *pos_of_row_host);
pmg_status pmg_mcsor_to_layout(pmg_mcsor mc, const double *nat_dev, double *lay_dev, void *stream);
pmg_status pmg_mcsor_from_layout(pmg_mcsor mc, const double *lay_dev, double *nat_dev, void *stream);
pmg_status pmg_mcsor_apply_layout(pmg_mcsor mc, const double *b_lay, double *y_lay, void *stream);
pmg_status pmg_mcsor_sample_layout(pmg_mcsor mc, const double *b_lay, double *y_lay, int32_t its, int scaled, uint64_t seed, uint64_t counter0, uint64_t *counter_out, void *stream);
pmg_status pmg_mcsor_residual_layout(pmg_mcsor mc, const double *b_lay, const double *y_lay, double *r_lay, void *stream);
/* Building blocks of the ROW-BLOCK distributed sampler (MCSORApply_MPIAIJ, src/mc_sor.c:298-381: for every colour,
   update the ghost values, then sweep the colour's rows): a rank holds its rows with the off-process columns appended
   as ghost rows (identity rows in an extra, never swept colour); the caller moves ghost values between the per-colour
   sweeps (parmgmc_amd/dist.py: DistMCSOR over torch.distributed).  row0 = global index of local row 0, so that the
   noise is that of the global row and the chain is the single-process chain bit for bit. */
pmg_status pmg_mcsor_set_noise_row_offset(pmg_mcsor mc, int64_t row0);
/* on != 0: idiag = omega / d in ONE rounding, PCPARSOR's rule (LocalMatInvertDiagonalForSOR, src/pc_parsor.c:53-86), instead
   of MCSOR's (1/d) * omega (src/mc_sor.c:114-124).  Before set-up. */
pmg_status pmg_mcsor_set_idiag_by_division(pmg_mcsor mc, int on);
pmg_status pmg_mcsor_sweep_color_layout(pmg_mcsor mc, int32_t color, int noisy, int scaled, uint64_t seed, uint64_t sweep, const double *b_layout_dev, double *y_layout_dev, void *stream);
/* MATLRC operators A + B S B^T (MCSORSetUp's LRC branch, src/mc_sor.c:572-595; MCSORBuildLRCCorrection :480-544):
   B is n x k column-major on the host in the matrix's row numbering, S the k diagonal entries of Sigma^-1.
   Call after pmg_mcsor_setup (uses the current omega).  Afterwards every directional sweep is followed by
   y -= Bb (B^T y) (:101-112) and every noisy right-hand side gets + B (sqrt(S) o eta) (src/pc_mcgibbs.c:130-140,
   src/pc_sorgibbs.c:86-90).  k = 0 removes the update. */
pmg_status pmg_mcsor_set_lowrank(pmg_mcsor mc, int32_t k, const double *B_host, const double *S_host);
/* MCSORDestroy (src/mc_sor.c:60-90); *mc = NULL afterwards; NULL handle is a no-op. */
pmg_status pmg_mcsor_destroy(pmg_mcsor *mc);

/* ------------------------------------------------------------------------------------------------------ */
/* Matrix-free MCSOR on a DMDA grid: the same interface for the operator of                                */
/* MatAssembleShiftedLaplaceFD (src/problems.c:14-75; 3-D analogue, nz = 1 is the reference 2-D matrix)    */
/* ------------------------------------------------------------------------------------------------------ */
typedef struct pmg_grid_s *pmg_grid;

/* One device owns planes [kz0, kz0+nz) of an nx*ny*nzg grid (single device: kz0 = 0, nz = nzg).
   Red-black colouring c = (i+j+k)&1 (valid for the 5/7-point star), omega = 1, sweep = forward. */
pmg_status pmg_grid_create(int32_t nx, int32_t ny, int32_t nzg, int32_t kz0, int32_t nz, double kappa, pmg_grid *g);
pmg_status pmg_grid_set_omega(pmg_grid g, double omega);
pmg_status pmg_grid_set_sweep_type(pmg_grid g, int type);
pmg_status pmg_grid_get_sweep_type(pmg_grid g, int *type);
pmg_status pmg_grid_get_num_colors(pmg_grid g, int32_t *ncolors);
/* colours of the owned points in DMDA natural order (i fastest), nx*ny*nz values */
pmg_status pmg_grid_get_coloring(pmg_grid g, int32_t *colors_host);
/* Number of doubles of a colour-partitioned device vector ("cvec") of this grid, ghost planes included. */
pmg_status pmg_grid_cvec_len(pmg_grid g, int64_t *len);
/* cvec position of every owned point, DMDA natural order (nx*ny*nz values) */
pmg_status pmg_grid_get_layout(pmg_grid g, int64_t *pos_of_point_host);
/* natural (DMDA global vector of the owned planes, i fastest) <-> cvec */
pmg_status pmg_grid_to_cvec(pmg_grid g, const double *nat_dev, double *cvec_dev, void *stream);
pmg_status pmg_grid_from_cvec(pmg_grid g, const double *cvec_dev, double *nat_dev, void *stream);
/* MCSORApply on natural-order device vectors (converts in and out; convenience / PCSHELL route,
   examples/ex3.c:59-67). */
pmg_status pmg_grid_apply(pmg_grid g, const double *b_nat_dev, double *y_nat_dev, void *stream);
/* MCSORApply / sample loop / residual directly on cvecs (no conversion; what the V-cycle and bench use). */
pmg_status pmg_grid_apply_cvec(pmg_grid g, const double *b_cvec, double *y_cvec, void *stream);
pmg_status pmg_grid_sample_cvec(pmg_grid g, const double *b_cvec, double *y_cvec, int32_t its, int scaled, uint64_t seed, uint64_t counter0, uint64_t *counter_out, void *stream);
pmg_status pmg_grid_residual_cvec(pmg_grid g, const double *b_cvec, const double *y_cvec, double *r_cvec, void *stream);
/* One colour of one sweep (the body of the colour loop of MCSORApply_MPIAIJ, src/mc_sor.c:317-340), for
   callers that interleave the per-colour ghost exchange themselves (multi-GPU).  noisy != 0 forms
   w = xi*sqrtdiag + b with noise counter `counter`; the colour-`color` points read the OTHER colour, so the
   ghost planes of colour 1-color must be current. */
pmg_status pmg_grid_sweep_color_cvec(pmg_grid g, int color, int noisy, int scaled, uint64_t seed, uint64_t counter, const double *b_cvec, double *y_cvec, void *stream);
/* The same restricted to owned planes [kbegin, kbegin+kcount): lets a multi-GPU driver sweep the two boundary
   planes first, start their halo exchange, and sweep the interior while the exchange is in flight (the overlap
   PCPARSOR gets from starting `botsct` before the INT1 rows, src/pc_parsor.c:739-745). */
pmg_status pmg_grid_sweep_color_planes_cvec(pmg_grid g, int color, int32_t kbegin, int32_t kcount, int noisy, int scaled, uint64_t seed, uint64_t counter, const double *b_cvec, double *y_cvec, void *stream);
/* Where the halo of colour `color` lives inside a cvec (offsets and count in doubles): the owned boundary
   plane on `side` (0 = low k, 1 = high k) that the neighbour needs, and the ghost plane on that side that
   receives the neighbour's plane.  A plane of one colour is one contiguous block, so the exchange that replaces
   the reference's per-colour VecScatter (src/mc_sor.c:318-319) is a single contiguous send/recv. */
pmg_status pmg_grid_halo_plane(pmg_grid g, int color, int side, int64_t *owned_offset, int64_t *ghost_offset, int64_t *count);
/* the same MATLRC update for the grid operator (B in DMDA natural order); single-device grids */
pmg_status pmg_grid_set_lowrank(pmg_grid g, int32_t k, const double *B_host, const double *S_host);
/* Sample loop on natural-order vectors: converts in once, runs `its` sweeps, converts out once. */
pmg_status pmg_grid_sample(pmg_grid g, const double *b_nat_dev, double *y_nat_dev, int32_t its, int scaled, uint64_t seed, uint64_t counter0, uint64_t *counter_out, void *stream);
pmg_status pmg_grid_destroy(pmg_grid *g);

/* ------------------------------------------------------------------------------------------------------ */
/* Multi-GPU sample loop on a z-slab decomposition: one process per GPU, per-colour halo exchange over RCCL     */
/* (ncclSend/ncclRecv across xGMI) overlapped with the interior sweep.  Replaces MCSORApply_MPIAIJ's per-colour */
/* VecScatter (src/mc_sor.c:317-340).                                                                           */
/* ------------------------------------------------------------------------------------------------------ */
typedef struct pmg_dist_s *pmg_dist;
/* rank 0: ncclGetUniqueId (128 bytes) to be broadcast to the other ranks by the launcher (torch.distributed / MPI).
   rccl_path: shared object to dlopen (NULL = "librccl.so.1"); beside PyTorch pass torch's bundled librccl.so. */
pmg_status pmg_dist_get_unique_id(const char *rccl_path, void *id128);
/* `g` owns this rank's planes (pmg_grid_create with kz0/nz).  nranks == 1 needs no RCCL (id128 may be NULL) unless
   loopback != 0, which makes the single rank its own z-neighbour for the HALO ONLY (exercises ncclSend/ncclRecv on
   one GPU; the ghost planes are written but never read because both slab faces are physical boundaries). */
pmg_status pmg_dist_create(pmg_grid g, int32_t rank, int32_t nranks, const void *id128, const char *rccl_path, int loopback, pmg_dist *d);
/* Transport "ipc": the face kernel stores its boundary planes straight into the neighbour's receive block (hipIpc
   memory mapped over xGMI) and raises a flag word there; the neighbour's face stream waits for the flag on the device
   -- no RCCL kernels, no host rendezvous; lower latency for the ~1 MB planes of a strong-scaled grid.  Bootstrap:
   every rank creates, exports its blob (pmg_dist_ipc_blob_bytes bytes), the launcher all-gathers the blobs, every
   rank connects to its two z-neighbours' blobs (NULL at the ends), barrier.  All ranks on one node. */
pmg_status pmg_dist_create_ipc(pmg_grid g, int32_t rank, int32_t nranks, pmg_dist *d);
pmg_status pmg_dist_ipc_blob_bytes(int32_t *bytes);
pmg_status pmg_dist_ipc_export(pmg_dist d, void *blob);
pmg_status pmg_dist_ipc_connect(pmg_dist d, const void *blob_lo, const void *blob_hi);
/* optional, after pmg_dist_ipc_connect: map every rank's block (blobs[0..nranks-1], own entry ignored) so that
   pmg_dist_allgather is one step instead of nranks - 1 rounds along the chain of neighbours */
pmg_status pmg_dist_ipc_connect_all(pmg_dist d, const void *const *blobs);
/* single rank as its own z-neighbour for the halo only (one-GPU timing / smoke test of the schedule) */
pmg_status pmg_dist_ipc_connect_loopback(pmg_dist d);
/* `its` samples of the sorgibbs (scaled = 0) / mcgibbs (scaled = 1) chain on this rank's slab, cvec vectors,
   sweep_type as PMG_SOR_*; noise counters counter0, counter0+1, ... exactly like pmg_grid_sample_cvec, so the
   chain is bit-identical for every number of ranks.  Work is enqueued on `stream` and an internal comm stream;
   `stream` is made to wait for the last exchange before the call returns. */
pmg_status pmg_dist_sample_cvec(pmg_dist d, const double *b_cvec, double *y_cvec, int32_t its, int scaled, int sweep_type, uint64_t seed, uint64_t counter0, uint64_t *counter_out, void *stream);
/* MCSORApply on the slabs: one deterministic sweep (forward / backward / symmetric) with the same halo schedule */
pmg_status pmg_dist_apply_cvec(pmg_dist d, const double *b_cvec, double *y_cvec, int sweep_type, void *stream);
/* vals[0..count) (device, count <= 4096) <- sum over all ranks, formed in rank order on every rank */
pmg_status pmg_dist_allreduce_sum(pmg_dist d, double *vals_dev, int32_t count, void *stream);
/* Building blocks of the distributed V-cycle on the same transports.  pmg_dist_exchange: one round trip with both
   z-neighbours on `stream`, nseg (<= 4) contiguous device segments per side; what is sent to the low neighbour
   arrives in its high receive segments and vice versa (segment sizes of a pair must agree; sides without a neighbour
   are skipped).  pmg_dist_allgather: block r (counts[r] doubles at buf + offsets[r]) is owned by rank r; afterwards
   every rank holds all blocks.  Collective: every rank makes the same sequence of calls. */
pmg_status pmg_dist_exchange(pmg_dist d, int nseg, const double *const *send_lo, const int64_t *nsend_lo, double *const *recv_lo, const int64_t *nrecv_lo, const double *const *send_hi, const int64_t *nsend_hi, double *const *recv_hi, const int64_t *nrecv_hi, void *stream);
pmg_status pmg_dist_allgather(pmg_dist d, double *buf_dev, const int64_t *offsets, const int64_t *counts, void *stream);
/* after synchronising the stream: PMG_ERR_LIB if a device-side wait for a halo flag gave up (lost or unreachable
   neighbour); the object then fails every later call */
pmg_status pmg_dist_check(pmg_dist d);
/* rank / number of ranks / largest message (doubles) the generic exchange can carry */
pmg_status pmg_dist_get_info(pmg_dist d, int32_t *rank, int32_t *nranks, int64_t *capacity);
/* A rank's own description of where it runs and how its halos travel -- printed per rank by `bench.py --gpus N` and
   `examples/pmg_bench -ranks N` so that a run on several physical GPUs can be judged from its record alone: device index
   and PCI bus id, peer access to the z-neighbours' devices (lo_device / hi_device: their device indices as this process
   sees them, -1 = no such neighbour; result 1 / 0 / -1 = yes / no / not applicable), the z-neighbours' ranks, the
   transport ("ipc" | "rccl"), ncclCommCount of the RCCL communicator (0 on ipc), and the polls of halo flag words that
   found them not yet raised since the object was created (ipc: 0 in steady state).  Synchronises the device. */
typedef struct {
  int32_t  rank, nranks, device, neighbour[2], peer_access[2], rccl_comm_count;
  uint64_t halo_wait_polls;
  char     pci_bus_id[32], transport[8];
} pmg_dist_description;
pmg_status pmg_dist_describe(pmg_dist d, int32_t lo_device, int32_t hi_device, pmg_dist_description *out);
/* ipc transport: unmap the peers' receive blocks.  Orderly tear-down when further transports follow: every rank
   disconnects, the caller runs a barrier, every rank destroys (frees its own block). */
pmg_status pmg_dist_ipc_disconnect(pmg_dist d);
pmg_status pmg_dist_destroy(pmg_dist *d);

/* ------------------------------------------------------------------------------------------------------ */
/* Exact coarse sampler: replaces PCCHOLSAMPLER's dense path (src/pc_chols.c:174-194, :220-291)             */
/* ------------------------------------------------------------------------------------------------------ */
typedef struct pmg_chol_s *pmg_chol;
/* PCSetUp_CholSampler dense branch: MatConvert to dense + potrf('L').  Host CSR of an SPD matrix; returns
   PMG_ERR_MAT_CH_ZRPVT naming the failing leading minor like src/pc_chols.c:190.  Synchronous. */
pmg_status pmg_chol_create_csr(int32_t n, const int32_t *rowptr_host, const int32_t *colidx_host, const double *vals_host, pmg_chol *ch);
/* the same sampler for a MATLRC operator A + B S B^T (src/pc_chols.c:119-153: the update is added to the matrix
   before it is factored); B is n x k column-major, S the k diagonal entries, both on the host */
pmg_status pmg_chol_create_csr_lowrank(int32_t n, const int32_t *rowptr, const int32_t *colidx, const double *vals, int32_t k, const double *B_host, const double *S_host, pmg_chol *out);
/* either PetscInt width (idx_width = 32 | 64); k = 0 for a plain AIJ matrix */
pmg_status pmg_chol_create_csr_idx(int64_t n, const void *rowptr, const void *colidx, const double *vals, int idx_width, int32_t k, const double *B_host, const double *S_host, pmg_chol *out);
/* the lower factor L, column-major n*n on the host (upper part zero) */
pmg_status pmg_chol_get_factor(pmg_chol ch, double *L_colmajor_host);
/* PCApply_CholSampler (src/pc_chols.c:262-291): y = L^-T (L^-1 b + xi), xi = row-stream normals of
   (seed, counter); noisy == 0 drops xi (plain solve y = A^-1 b). */
pmg_status pmg_chol_sample(pmg_chol ch, const double *b_dev, double *y_dev, int noisy, uint64_t seed, uint64_t counter, void *stream);
pmg_status pmg_chol_destroy(pmg_chol *ch);

/* ------------------------------------------------------------------------------------------------------ */
/* MCSOR on a general AIJ matrix distributed by row blocks: replaces MCSORApply_MPIAIJ (src/mc_sor.c:298-381)  */
/* ------------------------------------------------------------------------------------------------------ */
typedef struct pmg_distmcsor_s *pmg_distmcsor;
/* One rank per device owns a contiguous block of rows (MatGetOwnershipRange).  `mc`: its local operator -- the rows with
   the off-process COLUMNS appended as identity rows in one extra colour that is never swept (the diagonal / off-diagonal
   blocks of MatMPIAIJGetSeqAIJ, src/mc_sor.c:308, merged into one CSR), created with pmg_mcsor_create_csr[_idx], a USER
   colouring that is a valid distance-1 colouring of the GLOBAL matrix, pmg_mcsor_set_noise_row_offset(first global row),
   and set up.  `dist`: any transport (pmg_dist_create[_ipc]; its grid may be NULL).  The per-colour scatter plan
   (MatCreateScatters, src/mc_sor.c:152-214, de-duplicated per ghost column and split by the colour in which a value
   changes): send_ptr[ncolors+1] / send_pos = layout positions (pmg_mcsor_get_layout) of my rows of colour c that some other
   rank reads; counts[c*nranks + r] = length of rank r's list of colour c (the same array on every rank);
   recv_ptr[ncolors+1] / recv_src / recv_pos = for every ghost row that changes in colour c its index in the colour's
   gathered list (rank blocks in rank order) and its layout position.  Host arrays, copied; mc and dist are borrowed. */
pmg_status pmg_distmcsor_create(pmg_mcsor mc, pmg_dist dist, int32_t ncolors, const int64_t *send_ptr, const int32_t *send_pos, const int64_t *counts, const int64_t *recv_ptr, const int32_t *recv_src, const int32_t *recv_pos, pmg_distmcsor *out);
/* the sample loop / MCSORApply on layout vectors (my rows filled; ghost entries of y are refreshed inside): for every
   colour, sweep its rows on the device, then update the ghost values (src/mc_sor.c:317-340), all on `stream`, no host
   work between colours.  Collective.  Bit-identical to the single-process chain for any number of ranks. */
pmg_status pmg_distmcsor_sample_layout(pmg_distmcsor h, const double *b_lay, double *y_lay, int32_t its, int scaled, int sweep_type, uint64_t seed, uint64_t counter0, uint64_t *counter_out, void *stream);
pmg_status pmg_distmcsor_apply_layout(pmg_distmcsor h, const double *b_lay, double *y_lay, int sweep_type, void *stream);
/* the same on NATURAL-order device vectors of this rank's `nowned` owned rows -- what the local part of a Vec of the
   MATMPIAIJ holds (VecGetArray, src/mc_sor.c:252-255): converted to the layout and back inside.  y: state in, sample out */
pmg_status pmg_distmcsor_sample(pmg_distmcsor h, int32_t nowned, const double *b_owned_dev, double *y_owned_dev, int32_t its, int scaled, int sweep_type, uint64_t seed, uint64_t counter0, uint64_t *counter_out, void *stream);
pmg_status pmg_distmcsor_apply(pmg_distmcsor h, int32_t nowned, const double *b_owned_dev, double *y_owned_dev, int sweep_type, void *stream);
/* MATLRC operator A + B S B^T on row blocks (src/mc_sor.c:572-595 on a MATMPIAIJ base): B_local is nlocal x k column-major
   in the local row numbering (nlocal = rows of the local operator; this rank's `nowned` rows first; ghost entries ignored), S the k diagonal entries of
   Sigma^-1.  Afterwards every directional sweep of pmg_distmcsor_sample_layout / _apply_layout is followed by
   y -= Bb (B^T y) (:101-112; B^T y is all-reduced in rank order) and every noisy right-hand side gets
   + B (sqrt(S) o eta).  Collective (the correction is built with distributed sweeps); k = 0 removes the update.
   _dev: B already in the operator's layout on the device (ld x k), zero on the ghost rows.
   pmg_distmcsor_residual_layout: r = b - (A + B S B^T) y on the owned rows. */
pmg_status pmg_distmcsor_set_lowrank(pmg_distmcsor h, int32_t k, int32_t nlocal, int32_t nowned, const double *B_local_host, const double *S_host);
pmg_status pmg_distmcsor_set_lowrank_dev(pmg_distmcsor h, int32_t k, const double *B_layout_dev, const double *S_host);
pmg_status pmg_distmcsor_residual_layout(pmg_distmcsor h, const double *b_lay, const double *y_lay, double *r_lay, void *stream);
/* refresh every ghost row of a layout vector from its owner (collective) */
pmg_status pmg_distmcsor_refresh_layout(pmg_distmcsor h, double *v_lay, void *stream);
pmg_status pmg_distmcsor_destroy(pmg_distmcsor *h);

/* ------------------------------------------------------------------------------------------------------ */
/* Multigrid Monte Carlo on a DMDA hierarchy: replaces PCGAMGMC with -pc_gamgmc_mg_type mg                  */
/* (src/pc_gamgmc.c) and the PCMG V-cycle it drives                                                        */
/* ------------------------------------------------------------------------------------------------------ */
typedef struct pmg_mgmc_s *pmg_mgmc;
/* Sample callback, the analogue of PetscErrorCode (*cb)(PetscInt it, Vec y, void *ctx) set with
   PCSetSampleCallback (src/parmgmc.c:146-151): called on the host after sample `it` has been ENQUEUED on the
   stream, with the sample in DMDA natural order on the device (valid in stream order).  Non-zero return aborts
   the loop and is returned by pmg_mgmc_sample. */
typedef int (*pmg_sample_callback)(int32_t it, const double *y_nat_dev, int32_t n, void *ctx);
/* `levels` grids, the finest nx*ny*nz with the operator of MatAssembleShiftedLaplaceFD (src/problems.c:14-75),
   each coarser one (n-1)/2+1 points per refined direction (PETSc DMDA coarsening; PMG_ERR_ARG_SIZ if (n-1) is
   odd).  Defaults = the options PCGAMGMC injects (src/pc_gamgmc.c:299-350): level sampler sorgibbs, 1 sweep
   before and after, coarse cholsampler, Galerkin coarse operators. */
pmg_status pmg_mgmc_create_dmda(int32_t nx, int32_t ny, int32_t nz, double kappa, int32_t levels, pmg_mgmc *mg);
/* The same sampler on z-slabs of the DMDA, one rank per device (the reference distributes every PCMG level over the
   MPI ranks; GAMG reduces coarse grids to rank 0, src/pc_chols.c:38-47,272-282).  `g` = this rank's slab of the fine
   operator (pmg_grid_create with kz0 = cuts[rank], nz = cuts[rank+1] - cuts[rank]), `dist` = the halo transport
   created on it (both borrowed), cuts[0..nranks] = first fine plane of every rank.  Coarse plane K belongs to the
   owner of fine plane 2K; levels with <= 2^19 unknowns (env PMG_MG_REPLICATE_BELOW) or fewer planes than ranks are
   replicated on every rank after one all-gather of their right-hand side.  pmg_mgmc_sample then takes this rank's
   planes of b and y (nx*ny*nz_owned values, natural order) and is collective; samples are bit-identical for any
   number of ranks.  Low-rank updates and host copies are single-device features. */
pmg_status pmg_mgmc_create_dmda_slab(int32_t nx, int32_t ny, int32_t nz, double kappa, int32_t levels, pmg_grid g, pmg_dist dist, const int32_t *cuts, pmg_mgmc *mg);
/* The same sampler on a hierarchy handed over level by level (level 0 = coarsest): the level operators and
   interpolations PCGAMGMC finds inside PETSc's PCMG / PCGAMG after PCSetUp (PCMGGetSmoother + PCGetOperators,
   PCMGGetInterpolation: src/pc_gamgmc.c:165-176), e.g. a GAMG hierarchy of an unstructured P1 matrix
   (-pc_gamgmc_mg_type gamg, the reference's default).  Host CSR arrays are borrowed until pmg_mgmc_setup. */
pmg_status pmg_mgmc_create_hierarchy(int32_t levels, pmg_mgmc *mg);
pmg_status pmg_mgmc_set_level_operator(pmg_mgmc mg, int32_t level, int32_t n, const int32_t *rowptr_host, const int32_t *colidx_host, const double *vals_host);
/* interpolation from level-1 (ncols unknowns) to `level` (nrows unknowns), level >= 1 */
pmg_status pmg_mgmc_set_level_interpolation(pmg_mgmc mg, int32_t level, int32_t nrows, int32_t ncols, const int32_t *rowptr_host, const int32_t *colidx_host, const double *vals_host);
/* Caller-supplied hierarchy distributed by ROW BLOCKS over the ranks of `dist` (any pmg_dist object: only its transport is
   used), the reference's PCGAMGMC on a MATMPIAIJ (src/pc_gamgmc.c:157-223 with MCSORApply_MPIAIJ, src/mc_sor.c:298-381, as
   the level sampler).  The levels 0 .. F-1 are REPLICATED: passed whole on every rank (pmg_mgmc_set_level_operator /
   _interpolation as for one device; level 0 is the exact sampler, factored redundantly) and run with the single-device
   kernels on identical data; the right-hand side of level F-1 is all-gathered by the row blocks coarse_starts[0 .. nranks]
   of THAT level.  F >= 1 is the lowest level that has a row block.  A row-block level l >= F is this rank's rows in LOCAL numbering
   (pmg_mgmc_set_level_operator: owned rows in global order with their entries in global CSR order, then one identity row
   per ghost = every row of another rank that this rank's operator, restriction or the finer level's interpolation reads)
   plus pmg_mgmc_set_level_rowblock: global row of local row 0, number of owned rows, a globally valid distance-1 colouring
   of the owned rows, and the ghost-update plan of pmg_distmcsor_create with LOCAL ROW indices in place of layout
   positions.  pmg_mgmc_set_level_interpolation(l) then takes the owned rows of P_l with columns in the local numbering of
   level l-1 (global for the replicated level F-1) and pmg_mgmc_set_level_restriction(l) the rows of P_l^T this rank owns on level l-1,
   columns in the local numbering of level l, entries by ascending global fine row.  pmg_mgmc_sample's vectors have one
   entry per local row of the finest level (ghost entries ignored / undefined).  Same bits as the single-device chain of
   pmg_mgmc_create_hierarchy with the same colouring.  pmg_mgmc_set_lowrank takes this rank's rows of B (n_local x k, ghost
   entries ignored): the update is restricted level by level with the hierarchy's own P^T and lives in the levels' row-block
   samplers (equal to the single-device chain to rounding: B^T y is summed per rank, then over the ranks).  PMG_ERR_SUP: Gibbs
   coarse solver. */
pmg_status pmg_mgmc_set_rowblock_transport(pmg_mgmc mg, pmg_dist dist, const int64_t *coarse_starts);
pmg_status pmg_mgmc_set_level_rowblock(pmg_mgmc mg, int32_t level, int64_t row0, int32_t nowned, int32_t ncolors, const int32_t *colors_owned, const int64_t *send_ptr, const int32_t *send_rows, const int64_t *counts, const int64_t *recv_ptr, const int32_t *recv_src, const int32_t *recv_rows);
pmg_status pmg_mgmc_set_level_restriction(pmg_mgmc mg, int32_t level, int32_t nrows, int32_t ncols, const int32_t *rowptr, const int32_t *colidx, const double *vals);
/* both for either PetscInt width (idx_width = 32 | 64) */
pmg_status pmg_mgmc_set_level_operator_idx(pmg_mgmc mg, int32_t level, int64_t n, const void *rowptr_host, const void *colidx_host, const double *vals_host, int idx_width);
pmg_status pmg_mgmc_set_level_interpolation_idx(pmg_mgmc mg, int32_t level, int64_t nrows, int64_t ncols, const void *rowptr_host, const void *colidx_host, const double *vals_host, int idx_width);
/* -mg_levels_pc_type sorgibbs (scaled = 0, omega = 1) | mcgibbs (scaled = 1, any omega, any sweep type);
   its = -mg_levels_ksp_max_it */
pmg_status pmg_mgmc_set_smoother(pmg_mgmc mg, int scaled, double omega, int sweep_type, int32_t its);
/* -mg_coarse_pc_type cholsampler (type 0) | Gibbs sweeps with -mg_coarse_ksp_max_it its (type 1) */
pmg_status pmg_mgmc_set_coarse(pmg_mgmc mg, int type, int32_t its);
/* literal != 0: every sample computes w = b - A y, work = MG(w), y += work exactly as src/pc_gamgmc.c:253-256 writes
   it.  Default (0): the same V-cycle run in place on (b, y) -- for stationary linear sweeps S(b, y) = y + S(b - A y, 0)
   with the same noise, so both are the same chain up to rounding; the in-place form saves one fine residual, one
   axpy and one memset per sample. */
pmg_status pmg_mgmc_set_correction_form(pmg_mgmc mg, int literal);
/* on = 0: residual r = b - A x and restriction b_c = P^T r as two kernels on every level, a MATLRC term subtracted from r
   BEFORE the restriction: the reference's operation order (PCMG residual on the MATLRC level operator, src/pc_gamgmc.c:194,
   then MatRestrict).  Default (1): grid levels run ONE kernel that never stores r, and a MATLRC term is subtracted in
   restricted form, b_c -= B_{l-1} (S B_l^T x) with B_{l-1} = P^T B_l (src/pc_gamgmc.c:177-178) -- equal up to rounding
   (1e-12 on a whole sample, tests/test_gpu_benchsize_lowrank.py).  Single-device hierarchies: any time; z-slabs: before set-up. */
pmg_status pmg_mgmc_set_fused_transfers(pmg_mgmc mg, int on);
/* Colouring rule of the AIJ levels of a hierarchy (pmg_mgmc_create_hierarchy) on one device: PMG_COLORING_GREEDY (default) or
   PMG_COLORING_ITERATED (one class and one dependent launch fewer per sweep and level on P1 hierarchies); before set-up.  The
   reference colours every level with PETSc's JP (src/mc_sor.c:383-395): any valid distance-1 colouring is a valid sampler.
   Row-block levels keep the colouring their plan was built with. */
pmg_status pmg_mgmc_set_coloring(pmg_mgmc mg, int rule);
/* MATLRC fine operator A + B S B^T (examples/ex4.c): PCGAMGMC_SetUpHierarchy (src/pc_gamgmc.c:157-196) gives every
   level the operator A_l + B_l S B_l^T, B_{l-1} = P_l^T B_l, for its sampler and its residual; the coarse Cholesky
   sampler factors the explicit sum (src/pc_chols.c:119-153).  B: n_fine x k column-major in the finest level's
   natural numbering, S: k entries, host arrays (copied).  Call after the level operators are known, before set-up. */
pmg_status pmg_mgmc_set_lowrank(pmg_mgmc mg, int32_t k, const double *B_host, const double *S_host);
/* keep host copies of the Galerkin operators and interpolations for pmg_mgmc_get_level_matrix */
pmg_status pmg_mgmc_set_keep_host(pmg_mgmc mg, int keep);
/* PCSetUp(pg->mg) + PCGAMGMC_SetUpHierarchy (src/pc_gamgmc.c:145-225, :352-353).  Synchronous. */
pmg_status pmg_mgmc_setup(pmg_mgmc mg);
pmg_status pmg_mgmc_get_num_levels(pmg_mgmc mg, int32_t *levels);
pmg_status pmg_mgmc_get_level_dims(pmg_mgmc mg, int32_t level, int32_t *nx, int32_t *ny, int32_t *nz);
/* which = 0: Galerkin operator of `level` (< finest); which = 1: interpolation from level-1 to `level`.  CSR in
   natural numbering; pass NULL arrays to query sizes. */
pmg_status pmg_mgmc_get_level_matrix(pmg_mgmc mg, int32_t level, int which, int32_t *nrows, int32_t *nnz, int32_t *rowptr_host, int32_t *colidx_host, double *vals_host);
/* PCApplyRichardson_GAMGMC (src/pc_gamgmc.c:227-264): `its` samples of the chain y <- y + MG(b - A y) (first one
   y = MG(b) when guesszero != 0), natural-order device vectors.  Sample s = counter0 + it draws its noise from
   counters [64 s, 64 s + 64) of per-level streams, so a chain can be resumed at any sample. */
pmg_status pmg_mgmc_sample(pmg_mgmc mg, const double *b_nat_dev, double *y_nat_dev, int32_t its, int guesszero, uint64_t seed, uint64_t counter0, uint64_t *counter_out, pmg_sample_callback cb, void *cbctx, void *stream);
/* ALGORITHMIC bytes of one sample of pmg_mgmc_sample as the cycle is built for this hierarchy (this rank's share): each
   launch counted with the operands it must read and write once -- SURVEY 8(d)'s per-unit figures (24 B/unknown per
   structured sweep, 12 nnz + 40 N per sliced-ELL sweep) and their analogues for residual, transfers, coarse sample and
   low-rank steps, listed at the definition (pmg_mgmc.c) and in DESIGN.md section 6.  per_level_host: nlevels doubles or NULL.
   What bench.py divides by the measured time per sample for the roofline fraction of its V-cycle lines. */
pmg_status pmg_mgmc_get_algorithmic_bytes(pmg_mgmc mg, double *total, double *per_level_host);
/* Diagnostics: ONE kernel of the V-cycle on caller-supplied device vectors in the level's own layout (single device).
   They exist for the parity tests at 257^3 / 513^3, where a whole oracle cycle is out of reach: the tests run one
   kernel and compare sampled rows with the oracle's row arithmetic (PCMG pieces entered at reference
   src/pc_gamgmc.c:246,255; level sweeps = MCSORApply_SEQAIJ, src/mc_sor.c:241-296).
   layout kinds: 0 = grid level (colour-partitioned cvec), 1 = class-stencil level and 3 = dense coarsest level (natural
   order behind one ghost plane: natural index q at off + q), 2 = sliced-ELL level (pmg_mcsor layout). */
pmg_status pmg_mgmc_get_level_layout(pmg_mgmc mg, int32_t level, int32_t *kind, int64_t *ld, int64_t *off);
pmg_status pmg_mgmc_get_level_stencil(pmg_mgmc mg, int32_t level, double *coef_27x27_host, double *sqrtdiag_27_host);
pmg_status pmg_mgmc_level_sweep(pmg_mgmc mg, int32_t level, int backward, int noisy, uint64_t seed, uint64_t counter, const double *b_lvl, double *x_lvl, void *stream);
pmg_status pmg_mgmc_level_residual(pmg_mgmc mg, int32_t level, const double *b_lvl, const double *x_lvl, double *r_lvl, void *stream);
pmg_status pmg_mgmc_level_restrict(pmg_mgmc mg, int32_t level, double *r_fine_lvl, double *b_coarse_lvl, void *stream);
/* b_coarse = P^T (b - A x) as the V-cycle forms it on a single-device grid level (one kernel, same bits as
   level_residual + level_restrict); PMG_ERR_SUP on levels where the cycle runs the two steps */
pmg_status pmg_mgmc_level_residual_restrict(pmg_mgmc mg, int32_t level, const double *b_lvl, const double *x_lvl, double *b_coarse_lvl, void *stream);
pmg_status pmg_mgmc_level_prolong_add(pmg_mgmc mg, int32_t level, const double *e_coarse_lvl, double *x_fine_lvl, void *stream);
/* the MATLRC update of a level as its kernels hold it (row-compact form: ball observations touch << N rows, src/obs.c:39-50):
   ns layout positions of the support rows, ascending, and the ns x k column-major blocks of B_l (= P^T ... P^T B,
   src/pc_gamgmc.c:177-178), Bb forward and Bb backward (MCSORBuildLRCCorrection, src/mc_sor.c:480-544).  NULL arrays query k, ns. */
pmg_status pmg_mgmc_level_lowrank_factors(pmg_mgmc mg, int32_t level, int32_t *k, int64_t *ns, int64_t *rows_host, double *B_host, double *Bb_fwd_host, double *Bb_bwd_host);
/* y -= Bb (B^T y), MCSORPostSOR_LRC (src/mc_sor.c:101-112), on a vector in the level's layout */
pmg_status pmg_mgmc_level_lowrank_post(pmg_mgmc mg, int32_t level, int backward, double *y_lvl, void *stream);
/* restricted = 0: out (level layout) -= B_l (S B_l^T x); 1: out (layout of level-1) -= B_{l-1} (S B_l^T x) */
pmg_status pmg_mgmc_level_lowrank_residual_sub(pmg_mgmc mg, int32_t level, int restricted, const double *x_lvl, double *out_lvl, void *stream);
pmg_status pmg_mgmc_destroy(pmg_mgmc *mg);

/* ------------------------------------------------------------------------------------------------------ */
/* Row-block (MATMPIAIJ) set-up in C: what a caller with an MPI communicator needs to reach the multi-GPU   */
/* samplers -- replaces MatCreateScatters (src/mc_sor.c:152-214), the MATMPIAIJ branch of MCSORSetUp        */
/* (:553-605) and the distributed half of PCGAMGMC_SetUpHierarchy (src/pc_gamgmc.c:157-223)                */
/* ------------------------------------------------------------------------------------------------------ */
/* The ONE collective the set-up needs from its caller: a byte all-gather of equal-sized blocks over the ranks of the
   matrix' communicator (recv holds nranks * nbytes, block r from rank r; 0 = success).  A PETSc adapter passes
     int ag(void *ctx, const void *s, int64_t n, void *r) { return MPI_Allgather(s, (int)n, MPI_BYTE, r, (int)n, MPI_BYTE, *(MPI_Comm *)ctx); }
   (adapter/hip_petsc_common.h), the Python tests torch.distributed, examples/pmg_bench.c pipes between forked ranks.
   Every function that takes a pmg_host_comm is COLLECTIVE over its ranks; a failure on one rank fails on all. */
typedef int (*pmg_allgather_fn)(void *ctx, const void *send, int64_t nbytes, void *recv);
typedef struct {
  int32_t          rank, nranks;
  pmg_allgather_fn allgather; /* may be NULL when nranks == 1 */
  void            *ctx;
} pmg_host_comm;

/* (Ad, Ao, garray) of MatMPIAIJGetSeqAIJ (src/mc_sor.c:308) -> this rank's rows with GLOBAL columns.  Ad: nloc x nloc, local
   columns (global = col + cstart); Ao: compact columns (global = garray[col]); PetscInt arrays of idx_width bits.  order
   PMG_ROWBLOCK_ORDER_GLOBAL: entries by ascending global column = the row of the sequential matrix (the distributed chain
   then equals the single-process chain bit for bit); PMG_ROWBLOCK_ORDER_MPIAIJ: diagonal block first, then the off-diagonal
   block, as MCSORApply_MPIAIJ visits them (src/mc_sor.c:331-333).  rp_out: nloc + 1; ci_out, v_out: nnz(Ad) + nnz(Ao). */
#define PMG_ROWBLOCK_ORDER_GLOBAL 0
#define PMG_ROWBLOCK_ORDER_MPIAIJ 1
pmg_status pmg_rowblock_merge_mpiaij(int32_t nloc, int64_t cstart, const void *ad_rowptr, const void *ad_colidx, const double *ad_vals, const void *ao_rowptr, const void *ao_colidx, const double *ao_vals, const void *garray, int idx_width, int order, int64_t *rp_out, int64_t *ci_out, double *v_out);
/* The library's first-fit colouring of the GLOBAL matrix (rows ascending, smallest colour no coloured neighbour carries),
   computed rank after rank: the colouring one process computes.  row_starts[0..nranks] = MatGetOwnershipRanges. */
pmg_status pmg_rowblock_color_greedy(const pmg_host_comm *comm, const int64_t *row_starts, const int64_t *rowptr, const int64_t *colidx_global, int32_t *colors_owned, int32_t *ncolors);
/* the same for PMG_COLORING_ITERATED: the first-fit classes revisited last class first, every rank colouring its rows of a class
   at once, one exchange of the assigned colours per class -- the colouring pmg_mcsor's rule gives one process.  Collective. */
pmg_status pmg_rowblock_color_iterated(const pmg_host_comm *comm, const int64_t *row_starts, const int64_t *rowptr, const int64_t *colidx_global, int32_t *colors_owned, int32_t *ncolors);
/* Ghost rows + per-colour ghost-update plan of one row block: MatCreateScatters (src/mc_sor.c:152-214) de-duplicated per
   ghost row and laid out for ONE all-gather per colour.  cols: the global column indices of this rank's rows; extra:
   further rows of other ranks it reads (transfer columns).  Views (pmg_rowblock_plan_get, borrowed): ghosts = sorted global
   rows (local row nowned + q); send_ptr[ncolors+1] / send_rows = local owned rows this rank contributes per colour (ascending
   global row); counts[c * nranks + r]; recv_ptr / recv_src / recv_rows = for every ghost row the index of its value in the
   colour's gather buffer and its local row -- the arguments of pmg_mgmc_set_level_rowblock. */
typedef struct pmg_rowblock_plan_s *pmg_rowblock_plan;
pmg_status pmg_rowblock_plan_create(const pmg_host_comm *comm, const int64_t *row_starts, int64_t ncols, const int64_t *cols, int64_t nextra, const int64_t *extra, int32_t ncolors, const int32_t *colors_owned, pmg_rowblock_plan *plan);
/* Is `colors_owned` a distance-1 colouring as far as this rank's rows can tell (no owned row shares its colour with one of
   its columns, owned or ghost -- the ghost rows' colours are the plan's)?  PMG_ERR_ARG_WRONG on EVERY rank if any rank finds a
   conflict.  Every constructor that takes rows + a plan runs it (pmg_rowblock_sampler_create, pmg_rbh_build).  First fit
   yields such a colouring for STRUCTURALLY SYMMETRIC patterns (what PETSc assembles for these operators); a pattern in which r
   lists c but c does not list r can defeat it and is refused here instead of being swept with a race.  Collective. */
pmg_status pmg_rowblock_check_coloring(const pmg_host_comm *comm, pmg_rowblock_plan plan, const int64_t *rowptr, const int64_t *colidx_global, const int32_t *colors_owned);
pmg_status pmg_rowblock_plan_get(pmg_rowblock_plan plan, int32_t *nghost, const int64_t **ghosts, const int64_t **send_ptr, const int32_t **send_rows, const int64_t **counts, const int64_t **recv_ptr, const int32_t **recv_src, const int32_t **recv_rows);
void       pmg_rowblock_plan_destroy(pmg_rowblock_plan *plan);
/* A whole hierarchy (level 0 = coarsest) whose levels are MATMPIAIJ matrices -- what PCGAMGMC finds inside PCMG / PCGAMG
   (src/pc_gamgmc.c:165-176): per level this rank's rows of A_l and of P_l with global columns.  pmg_rbh_build (collective)
   replicates the levels with at most replicate_below rows (at least the coarsest; < 0: 50 000), colours the others (the
   caller's colouring or first-fit), forms the rows of P^T every rank owns, the ghost plans and the local matrices;
   pmg_rbh_create_mgmc hands all of it to pmg_mgmc_create_hierarchy / pmg_mgmc_set_level_*: afterwards pmg_mgmc_set_smoother,
   pmg_mgmc_set_lowrank, pmg_mgmc_setup, pmg_mgmc_sample as on one device.  Keep the handle until pmg_mgmc_setup returned. */
typedef struct pmg_rbh_s *pmg_rbh;
typedef struct {
  int64_t        n_global, row0;
  const int64_t *starts; /* nranks + 1 */
  int32_t        replicated, nowned, nlocal, nghost, ncolors, P_nrows, R_nrows, ncoarse_local;
  const int32_t *rp, *ci, *colors, *P_rp, *P_ci, *R_rp, *R_ci, *send_rows, *recv_src, *recv_rows;
  const double  *v, *P_v, *R_v;
  const int64_t *ghosts, *send_ptr, *counts, *recv_ptr;
} pmg_rbh_level_view;
pmg_status pmg_rbh_create(const pmg_host_comm *comm, int32_t nlevels, int64_t replicate_below, pmg_rbh *h);
pmg_status pmg_rbh_set_level_operator(pmg_rbh h, int32_t level, int64_t n_global, int64_t row0, int64_t nloc, const void *rowptr, const void *colidx_global, const double *vals, int idx_width);
pmg_status pmg_rbh_set_level_interpolation(pmg_rbh h, int32_t level, int64_t nloc_rows, const void *rowptr, const void *colidx_global_coarse, const double *vals, int idx_width);
pmg_status pmg_rbh_set_level_coloring(pmg_rbh h, int32_t level, int32_t ncolors, const int32_t *colors_owned);
pmg_status pmg_rbh_set_coloring(pmg_rbh h, int rule); /* PMG_COLORING_GREEDY (default) | PMG_COLORING_ITERATED for the levels without a caller's colouring; before pmg_rbh_build */
pmg_status pmg_rbh_build(pmg_rbh h);
pmg_status pmg_rbh_get_info(pmg_rbh h, int32_t *nlevels, int32_t *fold);
pmg_status pmg_rbh_get_level(pmg_rbh h, int32_t level, pmg_rbh_level_view *view);
pmg_status pmg_rbh_create_mgmc(pmg_rbh h, pmg_dist transport, pmg_mgmc *mg);
void       pmg_rbh_destroy(pmg_rbh *h);
/* The stand-alone multicolour sampler on a row block (MCSORCreate + MCSORSetUp on a MATMPIAIJ, src/mc_sor.c:553-605): local
   operator, device set-up with noise keyed on the global row, ghost plan, C sample loop.  colors_owned NULL: first-fit on
   the global matrix.  Use with pmg_distmcsor_sample_layout / _apply_layout; destroy *distmcsor, then *mc.  Needs a GPU. */
pmg_status pmg_rowblock_sampler_create(const pmg_host_comm *comm, pmg_dist transport, const int64_t *row_starts, const void *rowptr, const void *colidx_global, const double *vals, int idx_width, int32_t ncolors, const int32_t *colors_owned, double omega, pmg_mcsor *mc, pmg_distmcsor *distmcsor);
/* The halo / all-gather transports bootstrapped through the caller's all-gather instead of torch.distributed: kind "ipc"
   (hipIpc peer stores + flag words; the handle blobs travel through comm) or "rccl" (rank 0's ncclUniqueId travels through
   comm; rccl_path = the librccl.so to dlopen, NULL: default search).  g = this rank's z-slab (pmg_grid_create with kz0 / nz)
   for the DMDA samplers, NULL for a pure transport (row blocks).  pmg_dist_destroy_comm: unmap, barrier, free. */
pmg_status pmg_dist_create_comm(const pmg_host_comm *comm, const char *kind, pmg_grid g, const char *rccl_path, pmg_dist *d);
pmg_status pmg_dist_destroy_comm(const pmg_host_comm *comm, pmg_dist *d);

/* ------------------------------------------------------------------------------------------------------ */
/* The Woodbury term of PCWOODBURY (src/woodbury.c) for ANY sampler and solver, on one device or on rows    */
/* distributed over the ranks of a pmg_dist (z-slabs and row blocks alike)                                  */
/* ------------------------------------------------------------------------------------------------------ */
/* B_host: this rank's n rows of B (k columns, column-major, leading dimension ldb), S_host: the k entries of S = Sigma^-1
   (MatLRCGetMats, src/woodbury.c:162); dist NULL: one device, else B^T C and every B^T y are summed over its ranks in rank
   order (MatTransposeMatMult / MatMultTranspose of the reference's dense MPI matrix, :53,:280).
     set-up (:21-91):  for every column c: pmg_woodbury_column hands out B(:,c) and a zero-filled C(:,c) on the device, the
                       caller runs ITS solver on them (C(:,c) = solver(B(:,c)), :39-49); pmg_woodbury_finish forms
                       G = C (S^-1 + B^T C)^-1 (collective, synchronous);
     sample (:263-289): pmg_woodbury_noisy_rhs (w = b + B (sqrt|S| o xi), xi from (seed, counter), the same on every rank),
                       the caller's A-sampler on w, pmg_woodbury_correct (y -= G (B^T y), collective).
   All vectors: natural order over this rank's rows, device memory. */
typedef struct pmg_woodbury_s *pmg_woodbury;
pmg_status pmg_woodbury_create(int64_t n, int32_t k, const double *B_host, int64_t ldb, const double *S_host, pmg_dist dist, pmg_woodbury *w);
pmg_status pmg_woodbury_column(pmg_woodbury w, int32_t c, const double **B_col_dev, double **C_col_dev, void *stream);
/* C(:,c) <- a solver result held in the caller's own device vector (VecCopy(x, c), :46-48) */
pmg_status pmg_woodbury_set_c_column(pmg_woodbury w, int32_t c, const double *x_dev, void *stream);
pmg_status pmg_woodbury_finish(pmg_woodbury w);
pmg_status pmg_woodbury_noisy_rhs(pmg_woodbury w, const double *b_dev, double *w_dev, uint64_t seed, uint64_t counter, void *stream);
pmg_status pmg_woodbury_correct(pmg_woodbury w, double *y_dev, void *stream);
pmg_status pmg_woodbury_get_correction(pmg_woodbury w, double *G_host);
pmg_status pmg_woodbury_destroy(pmg_woodbury *w);

/* ------------------------------------------------------------------------------------------------------ */
/* VecSetRandomStandardNormal (src/parmgmc.c:70-116) on the counter-based source: entry r gets the (r&1)    */
/* branch of the Box-Muller pair with Philox counter {r>>1, sweep}, key = seed.                            */
/* ------------------------------------------------------------------------------------------------------ */
pmg_status pmg_vec_set_random_standard_normal(int64_t n, double *x_dev, uint64_t seed, uint64_t counter, void *stream);

/* MakeObservationMats (src/obs.c:135-180) on the unit-cube DMDA: ball-average observations for the low-rank update.
   Column i of B = (lumped mass h^dim) / (ball volume) inside the ball of radius radii[i] around coords[dim*i..],
   S = 1/sigma2, f = B (S o obsvals) (f_host / obsvals may be NULL).  Host arrays; rows = the planes
   [kz0, kz0+nz_owned) in natural order (a z-slab builds only its own rows); B is column-major with nobs columns. */
pmg_status pmg_make_observation_mats_dmda(int32_t nx, int32_t ny, int32_t nzg, int32_t kz0, int32_t nz_owned, int32_t nobs, double sigma2, const double *coords, const double *radii, const double *obsvals, double *B_host, double *S_host, double *f_host);

/* Measurement aid, no reference counterpart: ONE launch of c = a + 0.5 b over n doubles (n even, device vectors 16-byte
   aligned) with the colour sweep's access mix -- two streams read, one written, 16 bytes per lane, a tile per workgroup --
   and nothing else.  Timed by bench.py beside the sweep: the bandwidth a kernel of that mix reaches on the device at hand,
   reported next to the 8 TB/s vendor peak the roofline fraction is taken against. */
pmg_status pmg_stream_triad(int64_t n, const double *a_dev, const double *b_dev, double *c_dev, void *stream);

/* ---- chain diagnostics (host arrays, host arithmetic -- as in the reference) ---------------------------------- */
/* Autocorrelation (src/iact.c:17-47): acf[0..n) of the scalar series x[0..n) via a zero-padded FFT */
pmg_status pmg_autocorrelation(int64_t n, const double *x_host, double *acf_host);
/* IACT (src/iact.c:73-92; examples/ex2.c:107): integrated autocorrelation time with the automatic window
   M >= 5 tau(M); acf_host may be NULL; *valid = (500 tau <= n), may be NULL */
pmg_status pmg_iact(int64_t n, const double *x_host, double *tau, double *acf_host, int *valid);
/* EstimateCovarianceMatErrors (src/stats.c:94-117; examples/ex6.c:193): samples = host rows of length n ordered
   {sample 0 of chain 0, sample 0 of chain 1, ..., sample 1 of chain 0, ...}; errs[samples_per_chain] =
   ||C_i - A^-1||_F / ||A^-1||_F with the unbiased covariance over the chains at sample index i */
pmg_status pmg_estimate_covariance_errors(int32_t n, const int32_t *rowptr_host, const int32_t *colidx_host, const double *vals_host, int32_t chains, int32_t samples_per_chain, const double *samples_host, double *errs_host);

/* ------------------------------------------------------------------------------------------------------ */
/* The registration boundary without PETSc: PCRegister / PCSetType / pc->ops / PCSetSampleCallback /        */
/* PCSHELL / KSPRICHARDSON on raw device arrays (reference src/parmgmc.c:44-54,118-151; examples/ex1.c,     */
/* ex3.c, ex8.c).  With PETSc present the same constructors are bound through the adapter of INTEGRATION.md. */
/* ------------------------------------------------------------------------------------------------------ */
typedef struct pmg_pc_s  *pmg_pc;
typedef struct pmg_mat_s *pmg_mat;

/* ParMGMCInitialize / ParMGMCFinalize (src/parmgmc.c:118-137): registers "sorgibbs", "mcgibbs", "gamgmc",
   "cholsampler", "parsor" (include/parmgmc/parmgmc.h:26-31) and "shell". */
pmg_status pmg_initialize(void);
pmg_status pmg_finalize(void);
/* PCRegister(name, ctor) (src/parmgmc.c:44-54); the constructor fills the ops of a fresh PC. */
pmg_status pmg_pc_register(const char *name, pmg_status (*ctor)(pmg_pc));
/* seed of the library-wide random source (PetscRandomSetSeed on ParMGMCGetPetscRandom, src/parmgmc.c:56-68);
   every PC draws from its own counter-based stream derived from it */
pmg_status pmg_set_seed(uint64_t seed);
/* PetscOptionsSetValue / clear: option names as in the reference, e.g. "-pc_mcgibbs_omega" (src/pc_mcgibbs.c:197),
   "-pc_sorgibbs_forward" (src/pc_sorgibbs.c:271), "-gamgmc_mg_levels_ksp_max_it" (src/pc_gamgmc.c:324) */
pmg_status pmg_options_set_value(const char *name, const char *value);
pmg_status pmg_options_clear(void);

/* operators: a MATSEQAIJ as borrowed host CSR, or the DMDA operator of MatAssembleShiftedLaplaceFD */
pmg_status pmg_mat_create_csr(int32_t n, const int32_t *rowptr_host, const int32_t *colidx_host, const double *vals_host, pmg_mat *mat);
pmg_status pmg_mat_create_dmda(int32_t nx, int32_t ny, int32_t nz, double kappa, pmg_mat *mat);
/* MatCreateLRC(A, B, S, NULL, &Alrc) (examples/ex4.c, src/obs.c:176): A + B diag(S) B^T with B n x k column-major
   and S k entries as borrowed host arrays; mcgibbs / sorgibbs / cholsampler / gamgmc honour it as the reference does
   (src/mc_sor.c:572-595, src/pc_chols.c:119-153, src/pc_gamgmc.c:157-196) */
pmg_status pmg_mat_create_lrc(pmg_mat A, int32_t k, const double *B_host, const double *S_host, pmg_mat *mat);
pmg_status pmg_mat_get_size(pmg_mat mat, int32_t *n);
pmg_status pmg_mat_destroy(pmg_mat *mat);

/* PCCreate / PCSetType / PCSetOptionsPrefix / PCSetOperators / PCSetFromOptions / PCSetUp / PCView / PCReset /
   PCDestroy.  The operator is borrowed, not referenced (src/pc_sorgibbs.c:35-38). */
pmg_status pmg_pc_create(pmg_pc *pc);
pmg_status pmg_pc_set_type(pmg_pc pc, const char *type);
pmg_status pmg_pc_get_type(pmg_pc pc, char *buf, int32_t len);
pmg_status pmg_pc_set_options_prefix(pmg_pc pc, const char *prefix);
pmg_status pmg_pc_set_operators(pmg_pc pc, pmg_mat mat);
pmg_status pmg_pc_set_from_options(pmg_pc pc);
pmg_status pmg_pc_setup(pmg_pc pc);
pmg_status pmg_pc_view(pmg_pc pc, char *buf, int32_t len);
pmg_status pmg_pc_reset(pmg_pc pc);
pmg_status pmg_pc_destroy(pmg_pc *pc);
/* PCApply: sorgibbs (zero y + one sample, src/pc_sorgibbs.c:105-113), cholsampler, parsor, shell; mcgibbs and
   gamgmc only provide applyrichardson (src/pc_mcgibbs.c:318-325) -> PMG_ERR_SUP, as PETSc reports. */
pmg_status pmg_pc_apply(pmg_pc pc, const double *b_dev, double *y_dev, void *stream);
/* PCApplyRichardson: `its` samples, callback after each; tolerances / work vector of the PETSc signature are
   ignored by every sampler (src/pc_sorgibbs.c:117-120) and therefore absent; *outits = its, *reason = 4
   (PCRICHARDSON_CONVERGED_ITS). */
pmg_status pmg_pc_apply_richardson(pmg_pc pc, const double *b_dev, double *y_dev, int32_t its, int guesszero, int32_t *outits, int32_t *reason, void *stream);
/* KSPSolve with -ksp_type richardson and KSP_NORM_NONE (examples/ex1.c:97-129): all max_it iterations in one
   applyrichardson call; guess_nonzero = KSPSetInitialGuessNonzero. */
pmg_status pmg_ksp_richardson_solve(pmg_pc pc, const double *b_dev, double *y_dev, int32_t max_it, int guess_nonzero, void *stream);
/* PCSetSampleCallback(pc, cb, ctx, deleter) (src/parmgmc.c:146-151): replaces and deletes a previous context;
   the deleter also runs in reset/destroy (src/pc_sorgibbs.c:153-156,173-176). */
pmg_status pmg_pc_set_sample_callback(pmg_pc pc, pmg_sample_callback cb, void *ctx, int (*deleter)(void *ctx));
/* Checkpoint / resume (the reference has none: chain state is the caller's Vec plus the RNG state; with the
   counter-based source that state is two integers): the effective Philox seed of this PC's stream and the
   counter of its next draw.  Restoring y and the counter continues a chain bit for bit. */
pmg_status pmg_pc_get_noise_state(pmg_pc pc, uint64_t *seed, uint64_t *counter);
pmg_status pmg_pc_set_noise_counter(pmg_pc pc, uint64_t counter);
/* programmatic setters of the headers under include/parmgmc/pc/ */
pmg_status pmg_pc_mcgibbs_set_omega(pmg_pc pc, double omega);
pmg_status pmg_pc_mcgibbs_set_sweep_type(pmg_pc pc, int type);
pmg_status pmg_pc_parsor_set_omega(pmg_pc pc, double omega);
pmg_status pmg_pc_parsor_set_iterations(pmg_pc pc, int32_t its);
pmg_status pmg_pc_parsor_apply_sor(pmg_pc pc, const double *b_dev, int32_t its, int zero_initial_guess, double *x_dev, void *stream);
/* PCPARSOR's result depends on the number of MPI ranks: every rank sweeps its rows in the order TOP, INT1, MID, INT2,
   BOT and reads old or new off-rank values depending on the ranks' colours (ParallelSORApply, src/pc_parsor.c:703-878;
   ParallelSORPartitionNodes :272-592; ColorProcessors :187-270).  set_partition makes this PC reproduce the sweep of
   `nparts` ranks owning the contiguous row blocks [row_starts[p], row_starts[p+1]) on one device (rows become the
   nodes of a data-flow graph whose levels are swept as colours); proc_colors = the ranks' colours or NULL for first
   fit in rank order (the reference's JP colouring is randomised and unpinned).  nparts = 0: single-rank order.
   get_partition_info (after set-up): number of levels, rank colours, row classes (0 INT, 1 TOP, 2 MID, 3 BOT). */
pmg_status pmg_pc_parsor_set_partition(pmg_pc pc, int32_t nparts, const int32_t *row_starts_host, const int32_t *proc_colors_host);
pmg_status pmg_pc_parsor_get_partition_info(pmg_pc pc, int32_t *nlevels, int32_t *proc_colors_host, int32_t *node_classes_host);
/* PCWOODBURY (src/woodbury.c): sampler for a MATLRC operator A + B S B^T built from any sampler of A plus a solver
   (a PC with `apply`) of A: set-up forms G = C (S^-1 + B^T C)^-1 with C = solver(B) (:21-91) and drops the solver;
   every sample adds B (sqrt(S) o eta) to the rhs, draws one sample of the inner sampler and applies y -= G (B^T y)
   (:263-289).  PCWoodburySetSolver / PCWoodburySetSampler (:185-213): the woodbury PC takes over the caller's
   reference (do not destroy the inner PC afterwards); options -pc_woodbury_solver <type>, -pc_woodbury_sampler <type>
   create them (:245-261); inner option prefixes are "pc_woodbury_solver_" and "pc_woodbury_sampler" (sic, :208). */
pmg_status pmg_pc_woodbury_set_solver(pmg_pc pc, pmg_pc solver);
pmg_status pmg_pc_woodbury_set_sampler(pmg_pc pc, pmg_pc sampler);
pmg_status pmg_pc_gamgmc_set_levels(pmg_pc pc, int32_t levels);
/* PCSHELL: PCShellSetApply / PCShellSetContext / PCShellGetContext (examples/ex3.c:59-67,128-131) */
pmg_status pmg_pc_shell_set_apply(pmg_pc pc, pmg_status (*apply)(pmg_pc pc, const double *x_dev, double *y_dev, void *stream));
pmg_status pmg_pc_shell_set_context(pmg_pc pc, void *ctx);
pmg_status pmg_pc_shell_get_context(pmg_pc pc, void **ctx);

#ifdef __cplusplus
}
#endif
#endif /* PARMGMC_HIP_H */
