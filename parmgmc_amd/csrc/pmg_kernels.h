/* Internal launch interface between the host C layer and the HIP kernels (not part of the public C-ABI).
   Plain C so that the host side (gcc, C11) and the device side (hipcc) share one declaration. */
#ifndef PMG_KERNELS_H
#define PMG_KERNELS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Colour-partitioned storage of a vector on an nx*ny*nz vertex grid ("cvec").

   Points are split by parity c = (i+j+kg)&1 (kg = global plane index).  Each colour is stored as its own
   array of planes; a plane holds ny grid lines of `sx` doubles; along a line (j,k) the points of colour c are
   i = 2m + p, p = (c+j+kg)&1, and sit at m = 0,1,...  One ghost plane below and one above the nz owned
   planes hold the neighbouring device's boundary planes (multi-GPU) and are never read at a physical
   boundary.  Element (c, k, j, m) lives at  c*cs + (k+1)*sp + j*sx + m.  sx is a multiple of 16 doubles
   (128 B) so that every line starts on a cache line; pad slots hold zeros (the sweep stores 0 there). */
typedef struct {
  int32_t nx, ny, nz;  /* owned extent (nz = owned planes) */
  int32_t kz0, nzg;    /* global index of owned plane 0, global number of planes */
  int32_t sx;          /* line stride (doubles) */
  int64_t sp;          /* plane stride = ny*sx */
  int64_t cs;          /* colour stride = (nz+2)*sp */
} pmgk_grid_layout;

/* Operator + sweep parameters of the matrix-free 7-point kernel: the matrix of
   MatAssembleShiftedLaplaceFD (reference src/problems.c:14-75) generalised to 3-D.  Off-diagonals are -h2
   for every in-domain neighbour; the diagonal takes one of 7 values indexed by the number of in-domain
   neighbours, so idiag (= (1/d)*omega, reference src/mc_sor.c:114-124) and sqrtdiag (= sqrt|d| *
   sqrt((2-omega)/omega), reference src/pc_mcgibbs.c:142-153) are 7-entry tables computed on the host with
   the reference's rounding sequence. */
typedef struct {
  double   h2;
  double   one_minus_omega;
  double   idiag[8];
  double   sqrtdiag[8];
  double   diag[8];
  uint32_t key0, key1; /* Philox key = seed */
  uint64_t sweep;      /* noise counter of this sweep */
  int32_t  noisy;      /* 0: w = b (MCSORApply); 1: w = xi*sqrtdiag + b (PrepareRHS_Default) */
  int32_t  omega_is_one;
} pmgk_grid_op;

/* multi-GPU face planes: ghost planes of the OTHER colour to read (this rank's receive block) and destinations in
   the z-neighbours' receive blocks (peer memory) for the freshly swept planes 0 / nz-1; null = unused.
   full != 0: ONE launch sweeps all owned planes of the colour, face planes first (blockIdx.z 0, 1 -> planes 0, nz-1);
   the wavefronts of the face planes first wait until the flag words wlo / whi (this rank's block) have reached wval
   -- the neighbours' planes of the other colour have landed -- and, after their stores, the LAST of them raises the
   neighbours' flag words slo / shi to sval (counter: device word, zero between launches; err: set if a wait gives
   up).  The receive block must be fine-grained memory: the ghost planes are read in the kernel that waited. */
typedef struct {
  const double *glo, *ghi;
  double       *plo, *phi;
  int32_t       full;
  const uint64_t *wlo, *whi;
  uint64_t       wval;
  uint64_t      *slo, *shi;
  uint64_t       sval;
  unsigned      *counter, *err;
  unsigned long long *spins; /* device word (may be null): polls of the flag words that found them not yet raised, summed over
                                the face wavefronts of all launches -- 0 in steady state (the planes a colour needs were pushed
                                a whole colour pass earlier); what a multi-GPU bench record reports as halo_wait_polls */
} pmgk_grid_halo;
/* sweeps the colour-`color` points of the kcount owned planes kbegin, kbegin+kstride, ...; halo may be NULL */
int pmgk_grid_color_sweep(const pmgk_grid_layout *L, const pmgk_grid_op *op, int color, int kbegin, int kcount, int kstride, const pmgk_grid_halo *halo, const double *b_cvec, double *y_cvec, void *stream);
int pmgk_grid_to_cvec(const pmgk_grid_layout *L, const double *nat, double *cvec, void *stream);
int pmgk_grid_from_cvec(const pmgk_grid_layout *L, const double *cvec, double *nat, void *stream);
/* r = b - A y on cvecs (PCMGResidualDefault / src/pc_gamgmc.c:253-254) */
int pmgk_grid_residual(const pmgk_grid_layout *L, const pmgk_grid_op *op, const double *b_cvec, const double *y_cvec, double *r_cvec, void *stream);

/* Sliced-ELL multicolour operator ("colour-partitioned CSR"): rows permuted so that each colour is
   contiguous and starts on a 64-row slice boundary; slice s stores its off-diagonal entries column-major
   (entry j of lane l at soff[s] + j*64 + l) in the CSR storage order of the original row. */
typedef struct {
  int32_t        n;        /* rows (original) */
  int32_t        ld;       /* padded length of a permuted vector (multiple of 64) */
  int32_t        nslices;
  const int64_t *soff;     /* [nslices+1] device */
  const int32_t *swidth;   /* [nslices] device */
  const double  *vals;     /* device */
  const int32_t *cols;     /* device, permuted numbering */
  const double  *idiag;    /* [ld] device, 0 in pad rows */
  const double  *sqrtdiag; /* [ld] device */
  const double  *diag;     /* [ld] device */
  const int32_t *orig;     /* [ld] device: original row of permuted row, -1 in pad rows */
  int64_t        noise_row0; /* the noise of original row r is the row stream of noise_row0 + r (row block of a distributed matrix) */
} pmgk_sell;

int pmgk_sell_color_sweep(const pmgk_sell *S, int slice0, int nsl, double omega, int noisy, uint64_t seed, uint64_t sweep, const double *b, double *y, void *stream);
int pmgk_sell_residual(const pmgk_sell *S, const double *b, const double *y, double *r, void *stream);
int pmgk_permute_in(int32_t ld, const int32_t *orig, const double *nat, double *perm, void *stream);
int pmgk_permute_out(int32_t ld, const int32_t *orig, const double *perm, double *nat, void *stream);

/* generic CSR product on device arrays: y = alpha*A x + beta*y  (rows in any layout) */
int pmgk_csr_spmv(int32_t nrows, const int32_t *rowptr, const int32_t *colidx, const double *vals, double alpha, const double *x, double beta, double *y, void *stream);
/* y[rowpos[r]] (+)= alpha * sum_k vals[k] x[colidx[k]] over the rows of a CSR block whose output positions are
   given explicitly (transfer operators between level layouts); accumulate != 0 adds to y */
int pmgk_csr_spmv_rows(int32_t nrows, const int32_t *rowpos, const int32_t *rowptr, const int32_t *colidx, const double *vals, const double *x, double *y, int accumulate, double *zero, void *stream); /* zero != NULL: zero[rowpos[r]] = 0 as well */
/* dense lower Cholesky in place + W = L^-1 on the device (npad multiple of 32; MFMA f64 trailing updates) */
int pmgk_potrf_inverse(int32_t npad, double *A_colmajor, double *W_colmajor, double *T_scratch, double *Dinv_scratch, int *info_dev, void *stream);
int pmgk_pack_rowmajor(int32_t n, const double *in_colmajor, int64_t ld, int transpose, double *out_rowmajor, void *stream);
/* class-stencil form of a 27-point (9-point) Galerkin operator on an nx*ny*nzg grid: coef[27*cls + e], cls = position
   class (first / interior / last per direction), e = 9(dz+1) + 3(dy+1) + (dx+1); idiag = (1/d)*omega, sqrtdiag per
   class; device arrays.  Vectors hold the owned planes kz0 .. kz0+nz-1 between two ghost planes:
   element (i, j, k) at i + nx*(j + ny*(k - kz0 + 1)); single device: kz0 = 0, nz = nzg. */
typedef struct {
  int32_t       nx, ny, nz;
  int32_t       kz0, nzg;
  const double *coef, *idiag, *sqrtdiag;
} pmgk_st27;
typedef struct {
  int32_t nx, ny, nz, kz0, nzg;
} pmgk_st27_dims;
int pmgk_st27_sweep(const pmgk_st27 *S, int backward, double omega, int noisy, uint64_t seed, uint64_t sweep, const double *b, double *y, void *stream);
int pmgk_st27_sweep_phase(const pmgk_st27 *S, int backward, int phase, double omega, int noisy, uint64_t seed, uint64_t sweep, const double *b, double *y, void *stream);
int pmgk_st27_residual(const pmgk_st27 *S, const double *b, const double *y, double *r, void *stream);
/* single-device levels (kz0 = 0, nz = nzg), kernels_stencil27_pair.hip: one directional sweep OUT OF PLACE (y_out <- sweep(b,
   y_in), two launches: one per z-parity phase, the four in-plane colours fused) and the residual with dense paired loads;
   same bits as the per-colour kernels */
int pmgk_st27_sweep_pp(const pmgk_st27 *S, int backward, double omega, int noisy, uint64_t seed, uint64_t sweep, const double *b, const double *y_in, double *y_out, void *stream);
/* one z-parity phase of the out-of-place sweep, z-slabs included (the caller exchanges y_out's boundary planes between the phases) */
int pmgk_st27_sweep_pp_phase(const pmgk_st27 *S, int backward, int phase, double omega, int noisy, uint64_t seed, uint64_t sweep, const double *b, const double *y_in, double *y_out, void *stream);
int pmgk_st27_residual_pair(const pmgk_st27 *S, const double *b, const double *y, double *r, void *stream);
int pmgk_st27_restrict(const pmgk_st27_dims *F, const pmgk_st27_dims *C, const double *r, double *bc, void *stream);
/* fine planes kbegin .. kbegin+kcount-1 (global indices; may include the in-domain ghost planes of a slab) */
int pmgk_st27_prolong_add(const pmgk_st27_dims *F, const pmgk_st27_dims *C, int kbegin, int kcount, const double *ec, double *x, void *stream);
/* matrix-free Q1 transfers between a grid level (cvec, possibly a z-slab) and the next coarser level: C gives the
   coarse extents (global nzg, owned planes kz0 .. kz0+nz-1; coarse plane K belongs to the owner of fine plane 2K).
   cpos != NULL: coarse layout given by cpos[global natural coarse index] (single device only); cpos == NULL: the
   plane-padded natural layout of pmgk_st27.  The restriction reads r on the ghost planes of a slab; the prolongation
   covers the fine local planes kbegin .. kbegin+kcount-1 (-1 and nz are the ghost planes). */
int pmgk_q1_restrict(const pmgk_grid_layout *L, const pmgk_st27_dims *C, const int32_t *cpos, const double *r_cvec, double *bc, void *stream);
/* b_coarse = P^T (b - A y) in one launch (same bits as pmgk_grid_residual + pmgk_q1_restrict); -1 = not applicable, nothing launched */
int pmgk_grid_residual_restrict_applies(const pmgk_grid_layout *L, const pmgk_st27_dims *C, int have_lo2, int have_hi2); /* 1 / 0, launches nothing */
int pmgk_grid_residual_restrict(const pmgk_grid_layout *L, const pmgk_grid_op *op, const pmgk_st27_dims *C, const double *b_cvec, const double *y_cvec, const double *ylo2, const double *yhi2, double *bc, void *stream);
int pmgk_q1_prolong_add(const pmgk_grid_layout *L, const pmgk_st27_dims *C, const int32_t *cpos, int kbegin, int kcount, int only_color, const double *ec, double *x_cvec, void *stream);
/* triangular matrix-vector products of the coarse exact sampler (row-major n x n, lower or upper part) */
int pmgk_tri_gemv(int32_t n, int upper, const double *M, const double *x, const double *add, double *out, void *stream);
/* low-rank (MATLRC) pieces: M is n x k column-major with leading dimension ld, k <= 64 */
int pmgk_lrc_nblocks(int64_t n);
int pmgk_lrc_rows_per_block(void);
int pmgk_lrc_rows_nblocks(int64_t ns);
int pmgk_lrc_btx(int64_t n, int k, const double *M, int64_t ld, const double *y, double *partial, const double *scale, double *out, void *stream);
int pmgk_lrc_axpy_cols(int64_t n, int k, const double *M, int64_t ld, const double *coef, double sign, const double *in, double *out, void *stream);
int pmgk_lrc_gemm_small(int64_t n, int k, const double *Cm, int64_t ld, const double *Sb, double *Bb, void *stream);
/* row-compact form (support rows only): Mc is ns x k column-major, rows[q] the position of compact row q in the vectors */
int pmgk_lrc_mark_rows(int64_t n, int k, const double *A0, const double *A1, const double *A2, int64_t ld, unsigned char *mask, void *stream);
int pmgk_lrc_gather_rows(int64_t ns, int k, const double *M, int64_t ld, const int64_t *rows, double *Mc, void *stream);
int pmgk_lrc_reduce_axpy_rows(int64_t ns, int k, const double *Mc, const int64_t *rows, int nb, const double *partial, const double *scale, double sign, double *v, double *save, void *stream);
int pmgk_lrc_reduce_axpy_btx_rows(int64_t ns, int k, const double *Mb, const int64_t *rows, const double *partial_in, double sign, double *v, const double *Mc, double *partial_out, void *stream);
int pmgk_lrc_btx_rows(int64_t ns, int k, const double *Mc, const int64_t *rows, const double *y, double *partial, const double *scale, double *out, const double *save, double *w, void *stream); /* save != NULL: also w[rows] = save */
int pmgk_lrc_axpy_rows(int64_t ns, int k, const double *Mc, const int64_t *rows, const double *coef, double sign, double *v, double *save, void *stream);
int pmgk_lrc_scatter_rows(int64_t ns, const int64_t *rows, const double *save, double *v, void *stream);
/* fused forms (round 4): update + restore of the right-hand side; noise draw + scale + B eta;
   B^T y and the update that consumes it in one workgroup (small supports) */
int pmgk_lrc_axpy_restore_rows(int64_t ns, int k, const double *Mc, const int64_t *rows, const double *coef, double sign, double *v, const double *save, double *w, void *stream);
int pmgk_lrc_rhs_rows(int64_t ns, int k, const double *Mc, const int64_t *rows, const double *sqrtS, uint64_t seed, uint64_t sweep, double *b, double *save, void *stream);
int pmgk_lrc_btx_axpy_small(int64_t ns1, int k, const double *M1, const int64_t *rows1, const double *y, const double *scale, double *wk, int64_t ns2, const double *M2, const int64_t *rows2, double sign, double *v, const double *save, double *w, void *stream);
int pmgk_axpy(int64_t n, double alpha, const double *x, double *y, void *stream);
/* out[c] = sum over r = 0..nrows-1 (in that order) of in[r*count + c] */
int pmgk_fill_zero(double *p, int64_t n, void *stream); /* p 16-byte aligned */
int pmgk_stream_triad(int64_t n, const double *a, const double *b, double *c, void *stream); /* n even, 16-byte aligned */
int pmgk_sum_rows(int nrows, int count, const double *in, double *out, void *stream);
/* exchange in two launches: push = copy segments (src -> dst, dst in the neighbours' slots) then raise flag[q] to
   value[q] from the last block; pull = wait for flag[q] >= value[q], then copy segments out of my slots */
#define PMGK_XCH_MAXSEG 8
typedef struct {
  int           nseg;
  const double *src[PMGK_XCH_MAXSEG];
  double       *dst[PMGK_XCH_MAXSEG];
  int64_t       n[PMGK_XCH_MAXSEG];
  uint64_t     *flag[4]; /* null entries are skipped */
  uint64_t      value[4];
} pmgk_xch_args;
int pmgk_xch_push(const pmgk_xch_args *a, unsigned *counter, void *stream);
int pmgk_xch_push_dbg(const pmgk_xch_args *a, unsigned *counter, unsigned *dbg, void *stream); /* dbg[0..2]: pushes that raised their flags, last sequence numbers, a counter that did not start at zero */
/* all-gather over all-peer mappings: push = copy my block (n doubles at src) to dst[p] + dst_off for every peer
   p != me (device array of nranks pointers; null = skip), then raise flag[p] (device array) to value from the last
   block; wait = until myflags[p] >= value for every p != me */
int pmgk_allgather_push(int nranks, int me, const double *src, int64_t n, double *const *dst_dev, int64_t dst_off, uint64_t *const *flag_dev, uint64_t value, unsigned *counter, void *stream);
int pmgk_allgather_wait(int nranks, int me, const uint64_t *myflags, uint64_t value, unsigned *err, void *stream);
int pmgk_xch_pull(const pmgk_xch_args *a, unsigned *err, void *stream);
/* dst[q] = src[idx[q]];  dst[dst_idx[q]] = src[src_idx[q]]  (ghost updates of the row-block distributed sampler) */
int pmgk_gather_idx(int64_t n, const int32_t *idx, const double *src, double *dst, void *stream);
int pmgk_scatter_idx(int64_t n, const int32_t *src_idx, const int32_t *dst_idx, const double *src, double *dst, void *stream);
int pmgk_fill_normal_rows(int64_t n, uint64_t seed, uint64_t sweep, double *xi, void *stream);
#define PMGK_NORMAL_BATCH_MAX 64
int pmgk_fill_normal_batch(int nstreams, int64_t n, const uint64_t *seed, const uint64_t *sweep, const double *scale, double *xi, int64_t stride, void *stream); /* host arrays of seeds / sweeps */
int pmgk_fill_normal_rows_scaled(int64_t n, uint64_t seed, uint64_t sweep, const double *scale, double *xi, void *stream); /* xi = scale o z */

#ifdef __cplusplus
}
#endif
#endif
