/* Low-rank (MATLRC) support of the samplers -- host side (C11).
 *
 * For a precision A_post = A + B S B^T (B dense N x k, S = Sigma^-1 diagonal; Bayesian update with k observations)
 * the reference sweeps on the base matrix A and repairs every sweep with a rank-k Woodbury correction:
 *   - set-up, MCSORBuildLRCCorrection (src/mc_sor.c:480-544): C = M_A^-1 B column by column with ONE deterministic
 *     sweep from a zero guess per column (:499-510), T = B^T C + S^-1 (:514-527), Sb = T^-1 (:528-533),
 *     Bb = C Sb (:535); one Bb per sweep direction (src/mc_sor.c:578-590);
 *   - after every directional sweep, MCSORPostSOR_LRC (src/mc_sor.c:101-112): y -= Bb (B^T y);
 *   - the noisy right-hand side gets the extra term B (sqrt(S) o eta), eta ~ N(0, I_k):
 *     PrepareRHS_LRC (src/pc_mcgibbs.c:130-140), PCSORGibbsSample (src/pc_sorgibbs.c:86-90).
 * All N x k operands live on the device in the sampler's storage layout; the k x k inverse is formed on the host
 * (exact LU; the reference's KSPMatSolve on a k x k system converges to machine precision in <= k GMRES steps).
 */
#include "pmg_internal.h"
#include <math.h>
#include <stdlib.h>

struct pmg_lrc_s {
  int      k;
  int64_t  ld;
  double  *B, *Bb[2]; /* device, ld x k column-major; Bb[0] forward, Bb[1] backward */
  double  *S, *sqrtS; /* device, k */
  double  *wk, *eta, *partial, *beff, *col; /* device work space */
  /* row-compact form (ns > 0): only the support rows of B, Bb are kept -- ball observations touch << N rows
     (src/obs.c:39-50) and one sweep from a zero guess widens the support by a layer per colour only */
  int64_t  ns;
  int64_t *rows;             /* device, ns layout positions, ascending */
  double  *Bc, *Bbc[2];      /* device, ns x k column-major */
  double  *saved;            /* device, ns: the right-hand side entries under the noise term */
  double  *b_mod;            /* the vector whose support rows currently carry the noise term */
  int      restore_pending;  /* pmg_lrc_rhs_done has been called: the saved entries go back with the next pass over the support rows
                                (the post-correction's update kernel), or by themselves if something else comes first */
  /* Round 4 built the per-sweep chain in fewer launches -- the noise term in ONE kernel (draw + scale + B eta) instead of
     three, the restore of the right-hand side inside the post-correction's update, one workgroup for B^T y and its update on
     supports of at most 4096 rows -- bit-identical (tests/test_gpu_lrc_fused.py) and 101 -> 65 launches per 257^3 sample,
     but NOT faster: 0.845 ms per sample against 0.826 for the chain of small kernels on the same box (either fusion alone:
     0.853 / 0.870; tools/lrcbench.py, three interleaved runs, gpurun_out/r4_lrc3.log).  The chain of small kernels stays the
     default; PMG_LRC_FUSED=1 (read when the object is built) selects the fused forms. */
  int      unfused_rhs, unfused_restore;
  int      unfused;
  /* the repair in front of a residual (down leg of a V-cycle, pmg_lrc_expect_residual) also leaves the partial sums of B^T y_new
     in partial2; the residual's low-rank term then starts from them instead of passing over the support rows again */
  double       *partial2;
  const double *bty_vec; /* the vector partial2 belongs to, NULL = none */
  int           want_bty, bty_in_post; /* PMG_LRC_BTY=1 (not the default: no gain measured) */
  /* noise terms drawn ahead (pmg_lrc_preset_eta): pre_eta + i * pre_stride = sqrt(S) o eta(pre_seed, pre_ctr0 + i), i < pre_n */
  const double *pre_eta;
  uint64_t      pre_seed, pre_ctr0;
  int           pre_n;
  int64_t       pre_stride;
  int      reduce_in_axpy; /* the partial sums of B^T y are added by the update kernel that consumes them (default); PMG_LRC_REDUCE=0: lrc_reduce_kernel */
  int      restore_in_btx; /* the saved right-hand side entries go back in the B^T y pass of the repair (default); PMG_LRC_RESTORE=0: a kernel of their own */
  int      empty;            /* this rank's rows do not meet the support of B at all (row-distributed operator) */
  pmg_lrc_reduce_fn reduce;  /* row-distributed operator: sum of the k-vectors over the ranks */
  void             *rctx;
};

void pmg_lrc_destroy(pmg_lrc *p)
{
  if (!p || !*p) return;
  pmg_lrc l = *p;
  pmg_dev_free(l->B);
  pmg_dev_free(l->Bb[0]);
  pmg_dev_free(l->Bb[1]);
  pmg_dev_free(l->S);
  pmg_dev_free(l->sqrtS);
  pmg_dev_free(l->wk);
  pmg_dev_free(l->eta);
  pmg_dev_free(l->partial);
  pmg_dev_free(l->partial2);
  pmg_dev_free(l->beff);
  pmg_dev_free(l->col);
  pmg_dev_free(l->rows);
  pmg_dev_free(l->Bc);
  pmg_dev_free(l->Bbc[0]);
  pmg_dev_free(l->Bbc[1]);
  pmg_dev_free(l->saved);
  free(l);
  *p = NULL;
}

/* in-place inverse of a small dense matrix (column-major k x k) by Gauss-Jordan with partial pivoting */
int pmg_invert_small(int k, double *a, double *inv)
{
  for (int i = 0; i < k * k; ++i) inv[i] = 0.0;
  for (int i = 0; i < k; ++i) inv[i + k * i] = 1.0;
  for (int c = 0; c < k; ++c) {
    int    piv = c;
    double mx  = fabs(a[c + k * c]);
    for (int r = c + 1; r < k; ++r)
      if (fabs(a[r + k * c]) > mx) {
        mx  = fabs(a[r + k * c]);
        piv = r;
      }
    if (mx == 0.0) return 1;
    if (piv != c)
      for (int j = 0; j < k; ++j) {
        double t = a[c + k * j]; a[c + k * j] = a[piv + k * j]; a[piv + k * j] = t;
        t = inv[c + k * j]; inv[c + k * j] = inv[piv + k * j]; inv[piv + k * j] = t;
      }
    const double d = 1.0 / a[c + k * c];
    for (int j = 0; j < k; ++j) {
      a[c + k * j] *= d;
      inv[c + k * j] *= d;
    }
    for (int r = 0; r < k; ++r)
      if (r != c) {
        const double f = a[r + k * c];
        if (f != 0.0)
          for (int j = 0; j < k; ++j) {
            a[r + k * j] -= f * a[c + k * j];
            inv[r + k * j] -= f * inv[c + k * j];
          }
      }
  }
  return 0;
}

/* switch to the row-compact form when the joint support of B, Bb[0], Bb[1] is at most a quarter of the rows */
static pmg_status lrc_compact(pmg_lrc l)
{
  unsigned char *mask_dev = NULL, *mask = (unsigned char *)malloc((size_t)l->ld);
  PMG_CHECK(mask, PMG_ERR_MEM, "out of host memory");
  pmg_status st = pmg_dev_alloc((void **)&mask_dev, (size_t)l->ld);
  if (!st && pmgk_lrc_mark_rows(l->ld, l->k, l->B, l->Bb[0], l->Bb[1], l->ld, mask_dev, NULL)) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "kernel launch failed");
  if (!st && hipMemcpy(mask, mask_dev, (size_t)l->ld, hipMemcpyDeviceToHost) != hipSuccess) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "download failed");
  pmg_dev_free(mask_dev);
  int64_t ns = 0;
  for (int64_t r = 0; r < l->ld && !st; ++r) ns += mask[r];
  if (!st && ns == 0 && l->reduce) { /* nothing of B on this rank: only the (collective) reductions remain */
    free(mask);
    l->empty = 1;
    pmg_dev_free(l->B);
    pmg_dev_free(l->Bb[0]);
    pmg_dev_free(l->Bb[1]);
    pmg_dev_free(l->col);
    pmg_dev_free(l->beff);
    l->B = l->Bb[0] = l->Bb[1] = l->col = l->beff = NULL;
    return PMG_SUCCESS;
  }
  if (st || ns == 0 || ns > l->ld / 4) {
    free(mask);
    return st;
  }
  int64_t *rows = (int64_t *)malloc(sizeof(int64_t) * (size_t)ns);
  if (!rows) {
    free(mask);
    PMG_FAIL(PMG_ERR_MEM, "out of host memory");
  }
  for (int64_t r = 0, q = 0; r < l->ld; ++r)
    if (mask[r]) rows[q++] = r;
  free(mask);
  st = pmg_dev_upload((void **)&l->rows, rows, sizeof(int64_t) * (size_t)ns);
  free(rows);
  const size_t cb = sizeof(double) * (size_t)ns * (size_t)l->k;
  if (!st) st = pmg_dev_alloc((void **)&l->Bc, cb);
  if (!st) st = pmg_dev_alloc((void **)&l->Bbc[0], cb);
  if (!st) st = pmg_dev_alloc((void **)&l->Bbc[1], cb);
  if (!st) st = pmg_dev_alloc((void **)&l->saved, sizeof(double) * (size_t)ns);
  if (!st && pmgk_lrc_rows_nblocks(ns) > pmgk_lrc_nblocks(l->ld)) { /* the row-compact kernels cut the support into smaller blocks than the dense ones the rows */
    pmg_dev_free(l->partial);
    l->partial = NULL;
    st         = pmg_dev_alloc((void **)&l->partial, sizeof(double) * (size_t)pmgk_lrc_rows_nblocks(ns) * (size_t)l->k);
  }
  if (!st) st = pmg_dev_alloc((void **)&l->partial2, sizeof(double) * (size_t)pmgk_lrc_rows_nblocks(ns) * (size_t)l->k);
  if (!st && (pmgk_lrc_gather_rows(ns, l->k, l->B, l->ld, l->rows, l->Bc, NULL) || pmgk_lrc_gather_rows(ns, l->k, l->Bb[0], l->ld, l->rows, l->Bbc[0], NULL) || pmgk_lrc_gather_rows(ns, l->k, l->Bb[1], l->ld, l->rows, l->Bbc[1], NULL))) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "kernel launch failed");
  if (!st && hipDeviceSynchronize() != hipSuccess) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "device error while compacting the low-rank factors");
  if (st) return st;
  l->ns = ns;
  pmg_dev_free(l->B); /* the dense copies are not needed any more */
  pmg_dev_free(l->Bb[0]);
  pmg_dev_free(l->Bb[1]);
  pmg_dev_free(l->col);
  pmg_dev_free(l->beff);
  l->B = l->Bb[0] = l->Bb[1] = l->col = l->beff = NULL;
  return PMG_SUCCESS;
}

/* the correction from B already in the sampler's layout on the device (ld x k, zero outside the rows it touches);
   B_lay_dev is copied.  reduce != NULL: the k-vectors B^T y are partial sums of a row-distributed operator and
   reduce(rctx, wk_dev, count, stream) must turn `count` device doubles into their sum over all ranks (same result on
   every rank) -- the hook of the distributed samplers. */
pmg_status pmg_lrc_build_dev(pmg_lrc *out, int32_t k, int64_t ld, const double *B_lay_dev, const double *S_host, pmg_det_sweep_fn det, void *ctx, pmg_lrc_reduce_fn reduce, void *rctx)
{
  PMG_CHECK(out && B_lay_dev && S_host && det, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(k >= 1 && k <= 64, PMG_ERR_ARG_OUTOFRANGE, "rank k = %d (1..64 supported)", k);
  *out      = NULL;
  pmg_lrc l = (pmg_lrc)calloc(1, sizeof *l);
  PMG_CHECK(l, PMG_ERR_MEM, "out of host memory");
  l->k      = k;
  l->ld     = ld;
  l->reduce = reduce;
  l->rctx   = rctx;
  {
    const char *e = getenv("PMG_LRC_FUSED"); /* 1: both fusions; 2: the noise term only; 3: the restore only */
    l->unfused         = !(e && e[0] == '1');
    l->unfused_rhs     = !(e && (e[0] == '1' || e[0] == '2'));
    l->unfused_restore = !(e && (e[0] == '1' || e[0] == '3'));
    const char *er     = getenv("PMG_LRC_RESTORE");
    l->restore_in_btx  = l->unfused_restore && !(er && er[0] == '0');
    const char *ed     = getenv("PMG_LRC_REDUCE");
    l->reduce_in_axpy  = !(ed && ed[0] == '0');
    const char *eb     = getenv("PMG_LRC_BTY");
    l->bty_in_post     = eb && eb[0] == '1'; /* measured at 257^3, k = 3: 0.741-0.746 ms per sample with it, 0.737-0.745 without (one launch less per level, but blocks of 1024 rows with the update's and the sums' fetch chains one behind the other): not the default */
  }
  double sq[64];
  for (int c = 0; c < k; ++c) sq[c] = sqrt(fabs(S_host[c])); /* VecSqrtAbs(sqrtS), src/pc_mcgibbs.c:240-242 */
  pmg_status st = pmg_dev_alloc((void **)&l->B, sizeof(double) * (size_t)ld * k);
  if (!st && hipMemcpy(l->B, B_lay_dev, sizeof(double) * (size_t)ld * k, hipMemcpyDeviceToDevice) != hipSuccess) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "copy failed");
  if (!st) st = pmg_dev_upload((void **)&l->S, S_host, sizeof(double) * (size_t)k);
  if (!st) st = pmg_dev_upload((void **)&l->sqrtS, sq, sizeof(double) * (size_t)k);
  for (int d = 0; d < 2 && !st; ++d) st = pmg_dev_alloc((void **)&l->Bb[d], sizeof(double) * (size_t)ld * k);
  if (!st) st = pmg_dev_alloc((void **)&l->wk, sizeof(double) * 64 * 64); /* k values per sweep; k*k while T is formed */
  if (!st) st = pmg_dev_alloc((void **)&l->eta, sizeof(double) * 64);
  if (!st) st = pmg_dev_alloc((void **)&l->partial, sizeof(double) * (size_t)pmgk_lrc_nblocks(ld) * k);
  if (!st) st = pmg_dev_alloc((void **)&l->beff, sizeof(double) * (size_t)ld);
  if (!st) st = pmg_dev_alloc((void **)&l->col, sizeof(double) * (size_t)ld * k); /* C = M_A^-1 B */
  double *T = (double *)malloc(sizeof(double) * (size_t)k * k), *Sb = (double *)malloc(sizeof(double) * (size_t)k * k), *Sb_dev = NULL;
  if (!st && (!T || !Sb)) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
  if (!st) st = pmg_dev_alloc((void **)&Sb_dev, sizeof(double) * (size_t)k * k);
  for (int d = 0; d < 2 && !st; ++d) {
    const int dir = d == 0 ? PMG_SOR_FORWARD_SWEEP : PMG_SOR_BACKWARD_SWEEP;
    for (int c = 0; c < k && !st; ++c) { /* C(:,c) = one deterministic sweep on B(:,c) from x = 0, src/mc_sor.c:499-510 */
      if (hipMemsetAsync(l->col + ld * c, 0, sizeof(double) * (size_t)ld, NULL) != hipSuccess) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "memset failed");
      if (!st) st = det(ctx, dir, l->B + ld * c, l->col + ld * c, NULL);
    }
    for (int c = 0; c < k && !st; ++c) /* T(:,c) = B^T C(:,c), src/mc_sor.c:514 */
      if (pmgk_lrc_btx(ld, k, l->B, ld, l->col + ld * c, l->partial, NULL, l->wk + (size_t)k * c, NULL)) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "kernel launch failed");
    if (!st && l->reduce) st = l->reduce(l->rctx, l->wk, k * k, NULL);
    if (!st && hipMemcpy(T, l->wk, sizeof(double) * (size_t)k * k, hipMemcpyDeviceToHost) != hipSuccess) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "download failed");
    if (st) break;
    for (int c = 0; c < k; ++c) T[c + (size_t)k * c] += 1.0 / S_host[c]; /* + S^-1, src/mc_sor.c:525-527 */
    if (pmg_invert_small(k, T, Sb)) st = pmg_set_error(PMG_ERR_LIB, __FILE__, __LINE__, "S^-1 + B^T M^-1 B is singular");
    if (!st && hipMemcpy(Sb_dev, Sb, sizeof(double) * (size_t)k * k, hipMemcpyHostToDevice) != hipSuccess) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "upload failed");
    if (!st && pmgk_lrc_gemm_small(ld, k, l->col, ld, Sb_dev, l->Bb[d], NULL)) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "kernel launch failed"); /* Bb = C Sb, :535 */
    if (!st && hipDeviceSynchronize() != hipSuccess) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "device error while building the low-rank correction");
  }
  free(T);
  free(Sb);
  pmg_dev_free(Sb_dev);
  if (!st && !getenv("PMG_LRC_DENSE")) st = lrc_compact(l);
  if (st) {
    pmg_lrc_destroy(&l);
    return st;
  }
  *out = l;
  return PMG_SUCCESS;
}

/* the same from a host matrix in natural numbering: pos[r] = layout position of natural row r */
pmg_status pmg_lrc_build(pmg_lrc *out, int32_t k, int64_t ld, int32_t n, const double *B_nat_host, const int64_t *pos, const double *S_host, pmg_det_sweep_fn det, void *ctx)
{
  PMG_CHECK(out && B_nat_host && pos && S_host && det, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(k >= 1 && k <= 64, PMG_ERR_ARG_OUTOFRANGE, "rank k = %d (1..64 supported)", k);
  double *Bl = (double *)calloc((size_t)ld * k, sizeof(double)), *Bd = NULL;
  PMG_CHECK(Bl, PMG_ERR_MEM, "out of host memory");
  for (int c = 0; c < k; ++c)
    for (int32_t r = 0; r < n; ++r) Bl[pos[r] + ld * c] = B_nat_host[r + (size_t)n * c];
  pmg_status st = pmg_dev_upload((void **)&Bd, Bl, sizeof(double) * (size_t)ld * k);
  free(Bl);
  if (!st) st = pmg_lrc_build_dev(out, k, ld, Bd, S_host, det, ctx, NULL, NULL);
  pmg_dev_free(Bd);
  return st;
}

/* the saved right-hand side entries go back by themselves (nothing that passes over the support rows came in between) */
static pmg_status lrc_flush_restore(pmg_lrc l, void *stream)
{
  if (l->restore_pending) {
    PMG_KERNEL(pmgk_lrc_scatter_rows(l->ns, l->rows, l->saved, l->b_mod, stream));
    l->restore_pending = 0;
    l->b_mod           = NULL;
  }
  return PMG_SUCCESS;
}

/* one workgroup does B^T y and the update that consumes it (kernels_lrc.hip) when the support is one block of rows, the
   update's row set is small and nothing has to be summed over ranks in between */
static int lrc_small(pmg_lrc l, int64_t ns2) { return l->ns > 0 && l->ns <= pmgk_lrc_rows_per_block() && ns2 <= 16384 && !l->reduce && !l->unfused; }

/* b_eff = b + B (sqrt(S) o eta), eta = row-stream normals of (seed + tag, counter); returns the device vector to
   sweep with.  Row-compact form: the noise term is added to the support rows of b IN PLACE (old values saved) and
   b itself is returned -- pmg_lrc_rhs_done puts the saved values back after the sweep, bit for bit.  PMG_LRC_FUSED=1: ONE launch in the
   row-compact form (every block draws the k normals itself). */
pmg_status pmg_lrc_rhs(pmg_lrc l, const double *b_lay, uint64_t seed, uint64_t counter, const double **beff, void *stream)
{
  if (l->empty) {
    *beff = b_lay;
    return PMG_SUCCESS;
  }
  const uint64_t nseed = pmg_lrc_noise_seed(seed);
  l->bty_vec           = NULL;
  if (l->ns) {
    PMG_CALL(lrc_flush_restore(l, stream));
    PMG_CHECK(!l->b_mod, PMG_ERR_ARG_WRONGSTATE, "pmg_lrc_rhs_done missing");
    l->b_mod = (double *)b_lay;
    *beff    = b_lay;
    if (!l->unfused_rhs) {
      PMG_KERNEL(pmgk_lrc_rhs_rows(l->ns, l->k, l->Bc, l->rows, l->sqrtS, nseed, counter, l->b_mod, l->saved, stream)); /* VecSetRandomStandardNormal, VecPointwiseMult, MatMultAdd: src/pc_mcgibbs.c:130-140 */
      return PMG_SUCCESS;
    }
  }
  const double *eta = l->eta;
  if (l->pre_eta && nseed == l->pre_seed && counter >= l->pre_ctr0 && counter - l->pre_ctr0 < (uint64_t)l->pre_n) eta = l->pre_eta + (int64_t)(counter - l->pre_ctr0) * l->pre_stride; /* drawn with the cycle's other noise terms: the same numbers */
  else PMG_KERNEL(pmgk_fill_normal_rows_scaled(l->k, nseed, counter, l->sqrtS, l->eta, stream)); /* VecSetRandomStandardNormal(pg->w), VecPointwiseMult(w, w, sqrtS) */
  if (l->ns) {
    PMG_KERNEL(pmgk_lrc_axpy_rows(l->ns, l->k, l->Bc, l->rows, eta, 1.0, l->b_mod, l->saved, stream)); /* MatMultAdd(B, w, rhs, rhs) */
    return PMG_SUCCESS;
  }
  PMG_KERNEL(pmgk_lrc_axpy_cols(l->ld, l->k, l->B, l->ld, eta, 1.0, b_lay, l->beff, stream));  /* MatMultAdd(B, w, rhs, rhs)      */
  *beff = l->beff;
  return PMG_SUCCESS;
}

/* after the sweep that used the vector of pmg_lrc_rhs: the saved entries are put back by the next pass over the support
   rows -- pmg_lrc_post's update, which every sampler calls next -- or by a kernel of their own */
pmg_status pmg_lrc_rhs_done(pmg_lrc l, void *stream)
{
  if (l->ns && l->b_mod) {
    l->restore_pending = 1;
    if (l->unfused_restore && !l->restore_in_btx) PMG_CALL(lrc_flush_restore(l, stream));
  }
  return PMG_SUCCESS;
}

/* wk = scale o (B^T x) over the support rows: partial sums per block of rows, then their sum in a fixed order */
/* up to 8 columns (two per wavefront of the update's workgroups); measured at 257^3: k = 3 0.754 -> 0.730 ms per sample, but
   k = 17 1.249 -> 1.337: five rounds of column sums in every block cost more than the launch they replace */
static int lrc_reduce_later(pmg_lrc l) { return l->reduce_in_axpy && !l->reduce && l->k <= 8; }

/* partial_only: the sums stay per block in l->partial for pmgk_lrc_reduce_axpy_rows */
static pmg_status lrc_btx_compact(pmg_lrc l, const double *x_lay, const double *scale, const double *save, double *w, int partial_only, void *stream)
{
  PMG_KERNEL(pmgk_lrc_btx_rows(l->ns, l->k, l->Bc, l->rows, x_lay, l->partial, scale, partial_only ? NULL : l->wk, save, w, stream));
  return PMG_SUCCESS;
}

/* r -= B (S o (B^T x)): the low-rank part of MatMult(A_post, x) in a residual r = b - A_post x (what PCMG's
   residual operator is pointed at for MATLRC levels, src/pc_gamgmc.c:186-194) */
pmg_status pmg_lrc_residual_sub(pmg_lrc l, const double *x_lay, double *r_lay, void *stream)
{
  if (l->empty) { /* contribute zeros to the collective sum */
    PMG_HIP(hipMemsetAsync(l->wk, 0, sizeof(double) * (size_t)l->k, (hipStream_t)stream));
    return l->reduce(l->rctx, l->wk, l->k, stream);
  }
  if (l->ns) {
    PMG_CALL(lrc_flush_restore(l, stream));
    if (lrc_small(l, l->ns)) {
      PMG_KERNEL(pmgk_lrc_btx_axpy_small(l->ns, l->k, l->Bc, l->rows, x_lay, l->S, l->wk, l->ns, l->Bc, l->rows, -1.0, r_lay, NULL, NULL, stream));
      return PMG_SUCCESS;
    }
    const int later = lrc_reduce_later(l);
    const int have  = later && l->bty_vec == x_lay; /* the repair in front left the partial sums of B^T x */
    if (!have) PMG_CALL(lrc_btx_compact(l, x_lay, l->S, NULL, NULL, later, stream));
    if (l->reduce) PMG_CALL(l->reduce(l->rctx, l->wk, l->k, stream)); /* S scales every partial sum alike */
    if (later) PMG_KERNEL(pmgk_lrc_reduce_axpy_rows(l->ns, l->k, l->Bc, l->rows, pmgk_lrc_rows_nblocks(l->ns), have ? l->partial2 : l->partial, l->S, -1.0, r_lay, NULL, stream));
    else PMG_KERNEL(pmgk_lrc_axpy_rows(l->ns, l->k, l->Bc, l->rows, l->wk, -1.0, r_lay, NULL, stream));
    return PMG_SUCCESS;
  }
  PMG_KERNEL(pmgk_lrc_btx(l->ld, l->k, l->B, l->ld, x_lay, l->partial, l->S, l->wk, stream));
  if (l->reduce) PMG_CALL(l->reduce(l->rctx, l->wk, l->k, stream));
  PMG_KERNEL(pmgk_lrc_axpy_cols(l->ld, l->k, l->B, l->ld, l->wk, -1.0, r_lay, r_lay, stream));
  return PMG_SUCCESS;
}

/* the row-compact factors as the kernels use them (diagnostics): layout positions of the support rows (ascending) and the
   ns x k column-major blocks of B, Bb forward, Bb backward; NULL arrays query the sizes.  PMG_ERR_SUP for the dense form */
pmg_status pmg_lrc_get_compact(pmg_lrc l, int32_t *k, int64_t *ns, int64_t *rows_host, double *B_host, double *Bbf_host, double *Bbb_host)
{
  PMG_CHECK(l, PMG_ERR_ARG_NULL, "null handle");
  PMG_CHECK(l->ns > 0, PMG_ERR_SUP, "the low-rank factors of this level are kept dense");
  PMG_CALL(lrc_flush_restore(l, NULL));
  if (k) *k = l->k;
  if (ns) *ns = l->ns;
  const size_t cb = sizeof(double) * (size_t)l->ns * (size_t)l->k;
  if (rows_host) PMG_HIP(hipMemcpy(rows_host, l->rows, sizeof(int64_t) * (size_t)l->ns, hipMemcpyDeviceToHost));
  if (B_host) PMG_HIP(hipMemcpy(B_host, l->Bc, cb, hipMemcpyDeviceToHost));
  if (Bbf_host) PMG_HIP(hipMemcpy(Bbf_host, l->Bbc[0], cb, hipMemcpyDeviceToHost));
  if (Bbb_host) PMG_HIP(hipMemcpy(Bbb_host, l->Bbc[1], cb, hipMemcpyDeviceToHost));
  return PMG_SUCCESS;
}

/* what a directional sweep touches of this update on THIS rank: the rank k and the number of rows its passes run over --
   the support rows of the row-compact form, all ld rows of the dense form (*dense = 1), none (ns = 0) for a rank whose rows
   miss B's support altogether (row-distributed operator).  A plain query: it cannot fail and leaves no error message. */
void pmg_lrc_get_sizes(pmg_lrc l, int32_t *k, int64_t *ns, int *dense)
{
  const int is_dense = l && !l->empty && l->ns == 0;
  if (k) *k = l ? l->k : 0;
  if (ns) *ns = !l || l->empty ? 0 : (l->ns ? l->ns : l->ld);
  if (dense) *dense = is_dense;
}

/* The noise terms of the next n directional sweeps are in eta_dev already: eta_dev + i * stride holds the k numbers
   sqrt(S) o eta the sweep with (seed, counter0 + i) would draw (pmgk_fill_normal_batch with pmg_lrc_noise_seed(seed) and
   pmg_lrc_sqrtS), written by work queued on the stream the sweeps run on.  pmg_lrc_rhs takes them from there when it is
   called with a matching (seed, counter) and draws as before otherwise; eta_dev = NULL forgets. */
void pmg_lrc_preset_eta(pmg_lrc l, uint64_t seed, uint64_t counter0, int n, const double *eta_dev, int64_t stride)
{
  if (!l) return;
  l->pre_eta    = n > 0 ? eta_dev : NULL;
  l->pre_seed   = pmg_lrc_noise_seed(seed);
  l->pre_ctr0   = counter0;
  l->pre_n      = n;
  l->pre_stride = stride;
}
uint64_t      pmg_lrc_noise_seed(uint64_t seed) { return seed + 0x632BE59BD9B4E019ull; } /* the stream of the noise term beside the sweep's own (pmg_lrc_rhs) */
const double *pmg_lrc_sqrtS(pmg_lrc l) { return l ? l->sqrtS : NULL; }
int           pmg_lrc_rank(pmg_lrc l) { return l ? l->k : 0; }

/* on: the sweeps that follow are the pre-smoothing of a V-cycle level -- a residual of the same vector comes next (off: whatever
   the last repair left behind is forgotten) */
void pmg_lrc_expect_residual(pmg_lrc l, int on)
{
  if (!l) return;
  l->want_bty = on;
  if (!on) l->bty_vec = NULL;
}

/* 1: the update lives on one device (no reduction over ranks, rows on this rank) */
int pmg_lrc_is_local(pmg_lrc l) { return l && !l->reduce && !l->empty; }

/* The low-rank term of a residual, RESTRICTED: P^T (B S B^T x) = B_c (S B^T x) with B_c = P^T B the block of the next
   coarser level (src/pc_gamgmc.c:177-178).  lf: the fine level's update (x_fine in ITS layout), lc: the coarse level's
   (b_coarse in its layout); b_coarse -= B_c (S B_f^T x_fine).  Used behind the fused residual + restriction, which never
   forms the fine residual; equal to restricting r - B S B^T x up to rounding. */
pmg_status pmg_lrc_residual_sub_restricted(pmg_lrc lf, pmg_lrc lc, const double *x_fine_lay, double *b_coarse_lay, void *stream)
{
  PMG_CHECK(lf && lc && lf->k == lc->k, PMG_ERR_ARG_WRONG, "low-rank updates of two consecutive levels expected");
  /* S B_f^T x: per rank over its rows, summed over the ranks of a distributed level (a rank without rows contributes zeros) */
  if (lf->ns) PMG_CALL(lrc_flush_restore(lf, stream));
  if (lc->ns) PMG_CALL(lrc_flush_restore(lc, stream));
  if (!lf->empty && !lc->empty && lc->ns && lrc_small(lf, lc->ns)) {
    PMG_KERNEL(pmgk_lrc_btx_axpy_small(lf->ns, lf->k, lf->Bc, lf->rows, x_fine_lay, lf->S, lf->wk, lc->ns, lc->Bc, lc->rows, -1.0, b_coarse_lay, NULL, NULL, stream));
    return PMG_SUCCESS;
  }
  const int later = !lf->empty && lf->ns && !lc->empty && lc->ns && lrc_reduce_later(lf);
  const int have = later && lf->bty_vec == x_fine_lay;
  if (lf->empty) PMG_HIP(hipMemsetAsync(lf->wk, 0, sizeof(double) * (size_t)lf->k, (hipStream_t)stream));
  else if (have) { /* the repair in front left the partial sums of B_f^T x */
  } else if (lf->ns) PMG_CALL(lrc_btx_compact(lf, x_fine_lay, lf->S, NULL, NULL, later, stream));
  else PMG_KERNEL(pmgk_lrc_btx(lf->ld, lf->k, lf->B, lf->ld, x_fine_lay, lf->partial, lf->S, lf->wk, stream));
  if (lf->reduce) PMG_CALL(lf->reduce(lf->rctx, lf->wk, lf->k, stream));
  if (lc->empty) return PMG_SUCCESS; /* none of B_c's rows on this rank */
  if (later) PMG_KERNEL(pmgk_lrc_reduce_axpy_rows(lc->ns, lc->k, lc->Bc, lc->rows, pmgk_lrc_rows_nblocks(lf->ns), have ? lf->partial2 : lf->partial, lf->S, -1.0, b_coarse_lay, NULL, stream));
  else if (lc->ns) PMG_KERNEL(pmgk_lrc_axpy_rows(lc->ns, lc->k, lc->Bc, lc->rows, lf->wk, -1.0, b_coarse_lay, NULL, stream));
  else PMG_KERNEL(pmgk_lrc_axpy_cols(lc->ld, lc->k, lc->B, lc->ld, lf->wk, -1.0, b_coarse_lay, b_coarse_lay, stream));
  return PMG_SUCCESS;
}

/* y -= Bb_dir (B^T y), src/mc_sor.c:101-112 */
pmg_status pmg_lrc_post(pmg_lrc l, int dir, double *y_lay, void *stream)
{
  const int d = dir == PMG_SOR_FORWARD_SWEEP ? 0 : 1;
  l->bty_vec  = NULL;
  if (l->empty) {
    PMG_HIP(hipMemsetAsync(l->wk, 0, sizeof(double) * (size_t)l->k, (hipStream_t)stream));
    return l->reduce(l->rctx, l->wk, l->k, stream);
  }
  if (l->ns) {
    /* the right-hand side entries under the noise term go back in the update's pass over the support rows */
    const double *save = l->restore_pending && l->b_mod != y_lay ? l->saved : NULL;
    double       *bmod = l->b_mod;
    if (l->restore_pending && !save) PMG_CALL(lrc_flush_restore(l, stream));
    if (lrc_small(l, l->ns)) PMG_KERNEL(pmgk_lrc_btx_axpy_small(l->ns, l->k, l->Bc, l->rows, y_lay, NULL, l->wk, l->ns, l->Bbc[d], l->rows, -1.0, y_lay, save, bmod, stream));
    else {
      const int early = save && l->restore_in_btx; /* in the first pass over the support rows instead of the last */
      const int later = lrc_reduce_later(l) && (early || !save);
      PMG_CALL(lrc_btx_compact(l, y_lay, NULL, early ? save : NULL, early ? bmod : NULL, later, stream));
      if (l->reduce) PMG_CALL(l->reduce(l->rctx, l->wk, l->k, stream));
      if (later && l->want_bty && l->bty_in_post) {
        PMG_KERNEL(pmgk_lrc_reduce_axpy_btx_rows(l->ns, l->k, l->Bbc[d], l->rows, l->partial, -1.0, y_lay, l->Bc, l->partial2, stream));
        l->bty_vec = y_lay;
      } else if (later) PMG_KERNEL(pmgk_lrc_reduce_axpy_rows(l->ns, l->k, l->Bbc[d], l->rows, pmgk_lrc_rows_nblocks(l->ns), l->partial, NULL, -1.0, y_lay, NULL, stream));
      else if (save && !early) PMG_KERNEL(pmgk_lrc_axpy_restore_rows(l->ns, l->k, l->Bbc[d], l->rows, l->wk, -1.0, y_lay, save, bmod, stream));
      else PMG_KERNEL(pmgk_lrc_axpy_rows(l->ns, l->k, l->Bbc[d], l->rows, l->wk, -1.0, y_lay, NULL, stream));
    }
    if (save) {
      l->restore_pending = 0;
      l->b_mod           = NULL;
    }
    return PMG_SUCCESS;
  }
  PMG_KERNEL(pmgk_lrc_btx(l->ld, l->k, l->B, l->ld, y_lay, l->partial, NULL, l->wk, stream));
  if (l->reduce) PMG_CALL(l->reduce(l->rctx, l->wk, l->k, stream));
  PMG_KERNEL(pmgk_lrc_axpy_cols(l->ld, l->k, l->Bb[d], l->ld, l->wk, -1.0, y_lay, y_lay, stream));
  return PMG_SUCCESS;
}
