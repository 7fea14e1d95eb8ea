/* Host-side internals of libparmgmc_hip (C11). */
#ifndef PMG_INTERNAL_H
#define PMG_INTERNAL_H
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/parmgmc_hip.h"
#include "pmg_kernels.h"

#define PMG_VERSION_STRING "0.1.0"

/* Records file:line + message for pmg_last_error_string() and returns `code` (PetscCheck analogue:
   reference code raises through PetscCheck(cond, comm, PETSC_ERR_*, fmt...), e.g. src/mc_sor.c:427). */
pmg_status pmg_set_error(pmg_status code, const char *file, int line, const char *fmt, ...);

#define PMG_FAIL(code, ...) return pmg_set_error((code), __FILE__, __LINE__, __VA_ARGS__)
#define PMG_CHECK(cond, code, ...) \
  do { \
    if (!(cond)) return pmg_set_error((code), __FILE__, __LINE__, __VA_ARGS__); \
  } while (0)
/* PetscCall analogue */
#define PMG_CALL(expr) \
  do { \
    pmg_status pmg_s_ = (expr); \
    if (pmg_s_) return pmg_s_; \
  } while (0)
#define PMG_HIP(expr) \
  do { \
    hipError_t pmg_e_ = (expr); \
    if (pmg_e_ != hipSuccess) return pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "%s: %s", #expr, hipGetErrorString(pmg_e_)); \
  } while (0)
#define PMG_KERNEL(expr) \
  do { \
    if ((expr) != 0) return pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "kernel launch failed: %s", #expr); \
  } while (0)

static inline int pmg_sweep_type_ok(int t) { return t == PMG_SOR_FORWARD_SWEEP || t == PMG_SOR_BACKWARD_SWEEP || t == PMG_SOR_SYMMETRIC_SWEEP; }

pmg_status pmg_grid_set_lowrank_dev(pmg_grid g, int32_t k, const double *B_cvec_dev, const double *S_host);
pmg_status pmg_grid_sweep_color_halo_cvec(pmg_grid g, int color, int noisy, int scaled, uint64_t seed, uint64_t counter, const pmgk_grid_halo *halo, const double *b, double *y, void *stream);
pmg_status pmg_grid_sweep_color_faces_cvec(pmg_grid g, int color, int noisy, int scaled, uint64_t seed, uint64_t counter, const pmgk_grid_halo *halo, const double *b, double *y, void *stream);
/* kernel-side description of a grid object (internal) */
pmg_status pmg_grid_get_kernel_layout(pmg_grid g, pmgk_grid_layout *L);
/* residual and Q1 restriction fused; *done = 0 when the fused kernel does not apply (the caller runs the two steps) */
int32_t    pmg_grid_line_stride(int32_t nx, int32_t ny, int32_t nzg); /* doubles per line of a colour array */
int        pmg_grid_residual_restrict_applies(pmg_grid g, const pmgk_st27_dims *C, int have_lo2, int have_hi2);
pmg_status pmg_grid_residual_restrict(pmg_grid g, const double *b, const double *y, const double *ylo2, const double *yhi2, const pmgk_st27_dims *C, double *b_coarse, int *done, void *stream);

#define PMG_XCH_MAXSEG 4
/* low-rank (MATLRC) helper shared by pmg_mcsor and pmg_grid (pmg_lrc.c); vectors in the sampler's layout */
typedef struct pmg_lrc_s *pmg_lrc;
typedef pmg_status (*pmg_det_sweep_fn)(void *ctx, int dir, const double *b_lay, double *y_lay, void *stream);
typedef pmg_status (*pmg_lrc_reduce_fn)(void *ctx, double *vals_dev, int count, void *stream);
pmg_status pmg_lrc_build_dev(pmg_lrc *out, int32_t k, int64_t ld, const double *B_lay_dev, const double *S_host, pmg_det_sweep_fn det, void *ctx, pmg_lrc_reduce_fn reduce, void *rctx);
pmg_status pmg_lrc_build(pmg_lrc *out, int32_t k, int64_t ld, int32_t n, const double *B_nat_host, const int64_t *pos, const double *S_host, pmg_det_sweep_fn det, void *ctx);
pmg_status pmg_lrc_rhs(pmg_lrc l, const double *b_lay, uint64_t seed, uint64_t counter, const double **beff, void *stream);
pmg_status pmg_lrc_rhs_done(pmg_lrc l, void *stream); /* after the sweep that used the vector pmg_lrc_rhs returned */
pmg_status pmg_lrc_post(pmg_lrc l, int dir, double *y_lay, void *stream);
pmg_status pmg_lrc_residual_sub(pmg_lrc l, const double *x_lay, double *r_lay, void *stream);
pmg_status pmg_lrc_residual_sub_restricted(pmg_lrc l_fine, pmg_lrc l_coarse, const double *x_fine_lay, double *b_coarse_lay, void *stream);
void       pmg_lrc_preset_eta(pmg_lrc l, uint64_t seed, uint64_t counter0, int n, const double *eta_dev, int64_t stride); /* noise terms drawn ahead */
uint64_t      pmg_lrc_noise_seed(uint64_t seed);
const double *pmg_lrc_sqrtS(pmg_lrc l); /* device, k entries */
int           pmg_lrc_rank(pmg_lrc l);
void       pmg_lrc_expect_residual(pmg_lrc l, int on); /* the sweeps that follow are in front of a residual of the same vector */
int        pmg_lrc_is_local(pmg_lrc l);
void       pmg_lrc_get_sizes(pmg_lrc l, int32_t *k, int64_t *ns, int *dense); /* ns = 0: none of B's support on this rank */
pmg_status pmg_lrc_get_compact(pmg_lrc l, int32_t *k, int64_t *ns, int64_t *rows_host, double *B_host, double *Bbf_host, double *Bbb_host);
pmg_lrc    pmg_grid_lrc(pmg_grid g); /* the grid operator's low-rank update, NULL if none (borrowed) */
void       pmg_lrc_destroy(pmg_lrc *l);
pmg_status pmg_mcsor_set_idiag_by_division(pmg_mcsor mc, int on); /* PCPARSOR's idiag = omega / d */
pmg_status pmg_mcsor_set_natural_order(pmg_mcsor mc, int on); /* no locality renumbering inside the colours (hierarchy levels) */
/* pmg_parsor.c: data-flow form of PCPARSOR's multi-rank sweep; the four arrays are malloc'ed, the caller frees them */
pmg_status pmg_parsor_build_dataflow(int32_t n, const int32_t *rowptr, const int32_t *colidx, const double *vals, int32_t nparts, const int32_t *row_starts, const int32_t *proccols_in, int32_t **e_rowptr, int32_t **e_colidx, double **e_vals, int32_t **e_colors, int32_t *nlevels_out, int32_t *proccols_out, int32_t *classes_out);
int        pmg_invert_small(int k, double *a_colmajor, double *inv); /* Gauss-Jordan, partial pivoting; a is overwritten; nonzero = singular */

pmg_status pmg_narrow_csr(int64_t nrows, int64_t ncols, const void *rowptr, const void *colidx, int idx_width, const int32_t **rp, const int32_t **ci, int32_t **rp_own, int32_t **ci_own);
/* trace ranges named like the reference's PetscLogEvents (src/parmgmc.c:118-127) */
#define PMG_EVENT_MULTICOL_SOR "MulticolSOR"
#define PMG_EVENT_VEC_SET_RANDOM_NORMAL "VecSetRandN"
void pmg_trace_begin(const char *name);
void pmg_trace_end(void);
/* device allocation helpers (zero-filled) */
pmg_status pmg_dev_alloc(void **p, size_t bytes);
pmg_status pmg_dev_upload(void **p, const void *host, size_t bytes);
void       pmg_dev_free(void *p);


/* pmg_rowblock.c */
void pmg_mcsor_adopt_arrays(pmg_mcsor mc, int32_t *rowptr, int32_t *colidx, double *vals);

#endif
