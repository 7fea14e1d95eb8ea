/* PC layer: the registration boundary of the reference restated without PETSc (C11).
 *
 * The reference plugs its samplers into PETSc with PCRegister(name, ctor) (src/parmgmc.c:44-54); a constructor
 * fills pc->ops->{setup, apply, applyrichardson, reset, destroy, setfromoptions, view} and composes the sample
 * callback setter (src/pc_sorgibbs.c:306-324, src/pc_mcgibbs.c:305-327).  PETSc is not available here, so this
 * file provides the same shapes on raw device arrays: a type registry, an ops table with the same seven entries,
 * an options database with the reference's option names, PCSetSampleCallback, the PCSHELL route
 * (examples/ex3.c:59-67) and the only KSP the samplers need, KSPRICHARDSON's fast path that hands ALL iterations
 * to applyrichardson (examples/ex1.c:20,123-129).  INTEGRATION.md shows the real PETSc adapter.
 */
#include "pmg_internal.h"
#include <math.h>

/* ---------------------------------------------------------------------------------------------------- */
/* options database (PetscOptionsSetValue / PetscOptionsGet*)                                           */
/* ---------------------------------------------------------------------------------------------------- */
#define PMG_MAX_OPTS 128
static struct {
  char name[96], value[64];
} pmg_opts[PMG_MAX_OPTS];
static int pmg_nopts = 0;

pmg_status pmg_options_set_value(const char *name, const char *value)
{
  PMG_CHECK(name && name[0] == '-', PMG_ERR_ARG_WRONG, "option names start with '-'");
  for (int i = 0; i < pmg_nopts; ++i)
    if (!strcmp(pmg_opts[i].name, name)) {
      snprintf(pmg_opts[i].value, sizeof pmg_opts[i].value, "%s", value ? value : "");
      return PMG_SUCCESS;
    }
  PMG_CHECK(pmg_nopts < PMG_MAX_OPTS, PMG_ERR_MEM, "options database full");
  snprintf(pmg_opts[pmg_nopts].name, sizeof pmg_opts[0].name, "%s", name);
  snprintf(pmg_opts[pmg_nopts].value, sizeof pmg_opts[0].value, "%s", value ? value : "");
  ++pmg_nopts;
  return PMG_SUCCESS;
}

pmg_status pmg_options_clear(void)
{
  pmg_nopts = 0;
  return PMG_SUCCESS;
}

static const char *opt_find(const char *prefix, const char *name)
{
  char full[320];
  snprintf(full, sizeof full, "-%s%s", prefix ? prefix : "", name + 1);
  for (int i = 0; i < pmg_nopts; ++i)
    if (!strcmp(pmg_opts[i].name, full)) return pmg_opts[i].value;
  return NULL;
}
static int opt_bool(const char *prefix, const char *name)
{
  const char *v = opt_find(prefix, name);
  return v && (v[0] == 0 || !strcmp(v, "1") || !strcmp(v, "true") || !strcmp(v, "yes"));
}
static int opt_real(const char *prefix, const char *name, double *out)
{
  const char *v = opt_find(prefix, name);
  if (!v) return 0;
  *out = strtod(v, NULL);
  return 1;
}
static int opt_int(const char *prefix, const char *name, int *out)
{
  const char *v = opt_find(prefix, name);
  if (!v) return 0;
  *out = atoi(v);
  return 1;
}

/* ---------------------------------------------------------------------------------------------------- */
/* operators (the Mat + DM a PC is handed through PCSetOperators / PCSetDM)                             */
/* ---------------------------------------------------------------------------------------------------- */
struct pmg_mat_s {
  int            kind; /* 0 = MATSEQAIJ given as borrowed host CSR, 1 = DMDA operator of src/problems.c:14-75 */
  int32_t        n;
  const int32_t *rowptr, *colidx;
  const double  *vals;
  int32_t        nx, ny, nz;
  double         kappa;
  /* MATLRC: this + B S B^T (borrowed host arrays: B n x k column-major, S k entries) */
  int32_t        lrc_k;
  const double  *lrc_B, *lrc_S;
};

pmg_status pmg_mat_create_csr(int32_t n, const int32_t *rowptr, const int32_t *colidx, const double *vals, pmg_mat *out)
{
  PMG_CHECK(out, PMG_ERR_ARG_NULL, "null output handle");
  PMG_CHECK(n >= 0 && rowptr, PMG_ERR_ARG_WRONG, "bad CSR");
  pmg_mat m = (pmg_mat)calloc(1, sizeof *m);
  PMG_CHECK(m, PMG_ERR_MEM, "out of host memory");
  m->kind   = 0;
  m->n      = n;
  m->rowptr = rowptr;
  m->colidx = colidx;
  m->vals   = vals;
  *out      = m;
  return PMG_SUCCESS;
}

pmg_status pmg_mat_create_dmda(int32_t nx, int32_t ny, int32_t nz, double kappa, pmg_mat *out)
{
  PMG_CHECK(out, PMG_ERR_ARG_NULL, "null output handle");
  PMG_CHECK(nx >= 2 && ny >= 1 && nz >= 1, PMG_ERR_ARG_OUTOFRANGE, "grid %d x %d x %d", nx, ny, nz);
  pmg_mat m = (pmg_mat)calloc(1, sizeof *m);
  PMG_CHECK(m, PMG_ERR_MEM, "out of host memory");
  m->kind  = 1;
  m->nx    = nx;
  m->ny    = ny;
  m->nz    = nz;
  m->n     = nx * ny * nz;
  m->kappa = kappa;
  *out     = m;
  return PMG_SUCCESS;
}

/* MatCreateLRC(A, B, S, NULL, &Alrc) as examples/ex4.c and src/obs.c use it: A + B diag(S) B^T.  The result refers
   to the same base operator (its arrays stay borrowed) and to B / S as borrowed host arrays. */
pmg_status pmg_mat_create_lrc(pmg_mat A, int32_t k, const double *B_host, const double *S_host, pmg_mat *out)
{
  PMG_CHECK(A && out, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(!A->lrc_k, PMG_ERR_SUP, "nested low-rank updates");
  PMG_CHECK(k >= 1 && k <= 64, PMG_ERR_ARG_OUTOFRANGE, "rank k = %d (1..64 supported)", k);
  PMG_CHECK(B_host && S_host, PMG_ERR_ARG_NULL, "null low-rank factor");
  pmg_mat m = (pmg_mat)malloc(sizeof *m);
  PMG_CHECK(m, PMG_ERR_MEM, "out of host memory");
  *m        = *A;
  m->lrc_k  = k;
  m->lrc_B  = B_host;
  m->lrc_S  = S_host;
  *out      = m;
  return PMG_SUCCESS;
}

pmg_status pmg_mat_get_size(pmg_mat m, int32_t *n)
{
  PMG_CHECK(m && n, PMG_ERR_ARG_NULL, "null argument");
  *n = m->n;
  return PMG_SUCCESS;
}

pmg_status pmg_mat_destroy(pmg_mat *m)
{
  if (m && *m) {
    free(*m);
    *m = NULL;
  }
  return PMG_SUCCESS;
}

/* ---------------------------------------------------------------------------------------------------- */
/* PC object                                                                                            */
/* ---------------------------------------------------------------------------------------------------- */
typedef struct {
  pmg_status (*setup)(pmg_pc);
  pmg_status (*apply)(pmg_pc, const double *, double *, void *);
  pmg_status (*applyrichardson)(pmg_pc, const double *, double *, int32_t, int, int32_t *, void *);
  pmg_status (*reset)(pmg_pc);
  pmg_status (*destroy)(pmg_pc);
  pmg_status (*setfromoptions)(pmg_pc);
  pmg_status (*view)(pmg_pc, char *, size_t);
} pmg_pc_ops;

struct pmg_pc_s {
  char       type[32], prefix[128];
  pmg_pc_ops ops;
  void      *data;
  pmg_mat    pmat; /* borrowed, like pc->pmat (src/pc_sorgibbs.c:35-38) */
  int        setupcalled;
  uint64_t   stream_id, counter; /* noise stream of this PC, samples drawn so far */
  /* sample callback (src/pc_sorgibbs.c:280-293) */
  pmg_sample_callback scb;
  void               *cbctx;
  int (*del_scb)(void *);
};

#define PMG_MAX_TYPES 16
static struct {
  char name[32];
  pmg_status (*ctor)(pmg_pc);
} pmg_types[PMG_MAX_TYPES];
static int      pmg_ntypes     = 0;
static uint64_t pmg_seed       = 0xCAFE; /* the seed of the global PetscRandom (src/parmgmc.c:56-68; examples/ex6.c:131) */
static uint64_t pmg_next_stream = 0;

pmg_status pmg_pc_register(const char *name, pmg_status (*ctor)(pmg_pc))
{
  PMG_CHECK(name && ctor, PMG_ERR_ARG_NULL, "null argument");
  for (int i = 0; i < pmg_ntypes; ++i)
    if (!strcmp(pmg_types[i].name, name)) {
      pmg_types[i].ctor = ctor; /* PCRegister replaces an existing entry */
      return PMG_SUCCESS;
    }
  PMG_CHECK(pmg_ntypes < PMG_MAX_TYPES, PMG_ERR_MEM, "type registry full");
  snprintf(pmg_types[pmg_ntypes].name, sizeof pmg_types[0].name, "%s", name);
  pmg_types[pmg_ntypes++].ctor = ctor;
  return PMG_SUCCESS;
}

pmg_status pmg_set_seed(uint64_t seed)
{
  pmg_seed = seed;
  return PMG_SUCCESS;
}

static uint64_t pc_seed(pmg_pc pc) { return pmg_seed + 0xD1B54A32D192ED03ull * (pc->stream_id + 1); }

pmg_status pmg_pc_create(pmg_pc *out)
{
  PMG_CHECK(out, PMG_ERR_ARG_NULL, "null output handle");
  pmg_pc pc = (pmg_pc)calloc(1, sizeof *pc);
  PMG_CHECK(pc, PMG_ERR_MEM, "out of host memory");
  pc->stream_id = pmg_next_stream++;
  *out          = pc;
  return PMG_SUCCESS;
}

static pmg_status pc_drop_callback(pmg_pc pc)
{
  if (pc->del_scb) {
    pc->del_scb(pc->cbctx);
    pc->del_scb = NULL;
  }
  return PMG_SUCCESS;
}

pmg_status pmg_pc_reset(pmg_pc pc)
{
  PMG_CHECK(pc, PMG_ERR_ARG_NULL, "null PC");
  if (pc->ops.reset) PMG_CALL(pc->ops.reset(pc));
  pc_drop_callback(pc); /* PCReset_* calls the deleter (src/pc_sorgibbs.c:153-156) */
  pc->setupcalled = 0;
  return PMG_SUCCESS;
}

pmg_status pmg_pc_set_type(pmg_pc pc, const char *type)
{
  PMG_CHECK(pc && type, PMG_ERR_ARG_NULL, "null argument");
  if (pc->ops.destroy) {
    PMG_CALL(pc->ops.destroy(pc));
    pc->data = NULL;
  }
  memset(&pc->ops, 0, sizeof pc->ops);
  for (int i = 0; i < pmg_ntypes; ++i)
    if (!strcmp(pmg_types[i].name, type)) {
      snprintf(pc->type, sizeof pc->type, "%s", type);
      pc->setupcalled = 0;
      return pmg_types[i].ctor(pc);
    }
  PMG_FAIL(PMG_ERR_ARG_UNKNOWN_TYPE, "Unable to find requested PC type %s (call pmg_initialize first)", type);
}

pmg_status pmg_pc_get_type(pmg_pc pc, char *buf, int32_t len)
{
  PMG_CHECK(pc && buf && len > 0, PMG_ERR_ARG_NULL, "null argument");
  snprintf(buf, (size_t)len, "%s", pc->type);
  return PMG_SUCCESS;
}

pmg_status pmg_pc_set_options_prefix(pmg_pc pc, const char *prefix)
{
  PMG_CHECK(pc, PMG_ERR_ARG_NULL, "null PC");
  snprintf(pc->prefix, sizeof pc->prefix, "%s", prefix ? prefix : "");
  return PMG_SUCCESS;
}

pmg_status pmg_pc_set_operators(pmg_pc pc, pmg_mat mat)
{
  PMG_CHECK(pc && mat, PMG_ERR_ARG_NULL, "null argument");
  pc->pmat        = mat;
  pc->setupcalled = 0;
  return PMG_SUCCESS;
}

pmg_status pmg_pc_set_from_options(pmg_pc pc)
{
  PMG_CHECK(pc, PMG_ERR_ARG_NULL, "null PC");
  const char *t = opt_find(pc->prefix, "-pc_type");
  if (t && strcmp(t, pc->type)) PMG_CALL(pmg_pc_set_type(pc, t));
  if (pc->ops.setfromoptions) PMG_CALL(pc->ops.setfromoptions(pc));
  return PMG_SUCCESS;
}

pmg_status pmg_pc_setup(pmg_pc pc)
{
  PMG_CHECK(pc, PMG_ERR_ARG_NULL, "null PC");
  PMG_CHECK(pc->type[0], PMG_ERR_ARG_WRONGSTATE, "PC type not set");
  if (pc->setupcalled) return PMG_SUCCESS;
  PMG_CHECK(pc->pmat || !strcmp(pc->type, "shell"), PMG_ERR_ARG_WRONGSTATE, "Matrix must be set first");
  if (pc->ops.setup) PMG_CALL(pc->ops.setup(pc));
  pc->setupcalled = 1;
  return PMG_SUCCESS;
}

pmg_status pmg_pc_apply(pmg_pc pc, const double *b, double *y, void *stream)
{
  PMG_CHECK(pc && b && y, PMG_ERR_ARG_NULL, "null argument");
  PMG_CALL(pmg_pc_setup(pc));
  PMG_CHECK(pc->ops.apply, PMG_ERR_SUP, "PC type %s does not have apply", pc->type); /* PETSc's PCApply check; mcgibbs and gamgmc set only applyrichardson (src/pc_mcgibbs.c:318-325) */
  return pc->ops.apply(pc, b, y, stream);
}

/* PCApplyRichardson: tolerances and the work vector of the PETSc signature are ignored by every sampler
   (src/pc_sorgibbs.c:117-120), so they are not part of this entry point; *reason is always
   PCRICHARDSON_CONVERGED_ITS (= 4). */
pmg_status pmg_pc_apply_richardson(pmg_pc pc, const double *b, double *y, int32_t its, int guesszero, int32_t *outits, int32_t *reason, void *stream)
{
  PMG_CHECK(pc && b && y, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(its >= 0, PMG_ERR_ARG_OUTOFRANGE, "its = %d", its);
  PMG_CALL(pmg_pc_setup(pc));
  PMG_CHECK(pc->ops.applyrichardson, PMG_ERR_SUP, "PC type %s does not have applyrichardson", pc->type);
  int32_t done = 0;
  PMG_CALL(pc->ops.applyrichardson(pc, b, y, its, guesszero, &done, stream));
  if (outits) *outits = done;
  if (reason) *reason = 4;
  return PMG_SUCCESS;
}

/* KSPSolve with -ksp_type richardson, KSP_NORM_NONE: one applyrichardson call for all max_it iterations
   (examples/ex1.c:97-129); guess_nonzero = KSPSetInitialGuessNonzero. */
pmg_status pmg_ksp_richardson_solve(pmg_pc pc, const double *b, double *y, int32_t max_it, int guess_nonzero, void *stream)
{
  PMG_CHECK(pc && b && y, PMG_ERR_ARG_NULL, "null argument");
  PMG_CALL(pmg_pc_setup(pc));
  if (!guess_nonzero) PMG_HIP(hipMemsetAsync(y, 0, sizeof(double) * (size_t)pc->pmat->n, (hipStream_t)stream));
  return pmg_pc_apply_richardson(pc, b, y, max_it, !guess_nonzero, NULL, NULL, stream);
}

pmg_status pmg_pc_set_sample_callback(pmg_pc pc, pmg_sample_callback cb, void *ctx, int (*deleter)(void *))
{
  PMG_CHECK(pc, PMG_ERR_ARG_NULL, "null PC");
  PMG_CHECK(pc->type[0], PMG_ERR_ARG_WRONGSTATE, "PC type not set"); /* the setter is composed by the constructor (src/parmgmc.c:139-151) */
  PMG_CHECK(strcmp(pc->type, "shell") && strcmp(pc->type, "parsor"), PMG_ERR_SUP, "PC type %s has no sample callback", pc->type);
  pc_drop_callback(pc); /* a previous context is deleted first (src/pc_sorgibbs.c:284-287) */
  pc->scb     = cb;
  pc->cbctx   = ctx;
  pc->del_scb = deleter;
  return PMG_SUCCESS;
}

/* checkpoint / resume of a chain (the reference has none: chain state = the caller's Vec + the RNG state;
   with a counter-based source the RNG state is two integers) */
pmg_status pmg_pc_get_noise_state(pmg_pc pc, uint64_t *seed, uint64_t *counter)
{
  PMG_CHECK(pc && seed && counter, PMG_ERR_ARG_NULL, "null argument");
  *seed    = pc_seed(pc);
  *counter = pc->counter;
  return PMG_SUCCESS;
}

pmg_status pmg_pc_set_noise_counter(pmg_pc pc, uint64_t counter)
{
  PMG_CHECK(pc, PMG_ERR_ARG_NULL, "null PC");
  pc->counter = counter;
  return PMG_SUCCESS;
}

pmg_status pmg_pc_view(pmg_pc pc, char *buf, int32_t len)
{
  PMG_CHECK(pc && buf && len > 0, PMG_ERR_ARG_NULL, "null argument");
  int n = snprintf(buf, (size_t)len, "PC Object: type: %s\n", pc->type);
  if (pc->ops.view && n < len) PMG_CALL(pc->ops.view(pc, buf + n, (size_t)(len - n)));
  return PMG_SUCCESS;
}

pmg_status pmg_pc_destroy(pmg_pc *pcp)
{
  if (!pcp || !*pcp) return PMG_SUCCESS;
  pmg_pc pc = *pcp;
  if (pc->ops.destroy) pc->ops.destroy(pc);
  pc_drop_callback(pc);
  free(pc);
  *pcp = NULL;
  return PMG_SUCCESS;
}

static pmg_status pc_notify(pmg_pc pc, int32_t it, const double *y, void *stream)
{
  (void)stream;
  if (pc->scb) {
    const int rc = pc->scb(it, y, pc->pmat->n, pc->cbctx);
    PMG_CHECK(rc == 0, rc, "sample callback returned %d", rc);
  }
  return PMG_SUCCESS;
}

/* ---------------------------------------------------------------------------------------------------- */
/* "sorgibbs" (src/pc_sorgibbs.c) and "mcgibbs" (src/pc_mcgibbs.c): one implementation, two option sets  */
/* ---------------------------------------------------------------------------------------------------- */
typedef struct {
  int       is_mc;      /* mcgibbs: scaled noise, omega, all sweep types; sorgibbs: omega = 1, forward */
  double    omega;
  int       type;
  int       coloring;
  pmg_grid  g;
  pmg_mcsor mc;
} pc_gibbs;

static pmg_status gibbs_reset(pmg_pc pc)
{
  pc_gibbs *d = (pc_gibbs *)pc->data;
  pmg_grid_destroy(&d->g);
  pmg_mcsor_destroy(&d->mc);
  return PMG_SUCCESS;
}

static pmg_status gibbs_destroy(pmg_pc pc)
{
  if (pc->data) {
    gibbs_reset(pc);
    free(pc->data);
    pc->data = NULL;
  }
  return PMG_SUCCESS;
}

static pmg_status gibbs_setup(pmg_pc pc) /* PCSetUp_SORGibbs :181-262 / PCSetUp_MulticolorGibbs :213-255 */
{
  pc_gibbs *d = (pc_gibbs *)pc->data;
  gibbs_reset(pc);
  if (pc->pmat->kind == 1) {
    PMG_CALL(pmg_grid_create(pc->pmat->nx, pc->pmat->ny, pc->pmat->nz, 0, pc->pmat->nz, pc->pmat->kappa, &d->g));
    PMG_CALL(pmg_grid_set_omega(d->g, d->omega));
    PMG_CALL(pmg_grid_set_sweep_type(d->g, d->type));
    if (pc->pmat->lrc_k) PMG_CALL(pmg_grid_set_lowrank(d->g, pc->pmat->lrc_k, pc->pmat->lrc_B, pc->pmat->lrc_S)); /* MATLRC, src/pc_mcgibbs.c:226-243 */
  } else {
    PMG_CALL(pmg_mcsor_create_csr(pc->pmat->n, pc->pmat->rowptr, pc->pmat->colidx, pc->pmat->vals, &d->mc));
    PMG_CALL(pmg_mcsor_set_coloring(d->mc, d->coloring, NULL));
    PMG_CALL(pmg_mcsor_set_omega(d->mc, d->omega));
    PMG_CALL(pmg_mcsor_set_sweep_type(d->mc, d->type));
    PMG_CALL(pmg_mcsor_setup(d->mc));
    if (pc->pmat->lrc_k) PMG_CALL(pmg_mcsor_set_lowrank(d->mc, pc->pmat->lrc_k, pc->pmat->lrc_B, pc->pmat->lrc_S));
  }
  return PMG_SUCCESS;
}

static pmg_status gibbs_draw(pmg_pc pc, const double *b, double *y, int32_t its, void *stream)
{
  pc_gibbs *d = (pc_gibbs *)pc->data;
  if (d->g) return pmg_grid_sample(d->g, b, y, its, d->is_mc, pc_seed(pc), pc->counter, &pc->counter, stream);
  return pmg_mcsor_sample(d->mc, b, y, its, d->is_mc, pc_seed(pc), pc->counter, &pc->counter, stream);
}

static pmg_status gibbs_applyrichardson(pmg_pc pc, const double *b, double *y, int32_t its, int guesszero, int32_t *outits, void *stream)
{
  (void)guesszero; /* ignored like the tolerances (src/pc_sorgibbs.c:117-120) */
  if (!pc->scb) {
    PMG_CALL(gibbs_draw(pc, b, y, its, stream));
  } else {
    for (int32_t it = 0; it < its; ++it) { /* sample, then the callback (src/pc_mcgibbs.c:183, src/pc_sorgibbs.c:126-129) */
      PMG_CALL(gibbs_draw(pc, b, y, 1, stream));
      PMG_CALL(pc_notify(pc, it, y, stream));
    }
  }
  *outits = its;
  return PMG_SUCCESS;
}

static pmg_status sorgibbs_apply(pmg_pc pc, const double *b, double *y, void *stream) /* PCApply_SORGibbs :105-113 */
{
  PMG_HIP(hipMemsetAsync(y, 0, sizeof(double) * (size_t)pc->pmat->n, (hipStream_t)stream));
  return gibbs_draw(pc, b, y, 1, stream);
}

static pmg_status gibbs_setfromoptions(pmg_pc pc)
{
  pc_gibbs *d = (pc_gibbs *)pc->data;
  double    om;
  if (d->is_mc) { /* src/pc_mcgibbs.c:190-211 */
    if (opt_real(pc->prefix, "-pc_mcgibbs_omega", &om)) {
      PMG_CHECK(om > 0.0 && om < 2.0, PMG_ERR_ARG_OUTOFRANGE, "-pc_mcgibbs_omega %g outside (0,2)", om);
      d->omega = om;
    }
    if (opt_bool(pc->prefix, "-pc_mcgibbs_forward")) d->type = PMG_SOR_FORWARD_SWEEP;
    if (opt_bool(pc->prefix, "-pc_mcgibbs_backward")) d->type = PMG_SOR_BACKWARD_SWEEP;
    if (opt_bool(pc->prefix, "-pc_mcgibbs_symmetric")) d->type = PMG_SOR_SYMMETRIC_SWEEP;
  } else { /* src/pc_sorgibbs.c:264-278: forward, or local forward (= forward on one device) */
    if (opt_bool(pc->prefix, "-pc_sorgibbs_forward") || opt_bool(pc->prefix, "-pc_sorgibbs_local_forward")) d->type = PMG_SOR_FORWARD_SWEEP;
  }
  const char *c = opt_find(pc->prefix, d->is_mc ? "-pc_mcgibbs_coloring" : "-pc_sorgibbs_coloring");
  if (c) {
    if (!strcmp(c, "greedy")) d->coloring = PMG_COLORING_GREEDY;
    else if (!strcmp(c, "lexlevels")) d->coloring = PMG_COLORING_LEXLEVELS;
    else if (!strcmp(c, "iterated")) d->coloring = PMG_COLORING_ITERATED;
    else PMG_FAIL(PMG_ERR_ARG_WRONG, "unknown colouring %s", c);
  }
  pc->setupcalled = 0;
  return PMG_SUCCESS;
}

static pmg_status gibbs_view(pmg_pc pc, char *buf, size_t len)
{
  pc_gibbs *d  = (pc_gibbs *)pc->data;
  int32_t   nc = 0;
  if (d->g) pmg_grid_get_num_colors(d->g, &nc);
  else if (d->mc) pmg_mcsor_get_num_colors(d->mc, &nc);
  if (d->is_mc) snprintf(buf, len, "Number of colours: %d\n", nc); /* src/pc_mcgibbs.c:257-266 */
  else snprintf(buf, len, "Sweep type: Forward\n");                /* src/pc_sorgibbs.c:295-304 */
  return PMG_SUCCESS;
}

static pmg_status gibbs_ctor(pmg_pc pc, int is_mc)
{
  pc_gibbs *d = (pc_gibbs *)calloc(1, sizeof *d);
  PMG_CHECK(d, PMG_ERR_MEM, "out of host memory");
  d->is_mc                = is_mc;
  d->omega                = 1.0;
  d->type                 = PMG_SOR_FORWARD_SWEEP;
  d->coloring             = PMG_COLORING_GREEDY;
  pc->data                = d;
  pc->ops.setup           = gibbs_setup;
  pc->ops.applyrichardson = gibbs_applyrichardson;
  pc->ops.apply           = is_mc ? NULL : sorgibbs_apply; /* mcgibbs sets no apply (src/pc_mcgibbs.c:318-325) */
  pc->ops.reset           = gibbs_reset;
  pc->ops.destroy         = gibbs_destroy;
  pc->ops.setfromoptions  = gibbs_setfromoptions;
  pc->ops.view            = gibbs_view;
  return PMG_SUCCESS;
}
static pmg_status PCCreate_SORGibbs(pmg_pc pc) { return gibbs_ctor(pc, 0); }
static pmg_status PCCreate_MulticolorGibbs(pmg_pc pc) { return gibbs_ctor(pc, 1); }

/* PCMulticolorGibbsSetOmega / SetSweepType (include/parmgmc/pc/pc_mcgibbs.h:17-18) */
pmg_status pmg_pc_mcgibbs_set_omega(pmg_pc pc, double omega)
{
  PMG_CHECK(pc && !strcmp(pc->type, "mcgibbs"), PMG_ERR_ARG_WRONG, "not a mcgibbs PC");
  ((pc_gibbs *)pc->data)->omega = omega;
  pc->setupcalled               = 0;
  return PMG_SUCCESS;
}
pmg_status pmg_pc_mcgibbs_set_sweep_type(pmg_pc pc, int type)
{
  PMG_CHECK(pc && !strcmp(pc->type, "mcgibbs"), PMG_ERR_ARG_WRONG, "not a mcgibbs PC");
  PMG_CHECK(pmg_sweep_type_ok(type), PMG_ERR_SUP, "Only forward, backward and symmetric sweep supported");
  ((pc_gibbs *)pc->data)->type = type;
  pc->setupcalled              = 0;
  return PMG_SUCCESS;
}

/* ---------------------------------------------------------------------------------------------------- */
/* "parsor" (src/pc_parsor.c): deterministic forward SOR in the reference's lexicographic order          */
/* ---------------------------------------------------------------------------------------------------- */
typedef struct {
  double    omega;
  int       its;
  pmg_mcsor mc;
  pmg_grid  g;
  /* the sweep of `nparts` MPI ranks owning contiguous row blocks, reproduced on this device (pmg_parsor.c) */
  int32_t  nparts, nlevels, *row_starts, *proccols_user, *proccols, *classes;
  int32_t  ld;                     /* layout length of the 2n-row operator */
  double  *xe, *be, *lay_b, *lay_y; /* [x; snapshot], [b; 0] in natural order and in the sweep layout */
} pc_parsor;

static pmg_status parsor_reset(pmg_pc pc)
{
  pc_parsor *d = (pc_parsor *)pc->data;
  pmg_mcsor_destroy(&d->mc);
  pmg_grid_destroy(&d->g);
  pmg_dev_free(d->xe), pmg_dev_free(d->be), pmg_dev_free(d->lay_b), pmg_dev_free(d->lay_y);
  d->xe = d->be = d->lay_b = d->lay_y = NULL;
  free(d->proccols), free(d->classes);
  d->proccols = d->classes = NULL;
  return PMG_SUCCESS;
}
static pmg_status parsor_destroy(pmg_pc pc)
{
  if (pc->data) {
    pc_parsor *d = (pc_parsor *)pc->data;
    parsor_reset(pc);
    free(d->row_starts), free(d->proccols_user);
    free(pc->data);
    pc->data = NULL;
  }
  return PMG_SUCCESS;
}
static pmg_status parsor_setup(pmg_pc pc)
{
  pc_parsor *d = (pc_parsor *)pc->data;
  parsor_reset(pc);
  PMG_CHECK(!pc->pmat->lrc_k, PMG_ERR_SUP, "parsor works on assembled matrices (MATAIJ), not MATLRC");
  if (d->nparts > 0) { /* the order nparts ranks would sweep in (src/pc_parsor.c:703-878) */
    const int32_t n = pc->pmat->n;
    PMG_CHECK(pc->pmat->kind == 0, PMG_ERR_SUP, "a rank partition needs an assembled matrix (MATAIJ)");
    PMG_CHECK(d->row_starts[d->nparts] == n, PMG_ERR_ARG_WRONG, "the partition covers %d rows, the matrix has %d", d->row_starts[d->nparts], n);
    int32_t *rp = NULL, *ci = NULL, *cols = NULL;
    double  *va = NULL;
    d->proccols = (int32_t *)malloc(sizeof(int32_t) * (size_t)d->nparts);
    d->classes  = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    PMG_CHECK(d->proccols && d->classes, PMG_ERR_MEM, "out of host memory");
    PMG_CALL(pmg_parsor_build_dataflow(n, pc->pmat->rowptr, pc->pmat->colidx, pc->pmat->vals, d->nparts, d->row_starts, d->proccols_user, &rp, &ci, &va, &cols, &d->nlevels, d->proccols, d->classes));
    pmg_status st = pmg_mcsor_create_csr(2 * n, rp, ci, va, &d->mc);
    if (!st) st = pmg_mcsor_set_coloring(d->mc, PMG_COLORING_USER, cols);
    if (!st) st = pmg_mcsor_set_omega(d->mc, d->omega);
    if (!st) st = pmg_mcsor_set_idiag_by_division(d->mc, 1);
    if (!st) st = pmg_mcsor_setup(d->mc); /* copies what it needs */
    free(rp), free(ci), free(va), free(cols);
    PMG_CALL(st);
    PMG_CALL(pmg_mcsor_layout_len(d->mc, &d->ld));
    PMG_CALL(pmg_dev_alloc((void **)&d->xe, sizeof(double) * 2 * (size_t)n));
    PMG_CALL(pmg_dev_alloc((void **)&d->be, sizeof(double) * 2 * (size_t)n));
    PMG_CALL(pmg_dev_alloc((void **)&d->lay_b, sizeof(double) * (size_t)d->ld));
    PMG_CALL(pmg_dev_alloc((void **)&d->lay_y, sizeof(double) * (size_t)d->ld));
  } else if (pc->pmat->kind == 1) { /* red-black order on the grid: a different but equally valid sweep order */
    PMG_CALL(pmg_grid_create(pc->pmat->nx, pc->pmat->ny, pc->pmat->nz, 0, pc->pmat->nz, pc->pmat->kappa, &d->g));
    PMG_CALL(pmg_grid_set_omega(d->g, d->omega));
  } else { /* dependency levels == the lexicographic order PCPARSOR preserves (src/pc_parsor.c:703-878) */
    PMG_CALL(pmg_mcsor_create_csr(pc->pmat->n, pc->pmat->rowptr, pc->pmat->colidx, pc->pmat->vals, &d->mc));
    PMG_CALL(pmg_mcsor_set_coloring(d->mc, PMG_COLORING_LEXLEVELS, NULL));
    PMG_CALL(pmg_mcsor_set_omega(d->mc, d->omega));
    PMG_CALL(pmg_mcsor_set_idiag_by_division(d->mc, 1));
    PMG_CALL(pmg_mcsor_setup(d->mc));
  }
  return PMG_SUCCESS;
}
/* PCPARSORApplySOR(pc, b, its, zero_initial_guess, x) (src/pc_parsor.c:892-904) */
pmg_status pmg_pc_parsor_apply_sor(pmg_pc pc, const double *b, int32_t its, int zero_initial_guess, double *x, void *stream)
{
  PMG_CHECK(pc && b && x && !strcmp(pc->type, "parsor"), PMG_ERR_ARG_WRONG, "not a parsor PC");
  PMG_CALL(pmg_pc_setup(pc));
  pc_parsor *d = (pc_parsor *)pc->data;
  if (zero_initial_guess) PMG_HIP(hipMemsetAsync(x, 0, sizeof(double) * (size_t)pc->pmat->n, (hipStream_t)stream));
  if (d->nparts > 0) {
    const size_t   nb = sizeof(double) * (size_t)pc->pmat->n;
    hipStream_t    s  = (hipStream_t)stream;
    PMG_HIP(hipMemcpyAsync(d->be, b, nb, hipMemcpyDeviceToDevice, s));
    PMG_HIP(hipMemsetAsync(d->be + pc->pmat->n, 0, nb, s));
    PMG_CALL(pmg_mcsor_to_layout(d->mc, d->be, d->lay_b, stream));
    for (int32_t it = 0; it < its; ++it) {
      PMG_HIP(hipMemcpyAsync(d->xe, x, nb, hipMemcpyDeviceToDevice, s));
      PMG_HIP(hipMemcpyAsync(d->xe + pc->pmat->n, x, nb, hipMemcpyDeviceToDevice, s)); /* values from before the iteration */
      PMG_CALL(pmg_mcsor_to_layout(d->mc, d->xe, d->lay_y, stream));
      for (int32_t c = 0; c < d->nlevels; ++c) PMG_CALL(pmg_mcsor_sweep_color_layout(d->mc, c, 0, 0, 0, 0, d->lay_b, d->lay_y, stream));
      PMG_CALL(pmg_mcsor_from_layout(d->mc, d->lay_y, d->xe, stream));
      PMG_HIP(hipMemcpyAsync(x, d->xe, nb, hipMemcpyDeviceToDevice, s));
    }
    return PMG_SUCCESS;
  }
  for (int32_t it = 0; it < its; ++it) {
    if (d->g) PMG_CALL(pmg_grid_apply(d->g, b, x, stream));
    else PMG_CALL(pmg_mcsor_apply(d->mc, b, x, stream));
  }
  return PMG_SUCCESS;
}
static pmg_status parsor_apply(pmg_pc pc, const double *b, double *y, void *stream) { return pmg_pc_parsor_apply_sor(pc, b, ((pc_parsor *)pc->data)->its, 1, y, stream); }
static pmg_status parsor_setfromoptions(pmg_pc pc) /* src/pc_parsor.c:970-980 */
{
  pc_parsor *d = (pc_parsor *)pc->data;
  opt_real(pc->prefix, "-pc_parsor_omega", &d->omega);
  opt_int(pc->prefix, "-pc_parsor_its", &d->its);
  pc->setupcalled = 0;
  return PMG_SUCCESS;
}
static pmg_status PCCreate_PARSOR(pmg_pc pc)
{
  pc_parsor *d = (pc_parsor *)calloc(1, sizeof *d);
  PMG_CHECK(d, PMG_ERR_MEM, "out of host memory");
  d->omega               = 1.0;
  d->its                 = 1;
  pc->data               = d;
  pc->ops.setup          = parsor_setup;
  pc->ops.apply          = parsor_apply;
  pc->ops.reset          = parsor_reset;
  pc->ops.destroy        = parsor_destroy;
  pc->ops.setfromoptions = parsor_setfromoptions;
  return PMG_SUCCESS;
}
pmg_status pmg_pc_parsor_set_omega(pmg_pc pc, double omega)
{
  PMG_CHECK(pc && !strcmp(pc->type, "parsor"), PMG_ERR_ARG_WRONG, "not a parsor PC");
  ((pc_parsor *)pc->data)->omega = omega;
  pc->setupcalled                = 0;
  return PMG_SUCCESS;
}
/* Reproduce the sweep of `nparts` MPI ranks owning the contiguous row blocks [row_starts[p], row_starts[p+1])
   (ParallelSORApply, src/pc_parsor.c:703-878; see pmg_parsor.c).  proc_colors: the colouring of the ranks
   (ColorProcessors, :187-270), NULL = first fit in rank order.  nparts = 0 returns to the single-rank order. */
pmg_status pmg_pc_parsor_set_partition(pmg_pc pc, int32_t nparts, const int32_t *row_starts, const int32_t *proc_colors)
{
  PMG_CHECK(pc && !strcmp(pc->type, "parsor"), PMG_ERR_ARG_WRONG, "not a parsor PC");
  PMG_CHECK(nparts >= 0 && (nparts == 0 || row_starts), PMG_ERR_ARG_WRONG, "bad partition");
  pc_parsor *d = (pc_parsor *)pc->data;
  free(d->row_starts), free(d->proccols_user);
  d->row_starts = d->proccols_user = NULL;
  d->nparts                        = nparts;
  if (nparts > 0) {
    d->row_starts = (int32_t *)malloc(sizeof(int32_t) * ((size_t)nparts + 1));
    PMG_CHECK(d->row_starts, PMG_ERR_MEM, "out of host memory");
    memcpy(d->row_starts, row_starts, sizeof(int32_t) * ((size_t)nparts + 1));
    if (proc_colors) {
      d->proccols_user = (int32_t *)malloc(sizeof(int32_t) * (size_t)nparts);
      PMG_CHECK(d->proccols_user, PMG_ERR_MEM, "out of host memory");
      memcpy(d->proccols_user, proc_colors, sizeof(int32_t) * (size_t)nparts);
    }
  }
  pc->setupcalled = 0;
  return PMG_SUCCESS;
}
/* after set-up: number of dependency levels (= launches per iteration), the rank colours used and the class of every
   row (0 INT, 1 TOP, 2 MID, 3 BOT; src/pc_parsor.c:283-288); any output may be NULL */
pmg_status pmg_pc_parsor_get_partition_info(pmg_pc pc, int32_t *nlevels, int32_t *proc_colors, int32_t *node_classes)
{
  PMG_CHECK(pc && !strcmp(pc->type, "parsor"), PMG_ERR_ARG_WRONG, "not a parsor PC");
  PMG_CALL(pmg_pc_setup(pc));
  pc_parsor *d = (pc_parsor *)pc->data;
  PMG_CHECK(d->nparts > 0, PMG_ERR_ARG_WRONGSTATE, "no rank partition set");
  if (nlevels) *nlevels = d->nlevels;
  if (proc_colors) memcpy(proc_colors, d->proccols, sizeof(int32_t) * (size_t)d->nparts);
  if (node_classes) memcpy(node_classes, d->classes, sizeof(int32_t) * (size_t)pc->pmat->n);
  return PMG_SUCCESS;
}
pmg_status pmg_pc_parsor_set_iterations(pmg_pc pc, int32_t its)
{
  PMG_CHECK(pc && !strcmp(pc->type, "parsor"), PMG_ERR_ARG_WRONG, "not a parsor PC");
  ((pc_parsor *)pc->data)->its = its;
  return PMG_SUCCESS;
}

/* ---------------------------------------------------------------------------------------------------- */
/* "cholsampler" (src/pc_chols.c)                                                                        */
/* ---------------------------------------------------------------------------------------------------- */
typedef struct {
  pmg_chol ch;
} pc_chols;
static pmg_status chols_reset(pmg_pc pc) { return pmg_chol_destroy(&((pc_chols *)pc->data)->ch); }
static pmg_status chols_destroy(pmg_pc pc)
{
  if (pc->data) {
    chols_reset(pc);
    free(pc->data);
    pc->data = NULL;
  }
  return PMG_SUCCESS;
}
static pmg_status chols_setup(pmg_pc pc)
{
  pc_chols *d = (pc_chols *)pc->data;
  chols_reset(pc);
  PMG_CHECK(pc->pmat->kind == 0, PMG_ERR_SUP, "cholsampler needs an assembled matrix");
  return pmg_chol_create_csr_lowrank(pc->pmat->n, pc->pmat->rowptr, pc->pmat->colidx, pc->pmat->vals, pc->pmat->lrc_k, pc->pmat->lrc_B, pc->pmat->lrc_S, &d->ch); /* :119-153 */
}
static pmg_status chols_apply(pmg_pc pc, const double *b, double *y, void *stream) /* PCApply_CholSampler :262-291 */
{
  PMG_CALL(pmg_chol_sample(((pc_chols *)pc->data)->ch, b, y, 1, pc_seed(pc), pc->counter++, stream));
  return PMG_SUCCESS;
}
static pmg_status chols_applyrichardson(pmg_pc pc, const double *b, double *y, int32_t its, int guesszero, int32_t *outits, void *stream)
{
  (void)guesszero;
  for (int32_t it = 0; it < its; ++it) { /* independent exact samples, callback after each (:293-342) */
    PMG_CALL(chols_apply(pc, b, y, stream));
    PMG_CALL(pc_notify(pc, it, y, stream));
  }
  *outits = its;
  return PMG_SUCCESS;
}
static pmg_status PCCreate_CholSampler(pmg_pc pc)
{
  pc_chols *d = (pc_chols *)calloc(1, sizeof *d);
  PMG_CHECK(d, PMG_ERR_MEM, "out of host memory");
  pc->data                = d;
  pc->ops.setup           = chols_setup;
  pc->ops.apply           = chols_apply;
  pc->ops.applyrichardson = chols_applyrichardson;
  pc->ops.reset           = chols_reset;
  pc->ops.destroy         = chols_destroy;
  return PMG_SUCCESS;
}

/* ---------------------------------------------------------------------------------------------------- */
/* "gamgmc" (src/pc_gamgmc.c) with -pc_gamgmc_mg_type mg on a DMDA                                       */
/* ---------------------------------------------------------------------------------------------------- */
typedef struct {
  pmg_mgmc mg;
  int      levels, nu, coarse_its, smoother_mc, coarse_kind, sweep;
  double   omega;
} pc_gamgmc;
static pmg_status gamgmc_reset(pmg_pc pc) { return pmg_mgmc_destroy(&((pc_gamgmc *)pc->data)->mg); }
static pmg_status gamgmc_destroy(pmg_pc pc)
{
  if (pc->data) {
    gamgmc_reset(pc);
    free(pc->data);
    pc->data = NULL;
  }
  return PMG_SUCCESS;
}
static pmg_status gamgmc_setfromoptions(pmg_pc pc) /* src/pc_gamgmc.c:299-366: defaults injected, overridable */
{
  pc_gamgmc  *d = (pc_gamgmc *)pc->data;
  char        pre[160];
  const char *v;
  snprintf(pre, sizeof pre, "%sgamgmc_", pc->prefix);
  if ((v = opt_find(pc->prefix, "-pc_gamgmc_mg_type")) && strcmp(v, "mg")) PMG_FAIL(PMG_ERR_SUP, "-pc_gamgmc_mg_type %s: only the geometric hierarchy (mg) on a DMDA is built here; GAMG aggregation is PETSc's", v);
  opt_int(pre, "-pc_mg_levels", &d->levels);
  opt_int(pre, "-mg_levels_ksp_max_it", &d->nu);
  opt_int(pre, "-mg_coarse_ksp_max_it", &d->coarse_its);
  if ((v = opt_find(pre, "-mg_levels_pc_type"))) {
    if (!strcmp(v, "mcgibbs")) d->smoother_mc = 1;
    else if (!strcmp(v, "sorgibbs")) d->smoother_mc = 0;
    else PMG_FAIL(PMG_ERR_SUP, "level sampler %s", v);
  }
  if ((v = opt_find(pre, "-mg_coarse_pc_type"))) {
    if (!strcmp(v, "cholsampler")) d->coarse_kind = 0;
    else if (!strcmp(v, "mcgibbs") || !strcmp(v, "sorgibbs")) d->coarse_kind = 1;
    else PMG_FAIL(PMG_ERR_SUP, "coarse sampler %s", v);
  }
  opt_real(pre, "-mg_levels_pc_mcgibbs_omega", &d->omega);
  if (opt_bool(pre, "-mg_levels_pc_mcgibbs_symmetric")) d->sweep = PMG_SOR_SYMMETRIC_SWEEP;
  if (opt_bool(pre, "-mg_levels_pc_mcgibbs_backward")) d->sweep = PMG_SOR_BACKWARD_SWEEP;
  pc->setupcalled = 0;
  return PMG_SUCCESS;
}
static pmg_status gamgmc_setup(pmg_pc pc)
{
  pc_gamgmc *d = (pc_gamgmc *)pc->data;
  gamgmc_reset(pc);
  PMG_CHECK(pc->pmat->kind == 1, PMG_ERR_SUP, "gamgmc here needs a DMDA operator (PCSetDM, src/pc_gamgmc.c:290)");
  PMG_CALL(pmg_mgmc_create_dmda(pc->pmat->nx, pc->pmat->ny, pc->pmat->nz, pc->pmat->kappa, d->levels, &d->mg));
  PMG_CALL(pmg_mgmc_set_smoother(d->mg, d->smoother_mc, d->smoother_mc ? d->omega : 1.0, d->smoother_mc ? d->sweep : PMG_SOR_FORWARD_SWEEP, d->nu));
  PMG_CALL(pmg_mgmc_set_coarse(d->mg, d->coarse_kind, d->coarse_its));
  if (pc->pmat->lrc_k) PMG_CALL(pmg_mgmc_set_lowrank(d->mg, pc->pmat->lrc_k, pc->pmat->lrc_B, pc->pmat->lrc_S)); /* src/pc_gamgmc.c:157-196 */
  return pmg_mgmc_setup(d->mg);
}
static pmg_status gamgmc_applyrichardson(pmg_pc pc, const double *b, double *y, int32_t its, int guesszero, int32_t *outits, void *stream)
{
  pc_gamgmc *d = (pc_gamgmc *)pc->data;
  PMG_CALL(pmg_mgmc_sample(d->mg, b, y, its, guesszero, pc_seed(pc), pc->counter, &pc->counter, pc->scb, pc->cbctx, stream));
  *outits = its;
  return PMG_SUCCESS;
}
static pmg_status PCCreate_GAMGMC(pmg_pc pc)
{
  pc_gamgmc *d = (pc_gamgmc *)calloc(1, sizeof *d);
  PMG_CHECK(d, PMG_ERR_MEM, "out of host memory");
  d->levels               = 2;
  d->nu                   = 1;
  d->coarse_its           = 1;
  d->omega                = 1.0;
  d->sweep                = PMG_SOR_FORWARD_SWEEP;
  pc->data                = d;
  pc->ops.setup           = gamgmc_setup;
  pc->ops.applyrichardson = gamgmc_applyrichardson; /* no apply: src/pc_gamgmc.c:405-410 */
  pc->ops.reset           = gamgmc_reset;
  pc->ops.destroy         = gamgmc_destroy;
  pc->ops.setfromoptions  = gamgmc_setfromoptions;
  return PMG_SUCCESS;
}
/* PCGAMGMCSetLevels (include/parmgmc/pc/pc_gamgmc.h:15) */
pmg_status pmg_pc_gamgmc_set_levels(pmg_pc pc, int32_t levels)
{
  PMG_CHECK(pc && !strcmp(pc->type, "gamgmc"), PMG_ERR_ARG_WRONG, "not a gamgmc PC");
  ((pc_gamgmc *)pc->data)->levels = levels;
  pc->setupcalled                 = 0;
  return PMG_SUCCESS;
}

/* ---------------------------------------------------------------------------------------------------- */
/* "shell": PCSHELL (examples/ex3.c:59-67,128-131)                                                       */
/* ---------------------------------------------------------------------------------------------------- */
typedef struct {
  pmg_status (*apply)(pmg_pc, const double *, double *, void *);
  void *ctx;
} pc_shell;
static pmg_status shell_apply(pmg_pc pc, const double *b, double *y, void *stream)
{
  pc_shell *d = (pc_shell *)pc->data;
  PMG_CHECK(d->apply, PMG_ERR_ARG_WRONGSTATE, "No apply() routine provided to Shell PC");
  return d->apply(pc, b, y, stream);
}
static pmg_status shell_destroy(pmg_pc pc)
{
  free(pc->data);
  pc->data = NULL;
  return PMG_SUCCESS;
}
static pmg_status PCCreate_Shell(pmg_pc pc)
{
  pc_shell *d = (pc_shell *)calloc(1, sizeof *d);
  PMG_CHECK(d, PMG_ERR_MEM, "out of host memory");
  pc->data        = d;
  pc->ops.apply   = shell_apply;
  pc->ops.destroy = shell_destroy;
  return PMG_SUCCESS;
}
pmg_status pmg_pc_shell_set_apply(pmg_pc pc, pmg_status (*apply)(pmg_pc, const double *, double *, void *))
{
  PMG_CHECK(pc && !strcmp(pc->type, "shell"), PMG_ERR_ARG_WRONG, "not a shell PC");
  ((pc_shell *)pc->data)->apply = apply;
  return PMG_SUCCESS;
}
pmg_status pmg_pc_shell_set_context(pmg_pc pc, void *ctx)
{
  PMG_CHECK(pc && !strcmp(pc->type, "shell"), PMG_ERR_ARG_WRONG, "not a shell PC");
  ((pc_shell *)pc->data)->ctx = ctx;
  return PMG_SUCCESS;
}
pmg_status pmg_pc_shell_get_context(pmg_pc pc, void **ctx)
{
  PMG_CHECK(pc && ctx && !strcmp(pc->type, "shell"), PMG_ERR_ARG_WRONG, "not a shell PC");
  *ctx = ((pc_shell *)pc->data)->ctx;
  return PMG_SUCCESS;
}

/* ---------------------------------------------------------------------------------------------------- */
/* "woodbury" (src/woodbury.c): a sampler for A + B S B^T assembled from ANY sampler of A plus a solver   */
/* ---------------------------------------------------------------------------------------------------- */
typedef struct {
  pmg_pc       solver, sampler; /* owned */
  pmg_mat      Abase;           /* the base operator of the MATLRC matrix, owned copy of the descriptor */
  pmg_woodbury wb;              /* B, G = C (S^-1 + B^T C)^-1 and the dense products (pmg_woodbury.c) */
  double      *w;               /* device: noisy right-hand side (n) */
} pc_woodbury;

static pmg_status woodbury_free_setup(pc_woodbury *d)
{
  pmg_woodbury_destroy(&d->wb);
  pmg_dev_free(d->w);
  d->w = NULL;
  pmg_mat_destroy(&d->Abase);
  return PMG_SUCCESS;
}
static pmg_status woodbury_reset(pmg_pc pc) /* PCReset_Woodbury :93-109 */
{
  pc_woodbury *d = (pc_woodbury *)pc->data;
  woodbury_free_setup(d);
  if (d->solver) PMG_CALL(pmg_pc_reset(d->solver));
  if (d->sampler) PMG_CALL(pmg_pc_reset(d->sampler));
  return PMG_SUCCESS;
}
static pmg_status woodbury_destroy(pmg_pc pc) /* PCDestroy_Woodbury :111-125 */
{
  pc_woodbury *d = (pc_woodbury *)pc->data;
  if (d) {
    woodbury_free_setup(d);
    pmg_pc_destroy(&d->solver);
    pmg_pc_destroy(&d->sampler);
    free(d);
    pc->data = NULL;
  }
  return PMG_SUCCESS;
}

/* PCWoodburySetSolver / PCWoodburySetSampler (:185-213): the inner PC gets the outer prefix plus
   "pc_woodbury_solver_" / "pc_woodbury_sampler" (the reference appends the sampler prefix WITHOUT a trailing
   underscore, :208 -- kept, so the option keys are the reference's).  The woodbury PC takes the reference the
   caller held: do not destroy `inner` afterwards. */
static pmg_status woodbury_adopt(pmg_pc pc, pmg_pc inner, int is_solver)
{
  PMG_CHECK(pc && inner && !strcmp(pc->type, "woodbury"), PMG_ERR_ARG_WRONG, "not a woodbury PC");
  pc_woodbury *d = (pc_woodbury *)pc->data;
  char         pre[sizeof inner->prefix + 32];
  snprintf(pre, sizeof pre, "%s%s", pc->prefix, is_solver ? "pc_woodbury_solver_" : "pc_woodbury_sampler");
  PMG_CALL(pmg_pc_set_options_prefix(inner, pre));
  pmg_pc *slot = is_solver ? &d->solver : &d->sampler;
  if (*slot && *slot != inner) pmg_pc_destroy(slot);
  *slot           = inner;
  pc->setupcalled = 0;
  return PMG_SUCCESS;
}
pmg_status pmg_pc_woodbury_set_solver(pmg_pc pc, pmg_pc solver) { return woodbury_adopt(pc, solver, 1); }
pmg_status pmg_pc_woodbury_set_sampler(pmg_pc pc, pmg_pc sampler) { return woodbury_adopt(pc, sampler, 0); }

static pmg_status woodbury_set_inner_type(pmg_pc pc, const char *type, int is_solver) /* :215-243 */
{
  pmg_pc inner = NULL;
  PMG_CALL(pmg_pc_create(&inner));
  pmg_status st = woodbury_adopt(pc, inner, is_solver);
  if (st) {
    pmg_pc_destroy(&inner);
    return st;
  }
  return pmg_pc_set_type(inner, type);
}
static pmg_status woodbury_setfromoptions(pmg_pc pc) /* PCSetFromOptions_Woodbury :245-261 */
{
  pc_woodbury *d = (pc_woodbury *)pc->data;
  const char  *v;
  if ((v = opt_find(pc->prefix, "-pc_woodbury_solver"))) PMG_CALL(woodbury_set_inner_type(pc, v, 1));
  if ((v = opt_find(pc->prefix, "-pc_woodbury_sampler"))) PMG_CALL(woodbury_set_inner_type(pc, v, 0));
  if (d->solver && d->solver->ops.setfromoptions) PMG_CALL(d->solver->ops.setfromoptions(d->solver));
  if (d->sampler && d->sampler->ops.setfromoptions) PMG_CALL(d->sampler->ops.setfromoptions(d->sampler));
  pc->setupcalled = 0;
  return PMG_SUCCESS;
}

static pmg_status woodbury_setup(pmg_pc pc) /* PCSetUp_Woodbury :142-183 + PCWoodburyBuildLRCCorrection :21-91 */
{
  pc_woodbury *d = (pc_woodbury *)pc->data;
  PMG_CHECK(d->solver && d->sampler, PMG_ERR_SUP, "Must provide sampler and solver");          /* :151 */
  PMG_CHECK(pc->pmat->lrc_k, PMG_ERR_SUP, "PCWoodbury only supports matrices of type LRC");    /* :161 */
  woodbury_free_setup(d);
  const int32_t n = pc->pmat->n, k = pc->pmat->lrc_k;
  d->Abase = (pmg_mat)malloc(sizeof *d->Abase);
  PMG_CHECK(d->Abase, PMG_ERR_MEM, "out of host memory");
  *d->Abase       = *pc->pmat; /* MatLRCGetMats(pc->pmat, &A, &B, &S, NULL), :162 */
  d->Abase->lrc_k = 0;
  d->Abase->lrc_B = d->Abase->lrc_S = NULL;
  PMG_CALL(pmg_woodbury_create(n, k, pc->pmat->lrc_B, n, pc->pmat->lrc_S, NULL, &d->wb));
  PMG_CALL(pmg_dev_alloc((void **)&d->w, sizeof(double) * (size_t)n));
  PMG_CALL(pmg_pc_set_operators(d->solver, d->Abase)); /* :178-181 */
  PMG_CALL(pmg_pc_set_operators(d->sampler, d->Abase));
  PMG_CALL(pmg_pc_setup(d->solver));
  PMG_CALL(pmg_pc_setup(d->sampler));
  PMG_CHECK(d->sampler->ops.applyrichardson, PMG_ERR_SUP, "PC type %s does not have applyrichardson", d->sampler->type);
  for (int c = 0; c < k; ++c) { /* C = solver(B) column by column from a zero guess (:35-50) */
    const double *bc;
    double       *cc;
    PMG_CALL(pmg_woodbury_column(d->wb, c, &bc, &cc, NULL));
    PMG_CALL(pmg_pc_apply(d->solver, bc, cc, NULL)); /* PCApply(wb->solver, b, x), :45 */
  }
  PMG_CALL(pmg_woodbury_finish(d->wb)); /* G = C (S^-1 + B^T C)^-1, :52-81 */
  pmg_pc_destroy(&d->solver);           /* :182 */
  return PMG_SUCCESS;
}

static pmg_status woodbury_applyrichardson(pmg_pc pc, const double *b, double *y, int32_t its, int guesszero, int32_t *outits, void *stream)
{ /* PCApplyRichardson_Woodbury :263-289 */
  (void)guesszero;
  pc_woodbury *d = (pc_woodbury *)pc->data;
  for (int32_t it = 0; it < its; ++it) {
    PMG_CALL(pmg_woodbury_noisy_rhs(d->wb, b, d->w, pc_seed(pc), pc->counter++, stream)); /* w = b + B (sqrt|S| o xi), :275-277 */
    int32_t done = 0;
    PMG_CALL(d->sampler->ops.applyrichardson(d->sampler, d->w, y, 1, 0, &done, stream)); /* one sample of the A-sampler, :278    */
    PMG_CALL(pmg_woodbury_correct(d->wb, y, stream));                                     /* y -= G (B^T y), :280-282            */
    PMG_CALL(pc_notify(pc, it, y, stream));
  }
  *outits = its;
  return PMG_SUCCESS;
}

static pmg_status PCCreate_Woodbury(pmg_pc pc) /* :291-302 */
{
  pc_woodbury *d = (pc_woodbury *)calloc(1, sizeof *d);
  PMG_CHECK(d, PMG_ERR_MEM, "out of host memory");
  pc->data                = d;
  pc->ops.setup           = woodbury_setup;
  pc->ops.reset           = woodbury_reset;
  pc->ops.destroy         = woodbury_destroy;
  pc->ops.setfromoptions  = woodbury_setfromoptions;
  pc->ops.applyrichardson = woodbury_applyrichardson;
  return PMG_SUCCESS;
}

/* ---------------------------------------------------------------------------------------------------- */
/* ParMGMCInitialize / ParMGMCFinalize (src/parmgmc.c:118-137)                                           */
/* ---------------------------------------------------------------------------------------------------- */
pmg_status pmg_initialize(void)
{
  PMG_CALL(pmg_pc_register("sorgibbs", PCCreate_SORGibbs));
  PMG_CALL(pmg_pc_register("mcgibbs", PCCreate_MulticolorGibbs));
  PMG_CALL(pmg_pc_register("gamgmc", PCCreate_GAMGMC));
  PMG_CALL(pmg_pc_register("cholsampler", PCCreate_CholSampler));
  PMG_CALL(pmg_pc_register("parsor", PCCreate_PARSOR));
  PMG_CALL(pmg_pc_register("woodbury", PCCreate_Woodbury));
  PMG_CALL(pmg_pc_register("shell", PCCreate_Shell));
  return PMG_SUCCESS;
}

pmg_status pmg_finalize(void)
{
  pmg_ntypes = 0;
  pmg_nopts  = 0;
  return PMG_SUCCESS;
}
