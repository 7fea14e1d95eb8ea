/*
 * pmg_oracle.c -- CPU restatement of the ParMGMC Gibbs/SOR hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke() check of
 * __graft_entry__.py and the cpu_baseline leg of bench.py may load it.  The shipped path is
 * the HIP library under parmgmc_amd/csrc and never calls into this file.
 *
 * Every function restates, in plain scalar C on raw arrays, the loop of the reference
 * (nilsfriess/ParMGMC) named in its header comment.  PETSc is not available in this
 * container, so the reference itself cannot be compiled (oracle/_ref is "unbuildable":
 * every src file includes petsc headers); the restatement is pinned instead by the
 * reference's own known-answer tests (examples/ex5.c, ex1.c, ex6.c + src/stats.c), see
 * tests/test_oracle_*.py.
 *
 * Floating point: compile with -ffp-contract=off so that every a*b+c rounds twice, exactly
 * as the reference does when built by PETSc's default flags (no FMA contraction by gcc
 * without -march flags).  The HIP kernels are built the same way, which is what makes the
 * deterministic-sweep parity tests bit-exact.
 *
 * Index type: 32-bit (PetscInt default).  Scalars: double (PetscScalar = PetscReal).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_SOR_FORWARD 1   /* SOR_FORWARD_SWEEP  */
#define ORC_SOR_BACKWARD 2  /* SOR_BACKWARD_SWEEP */
#define ORC_SOR_SYMMETRIC 3 /* SOR_SYMMETRIC_SWEEP = fwd|bwd */

/* ------------------------------------------------------------------------------------------
 * Operator generators
 * ---------------------------------------------------------------------------------------- */

/* Number of stored entries of the 2-D 5-point / 3-D 7-point operator on an nx*ny*nz vertex
   grid (nz = 1 gives the 2-D operator). */
int64_t orc_laplace_nnz(int nx, int ny, int nz)
{
  int64_t n = (int64_t)nx * ny * nz;
  return n + 2 * ((int64_t)(nx - 1) * ny * nz + (int64_t)nx * (ny - 1) * nz + (int64_t)nx * ny * (nz - 1));
}

/* MatAssembleShiftedLaplaceFD, reference src/problems.c:14-75, generalised from 2-D to 3-D
   (nz = 1 is the reference operator exactly).
     - "hinv2" is 1/((mx-1)*(mx-1)): an INTEGER product divided into 1.0 (problems.c:24) and
       mx is used for both directions;
     - every in-domain neighbour contributes the off-diagonal -hinv2 and adds hinv2 to the
       diagonal BY REPEATED ADDITION starting from kappa*kappa (problems.c:27,37,44,51,58),
       not by one multiplication;
     - rows in DMDA natural order idx = i + nx*(j + ny*k); PETSc AIJ keeps the columns of a
       row sorted ascending whatever the insertion order, so the stored order is
       (k-1) (j-1) (i-1) diag (i+1) (j+1) (k+1). */
/* one row of that matrix: columns (64-bit) and values in storage order, *dpos = position of the diagonal;
   returns the number of entries (<= 7).  Shared by the assembly below and by the sampled-row checks further down,
   so that the rows recomputed at 512^3 are the rows the pinned assembly produces. */
static inline int orc_laplace_row(int nx, int ny, int nz, double kappa, double hinv2, int i, int j, int k, int64_t *cols, double *vals, int *dpos)
{
  const int64_t row  = i + (int64_t)nx * (j + (int64_t)ny * k);
  double        diag = kappa * kappa;
  int           p    = 0;
  /* the reference adds in the order south, west, north, east; all addends are equal so
     the sum is order independent -- only the COUNT of additions matters */
  if (k > 0) diag += hinv2;
  if (j > 0) diag += hinv2;
  if (i > 0) diag += hinv2;
  if (k < nz - 1) diag += hinv2;
  if (j < ny - 1) diag += hinv2;
  if (i < nx - 1) diag += hinv2;
  if (k > 0) { cols[p] = row - (int64_t)nx * ny; vals[p++] = -hinv2; }
  if (j > 0) { cols[p] = row - nx; vals[p++] = -hinv2; }
  if (i > 0) { cols[p] = row - 1; vals[p++] = -hinv2; }
  *dpos = p;
  cols[p] = row; vals[p++] = diag;
  if (i < nx - 1) { cols[p] = row + 1; vals[p++] = -hinv2; }
  if (j < ny - 1) { cols[p] = row + nx; vals[p++] = -hinv2; }
  if (k < nz - 1) { cols[p] = row + (int64_t)nx * ny; vals[p++] = -hinv2; }
  return p;
}

void orc_assemble_shifted_laplace(int nx, int ny, int nz, double kappa, int32_t *rowptr, int32_t *colidx, double *vals)
{
  const double hinv2 = 1. / ((nx - 1) * (nx - 1));
  int64_t      p     = 0;
  for (int k = 0; k < nz; k++)
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++) {
        const int64_t row = i + (int64_t)nx * (j + (int64_t)ny * k);
        int64_t       c[7];
        double        v[7];
        int           dpos;
        const int     n = orc_laplace_row(nx, ny, nz, kappa, hinv2, i, j, k, c, v, &dpos);
        rowptr[row]     = (int32_t)p;
        for (int q = 0; q < n; ++q) { colidx[p] = (int32_t)c[q]; vals[p++] = v[q]; }
      }
  rowptr[(int64_t)nx * ny * nz] = (int32_t)p;
}

/* AssembleMatrix of reference examples/ex6.c:69-127: off-diagonals -1, diagonal
   (number of neighbours) + kappa, where the count is an int added to the double kappa
   (ex6.c:113 "values[k] = k + kappa").  2-D, n x n. */
void orc_assemble_ex6(int n, double kappa, int32_t *rowptr, int32_t *colidx, double *vals)
{
  int p = 0;
  for (int j = 0; j < n; j++)
    for (int i = 0; i < n; i++) {
      const int row = i + n * j;
      int       cnt = (i != 0) + (i != n - 1) + (j != 0) + (j != n - 1);
      rowptr[row]   = p;
      if (j != 0) { colidx[p] = row - n; vals[p++] = -1; }
      if (i != 0) { colidx[p] = row - 1; vals[p++] = -1; }
      colidx[p] = row; vals[p++] = cnt + kappa;
      if (i != n - 1) { colidx[p] = row + 1; vals[p++] = -1; }
      if (j != n - 1) { colidx[p] = row + n; vals[p++] = -1; }
    }
  rowptr[n * n] = p;
}

/* ------------------------------------------------------------------------------------------
 * MCSOR set-up pieces
 * ---------------------------------------------------------------------------------------- */

/* MatGetDiagonalPointers, reference src/mc_sor.c:126-150: CSR position of the diagonal entry
   of every row (last match wins, as in the reference loop). */
void orc_diag_pointers(int n, const int32_t *rowptr, const int32_t *colidx, int32_t *diagptr)
{
  for (int row = 0; row < n; ++row)
    for (int k = rowptr[row]; k < rowptr[row + 1]; ++k)
      if (colidx[k] == row) diagptr[row] = k;
}

/* MCSORUpdateIDiag, reference src/mc_sor.c:114-124: MatGetDiagonal, VecReciprocal,
   VecScale(omega) -- i.e. idiag = (1/d) * omega, TWO roundings (not omega/d). */
void orc_idiag(int n, const int32_t *diagptr, const double *vals, double omega, double *idiag)
{
  for (int r = 0; r < n; ++r) {
    double t = 1.0 / vals[diagptr[r]];
    idiag[r] = t * omega;
  }
}

/* PCMulticolorGibbsUpdateSqrtDiag, reference src/pc_mcgibbs.c:142-153: sqrt(|d|) scaled by
   sqrt((2-omega)/omega); PCSetUp_SORGibbs (src/pc_sorgibbs.c:233-236) is the omega = 1 case
   WITHOUT the scale call. Pass scale = 0 to get the unscaled sorgibbs variant. */
void orc_sqrtdiag(int n, const int32_t *diagptr, const double *vals, double omega, int scale, double *sqrtdiag)
{
  const double s = sqrt((2 - omega) / omega);
  for (int r = 0; r < n; ++r) {
    double t    = sqrt(fabs(vals[diagptr[r]]));
    sqrtdiag[r] = scale ? t * s : t;
  }
}

/* ------------------------------------------------------------------------------------------
 * Sweeps
 * ---------------------------------------------------------------------------------------- */

/* One row update: the body shared by reference src/mc_sor.c:261-267, :278-284 and
   src/pc_parsor.c:676-697 (SparseDenseMinusDot :88-91): start from b[r], subtract the
   strictly-lower then the strictly-upper part in storage order, then
   y[r] = (1-omega)*y[r] + idiag[r]*sum. */
static inline void orc_row_update(int r, const int32_t *rowptr, const int32_t *colidx, const double *vals, const int32_t *diagptr, const double *idiag, double omega, const double *b, double *y)
{
  double sum = b[r];
  for (int k = rowptr[r]; k < diagptr[r]; ++k) sum -= vals[k] * y[colidx[k]];
  for (int k = diagptr[r] + 1; k < rowptr[r + 1]; ++k) sum -= vals[k] * y[colidx[k]];
  y[r] = (1. - omega) * y[r] + idiag[r] * sum;
}

/* MCSORApply_SEQAIJ, reference src/mc_sor.c:241-296.  Colours are given as an index-set
   list: rows of colour c are colorrows[colorptr[c] .. colorptr[c+1]).  Forward visits
   colours ascending and rows in list order; backward visits colours descending AND the rows
   of a colour in reverse list order (mc_sor.c:274,277). `type` is FORWARD or BACKWARD. */
void orc_mcsor_sweep_seq(int type, int ncolors, const int32_t *colorptr, const int32_t *colorrows, const int32_t *rowptr, const int32_t *colidx, const double *vals, const int32_t *diagptr, const double *idiag, double omega, const double *b, double *y)
{
  if (type == ORC_SOR_FORWARD) {
    for (int c = 0; c < ncolors; ++c)
      for (int i = colorptr[c]; i < colorptr[c + 1]; ++i) orc_row_update(colorrows[i], rowptr, colidx, vals, diagptr, idiag, omega, b, y);
  }
  if (type == ORC_SOR_BACKWARD) {
    for (int c = ncolors - 1; c >= 0; --c)
      for (int i = colorptr[c + 1] - 1; i >= colorptr[c]; --i) orc_row_update(colorrows[i], rowptr, colidx, vals, diagptr, idiag, omega, b, y);
  }
}

/* MCSORApply, reference src/mc_sor.c:216-239: symmetric = one forward then one backward
   sweep with the SAME right-hand side (the low-rank post-correction hook is separate). */
void orc_mcsor_apply(int type, int ncolors, const int32_t *colorptr, const int32_t *colorrows, const int32_t *rowptr, const int32_t *colidx, const double *vals, const int32_t *diagptr, const double *idiag, double omega, const double *b, double *y)
{
  if (type == ORC_SOR_SYMMETRIC) {
    orc_mcsor_sweep_seq(ORC_SOR_FORWARD, ncolors, colorptr, colorrows, rowptr, colidx, vals, diagptr, idiag, omega, b, y);
    orc_mcsor_sweep_seq(ORC_SOR_BACKWARD, ncolors, colorptr, colorrows, rowptr, colidx, vals, diagptr, idiag, omega, b, y);
  } else {
    orc_mcsor_sweep_seq(type, ncolors, colorptr, colorrows, rowptr, colidx, vals, diagptr, idiag, omega, b, y);
  }
}

/* MCSORApply_MPIAIJ, reference src/mc_sor.c:298-381, for ONE rank and ONE colour (the colour
   loop and the ghost scatter live in the caller, oracle/__init__.py:mcsor_sweep_domains,
   which restates MatCreateScatters :152-214).  The rank's matrix is split the PETSc way into
   a diagonal block `a*` (local column numbering) and an off-diagonal block `b*`.  `ghost`
   holds one entry per off-process nonzero of every row of this colour, in row-visit order,
   duplicates included (:197-198); the forward sweep consumes it sequentially (gcnt++, :332),
   the backward sweep walks the rows in reverse and recomputes each row's offset from the end
   (:356-366).  Note the operand order differs from the sequential kernel: the sum starts at
   0 and b is added last, idiag*(sum + b) (:334). */
void orc_mcsor_rank_color(int type, int nrows_c, const int32_t *rows, const int32_t *aR, const int32_t *aC, const double *aV, const int32_t *diagptr, const int32_t *bR, const double *bV, const double *ghost, const double *idiag, double omega, const double *b, double *y)
{
  if (type == ORC_SOR_FORWARD) {
    int gcnt = 0;
    for (int i = 0; i < nrows_c; ++i) {
      const int r   = rows[i];
      double    sum = 0;
      for (int k = aR[r]; k < diagptr[r]; ++k) sum -= aV[k] * y[aC[k]];
      for (int k = diagptr[r] + 1; k < aR[r + 1]; ++k) sum -= aV[k] * y[aC[k]];
      for (int k = bR[r]; k < bR[r + 1]; ++k) sum -= bV[k] * ghost[gcnt++];
      y[r] = (1 - omega) * y[r] + idiag[r] * (sum + b[r]);
    }
  } else {
    int gcnt = 0;
    for (int i = 0; i < nrows_c; ++i) gcnt += bR[rows[i] + 1] - bR[rows[i]]; /* = ghost vec size */
    for (int i = nrows_c - 1; i >= 0; --i) {
      const int r   = rows[i];
      double    sum = 0;
      gcnt -= bR[r + 1] - bR[r];
      for (int k = aR[r]; k < diagptr[r]; ++k) sum -= aV[k] * y[aC[k]];
      for (int k = diagptr[r] + 1; k < aR[r + 1]; ++k) sum -= aV[k] * y[aC[k]];
      int go = gcnt;
      for (int k = bR[r]; k < bR[r + 1]; ++k) sum -= bV[k] * ghost[go++];
      y[r] = (1 - omega) * y[r] + idiag[r] * (sum + b[r]);
    }
  }
}

/* SORLocalForwardSweepIS, reference src/pc_parsor.c:666-701: the row update over an
   arbitrary row list with an optional off-diagonal block applied to a ghost vector `lv`
   that is indexed by LOCAL GHOST COLUMN (unlike the sequential buffer of MCSOR).  Note the
   operand order of the last line differs from mc_sor.c: sum*idiag, same value. */
void orc_parsor_rows(int nrows, const int32_t *rows, const int32_t *rowptr, const int32_t *colidx, const double *vals, const int32_t *diagptr, const double *idiag, double omega, const double *b, double *x, const int32_t *browptr, const int32_t *bcolidx, const double *bvals, const double *lv)
{
  for (int j = 0; j < nrows; ++j) {
    const int i   = rows[j];
    double    sum = b[i];
    for (int k = rowptr[i]; k < diagptr[i]; ++k) sum -= vals[k] * x[colidx[k]];
    for (int k = diagptr[i] + 1; k < rowptr[i + 1]; ++k) sum -= vals[k] * x[colidx[k]];
    if (browptr)
      for (int k = browptr[i]; k < browptr[i + 1]; ++k) sum -= bvals[k] * lv[bcolidx[k]];
    x[i] = (1. - omega) * x[i] + sum * idiag[i];
  }
}

/* ------------------------------------------------------------------------------------------
 * Noise
 * ---------------------------------------------------------------------------------------- */

/* Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
   SC'11; the Random123 library's philox4x32_R(10, ...)).  Multipliers 0xD2511F53 and
   0xCD9E8D57, Weyl key increments 0x9E3779B9 and 0xBB67AE85.  This is the uniform source
   that replaces PetscRandom (reference src/parmgmc.c:56-68), whose streams a GPU cannot
   reproduce; the known-answer vectors of the Random123 distribution pin it. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* 53-bit uniform in (0,1] from two 32-bit words: never 0, so log() below is finite
   (the reference's PetscRandom can return 0 and then produces inf, src/parmgmc.c:103-106). */
static inline double orc_u53(uint32_t lo, uint32_t hi)
{
  uint64_t x = ((uint64_t)hi << 32) | lo;
  return (double)((x >> 11) + 1) * 0x1.0p-53;
}

/* One Box-Muller pair, reference src/parmgmc.c:99-110:
     radius = sqrt(-2 log u1); theta = 2 pi u2; z0 = radius cos(theta); z1 = radius sin(theta). */
static inline void orc_box_muller_pair(double u1, double u2, double *z0, double *z1)
{
  const double radius = sqrt(-2.0 * log(u1));
  const double theta  = 2.0 * 3.14159265358979323846264338327950288 * u2; /* PETSC_PI */
  *z0                 = radius * cos(theta);
  *z1                 = radius * sin(theta);
}

/* The pair of standard normals the library defines for Philox counter `ctr` and key `key`
   (uniform mapping above, Box-Muller as the reference). */
void orc_normal_pair(const uint32_t ctr[4], const uint32_t key[2], double z[2])
{
  uint32_t r[4];
  orc_philox4x32_10(ctr, key, r);
  orc_box_muller_pair(orc_u53(r[0], r[1]), orc_u53(r[2], r[3]), &z[0], &z[1]);
}

/* VecSetRandomStandardNormal, Box-Muller branch, reference src/parmgmc.c:99-110, on a
   pluggable uniform source: consecutive entries (i, i+1) share one (u1,u2) pair, the odd-n
   tail uses only the cosine branch.  `u` holds the uniforms in draw order (2*ceil(n/2)). */
void orc_vec_set_random_standard_normal(int n, const double *u, double *array)
{
  int q = 0;
  for (int i = 0; i < n; i += 2) {
    double z0, z1;
    orc_box_muller_pair(u[q], u[q + 1], &z0, &z1);
    q += 2;
    array[i] = z0;
    if (i + 1 < n) array[i + 1] = z1;
  }
}

/* Noise layout of the library for a CSR operator ("row stream"): entry r of sweep number s
   is the (r&1) branch of the normal pair with counter {r>>1 (low 32 bits), r>>33, s low,
   s high} -- i.e. exactly VecSetRandomStandardNormal's pairing of entries (2q, 2q+1) with the
   counter-based uniform source in place of the sequential one. */
void orc_noise_rows(int64_t n, uint64_t seed, uint64_t sweep, double *xi)
{
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  for (int64_t r = 0; r < n; r += 2) {
    const uint64_t q      = (uint64_t)r >> 1;
    const uint32_t ctr[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)sweep, (uint32_t)(sweep >> 32)};
    double         z[2];
    orc_normal_pair(ctr, key, z);
    xi[r] = z[0];
    if (r + 1 < n) xi[r + 1] = z[1];
  }
}

/* Noise layout of the library for a structured grid ("grid stream"), written to a vector in
   DMDA natural order.  Points of one colour c = (i+j+k)&1 along a grid line (j,k) are
   numbered m = i>>1; consecutive same-colour points (m = 2q, 2q+1) share one Box-Muller
   pair -- this is VecSetRandomStandardNormal's (i,i+1) pairing applied to the
   colour-partitioned storage order the device uses.  Counter = {q, line = j + ny*k (global),
   sweep low, (sweep high & 0x7fffffff) | c<<31}.  The definition uses GLOBAL indices only, so
   the noise does not depend on how the grid is split over devices. */
static inline double orc_noise_grid_point(int ny, int i, int j, int k, const uint32_t key[2], uint64_t sweep)
{
  const uint32_t line   = (uint32_t)(j + (int64_t)ny * k);
  const int      c      = (i + j + k) & 1;
  const int      m      = i >> 1;
  const uint32_t ctr[4] = {(uint32_t)(m >> 1), line, (uint32_t)sweep, ((uint32_t)(sweep >> 32) & 0x7fffffffu) | ((uint32_t)c << 31)};
  double         z[2];
  orc_normal_pair(ctr, key, z);
  return z[m & 1];
}

void orc_noise_grid(int nx, int ny, int nz, uint64_t seed, uint64_t sweep, double *xi)
{
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) xi[i + (int64_t)nx * (j + (int64_t)ny * k)] = orc_noise_grid_point(ny, i, j, k, key, sweep);
}

/* PrepareRHS_Default, reference src/pc_mcgibbs.c:119-128 (and its twin in PCSORGibbsSample,
   src/pc_sorgibbs.c:81-83): three separate vector passes
     w = xi ; w = w .* sqrtdiag ; w = w + b     (VecAXPY with alpha = 1 adds without scaling). */
void orc_prepare_rhs(int64_t n, const double *xi, const double *sqrtdiag, const double *b, double *w)
{
  for (int64_t r = 0; r < n; ++r) w[r] = xi[r];
  for (int64_t r = 0; r < n; ++r) w[r] = w[r] * sqrtdiag[r];
  if (b)
    for (int64_t r = 0; r < n; ++r) w[r] = w[r] + b[r];
}

/* ------------------------------------------------------------------------------------------
 * Lexicographic single-colour CPU baseline (the reference's serial behaviour)
 * ---------------------------------------------------------------------------------------- */

/* One full sorgibbs/mcgibbs sample in serial: noise (Box-Muller on the row stream), the three
   RHS passes, one lexicographic forward sweep (reference serial colouring = one colour,
   src/mc_sor.c:397-410, so MCSORApply_SEQAIJ is plain Gauss-Seidel).  Used by bench.py as
   the timed CPU baseline ("port"). */
void orc_gibbs_sample_serial(int n, const int32_t *rowptr, const int32_t *colidx, const double *vals, const int32_t *diagptr, const double *idiag, const double *sqrtdiag, double omega, const double *b, double *y, double *w, uint64_t seed, uint64_t sweep)
{
  orc_noise_rows(n, seed, sweep, w);
  for (int r = 0; r < n; ++r) w[r] = w[r] * sqrtdiag[r];
  for (int r = 0; r < n; ++r) w[r] = w[r] + b[r];
  for (int r = 0; r < n; ++r) orc_row_update(r, rowptr, colidx, vals, diagptr, idiag, omega, w, y);
}

/* The same sample with a distance-1 colouring on ALL host cores (OpenMP where the library is built with -fopenmp;
   plain loops otherwise): rows of one colour are independent (reference src/mc_sor.c:256-271 over the rows of an IS),
   so each colour is one parallel loop -- what the reference does with one MPI rank per core.  Used by bench.py as the
   second CPU baseline (cores = the threads OpenMP reports). */
void orc_set_num_threads(int n)
{
#ifdef _OPENMP
  extern void omp_set_num_threads(int);
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int orc_num_threads(void)
{
#ifdef _OPENMP
  extern int omp_get_max_threads(void);
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void orc_gibbs_sample_colored_parallel(int n, int ncolors, const int32_t *colorptr, const int32_t *colorrows, const int32_t *rowptr, const int32_t *colidx, const double *vals, const int32_t *diagptr, const double *idiag, const double *sqrtdiag, double omega, const double *b, double *y, double *w, uint64_t seed, uint64_t sweep)
{
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  const int64_t  npairs = ((int64_t)n + 1) / 2;
#pragma omp parallel for schedule(static)
  for (int64_t q = 0; q < npairs; ++q) { /* orc_noise_rows + the two RHS passes, fused per pair */
    const uint32_t ctr[4] = {(uint32_t)q, (uint32_t)((uint64_t)q >> 32), (uint32_t)sweep, (uint32_t)(sweep >> 32)};
    double         z[2];
    orc_normal_pair(ctr, key, z);
    const int64_t r = 2 * q;
    w[r]            = z[0] * sqrtdiag[r] + b[r];
    if (r + 1 < n) w[r + 1] = z[1] * sqrtdiag[r + 1] + b[r + 1];
  }
  for (int c = 0; c < ncolors; ++c) {
#pragma omp parallel for schedule(static)
    for (int32_t p = colorptr[c]; p < colorptr[c + 1]; ++p) orc_row_update(colorrows[p], rowptr, colidx, vals, diagptr, idiag, omega, w, y);
  }
}

/* ------------------------------------------------------------------------------------------
 * Coarse exact sampler
 * ---------------------------------------------------------------------------------------- */

/* Dense lower Cholesky in place, column-major, what LAPACKpotrf_("L",...) computes at
   reference src/pc_chols.c:188 (unblocked left-looking here; entries above the diagonal are
   left untouched as LAPACK does).  Returns 0 or the 1-based order of the failing minor. */
int orc_potrf_lower(int n, double *a)
{
  for (int j = 0; j < n; ++j) {
    double d = a[j + (size_t)n * j];
    for (int k = 0; k < j; ++k) d -= a[j + (size_t)n * k] * a[j + (size_t)n * k];
    if (!(d > 0)) return j + 1;
    d                    = sqrt(d);
    a[j + (size_t)n * j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = a[i + (size_t)n * j];
      for (int k = 0; k < j; ++k) s -= a[i + (size_t)n * k] * a[j + (size_t)n * k];
      a[i + (size_t)n * j] = s / d;
    }
  }
  return 0;
}

/* PCApply_CholSampler dense path, reference src/pc_chols.c:220-260,284-288:
   v = L^{-1} x (trsv L,N,N); v += xi; y = L^{-T} v (trsv L,T,N). */
void orc_chol_sample(int n, const double *L, const double *x, const double *xi, double *y)
{
  double *v = (double *)malloc(sizeof(double) * (size_t)n);
  for (int i = 0; i < n; ++i) {
    double s = x[i];
    for (int k = 0; k < i; ++k) s -= L[i + (size_t)n * k] * v[k];
    v[i] = s / L[i + (size_t)n * i];
  }
  for (int i = 0; i < n; ++i) v[i] = v[i] + xi[i];
  for (int i = n - 1; i >= 0; --i) {
    double s = v[i];
    for (int k = i + 1; k < n; ++k) s -= L[k + (size_t)n * i] * y[k];
    y[i] = s / L[i + (size_t)n * i];
  }
  free(v);
}

/* ------------------------------------------------------------------------------------------
 * Plain CSR products used by the V-cycle restatement (PETSc MatMult / MatMultTranspose)
 * ---------------------------------------------------------------------------------------- */
void orc_spmv(int nrows, const int32_t *rowptr, const int32_t *colidx, const double *vals, const double *x, double *y)
{
  for (int r = 0; r < nrows; ++r) {
    double s = 0;
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) s += vals[k] * x[colidx[k]];
    y[r] = s;
  }
}

/* ------------------------------------------------------------------------------------------
 * Sampled-row restatements for parity checks at sizes where a whole sweep takes the CPU too long
 * (256^3 ... 513^3): the SAME row arithmetic as above, for a list of rows only.
 *
 * A multicolour sweep visits colour after colour, so when row r of colour c is updated its
 * neighbours of a lower colour already hold their new values and those of a higher colour
 * still hold the old ones (forward order; reversed for a backward sweep).  Given the vector
 * before (y0) and after (y1) the whole sweep, the value the reference loop computes for row r
 * is therefore a function of y0, y1 and the colour rule alone.
 * ---------------------------------------------------------------------------------------- */

/* rows of the 7-point grid operator, red-black colouring c = (i+j+k)&1 (colour 0 first when forward).
   Row update per reference src/mc_sor.c:261-267 (orc_row_update), right-hand side per
   src/pc_mcgibbs.c:119-128 (orc_prepare_rhs) with the grid noise stream when noisy. */
void orc_grid7_rows_sweep(int nx, int ny, int nz, double kappa, double omega, int backward, int noisy, int scaled, uint64_t seed, uint64_t sweep, int64_t nrows, const int64_t *rows, const double *b, const double *y0, const double *y1, double *out)
{
  const double   hinv2  = 1. / ((nx - 1) * (nx - 1));
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  const double   sc     = sqrt((2 - omega) / omega);
  for (int64_t q = 0; q < nrows; ++q) {
    const int64_t r = rows[q];
    const int     i = (int)(r % nx), j = (int)((r / nx) % ny), k = (int)(r / ((int64_t)nx * ny));
    int64_t       c[7];
    double        v[7];
    int           dpos;
    const int     n    = orc_laplace_row(nx, ny, nz, kappa, hinv2, i, j, k, c, v, &dpos);
    const int     mine = (i + j + k) & 1;
    /* neighbours have the other colour: new values iff that colour was swept before mine */
    const int     nb_new = backward ? (mine == 0) : (mine == 1);
    const double *yn     = nb_new ? y1 : y0;
    const double  t      = 1.0 / v[dpos];
    const double  idiag  = t * omega; /* orc_idiag */
    double        w      = b[r];
    if (noisy) {
      const double sd0 = sqrt(fabs(v[dpos]));
      const double sd  = scaled ? sd0 * sc : sd0; /* orc_sqrtdiag */
      double       xi  = orc_noise_grid_point(ny, i, j, k, key, sweep);
      xi               = xi * sd; /* orc_prepare_rhs: three passes */
      w                = xi + b[r];
    }
    double sum = w;
    for (int e = 0; e < dpos; ++e) sum -= v[e] * yn[c[e]];
    for (int e = dpos + 1; e < n; ++e) sum -= v[e] * yn[c[e]];
    out[q] = (1. - omega) * y0[r] + idiag * sum;
  }
}

/* r = b - A y for sampled rows of the grid operator: MatMult row sum in storage order including the
   diagonal at its place, then VecAYPX(w,-1,b) (reference src/pc_gamgmc.c:253-254; orc_spmv) */
void orc_grid7_rows_residual(int nx, int ny, int nz, double kappa, int64_t nrows, const int64_t *rows, const double *b, const double *y, double *out)
{
  const double hinv2 = 1. / ((nx - 1) * (nx - 1));
  for (int64_t q = 0; q < nrows; ++q) {
    const int64_t r = rows[q];
    const int     i = (int)(r % nx), j = (int)((r / nx) % ny), k = (int)(r / ((int64_t)nx * ny));
    int64_t       c[7];
    double        v[7];
    int           dpos;
    const int     n = orc_laplace_row(nx, ny, nz, kappa, hinv2, i, j, k, c, v, &dpos);
    double        s = 0.0;
    for (int e = 0; e < n; ++e) s += v[e] * y[c[e]];
    out[q] = b[r] - s;
  }
}

/* Rows of a 27-point class-stencil operator (coef[27*cls + e], e = 9(dz+1)+3(dy+1)+(dx+1); the assembled row holds
   the in-domain entries in ascending column order), 8 parity colours (i&1) + 2(j&1) + 4(k&1) swept ascending
   (descending when backward), noise = row stream of the natural index (orc_noise_rows).  which = 0: sweep value as
   in orc_grid7_rows_sweep (sqrtd[cls] is the noise scale); which = 1: residual b - A y0 with the diagonal added
   LAST (the order of the library's sliced-ELL residual, which keeps the diagonal apart). */
void orc_st27_rows(int which, int nx, int ny, int nz, const double *coef, const double *sqrtd, double omega, int backward, int noisy, uint64_t seed, uint64_t sweep, int64_t nrows, const int64_t *rows, const double *b, const double *y0, const double *y1, double *out)
{
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  for (int64_t q = 0; q < nrows; ++q) {
    const int64_t r = rows[q];
    const int     i = (int)(r % nx), j = (int)((r / nx) % ny), k = (int)(r / ((int64_t)nx * ny));
    const int     cls = (i == 0 ? 0 : (i == nx - 1 ? 2 : 1)) + 3 * (j == 0 ? 0 : (j == ny - 1 ? 2 : 1)) + 9 * (k == 0 ? 0 : (k == nz - 1 ? 2 : 1));
    const double *cf   = coef + 27 * cls;
    const int     mine = (i & 1) + 2 * (j & 1) + 4 * (k & 1);
    const double  d    = cf[13];
    if (which == 1) {
      double s = 0.0;
      int    e = 0;
      for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx, ++e) {
            if (e == 13 || i + dx < 0 || i + dx >= nx || j + dy < 0 || j + dy >= ny || k + dz < 0 || k + dz >= nz) continue;
            s += cf[e] * y0[r + dx + (int64_t)nx * (dy + (int64_t)ny * dz)];
          }
      s += d * y0[r];
      out[q] = b[r] - s;
      continue;
    }
    double w = b[r];
    if (noisy) {
      const uint64_t pr     = (uint64_t)r >> 1;
      const uint32_t ctr[4] = {(uint32_t)pr, (uint32_t)(pr >> 32), (uint32_t)sweep, (uint32_t)(sweep >> 32)};
      double         z[2];
      orc_normal_pair(ctr, key, z);
      double xi = z[r & 1];
      xi        = xi * sqrtd[cls];
      w         = xi + b[r];
    }
    double sum = w;
    int    e   = 0;
    for (int dz = -1; dz <= 1; ++dz)
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx, ++e) {
          if (e == 13 || i + dx < 0 || i + dx >= nx || j + dy < 0 || j + dy >= ny || k + dz < 0 || k + dz >= nz) continue;
          const int     other = ((i + dx) & 1) + 2 * ((j + dy) & 1) + 4 * ((k + dz) & 1);
          const int     isnew = backward ? other > mine : other < mine;
          const int64_t cidx  = r + dx + (int64_t)nx * (dy + (int64_t)ny * dz);
          sum -= cf[e] * (isnew ? y1[cidx] : y0[cidx]);
        }
    const double t = 1.0 / d;
    out[q]         = (1. - omega) * y0[r] + (t * omega) * sum;
  }
}

/* Q1 transfers of a DMDA hierarchy for sampled rows (PETSc DMCreateInterpolation semantics as restated by
   q1_interp in oracle/__init__.py: fine 2I = coarse I, fine 2I+1 midway with weights 1/2, 1/2, tensor product).
   restrict: out = (P^T r)[I] -- a CSR row of P^T, fine columns ascending, summed from 0;
   prolong:  out = x[f] + (P e)[f] -- a CSR row of P, coarse columns ascending, summed from 0, then MatInterpolateAdd. */
void orc_q1_rows(int which, const int32_t nf[3], const int32_t nc[3], int64_t nrows, const int64_t *rows, const double *fine, const double *coarse, double *out)
{
  for (int64_t q = 0; q < nrows; ++q) {
    const int64_t r = rows[q];
    if (which == 0) { /* coarse row I of P^T */
      const int I[3] = {(int)(r % nc[0]), (int)((r / nc[0]) % nc[1]), (int)(r / ((int64_t)nc[0] * nc[1]))};
      int       lo[3], hi[3];
      for (int d = 0; d < 3; ++d) {
        const int ref = nf[d] != nc[d];
        const int f   = ref ? 2 * I[d] : I[d];
        lo[d]         = ref ? (f - 1 < 0 ? 0 : f - 1) : f;
        hi[d]         = ref ? (f + 1 > nf[d] - 1 ? nf[d] - 1 : f + 1) : f;
      }
      double s = 0.0;
      for (int k = lo[2]; k <= hi[2]; ++k)
        for (int j = lo[1]; j <= hi[1]; ++j)
          for (int i = lo[0]; i <= hi[0]; ++i) {
            const int    f[3] = {i, j, k};
            double       w    = 1.0;
            for (int d = 0; d < 3; ++d)
              if (nf[d] != nc[d] && (f[d] & 1)) w *= 0.5;
            s += w * fine[i + (int64_t)nf[0] * (j + (int64_t)nf[1] * k)];
          }
      out[q] = s;
    } else { /* fine row f of P */
      const int f[3] = {(int)(r % nf[0]), (int)((r / nf[0]) % nf[1]), (int)(r / ((int64_t)nf[0] * nf[1]))};
      int       c0[3], m[3];
      double    w = 1.0;
      for (int d = 0; d < 3; ++d) {
        const int ref = nf[d] != nc[d];
        c0[d]         = ref ? f[d] / 2 : f[d];
        m[d]          = (ref && (f[d] & 1)) ? 2 : 1;
        if (m[d] == 2) w *= 0.5;
      }
      double s = 0.0;
      for (int c = 0; c < m[2]; ++c)
        for (int bq = 0; bq < m[1]; ++bq)
          for (int a = 0; a < m[0]; ++a) s += w * coarse[(c0[0] + a) + (int64_t)nc[0] * ((c0[1] + bq) + (int64_t)nc[1] * (c0[2] + c))];
      out[q] = fine[r] + s;
    }
  }
}
