/*
 * petsc_decl_mock.h -- NOT PETSc.  Declarations only (no definitions, nothing can link or run) of the subset of the
 * PETSc C API that the files under adapter/ calls, transcribed from PETSc's manual pages, plus the five public declarations of
 * ParMGMC's own header the adapter needs.  Its single purpose: tests/test_adapter_syntax.py runs
 * `gcc -fsyntax-only -DPARMGMC_HIP_HAVE_PETSC` over the adapter so that typos, missing arguments and type errors in
 * the adapter are caught in an image that has no PETSc.  A clean syntax check says nothing about PETSc's behaviour;
 * the adapter remains untested against a real PETSc.
 */
#ifndef PETSC_DECL_MOCK_H
#define PETSC_DECL_MOCK_H
#include <math.h>
#include <stddef.h>
#include <stdint.h>

typedef int PetscErrorCode;
#ifdef PETSC_DECL_MOCK_64BIT_INDICES
typedef int64_t PetscInt;
#define PetscInt_FMT "lld"
#else
typedef int PetscInt;
#define PetscInt_FMT "d"
#endif
typedef int64_t PetscInt64;
typedef int     PetscMPIInt;
typedef double  PetscScalar;
typedef double  PetscReal;
typedef enum { PETSC_FALSE, PETSC_TRUE } PetscBool;
typedef int PetscLogEvent;
typedef int PetscClassId;
typedef int MPI_Comm;
typedef int MPI_Datatype;
typedef int MPI_Op;
#define MPI_BYTE 1
#define MPI_INT 2
#define MPI_MIN 3
#define MPI_DOUBLE 4
#define MPI_SUM 5
#define MPI_SUCCESS 0
#define MPI_COMM_WORLD 2
/* the mock stands for a current PETSc (the reference's CI pins 3.25.1, .github/workflows/linux.yml:38) */
#define PETSC_VERSION_LT(a, b, c) 0
#define PETSC_VERSION_LE(a, b, c) 0
#define PETSC_VERSION_GT(a, b, c) 1
#define PETSC_VERSION_GE(a, b, c) 1
typedef enum { DMDA_STENCIL_STAR, DMDA_STENCIL_BOX } DMDAStencilType;
typedef enum { DM_BOUNDARY_NONE, DM_BOUNDARY_GHOSTED, DM_BOUNDARY_MIRROR, DM_BOUNDARY_PERIODIC } DMBoundaryType;
#define PetscAbsReal(x) ((x) < 0 ? -(x) : (x))
#define PetscSqrtReal(x) sqrt(x)
typedef enum { PETSC_MEMTYPE_HOST = 0, PETSC_MEMTYPE_DEVICE = 1 } PetscMemType;
#define PetscMemTypeHost(m) (((m) & 0x1) == PETSC_MEMTYPE_HOST)
#define PetscMemTypeDevice(m) (((m) & 0x1) == PETSC_MEMTYPE_DEVICE)
#define PETSC_SUCCESS 0
#define PETSC_ERR_SUP 56
#define PETSC_ERR_ORDER 58
#define PETSC_ERR_ARG_SIZ 60
#define PETSC_ERR_MAT_LU_ZRPVT 71
#define PETSC_ERR_PLIB 77
#define PETSC_DECIDE (-1)
#define PetscAbs(a) (((a) >= 0) ? (a) : (-(a)))
#define PETSC_ERR_GPU 97
#define PETSC_COMM_SELF 1
#define PETSC_EXTERN extern
typedef struct _p_PetscObject *PetscObject;
typedef struct _p_Vec         *Vec;
typedef struct _p_Mat         *Mat;
typedef struct _p_KSP         *KSP;
typedef struct _p_DM          *DM;
typedef struct _p_PetscViewer *PetscViewer;
typedef struct _p_PetscRandom *PetscRandom;
typedef struct _n_PetscOptions *PetscOptions;
typedef struct _p_PetscOptionItems *PetscOptionItems;
typedef struct _p_PC *PC;
typedef struct _p_PetscDS     *PetscDS;
typedef struct _n_ISColoring  *ISColoring;
typedef unsigned short         ISColoringValue;
typedef enum { IS_COLORING_GLOBAL, IS_COLORING_LOCAL } ISColoringType;
typedef enum { PETSC_COPY_VALUES, PETSC_OWN_POINTER, PETSC_USE_POINTER } PetscCopyMode;
typedef enum { MAT_DO_NOT_COPY_VALUES, MAT_COPY_VALUES, MAT_SHARE_NONZERO_PATTERN } MatDuplicateOption;
typedef enum { NORM_1 = 0, NORM_2 = 1, NORM_FROBENIUS = 2, NORM_INFINITY = 3 } NormType;
typedef enum { FILE_MODE_UNDEFINED = -1, FILE_MODE_READ = 0, FILE_MODE_WRITE } PetscFileMode;
typedef enum { KSP_DMACTIVE_OPERATOR = 1, KSP_DMACTIVE_RHS = 2, KSP_DMACTIVE_INITIAL_GUESS = 4 } KSPDMActive;
#define PCSHELL "shell"
typedef const char *MatType;
typedef const char *PCType;
typedef enum { SOR_FORWARD_SWEEP = 1, SOR_BACKWARD_SWEEP = 2, SOR_SYMMETRIC_SWEEP = 3, SOR_LOCAL_FORWARD_SWEEP = 4 } MatSORType;
typedef enum { PCRICHARDSON_NOT_SET = 0, PCRICHARDSON_CONVERGED_RTOL = 2, PCRICHARDSON_CONVERGED_ATOL = 3, PCRICHARDSON_CONVERGED_ITS = 4 } PCRichardsonConvergedReason;
#define MATSEQAIJ "seqaij"
#define MATMPIAIJ "mpiaij"
#define DMDA "da"
#define MATLRC "lrc"
#define PCMG "mg"
#define PCGAMG "gamg"
#define KSPRICHARDSON "richardson"

/* pcimpl.h: the part of the private PC struct the constructors fill */
struct _PCOps {
  PetscErrorCode (*setup)(PC);
  PetscErrorCode (*apply)(PC, Vec, Vec);
  PetscErrorCode (*applyrichardson)(PC, Vec, Vec, Vec, PetscReal, PetscReal, PetscReal, PetscInt, PetscBool, PetscInt *, PCRichardsonConvergedReason *);
  PetscErrorCode (*setfromoptions)(PC, PetscOptionItems);
  PetscErrorCode (*reset)(PC);
  PetscErrorCode (*destroy)(PC);
  PetscErrorCode (*view)(PC, PetscViewer);
};
struct _p_PC {
  struct _PCOps *ops;
  DM             dm;
  Mat            mat, pmat;
  void          *data;
};

/* error handling / bookkeeping macros */
PetscErrorCode PetscError(MPI_Comm, int, const char *, const char *, PetscErrorCode, int, const char *, ...);
#define PetscFunctionBeginUser do { } while (0)
#define PetscFunctionBegin do { } while (0)
#define PetscFunctionReturn(x) return (x)
#define PetscCall(...) do { PetscErrorCode ierr_q_ = (__VA_ARGS__); if (ierr_q_) return ierr_q_; } while (0)
#define PetscCallMPI(...) do { int ierr_m_ = (__VA_ARGS__); if (ierr_m_) return 98; } while (0)
#define PetscCheck(cond, comm, ierr, ...) do { if (!(cond)) return PetscError(comm, __LINE__, __func__, __FILE__, ierr, 0, __VA_ARGS__); } while (0)
#define PetscNew(p) PetscMallocA_mock(sizeof(**(p)), (void **)(p))
#define PetscMalloc1(n, p) PetscMallocA_mock((size_t)(n) * sizeof(**(p)), (void **)(p))
#define PetscCalloc1(n, p) PetscMallocA_mock((size_t)(n) * sizeof(**(p)), (void **)(p))
#define PetscFree(p) (PetscFree_mock((void *)(p)), (p) = NULL, PETSC_SUCCESS)
#define PetscArraycpy(d, s, n) PetscMemcpy((d), (s), (size_t)(n) * sizeof(*(d)))
PetscErrorCode PetscMallocA_mock(size_t, void **);
PetscErrorCode PetscFree_mock(void *);
PetscErrorCode PetscMemcpy(void *, const void *, size_t);
PetscErrorCode PetscStrncpy(char[], const char[], size_t);
PetscErrorCode PetscSNPrintf(char *, size_t, const char[], ...);
int            MPI_Comm_size(MPI_Comm, int *);
int            MPI_Comm_rank(MPI_Comm, int *);
int            MPI_Allgather(const void *, int, MPI_Datatype, void *, int, MPI_Datatype, MPI_Comm);
int            MPI_Allreduce(const void *, void *, int, MPI_Datatype, MPI_Op, MPI_Comm);
PetscErrorCode PetscObjectReference(PetscObject);
MPI_Comm       PetscObjectComm(PetscObject);
PetscErrorCode PetscObjectTypeCompare(PetscObject, const char[], PetscBool *);
PetscErrorCode PetscObjectComposeFunction_Private(PetscObject, const char[], void (*)(void));
#define PetscObjectComposeFunction(o, n, f) PetscObjectComposeFunction_Private((o), (n), (void (*)(void))(f))
PetscErrorCode PetscLogEventBegin(PetscLogEvent, void *, void *, void *, void *);
PetscErrorCode PetscLogEventEnd(PetscLogEvent, void *, void *, void *, void *);
/* options */
#define PetscOptionsHeadBegin(obj, head) do { (void)(obj); } while (0)
#define PetscOptionsHeadEnd() do { } while (0)
PetscErrorCode PetscOptionsBool(const char[], const char[], const char[], PetscBool, PetscBool *, PetscBool *);
PetscErrorCode PetscOptionsRangeReal(const char[], const char[], const char[], PetscReal, PetscReal *, PetscBool *, PetscReal, PetscReal);
PetscErrorCode PetscOptionsString(const char[], const char[], const char[], const char[], char[], size_t, PetscBool *);
PetscErrorCode PetscOptionsReal(const char[], const char[], const char[], PetscReal, PetscReal *, PetscBool *);
PetscErrorCode PetscOptionsInt(const char[], const char[], const char[], PetscInt, PetscInt *, PetscBool *);
PetscErrorCode PetscOptionsHasName(PetscOptions, const char[], const char[], PetscBool *);
PetscErrorCode PetscOptionsSetValue(PetscOptions, const char[], const char[]);
PetscErrorCode PetscOptionsGetString(PetscOptions, const char[], const char[], char[], size_t, PetscBool *);
PetscErrorCode PetscOptionsGetInt(PetscOptions, const char[], const char[], PetscInt *, PetscBool *);
PetscErrorCode PetscOptionsGetReal(PetscOptions, const char[], const char[], PetscReal *, PetscBool *);
PetscErrorCode PetscOptionsGetBool(PetscOptions, const char[], const char[], PetscBool *, PetscBool *);
PetscErrorCode PetscViewerASCIIPrintf(PetscViewer, const char[], ...);
PetscErrorCode PetscRandomGetSeed(PetscRandom, PetscInt64 *);
PetscErrorCode PetscRandomDestroy(PetscRandom *);
/* what the reference's examples ex3.c / ex5.c call besides the adapter's own needs (tests/test_adapter_syntax.py compiles them
   UNCHANGED from /root/reference against these declarations and the MCSOR binding of adapter/mc_sor_hip.c) */
PetscErrorCode PetscInitialize(int *, char ***, const char[], const char[]);
PetscErrorCode PetscFinalize(void);
PetscErrorCode PetscObjectSetName(PetscObject, const char[]);
PetscErrorCode PetscViewerVTKOpen(MPI_Comm, const char[], PetscFileMode, PetscViewer *);
PetscErrorCode PetscViewerDestroy(PetscViewer *);
PetscErrorCode DMDACreate2d(MPI_Comm, DMBoundaryType, DMBoundaryType, DMDAStencilType, PetscInt, PetscInt, PetscInt, PetscInt, PetscInt, PetscInt, const PetscInt[], const PetscInt[], DM *);
PetscErrorCode DMSetFromOptions(DM);
PetscErrorCode DMSetUp(DM);
PetscErrorCode DMDestroy(DM *);
PetscErrorCode DMDASetUniformCoordinates(DM, PetscReal, PetscReal, PetscReal, PetscReal, PetscReal, PetscReal);
PetscErrorCode DMCreateMatrix(DM, Mat *);
PetscErrorCode DMCreateGlobalVector(DM, Vec *);
PetscErrorCode KSPCreate(MPI_Comm, KSP *);
PetscErrorCode KSPDestroy(KSP *);
PetscErrorCode KSPSetOperators(KSP, Mat, Mat);
PetscErrorCode KSPSetDM(KSP, DM);
PetscErrorCode KSPSetDMActive(KSP, KSPDMActive, PetscBool);
PetscErrorCode KSPSetFromOptions(KSP);
PetscErrorCode KSPSetUp(KSP);
PetscErrorCode KSPSetInitialGuessNonzero(KSP, PetscBool);
PetscErrorCode KSPSolve(KSP, Vec, Vec);
PetscErrorCode PCShellSetApply(PC, PetscErrorCode (*)(PC, Vec, Vec));
PetscErrorCode PCShellSetContext(PC, void *);
PetscErrorCode PCShellGetContext(PC, void *);
PetscErrorCode MatCreateLRC(Mat, Mat, Vec, Mat, Mat *);
PetscErrorCode MatMult(Mat, Vec, Vec);
PetscErrorCode MatDestroy(Mat *);
PetscErrorCode MatDuplicate(Mat, MatDuplicateOption, Mat *);
PetscErrorCode MatDenseGetColumnVecWrite(Mat, PetscInt, Vec *);
PetscErrorCode MatDenseRestoreColumnVecWrite(Mat, PetscInt, Vec *);
PetscErrorCode MatDenseGetArrayWrite(Mat, PetscScalar **);
PetscErrorCode MatDenseRestoreArrayWrite(Mat, PetscScalar **);
PetscErrorCode ISColoringCreate(MPI_Comm, PetscInt, PetscInt, const ISColoringValue[], PetscCopyMode, ISColoring *);
PetscErrorCode ISColoringSetType(ISColoring, ISColoringType);
PetscErrorCode ISColoringDestroy(ISColoring *);
PetscErrorCode VecSetRandom(Vec, PetscRandom);
PetscErrorCode VecDuplicate(Vec, Vec *);
PetscErrorCode VecCopy(Vec, Vec);
PetscErrorCode VecAXPY(Vec, PetscScalar, Vec);
PetscErrorCode VecNorm(Vec, NormType, PetscReal *);
PetscErrorCode VecView(Vec, PetscViewer);
PetscErrorCode VecGetOwnershipRange(Vec, PetscInt *, PetscInt *);
/* Vec */
PetscErrorCode VecGetLocalSize(Vec, PetscInt *);
PetscErrorCode VecZeroEntries(Vec);
PetscErrorCode VecDestroy(Vec *);
PetscErrorCode VecGetArrayRead(Vec, const PetscScalar **);
PetscErrorCode VecRestoreArrayRead(Vec, const PetscScalar **);
PetscErrorCode VecGetArrayAndMemType(Vec, PetscScalar **, PetscMemType *);
PetscErrorCode VecRestoreArrayAndMemType(Vec, PetscScalar **);
PetscErrorCode VecGetArrayReadAndMemType(Vec, const PetscScalar **, PetscMemType *);
PetscErrorCode VecRestoreArrayReadAndMemType(Vec, const PetscScalar **);
/* Mat */
PetscErrorCode MatGetSize(Mat, PetscInt *, PetscInt *);
PetscErrorCode MatSeqAIJGetCSRAndMemType(Mat, const PetscInt **, const PetscInt **, PetscScalar **, PetscMemType *);
PetscErrorCode MatLRCGetMats(Mat, Mat *, Mat *, Vec *, Mat *);
PetscErrorCode MatDenseGetLDA(Mat, PetscInt *);
PetscErrorCode MatDenseGetArrayRead(Mat, const PetscScalar **);
PetscErrorCode MatDenseRestoreArrayRead(Mat, const PetscScalar **);
PetscErrorCode MatDenseGetColumnVecRead(Mat, PetscInt, Vec *);
PetscErrorCode MatDenseRestoreColumnVecRead(Mat, PetscInt, Vec *);
PetscErrorCode MatCreateVecs(Mat, Vec *, Vec *);
PetscErrorCode MatGetLocalSize(Mat, PetscInt *, PetscInt *);
PetscErrorCode MatGetOwnershipRange(Mat, PetscInt *, PetscInt *);
PetscErrorCode MatGetOwnershipRangeColumn(Mat, PetscInt *, PetscInt *);
PetscErrorCode MatGetOwnershipRanges(Mat, const PetscInt **);
PetscErrorCode MatMPIAIJGetSeqAIJ(Mat, Mat *, Mat *, const PetscInt *[]);
PetscErrorCode MatGetRow(Mat, PetscInt, PetscInt *, const PetscInt *[], const PetscScalar *[]);
PetscErrorCode MatRestoreRow(Mat, PetscInt, PetscInt *, const PetscInt *[], const PetscScalar *[]);
/* DMDA */
typedef enum { DMDA_Q0, DMDA_Q1 } DMDAInterpolationType;
typedef enum { PC_MG_GALERKIN_BOTH, PC_MG_GALERKIN_PMAT, PC_MG_GALERKIN_MAT, PC_MG_GALERKIN_NONE, PC_MG_GALERKIN_EXTERNAL } PCMGGalerkinType;
PetscErrorCode DMDAGetInterpolationType(DM, DMDAInterpolationType *);
PetscErrorCode PCMGGetGalerkin(PC, PCMGGalerkinType *);
PetscErrorCode DMDAGetInfo(DM, PetscInt *, PetscInt *, PetscInt *, PetscInt *, PetscInt *, PetscInt *, PetscInt *, PetscInt *, PetscInt *, DMBoundaryType *, DMBoundaryType *, DMBoundaryType *, DMDAStencilType *);
PetscErrorCode DMDAGetCorners(DM, PetscInt *, PetscInt *, PetscInt *, PetscInt *, PetscInt *, PetscInt *);
/* PC / KSP / PCMG */
PetscErrorCode PCCreate(MPI_Comm, PC *);
PetscErrorCode PCDestroy(PC *);
PetscErrorCode PCReset(PC);
PetscErrorCode PCSetType(PC, PCType);
PetscErrorCode PCSetUp(PC);
PetscErrorCode PCApply(PC, Vec, Vec);
PetscErrorCode PCApplyRichardson(PC, Vec, Vec, Vec, PetscReal, PetscReal, PetscReal, PetscInt, PetscBool, PetscInt *, PCRichardsonConvergedReason *);
PetscErrorCode PCSetFromOptions(PC);
PetscErrorCode PCView(PC, PetscViewer);
PetscErrorCode PCSetDM(PC, DM);
PetscErrorCode PCSetOperators(PC, Mat, Mat);
PetscErrorCode PCGetOperators(PC, Mat *, Mat *);
PetscErrorCode PCGetOptionsPrefix(PC, const char *[]);
PetscErrorCode PCSetOptionsPrefix(PC, const char[]);
PetscErrorCode PCAppendOptionsPrefix(PC, const char[]);
PetscErrorCode PCRegister(const char[], PetscErrorCode (*)(PC));
PetscErrorCode PCMGGetLevels(PC, PetscInt *);
PetscErrorCode PCMGSetLevels(PC, PetscInt, MPI_Comm *);
PetscErrorCode PCMGGetSmoother(PC, PetscInt, KSP *);
PetscErrorCode PCMGGetInterpolation(PC, PetscInt, Mat *);
PetscErrorCode KSPGetPC(KSP, PC *);
#endif
