/* NOT ParMGMC's header: the declarations of its MCSOR interface (reference include/parmgmc/mc_sor.h:17-30) that
   adapter/mc_sor_hip.c implements, for the syntax check only -- see ../petsc_decl_mock.h.  Inside a ParMGMC tree the
   reference's own header is the one that is included. */
#include "../petsc_decl_mock.h"
typedef struct _MCSOR {
  void *ctx;
} *MCSOR;
PETSC_EXTERN PetscErrorCode MCSORCreate(Mat, MCSOR *);
PETSC_EXTERN PetscErrorCode MCSORSetUp(MCSOR);
PETSC_EXTERN PetscErrorCode MCSORDestroy(MCSOR *);
PETSC_EXTERN PetscErrorCode MCSORApply(MCSOR, Vec, Vec);
PETSC_EXTERN PetscErrorCode MCSORSetOmega(MCSOR, PetscReal);
PETSC_EXTERN PetscErrorCode MCSORSetSweepType(MCSOR, MatSORType);
PETSC_EXTERN PetscErrorCode MCSORGetSweepType(MCSOR, MatSORType *);
PETSC_EXTERN PetscErrorCode MCSORGetISColoring(MCSOR, ISColoring *);
PETSC_EXTERN PetscErrorCode MCSORGetNumColors(MCSOR, PetscInt *);
PETSC_EXTERN PetscErrorCode MCSORBuildLRCCorrection(PetscErrorCode (*det_sor)(void *, Vec, Vec), void *, Mat, Mat, Vec, Mat *);
