/* NOT ParMGMC's header: the declarations of its public interface that the files under adapter/ uses (reference
   include/parmgmc/parmgmc.h:18-47), for the syntax check only -- see ../petsc_decl_mock.h */
#include "../petsc_decl_mock.h"
#define PetscOptionItems_ARG PetscOptionItems
#define PCMCGIBBS "mcgibbs"
#define PCGAMGMC "gamgmc"
#define PCSORGIBBS "sorgibbs"
#define PCCHOLSAMPLER "cholsampler"
#define PCPARSOR "parsor"
#define PCWOODBURY "woodbury"
PETSC_EXTERN PetscLogEvent  MULTICOL_SOR;
PETSC_EXTERN PetscErrorCode PCRegisterSetSampleCallback(PC, PetscErrorCode (*)(PC, PetscErrorCode (*)(PetscInt, Vec, void *), void *, PetscErrorCode (*)(void *)));
PETSC_EXTERN PetscErrorCode ParMGMCGetPetscRandom(PetscRandom *);
