/* NOT a PETSc header: see petsc_decl_mock.h (declarations only, for tests/test_adapter_syntax.py) */
#include "petsc_decl_mock.h"
