#include "petsc_decl_mock.h"
