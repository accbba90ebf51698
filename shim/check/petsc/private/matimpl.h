/* stand-in for the check of the shim files only: see ../../petsc_decls.h */
#include "../../petsc_decls.h"
