// ea_kernels_var.hip -- second translation unit of the fused evaluation: the residual variants (EAResidueEx, EAResidueSecondCam,
// EAResidueSecondCamEx; standalone/utils.h:102-421) of ea_kernels.hip's kernel template, compiled with the compiler's default
// machine-scheduling strategy (the plain functor's instantiations are built with max-ilp; see launch_eval_fused there
// and build.py).  Nothing else is instantiated here.
#define EA_TU_VARIANT 1
#undef EA_STAMPS  // (the diagnostic stamps belong to the plain kernels' unit)
#include "ea_kernels.hip"
