// One physics kernel per translation unit: compiled once per kernel with -DJACO_TU=<n> (physics_kernel.h, "translation units"), so that
// the eight long device compiles of a library build run in parallel.  Each unit defines its kernel and that kernel's launcher.
#include <hip/hip_runtime.h>
#ifndef JACO_TU
#error "kernels.hip is compiled with -DJACO_TU=<kernel index>"
#endif
#include "physics_kernel.h"
