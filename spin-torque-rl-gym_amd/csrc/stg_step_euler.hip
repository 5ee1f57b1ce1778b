// stg_step_euler.hip -- instantiations of the env-step kernel for STG_SOLVER_EULER (see stg_kernels.hpp)
#include "stg_kernels.hpp"

void stg_dispatch_step_euler(const StepArgs& a, bool thermal, int multi, bool axis_z, bool devphys, int act_f64, bool pc, hipStream_t st) {
    dispatch_step<STG_SOLVER_EULER>(a, thermal, multi, axis_z, devphys, act_f64, pc, st);
}
