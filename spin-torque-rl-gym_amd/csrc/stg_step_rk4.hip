// stg_step_rk4.hip -- instantiations of the env-step kernel for STG_SOLVER_RK4 (see stg_kernels.hpp)
#include "stg_kernels.hpp"

void stg_dispatch_step_rk4(const StepArgs& a, bool thermal, int multi, bool axis_z, bool devphys, int act_f64, bool pc, hipStream_t st) {
    dispatch_step<STG_SOLVER_RK4>(a, thermal, multi, axis_z, devphys, act_f64, pc, st);
}
