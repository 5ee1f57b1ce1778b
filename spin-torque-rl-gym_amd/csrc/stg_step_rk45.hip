// stg_step_rk45.hip -- instantiations of the env-step kernel for STG_SOLVER_RK45 (see stg_kernels.hpp)
#include "stg_kernels.hpp"

void stg_dispatch_step_rk45(const StepArgs& a, bool thermal, int multi, bool axis_z, int act_f64, bool pc, hipStream_t st) {
    dispatch_step<STG_SOLVER_RK45>(a, thermal, multi, axis_z, false, act_f64, pc, st);
}

void stg_dispatch_step_rk45_refill(const StepArgs& a, bool thermal, bool multi, bool axis_z, int act_f64, hipStream_t st) {
    dispatch_refill(a, thermal, multi, axis_z, act_f64, st);
}

#ifdef STG_PROFILE_LOOP
extern "C" int stg_debug_waves(long long* out, int n_waves) {   // n_waves x 6 records, see g_stg_wave
    if (n_waves > STG_PROF_WAVES) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(stg::g_stg_wave), (size_t)n_waves * 6 * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif
