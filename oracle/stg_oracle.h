/*
 * stg_oracle.h -- CPU restatement (plain C, IEEE fp64) of the SpinTorque-v0 step path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / reported baseline.  The product path (spin-torque-rl-gym_amd/) never calls it.
 *
 * Parity status: PINNED by golden vectors generated in the build container by importing the
 * unmodified Python reference (tests/golden/make_golden.py writes the .npz files under tests/golden/); see
 * tests/test_oracle_golden.py.  The reference's own tests hold no numeric vectors for this path
 * (SURVEY.md section 4), and the Dormand-Prince integrator is SciPy's (scipy 1.15.3,
 * scipy/integrate/_ivp/{rk,common,base}.py), restated here from its published algorithm.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/spin_torque_gym/).  Operation order follows the NumPy expressions literally;
 * build with -ffp-contract=off so no FMA is formed that NumPy would not form.
 */
#ifndef STG_ORACLE_H
#define STG_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* device_params dict of the reference, flattened (devices/device_factory.py:118-172,
 * envs/spin_torque_env.py:156-182).  Raw values exactly as the dict holds them. */
typedef struct {
    double damping;                 /* 'damping' */
    double ms;                      /* 'saturation_magnetization' */
    double ku;                      /* 'uniaxial_anisotropy' */
    double volume;                  /* 'volume' */
    double polarization;            /* 'polarization' */
    double easy_axis[3];            /* 'easy_axis' (raw, not normalised) */
    double demag[3];                /* 'demag_factors' (LLGSSolver only, default 0,0,1) */
    double a_ex;                    /* 'exchange_constant' (LLGSSolver only, default 20e-12) */
    double area;                    /* 'area' (env energy, default 1e-14) */
    double r_p, r_ap;               /* 'resistance_parallel' / 'resistance_antiparallel' */
    double ref_m[3];                /* 'reference_magnetization' (raw) */
    double r_series;                /* SOT only: 0.1 * (rho_hm/t_hm) / (area*1e-12)  (devices/sot_mram.py:218-223) */
    double sot_tau_dl, sot_tau_fl;  /* SOT: tau_dl_factor, tau_fl_factor (devices/sot_mram.py:61-72) */
    double sot_sigma[3];            /* SOT: z x current_direction (devices/sot_mram.py:180-186) */
    double vcma_xi, vcma_td, vcma_vbd; /* VCMA: vcma_coefficient, dielectric_thickness, breakdown_voltage */
    double shape_demag[3];          /* SOT/VCMA compute_effective_field: demag factors from 'aspect_ratio' (devices/sot_mram.py:114-132) */
    int32_t dev_type;               /* 0 stt_mram, 1 sot_mram, 2 vcma_mram */
    int32_t params_valid;           /* result of utils/validation.py:176-234 as 'stt_mram' (host-evaluated) */
} stgo_params;

/* env + solver configuration (envs/spin_torque_env.py:36-53, physics/llgs_solver.py:24-31,
 * physics/simple_solver.py:24-31) */
typedef struct {
    int32_t solver;                 /* 0 = SimpleLLGSSolver rk4, 1 = SimpleLLGSSolver euler, 2 = LLGSSolver RK45 */
    int32_t thermal;                /* include_thermal_fluctuations */
    double temperature;             /* K */
    double gamma;                   /* 2.21e5 */
    double max_step;                /* 1e-12 */
    double rtol, atol;              /* RK45 only: 1e-6 / 1e-9 */
    int32_t max_steps;              /* 100 */
    double max_current;             /* 2e6 */
    double max_duration;            /* 5e-9 */
    double success_threshold;       /* 0.9 */
    double energy_penalty_weight;   /* 0.1 */
    uint64_t seed;                  /* thermal-field Philox key */
    int64_t max_attempts;           /* RK45 attempt budget per solve (guard; reference has none) */
    int32_t torque_model;           /* 0 reference RHS; 1 device-physics torque model (include/spintorque_hip.h) */
    int32_t noise_model;            /* fixed-step solvers: 0 white field per RHS call (simple_solver.py:378-388);
                                       1 Ornstein-Uhlenbeck field per sub-step (physics/thermal_model.py:113-137) */
    double noise_corr_time;         /* ThermalFluctuations.correlation_time */
} stgo_config;

/* ThermalFluctuations._generate_correlated_noise (physics/thermal_model.py:113-137): one update of the unit-variance
 * Ornstein-Uhlenbeck state x with the white sample xi; the field is strength * x */
void stgo_ou_update(double x[3], const double xi[3], double dt, double corr_time);

/* per-env mutable state (envs/spin_torque_env.py:133-139) */
typedef struct {
    double m[3];
    double target[3];
    double total_energy;
    int32_t step_count;
    uint32_t rng_step;              /* steps taken by this env since creation (never reset): Philox counter word */
    double last_action[2];
} stgo_env_state;

typedef struct {
    float obs[12];
    double reward;
    uint8_t terminated, truncated;
    uint8_t status;                 /* 0 ok, 1 solver failed -> m unchanged, 2 a sub-step hit the non-finite reset */
    double energy;                  /* energy_consumed of this step */
    int32_t n_sub;                  /* RK4: sub-steps; RK45: accepted points (excluding t0) */
} stgo_step_out;

/* ---- A1/A2: SimpleLLGSSolver right-hand side (physics/simple_solver.py:297-388) ---- */
void stgo_simple_dmdt(const double m[3], const stgo_params* p, double gamma, double J,
                      const double h_thermal[3], double out[3]);

/* ---- A3/A4/A5: RobustLLGSSolver.solve as the env calls it (utils/robust_solver.py:75-150,
 *      physics/simple_solver.py:71-229).  Returns success (1) or fallback (0).
 *      traj (optional) receives the (n+1) x 3 normalised rows. first_zero_row = index of the first
 *      all-zero row (or -1); n_reset = number of sub-steps that took the non-finite -> [0,0,1] branch. */
int stgo_simple_solve(const double m0[3], double T, const stgo_params* p, const stgo_config* c,
                      double J, uint64_t env_id, uint32_t env_step,
                      double m_final[3], int32_t* n_steps, int32_t* first_zero_row, int32_t* n_reset,
                      double* traj, int64_t traj_cap_rows);

/* ---- A6: LLGSSolver.solve::llgs_rhs (physics/llgs_solver.py:92-126, 182-237) ---- */
void stgo_llgs_rhs(const double y[3], const stgo_params* p, double gamma, double J,
                   const double h_thermal[3], double out[3]);

/* ---- A7/A8: LLGSSolver.solve with scipy RK45 (physics/llgs_solver.py:51-180; scipy rk.py).
 *      Writes up to cap accepted points (t, normalised m, energy, torque-norm sum); returns number of
 *      points (including t0); *success as sol.success; m_final = last normalised row. ---- */
int64_t stgo_llgs_solve(const double m0[3], double T, const stgo_params* p, const stgo_config* c,
                        double J, uint64_t env_id, uint32_t env_step,
                        double m_final[3], int32_t* success, int64_t* n_attempts,
                        double* t_out, double* m_out, double* e_out, double* tq_out, int64_t cap);

/* ---- opt-in device-physics terms: SOTMRAMDevice.compute_spin_torque (devices/sot_mram.py:163-194)
 *      and VCMAMRAMDevice._compute_effective_anisotropy (devices/vcma_mram.py:122-147) ---- */
void stgo_sot_torque(const double m[3], double J, const stgo_params* p, double tau_dl[3], double tau_fl[3]);
double stgo_vcma_keff(double volt, const stgo_params* p);

/* ---- A9: compute_resistance (devices/stt_mram.py:78-94, sot_mram.py:196-228, vcma_mram.py:236-257) ---- */
double stgo_resistance(const double m[3], const stgo_params* p);

/* ---- A16 / A1 / A6: Brown thermal-field strength.  which = 0: SimpleLLGSSolver (kb = 1.38e-23,
 *      simple_solver.py:378-383); 1: LLGSSolver / ThermalFluctuations (k_b = 1.380649e-23,
 *      llgs_solver.py:85-90, thermal_model.py:46-73). ---- */
double stgo_thermal_strength(const stgo_params* p, double gamma, double temperature, int which);

/* ---- A10: SafetyWrapper.validate_action + _parse_action (utils/monitoring.py:288-315,
 *      envs/spin_torque_env.py:409-433) ---- */
void stgo_parse_action(const float action[2], const stgo_config* c, double* J, double* T);

/* ---- A12: _get_observation, vector mode (envs/spin_torque_env.py:490-524) ---- */
void stgo_observation(const stgo_env_state* s, const stgo_params* p, const stgo_config* c, float obs[12]);

/* ---- A10-A14: one SpinTorqueEnv.step (envs/spin_torque_env.py:310-407) ---- */
void stgo_env_step(stgo_env_state* s, const float action[2], const stgo_params* p, const stgo_config* c,
                   uint64_t env_id, stgo_step_out* out);

/* batch form used by bench.py's cpu_baseline leg and the gloo tests: env i uses params[cls[i]]
 * (cls == NULL -> params[0]); OpenMP-parallel over envs with n_threads threads (<=0 -> default). */
void stgo_env_step_batch(int64_t n, stgo_env_state* s, const float* actions /*[n][2]*/,
                         const stgo_params* params, const uint8_t* cls, const stgo_config* c,
                         uint64_t env_id0, stgo_step_out* out, int n_threads);

/* thermal-field generator, the same construction as the HIP kernels' (restated, not shared code): per (env, env step)
 * one xoshiro128+ stream seeded by Philox4x32-10(key = seed, counter = (env_id, env_step, tag)), 23-bit uniforms,
 * fp32 Box-Muller pairs, RHS call j consuming normals 3j..3j+2.  stgo_thermal_normals returns the normals of call
 * `call_idx` (replaying the stream from call 0). */
void stgo_thermal_normals(uint64_t seed, uint64_t env_id, uint32_t env_step, uint32_t call_idx, double z[3]);
void stgo_reset_draw(uint64_t seed, uint64_t env_id, uint32_t rng_step, int n_targets, double z[3], int* target_idx);
void stgo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* ---- SpinTorqueArray-v0: envs/array_env.py:362-520 ---- */
typedef struct {
    int32_t rows, cols;
    int32_t action_mode;            /* 0 individual, 1 row, 2 column, 3 global (array_env.py:427-445) */
    int32_t include_coupling;
    int32_t max_steps;
    int32_t obs_mode;               /* 0 'array' (rows*cols*6), 1 'vector' (rows*cols*6 + 4) */
    double max_current, max_duration, success_threshold, energy_penalty_weight, temperature;
} stgo_array_config;

/* coupling matrix of array_env.py:301-334; type 0 dipolar, 1 exchange, 2 stray_field; out [n][n] */
void stgo_array_coupling(int rows, int cols, int type, double strength, double* out);
/* device.compute_effective_field(m, 0) for the three device classes (stt_mram.py:55-76, sot_mram.py:79-112,
 * vcma_mram.py:85-120) */
void stgo_device_field(const double m[3], const stgo_params* p, double h[3]);
/* one SpinTorqueArrayEnv.step; pattern/target [n][3] row-major; action has n_action floats; obs has
 * n*6 (+4) floats; returns nothing, fills reward/terminated/truncated/energy */
void stgo_array_step(const stgo_array_config* c, const stgo_params* p, const double* coupling, double* pattern,
                     const double* target, double* total_energy, int32_t* step_count, const float* action, int n_action,
                     float* obs, double* reward, uint8_t* terminated, uint8_t* truncated, double* energy);
void stgo_array_observation(const stgo_array_config* c, const double* pattern, const double* target, double total_energy,
                            int32_t step_count, float* obs);

int stgo_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
