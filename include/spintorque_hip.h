/*
 * spintorque_hip.h -- C-ABI of libspintorque_hip.so: the MI355X (gfx950) implementation of the
 * SpinTorque-v0 env.step() hot path, vectorised over N independent macrospin environments.
 *
 * The reference (danieleschmidt/spin-torque-rl-gym) is pure Python and has no FFI of its own; the
 * seam this library replaces is Python duck-typing at two levels:
 *   env level    : SpinTorqueEnv.reset/step            (spin_torque_gym/envs/spin_torque_env.py:250-407)
 *   solver level : RobustLLGSSolver.solve / LLGSSolver.solve
 *                                                       (utils/robust_solver.py:75-150, physics/llgs_solver.py:51-180)
 * Each entry point below cites the reference code it stands in for.  INTEGRATION.md shows the ctypes
 * stub a maintainer of the reference would add to bind it.
 *
 * Conventions
 *   - plain C types only; every pointer marked [dev] is a device (HBM) pointer owned by the caller
 *     (any allocator: hipMalloc, a PyTorch-ROCm tensor's data_ptr(), ...); [host] pointers are host memory.
 *   - per-env arrays are structure-of-arrays: component-major, env index fastest (m is [3][N], obs is [12][N]),
 *     so that lane i of a wavefront touches element i of each row (coalesced).
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *     nothing synchronises with the host except stg_create, stg_destroy and stg_set_params (stg_get_state / stg_set_state enqueue copies on `stream`).
 *   - return value 0 = success, negative = error (STG_E_*); stg_last_error() returns a thread-local message.
 *   - one context per GPU; a context is not thread-safe.  Contexts are independent (no global state).
 *   - IEEE fp64 state and arithmetic (the reference's NumPy float64); observations are rounded to fp32 last,
 *     as the reference does (spin_torque_env.py:520).
 */
#ifndef SPINTORQUE_HIP_H
#define SPINTORQUE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STG_ABI_VERSION 4

typedef struct stg_ctx stg_ctx;

enum {
    STG_OK = 0,
    STG_E_INVALID = -1,   /* bad argument */
    STG_E_HIP = -2,       /* a HIP runtime call failed */
    STG_E_NOMEM = -3,
    STG_E_STATE = -4      /* call order (e.g. step before set_params/reset) */
};

/* integrator: which reference solver's semantics a step/solve follows */
enum {
    STG_SOLVER_RK4 = 0,   /* SimpleLLGSSolver(method='rk4') behind RobustLLGSSolver -- what SpinTorqueEnv builds
                             (spin_torque_env.py:93-102; simple_solver.py:278-295) */
    STG_SOLVER_EULER = 1, /* SimpleLLGSSolver(method='euler') (simple_solver.py:263-276) */
    STG_SOLVER_RK45 = 2   /* LLGSSolver: SciPy solve_ivp(method='RK45') (llgs_solver.py:130-139) */
};

enum { STG_DEV_STT = 0, STG_DEV_SOT = 1, STG_DEV_VCMA = 2 };

/* per-lane status written by stg_step / stg_solve */
enum {
    STG_STATUS_OK = 0,
    STG_STATUS_NOOP = 1,       /* solver reported success=False -> magnetisation left unchanged
                                  (spin_torque_env.py:461-467; robust_solver.py:140-150) */
    STG_STATUS_RESET = 2,      /* success, but >= 1 sub-step took the non-finite -> [0,0,1] branch
                                  (simple_solver.py:213-216) */
    STG_STATUS_INACTIVE = 3    /* env was already terminated/truncated and autoreset is off: not stepped */
};

enum { STG_OUT_SOA = 0, STG_OUT_RECORDS = 1 };
#define STG_RECORD_BYTES 56        /* 12 x f32 obs | f32 reward | u8 terminated | u8 truncated | u8 status | u8 0 */

#define STG_MAX_TARGETS 8
#define STG_MAX_CLASSES 64

/* SpinTorqueEnv.__init__ keyword arguments + solver constructor arguments
 * (spin_torque_env.py:36-53,93-102; llgs_solver.py:24-31; simple_solver.py:24-31) */
typedef struct {
    int32_t solver;                 /* STG_SOLVER_* */
    int32_t thermal;                /* include_thermal_fluctuations */
    double temperature;             /* K */
    double gamma;                   /* 2.21e5 m/(A s) */
    double max_step;                /* 1e-12 s */
    double rtol, atol;              /* RK45: 1e-6, 1e-9 */
    int32_t max_steps;              /* 100 */
    int32_t n_targets;              /* len(target_states), <= STG_MAX_TARGETS */
    double max_current;             /* 2e6 A/m^2 */
    double max_duration;            /* 5e-9 s */
    double success_threshold;       /* 0.9 */
    double energy_penalty_weight;   /* 0.1 */
    double targets[STG_MAX_TARGETS][3]; /* target_states, unit vectors (default +z, -z) */
    uint64_t seed;                  /* key of the in-kernel Philox4x32-10 (thermal field, device-side resets) */
    int64_t max_attempts;           /* RK45 attempt budget per solve; exceeded -> STG_STATUS_NOOP.  The reference has
                                       no such guard (its stiff cases simply never return, SURVEY.md headline 3). */
    int32_t skip_done;              /* 1: lanes whose episode already ended are not integrated (wavefront-level
                                       early-out) and report STG_STATUS_INACTIVE; 0: reference behaviour (step anyway) */
    int32_t torque_model;           /* 0 (default): the reference env's type-agnostic RHS for every device type.
                                       1: opt-in device-physics torque model for the fixed-step solvers (SURVEY 8f #1,
                                       BASELINE config 4): SOT classes use SOTMRAMDevice.compute_spin_torque
                                       (sot_mram.py:163-194) instead of the Slonczewski term, VCMA classes use
                                       H_k from VCMAMRAMDevice._compute_effective_anisotropy (vcma_mram.py:122-147)
                                       at V = J R(m) A while the pulse is on.  The reference env never calls these
                                       formulas; the model is pinned against the device classes, not against env runs. */
    int32_t wave_spec;              /* thermal kernels: every integrating wavefront gets a second wavefront that runs the
                                       envs' normal streams ahead into LDS (same values, same order: results are
                                       bit-identical).  0 = automatic (launches of <= 65536 envs), 1 always, -1 never */
    int32_t lane_sort;              /* schedule of envs onto lanes: 0 = automatic (sort), 1 = always sort the envs by pulse duration
                                       on the device before each step so that the lanes of a wavefront have equal trip
                                       counts, -1 = identity.  Results are unaffected: an env's arithmetic does not depend
                                       on the lane that runs it. */
    int32_t noise_model;            /* thermal field of the fixed-step solvers: 0 (default) = white, three fresh normals
                                       per RHS call (what the reference solvers do, simple_solver.py:378-388);
                                       1 = Ornstein-Uhlenbeck, the reference's ThermalFluctuations model
                                       (physics/thermal_model.py:113-137): once per sub-step
                                       x <- d x + sqrt(1 - d^2) xi, d = exp(-dt / noise_corr_time), field = strength * x
                                       for all stages of the sub-step, x = 0 at the start of every pulse.
                                       Not available with STG_SOLVER_RK45 (no fixed dt). */
    int32_t out_layout;             /* layout of a step's RL-facing outputs (obs, reward, terminated, truncated):
                                       STG_OUT_SOA (0, default): four caller arrays, obs component-major float[12][N];
                                       STG_OUT_RECORDS (1): ONE array of STG_RECORD_BYTES-byte records, env index major --
                                       record i = { float obs[12]; float reward; uint8 terminated, truncated, status, 0 } --
                                       passed as `obs` (reward/terminated/truncated arguments are ignored, may be NULL);
                                       final_obs is then env-major too, float[N][12].
                                       A shard's records are one contiguous block, so the multi-GPU exchange is a single
                                       all-gather straight into the learner's [N_global] record array: obs is then the
                                       [N_global, 12] strided view of it (Gym's own orientation), no transposition or
                                       concatenation copy anywhere. */
    double noise_corr_time;         /* correlation_time of ThermalFluctuations (default 1e-12 s); used when noise_model = 1 */
    int32_t lane_refill;            /* STG_SOLVER_RK45, single-step launches: persistent wavefronts share one global queue of the launch's
                                       envs in the sorted schedule's longest-first order, and a lane that has finished its env takes the next
                                       entry while its neighbours keep integrating (ABI v3; since round 4 the queue is global, striped over 64 cursors so that the atomics do not serialise).  0 = automatic
                                       (launches of more than 131072 envs, more than 98304 with the thermal field: 1024 wavefronts -- one per
                                       SIMD -- up to 8 envs per lane on average, 2048 beyond), -1 = never, >= 2 = that many envs per lane on
                                       average (ceil(blocks / lane_refill) wavefronts).  Per-env arithmetic is untouched: results are
                                       bit-identical to the one-env-per-lane launch.  Not used with per-env parameter records, fused steps
                                       (K > 1) or a forced wave_spec = 1. */
    int32_t reserved0;              /* must be 0 */
} stg_config;

/* one reference-style device_params dict, flattened with the defaults the reference's .get() calls use
 * (simple_solver.py:126-131; llgs_solver.py:79-82,192-205; spin_torque_env.py:476,502; devices/ (all three device modules)) */
typedef struct {
    double damping;                 /* 'damping' */
    double ms;                      /* 'saturation_magnetization' */
    double ku;                      /* 'uniaxial_anisotropy' */
    double volume;                  /* 'volume' */
    double polarization;            /* 'polarization' */
    double easy_axis[3];            /* 'easy_axis', raw */
    double demag[3];                /* 'demag_factors' (LLGSSolver), default 0,0,1 */
    double a_ex;                    /* 'exchange_constant' (LLGSSolver), default 20e-12 */
    double area;                    /* 'area', default 1e-14 */
    double r_p, r_ap;               /* 'resistance_parallel', 'resistance_antiparallel' */
    double ref_m[3];                /* 'reference_magnetization', raw */
    double r_series;                /* SOT: 0.1*(rho_hm/t_hm)/(area*1e-12) (sot_mram.py:218-223); else 0 */
    double sot_tau_dl, sot_tau_fl;  /* SOT: tau_dl_factor, tau_fl_factor (sot_mram.py:61-72); else 0 */
    double sot_sigma[3];            /* SOT: z x current_direction (sot_mram.py:180-186), default (0,1,0) */
    double vcma_xi;                 /* VCMA: 'vcma_coefficient' */
    double vcma_td;                 /* VCMA: 'dielectric_thickness' */
    double vcma_vbd;                /* VCMA: 'breakdown_voltage' */
    double shape_demag[3];          /* SOT/VCMA: demag factors of compute_effective_field from 'aspect_ratio'
                                       (sot_mram.py:114-132, vcma_mram.py:149-166); zeros for STT.  Array env only. */
    int32_t dev_type;               /* STG_DEV_* : selects the compute_resistance form */
    int32_t params_valid;           /* outcome of validate_parameters(params,'stt_mram') (utils/validation.py:176-234),
                                       evaluated by the host mirror; 0 -> every solve falls back (no-op) */
} stg_device_params;

/* ---- lifecycle -------------------------------------------------------------------------------------- */

/* Allocates the SoA state for n_envs environments on GPU `device_id`.
 * env_id0 = global index of this context's first env (multi-GPU shards: rank r owns [env_id0, env_id0+n_envs)),
 * used only as the Philox counter so that results do not depend on the partition. */
int stg_create(stg_ctx** out, int device_id, int64_t n_envs, int64_t env_id0, const stg_config* cfg);
void stg_destroy(stg_ctx* ctx);
const char* stg_last_error(void);
int stg_abi_version(void);

/* ---- device parameter surface (spin_torque_gym.devices; DeviceFactory.create_device, device_factory.py:49-77) --- */

/* n_classes parameter sets (<= STG_MAX_CLASSES) [host]; cls [dev] uint8[N] selects the set of each env
 * (NULL: every env uses table[0]).  The derived per-class constants are staged in LDS by the kernels. */
int stg_set_params(stg_ctx* ctx, const stg_device_params* table, int32_t n_classes, const uint8_t* cls);

/* ---- env level ---------------------------------------------------------------------------------------- */

/* Per-env parameters (SURVEY 8b `stg_set_params_per_env`; domain randomisation / device-to-device variation): instead of
 * a class table, every env carries its own parameter record.  soa_params is a DEVICE array [STG_NPARAM][N] of doubles
 * whose rows are the double-valued fields of stg_device_params in declaration order (damping, ms, ku, volume,
 * polarization, easy_axis[3], demag[3], a_ex, area, r_p, r_ap, ref_m[3], r_series, sot_tau_dl, sot_tau_fl,
 * sot_sigma[3], vcma_xi, vcma_td, vcma_vbd, shape_demag[3]); dev_type and params_valid are DEVICE arrays uint8[N]
 * (STG_DEV_*; the host-evaluated outcome of utils/validation.py:176-234 per env).  The library copies all three.
 * Each lane derives its constants in the kernel prologue with the same arithmetic the host uses for a class table, so
 * an env gives bit-identical results either way.  Replaces a previous stg_set_params (and vice versa); stg_solve*,
 * stg_device_terms and stg_thermal_strength work on class tables only. */
#define STG_NPARAM 30
int stg_set_params_per_env(stg_ctx* ctx, const double* soa_params, const uint8_t* dev_type, const uint8_t* params_valid);

/* SpinTorqueEnv.reset (spin_torque_env.py:250-308) for the envs with mask[i] != 0 (mask NULL: all).
 * init_m / target [dev] double[3][N]: options['initial_state'] / options['target_state'] (normalised by the kernel as
 * device.validate_magnetization does); NULL: drawn on the device (normal(0,1,3) normalised; uniform choice among
 * cfg.targets) from Philox(seed, env_id, episode).  obs_out [dev] float[12][N] (cfg.out_layout = STG_OUT_RECORDS: the
 * record array, 8-byte aligned: the obs fields of every env are written -- a masked call reports the current observation of
 * the envs it leaves alone --, the reward/flag fields are zeroed for the reset envs and left untouched for the others)
 * may be NULL. */
int stg_reset(stg_ctx* ctx, const uint8_t* mask, const double* init_m, const double* target,
              uint64_t seed, float* obs_out, void* stream);

/* SpinTorqueEnv.step (spin_torque_env.py:310-407) for all N envs.
 * actions [dev]: [2][N] (row 0 current density A/m^2, row 1 pulse duration s), float32 (act_f64 = 0) or float64.
 * obs float[12][N]; reward float[N]; terminated/truncated uint8[N] (cfg.out_layout = STG_OUT_RECORDS: obs is the record
 * array uint8[N][STG_RECORD_BYTES] and the other three are ignored); optional (may be NULL): reward_f64 double[N]
 * (the reward before rounding to fp32), energy double[N] (info['energy_consumed'], spin_torque_env.py:474-480),
 * status uint8[N] (STG_STATUS_*). */
int stg_step(stg_ctx* ctx, const void* actions, int32_t act_f64, float* obs, float* reward, double* reward_f64,
             double* energy, uint8_t* terminated, uint8_t* truncated, uint8_t* status, void* stream);

/* K consecutive env steps in one launch (state stays in registers between steps).
 * actions [K][2][N]; outputs as stg_step with a leading [K] dimension (records: [K][N][STG_RECORD_BYTES]); out_every = 1
 * writes every step's outputs, 0 only the last step's (leading dimension 1).
 * autoreset != 0 (same-step auto-reset, the usual GPU vector-env convention): an env whose episode ends at step k
 * reports that step's reward / terminated / truncated, is then reset on the device (as stg_reset with NULL
 * init_m/target, Philox key cfg.seed) and its obs row holds the NEW episode's first observation; final_obs
 * (float[K or 1][12][N] -- float[K or 1][N][12] with STG_OUT_RECORDS --, may be NULL) receives the terminal observation of
 * such envs (entries of other envs untouched).  With STG_OUT_RECORDS the record array and final_obs must be 8-byte
 * aligned (the kernel stores float pairs); a misaligned pointer is rejected with STG_E_INVALID. */
int stg_step_many(stg_ctx* ctx, int32_t K, const void* actions, int32_t act_f64, int32_t out_every, int32_t autoreset,
                  float* obs, float* final_obs, float* reward, double* reward_f64, double* energy, uint8_t* terminated,
                  uint8_t* truncated, uint8_t* status, void* stream);

/* state access for parity checks and checkpoint/resume; all pointers [dev], any may be NULL.
 * m, target: double[3][N]; total_energy: double[N]; step_count: int32[N]; rng_step: uint32[N]; done: uint8[N] */
int stg_get_state(stg_ctx* ctx, double* m, double* target, double* total_energy, int32_t* step_count,
                  uint32_t* rng_step, uint8_t* done, void* stream);
int stg_set_state(stg_ctx* ctx, const double* m, const double* target, const double* total_energy,
                  const int32_t* step_count, const uint32_t* rng_step, const uint8_t* done, void* stream);

/* On-device metrics since creation / the last reset of the counters ([host] out[4]): env steps taken, integrator
 * work units (RK4/Euler sub-steps, RK45 attempted steps), reserved, steps that ended as STG_STATUS_NOOP.
 * Stands in for the host-side bookkeeping of EnvironmentMonitor / RobustLLGSSolver.get_statistics
 * (utils/monitoring.py:30-268, utils/robust_solver.py:311-328).  Synchronises the device. */
int stg_get_counters(stg_ctx* ctx, uint64_t* out, int32_t reset);

/* Where the dispatcher placed the wavefronts of a recent step launch, and when each ran (ABI v4).  The launch schedules of this
 * library (which 64-env block a wavefront takes) are speed heuristics built on observed dispatcher behaviour; every step launch
 * records, once per wavefront, the SIMD it ran on and the real-time counter when it started and when it retired, so that a caller
 * (bench.py: `roofline.placement`) can tell "this run got an unlucky placement" from "this build is slower" and see the per-SIMD
 * timeline of a launch.  launches_back = 0 addresses the most recent step launch of the context, 1 the one before, ... (the last 32
 * are kept).  out [host] receives STG_PLACEMENT_WORDS_PER_WAVE words per wavefront, wavefront index = workgroup *
 * waves_per_workgroup + wavefront (the first 4096 wavefronts of the launch; cap = size of out in words):
 *   word 0: bits 0-15 = HW_ID[15:0] (bits 5:4 SIMD, 11:8 CU, 12 SH, 15:13 SE), bits 16-19 = XCC_ID, bit 20 = producer wavefront of a
 *           wave-specialised pair, bit 31 = valid (a workgroup beyond the batch leaves 0)
 *   word 1, word 2: low 32 bits of the 100 MHz real-time counter (s_memrealtime) at the wavefront's start / when it retired (0: the
 *           wavefront had no env)
 *   word 3: the wavefront's work -- the largest number of integrator work units (sub-steps / RK45 attempts) any of its lanes did in
 *           this launch; 0 for a producer
 * Returns the number of WAVEFRONTS written or a negative error.  Synchronises the device.  Cost per launch: two s_getreg, two
 * s_memrealtime, a six-step lane reduction and four 4-byte stores per wavefront, nothing inside any loop.  The reference has no counterpart (its bookkeeping is
 * host-side: utils/monitoring.py:30-268). */
#define STG_PLACEMENT_WORDS_PER_WAVE 4
int stg_get_placement(stg_ctx* ctx, int32_t launches_back, uint32_t* out, int32_t cap, int32_t* n_workgroups,
                      int32_t* waves_per_workgroup);

/* ---- solver level ------------------------------------------------------------------------------------- */

/* RobustLLGSSolver.solve / LLGSSolver.solve over (0, T[i]) with current_func(t) = J[i] if t <= T[i] else 0 and zero
 * applied field, as SpinTorqueEnv._simulate_dynamics calls it (spin_torque_env.py:442-459), for N independent
 * problems.  m0 double[3][N]; J, T double[N]; m_final double[3][N] (last trajectory row; m0 when success = 0);
 * n_points int32[N] (RK4/Euler: sub-steps n; RK45: accepted points excluding t0); success uint8[N].
 * env_step: Philox counter word (thermal on only).  Uses the context's config and parameter table/classes. */
int stg_solve(stg_ctx* ctx, const double* m0, const double* J, const double* T, uint32_t env_step,
              double* m_final, int32_t* n_points, uint8_t* success, void* stream);

/* as stg_solve, additionally recording the trajectory: the first traj_cap rows of t [traj_cap][N],
 * m [traj_cap][3][N] (normalised rows, llgs_solver.py:152-153), energy [traj_cap][N] (llgs_solver.py:239-262) and
 * torques [traj_cap][N] = |tau_stt| + |tau_fl| at each accepted point (llgs_solver.py:159-172); energy and torques are
 * the LLGSSolver result dict's by-products: RK45 only, either may be NULL.  Row 0 is t0. */
int stg_solve_traj(stg_ctx* ctx, const double* m0, const double* J, const double* T, uint32_t env_step,
                   int32_t traj_cap, double* t, double* m, double* energy, double* torques,
                   double* m_final, int32_t* n_points, uint8_t* success, void* stream);

/* ---- device-class formulas (opt-in torque model) ------------------------------------------- */

/* Evaluates, per env, the device-class formulas the torque model uses: for SOT classes tau_dl, tau_fl of
 * SOTMRAMDevice.compute_spin_torque(J, m) (double[3][N] each, zeros for other classes), and for VCMA classes
 * K_eff(volt) of VCMAMRAMDevice._compute_effective_anisotropy (double[N]; the class' K for other classes).
 * m double[3][N], J, volt double[N]; any output may be NULL. */
int stg_device_terms(stg_ctx* ctx, const double* m, const double* J, const double* volt, double* tau_dl, double* tau_fl,
                     double* k_eff, void* stream);

/* ---- thermal field (physics/thermal_model.py:12-137; simple_solver.py:378-386; llgs_solver.py:85-90,111-113) ---- */

/* Brown field strength of class `cls` for the context's solver kind and temperature ([host] out). */
int stg_thermal_strength(stg_ctx* ctx, int32_t cls, double* out);
/* Dumps the standard normals the kernels would draw for RHS calls call0..call0+n_calls-1 of env step `env_step`:
 * z [dev] double[n_calls][3][N].  Diagnostic entry used by the distribution tests. */
int stg_thermal_normals(stg_ctx* ctx, uint32_t env_step, uint32_t call0, int32_t n_calls, double* z, void* stream);

/* ---- SpinTorqueArray-v0 (spin_torque_gym/envs/array_env.py) --------------------------------- */

typedef struct stg_array_ctx stg_array_ctx;

/* SpinTorqueArrayEnv.__init__ keyword arguments that reach the step path (array_env.py:30-52) */
typedef struct {
    int32_t rows, cols;             /* array_size; rows*cols <= 64 */
    int32_t action_mode;            /* 0 'individual' [device, J, T], 1 'row', 2 'column', 3 'global' [J, T] (:427-445) */
    int32_t include_coupling;
    int32_t max_steps;              /* 200 */
    int32_t obs_mode;               /* 0 'array' (rows*cols*6 floats), 1 'vector' (rows*cols*6 + 4) (:533-557) */
    double max_current, max_duration, success_threshold, energy_penalty_weight, temperature;
} stg_array_config;

/* n_arrays independent arrays on GPU device_id; dev = the device class of every cell (DeviceFactory.create_device per
 * cell, array_env.py:113-120); coupling [host] double[n][n] = _compute_coupling_matrix (:301-334), may be NULL when
 * include_coupling = 0.  env_id0: global index of the first array (device-side reset draws). */
int stg_array_create(stg_array_ctx** out, int device_id, int64_t n_arrays, int64_t env_id0, const stg_array_config* cfg,
                     const stg_device_params* dev, const double* coupling);
void stg_array_destroy(stg_array_ctx* ctx);

/* SpinTorqueArrayEnv.reset (array_env.py:336-360) for arrays with mask[i] != 0 (NULL: all).  init_pattern / target [dev]
 * double[rows*cols][3][N] (options['initial_pattern'] / options['target_pattern']); NULL init_pattern: normalised normal
 * draws on the device; NULL target: keep the current target (the +-z checkerboard of :161-170 at first).
 * obs_out float[obs_dim][N] or NULL. */
int stg_array_reset(stg_array_ctx* ctx, const uint8_t* mask, const double* init_pattern, const double* target,
                    uint64_t seed, float* obs_out, void* stream);

/* SpinTorqueArrayEnv.step (array_env.py:362-409) for all N arrays.  actions [dev] float[A][N], A = 3 ([index, J, T]) or
 * 2 in 'global' mode -- where, as in the reference, action[1] is read as the current density and the duration defaults
 * to 1 ns.  obs float[obs_dim][N]; reward float[N]; reward_f64 / energy double[N] or NULL; terminated / truncated uint8[N]. */
int stg_array_step(stg_array_ctx* ctx, const float* actions, float* obs, float* reward, double* reward_f64, double* energy,
                   uint8_t* terminated, uint8_t* truncated, void* stream);

/* pattern, target: double[rows*cols][3][N]; total_energy double[N]; step_count int32[N]; any may be NULL */
int stg_array_get_state(stg_array_ctx* ctx, double* pattern, double* target, double* total_energy, int32_t* step_count,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPINTORQUE_HIP_H */
