// spintorque_hip.hip -- kernels and C-ABI of libspintorque_hip.so (see include/spintorque_hip.h).
//
// Layout in HBM (all owned by the context, structure-of-arrays, env index fastest):
//   mx,my,mz, tx,ty,tz, e_tot : f64[N]   step : i32[N]   rng : u32[N]   done : u8[N]       = 65 B/env
//   class table                : f64[n_classes][C_COUNT]  (derived constants, <= 64 classes)
//   cls                        : u8[N] (caller-owned) when n_classes > 1
//   per-env parameters         : f64[STG_NPARAM][N] + type/valid u8[N] (library-owned copy), alternative to the class table
// One lane per env; the step kernel's workgroups hold one integrating wavefront (launches below 65 536 envs: small
// batches spread over as many CUs as they have wavefronts) or four (one per SIMD of a CU), plus, in the wave-specialised
// thermal form, one producer wavefront each; the hot loops live entirely in registers.  With one device class the
// constants are read through scalar loads (SGPRs); with several (or per-env parameters) each lane reads its row of
// derived constants from LDS.
#include "../../include/spintorque_hip.h"
#include "stg_physics.hpp"
#include <type_traits>

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

using namespace stg;

// ------------------------------------------------------------------------------------------------
// kernel argument blocks
// ------------------------------------------------------------------------------------------------
struct StateView {
    double *mx, *my, *mz, *tx, *ty, *tz, *etot;
    int32_t* step;
    uint32_t* rng;
    uint8_t* done;
};

struct CfgView {
    double temperature, max_step, rtol, atol, max_current, max_duration, thr, w_energy;
    double temp_norm, inv_max_current, inv_max_duration;   // temperature/300, 1/max_current, 1/max_duration (observation)
    double inv_tau;               // 1/noise_corr_time when the Ornstein-Uhlenbeck field is selected, else 0 (white field)
    double targets[STG_MAX_TARGETS][3];
    uint64_t seed;
    int64_t max_attempts;
    int32_t max_steps, n_targets, skip_done;
};

// Per-env parameter block (stg_set_params_per_env): STG_NPARAM rows of N doubles in the field order of
// stg_device_params, plus the device type and the host-evaluated validity flag per env.
struct EnvParams {
    const double* soa;        // [STG_NPARAM][N] or nullptr
    const uint8_t* type;      // [N]
    const uint8_t* valid;     // [N]
    double gamma, temperature;
};

struct StepArgs {
    StateView s;
    CfgView c;
    int64_t N, env_id0;
    const double* ctab;
    const uint8_t* cls;
    int32_t ncls;
    EnvParams ep;                 // per-env parameters (soa == nullptr: class table)
    int32_t force_wg1;
    const void* actions;          // [K][2][N]
    int32_t K, out_every, autoreset;
    const uint32_t* perm;         // lane -> env (duration-sorted schedule) or nullptr
    unsigned long long* counters; // [4]: env-steps, integrator sub-steps/attempts, RHS evaluations, no-op steps
    float* obs;                   // [K or 1][12][N]
    float* final_obs;             // [K or 1][12][N] or nullptr: terminal observation of envs auto-reset at that step
    float* reward;                // [K or 1][N]
    double *reward64, *energy;
    uint8_t *term, *trunc, *status;
};

struct SolveArgs {
    CfgView c;
    int64_t N, env_id0;
    const double* ctab;
    const uint8_t* cls;
    int32_t ncls;
    const double *m0, *J, *T;
    uint32_t env_step;
    double* m_final;
    int32_t* n_points;
    uint8_t* success;
    int32_t traj_cap;
    double *traj_t, *traj_m, *traj_e;
};

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_env_params(const EnvParams& e, int64_t N, int64_t i, stg_device_params& p) {
    const double* q = e.soa + i;
    int r = 0;
    auto nx = [&]() { const double v = q[(int64_t)r * N]; ++r; return v; };
    p.damping = nx(); p.ms = nx(); p.ku = nx(); p.volume = nx(); p.polarization = nx();
    for (int k = 0; k < 3; ++k) p.easy_axis[k] = nx();
    for (int k = 0; k < 3; ++k) p.demag[k] = nx();
    p.a_ex = nx(); p.area = nx(); p.r_p = nx(); p.r_ap = nx();
    for (int k = 0; k < 3; ++k) p.ref_m[k] = nx();
    p.r_series = nx(); p.sot_tau_dl = nx(); p.sot_tau_fl = nx();
    for (int k = 0; k < 3; ++k) p.sot_sigma[k] = nx();
    p.vcma_xi = nx(); p.vcma_td = nx(); p.vcma_vbd = nx();
    for (int k = 0; k < 3; ++k) p.shape_demag[k] = nx();
    p.dev_type = (int32_t)e.type[i];
    p.params_valid = (int32_t)e.valid[i];
}

// Returns this lane's row of derived constants.  One class: the (wave-uniform) global table row, which the compiler
// turns into scalar loads.  MULTI: the class table staged in LDS, or -- per-env parameters -- a row per lane derived
// here from the env's own record (the LDS block holds exactly 64 rows: per-env launches use 64 integrating lanes per
// workgroup; lanes of a producer wavefront read the row of the integrating lane they mirror).
template <bool MULTI>
__device__ __forceinline__ const double* class_row(const double* ctab, const uint8_t* cls, int32_t ncls, int64_t i,
                                                   bool in_range, double* lds, const EnvParams& ep, int64_t N) {
    if (MULTI) {
        if (ep.soa) {
            const int lane = (int)(threadIdx.x & 63u);
            if (threadIdx.x < 64 && in_range) {
                stg_device_params p;
                load_env_params(ep, N, i, p);
                derive_row(p, ep.gamma, ep.temperature, lds + lane * C_COUNT);
            }
            __syncthreads();
            return lds + lane * C_COUNT;
        }
        for (int j = threadIdx.x; j < ncls * C_COUNT; j += blockDim.x) lds[j] = ctab[j];
        __syncthreads();
        const int c = in_range ? (int)cls[i] : 0;
        return lds + (c < ncls ? c : 0) * C_COUNT;
    }
    return ctab;
}

__device__ __forceinline__ SimpleK load_simple(const double* r) {
    return SimpleK{V3{r[C_EX], r[C_EY], r[C_EZ]}, r[C_HK], r[C_MS], r[C_ALPHA], r[C_GEFF], r[C_HS_SIMPLE]};
}
__device__ __forceinline__ LlgsK load_llgs(const double* r) {
    return make_llgs(V3{r[C_RX], r[C_RY], r[C_RZ]}, V3{r[C_DX], r[C_DY], r[C_DZ]}, r[C_HK], r[C_HEX], r[C_ALPHA], r[C_GAMMA],
                     r[C_HS_LLGS]);
}
__device__ __forceinline__ LlgsEnergyK load_energy(const double* r) {
    return LlgsEnergyK{r[C_KUV], r[C_EDEMAG], V3{r[C_RX], r[C_RY], r[C_RZ]}, V3{r[C_NX], r[C_NY], r[C_NZ]}};
}

template <int SOLVER, bool THERMAL, bool RECORD, bool AXIS_Z, bool DEVPHYS, class NSRC>
__device__ __forceinline__ SolveOut run_solver(const V3& m, double J, double T, const double* row, const CfgView& c,
                                               const RngKey& rk, const Recorder& rec, NSRC& ns, bool enabled) {
    if (SOLVER == STG_SOLVER_RK45) {
        const LlgsK k = load_llgs(row);
        LlgsEnergyK ek{};
        if (RECORD) ek = load_energy(row);
        return llgs_solve<THERMAL, RECORD, AXIS_Z>(m, J, T, k, row[C_BETA], row[C_BETAP], c.rtol, c.atol, c.max_step,
                                                   c.max_attempts, rk, rec, ek, ns, enabled);
    }
    const SimpleK k = load_simple(row);
    DevTorque dv{0.0, 0.0, V3{0.0, 1.0, 0.0}, k.hk, false};
    double pol = row[C_POL];
    if (DEVPHYS) {
        // per-lane coefficients of the device-physics torque model (stg_physics.hpp: DevTorque)
        const int kind = (int)row[C_DEVTYPE];
        const bool sot = kind == STG_DEV_SOT;
        dv.any_sot = __ballot(sot) != 0ull;
        dv.sigma = V3{row[C_SIGX], row[C_SIGY], row[C_SIGZ]};
        if (sot) {
            pol = 0.0;                                           // no Slonczewski term for SOT lanes
            dv.sdl = row[C_SOT_DL] * J / row[C_MSV];
            dv.sfl = row[C_SOT_FL] * J / row[C_MSV];
        }
        if (kind == STG_DEV_VCMA) {
            // V = J R(m_before) A, the voltage the env computes for this pulse (spin_torque_env.py:475-477)
            const V3 ref{row[C_REFX], row[C_REFY], row[C_REFZ]};
            const double r = resistance(m, kind, row[C_RP], row[C_RAP], row[C_TMR], ref, row[C_RSERIES]);
            const double keff = vcma_keff(J * r * row[C_AREA], row[C_KU], row[C_VCMA_XI], row[C_VCMA_TD2], row[C_VCMA_VBD]);
            dv.hk_pulse = (2 * keff) / row[C_MU0MS];
        }
    }
    return simple_solve<SOLVER == STG_SOLVER_EULER ? 1 : 0, THERMAL, RECORD, AXIS_Z, DEVPHYS>(
        m, J, T, k, pol, row[C_MSV], row[C_VALID] != 0.0, c.temperature, c.max_step, rk, rec, dv, ns, c.inv_tau, enabled);
}

// SpinTorqueEnv.reset draws (spin_torque_env.py:286-299) from the device generator: normal(0,1,3) normalised and a
// uniform choice among the target states, from this env's stream tagged 0xFFFFFFFF (the thermal field uses tag 0).
__device__ __forceinline__ void device_reset_draw(uint64_t seed, uint64_t env_id, uint32_t rng_step, const CfgView& c,
                                                  bool draw_m, bool draw_t, V3& m, V3& tgt) {
#pragma clang fp contract(off)
    NormalStream ns;
    ns.init(seed ^ 0x9E3779B97F4A7C15ull, env_id, rng_step, 0xFFFFFFFFu);
    const V3 z = ns.draw3_even();
    const uint32_t r = ns.next();
    if (draw_m) {
        const double s2 = dot(z, z), inv = rsqrt_fast(s2);
        m = (s2 < 1e-24) ? V3{0.0, 0.0, 1.0} : V3{z.x * inv, z.y * inv, z.z * inv};
    }
    if (draw_t) {
        const int idx = (int)__umulhi(r, (uint32_t)c.n_targets);
        tgt = V3{c.targets[idx][0], c.targets[idx][1], c.targets[idx][2]};
    }
}

// A12: _get_observation (vector mode), written component-major.
__device__ __forceinline__ void write_obs(float* obs, int64_t N, int64_t i, const V3& m, const V3& tgt, const double* row,
                                          const CfgView& c, int32_t step, double etot, double J, double T) {
#pragma clang fp contract(off)
    const V3 ref{row[C_REFX], row[C_REFY], row[C_REFZ]};
    const double r = resistance(m, (int)row[C_DEVTYPE], row[C_RP], row[C_RAP], row[C_TMR], ref, row[C_RSERIES]);
    obs[0 * N + i] = obs_cast(m.x);
    obs[1 * N + i] = obs_cast(m.y);
    obs[2 * N + i] = obs_cast(m.z);
    obs[3 * N + i] = obs_cast(tgt.x);
    obs[4 * N + i] = obs_cast(tgt.y);
    obs[5 * N + i] = obs_cast(tgt.z);
    obs[6 * N + i] = obs_cast(r / row[C_RP]);
    // (the normalisations by run constants are multiplications by host-computed reciprocals: <= 1 ulp of fp64 away
    // from the reference's quotients, before the cast to fp32)
    obs[7 * N + i] = obs_cast(c.temp_norm);
    obs[8 * N + i] = obs_cast((double)(c.max_steps - step) / (double)c.max_steps);
    obs[9 * N + i] = obs_cast(etot * 1e12);
    obs[10 * N + i] = obs_cast(J * c.inv_max_current);
    obs[11 * N + i] = obs_cast(T * c.inv_max_duration);
}

constexpr int COUNTER_STRIPES = 1024;     // copies of the on-device counters (one 64-byte line each)
constexpr int COUNTER_STRIDE = 8;         // u64 per stripe

// Adds the sums of three per-lane counters over the executing lanes of the wavefront to dst[0], dst[1], dst[3] with one
// global atomic each.  The lanes accumulate in LDS (ds_add_u64 on one address: the LDS unit serialises the lanes while
// the wavefront issues nothing); LDS operations of one wavefront complete in order, so no barrier is needed.
__device__ __forceinline__ void wave_add3(unsigned long long* dst, unsigned long long* lds3, unsigned long long v0,
                                          unsigned long long v1, unsigned long long v3) {
    const bool first = (int)__lane_id() == __builtin_ctzll(__ballot(1));
    if (first) { lds3[0] = 0; lds3[1] = 0; lds3[2] = 0; }
    atomicAdd(&lds3[0], v0);
    atomicAdd(&lds3[1], v1);
    if (__ballot(v3 != 0) != 0ull) atomicAdd(&lds3[2], v3);
    if (first) {
        const unsigned long long s0 = lds3[0], s1 = lds3[1], s3 = lds3[2];
        if (s0) atomicAdd(dst + 0, s0);
        if (s1) atomicAdd(dst + 1, s1);
        if (s3) atomicAdd(dst + 3, s3);
    }
}

constexpr int PLAN_DUR = 256;         // 20 ps of pulse duration per bucket at the default 5 ns maximum
constexpr int PLAN_BUCKETS = 3 * PLAN_DUR;   // x device kind (device-physics torque model: type-uniform wavefronts)
constexpr int PLAN_THREADS = 1024;
constexpr int PLAN_ITEMS = 4;
constexpr int TILE_ENVS = PLAN_THREADS * PLAN_ITEMS;   // 4096 envs sorted together (one plan workgroup)
constexpr int TILE_WAVES = TILE_ENVS / 64;             // = 64 wavefronts of the step launch

// Which 64-slot block of the schedule integrating wavefront `cw` of workgroup `b` takes; a workgroup holds WGW = 1 or 4
// integrating wavefronts.
//  * WGW = 4 (launches of at least one such workgroup per CU): the dispatcher puts the wavefronts of a 256-thread
//    workgroup on the four SIMDs of a CU one each, deterministically -- 64-thread workgroups were observed to double up
//    on some SIMDs and leave others empty (tools/probes/wave_placement.hip: up to 104 of 1024 SIMDs with two wavefronts
//    at 65 536 envs), which a launch with one wavefront per SIMD pays for in full.
//  * With the sorted schedule, a tile's workgroups share an XCD (so its L2 merges their scattered accesses): workgroups
//    are observed to be dealt round-robin over the 8 XCDs (b % 8 labels the XCD group; a speed heuristic only, never a
//    correctness assumption), XCD group r takes tiles r, r+8, r+16, ... and walks them rank-major (longest wavefronts of
//    every tile first).  Tiles beyond the last complete group of 8 keep the identity map.
//  * Which wavefronts of a tile share a workgroup (= a CU): with one workgroup per CU, strided ranks u, u+16, u+32, u+48
//    -- measured 1.96 ms against 2.23 ms for consecutive ranks on the RK45 step at 65 536 envs: four wavefronts that
//    are busy for the whole launch slow each other down, a long one next to progressively shorter ones does not.
template <int WGW>
__device__ __forceinline__ int64_t stg_slot_block(uint32_t b, uint32_t nwg, bool sorted, int cw, bool pairs) {
    constexpr uint32_t TILE_WGS = TILE_WAVES / WGW;                   // workgroups per tile
    if (!sorted) return (int64_t)b * WGW + cw;
    if (WGW == 1 && pairs && nwg == 1024) {
        // Wave-specialised launch with exactly one integrating wavefront per SIMD (65 536 envs; 16 tiles, two per XCD
        // group).  Observed placement (tools/probes/wave_placement.hip, 1024 x 128 threads): a CU takes the workgroups
        // q, q+32, q+64, q+96 of its XCD group and the producer of arrival g shares a SIMD with the integrating
        // wavefront of arrival g+1 (cyclically).  Arrivals alternate between the long half of a tile (ranks j) and the
        // short half (ranks 63-j): every long integrating wavefront then shares its SIMD with the producer of a short
        // one, which retires early, and its own producer runs next to a short integrating wavefront.
        const uint32_t r = b % 8, q = b / 8, g = q / 32, j = q % 32;
        return (int64_t)((g >> 1) * 8 + r) * TILE_WAVES + ((g & 1) ? (TILE_WAVES - 1 - j) : j);
    }
    const uint32_t tiles8 = (nwg / (8 * TILE_WGS)) * 8;              // tiles in complete groups of 8
    if (b >= tiles8 * TILE_WGS) return (int64_t)b * WGW + cw;
    const uint32_t r = b % 8, q = b / 8;                              // XCD group, position inside the group
    const uint32_t tiles_per_xcd = tiles8 / 8;
    const uint32_t u = q / tiles_per_xcd, t = (q % tiles_per_xcd) * 8 + r;
    // one workgroup per CU at most (everything resident from the start): spread; otherwise keep wavefronts of similar
    // duration together, so that a workgroup's four SIMD slots come free together for the next one (measured 6.0 ms
    // against 7.6 ms at 262 144 envs)
    const uint32_t rank = (nwg <= 256) ? (u + TILE_WGS * cw) : (WGW * u + cw);
    return (int64_t)t * TILE_WAVES + rank;
}

// ------------------------------------------------------------------------------------------------
// env.step kernel (A10-A14 around the solver), K fused steps per launch
// ------------------------------------------------------------------------------------------------
// Workgroup = WGW integrating wavefronts: 4 (256 envs, one wavefront per SIMD of the CU) for launches that fill the chip,
// 1 for smaller ones (which then spread over four times as many CUs).
// PC = producer/consumer wave specialisation (thermal only, launches of at most 65 536 envs): every integrating
// wavefront gets a second wavefront that runs the normal streams of the same envs one chunk ahead into LDS
// (stg_physics.hpp: SharedNormalsT).  Same values in the same order, so results are identical.  It is launched with
// WGW = 1 (128-thread workgroups): the dispatcher was observed to put the two wavefronts on different SIMDs and to give
// every SIMD one integrating and one producing wavefront at 65 536 envs (tools/probes/wave_placement.hip).  The code
// supports WGW = 4 (producer 4+w serves integrating wavefront 3-w, all eight in lockstep) but that form measured no
// better for RK45 and 8 % worse for RK4, so it is not instantiated.
template <int SOLVER, bool THERMAL, bool MULTI, bool AXIS_Z, bool DEVPHYS, typename AT, bool PC, int WGW>
#ifndef STG_STEP_ATTR
#define STG_STEP_ATTR
#endif
__global__ void __launch_bounds__(PC ? 2 * WGW * 64 : WGW * 64) STG_STEP_ATTR stg_step_kernel(const StepArgs a) {
    // the env-step arithmetic around the solver (energy, reward, flags) has no contraction: same roundings in every
    // instantiation, and the same as NumPy's
#pragma clang fp contract(off)
    static_assert(!PC || THERMAL, "wave specialisation only exists for the thermal kernels");
    __shared__ double s_tab[MULTI ? STG_MAX_CLASSES * C_COUNT : 1];
    // normals rings of the wave-specialised kernels (one per integrating wavefront): RK45 hands over finished fields
    // (double), the fixed-step solvers raw normals (float)
    constexpr bool FIELD = SOLVER == STG_SOLVER_RK45;
    using NT = typename std::conditional<FIELD, double, float>::type;
    constexpr int RING = 2 * SHARED_CHUNK_MAX * 64;
    __shared__ NT s_norm[PC ? WGW * RING : 1];
    __shared__ int s_alive[2 * WGW], s_go[2 * WGW];
    __shared__ unsigned long long s_cnt[WGW * 3];
    __shared__ uint32_t s_rng[PC ? WGW * 64 : 1];
    const int lane = (int)(threadIdx.x & 63u);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool producer = PC && wave >= WGW;
    const int cw = producer ? (2 * WGW - 1 - wave) : wave;          // the integrating wavefront this one is, or serves
    const int64_t lane_slot = stg_slot_block<WGW>(blockIdx.x, gridDim.x, a.perm != nullptr, cw, PC) * 64 + lane;
    const bool live = lane_slot < a.N;
    // duration-sorted schedule: slot j of the launch integrates env perm[j], so the 64 lanes of a wavefront have
    // (nearly) equal trip counts; all state and outputs stay at the env's own index
    const int64_t i = live ? (a.perm ? (int64_t)a.perm[lane_slot] : lane_slot) : 0;
    const double* row = class_row<MULTI>(a.ctab, a.cls, a.ncls, i, live, s_tab, a.ep, a.N);
    // Lanes without an env: the one-wavefront form has no rendezvous after this point and lets them go; in the
    // wave-specialised form they stay (inert) because every wavefront of the workgroup takes part in every s_barrier.
    if (!PC && !live) return;
    const int64_t N = a.N;
    const uint64_t env_id = (uint64_t)(a.env_id0 + i);

    if (producer) {
        // per env-step: wait for the stream positions (H1), then stay one chunk ahead of integrating wavefront `cw`.
        // Normals per chunk: RK45 6 (initial step) then 18 per attempt; RK4 12 per sub-step; Euler and the
        // Ornstein-Uhlenbeck field 3 per sub-step
        constexpr int n_first = SOLVER == STG_SOLVER_RK45 ? 6 : (SOLVER == STG_SOLVER_RK4 ? 12 : 3);
        constexpr int n_chunk = SOLVER == STG_SOLVER_RK45 ? 18 : n_first;
        const bool ou = a.c.inv_tau > 0.0;
        const double ghs = FIELD ? load_llgs(row).ghs : 0.0;
        for (int k = 0; k < a.K; ++k) {
            __syncthreads();                                       // H1: s_rng / s_go[k & 1] published
            const int p = (k & 1) * WGW;
            if (!any_flag<WGW>(s_go + p)) continue;
            const bool serve = s_go[p + cw] != 0;
            const RngKey rk{a.c.seed, env_id, s_rng[cw * 64 + lane]};
            if (SOLVER == STG_SOLVER_RK4 && ou) produce_normals<NT, FIELD, WGW>(s_norm + cw * RING, s_alive, cw, lane, rk, 3, 3, ghs, serve);
            else produce_normals<NT, FIELD, WGW>(s_norm + cw * RING, s_alive, cw, lane, rk, n_first, n_chunk, ghs, serve);
        }
        return;
    }

    V3 m{a.s.mx[i], a.s.my[i], a.s.mz[i]};          // (lanes without an env read env 0 and never write)
    V3 tgt{a.s.tx[i], a.s.ty[i], a.s.tz[i]};
    double etot = a.s.etot[i];
    int32_t step = a.s.step[i];
    uint32_t rng = a.s.rng[i];
    bool done = a.s.done[i] != 0;
    const AT* act = (const AT*)a.actions;
    const Recorder norec{};
    unsigned long long c_steps = 0, c_sub = 0, c_noop = 0;

    for (int k = 0; k < a.K; ++k) {
        double J, T;
        parse_action<AT>(act[((int64_t)k * 2 + 0) * N + i], act[((int64_t)k * 2 + 1) * N + i], a.c.max_current,
                         a.c.max_duration, J, T);
        const bool last = (k == a.K - 1);
        const int64_t ko = a.out_every ? k : 0;
        const bool wr = (a.out_every || last) && live;
        uint8_t st;
        double reward, energy = 0.0;
        bool is_success, truncated;
        // with skip_done, finished envs are not integrated: a wavefront whose lanes are all done skips the integrator
        const bool lane_solves = live && !(a.c.skip_done && done);
        const RngKey rk{a.c.seed, env_id, rng};
        SolveOut so{m, 0, 0, 0, false};
        if (PC) {
            // H1: this wavefront's stream positions and whether it integrates at all; then, if anybody in the workgroup
            // does, H2 (chunk 0 is in LDS) and the solve, which every integrating wavefront of the workgroup walks chunk
            // for chunk (lanes that do not integrate are inert)
            SharedNormalsT<NT, FIELD, WGW> shared{s_norm + cw * RING, s_alive, cw, lane, 0, 0};
            const int p = (k & 1) * WGW;
            s_rng[cw * 64 + lane] = rng;
            const bool mine = __ballot(lane_solves) != 0ull;
            s_go[p + cw] = mine ? 1 : 0;
            __syncthreads();                                       // H1
            if ((WGW == 1) ? mine : any_flag<WGW>(s_go + p)) {
                __syncthreads();                                   // H2
                so = run_solver<SOLVER, THERMAL, false, AXIS_Z, DEVPHYS>(m, J, T, row, a.c, rk, norec, shared, lane_solves);
            }
        }
        if (!lane_solves) {
            st = STG_STATUS_INACTIVE; reward = 0.0;
            is_success = dot(m, tgt) >= a.c.thr;
            truncated = step >= a.c.max_steps;
        } else {
            const double prev_align = dot(m, tgt);                                   // spin_torque_env.py:338-339
            if (!PC) {
                InlineNormals inl;
                so = run_solver<SOLVER, THERMAL, false, AXIS_Z, DEVPHYS>(m, J, T, row, a.c, rk, norec, inl, true);
            }
            if (fabs(J) > 1e-12) {                                                   // spin_torque_env.py:474-480
                const V3 ref{row[C_REFX], row[C_REFY], row[C_REFZ]};
                const double r = resistance(m, (int)row[C_DEVTYPE], row[C_RP], row[C_RAP], row[C_TMR], ref, row[C_RSERIES]);
                const double v = J * r * row[C_AREA];
                energy = (v * v) / r * T;
            }
            if (so.ok) {                                                             // spin_torque_env.py:461-467
                const double inv = rsqrt_fast(dot(so.m, so.m));
                m = V3{so.m.x * inv, so.m.y * inv, so.m.z * inv};
            }
            etot += energy;
            step += 1;
            rng += 1;
            const double align = dot(m, tgt);                                        // spin_torque_env.py:350-353
            is_success = align >= a.c.thr;
            // default reward (spin_torque_env.py:184-207; rewards/composite_reward.py:65-126), H7 sign kept
            reward = 10.0 * (is_success ? 10.0 : 0.0);
            reward += (-a.c.w_energy) * (-energy / 1e-12);
            reward += (align - prev_align);
            if (isnan(reward) || isinf(reward)) reward = -1.0;                       // monitoring.py:332-348
            reward = fmin(fmax(reward, -1e6), 1e6);
            truncated = step >= a.c.max_steps;                                       // spin_torque_env.py:371-372
            st = so.ok ? (so.resets > 0 ? STG_STATUS_RESET : STG_STATUS_OK) : STG_STATUS_NOOP;
            c_steps += 1; c_sub += (unsigned long long)so.work; c_noop += so.ok ? 0 : 1;
            done = is_success || truncated;
        }
        // same-step auto-reset: the finished episode's reward/flags go out with this step, the state is redrawn on the
        // device and the observation handed to the agent is the NEW episode's first one (the terminal observation goes
        // to final_obs when the caller asked for it)
        const bool do_reset = a.autoreset && done && live;
        if (wr && do_reset && a.final_obs) write_obs(a.final_obs + ko * 12 * N, N, i, m, tgt, row, a.c, step, etot, J, T);
        if (do_reset) {
            device_reset_draw(a.c.seed, env_id, rng, a.c, true, true, m, tgt);
            etot = 0.0; step = 0; done = false;
            J = 0.0; T = 0.0;                       // last_action = zeros after reset (spin_torque_env.py:283)
        }
        if (wr) {
            write_obs(a.obs + ko * 12 * N, N, i, m, tgt, row, a.c, step, etot, J, T);
            a.reward[ko * N + i] = (float)reward;
            if (a.reward64) a.reward64[ko * N + i] = reward;
            if (a.energy) a.energy[ko * N + i] = energy;
            a.term[ko * N + i] = is_success ? 1 : 0;
            a.trunc[ko * N + i] = truncated ? 1 : 0;
            if (a.status) a.status[ko * N + i] = st;
        }
    }
    if (live) {
        a.s.mx[i] = m.x; a.s.my[i] = m.y; a.s.mz[i] = m.z;
        a.s.tx[i] = tgt.x; a.s.ty[i] = tgt.y; a.s.tz[i] = tgt.z;
        a.s.etot[i] = etot;
        a.s.step[i] = step;
        a.s.rng[i] = rng;
        a.s.done[i] = done ? 1 : 0;
    }
    // on-device metrics (the reference's EnvironmentMonitor/solver stats are host-side bookkeeping): one atomic per
    // counter per wavefront, into one of COUNTER_STRIPES copies
    wave_add3(a.counters + (size_t)((blockIdx.x * WGW + cw) % COUNTER_STRIPES) * COUNTER_STRIDE, s_cnt + cw * 3, c_steps, c_sub, c_noop);
}

// ------------------------------------------------------------------------------------------------
// solver-level seam
// ------------------------------------------------------------------------------------------------
template <int SOLVER, bool THERMAL, bool MULTI, bool RECORD>
__global__ void __launch_bounds__(64) stg_solve_kernel(const SolveArgs a) {
    __shared__ double s_tab[MULTI ? STG_MAX_CLASSES * C_COUNT : 1];
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool in_range = i < a.N;
    const double* row = class_row<MULTI>(a.ctab, a.cls, a.ncls, i, in_range, s_tab, EnvParams{}, a.N);
    if (!in_range) return;
    const int64_t N = a.N;
    const V3 m0{a.m0[i], a.m0[N + i], a.m0[2 * N + i]};
    const RngKey rk{a.c.seed, (uint64_t)(a.env_id0 + i), a.env_step};
    const Recorder rec{a.traj_t, a.traj_m, a.traj_e, N, i, a.traj_cap};
    InlineNormals ns;
    const SolveOut so = run_solver<SOLVER, THERMAL, RECORD, false, false>(m0, a.J[i], a.T[i], row, a.c, rk, rec, ns, true);
    a.m_final[i] = so.m.x; a.m_final[N + i] = so.m.y; a.m_final[2 * N + i] = so.m.z;
    if (a.n_points) a.n_points[i] = so.n;
    if (a.success) a.success[i] = so.ok ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// reset / state / diagnostics kernels
// ------------------------------------------------------------------------------------------------
struct ResetArgs {
    StateView s;
    CfgView c;
    int64_t N, env_id0;
    const double* ctab;
    const uint8_t* cls;
    int32_t ncls;
    const uint8_t* mask;
    const double *init_m, *target;
    uint64_t seed;
    float* obs;
    EnvParams ep;
};

__global__ void __launch_bounds__(64) stg_reset_kernel(const ResetArgs a) {
    __shared__ double s_tab[STG_MAX_CLASSES * C_COUNT];
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool in_range = i < a.N;
    const double* row = (a.ncls > 1 || a.ep.soa) ? class_row<true>(a.ctab, a.cls, a.ncls, i, in_range, s_tab, a.ep, a.N) : a.ctab;
    if (!in_range) return;
    const int64_t N = a.N;
    if (a.mask && !a.mask[i]) {
        if (a.obs) {   // unchanged env: report its current observation (last action unknown -> 0, as after reset)
            const V3 m{a.s.mx[i], a.s.my[i], a.s.mz[i]}, t{a.s.tx[i], a.s.ty[i], a.s.tz[i]};
            write_obs(a.obs, N, i, m, t, row, a.c, a.s.step[i], a.s.etot[i], 0.0, 0.0);
        }
        return;
    }
    V3 m{0, 0, 1}, t{0, 0, 1};
    device_reset_draw(a.seed, (uint64_t)(a.env_id0 + i), a.s.rng[i], a.c, a.init_m == nullptr, a.target == nullptr, m, t);
    if (a.init_m) {   // device.validate_magnetization (base_device.py:94-116): v / |v|
        const V3 v{a.init_m[i], a.init_m[N + i], a.init_m[2 * N + i]};
        const double n = sqrt(dot(v, v));
        m = V3{v.x / n, v.y / n, v.z / n};
    }
    if (a.target) {
        const V3 v{a.target[i], a.target[N + i], a.target[2 * N + i]};
        const double n = sqrt(dot(v, v));
        t = V3{v.x / n, v.y / n, v.z / n};
    }
    a.s.mx[i] = m.x; a.s.my[i] = m.y; a.s.mz[i] = m.z;
    a.s.tx[i] = t.x; a.s.ty[i] = t.y; a.s.tz[i] = t.z;
    a.s.etot[i] = 0.0;
    a.s.step[i] = 0;
    a.s.done[i] = 0;
    if (a.obs) write_obs(a.obs, N, i, m, t, row, a.c, 0, 0.0, 0.0, 0.0);
}

__global__ void stg_normals_kernel(uint64_t seed, int64_t env_id0, int64_t N, uint32_t env_step, uint32_t call0,
                                   int32_t n_calls, double* z) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    NormalStream ns;
    ns.init(seed, (uint64_t)(env_id0 + i), env_step, 0u);
    for (uint32_t c = 0; c < call0 + (uint32_t)n_calls; ++c) {       // the stream is sequential: replay from call 0
        const V3 v = (c & 1u) ? ns.draw3_odd() : ns.draw3_even();
        if (c >= call0) {
            double* b = z + (int64_t)(c - call0) * 3 * N + i;
            b[0] = v.x; b[N] = v.y; b[2 * N] = v.z;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// lane schedule: tile-local counting sort of the envs by this step's integration work (descending)
// ------------------------------------------------------------------------------------------------
// The trip count of a lane is set by its pulse duration (RK4: n = T/dt sub-steps; RK45: attempts ~ T / 0.65 ps), which
// the agent chooses per env: U[0.1, 1] ns gives a mean/max ratio of 0.55 inside a wavefront.  One kernel sorts each
// TILE of 4096 consecutive envs (= 64 wavefronts) by work in LDS (histogram -> scan -> scatter, no global atomics)
// and writes a tile-local permutation; slot j of the step launch then integrates env perm[j], so the lanes of a
// wavefront have (nearly) equal trip counts while every access of a wavefront stays inside its tile's 32 KB window of
// each SoA row.  The step kernel places a tile's 64 wavefronts on ONE XCD (stg_slot_block), whose L2 then merges the
// tile's scattered 4-8 B accesses into full lines before they reach HBM.

struct PlanArgs {
    const void* actions;      // [2][N] of the first fused step
    int32_t act_f64;
    int64_t N;
    double max_current, max_duration;
    const uint8_t* done;      // with skip_done: finished envs go last (no work)
    int32_t skip_done;
    const uint8_t* cls;       // device-physics torque model with several classes: group lanes by device kind
    const double* ctab;
    const uint8_t* env_type;  // (per-env parameters: the kind of each env)
    int32_t by_kind;
    uint32_t* perm;
};

__device__ __forceinline__ int plan_key(const PlanArgs& a, int64_t i) {
    double J, T;
    if (a.act_f64) parse_action<double>(((const double*)a.actions)[i], ((const double*)a.actions)[a.N + i], a.max_current, a.max_duration, J, T);
    else parse_action<float>(((const float*)a.actions)[i], ((const float*)a.actions)[a.N + i], a.max_current, a.max_duration, J, T);
    if (a.skip_done && a.done[i]) return (a.by_kind ? PLAN_BUCKETS : PLAN_DUR) - 1;
    const double w = fmax(T, 1e-10) / a.max_duration;          // below 0.1 ns the RK4 sub-step count stays at ~100
    int b = (int)(w * (PLAN_DUR - 1));
    b = b < 0 ? 0 : (b > PLAN_DUR - 2 ? PLAN_DUR - 2 : b);
    const int kind = a.by_kind ? (a.env_type ? (int)a.env_type[i] : (int)a.ctab[(int)a.cls[i] * C_COUNT + C_DEVTYPE]) : 0;
    return kind * PLAN_DUR + (PLAN_DUR - 2) - b;               // descending work inside each kind: long pulses first
}

__global__ void __launch_bounds__(PLAN_THREADS) stg_plan_tile_kernel(const PlanArgs a) {
    __shared__ uint32_t cnt[PLAN_BUCKETS], start[PLAN_BUCKETS];
    const int nb = a.by_kind ? PLAN_BUCKETS : PLAN_DUR;       // buckets in use
    const int tid = threadIdx.x;
    if (tid < nb) cnt[tid] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * TILE_ENVS;
    uint32_t rank[PLAN_ITEMS];
    int key[PLAN_ITEMS];
#pragma unroll
    for (int r = 0; r < PLAN_ITEMS; ++r) {
        const int64_t i = base + r * PLAN_THREADS + tid;
        key[r] = -1;
        if (i < a.N) {
            key[r] = plan_key(a, i);
            rank[r] = atomicAdd(&cnt[key[r]], 1u);             // rank inside the bucket (LDS atomic)
        }
    }
    __syncthreads();
    if (tid < nb) start[tid] = cnt[tid];
    __syncthreads();
    for (int off = 1; off < nb; off <<= 1) {                   // Hillis-Steele inclusive scan over the buckets
        uint32_t add = 0;
        if (tid < nb && tid >= off) add = start[tid - off];
        __syncthreads();
        if (tid < nb) start[tid] += add;
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < PLAN_ITEMS; ++r) {
        const int64_t i = base + r * PLAN_THREADS + tid;
        if (key[r] >= 0) a.perm[base + (start[key[r]] - cnt[key[r]]) + rank[r]] = (uint32_t)i;   // exclusive start
    }
}

// device-class formulas evaluated per env (stg_device_terms)
__global__ void stg_device_terms_kernel(int64_t N, const double* ctab, const uint8_t* cls, int32_t ncls, const double* m,
                                        const double* J, const double* volt, double* tau_dl, double* tau_fl, double* k_eff) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int c = (ncls > 1 && cls) ? (int)cls[i] : 0;
    const double* row = ctab + (c < ncls ? c : 0) * C_COUNT;
    const int kind = (int)row[C_DEVTYPE];
    if (tau_dl || tau_fl) {
        V3 dl{0.0, 0.0, 0.0}, fl{0.0, 0.0, 0.0};
        if (kind == STG_DEV_SOT) {      // sot_mram.py:186-192: tau_dl_factor*J*cross(sigma, m), tau_fl_factor*J*sigma
            const V3 mm{m[i], m[N + i], m[2 * N + i]}, sg{row[C_SIGX], row[C_SIGY], row[C_SIGZ]};
            const V3 sxm = cross(sg, mm);
            const double a = row[C_SOT_DL] * J[i], b = row[C_SOT_FL] * J[i];
            dl = V3{a * sxm.x, a * sxm.y, a * sxm.z};
            fl = V3{b * sg.x, b * sg.y, b * sg.z};
        }
        if (tau_dl) { tau_dl[i] = dl.x; tau_dl[N + i] = dl.y; tau_dl[2 * N + i] = dl.z; }
        if (tau_fl) { tau_fl[i] = fl.x; tau_fl[N + i] = fl.y; tau_fl[2 * N + i] = fl.z; }
    }
    if (k_eff)
        k_eff[i] = (kind == STG_DEV_VCMA) ? vcma_keff(volt[i], row[C_KU], row[C_VCMA_XI], row[C_VCMA_TD2], row[C_VCMA_VBD]) : row[C_KU];
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(STG_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                  \
    } while (0)

struct stg_ctx {
    int device;
    int64_t N, env_id0;
    stg_config cfg;
    StateView s{};
    void* slab = nullptr;
    double* ctab = nullptr;           // device, [ncls][C_COUNT]
    double h_ctab[STG_MAX_CLASSES][C_COUNT];
    int32_t ncls = 0;
    const uint8_t* cls = nullptr;     // caller-owned device pointer
    unsigned long long* counters = nullptr;
    uint32_t* perm = nullptr;
    bool have_params = false, have_state = false;
    bool axis_z = false;              // every class has easy axis = +z exactly: the specialised Simple RHS applies
    bool axis_z_llgs = false;         // every class has raw easy axis (0,0,rz) and demag (0,0,Nz): specialised LLGS RHS
    // per-env parameters (stg_set_params_per_env): library-owned copies
    double* env_soa = nullptr;        // [STG_NPARAM][N]
    uint8_t *env_type = nullptr, *env_valid = nullptr;
    bool per_env = false;
};

static EnvParams env_params_of(const stg_ctx* ctx) {
    EnvParams e{};
    if (ctx->per_env) { e.soa = ctx->env_soa; e.type = ctx->env_type; e.valid = ctx->env_valid; }
    e.gamma = ctx->cfg.gamma; e.temperature = ctx->cfg.temperature;
    return e;
}

static CfgView cfg_view(const stg_config& c) {
    CfgView v{};
    v.temperature = c.temperature; v.max_step = c.max_step; v.rtol = c.rtol; v.atol = c.atol;
    v.max_current = c.max_current; v.max_duration = c.max_duration; v.thr = c.success_threshold;
    v.w_energy = c.energy_penalty_weight;
    v.inv_tau = (c.noise_model == 1 && c.solver != STG_SOLVER_RK45) ? 1.0 / c.noise_corr_time : 0.0;
    v.temp_norm = c.temperature / 300.0; v.inv_max_current = 1.0 / c.max_current; v.inv_max_duration = 1.0 / c.max_duration;
    std::memcpy(v.targets, c.targets, sizeof(v.targets));
    v.seed = c.seed; v.max_attempts = c.max_attempts; v.max_steps = c.max_steps; v.n_targets = c.n_targets;
    v.skip_done = c.skip_done;
    return v;
}

static int check_cfg(const stg_config* c) {
    if (!c) return fail(STG_E_INVALID, "cfg is NULL");
    if (c->solver < 0 || c->solver > 2) return fail(STG_E_INVALID, "cfg.solver must be STG_SOLVER_RK4/EULER/RK45");
    if (c->n_targets < 1 || c->n_targets > STG_MAX_TARGETS) return fail(STG_E_INVALID, "cfg.n_targets out of range");
    if (!(c->max_step > 0)) return fail(STG_E_INVALID, "cfg.max_step must be positive");
    if (c->max_steps < 1) return fail(STG_E_INVALID, "cfg.max_steps must be >= 1");
    if (!(c->max_current > 0) || !(c->max_duration > 0)) return fail(STG_E_INVALID, "cfg.max_current/max_duration must be positive");
    if (c->solver == STG_SOLVER_RK45 && (!(c->rtol > 0) || !(c->atol >= 0))) return fail(STG_E_INVALID, "cfg.rtol/atol invalid");
    if (c->noise_model != 0 && c->noise_model != 1) return fail(STG_E_INVALID, "cfg.noise_model must be 0 (white) or 1 (Ornstein-Uhlenbeck)");
    if (c->noise_model == 1 && c->solver == STG_SOLVER_RK45) return fail(STG_E_INVALID, "cfg.noise_model = 1 needs a fixed-step solver (RK4 or Euler)");
    if (c->noise_model == 1 && !(c->noise_corr_time > 0)) return fail(STG_E_INVALID, "cfg.noise_corr_time must be positive");
    if (c->solver == STG_SOLVER_RK45 && c->max_attempts < 1) return fail(STG_E_INVALID, "cfg.max_attempts must be >= 1");
    if (c->torque_model < 0 || c->torque_model > 1) return fail(STG_E_INVALID, "cfg.torque_model must be 0 or 1");
    if (c->torque_model == 1 && c->solver == STG_SOLVER_RK45)
        return fail(STG_E_INVALID, "the device-physics torque model is implemented for the fixed-step solvers (rk4, euler)");
    return STG_OK;
}

extern "C" {

int stg_internal_fail(int code, const char* msg) { return fail(code, msg ? msg : ""); }   // used by stg_array.hip
const char* stg_last_error(void) { return g_err.c_str(); }
int stg_abi_version(void) { return STG_ABI_VERSION; }

int stg_create(stg_ctx** out, int device_id, int64_t n_envs, int64_t env_id0, const stg_config* cfg) {
    if (!out) return fail(STG_E_INVALID, "out is NULL");
    *out = nullptr;
    if (n_envs < 1) return fail(STG_E_INVALID, "n_envs must be >= 1");
    if (env_id0 < 0) return fail(STG_E_INVALID, "env_id0 must be >= 0");
    if (int rc = check_cfg(cfg)) return rc;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail(STG_E_INVALID, "device_id out of range");
    HIP_TRY(hipSetDevice(device_id));
    stg_ctx* c = new (std::nothrow) stg_ctx();
    if (!c) return fail(STG_E_NOMEM, "out of host memory");
    c->device = device_id; c->N = n_envs; c->env_id0 = env_id0; c->cfg = *cfg;
    // one slab: 7 f64 rows, then i32, u32, u8 rows (each row 256-B aligned)
    auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
    const size_t N = (size_t)n_envs;
    const size_t r8 = al(N * 8), r4 = al(N * 4), r1 = al(N);
    const size_t total = 7 * r8 + 3 * r4 + r1 + al(sizeof(double) * STG_MAX_CLASSES * C_COUNT) +
                         COUNTER_STRIPES * COUNTER_STRIDE * sizeof(unsigned long long);
    hipError_t e = hipMalloc(&c->slab, total);
    if (e != hipSuccess) { delete c; return fail(STG_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    e = hipMemset(c->slab, 0, total);
    if (e != hipSuccess) { (void)hipFree(c->slab); delete c; return fail(STG_E_HIP, std::string("hipMemset: ") + hipGetErrorString(e)); }
    char* p = (char*)c->slab;
    double** rows[7] = {&c->s.mx, &c->s.my, &c->s.mz, &c->s.tx, &c->s.ty, &c->s.tz, &c->s.etot};
    for (auto r : rows) { *r = (double*)p; p += r8; }
    c->s.step = (int32_t*)p; p += r4;
    c->s.rng = (uint32_t*)p; p += r4;
    c->s.done = (uint8_t*)p; p += r1;
    c->ctab = (double*)p; p += al(sizeof(double) * STG_MAX_CLASSES * C_COUNT);
    c->counters = (unsigned long long*)p; p += COUNTER_STRIPES * COUNTER_STRIDE * sizeof(unsigned long long);
    c->perm = (uint32_t*)p; p += r4;
    *out = c;
    return STG_OK;
}

void stg_destroy(stg_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->slab) (void)hipFree(ctx->slab);
    if (ctx->env_soa) (void)hipFree(ctx->env_soa);
    if (ctx->env_type) (void)hipFree(ctx->env_type);
    delete ctx;
}

static void derive_class(const stg_device_params& p, const stg_config& cfg, double* r) {
    derive_row(p, cfg.gamma, cfg.temperature, r);      // stg_physics.hpp (shared with the per-env device path)
}

int stg_set_params(stg_ctx* ctx, const stg_device_params* table, int32_t n_classes, const uint8_t* cls) {
    if (!ctx || !table) return fail(STG_E_INVALID, "ctx/table is NULL");
    if (n_classes < 1 || n_classes > STG_MAX_CLASSES) return fail(STG_E_INVALID, "n_classes must be in [1, STG_MAX_CLASSES]");
    if (n_classes > 1 && !cls) return fail(STG_E_INVALID, "cls is required when n_classes > 1");
    for (int k = 0; k < n_classes; ++k) {
        const stg_device_params& p = table[k];
        if (p.dev_type < 0 || p.dev_type > 2) return fail(STG_E_INVALID, "dev_type must be STG_DEV_STT/SOT/VCMA");
        // BaseSpintronicDevice/STTMRAMDevice._validate_parameters raise on these at construction
        // (devices/stt_mram.py:46-53); the host mirror raises before getting here.
        if (!(p.volume > 0) || !(p.ms > 0)) return fail(STG_E_INVALID, "volume and saturation_magnetization must be positive");
        derive_class(p, ctx->cfg, ctx->h_ctab[k]);
    }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpy(ctx->ctab, ctx->h_ctab, sizeof(double) * n_classes * C_COUNT, hipMemcpyHostToDevice));
    ctx->ncls = n_classes;
    ctx->axis_z = true;
    for (int k = 0; k < n_classes; ++k)
        ctx->axis_z = ctx->axis_z && ctx->h_ctab[k][C_EX] == 0.0 && ctx->h_ctab[k][C_EY] == 0.0 && ctx->h_ctab[k][C_EZ] == 1.0;
    ctx->axis_z_llgs = true;
    for (int k = 0; k < n_classes; ++k) {
        const double* r = ctx->h_ctab[k];
        ctx->axis_z_llgs = ctx->axis_z_llgs && r[C_RX] == 0.0 && r[C_RY] == 0.0 && r[C_DX] == 0.0 && r[C_DY] == 0.0;
    }
    ctx->cls = n_classes > 1 ? cls : nullptr;
    ctx->per_env = false;
    ctx->have_params = true;
    return STG_OK;
}

// axis flags of a per-env parameter block: flag[0] = some env's easy axis is not exactly +z after normalisation,
// flag[1] = some env's raw axis or demag factors have x/y components (LLGS specialisation)
__global__ void stg_env_axis_kernel(const double* soa, int64_t N, int32_t* flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double ex = soa[5 * N + i], ey = soa[6 * N + i], ez = soa[7 * N + i];
    const double dx = soa[8 * N + i], dy = soa[9 * N + i];
    const double en = sqrt((ex * ex + ey * ey) + ez * ez);
    if (!(ex / en == 0.0 && ey / en == 0.0 && ez / en == 1.0)) flag[0] = 1;
    if (!(ex == 0.0 && ey == 0.0 && -dx == 0.0 && -dy == 0.0)) flag[1] = 1;
}

int stg_set_params_per_env(stg_ctx* ctx, const double* soa_params, const uint8_t* dev_type, const uint8_t* params_valid) {
    if (!ctx || !soa_params || !dev_type || !params_valid) return fail(STG_E_INVALID, "ctx/soa_params/dev_type/params_valid is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t N = (size_t)ctx->N;
    if (!ctx->env_soa) {
        HIP_TRY(hipMalloc(&ctx->env_soa, sizeof(double) * STG_NPARAM * N));
        HIP_TRY(hipMalloc(&ctx->env_type, 2 * N));
        ctx->env_valid = ctx->env_type + N;
    }
    HIP_TRY(hipMemcpy(ctx->env_soa, soa_params, sizeof(double) * STG_NPARAM * N, hipMemcpyDeviceToDevice));
    HIP_TRY(hipMemcpy(ctx->env_type, dev_type, N, hipMemcpyDeviceToDevice));
    HIP_TRY(hipMemcpy(ctx->env_valid, params_valid, N, hipMemcpyDeviceToDevice));
    // the specialised right-hand sides apply only if EVERY env has the default axis geometry
    int32_t h_flag[2] = {0, 0};
    int32_t* d_flag = nullptr;
    HIP_TRY(hipMalloc(&d_flag, 8));
    HIP_TRY(hipMemset(d_flag, 0, 8));
    hipLaunchKernelGGL(stg_env_axis_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, 0, ctx->env_soa, (int64_t)N, d_flag);
    HIP_TRY(hipMemcpy(h_flag, d_flag, 8, hipMemcpyDeviceToHost));
    (void)hipFree(d_flag);
    ctx->axis_z = h_flag[0] == 0;
    ctx->axis_z_llgs = h_flag[1] == 0;
    ctx->ncls = 0; ctx->cls = nullptr;
    ctx->per_env = true;
    ctx->have_params = true;
    return STG_OK;
}

int stg_thermal_strength(stg_ctx* ctx, int32_t cls, double* out) {
    if (!ctx || !out) return fail(STG_E_INVALID, "ctx/out is NULL");
    if (!ctx->have_params) return fail(STG_E_STATE, "stg_set_params has not been called");
    if (ctx->per_env) return fail(STG_E_STATE, "per-env parameters have no class table");
    if (cls < 0 || cls >= ctx->ncls) return fail(STG_E_INVALID, "cls out of range");
    *out = ctx->h_ctab[cls][ctx->cfg.solver == STG_SOLVER_RK45 ? C_HS_LLGS : C_HS_SIMPLE];
    return STG_OK;
}

// largest launch the automatic wave specialisation applies to: one integrating wavefront per SIMD (256 CUs x 4 SIMDs x
// 64 lanes); beyond that the launch is throughput-bound and the rendezvous costs more than it gives
constexpr int64_t STG_WAVE_SPEC_MAX_ENVS = 65536;

static inline dim3 grid_for(int64_t N) { return dim3((unsigned)((N + 63) / 64)); }

int stg_reset(stg_ctx* ctx, const uint8_t* mask, const double* init_m, const double* target, uint64_t seed,
              float* obs_out, void* stream) {
    if (!ctx) return fail(STG_E_INVALID, "ctx is NULL");
    if (!ctx->have_params) return fail(STG_E_STATE, "stg_set_params must be called before stg_reset");
    HIP_TRY(hipSetDevice(ctx->device));
    ResetArgs a{};
    a.s = ctx->s; a.c = cfg_view(ctx->cfg); a.N = ctx->N; a.env_id0 = ctx->env_id0;
    a.ctab = ctx->ctab; a.cls = ctx->cls; a.ncls = ctx->ncls; a.ep = env_params_of(ctx);
    a.mask = mask; a.init_m = init_m; a.target = target; a.seed = seed; a.obs = obs_out;
    hipLaunchKernelGGL(stg_reset_kernel, grid_for(ctx->N), dim3(64), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    ctx->have_state = true;
    return STG_OK;
}

extern "C++" {
constexpr int64_t STG_WG4_MIN_ENVS = 65536;       // 256 CUs x 4 SIMDs x 64 lanes

template <int SOLVER, bool THERMAL, bool MULTI, bool AXIS_Z, bool DEVPHYS, int WGW>
static void launch_step_w(const StepArgs& a, int act_f64, bool pc, hipStream_t st) {
    const dim3 grid((unsigned)((a.N + WGW * 64 - 1) / (WGW * 64)));
    if (act_f64)
        hipLaunchKernelGGL((stg_step_kernel<SOLVER, THERMAL, MULTI, AXIS_Z, DEVPHYS, double, false, WGW>), grid, dim3(WGW * 64), 0, st, a);
    else
        hipLaunchKernelGGL((stg_step_kernel<SOLVER, THERMAL, MULTI, AXIS_Z, DEVPHYS, float, false, WGW>), grid, dim3(WGW * 64), 0, st, a);
}

template <int SOLVER, bool THERMAL, bool MULTI, bool AXIS_Z, bool DEVPHYS>
static void launch_step(const StepArgs& a, int act_f64, bool pc, hipStream_t st) {
    if (THERMAL && !DEVPHYS && pc) {
        // wave-specialised variant: one integrating + one producing wavefront per workgroup; not built for the
        // device-physics model
        constexpr bool PC = THERMAL && !DEVPHYS;
        const dim3 grid((unsigned)((a.N + 63) / 64));
        if (act_f64)
            hipLaunchKernelGGL((stg_step_kernel<SOLVER, THERMAL, MULTI, AXIS_Z, DEVPHYS, double, PC, 1>), grid, dim3(128), 0, st, a);
        else
            hipLaunchKernelGGL((stg_step_kernel<SOLVER, THERMAL, MULTI, AXIS_Z, DEVPHYS, float, PC, 1>), grid, dim3(128), 0, st, a);
        return;
    }
    // workgroups of 4 integrating wavefronts once there is one per CU, of 1 below that
    if (a.N >= STG_WG4_MIN_ENVS && !a.force_wg1) launch_step_w<SOLVER, THERMAL, MULTI, AXIS_Z, DEVPHYS, 4>(a, act_f64, pc, st);
    else launch_step_w<SOLVER, THERMAL, MULTI, AXIS_Z, DEVPHYS, 1>(a, act_f64, pc, st);
}
template <int SOLVER, bool AXIS_Z, bool DEVPHYS>
static void dispatch_step2(const StepArgs& a, bool thermal, bool multi, int act_f64, bool pc, hipStream_t st) {
    if (thermal) { if (multi) launch_step<SOLVER, true, true, AXIS_Z, DEVPHYS>(a, act_f64, pc, st); else launch_step<SOLVER, true, false, AXIS_Z, DEVPHYS>(a, act_f64, pc, st); }
    else { if (multi) launch_step<SOLVER, false, true, AXIS_Z, DEVPHYS>(a, act_f64, false, st); else launch_step<SOLVER, false, false, AXIS_Z, DEVPHYS>(a, act_f64, false, st); }
}
// axis_z selects the easy-axis = z specialisation of the RHS (Simple: e = +z; LLGS: raw axis and demag along z);
// devphys the opt-in device-physics torque model (fixed-step solvers only)
template <int SOLVER>
static void dispatch_step(const StepArgs& a, bool thermal, bool multi, bool axis_z, bool devphys, int act_f64, bool pc, hipStream_t st) {
    if (SOLVER != STG_SOLVER_RK45 && devphys) {
        if (axis_z) dispatch_step2<SOLVER, true, true>(a, thermal, multi, act_f64, pc, st);
        else dispatch_step2<SOLVER, false, true>(a, thermal, multi, act_f64, pc, st);
    } else {
        if (axis_z) dispatch_step2<SOLVER, true, false>(a, thermal, multi, act_f64, pc, st);
        else dispatch_step2<SOLVER, false, false>(a, thermal, multi, act_f64, pc, st);
    }
}
}  // extern "C++"

int stg_step_many(stg_ctx* ctx, int32_t K, const void* actions, int32_t act_f64, int32_t out_every, int32_t autoreset,
                  float* obs, float* final_obs, float* reward, double* reward_f64, double* energy, uint8_t* terminated,
                  uint8_t* truncated, uint8_t* status, void* stream) {
    if (!ctx) return fail(STG_E_INVALID, "ctx is NULL");
    if (!ctx->have_params || !ctx->have_state) return fail(STG_E_STATE, "stg_set_params and stg_reset must precede stg_step");
    if (K < 1) return fail(STG_E_INVALID, "K must be >= 1");
    if (!actions || !obs || !reward || !terminated || !truncated) return fail(STG_E_INVALID, "actions/obs/reward/terminated/truncated must not be NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    StepArgs a{};
    a.s = ctx->s; a.c = cfg_view(ctx->cfg); a.N = ctx->N; a.env_id0 = ctx->env_id0;
    a.ctab = ctx->ctab; a.cls = ctx->cls; a.ncls = ctx->ncls; a.ep = env_params_of(ctx);
    a.counters = ctx->counters;
    hipStream_t st = (hipStream_t)stream;
    a.perm = nullptr;
    // lane_sort: 0 = automatic (on: the single LDS-only plan kernel costs ~5 us and the sorted schedule is never slower
    // once there is more than one wavefront), 1 = always, -1 = never (identity schedule)
    const bool want_sort = ctx->cfg.lane_sort >= 0;
    if (want_sort && ctx->N > 64) {
        if (ctx->N > 0xFFFFFFFFll) return fail(STG_E_INVALID, "lane sort supports up to 2^32 envs per context");
        PlanArgs pa{};
        pa.actions = actions; pa.act_f64 = act_f64; pa.N = ctx->N;
        pa.max_current = ctx->cfg.max_current; pa.max_duration = ctx->cfg.max_duration;
        pa.done = ctx->s.done; pa.skip_done = (ctx->cfg.skip_done && !autoreset) ? 1 : 0;
        pa.perm = ctx->perm;
        pa.cls = ctx->cls; pa.ctab = ctx->ctab;
        pa.env_type = ctx->per_env ? ctx->env_type : nullptr;
        pa.by_kind = (ctx->cfg.torque_model == 1 && ((ctx->ncls > 1 && ctx->cls) || ctx->per_env)) ? 1 : 0;
        const dim3 g((unsigned)((ctx->N + TILE_ENVS - 1) / TILE_ENVS));
        hipLaunchKernelGGL(stg_plan_tile_kernel, g, dim3(PLAN_THREADS), 0, st, pa);
        a.perm = ctx->perm;
    }
    a.actions = actions; a.K = K; a.out_every = out_every ? 1 : 0; a.autoreset = autoreset ? 1 : 0;
    a.obs = obs; a.final_obs = final_obs; a.reward = reward; a.reward64 = reward_f64; a.energy = energy; a.term = terminated; a.trunc = truncated; a.status = status;
    // the Simple solver only draws a thermal field when temperature > 0 (simple_solver.py:321,378)
    const bool thermal = ctx->cfg.thermal && ctx->cfg.temperature > 0;
    const bool multi = ctx->ncls > 1 || ctx->per_env;
    const bool devphys = ctx->cfg.torque_model == 1;
    a.force_wg1 = ctx->per_env ? 1 : 0;       // per-env rows fill the 64-row LDS block: 64 integrating lanes per workgroup
    // wave_spec: 0 = automatic (thermal launches of at most STG_WAVE_SPEC_MAX_ENVS envs, i.e. latency-bound ones),
    // 1 = always, -1 = never.  Results do not depend on it.
    const bool pc = ctx->cfg.wave_spec > 0 || (ctx->cfg.wave_spec == 0 && ctx->N <= STG_WAVE_SPEC_MAX_ENVS);
    switch (ctx->cfg.solver) {
        case STG_SOLVER_RK4: dispatch_step<STG_SOLVER_RK4>(a, thermal, multi, ctx->axis_z, devphys, act_f64, pc, st); break;
        case STG_SOLVER_EULER: dispatch_step<STG_SOLVER_EULER>(a, thermal, multi, ctx->axis_z, devphys, act_f64, pc, st); break;
        default: dispatch_step<STG_SOLVER_RK45>(a, ctx->cfg.thermal != 0, multi, ctx->axis_z_llgs, false, act_f64, pc, st); break;
    }
    HIP_TRY(hipGetLastError());
    return STG_OK;
}

int stg_step(stg_ctx* ctx, const void* actions, int32_t act_f64, float* obs, float* reward, double* reward_f64,
             double* energy, uint8_t* terminated, uint8_t* truncated, uint8_t* status, void* stream) {
    return stg_step_many(ctx, 1, actions, act_f64, 1, 0, obs, nullptr, reward, reward_f64, energy, terminated, truncated,
                         status, stream);
}

extern "C++" {
template <int SOLVER, bool RECORD>
static void dispatch_solve(const SolveArgs& a, bool thermal, bool multi, hipStream_t st) {
    const dim3 g = grid_for(a.N), b(64);
    if (thermal) {
        if (multi) hipLaunchKernelGGL((stg_solve_kernel<SOLVER, true, true, RECORD>), g, b, 0, st, a);
        else hipLaunchKernelGGL((stg_solve_kernel<SOLVER, true, false, RECORD>), g, b, 0, st, a);
    } else {
        if (multi) hipLaunchKernelGGL((stg_solve_kernel<SOLVER, false, true, RECORD>), g, b, 0, st, a);
        else hipLaunchKernelGGL((stg_solve_kernel<SOLVER, false, false, RECORD>), g, b, 0, st, a);
    }
}
}  // extern "C++"

static int solve_common(stg_ctx* ctx, const double* m0, const double* J, const double* T, uint32_t env_step,
                        int32_t traj_cap, double* tt, double* tm, double* te, double* m_final, int32_t* n_points,
                        uint8_t* success, void* stream, bool record) {
    if (!ctx) return fail(STG_E_INVALID, "ctx is NULL");
    if (!ctx->have_params) return fail(STG_E_STATE, "stg_set_params must precede stg_solve");
    if (ctx->per_env) return fail(STG_E_STATE, "stg_solve* works on the class table (stg_set_params), not on per-env parameters");
    if (!m0 || !J || !T || !m_final) return fail(STG_E_INVALID, "m0/J/T/m_final must not be NULL");
    if (record && traj_cap < 1) return fail(STG_E_INVALID, "traj_cap must be >= 1");
    HIP_TRY(hipSetDevice(ctx->device));
    SolveArgs a{};
    a.c = cfg_view(ctx->cfg); a.N = ctx->N; a.env_id0 = ctx->env_id0;
    a.ctab = ctx->ctab; a.cls = ctx->cls; a.ncls = ctx->ncls;
    a.m0 = m0; a.J = J; a.T = T; a.env_step = env_step; a.m_final = m_final; a.n_points = n_points; a.success = success;
    a.traj_cap = traj_cap; a.traj_t = tt; a.traj_m = tm; a.traj_e = te;
    const bool multi = ctx->ncls > 1;
    hipStream_t st = (hipStream_t)stream;
    const bool th_simple = ctx->cfg.thermal && ctx->cfg.temperature > 0, th_llgs = ctx->cfg.thermal != 0;
    switch (ctx->cfg.solver) {
        case STG_SOLVER_RK4:
            if (record) dispatch_solve<STG_SOLVER_RK4, true>(a, th_simple, multi, st); else dispatch_solve<STG_SOLVER_RK4, false>(a, th_simple, multi, st);
            break;
        case STG_SOLVER_EULER:
            if (record) dispatch_solve<STG_SOLVER_EULER, true>(a, th_simple, multi, st); else dispatch_solve<STG_SOLVER_EULER, false>(a, th_simple, multi, st);
            break;
        default:
            if (record) dispatch_solve<STG_SOLVER_RK45, true>(a, th_llgs, multi, st); else dispatch_solve<STG_SOLVER_RK45, false>(a, th_llgs, multi, st);
            break;
    }
    HIP_TRY(hipGetLastError());
    return STG_OK;
}

int stg_solve(stg_ctx* ctx, const double* m0, const double* J, const double* T, uint32_t env_step, double* m_final,
              int32_t* n_points, uint8_t* success, void* stream) {
    return solve_common(ctx, m0, J, T, env_step, 0, nullptr, nullptr, nullptr, m_final, n_points, success, stream, false);
}

int stg_solve_traj(stg_ctx* ctx, const double* m0, const double* J, const double* T, uint32_t env_step, int32_t traj_cap,
                   double* t, double* m, double* energy, double* m_final, int32_t* n_points, uint8_t* success,
                   void* stream) {
    return solve_common(ctx, m0, J, T, env_step, traj_cap, t, m, energy, m_final, n_points, success, stream, true);
}

int stg_get_state(stg_ctx* ctx, double* m, double* target, double* total_energy, int32_t* step_count, uint32_t* rng_step,
                  uint8_t* done, void* stream) {
    if (!ctx) return fail(STG_E_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t N = (size_t)ctx->N;
    if (m) {
        HIP_TRY(hipMemcpyAsync(m, ctx->s.mx, N * 8, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(m + N, ctx->s.my, N * 8, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(m + 2 * N, ctx->s.mz, N * 8, hipMemcpyDeviceToDevice, st));
    }
    if (target) {
        HIP_TRY(hipMemcpyAsync(target, ctx->s.tx, N * 8, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(target + N, ctx->s.ty, N * 8, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(target + 2 * N, ctx->s.tz, N * 8, hipMemcpyDeviceToDevice, st));
    }
    if (total_energy) HIP_TRY(hipMemcpyAsync(total_energy, ctx->s.etot, N * 8, hipMemcpyDeviceToDevice, st));
    if (step_count) HIP_TRY(hipMemcpyAsync(step_count, ctx->s.step, N * 4, hipMemcpyDeviceToDevice, st));
    if (rng_step) HIP_TRY(hipMemcpyAsync(rng_step, ctx->s.rng, N * 4, hipMemcpyDeviceToDevice, st));
    if (done) HIP_TRY(hipMemcpyAsync(done, ctx->s.done, N, hipMemcpyDeviceToDevice, st));
    return STG_OK;
}

int stg_set_state(stg_ctx* ctx, const double* m, const double* target, const double* total_energy,
                  const int32_t* step_count, const uint32_t* rng_step, const uint8_t* done, void* stream) {
    if (!ctx) return fail(STG_E_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t N = (size_t)ctx->N;
    if (m) {
        HIP_TRY(hipMemcpyAsync(ctx->s.mx, m, N * 8, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(ctx->s.my, m + N, N * 8, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(ctx->s.mz, m + 2 * N, N * 8, hipMemcpyDeviceToDevice, st));
    }
    if (target) {
        HIP_TRY(hipMemcpyAsync(ctx->s.tx, target, N * 8, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(ctx->s.ty, target + N, N * 8, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(ctx->s.tz, target + 2 * N, N * 8, hipMemcpyDeviceToDevice, st));
    }
    if (total_energy) HIP_TRY(hipMemcpyAsync(ctx->s.etot, total_energy, N * 8, hipMemcpyDeviceToDevice, st));
    if (step_count) HIP_TRY(hipMemcpyAsync(ctx->s.step, step_count, N * 4, hipMemcpyDeviceToDevice, st));
    if (rng_step) HIP_TRY(hipMemcpyAsync(ctx->s.rng, rng_step, N * 4, hipMemcpyDeviceToDevice, st));
    if (done) HIP_TRY(hipMemcpyAsync(ctx->s.done, done, N, hipMemcpyDeviceToDevice, st));
    if (m && target) ctx->have_state = true;
    return STG_OK;
}

int stg_device_terms(stg_ctx* ctx, const double* m, const double* J, const double* volt, double* tau_dl, double* tau_fl,
                     double* k_eff, void* stream) {
    if (!ctx) return fail(STG_E_INVALID, "ctx is NULL");
    if (!ctx->have_params) return fail(STG_E_STATE, "stg_set_params must precede stg_device_terms");
    if (ctx->per_env) return fail(STG_E_STATE, "stg_device_terms works on the class table (stg_set_params), not on per-env parameters");
    if ((tau_dl || tau_fl) && (!m || !J)) return fail(STG_E_INVALID, "m and J are required for the SOT torques");
    if (k_eff && !volt) return fail(STG_E_INVALID, "volt is required for K_eff");
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(stg_device_terms_kernel, dim3((unsigned)((ctx->N + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       ctx->N, ctx->ctab, ctx->cls, ctx->ncls, m, J, volt, tau_dl, tau_fl, k_eff);
    HIP_TRY(hipGetLastError());
    return STG_OK;
}

#ifdef STG_PROFILE_LOOP
int stg_debug_prof(long long* out) {          // experiment builds only
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(stg::g_stg_prof), 16 * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif

int stg_get_counters(stg_ctx* ctx, uint64_t* out, int32_t reset) {
    if (!ctx || !out) return fail(STG_E_INVALID, "ctx/out is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    // the kernels add into COUNTER_STRIPES copies (a workgroup picks one by its index: thousands of wavefronts on one
    // address would serialise in the memory-side atomic unit); the stripes are summed here
    static_assert(sizeof(uint64_t) == sizeof(unsigned long long), "counter width");
    uint64_t h[COUNTER_STRIPES * COUNTER_STRIDE];
    HIP_TRY(hipMemcpy(h, ctx->counters, sizeof(h), hipMemcpyDeviceToHost));
    for (int j = 0; j < 4; ++j) {
        out[j] = 0;
        for (int st = 0; st < COUNTER_STRIPES; ++st) out[j] += h[st * COUNTER_STRIDE + j];
    }
    if (reset) HIP_TRY(hipMemset(ctx->counters, 0, sizeof(h)));
    return STG_OK;
}

int stg_thermal_normals(stg_ctx* ctx, uint32_t env_step, uint32_t call0, int32_t n_calls, double* z, void* stream) {
    if (!ctx || !z) return fail(STG_E_INVALID, "ctx/z is NULL");
    if (n_calls < 1) return fail(STG_E_INVALID, "n_calls must be >= 1");
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(stg_normals_kernel, dim3((unsigned)((ctx->N + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       ctx->cfg.seed, ctx->env_id0, ctx->N, env_step, call0, n_calls, z);
    HIP_TRY(hipGetLastError());
    return STG_OK;
}

}  // extern "C"
