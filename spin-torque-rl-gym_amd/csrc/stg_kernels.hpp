// stg_kernels.hpp -- device side of the env-step path shared by the translation units of libspintorque_hip.so:
// kernel argument blocks, device helpers, the step kernel template and its launch/dispatch templates.  The step kernel
// is instantiated once per solver in stg_step_{rk4,euler,rk45}.hip (so that the three compile in parallel);
// spintorque_hip.hip holds the C-ABI, the other kernels and the host logic.
#pragma once
#include "../../include/spintorque_hip.h"
#include "stg_physics.hpp"

#include <hip/hip_runtime.h>

#include <type_traits>

using namespace stg;

// ------------------------------------------------------------------------------------------------
// kernel argument blocks
// ------------------------------------------------------------------------------------------------
// One env's persistent state is ONE 64-byte record (env index major): its lane loads and stores it as four 16-byte
// accesses.  Under the identity schedule a wavefront touches 4 KB contiguous (as coalesced as rows of a structure of
// arrays); under the duration-sorted schedule, where a lane's env is anywhere in its 4096-env tile, a record is still two
// whole 32-byte sectors -- ten scattered 1-8 B row elements were ten partially written sectors.
struct alignas(16) EnvRec {
    double m[3];
    double tgt[3];
    double etot;
    uint32_t stepw;       // step_count | episode-finished flag in bit 31
    uint32_t rng;         // stream position: env steps taken since creation (Philox counter word)
};
static_assert(sizeof(EnvRec) == 64, "state record size");
constexpr uint32_t STG_DONE_BIT = 0x80000000u;

struct StateView {
    EnvRec* rec;
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

// Per-env parameters (stg_set_params_per_env): ONE record per env, env index major, holding only the fields the context's solver and
// torque model read (the caller hands over all STG_NPARAM rows of a structure of arrays; the library packs once):
//   core, 16 doubles = 128 B (the fixed-step solvers with the reference RHS -- what SpinTorqueEnv runs):
//     damping, ms, ku, volume, polarization, easy_axis[3], area, r_p, r_ap, ref_m[3], r_series, dev_type + 4 * params_valid
//   ENV_LAYOUT_LLGS (RK45), 20 doubles = 160 B: core + demag[3], a_ex
//   ENV_LAYOUT_DEV (fixed-step solvers with the device-physics torque model), 24 doubles = 192 B: core + sot_tau_dl, sot_tau_fl,
//     sot_sigma[3], vcma_xi, vcma_td, vcma_vbd
// (shape_demag is the array env's.)  Rows were the wrong layout for the step kernel for the same reason as for the state (DESIGN.md
// section 2): under the duration-sorted schedule a lane's env is anywhere in its tile, so each of its row elements was an 8-byte read
// out of its own 64-byte request -- measured 1.9 KB of HBM traffic per env-step for 242 B of parameters (profiles/r03e); a record is
// whole 32-byte sectors whatever the schedule.  Rounds 2-3 kept all 30 fields + 2 in a 256-byte record: 1.92 x the algorithmic bytes
// of the per-env cfg4 row (SURVEY 8d prices the parameters at 112 B); the core record makes it 128 B.
enum : int { ENV_LAYOUT_CORE = 0, ENV_LAYOUT_LLGS = 1, ENV_LAYOUT_DEV = 2 };
__host__ __device__ constexpr int env_layout_doubles(int layout) { return layout == ENV_LAYOUT_CORE ? 16 : (layout == ENV_LAYOUT_LLGS ? 20 : 24); }
struct EnvParams {
    const double* soa;        // records [N][env_layout_doubles(layout)] or nullptr (the name is historical: "per-env parameters present")
    double gamma, temperature;
    int32_t layout;           // ENV_LAYOUT_* the records were packed in (= what the context's solver / torque model select)
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
    int32_t walk;                 // tiles of an XCD group in flight together (sorted schedule, see stg_slot_block)
    int32_t records;              // STG_OUT_RECORDS: `obs` is the record array [K or 1][N][STG_RECORD_BYTES], reward/term/trunc unused
    int32_t refill, refill_check; // lane-refill launch (stg_step_refill_kernel): != 0 selects it; attempts between refill points
    int32_t refill_nw;            // ... and its number of (persistent) wavefronts
    unsigned long long* refill_cursor;   // ... the cursors of its global queue's stripes (all 0 when the launch starts)
    unsigned long long* refill_cursor_next;   // ... and the cursors of the NEXT refill launch, which this launch zeroes (two sets alternate)
    int32_t spread_max;           // sorted schedule, 4-wavefront workgroups: up to this many workgroups a workgroup takes ranks u, u+16, u+32,
                                  // u+48 of its tile (spread), beyond it four consecutive ranks (stg_slot_block)
    int32_t hybrid_prio;          // experiment knob (STG_HYB_PRIO=0): hybrid launch with the old numbering of the two-block workgroups
    int32_t hybrid;               // wave-specialised launch of 1024 workgroups over more than 1024 blocks: number of producer/consumer pairs + 1
                                  // (the other workgroups integrate two blocks with the normals inline); 0: every workgroup is a pair.
                                  // See stg_hybrid_block
    const uint32_t* perm;         // lane -> env (duration-sorted schedule) or nullptr
    const void* act_sorted;       // with perm: the first fused step's actions in slot order, [N][2] in their own dtype
    unsigned long long* counters; // [4]: env-steps, integrator sub-steps/attempts, RHS evaluations, no-op steps
    uint32_t* placement;          // this launch's placement table (record_placement) or nullptr
    float* obs;                   // [K or 1][12][N]
    float* final_obs;             // [K or 1][12][N] (records: [K or 1][N][12]) or nullptr: terminal observation of envs auto-reset at that step
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
    double *traj_t, *traj_m, *traj_e, *traj_tq;
};

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_state(const StateView& s, int64_t i, V3& m, V3& tgt, double& etot, int32_t& step,
                                           uint32_t& rng, bool& done) {
    const double2* p = reinterpret_cast<const double2*>(s.rec + i);
    const double2 a = p[0], b = p[1], c = p[2], d = p[3];
    m = V3{a.x, a.y, b.x};
    tgt = V3{b.y, c.x, c.y};
    etot = d.x;
    const unsigned long long w = (unsigned long long)__double_as_longlong(d.y);
    const uint32_t sw = (uint32_t)w;
    step = (int32_t)(sw & ~STG_DONE_BIT);
    done = (sw & STG_DONE_BIT) != 0u;
    rng = (uint32_t)(w >> 32);
}
__device__ __forceinline__ void store_state(const StateView& s, int64_t i, const V3& m, const V3& tgt, double etot, int32_t step,
                                            uint32_t rng, bool done) {
    double2* p = reinterpret_cast<double2*>(s.rec + i);
    const unsigned long long w = (unsigned long long)((uint32_t)step | (done ? STG_DONE_BIT : 0u)) | ((unsigned long long)rng << 32);
    p[0] = make_double2(m.x, m.y);
    p[1] = make_double2(m.z, tgt.x);
    p[2] = make_double2(tgt.y, tgt.z);
    p[3] = make_double2(etot, __longlong_as_double((long long)w));
}

// One env's record -> the reference-style parameter struct (fields that are not in the layout get the defaults of
// devices.flatten_params; the derived constants built from them belong to rows this kernel never reads).
template <int LAYOUT>
__device__ __forceinline__ void load_env_params(const EnvParams& e, int64_t i, stg_device_params& p) {
    constexpr int ND = env_layout_doubles(LAYOUT);
    double q[ND];
    const double2* src = reinterpret_cast<const double2*>(e.soa + i * ND);
#pragma unroll
    for (int k = 0; k < ND / 2; ++k) { const double2 v = src[k]; q[2 * k] = v.x; q[2 * k + 1] = v.y; }
    p.damping = q[0]; p.ms = q[1]; p.ku = q[2]; p.volume = q[3]; p.polarization = q[4];
    for (int k = 0; k < 3; ++k) p.easy_axis[k] = q[5 + k];
    p.area = q[8]; p.r_p = q[9]; p.r_ap = q[10];
    for (int k = 0; k < 3; ++k) p.ref_m[k] = q[11 + k];
    p.r_series = q[14];
    const int tv = (int)q[15];
    p.dev_type = tv & 3;
    p.params_valid = tv >> 2;
    p.demag[0] = 0.0; p.demag[1] = 0.0; p.demag[2] = 1.0; p.a_ex = 20e-12;
    p.sot_tau_dl = 0.0; p.sot_tau_fl = 0.0; p.sot_sigma[0] = 0.0; p.sot_sigma[1] = 1.0; p.sot_sigma[2] = 0.0;
    p.vcma_xi = 0.0; p.vcma_td = 1.0; p.vcma_vbd = 1.0;
    for (int k = 0; k < 3; ++k) p.shape_demag[k] = 0.0;
    if constexpr (LAYOUT == ENV_LAYOUT_LLGS) {
        for (int k = 0; k < 3; ++k) p.demag[k] = q[16 + k];
        p.a_ex = q[19];
    }
    if constexpr (LAYOUT == ENV_LAYOUT_DEV) {
        p.sot_tau_dl = q[16]; p.sot_tau_fl = q[17];
        for (int k = 0; k < 3; ++k) p.sot_sigma[k] = q[18 + k];
        p.vcma_xi = q[21]; p.vcma_td = q[22]; p.vcma_vbd = q[23];
    }
}

// Returns this lane's row of derived constants.  One class: the (wave-uniform) global table row, which the compiler
// turns into scalar loads.  MULTI: the class table staged in LDS, or -- per-env parameters -- a row per lane derived
// here from the env's own record (the LDS block holds exactly 64 rows: per-env launches use 64 integrating lanes per
// workgroup; lanes of a producer wavefront read the row of the integrating lane they mirror).
// `layout`: the per-env record layout -- a compile-time constant in the step kernel (its solver and torque model fix it), the
// context's value in the reset kernel.
template <bool MULTI>
__device__ __forceinline__ const double* class_row(const double* ctab, const uint8_t* cls, int32_t ncls, int64_t i,
                                                   bool in_range, double* lds, const EnvParams& ep, int64_t N, int layout = ENV_LAYOUT_CORE) {
    if (MULTI) {
        if (ep.soa) {
            const int lane = (int)(threadIdx.x & 63u);
            if (threadIdx.x < 64 && in_range) {
                stg_device_params p;
                if (layout == ENV_LAYOUT_CORE) load_env_params<ENV_LAYOUT_CORE>(ep, i, p);
                else if (layout == ENV_LAYOUT_LLGS) load_env_params<ENV_LAYOUT_LLGS>(ep, i, p);
                else load_env_params<ENV_LAYOUT_DEV>(ep, i, p);
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
    return make_simple(V3{r[C_EX], r[C_EY], r[C_EZ]}, r[C_HK], r[C_MS], r[C_ALPHA], r[C_GEFF], r[C_HS_SIMPLE]);
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
    DevTorque dv{0.0, 0.0, V3{0.0, 1.0, 0.0}, k.ghk, false};
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
            dv.ghk_pulse = -row[C_GEFF] * ((2 * keff) / row[C_MU0MS]);
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

// A12: _get_observation (vector mode): the twelve fp32 values.
__device__ __forceinline__ void make_obs(float (&o)[12], const V3& m, const V3& tgt, const double* row, const CfgView& c,
                                         int32_t step, double etot, double J, double T) {
#pragma clang fp contract(off)
    const V3 ref{row[C_REFX], row[C_REFY], row[C_REFZ]};
    const double r = resistance(m, (int)row[C_DEVTYPE], row[C_RP], row[C_RAP], row[C_TMR], ref, row[C_RSERIES]);
    o[0] = obs_cast(m.x);
    o[1] = obs_cast(m.y);
    o[2] = obs_cast(m.z);
    o[3] = obs_cast(tgt.x);
    o[4] = obs_cast(tgt.y);
    o[5] = obs_cast(tgt.z);
    o[6] = obs_cast(r / row[C_RP]);
    // (the normalisations by run constants are multiplications by host-computed reciprocals: <= 1 ulp of fp64 away
    // from the reference's quotients, before the cast to fp32)
    o[7] = obs_cast(c.temp_norm);
    o[8] = obs_cast((double)(c.max_steps - step) / (double)c.max_steps);
    o[9] = obs_cast(etot * 1e12);
    o[10] = obs_cast(J * c.inv_max_current);
    o[11] = obs_cast(T * c.inv_max_duration);
}
// ... written component-major (float[12][N], env index fastest)
__device__ __forceinline__ void write_obs(float* obs, int64_t N, int64_t i, const V3& m, const V3& tgt, const double* row,
                                          const CfgView& c, int32_t step, double etot, double J, double T) {
    float o[12];
    make_obs(o, m, tgt, row, c, step, etot, J, T);
#pragma unroll
    for (int k = 0; k < 12; ++k) obs[k * N + i] = o[k];
}
// ... or as env i's record (STG_OUT_RECORDS): { f32 obs[12]; f32 reward; u8 terminated, truncated, status, 0 } = 56 B, written
// as seven 8-byte stores (records are 8-byte aligned: 56 i on top of an allocation's base).  One contiguous block per
// env instead of fifteen scattered elements: with the sorted lane schedule a lane's stores then fill whole sectors.
__device__ __forceinline__ void write_record(void* base, int64_t i, const V3& m, const V3& tgt, const double* row,
                                             const CfgView& c, int32_t step, double etot, double J, double T, float reward,
                                             uint32_t flags) {
    float o[12];
    make_obs(o, m, tgt, row, c, step, etot, J, T);
    float2* rec = (float2*)((char*)base + i * STG_RECORD_BYTES);
#pragma unroll
    for (int k = 0; k < 6; ++k) rec[k] = make_float2(o[2 * k], o[2 * k + 1]);
    rec[6] = make_float2(reward, __uint_as_float(flags));
}

// ... or only the 48 observation bytes of env i's record (a masked stg_reset reporting an env it does NOT reset: the
// reward / flag fields of that record still belong to the env's last step and stay as they are)
__device__ __forceinline__ void write_record_obs(void* base, int64_t i, const V3& m, const V3& tgt, const double* row,
                                                 const CfgView& c, int32_t step, double etot, double J, double T) {
    float o[12];
    make_obs(o, m, tgt, row, c, step, etot, J, T);
    float2* rec = (float2*)((char*)base + i * STG_RECORD_BYTES);
#pragma unroll
    for (int k = 0; k < 6; ++k) rec[k] = make_float2(o[2 * k], o[2 * k + 1]);
}

// Where the dispatcher put this launch's wavefronts (stg_get_placement).  The schedules above lean on observed dispatcher behaviour
// (workgroup b on XCD b % 8, a CU takes workgroups q, q + 32, ..., which wavefronts share a SIMD); the release build records what a
// launch actually got so that a slow run can be told from a slow build: one lane of every wavefront stores its HW_ID / XCC_ID once
// per launch -- two s_getreg_b32 and one 4-byte store per wavefront, nothing in any loop.
//   table[0] = workgroups of the launch, table[1] = wavefronts per workgroup, then PLACEMENT_ENTRY words per wavefront (index
//   b * wpw + wave, the first PLACEMENT_CAP wavefronts):
//     [0] bits 0-15 HW_ID[15:0] (wave slot 3:0, SIMD 5:4, pipe 7:6, CU 11:8, SH 12, SE 15:13), bits 16-19 XCC_ID, bit 20 producer
//         wavefront of a wave-specialised pair, bit 31 valid
//     [1], [2] the low 32 bits of the 100 MHz real-time counter (s_memrealtime) when the wavefront started / retired (0: it had no
//         env): with [0] the per-SIMD timeline of the launch -- who ran where, next to whom, for how long
//     [3] the wavefront's work: the largest number of integrator work units (RK4 / Euler sub-steps, RK45 attempts) any of its lanes
//         did in this launch (a refill wavefront: the largest per-lane total over its queue); 0 for a producer
constexpr int PLACEMENT_CAP = 4096;
constexpr int PLACEMENT_ENTRY = 4;
constexpr int PLACEMENT_WORDS = 2 + PLACEMENT_ENTRY * PLACEMENT_CAP;
constexpr int PLACEMENT_RING = 32;        // tables kept: one per launch, the last PLACEMENT_RING launches
__device__ __forceinline__ void record_placement(uint32_t* table, int wave, int lane, bool producer) {
    if (table == nullptr || lane != 0) return;
    const uint32_t wpw = blockDim.x >> 6, idx = blockIdx.x * wpw + (uint32_t)wave;
    if (idx == 0u) { table[0] = gridDim.x; table[1] = wpw; }
    if (idx >= (uint32_t)PLACEMENT_CAP) return;
    const uint32_t hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID bits 15:0
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);     // HW_REG_XCC_ID bits 3:0
    uint32_t* e = table + 2 + PLACEMENT_ENTRY * idx;
    e[0] = (hw & 0xFFFFu) | ((xcc & 0xFu) << 16) | (producer ? (1u << 20) : 0u) | (1u << 31);
    e[1] = (uint32_t)__builtin_amdgcn_s_memrealtime();
    e[2] = 0u;
    e[3] = 0u;
}
// ... and when the wavefront retires (every exit of a wavefront that had work); `work`: this lane's integrator work units
__device__ __forceinline__ void record_retired(uint32_t* table, int wave, int lane, unsigned long long work) {
    if (table == nullptr) return;
    uint32_t w = work > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)work;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t v = (uint32_t)__shfl_xor((int)w, o); w = v > w ? v : w; }
    if (lane != 0) return;
    const uint32_t wpw = blockDim.x >> 6, idx = blockIdx.x * wpw + (uint32_t)wave;
    if (idx >= (uint32_t)PLACEMENT_CAP) return;
    uint32_t* e = table + 2 + PLACEMENT_ENTRY * idx;
    e[2] = (uint32_t)__builtin_amdgcn_s_memrealtime();
    e[3] = w;
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
//  * How many tiles an XCD group has in flight (`walk`).  A tile's 64 wavefronts make scattered 4-8 B accesses over the
//    tile's whole window of every state/output row (~0.7 MB per tile); the XCD's 4 MB L2 merges them into full lines
//    only while the windows of all tiles in flight fit into it.  The group therefore walks its tiles `walk` at a time,
//    rank-major inside such a set (longest wavefronts of the set first): measured at 262 144 mixed envs, all 8 tiles of
//    a group at once = 363 MB of HBM traffic per launch for 42 MB of algorithmic bytes, see DESIGN.md section 3.
constexpr uint32_t STG_WALK_SNAKE_ON = 0x40000000u, STG_WALK_SNAKE_OFF = 0x20000000u;      // flag bits in `walk`
// with STG_WALK_SNAKE_ON: the number of leading rounds the boustrophedon order applies to (bits 21..28; 0 = every round)
constexpr uint32_t STG_WALK_ROUNDS_SHIFT = 21u, STG_WALK_ROUNDS_MASK = 0xFFu << STG_WALK_ROUNDS_SHIFT;
template <int WGW>
__device__ __forceinline__ int64_t stg_slot_block(uint32_t b, uint32_t nwg, bool sorted, int cw, bool pairs, uint32_t walk, uint32_t spread_max = 256u) {
    // (nwg: the workgroups of the batch, ceil(N / (WGW * 64)); the grid holds whole tiles, see step_grid)
    constexpr uint32_t TILE_WGS = TILE_WAVES / WGW;                   // workgroups per tile
    if (!sorted) return (int64_t)b * WGW + cw;
    // A ragged last tile (N no multiple of 4096) is treated as a complete one whose slots beyond N are empty: the plan kernel sorts its
    // envs into its first slots, so its real blocks sit at the low ranks and are dealt with the other tiles' long blocks (rank-major:
    // early), while its empty blocks sit at the high ranks -- dispatched last, gone at once.  (Round 3, first attempt: the ragged
    // tile's workgroups at the front of the grid, padded to 8 -- one extra workgroup upset the dispatcher's round structure that the
    // boustrophedon order relies on: RK4 at T = 0 K 131 136 envs 0.39 ms against 0.28 ms at 131 072.  Before that: at the end of the
    // grid, identity order -- its long wavefronts doubled up with other long ones: 100 000 envs RK45 + thermal 4.2 ms.)
    const uint32_t tiles = (nwg + TILE_WGS - 1) / TILE_WGS;           // tiles, the ragged one included
    // One workgroup per CU at most (everything resident from the start, nobody shares a SIMD): a workgroup takes ranks u, u+16,
    // u+32, u+48 of its tile (spread).  With more workgroups than CUs some CU holds two, wavefront w of both on the same SIMD: then
    // a workgroup takes four consecutive ranks, so that the workgroups dispatched last are short throughout and whoever doubles up
    // with them loses little (spread there puts a long wavefront into EVERY workgroup: 73 728 envs RK45 2.38 ms against 1.74 ms).
    const bool spread = nwg <= spread_max;
    const uint32_t snake_rounds = (walk & STG_WALK_ROUNDS_MASK) >> STG_WALK_ROUNDS_SHIFT;
    if (b >= tiles * TILE_WGS) return (int64_t)tiles * TILE_WAVES;    // (beyond the grid of step_grid: no env)
    nwg = tiles * TILE_WGS;
    const uint32_t r = b % 8;                                         // XCD group
    uint32_t q = b / 8;                                               // position inside the group
    if (WGW == 1 && pairs && tiles > 8u && tiles <= 16u && !(walk & STG_WALK_SNAKE_OFF)) {
        // Wave-specialised launch with three or four workgroups per CU (32 768 < N <= 65 536 envs: 128-thread workgroups, everything
        // resident at once).  Observed (tools/probes/wave_placement.hip, 513 ... 1024 x 128 threads; profiles/r03_pair_placement.txt):
        // workgroup b runs on XCD b % 8, a CU takes the workgroups q, q+32, q+64, q+96 of its XCD group, and whatever the workgroup
        // count the integrating wavefront of a CU's arrival g+1 shares its SIMD with the producer of arrival g (the last arrival's
        // producer with the first arrival's integrating wavefront).  Even arrivals therefore take the long end of the group's share of
        // the rank-major order and odd arrivals the short end: every long integrating wavefront shares its SIMD with the producer of a
        // short one, which retires early, and its own producer runs next to a short integrating wavefront.  (Plain rank-major order:
        // arrivals j, 256+j, 512+j, 768+j -- a long producer next to a three-quarter-length integrating wavefront: RK45 + thermal at
        // 60 000 envs 2.40 ms against 2.14, RK4 + thermal 0.65 against 0.53.  At exactly 16 tiles the rule keeps each tile on one XCD
        // group (position q of group r is block 8 q + r of the order = tile r or 8 + r); rounds 1-2 had a separate map for that size
        // -- ranks j, 63-j of both tiles on CU j -- which this one replaces: RK45 the same, RK4 0.592 -> 0.559 ms.)
        const uint32_t n_q = tiles * TILE_WGS / 8u, round = q / 32u, p = q % 32u;
        // (the fourth arrival -- whose producer shares a SIMD with the FIRST arrival's integrating wavefront, the longest of the CU --
        // takes the very shortest blocks, the second arrival the next shortest)
        const uint32_t len3 = n_q > 96u ? n_q - 96u : 0u;                     // workgroups of the group's fourth round
        q = (round == 0u) ? p : (round == 2u) ? (32u + p) : (round == 3u) ? (n_q - 1u - p) : (n_q - 1u - (len3 + p));
        const uint32_t o = q * 8u + r, u = o / tiles, t = o % tiles;
        return (int64_t)t * TILE_WAVES + u;
    }
    if (tiles % 8u != 0u) {
        // Tile counts that are no multiple of 8 (round 3: until then the tiles beyond the last complete group of 8 kept the identity
        // map at the END of the grid, where their long wavefronts doubled up with other long ones -- 81 920 envs RK45 + thermal took
        // 4.9 ms, more than 131 072).  No tile-to-XCD affinity is possible here (the dispatcher deals the workgroups evenly over the XCDs, the tiles
        // do not divide evenly), and with per-env records none is needed; what matters is the order: rank-major over ALL tiles
        // (every tile's longest workgroup first), position o = q * 8 + r of that order, with the same boustrophedon rule per XCD
        // group.  For tile counts that are multiples of 8 this formula IS the map below (u = q / tiles_per_xcd, t = (q %
        // tiles_per_xcd) * 8 + r).
        const uint32_t n_q = tiles * TILE_WGS / 8u, round = q / 32u, p = q % 32u;          // (TILE_WGS is a multiple of 8)
        const bool snake = (walk & STG_WALK_SNAKE_ON) ? true : ((walk & STG_WALK_SNAKE_OFF) ? false : n_q <= 64u);
        const uint32_t len = (n_q - round * 32u) < 32u ? (n_q - round * 32u) : 32u;
        if (snake && (round & 1u) && (snake_rounds == 0u || round < snake_rounds)) q = round * 32u + (len - 1u - p);
        const uint32_t o = q * 8u + r, u = o / tiles, t = o % tiles;
        const uint32_t rank = spread ? (u + TILE_WGS * cw) : (WGW * u + cw);
        return (int64_t)t * TILE_WAVES + rank;
    }
    const uint32_t tiles_per_xcd = tiles / 8;
    {
        // Boustrophedon order over the XCD's 32 CUs.  The dispatcher deals a group's workgroups to its CUs in rounds of 32;
        // when every workgroup is resident from the start (at most two rounds: every kernel fits two workgroups per CU) the
        // assignment is static, and with the plain longest-first order CU j gets the j-th longest workgroup of BOTH rounds
        // (131 072 envs: 1.5 work units on the first CUs, 0.6 on the last).  Reversing the second round gives every CU the
        // same sum: RK45 + thermal at 131 072 envs 4.03 -> 3.09 ms, RK4 + thermal 0.95 -> 0.77 ms.  With more rounds the
        // later workgroups go to whichever CU frees up first, where longest-first is the better order (262 144 envs:
        // 5.61 ms against 6.00 ms reversed) -- so: two rounds at most (STG_SNAKE=0/1 forces it off/on for experiments).
        const uint32_t n_q = tiles_per_xcd * TILE_WGS, round = q / 32u, p = q % 32u;
        const bool snake = (walk & STG_WALK_SNAKE_ON) ? true : ((walk & STG_WALK_SNAKE_OFF) ? false : n_q <= 64u);
        const uint32_t len = (n_q - round * 32u) < 32u ? (n_q - round * 32u) : 32u;
        if (snake && (round & 1u) && (snake_rounds == 0u || round < snake_rounds)) q = round * 32u + (len - 1u - p);
    }
    walk &= ~(STG_WALK_SNAKE_ON | STG_WALK_SNAKE_OFF | STG_WALK_ROUNDS_MASK);
    const uint32_t W = walk < tiles_per_xcd ? (walk ? walk : 1u) : tiles_per_xcd;        // tiles walked together
    const uint32_t set = q / (W * TILE_WGS), within = q % (W * TILE_WGS);
    const uint32_t Ws = (tiles_per_xcd - set * W) < W ? (tiles_per_xcd - set * W) : W;   // (the last set may be smaller)
    const uint32_t u = within / Ws, t = (set * W + within % Ws) * 8 + r;
    // one workgroup per CU at most (everything resident from the start): spread; otherwise keep wavefronts of similar
    // duration together, so that a workgroup's four SIMD slots come free together for the next one (measured 6.0 ms
    // against 7.6 ms at 262 144 envs)
    const uint32_t rank = spread ? (u + TILE_WGS * cw) : (WGW * u + cw);
    return (int64_t)t * TILE_WAVES + rank;
}

// 64-slot block `idx` of the longest-first order -> first slot: rank-major over the tiles, a ragged last tile included (every tile's
// longest block, second longest, ...; the ragged tile's blocks beyond N are empty).  idx < tiles * 64.
__device__ __forceinline__ int64_t refill_slot_base(int64_t idx, int64_t tiles) {
    const int64_t j = idx / tiles, t = idx - j * tiles;
    return t * TILE_ENVS + j * 64;
}

// Hybrid wave-specialised launch (RK45 / RK4 + thermal, 65 536 < N <= 131 072 envs, sorted schedule).  Such a launch is a makespan: it
// ends with its longest wavefront, which runs nearly alone on its SIMD for most of the time -- and with the normals inline that
// wavefront issues 725 instructions per RK45 attempt, with a producer 484.  There are 2048 wave slots (227 VGPRs: two wavefronts per
// SIMD) and nblk > 1024 blocks, so not every block can have a producer; the launch is ALWAYS 1024 two-wavefront workgroups:
//   * the n_pair = 2048 - nblk LONGEST blocks of the rank-major order run as producer/consumer pairs (one block per workgroup),
//   * the other 2 (nblk - 1024) blocks two per workgroup, both wavefronts integrating with the normals inline.
// Every wavefront is resident from the start and the dispatcher's arrival structure is that of the 65 536-env launch (stg_slot_block:
// workgroup b on XCD b % 8, a CU takes q, q+32, q+64, q+96 of its group, wavefront 1 of arrival g shares its SIMD with wavefront 0 of
// arrival g+1), so the same rule places the workgroups: ranked longest-first (pairs, then the two-block workgroups), even arrivals take
// the long end of the group's share and odd arrivals the short end.  (Rounds 2-3 launched nblk workgroups -- the unpaired ones with a
// second wavefront that retired at once -- which broke that structure beyond 1024 workgroups: 66 000 envs 2.60 ms.)
// Speed heuristics only: results never depend on the schedule.
// -> first slot of the block of wavefront `wave` (0 / 1) of workgroup b; `paired`: the workgroup is a producer/consumer pair
__device__ __forceinline__ int64_t stg_hybrid_block(uint32_t b, int wave, uint32_t n_pair, int64_t tiles, bool& paired, bool a_split = true) {
    const uint32_t r = b % 8u, q = b / 8u, round = q / 32u, p = q % 32u;            // 1024 workgroups: 128 per XCD group
    const uint32_t qq = (round == 0u) ? p : (round == 2u) ? (32u + p) : (round == 3u) ? (127u - p) : (95u - p);   // (as in stg_slot_block)
    const uint32_t k = qq * 8u + r;                                                  // rank of the workgroup, longest first
    paired = k < n_pair;
    // A two-block workgroup takes one block of the longer half of the unpaired blocks (wavefront 0) and one of the shorter half
    // (wavefront 1).  The CU's arrivals share SIMDs in a ring -- wavefront 1 of arrival g with wavefront 0 of arrival g+1, of the
    // fourth with the first arrival's (measured: profiles/r04_simd_timeline.txt) -- and SIMD arbitration is by age: wavefront 1 of a
    // two-block workgroup (the fourth, youngest arrival) sits next to the CU's LONGEST integrating wavefront and gets only the leftover
    // slots until that one retires.  It used to take block 2j+1 next to wavefront 0's 2j -- at 81 920 envs an 867-attempt block that
    // ended the launch at 2.65 ms, 0.9 ms after its SIMD's pair; with the shortest blocks there it ends sooner (a_split = false: the old
    // numbering, experiments).
    const int64_t n_in = 1024 - (int64_t)n_pair, j = (int64_t)k - (int64_t)n_pair;
    const int64_t blk = paired ? (int64_t)k : (a_split ? (int64_t)n_pair + (wave ? n_in + j : j) : (int64_t)n_pair + 2 * j + wave);
    return blk < tiles * TILE_WAVES ? refill_slot_base(blk, tiles) : tiles * TILE_ENVS;
}

// ------------------------------------------------------------------------------------------------
// what follows the solve in one env-step (A11-A14): energy, state update, reward, flags, same-step auto-reset, outputs
// ------------------------------------------------------------------------------------------------
// m, tgt, etot, step, rng, done: the env's state before the step in, after it out (the caller stores it); so: the solve's result
// (ignored when !lane_solves: an env that is not stepped -- skip_done, or a lane without an env).
__device__ __forceinline__ void env_step_tail(const StepArgs& a, int64_t i, int64_t ko, bool wr, bool live, bool lane_solves,
                                              const double* row, uint64_t env_id, V3& m, V3& tgt, double& etot, int32_t& step,
                                              uint32_t& rng, bool& done, double J, double T, const SolveOut& so,
                                              unsigned long long& c_steps, unsigned long long& c_sub, unsigned long long& c_noop) {
    // the env-step arithmetic around the solver (energy, reward, flags) has no contraction: same roundings in every
    // instantiation, and the same as NumPy's
#pragma clang fp contract(off)
    const int64_t N = a.N;
    uint8_t st;
    double reward, energy = 0.0;
    bool is_success, truncated;
    if (!lane_solves) {
        st = STG_STATUS_INACTIVE; reward = 0.0;
        is_success = dot(m, tgt) >= a.c.thr;
        truncated = step >= a.c.max_steps;
    } else {
        const double prev_align = dot(m, tgt);                                   // spin_torque_env.py:338-339
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
    if (wr && do_reset && a.final_obs) {
        if (a.records) {      // env-major float[N][12]: one 48-byte block per env
            float o[12];
            make_obs(o, m, tgt, row, a.c, step, etot, J, T);
            float2* fo = (float2*)(a.final_obs + (ko * N + i) * 12);
#pragma unroll
            for (int q = 0; q < 6; ++q) fo[q] = make_float2(o[2 * q], o[2 * q + 1]);
        } else {
            write_obs(a.final_obs + ko * 12 * N, N, i, m, tgt, row, a.c, step, etot, J, T);
        }
    }
    if (do_reset) {
        device_reset_draw(a.c.seed, env_id, rng, a.c, true, true, m, tgt);
        etot = 0.0; step = 0; done = false;
        J = 0.0; T = 0.0;                       // last_action = zeros after reset (spin_torque_env.py:283)
    }
    if (wr) {
        if (a.records) {
            write_record((char*)a.obs + ko * N * STG_RECORD_BYTES, i, m, tgt, row, a.c, step, etot, J, T, (float)reward,
                         (is_success ? 1u : 0u) | (truncated ? 0x100u : 0u) | ((uint32_t)st << 16));
        } else {
            write_obs(a.obs + ko * 12 * N, N, i, m, tgt, row, a.c, step, etot, J, T);
            a.reward[ko * N + i] = (float)reward;
            a.term[ko * N + i] = is_success ? 1 : 0;
            a.trunc[ko * N + i] = truncated ? 1 : 0;
        }
        if (a.reward64) a.reward64[ko * N + i] = reward;
        if (a.energy) a.energy[ko * N + i] = energy;
        if (a.status) a.status[ko * N + i] = st;
    }
}

constexpr int REFILL_STRIPES = 64;            // interleaved stripes of the refill queue, one cursor each
constexpr int REFILL_CURSOR_STRIDE = 16;      // u64 between two cursors (a 128-byte line each)

// The refill loop of ONE persistent wavefront (see stg_step_refill_kernel below, which is this loop for every wavefront of a launch).  The
// queue holds the blocks [blk0, blk0 + nblk_q) of the rank-major order; this is wavefront w of the nw that share it.  (A device function
// of its own since round 4's experiment of running it in the non-pair workgroups of the hybrid launch: profiles/EXPERIMENTS.md.)
template <bool THERMAL, bool MULTI, bool AXIS_Z, typename AT>
__device__ __forceinline__ void refill_wave(const StepArgs& a, int64_t w, int64_t nw, int64_t blk0, int64_t nblk_q, int lane, const double* s_tab,
                                            unsigned long long& c_steps, unsigned long long& c_sub, unsigned long long& c_noop) {
    const int64_t N = a.N;
    const int64_t tiles = (N + TILE_ENVS - 1) / TILE_ENVS;
    // (the next refill launch's cursors: nobody reads them during this launch)
    if (w == 0 && lane < REFILL_STRIPES) a.refill_cursor_next[lane * REFILL_CURSOR_STRIDE] = 0ull;
    const AT* act = (const AT*)a.actions;
    const Recorder norec{};
    const LlgsEnergyK noek{};
    const Dp5Tab tb = make_dp5_tab();
    InlineNormals ns;

    // the lane's env in flight
    int64_t i = 0;
    bool has_env = false;
    double J = 0.0, T = 0.0;
    V3 e_tgt{0.0, 0.0, 1.0};                                    // the env's record as loaded (target, energy, step count, stream position)
    double e_etot = 0.0;
    int32_t e_step = 0;
    uint32_t e_rng = 0;
    bool e_skip = false;
    const double* row = a.ctab;
    LlgsK k = load_llgs(row);
    LlgsLane L;
    L.active = false; L.ok = true; L.rejected = false; L.attempts = 0; L.npts = 0;
    L.y = L.f = L.m0 = V3{0.0, 0.0, 1.0};
    L.t = L.T = L.h_abs = L.min_step = L.bJ = L.bpJ = 0.0;
    V3 out_m{0.0, 0.0, 1.0};

    // takes queue entry p (if there is one): state, action, the solve's prologue
    auto take = [&](int64_t p, bool want) {
        const int64_t idx = blk0 + (p >> 6);                     // block of the rank-major order
        const bool valid_blk = want && (p >> 6) < nblk_q;
        const int64_t slot = (valid_blk ? refill_slot_base(idx, tiles) : 0) + (p & 63);
        const bool valid = valid_blk && slot < N;
        if (!valid) return;
        i = a.perm ? (int64_t)a.perm[slot] : slot;
        V3 m;
        bool done;
        load_state(a.s, i, m, e_tgt, e_etot, e_step, e_rng, done);           // (kept for the tail of the env-step: no second read)
        const uint32_t rng = e_rng;
        if (a.perm) {
            typedef typename std::conditional<std::is_same<AT, double>::value, double2, float2>::type AT2;
            const AT2 aa = ((const AT2*)a.act_sorted)[slot];
            parse_action<AT>(aa.x, aa.y, a.c.max_current, a.c.max_duration, J, T);
        } else {
            parse_action<AT>(act[i], act[N + i], a.c.max_current, a.c.max_duration, J, T);
        }
        if (MULTI) {
            const int c = (int)a.cls[i];
            row = s_tab + (c < a.ncls ? c : 0) * C_COUNT;
            k = load_llgs(row);
        }
        const RngKey rk{a.c.seed, (uint64_t)(a.env_id0 + i), rng};
        // with skip_done an env whose episode has ended is not integrated: it passes through the tail as inactive at the next refill point
        e_skip = a.c.skip_done && done;
        llgs_lane_begin<THERMAL, false, AXIS_Z>(L, out_m, m, J, T, k, row[C_BETA], row[C_BETAP], a.c.rtol, a.c.atol, a.c.max_step, rk,
                                                norec, noek, ns, !e_skip);
        has_env = true;
    };
    // finishes the lane's env: the rest of the env-step after the solve, outputs, state
    auto finish = [&]() {
        const SolveOut so = llgs_lane_finish<false>(L, out_m, norec, noek, ns);
        V3 m = L.m0, tgt = e_tgt;                                // (the row the solve started from; the rest of the record from take())
        double etot = e_etot;
        int32_t step = e_step;
        uint32_t rng = e_rng;
        bool done = e_skip;                                      // (a stepped env's flag is recomputed by the tail)
        env_step_tail(a, i, 0, true, true, !e_skip, row, (uint64_t)(a.env_id0 + i), m, tgt, etot, step, rng, done, J, T, so, c_steps, c_sub, c_noop);
        store_state(a.s, i, m, tgt, etot, step, rng, done);
        has_env = false;
        L.active = false;
    };

    take(w * 64 + lane, true);
    // The blocks behind the nw initial ones, r = 0 ... n_rest - 1 (block nw + r of the queue), are dealt over REFILL_STRIPES interleaved
    // stripes -- stripe s holds r = s, s + S, s + 2 S, ...: every stripe runs from long to short envs -- each with a cursor of its own:
    // a wavefront draws from stripe w % S and moves on to the next stripe when that one is empty.  (One cursor for the whole queue cost
    // the short-pulse launch -- 1 048 576 envs of one attempt each: 14 000 atomics on ONE address in 120 us -- half its speed.)
    const int64_t n_rest = nblk_q > nw ? nblk_q - nw : 0;
    static_assert(REFILL_STRIPES == 64, "one lane per stripe when a wavefront looks for a stripe that still has entries");
    int stripe = (int)(w % REFILL_STRIPES);                     // wave-uniform: the stripe the next reservation comes from
    bool queue_open = true;                                     // wave-uniform: some stripe may still have entries
    // entries reserved but not handed out yet (wave-uniform): [loc_q, loc_end) of stripe loc_stripe.  A wavefront whose 64 lanes are ALL
    // idle at two refill points in a row (envs of a few attempts, all through together: the short-pulse regime) reserves four blocks at
    // once and hands them out over the next refill points without touching memory; otherwise it reserves exactly what its idle lanes
    // take (a reserved block waits for THIS wavefront's lanes: long envs must not be hoarded).
    int64_t loc_q = 0, loc_end = 0;
    int loc_stripe = 0, full_streak = 0;
    bool more = n_rest > 0;                                     // wave-uniform: the queue may still hold entries for this wavefront
    const int check = a.refill_check > 0 ? a.refill_check : 1;
    for (;;) {
        // up to `check` attempts of the whole wavefront (lanes that are through walk along, frozen) ...
        for (int c = 0; c < check; ++c) {
            llgs_lane_gate(L, a.c.max_attempts);
            if (__ballot(L.active) == 0ull) break;
            V3 z2{0.0, 0.0, 0.0}, z3{0.0, 0.0, 0.0};
            llgs_lane_attempt<THERMAL, false, AXIS_Z>(L, out_m, k, tb, a.c.rtol, a.c.atol, a.c.max_step, norec, noek, ns, z2, z3);
        }
        // ... then a refill point: finished lanes write their env; every lane without an env -- finished just now, or one that drew an
        // empty slot earlier -- takes the next entries in lane order: first what the wavefront has reserved, then a new reservation from
        // its current stripe (ONE atomic)
        const bool fin = has_env && !L.active;
        if (__ballot(fin) != 0ull || (more && __ballot(!has_env) != 0ull)) {
            if (fin) finish();
            const unsigned long long takers = __ballot(!has_env);
            if (more && takers != 0ull) {
                const int n_take = (int)__builtin_popcountll(takers), first = (int)__builtin_ctzll(takers);
                const int rank = (int)__builtin_popcountll(takers & ((1ull << lane) - 1ull));
                const int64_t avail = loc_end - loc_q;
                const int need_new = n_take > avail ? (int)(n_take - avail) : 0;
                int64_t new_q = 0;
                const int new_stripe = stripe;
                if (need_new > 0 && queue_open) {
                    const int reserve = (takers == ~0ull && full_streak >= 1) ? 4 * 64 : need_new;
                    unsigned long long base = 0;
                    if (lane == first) base = atomicAdd(a.refill_cursor + stripe * REFILL_CURSOR_STRIDE, (unsigned long long)reserve);
                    base = (unsigned long long)__shfl((long long)base, first);
                    new_q = (int64_t)base;
                    // (entries beyond the stripe's last block do not exist; a stripe whose last entry has been reserved is used up: on to
                    // the next one -- all of them tried: nothing left to reserve)
                    const int64_t stripe_end = ((n_rest - stripe + REFILL_STRIPES - 1) / REFILL_STRIPES) * 64;
                    int64_t new_end = new_q + reserve;
                    if (new_end >= stripe_end) {
                        new_end = stripe_end;
                        // ... which one?  Lane s reads cursor s (a snapshot: cursors only grow, a stripe seen empty is empty): the next
                        // stripe after this one that still has entries -- ONE vector load instead of probing the stripes one atomic
                        // at a time (which cost the short-pulse launch a tail of up to 64 atomic round trips per wavefront)
                        const int64_t end_l = ((n_rest - lane + REFILL_STRIPES - 1) / REFILL_STRIPES) * 64;
                        const unsigned long long cur_l = __hip_atomic_load(a.refill_cursor + lane * REFILL_CURSOR_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        unsigned long long open_s = __ballot((int64_t)cur_l < end_l);
                        open_s &= ~(1ull << stripe);
                        if (open_s == 0ull) queue_open = false;
                        else {
                            const unsigned long long after = stripe == 63 ? 0ull : (open_s >> (stripe + 1)) << (stripe + 1);
                            stripe = (int)__builtin_ctzll(after != 0ull ? after : open_s);
                        }
                    }
                    // this lane's entry: of the old reservation while it lasts, then of the new one
                    const bool from_old = rank < avail;
                    const int64_t q = from_old ? loc_q + rank : new_q + (rank - avail);
                    const int st = from_old ? loc_stripe : new_stripe;
                    const int64_t r = (q >> 6) * REFILL_STRIPES + st;                     // rest block r, slot q % 64
                    take((nw + r) * 64 + (q & 63), !has_env && r < n_rest && (from_old || q < new_end));
                    loc_q = new_q + need_new < new_end ? new_q + need_new : new_end;
                    loc_end = new_end;
                    loc_stripe = new_stripe;
                } else {
                    const int64_t q = loc_q + rank;
                    const int64_t r = (q >> 6) * REFILL_STRIPES + loc_stripe;
                    take((nw + r) * 64 + (q & 63), !has_env && q < loc_end && r < n_rest);
                    loc_q = loc_q + n_take < loc_end ? loc_q + n_take : loc_end;
                }
                more = queue_open || loc_q < loc_end;
                full_streak = takers == ~0ull ? full_streak + 1 : 0;
            }
        }
        if (__ballot(has_env) == 0ull && !more) break;
    }
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
// every SIMD one integrating and one producing wavefront at 65 536 envs (tools/probes/wave_placement.hip).  (A 4 + 4
// form -- producer 4+w serving integrating wavefront 3-w, all eight in lockstep -- was built and measured: no better
// for RK45, 8 % worse for RK4; it is gone.)
// MULTI: 0 = one device class (constants through scalar loads), 1 = class table staged in LDS, 2 = per-env parameter records: every lane
// derives its constants from its env's record straight into REGISTERS (a private array with compile-time indices only: the constants a
// kernel never reads cost nothing).  Rounds 2-3 derived them into a 64-row LDS block -- 23.5 KB per 64-lane workgroup, which held a CU
// to six wavefronts (1.5 per SIMD) and forced one-wavefront workgroups, which the dispatcher spreads unevenly (1 ... 6 per SIMD): the
// per-env cfg4 row ran at a SIMD busy share of 0.78 (profiles/r04_simd_timeline.txt).
template <int SOLVER, bool THERMAL, int MULTI, bool AXIS_Z, bool DEVPHYS, typename AT, bool PC, int WGW>
#ifndef STG_STEP_ATTR
#define STG_STEP_ATTR
#endif
// (second bound = minimum wavefronts per SIMD: the fixed-step T = 0 K kernels with the reference RHS and the easy axis along z -- every
// factory default -- sit at 127-129 VGPRs: one
// register decides between three and four resident wavefronts per SIMD, so they are held to four: 127 VGPRs, no spills (cfg4 class
// table 0.516 -> 0.505 ms, RK4 at T = 0 K 262 144 envs 0.513 -> 0.497 ms).  Not the class-table kernel with 64-thread workgroups (per-env
// parameter records: its 23.5 KB LDS block per workgroup bounds the occupancy anyway, and the tighter allocation cost it 6 %).)
__global__ void __launch_bounds__(PC ? 2 * WGW * 64 : WGW * 64, (SOLVER != STG_SOLVER_RK45 && !THERMAL && !DEVPHYS && AXIS_Z && MULTI != 2 && !(MULTI && WGW == 1)) ? 4 : 1) STG_STEP_ATTR
stg_step_kernel(const StepArgs a) {
    // the env-step arithmetic around the solver (energy, reward, flags) has no contraction: same roundings in every
    // instantiation, and the same as NumPy's
#pragma clang fp contract(off)
    static_assert(!PC || THERMAL, "wave specialisation only exists for the thermal kernels");
    // class table (MULTI): dynamic LDS sized by the launch -- n_classes rows, or one row per lane with per-env
    // parameters -- so that a three-class batch does not reserve the 23.5 KB of STG_MAX_CLASSES rows (which kept the
    // fourth wave-specialised workgroup off a CU)
    extern __shared__ double s_tab[];
    // normals rings of the wave-specialised kernels (one per integrating wavefront): RK45 hands over finished fields
    // (double), the fixed-step solvers raw normals (float)
    constexpr bool FIELD = SOLVER == STG_SOLVER_RK45;
    using NT = typename std::conditional<FIELD, double, float>::type;
    // how the integrating and the producing wavefront keep in step (stg_physics.hpp, SharedNormalsT), by measurement
    // at 4096 and 65 536 envs: RK4 by handshake words with a ring of 4 chunks (8-12 % faster than a barrier per
    // sub-step), RK45 and Euler by one s_barrier per chunk with a ring of 2 (the words are 5-10 % slower there: RK45's
    // producer has slack and parks at the barrier for free, Euler's chunk is too short to pay for the polling)
    constexpr bool BARRIER = SOLVER != STG_SOLVER_RK4;
    constexpr int DEPTH = BARRIER ? 2 : 4;
    constexpr int RING = DEPTH * (FIELD ? SHARED_CHUNK_RK45 : SHARED_CHUNK_FIXED) * 64;
    static_assert(!PC || WGW == 1, "the wave-specialised form is one integrating + one producing wavefront");
    __shared__ NT s_norm[PC ? RING : 1];
    __shared__ int s_hs[2], s_go[2 * WGW];
    __shared__ unsigned long long s_cnt[(PC ? 2 : 1) * WGW * 3];    // (one triple per wavefront that may integrate)
    __shared__ uint32_t s_rng[PC ? WGW * 64 : 1];
    const int lane = (int)(threadIdx.x & 63u);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // hybrid launch (PC kernels only, a.hybrid = number of pairs + 1): the workgroups that are not pairs have no producer -- both their
    // wavefronts integrate a block of their own with the normals inline (stg_hybrid_block)
    bool paired = PC;
    int64_t hyb_slot = 0;
    if (PC && a.hybrid) hyb_slot = stg_hybrid_block(blockIdx.x, wave, (uint32_t)(a.hybrid - 1), (a.N + TILE_ENVS - 1) / TILE_ENVS, paired, a.hybrid_prio != 0);
    const bool producer = PC && paired && wave >= WGW;
    const int cw = producer ? (2 * WGW - 1 - wave) : (wave < WGW ? wave : wave - WGW);   // the integrating wavefront this one is, or serves
    record_placement(a.placement, wave, lane, producer);
    const int64_t lane_slot = ((PC && a.hybrid) ? hyb_slot
                                                : stg_slot_block<WGW>(blockIdx.x, (uint32_t)((a.N + WGW * 64 - 1) / (WGW * 64)), a.perm != nullptr, cw, PC,
                                                                      (uint32_t)a.walk, (uint32_t)a.spread_max) * 64) + lane;
    const bool live = lane_slot < a.N;
    // duration-sorted schedule: slot j of the launch integrates env perm[j], so the 64 lanes of a wavefront have
    // (nearly) equal trip counts; all state and outputs stay at the env's own index
    const int64_t i = live ? (a.perm ? (int64_t)a.perm[lane_slot] : lane_slot) : 0;
    constexpr int ENV_LAYOUT = SOLVER == STG_SOLVER_RK45 ? ENV_LAYOUT_LLGS : (DEVPHYS ? ENV_LAYOUT_DEV : ENV_LAYOUT_CORE);
    double own_row[MULTI == 2 ? C_COUNT : 1];
    const double* row;
    if constexpr (MULTI == 2) {
        stg_device_params p;                                      // (a lane without an env reads env 0's record and never writes)
        load_env_params<ENV_LAYOUT>(a.ep, i, p);
        derive_row(p, a.ep.gamma, a.ep.temperature, own_row);
        row = own_row;
    } else {
        row = class_row<MULTI != 0>(a.ctab, a.cls, a.ncls, i, live, s_tab, a.ep, a.N, ENV_LAYOUT);
    }
    // Lanes without an env: the one-wavefront form has no rendezvous after this point and lets them go; in the
    // wave-specialised form they stay (inert) because every wavefront of the workgroup takes part in every s_barrier.
    if (!PC && !live) return;
    if (PC && !paired && !live) return;                             // (no rendezvous in a workgroup that is not a pair)
    const int64_t N = a.N;
    const uint64_t env_id = (uint64_t)(a.env_id0 + i);

    if (producer) {
        // per env-step: wait for the stream positions (H1), then stay one chunk ahead of integrating wavefront `cw`.
        // Normals per chunk: RK45 6 (initial step) then 18 per attempt; the fixed-step solvers SHARED_CHUNK_FIXED = 24: two RK4
        // sub-steps with the white field, or eight sub-steps of Euler / of the Ornstein-Uhlenbeck field (3 each)
        constexpr int n_first = SOLVER == STG_SOLVER_RK45 ? 6 : SHARED_CHUNK_FIXED;     // (Euler / OU: SHARED_SUBS3 sub-steps of 3)
        static_assert(3 * SHARED_SUBS3 == SHARED_CHUNK_FIXED, "chunk size of the 3-normal sub-steps");
        constexpr int n_chunk = SOLVER == STG_SOLVER_RK45 ? SHARED_CHUNK_RK45 : n_first;
        const double ghs = FIELD ? load_llgs(row).ghs : 0.0;
        for (int k = 0; k < a.K; ++k) {
            __syncthreads();                                       // H1: s_rng / s_go[k & 1] published
            const int p = (k & 1) * WGW;
            if (!any_flag<WGW>(lds_flag(s_go + p))) continue;
            const RngKey rk{a.c.seed, env_id, s_rng[cw * 64 + lane]};
            produce_normals<NT, FIELD, DEPTH, BARRIER>(s_norm, s_hs, lane, rk, n_first, n_chunk, ghs);
        }
        record_retired(a.placement, wave, lane, 0ull);
        return;
    }

    V3 m, tgt;                                      // (lanes without an env read env 0 and never write)
    double etot;
    int32_t step;
    uint32_t rng;
    bool done;
    load_state(a.s, i, m, tgt, etot, step, rng, done);
    const AT* act = (const AT*)a.actions;
    const Recorder norec{};
    unsigned long long c_steps = 0, c_sub = 0, c_noop = 0;

    for (int k = 0; k < a.K; ++k) {
        double J, T;
        if (k == 0 && a.perm) {
            // slot order: one coalesced 8/16-byte load per lane (the plan kernel wrote it next to the permutation)
            typedef typename std::conditional<std::is_same<AT, double>::value, double2, float2>::type AT2;
            const AT2 aa = ((const AT2*)a.act_sorted)[live ? lane_slot : 0];
            parse_action<AT>(aa.x, aa.y, a.c.max_current, a.c.max_duration, J, T);
        } else {
            parse_action<AT>(act[((int64_t)k * 2 + 0) * N + i], act[((int64_t)k * 2 + 1) * N + i], a.c.max_current,
                             a.c.max_duration, J, T);
        }
        const bool last = (k == a.K - 1);
        const int64_t ko = a.out_every ? k : 0;
        const bool wr = (a.out_every || last) && live;
        // with skip_done, finished envs are not integrated: a wavefront whose lanes are all done skips the integrator
        const bool lane_solves = live && !(a.c.skip_done && done);
        const RngKey rk{a.c.seed, env_id, rng};
        SolveOut so{m, 0, 0, 0, false};
        if (PC && paired) {
            // H1: this wavefront's stream positions and whether it integrates at all; then, if it does, H2 (chunk 0 is in
            // LDS, the handshake words are reset) and the solve (lanes that do not integrate are inert); inside the solve
            // the two wavefronts keep in step through the handshake words, not barriers
            SharedNormalsT<NT, FIELD, DEPTH, BARRIER> shared{s_norm, s_hs, lane, 0, 0, 1};
            const int p = (k & 1) * WGW;
            s_rng[cw * 64 + lane] = rng;
            const bool mine = __ballot(lane_solves) != 0ull;
            s_go[p + cw] = mine ? 1 : 0;
            __syncthreads();                                       // H1
            if ((WGW == 1) ? mine : any_flag<WGW>(lds_flag(s_go + p))) {
                if (!BARRIER) s_hs[1] = 0;                         // (the producer is past its last read of the previous solve's value: H1)
                __syncthreads();                                   // H2
                so = run_solver<SOLVER, THERMAL, false, AXIS_Z, DEVPHYS>(m, J, T, row, a.c, rk, norec, shared, lane_solves);
            }
        }
        if (!(PC && paired) && lane_solves) {
            InlineNormals inl;
            so = run_solver<SOLVER, THERMAL, false, AXIS_Z, DEVPHYS>(m, J, T, row, a.c, rk, norec, inl, true);
        }
        env_step_tail(a, i, ko, wr, live, lane_solves, row, env_id, m, tgt, etot, step, rng, done, J, T, so, c_steps, c_sub, c_noop);
    }
    if (live) store_state(a.s, i, m, tgt, etot, step, rng, done);
    record_retired(a.placement, wave, lane, c_sub);
    // on-device metrics (the reference's EnvironmentMonitor/solver stats are host-side bookkeeping): one atomic per
    // counter per wavefront, into one of COUNTER_STRIPES copies
    const int ci = PC ? wave : cw;                                  // (this wavefront's counter triple)
    wave_add3(a.counters + (size_t)((blockIdx.x * (PC ? 2 : 1) * WGW + ci) % COUNTER_STRIPES) * COUNTER_STRIDE, s_cnt + ci * 3, c_steps, c_sub, c_noop);
}

// ------------------------------------------------------------------------------------------------
// env.step with LANE REFILL (RK45, one env-step per launch): throughput launches with several envs per lane
// ------------------------------------------------------------------------------------------------
// In stg_step_kernel a lane integrates ONE env and then idles until the slowest lane of its wavefront is through: after the
// duration sort the lanes of a wavefront still differ in attempts (attempts per picosecond vary 1.0-1.9 with nothing known
// before the solve: the mean lane of a 64-env block does 0.5-0.7 of the attempts of its slowest lane), and the issue slots of an
// idle lane are paid all the same.  Here nw PERSISTENT wavefronts (one or two per SIMD) share ONE GLOBAL QUEUE of all the launch's
// envs in the rank-major order of the sorted schedule (every tile's longest block first, ...): a wavefront starts with block w of that
// order, and a lane that has finished its env writes that env's outputs and takes the next entry of the queue while its neighbours
// keep integrating.  Longest envs first, handed to whichever lane ANYWHERE on the chip runs dry first: the longest-processing-time
// rule at lane granularity -- the launch ends with the shortest envs, every SIMD retires within one short env of the others
// (round 4, profiles/r04_refill_global_ab.txt; rounds 2-3 gave every wavefront a fixed queue of R blocks: a lane that finished early
// got only what its own wavefront's queue held -- 196 608 envs 4.40 -> 3.67 ms, 262 144 envs 5.18 -> 4.49 ms).  Per-env arithmetic
// is exactly that of stg_step_kernel (llgs_lane_begin / _attempt / _finish, env_step_tail), and an env's arithmetic never depends on
// the lane that runs it, so results are bit-identical whatever the order the entries are taken in (tested).
//  * Queue entry p = slot p % 64 of block p / 64 of the rank-major order; entries [0, nw * 64) are the initial envs, the blocks behind
//    them are dealt over 64 interleaved stripes with a cursor each: ONE atomic per wavefront and refill point (the wavefront's takers
//    get consecutive entries of its current stripe; an empty stripe sends it to the next).  Two sets of cursors alternate between
//    launches: a launch finds its own at 0 and zeroes the next one's (no memset between launches).
//  * A refill point -- finish the env (tail of the env-step, stores), take the next entry (state load, action, the solve's
//    prologue: two RHS calls, initial step) -- is ~2700 instructions of lane-divergent code that the whole wavefront waits for; it
//    is entered at most every `refill_check` attempts (lanes that finished within that window go together) or when no lane is
//    integrating.  A lane that draws an empty slot (a ragged tile's slots beyond N sit among the others) draws again next time.
//  * Two wavefronts per SIMD: the SIMD serves the older one first (arbitration is by priority, then age:
//    tools/probes/simd_fairness.hip), which with a shared queue only means that it takes more entries.
template <bool THERMAL, bool MULTI, bool AXIS_Z, typename AT, int WGW>
__global__ void __launch_bounds__(WGW * 64) stg_step_refill_kernel(const StepArgs a) {
    extern __shared__ double s_tab[];
    __shared__ unsigned long long s_cnt[WGW * 3];
    const int lane = (int)(threadIdx.x & 63u);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (MULTI) {       // class table -> LDS (per-env parameter records do not take this kernel)
        for (int j = threadIdx.x; j < a.ncls * C_COUNT; j += blockDim.x) s_tab[j] = a.ctab[j];
        __syncthreads();
    }
    const int64_t tiles = (a.N + TILE_ENVS - 1) / TILE_ENVS, nblk = tiles * TILE_WAVES, nw = a.refill_nw;   // (blocks of whole tiles)
    const int64_t w = (int64_t)blockIdx.x * WGW + wave;
    record_placement(a.placement, wave, lane, false);
    if (w >= nw) return;
    unsigned long long c_steps = 0, c_sub = 0, c_noop = 0;
    refill_wave<THERMAL, MULTI, AXIS_Z, AT>(a, w, nw, 0, nblk, lane, s_tab, c_steps, c_sub, c_noop);
    record_retired(a.placement, wave, lane, c_sub);
    wave_add3(a.counters + (size_t)((blockIdx.x * WGW + wave) % COUNTER_STRIPES) * COUNTER_STRIDE, s_cnt + wave * 3, c_steps, c_sub, c_noop);
}

// ------------------------------------------------------------------------------------------------
// launch / dispatch of the step kernel (one instantiation set per solver, see stg_step_*.hip)
// ------------------------------------------------------------------------------------------------
constexpr int64_t STG_WG4_MIN_ENVS = 65536;       // 256 CUs x 4 SIMDs x 64 lanes

// dynamic LDS of a launch: the class table of a MULTI kernel (see stg_step_kernel)
template <int MULTI>
static size_t step_dyn_lds(const StepArgs& a) {
    return MULTI == 1 ? (size_t)a.ncls * C_COUNT * sizeof(double) : 0;
}

// Grid of a step launch: under the sorted schedule whole tiles (a ragged last tile counts as one, see stg_slot_block).
static inline unsigned step_grid(const StepArgs& a, int wgw) {
    const unsigned nwg = (unsigned)((a.N + wgw * 64 - 1) / (wgw * 64));
    if (!a.perm) return nwg;
    const unsigned tile_wgs = (unsigned)(TILE_WAVES / wgw);
    return ((nwg + tile_wgs - 1) / tile_wgs) * tile_wgs;
}

// Workgroups of this kernel a CU holds at once (registers, LDS), asked of the runtime once per (kernel, LDS size) and thread.
static int resident_workgroups_per_cu(const void* kernel, int block, size_t lds) {
    struct Key { const void* f; size_t lds; int v; };
    static thread_local Key cache[8] = {};
    static thread_local int used = 0;
    for (int j = 0; j < used; ++j) if (cache[j].f == kernel && cache[j].lds == lds) return cache[j].v;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block, lds) != hipSuccess || nb < 1) nb = 2;
    if (used < 8) cache[used++] = Key{kernel, lds, nb};
    return nb;
}

// The boustrophedon workgroup order (stg_slot_block) balances the rounds of workgroups that are resident from the start -- for
// those the dispatcher's assignment to the 32 CUs of an XCD is static, and the plain longest-first order stacks the longest
// workgroup of EVERY such round on the first CUs.  Later rounds go to whichever CU frees up first, where longest-first is the
// better order.  How many rounds are resident is the kernel's occupancy: the T = 0 K fixed-step kernels hold four workgroups of
// four wavefronts per CU (262 144 envs = four rounds, all resident: cfg4 0.532 -> 0.500 ms, RK4 at T = 0 K 0.54 -> 0.49 ms), the
// device-physics and thermal ones three or two.  So the host asks the runtime and the boustrophedon applies to that many leading
// rounds (a launch a little over two rounds keeps its first two balanced: RK4 + thermal 132 000 envs).
static StepArgs with_snake_rule(const StepArgs& a, const void* kernel, int wgw, size_t lds, unsigned nwg) {
    StepArgs b = a;
    if (!a.perm || (a.walk & (int32_t)(STG_WALK_SNAKE_ON | STG_WALK_SNAKE_OFF))) return b;          // identity schedule / forced by STG_SNAKE
    const unsigned tile_wgs = (unsigned)(TILE_WAVES / wgw), n_q = ((nwg + tile_wgs - 1) / tile_wgs) * tile_wgs / 8u;   // workgroups per XCD group
    const int nb = resident_workgroups_per_cu(kernel, wgw * 64, lds);
    // ... while the launch is at most half a round over what is resident: with a whole further round of workgroups waiting, plain
    // longest-first measured better throughout (cfg4 device-physics kernel, three of four rounds resident: 0.72 ms against 0.765;
    // RK4 + thermal 262 144 envs, two of four: 1.47 against 1.51)
    if (n_q > 32u && nb >= 2 && n_q <= 32u * (unsigned)nb + 16u)
        b.walk |= (int32_t)(STG_WALK_SNAKE_ON | ((unsigned)(nb > 255 ? 255 : nb) << STG_WALK_ROUNDS_SHIFT));
    else b.walk |= (int32_t)STG_WALK_SNAKE_OFF;
    return b;
}

template <int SOLVER, bool THERMAL, int MULTI, bool AXIS_Z, bool DEVPHYS, int WGW>
static void launch_step_w(const StepArgs& a, int act_f64, bool pc, hipStream_t st) {
    const dim3 grid(step_grid(a, WGW));
    const unsigned nwg = (unsigned)((a.N + WGW * 64 - 1) / (WGW * 64));
    const size_t lds = step_dyn_lds<MULTI>(a);
    // (launched by name, not through a function-pointer variable: a host build with -fsanitize=address was seen to push the call
    // configuration and then NOT launch through the pointer -- no error, no kernel; the pointer only serves the occupancy query)
    if (act_f64) {
        const StepArgs b = with_snake_rule(a, (const void*)&stg_step_kernel<SOLVER, THERMAL, MULTI, AXIS_Z, DEVPHYS, double, false, WGW>, WGW, lds, nwg);
        hipLaunchKernelGGL((stg_step_kernel<SOLVER, THERMAL, MULTI, AXIS_Z, DEVPHYS, double, false, WGW>), grid, dim3(WGW * 64), lds, st, b);
    } else {
        const StepArgs b = with_snake_rule(a, (const void*)&stg_step_kernel<SOLVER, THERMAL, MULTI, AXIS_Z, DEVPHYS, float, false, WGW>, WGW, lds, nwg);
        hipLaunchKernelGGL((stg_step_kernel<SOLVER, THERMAL, MULTI, AXIS_Z, DEVPHYS, float, false, WGW>), grid, dim3(WGW * 64), lds, st, b);
    }
}

template <int SOLVER, bool THERMAL, int MULTI, bool AXIS_Z, bool DEVPHYS>
static void launch_step(const StepArgs& a, int act_f64, bool pc, hipStream_t st) {
    if (THERMAL && !DEVPHYS && pc) {
        // wave-specialised variant: one integrating + one producing wavefront per workgroup; not built for the
        // device-physics model
        constexpr bool PC = THERMAL && !DEVPHYS;
        const dim3 grid(a.hybrid ? 1024u : step_grid(a, 1));          // (hybrid: always 1024 workgroups, see stg_hybrid_block)
        if (act_f64)
            hipLaunchKernelGGL((stg_step_kernel<SOLVER, THERMAL, MULTI, AXIS_Z, DEVPHYS, double, PC, 1>), grid, dim3(128), step_dyn_lds<MULTI>(a), st, a);
        else
            hipLaunchKernelGGL((stg_step_kernel<SOLVER, THERMAL, MULTI, AXIS_Z, DEVPHYS, float, PC, 1>), grid, dim3(128), step_dyn_lds<MULTI>(a), st, a);
        return;
    }
    // workgroups of 4 integrating wavefronts once there is one per CU, of 1 below that
    if (a.N >= STG_WG4_MIN_ENVS && !a.force_wg1) launch_step_w<SOLVER, THERMAL, MULTI, AXIS_Z, DEVPHYS, 4>(a, act_f64, pc, st);
    else launch_step_w<SOLVER, THERMAL, MULTI, AXIS_Z, DEVPHYS, 1>(a, act_f64, pc, st);
}
template <int SOLVER, bool AXIS_Z, bool DEVPHYS>
static void dispatch_step2(const StepArgs& a, bool thermal, int multi, int act_f64, bool pc, hipStream_t st) {
    if (thermal) {
        if (multi == 2) launch_step<SOLVER, true, 2, AXIS_Z, DEVPHYS>(a, act_f64, pc, st);
        else if (multi) launch_step<SOLVER, true, 1, AXIS_Z, DEVPHYS>(a, act_f64, pc, st);
        else launch_step<SOLVER, true, 0, AXIS_Z, DEVPHYS>(a, act_f64, pc, st);
    } else {
        if (multi == 2) launch_step<SOLVER, false, 2, AXIS_Z, DEVPHYS>(a, act_f64, false, st);
        else if (multi) launch_step<SOLVER, false, 1, AXIS_Z, DEVPHYS>(a, act_f64, false, st);
        else launch_step<SOLVER, false, 0, AXIS_Z, DEVPHYS>(a, act_f64, false, st);
    }
}
// axis_z selects the easy-axis = z specialisation of the RHS (Simple: e = +z; LLGS: raw axis and demag along z);
// devphys the opt-in device-physics torque model (fixed-step solvers only)
template <int SOLVER>
static void dispatch_step(const StepArgs& a, bool thermal, int multi, bool axis_z, bool devphys, int act_f64, bool pc, hipStream_t st) {
    if constexpr (SOLVER != STG_SOLVER_RK45) {          // (the device-physics torque model exists for the fixed-step solvers only)
        if (devphys) {
            if (axis_z) dispatch_step2<SOLVER, true, true>(a, thermal, multi, act_f64, pc, st);
            else dispatch_step2<SOLVER, false, true>(a, thermal, multi, act_f64, pc, st);
            return;
        }
    }
    if (axis_z) dispatch_step2<SOLVER, true, false>(a, thermal, multi, act_f64, pc, st);
    else dispatch_step2<SOLVER, false, false>(a, thermal, multi, act_f64, pc, st);
}

// lane-refill launch of the RK45 step (a.refill = envs per lane >= 2): ceil(ceil(N / 64) / refill) wavefronts
template <bool THERMAL, bool MULTI, bool AXIS_Z>
static void launch_refill(const StepArgs& a, int act_f64, hipStream_t st) {
    constexpr int WGW = 4;
    const int64_t nw = a.refill_nw;
    const dim3 grid((unsigned)((nw + WGW - 1) / WGW));
    const size_t lds = MULTI ? (size_t)a.ncls * C_COUNT * sizeof(double) : 0;
    if (act_f64) hipLaunchKernelGGL((stg_step_refill_kernel<THERMAL, MULTI, AXIS_Z, double, WGW>), grid, dim3(WGW * 64), lds, st, a);
    else hipLaunchKernelGGL((stg_step_refill_kernel<THERMAL, MULTI, AXIS_Z, float, WGW>), grid, dim3(WGW * 64), lds, st, a);
}
static void dispatch_refill(const StepArgs& a, bool thermal, bool multi, bool axis_z, int act_f64, hipStream_t st) {
    if (thermal) {
        if (multi) { if (axis_z) launch_refill<true, true, true>(a, act_f64, st); else launch_refill<true, true, false>(a, act_f64, st); }
        else { if (axis_z) launch_refill<true, false, true>(a, act_f64, st); else launch_refill<true, false, false>(a, act_f64, st); }
    } else {
        if (multi) { if (axis_z) launch_refill<false, true, true>(a, act_f64, st); else launch_refill<false, true, false>(a, act_f64, st); }
        else { if (axis_z) launch_refill<false, false, true>(a, act_f64, st); else launch_refill<false, false, false>(a, act_f64, st); }
    }
}

// defined in stg_step_{rk4,euler,rk45}.hip
// (multi: 0 one class, 1 class table, 2 per-env parameter records)
void stg_dispatch_step_rk4(const StepArgs& a, bool thermal, int multi, bool axis_z, bool devphys, int act_f64, bool pc, hipStream_t st);
void stg_dispatch_step_euler(const StepArgs& a, bool thermal, int multi, bool axis_z, bool devphys, int act_f64, bool pc, hipStream_t st);
void stg_dispatch_step_rk45(const StepArgs& a, bool thermal, int multi, bool axis_z, int act_f64, bool pc, hipStream_t st);
void stg_dispatch_step_rk45_refill(const StepArgs& a, bool thermal, bool multi, bool axis_z, int act_f64, hipStream_t st);
