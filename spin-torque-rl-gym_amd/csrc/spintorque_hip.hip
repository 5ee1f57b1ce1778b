// spintorque_hip.hip -- kernels and C-ABI of libspintorque_hip.so (see include/spintorque_hip.h).
//
// Layout in HBM (all owned by the context, structure-of-arrays, env index fastest):
//   state : EnvRec[N] = { f64 m[3], target[3], e_tot; u32 step | done << 31; u32 rng }        = 64 B/env, one record per env
//   class table                : f64[n_classes][C_COUNT]  (derived constants, <= 64 classes)
//   cls                        : u8[N] (caller-owned) when n_classes > 1
//   per-env parameters         : records f64[N][16 | 20 | 24] (library-owned, packed from the caller's rows: the fields the context's
//                                solver / torque model read, stg_kernels.hpp: EnvParams), alternative to the class table
// One lane per env; the step kernel's workgroups hold one integrating wavefront (launches below 65 536 envs: small
// batches spread over as many CUs as they have wavefronts) or four (one per SIMD of a CU), plus, in the wave-specialised
// thermal form, one producer wavefront each; the hot loops live entirely in registers.  With one device class the
// constants are read through scalar loads (SGPRs); with several (or per-env parameters) each lane reads its row of
// derived constants from LDS.
#include "stg_kernels.hpp"

#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

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
    const Recorder rec{a.traj_t, a.traj_m, a.traj_e, a.traj_tq, N, i, a.traj_cap};
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
    int32_t records;          // STG_OUT_RECORDS: obs is the record array
    EnvParams ep;
};

__global__ void __launch_bounds__(64) stg_reset_kernel(const ResetArgs a) {
    __shared__ double s_tab[STG_MAX_CLASSES * C_COUNT];
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool in_range = i < a.N;
    const double* row = (a.ncls > 1 || a.ep.soa) ? class_row<true>(a.ctab, a.cls, a.ncls, i, in_range, s_tab, a.ep, a.N, a.ep.layout) : a.ctab;
    if (!in_range) return;
    const int64_t N = a.N;
    V3 m0, t0;
    double etot0;
    int32_t step0;
    uint32_t rng0;
    bool done0;
    load_state(a.s, i, m0, t0, etot0, step0, rng0, done0);
    if (a.mask && !a.mask[i]) {
        if (a.obs) {   // unchanged env: report its current observation (last action unknown -> 0, as after reset)
            // (records: only the observation fields -- reward and flags of an env that is not reset are its last step's,
            // which the caller may still be reading: `reward`/`terminated` of SpinTorqueVecEnv.step are views of this array)
            if (a.records) write_record_obs(a.obs, i, m0, t0, row, a.c, step0, etot0, 0.0, 0.0);
            else write_obs(a.obs, N, i, m0, t0, row, a.c, step0, etot0, 0.0, 0.0);
        }
        return;
    }
    V3 m{0, 0, 1}, t{0, 0, 1};
    device_reset_draw(a.seed, (uint64_t)(a.env_id0 + i), rng0, a.c, a.init_m == nullptr, a.target == nullptr, m, t);
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
    store_state(a.s, i, m, t, 0.0, 0, rng0, false);
    if (a.obs) {
        if (a.records) write_record(a.obs, i, m, t, row, a.c, 0, 0.0, 0.0, 0.0, 0.0f, 0u);
        else write_obs(a.obs, N, i, m, t, row, a.c, 0, 0.0, 0.0, 0.0);
    }
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

// stg_get_state / stg_set_state: between the library's state records and the caller's component-major arrays
// (m, target: double[3][N]; the others [N]; any may be NULL = field not copied)
template <bool SET>
__global__ void stg_state_copy_kernel(StateView s, int64_t N, double* m, double* target, double* etot, int32_t* step,
                                      uint32_t* rng, uint8_t* done) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    V3 mm, tt;
    double e;
    int32_t st;
    uint32_t r;
    bool d;
    load_state(s, i, mm, tt, e, st, r, d);
    if (SET) {
        if (m) mm = V3{m[i], m[N + i], m[2 * N + i]};
        if (target) tt = V3{target[i], target[N + i], target[2 * N + i]};
        if (etot) e = etot[i];
        if (step) st = step[i];
        if (rng) r = rng[i];
        if (done) d = done[i] != 0;
        store_state(s, i, mm, tt, e, st, r, d);
    } else {
        if (m) { m[i] = mm.x; m[N + i] = mm.y; m[2 * N + i] = mm.z; }
        if (target) { target[i] = tt.x; target[N + i] = tt.y; target[2 * N + i] = tt.z; }
        if (etot) etot[i] = e;
        if (step) step[i] = st;
        if (rng) rng[i] = r;
        if (done) done[i] = d ? 1 : 0;
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
    const EnvRec* state;      // with skip_done: finished envs go last (no work)
    int32_t skip_done;
    const uint8_t* cls;       // device-physics torque model with several classes: group lanes by device kind
    const double* ctab;
    const uint8_t* env_type;  // (per-env parameters: the kind of each env)
    int32_t by_kind;
    int32_t regroup;          // by_kind: renumber the tile's groups of four blocks longest-first (launches that are resident as a whole)
    uint32_t* perm;
    void* act_sorted;         // [N][2] in the actions' dtype, slot order: the step kernel reads its action with one coalesced load
                              // instead of two scattered 4-byte reads through the permutation
};

__device__ __forceinline__ int plan_key(const PlanArgs& a, int64_t i) {
    double J, T;
    if (a.act_f64) parse_action<double>(((const double*)a.actions)[i], ((const double*)a.actions)[a.N + i], a.max_current, a.max_duration, J, T);
    else parse_action<float>(((const float*)a.actions)[i], ((const float*)a.actions)[a.N + i], a.max_current, a.max_duration, J, T);
    if (a.skip_done && (a.state[i].stepw & STG_DONE_BIT)) return (a.by_kind ? PLAN_BUCKETS : PLAN_DUR) - 1;
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
    // Grouped by device kind, the tile's slot sequence is one long-to-short ramp PER KIND, so its 64-slot blocks -- the step launch's
    // wavefronts, dealt longest rank first -- are not in order of work (the longest block of the last kind has rank ~43).  GROUPS OF
    // FOUR consecutive blocks (= one 4-wavefront workgroup of the step launch) are therefore renumbered by the pulse duration of their
    // first env (stable): rank order becomes longest-first again, which is what the step kernel's schedule (stg_slot_block) assumes,
    // while a workgroup's four wavefronts stay of one kind and of nearly one duration -- they have to finish together for their four
    // SIMD slots to come free together, and a sub-step of one kind costs up to 1.5x that of another (renumbering single blocks mixed
    // the kinds inside workgroups: 262 144 envs 0.72 -> 0.96 ms).  Only where it measured faster (tools/devphys_ab.py): launches of
    // 98 304 ... 131 072 envs with 4-wavefront workgroups, i.e. one and a half to two workgroups per CU, which the boustrophedon order
    // pairs long with short -- 131 072 envs 0.634 -> 0.508 ms, with the thermal field 1.51 -> 1.28 ms; 100 000 envs 0.623 -> 0.555 ms.
    // Elsewhere the kind-major order is as good or better: up to 65 536 envs nothing shares a SIMD; at 70 000 envs the few workgroups
    // of the second round land on the CUs holding the longest ones, which longest-first makes the costliest kind's (0.39 -> 0.60 ms);
    // with more rounds than fit 262 144 envs 0.72 against 0.82 ms, 1 048 576 envs 2.30 against 2.47; per-env records (64-thread
    // workgroups) 131 072 envs 0.557 against 0.745 ms.
    __shared__ int grp_d[TILE_WAVES / 4];
    __shared__ uint8_t grp_rank[TILE_WAVES / 4];
    // Kind-pure groups (regroup >= 2).  A kind's ramp rarely ends on a group boundary (256 slots), so one group per boundary held the
    // SHORT end of one kind's ramp and the LONG start of the next: a 4-wavefront workgroup whose wavefronts run 150 ... 1000 sub-steps
    // keeps its CU slot until the longest is through.  Each kind therefore keeps a whole number of groups of its longest envs in
    // place and hands its remainder (< 256 slots, its SHORTEST envs) to a common tail behind all kinds: every group but the (at most
    // three) tail groups is of one kind and one duration, and the tail groups are short throughout.
    __shared__ uint32_t k_start[3], k_main[3], k_main_off[3], k_tail_off[3];
    const bool kind_pure = a.by_kind && a.regroup >= 2;
    if (kind_pure) {
        if (tid == 0) {
            uint32_t main_off = 0, tail_tot = 0, cnt_k[3];
            for (int k = 0; k < 3; ++k) {
                const uint32_t s0 = start[k * PLAN_DUR] - cnt[k * PLAN_DUR], e0 = start[(k + 1) * PLAN_DUR - 1];
                k_start[k] = s0; cnt_k[k] = e0 - s0;
                k_main[k] = cnt_k[k] & ~255u;
                k_main_off[k] = main_off; main_off += k_main[k];
            }
            for (int k = 0; k < 3; ++k) { k_tail_off[k] = main_off + tail_tot; tail_tot += cnt_k[k] - k_main[k]; }
        }
        __syncthreads();
    }
    auto slot_of = [&](int r) -> uint32_t {
        uint32_t sl = (start[key[r]] - cnt[key[r]]) + rank[r];                                     // exclusive start of the bucket + rank in it
        if (kind_pure) {
            const int k = key[r] / PLAN_DUR;
            const uint32_t pos = sl - k_start[k];
            sl = pos < k_main[k] ? k_main_off[k] + pos : k_tail_off[k] + (pos - k_main[k]);
        }
        return sl;
    };
    if (a.by_kind && a.regroup) {
        constexpr int NG = TILE_WAVES / 4;
        if (tid < NG) grp_d[tid] = 0x7fffffff;                 // (groups beyond N: last)
        __syncthreads();
#pragma unroll
        for (int r = 0; r < PLAN_ITEMS; ++r)
            if (key[r] >= 0) {
                const uint32_t sl = slot_of(r);
                // regroup 1: by the pulse duration of the group's first env; 2 / 3: by the group's LONGEST block (a group that straddles the
                // boundary between two kinds holds the short end of one ramp and the long start of the next: keyed by its first env it sorts
                // last and its long block then runs at the very end) -- 2: estimated cost = duration x the kind's cost of a sub-step (measured
                // on homogeneous batches, profiles/r03_devphys_regroup.txt: STT 0.466, SOT 0.646, VCMA 0.419 ms per step -> 32 : 44 : 29),
                // 3: duration
                const int dq = key[r] % PLAN_DUR, kind = key[r] / PLAN_DUR;
                const int f = kind == 1 ? 44 : (kind == 2 ? 29 : 32);
                if (a.regroup == 1) { if ((sl & 255u) == 0u) grp_d[sl >> 8] = dq; }
                else if ((sl & 63u) == 0u) atomicMin(&grp_d[sl >> 8], a.regroup == 2 ? (1 << 20) - (PLAN_DUR - dq) * f : dq);
            }
        __syncthreads();
        // (a ragged tile's last, partly filled group keeps its place behind the full ones: the step kernel's "slot < N" test relies on
        // the tile's envs filling its first slots)
        const int64_t n_t = a.N - base < TILE_ENVS ? a.N - base : TILE_ENVS;
        if (tid == (int)(n_t >> 8) && (n_t & 255)) grp_d[tid] = 0x7ffffffe;
        __syncthreads();
        if (tid < NG) {
            const int d = grp_d[tid];
            int before = 0;
            for (int j = 0; j < NG; ++j) before += (grp_d[j] < d || (grp_d[j] == d && j < tid)) ? 1 : 0;
            grp_rank[tid] = (uint8_t)before;
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < PLAN_ITEMS; ++r) {
        const int64_t i = base + r * PLAN_THREADS + tid;
        if (key[r] >= 0) {
            uint32_t sl = slot_of(r);
            if (a.by_kind && a.regroup) sl = ((uint32_t)grp_rank[sl >> 8] << 8) | (sl & 255u);
            const int64_t slot = base + sl;
            a.perm[slot] = (uint32_t)i;
            if (a.act_f64) ((double2*)a.act_sorted)[slot] = make_double2(((const double*)a.actions)[i], ((const double*)a.actions)[a.N + i]);
            else ((float2*)a.act_sorted)[slot] = make_float2(((const float*)a.actions)[i], ((const float*)a.actions)[a.N + i]);
        }
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

// automatic lane refill of the RK45 step (stg_step_refill_kernel: persistent wavefronts sharing one global queue; measured on 81 921 ...
// 1 048 576 envs, profiles/r04_refill_global_ab.txt): 1024 wavefronts -- one per SIMD -- while that leaves at most 8 envs per lane (up
// to 524 288 envs), 2048 beyond.  From 131 073 envs at T = 0 K (up to there the one-env-per-lane launch with its two wavefronts per SIMD
// is as fast: 131 072 envs 1.96 against 1.94 ms, 98 304 envs 1.81 against 1.87) and from 98 305 envs with the thermal field (just
// above the hybrid wave-specialised launch: 98 304 envs hybrid 2.81 against 2.90 ms, 106 496 envs 3.08 against 2.93; up to 131 072 envs
// 3.1-3.2 -> 2.9-3.0 ms against one env per lane: there the launch is bound by its longest env at the inline-normal loop's
// lone-wavefront speed either way).  Attempts between refill points: 32 (16 with 2048 wavefronts).
constexpr int64_t STG_REFILL_AUTO_ENVS = 131073, STG_REFILL_AUTO_ENVS_THERMAL = 98305;
constexpr int32_t STG_REFILL_CHECK_DEFAULT = 32;
static inline void refill_auto(int64_t n, bool thermal, int& r, int64_t& nw) {
    r = 0; nw = 0;
    static const int64_t min_env = std::getenv("STG_REFILL_MIN") ? std::atoll(std::getenv("STG_REFILL_MIN")) : 0;   // (experiments)
    if (n < (min_env > 0 ? min_env : (thermal ? STG_REFILL_AUTO_ENVS_THERMAL : STG_REFILL_AUTO_ENVS))) return;
    const int64_t nblk = ((n + TILE_ENVS - 1) / TILE_ENVS) * TILE_WAVES;      // blocks of whole tiles (a ragged tile's empty blocks included)
    nw = 1024;
    int64_t rr = (nblk + nw - 1) / nw;
    if (rr > 8) { nw = 2048; rr = (nblk + nw - 1) / nw; }
    r = (int)(rr < 2 ? 2 : (rr > 0x7FFFFFFF ? 0x7FFFFFFF : rr));              // (envs per lane on average; only != 0 matters to the launch)
}
constexpr int32_t STG_WALK_TILES_DEFAULT = 1 << 20;   // all tiles of the group (fastest, see stg_slot_block)

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
    unsigned long long* refill_cursor = nullptr;   // the refill launches' two alternating sets of stripe cursors (each cursor in a 128-byte line of its own)
    uint64_t refill_seq = 0;                       // refill launches so far: launch k uses cursor k & 1 and zeroes the other
    uint32_t* placement = nullptr;    // [PLACEMENT_RING][PLACEMENT_WORDS]: where the wavefronts of the last launches ran (stg_get_placement)
    uint64_t launch_seq = 0;          // step launches so far
    uint32_t* perm = nullptr;
    void* act_sorted = nullptr;       // [N][2] actions of the step in slot order (written by the plan kernel)
    bool have_params = false, have_state = false;
    bool axis_z = false;              // every class has easy axis = +z exactly: the specialised Simple RHS applies
    bool axis_z_llgs = false;         // every class has raw easy axis (0,0,rz) and demag (0,0,Nz): specialised LLGS RHS
    // per-env parameters (stg_set_params_per_env): library-owned copies
    double* env_soa = nullptr;        // records [N][env_layout_doubles(env_layout)] (stg_kernels.hpp: EnvParams)
    int32_t env_layout = ENV_LAYOUT_CORE;
    uint8_t* env_type = nullptr;      // [N]: the device kind by itself, for the plan kernel (read in env order)
    bool per_env = false;
    int32_t walk_tiles = STG_WALK_TILES_DEFAULT;   // sorted schedule: tiles an XCD group keeps in flight (stg_slot_block)
    int32_t spread_max = 256;                      // STG_SPREAD_MAX (experiments), see StepArgs
    int32_t hybrid = 1;                            // STG_HYBRID=0 switches the hybrid wave-specialised launch off (experiments)
    int32_t hybrid_min = -1;                       // fewest producer/consumer pairs for which the hybrid launch is used (STG_HYBRID_MIN; -1: by solver)
    int32_t refill = -1, refill_check = STG_REFILL_CHECK_DEFAULT;        // STG_REFILL experiment override of cfg.lane_refill (-1: none)
};

static int32_t walk_tiles_from_env() {
    // experiment knob (A/B runs of the schedule): STG_WALK_TILES=<n>; results never depend on it
    const char* e = std::getenv("STG_WALK_TILES");
    const int v = e ? std::atoi(e) : 0;
    int32_t w = v > 0 ? v : STG_WALK_TILES_DEFAULT;
    const char* sn = std::getenv("STG_SNAKE");       // unset: automatic (see stg_slot_block)
    if (sn) w |= (int32_t)(std::atoi(sn) != 0 ? STG_WALK_SNAKE_ON : STG_WALK_SNAKE_OFF);
    // STG_SNAKE_ROUNDS=<n> (with STG_SNAKE=1): the boustrophedon applies to the first n rounds only (0 = every round)
    if (const char* sr = std::getenv("STG_SNAKE_ROUNDS")) w |= (int32_t)(((unsigned)std::atoi(sr) & 0xFFu) << STG_WALK_ROUNDS_SHIFT);
    return w;
}

static EnvParams env_params_of(const stg_ctx* ctx) {
    EnvParams e{};
    if (ctx->per_env) e.soa = ctx->env_soa;
    e.layout = ctx->env_layout;
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
    if (c->out_layout != STG_OUT_SOA && c->out_layout != STG_OUT_RECORDS) return fail(STG_E_INVALID, "cfg.out_layout must be STG_OUT_SOA or STG_OUT_RECORDS");
    if (c->lane_refill < -1 || c->lane_refill == 1 || c->lane_refill > 1024) return fail(STG_E_INVALID, "cfg.lane_refill must be -1 (never), 0 (automatic) or the number of envs per lane (2..1024)");
    if (c->reserved0 != 0) return fail(STG_E_INVALID, "cfg.reserved0 must be 0");
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
    // per-env parameter records hold what this context's kernels read: their solver and torque model are fixed here
    c->env_layout = cfg->solver == STG_SOLVER_RK45 ? ENV_LAYOUT_LLGS : (cfg->torque_model == 1 ? ENV_LAYOUT_DEV : ENV_LAYOUT_CORE);
    c->walk_tiles = walk_tiles_from_env();
    if (const char* e = std::getenv("STG_HYBRID")) c->hybrid = std::atoi(e);
    if (const char* e = std::getenv("STG_HYBRID_MIN")) c->hybrid_min = std::atoi(e);
    if (const char* e = std::getenv("STG_SPREAD_MAX")) c->spread_max = std::atoi(e);
    if (const char* e = std::getenv("STG_REFILL")) {
        int r = 0, chk = 0;
        if (std::sscanf(e, "%d,%d", &r, &chk) >= 1) { c->refill = r; if (chk > 0) c->refill_check = chk; }
    }
    // one slab: the state records, the class table, the counter stripes, the lane permutation (each 256-B aligned)
    auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
    const size_t N = (size_t)n_envs;
    const size_t rs = al(N * sizeof(EnvRec)), r4 = al(N * 4), ra = al(N * 16);
    const size_t rp = al(sizeof(uint32_t) * PLACEMENT_RING * PLACEMENT_WORDS);
    const size_t rc = al(2 * REFILL_STRIPES * REFILL_CURSOR_STRIDE * sizeof(unsigned long long));
    const size_t total = rs + r4 + ra + rp + rc + al(sizeof(double) * STG_MAX_CLASSES * C_COUNT) +
                         COUNTER_STRIPES * COUNTER_STRIDE * sizeof(unsigned long long);
    hipError_t e = hipMalloc(&c->slab, total);
    if (e != hipSuccess) { delete c; return fail(STG_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    e = hipMemset(c->slab, 0, total);
    if (e != hipSuccess) { (void)hipFree(c->slab); delete c; return fail(STG_E_HIP, std::string("hipMemset: ") + hipGetErrorString(e)); }
    char* p = (char*)c->slab;
    c->s.rec = (EnvRec*)p; p += rs;
    c->ctab = (double*)p; p += al(sizeof(double) * STG_MAX_CLASSES * C_COUNT);
    c->counters = (unsigned long long*)p; p += COUNTER_STRIPES * COUNTER_STRIDE * sizeof(unsigned long long);
    c->perm = (uint32_t*)p; p += r4;
    c->act_sorted = (void*)p; p += ra;
    c->placement = (uint32_t*)p; p += rp;
    c->refill_cursor = (unsigned long long*)p; p += rc;
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
// rows of the caller's structure of arrays -> the library's per-env records (once per stg_set_params_per_env)
__global__ void stg_env_pack_kernel(const double* soa, const uint8_t* type, const uint8_t* valid, int64_t N, double* rec, int layout) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    // rows of the caller's block = the double fields of stg_device_params in declaration order (include/spintorque_hip.h):
    // 0 damping, 1 ms, 2 ku, 3 volume, 4 polarization, 5-7 easy_axis, 8-10 demag, 11 a_ex, 12 area, 13 r_p, 14 r_ap, 15-17 ref_m,
    // 18 r_series, 19 sot_tau_dl, 20 sot_tau_fl, 21-23 sot_sigma, 24 vcma_xi, 25 vcma_td, 26 vcma_vbd, 27-29 shape_demag
    constexpr int core[15] = {0, 1, 2, 3, 4, 5, 6, 7, 12, 13, 14, 15, 16, 17, 18};
    constexpr int llgs[4] = {8, 9, 10, 11};
    constexpr int dev[8] = {19, 20, 21, 22, 23, 24, 25, 26};
    double* r = rec + i * env_layout_doubles(layout);
    for (int k = 0; k < 15; ++k) r[k] = soa[(int64_t)core[k] * N + i];
    r[15] = (double)((int)(type[i] & 3) + 4 * (valid[i] ? 1 : 0));
    if (layout == ENV_LAYOUT_LLGS) for (int k = 0; k < 4; ++k) r[16 + k] = soa[(int64_t)llgs[k] * N + i];
    if (layout == ENV_LAYOUT_DEV) for (int k = 0; k < 8; ++k) r[16 + k] = soa[(int64_t)dev[k] * N + i];
}

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
        // both buffers or neither: a context is never left with one of them set
        double* soa = nullptr;
        uint8_t* bytes = nullptr;
        HIP_TRY(hipMalloc(&soa, sizeof(double) * env_layout_doubles(ctx->env_layout) * N));
        const hipError_t e2 = hipMalloc(&bytes, N);
        if (e2 != hipSuccess) {
            (void)hipFree(soa);
            return fail(STG_E_NOMEM, std::string("hipMalloc(env_type): ") + hipGetErrorString(e2));
        }
        ctx->env_soa = soa;
        ctx->env_type = bytes;
    }
    hipLaunchKernelGGL(stg_env_pack_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, 0, soa_params, dev_type, params_valid,
                       (int64_t)N, ctx->env_soa, (int)ctx->env_layout);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(ctx->env_type, dev_type, N, hipMemcpyDeviceToDevice));
    // the specialised right-hand sides apply only if EVERY env has the default axis geometry
    int32_t h_flag[2] = {0, 0};
    int32_t* d_flag = nullptr;
    HIP_TRY(hipMalloc(&d_flag, 8));
    hipError_t ef = hipMemset(d_flag, 0, 8);
    if (ef == hipSuccess) {
        hipLaunchKernelGGL(stg_env_axis_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, 0, soa_params, (int64_t)N, d_flag);
        ef = hipMemcpy(h_flag, d_flag, 8, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_flag);                                  // on every path
    if (ef != hipSuccess) return fail(STG_E_HIP, std::string("axis flags: ") + hipGetErrorString(ef));
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
    if (ctx->cfg.out_layout == STG_OUT_RECORDS && ((uintptr_t)obs_out & 7u))
        return fail(STG_E_INVALID, "the record array must be 8-byte aligned");
    HIP_TRY(hipSetDevice(ctx->device));
    ResetArgs a{};
    a.s = ctx->s; a.c = cfg_view(ctx->cfg); a.N = ctx->N; a.env_id0 = ctx->env_id0;
    a.ctab = ctx->ctab; a.cls = ctx->cls; a.ncls = ctx->ncls; a.ep = env_params_of(ctx);
    a.mask = mask; a.init_m = init_m; a.target = target; a.seed = seed; a.obs = obs_out;
    a.records = ctx->cfg.out_layout == STG_OUT_RECORDS ? 1 : 0;
    hipLaunchKernelGGL(stg_reset_kernel, grid_for(ctx->N), dim3(64), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    ctx->have_state = true;
    return STG_OK;
}

int stg_step_many(stg_ctx* ctx, int32_t K, const void* actions, int32_t act_f64, int32_t out_every, int32_t autoreset,
                  float* obs, float* final_obs, float* reward, double* reward_f64, double* energy, uint8_t* terminated,
                  uint8_t* truncated, uint8_t* status, void* stream) {
    if (!ctx) return fail(STG_E_INVALID, "ctx is NULL");
    if (!ctx->have_params || !ctx->have_state) return fail(STG_E_STATE, "stg_set_params and stg_reset must precede stg_step");
    if (K < 1) return fail(STG_E_INVALID, "K must be >= 1");
    const bool records = ctx->cfg.out_layout == STG_OUT_RECORDS;
    if (!actions || !obs) return fail(STG_E_INVALID, "actions/obs must not be NULL");
    if (!records && (!reward || !terminated || !truncated)) return fail(STG_E_INVALID, "reward/terminated/truncated must not be NULL (cfg.out_layout = STG_OUT_SOA)");
    if (records && ((uintptr_t)obs & 7u)) return fail(STG_E_INVALID, "the record array must be 8-byte aligned");
    if (records && ((uintptr_t)final_obs & 7u)) return fail(STG_E_INVALID, "final_obs must be 8-byte aligned (cfg.out_layout = STG_OUT_RECORDS)");
    HIP_TRY(hipSetDevice(ctx->device));
    StepArgs a{};
    a.s = ctx->s; a.c = cfg_view(ctx->cfg); a.N = ctx->N; a.env_id0 = ctx->env_id0;
    a.ctab = ctx->ctab; a.cls = ctx->cls; a.ncls = ctx->ncls; a.ep = env_params_of(ctx);
    a.counters = ctx->counters;
    a.placement = ctx->placement + (size_t)(ctx->launch_seq % PLACEMENT_RING) * PLACEMENT_WORDS;
    ctx->launch_seq += 1;
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
        pa.state = ctx->s.rec; pa.skip_done = (ctx->cfg.skip_done && !autoreset) ? 1 : 0;
        pa.perm = ctx->perm; pa.act_sorted = ctx->act_sorted;
        pa.cls = ctx->cls; pa.ctab = ctx->ctab;
        pa.env_type = ctx->per_env ? ctx->env_type : nullptr;
        pa.by_kind = (ctx->cfg.torque_model == 1 && ((ctx->ncls > 1 && ctx->cls) || ctx->per_env)) ? 1 : 0;
        // (measured range, tools/devphys_ab.py: 4-wavefront workgroups, between one and a half and two workgroups per CU)
        // device-physics model with a class table: kind-pure groups of four blocks (= whole workgroups of the step launch), dealt by the
        // estimated cost of their longest block (round 4, profiles/r04_devphys_order_ab.txt: 1.08-1.37x at 98 304 ... 1 048 576 envs;
        // rounds 2-3 keyed the groups by their first env and only at 98 304 ... 131 072 envs).  Per-env parameter records launch
        // one-wavefront workgroups: no groups there.  STG_REGROUP=0/1/2/3 forces a mode (experiments).
        pa.regroup = (pa.by_kind && !ctx->per_env) ? 2 : 0;
        if (const char* rg = std::getenv("STG_REGROUP")) pa.regroup = std::atoi(rg);
        const dim3 g((unsigned)((ctx->N + TILE_ENVS - 1) / TILE_ENVS));
        hipLaunchKernelGGL(stg_plan_tile_kernel, g, dim3(PLAN_THREADS), 0, st, pa);
        a.perm = ctx->perm;
        a.act_sorted = ctx->act_sorted;
    }
    a.actions = actions; a.K = K; a.out_every = out_every ? 1 : 0; a.autoreset = autoreset ? 1 : 0;
    a.records = records ? 1 : 0;
    a.walk = ctx->walk_tiles;
    a.spread_max = ctx->spread_max;
    a.obs = obs; a.final_obs = final_obs; a.reward = reward; a.reward64 = reward_f64; a.energy = energy; a.term = terminated; a.trunc = truncated; a.status = status;
    // the Simple solver only draws a thermal field when temperature > 0 (simple_solver.py:321,378)
    const bool thermal = ctx->cfg.thermal && ctx->cfg.temperature > 0;
    const int multi = ctx->per_env ? 2 : (ctx->ncls > 1 ? 1 : 0);      // (2: every lane derives its constants from its env's record, in registers)
    const bool devphys = ctx->cfg.torque_model == 1;
    a.force_wg1 = 0;
    // wave_spec: 0 = automatic (thermal launches of at most STG_WAVE_SPEC_MAX_ENVS envs, i.e. latency-bound ones),
    // 1 = always, -1 = never.  Results do not depend on it.
    bool pc = ctx->cfg.wave_spec > 0 || (ctx->cfg.wave_spec == 0 && ctx->N <= STG_WAVE_SPEC_MAX_ENVS);
    // hybrid (RK45 / RK4 + thermal, sorted schedule, 65 536 < N <= 131 072, automatic mode): 1024 two-wavefront workgroups -- producer /
    // consumer pairs for the 2048 - nblk longest blocks, two blocks with inline normals in each of the others (stg_kernels.hpp:
    // stg_hybrid_block); experiment knobs STG_HYBRID=0/1, STG_HYBRID_MIN=<fewest pairs worth it>
    a.hybrid = 0;
    if ((ctx->cfg.solver == STG_SOLVER_RK45 || (ctx->cfg.solver == STG_SOLVER_RK4 && thermal && !devphys)) &&
        ctx->cfg.thermal && ctx->cfg.wave_spec == 0 && a.perm && !ctx->per_env &&
        ctx->N > STG_WAVE_SPEC_MAX_ENVS && ctx->hybrid != 0) {
        const int64_t nblk = ((ctx->N + TILE_ENVS - 1) / TILE_ENVS) * TILE_WAVES;      // blocks of whole tiles
        // (a) up to 131 072 envs: pairs for the 2048 - nblk longest blocks, two fixed blocks in each other workgroup -- measured
        // (profiles/r04_hybrid_range_ab.txt) ahead of the alternatives down to 512 pairs (RK45, 98 304 envs) / 640 pairs (RK4, 90 112 envs)
        const int64_t n_pair = 2048 - nblk;
        const int64_t min_pairs = ctx->hybrid_min >= 0 ? ctx->hybrid_min : (ctx->cfg.solver == STG_SOLVER_RK45 ? 512 : 640);
        if (ctx->N <= 2 * STG_WAVE_SPEC_MAX_ENVS && n_pair >= min_pairs) {
            pc = true; a.hybrid = (int32_t)n_pair + 1; a.hybrid_prio = getenv("STG_HYB_PRIO") ? atoi(getenv("STG_HYB_PRIO")) : 1;
        }
    }
    // lane refill (RK45 throughput launches, see stg_step_refill_kernel).  cfg.lane_refill: 0 = automatic, -1 never, >= 2 forced;
    // experiment knob STG_REFILL=<envs per lane>[,<attempts between refill points>] overrides the configuration
    a.refill = 0; a.refill_check = STG_REFILL_CHECK_DEFAULT; a.refill_nw = 0;
    if (ctx->cfg.solver == STG_SOLVER_RK45 && K == 1 && !ctx->per_env) {
        const int64_t nblk = ((ctx->N + TILE_ENVS - 1) / TILE_ENVS) * TILE_WAVES;
        int r = 0, chk = STG_REFILL_CHECK_DEFAULT;
        int64_t nw = 0;
        if (ctx->cfg.lane_refill == 0) {
            refill_auto(ctx->N, ctx->cfg.thermal != 0, r, nw);
            // (two wavefronts per SIMD: a refill point every 16 attempts -- 1 048 576 envs 13.3 against 13.6 ms; with one per SIMD 16 ... 64
            // are alike, 8 and 128 worse: tools/refill_check_sweep.py)
            if (nw >= 2048) chk = 16;
        }
        else if (ctx->cfg.lane_refill > 0) { r = ctx->cfg.lane_refill; nw = (nblk + r - 1) / r; }
        if (ctx->refill >= 0) { r = ctx->refill; chk = ctx->refill_check; nw = r >= 2 ? (nblk + r - 1) / r : 0; }
        // (not combined with the wave-specialised launch: a forced wave_spec = 1 keeps the one-env-per-lane kernel)
        if (r >= 2 && nw >= 1 && !(ctx->cfg.thermal && ctx->cfg.wave_spec > 0)) {
            if (nw > 0x7FFFFFFFll) return fail(STG_E_INVALID, "lane refill: too many wavefronts");
            a.refill = r; a.refill_check = chk > 0 ? chk : STG_REFILL_CHECK_DEFAULT; a.refill_nw = (int32_t)nw;
            // two cursors alternate: this launch finds its own at 0 (zeroed by the previous refill launch, or by stg_create) and
            // zeroes the next one's -- launches of a context are ordered on their stream
            a.refill_cursor = ctx->refill_cursor + (ctx->refill_seq & 1) * (REFILL_STRIPES * REFILL_CURSOR_STRIDE);
            a.refill_cursor_next = ctx->refill_cursor + ((ctx->refill_seq + 1) & 1) * (REFILL_STRIPES * REFILL_CURSOR_STRIDE);
            ctx->refill_seq += 1;
            stg_dispatch_step_rk45_refill(a, ctx->cfg.thermal != 0, multi != 0, ctx->axis_z_llgs, act_f64, st);
            HIP_TRY(hipGetLastError());
            return STG_OK;
        }
    }
    switch (ctx->cfg.solver) {
        case STG_SOLVER_RK4: stg_dispatch_step_rk4(a, thermal, multi, ctx->axis_z, devphys, act_f64, pc, st); break;
        case STG_SOLVER_EULER: stg_dispatch_step_euler(a, thermal, multi, ctx->axis_z, devphys, act_f64, pc, st); break;
        default: stg_dispatch_step_rk45(a, ctx->cfg.thermal != 0, multi, ctx->axis_z_llgs, act_f64, pc, st); break;
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
                        int32_t traj_cap, double* tt, double* tm, double* te, double* ttq, double* m_final, int32_t* n_points,
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
    a.traj_cap = traj_cap; a.traj_t = tt; a.traj_m = tm; a.traj_e = te; a.traj_tq = ttq;
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
    return solve_common(ctx, m0, J, T, env_step, 0, nullptr, nullptr, nullptr, nullptr, m_final, n_points, success, stream, false);
}

int stg_solve_traj(stg_ctx* ctx, const double* m0, const double* J, const double* T, uint32_t env_step, int32_t traj_cap,
                   double* t, double* m, double* energy, double* torques, double* m_final, int32_t* n_points, uint8_t* success,
                   void* stream) {
    return solve_common(ctx, m0, J, T, env_step, traj_cap, t, m, energy, torques, m_final, n_points, success, stream, true);
}

int stg_get_state(stg_ctx* ctx, double* m, double* target, double* total_energy, int32_t* step_count, uint32_t* rng_step,
                  uint8_t* done, void* stream) {
    if (!ctx) return fail(STG_E_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(stg_state_copy_kernel<false>, dim3((unsigned)((ctx->N + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       ctx->s, ctx->N, m, target, total_energy, step_count, rng_step, done);
    HIP_TRY(hipGetLastError());
    return STG_OK;
}

int stg_set_state(stg_ctx* ctx, const double* m, const double* target, const double* total_energy,
                  const int32_t* step_count, const uint32_t* rng_step, const uint8_t* done, void* stream) {
    if (!ctx) return fail(STG_E_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(stg_state_copy_kernel<true>, dim3((unsigned)((ctx->N + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       ctx->s, ctx->N, const_cast<double*>(m), const_cast<double*>(target), const_cast<double*>(total_energy),
                       const_cast<int32_t*>(step_count), const_cast<uint32_t*>(rng_step), const_cast<uint8_t*>(done));
    HIP_TRY(hipGetLastError());
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

int stg_get_placement(stg_ctx* ctx, int32_t launches_back, uint32_t* out, int32_t cap, int32_t* n_workgroups, int32_t* waves_per_workgroup) {
    if (!ctx || !out || cap < 1) return fail(STG_E_INVALID, "ctx/out is NULL or cap < 1");
    if (launches_back < 0 || launches_back >= PLACEMENT_RING || (uint64_t)launches_back >= ctx->launch_seq)
        return fail(STG_E_INVALID, "launches_back must address one of the last launches (at most 32 are kept)");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    static thread_local uint32_t h[PLACEMENT_WORDS];
    const uint64_t seq = ctx->launch_seq - 1 - (uint64_t)launches_back;
    HIP_TRY(hipMemcpy(h, ctx->placement + (size_t)(seq % PLACEMENT_RING) * PLACEMENT_WORDS, sizeof(h), hipMemcpyDeviceToHost));
    const int64_t waves = (int64_t)h[0] * (int64_t)h[1];
    int32_t n = (int32_t)(waves < PLACEMENT_CAP ? waves : PLACEMENT_CAP);
    if (n > cap / STG_PLACEMENT_WORDS_PER_WAVE) n = cap / STG_PLACEMENT_WORDS_PER_WAVE;
    if (n_workgroups) *n_workgroups = (int32_t)h[0];
    if (waves_per_workgroup) *waves_per_workgroup = (int32_t)h[1];
    static_assert(STG_PLACEMENT_WORDS_PER_WAVE == PLACEMENT_ENTRY, "entry size of the placement table");
    std::memcpy(out, h + 2, sizeof(uint32_t) * (size_t)n * PLACEMENT_ENTRY);
    return n;
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
