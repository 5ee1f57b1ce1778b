// stg_array.hip -- SpinTorqueArray-v0 step path: N independent R x C device arrays per launch.
//
// Reference: spin_torque_gym/envs/array_env.py -- step :362-409, _apply_action :411-476, _compute_effective_field :478-494,
// _simulate_device_dynamics :496-521, _compute_pattern_similarity :523-531, _get_observation :533-569,
// default reward :183-224, coupling matrix :301-334.
//
// Unlike SpinTorque-v0 this env is memory-shaped: a step touches every device's magnetisation (coupling sum, similarity,
// observation) but integrates only the addressed devices with ten Euler sub-steps of ONE derivative -- ~1 KB of HBM
// traffic per array-step against a few hundred flops in 'individual' mode.  Layout: one array per lane; the array's
// pattern lives in LDS for the duration of the step (lane-fastest: [device*3+component][64 lanes], conflict-free
// ds_read_b64), because the addressed devices update sequentially and each sees its predecessors' new states (the
// reference updates current_pattern in place).  Global state is SoA [device][component][N] so every wavefront access is
// a coalesced 512-B row.
#include "../../include/spintorque_hip.h"
#include "stg_physics.hpp"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

using namespace stg;

namespace {

constexpr int ARR_MAX_DEV = 64;

struct ArrDev {            // per-class constants of the (single) device class of the array
    double hk, ms;
    double ex, ey, ez;     // raw easy axis
    double nx, ny, nz;     // shape demag factors (SOT/VCMA)
    double area, r_p, r_ap, tmr, refx, refy, refz, r_series;
    int32_t dev_type;
};

struct ArrArgs {
    int64_t N;
    int32_t rows, cols, mode, include_coupling, max_steps, obs_mode;
    double max_current, max_duration, thr, w_energy, temperature;
    ArrDev dev;
    const double* coupling;        // [n][n] device memory
    double *pattern, *target;      // [n][3][N]
    double* etot;                  // [N]
    int32_t* step;                 // [N]
    const float* actions;          // [A][N]
    float* obs;                    // [obs_dim][N]
    float* reward;
    double *reward64, *energy;
    uint8_t *term, *trunc;
};

// device.compute_effective_field(m, 0): stt_mram.py:55-76 (re-normalised m, anisotropy only); sot_mram.py:79-112 and
// vcma_mram.py:85-120 (raw m, anisotropy + shape demagnetisation; thermal/DMI/exchange terms are zero there)
__device__ __forceinline__ V3 device_field(const V3& m_in, const ArrDev& d) {
    if (d.dev_type == STG_DEV_STT) {
        const double inv = rsqrt_fast(dot(m_in, m_in));
        const V3 m{m_in.x * inv, m_in.y * inv, m_in.z * inv};
        const double c = d.hk * (m.x * d.ex + m.y * d.ey + m.z * d.ez);
        return V3{c * d.ex, c * d.ey, c * d.ez};
    }
    const double c = d.hk * (m_in.x * d.ex + m_in.y * d.ey + m_in.z * d.ez);
    return V3{c * d.ex - d.ms * d.nx * m_in.x, c * d.ey - d.ms * d.ny * m_in.y, c * d.ez - d.ms * d.nz * m_in.z};
}

// Observation rows: 'array' mode [R,C,6] = (pattern, target) per cell; 'vector' mode = flattened pattern, flattened
// target, then 4 global values (array_env.py:533-557).
__device__ __forceinline__ int64_t obs_row_pattern(int obs_mode, int n, int d, int k) { return obs_mode == 0 ? d * 6 + k : d * 3 + k; }
__device__ __forceinline__ int64_t obs_row_target(int obs_mode, int n, int d, int k) { return obs_mode == 0 ? d * 6 + 3 + k : 3 * n + d * 3 + k; }

// NDEV > 0: the number of cells is a compile-time constant and the coupling sum unrolls (4 x 4, the registered
// SpinTorqueArray-v0: 'global' mode 0.178 -> 0.167 ms per launch at 262 144 arrays, 'row' / 'column' unchanged).
// (Round 3, measured and rejected: a software-pipelined sweep that forms the coupling sum of the NEXT addressed cell -- all
// terms but the current cell's -- before the current cell's ten dependent Euler sub-steps and adds that one term afterwards:
// 0.169-0.172 ms against 0.167-0.168 ms for the plain unrolled form; the launch is not bound by that dependency chain.)
template <int NDEV>
__global__ void __launch_bounds__(64) stg_array_step_kernel(const ArrArgs a) {
    extern __shared__ double lds[];           // pattern [n*3][64] then coupling [n*n]
    const int n = NDEV > 0 ? NDEV : a.rows * a.cols;
    double* lp = lds;
    double* lc = lds + (size_t)n * 3 * 64;
    const int lane = threadIdx.x;
    if (a.include_coupling) {
        for (int q = lane; q < n * n; q += 64) lc[q] = a.coupling[q];
    }
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    const bool in_range = i < a.N;
    const int64_t N = a.N;
    // ONE pass over the state: pattern -> LDS, target -> its observation rows, and the similarity sum on the fly
    // (np.mean of the per-cell dot products, array_env.py:523-531); the target is not touched again except for the
    // addressed cells.
    double sim_sum = 0.0;
    if (in_range) {
        for (int d = 0; d < n; ++d) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double pv = a.pattern[(int64_t)(d * 3 + k) * N + i];
                const double tv = a.target[(int64_t)(d * 3 + k) * N + i];
                lp[(d * 3 + k) * 64 + lane] = pv;
                a.obs[obs_row_target(a.obs_mode, n, d, k) * N + i] = (float)tv;
                sim_sum += pv * tv;
            }
        }
    }
    __syncthreads();
    if (!in_range) return;
    const double prev_sim = sim_sum / n;                                               // array_env.py:372-373
    // _apply_action (array_env.py:411-445).  In 'global' mode the action is [current, duration]: action[1] (the
    // duration) is what the reference reads as the current density, and the duration defaults to 1 ns -- kept as is.
    const int n_act = a.mode == 3 ? 2 : 3;
    double J = (double)a.actions[N + i];
    double T = n_act > 2 ? (double)a.actions[2 * N + i] : 1e-9;
    J = isnan(J) ? J : fmin(fmax(J, -a.max_current), a.max_current);
    T = isnan(T) ? T : fmin(fmax(T, 1e-12), a.max_duration);
    int first = 0, count = n, stride = 1;
    if (a.mode != 3) {
        const double lim = a.mode == 0 ? n - 1 : (a.mode == 1 ? a.rows - 1 : a.cols - 1);
        const double a0 = (double)a.actions[i];
        // int(np.clip(action[0], 0, lim)); a NaN index raises in the reference -- here it addresses nothing
        const int idx = isnan(a0) ? -1 : (int)fmin(fmax(a0, 0.0), lim);
        if (a.mode == 0) { first = idx; count = 1; stride = 1; }
        else if (a.mode == 1) { first = idx * a.cols; count = a.cols; stride = 1; }
        else { first = idx; count = a.rows; stride = a.cols; }
        if (idx < 0) count = 0;
    }
    const bool drive = fabs(J) > 1e-12;                                                // array_env.py:506
    double e_total = 0.0;
    if (drive) {
        // The addressed cells update sequentially and each sees its predecessors' new states through the coupling sum
        // (the reference updates current_pattern in place, array_env.py:447-476).
#pragma unroll 1
        for (int q = 0; q < count; ++q) {
            const int d = first + q * stride;
            const V3 m0{lp[(d * 3) * 64 + lane], lp[(d * 3 + 1) * 64 + lane], lp[(d * 3 + 2) * 64 + lane]};
            V3 h = device_field(m0, a.dev);
            if (a.include_coupling) {                                                  // array_env.py:485-492
                V3 hc{0.0, 0.0, 0.0};
#pragma unroll
                for (int j = 0; j < (NDEV > 0 ? NDEV : n); ++j) {
                    const double c = (j == d) ? 0.0 : lc[d * n + j];
                    hc = V3{hc.x + c * lp[(j * 3) * 64 + lane], hc.y + c * lp[(j * 3 + 1) * 64 + lane],
                            hc.z + c * lp[(j * 3 + 2) * 64 + lane]};
                }
                h = V3{h.x + hc.x, h.y + hc.y, h.z + hc.z};
            }
            // _simulate_device_dynamics (array_env.py:496-521): alpha = 0.01, gamma = 2.21e5, p_hat = z
            const V3 mxp{m0.y, -m0.x, 0.0};
            const V3 t2 = cross(m0, mxp);
            const double tj = 0.1 * J;
            const V3 mxh = cross(m0, h);
            V3 dm{-2.21e5 * mxh.x, -2.21e5 * mxh.y, -2.21e5 * mxh.z};
            const V3 mxdm = cross(m0, dm);
            dm = V3{dm.x + 0.01 * mxdm.x + tj * t2.x, dm.y + 0.01 * mxdm.y + tj * t2.y, dm.z + 0.01 * mxdm.z + tj * t2.z};
            const double dt = T / 10;
            V3 m = m0;
#pragma unroll
            for (int it = 0; it < 10; ++it) {
                m = V3{m.x + dm.x * dt, m.y + dm.y * dt, m.z + dm.z * dt};
                const double inv = rsqrt_fast(dot(m, m));       // m / |m| (array_env.py:518), <= 2 ulp per component
                m = V3{m.x * inv, m.y * inv, m.z * inv};
            }
            lp[(d * 3) * 64 + lane] = m.x; lp[(d * 3 + 1) * 64 + lane] = m.y; lp[(d * 3 + 2) * 64 + lane] = m.z;
            a.pattern[(int64_t)(d * 3) * N + i] = m.x;
            a.pattern[(int64_t)(d * 3 + 1) * N + i] = m.y;
            a.pattern[(int64_t)(d * 3 + 2) * N + i] = m.z;
            // similarity: only this cell's dot product changed
            const V3 tg{a.target[(int64_t)(d * 3) * N + i], a.target[(int64_t)(d * 3 + 1) * N + i], a.target[(int64_t)(d * 3 + 2) * N + i]};
            sim_sum += dot(m, tg) - dot(m0, tg);
            // energy with the resistance of the UPDATED state (current_m is a view of the pattern, array_env.py:455-463)
            const V3 ref{a.dev.refx, a.dev.refy, a.dev.refz};
            const double r = resistance(m, a.dev.dev_type, a.dev.r_p, a.dev.r_ap, a.dev.tmr, ref, a.dev.r_series);
            const double v = J * r * a.dev.area;
            e_total += (v * v) / r * T;
        }
    }
    const double etot = a.etot[i] + e_total;
    const int32_t step = a.step[i] + 1;
    a.etot[i] = etot;
    a.step[i] = step;
    const double sim = sim_sum / n;
    const bool is_success = sim >= a.thr;
    // pattern rows of the observation + uniformity (1 - population std of the cell magnitudes, array_env.py:216-224)
    double mean = 0.0;
    for (int d = 0; d < n; ++d) {
        const V3 m{lp[(d * 3) * 64 + lane], lp[(d * 3 + 1) * 64 + lane], lp[(d * 3 + 2) * 64 + lane]};
        a.obs[obs_row_pattern(a.obs_mode, n, d, 0) * N + i] = (float)m.x;
        a.obs[obs_row_pattern(a.obs_mode, n, d, 1) * N + i] = (float)m.y;
        a.obs[obs_row_pattern(a.obs_mode, n, d, 2) * N + i] = (float)m.z;
        mean += sqrt(dot(m, m));
    }
    mean /= n;
    double var = 0.0;
    for (int d = 0; d < n; ++d) {
        const V3 m{lp[(d * 3) * 64 + lane], lp[(d * 3 + 1) * 64 + lane], lp[(d * 3 + 2) * 64 + lane]};
        const double dv = sqrt(dot(m, m)) - mean;
        var += dv * dv;
    }
    if (a.obs_mode == 1) {
        a.obs[(int64_t)(6 * n + 0) * N + i] = (float)sim;
        a.obs[(int64_t)(6 * n + 1) * N + i] = (float)((double)(a.max_steps - step) / (double)a.max_steps);
        a.obs[(int64_t)(6 * n + 2) * N + i] = (float)(etot / 1e-12);
        a.obs[(int64_t)(6 * n + 3) * N + i] = (float)(a.temperature / 300.0);
    }
    const double uniformity = fmax(0.0, 1.0 - sqrt(var / n));
    // default reward (array_env.py:183-224): pattern match, energy (sign as written), progress, uniformity
    double reward = 10.0 * (is_success ? 10.0 : sim * 5.0);
    reward += (-a.w_energy) * (-e_total / 1e-12);
    reward += (sim - prev_sim);
    reward += 2.0 * uniformity;
    a.reward[i] = (float)reward;
    if (a.reward64) a.reward64[i] = reward;
    if (a.energy) a.energy[i] = e_total;
    a.term[i] = is_success ? 1 : 0;
    a.trunc[i] = step >= a.max_steps ? 1 : 0;
}

// 'global' action mode on NDEV cells (4 x 4, the registered SpinTorqueArray-v0): every cell is addressed, in order.  The whole pattern
// lives in REGISTERS (48 doubles) instead of a 24.6 KB per-wavefront copy in LDS: that copy bounded the general kernel at six
// single-wavefront workgroups per CU (1.5 wavefronts per SIMD) for a launch that is a chain of 160 dependent normalise-and-step
// iterations per lane -- latency-bound.  Registers cannot be indexed by the running cell, so the register file is ROTATED instead: at
// the top of iteration d, pm[0] is cell d and pm[k] cell (d + k) mod n; after the update everything moves down one place (48 64-bit
// moves per cell) and after n iterations it is back in place.  The coupling matrix is staged in LDS in that rotated order
// (lc[d][k] = C[d][(d + k) mod n]), so a row is sixteen consecutive doubles.  All row accesses to the pattern / target / observation
// arrays are BUFFER instructions: descriptor and row offset in scalar registers, one 32-bit lane offset in a VGPR -- no per-row 64-bit
// VGPR addresses.  (A fully unrolled sweep, the other way to keep the pattern in registers, needed more than 340 of them: one
// wavefront per SIMD.)  The coupling sum runs over the cells in rotated order (d+1, ..., n-1, 0, ..., d-1) instead of 0 ... n-1: the
// only arithmetic difference to the general kernel, at the rounding level of a term that is ~1e-3 of the field.
template <int NDEV>
__global__ void __launch_bounds__(256, 3) stg_array_step_global_kernel(const ArrArgs a) {
    extern __shared__ double lds[];           // rotated coupling [n][n]
    constexpr int n = NDEV;
    double* lc = lds;
    if (a.include_coupling) {
        for (int q = threadIdx.x; q < n * n; q += blockDim.x) {
            const int d = q / n, k = q % n;
            lc[q] = a.coupling[d * n + (d + k) % n];
        }
        __syncthreads();
    }
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N) return;
    const int64_t N = a.N;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const uint32_t off8 = (uint32_t)i * 8u, off4 = (uint32_t)i * 4u;
    const uint32_t row8 = (uint32_t)N * 8u, row4 = (uint32_t)N * 4u;
    const int obs_rows = 6 * n + (a.obs_mode == 1 ? 4 : 0);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)a.pattern, 0, (int)(3u * n * row8), 0x00020000);
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)a.target, 0, (int)(3u * n * row8), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)a.obs, 0, (int)((uint32_t)obs_rows * row4), 0x00020000);
    auto ldp = [&](int row) -> double { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rp, off8, (uint32_t)row * row8, 0)); };
    auto ldt = [&](int row) -> double { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rt, off8, (uint32_t)row * row8, 0)); };
    auto stp = [&](int row, double v) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rp, off8, (uint32_t)row * row8, 0); };
    auto sto = [&](int64_t row, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), ro, off4, (uint32_t)row * row4, 0); };
    V3 pm[NDEV];
    double sim_sum = 0.0;
#pragma unroll
    for (int d = 0; d < n; ++d) {
        double pv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            pv[k] = ldp(d * 3 + k);
            const double tv = ldt(d * 3 + k);
            sto(obs_row_target(a.obs_mode, n, d, k), (float)tv);
            sim_sum += pv[k] * tv;
        }
        pm[d] = V3{pv[0], pv[1], pv[2]};
    }
    const double prev_sim = sim_sum / n;                                               // array_env.py:372-373
    // _apply_action in 'global' mode (array_env.py:411-445): the action is [current, duration]; action[1] is what the reference reads
    // as the current density and the duration defaults to 1 ns -- kept as is
    double J = (double)a.actions[N + i];
    double T = 1e-9;
    J = isnan(J) ? J : fmin(fmax(J, -a.max_current), a.max_current);
    T = isnan(T) ? T : fmin(fmax(T, 1e-12), a.max_duration);
    const bool drive = fabs(J) > 1e-12;                                                // array_env.py:506
    double e_total = 0.0;
    if (drive) {
#pragma unroll 1
        for (int d = 0; d < n; ++d) {
            const V3 m0 = pm[0];
            V3 h = device_field(m0, a.dev);
            if (a.include_coupling) {                                                  // array_env.py:485-492
                const double* crow = lc + d * n;                                       // crow[k] = C[d][(d + k) mod n]
                V3 hc{0.0, 0.0, 0.0};
#pragma unroll
                for (int k = 1; k < n; ++k) {
                    const double c = crow[k];
                    hc = V3{hc.x + c * pm[k].x, hc.y + c * pm[k].y, hc.z + c * pm[k].z};
                }
                h = V3{h.x + hc.x, h.y + hc.y, h.z + hc.z};
            }
            // _simulate_device_dynamics (array_env.py:496-521): alpha = 0.01, gamma = 2.21e5, p_hat = z
            const V3 mxp{m0.y, -m0.x, 0.0};
            const V3 t2 = cross(m0, mxp);
            const double tj = 0.1 * J;
            const V3 mxh = cross(m0, h);
            V3 dm{-2.21e5 * mxh.x, -2.21e5 * mxh.y, -2.21e5 * mxh.z};
            const V3 mxdm = cross(m0, dm);
            dm = V3{dm.x + 0.01 * mxdm.x + tj * t2.x, dm.y + 0.01 * mxdm.y + tj * t2.y, dm.z + 0.01 * mxdm.z + tj * t2.z};
            const double dt = T / 10;
            V3 m = m0;
#pragma unroll
            for (int it = 0; it < 10; ++it) {
                m = V3{m.x + dm.x * dt, m.y + dm.y * dt, m.z + dm.z * dt};
                const double inv = rsqrt_fast(dot(m, m));       // m / |m| (array_env.py:518), <= 2 ulp per component
                m = V3{m.x * inv, m.y * inv, m.z * inv};
            }
            stp(d * 3, m.x);
            stp(d * 3 + 1, m.y);
            stp(d * 3 + 2, m.z);
            // similarity: only this cell's dot product changed
            const V3 tg{ldt(d * 3), ldt(d * 3 + 1), ldt(d * 3 + 2)};
            sim_sum += dot(m, tg) - dot(m0, tg);
            // energy with the resistance of the UPDATED state (current_m is a view of the pattern, array_env.py:455-463)
            const V3 ref{a.dev.refx, a.dev.refy, a.dev.refz};
            const double r = resistance(m, a.dev.dev_type, a.dev.r_p, a.dev.r_ap, a.dev.tmr, ref, a.dev.r_series);
            const double v = J * r * a.dev.area;
            e_total += (v * v) / r * T;
            // rotate: cell d (updated) goes to the end, cell d+1 to the front
#pragma unroll
            for (int k = 0; k + 1 < n; ++k) pm[k] = pm[k + 1];
            pm[n - 1] = m;
        }
    }
    const double etot = a.etot[i] + e_total;
    const int32_t step = a.step[i] + 1;
    a.etot[i] = etot;
    a.step[i] = step;
    const double sim = sim_sum / n;
    const bool is_success = sim >= a.thr;
    // pattern rows of the observation + uniformity (1 - population std of the cell magnitudes, array_env.py:216-224)
    double mean = 0.0;
#pragma unroll
    for (int d = 0; d < n; ++d) {
        const V3 m = pm[d];
        sto(obs_row_pattern(a.obs_mode, n, d, 0), (float)m.x);
        sto(obs_row_pattern(a.obs_mode, n, d, 1), (float)m.y);
        sto(obs_row_pattern(a.obs_mode, n, d, 2), (float)m.z);
        mean += sqrt(dot(m, m));
    }
    mean /= n;
    double var = 0.0;
#pragma unroll
    for (int d = 0; d < n; ++d) {
        const double dv = sqrt(dot(pm[d], pm[d])) - mean;
        var += dv * dv;
    }
    if (a.obs_mode == 1) {
        sto(6 * n + 0, (float)sim);
        sto(6 * n + 1, (float)((double)(a.max_steps - step) / (double)a.max_steps));
        sto(6 * n + 2, (float)(etot / 1e-12));
        sto(6 * n + 3, (float)(a.temperature / 300.0));
    }
    const double uniformity = fmax(0.0, 1.0 - sqrt(var / n));
    // default reward (array_env.py:183-224): pattern match, energy (sign as written), progress, uniformity
    double reward = 10.0 * (is_success ? 10.0 : sim * 5.0);
    reward += (-a.w_energy) * (-e_total / 1e-12);
    reward += (sim - prev_sim);
    reward += 2.0 * uniformity;
    a.reward[i] = (float)reward;
    if (a.reward64) a.reward64[i] = reward;
    if (a.energy) a.energy[i] = e_total;
    a.term[i] = is_success ? 1 : 0;
    a.trunc[i] = step >= a.max_steps ? 1 : 0;
}

// 'individual' action mode (one addressed cell per step): nothing but that one cell changes, so the step is ONE streaming
// pass over the array -- pattern and target in, both halves of the observation out, similarity, norm statistics and the
// addressed cell's coupling sum accumulated on the way -- with no per-lane copy of the pattern in LDS.  Only the coupling
// matrix sits in LDS (shared by the workgroup), so occupancy is set by registers, not by 24 KB of LDS per wavefront, and
// the kernel keeps far more loads in flight.  Same arithmetic as the general kernel except the population standard
// deviation of the cell norms (uniformity term), which is formed in one pass from e_j = |m_j| - 1 (the norms are 1 to
// rounding, so sum(e^2)/n - mean(e)^2 has no cancellation problem at the 1e-16 level the quantity lives at).
__global__ void __launch_bounds__(256) stg_array_step_individual_kernel(const ArrArgs a) {
    extern __shared__ double lds[];           // coupling [n*n]
    const int n = a.rows * a.cols;
    double* lc = lds;
    if (a.include_coupling) {
        for (int q = threadIdx.x; q < n * n; q += blockDim.x) lc[q] = a.coupling[q];
        __syncthreads();
    }
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N) return;
    const int64_t N = a.N;
    // _apply_action (array_env.py:411-445)
    double J = (double)a.actions[N + i];
    double T = (double)a.actions[2 * N + i];
    J = isnan(J) ? J : fmin(fmax(J, -a.max_current), a.max_current);
    T = isnan(T) ? T : fmin(fmax(T, 1e-12), a.max_duration);
    const double a0 = (double)a.actions[i];
    // int(np.clip(action[0], 0, n-1)); a NaN index raises in the reference -- here it addresses nothing
    const int d = isnan(a0) ? -1 : (int)fmin(fmax(a0, 0.0), (double)(n - 1));
    const bool drive = fabs(J) > 1e-12 && d >= 0;                                      // array_env.py:506
    const double* crow = lc + (d >= 0 ? d : 0) * n;
    double sim_sum = 0.0, se = 0.0, se2 = 0.0;
    V3 m0{0.0, 0.0, 1.0}, tg{0.0, 0.0, 1.0}, hc{0.0, 0.0, 0.0};
    double e_d = 0.0;
    for (int j = 0; j < n; ++j) {
        const V3 pv{a.pattern[(int64_t)(j * 3) * N + i], a.pattern[(int64_t)(j * 3 + 1) * N + i], a.pattern[(int64_t)(j * 3 + 2) * N + i]};
        const V3 tv{a.target[(int64_t)(j * 3) * N + i], a.target[(int64_t)(j * 3 + 1) * N + i], a.target[(int64_t)(j * 3 + 2) * N + i]};
        a.obs[obs_row_target(a.obs_mode, n, j, 0) * N + i] = (float)tv.x;
        a.obs[obs_row_target(a.obs_mode, n, j, 1) * N + i] = (float)tv.y;
        a.obs[obs_row_target(a.obs_mode, n, j, 2) * N + i] = (float)tv.z;
        a.obs[obs_row_pattern(a.obs_mode, n, j, 0) * N + i] = (float)pv.x;          // (the addressed cell is rewritten below)
        a.obs[obs_row_pattern(a.obs_mode, n, j, 1) * N + i] = (float)pv.y;
        a.obs[obs_row_pattern(a.obs_mode, n, j, 2) * N + i] = (float)pv.z;
        sim_sum += pv.x * tv.x; sim_sum += pv.y * tv.y; sim_sum += pv.z * tv.z;       // same order as the general kernel
        const double e = sqrt(dot(pv, pv)) - 1.0;
        se += e; se2 += e * e;
        if (j == d) { m0 = pv; tg = tv; e_d = e; }
        if (a.include_coupling && drive) {                                             // array_env.py:485-492
            const double c = (j == d) ? 0.0 : crow[j];
            hc = V3{hc.x + c * pv.x, hc.y + c * pv.y, hc.z + c * pv.z};
        }
    }
    const double prev_sim = sim_sum / n;                                               // array_env.py:372-373
    double e_total = 0.0;
    if (drive) {
        V3 h = device_field(m0, a.dev);
        if (a.include_coupling) h = V3{h.x + hc.x, h.y + hc.y, h.z + hc.z};
        // _simulate_device_dynamics (array_env.py:496-521): alpha = 0.01, gamma = 2.21e5, p_hat = z
        const V3 mxp{m0.y, -m0.x, 0.0};
        const V3 t2 = cross(m0, mxp);
        const double tj = 0.1 * J;
        const V3 mxh = cross(m0, h);
        V3 dm{-2.21e5 * mxh.x, -2.21e5 * mxh.y, -2.21e5 * mxh.z};
        const V3 mxdm = cross(m0, dm);
        dm = V3{dm.x + 0.01 * mxdm.x + tj * t2.x, dm.y + 0.01 * mxdm.y + tj * t2.y, dm.z + 0.01 * mxdm.z + tj * t2.z};
        const double dt = T / 10;
        V3 m = m0;
#pragma unroll
        for (int it = 0; it < 10; ++it) {
            m = V3{m.x + dm.x * dt, m.y + dm.y * dt, m.z + dm.z * dt};
            const double inv = rsqrt_fast(dot(m, m));           // m / |m| (array_env.py:518), <= 2 ulp per component
            m = V3{m.x * inv, m.y * inv, m.z * inv};
        }
        a.pattern[(int64_t)(d * 3) * N + i] = m.x;
        a.pattern[(int64_t)(d * 3 + 1) * N + i] = m.y;
        a.pattern[(int64_t)(d * 3 + 2) * N + i] = m.z;
        a.obs[obs_row_pattern(a.obs_mode, n, d, 0) * N + i] = (float)m.x;
        a.obs[obs_row_pattern(a.obs_mode, n, d, 1) * N + i] = (float)m.y;
        a.obs[obs_row_pattern(a.obs_mode, n, d, 2) * N + i] = (float)m.z;
        sim_sum += dot(m, tg) - dot(m0, tg);
        const double e_new = sqrt(dot(m, m)) - 1.0;
        se += e_new - e_d; se2 += e_new * e_new - e_d * e_d;
        // energy with the resistance of the UPDATED state (current_m is a view of the pattern, array_env.py:455-463)
        const V3 ref{a.dev.refx, a.dev.refy, a.dev.refz};
        const double r = resistance(m, a.dev.dev_type, a.dev.r_p, a.dev.r_ap, a.dev.tmr, ref, a.dev.r_series);
        const double v = J * r * a.dev.area;
        e_total = (v * v) / r * T;
    }
    const double etot = a.etot[i] + e_total;
    const int32_t step = a.step[i] + 1;
    a.etot[i] = etot;
    a.step[i] = step;
    const double sim = sim_sum / n;
    const bool is_success = sim >= a.thr;
    if (a.obs_mode == 1) {
        a.obs[(int64_t)(6 * n + 0) * N + i] = (float)sim;
        a.obs[(int64_t)(6 * n + 1) * N + i] = (float)((double)(a.max_steps - step) / (double)a.max_steps);
        a.obs[(int64_t)(6 * n + 2) * N + i] = (float)(etot / 1e-12);
        a.obs[(int64_t)(6 * n + 3) * N + i] = (float)(a.temperature / 300.0);
    }
    // uniformity = 1 - population std of the cell norms (array_env.py:216-224)
    const double me = se / n;
    const double var = fmax(se2 / n - me * me, 0.0);
    const double uniformity = fmax(0.0, 1.0 - sqrt(var));
    double reward = 10.0 * (is_success ? 10.0 : sim * 5.0);                            // array_env.py:183-224
    reward += (-a.w_energy) * (-e_total / 1e-12);
    reward += (sim - prev_sim);
    reward += 2.0 * uniformity;
    a.reward[i] = (float)reward;
    if (a.reward64) a.reward64[i] = reward;
    if (a.energy) a.energy[i] = e_total;
    a.term[i] = is_success ? 1 : 0;
    a.trunc[i] = step >= a.max_steps ? 1 : 0;
}

struct ArrResetArgs {
    int64_t N, env_id0;
    int32_t rows, cols, max_steps, obs_mode;
    double temperature;
    const uint8_t* mask;
    const double *init_pattern, *target_in;
    double *pattern, *target, *etot;
    int32_t* step;
    uint32_t* resets;
    uint64_t seed;
    int32_t default_target;
    float* obs;
};

__global__ void __launch_bounds__(64) stg_array_reset_kernel(const ArrResetArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= a.N) return;
    const int n = a.rows * a.cols;
    const int64_t N = a.N;
    const bool sel = !a.mask || a.mask[i];
    if (sel) {
        if (a.init_pattern) {
            for (int q = 0; q < 3 * n; ++q) a.pattern[(int64_t)q * N + i] = a.init_pattern[(int64_t)q * N + i];
        } else {   // normal(0,1,3) normalised per device (array_env.py:350-356), from the device generator
            NormalStream ns;
            ns.init(a.seed ^ 0x9E3779B97F4A7C15ull, (uint64_t)(a.env_id0 + i), a.resets[i], 0xFFFFFFFDu);
            for (int d = 0; d < n; ++d) {
                const V3 z = (d & 1) ? ns.draw3_odd() : ns.draw3_even();
                const double inv = 1.0 / sqrt(dot(z, z));
                a.pattern[(int64_t)(d * 3) * N + i] = z.x * inv;
                a.pattern[(int64_t)(d * 3 + 1) * N + i] = z.y * inv;
                a.pattern[(int64_t)(d * 3 + 2) * N + i] = z.z * inv;
            }
            a.resets[i] += 1;
        }
        if (a.target_in) {
            for (int q = 0; q < 3 * n; ++q) a.target[(int64_t)q * N + i] = a.target_in[(int64_t)q * N + i];
        } else if (a.default_target) {   // checkerboard of +-z (array_env.py:161-170)
            for (int d = 0; d < n; ++d) {
                const int r = d / a.cols, c = d % a.cols;
                a.target[(int64_t)(d * 3) * N + i] = 0.0;
                a.target[(int64_t)(d * 3 + 1) * N + i] = 0.0;
                a.target[(int64_t)(d * 3 + 2) * N + i] = ((r + c) % 2 == 0) ? 1.0 : -1.0;
            }
        }
        a.etot[i] = 0.0;
        a.step[i] = 0;
    }
    if (a.obs) {
        double s = 0.0;
        for (int d = 0; d < n; ++d)
            for (int k = 0; k < 3; ++k) s += a.pattern[(int64_t)(d * 3 + k) * N + i] * a.target[(int64_t)(d * 3 + k) * N + i];
        const double sim = s / n;
        if (a.obs_mode == 0) {
            for (int d = 0; d < n; ++d)
                for (int k = 0; k < 3; ++k) {
                    a.obs[(int64_t)(d * 6 + k) * N + i] = (float)a.pattern[(int64_t)(d * 3 + k) * N + i];
                    a.obs[(int64_t)(d * 6 + 3 + k) * N + i] = (float)a.target[(int64_t)(d * 3 + k) * N + i];
                }
        } else {
            for (int q = 0; q < 3 * n; ++q) {
                a.obs[(int64_t)q * N + i] = (float)a.pattern[(int64_t)q * N + i];
                a.obs[(int64_t)(3 * n + q) * N + i] = (float)a.target[(int64_t)q * N + i];
            }
            a.obs[(int64_t)(6 * n + 0) * N + i] = (float)sim;
            a.obs[(int64_t)(6 * n + 1) * N + i] = (float)((double)(a.max_steps - a.step[i]) / (double)a.max_steps);
            a.obs[(int64_t)(6 * n + 2) * N + i] = (float)(a.etot[i] / 1e-12);
            a.obs[(int64_t)(6 * n + 3) * N + i] = (float)(a.temperature / 300.0);
        }
    }
}

}  // namespace

extern "C" int stg_internal_fail(int code, const char* msg);     // spintorque_hip.hip: sets stg_last_error()
static int afail(int code, const std::string& msg) { return stg_internal_fail(code, msg.c_str()); }

#define AHIP_TRY(expr)                                                                                  \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return afail(STG_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct stg_array_ctx {
    int device;
    int64_t N, env_id0;
    stg_array_config cfg;
    ArrDev dev;
    void* slab = nullptr;
    double *pattern, *target, *etot, *coupling;
    int32_t* step;
    uint32_t* resets;
    bool have_state = false, have_target = false;
    int variant = 1;          // experiment knob STG_ARRAY_VARIANT: 0 generic kernel for every size, 1 (default) the 4 x 4 specialisations,
                              // 2 the 4 x 4 kernel with the pattern in LDS in 'global' mode too
};

extern "C" {

int stg_array_create(stg_array_ctx** out, int device_id, int64_t n_arrays, int64_t env_id0, const stg_array_config* cfg,
                     const stg_device_params* p, const double* coupling) {
    if (!out || !cfg || !p) return afail(STG_E_INVALID, "out/cfg/dev is NULL");
    *out = nullptr;
    if (n_arrays < 1) return afail(STG_E_INVALID, "n_arrays must be >= 1");
    const int n = cfg->rows * cfg->cols;
    if (cfg->rows < 1 || cfg->cols < 1 || n > ARR_MAX_DEV) return afail(STG_E_INVALID, "array size must be between 1 and 64 devices");
    if (cfg->action_mode < 0 || cfg->action_mode > 3) return afail(STG_E_INVALID, "action_mode must be 0..3");
    if (cfg->obs_mode < 0 || cfg->obs_mode > 1) return afail(STG_E_INVALID, "obs_mode must be 0 ('array') or 1 ('vector')");
    if (cfg->include_coupling && !coupling) return afail(STG_E_INVALID, "coupling matrix required when include_coupling is set");
    if (cfg->max_steps < 1) return afail(STG_E_INVALID, "max_steps must be >= 1");
    if (p->dev_type < 0 || p->dev_type > 2) return afail(STG_E_INVALID, "dev_type must be STG_DEV_STT/SOT/VCMA");
    int ndev = 0;
    AHIP_TRY(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return afail(STG_E_INVALID, "device_id out of range");
    AHIP_TRY(hipSetDevice(device_id));
    stg_array_ctx* c = new (std::nothrow) stg_array_ctx();
    if (!c) return afail(STG_E_NOMEM, "out of host memory");
    c->device = device_id; c->N = n_arrays; c->env_id0 = env_id0; c->cfg = *cfg;
    if (const char* e = std::getenv("STG_ARRAY_VARIANT")) c->variant = std::atoi(e);
    const double mu0 = 4 * M_PI * 1e-7;
    ArrDev& d = c->dev;
    d.hk = 2 * p->ku / (mu0 * p->ms); d.ms = p->ms;                       // stt_mram.py:71, sot_mram.py:93
    d.ex = p->easy_axis[0]; d.ey = p->easy_axis[1]; d.ez = p->easy_axis[2];
    d.nx = p->shape_demag[0]; d.ny = p->shape_demag[1]; d.nz = p->shape_demag[2];
    d.area = p->area; d.r_p = p->r_p; d.r_ap = p->r_ap; d.tmr = (p->r_ap - p->r_p) / p->r_p;
    const double rn = std::sqrt((p->ref_m[0] * p->ref_m[0] + p->ref_m[1] * p->ref_m[1]) + p->ref_m[2] * p->ref_m[2]);
    d.refx = p->ref_m[0] / rn; d.refy = p->ref_m[1] / rn; d.refz = p->ref_m[2] / rn;
    d.r_series = p->r_series; d.dev_type = p->dev_type;
    auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
    const size_t N = (size_t)n_arrays;
    const size_t rp = al(N * 8 * 3 * n), r8 = al(N * 8), r4 = al(N * 4), rc = al(sizeof(double) * n * n);
    const size_t total = 2 * rp + r8 + 2 * r4 + rc;
    hipError_t e = hipMalloc(&c->slab, total);
    if (e != hipSuccess) { delete c; return afail(STG_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    (void)hipMemset(c->slab, 0, total);
    char* q = (char*)c->slab;
    c->pattern = (double*)q; q += rp;
    c->target = (double*)q; q += rp;
    c->etot = (double*)q; q += r8;
    c->step = (int32_t*)q; q += r4;
    c->resets = (uint32_t*)q; q += r4;
    c->coupling = (double*)q;
    if (cfg->include_coupling) {
        e = hipMemcpy(c->coupling, coupling, sizeof(double) * n * n, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(c->slab); delete c; return afail(STG_E_HIP, "hipMemcpy(coupling) failed"); }
    }
    *out = c;
    return STG_OK;
}

void stg_array_destroy(stg_array_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->slab) (void)hipFree(ctx->slab);
    delete ctx;
}

int stg_array_reset(stg_array_ctx* ctx, const uint8_t* mask, const double* init_pattern, const double* target,
                    uint64_t seed, float* obs_out, void* stream) {
    if (!ctx) return afail(STG_E_INVALID, "ctx is NULL");
    AHIP_TRY(hipSetDevice(ctx->device));
    ArrResetArgs a{};
    a.N = ctx->N; a.env_id0 = ctx->env_id0; a.rows = ctx->cfg.rows; a.cols = ctx->cfg.cols; a.max_steps = ctx->cfg.max_steps;
    a.obs_mode = ctx->cfg.obs_mode; a.temperature = ctx->cfg.temperature; a.mask = mask; a.init_pattern = init_pattern;
    a.target_in = target; a.pattern = ctx->pattern; a.target = ctx->target; a.etot = ctx->etot; a.step = ctx->step;
    a.resets = ctx->resets; a.seed = seed; a.default_target = ctx->have_target ? 0 : 1; a.obs = obs_out;
    hipLaunchKernelGGL(stg_array_reset_kernel, dim3((unsigned)((ctx->N + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a);
    AHIP_TRY(hipGetLastError());
    ctx->have_state = true;
    ctx->have_target = true;
    return STG_OK;
}

int stg_array_step(stg_array_ctx* ctx, const float* actions, float* obs, float* reward, double* reward_f64, double* energy,
                   uint8_t* terminated, uint8_t* truncated, void* stream) {
    if (!ctx) return afail(STG_E_INVALID, "ctx is NULL");
    if (!ctx->have_state) return afail(STG_E_STATE, "stg_array_reset must precede stg_array_step");
    if (!actions || !obs || !reward || !terminated || !truncated) return afail(STG_E_INVALID, "actions/obs/reward/terminated/truncated must not be NULL");
    AHIP_TRY(hipSetDevice(ctx->device));
    const stg_array_config& c = ctx->cfg;
    ArrArgs a{};
    a.N = ctx->N; a.rows = c.rows; a.cols = c.cols; a.mode = c.action_mode; a.include_coupling = c.include_coupling;
    a.max_steps = c.max_steps; a.obs_mode = c.obs_mode; a.max_current = c.max_current; a.max_duration = c.max_duration;
    a.thr = c.success_threshold; a.w_energy = c.energy_penalty_weight; a.temperature = c.temperature; a.dev = ctx->dev;
    a.coupling = ctx->coupling; a.pattern = ctx->pattern; a.target = ctx->target; a.etot = ctx->etot; a.step = ctx->step;
    a.actions = actions; a.obs = obs; a.reward = reward; a.reward64 = reward_f64; a.energy = energy; a.term = terminated;
    a.trunc = truncated;
    const int n = c.rows * c.cols;
    if (c.action_mode == 0) {
        // one addressed cell per step: streaming kernel, only the coupling matrix in LDS (<= 32 KB at 8 x 8)
        const size_t lds_c = c.include_coupling ? sizeof(double) * (size_t)n * n : 0;
        hipLaunchKernelGGL(stg_array_step_individual_kernel, dim3((unsigned)((ctx->N + 255) / 256)), dim3(256), lds_c,
                           (hipStream_t)stream, a);
        AHIP_TRY(hipGetLastError());
        return STG_OK;
    }
    if (c.action_mode == 3 && n == 16 && ctx->variant >= 1 && ctx->variant != 2 && ctx->N <= (4ll << 20)) {
        // 'global' mode on 4 x 4 arrays: fully unrolled sweep, pattern in registers (STG_ARRAY_VARIANT=2: the LDS form of this size)
        const size_t lds_c = c.include_coupling ? sizeof(double) * (size_t)n * n : 0;
        hipLaunchKernelGGL((stg_array_step_global_kernel<16>), dim3((unsigned)((ctx->N + 255) / 256)), dim3(256), lds_c, (hipStream_t)stream, a);
        AHIP_TRY(hipGetLastError());
        return STG_OK;
    }
    const size_t lds = sizeof(double) * ((size_t)n * 3 * 64 + (c.include_coupling ? (size_t)n * n : 0));
    const dim3 grid((unsigned)((ctx->N + 63) / 64));
    if (n == 16 && ctx->variant != 0) {
        // 4 x 4 arrays (the registered SpinTorqueArray-v0): the coupling sum unrolled
        hipLaunchKernelGGL((stg_array_step_kernel<16>), grid, dim3(64), lds, (hipStream_t)stream, a);
    } else {
        if (lds > 48 * 1024)
            AHIP_TRY(hipFuncSetAttribute((const void*)stg_array_step_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((stg_array_step_kernel<0>), grid, dim3(64), lds, (hipStream_t)stream, a);
    }
    AHIP_TRY(hipGetLastError());
    return STG_OK;
}

int stg_array_get_state(stg_array_ctx* ctx, double* pattern, double* target, double* total_energy, int32_t* step_count,
                        void* stream) {
    if (!ctx) return afail(STG_E_INVALID, "ctx is NULL");
    AHIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t N = (size_t)ctx->N, n = (size_t)ctx->cfg.rows * ctx->cfg.cols;
    if (pattern) AHIP_TRY(hipMemcpyAsync(pattern, ctx->pattern, N * 8 * 3 * n, hipMemcpyDeviceToDevice, st));
    if (target) AHIP_TRY(hipMemcpyAsync(target, ctx->target, N * 8 * 3 * n, hipMemcpyDeviceToDevice, st));
    if (total_energy) AHIP_TRY(hipMemcpyAsync(total_energy, ctx->etot, N * 8, hipMemcpyDeviceToDevice, st));
    if (step_count) AHIP_TRY(hipMemcpyAsync(step_count, ctx->step, N * 4, hipMemcpyDeviceToDevice, st));
    return STG_OK;
}

}  // extern "C"
