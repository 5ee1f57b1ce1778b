// stg_physics.hpp -- device-side physics of the SpinTorque-v0 step path for gfx950 (CDNA4).
//
// One macrospin per lane: every function here is straight-line fp64 VALU work on registers, so a
// 64-wide wavefront integrates 64 independent environments in lock step.  Divergence (per-lane trip
// counts, accept/reject) is handled by EXEC masking of the enclosing loops; there is no cross-lane
// traffic and no MFMA (the path is an elementwise 3-vector ODE, not a contraction).
//
// Reference semantics followed (paths relative to /root/reference/spin_torque_gym/):
//   SimpleLLGSSolver RHS / RK4 / Euler / normalisation ... physics/simple_solver.py:137-168,208-229,263-388
//   RobustLLGSSolver input/output gates ................. utils/robust_solver.py:152-205
//   LLGSSolver RHS / energy ............................. physics/llgs_solver.py:92-126,182-262
//   SciPy RK45 (Dormand-Prince 5(4)) controller ......... scipy/integrate/_ivp/rk.py, common.py (scipy 1.15.3)
//   compute_resistance .................................. devices/stt_mram.py:78-94, sot_mram.py:196-228, vcma_mram.py:236-257
//   action clamp, observation, reward, termination ...... utils/monitoring.py:288-348, envs/spin_torque_env.py:310-524
//
// Floating point: IEEE fp64, denormals on, no fast-math.  FMA contraction is allowed in the RHS
// algebra (it only changes roundings at the 1e-16 level; parity bound is 1e-5, measured ~1e-12) but
// NOT where the reference's branch decisions hang on an exact rounding: sub-step and stage times (SURVEY
// H4/H5) go through mul_x/add_x/sub_x below, which are compiled with contraction off (ROCm's __dadd_rn/__dmul_rn are
// plain operators and DO get fused; the build uses -ffp-contract=fast-honor-pragmas so the pragma holds).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace stg {

// ---- per-class derived constants (host-computed in fp64, staged in LDS by the kernels) ----------
enum ClassConst : int {
    // SimpleLLGSSolver
    C_EX = 0, C_EY, C_EZ,      // easy_axis / |easy_axis|                      simple_solver.py:318
    C_HK,                      // 2*k_u/(mu_0*ms)                              simple_solver.py:370
    C_MS,                      // saturation magnetisation
    C_ALPHA,                   // damping
    C_GEFF,                    // gamma/(1+alpha^2)                            simple_solver.py:337
    C_POL,                     // polarization
    C_MSV,                     // ms*volume                                    simple_solver.py:330
    C_HS_SIMPLE,               // Brown strength, kb = 1.38e-23                simple_solver.py:380-383
    // LLGSSolver
    C_RX, C_RY, C_RZ,          // raw easy axis (not normalised)               llgs_solver.py:194-196
    C_DX, C_DY, C_DZ,          // -ms*demag_factors                            llgs_solver.py:200-201
    C_HEX,                     // (2*a_ex/(mu_0*ms))*0.1 or 0                  llgs_solver.py:205-209
    C_BETA, C_BETAP,           // P*gamma/(2*ms*V), 0.1*beta                   llgs_solver.py:229-230
    C_GAMMA,
    C_HS_LLGS,                 // Brown strength, k_b = 1.380649e-23           llgs_solver.py:87-90
    C_KUV,                     // k_u*volume          (energy, llgs_solver.py:256)
    C_EDEMAG,                  // 0.5*mu_0*ms^2*volume (energy, llgs_solver.py:260)
    C_NX, C_NY, C_NZ,          // demag_factors
    // env
    C_AREA, C_RP, C_RAP, C_TMR,
    C_REFX, C_REFY, C_REFZ,    // reference_magnetization / |.|
    C_RSERIES,
    C_DEVTYPE,                 // STG_DEV_* as double
    C_VALID,                   // validate_parameters outcome as double (0/1)
    // opt-in device-physics torque model
    C_SOT_DL, C_SOT_FL,        // tau_dl_factor, tau_fl_factor                 sot_mram.py:61-72
    C_SIGX, C_SIGY, C_SIGZ,    // sigma = z x j_hat                            sot_mram.py:183-186
    C_KU,                      // uniaxial anisotropy K
    C_VCMA_XI, C_VCMA_TD2,     // vcma_coefficient, dielectric_thickness**2    vcma_mram.py:136-139
    C_VCMA_VBD,                // breakdown_voltage                            vcma_mram.py:132
    C_MU0MS,                   // mu_0 * ms (denominator of h_k)
    C_COUNT
};

// Derived constants of one device-parameter record, in the reference's own operation order (each line cites the
// expression it evaluates).  Host: stg_set_params builds the class table with it; device: per-env parameters
// (stg_set_params_per_env) are derived by each lane in the kernel prologue -- same arithmetic, no contraction.
template <class P>
__host__ __device__ inline void derive_row(const P& p, double gamma, double temperature, double* r) {
#pragma clang fp contract(off)
    const double mu0 = 4 * 3.14159265358979323846 * 1e-7;                        // simple_solver.py:60
    const double en = sqrt((p.easy_axis[0] * p.easy_axis[0] + p.easy_axis[1] * p.easy_axis[1]) + p.easy_axis[2] * p.easy_axis[2]);
    r[C_EX] = p.easy_axis[0] / en; r[C_EY] = p.easy_axis[1] / en; r[C_EZ] = p.easy_axis[2] / en;   // simple_solver.py:318
    r[C_HK] = (2 * p.ku) / (mu0 * p.ms);                                         // simple_solver.py:370 == llgs_solver.py:196
    r[C_MS] = p.ms;
    r[C_ALPHA] = p.damping;
    r[C_GEFF] = gamma / (1 + p.damping * p.damping);                             // simple_solver.py:337
    r[C_POL] = p.polarization;
    r[C_MSV] = p.ms * p.volume;                                                  // simple_solver.py:330
    r[C_HS_SIMPLE] = sqrt(2 * p.damping * 1.38e-23 * temperature / (mu0 * p.ms * p.volume * gamma));   // :380-383
    r[C_RX] = p.easy_axis[0]; r[C_RY] = p.easy_axis[1]; r[C_RZ] = p.easy_axis[2];
    r[C_DX] = -p.ms * p.demag[0]; r[C_DY] = -p.ms * p.demag[1]; r[C_DZ] = -p.ms * p.demag[2];   // llgs_solver.py:201
    r[C_HEX] = p.a_ex > 0 ? (2 * p.a_ex / (mu0 * p.ms)) * 0.1 : 0.0;             // llgs_solver.py:205-209
    r[C_BETA] = p.polarization * gamma / (2 * p.ms * p.volume);                  // llgs_solver.py:229
    r[C_BETAP] = 0.1 * r[C_BETA];                                                // llgs_solver.py:230
    r[C_GAMMA] = gamma;
    r[C_HS_LLGS] = sqrt(2 * p.damping * 1.380649e-23 * temperature / (gamma * mu0 * p.ms * p.volume));  // llgs_solver.py:87-90
    r[C_KUV] = p.ku * p.volume;                                                  // llgs_solver.py:256
    r[C_EDEMAG] = 0.5 * mu0 * (p.ms * p.ms) * p.volume;                          // llgs_solver.py:260
    r[C_NX] = p.demag[0]; r[C_NY] = p.demag[1]; r[C_NZ] = p.demag[2];
    r[C_AREA] = p.area; r[C_RP] = p.r_p; r[C_RAP] = p.r_ap;
    r[C_TMR] = (p.r_ap - p.r_p) / p.r_p;                                         // stt_mram.py:88
    const double rn = sqrt((p.ref_m[0] * p.ref_m[0] + p.ref_m[1] * p.ref_m[1]) + p.ref_m[2] * p.ref_m[2]);
    r[C_REFX] = p.ref_m[0] / rn; r[C_REFY] = p.ref_m[1] / rn; r[C_REFZ] = p.ref_m[2] / rn;
    r[C_RSERIES] = p.r_series;
    r[C_DEVTYPE] = (double)p.dev_type;
    r[C_VALID] = p.params_valid ? 1.0 : 0.0;
    r[C_SOT_DL] = p.sot_tau_dl; r[C_SOT_FL] = p.sot_tau_fl;
    r[C_SIGX] = p.sot_sigma[0]; r[C_SIGY] = p.sot_sigma[1]; r[C_SIGZ] = p.sot_sigma[2];
    r[C_KU] = p.ku;
    r[C_VCMA_XI] = p.vcma_xi;
    r[C_VCMA_TD2] = p.vcma_td * p.vcma_td;                                       // dielectric_thickness**2, vcma_mram.py:139
    r[C_VCMA_VBD] = p.vcma_vbd;
    r[C_MU0MS] = mu0 * p.ms;
}

struct V3 {
    double x, y, z;
};

// single IEEE operations that must never be contracted into an FMA with a neighbour
__device__ __forceinline__ double mul_x(double a, double b) {
#pragma clang fp contract(off)
    const double r = a * b;
    return r;
}
__device__ __forceinline__ double add_x(double a, double b) {
#pragma clang fp contract(off)
    const double r = a + b;
    return r;
}
__device__ __forceinline__ double sub_x(double a, double b) {
#pragma clang fp contract(off)
    const double r = a - b;
    return r;
}

// 1/sqrt(s) for s in [1e-24, DBL_MAX]: hardware seed (v_rsq_f64) + one third-order correction, ~1 ulp.
// Replaces sqrt + three IEEE divisions per normalisation (the reference's m / |m|); the quotient differs from the
// correctly rounded one by <= 2 ulp per component.  s = 0, +inf or NaN give NaN (0 * inf in the correction): a row whose squared
// norm over- or underflowed poisons its solve, which then fails and returns the row it was given
// (tests/test_gpu_parity.py: test_rk45_pathological_start_rows_vs_oracle; the fixed-step path has its own tiers, simple_validate).
__device__ __forceinline__ double rsqrt_fast(double s) {
    const double y = __builtin_amdgcn_rsq(s);
    const double e = __builtin_fma(-(s * y), y, 1.0);
    return __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);
}

// a wave-uniform constant held in a VGPR pair (opaque to the optimiser: it is not moved back into SGPRs)
__device__ __forceinline__ double vgpr_const(double c) {
    asm volatile("" : "+v"(c));
    return c;
}

// max(|a|, |b|) in ONE v_max_f64 with source modifiers.  (fmax(fabs(a), fabs(b)) compiles to three: the IEEE-mode lowering of fmax
// first quiets each operand with v_max x, x, and fabs of a possibly-signalling NaN does not count as quiet.  No signalling NaN
// exists on this path -- every value is the result of an arithmetic instruction -- and for quiet NaNs the instruction returns the
// other operand, as fmax does.)
__device__ __forceinline__ double fmax_abs(double a, double b) {
    double r;
    asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// dst_i = src_i in the lanes of `mask` (a ballot), seven doubles at once: seven v_mov_b64 under EXEC = mask instead of fourteen
// v_cndmask_b32 -- the commit of an accepted attempt (state, f(state), time).
__device__ __forceinline__ void commit7(unsigned long long mask, double& d0, double s0, double& d1, double s1, double& d2, double s2,
                                        double& d3, double s3, double& d4, double s4, double& d5, double s5, double& d6, double s6) {
    // (EXEC and the mask change places by three XORs, so that no further SGPR pair is needed: the kernels sit at the SGPR limit)
    asm("s_xor_b64 exec, exec, %[m]\n\t"
        "s_xor_b64 %[m], exec, %[m]\n\t"
        "s_xor_b64 exec, exec, %[m]\n\t"
        "v_mov_b64 %[d0], %[s0]\n\t"
        "v_mov_b64 %[d1], %[s1]\n\t"
        "v_mov_b64 %[d2], %[s2]\n\t"
        "v_mov_b64 %[d3], %[s3]\n\t"
        "v_mov_b64 %[d4], %[s4]\n\t"
        "v_mov_b64 %[d5], %[s5]\n\t"
        "v_mov_b64 %[d6], %[s6]\n\t"
        "s_mov_b64 exec, %[m]"
        : [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2), [d3] "+v"(d3), [d4] "+v"(d4), [d5] "+v"(d5), [d6] "+v"(d6), [m] "+s"(mask)
        : [s0] "v"(s0), [s1] "v"(s1), [s2] "v"(s2), [s3] "v"(s3), [s4] "v"(s4), [s5] "v"(s5), [s6] "v"(s6)
        : "scc");
}

// 1/x: hardware seed (v_rcp_f64) + one Newton step, ~1 ulp.
__device__ __forceinline__ double rcp_fast(double x) {
    const double y = __builtin_amdgcn_rcp(x);
    return __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
}

// e2^(-1/10) for e2 = err^2 in (3.5e-11, 3.4e6), i.e. err^(-1/5): fp32 transcendental seed + one third-order Newton step
// on y^-10 = e2 (~1 ulp).  Stands in for the libm pow of SciPy's step-size controller (rk.py:158-168).
__device__ __forceinline__ double inv_tenth_root(double e2) {
    const double y = (double)__builtin_amdgcn_exp2f(-0.1f * __builtin_amdgcn_logf((float)e2));
    const double y2 = y * y, y4 = y2 * y2, y8 = y4 * y4;
    const double e = __builtin_fma(-e2, y8 * y2, 1.0);                // e = 1 - e2 y^10 = -(10 d + 45 d^2), y = y*(1+d)
    return __builtin_fma(y * e, __builtin_fma(0.055, e, 0.1), y);     // y (1 + e/10 + 11 e^2/200)
}

// x^(1/5) for finite x > 0 (any exponent): r = x^(-1/5) from an fp32 transcendental seed on the mantissa plus one
// third-order Newton step on r^-5 = x (~1 ulp), then x * r^4.  Stands in for the libm pow of SciPy's
// select_initial_step (common.py:128); the result only seeds the first step size.
__device__ __forceinline__ double fifth_root(double x) {
    int e;
    const double mant = frexp(x, &e);                                     // x = mant 2^e, mant in [0.5, 1)
    const double t = -0.2 * ((double)e + (double)__builtin_amdgcn_logf((float)mant));   // log2(x^(-1/5))
    const double ti = rint(t);
    double r = ldexp((double)__builtin_amdgcn_exp2f((float)(t - ti)), (int)ti);
    const double r2 = r * r, r4 = r2 * r2;
    const double err = __builtin_fma(-x, r4 * r, 1.0);                    // 1 - x r^5
    r = __builtin_fma(r * err, __builtin_fma(0.12, err, 0.2), r);         // r (1 + e/5 + 3 e^2/25)
    const double q2 = r * r;
    return x * (q2 * q2);
}

// cross/dot spell their FMAs out (contraction of a*b - c*d is otherwise the backend's choice and can change with the
// surrounding code, i.e. between two instantiations of the same source)
__device__ __forceinline__ V3 cross(const V3& a, const V3& b) {
#pragma clang fp contract(off)
    return V3{__builtin_fma(a.y, b.z, -(a.z * b.y)), __builtin_fma(a.z, b.x, -(a.x * b.z)), __builtin_fma(a.x, b.y, -(a.y * b.x))};
}
__device__ __forceinline__ double dot(const V3& a, const V3& b) {
#pragma clang fp contract(off)
    return __builtin_fma(a.z, b.z, __builtin_fma(a.y, b.y, a.x * b.x));
}
// thermal field of one RHS call from its three normals (separate roundings: the producer wavefront of the
// wave-specialised kernels forms exactly this product before handing it over)
__device__ __forceinline__ V3 scale3(double c, const V3& z) {
#pragma clang fp contract(off)
    return V3{c * z.x, c * z.y, c * z.z};
}
__device__ __forceinline__ bool finite3(const V3& a) { return isfinite(a.x) && isfinite(a.y) && isfinite(a.z); }

// ---- counter-based RNG: Philox4x32-10 (Salmon et al., SC'11) ------------------------------------
// key = seed, counter = (env_id_lo, env_id_hi, env_step, call_idx): any lane can produce its draw for
// any RHS call without state, so results are independent of the partition of envs over GPUs.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// The thermal field needs three standard normals per RHS call.  Per (env, env step) the kernels run ONE stream:
//   state   : xoshiro128+ (Blackman & Vigna), seeded by Philox4x32-10(key = seed, counter = (env_id, env_step, tag))
//             -- counter-based seeding keeps every env's stream independent of the batch partition;
//   normals : consecutive outputs (u_2p, u_2p+1) -> 23-bit uniforms in (0,1) -> one fp32 Box-Muller pair on the
//             transcendental unit (v_log_f32 / v_sin_f32 / v_cos_f32 take log2 and revolutions natively);
//   calls   : RHS call j takes normals 3j..3j+2, so even calls draw two pairs and keep the 4th normal, odd calls draw
//             one pair and use the kept one first: 1.5 pairs per call, no waste.
// (The reference draws from NumPy's global MT19937 per RHS call -- SURVEY H6 -- only the distribution can match.)
struct NormalStream {
    uint32_t s0, s1, s2, s3;
    float carry;
    __device__ __forceinline__ void init(uint64_t seed, uint64_t env_id, uint32_t env_step, uint32_t tag) {
        uint32_t r[4];
        philox4x32_10((uint32_t)env_id, (uint32_t)(env_id >> 32), env_step, tag, (uint32_t)seed, (uint32_t)(seed >> 32), r);
        s0 = r[0]; s1 = r[1]; s2 = r[2]; s3 = r[3] | 1u;     // never the all-zero state
        carry = 0.0f;
    }
    // xoshiro128+ (Blackman & Vigna; the variant its authors recommend for floating-point generation: the weak low
    // bits are discarded, only the top 23 bits of each output are used)
    __device__ __forceinline__ uint32_t next() {
        const uint32_t result = s0 + s3;
        const uint32_t t = s1 << 9;
        s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t;
        s3 = (s3 << 11) | (s3 >> 21);
        return result;
    }
    // uniform in (0,1) with 23-bit resolution: top 23 bits become the mantissa of a float in [1,2), minus (1 - 2^-24)
    __device__ __forceinline__ float uniform() {
        // (v_alignbit_b32 forms {0x7f, x} >> 9 = 0x3F800000 | (x >> 9) in one instruction)
        return __uint_as_float(__builtin_amdgcn_alignbit(0x7fu, next(), 9u)) - 0.99999994f;
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    __device__ __forceinline__ void pair(float& a, float& b) {
        // the two mantissa-trick subtractions and the two final products go through packed fp32 (v_pk_add_f32 /
        // v_pk_mul_f32: one instruction for both lanes of the pair, same IEEE results)
#ifdef STG_EXP_CHEAP_NORMALS
        // EXPERIMENT BUILD ONLY (tools/build_variant.sh cheapn "-DSTG_EXP_CHEAP_NORMALS"; never shipped): a stand-in with the COST of a
        // hypothetical cheaper generator -- the two xoshiro words converted and scaled, no transcendental at all, ~9.5 issue slots per
        // value against ~18.5 -- and the WRONG distribution (uniform, unit variance).  It measures the ceiling a cheaper normal could
        // buy before anyone designs one (profiles/EXPERIMENTS.md).
        const f32x2 u{(float)(int32_t)next(), (float)(int32_t)next()};
        const f32x2 v = u * f32x2{8.0654e-10f, 8.0654e-10f};          // sqrt(3) / 2^31
        a = v.x;
        b = v.y;
#else
        float t, c, s_;
        pair_head(t, c, s_);
        finish_pair(t, c, s_, a, b);
#endif
    }
    // the same pair in two halves: pair() == finish_pair(pair_head(...)); a producer wavefront can hand over the head
    // (t = -2 ln u0, cos, sin) and leave the square root and the products to the integrating wavefront
    __device__ __forceinline__ void pair_head(float& t, float& c, float& s_) {
        f32x2 u{__uint_as_float(__builtin_amdgcn_alignbit(0x7fu, next(), 9u)), __uint_as_float(__builtin_amdgcn_alignbit(0x7fu, next(), 9u))};
        u = u - f32x2{0.99999994f, 0.99999994f};
        t = -1.3862943611198906f * __builtin_amdgcn_logf(u.x);
        c = __builtin_amdgcn_cosf(u.y);
        s_ = __builtin_amdgcn_sinf(u.y);
    }
    static __device__ __forceinline__ void finish_pair(float t, float c, float s_, float& a, float& b) {
        const float r = __builtin_amdgcn_sqrtf(t);
        const f32x2 cs = f32x2{c, s_} * f32x2{r, r};
        a = cs.x;
        b = cs.y;
    }
    __device__ __forceinline__ V3 draw3_even() {
        float a, b, c, d;
        pair(a, b);
        pair(c, d);
        carry = d;
        return V3{(double)a, (double)b, (double)c};
    }
    __device__ __forceinline__ V3 draw3_odd() {
        float a, b;
        pair(a, b);
        return V3{(double)carry, (double)a, (double)b};
    }
};

// identifies one env-step's thermal stream
struct RngKey {
    uint64_t seed, env_id;
    uint32_t env_step;
};

// ---- where a solve gets its normals from ---------------------------------------------------------------------
// InlineNormals: the integrating lane runs the stream itself (default).
// SharedNormals: wave specialisation.  The workgroup has a second wavefront (the producer) that runs the SAME per-env
// streams ahead of the integrating wavefront, one chunk (= one RK4 sub-step / one RK45 attempt) at a time, and leaves
// the normals in an LDS ring; the integrating wavefront only reads them.  The RNG + Box-Muller work (40 % of a thermal
// RK45 attempt, 65 % of a thermal RK4 sub-step) leaves the critical path of the longest lane; the values, their order
// and therefore the results are identical.  Two ways of keeping the pair in step, chosen per solver by measurement:
//  * BARRIER (RK45, Euler): one s_barrier per chunk, ring of 2; the producer fills chunk k+1 while the integrator works
//    on chunk k and then parks at the barrier, where it costs the SIMD it shares with another workgroup's integrating
//    wavefront nothing.
//  * handshake words (RK4, where the producer is as busy as the integrator and a barrier per sub-step makes each
//    wait for the other's jitter: 0.61 -> 0.56 ms per step at 65 536 envs): hs[0] = chunks the producer has
//    published, hs[1] = chunks the integrator is through with (| PC_STOP: it stops here); ring of DEPTH = 4.  The
//    producer fills chunk k (slot k % DEPTH) once hs[1] >= k - DEPTH + 1; the integrator reads chunk k once
//    hs[0] > k.  Release stores / acquire loads at workgroup scope order the ring accesses around the words.  Both
//    wavefronts belong to one workgroup, so both are resident and each one's wait ends by the other's progress;
//    PC_STOP ends the producer's loop, and a poll budget (PC_SPIN_CAP, ~1 s) turns a protocol error into a failed
//    solve instead of a hung GPU.  (For RK45 and Euler this form measured 5-10 % slower than the barrier.)
struct InlineNormals {
    static constexpr bool kShared = false, kScaled = false;
    NormalStream ns;
    __device__ __forceinline__ void begin(const RngKey& rk) { ns.init(rk.seed, rk.env_id, rk.env_step, 0u); }
    __device__ __forceinline__ V3 draw(bool even) { return even ? ns.draw3_even() : ns.draw3_odd(); }
    static constexpr bool broken = false;
    __device__ __forceinline__ void peek() {}
    __device__ __forceinline__ bool chunk_end(bool lane_continues) { return lane_continues; }
};

// A chunk of the shared stream (one rendezvous of an integrating wavefront with its producer): an RK45 attempt (18 normals; the
// prologue's chunk 0 holds 6), or -- fixed-step solvers -- SHARED_SUBS_RK4 RK4 sub-steps with the white field (12 normals each) =
// SHARED_SUBS3 sub-steps of the forms that draw 3 per sub-step (Euler, the Ornstein-Uhlenbeck field).  Two RK4 sub-steps per chunk
// instead of one (round 4): RK4 + thermal 32 768 envs 0.518 -> 0.489 ms, 65 536 envs 0.544 -> 0.534, Euler 0.339 -> 0.330; four per
// chunk with a ring of two chunks: worse (0.546 / 0.570); a ring of 8 two-sub-step chunks no longer fits four workgroups per CU.
constexpr int SHARED_SUBS_RK4 = 2;
constexpr int SHARED_SUBS3 = 4 * SHARED_SUBS_RK4;
constexpr int SHARED_CHUNK_FIXED = 12 * SHARED_SUBS_RK4;     // normals per chunk of the fixed-step solvers
constexpr int SHARED_CHUNK_RK45 = 18;
constexpr int PC_STOP = 1 << 30;
constexpr int PC_SPIN_CAP = 1 << 22;

// LDS words that both wavefronts of a pair touch are accessed through an LDS-typed volatile pointer: through a generic one the
// compiler must keep a volatile access as a FLAT instruction (address-space inference does not rewrite volatile accesses), and a
// flat store to LDS in front of the per-attempt barrier -- `flat_store_dword ... sc0 sc1; s_waitcnt vmcnt(0) lgkmcnt(0)` -- sits on
// the integrating wavefront's critical path; typed, it is a `ds_write_b32`.
typedef __attribute__((address_space(3))) int lds_int;
__device__ __forceinline__ volatile lds_int* lds_flag(const int* p) { return (volatile lds_int*)p; }

template <int WGW>
__device__ __forceinline__ bool any_flag(volatile lds_int* f) {
    int v = f[0];
#pragma unroll
    for (int j = 1; j < WGW; ++j) v |= f[j];
    return v != 0;
}
__device__ __forceinline__ int pc_load(const int* p) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void pc_store(int* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// T = float: raw normals (fixed-step solvers, whose producer is the critical wavefront and should do no extra work);
// T = double, SCALED: the finished thermal field c * z (RK45: the integrating wavefront saves the conversions and the
// scaling of every RHS call).  DEPTH = chunks in the ring (a power of two; 2 with BARRIER).
template <typename T, bool SCALED, int DEPTH, bool BARRIER>
struct SharedNormalsT {
    static constexpr int CHUNK = SCALED ? SHARED_CHUNK_RK45 : SHARED_CHUNK_FIXED;    // (the RK45 kernels hand over finished fields)
    static_assert(!BARRIER || DEPTH == 2, "the barrier form runs the producer exactly one chunk ahead");
    static constexpr bool kShared = true, kScaled = SCALED;
    const T* buf;           // this workgroup's LDS ring [DEPTH][CHUNK][64]
    int* hs;                // LDS: the handshake words / BARRIER: the integrator's "continues" flag, two alternating copies
    int lane, it, idx, seen;   // seen: the producer's count as last read (a lane-uniform value in a VGPR)
    bool broken;               // the poll budget ran out (never, unless the protocol is broken): the solve reports failure
    __device__ __forceinline__ void begin(const RngKey&) { it = 0; idx = 0; seen = 1; broken = false; }    // chunk 0 is there (H2)
    __device__ __forceinline__ V3 draw(bool) {
        const T* b = buf + ((it & (DEPTH - 1)) * CHUNK + idx) * 64 + lane;
        idx += 3;
        return V3{(double)b[0], (double)b[64], (double)b[128]};
    }
    // an early look at the producer's count (relaxed: nothing waits for it here), so that chunk_end usually finds the
    // next chunk published without an LDS round trip of its own
    __device__ __forceinline__ void peek() {
        if (!BARRIER) seen = __hip_atomic_load(hs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // end of a chunk: publishes that the slot is free and whether this (wave-uniform) wavefront continues; returns
    // that, once the next chunk is in LDS.  Straight-line when the producer is ahead (the normal case).
    __device__ __forceinline__ bool chunk_end(bool lane_continues) {
        const bool mine = __ballot(lane_continues) != 0ull;
        if (BARRIER) {
            lds_flag(hs)[it & 1] = mine ? 1 : 0;
            __syncthreads();
            ++it;
            idx = 0;
            return mine;
        }
        ++it;
        idx = 0;
        pc_store(hs + 1, it | (mine ? 0 : PC_STOP));
        int have = __builtin_amdgcn_readfirstlane(seen);
        if (__builtin_expect(mine && have <= it, 0)) {
            int spins = 0;
#pragma unroll 1
            do {
                if (spins) __builtin_amdgcn_s_sleep(1);
                have = pc_load(hs);
            } while (have <= it && ++spins <= PC_SPIN_CAP);
            broken = broken || have <= it;
        }
        seen = have;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");      // the ring reads below stay below
        return mine && !broken;
    }
};

// The producer wavefront's side for one solve: chunk 0 has n_first normals, later chunks n_chunk; calls alternate
// between the even and odd phase of the stream across chunk boundaries, exactly as the integrator's calls do.
template <typename T, bool SCALED, int DEPTH, bool BARRIER>
__device__ __forceinline__ void produce_normals(T* buf, int* hs, int lane, const RngKey& rk, int n_first, int n_chunk, double c) {
    NormalStream ns;
    ns.init(rk.seed, rk.env_id, rk.env_step, 0u);
    bool even = true;
    auto fill = [&](int slot, int count) {
        T* b = buf + (slot * (SCALED ? SHARED_CHUNK_RK45 : SHARED_CHUNK_FIXED)) * 64 + lane;
        for (int j = 0; j < count; j += 3) {
            V3 z = even ? ns.draw3_even() : ns.draw3_odd();
            if (SCALED) z = scale3(c, z);
            even = !even;
            b[(j + 0) * 64] = (T)z.x; b[(j + 1) * 64] = (T)z.y; b[(j + 2) * 64] = (T)z.z;
        }
    };
    fill(0, n_first);
    if (!BARRIER) pc_store(hs, 1);
    __syncthreads();                                   // H2: chunk 0 ready (the integrating wavefront waits here too)
    if (BARRIER) {
        for (int k = 1;; ++k) {
            fill(k & 1, n_chunk);                      // chunk k, while the integrating wavefront works on chunk k - 1
            __syncthreads();                           // = the integrating wavefront's chunk_end rendezvous
            if (lds_flag(hs)[(k - 1) & 1] == 0) return;
        }
    }
    for (int k = 1;; ++k) {
        // slot k % DEPTH held chunk k - DEPTH: wait until the integrator is through with it
        int v = pc_load(hs + 1);
#pragma unroll 1
        for (int spins = 0; (v & (PC_STOP - 1)) < k - DEPTH + 1 && !(v & PC_STOP); ++spins) {
            if (spins > PC_SPIN_CAP) return;
            __builtin_amdgcn_s_sleep(1);
            v = pc_load(hs + 1);
        }
        if (v & PC_STOP) return;
        fill(k & (DEPTH - 1), n_chunk);
        pc_store(hs, k + 1);
    }
}

// ---- per-lane constant sets ----------------------------------------------------------------------
struct SimpleK {            // A1/A2 constants of this lane's device class, with -gamma' = -gamma/(1+alpha^2) folded in
    V3 e;                   // easy axis, normalised
    double ghk;             // -gamma' * H_k
    double gdm;             // +gamma' * Ms          (-gamma' times the demagnetising field -Ms m_z z^)
    double alpha;
    double ghs;             // -gamma' * Brown strength
};
__device__ __forceinline__ SimpleK make_simple(const V3& e, double hk, double ms, double alpha, double geff, double hs) {
#pragma clang fp contract(off)
    return SimpleK{e, -geff * hk, geff * ms, alpha, -geff * hs};
}
struct LlgsK {              // A6 constants, with -gamma folded in (dm0 = -gamma m x H = m x (-gamma H))
    V3 r;                   // raw easy axis
    V3 gd;                  // -gamma * (-ms * demag_factors)
    double ghk, ghex;       // -gamma * hk, -gamma * hex
    double alpha;
    double ghs;             // -gamma * Brown strength: the thermal field enters as ghs * z
    double gz;              // AXIS_Z: -gamma * (hk r_z^2 - ms N_z), so that -gamma H_z(det) = gz * m_z
    double agz;             // alpha * gz
};
// builds the folded constants from the class-table values (once per lane per launch)
__device__ __forceinline__ LlgsK make_llgs(const V3& r, const V3& d, double hk, double hex, double alpha, double gamma, double hs) {
#pragma clang fp contract(off)
    const double ng = -gamma;
    const double gz = ng * ((hk * r.z) * r.z + d.z);
    return LlgsK{r, V3{ng * d.x, ng * d.y, ng * d.z}, ng * hk, ng * hex, alpha, ng * hs, gz, alpha * gz};
}


// A1 + A2: SimpleLLGSSolver._compute_dmdt with h_applied = 0 (simple_solver.py:297-388), regrouped.  With
//   H = hk (m.e) e - Ms m_z z^ (+ hs z),   g' = gamma/(1+alpha^2),   G = -g' H,   t = m x e:
//   dm/dt = -g' (m x H + alpha m x (m x H)) + aJ m x (m x e)  =  p + m x (alpha p + aJ t),   p = m x G.
// -g' rides in the constants (ghk, gdm, ghs), so nothing is scaled at the end: 27 fp64 instructions at T = 0 K instead
// of 42 for the literal form (11 with the easy axis along z); the intermediates are the reference's own g'-scaled
// quantities, so overflow happens at the same sub-step.
// ghk: -g' H_k of this stage (a VCMA lane of the device-physics model has a different one while its pulse is on).
template <bool THERMAL, bool AXIS_Z>
__device__ __forceinline__ V3 simple_rhs(const V3& m, const SimpleK& k, double ghk, double aJ, const V3& z) {
    // every FMA of the fixed-step path is written out and contraction is off, so all instantiations of this source
    // (one or two wavefronts per workgroup, any launch size) round identically: results do not depend on the partition
#pragma clang fp contract(off)
    if (AXIS_Z) {
        // easy axis = +z exactly (every factory default, device_factory.py:129-172): t = (my, -mx, 0), G = (0, 0, gz),
        // so the products with the axis' zero components drop out.  (They only differ from the general form when a
        // factor is already inf/NaN, where both forms end non-finite.)
        const double gz = (ghk + k.gdm) * m.z;
        V3 p;
        if (THERMAL) {
            const V3 g{k.ghs * z.x, k.ghs * z.y, __builtin_fma(k.ghs, z.z, gz)};
            p = cross(m, g);
        } else {
            p = V3{m.y * gz, -(m.x * gz), 0.0};
        }
        const double wx = __builtin_fma(k.alpha, p.x, aJ * m.y), wy = __builtin_fma(k.alpha, p.y, -(aJ * m.x));
        if (THERMAL) {
            const double wz = k.alpha * p.z;
            return V3{__builtin_fma(m.y, wz, __builtin_fma(-m.z, wy, p.x)), __builtin_fma(m.z, wx, __builtin_fma(-m.x, wz, p.y)),
                      __builtin_fma(m.x, wy, __builtin_fma(-m.y, wx, p.z))};
        }
        return V3{__builtin_fma(-m.z, wy, p.x), __builtin_fma(m.z, wx, p.y), __builtin_fma(m.x, wy, -(m.y * wx))};
    }
    const V3 t = cross(m, k.e);
    const double c = ghk * dot(m, k.e);
    const double d = k.gdm * m.z;
    V3 p;
    if (THERMAL) {
        const V3 g{__builtin_fma(k.ghs, z.x, c * k.e.x), __builtin_fma(k.ghs, z.y, c * k.e.y),
                   __builtin_fma(k.ghs, z.z, __builtin_fma(c, k.e.z, d))};                // simple_solver.py:384,388
        p = cross(m, g);
    } else {
        p = V3{__builtin_fma(c, t.x, d * m.y), __builtin_fma(c, t.y, -(d * m.x)), c * t.z};
    }
    const V3 w{__builtin_fma(k.alpha, p.x, aJ * t.x), __builtin_fma(k.alpha, p.y, aJ * t.y),
               __builtin_fma(k.alpha, p.z, aJ * t.z)};
    return V3{__builtin_fma(m.y, w.z, __builtin_fma(-m.z, w.y, p.x)), __builtin_fma(m.z, w.x, __builtin_fma(-m.x, w.z, p.y)),
              __builtin_fma(m.x, w.y, __builtin_fma(-m.y, w.x, p.z))};
}

// ---- opt-in device-physics torque model (BASELINE config 4 "divergent torque terms") -------------------
// The reference env integrates the same type-agnostic RHS for every device type; its device classes carry torque
// formulas the env never calls.  With cfg.torque_model = 1 the Simple RHS uses them per lane:
//   STT  : Slonczewski term of the reference RHS (unchanged)
//   SOT  : no Slonczewski term; + [tau_dl J (sigma x m) + tau_fl J sigma] / (ms V)      SOTMRAMDevice.compute_spin_torque
//   VCMA : Slonczewski term, with H_k from K_eff(V), V = J R(m) A while the pulse is on    VCMAMRAMDevice._compute_effective_anisotropy
// (the 1/(ms V) normalisation is the one the reference applies to its own torque vector, simple_solver.py:330).
// Lanes differ only in coefficients (predication, no divergent control flow); `any_sot` is wave-uniform (ballot), so
// wavefronts without a SOT lane skip the extra cross product -- and the lane schedule groups lanes by type.
struct DevTorque {
    double sdl, sfl;      // tau_dl J/(ms V), tau_fl J/(ms V) for SOT lanes, else 0
    V3 sigma;
    double ghk_pulse;     // -gamma' H_k while the pulse is on (VCMA: from K_eff(V)); = ghk for other types
    bool any_sot;         // wave-uniform
};

// K_eff(V) of vcma_mram.py:122-147
__device__ __forceinline__ double vcma_keff(double volt, double ku, double xi, double td2, double vbd) {
    const double v = volt < -vbd ? -vbd : (volt > vbd ? vbd : volt);
    const double keff = ku + (-xi * fabs(v) / td2);
    return fmax(keff, -0.5 * ku);
}

template <bool THERMAL, bool AXIS_Z, bool DEVPHYS>
__device__ __forceinline__ V3 simple_stage(const V3& m, const SimpleK& k, double aJ, const V3& z, const DevTorque& dv,
                                           bool on) {
#pragma clang fp contract(off)
    if (!DEVPHYS) return simple_rhs<THERMAL, AXIS_Z>(m, k, k.ghk, aJ, z);
    V3 f = simple_rhs<THERMAL, AXIS_Z>(m, k, on ? dv.ghk_pulse : k.ghk, aJ, z);
    if (dv.any_sot) {
        const double a = on ? dv.sdl : 0.0, b = on ? dv.sfl : 0.0;
        const V3 sxm = cross(dv.sigma, m);
        f = V3{f.x + __builtin_fma(a, sxm.x, b * dv.sigma.x), f.y + __builtin_fma(a, sxm.y, b * dv.sigma.y),
               f.z + __builtin_fma(a, sxm.z, b * dv.sigma.z)};
    }
    return f;
}

// SimpleLLGSSolver._validate_magnetization (simple_solver.py:208-229).
// returns 0 = normalised, 1 = reset to +z; sets zero_row when the quotient is the all-zero row m/inf.
__device__ __forceinline__ int simple_validate(V3& m, bool& zero_row) {
#pragma clang fp contract(off)
    zero_row = false;
    // First tier (every RK4 sub-step in practice): |m|^2 within 2^-10 of 1 for the whole wavefront -- a step of a
    // unit vector leaves it there -- so 1/|m| = (1 + e)^-1/2 comes from four Horner FMAs (truncation 0.25 |e|^5 < 2.3e-16)
    // instead of v_rsq_f64 and its correction.
    // A lane's arithmetic must not depend on its wavefront-mates (the lane schedule changes them from run to run): the
    // wave-uniform tests below only skip work, a lane near |m| = 1 uses the series on either path.
    const double e = __builtin_fma(m.z, m.z, __builtin_fma(m.y, m.y, __builtin_fma(m.x, m.x, -1.0)));
    const bool near1 = fabs(e) < 0.0009765625;
    const double inv1 = __builtin_fma(e, __builtin_fma(e, __builtin_fma(e, __builtin_fma(e, 0.2734375, -0.3125), 0.375), -0.5), 1.0);
    if (__builtin_expect(__ballot(!near1) == 0ull, 1)) {
        m = V3{m.x * inv1, m.y * inv1, m.z * inv1};
        return 0;
    }
    const double s = dot(m, m);
    const double inv = near1 ? inv1 : rsqrt_fast(s);   // (s = inf -- finite m, overflowed norm: the reference's zero row -- is flagged below)
    // ordinary case for the whole wavefront: 1e-24 <= |m|^2 <= DBL_MAX implies finite components (a NaN or inf
    // component makes s NaN or inf) -- the special cases sit behind a wave-uniform branch
    const bool ordinary = (s >= 1e-24) && (s <= 1.7976931348623157e308);
    if (__builtin_expect(__ballot(!ordinary) == 0ull, 1)) {
        m = V3{m.x * inv, m.y * inv, m.z * inv};
        return 0;
    }
    // non-finite components, or |m| < 1e-12 (s < 1e-24): the reference's "safe default" [0,0,1]
    if (!finite3(m) || s < 1e-24) { m = V3{0.0, 0.0, 1.0}; return 1; }
    m = V3{m.x * inv, m.y * inv, m.z * inv};
    zero_row = isinf(s);
    return 0;
}

// validation.validate_magnetization as a predicate (utils/validation.py:46-51)
__device__ __forceinline__ bool validation_rejects(const V3& m) {
    return !finite3(m) || sqrt(dot(m, m)) < 1e-12;
}

struct SolveOut {
    V3 m;            // final row (unchanged input when !ok)
    int32_t n;       // RK4/Euler: sub-steps; RK45: accepted points excluding t0
    int32_t resets;  // sub-steps that took the non-finite -> +z branch
    int64_t work;    // integrator work units: RK4/Euler sub-steps, RK45 attempted steps
    bool ok;
};

// Optional trajectory recorder (stg_solve_traj): rows are written with env index fastest.
struct Recorder {
    double* t;       // [cap][N]
    double* m;       // [cap][3][N]
    double* e;       // [cap][N] or nullptr
    double* tq;      // [cap][N] or nullptr: |tau_stt| + |tau_fl| per accepted point (llgs_solver.py:159-172)
    int64_t N, i;
    int32_t cap;
    __device__ __forceinline__ void put(int32_t row, double tt, const V3& mm, double ee, double tqv = 0.0) const {
        if (row < cap) {
            if (t) t[(int64_t)row * N + i] = tt;
            if (m) {
                double* b = m + (int64_t)row * 3 * N + i;
                b[0] = mm.x; b[N] = mm.y; b[2 * N] = mm.z;
            }
            if (e) e[(int64_t)row * N + i] = ee;
            if (tq) tq[(int64_t)row * N + i] = tqv;
        }
    }
};

// A3 + A4 + A5: RobustLLGSSolver.solve -> SimpleLLGSSolver.solve, METHOD 0 = rk4, 1 = euler.
template <int METHOD, bool THERMAL, bool RECORD, bool AXIS_Z, bool DEVPHYS, class NSRC>
__device__ __forceinline__ SolveOut simple_solve(const V3& m0, double J, double T, const SimpleK& k, double pol,
                                                 double msv, bool class_valid, double temperature, double max_step,
                                                 const RngKey& rk, const Recorder& rec, const DevTorque& dv, NSRC& ns,
                                                 double inv_tau, bool enabled) {
#pragma clang fp contract(off)
    SolveOut o{m0, 0, 0, 0, false};
    // robust_solver.py:152-190 (_validate_inputs); any failure ends in the fallback result (:140-150)
    // `enabled` = false: a lane that only walks the workgroup's chunk loop (SharedNormals; no env behind it, or one
    // that is not stepped this time)
    const bool rejected_in = !enabled || validation_rejects(m0) || !(T > 0.0) || !class_valid || !(temperature > 0.0);
    if (!NSRC::kShared && rejected_in) return o;
    V3 m = m0;
    bool zr;
    int resets = simple_validate(m, zr);                                   // simple_solver.py:119
    // simple_solver.py:137-139, in exactly these roundings
    double dt = fmin(max_step, (T / 100.0));
    int n = (int)(T / dt);
    n = n < 10 ? 10 : n;
    dt = T / (double)n;
    // SharedNormals: every lane of the wavefront walks the (wave-uniform) chunk loop; a rejected lane has no sub-steps
    if (NSRC::kShared && rejected_in) n = 0;
    o.n = n;
    o.work = n;
    const double half_dt = 0.5 * dt, sixth_dt = dt / 6.0;
    const bool useJ = fabs(J) > 1e-12;                                     // simple_solver.py:326
    const double aJ = useJ ? (pol * J) / msv : 0.0;                        // simple_solver.py:330
    const double kJ = aJ;                                                  // (the RHS takes a_J itself, see simple_rhs)
    // Stage times are t_i = i*dt (np.linspace), t_i + dt/2, t_i + dt and the pulse is on while t <= T
    // (spin_torque_env.py:442-443).  For i <= n-2 every stage time is below T by at least dt/2; only the last
    // sub-step's k2/k3/k4 stages can land an ulp beyond T and see J = 0, so only that one is tested,
    // in exactly the reference's roundings.
    const double t_last = mul_x((double)(n - 1), dt);
    const bool on2_last = add_x(t_last, mul_x(dt, 0.5)) <= T, on4_last = add_x(t_last, dt) <= T;
    const double kJ2_last = on2_last ? kJ : 0.0;
    const double kJ4_last = on4_last ? kJ : 0.0;
    bool fail = false;
    const V3 zero{0.0, 0.0, 0.0};
    if (THERMAL) ns.begin(rk);
    // Ornstein-Uhlenbeck field (ThermalFluctuations._generate_correlated_noise, thermal_model.py:113-137), selected by a
    // wave-uniform inv_tau > 0: one update per sub-step, x <- d x + sqrt(1 - d^2) xi with d = exp(-dt/tau), and the same
    // x for every stage of the sub-step; the stream hands out three normals per sub-step, alternating phase like Euler's
    const bool ou_sel = THERMAL && inv_tau > 0.0;
    double ou_d = 0.0, ou_c = 1.0;
    V3 ou_x = zero;
    if (ou_sel) {
        ou_d = exp(-dt * inv_tau);
        ou_c = sqrt(1.0 - ou_d * ou_d);
    }
    if (RECORD) rec.put(0, 0.0, m, 0.0);
    // per-lane trip count n; with SharedNormals the loop is wave-uniform (lanes past their n idle inside the body).
    // The loop exists twice (white / Ornstein-Uhlenbeck field), chosen once outside: `ou` is wave-uniform.
    auto run = [&](auto ou_tag) {
    constexpr bool ou = decltype(ou_tag)::value;
    // one sub-step; WHICH: 0 = not the last one (pulse on in every stage), 1 = the last one (stage gates of H4),
    // 2 = decided at run time (SharedNormals: every sub-step is a chunk of the wave-uniform loop)
    auto substep = [&](int i, auto which_tag) {
        constexpr int WHICH = decltype(which_tag)::value;
        const bool last = WHICH == 2 ? (i == n - 1) : (WHICH == 1);
        const double kJ2 = last ? kJ2_last : kJ, kJ4 = last ? kJ4_last : kJ;
        const bool on2 = !last || on2_last, on4 = !last || on4_last;
        V3 mn;
        if (METHOD == 1) {
            V3 z0 = zero;
            if (THERMAL) z0 = ns.draw((i & 1) == 0);
            if (ou) {
                ou_x = V3{__builtin_fma(ou_d, ou_x.x, ou_c * z0.x), __builtin_fma(ou_d, ou_x.y, ou_c * z0.y),
                          __builtin_fma(ou_d, ou_x.z, ou_c * z0.z)};
                z0 = ou_x;
            }
            const V3 f = simple_stage<THERMAL, AXIS_Z, DEVPHYS>(m, k, kJ, z0, dv, true);
            mn = V3{__builtin_fma(dt, f.x, m.x), __builtin_fma(dt, f.y, m.y), __builtin_fma(dt, f.z, m.z)};   // simple_solver.py:275-276
        } else {
            V3 z0 = zero, z1 = zero, z2 = zero, z3 = zero;
            if (ou) {
                const V3 xi = ns.draw((i & 1) == 0);
                ou_x = V3{__builtin_fma(ou_d, ou_x.x, ou_c * xi.x), __builtin_fma(ou_d, ou_x.y, ou_c * xi.y),
                          __builtin_fma(ou_d, ou_x.z, ou_c * xi.z)};
                z0 = ou_x; z1 = ou_x; z2 = ou_x; z3 = ou_x;
            } else if (THERMAL) {   // 12 normals = 6 Box-Muller pairs per sub-step
                z0 = ns.draw(true); z1 = ns.draw(false); z2 = ns.draw(true); z3 = ns.draw(false);
            }
            const V3 f1 = simple_stage<THERMAL, AXIS_Z, DEVPHYS>(m, k, kJ, z0, dv, true);
            const V3 y2{__builtin_fma(half_dt, f1.x, m.x), __builtin_fma(half_dt, f1.y, m.y), __builtin_fma(half_dt, f1.z, m.z)};
            const V3 f2 = simple_stage<THERMAL, AXIS_Z, DEVPHYS>(y2, k, kJ2, z1, dv, on2);
            const V3 y3{__builtin_fma(half_dt, f2.x, m.x), __builtin_fma(half_dt, f2.y, m.y), __builtin_fma(half_dt, f2.z, m.z)};
            const V3 f3 = simple_stage<THERMAL, AXIS_Z, DEVPHYS>(y3, k, kJ2, z2, dv, on2);
            if (THERMAL) ns.peek();
            const V3 y4{__builtin_fma(dt, f3.x, m.x), __builtin_fma(dt, f3.y, m.y), __builtin_fma(dt, f3.z, m.z)};
            const V3 f4 = simple_stage<THERMAL, AXIS_Z, DEVPHYS>(y4, k, kJ4, z3, dv, on4);
            // m + (k1 + 2 k2 + 2 k3 + k4)/6 with k = dt*f                  simple_solver.py:290-295
            mn = V3{__builtin_fma(sixth_dt, __builtin_fma(2.0, f2.x, f1.x) + __builtin_fma(2.0, f3.x, f4.x), m.x),
                    __builtin_fma(sixth_dt, __builtin_fma(2.0, f2.y, f1.y) + __builtin_fma(2.0, f3.y, f4.y), m.y),
                    __builtin_fma(sixth_dt, __builtin_fma(2.0, f2.z, f1.z) + __builtin_fma(2.0, f3.z, f4.z), m.z)};
        }
        resets += simple_validate(mn, zr);                                 // simple_solver.py:168
        fail |= zr;                                                        // robust_solver.py:192-205
        m = mn;
        if (RECORD) rec.put(i + 1, last ? T : mul_x((double)(i + 1), dt), m, 0.0);
    };
    if (NSRC::kShared) {
        // a chunk of the shared stream is SHARED_CHUNK_FIXED normals: SHARED_SUBS_RK4 RK4 sub-steps with the white field, or
        // SHARED_SUBS3 sub-steps of the forms that take three per sub-step (Euler, the Ornstein-Uhlenbeck field) -- the
        // rendezvous with the producer costs about as much as one Euler sub-step
        constexpr int SUBS = (METHOD == 1 || ou) ? SHARED_SUBS3 : SHARED_SUBS_RK4;
        for (int i = 0;; ++i) {
            if (i < n) substep(i, std::integral_constant<int, 2>{});
            if (SUBS == 1 || (i & (SUBS - 1)) == SUBS - 1) {
                if (!ns.chunk_end(i + 1 < n)) break;
            }
        }
    } else {
        // a counted, lane-divergent loop over the first n-1 sub-steps, then the last one with the stage gates (n >= 10)
        for (int i = 0; i < n - 1; ++i) substep(i, std::integral_constant<int, 0>{});
        substep(n - 1, std::integral_constant<int, 1>{});
    }
    };
    if (THERMAL && ou_sel) run(std::true_type{}); else run(std::false_type{});
    const bool rejected_shared = NSRC::kShared && rejected_in;
    o.resets = rejected_shared ? 0 : resets;
    if (fail || rejected_shared || ns.broken) return o;
    o.m = m;
    o.ok = true;
    return o;
}

// A6: LLGSSolver.solve::llgs_rhs.  bJ = beta*J, bpJ = beta'*J (0 when |J| < 1e-12 or the pulse is over); ht = ghs * z is the
// thermal field of this call already times -gamma (scale3).  With G = -gamma H and p_hat = z (llgs_solver.py:226-235) the
// reference's sum  m x G + alpha m x (m x G) + bJ m x (m x z) + bpJ m x z  is regrouped as
//     dm = m x U + m x (m x W),   U = G + bpJ z,   W = alpha G + bJ z,
// and, m being the unit vector formed two lines above,  m x (m x W) = m (m.W) - W:
//     dm = m x U + m (m.W) - W
// -- 18 fp64 instructions after the normalisation with the thermal field (8 at T = 0 K) instead of 26 (19) for the four separate
// products; the two forms differ by (|m|^2 - 1) W, i.e. by the rounding of the normalisation (<= 4e-16 |W|).
// AXIS_Z: raw easy axis = (0,0,rz) and demag factors = (0,0,Nz) (every factory default): G = ghex*m + (0,0,gz m_z) + ht.
// The exchange placeholder hex*m (llgs_solver.py:205-209) is parallel to m, so it cancels in both products up to a rounding
// residue of ~1e-16 * hex (hex ~ 4e-6 A/m against H_k ~ 2.4e6 A/m) and is dropped there; the general form keeps it.
// CHECKED: |y| > 1e-12 -> y/|y|, else +z (llgs_solver.py:97-101), behind a wave-uniform branch that is never taken in practice.
// The prologue's two calls are CHECKED.  Inside the attempt loop the test cannot fire on a finite state (y starts as a unit vector,
// an accepted step keeps it one to ~1e-6 and a stage point is y + h (...) with h |f| << 1), and for a NaN state the replacement is
// unobservable: y, hence every stage point, y_new and the error norm are NaN with or without it, the attempt is rejected with
// factor 0.2 either way, and the solve fails after the same number of attempts -- so the loop's seven calls skip the compare and
// the branch (a lone wavefront stalls on each: compare -> VCC -> branch).
template <bool THERMAL, bool AXIS_Z, bool CHECKED = true>
__device__ __forceinline__ V3 llgs_rhs(const V3& y, const LlgsK& k, double bJ, double bpJ, const V3& ht) {
#pragma clang fp contract(off)
    const double ss = dot(y, y);
    const double inv = rsqrt_fast(ss);
    V3 m{y.x * inv, y.y * inv, y.z * inv};
    if (CHECKED) {
        const bool unit = ss > 1e-24;
        if (__builtin_expect(__ballot(!unit) != 0ull, 0)) m = V3{unit ? m.x : 0.0, unit ? m.y : 0.0, unit ? m.z : 1.0};
    }
    if (AXIS_Z) {
        const double u = __builtin_fma(k.gz, m.z, bpJ);                         // U = ht + u z
        const double w = __builtin_fma(k.agz, m.z, bJ);                         // W = alpha ht + w z
        if (!THERMAL) {
            const double c = m.z * w;                                           // m.W
            return V3{__builtin_fma(m.x, c, m.y * u), __builtin_fma(m.y, c, -(m.x * u)), __builtin_fma(m.z, c, -w)};
        }
        const double uz = ht.z + u;                                             // llgs_solver.py:111-113
        const V3 W{k.alpha * ht.x, k.alpha * ht.y, __builtin_fma(k.alpha, ht.z, w)};
        const double s = dot(m, W);
        return V3{__builtin_fma(m.x, s, __builtin_fma(m.y, uz, __builtin_fma(-m.z, ht.y, -W.x))),
                  __builtin_fma(m.y, s, __builtin_fma(m.z, ht.x, __builtin_fma(-m.x, uz, -W.y))),
                  __builtin_fma(m.z, s, __builtin_fma(m.x, ht.y, __builtin_fma(-m.y, ht.x, -W.z)))};
    }
    const double c = k.ghk * dot(m, k.r);
    V3 g{(c * k.r.x + k.gd.x * m.x) + k.ghex * m.x, (c * k.r.y + k.gd.y * m.y) + k.ghex * m.y,
         (c * k.r.z + k.gd.z * m.z) + k.ghex * m.z};
    if (THERMAL) g = V3{g.x + ht.x, g.y + ht.y, g.z + ht.z};                                // llgs_solver.py:111-113
    const V3 U{g.x, g.y, g.z + bpJ};
    const V3 W{k.alpha * g.x, k.alpha * g.y, __builtin_fma(k.alpha, g.z, bJ)};
    const double s = dot(m, W);
    return V3{__builtin_fma(m.x, s, __builtin_fma(m.y, U.z, __builtin_fma(-m.z, U.y, -W.x))),
              __builtin_fma(m.y, s, __builtin_fma(m.z, U.x, __builtin_fma(-m.x, U.z, -W.y))),
              __builtin_fma(m.z, s, __builtin_fma(m.x, U.y, __builtin_fma(-m.y, U.x, -W.z)))};
}

struct LlgsEnergyK {
    double kuv, edemag;
    V3 r, nfac;
};
// llgs_solver.py:239-262 with h_applied = 0
__device__ __forceinline__ double llgs_energy(const V3& m, const LlgsEnergyK& k) {
    const double ct = dot(m, k.r);
    return -k.kuv * (ct * ct) + k.edemag * (k.nfac.x * (m.x * m.x) + k.nfac.y * (m.y * m.y) + k.nfac.z * (m.z * m.z));
}

// A8: |tau_stt| + |tau_fl| of LLGSSolver._compute_spin_torques at a (renormalised) trajectory point, p_hat = z
// (llgs_solver.py:159-172,213-237): tau_stt = beta J m x (m x z), tau_fl = beta' J (m x z); zero when |J| < 1e-12.
__device__ __forceinline__ double llgs_torque_norms(const V3& m, double bJ, double bpJ) {
#pragma clang fp contract(off)
    const V3 mxp{m.y, -m.x, 0.0};                           // m x z
    const V3 mm = cross(m, mxp);
    const V3 a{bJ * mm.x, bJ * mm.y, bJ * mm.z}, b{bpJ * mxp.x, bpJ * mxp.y, bpJ * mxp.z};
    return sqrt((a.x * a.x + a.y * a.y) + a.z * a.z) + sqrt((b.x * b.x + b.y * b.y) + b.z * b.z);
}

__device__ __forceinline__ double rms3(const V3& a) { return sqrt(dot(a, a)) / 1.7320508075688772; }   // common.py:63-65

#ifdef STG_PROFILE_LOOP
// experiment builds only (-DSTG_PROFILE_LOOP): per integrating wavefront (first STG_PROF_WAVES of the launch) start/end
// s_memtime, start/end s_memrealtime (100 MHz), HW_ID, attempts of its worst lane -- where each wavefront ran, for how long,
// and at which shader clock; read back with stg_debug_waves() (tools/probe_wave_records.py)
#define STG_PROF_WAVES 8192
static __device__ long long g_stg_wave[STG_PROF_WAVES * 6];
struct WaveProf {
    long long t0, r0;
    __device__ __forceinline__ void start() { t0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
    __device__ __forceinline__ void stop(long long attempts) {
        const int wid_ = blockIdx.x * ((int)blockDim.x / 64) + (int)threadIdx.x / 64;
        long long att_max_ = attempts;
        for (int o_ = 32; o_ > 0; o_ >>= 1) { const long long v_ = __shfl_xor(att_max_, o_); att_max_ = v_ > att_max_ ? v_ : att_max_; }
        if ((threadIdx.x & 63) == 0 && wid_ < STG_PROF_WAVES) {
            long long* r_ = g_stg_wave + 6 * wid_;
            r_[0] = t0; r_[1] = __builtin_readcyclecounter(); r_[2] = r0; r_[3] = __builtin_amdgcn_s_memrealtime();
            r_[4] = (long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11))      // HW_REG_HW_ID, 32 bits
                    | ((long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32);    // HW_REG_XCC_ID
            r_[5] = att_max_;
        }
    }
};
#else
struct WaveProf {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void stop(long long) {}
};
#endif

// The solve is split into begin / attempt / finish on a per-lane state (LlgsLane), so that the same arithmetic serves the
// one-env-per-lane loop (llgs_solve) and the lane-refill step kernel (stg_kernels.hpp: stg_step_refill_kernel), in which a
// lane that has finished its env takes the next one of its wavefront's queue while its neighbours keep integrating.
//
// Dormand-Prince tableau (rk.py:380-391).  The 25 tableau constants do not fit next to everything else that is wave-uniform:
// with all of them in SGPR pairs the kernel sits at the SGPR limit and the compiler re-materialises 15-21 halves per attempt
// inside the loop (s_mov_b32: one issue slot each for a lone wavefront).  The rows used once per attempt live in VGPRs instead
// (opaque to the optimiser, set once before the loop; same doubles): 439 -> 420 instructions per attempt at T = 0 K, 504 -> 492
// for the wave-specialised thermal kernel; measured 1.555 -> 1.475 ms at 4096 envs, 2.16 -> 2.11 ms on the headline launch.
// (stage nodes C = 1/5, 3/10, 4/5, 8/9, 1, 1: the RHS is autonomous but for the pulse gate, see llgs_lane_attempt)
struct Dp5Tab {
    double A51, A52, A53, A54, A61, A62, A63, A64, A65, E1, E3, E4, E5, E6, E7;
};
__device__ __forceinline__ Dp5Tab make_dp5_tab() {
    Dp5Tab t;
    t.A51 = vgpr_const(19372.0 / 6561); t.A52 = vgpr_const(-25360.0 / 2187); t.A53 = vgpr_const(64448.0 / 6561); t.A54 = vgpr_const(-212.0 / 729);
    t.A61 = vgpr_const(9017.0 / 3168); t.A62 = vgpr_const(-355.0 / 33); t.A63 = vgpr_const(46732.0 / 5247); t.A64 = vgpr_const(49.0 / 176);
    t.A65 = vgpr_const(-5103.0 / 18656);
    t.E1 = vgpr_const(-71.0 / 57600); t.E3 = vgpr_const(71.0 / 16695); t.E4 = vgpr_const(-71.0 / 1920); t.E5 = vgpr_const(17253.0 / 339200);
    t.E6 = vgpr_const(-22.0 / 525); t.E7 = vgpr_const(1.0 / 40);
    return t;
}

// one lane's solve in flight
struct LlgsLane {
    V3 y, f;                 // state and f(y) (FSAL)
    V3 m0;                   // the input row (returned when the solve fails)
    double t, T, h_abs, min_step;
    double bJ, bpJ;          // beta J, beta' J (0 when |J| < 1e-12)
    int64_t attempts;
    int32_t npts;
    bool ok, rejected, active;
};

__device__ __forceinline__ double llgs_min_step_at(double tt) {   // 10 * |nextafter(t, inf) - t|, t >= 0                      rk.py:119
    return 10.0 * (__longlong_as_double(__double_as_longlong(tt) + 1) - tt);
}

// every accepted point is renormalised on output (llgs_solver.py:152-153); without a recorder only the last one is ever
// read, so it is formed once after the loop (llgs_lane_finish)
template <bool RECORD>
__device__ __forceinline__ void llgs_lane_emit(LlgsLane& L, V3& out_m, const Recorder& rec, const LlgsEnergyK& ek) {
    const double inv = rsqrt_fast(dot(L.y, L.y));
    out_m = V3{L.y.x * inv, L.y.y * inv, L.y.z * inv};
    // (the recorded time points never pass T, so current_func(t) = J at every one of them)
    if (RECORD) rec.put(L.npts, L.t, out_m, rec.e ? llgs_energy(out_m, ek) : 0.0, rec.tq ? llgs_torque_norms(out_m, L.bJ, L.bpJ) : 0.0);
    ++L.npts;
}

// RHS call; the pulse gate of spin_torque_env.py:442-443 (J while t <= T) can only close for stage times of the form
// fl(t + h) on the step that is clamped to end at T: every other stage time is fl(t + fl(c h)) with c <= 8/9, hence
// <= t_new <= T by monotonic rounding, and an unclamped fl(t + h) is within an ulp of t_new < T.  So only k6 / f_new of an
// attempt test the gate (`on`).
template <bool THERMAL, bool AXIS_Z, bool CHECKED = true>
__device__ __forceinline__ V3 llgs_fun(const LlgsLane& L, const LlgsK& k, const V3& y, const V3& ht, bool on) {
    return llgs_rhs<THERMAL, AXIS_Z, CHECKED>(y, k, on ? L.bJ : 0.0, on ? L.bpJ : 0.0, ht);
}
// the thermal field of one RHS call (already times -gamma); EVEN selects the normal-stream phase (calls alternate).  Fetch and
// evaluation are separate so that a loop can fetch call j+1's field before it evaluates call j: with the shared source a
// draw is three LDS reads, and issued at the point of use each of them stalls the lone integrating wavefront for the LDS
// latency (six stalls per attempt).
template <bool THERMAL, class NSRC>
__device__ __forceinline__ V3 llgs_draw(NSRC& ns, const LlgsK& k, bool even) {
    if (!THERMAL) return V3{0.0, 0.0, 0.0};
    return NSRC::kScaled ? ns.draw(even) : scale3(k.ghs, ns.draw(even));
}

// Prologue of a solve: normalise m0 (llgs_solver.py:76), f(t0, y0), select_initial_step (common.py:68-134).
// `enabled` = false: a lane that only walks its workgroup's chunk loop (SharedNormals).
template <bool THERMAL, bool RECORD, bool AXIS_Z, class NSRC>
__device__ __forceinline__ void llgs_lane_begin(LlgsLane& L, V3& out_m, const V3& m0, double J, double T, const LlgsK& k, double beta,
                                                double betap, double rtol, double atol, double max_step, const RngKey& rk,
                                                const Recorder& rec, const LlgsEnergyK& ek, NSRC& ns, bool enabled) {
    const bool useJ = !(fabs(J) < 1e-12);                                   // llgs_solver.py:222
    L.bJ = useJ ? beta * J : 0.0;
    L.bpJ = useJ ? betap * J : 0.0;
    L.m0 = m0;
    L.T = T;
    if (THERMAL) ns.begin(rk);
    const double n0 = rsqrt_fast(dot(m0, m0));                              // llgs_solver.py:76
    L.y = V3{m0.x * n0, m0.y * n0, m0.z * n0};
    L.t = 0.0;
    L.npts = 0;
    if (RECORD) llgs_lane_emit<RECORD>(L, out_m, rec, ek); else ++L.npts;
    L.f = llgs_fun<THERMAL, AXIS_Z>(L, k, L.y, llgs_draw<THERMAL>(ns, k, true), true);        // t = 0 <= T
    double h_abs;
    {   // select_initial_step (common.py:68-134), order = error_estimator_order = 4
        // (quotients by reciprocal-multiply, ~1 ulp: these norms only seed the first step size)
        const V3& y = L.y;
        const V3& f = L.f;
        const V3 sc{atol + fabs(y.x) * rtol, atol + fabs(y.y) * rtol, atol + fabs(y.z) * rtol};
        const V3 isc{rcp_fast(sc.x), rcp_fast(sc.y), rcp_fast(sc.z)};
        const double d0 = rms3(V3{y.x * isc.x, y.y * isc.y, y.z * isc.z});
        const double d1 = rms3(V3{f.x * isc.x, f.y * isc.y, f.z * isc.z});
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : (0.01 * d0) * rcp_fast(d1);
        h0 = fmin(h0, T);
        const V3 y1{y.x + h0 * f.x, y.y + h0 * f.y, y.z + h0 * f.z};
        const V3 f1 = llgs_fun<THERMAL, AXIS_Z>(L, k, y1, llgs_draw<THERMAL>(ns, k, false), true);   // t + h0 = h0 <= T (h0 = min(h0, T))
        const double d2 = rms3(V3{(f1.x - f.x) * isc.x, (f1.y - f.y) * isc.y, (f1.z - f.z) * isc.z}) * rcp_fast(h0);
        const double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : fifth_root(0.01 * rcp_fast(fmax(d1, d2)));
        h_abs = fmin(fmin(100.0 * h0, h1), fmin(T, max_step));
    }
    L.ok = true;
    L.rejected = false;
    L.attempts = 0;
    L.min_step = llgs_min_step_at(L.t);
    L.h_abs = h_abs > max_step ? max_step : (h_abs < L.min_step ? L.min_step : h_abs);          // rk.py:121-126
    L.active = enabled && (L.t != T);       // (a disabled lane only walks the workgroup's chunk loop)
}

// The budget / minimum-step test that opens an attempt (rk.py:132-133 + the attempt budget); a lane that fails it stops.
__device__ __forceinline__ void llgs_lane_gate(LlgsLane& L, int64_t max_attempts) {
    // (bitwise on purpose: no short-circuit control flow inside the wave-uniform attempt loop)
    const bool fail_now = L.active & ((L.h_abs < L.min_step) | (L.attempts >= max_attempts));
    L.ok = L.ok & !fail_now;
    L.active = L.active & !fail_now;
}

// ONE attempted step (rk.py:111-181) of a lane; the body has no lane-divergent control flow: a lane that is through (or
// never started) walks along with its state frozen by the selects below.  SciPy nests "while not finished: step()" around
// "while not step_accepted: attempt" (ivp.py:654-661, base.py:175-206); here an accepted attempt performs the outer loop's
// bookkeeping itself: same sequence of attempts per lane, but a wavefront needs max-over-lanes(total attempts) iterations
// instead of sum-over-steps(max-over-lanes(attempts of that step)).
// z2, z3: the thermal fields of the first two RHS calls of the attempt (fetched by the caller, see llgs_draw).
template <bool THERMAL, bool RECORD, bool AXIS_Z, class NSRC>
__device__ __forceinline__ void llgs_lane_attempt(LlgsLane& L, V3& out_m, const LlgsK& k, const Dp5Tab& tb, double rtol, double atol,
                                                  double max_step, const Recorder& rec, const LlgsEnergyK& ek, NSRC& ns, V3 z2, V3 z3) {
    constexpr double A21 = 1.0 / 5;
    constexpr double A31 = 3.0 / 40, A32 = 9.0 / 40;
    constexpr double A41 = 44.0 / 45, A42 = -56.0 / 15, A43 = 32.0 / 9;
    constexpr double B1 = 35.0 / 384, B3 = 500.0 / 1113, B4 = 125.0 / 192, B5 = -2187.0 / 6784, B6 = 11.0 / 84;
    const double A51 = tb.A51, A52 = tb.A52, A53 = tb.A53, A54 = tb.A54, A61 = tb.A61, A62 = tb.A62, A63 = tb.A63, A64 = tb.A64, A65 = tb.A65;
    const double E1 = tb.E1, E3 = tb.E3, E4 = tb.E4, E5 = tb.E5, E6 = tb.E6, E7 = tb.E7;
    const double T = L.T;
    const bool active = L.active;
    const V3 y = L.y;
    const double t = L.t;
    auto fun = [&](const V3& yy, const V3& ht, bool on) -> V3 { return llgs_fun<THERMAL, AXIS_Z, false>(L, k, yy, ht, on); };
    auto draw = [&](bool even) -> V3 { return llgs_draw<THERMAL>(ns, k, even); };
    L.attempts += active ? 1 : 0;
    const double t_new = fmin(add_x(t, L.h_abs), T);                       // rk.py:135-138: if t_new - T > 0: t_new = T
    const double h = sub_x(t_new, t);
    const double h_try = fabs(h);
    // rk_step (rk.py:14-70); the one stage time that can pass T is formed without contraction
    const bool on_end = add_x(t, h) <= T;
    const V3 k1 = L.f;
    // normal-stream phases alternate per call (k2: even, k3: odd, ...); each field is fetched one call ahead
    if (THERMAL && !NSRC::kShared) { z2 = draw(true); z3 = draw(false); }
    const V3 k2 = fun(V3{y.x + (k1.x * A21) * h, y.y + (k1.y * A21) * h, y.z + (k1.z * A21) * h}, z2, true);
    const V3 z4 = draw(true);
    const V3 k3 = fun(V3{y.x + (k1.x * A31 + k2.x * A32) * h, y.y + (k1.y * A31 + k2.y * A32) * h,
                         y.z + (k1.z * A31 + k2.z * A32) * h}, z3, true);
    const V3 z5 = draw(false);
    const V3 k4 = fun(V3{y.x + (k1.x * A41 + k2.x * A42 + k3.x * A43) * h, y.y + (k1.y * A41 + k2.y * A42 + k3.y * A43) * h,
                         y.z + (k1.z * A41 + k2.z * A42 + k3.z * A43) * h}, z4, true);
    const V3 z6 = draw(true);
    const V3 k5 = fun(V3{y.x + (k1.x * A51 + k2.x * A52 + k3.x * A53 + k4.x * A54) * h,
                         y.y + (k1.y * A51 + k2.y * A52 + k3.y * A53 + k4.y * A54) * h,
                         y.z + (k1.z * A51 + k2.z * A52 + k3.z * A53 + k4.z * A54) * h}, z5, true);
    if (THERMAL) ns.peek();
    const V3 z7 = draw(false);
    const V3 k6 = fun(V3{y.x + (k1.x * A61 + k2.x * A62 + k3.x * A63 + k4.x * A64 + k5.x * A65) * h,
                         y.y + (k1.y * A61 + k2.y * A62 + k3.y * A63 + k4.y * A64 + k5.y * A65) * h,
                         y.z + (k1.z * A61 + k2.z * A62 + k3.z * A63 + k4.z * A64 + k5.z * A65) * h}, z6, on_end);
    const V3 y_new{y.x + h * (k1.x * B1 + k3.x * B3 + k4.x * B4 + k5.x * B5 + k6.x * B6),
                   y.y + h * (k1.y * B1 + k3.y * B3 + k4.y * B4 + k5.y * B5 + k6.y * B6),
                   y.z + h * (k1.z * B1 + k3.z * B3 + k4.z * B4 + k5.z * B5 + k6.z * B6)};
    const V3 f_new = fun(y_new, z7, on_end);
    const V3 ev{(k1.x * E1 + k3.x * E3 + k4.x * E4 + k5.x * E5 + k6.x * E6 + f_new.x * E7) * h,
                (k1.y * E1 + k3.y * E3 + k4.y * E4 + k5.y * E5 + k6.y * E6 + f_new.y * E7) * h,
                (k1.z * E1 + k3.z * E3 + k4.z * E4 + k5.z * E5 + k6.z * E6 + f_new.z * E7) * h};
    const V3 sc{atol + fmax_abs(y.x, y_new.x) * rtol, atol + fmax_abs(y.y, y_new.y) * rtol, atol + fmax_abs(y.z, y_new.z) * rtol};
    // error_norm = rms(ev / scale) (rk.py:104-109); the controller only needs err < 1 and err^-0.2, so the kernel
    // carries err^2 (no sqrt) and divides by reciprocal-multiply (v_rcp_f64 + one Newton step, ~1 ulp)
    const V3 q{ev.x * rcp_fast(sc.x), ev.y * rcp_fast(sc.y), ev.z * rcp_fast(sc.z)};
    const double err2 = dot(q, q) * (1.0 / 3.0);
    const double err = err2;      // compared against squared thresholds below
    // Controller (rk.py:158-181), branch-free: both outcomes share err^-0.2 and differ in a handful of selects.
    // 0.9 * err^-0.2 saturates at MAX_FACTOR = 10 for err <= 0.09^5 and at MIN_FACTOR = 0.2 for err >= 4.5^5 (err is
    // the SQUARED norm here); NaN error norms reject (nan < 1 is False) with fmax(0.2, NaN) = 0.2, as in SciPy.
    const bool acc = active && err < 1.0;
    const double r9 = 0.9 * inv_tenth_root(err);
    // (the clamps also cover the ends of the range: err -> 0 makes r9 huge, inf or -- at exactly 0 -- NaN, and
    // fmin/fmax return their other operand for a NaN; err -> inf makes it 0 or NaN)
    double fa = fmin(10.0, r9);
    fa = L.rejected ? fmin(1.0, fa) : fa;
    const double fr = fmax(0.2, r9);
    // h_abs of a lane that is through (or never started) is dead: nothing reads it before llgs_lane_begin writes it again
    const double h_next = h_try * (acc ? fa : fr);
    // an accepted attempt advances, records, and does the next step()'s prologue
    commit7(__ballot(acc), L.t, t_new, L.y.x, y_new.x, L.y.y, y_new.y, L.y.z, y_new.z, L.f.x, f_new.x, L.f.y, f_new.y, L.f.z, f_new.z);
    if (RECORD) { if (acc) llgs_lane_emit<RECORD>(L, out_m, rec, ek); } else L.npts += acc ? 1 : 0;
    L.rejected = active ? !acc : L.rejected;
    L.min_step = llgs_min_step_at(L.t);                                                  // unchanged t -> unchanged value
    // the next step()'s clamp of h_abs to [min_step, max_step] (rk.py:121-126) applies after an accepted attempt only; a rejected
    // one has shrunk h_try <= max_step, so the upper clamp is a no-op there and runs unconditionally
    L.h_abs = fmax(fmin(h_next, max_step), acc ? L.min_step : 0.0);
    L.active = active && (L.t != T);
}

// end of a solve: the final row (the input row when the solve failed), accepted points, attempts
template <bool RECORD, class NSRC>
__device__ __forceinline__ SolveOut llgs_lane_finish(LlgsLane& L, V3& out_m, const Recorder& rec, const LlgsEnergyK& ek, const NSRC& ns) {
    SolveOut o{L.m0, 0, 0, 0, false};
    if (!RECORD) { llgs_lane_emit<RECORD>(L, out_m, rec, ek); --L.npts; }
    o.m = out_m;
    o.n = L.npts - 1;
    o.work = L.attempts;
    o.ok = L.ok && !ns.broken;
    if (!L.ok) o.m = L.m0;
    return o;
}

// A7 (+A8 when RECORD): scipy solve_ivp(RK45) as LLGSSolver.solve drives it, one env per lane from start to end.
template <bool THERMAL, bool RECORD, bool AXIS_Z, class NSRC>
__device__ __forceinline__ SolveOut llgs_solve(const V3& m0, double J, double T, const LlgsK& k, double beta,
                                               double betap, double rtol, double atol, double max_step,
                                               int64_t max_attempts, const RngKey& rk, const Recorder& rec,
                                               const LlgsEnergyK& ek, NSRC& ns, bool enabled) {
    const Dp5Tab tb = make_dp5_tab();
    const V3 zero{0.0, 0.0, 0.0};
    LlgsLane L;
    V3 out_m = m0;
    llgs_lane_begin<THERMAL, RECORD, AXIS_Z>(L, out_m, m0, J, T, k, beta, betap, rtol, atol, max_step, rk, rec, ek, ns, enabled);
    // SharedNormals: the prologue's two RHS calls were chunk 0; every attempt is one further chunk and the loop is
    // wave-uniform (a finished lane idles until the wavefront's last lane is through)
    WaveProf prof;
    prof.start();
    bool wave_go = true;
    if (NSRC::kShared) wave_go = ns.chunk_end(L.active);
    if (wave_go)
    for (;;) {
      // The loop is wave-uniform and its body has no lane-divergent control flow (see llgs_lane_attempt).  (The
      // lane-divergent form -- break per lane, exec-masked body -- spent as long on its mask bookkeeping at the top of every
      // iteration as on one RHS.)
      // (shared source: the first two fields of the attempt are fetched before anything else, their LDS latency runs
      // under the step-size bookkeeping)
      V3 z2 = zero, z3 = zero;
      if (THERMAL && NSRC::kShared) { z2 = llgs_draw<THERMAL>(ns, k, true); z3 = llgs_draw<THERMAL>(ns, k, false); }
      llgs_lane_gate(L, max_attempts);
      if (!NSRC::kShared && __ballot(L.active) == 0ull) break;
      llgs_lane_attempt<THERMAL, RECORD, AXIS_Z>(L, out_m, k, tb, rtol, atol, max_step, rec, ek, ns, z2, z3);
      if (NSRC::kShared && !ns.chunk_end(L.active)) break;
    }
    prof.stop(L.attempts);
    return llgs_lane_finish<RECORD>(L, out_m, rec, ek, ns);
}

// A9: compute_resistance.  ref = normalised reference layer.
__device__ __forceinline__ double resistance(const V3& m_in, int dev_type, double r_p, double r_ap, double tmr,
                                             const V3& ref, double r_series) {
#pragma clang fp contract(off)
    if (dev_type == 0) {
        const double inv = rsqrt_fast(dot(m_in, m_in));        // validate_magnetization, base_device.py:112-116
        const V3 m{m_in.x * inv, m_in.y * inv, m_in.z * inv};
        const double r = r_p * (1.0 + tmr * (1.0 - dot(m, ref)) * 0.5);
        return fmax(r, r_p * 0.5);
    }
    double r = r_p + (r_ap - r_p) * (1.0 - dot(m_in, ref)) * 0.5;
    if (dev_type == 1) r = r + r_series;
    return fmax(r, 1.0);
}

// A10: SafetyWrapper.validate_action (in the action's own dtype) + _parse_action (fp64)
template <typename AT>
__device__ __forceinline__ void parse_action(AT a0, AT a1, double max_current, double max_duration, double& J, double& T) {
    const AT cmax = (AT)1e8, dmin = (AT)1e-12, dmax = (AT)1e-6;
    if (!isnan(a0)) a0 = a0 < -cmax ? -cmax : (a0 > cmax ? cmax : a0);
    if (!isnan(a1)) a1 = a1 < dmin ? dmin : (a1 > dmax ? dmax : a1);
    if (isnan(a0) || isnan(a1) || isinf(a0) || isinf(a1)) { a0 = (AT)0; a1 = dmin; }
    const double j = (double)a0, t = (double)a1;
    J = j < -max_current ? -max_current : (j > max_current ? max_current : j);
    T = t < 1e-12 ? 1e-12 : (t > max_duration ? max_duration : t);
}

__device__ __forceinline__ float obs_cast(double v) {
    const float f = (float)v;
    // SafetyWrapper.validate_observation: np.nan_to_num(nan=0, posinf=1e6, neginf=-1e6)
    return isnan(f) ? 0.0f : (isinf(f) ? (f > 0 ? 1e6f : -1e6f) : f);
}

}  // namespace stg
