/*
 * stg_oracle.c -- CPU restatement of the SpinTorque-v0 step path.  TEST INFRASTRUCTURE ONLY
 * (see stg_oracle.h for the rules and the parity status).  Reference paths are relative to
 * /root/reference/spin_torque_gym/.  Build: oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include "stg_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MU0 (4.0 * M_PI * 1e-7)          /* 4*np.pi*1e-7: simple_solver.py:60, llgs_solver.py:48 */
#define KB_SIMPLE 1.38e-23               /* simple_solver.py:380 */
#define KB_LLGS 1.380649e-23             /* llgs_solver.py:49, thermal_model.py:32 */

/* np.cross for 3-vectors: c0 = a1*b2 - a2*b1, ... */
static inline void cross3(const double a[3], const double b[3], double c[3]) {
    double c0 = a[1] * b[2] - a[2] * b[1];
    double c1 = a[2] * b[0] - a[0] * b[2];
    double c2 = a[0] * b[1] - a[1] * b[0];
    c[0] = c0; c[1] = c1; c[2] = c2;
}
/* np.dot / the sum of squares inside np.linalg.norm: left-to-right accumulation */
static inline double dot3(const double a[3], const double b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static inline double norm3(const double a[3]) { return sqrt(dot3(a, a)); }
static inline int finite3(const double a[3]) { return isfinite(a[0]) && isfinite(a[1]) && isfinite(a[2]); }

/* ------------------------------------------------------------------------------------------------
 * Thermal-field random numbers.  The reference draws np.random.normal(0,1,3) from the process-global
 * MT19937 per RHS call (simple_solver.py:384, llgs_solver.py:112) -- not reproducible per env, so only
 * the distribution can be matched.  The build keys a counter-based generator instead:
 * Philox4x32-10 (Salmon et al., SC'11), key = seed, counter = (env_id, env_step, rhs_call_index).
 * ------------------------------------------------------------------------------------------------ */
void stgo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Normal stream of one (env, env step): xoshiro128+ seeded by Philox4x32-10(key = seed, counter = (env_id, env_step,
 * tag)); consecutive outputs -> 23-bit uniforms -> fp32 Box-Muller pairs; RHS call j consumes normals 3j..3j+2 (even
 * calls draw two pairs and keep the 4th normal for the next, odd, call).  Same construction as the HIP kernels
 * (csrc/stg_physics.hpp: NormalStream), restated. */
typedef struct { uint32_t s[4]; float carry; uint32_t calls; } nstream;

static void ns_init(nstream* g, uint64_t seed, uint64_t env_id, uint32_t env_step, uint32_t tag) {
    uint32_t ctr[4] = {(uint32_t)env_id, (uint32_t)(env_id >> 32), env_step, tag};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    stgo_philox4x32_10(ctr, key, g->s);
    g->s[3] |= 1u;
    g->carry = 0.0f;
    g->calls = 0;
}
static uint32_t ns_next(nstream* g) {
    uint32_t* s = g->s;
    uint32_t result = s[0] + s[3];                 /* xoshiro128+ */
    uint32_t t = s[1] << 9;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t;
    s[3] = (s[3] << 11) | (s[3] >> 21);
    return result;
}
/* 23-bit uniform in (0,1): top 23 bits as the mantissa of a float in [1,2), minus (1 - 2^-24) */
static float ns_uniform(nstream* g) {
    union { uint32_t u; float f; } v;
    v.u = (ns_next(g) >> 9) | 0x3F800000u;
    return v.f - 0.99999994f;
}
static void ns_pair(nstream* g, float* a, float* b) {
    float u0 = ns_uniform(g);
    float u1 = ns_uniform(g);
    float r = sqrtf(-2.0f * logf(u0));
    const float two_pi = 6.28318530717958647692f;
    *a = r * cosf(two_pi * u1);
    *b = r * sinf(two_pi * u1);
}
static void ns_draw3(nstream* g, double z[3]) {
    float a, b, c, d;
    if ((g->calls & 1u) == 0) {
        ns_pair(g, &a, &b);
        ns_pair(g, &c, &d);
        g->carry = d;
        z[0] = a; z[1] = b; z[2] = c;
    } else {
        ns_pair(g, &a, &b);
        z[0] = g->carry; z[1] = a; z[2] = b;
    }
    g->calls++;
}

void stgo_thermal_normals(uint64_t seed, uint64_t env_id, uint32_t env_step, uint32_t call_idx, double z[3]) {
    nstream g;
    ns_init(&g, seed, env_id, env_step, 0u);
    for (uint32_t c = 0; c <= call_idx; ++c) ns_draw3(&g, z);
}

/* device-side reset draw of the kernels (csrc/spintorque_hip.hip: device_reset_draw), restated */
void stgo_reset_draw(uint64_t seed, uint64_t env_id, uint32_t rng_step, int n_targets, double z[3], int* target_idx) {
    nstream g;
    ns_init(&g, seed ^ 0x9E3779B97F4A7C15ull, env_id, rng_step, 0xFFFFFFFFu);
    ns_draw3(&g, z);
    uint32_t r = ns_next(&g);
    *target_idx = (int)(((uint64_t)r * (uint64_t)n_targets) >> 32);
}

double stgo_thermal_strength(const stgo_params* p, double gamma, double temperature, int which) {
    if (which == 0) {
        /* simple_solver.py:382-383: sqrt(2*alpha*kb*T / (mu_0*ms*volume*gamma)) */
        return sqrt(2 * p->damping * KB_SIMPLE * temperature / (MU0 * p->ms * p->volume * gamma));
    }
    /* llgs_solver.py:87-90 / thermal_model.py:68-73: sqrt(2*alpha*k_b*T / (gamma*mu_0*ms*volume)) */
    return sqrt(2 * p->damping * KB_LLGS * temperature / (gamma * MU0 * p->ms * p->volume));
}

/* ------------------------------------------------------------------------------------------------
 * A1 + A2  SimpleLLGSSolver._compute_effective_field / _compute_dmdt
 * ------------------------------------------------------------------------------------------------ */
void stgo_simple_dmdt(const double m[3], const stgo_params* p, double gamma, double J,
                      const double h_thermal[3], double out[3]) {
    /* simple_solver.py:318 / :361  easy_axis / np.linalg.norm(easy_axis) */
    double en = norm3(p->easy_axis);
    double e[3] = {p->easy_axis[0] / en, p->easy_axis[1] / en, p->easy_axis[2] / en};
    /* :370-371  h_k = 2*k_u/(mu_0*ms); h_anis = h_k*dot(m,e)*e */
    double hk = (2 * p->ku) / (MU0 * p->ms);
    double c = hk * dot3(m, e);
    double h_anis[3] = {c * e[0], c * e[1], c * e[2]};
    /* :375  h_demag = -ms*m[2]*[0,0,1] */
    double d = -p->ms * m[2];
    double h_demag[3] = {d * 0.0, d * 0.0, d * 1.0};
    /* :388  h_applied(=0) + h_anis + h_demag + h_thermal */
    double h[3];
    for (int i = 0; i < 3; ++i) h[i] = ((0.0 + h_anis[i]) + h_demag[i]) + (h_thermal ? h_thermal[i] : 0.0);
    /* :324-334  spin torque, p = easy axis */
    double tq[3] = {0.0, 0.0, 0.0};
    if (fabs(J) > 1e-12) {
        double mxp[3], mxmxp[3];
        cross3(m, e, mxp);
        cross3(m, mxp, mxmxp);
        double a = p->polarization * J / (p->ms * p->volume);
        for (int i = 0; i < 3; ++i) tq[i] = a * mxmxp[i];
    }
    /* :337-342 */
    double gamma_eff = gamma / (1 + p->damping * p->damping);
    double prec[3], mxprec[3];
    cross3(m, h, prec);
    cross3(m, prec, mxprec);
    for (int i = 0; i < 3; ++i) {
        double damp = p->damping * mxprec[i];
        out[i] = -gamma_eff * (prec[i] + damp) + tq[i];
    }
}

/* SOTMRAMDevice.compute_spin_torque (devices/sot_mram.py:163-194): tau_DL = f_dl J (sigma x m), tau_FL = f_fl J sigma */
void stgo_sot_torque(const double m[3], double J, const stgo_params* p, double tau_dl[3], double tau_fl[3]) {
    double sxm[3];
    cross3(p->sot_sigma, m, sxm);
    for (int i = 0; i < 3; ++i) {
        tau_dl[i] = p->sot_tau_dl * J * sxm[i];
        tau_fl[i] = p->sot_tau_fl * J * p->sot_sigma[i];
    }
}

/* VCMAMRAMDevice._compute_effective_anisotropy (devices/vcma_mram.py:122-147) */
double stgo_vcma_keff(double volt, const stgo_params* p) {
    double v = fmin(fmax(volt, -p->vcma_vbd), p->vcma_vbd);
    double change = -p->vcma_xi * fabs(v) / (p->vcma_td * p->vcma_td);
    double keff = p->ku + change;
    return fmax(keff, -0.5 * p->ku);
}

/* Right-hand side of the opt-in device-physics torque model (include/spintorque_hip.h: stg_config.torque_model = 1):
 * the reference RHS (stgo_simple_dmdt) with, per device type, SOT: no Slonczewski term, + (tau_DL + tau_FL)/(ms V);
 * VCMA: K_u replaced by K_eff(volt) while the pulse is on.  `on` = the pulse gate of this stage. */
static void device_dmdt(const double m[3], const stgo_params* p, double gamma, double J, double volt, int on,
                        const double h_thermal[3], double out[3]) {
    stgo_params q = *p;
    double Js = on ? J : 0.0;
    if (p->dev_type == 2 && on) q.ku = stgo_vcma_keff(volt, p);
    if (p->dev_type == 1) {
        stgo_simple_dmdt(m, &q, gamma, 0.0, h_thermal, out);
        double dl[3], fl[3];
        stgo_sot_torque(m, Js, p, dl, fl);
        double msv = p->ms * p->volume;
        for (int i = 0; i < 3; ++i) out[i] += (dl[i] + fl[i]) / msv;
    } else {
        stgo_simple_dmdt(m, &q, gamma, Js, h_thermal, out);
    }
}

/* SimpleLLGSSolver._validate_magnetization (simple_solver.py:208-229).  Returns 0 normal, 1 reset to
 * [0,0,1] (non-finite input, tiny norm or non-finite quotient). */
static int simple_validate(double m[3]) {
    if (!finite3(m)) { m[0] = 0; m[1] = 0; m[2] = 1; return 1; }
    double mag = norm3(m);
    if (mag < 1e-12) { m[0] = 0; m[1] = 0; m[2] = 1; return 1; }
    double q[3] = {m[0] / mag, m[1] / mag, m[2] / mag};
    if (!finite3(q)) { m[0] = 0; m[1] = 0; m[2] = 1; return 1; }
    m[0] = q[0]; m[1] = q[1]; m[2] = q[2];
    return 0;
}

/* validation.validate_magnetization as a predicate (utils/validation.py:24-59): raises when the
 * vector is non-finite or shorter than 1e-12. */
static int validation_rejects(const double m[3]) {
    if (!finite3(m)) return 1;
    if (norm3(m) < 1e-12) return 1;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * A3 + A4 + A5  RobustLLGSSolver.solve -> SimpleLLGSSolver.solve (rk4 / euler)
 * ------------------------------------------------------------------------------------------------ */
/* physics/thermal_model.py:113-137 */
void stgo_ou_update(double x[3], const double xi[3], double dt, double corr_time) {
    const double decay = corr_time > 0 ? exp(-dt / corr_time) : 0.0;       /* :120-123 */
    const double w = sqrt(1 - decay * decay);                              /* :129-132 */
    for (int j = 0; j < 3; ++j) x[j] = decay * x[j] + w * xi[j];
}

int stgo_simple_solve(const double m0[3], double T, const stgo_params* p, const stgo_config* c,
                      double J, uint64_t env_id, uint32_t env_step,
                      double m_final[3], int32_t* n_steps, int32_t* first_zero_row, int32_t* n_reset,
                      double* traj, int64_t traj_cap_rows) {
    if (n_steps) *n_steps = 0;
    if (first_zero_row) *first_zero_row = -1;
    if (n_reset) *n_reset = 0;
    m_final[0] = m0[0]; m_final[1] = m0[1]; m_final[2] = m0[2];
    /* robust_solver.py:152-190 _validate_inputs: any failure -> except Exception -> fallback (:140-150) */
    if (validation_rejects(m0)) return 0;
    if (!(T > 0.0)) return 0;                      /* t_end <= t_start */
    if (!p->params_valid) return 0;                /* validate_parameters(device_params) as 'stt_mram' */
    if (!(c->temperature > 0.0)) return 0;         /* "Temperature must be positive number" */

    double m[3] = {m0[0], m0[1], m0[2]};
    int resets = simple_validate(m);               /* simple_solver.py:119 */
    /* simple_solver.py:137-139 */
    double dt = fmin(c->max_step, (T - 0.0) / 100);
    double q = (T - 0.0) / dt;
    int n = (int)q; if (n < 10) n = 10;
    dt = (T - 0.0) / n;
    if (n_steps) *n_steps = n;
    const int thermal = c->thermal && c->temperature > 0;
    const double hs = thermal ? stgo_thermal_strength(p, c->gamma, c->temperature, 0) : 0.0;
    int zero_row = -1;
    /* device-physics model: the voltage the env itself computes for the pulse, V = J R(m_before) A
     * (spin_torque_env.py:475-477), sets K_eff for VCMA classes */
    const double volt = c->torque_model ? J * stgo_resistance(m0, p) * p->area : 0.0;
    nstream g;
    if (thermal) ns_init(&g, c->seed, env_id, env_step, 0u);
    if (traj && traj_cap_rows > 0) { traj[0] = m[0]; traj[1] = m[1]; traj[2] = m[2]; }

    /* noise_model 1: ThermalFluctuations' correlated field, one update per sub-step (x = 0 when the pulse starts),
     * held for all stages of the sub-step */
    const int ou = thermal && c->noise_model == 1;
    double oux[3] = {0.0, 0.0, 0.0};
    for (int i = 0; i < n; ++i) {
        double t = (double)i * dt + 0.0;           /* np.linspace(0, T, n+1)[i] = i*step + start */
        double k[4][3], y[3], hth[3], z[3];
        const int nstage = (c->solver == 1) ? 1 : 4;
        if (ou) {
            ns_draw3(&g, z);
            stgo_ou_update(oux, z, dt, c->noise_corr_time);
            hth[0] = hs * oux[0]; hth[1] = hs * oux[1]; hth[2] = hs * oux[2];
        }
        for (int s = 0; s < nstage; ++s) {
            double ts;
            if (s == 0) { ts = t; y[0] = m[0]; y[1] = m[1]; y[2] = m[2]; }
            else if (s < 3) { ts = t + dt / 2; for (int j = 0; j < 3; ++j) y[j] = m[j] + k[s - 1][j] / 2; }
            else { ts = t + dt; for (int j = 0; j < 3; ++j) y[j] = m[j] + k[2][j]; }
            /* spin_torque_env.py:442-443  current_func(t) = J if t <= T else 0 */
            double Jt = (ts <= T) ? J : 0.0;
            if (thermal && !ou) {
                ns_draw3(&g, z);
                hth[0] = hs * z[0]; hth[1] = hs * z[1]; hth[2] = hs * z[2];
            }
            double f[3];
            if (c->torque_model) device_dmdt(y, p, c->gamma, J, volt, ts <= T, thermal ? hth : 0, f);
            else stgo_simple_dmdt(y, p, c->gamma, Jt, thermal ? hth : 0, f);
            for (int j = 0; j < 3; ++j) k[s][j] = dt * f[j];
        }
        double mn[3];
        if (nstage == 1) {
            /* simple_solver.py:275-276  m + dt*dmdt */
            for (int j = 0; j < 3; ++j) mn[j] = m[j] + k[0][j];
        } else {
            /* simple_solver.py:295  m + (k1 + 2*k2 + 2*k3 + k4)/6 */
            for (int j = 0; j < 3; ++j) mn[j] = m[j] + (((k[0][j] + 2 * k[1][j]) + 2 * k[2][j]) + k[3][j]) / 6;
        }
        resets += simple_validate(mn);             /* simple_solver.py:168 */
        m[0] = mn[0]; m[1] = mn[1]; m[2] = mn[2];
        /* robust_solver.py:192-205: a row shorter than 1e-12 (the m/inf = 0 row of SURVEY H3) makes
         * _validate_output raise validation.ValidationError, which is not the class it catches. */
        if (zero_row < 0 && validation_rejects(m)) zero_row = i + 1;
        if (traj && (int64_t)(i + 1) < traj_cap_rows) { traj[3 * (i + 1)] = m[0]; traj[3 * (i + 1) + 1] = m[1]; traj[3 * (i + 1) + 2] = m[2]; }
    }
    if (first_zero_row) *first_zero_row = zero_row;
    if (n_reset) *n_reset = resets;
    if (zero_row >= 0) return 0;
    m_final[0] = m[0]; m_final[1] = m[1]; m_final[2] = m[2];
    return 1;
}

/* ------------------------------------------------------------------------------------------------
 * A6  LLGSSolver.solve::llgs_rhs
 * ------------------------------------------------------------------------------------------------ */
void stgo_llgs_rhs(const double y[3], const stgo_params* p, double gamma, double J,
                   const double h_thermal[3], double out[3]) {
    /* llgs_solver.py:94-101 */
    double m[3];
    double mn = norm3(y);
    if (mn > 1e-12) { m[0] = y[0] / mn; m[1] = y[1] / mn; m[2] = y[2] / mn; }
    else { m[0] = 0; m[1] = 0; m[2] = 1; }
    /* :182-211 effective field; easy axis is NOT normalised here */
    double c = (2 * p->ku / (MU0 * p->ms)) * dot3(m, p->easy_axis);
    double h[3];
    for (int i = 0; i < 3; ++i) {
        double hi = 0.0;                                   /* h_applied.copy() = zeros */
        hi += c * p->easy_axis[i];                         /* h_anis */
        hi += (-p->ms * p->demag[i]) * m[i];               /* -ms*demag_factors*m */
        if (p->a_ex > 0) hi += ((2 * p->a_ex / (MU0 * p->ms)) * 0.1) * m[i];   /* :205-209 */
        if (h_thermal) hi += h_thermal[i];                 /* :111-113 */
        h[i] = hi;
    }
    /* :213-237 torques, p_hat = z */
    double tstt[3] = {0, 0, 0}, tfl[3] = {0, 0, 0};
    if (!(fabs(J) < 1e-12)) {
        const double z[3] = {0, 0, 1};
        double beta = p->polarization * gamma / (2 * p->ms * p->volume);
        double betap = 0.1 * beta;
        double mxp[3], mxmxp[3];
        cross3(m, z, mxp);
        cross3(m, mxp, mxmxp);
        for (int i = 0; i < 3; ++i) { tstt[i] = beta * J * mxmxp[i]; tfl[i] = betap * J * mxp[i]; }
    }
    /* :121-124 */
    double mxh[3], dm[3], mxdm[3];
    cross3(m, h, mxh);
    for (int i = 0; i < 3; ++i) dm[i] = -gamma * mxh[i];
    cross3(m, dm, mxdm);
    for (int i = 0; i < 3; ++i) dm[i] += p->damping * mxdm[i];
    for (int i = 0; i < 3; ++i) out[i] = dm[i] + (tstt[i] + tfl[i]);
}

/* llgs_solver.py:239-262 */
static double llgs_energy(const double m[3], const stgo_params* p) {
    const double happ[3] = {0, 0, 0};
    double e_zeeman = -MU0 * p->ms * p->volume * dot3(m, happ);
    double ct = dot3(m, p->easy_axis);
    double e_anis = -p->ku * p->volume * (ct * ct);
    double s = (p->demag[0] * (m[0] * m[0]) + p->demag[1] * (m[1] * m[1])) + p->demag[2] * (m[2] * m[2]);
    double e_demag = 0.5 * MU0 * (p->ms * p->ms) * p->volume * s;
    return e_zeeman + e_anis + e_demag;
}

/* |tau_stt| + |tau_fl| of llgs_solver.py:168-172 */
static double llgs_torque_norms(const double m[3], const stgo_params* p, double gamma, double J) {
    if (fabs(J) < 1e-12) return 0.0;
    const double z[3] = {0, 0, 1};
    double beta = p->polarization * gamma / (2 * p->ms * p->volume);
    double betap = 0.1 * beta;
    double mxp[3], mxmxp[3], a[3], b[3];
    cross3(m, z, mxp);
    cross3(m, mxp, mxmxp);
    for (int i = 0; i < 3; ++i) { a[i] = beta * J * mxmxp[i]; b[i] = betap * J * mxp[i]; }
    return norm3(a) + norm3(b);
}

/* scipy/integrate/_ivp/common.py:63-65  norm(x) = np.linalg.norm(x) / x.size**0.5 */
static inline double rms3(const double x[3]) { return norm3(x) / sqrt(3.0); }

/* Dormand-Prince 5(4) tableau, scipy/integrate/_ivp/rk.py:380-391 */
static const double DP_C[6] = {0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1};
static const double DP_A[6][5] = {
    {0, 0, 0, 0, 0},
    {1.0 / 5, 0, 0, 0, 0},
    {3.0 / 40, 9.0 / 40, 0, 0, 0},
    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
static const double DP_B[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
static const double DP_E[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525, 1.0 / 40};

typedef struct {
    const stgo_params* p; const stgo_config* c; double J, T, hs; int thermal;
    nstream g;
} rhs_ctx;

static void llgs_fun(rhs_ctx* x, double t, const double y[3], double out[3]) {
    double Jt = (t <= x->T) ? x->J : 0.0;           /* spin_torque_env.py:442-443 */
    double hth[3], z[3];
    if (x->thermal) {
        ns_draw3(&x->g, z);
        hth[0] = x->hs * z[0]; hth[1] = x->hs * z[1]; hth[2] = x->hs * z[2];
    }
    stgo_llgs_rhs(y, x->p, x->c->gamma, Jt, x->thermal ? hth : 0, out);
}

/* ------------------------------------------------------------------------------------------------
 * A7 + A8  LLGSSolver.solve: scipy solve_ivp(method='RK45', rtol, atol, max_step) then the epilogue
 * ------------------------------------------------------------------------------------------------ */
int64_t stgo_llgs_solve(const double m0[3], double T, const stgo_params* p, const stgo_config* c,
                        double J, uint64_t env_id, uint32_t env_step,
                        double m_final[3], int32_t* success, int64_t* n_attempts,
                        double* t_out, double* m_out, double* e_out, double* tq_out, int64_t cap) {
    rhs_ctx x;
    x.p = p; x.c = c; x.J = J; x.T = T; x.hs = 0.0; x.thermal = c->thermal ? 1 : 0;
    if (x.thermal) {
        x.hs = stgo_thermal_strength(p, c->gamma, c->temperature, 1);               /* llgs_solver.py:85-90 */
        ns_init(&x.g, c->seed, env_id, env_step, 0u);
    }
    const double rtol = c->rtol, atol = c->atol, max_step = c->max_step;
    double y[3], f[3];
    double n0 = norm3(m0);                            /* llgs_solver.py:76 */
    for (int i = 0; i < 3; ++i) y[i] = m0[i] / n0;
    double t = 0.0;
    const double t_bound = T;
    int64_t npts = 0, attempts = 0;
#define EMIT_POINT()                                                                              \
    do {                                                                                          \
        double nn = norm3(y);                                                                     \
        double mr[3] = {y[0] / nn, y[1] / nn, y[2] / nn};           /* llgs_solver.py:152-153 */   \
        if (npts < cap) {                                                                         \
            if (t_out) t_out[npts] = t;                                                           \
            if (m_out) { m_out[3 * npts] = mr[0]; m_out[3 * npts + 1] = mr[1]; m_out[3 * npts + 2] = mr[2]; } \
            if (e_out) e_out[npts] = llgs_energy(mr, p);                                          \
            if (tq_out) tq_out[npts] = llgs_torque_norms(mr, p, c->gamma, (t <= T) ? J : 0.0);    \
        }                                                                                         \
        m_final[0] = mr[0]; m_final[1] = mr[1]; m_final[2] = mr[2];                               \
        ++npts;                                                                                   \
    } while (0)
    EMIT_POINT();
    /* RungeKutta.__init__ (rk.py:93-103): f0, then select_initial_step (common.py:68-134) */
    llgs_fun(&x, t, y, f);
    double h_abs;
    {
        double interval = fabs(t_bound - t);
        double scale[3], a[3], b[3];
        for (int i = 0; i < 3; ++i) { scale[i] = atol + fabs(y[i]) * rtol; a[i] = y[i] / scale[i]; b[i] = f[i] / scale[i]; }
        double d0 = rms3(a), d1 = rms3(b), h0;
        if (d0 < 1e-5 || d1 < 1e-5) h0 = 1e-6; else h0 = 0.01 * d0 / d1;
        h0 = fmin(h0, interval);
        double y1[3], f1[3], dd[3];
        for (int i = 0; i < 3; ++i) y1[i] = y[i] + h0 * 1.0 * f[i];
        llgs_fun(&x, t + h0 * 1.0, y1, f1);
        for (int i = 0; i < 3; ++i) dd[i] = (f1[i] - f[i]) / scale[i];
        double d2 = rms3(dd) / h0, h1;
        if (d1 <= 1e-15 && d2 <= 1e-15) h1 = fmax(1e-6, h0 * 1e-3);
        else h1 = pow(0.01 / fmax(d1, d2), 1.0 / (4 + 1));
        h_abs = fmin(fmin(100 * h0, h1), fmin(interval, max_step));
    }
    int ok = 1;
    /* solve_ivp loop (ivp.py:654-661) over OdeSolver.step (base.py:175-206) / _step_impl (rk.py:111-181) */
    while (t != t_bound) {
        double min_step = 10 * fabs(nextafter(t, INFINITY) - t);
        if (h_abs > max_step) h_abs = max_step; else if (h_abs < min_step) h_abs = min_step;
        int accepted = 0, rejected = 0;
        double K[7][3], y_new[3], t_new = t;
        while (!accepted) {
            if (h_abs < min_step || attempts >= c->max_attempts) { ok = 0; break; }
            ++attempts;
            double h = h_abs * 1.0;
            t_new = t + h;
            if (1.0 * (t_new - t_bound) > 0) t_new = t_bound;
            h = t_new - t;
            h_abs = fabs(h);
            /* rk_step (rk.py:14-70) */
            for (int i = 0; i < 3; ++i) K[0][i] = f[i];
            for (int s = 1; s < 6; ++s) {
                double ys[3];
                for (int i = 0; i < 3; ++i) {
                    double acc = 0.0;                              /* np.dot(K[:s].T, a[:s]) */
                    for (int j = 0; j < s; ++j) acc += K[j][i] * DP_A[s][j];
                    ys[i] = y[i] + acc * h;
                }
                llgs_fun(&x, t + DP_C[s] * h, ys, K[s]);
            }
            for (int i = 0; i < 3; ++i) {
                double acc = 0.0;                                  /* np.dot(K[:-1].T, B) */
                for (int j = 0; j < 6; ++j) acc += K[j][i] * DP_B[j];
                y_new[i] = y[i] + h * acc;
            }
            llgs_fun(&x, t + h, y_new, K[6]);
            double en[3];
            for (int i = 0; i < 3; ++i) {
                double scale = atol + fmax(fabs(y[i]), fabs(y_new[i])) * rtol;
                double acc = 0.0;                                  /* np.dot(K.T, E) * h */
                for (int j = 0; j < 7; ++j) acc += K[j][i] * DP_E[j];
                en[i] = (acc * h) / scale;
            }
            double err = rms3(en);
            if (err < 1) {
                double factor;
                if (err == 0) factor = 10; else factor = fmin(10, 0.9 * pow(err, -0.2));
                if (rejected) factor = fmin(1, factor);
                h_abs *= factor;
                accepted = 1;
            } else {
                /* NaN error norms land here too (nan < 1 is False), as in SciPy */
                h_abs *= fmax(0.2, 0.9 * pow(err, -0.2));
                rejected = 1;
            }
        }
        if (!ok) break;
        t = t_new;
        for (int i = 0; i < 3; ++i) { y[i] = y_new[i]; f[i] = K[6][i]; }
        EMIT_POINT();
    }
#undef EMIT_POINT
    if (success) *success = ok;
    if (n_attempts) *n_attempts = attempts;
    return npts;
}

/* ------------------------------------------------------------------------------------------------
 * A9  compute_resistance
 * ------------------------------------------------------------------------------------------------ */
double stgo_resistance(const double m_in[3], const stgo_params* p) {
    double rn = norm3(p->ref_m);
    double ref[3] = {p->ref_m[0] / rn, p->ref_m[1] / rn, p->ref_m[2] / rn};
    if (p->dev_type == 0) {
        /* devices/stt_mram.py:78-94; m re-normalised by validate_magnetization (base_device.py:94-116) */
        double mn = norm3(m_in);
        double m[3] = {m_in[0] / mn, m_in[1] / mn, m_in[2] / mn};
        double tmr = (p->r_ap - p->r_p) / p->r_p;
        double ct = dot3(m, ref);
        double r = p->r_p * (1 + tmr * (1 - ct) / 2);
        return fmax(r, p->r_p * 0.5);
    }
    double ct = dot3(m_in, ref);
    double r = p->r_p + (p->r_ap - p->r_p) * (1 - ct) / 2;
    if (p->dev_type == 1) {
        /* devices/sot_mram.py:196-228: r_mtj + r_hm*0.1, floor 1 */
        r = r + p->r_series;
    }
    /* devices/vcma_mram.py:236-257: floor 1 */
    return fmax(r, 1.0);
}

/* ------------------------------------------------------------------------------------------------
 * A10  SafetyWrapper.validate_action (float32 arithmetic) + _parse_action (float64)
 * ------------------------------------------------------------------------------------------------ */
void stgo_parse_action(const float action[2], const stgo_config* c, double* J, double* T) {
    /* utils/monitoring.py:304-313: np.clip on the float32 array elements, then the NaN/Inf test */
    float a0 = action[0], a1 = action[1];
    const float cmax = (float)1e8, dmin = (float)1e-12, dmax = (float)1e-6;
    if (!isnan(a0)) a0 = fminf(fmaxf(a0, -cmax), cmax);
    if (!isnan(a1)) a1 = fminf(fmaxf(a1, dmin), dmax);
    if (isnan(a0) || isnan(a1) || isinf(a0) || isinf(a1)) { a0 = 0.0f; a1 = dmin; }
    /* envs/spin_torque_env.py:417-431 */
    double j = (double)a0, t = (double)a1;
    j = fmin(fmax(j, -c->max_current), c->max_current);
    t = fmin(fmax(t, 1e-12), c->max_duration);
    *J = j; *T = t;
}

/* ------------------------------------------------------------------------------------------------
 * A12  _get_observation (vector mode)
 * ------------------------------------------------------------------------------------------------ */
void stgo_observation(const stgo_env_state* s, const stgo_params* p, const stgo_config* c, float obs[12]) {
    double r = stgo_resistance(s->m, p);
    double v[12];
    v[0] = s->m[0]; v[1] = s->m[1]; v[2] = s->m[2];
    v[3] = s->target[0]; v[4] = s->target[1]; v[5] = s->target[2];
    v[6] = r / p->r_p;
    v[7] = c->temperature / 300.0;
    v[8] = (double)(c->max_steps - s->step_count) / (double)c->max_steps;
    v[9] = s->total_energy / 1e-12;
    v[10] = s->last_action[0] / c->max_current;
    v[11] = s->last_action[1] / c->max_duration;
    for (int i = 0; i < 12; ++i) {
        float f = (float)v[i];
        /* SafetyWrapper.validate_observation (utils/monitoring.py:317-330): np.nan_to_num */
        if (isnan(f)) f = 0.0f; else if (isinf(f)) f = f > 0 ? 1e6f : -1e6f;
        obs[i] = f;
    }
}

/* ------------------------------------------------------------------------------------------------
 * A10-A14  SpinTorqueEnv.step
 * ------------------------------------------------------------------------------------------------ */
void stgo_env_step(stgo_env_state* s, const float action[2], const stgo_params* p, const stgo_config* c,
                   uint64_t env_id, stgo_step_out* out) {
    double J, T;
    stgo_parse_action(action, c, &J, &T);
    s->last_action[0] = J; s->last_action[1] = T;
    double prev_align = dot3(s->m, s->target);                      /* spin_torque_env.py:338-339 */
    /* _simulate_dynamics (spin_torque_env.py:435-488) */
    double mf[3];
    int ok;
    int32_t nsub = 0, zr = -1, nreset = 0;
    if (c->solver == 2) {
        int32_t succ = 0; int64_t att = 0;
        int64_t npts = stgo_llgs_solve(s->m, T, p, c, J, env_id, s->rng_step, mf, &succ, &att, 0, 0, 0, 0, 0);
        ok = succ; nsub = (int32_t)(npts - 1);
    } else {
        ok = stgo_simple_solve(s->m, T, p, c, J, env_id, s->rng_step, mf, &nsub, &zr, &nreset, 0, 0);
    }
    double m_before[3] = {s->m[0], s->m[1], s->m[2]};
    if (ok) {
        double nn = norm3(mf);                                      /* :462-464 */
        s->m[0] = mf[0] / nn; s->m[1] = mf[1] / nn; s->m[2] = mf[2] / nn;
    }
    double energy = 0.0;
    if (fabs(J) > 1e-12) {                                          /* :474-480 */
        double r = stgo_resistance(m_before, p);
        double v = J * r * p->area;
        energy = (v * v) / r * T;
    }
    s->total_energy += energy;
    s->step_count += 1;
    s->rng_step += 1;
    double align = dot3(s->m, s->target);                           /* :350-353 */
    double improve = align - prev_align;
    int is_success = align >= c->success_threshold;
    stgo_observation(s, p, c, out->obs);
    /* default reward (spin_torque_env.py:184-207) through CompositeReward.compute
     * (rewards/composite_reward.py:65-126): total = sum of weight*component in dict order */
    double total = 0.0;
    total += 10.0 * (is_success ? 10.0 : 0.0);
    total += (-c->energy_penalty_weight) * (-energy / 1e-12);
    total += 1.0 * improve;
    total += -2.0 * 0.0;
    /* SafetyWrapper.validate_reward (utils/monitoring.py:332-348) */
    if (isnan(total) || isinf(total)) total = -1.0;
    total = fmin(fmax(total, -1e6), 1e6);
    out->reward = total;
    out->terminated = (uint8_t)is_success;
    out->truncated = (uint8_t)(s->step_count >= c->max_steps);
    out->status = ok ? (nreset > 0 ? 2 : 0) : 1;
    out->energy = energy;
    out->n_sub = nsub;
}

void stgo_env_step_batch(int64_t n, stgo_env_state* s, const float* actions, const stgo_params* params,
                         const uint8_t* cls, const stgo_config* c, uint64_t env_id0, stgo_step_out* out,
                         int n_threads) {
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads)
#endif
    for (int64_t i = 0; i < n; ++i) {
        const stgo_params* p = &params[cls ? cls[i] : 0];
        stgo_env_step(&s[i], &actions[2 * i], p, c, env_id0 + (uint64_t)i, &out[i]);
    }
}

/* ------------------------------------------------------------------------------------------------
 * SpinTorqueArray-v0
 * ------------------------------------------------------------------------------------------------ */
void stgo_array_coupling(int rows, int cols, int type, double strength, double* out) {
    const int n = rows * cols;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double v = 0.0;
            if (i != j) {
                /* array_env.py:311-332 */
                int ir = i / cols, ic = i % cols, jr = j / cols, jc = j % cols;
                double dist = sqrt((double)((ir - jr) * (ir - jr) + (ic - jc) * (ic - jc)));
                if (type == 0 && dist > 0) v = strength / (dist * dist * dist);
                else if (type == 1 && dist == 1) v = strength;
                else if (type == 2 && dist > 0) v = strength / (dist * dist);
            }
            out[i * n + j] = v;
        }
}

void stgo_device_field(const double m_in[3], const stgo_params* p, double h[3]) {
    const double hk = 2 * p->ku / (MU0 * p->ms);
    if (p->dev_type == 0) {
        /* stt_mram.py:55-76: m re-normalised, raw easy axis, anisotropy only */
        double mn = norm3(m_in);
        double m[3] = {m_in[0] / mn, m_in[1] / mn, m_in[2] / mn};
        double c = hk * dot3(m, p->easy_axis);
        for (int i = 0; i < 3; ++i) h[i] = 0.0 + c * p->easy_axis[i];
        return;
    }
    /* sot_mram.py:79-112 / vcma_mram.py:85-120: applied(0) + h_anis + h_exchange(0) + h_demag + h_thermal(0) (+ h_dmi(0));
     * K_eff(0 V) = K for VCMA */
    double c = hk * dot3(m_in, p->easy_axis);
    for (int i = 0; i < 3; ++i) {
        double v = 0.0 + c * p->easy_axis[i];
        v = v + 0.0;
        v = v + (-p->ms * p->shape_demag[i]) * m_in[i];
        h[i] = v + 0.0;
    }
}

/* numpy's pairwise summation for n < 128 (np.mean / np.std over the device list): 8 interleaved accumulators */
static double np_sum(const double* a, int n) {
    if (n < 8) { double s = 0.0; for (int i = 0; i < n; ++i) s += a[i]; return s; }
    double r[8];
    for (int k = 0; k < 8; ++k) r[k] = a[k];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) for (int k = 0; k < 8; ++k) r[k] += a[i + k];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

static double array_similarity(const stgo_array_config* c, const double* pattern, const double* target) {
    const int n = c->rows * c->cols;
    double d[256];
    for (int i = 0; i < n; ++i) d[i] = dot3(&pattern[3 * i], &target[3 * i]);       /* array_env.py:523-531 */
    return np_sum(d, n) / n;
}

void stgo_array_observation(const stgo_array_config* c, const double* pattern, const double* target, double total_energy,
                            int32_t step_count, float* obs) {
    const int n = c->rows * c->cols;
    if (c->obs_mode == 0) {                                  /* array_env.py:535-541: [R,C,6] */
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < 3; ++k) { obs[6 * i + k] = (float)pattern[3 * i + k]; obs[6 * i + 3 + k] = (float)target[3 * i + k]; }
        return;
    }
    for (int i = 0; i < 3 * n; ++i) { obs[i] = (float)pattern[i]; obs[3 * n + i] = (float)target[i]; }   /* :543-557 */
    obs[6 * n + 0] = (float)array_similarity(c, pattern, target);
    obs[6 * n + 1] = (float)((double)(c->max_steps - step_count) / (double)c->max_steps);
    obs[6 * n + 2] = (float)(total_energy / 1e-12);
    obs[6 * n + 3] = (float)(c->temperature / 300.0);
}

void stgo_array_step(const stgo_array_config* c, const stgo_params* p, const double* coupling, double* pattern,
                     const double* target, double* total_energy, int32_t* step_count, const float* action, int n_action,
                     float* obs, double* reward, uint8_t* terminated, uint8_t* truncated, double* energy_out) {
    const int n = c->rows * c->cols;
    const double prev_sim = array_similarity(c, pattern, target);                     /* array_env.py:372-373 */
    /* _apply_action (array_env.py:411-476).  Note: in 'global' mode the action is [current, duration], so action[1]
     * (the duration) is what the reference reads as the current density and the duration defaults to 1 ns. */
    double J = n_action > 1 ? (double)action[1] : 0.0;
    double T = n_action > 2 ? (double)action[2] : 1e-9;
    J = fmin(fmax(J, -c->max_current), c->max_current);
    T = fmin(fmax(T, 1e-12), c->max_duration);
    int first = 0, count = 0, stride = 1;
    if (c->action_mode == 3) { first = 0; count = n; stride = 1; }
    else {
        double lim = c->action_mode == 0 ? n - 1 : (c->action_mode == 1 ? c->rows - 1 : c->cols - 1);
        int idx = (int)fmin(fmax((double)action[0], 0.0), lim);                      /* int(np.clip(action[0], 0, lim)) */
        if (c->action_mode == 0) { first = idx; count = 1; stride = 1; }
        else if (c->action_mode == 1) { first = idx * c->cols; count = c->cols; stride = 1; }
        else { first = idx; count = c->rows; stride = c->cols; }
    }
    double e_total = 0.0;
    for (int q = 0; q < count; ++q) {
        const int d = first + q * stride;
        double* m0 = &pattern[3 * d];
        /* _compute_effective_field (array_env.py:478-494): intrinsic + coupling with the pattern as updated so far */
        double h[3], hc[3] = {0, 0, 0};
        stgo_device_field(m0, p, h);
        if (c->include_coupling)
            for (int j = 0; j < n; ++j)
                if (j != d) for (int k = 0; k < 3; ++k) hc[k] += coupling[d * n + j] * pattern[3 * j + k];
        for (int k = 0; k < 3; ++k) h[k] = h[k] + hc[k];
        /* _simulate_device_dynamics (array_env.py:496-521): one derivative at m0, ten normalised Euler sub-steps */
        if (fabs(J) > 1e-12) {
            const double z[3] = {0, 0, 1};
            double mxp[3], tau[3], mxh[3], dm[3], mxdm[3];
            cross3(m0, z, mxp);
            cross3(m0, mxp, tau);
            for (int k = 0; k < 3; ++k) tau[k] = 0.1 * J * tau[k];
            cross3(m0, h, mxh);
            for (int k = 0; k < 3; ++k) dm[k] = -2.21e5 * mxh[k];
            cross3(m0, dm, mxdm);
            for (int k = 0; k < 3; ++k) dm[k] += 0.01 * mxdm[k];
            for (int k = 0; k < 3; ++k) dm[k] += tau[k];
            const double dt = T / 10;
            double m[3] = {m0[0], m0[1], m0[2]};
            for (int it = 0; it < 10; ++it) {
                for (int k = 0; k < 3; ++k) m[k] += dm[k] * dt;
                double nn = norm3(m);
                for (int k = 0; k < 3; ++k) m[k] = m[k] / nn;
            }
            m0[0] = m[0]; m0[1] = m[1]; m0[2] = m[2];
            /* energy with the resistance of the UPDATED state (current_m is a view of the pattern, :455-463) */
            double r = stgo_resistance(m0, p);
            double v = J * r * p->area;
            e_total += (v * v) / r * T;
        }
    }
    *total_energy += e_total;
    *step_count += 1;
    const double sim = array_similarity(c, pattern, target);
    const int is_success = sim >= c->success_threshold;
    stgo_array_observation(c, pattern, target, *total_energy, *step_count, obs);
    /* default reward (array_env.py:183-224) through CompositeReward */
    double mag[256];
    for (int i = 0; i < n; ++i) mag[i] = norm3(&pattern[3 * i]);
    double mean = np_sum(mag, n) / n, dev2[256];
    for (int i = 0; i < n; ++i) dev2[i] = (mag[i] - mean) * (mag[i] - mean);
    double uniformity = 1.0 - sqrt(np_sum(dev2, n) / n);
    if (uniformity < 0) uniformity = 0;
    double total = 0.0;
    total += 10.0 * (is_success ? 10.0 : sim * 5.0);
    total += (-c->energy_penalty_weight) * (-e_total / 1e-12);
    total += 1.0 * (sim - prev_sim);
    total += 2.0 * uniformity;
    *reward = total;
    *terminated = (uint8_t)is_success;
    *truncated = (uint8_t)(*step_count >= c->max_steps);
    if (energy_out) *energy_out = e_total;
}

int stgo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
