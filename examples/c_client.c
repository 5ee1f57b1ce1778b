/*
 * c_client.c -- a plain C99 caller of libspintorque_hip.so: no Python, no PyTorch, no HIP headers.
 *
 * Shows (and, through tests/test_c_client.py, checks) that the drop-in boundary is the C-ABI of include/spintorque_hip.h:
 * device memory from the HIP runtime's C entry points, a stg_config / stg_device_params filled by hand with the values the
 * reference's constructors use (SpinTorqueEnv.__init__, spin_torque_env.py:36-53; DeviceFactory defaults for 'stt_mram',
 * device_factory.py:129-145, with the volume of the switching regime of SURVEY G2), then
 *     stg_create -> stg_set_params -> stg_reset(initial_state, target_state) -> K x stg_step -> stg_get_state
 * i.e. SpinTorqueEnv.reset(options=...) and K env.step(action) calls for N envs at once.  Inputs and outputs are written to
 * a file; the test replays the same inputs through the CPU oracle and compares.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_client.c -o examples/_build/c_client -L<dir of libspintorque_hip.so> -lspintorque_hip \
 *       -L/opt/rocm/lib -lamdhip64 -lm          (see __graft_entry__.build)
 *   c_client <rk4|rk45> <n_envs> <steps> <out.bin>
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spintorque_hip.h"

/* the four HIP runtime entry points a caller needs for device memory (hip_runtime_api.h; hipError_t is an int enum,
 * hipMemcpyKind: 1 = host to device, 2 = device to host) */
extern int hipMalloc(void** ptr, size_t size);
extern int hipFree(void* ptr);
extern int hipMemcpy(void* dst, const void* src, size_t size, int kind);
extern int hipDeviceSynchronize(void);

#define CHECK_STG(call) do { int rc_ = (call); if (rc_ != STG_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, stg_last_error()); return 2; } } while (0)
#define CHECK_HIP(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s -> hipError %d\n", #call, rc_); return 3; } } while (0)

/* a small deterministic generator for the inputs (SplitMix64) */
static uint64_t sm_state;
static uint64_t sm_next(void) {
    uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double sm_uniform(void) { return (double)(sm_next() >> 11) * (1.0 / 9007199254740992.0); }

int main(int argc, char** argv) {
    if (argc != 5) { fprintf(stderr, "usage: %s <rk4|rk45> <n_envs> <steps> <out.bin>\n", argv[0]); return 1; }
    const int rk45 = strcmp(argv[1], "rk45") == 0;
    const int64_t n = atoll(argv[2]);
    const int64_t K = atoll(argv[3]);
    if (n <= 0 || K <= 0) { fprintf(stderr, "n_envs and steps must be positive\n"); return 1; }
    if (stg_abi_version() != STG_ABI_VERSION) { fprintf(stderr, "ABI %d, header %d\n", stg_abi_version(), STG_ABI_VERSION); return 1; }

    /* SpinTorqueEnv(device_type='stt_mram', include_thermal_fluctuations=False, ...) with its defaults */
    stg_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.solver = rk45 ? STG_SOLVER_RK45 : STG_SOLVER_RK4;
    cfg.thermal = 0;
    cfg.temperature = 300.0;
    cfg.gamma = 2.21e5;
    cfg.max_step = 1e-12;
    cfg.rtol = 1e-6; cfg.atol = 1e-9;
    cfg.max_steps = 100;
    cfg.n_targets = 2;
    cfg.targets[0][2] = 1.0; cfg.targets[1][2] = -1.0;
    cfg.max_current = 2e6;
    cfg.max_duration = 5e-9;
    cfg.success_threshold = 0.9;
    cfg.energy_penalty_weight = 0.1;
    cfg.seed = 0;
    cfg.max_attempts = 200000;
    cfg.noise_corr_time = 1e-12;
    cfg.out_layout = STG_OUT_SOA;

    /* DeviceFactory().get_default_parameters('stt_mram') with 'volume' replaced (G2: a volume at which the current switches
     * the magnetisation: 8.75e-11 for the env's RK4 solver, 9.7e-6 for LLGSSolver) */
    stg_device_params dp;
    memset(&dp, 0, sizeof dp);
    dp.damping = 0.01; dp.ms = 800e3; dp.ku = 1.2e6; dp.volume = rk45 ? 9.7e-6 : 8.75e-11; dp.polarization = 0.7;
    dp.easy_axis[2] = 1.0; dp.demag[2] = 1.0; dp.a_ex = 2e-11; dp.area = 4.999999999999999e-15;
    dp.r_p = 1e3; dp.r_ap = 2e3; dp.ref_m[2] = 1.0; dp.sot_sigma[1] = 1.0; dp.vcma_td = 1e-9; dp.vcma_vbd = 2.0;
    dp.dev_type = STG_DEV_STT; dp.params_valid = 1;

    /* inputs: unit initial states, targets +-z, actions [J, T] as float32 like the env's action space */
    double* h_m0 = malloc(sizeof(double) * 3 * n);
    double* h_tg = malloc(sizeof(double) * 3 * n);
    float* h_act = malloc(sizeof(float) * K * 2 * n);
    sm_state = 0x5EEDull + (uint64_t)n;
    for (int64_t i = 0; i < n; ++i) {
        const double z = 2.0 * sm_uniform() - 1.0, phi = 6.283185307179586 * sm_uniform(), r = sqrt(1.0 - z * z);
        h_m0[0 * n + i] = r * cos(phi); h_m0[1 * n + i] = r * sin(phi); h_m0[2 * n + i] = z;
        h_tg[0 * n + i] = 0.0; h_tg[1 * n + i] = 0.0; h_tg[2 * n + i] = (sm_next() & 1) ? 1.0 : -1.0;
    }
    for (int64_t k = 0; k < K; ++k)
        for (int64_t i = 0; i < n; ++i) {
            h_act[(k * 2 + 0) * n + i] = (float)((2.0 * sm_uniform() - 1.0) * 2e6);
            h_act[(k * 2 + 1) * n + i] = (float)((rk45 ? 2e-11 : 1e-10) + sm_uniform() * (rk45 ? 1.8e-10 : 9e-10));
        }

    double *d_m0, *d_tg, *d_m, *d_etot, *d_energy, *d_rew64;
    float *d_act, *d_obs, *d_rew;
    uint8_t *d_te, *d_tr, *d_st;
    CHECK_HIP(hipMalloc((void**)&d_m0, sizeof(double) * 3 * n));
    CHECK_HIP(hipMalloc((void**)&d_tg, sizeof(double) * 3 * n));
    CHECK_HIP(hipMalloc((void**)&d_m, sizeof(double) * 3 * n));
    CHECK_HIP(hipMalloc((void**)&d_etot, sizeof(double) * n));
    CHECK_HIP(hipMalloc((void**)&d_energy, sizeof(double) * n));
    CHECK_HIP(hipMalloc((void**)&d_rew64, sizeof(double) * n));
    CHECK_HIP(hipMalloc((void**)&d_act, sizeof(float) * 2 * n));
    CHECK_HIP(hipMalloc((void**)&d_obs, sizeof(float) * 12 * n));
    CHECK_HIP(hipMalloc((void**)&d_rew, sizeof(float) * n));
    CHECK_HIP(hipMalloc((void**)&d_te, n));
    CHECK_HIP(hipMalloc((void**)&d_tr, n));
    CHECK_HIP(hipMalloc((void**)&d_st, n));
    CHECK_HIP(hipMemcpy(d_m0, h_m0, sizeof(double) * 3 * n, 1));
    CHECK_HIP(hipMemcpy(d_tg, h_tg, sizeof(double) * 3 * n, 1));

    stg_ctx* ctx = NULL;
    CHECK_STG(stg_create(&ctx, 0, n, 0, &cfg));
    CHECK_STG(stg_set_params(ctx, &dp, 1, NULL));
    CHECK_STG(stg_reset(ctx, NULL, d_m0, d_tg, 0, d_obs, NULL));

    FILE* f = fopen(argv[4], "wb");
    if (!f) { perror(argv[4]); return 1; }
    const int64_t hdr[4] = {n, K, rk45, STG_ABI_VERSION};
    fwrite(hdr, sizeof hdr, 1, f);
    fwrite(h_m0, sizeof(double), 3 * n, f);
    fwrite(h_tg, sizeof(double), 3 * n, f);
    fwrite(h_act, sizeof(float), K * 2 * n, f);
    float* h_obs = malloc(sizeof(float) * 12 * n);
    float* h_rew = malloc(sizeof(float) * n);
    double* h_d = malloc(sizeof(double) * 3 * n);
    uint8_t* h_b = malloc(n);
    CHECK_HIP(hipMemcpy(h_obs, d_obs, sizeof(float) * 12 * n, 2));                 /* (hipMemcpy waits for the reset kernel) */
    fwrite(h_obs, sizeof(float), 12 * n, f);                                       /* reset observation */
    for (int64_t k = 0; k < K; ++k) {
        CHECK_HIP(hipMemcpy(d_act, h_act + k * 2 * n, sizeof(float) * 2 * n, 1));
        CHECK_STG(stg_step(ctx, d_act, 0, d_obs, d_rew, d_rew64, d_energy, d_te, d_tr, d_st, NULL));
        CHECK_STG(stg_get_state(ctx, d_m, NULL, d_etot, NULL, NULL, NULL, NULL));
        CHECK_HIP(hipDeviceSynchronize());
        CHECK_HIP(hipMemcpy(h_obs, d_obs, sizeof(float) * 12 * n, 2)); fwrite(h_obs, sizeof(float), 12 * n, f);
        CHECK_HIP(hipMemcpy(h_rew, d_rew, sizeof(float) * n, 2));      fwrite(h_rew, sizeof(float), n, f);
        CHECK_HIP(hipMemcpy(h_d, d_rew64, sizeof(double) * n, 2));     fwrite(h_d, sizeof(double), n, f);
        CHECK_HIP(hipMemcpy(h_d, d_energy, sizeof(double) * n, 2));    fwrite(h_d, sizeof(double), n, f);
        CHECK_HIP(hipMemcpy(h_b, d_te, n, 2));                         fwrite(h_b, 1, n, f);
        CHECK_HIP(hipMemcpy(h_b, d_tr, n, 2));                         fwrite(h_b, 1, n, f);
        CHECK_HIP(hipMemcpy(h_b, d_st, n, 2));                         fwrite(h_b, 1, n, f);
        CHECK_HIP(hipMemcpy(h_d, d_m, sizeof(double) * 3 * n, 2));     fwrite(h_d, sizeof(double), 3 * n, f);
        CHECK_HIP(hipMemcpy(h_d, d_etot, sizeof(double) * n, 2));      fwrite(h_d, sizeof(double), n, f);
    }
    fclose(f);
    uint64_t counters[4];
    CHECK_STG(stg_get_counters(ctx, counters, 0));
    printf("c_client: %s, %lld envs x %lld steps: %llu env-steps, %llu integrator work units, %llu no-op steps\n", argv[1],
           (long long)n, (long long)K, (unsigned long long)counters[0], (unsigned long long)counters[1], (unsigned long long)counters[3]);
    stg_destroy(ctx);
    hipFree(d_m0); hipFree(d_tg); hipFree(d_m); hipFree(d_etot); hipFree(d_energy); hipFree(d_rew64); hipFree(d_act);
    hipFree(d_obs); hipFree(d_rew); hipFree(d_te); hipFree(d_tr); hipFree(d_st);
    free(h_m0); free(h_tg); free(h_act); free(h_obs); free(h_rew); free(h_d); free(h_b);
    return 0;
}
