// Probe: one macrospin per QUAD (x, y, z on three lanes of a quad, cross products through quad_perm
// DPP moves of the two 32-bit halves of every fp64 operand) against one macrospin per LANE, on the RK45 attempt of the
// T = 0 K, easy-axis-z kernel (LLGSSolver RHS, llgs_solver.py:92-126, + Dormand-Prince stage combinations, error norm and
// the err^-1/5 root): same arithmetic, every FMA spelled out in both forms, so the results must agree BIT FOR BIT -- and
// the question is the time per attempt of a LONE wavefront per SIMD (cfg 2: 4096 envs = 64 one-lane wavefronts, or 256
// quad wavefronts, on 1024 SIMDs) and the instruction count per attempt.
//   build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o quad_lane quad_lane.hip
//   run:   ./quad_lane [envs=4096] [attempts=2000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct V3 { double x, y, z; };
struct K { double gz, alpha, bJ, bpJ, h; };

__device__ __forceinline__ double rsqrt_fast(double s) {
    const double y = __builtin_amdgcn_rsq(s);
    const double e = __builtin_fma(-(s * y), y, 1.0);
    return __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);
}
__device__ __forceinline__ double rcp_fast(double x) {
    const double y = __builtin_amdgcn_rcp(x);
    return __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
}
__device__ __forceinline__ double inv_tenth_root(double e2) {
    const double y = (double)__builtin_amdgcn_exp2f(-0.1f * __builtin_amdgcn_logf((float)e2));
    const double y2 = y * y, y4 = y2 * y2, y8 = y4 * y4;
    const double e = __builtin_fma(-e2, y8 * y2, 1.0);
    return __builtin_fma(y * e, __builtin_fma(0.055, e, 0.1), y);
}

// Dormand-Prince tableau
#define A21 (1.0 / 5)
#define A31 (3.0 / 40)
#define A32 (9.0 / 40)
#define A41 (44.0 / 45)
#define A42 (-56.0 / 15)
#define A43 (32.0 / 9)
#define A51 (19372.0 / 6561)
#define A52 (-25360.0 / 2187)
#define A53 (64448.0 / 6561)
#define A54 (-212.0 / 729)
#define A61 (9017.0 / 3168)
#define A62 (-355.0 / 33)
#define A63 (46732.0 / 5247)
#define A64 (49.0 / 176)
#define A65 (-5103.0 / 18656)
#define B1 (35.0 / 384)
#define B3 (500.0 / 1113)
#define B4 (125.0 / 192)
#define B5 (-2187.0 / 6784)
#define B6 (11.0 / 84)
#define E1 (-71.0 / 57600)
#define E3 (71.0 / 16695)
#define E4 (-71.0 / 1920)
#define E5 (17253.0 / 339200)
#define E6 (-22.0 / 525)
#define E7 (1.0 / 40)

// ---- the stage combinations, one component at a time (identical in both layouts) -----------------------------------
#define FMA __builtin_fma
__device__ __forceinline__ double st2(double y, double k1, double h) { return FMA(k1 * A21, h, y); }
__device__ __forceinline__ double st3(double y, double k1, double k2, double h) { return FMA(FMA(k2, A32, k1 * A31), h, y); }
__device__ __forceinline__ double st4(double y, double k1, double k2, double k3, double h) { return FMA(FMA(k3, A43, FMA(k2, A42, k1 * A41)), h, y); }
__device__ __forceinline__ double st5(double y, double k1, double k2, double k3, double k4, double h) {
    return FMA(FMA(k4, A54, FMA(k3, A53, FMA(k2, A52, k1 * A51))), h, y);
}
__device__ __forceinline__ double st6(double y, double k1, double k2, double k3, double k4, double k5, double h) {
    return FMA(FMA(k5, A65, FMA(k4, A64, FMA(k3, A63, FMA(k2, A62, k1 * A61)))), h, y);
}
__device__ __forceinline__ double ynew(double y, double k1, double k3, double k4, double k5, double k6, double h) {
    return FMA(h, FMA(k6, B6, FMA(k5, B5, FMA(k4, B4, FMA(k3, B3, k1 * B1)))), y);
}
__device__ __forceinline__ double errc(double k1, double k3, double k4, double k5, double k6, double k7, double h) {
    return FMA(k7, E7, FMA(k6, E6, FMA(k5, E5, FMA(k4, E4, FMA(k3, E3, k1 * E1))))) * h;
}
__device__ __forceinline__ double scalec(double y, double yn) { return FMA(fmax(fabs(y), fabs(yn)), 1e-6, 1e-9); }

// ---- one lane per macrospin ------------------------------------------------------------------------------------------
__device__ __forceinline__ V3 rhs1(const V3& y, const K& k) {
    const double ss = FMA(y.z, y.z, FMA(y.y, y.y, y.x * y.x));
    const double inv = rsqrt_fast(ss);
    const V3 m{y.x * inv, y.y * inv, y.z * inv};
    const double g = k.gz * m.z;
    const V3 dm{m.y * g, -(m.x * g), 0.0};
    const V3 mxdm{-(m.z * dm.y), m.z * dm.x, FMA(m.x, dm.y, -(m.y * dm.x))};
    const V3 d2{FMA(k.alpha, mxdm.x, dm.x), FMA(k.alpha, mxdm.y, dm.y), FMA(k.alpha, mxdm.z, dm.z)};
    const double uz = -FMA(m.x, m.x, m.y * m.y);
    return V3{FMA(k.bJ, m.x * m.z, FMA(k.bpJ, m.y, d2.x)), FMA(k.bJ, m.y * m.z, FMA(-k.bpJ, m.x, d2.y)), FMA(k.bJ, uz, d2.z)};
}

__global__ void __launch_bounds__(64) one_lane(const double* y0, double* out, double* hsum, long long* cyc, int n, int attempts, K k) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    V3 y{y0[i], y0[n + i], y0[2 * n + i]};
    V3 f = rhs1(y, k);
    double h = k.h, acc = 0.0;
    const long long t0 = __builtin_readcyclecounter();
    for (int a = 0; a < attempts; ++a) {
        const V3 k1 = f;
        const V3 k2 = rhs1(V3{st2(y.x, k1.x, h), st2(y.y, k1.y, h), st2(y.z, k1.z, h)}, k);
        const V3 k3 = rhs1(V3{st3(y.x, k1.x, k2.x, h), st3(y.y, k1.y, k2.y, h), st3(y.z, k1.z, k2.z, h)}, k);
        const V3 k4 = rhs1(V3{st4(y.x, k1.x, k2.x, k3.x, h), st4(y.y, k1.y, k2.y, k3.y, h), st4(y.z, k1.z, k2.z, k3.z, h)}, k);
        const V3 k5 = rhs1(V3{st5(y.x, k1.x, k2.x, k3.x, k4.x, h), st5(y.y, k1.y, k2.y, k3.y, k4.y, h), st5(y.z, k1.z, k2.z, k3.z, k4.z, h)}, k);
        const V3 k6 = rhs1(V3{st6(y.x, k1.x, k2.x, k3.x, k4.x, k5.x, h), st6(y.y, k1.y, k2.y, k3.y, k4.y, k5.y, h),
                              st6(y.z, k1.z, k2.z, k3.z, k4.z, k5.z, h)}, k);
        const V3 yn{ynew(y.x, k1.x, k3.x, k4.x, k5.x, k6.x, h), ynew(y.y, k1.y, k3.y, k4.y, k5.y, k6.y, h), ynew(y.z, k1.z, k3.z, k4.z, k5.z, k6.z, h)};
        const V3 k7 = rhs1(yn, k);
        const double qx = errc(k1.x, k3.x, k4.x, k5.x, k6.x, k7.x, h) * rcp_fast(scalec(y.x, yn.x));
        const double qy = errc(k1.y, k3.y, k4.y, k5.y, k6.y, k7.y, h) * rcp_fast(scalec(y.y, yn.y));
        const double qz = errc(k1.z, k3.z, k4.z, k5.z, k6.z, k7.z, h) * rcp_fast(scalec(y.z, yn.z));
        const double err2 = FMA(qz, qz, FMA(qy, qy, qx * qx)) * (1.0 / 3.0);
        const double fac = fmin(10.0, 0.9 * inv_tenth_root(err2));
        h = fmin(1e-12, fmax(h * fac, 1e-13));          // (SciPy's controller, reduced to what keeps the dependency)
        acc += h;
        y = yn; f = k7;
    }
    const long long t1 = __builtin_readcyclecounter();
    out[i] = y.x; out[n + i] = y.y; out[2 * n + i] = y.z;
    hsum[i] = acc;
    if (i == 0) *cyc = t1 - t0;
}

// ---- one QUAD per macrospin: lane 4q + c holds component c (c = 3 idles) ---------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
constexpr int ROT1 = 1 | (2 << 2) | (0 << 4) | (3 << 6);    // lane c reads c+1 (mod 3)
constexpr int ROT2 = 2 | (0 << 2) | (1 << 4) | (3 << 6);    // lane c reads c+2 (mod 3)
constexpr int BC0 = 0;                                       // every lane reads lane 0 of its quad
constexpr int BC2 = 2 | (2 << 2) | (2 << 4) | (2 << 6);     // ... lane 2

struct LaneK { double k1, k2, qa, qb, pa, pb; };   // per-lane 0/1 selectors (see rhsq)

// own = y_c.  Every operation is the one the one-lane form performs for component c (the generic cross product formula
// reproduces the specialised zero-skipping forms exactly: fma(a, b, -0) = a*b), selectors are exact 0/1 multiplications.
__device__ __forceinline__ double rhsq(double own, const K& k, const LaneK& s) {
    const double y1 = dpp64<ROT1>(own), y2 = dpp64<ROT2>(own);
    const double ss_l = FMA(y2, y2, FMA(y1, y1, own * own));         // lane 0 of the quad: x, y, z order
    const double ss = dpp64<BC0>(ss_l);
    const double inv = rsqrt_fast(ss);
    const double m0 = own * inv, m1 = y1 * inv, m2 = y2 * inv;        // m_c, m_{c+1}, m_{c+2}: the same products as their owners'
    const double mz = dpp64<BC2>(m0);
    const double g = k.gz * mz;
    const double sel = FMA(s.k1, m1, s.k2 * m2);                      // (m.y, -m.x, 0)[c]
    const double dm = sel * g;
    const double d1 = dpp64<ROT1>(dm), d2 = dpp64<ROT2>(dm);
    const double mxdm = FMA(m1, d2, -(m2 * d1));                      // (m x dm)_c
    const double dn = FMA(k.alpha, mxdm, dm);
    const double inner = FMA(s.qa * k.bpJ, m1, FMA(s.qb * k.bpJ, m2, dn));   // c0: +bpJ m.y, c1: -bpJ m.x, c2: nothing
    const double uz = -FMA(m1, m1, m2 * m2);                          // lane 2: -(mx^2 + my^2)
    const double P = FMA(s.pa, m0 * mz, s.pb * uz);                   // (mx mz, my mz, uz)[c]
    return FMA(k.bJ, P, inner);
}

__global__ void __launch_bounds__(64) quad_lane(const double* y0, double* out, double* hsum, long long* cyc, int n, int attempts, K k) {
    const int lane = blockIdx.x * 64 + threadIdx.x;
    const int env = lane >> 2, c = lane & 3;
    if (env >= n) return;
    const int cc = c < 3 ? c : 2;                                     // the idle lane mirrors component 2 (never read by others)
    const LaneK s{cc == 0 ? 1.0 : 0.0, cc == 1 ? -1.0 : 0.0, cc == 0 ? 1.0 : 0.0, cc == 1 ? -1.0 : 0.0, cc < 2 ? 1.0 : 0.0, cc == 2 ? 1.0 : 0.0};
    double y = y0[cc * n + env];
    double f = rhsq(y, k, s);
    double h = k.h, acc = 0.0;
    const long long t0 = __builtin_readcyclecounter();
    for (int a = 0; a < attempts; ++a) {
        const double k1 = f;
        const double k2 = rhsq(st2(y, k1, h), k, s);
        const double k3 = rhsq(st3(y, k1, k2, h), k, s);
        const double k4 = rhsq(st4(y, k1, k2, k3, h), k, s);
        const double k5 = rhsq(st5(y, k1, k2, k3, k4, h), k, s);
        const double k6 = rhsq(st6(y, k1, k2, k3, k4, k5, h), k, s);
        const double yn = ynew(y, k1, k3, k4, k5, k6, h);
        const double k7 = rhsq(yn, k, s);
        const double q = errc(k1, k3, k4, k5, k6, k7, h) * rcp_fast(scalec(y, yn));
        const double q1 = dpp64<ROT1>(q), q2 = dpp64<ROT2>(q);
        const double e_l = FMA(q2, q2, FMA(q1, q1, q * q)) * (1.0 / 3.0);    // x, y, z order in lane 0
        const double err2 = dpp64<BC0>(e_l);
        const double fac = fmin(10.0, 0.9 * inv_tenth_root(err2));
        h = fmin(1e-12, fmax(h * fac, 1e-13));
        acc += h;
        y = yn; f = k7;
    }
    const long long t1 = __builtin_readcyclecounter();
    if (c < 3) out[c * n + env] = y;
    if (c == 0) hsum[env] = acc;
    if (lane == 0) *cyc = t1 - t0;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4096, attempts = argc > 2 ? atoi(argv[2]) : 2000;
    std::vector<double> y0(3 * n);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / (1 << 24) * 2 - 1; };
    for (int i = 0; i < n; ++i) {
        double x = rnd(), y = rnd(), z = rnd(), r = sqrt(x * x + y * y + z * z) + 1e-9;
        y0[i] = x / r; y0[n + i] = y / r; y0[2 * n + i] = z / r;
    }
    // default STT constants at volume 9.7e-6: -gamma (H_k - Ms), alpha, beta J, beta' J (J = 1.5e6), h0 = 0.6 ps
    const K k{-2.21e5 * (2.0 * 1.2e6 / (4e-7 * 3.14159265358979 * 8e5) - 8e5), 0.01, 0.7 * 2.21e5 / (2 * 8e5 * 9.7e-6) * 1.5e6,
              0.07 * 2.21e5 / (2 * 8e5 * 9.7e-6) * 1.5e6, 6e-13};
    double *d_y0, *d_o1, *d_o2, *d_h1, *d_h2; long long* d_c;
    hipMalloc(&d_y0, 24 * n); hipMalloc(&d_o1, 24 * n); hipMalloc(&d_o2, 24 * n); hipMalloc(&d_h1, 8 * n); hipMalloc(&d_h2, 8 * n); hipMalloc(&d_c, 16);
    hipMemcpy(d_y0, y0.data(), 24 * n, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms1 = 0, ms2 = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(one_lane, dim3((n + 63) / 64), dim3(64), 0, 0, d_y0, d_o1, d_h1, d_c, n, attempts, k);
        hipEventRecord(e1); hipDeviceSynchronize(); hipEventElapsedTime(&ms1, e0, e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(quad_lane, dim3((4 * n + 63) / 64), dim3(64), 0, 0, d_y0, d_o2, d_h2, d_c + 1, n, attempts, k);
        hipEventRecord(e1); hipDeviceSynchronize(); hipEventElapsedTime(&ms2, e0, e1);
    }
    std::vector<double> o1(3 * n), o2(3 * n), h1(n), h2(n);
    long long cy[2];
    hipMemcpy(o1.data(), d_o1, 24 * n, hipMemcpyDeviceToHost); hipMemcpy(o2.data(), d_o2, 24 * n, hipMemcpyDeviceToHost);
    hipMemcpy(h1.data(), d_h1, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), d_h2, 8 * n, hipMemcpyDeviceToHost);
    hipMemcpy(cy, d_c, 16, hipMemcpyDeviceToHost);
    long long diff = 0;
    for (int i = 0; i < 3 * n; ++i) diff += memcmp(&o1[i], &o2[i], 8) != 0 && !(o1[i] == 0.0 && o2[i] == 0.0);
    for (int i = 0; i < n; ++i) diff += memcmp(&h1[i], &h2[i], 8) != 0;
    printf("envs %d, attempts %d: one lane per env %.3f ms (%d wavefronts, %.0f cycles/attempt)   one quad per env %.3f ms (%d wavefronts, %.0f cycles/attempt)   "
           "speed-up %.3f   values differing in any bit: %lld of %d   (sample m_z %.15f, sum h %.6e)\n",
           n, attempts, ms1, (n + 63) / 64, (double)cy[0] / attempts, ms2, (4 * n + 63) / 64, (double)cy[1] / attempts, ms1 / ms2, diff, 4 * n,
           o1[2 * n], h1[0]);
    return diff != 0;
}
