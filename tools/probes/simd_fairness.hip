// Probe: how does a SIMD arbitrate between TWO resident wavefronts that are both always ready?  One 512-thread workgroup = 8
// wavefronts = two per SIMD of one CU (wavefronts w and w + 4 share SIMD w % 4, checked through HW_ID); every wavefront runs the same
// loop of dependent fp64 FMAs (ILP chains per wavefront = template parameter) and records when it finished.  Fair (round-robin)
// arbitration: both wavefronts of a SIMD finish together; age-based (oldest first): one finishes at ~55-60 % of the other's time.
// A second launch gives the two wavefronts of a SIMD opposite s_setprio levels that swap every `slice` iterations.
// The only memory the kernel touches is out[0 .. 8*4): fixed indices, no data-dependent addressing.
// build: hipcc -O2 --offload-arch=gfx950 -o simd_fairness simd_fairness.hip ; run: ./simd_fairness
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int ILP, int MODE>      // MODE 0: plain; 1: static priority for the YOUNGER wavefront of each SIMD; 2: priorities swap every `slice`
                                  // iterations (by each wavefront's own count); 3: priority = bit 12 of the 100 MHz real-time counter (a
                                  // clock both wavefronts share: always opposite) XOR the wavefront's slot parity, re-read every `slice` iterations
__global__ void __launch_bounds__(512) probe(unsigned long long* out, int iters, int slice, double seed) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave-uniform in an SGPR: s_setprio ignores EXEC, so the
                                                                                  // branches around it must be scalar branches
    const uint32_t hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
    double x[ILP];
#pragma unroll
    for (int k = 0; k < ILP; ++k) x[k] = seed + k + threadIdx.x * 1e-3;
    const int second = wave >= 4 ? 1 : 0;
    if (MODE == 1) { if (second) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 2 && (i % slice) == 0) {
            if ((((i / slice) & 1) ^ second) != 0) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
        }
        if (MODE == 3 && (i % slice) == 0) {
            const int ph = (int)((__builtin_amdgcn_s_memrealtime() >> 12) & 1ull);      // flips every 41 us
            if ((ph ^ (int)(hw & 1u)) != 0) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int k = 0; k < ILP; ++k) x[k] = __builtin_fma(x[k], 1.0000001, 1e-9);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) s += x[k];
    if ((threadIdx.x & 63) == 0) {
        out[wave * 4 + 0] = t0; out[wave * 4 + 1] = t1; out[wave * 4 + 2] = hw; out[wave * 4 + 3] = (s == 12345.678) ? 1ull : 0ull;
    }
}

template <int ILP, int MODE>
static void run(const char* tag, unsigned long long* d, int iters, int slice) {
    unsigned long long h[32];
    hipLaunchKernelGGL((probe<ILP, MODE>), dim3(1), dim3(512), 0, 0, d, iters, slice, 1.0);
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { printf("%s: HIP error\n", tag); return; }
    unsigned long long base = h[0];
    for (int w = 0; w < 8; ++w) if (h[w * 4] < base) base = h[w * 4];
    printf("%s (ILP %d): per wavefront [SIMD: start..end in kilocycles]:", tag, ILP);
    for (int w = 0; w < 8; ++w)
        printf(" w%d[s%d: %.1f..%.1f]", w, (int)((h[w * 4 + 2] >> 4) & 3), (h[w * 4] - base) * 1e-3, (h[w * 4 + 1] - base) * 1e-3);
    double fin_a = 0, fin_b = 0;
    for (int w = 0; w < 4; ++w) { fin_a += (h[w * 4 + 1] - base) * 0.25e-3; fin_b += (h[(w + 4) * 4 + 1] - base) * 0.25e-3; }
    printf("\n   mean finish of wavefronts 0-3: %.1f, of wavefronts 4-7: %.1f kilocycles; per 16*ILP-FMA iteration: %.1f / %.1f cycles\n",
           fin_a, fin_b, fin_a * 1e3 / iters, fin_b * 1e3 / iters);
}

int main() {
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, 32 * sizeof(unsigned long long)) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    const int iters = 20000;
    run<1, 0>("plain", d, iters, 1);
    run<3, 0>("plain", d, iters, 1);
    run<3, 1>("younger wavefront at s_setprio 2", d, iters, 1);
    run<3, 2>("priorities swap every 256 iterations", d, iters, 256);
    run<1, 2>("priorities swap every 256 iterations", d, iters, 256);
    run<3, 3>("priority from the shared real-time clock, re-read every 64 iterations", d, iters, 64);
    run<1, 3>("priority from the shared real-time clock, re-read every 64 iterations", d, iters, 64);
    run<1, 1>("younger wavefront at s_setprio 2", d, iters, 1);
    hipFree(d);
    return 0;
}
