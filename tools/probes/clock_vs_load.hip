// Probe: does the shader clock depend on how many wavefronts run fp64 work?  Every wavefront runs the same FMA loop and
// records s_memtime (shader-clock ticks) and s_memrealtime (constant 100 MHz) at its start and end.
//   clock = d(memtime) / d(memrealtime) x 100 MHz;  rounds = whether all wavefronts were resident at once.
// build: hipcc -O2 --offload-arch=gfx950 -o clock_vs_load clock_vs_load.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int OP>
__global__ void work(double* out, long long* rec, int iters, double a, double b) {
    double x[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) x[j] = threadIdx.x * 1e-3 + j;
    float f[3] = {1.0f + threadIdx.x, 2.0f, 3.0f};
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (OP == 0) x[j] = __builtin_fma(x[j], a, b);
                else f[j] = __builtin_fmaf(f[j], (float)a, (float)b);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2] + f[0] + f[1] + f[2];
    if ((threadIdx.x & 63) == 0) {
        long long* r = rec + 4 * ((size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64);
        r[0] = t0; r[1] = t1; r[2] = r0; r[3] = r1;
    }
}

template <int OP>
void run(const char* name, int blocks, int threads, int iters) {
    const int waves = blocks * (threads / 64);
    double* o; long long* rec;
    hipMalloc(&o, (size_t)blocks * threads * 8); hipMalloc(&rec, (size_t)waves * 32);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((work<OP>), dim3(blocks), dim3(threads), 0, 0, o, rec, 10, 1.0000001, 1e-9);
    hipEventRecord(e0);
    hipLaunchKernelGGL((work<OP>), dim3(blocks), dim3(threads), 0, 0, o, rec, iters, 1.0000001, 1e-9);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h((size_t)waves * 4);
    hipMemcpy(h.data(), rec, h.size() * 8, hipMemcpyDeviceToHost);
    double ticks = 0, real = 0; long long rmin = h[2], rmax_start = h[2], rend = h[3];
    for (int w = 0; w < waves; ++w) {
        ticks += (double)(h[4 * w + 1] - h[4 * w]); real += (double)(h[4 * w + 3] - h[4 * w + 2]);
        rmin = std::min(rmin, h[4 * w + 2]); rmax_start = std::max(rmax_start, h[4 * w + 2]); rend = std::max(rend, h[4 * w + 3]);
    }
    const double n = (double)iters * 48;
    printf("%-8s %5d x %3d (%5d waves): launch %.3f ms; per wave %.3f ticks/instr, lifetime %.3f ms, clock %.0f MHz; "
           "last start %.3f ms after first, span %.3f ms\n", name, blocks, threads, waves, ms, ticks / waves / n,
           real / waves * 1e-5, ticks / real * 100.0, (rmax_start - rmin) * 1e-5, (rend - rmin) * 1e-5);
    hipFree(o); hipFree(rec);
}

int main(int argc, char** argv) {
    if (argc > 1) {   // sustained load: clock_vs_load <iters> [reps]: 1024 x 256 and 256 x 256 fp64 launches back to back
        const int it = atoi(argv[1]), reps = argc > 2 ? atoi(argv[2]) : 5;
        for (int r = 0; r < reps; ++r) { run<0>("fma_f64", 1024, 256, it); run<0>("fma_f64", 256, 256, it); }
        return 0;
    }
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("fma_f64", 64, 64, iters); run<0>("fma_f64", 256, 64, iters); run<0>("fma_f64", 1024, 64, iters);
        run<0>("fma_f64", 2048, 64, iters); run<0>("fma_f64", 4096, 64, iters);
        run<0>("fma_f64", 256, 256, iters); run<0>("fma_f64", 512, 256, iters); run<0>("fma_f64", 1024, 256, iters);
        run<0>("fma_f64", 2048, 256, iters);
        run<1>("fma_f32", 256, 256, iters); run<1>("fma_f32", 1024, 256, iters); run<1>("fma_f32", 2048, 256, iters);
    }
    return 0;
}
