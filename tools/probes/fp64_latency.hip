// Probe: issue interval of dependent / independent v_fma_f64 (and v_mul/v_add/v_rsq/v_rcp f64) for ONE wavefront per SIMD.
// build: hipcc -O2 --offload-arch=gfx950 -o fp64_latency fp64_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ int g_active = 64;
template <int ILP, int OP>
__global__ void chain(double* out, long long* cyc, int iters, double a, double b) {
    if ((int)threadIdx.x >= g_active) return;
    double x[ILP];
#pragma unroll
    for (int j = 0; j < ILP; ++j) x[j] = threadIdx.x * 1e-3 + j;
    const long long t0 = __builtin_readcyclecounter();   // s_memtime
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int j = 0; j < ILP; ++j) {
                if (OP == 0) x[j] = __builtin_fma(x[j], a, b);
                else if (OP == 1) x[j] = x[j] * a;
                else if (OP == 2) x[j] = x[j] + b;
                else if (OP == 3) x[j] = __builtin_amdgcn_rsq(x[j]);
                else if (OP == 4) x[j] = __builtin_amdgcn_rcp(x[j]);
                else if (OP == 5) { float f = (float)x[j]; f = __builtin_fmaf(f, (float)a, (float)b); x[j] = f; }
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int j = 0; j < ILP; ++j) s += x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int ILP, int OP>
void run(const char* name, int blocks) {
    double* o; long long* c; hipMalloc(&o, blocks * 64 * 8); hipMalloc(&c, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((chain<ILP, OP>), dim3(blocks), dim3(64), 0, 0, o, c, 10, 1.0000001, 1e-9);
    hipEventRecord(e0);
    hipLaunchKernelGGL((chain<ILP, OP>), dim3(blocks), dim3(64), 0, 0, o, c, iters, 1.0000001, 1e-9);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 16 * ILP;
    printf("%-10s ILP=%d blocks=%5d: %.2f ns/instr/wave (wall), memtime ticks/instr %.3f\n", name, ILP, blocks, ms * 1e6 / n, (double)cy / n);
    hipFree(o); hipFree(c);
}

int main() {
    for (int act : {16, 32, 48}) {
        hipMemcpyToSymbol(HIP_SYMBOL(g_active), &act, 4);
        printf("active lanes = %d\n", act);
        run<1, 0>("fma_f64", 1024); run<3, 0>("fma_f64", 1024); run<3, 0>("fma_f64", 2048);
    }
    { int act = 64; hipMemcpyToSymbol(HIP_SYMBOL(g_active), &act, 4); }
    printf("-- one wavefront per CU (256 x 64) vs four (1024 x 64 / 256 x 256 below)\n");
    run<1, 3>("rsq_f64", 256); run<3, 3>("rsq_f64", 256); run<1, 4>("rcp_f64", 256); run<3, 0>("fma_f64", 256);
    for (int blocks : {1024, 2048}) {
        run<1, 0>("fma_f64", blocks); run<2, 0>("fma_f64", blocks); run<3, 0>("fma_f64", blocks); run<4, 0>("fma_f64", blocks);
        run<1, 1>("mul_f64", blocks); run<3, 1>("mul_f64", blocks);
        run<1, 2>("add_f64", blocks); run<3, 2>("add_f64", blocks);
        run<1, 3>("rsq_f64", blocks); run<3, 3>("rsq_f64", blocks);
        run<1, 4>("rcp_f64", blocks); run<3, 4>("rcp_f64", blocks);
        run<1, 5>("cvt+fma32", blocks);
    }
    return 0;
}
