// Probe: does the issue interval of v_fma_f64 / v_mul_f64 depend on how many operands come from VGPRs?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(double* out, const double* in, int iters, double sa, double sb) {
    double x0 = in[threadIdx.x], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    const double va = in[64 + threadIdx.x], vb = in[128 + threadIdx.x];   // per-lane (VGPR) operands
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) { x0 = __builtin_fma(x0, sa, sb); x1 = __builtin_fma(x1, sa, sb); x2 = __builtin_fma(x2, sa, sb); x3 = __builtin_fma(x3, sa, sb); }
            if (MODE == 1) { x0 = __builtin_fma(x0, va, sb); x1 = __builtin_fma(x1, va, sb); x2 = __builtin_fma(x2, va, sb); x3 = __builtin_fma(x3, va, sb); }
            if (MODE == 2) { x0 = __builtin_fma(x0, va, vb); x1 = __builtin_fma(x1, va, vb); x2 = __builtin_fma(x2, va, vb); x3 = __builtin_fma(x3, va, vb); }
            if (MODE == 3) { x0 = x0 * va; x1 = x1 * va; x2 = x2 * va; x3 = x3 * va; }
            if (MODE == 4) { x0 = __builtin_fma(x1, x2, x0); x1 = __builtin_fma(x2, x3, x1); x2 = __builtin_fma(x3, x0, x2); x3 = __builtin_fma(x0, x1, x3); }   // all-VGPR, cross-dependent at distance >= 1..3
            if (MODE == 5) { x0 = __builtin_fma(x0, va, vb); x1 = __builtin_fma(x1, vb, va); x2 = __builtin_fma(x2, va, x0); x3 = __builtin_fma(x3, vb, x1); }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3;
}
template <int MODE> void run(const char* name, int blocks, double* o, double* in) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, o, in, 10, 1.0000001, 1e-9);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, o, in, iters, 1.0000001, 1e-9);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s blocks=%d: %.2f ns/instr/wave\n", name, blocks, ms * 1e6 / (iters * 64.0));
}
int main() {
    double *o, *in; hipMalloc(&o, 4096 * 64 * 8); hipMalloc(&in, 192 * 8);
    double h[192]; for (int i = 0; i < 192; ++i) h[i] = 1.0 + 1e-9 * i; hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    for (int blocks : {1024, 2048, 4096}) {
        run<0>("fma x,S,S (1 VGPR src)", blocks, o, in);
        run<1>("fma x,V,S (2 VGPR src)", blocks, o, in);
        run<2>("fma x,V,V (3 VGPR src)", blocks, o, in);
        run<3>("mul x,V   (2 VGPR src)", blocks, o, in);
        run<4>("fma all-VGPR cross-dependent", blocks, o, in);
        run<5>("fma 3 VGPR mixed", blocks, o, in);
    }
    return 0;
}
