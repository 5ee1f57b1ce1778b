// Probe: does a lone wavefront per SIMD issue fp64 instructions slower from a long straight-line loop body (instruction
// fetch) than from a short one?  Body = R x 4 independent v_fma_f64 (8-byte VOP3 encodings) or v_fmac_f64_e32 (4 bytes).
#include <hip/hip_runtime.h>
#include <cstdio>
#define FMA4 asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
#define FMAC4 asm volatile("v_fmac_f64_e32 %0, %4, %5\n v_fmac_f64_e32 %1, %4, %5\n v_fmac_f64_e32 %2, %4, %5\n v_fmac_f64_e32 %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
#define R4(X) X X X X
#define R16(X) R4(R4(X))
#define R64(X) R4(R16(X))
#define R128(X) R64(X) R64(X)
__device__ int g_phase = 0;
template <int MODE>
__global__ void k(double* out, const double* in, int iters) {
    double x0 = in[threadIdx.x & 63], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    const double a = in[64 + (threadIdx.x & 63)], b = in[128 + (threadIdx.x & 63)];
    if (g_phase) {                              // put the wavefronts of a CU out of phase with each other
        const int skew = ((threadIdx.x >> 6) * 131 + blockIdx.x * 37) % 509;
        for (int i = 0; i < skew; ++i) { FMA4 }
    }
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { R4(FMA4) }            // 16 instr, 128 B
        if (MODE == 1) { R128(FMA4) }          // 512 instr, 4 KB
        if (MODE == 2) { R128(FMA4) R128(FMA4) R128(FMA4) R128(FMA4) }   // 2048 instr, 16 KB
        if (MODE == 3) { R128(FMAC4) }         // 512 instr, 2 KB
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}
template <int MODE> void run(const char* name, int blocks, int threads, int per_iter, double* o, double* in) {
    const int iters = 4000000 / per_iter;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 0, 0, o, in, 2);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 0, 0, o, in, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %4d x %3d threads: %.2f ns/instr/wave\n", name, blocks, threads, ms * 1e6 / ((double)iters * per_iter));
}
int main() {
    double *o, *in; hipMalloc(&o, 4096 * 512 * 8); hipMalloc(&in, 192 * 8);
    double h[192]; for (int i = 0; i < 192; ++i) h[i] = 1.0 + 1e-9 * i; hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemcpyToSymbol(HIP_SYMBOL(g_phase), &rep, 4);
        printf("-- wavefronts %s\n", rep ? "out of phase" : "in phase");
        run<0>("body 16 x fma (128 B)", 256, 256, 16, o, in);
        run<1>("body 512 x fma (4 KB)", 256, 256, 512, o, in);
        run<2>("body 2048 x fma (16 KB)", 256, 256, 2048, o, in);
        run<3>("body 512 x fmac_e32 (2 KB)", 256, 256, 512, o, in);
        run<1>("body 512 x fma (4 KB)", 64, 256, 512, o, in);      // 64 CUs busy, 4 waves each
        run<1>("body 512 x fma (4 KB)", 256, 64, 512, o, in);      // ~1 wave per CU
        run<1>("body 512 x fma (4 KB)", 512, 256, 512, o, in);     // 2 waves per SIMD
    }
    return 0;
}
