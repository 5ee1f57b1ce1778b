// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for the access widths of the env-step kernel (8 B, 4 B and 1 B per
// lane, coalesced, one element per lane), as MI355X_MICROARCH.md asks for widths other than 16 B/lane.
// build: hipcc -O2 --offload-arch=gfx950 -o hbm_calib hbm_calib.hip
// run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./hbm_calib   (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <typename T> __global__ void rd(const T* in, double* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0;
    if (i < n) s = (double)in[i];
    if (s == -1.0) out[0] = s;            // never true: keeps the load
}
template <typename T> __global__ void wr(T* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (T)i;
}
__global__ void rd16(const double2* in, double* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0;
    if (i < n) { const double2 v = in[i]; s = v.x + v.y; }
    if (s == -1.0) out[0] = s;
}
int main() {
    const size_t n = 64u << 20;                 // 64 Mi elements: 512 MiB of f64 (past the 256 MiB Infinity Cache)
    void *a, *b; double* o;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 8); hipMalloc(&o, 8);
    hipMemset(a, 1, n * 16); hipMemset(b, 0, n * 8);
    const dim3 g((unsigned)(n / 64)), blk(64);  // 64-thread workgroups like the step kernel
    hipLaunchKernelGGL(rd<double>, g, blk, 0, 0, (const double*)a, o, n);     // 8 B/lane loads:  n*8 bytes
    hipLaunchKernelGGL(rd<float>, g, blk, 0, 0, (const float*)a, o, n);       // 4 B/lane loads:  n*4
    hipLaunchKernelGGL(rd<uint8_t>, g, blk, 0, 0, (const uint8_t*)a, o, n);   // 1 B/lane loads:  n
    hipLaunchKernelGGL(rd16, g, blk, 0, 0, (const double2*)a, o, n);          // 16 B/lane loads: n*16 (the guide's case)
    hipLaunchKernelGGL(wr<double>, g, blk, 0, 0, (double*)b, n);              // 8 B/lane stores
    hipLaunchKernelGGL(wr<float>, g, blk, 0, 0, (float*)b, n);                // 4 B/lane stores
    hipLaunchKernelGGL(wr<uint8_t>, g, blk, 0, 0, (uint8_t*)b, n);            // 1 B/lane stores
    hipDeviceSynchronize();
    printf("n = %zu elements\n", n);
    return 0;
}
