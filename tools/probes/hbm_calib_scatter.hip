// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for the SCATTERED record accesses of the env-step kernels: a lane reads its env's
// 64-byte state record as four 16-byte loads and writes a 56-byte output record as seven 8-byte stores; under the duration-sorted
// schedule the records of a wavefront are scattered over its tile, and at a refill point of the lane-refill kernel only a few lanes
// of the wavefront load at all.  hbm_calib.hip covers coalesced full-wavefront accesses (FETCH_SIZE reports half the bytes there).
// build: hipcc -O2 --offload-arch=gfx950 -o hbm_calib_scatter hbm_calib_scatter.hip
// run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./hbm_calib_scatter   (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr uint64_t NREC = 1ull << 24;                 // 16 Mi records of 64 B = 1 GiB (past the 256 MiB Infinity Cache)
constexpr uint64_t BYTES = NREC * 64;
// every access is bounds-checked against the allocation (a violation is counted, not performed): the probe cannot fault whatever the
// index arithmetic does
__device__ unsigned long long g_violations;
__device__ __forceinline__ bool ok(uint64_t byte_off, uint64_t len) {
    if (byte_off + len <= BYTES) return true;
    atomicAdd(&g_violations, 1ull);
    return false;
}
// odd multiplier: a permutation of [0, NREC).  The masked value goes through an opaque register barrier: hipcc 7.2 (-O2, gfx950) was
// seen to turn `(x & 0xFFFFFF) * 56 + base` into v_mul_lo_u32 / v_add_u32 / v_mad_u64_u32 WITHOUT the mask (the 24-bit-multiply
// combine dropped it and the 64-bit multiply-add does not truncate) and then to remove the bounds check as provably true -- the first
// two runs of this probe wrote up to 240 GB past the buffer and faulted the GPU (round 4, profiles/EXPERIMENTS.md).  Check the ISA
// (`v_and_b32 ..., 0xffffff`) before running it.
__device__ __forceinline__ uint64_t scatter(uint64_t i) {
    uint32_t r = (uint32_t)((i * 0x9E3779B1ull + 12345ull) & (NREC - 1));
    asm volatile("" : "+v"(r));
    r &= (uint32_t)(NREC - 1);
    asm volatile("" : "+v"(r));
    return (uint64_t)r;
}

// every ACTIVE lane reads one whole 64-byte record (4 x 16 B) at a scattered position; `active` of the 64 lanes of a wavefront take part
template <int ACTIVE>
__global__ void rd_rec(const double2* in, double* out, uint64_t nwaves) {
    const uint64_t w = (uint64_t)blockIdx.x, lane = threadIdx.x;
    double s = 0;
    if (lane < ACTIVE) {
        const uint64_t r = scatter(w * ACTIVE + lane);
        if (ok(r * 64, 64)) {
            const double2* p = in + r * 4;
            const double2 a = p[0], b = p[1], c = p[2], d = p[3];
            s = a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
        }
    }
    if (s == -1.0) out[0] = s;
}
// the same record read with the records of a wavefront CONTIGUOUS (identity schedule): 4 KB per wavefront
__global__ void rd_rec_contig(const double2* in, double* out, uint64_t nwaves) {
    const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    const uint64_t r = i & (NREC - 1);
    double s = 0;
    if (ok(r * 64, 64)) {
        const double2* p = in + r * 4;
        const double2 a = p[0], b = p[1], c = p[2], d = p[3];
        s = a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
    }
    if (s == -1.0) out[0] = s;
}
// every active lane writes one 56-byte record (7 x 8 B) at a scattered 56-byte-strided position
template <int ACTIVE>
__global__ void wr_rec56(float2* outp, uint64_t nwaves) {
    const uint64_t w = (uint64_t)blockIdx.x, lane = threadIdx.x;
    if (lane < ACTIVE) {
        const uint64_t r = scatter(w * ACTIVE + lane);
        if (ok(r * 56, 56)) {
            float2* p = (float2*)((char*)outp + r * 56);
#pragma unroll
            for (int k = 0; k < 7; ++k) p[k] = make_float2((float)lane, (float)k);
        }
    }
}
// ... and one 64-byte state record (4 x 16 B)
template <int ACTIVE>
__global__ void wr_rec64(double2* outp, uint64_t nwaves) {
    const uint64_t w = (uint64_t)blockIdx.x, lane = threadIdx.x;
    if (lane < ACTIVE) {
        const uint64_t r = scatter(w * ACTIVE + lane);
        if (ok(r * 64, 64)) {
            double2* p = outp + r * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) p[k] = make_double2((double)lane, (double)k);
        }
    }
}

int main() {
    void *a = nullptr, *b = nullptr; double* o = nullptr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
    CK(hipMalloc(&a, BYTES)); CK(hipMalloc(&b, BYTES)); CK(hipMalloc(&o, 8));
    CK(hipMemset(a, 1, BYTES)); CK(hipMemset(b, 0, BYTES)); CK(hipMemset(o, 0, 8));
    CK(hipDeviceSynchronize());
    printf("buffers: a = %p, b = %p, %llu bytes each\n", a, b, (unsigned long long)BYTES); fflush(stdout);
    const uint64_t nrec = 1ull << 22;                 // records touched per kernel: 4 Mi x 64 B = 256 MiB expected (reads), 224 MiB (56-B writes)
    hipLaunchKernelGGL(rd_rec<64>, dim3((unsigned)(nrec / 64)), dim3(64), 0, 0, (const double2*)a, o, nrec / 64);
    hipLaunchKernelGGL(rd_rec<8>, dim3((unsigned)(nrec / 8)), dim3(64), 0, 0, (const double2*)a, o, nrec / 8);
    hipLaunchKernelGGL(rd_rec<1>, dim3((unsigned)(nrec / 1)), dim3(64), 0, 0, (const double2*)a, o, nrec / 1);
    hipLaunchKernelGGL(rd_rec_contig, dim3((unsigned)(nrec / 64)), dim3(64), 0, 0, (const double2*)a, o, nrec / 64);
    hipLaunchKernelGGL(wr_rec56<64>, dim3((unsigned)(nrec / 64)), dim3(64), 0, 0, (float2*)b, nrec / 64);
    hipLaunchKernelGGL(wr_rec56<8>, dim3((unsigned)(nrec / 8)), dim3(64), 0, 0, (float2*)b, nrec / 8);
    hipLaunchKernelGGL(wr_rec64<64>, dim3((unsigned)(nrec / 64)), dim3(64), 0, 0, (double2*)b, nrec / 64);
    hipLaunchKernelGGL(wr_rec64<8>, dim3((unsigned)(nrec / 8)), dim3(64), 0, 0, (double2*)b, nrec / 8);
    CK(hipDeviceSynchronize());
    unsigned long long viol = 0;
    CK(hipMemcpyFromSymbol(&viol, HIP_SYMBOL(g_violations), sizeof(viol)));
    printf("bounds violations: %llu\n", viol);
    printf("records per kernel = %llu: expected 262144 KB per read kernel and per 64-B write kernel, 229376 KB per 56-B write kernel\n", (unsigned long long)nrec);
    return 0;
}
