// Probe: where does the dispatcher put the wavefronts of a workgroup?  Records HW_ID (SIMD, CU, SE) and XCC_ID per wave.
// build: hipcc -O2 --offload-arch=gfx950 -o wave_placement wave_placement.hip ; run: ./wave_placement <blocks> <threads>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <tuple>

__global__ void probe(uint32_t* out, int spin) {
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID, 32 bits
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID, 4 bits
    double x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0000001 + 1e-9;                       // keep every wave resident for a while
    const int wave = threadIdx.x / 64;
    if ((threadIdx.x & 63) == 0) {
        const size_t o = ((size_t)blockIdx.x * (blockDim.x / 64) + wave) * 2;
        out[o] = hw; out[o + 1] = xcc | (x == 12345.0 ? 1u << 31 : 0);
    }
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 1024, threads = argc > 2 ? atoi(argv[2]) : 128;
    const int wpb = threads / 64;
    uint32_t* d; hipMalloc(&d, (size_t)blocks * wpb * 8);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<uint32_t> h((size_t)blocks * wpb * 2);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // per (xcc, se, cu, simd): which (block, wave) landed there
    std::map<std::tuple<int,int,int,int>, std::vector<std::pair<int,int>>> m;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < wpb; ++w) {
        const uint32_t hw = h[((size_t)b * wpb + w) * 2], xcc = h[((size_t)b * wpb + w) * 2 + 1] & 0xF;
        const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        m[{(int)xcc, se * 2 + sh, cu, simd}].push_back({b, w});
    }
    int shown = 0; std::map<int,int> hist; std::map<int,int> same_role;
    for (auto& kv : m) {
        hist[(int)kv.second.size()]++;
        int w0 = 0; for (auto& p : kv.second) w0 += (p.second % 2 == 0);
        same_role[w0 * 10 + (int)kv.second.size()]++;
        if (shown < 24) {
            printf("xcc %d se %d cu %2d simd %d:", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first));
            for (auto& p : kv.second) printf(" (b%d,w%d)", p.first, p.second);
            printf("\n"); ++shown;
        }
    }
    // for SIMDs holding waves of several workgroups: histogram of (block of an even wave - block of an odd wave) / 8
    std::map<int,int> dh;
    for (auto& kv : m) for (auto& p : kv.second) for (auto& q : kv.second)
        if (p.second % 2 == 0 && q.second % 2 == 1 && p.first != q.first) dh[((p.first - q.first) % blocks + blocks) % blocks]++;
    printf("block(even wave) - block(odd wave) mod nblocks on shared SIMDs:");
    for (auto& kv : dh) printf(" %d:%d", kv.first, kv.second);
    printf("\n");
    // any two wavefronts on one SIMD: histogram of the difference of their block ids (one-wavefront workgroups: which arrivals double up)
    std::map<int,int> ph;
    for (auto& kv : m) for (size_t x = 0; x < kv.second.size(); ++x) for (size_t y = x + 1; y < kv.second.size(); ++y)
        ph[abs(kv.second[x].first - kv.second[y].first)]++;
    printf("|block - block| of wavefront pairs sharing a SIMD:");
    { int shown2 = 0; for (auto& kv : ph) if (shown2++ < 12) printf(" %d:%d", kv.first, kv.second); }
    printf("\n");
    printf("SIMDs used: %zu\n", m.size());
    for (auto& kv : hist) printf("  SIMDs holding %d waves: %d\n", kv.first, kv.second);
    for (auto& kv : same_role) printf("  SIMDs with %d even-index waves out of %d: %d\n", kv.first / 10, kv.first % 10, kv.second);
    return 0;
}
