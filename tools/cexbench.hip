// cexbench.hip — width sweep of the register sorting-network commit kernel
// (k_commit_regs) with a checksum, used to A/B code shapes of the
// compare-exchange and of the row addressing (development tool).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels.h"
using namespace rafthip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int N, typename MT>
float run(const u64* m, u64 stride, u64 G, int S, u64* out, hipStream_t st, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const unsigned blocks = (unsigned)(stride / kBlock);
    LogBounds lb{nullptr, nullptr};
    for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((k_commit_regs<N, MT, kStore>), dim3(blocks), dim3(kBlock), 0, st, m, stride, G, S,
                           (const MT*)nullptr, (const MT*)nullptr, S, out, lb, (Pair*)nullptr, (u32*)nullptr);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL((k_commit_regs<N, MT, kStore>), dim3(blocks), dim3(kBlock), 0, st, m, stride, G, S,
                           (const MT*)nullptr, (const MT*)nullptr, S, out, lb, (Pair*)nullptr, (u32*)nullptr);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

int main() {
    const u64 G = 1ull << 22, stride = G;
    const int S = 64;
    std::vector<u64> host((size_t)S * stride);
    u64 x = 0x5EED;
    for (auto& h : host) {
        x += 0x9E3779B97F4A7C15ull;
        u64 z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        h = z ^ (z >> 31);
    }
    u64 *m, *out;
    CK(hipMalloc(&m, host.size() * 8));
    CK(hipMalloc(&out, stride * 8));
    CK(hipMemcpy(m, host.data(), host.size() * 8, hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    // checksum so the variants can be compared for equality
    auto sum = [&]() {
        std::vector<u64> h(G);
        CK(hipMemcpy(h.data(), out, G * 8, hipMemcpyDeviceToHost));
        u64 s = 0;
        for (u64 v : h) s = s * 1000003ull + v;
        return s;
    };
    float t;
    t = run<12, uint16_t>(m, stride, G, 9, out, st, 10);
    printf("N=12 S=9   %8.1f us  %6.0f GB/s  sum %016llx\n", t, G * 82.0 / t / 1e3, sum());
    t = run<16, uint16_t>(m, stride, G, 16, out, st, 10);
    printf("N=16 S=16  %8.1f us  %6.0f GB/s  sum %016llx\n", t, G * 136.0 / t / 1e3, sum());
    t = run<32, uint32_t>(m, stride, G, 32, out, st, 10);
    printf("N=32 S=32  %8.1f us  %6.0f GB/s  sum %016llx\n", t, G * 264.0 / t / 1e3, sum());
    t = run<64, uint64_t>(m, stride, G, 64, out, st, 10);
    printf("N=64 S=64  %8.1f us  %6.0f GB/s  sum %016llx\n", t, G * 520.0 / t / 1e3, sum());
    return 0;
}
