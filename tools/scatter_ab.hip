// scatter_ab.hip — what limits K5?  The same 2^24 random (cell, index) records
// applied several ways to a Match plane of a given size: 64-bit atomicMax (what
// k_scatter_acks does), 32-bit atomicMax on the low word (is it the 64-bit RMW?),
// plain load + conditional plain store (racy on duplicates; the basis of a
// store-then-verify scheme), load only, the atomic at workgroup scope (where does
// the RMW execute?) and read-before-atomic.  Indexes grow per launch so every
// launch updates its cells.  Development probe, not part of the library.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Rec { u32 group, slot; u64 index; };

template <int MODE>
__global__ __launch_bounds__(256) void k(u64* __restrict__ match, u64 stride, const Rec* __restrict__ recs, u64 n,
                                         u64* __restrict__ sink, u64 bump) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Rec r = recs[i];
    r.index += bump;  // grows per launch, so every launch really updates its cells
    u64* cell = match + (u64)r.slot * stride + r.group;
    if (MODE == 0) {
        atomicMax(cell, r.index);
    } else if (MODE == 1) {
        atomicMax(reinterpret_cast<u32*>(cell), (u32)r.index);
    } else if (MODE == 2) {
        const u64 old = __builtin_nontemporal_load(cell);
        if (r.index > old) __builtin_nontemporal_store(r.index, cell);
    } else if (MODE == 3) {
        const u64 old = *cell;
        if (r.index > old) *cell = r.index;
    } else if (MODE == 4) {
        const u64 old = *cell;
        if (old == 0x123456789abcdefull) *sink = old;
    } else if (MODE == 5) {  // same RMW, workgroup scope: may execute in the XCD's own L2 (NOT coherent across XCDs)
        __hip_atomic_fetch_max(cell, r.index, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {                 // read-before-atomic: the RMW only when it would change the cell
        if (*cell < r.index) atomicMax(cell, r.index);
    }
}

int main() {
    const u64 N = 1ull << 24;
    const int S = 9;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int lg : {16, 20, 22, 24}) {
        const u64 G = 1ull << lg;
        std::vector<Rec> h(N);
        u64 x = 0x5EED + lg;
        for (auto& r : h) {
            x += 0x9E3779B97F4A7C15ull;
            u64 z = x;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            r.group = (u32)(z % G);
            r.slot = (u32)((z >> 40) % S);
            r.index = z >> 20;  // < 2^44: the per-launch bump (i << 48) dominates
        }
        Rec* d;
        u64 *m, *sink;
        CK(hipMalloc(&d, N * sizeof(Rec)));
        CK(hipMalloc(&m, (size_t)S * G * 8));
        CK(hipMalloc(&sink, 8));
        CK(hipMemcpy(d, h.data(), N * sizeof(Rec), hipMemcpyHostToDevice));
        const char* names[] = {"atomicMax u64", "atomicMax u32", "load + nt cond store", "load + cond store", "load only",
                               "atomicMax u64 wg-scope", "load, atomicMax if larger"};
        auto run = [&](auto kern, int mode) {
            CK(hipMemset(m, 0, (size_t)S * G * 8));
            for (int i = 0; i < 2; ++i)
                hipLaunchKernelGGL(kern, dim3((unsigned)(N / 256)), dim3(256), 0, st, m, G, d, N, sink, (u64)i << 48);
            CK(hipEventRecord(e0, st));
            for (int i = 2; i < 12; ++i)
                hipLaunchKernelGGL(kern, dim3((unsigned)(N / 256)), dim3(256), 0, st, m, G, d, N, sink, (u64)i << 48);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("G=2^%d planes %7.1f MB  %-22s %8.1f us  %5.1f G rec/s\n", lg, S * G * 8 / 1e6, names[mode], ms * 100,
                   N / (ms * 100) / 1e3);
        };
        run(k<0>, 0);
        run(k<1>, 1);
        run(k<2>, 2);
        run(k<3>, 3);
        run(k<4>, 4);
        run(k<5>, 5);
        run(k<6>, 6);
        CK(hipFree(d));
        CK(hipFree(m));
        CK(hipFree(sink));
    }
    return 0;
}
