// kbench.hip — in-process A/B of launch shapes for the commit kernel (K1).
// Development tool, not part of the library: variants are timed interleaved in
// one process (HIP events), over R rotating cold batches, so the numbers are
// comparable.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iraft_amd/csrc
//                     -o gpurun_out/kbench tools/kbench.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "kernels.h"

using namespace rafthip;

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

template <bool NT>
__device__ __forceinline__ Vec<u64, 2> ld2(const u64* p) {
    if constexpr (NT) {
        Vec<u64, 2> r;
        r.v[0] = __builtin_nontemporal_load(p);
        r.v[1] = __builtin_nontemporal_load(p + 1);
        return r;
    } else {
        return load_vec<u64, 2>(p);
    }
}
template <bool NT>
__device__ __forceinline__ void st2(u64* p, const Vec<u64, 2>& x) {
    if constexpr (NT) {
        __builtin_nontemporal_store(x.v[0], p);
        __builtin_nontemporal_store(x.v[1], p + 1);
    } else {
        store_vec<u64, 2>(p, x);
    }
}

// CH chunks of 2 groups per lane; chunk c of a block covers a contiguous run of
// BLOCK*2 groups, so each wave instruction still reads 1 KiB contiguous.
template <int V, int CH, int BLOCK, int NT, bool SUM>
__global__ __launch_bounds__(BLOCK) void kv(const u64* __restrict__ match, u64 stride, u64 G,
                                            u64* __restrict__ commit, u64 tiles) {
    for (u64 tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const u64 base = tile * (u64)(BLOCK * 2 * CH) + threadIdx.x * 2;
        Vec<u64, 2> in[CH][V];
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int s = 0; s < V; ++s) in[c][s] = ld2<(NT == 1 || NT == 2)>(match + (u64)s * stride + base + (u64)c * BLOCK * 2);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            Vec<u64, 2> res;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                u64 a[V];
#pragma unroll
                for (int s = 0; s < V; ++s) a[s] = in[c][s].v[j];
                if constexpr (SUM) {
                    u64 t = 0;
#pragma unroll
                    for (int s = 0; s < V; ++s) t += a[s];
                    res.v[j] = t;
                } else {
                    res.v[j] = committed_fixed<V>(a);
                }
            }
            st2<(NT == 1 || NT == 3)>(commit + base + (u64)c * BLOCK * 2, res);
        }
    }
}

// Buffer-instruction form with explicit cache-policy bits (aux: 1 = sc0, 2 = nt,
// 16 = sc1) and an optional XCD-chunked tile order: workgroup b runs on XCD b % 8,
// so tile = (b % 8) * (tiles / 8) + b / 8 gives every XCD one contiguous eighth.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int V, int AUXL, int AUXS, bool XCD>
__global__ __launch_bounds__(512) void kb(const u64* __restrict__ match, u64 stride, u64 G,
                                          u64* __restrict__ commit, u64 tiles) {
    constexpr int CH = 2, BLOCK = 512;
    u64 tile = blockIdx.x;
    if (XCD && tiles % 8 == 0) tile = (u64)(blockIdx.x % 8) * (tiles / 8) + blockIdx.x / 8;
    const u64 tbase = tile * (u64)(BLOCK * 2 * CH);
    const u32 voff = threadIdx.x * 16;
    u32x4 in[CH][V];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int s = 0; s < V; ++s) {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(match + (u64)s * stride + tbase + (u64)c * BLOCK * 2), 0, BLOCK * 16, 0x00020000);
            in[c][s] = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, AUXL);
        }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        u32x4 res;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            u64 a[V];
#pragma unroll
            for (int s = 0; s < V; ++s) a[s] = ((u64)in[c][s][2 * j + 1] << 32) | in[c][s][2 * j];
            const u64 r = committed_fixed<V>(a);
            res[2 * j] = (u32)r;
            res[2 * j + 1] = (u32)(r >> 32);
        }
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(commit + tbase + (u64)c * BLOCK * 2), 0, BLOCK * 16, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(res, r, voff, 0, AUXS);
    }
}
template <int V, int AUXL, int AUXS, bool XCD>
void launch_kb(const u64* m, u64 stride, u64 G, u64* c, hipStream_t st) {
    const u64 tiles = (G + 2047) / 2048;
    hipLaunchKernelGGL((kb<V, AUXL, AUXS, XCD>), dim3((unsigned)tiles), dim3(512), 0, st, m, stride, G, c, tiles);
}

struct Variant {
    std::string name;
    void (*launch)(const u64*, u64, u64, u64*, hipStream_t);
};

template <int V, int CH, int BLOCK, int NT, bool SUM, int PERSIST /*blocks per CU, 0 = one tile per block*/>
void launch_kv(const u64* m, u64 stride, u64 G, u64* c, hipStream_t st) {
    constexpr u64 per = (u64)BLOCK * 2 * CH;
    static_assert(4096 % per == 0, "tile must divide the plane stride granule, or the last tile runs off the plane");
    const u64 tiles = (G + per - 1) / per;
    u64 grid = tiles;
    if (PERSIST > 0) grid = std::min<u64>(tiles, (u64)256 * PERSIST);
    hipLaunchKernelGGL((kv<V, CH, BLOCK, NT, SUM>), dim3((unsigned)grid), dim3(BLOCK), 0, st, m, stride, G, c, tiles);
}

int main(int argc, char** argv) {
    const u64 G = argc > 1 ? strtoull(argv[1], 0, 0) : (1ull << 20);
    const int R = argc > 2 ? atoi(argv[2]) : 12;
    const int rounds = argc > 3 ? atoi(argv[3]) : 7;
    const int iters = argc > 4 ? atoi(argv[4]) : 240;
    constexpr int V = 5;
    const u64 stride = (G + 4095) / 4096 * 4096;
    std::vector<u64*> match(R), commit(R);
    std::vector<u64> host((size_t)V * stride);
    u64 x = 0x5EED;
    for (int r = 0; r < R; ++r) {
        for (auto& h : host) {
            x += 0x9E3779B97F4A7C15ull;
            u64 z = x;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            h = (z ^ (z >> 31)) >> 1;
        }
        CK(hipMalloc(&match[r], host.size() * 8));
        CK(hipMalloc(&commit[r], stride * 8));
        CK(hipMemcpy(match[r], host.data(), host.size() * 8, hipMemcpyHostToDevice));
    }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    std::vector<Variant> vs = {
        {"ch2_b256", launch_kv<V, 2, 256, 0, false, 0>},
        {"ch2_b256_nt", launch_kv<V, 2, 256, 1, false, 0>},
        {"ch2_b256_ntL", launch_kv<V, 2, 256, 2, false, 0>},
        {"ch2_b256_ntS", launch_kv<V, 2, 256, 3, false, 0>},
        {"ch1_b256_nt", launch_kv<V, 1, 256, 1, false, 0>},
        {"ch4_b256_nt", launch_kv<V, 4, 256, 1, false, 0>},
        {"ch2_b64_nt", launch_kv<V, 2, 64, 1, false, 0>},
        {"ch2_b128_nt", launch_kv<V, 2, 128, 1, false, 0>},
        {"ch2_b512_nt", launch_kv<V, 2, 512, 1, false, 0>},
        {"ch2_b1024_nt", launch_kv<V, 2, 1024, 1, false, 0>},
        {"ch1_b1024_nt", launch_kv<V, 1, 1024, 1, false, 0>},
        {"ch4_b64_nt", launch_kv<V, 4, 64, 1, false, 0>},
        {"ch2_b512_nt_p2", launch_kv<V, 2, 512, 1, false, 2>},
        {"SUM_ch2_b256_nt", launch_kv<V, 2, 256, 1, true, 0>},
        {"buf_plain", launch_kb<V, 0, 0, false>},
        {"buf_nt_nt", launch_kb<V, 2, 2, false>},
        {"buf_nt_nt_xcd", launch_kb<V, 2, 2, true>},
        {"buf_sc1_nt", launch_kb<V, 16, 2, false>},
        {"buf_sc1nt_nt", launch_kb<V, 18, 2, false>},
        {"buf_sc0sc1nt_nt", launch_kb<V, 19, 2, false>},
        {"buf_nt_sc1nt", launch_kb<V, 2, 18, false>},
        {"buf_nt_sc0sc1nt", launch_kb<V, 2, 19, false>},
        {"buf_nt_plain_xcd", launch_kb<V, 2, 0, true>},
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<std::vector<float>> us(vs.size());
    for (int rd = 0; rd < rounds; ++rd) {
        for (size_t v = 0; v < vs.size(); ++v) {
            for (int i = 0; i < 24; ++i) vs[v].launch(match[i % R], stride, G, commit[i % R], st);
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) vs[v].launch(match[i % R], stride, G, commit[i % R], st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            us[v].push_back(ms * 1e3f / iters);
        }
    }
    CK(hipGetLastError());
    printf("G=%llu R=%d rounds=%d iters=%d bytes/launch=%.1f MB\n", G, R, rounds, iters, G * 48 / 1e6);
    for (size_t v = 0; v < vs.size(); ++v) {
        std::sort(us[v].begin(), us[v].end());
        const float med = us[v][us[v].size() / 2], mn = us[v][0];
        printf("%-18s median %8.2f us  min %8.2f us  -> %7.1f GB/s (median)\n", vs[v].name.c_str(), med, mn,
               G * 48.0 / med / 1e3);
    }
    return 0;
}
