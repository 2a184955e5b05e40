// scatter_bin.hip — the tile-binned alternative to K5 (k_scatter_acks), built and measured
// (DESIGN.md §4).  Same 2^24 uniformly random acks over 2^22 groups x 9 slots:
//
//   atomic   one 64-bit atomicMax per ack on a random cell (what the library does)
//   binned   count   every workgroup histograms its slice of the acks by tile of T groups (LDS)
//            scan    per tile: offsets of every workgroup's run, tile bases
//            bin     the slice again: each ack goes to its tile's bin (LDS cursors, no global atomics)
//            apply   one workgroup per tile: the tile's S x T cells staged in LDS, the bin's acks
//                    applied with LDS 64-bit max atomics, the tile written back with whole-line stores
//
// Prints the duration of every kernel and checks that both forms leave the same Match planes.
// Development probe, not part of the library.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/scatter_bin tools/scatter_bin.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Rec { u32 group, slot; u64 index; };

constexpr int S = 9;
constexpr u32 T = 1024;        // groups per tile: S*T*8 = 72 KB of LDS in the apply pass
constexpr u32 kBinBlock = 1024;

__global__ __launch_bounds__(256) void k_atomic(u64* __restrict__ match, u64 stride, const Rec* __restrict__ recs, u64 n, u64 bump) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Rec r = recs[i];
    atomicMax(match + (u64)r.slot * stride + r.group, r.index + bump);
}

// counts[wg][tile] = acks of workgroup wg's slice that fall into tile
__global__ __launch_bounds__(kBinBlock) void k_count(const Rec* __restrict__ recs, u64 n, u64 slice, u32 ntiles,
                                                     u32* __restrict__ counts) {
    extern __shared__ u32 hist[];
    for (u32 t = threadIdx.x; t < ntiles; t += kBinBlock) hist[t] = 0;
    __syncthreads();
    const u64 lo = (u64)blockIdx.x * slice, hi = lo + slice < n ? lo + slice : n;
    for (u64 i = lo + threadIdx.x; i < hi; i += kBinBlock) atomicAdd(&hist[recs[i].group / T], 1u);
    __syncthreads();
    for (u32 t = threadIdx.x; t < ntiles; t += kBinBlock) counts[(u64)blockIdx.x * ntiles + t] = hist[t];
}

// per tile: exclusive scan of counts over the workgroups (in place -> run offsets inside the tile), tile totals
__global__ __launch_bounds__(256) void k_scan_wg(u32* __restrict__ counts, u32 nwg, u32 ntiles, u32* __restrict__ tile_total) {
    const u32 t = blockIdx.x * 256 + threadIdx.x;
    if (t >= ntiles) return;
    u32 run = 0;
    for (u32 w = 0; w < nwg; ++w) {
        const u32 c = counts[(u64)w * ntiles + t];
        counts[(u64)w * ntiles + t] = run;
        run += c;
    }
    tile_total[t] = run;
}
// exclusive scan of the tile totals (one workgroup)
__global__ __launch_bounds__(1024) void k_scan_tiles(const u32* __restrict__ tile_total, u32 ntiles, u32* __restrict__ tile_base) {
    __shared__ u32 part[1024];
    const u32 t = threadIdx.x, per = (ntiles + 1023) / 1024;
    const u32 lo = t * per, hi = lo + per < ntiles ? lo + per : ntiles;
    u32 sum = 0;
    for (u32 i = lo; i < hi; ++i) sum += tile_total[i];
    part[t] = sum;
    __syncthreads();
    for (u32 d = 1; d < 1024; d <<= 1) {
        const u32 add = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    u32 run = part[t] - sum;
    for (u32 i = lo; i < hi; ++i) {
        tile_base[i] = run;
        run += tile_total[i];
    }
    if (t == 1023) tile_base[ntiles] = part[1023];
}

// the slice again: every ack to bins[tile_base[tile] + offset of this workgroup's run + rank within the run]
__global__ __launch_bounds__(kBinBlock) void k_bin(const Rec* __restrict__ recs, u64 n, u64 slice, u32 ntiles,
                                                   const u32* __restrict__ counts, const u32* __restrict__ tile_base,
                                                   Rec* __restrict__ bins, u64 bump) {
    extern __shared__ u32 cursor[];
    for (u32 t = threadIdx.x; t < ntiles; t += kBinBlock) cursor[t] = tile_base[t] + counts[(u64)blockIdx.x * ntiles + t];
    __syncthreads();
    const u64 lo = (u64)blockIdx.x * slice, hi = lo + slice < n ? lo + slice : n;
    for (u64 i = lo + threadIdx.x; i < hi; i += kBinBlock) {
        Rec r = recs[i];
        r.index += bump;
        const u32 pos = atomicAdd(&cursor[r.group / T], 1u);
        bins[pos] = r;
    }
}

// one workgroup per tile: cells in LDS, LDS max atomics, whole tile back
__global__ __launch_bounds__(512) void k_apply(u64* __restrict__ match, u64 stride, const Rec* __restrict__ bins,
                                               const u32* __restrict__ tile_base) {
    extern __shared__ u64 cells[];  // [S][T]
    const u32 tile = blockIdx.x;
    const u32 lo = tile_base[tile], hi = tile_base[tile + 1];
    if (lo == hi) return;
    const u64 g0 = (u64)tile * T;
    for (u32 i = threadIdx.x * 2; i < S * T; i += 1024) {
        const u32 s = i / T, t = i % T;
        const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(match + (u64)s * stride + g0 + t);
        cells[i] = v.x;
        cells[i + 1] = v.y;
    }
    __syncthreads();
    for (u32 i = lo + threadIdx.x; i < hi; i += 512) {
        const Rec r = bins[i];
        atomicMax(&cells[r.slot * T + (r.group - (u32)g0)], r.index);
    }
    __syncthreads();
    for (u32 i = threadIdx.x * 2; i < S * T; i += 1024) {
        const u32 s = i / T, t = i % T;
        ulonglong2 v;
        v.x = cells[i];
        v.y = cells[i + 1];
        *reinterpret_cast<ulonglong2*>(match + (u64)s * stride + g0 + t) = v;
    }
}

int main(int argc, char** argv) {
    // usage: scatter_bin [slices [log2 acks [log2 groups]]]   (defaults: 512 slices, 2^24 acks over 2^22 groups)
    const u64 N = 1ull << (argc > 2 ? atoi(argv[2]) : 24), G = 1ull << (argc > 3 ? atoi(argv[3]) : 22);
    const u32 ntiles = (u32)(G / T);
    const u32 nwg = argc > 1 ? (u32)atoi(argv[1]) : 512;  // slices
    const u64 slice = (N + nwg - 1) / nwg;
    std::vector<Rec> h(N);
    u64 x = 0x5EED;
    for (auto& r : h) {
        x += 0x9E3779B97F4A7C15ull;
        u64 z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        r.group = (u32)(z % G);
        r.slot = (u32)((z >> 40) % S);
        r.index = z >> 20;
    }
    Rec *d, *bins;
    u64 *m_a, *m_b;
    u32 *counts, *tile_total, *tile_base;
    CK(hipMalloc(&d, N * sizeof(Rec)));
    CK(hipMalloc(&bins, N * sizeof(Rec)));
    CK(hipMalloc(&m_a, (size_t)S * G * 8));
    CK(hipMalloc(&m_b, (size_t)S * G * 8));
    CK(hipMalloc(&counts, (size_t)nwg * ntiles * 4));
    CK(hipMalloc(&tile_total, (size_t)ntiles * 4));
    CK(hipMalloc(&tile_base, (size_t)(ntiles + 1) * 4));
    CK(hipMemcpy(d, h.data(), N * sizeof(Rec), hipMemcpyHostToDevice));
    CK(hipMemset(m_a, 0, (size_t)S * G * 8));
    CK(hipMemset(m_b, 0, (size_t)S * G * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_apply), hipFuncAttributeMaxDynamicSharedMemorySize, S * T * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_count), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(ntiles * 4)));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bin), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(ntiles * 4)));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t ev[6];
    for (auto& e : ev) CK(hipEventCreate(&e));
    const int R = 12;
    double t[5] = {0, 0, 0, 0, 0};
    float ms;
    for (int rep = 0; rep < R; ++rep) {
        const u64 bump = (u64)(rep + 1) << 44;
        CK(hipEventRecord(ev[0], st));
        hipLaunchKernelGGL(k_atomic, dim3((unsigned)(N / 256)), dim3(256), 0, st, m_a, G, d, N, bump);
        CK(hipEventRecord(ev[1], st));
        hipLaunchKernelGGL(k_count, dim3(nwg), dim3(kBinBlock), ntiles * 4, st, d, N, slice, ntiles, counts);
        CK(hipEventRecord(ev[2], st));
        hipLaunchKernelGGL(k_scan_wg, dim3((ntiles + 255) / 256), dim3(256), 0, st, counts, nwg, ntiles, tile_total);
        hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, st, tile_total, ntiles, tile_base);
        CK(hipEventRecord(ev[3], st));
        hipLaunchKernelGGL(k_bin, dim3(nwg), dim3(kBinBlock), ntiles * 4, st, d, N, slice, ntiles, counts, tile_base, bins, bump);
        CK(hipEventRecord(ev[4], st));
        hipLaunchKernelGGL(k_apply, dim3(ntiles), dim3(512), S * T * 8, st, m_b, G, bins, tile_base);
        CK(hipEventRecord(ev[5], st));
        CK(hipStreamSynchronize(st));
        CK(hipGetLastError());
        if (rep >= 2)
            for (int k = 0; k < 5; ++k) {
                CK(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
                t[k] += ms * 1e3 / (R - 2);
            }
    }
    std::vector<u64> a((size_t)S * G), b((size_t)S * G);
    CK(hipMemcpy(a.data(), m_a, a.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), m_b, b.size() * 8, hipMemcpyDeviceToHost));
    const bool same = memcmp(a.data(), b.data(), a.size() * 8) == 0;
    printf("%llu random acks, %llu groups x 9 slots, tiles of %u groups, %u slices\n", N, G, T, nwg);
    printf("atomic  k_scatter_acks form              %8.1f us\n", t[0]);
    printf("binned  count %.1f + scan %.1f + bin %.1f + apply %.1f = %8.1f us\n", t[1], t[2], t[3], t[4], t[1] + t[2] + t[3] + t[4]);
    printf("planes %s\n", same ? "identical" : "DIFFER");
    return same ? 0 : 1;
}
