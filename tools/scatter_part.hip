// scatter_part.hip — the two-level, LDS-staged partition form of K5, as a
// stand-alone probe: same random acks through (a) one 64-bit atomicMax per ack and (b)
//
//   count   every workgroup histograms its slice of the acks by tile of T groups in LDS and adds the
//           histogram into a global per-tile count (contiguous atomics)
//   scan    one workgroup: exclusive scan of the tile counts -> tile bases; bucket / tile cursors;
//           chunk table of pass 2
//   pass 1  partition by bucket (128 tiles): a workgroup takes a chunk of C records, ranks them per
//           bucket in LDS, reserves one run per bucket with one global atomic, stages the records
//           grouped by bucket in LDS and writes whole runs (16 B per lane, consecutive lanes ->
//           consecutive addresses)
//   pass 2  the same inside every bucket, by tile
//   apply   one workgroup per tile: the tile's S x T cells staged in LDS, its bin applied with LDS
//           64-bit max, the tile written back with whole-line stores
//
// The kernels are the library's (raft_amd/csrc/scatter_part.h); this program times each of them
// and checks that both forms leave identical Match planes.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iraft_amd/csrc -o tools/bin/scatter_part tools/scatter_part.hip
// Usage: scatter_part [log2 acks [log2 groups [skew [slots]]]]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "scatter_part.h"
using namespace rafthip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_atomic(u64* __restrict__ match, u64 stride, const Ack* __restrict__ recs, u64 n) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Ack r = recs[i];
    atomicMax(match + (u64)r.slot * stride + r.group, r.index);
}
__global__ void k_bump(Ack* recs, u64 n, u64 bump) {  // every repetition really moves its cells
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < n) recs[i].index += bump;
}

int main(int argc, char** argv) {
    const u64 N = 1ull << (argc > 1 ? atoi(argv[1]) : 24), G = 1ull << (argc > 2 ? atoi(argv[2]) : 22);
    const double zipf = argc > 3 ? atof(argv[3]) : 0.0;
    const u32 S = argc > 4 ? (u32)atoi(argv[4]) : 9;
    u32 shift = 10;
    while (((size_t)S << shift) * 8 > 72u * 1024u) --shift;
    const u32 tiles = (u32)(G >> shift), tiles_pad = (tiles + 127) / 128 * 128, nb = tiles_pad >> kPartLB;
    if (tiles_pad > kPartMaxTiles) { fprintf(stderr, "too many tiles\n"); return 2; }
    std::vector<Ack> h(N);
    u64 x = 0x5EED;
    for (auto& r : h) {
        x += 0x9E3779B97F4A7C15ull;
        u64 z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        if (zipf > 0) {  // skewed groups: u^k concentrates the acks on the low group ids
            const double u = (double)(z & 0xFFFFFFFFFFFFull) / (double)(1ull << 48);
            r.group = (u32)((double)G * pow(u, zipf));
            if (r.group >= G) r.group = (u32)G - 1;
        } else {
            r.group = (u32)(z % G);
        }
        r.slot = (u32)((z >> 40) % S);
        r.index = z >> 20;
    }
    Ack *d, *b1, *b2;
    u64 *m_a, *m_b, *rej;
    u32* words;
    CK(hipMalloc(&d, N * sizeof(Ack)));
    CK(hipMalloc(&b1, N * sizeof(Ack)));
    CK(hipMalloc(&b2, N * sizeof(Ack)));
    CK(hipMalloc(&m_a, (size_t)S * G * 8));
    CK(hipMalloc(&m_b, (size_t)S * G * 8));
    CK(hipMalloc(&rej, 8));
    const size_t nwords = (size_t)kPartMaxTiles * 3 + 1 + kPartMaxBuckets * 2 + 1 + 1;
    CK(hipMalloc(&words, nwords * 4));
    CK(hipMemset(words, 0, nwords * 4));
    CK(hipMemset(rej, 0, 8));
    PartBufs pb;
    pb.tile_count = words;
    pb.tile_base = pb.tile_count + kPartMaxTiles;
    pb.tile_cursor = pb.tile_base + kPartMaxTiles + 1;
    pb.bucket_cursor = pb.tile_cursor + kPartMaxTiles;
    pb.chunk_first = pb.bucket_cursor + kPartMaxBuckets;
    pb.flag = pb.chunk_first + kPartMaxBuckets + 1;
    CK(hipMemcpy(d, h.data(), N * sizeof(Ack), hipMemcpyHostToDevice));
    CK(hipMemset(m_a, 0, (size_t)S * G * 8));
    CK(hipMemset(m_b, 0, (size_t)S * G * 8));
    const size_t lds_count = (size_t)tiles_pad * 4, lds_stage = (size_t)kPartChunk * 16, lds_cells = ((size_t)S << shift) * 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_count), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kPartMaxTiles * 4)));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_scan), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kPartMaxTiles * 4)));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_partition<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_stage));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_partition<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_stage));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_apply), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t ev[8];
    for (auto& e : ev) CK(hipEventCreate(&e));
    const int R = 10;
    double t[7] = {0, 0, 0, 0, 0, 0, 0};
    float ms;
    const u32 nwg = 512, persistent = 256;  // one 1024-lane partition workgroup per CU
    const u64 slice = (N + nwg - 1) / nwg;
    const u64 chunks1 = (N + kPartChunk - 1) / kPartChunk, chunks2 = chunks1 + nb;
    u32 flag = 0;
    for (int rep = 0; rep < R; ++rep) {
        hipLaunchKernelGGL(k_bump, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, d, N, 1ull << 44);
        CK(hipEventRecord(ev[0], st));
        hipLaunchKernelGGL(k_atomic, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, m_a, G, d, N);
        CK(hipEventRecord(ev[1], st));
        hipLaunchKernelGGL(k_part_count, dim3(nwg), dim3(kPartCountBlock), lds_count, st, d, N, slice, G, S, shift, tiles_pad, pb.tile_count, rej);
        CK(hipEventRecord(ev[2], st));
        hipLaunchKernelGGL(k_part_scan, dim3(1), dim3(1024), lds_count, st, pb, tiles_pad, nb);
        CK(hipEventRecord(ev[3], st));
        hipLaunchKernelGGL(k_part_partition<1>, dim3((u32)(chunks1 < persistent ? chunks1 : persistent)), dim3(kPartBlock), lds_stage, st, d, N, G, S, shift, nb, pb, b1);
        CK(hipEventRecord(ev[4], st));
        hipLaunchKernelGGL(k_part_partition<2>, dim3((u32)(chunks2 < persistent ? chunks2 : persistent)), dim3(kPartBlock), lds_stage, st, (const Ack*)b1, N, G, S, shift, nb, pb, b2);
        CK(hipEventRecord(ev[5], st));
        hipLaunchKernelGGL(k_part_apply, dim3(tiles), dim3(kPartApplyBlock), lds_cells, st, m_b, G, S, shift, (const Ack*)b2, pb);
        CK(hipEventRecord(ev[6], st));
        hipLaunchKernelGGL(k_scatter_acks_flagged, dim3((unsigned)((N + kBlock - 1) / kBlock < kPartFlaggedGrid ? (N + kBlock - 1) / kBlock : kPartFlaggedGrid)), dim3(kBlock), 0, st, m_b, G, G, S, (const Ack*)d, N, (const u32*)pb.flag);
        CK(hipEventRecord(ev[7], st));
        CK(hipStreamSynchronize(st));
        CK(hipGetLastError());
        CK(hipMemcpy(&flag, pb.flag, 4, hipMemcpyDeviceToHost));
        if (rep >= 2)
            for (int k = 0; k < 7; ++k) {
                CK(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
                t[k] += ms * 1e3 / (R - 2);
            }
    }
    std::vector<u64> a((size_t)S * G), b((size_t)S * G);
    CK(hipMemcpy(a.data(), m_a, a.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), m_b, b.size() * 8, hipMemcpyDeviceToHost));
    const bool same = memcmp(a.data(), b.data(), a.size() * 8) == 0;
    const double part = t[1] + t[2] + t[3] + t[4] + t[5] + t[6];
    printf("%llu acks (skew %.1f), %llu groups x %u slots, tiles of %u groups, %u buckets, chunks of %d records, skew flag %u\n", N, zipf, G, S,
           1u << shift, nb, kPartChunk, flag);
    printf("atomic       %9.1f us\n", t[0]);
    printf("partitioned  count %.1f + scan %.1f + pass1 %.1f + pass2 %.1f + apply %.1f + flagged atomics %.1f = %9.1f us   (%.2fx)\n", t[1], t[2],
           t[3], t[4], t[5], t[6], part, t[0] / part);
    printf("planes %s\n", same ? "identical" : "DIFFER");
    return same ? 0 : 1;
}
