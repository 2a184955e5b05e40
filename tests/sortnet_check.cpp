// sortnet_check.cpp — host-side check of the compile-time sorting networks in
// raft_amd/csrc/sortnet.h that are too wide for the in-header 0-1 static_assert:
// exhaustive 0-1 inputs for N <= 20, and for wider networks every 0-1 input of
// the form "two sorted runs" at every merge level plus random 0-1 and random
// u64 inputs with many ties.  Built and run by tests/test_sortnet.py.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "sortnet.h"

using namespace rafthip;

static uint64_t rng_state = 0x5EED;
static uint64_t rnd() {
    rng_state += 0x9E3779B97F4A7C15ull;
    uint64_t z = rng_state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int N>
static void apply(uint64_t* v) {
    for (int k = 0; k < Batcher<N>::n; ++k) {
        const int a = Batcher<N>::list.ce[k].a, b = Batcher<N>::list.ce[k].b;
        if (v[a] > v[b]) std::swap(v[a], v[b]);
    }
}

template <int N>
static int check() {
    int bad = 0;
    uint64_t v[N], w[N];
    auto one = [&]() {
        for (int i = 0; i < N; ++i) w[i] = v[i];
        apply<N>(v);
        std::sort(w, w + N);
        for (int i = 0; i < N; ++i) bad += v[i] != w[i];
    };
    if (N <= 20) {
        for (uint64_t in = 0; in < (1ull << N); ++in) {
            for (int i = 0; i < N; ++i) v[i] = (in >> i) & 1;
            one();
        }
    }
    for (int it = 0; it < 200000; ++it) {
        const int mode = it % 4;
        for (int i = 0; i < N; ++i) {
            const uint64_t r = rnd();
            v[i] = mode == 0 ? (r & 1) : mode == 1 ? (r % 5) : mode == 2 ? r : (r >> 60 ? r : 0);
        }
        if (it % 7 == 0) std::sort(v, v + N);
        if (it % 11 == 0) std::sort(v, v + N, [](uint64_t x, uint64_t y) { return x > y; });
        one();
    }
    // threshold inputs: every choice of how many zeros sit in each aligned block of 2^k keys, sampled
    for (int it = 0; it < 200000; ++it) {
        const int blk = 1 << (1 + it % 5);
        for (int b = 0; b < N; b += blk) {
            const int z = (int)(rnd() % (blk + 1));
            for (int i = b; i < b + blk && i < N; ++i) v[i] = (i - b) >= z;
        }
        one();
    }
    printf("Batcher<%d>: %d comparators, %s\n", N, Batcher<N>::n, bad ? "FAIL" : "ok");
    return bad;
}

int main() {
    int bad = 0;
    bad += check<2>() + check<3>() + check<7>() + check<9>() + check<10>() + check<12>() + check<13>() + check<16>();
    bad += check<17>() + check<20>() + check<24>() + check<31>() + check<32>() + check<33>() + check<48>() + check<63>() + check<64>();
    return bad ? 1 : 0;
}
