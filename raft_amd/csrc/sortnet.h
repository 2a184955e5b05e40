// sortnet.h — sorting networks used in registers by the lane-per-group commit
// kernels: minimal-size networks for 1..8 keys (SortNet), and Batcher's
// odd-even merge sort pruned to N keys for 9..64 (Batcher).  The reference sorts the <=7 voter indexes
// with an insertion sort on the stack (quorum/majority.go:115-122,141-147); a
// fixed compare-exchange network gives the same sorted order with no
// data-dependent branches, which is what a 64-wide wavefront wants.
//
// Each network is checked at compile time with the 0-1 principle (a network
// sorts every input iff it sorts every 0/1 input), so a typo cannot ship.
#pragma once
#include <stdint.h>

namespace rafthip {

struct CE {
    uint8_t a, b;  // compare-exchange: afterwards v[a] <= v[b]
};

template <int N>
struct SortNet;

template <>
struct SortNet<1> {
    static constexpr int n = 0;
    static constexpr CE ce[1] = {{0, 0}};
};
template <>
struct SortNet<2> {
    static constexpr int n = 1;
    static constexpr CE ce[1] = {{0, 1}};
};
template <>
struct SortNet<3> {
    static constexpr int n = 3;
    static constexpr CE ce[3] = {{0, 2}, {0, 1}, {1, 2}};
};
template <>
struct SortNet<4> {
    static constexpr int n = 5;
    static constexpr CE ce[5] = {{0, 2}, {1, 3}, {0, 1}, {2, 3}, {1, 2}};
};
template <>
struct SortNet<5> {
    static constexpr int n = 9;
    static constexpr CE ce[9] = {{0, 3}, {1, 4}, {0, 2}, {1, 3}, {0, 1}, {2, 4}, {1, 2}, {3, 4}, {2, 3}};
};
template <>
struct SortNet<6> {
    static constexpr int n = 12;
    static constexpr CE ce[12] = {{0, 5}, {1, 3}, {2, 4}, {1, 2}, {3, 4}, {0, 3},
                                  {2, 5}, {0, 1}, {2, 3}, {4, 5}, {1, 2}, {3, 4}};
};
template <>
struct SortNet<7> {
    static constexpr int n = 16;
    static constexpr CE ce[16] = {{0, 6}, {2, 3}, {4, 5}, {0, 2}, {1, 4}, {3, 6}, {0, 1}, {2, 5},
                                  {3, 4}, {1, 2}, {4, 6}, {2, 3}, {4, 5}, {1, 2}, {3, 4}, {5, 6}};
};
template <>
struct SortNet<8> {
    static constexpr int n = 19;
    static constexpr CE ce[19] = {{0, 2}, {1, 3}, {4, 6}, {5, 7}, {0, 4}, {1, 5}, {2, 6},
                                  {3, 7}, {0, 1}, {2, 3}, {4, 5}, {6, 7}, {2, 4}, {3, 5},
                                  {1, 4}, {3, 6}, {1, 2}, {3, 4}, {5, 6}};
};

template <int N>
constexpr bool sortnet_sorts_all_01() {
    for (unsigned in = 0; in < (1u << N); ++in) {
        unsigned v = in;  // bit i = key i
        for (int k = 0; k < SortNet<N>::n; ++k) {
            const unsigned a = SortNet<N>::ce[k].a, b = SortNet<N>::ce[k].b;
            const unsigned va = (v >> a) & 1u, vb = (v >> b) & 1u;
            if (va > vb) v ^= (1u << a) | (1u << b);
        }
        for (int i = 0; i + 1 < N; ++i)
            if (((v >> i) & 1u) > ((v >> (i + 1)) & 1u)) return false;
    }
    return true;
}

// ---------------------------------------------------------------------------
// Batcher's odd-even merge sort for P = 2^k keys (Knuth 5.2.2, merge exchange),
// pruned to the comparators that touch only keys < N: imagine keys N..P-1 as
// +infinity — a compare-exchange never moves them, so dropping those
// comparators sorts the first N keys all the same.  63 / 191 / 543
// comparators at N = 16 / 32 / 64.  Generated at compile time; the 0-1 check
// below covers N <= 12 (2^N inputs), tests/sortnet_check.cpp the rest.
// ---------------------------------------------------------------------------
template <int N>
struct Batcher {
    static constexpr int P = N <= 2 ? 2 : N <= 4 ? 4 : N <= 8 ? 8 : N <= 16 ? 16 : N <= 32 ? 32 : 64;
    static_assert(N >= 2 && N <= 64, "Batcher network: 2..64 keys");
    struct List {
        CE ce[543];
        int n;
    };
    static constexpr List make() {
        List l{};
        l.n = 0;
        for (int p = 1; p < P; p <<= 1)
            for (int k = p; k >= 1; k >>= 1)
                for (int j = k % p; j + k < P; j += 2 * k)
                    for (int i = 0; i < k && i + j + k < P; ++i)
                        if ((i + j) / (2 * p) == (i + j + k) / (2 * p)) {
                            const int a = i + j, b = i + j + k;
                            if (b < N) l.ce[l.n++] = CE{(uint8_t)a, (uint8_t)b};
                        }
        return l;
    }
    static constexpr List list = make();
    static constexpr int n = list.n;
};

template <int N>
constexpr bool batcher_sorts_all_01() {
    for (unsigned in = 0; in < (1u << N); ++in) {
        unsigned v = in;
        for (int k = 0; k < Batcher<N>::n; ++k) {
            const unsigned a = Batcher<N>::list.ce[k].a, b = Batcher<N>::list.ce[k].b;
            const unsigned va = (v >> a) & 1u, vb = (v >> b) & 1u;
            if (va > vb) v ^= (1u << a) | (1u << b);
        }
        for (int i = 0; i + 1 < N; ++i)
            if (((v >> i) & 1u) > ((v >> (i + 1)) & 1u)) return false;
    }
    return true;
}
static_assert(Batcher<16>::n == 63 && Batcher<32>::n == 191 && Batcher<64>::n == 543, "Batcher comparator counts");
static_assert(batcher_sorts_all_01<5>() && batcher_sorts_all_01<8>() && batcher_sorts_all_01<9>() &&
                  batcher_sorts_all_01<12>(),
              "pruned Batcher networks");

static_assert(sortnet_sorts_all_01<1>(), "sorting network 1");
static_assert(sortnet_sorts_all_01<2>(), "sorting network 2");
static_assert(sortnet_sorts_all_01<3>(), "sorting network 3");
static_assert(sortnet_sorts_all_01<4>(), "sorting network 4");
static_assert(sortnet_sorts_all_01<5>(), "sorting network 5");
static_assert(sortnet_sorts_all_01<6>(), "sorting network 6");
static_assert(sortnet_sorts_all_01<7>(), "sorting network 7");
static_assert(sortnet_sorts_all_01<8>(), "sorting network 8");

}  // namespace rafthip
