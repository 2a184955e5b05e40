// sortnet.h — minimal-size sorting networks for 1..8 keys, used in registers by
// the lane-per-group commit kernels.  The reference sorts the <=7 voter indexes
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

static_assert(sortnet_sorts_all_01<1>(), "sorting network 1");
static_assert(sortnet_sorts_all_01<2>(), "sorting network 2");
static_assert(sortnet_sorts_all_01<3>(), "sorting network 3");
static_assert(sortnet_sorts_all_01<4>(), "sorting network 4");
static_assert(sortnet_sorts_all_01<5>(), "sorting network 5");
static_assert(sortnet_sorts_all_01<6>(), "sorting network 6");
static_assert(sortnet_sorts_all_01<7>(), "sorting network 7");
static_assert(sortnet_sorts_all_01<8>(), "sorting network 8");

}  // namespace rafthip
