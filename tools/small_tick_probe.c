/* Latency of a small leader tick through the C ABI from plain C (what a cgo caller sees; tools/small_tick_probe.py
 * measures the same calls through Python / ctypes): n serialized MsgAppResp in ordinary malloc'ed memory ->
 * rafthip_step_wire (messages, status, counts and action bytes back) -> rafthip_commit_advance (pairs back); the same
 * acks as records -> rafthip_scatter_acks.  2^20 groups x 5 slots, median of 200 calls, microseconds.
 *
 *   gcc -O2 -std=c11 -Iinclude tools/small_tick_probe.c -Lraft_amd/lib -lrafthip -Wl,-rpath,$PWD/raft_amd/lib \
 *       -Wl,-rpath-link,/opt/rocm/lib -o tools/bin/small_tick_probe
 *   RAFTHIP_NO_SMALL_ARENA=1 tools/bin/small_tick_probe     # every pageable buffer through its own copy command */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rafthip.h"

#define G (1u << 20)
#define S 5u
#define ROUNDS 220
#define SKIP 20

#define OK(call)                                                                          \
    do {                                                                                  \
        if ((call) != 0) {                                                                \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #call, rafthip_last_error()); \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

static uint64_t seed = 0x5EED000Cull;
static uint64_t rnd(void) { /* splitmix64 */
    uint64_t z = (seed += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double now_us(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return 1e6 * (double)t.tv_sec + 1e-3 * (double)t.tv_nsec;
}
static int cmp_double(const void *a, const void *b) {
    const double x = *(const double *)a, y = *(const double *)b;
    return x < y ? -1 : x > y;
}
static double median(double *v, int n) {
    qsort(v, (size_t)n, sizeof *v, cmp_double);
    return v[n / 2];
}
static size_t put_varint(uint8_t *d, uint64_t v) {
    size_t n = 0;
    while (v >= 0x80) d[n++] = (uint8_t)(v | 0x80), v >>= 7;
    d[n++] = (uint8_t)v;
    return n;
}
/* Message.Marshal of a non-reject MsgAppResp (raftpb/raft.pb.go:937-1025): every scalar in field order */
static size_t put_app_resp(uint8_t *d, uint64_t to, uint64_t from, uint64_t term, uint64_t index) {
    size_t n = 0;
    const uint64_t f[6] = {4, to, from, term, 0, index};
    for (int k = 0; k < 6; k++) d[n++] = (uint8_t)((k + 1) << 3), n += put_varint(d + n, f[k]);
    d[n++] = 0x40, d[n++] = 0;
    d[n++] = 0x50, d[n++] = 0;
    d[n++] = 0x58, d[n++] = 0;
    d[n++] = 0x68, d[n++] = 0;
    return n;
}

int main(void) {
    rafthip_config cfg = {sizeof cfg, 0, G, S, 0, NULL};
    rafthip_engine *e = NULL;
    OK(rafthip_create(&cfg, &e));
    OK(rafthip_set_uniform_config(e, S));
    OK(rafthip_progress_config(e, 4, 0));
    uint64_t *plane = malloc((size_t)S * G * 8), *term = malloc((size_t)G * 8), *zero = calloc(G, 8);
    uint8_t *state = malloc((size_t)S * G), *zero8 = calloc(G, 1);
    uint64_t *match = calloc((size_t)S * G, 8);
    for (size_t i = 0; i < (size_t)S * G; i++) plane[i] = 1, state[i] = RAFTHIP_STATE_REPLICATE;
    OK(rafthip_upload_match(e, 0, G, match, G));
    OK(rafthip_upload_progress(e, 0, G, state, plane, NULL, G)); /* Next = 1 */
    for (size_t s = 0; s < S; s++)
        for (size_t g = 0; g < G; g++) plane[s * G + g] = (s + 1) + (g << 8);
    OK(rafthip_upload_ids(e, 0, G, plane, G));
    for (size_t g = 0; g < G; g++) term[g] = 5, plane[g] = ~0ull;
    OK(rafthip_upload_term(e, 0, G, term));
    OK(rafthip_upload_read_ctx(e, 0, G, zero, zero8));
    OK(rafthip_upload_term_range(e, 0, G, zero, plane)); /* every index is of the current term */

    const size_t sizes[] = {100, 1000, 10000, 30000, 100000};
    const size_t nmax = 100000;
    uint8_t *bytes = malloc(nmax * 64), *status = malloc(nmax), *actions = malloc(nmax);
    uint64_t *offsets = malloc((nmax + 1) * 8);
    uint32_t *groups = malloc(nmax * 4);
    rafthip_msg *msgs = malloc(nmax * sizeof *msgs);
    rafthip_ack *acks = malloc(nmax * sizeof *acks);
    rafthip_commit_pair *pairs = malloc((size_t)G * sizeof *pairs);
    uint64_t counts[RAFTHIP_WIRE_STATUSES], bump = 0;
    static double t_step[ROUNDS], t_commit[ROUNDS], t_scatter[ROUNDS];
    printf("{");
    for (size_t k = 0; k < sizeof sizes / sizeof sizes[0]; k++) {
        const size_t n = sizes[k];
        uint64_t applied = 0, advanced = 0;
        for (int it = 0; it < ROUNDS; it++) {
            size_t nb = 0;
            bump += 64;
            for (size_t i = 0; i < n; i++) {
                const uint32_t g = (uint32_t)(rnd() % G), s = (uint32_t)(rnd() % S);
                offsets[i] = nb, groups[i] = g;
                nb += put_app_resp(bytes + nb, 9, (s + 1) + ((uint64_t)g << 8), 5, bump + (rnd() & 31));
                acks[i].group = g, acks[i].slot = s, acks[i].index = bump + 32 + (rnd() & 31);
            }
            offsets[n] = nb;
            const double t0 = now_us();
            OK(rafthip_step_wire(e, bytes, offsets, groups, n, msgs, status, counts, actions));
            const double t1 = now_us();
            uint64_t adv = 0;
            OK(rafthip_commit_advance(e, pairs, G, &adv));
            const double t2 = now_us();
            OK(rafthip_scatter_acks(e, acks, n));
            const double t3 = now_us();
            if (it >= SKIP) {
                t_step[it - SKIP] = t1 - t0, t_commit[it - SKIP] = t2 - t1, t_scatter[it - SKIP] = t3 - t2;
                applied += counts[RAFTHIP_WIRE_APPLIED], advanced += adv;
            }
        }
        if (applied != (uint64_t)(ROUNDS - SKIP) * n) {
            fprintf(stderr, "n = %zu: %llu of %llu responses applied\n", n, (unsigned long long)applied,
                    (unsigned long long)((ROUNDS - SKIP) * n));
            return 1;
        }
        printf("%s\"%zu\": {\"step_wire_us\": %.1f, \"commit_advance_us\": %.1f, \"scatter_acks_us\": %.1f, "
               "\"advanced_per_call\": %.0f}", k ? ", " : "", n, median(t_step, ROUNDS - SKIP),
               median(t_commit, ROUNDS - SKIP), median(t_scatter, ROUNDS - SKIP), (double)advanced / (ROUNDS - SKIP));
        fflush(stdout);
    }
    printf("}\n");
    OK(rafthip_destroy(e));
    return 0;
}
