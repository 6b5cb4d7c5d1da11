// Host build of the straight-line Greedy pass (csrc/asm_greedy3.h) for the CPU test-suite: the very code the fast kernel
// runs per thread — g3_setup, g3_pass, the rank table — driven pair by pair on the host, so that tests/test_greedy3_host.py can
// diff it against the oracle without a GPU.  Test support only: nothing in the product links this file.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../csrc/asm_greedy3.h"

namespace {
struct HostStore {
    G3V v[8];
    void put(int j, G3V x) { v[j] = x; }
    G3V get(int j) const { return v[j]; }
};
struct HostTable {
    const uint2* t;
    uint2 get(uint32_t i) const { return t[i]; }
};

template <int K>
int run_batch(long n, const unsigned char* views, const uint32_t* lens, const double* probs, int32_t* costs, int32_t* passes,
              int64_t* slow_passes) {
    G3Sig sig = {log(probs[0] / 0.25), log(probs[1] / 0.25), log(probs[2] / 2 / 0.25)}; /* hurdle_matrix.h:536-538 */
    std::vector<uint2> tab;
    if (!g3_build_table(sig, K, tab)) return -1;
    tab.resize(tab.size() + 1024, make_uint2(0u, 0u)); /* the pass may read past the table when it takes the slow path */
    HostTable T{tab.data()};
    int64_t slow = 0;
    for (long i = 0; i < n; i++) {
        G3V P[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
        for (int s = 0; s < 2; s++)
            for (int q = 0; q < 128; q++) { /* bit_convert.cpp:340-355: exactly 'C','G','T' set bits */
                const unsigned char c = views[i * 256 + s * 128 + q];
                const int code = c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 0;
                if (code & 1) (q < 64 ? P[2 * s].lo : P[2 * s].hi) |= 1ull << (q & 63);
                if (code & 2) (q < 64 ? P[2 * s + 1].lo : P[2 * s + 1].hi) |= 1ull << (q & 63);
            }
        HostStore st;
        G3State<K> S;
        g3_setup<K>(S, P[0], P[1], P[2], P[3], lens[i], st);
        int np = 0;
        while (!S.finished) {
            bool was_slow = false;
            g3_pass<K>(S, T, sig, st, &was_slow);
            slow += was_slow;
            np++;
        }
        // final hop (hurdle_matrix.h:575-590), as the kernel's refill block does it
        const int dest_col = g3_dest(S.m, S.n, S.dest_lane);
        int cost = S.cost;
        if (S.cur_lane != S.dest_lane || S.cur_col < dest_col) {
            G3V dv;
            if (S.dest_lane >= -K && S.dest_lane <= K) {
                dv = st.get(S.dest_lane + K);
            } else {
                const int a = S.dest_lane < 0 ? -S.dest_lane : S.dest_lane;
                if (S.dest_lane < 0) {
                    const G3V x0 = g3_toward0(P[0], a), x1 = g3_toward0(P[1], a);
                    dv.lo = (x0.lo ^ P[2].lo) | (x1.lo ^ P[3].lo), dv.hi = (x0.hi ^ P[2].hi) | (x1.hi ^ P[3].hi);
                } else {
                    const G3V x0 = g3_toward0(P[2], a), x1 = g3_toward0(P[3], a);
                    dv.lo = (x0.lo ^ P[0].lo) | (x1.lo ^ P[1].lo), dv.hi = (x0.hi ^ P[0].hi) | (x1.hi ^ P[1].hi);
                }
            }
            const int d = S.cur_lane - S.dest_lane;
            const int from = S.cur_col + g3_fwd(S.cur_lane, S.dest_lane);
            const bool ok = (unsigned)from < 128u && (unsigned)(dest_col - from - 1) < 128u;
            const int dist = ok ? g3_ones_from(dv, (uint32_t)from) - g3_ones_from(dv, (uint32_t)dest_col) : 0;
            cost += (d < 0 ? -d : d) + dist;
        }
        costs[i] = cost;
        if (passes) passes[i] = np;
    }
    if (slow_passes) *slow_passes = slow;
    return 0;
}
}  // namespace

extern "C" int g3_host_batch(long n, const unsigned char* views /* n x (A[128], B[128]) as the conversion sees them */,
                             const uint32_t* lens /* m | n << 16 */, int K, const double* probs, int32_t* costs, int32_t* passes,
                             int64_t* slow_passes) {
    switch (K) {
        case 1: return run_batch<1>(n, views, lens, probs, costs, passes, slow_passes);
        case 2: return run_batch<2>(n, views, lens, probs, costs, passes, slow_passes);
        case 3: return run_batch<3>(n, views, lens, probs, costs, passes, slow_passes);
    }
    return -2;
}

/* 1 when the rank table can be built for these probabilities (the fast kernel is used), 0 when not */
extern "C" int g3_host_table_ok(const double* probs, int K) {
    G3Sig sig = {log(probs[0] / 0.25), log(probs[1] / 0.25), log(probs[2] / 2 / 0.25)};
    std::vector<uint2> tab;
    return g3_build_table(sig, K, tab) ? 1 : 0;
}

// ---- the pruned wide-band pass (csrc/asm_greedy_prune.h) on the host ------------------------------------------------------
#include "../csrc/asm_greedy_prune.h"

namespace {
struct HostLanes {
    G3V lo[PR_MAXL], lf[PR_MAXL];
    uint32_t hst[PR_HIST], cache[8];
    signed char zls[PR_MAXL];
    int K;
    void get(int lane, G3V& a, G3V& b) const { a = lo[lane + K], b = lf[lane + K]; }
    int zl(int lane) const { return (int)zls[lane + K]; }
    uint32_t hist(int q) const { return hst[q]; }
    void set_hist(int q, uint32_t w) { hst[q] = w; }
    uint32_t cache_get(int slot) const { return cache[slot]; }
    void cache_put(int slot, uint32_t w) { cache[slot] = w; }
};
}  // namespace

extern "C" int pr_host_batch(long n, const unsigned char* views, const uint32_t* lens, int K, const double* probs, int32_t* costs,
                             int32_t* passes, int64_t* counters /* [9]: passes, evals, need tests, ub tests, candidate evals, raise scans,
                                                                    overflow pairs, reaching passes, evals in them */) {
    if (K < 1 || K > PR_MAXK) return -2;
    G3Sig sig = {log(probs[0] / 0.25), log(probs[1] / 0.25), log(probs[2] / 2 / 0.25)};
    if (!(sig.mismatch <= 0.0 && sig.indel <= 0.0 && sig.match >= 0.0)) return -1; /* the bounds need these signs */
    PrStats st = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t overflow = 0;
    for (long i = 0; i < n; i++) {
        G3V P[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
        for (int s = 0; s < 2; s++)
            for (int q = 0; q < 128; q++) {
                const unsigned char c = views[i * 256 + s * 128 + q];
                const int code = c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 0;
                if (code & 1) (q < 64 ? P[2 * s].lo : P[2 * s].hi) |= 1ull << (q & 63);
                if (code & 2) (q < 64 ? P[2 * s + 1].lo : P[2 * s + 1].hi) |= 1ull << (q & 63);
            }
        int m = (int)(lens[i] & 0xffffu), nn = (int)(lens[i] >> 16);
        m = m > 128 ? 128 : m, nn = nn > 128 ? 128 : nn;
        HostLanes L;
        L.K = K;
        PrPairInfo pi;
        memset(&pi, 0, sizeof pi);
        unsigned long long taken = 0ull;
        for (int j = -K; j <= K; j++) { /* what the set-up kernel computes, one band lane per thread */
            pr_lane_vectors(P[0], P[1], P[2], P[3], j, L.lo[j + K], L.lf[j + K]);
            const PrLaneInfo f = pr_lane_info(L.lf[j + K], g3_dest(m, nn, j));
            L.zls[j + K] = (signed char)f.zl;
            for (int c = 1; c <= f.run; c++) pi.runs[c - 1] |= 1ull << (j + K);
        }
        for (int q = 0; q < 5; q++) { /* the four smallest zl (ties: the lower lane), and the fifth smallest value */
            int best_t = -1;
            for (int t = 0; t <= 2 * K; t++)
                if (!((taken >> t) & 1ull) && (best_t < 0 || L.zls[t] < L.zls[best_t])) best_t = t;
            if (q < 4) {
                pi.zl_lane[q] = (unsigned char)(best_t < 0 ? 127 : best_t);
                pi.zl_val[q] = (signed char)(best_t < 0 ? 127 : L.zls[best_t]);
            } else {
                pi.zl_next = best_t < 0 ? 127 : L.zls[best_t];
            }
            if (best_t >= 0) taken |= 1ull << best_t;
        }
        PrPair S;
        pr_begin(S, K, lens[i], pi, L);
        int np = 0;
        while (!S.finished) {
            pr_pass(S, sig, L, &st);
            np++;
        }
        overflow += S.overflow;
        const int dest_col = g3_dest(S.m, S.n, S.dest_lane);
        int cost = S.cost;
        if (S.cl != S.dest_lane || S.cc < dest_col) { /* final hop, hurdle_matrix.h:575-590 */
            G3V dv, dflip;
            pr_lane_vectors(P[0], P[1], P[2], P[3], S.dest_lane, dv, dflip);
            const int d = S.cl - S.dest_lane;
            const int from = S.cc + g3_fwd(S.cl, S.dest_lane);
            const bool ok = (unsigned)from < 128u && (unsigned)(dest_col - from - 1) < 128u;
            const int dist = ok ? g3_ones_from(dv, (uint32_t)from) - g3_ones_from(dv, (uint32_t)dest_col) : 0;
            cost += (d < 0 ? -d : d) + dist;
        }
        costs[i] = S.overflow ? INT32_MIN : cost;
        if (passes) passes[i] = np;
    }
    if (counters) {
        counters[0] = st.passes, counters[1] = st.evals, counters[2] = st.need_tests, counters[3] = st.ub_tests;
        counters[4] = st.cand_evals, counters[5] = st.raise_scans, counters[6] = overflow, counters[7] = st.reach_passes,
        counters[8] = st.reach_evals;
    }
    return 0;
}
