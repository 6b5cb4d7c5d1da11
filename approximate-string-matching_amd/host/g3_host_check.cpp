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
