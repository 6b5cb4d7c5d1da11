// Greedy hurdle-matrix aligner, narrow band (K <= 3), unit penalties, GLOBAL mode — the benchmark's configuration
// (benchmark.cpp:22: k = 3, x = o = e = 1) — as a STRAIGHT-LINE, branch-free pass.
//
// Same algorithm and the same per-pair results as greedy_persist_kernel<K, true> (asm_kernels.h), which follows
// hurdle_matrix<int_128bit>::_update_highway_list / _choose_best_highway / _step / run (GASMA/hurdle_matrix.h:285-434,
// 568-597); what changes is how a pass is evaluated:
//
//  * no FP64 on the vector ALU.  The reference ranks the band lanes by the log-odds score
//        fma(indel_sig, num_switches, fma(mismatch_sig, num_hurdles, match_sig * length))        (hurdle_matrix.h:328-330)
//    and only the ORDER (and the exact ties) of those doubles matters.  With the reference's default probabilities
//    mismatch_sig == indel_sig bit for bit, so the score of (length, hurdles, switches) lies within a few ulp of
//    match_sig*length + mismatch_sig*(hurdles + switches): the host evaluates every double of the domain with the same
//    three roundings, checks that the classes (length, hurdles + switches) do not interleave, and hands the kernel a table
//    of  class rank | rank inside the class for every split  (g3_build_table below; 66 KB, staged into LDS).  A lane's key
//    is then  rank << 6 | leap << 3 | lane  in ONE dword and the arg-max of hurdle_matrix.h:345-351 (larger heuristic, then
//    larger leap, then the lower lane) is three v_max3_u32.  The `reaching_destination` branch (:334-343) is integer
//    already and gets a key of the same shape.  Scores outside the table's domain (hurdles + switches >= 64: a lane that is
//    one solid block of mismatches) take a slow path that evaluates the doubles as before.
//  * no divergent branches inside a pass: every lane is looked up unconditionally and the cached highway is kept with
//    selects (the cache is what the reference computes — `if (starting_point < start_column)` :293 — including its stale
//    num_switches and the reaching flag that only recomputed lanes may raise).
//  * the lane vector of the winning lane (needed by _choose_best_highway's tail count, :392) is read from a copy of the
//    lane vectors in LDS with the lane index as address instead of a seven-way select over 4 x 7 registers.
//
// The header compiles for the host as well (G3_HD): tests/ and tools/ run the very same pass on the CPU against the oracle.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define G3_HD __host__ __device__ __forceinline__
#else
#define G3_HD inline
struct uint2 { /* the host-only build (g++, tests) has no HIP vector types */
    unsigned int x, y;
};
static inline uint2 make_uint2(unsigned int x, unsigned int y) {
    uint2 r;
    r.x = x, r.y = y;
    return r;
}
#endif

typedef unsigned long long g3_u64;
#if defined(G3_MARKERS) && defined(__HIP_DEVICE_COMPILE__)
#define G3_MARK(name) asm volatile("; MARK " name)
#else
#define G3_MARK(name)
#endif

#ifndef G3_TMAX
#define G3_TMAX 32  /* table domain: hurdles + switches < G3_TMAX; beyond it the FP64 path of the same kernel */
#endif
#define G3_LENS 129 /* highway lengths 0..128 */
#define G3_TABLE_ENTRIES (G3_LENS * G3_TMAX)
#define G3_INF 0xffffffffu

struct G3V {
    g3_u64 lo, hi;
};

// ---- small primitives (device: one or two instructions each; host: the plain meaning) -----------------------------
G3_HD uint32_t g3_ffbl(uint32_t x) { /* index of the lowest set bit, 0xffffffff when none (v_ffbl_b32) */
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r; /* the instruction's own "none" value is what the scans want; __ffs / __builtin_ctz wrap it in selects */
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
#else
    return x ? (uint32_t)__builtin_ctz(x) : G3_INF;
#endif
}
G3_HD uint32_t g3_ctz64(g3_u64 y) { /* 0xffffffff when y == 0 */
    const uint32_t a = g3_ffbl((uint32_t)y), b = g3_ffbl((uint32_t)(y >> 32)) | 32u;
    return a < b ? a : b;
}
G3_HD uint32_t g3_addsat(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_elementwise_add_sat(a, b);
#else
    const uint32_t s = a + b;
    return s < a ? G3_INF : s;
#endif
}
G3_HD uint32_t g3_umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
G3_HD uint32_t g3_umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
G3_HD int g3_popc64(g3_u64 y) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(y);
#else
    return __builtin_popcountll(y);
#endif
}
G3_HD uint32_t g3_absdiff(uint32_t a, uint32_t b) { /* |a - b| for a, b < 65536: one v_sad_u16 (the upper halves are zero) */
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sad_u16(a, b, 0u);
#else
    return a > b ? a - b : b - a;
#endif
}
G3_HD uint32_t g3_bfe(uint32_t v, uint32_t off, uint32_t width) { /* v_bfe_u32 */
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(v, off, width);
#else
    return (v >> off) & ((1u << width) - 1u);
#endif
}

// first set bit of v at or after `from`, 0xffffffff (or any value >= 128) when there is none.  from in [0, 191].
// Only the word `from` falls into is shifted (the hardware takes 64-bit shift counts mod 64); a hit in the lower word wins
// over the upper word's first bit.
// For from >= 128 the result is some value >= from (the upper word shifted by from mod 64): callers that clamp or compare
// against positions <= 128 need no guard.
G3_HD uint32_t g3_next_one(G3V v, uint32_t from) {
    const bool low = from < 64u;
    const g3_u64 w = low ? v.lo : v.hi;
    const g3_u64 y = w >> (from & 63u);
    const uint32_t p = g3_addsat(from, g3_ctz64(y));
    const uint32_t h = g3_addsat(64u, g3_ctz64(v.hi));
    return g3_umin(p, low ? h : G3_INF);
}

// ones of v at index >= s; 0 from 128 on
G3_HD int g3_ones_from(G3V v, uint32_t s) {
    const bool low = s < 64u;
    const g3_u64 w = low ? v.lo : v.hi;
    const g3_u64 y = w >> (s & 63u);
    const int c = g3_popc64(y) + (low ? g3_popc64(v.hi) : 0);
    return s >= 128u ? 0 : c;
}

G3_HD G3V g3_toward0_const(G3V v, int s) { /* utils.h:143-153 for a compile-time 0 <= s < 64 */
    G3V r;
    r.lo = s ? ((v.lo >> s) | (v.hi << (64 - s))) : v.lo;
    r.hi = v.hi >> s;
    return r;
}
G3_HD G3V g3_toward0(G3V v, int s) { /* any s; 0 outside [0,127] */
    G3V r;
    const int q = s & 63;
    const g3_u64 lo1 = (v.lo >> q) | ((v.hi << (63 - q)) << 1), hi1 = v.hi >> q;
    const bool big = (s & 64) != 0, dead = (unsigned)s >= 128u;
    r.lo = dead ? 0ull : (big ? hi1 : lo1);
    r.hi = (dead || big) ? 0ull : hi1;
    return r;
}
G3_HD G3V g3_flip1(G3V v) { /* utils.h:200-216, threshold 1: a set bit survives only next to another one */
    G3V r;
    r.lo = v.lo & ((v.lo >> 1) | (v.hi << 63) | (v.lo << 1));
    r.hi = v.hi & ((v.hi >> 1) | (v.hi << 1) | (v.lo >> 63));
    return r;
}

// utils.h:587-593
constexpr int g3_fwd(int l1, int l2) {
    const int a1 = l1 < 0 ? -l1 : l1, a2 = l2 < 0 ? -l2 : l2;
    return (l1 * l2 >= 0) ? (a1 > a2 ? a1 - a2 : 0) : a1;
}
// 2 bits per value of the variable lane: tab_from_cur[j] bits [2c+1:2c] = fwd(c-K, j-K); tab_to_best[j] = fwd(j-K, b-K)
template <int K>
constexpr uint32_t g3_tab_from_cur(int j) {
    uint32_t t = 0;
    for (int c = 0; c < 2 * K + 1; c++) t |= (uint32_t)g3_fwd(c - K, j - K) << (2 * c);
    return t;
}
template <int K>
constexpr uint32_t g3_tab_to_best(int j) {
    uint32_t t = 0;
    for (int b = 0; b < 2 * K + 1; b++) t |= (uint32_t)g3_fwd(j - K, b - K) << (2 * b);
    return t;
}

// hurdle_matrix.h:58-68 in closed form: columns of lane l end where either string does
G3_HD int g3_dest(int m, int n, int lane) {
    const int a = m + (lane < 0 ? lane : 0), b = n - (lane > 0 ? lane : 0);
    return a < b ? a : b;
}

struct G3Sig {
    double match, mismatch, indel; /* hurdle_matrix.h:536-538, from the host's libm */
};

// Per-pair state of one thread.  `Store` is where the copy of the lane vectors lives that is addressed by lane index
// (LDS on the device, a plain array on the host): store.put(j, v), store.get(j).
template <int K>
struct G3State {
    static constexpr int NL = 2 * K + 1;
    G3V lo[NL], lf[NL];
    int sp[NL], en[NL], nsw[NL], dst[NL];
    int m, n, dest_lane, cur_lane, cur_col, cost, guard;
    bool finished;
};

template <int K, class Store>
G3_HD void g3_setup(G3State<K>& s, G3V A0, G3V A1, G3V B0, G3V B1, uint32_t lens, Store& store) {
    constexpr int NL = 2 * K + 1;
    int m = (int)(lens & 0xffffu), n = (int)(lens >> 16);
    m = m > 128 ? 128 : m; /* hurdle_matrix.h:626-627 */
    n = n > 128 ? 128 : n;
    s.m = m, s.n = n, s.dest_lane = n - m; /* :649 */
#pragma unroll
    for (int j = 0; j < NL; j++) {
        const int lane = j - K;
        G3V x0, x1, v;
        if (lane < 0) { /* hurdle_matrix.h:441-455 */
            x0 = g3_toward0_const(A0, -lane), x1 = g3_toward0_const(A1, -lane);
            v.lo = (x0.lo ^ B0.lo) | (x1.lo ^ B1.lo), v.hi = (x0.hi ^ B0.hi) | (x1.hi ^ B1.hi);
        } else {
            x0 = g3_toward0_const(B0, lane), x1 = g3_toward0_const(B1, lane);
            v.lo = (x0.lo ^ A0.lo) | (x1.lo ^ A1.lo), v.hi = (x0.hi ^ A0.hi) | (x1.hi ^ A1.hi);
        }
        s.lo[j] = v;
        s.lf[j] = g3_flip1(v);
        store.put(j, v);
        s.sp[j] = -1; /* :106-119 */
        s.en[j] = -1;
        s.nsw[j] = 0;
        s.dst[j] = g3_dest(m, n, lane);
    }
    s.cur_lane = 0, s.cur_col = 0, s.cost = 0, s.guard = 0;
    s.finished = false;
}

// What a commit of _step emits (the CIGAR sink of the kernels consumes it; unused otherwise)
struct G3Step {
    int from_lane, to_lane, run;
    bool committed;
};

// One pass of run()'s loop (hurdle_matrix.h:568-574): _update_highway_list, _choose_best_highway, commit.
// table[len * G3_TMAX + T] = { class rank << 9, rank inside the class for num_switches = 0..6 at 4 bits each }.
template <int K, class Store, class Table>
G3_HD G3Step g3_pass(G3State<K>& s, const Table& table, const G3Sig& sig, Store& store, bool* took_slow_path = nullptr) {
    constexpr int NL = 2 * K + 1;
    constexpr int LB = 3; /* bits of the lane field of a key (NL <= 7) */
    G3Step out;
    out.committed = false, out.from_lane = s.cur_lane, out.to_lane = s.cur_lane, out.run = 0;
    const uint32_t ci = (uint32_t)(s.cur_lane + K), sh2 = 2u * ci;
    const uint32_t cc = (uint32_t)s.cur_col;

    G3_MARK("update_begin");
    // ---- _update_highway_list, first lane loop (:291-322) ----
    bool reaching = false;
    int sw[NL], inter[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) {
        const uint32_t start = cc + g3_bfe(g3_tab_from_cur<K>(j), sh2, 2u);
        sw[j] = (int)g3_absdiff(ci, (uint32_t)j); /* switch_lane_penalty with o = e = 1 (utils.h:576-579) */
        const bool need = s.sp[j] < (int)start; /* :293 */
        // next highway at or after `start`: first zero of the flipped vector, then the first one after it (:299-303)
        G3V nz;
        nz.lo = ~s.lf[j].lo, nz.hi = ~s.lf[j].hi;
        // no zero at or after start (or start >= 128: the shifted vector is empty, first_zero = 0): the highway "starts" at
        // max(128, start) with length 0
        const uint32_t spn = g3_umax(g3_umin(g3_next_one(nz, start), 128u), start);
        // first one at or after the highway's start; some value >= max(128, spn) when the highway runs to the end of the vector
        const uint32_t a1 = g3_next_one(s.lf[j], spn);
        // a lane's destination is negative when a string is shorter than the lane's offset: every highway then "reaches" it
        // with length 0 (the reference compares signed); a1 > max(dst, 0) is the same test, a1 being > 0 always
        const uint32_t dstc = (uint32_t)(s.dst[j] > 0 ? s.dst[j] : 0);
        const uint32_t lim = g3_umax(dstc, spn);
        const uint32_t e_new = g3_umin(a1, lim); /* :305-308: length clipped at the lane's destination, never negative */
        bool over = a1 > dstc;
        if (j == K) /* only lane 0 of a 128/128 pair has destination 128; there the reference's test is sp + length > 128
                       with length = 128 for "no further hurdle": false for sp = 0, true when no zero was found (sp = 128) */
            over = (over && !(spn == 0u && s.dst[j] == 128)) || spn >= 128u;
        reaching = reaching || (need && over);
        s.sp[j] = need ? (int)spn : s.sp[j];
        s.en[j] = need ? (int)e_new : s.en[j];
        s.nsw[j] = need ? 4 * sw[j] : s.nsw[j]; /* :294: refreshed only on recompute; kept times 4 (the table's nibble index) */
        // :318-320 pop_count_between(start, sp + len): end >= start always, both counts vanish from 128 on
        const int nh = g3_ones_from(s.lo[j], start) - g3_ones_from(s.lo[j], (uint32_t)s.en[j]);
        inter[j] = sw[j] + nh;
    }

    G3_MARK("argmax_begin");
    // ---- second lane loop (:325-352): arg-max by (heuristic, leap, lower lane) ----
    uint32_t key[NL];
    bool slow = false;
#pragma unroll
    for (int j = 0; j < NL; j++) {
        const int nh = inter[j] - sw[j];
        const uint32_t lanebits = (uint32_t)(NL - 1 - j);
        // reaching: heuristic = -(switch + hurdles) - final switch - (destination - end), leap = -(switch + final switch)
        const int fsw = (int)g3_absdiff((uint32_t)(s.dest_lane + 256), (uint32_t)(j - K + 256));
        const int cost_r = inter[j] + fsw + (s.dst[j] - s.en[j]);
        const uint32_t key_r = ((uint32_t)(1024 - cost_r) << 11) | ((uint32_t)(255 - (sw[j] + fsw)) << LB) | lanebits;
        // not reaching: rank of the significance score from the table
        const int len = s.en[j] - s.sp[j];
        const uint32_t T = (uint32_t)nh + ((uint32_t)s.nsw[j] >> 2);
        slow = slow || (T >= (uint32_t)G3_TMAX);
        const uint2 ent = table.get((uint32_t)len * G3_TMAX + T); /* beyond the table for T >= G3_TMAX: `slow` discards what is read */
        const uint32_t sub = g3_bfe(ent.y, (uint32_t)s.nsw[j], 4u);
        const uint32_t key_n = ent.x | (sub << 6) | ((uint32_t)(2 * K - sw[j]) << LB) | lanebits;
        key[j] = reaching ? key_r : key_n;
    }
    uint32_t kmax = key[0];
#pragma unroll
    for (int j = 1; j < NL; j++) kmax = g3_umax(kmax, key[j]);
    uint32_t bj = (uint32_t)(NL - 1) - (kmax & 7u);
    G3_MARK("slow_begin");
    if (slow && !reaching) { /* outside the table: the doubles themselves (hurdle_matrix.h:328-330,345-351) */
        if (took_slow_path) *took_slow_path = true;
        double best_h = -__builtin_inf();
        int best_leap = 0;
        bj = K;
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int nh = inter[j] - sw[j], len = s.en[j] - s.sp[j];
#if defined(__HIP_DEVICE_COMPILE__)
            const double heur = __fma_rn(sig.indel, (double)(s.nsw[j] >> 2), __fma_rn(sig.mismatch, (double)nh, __dmul_rn(sig.match, (double)len)));
#else
            const double heur = __builtin_fma(sig.indel, (double)(s.nsw[j] >> 2), __builtin_fma(sig.mismatch, (double)nh, sig.match * (double)len));
#endif
            const int leap = -sw[j];
            if (heur > best_h || (heur == best_h && leap > best_leap)) best_h = heur, best_leap = leap, bj = (uint32_t)j;
        }
    }
    G3_MARK("winner_begin");
    int bsp = s.sp[0], ben = s.en[0], bcost = inter[0];
#pragma unroll
    for (int j = 1; j < NL; j++) {
        const bool is = bj == (uint32_t)j;
        bsp = is ? s.sp[j] : bsp, ben = is ? s.en[j] : ben, bcost = is ? inter[j] : bcost;
    }
    s.guard++;
    if (ben - bsp <= 0 || s.guard > 4 * 128) { /* :358-361 */
        s.finished = true;
        return out;
    }

    G3_MARK("choose_begin");
    // ---- _choose_best_highway (:368-401) ----
    const G3V bv = store.get((int)bj);
    const int bfs = g3_ones_from(bv, (uint32_t)bsp);
    const uint32_t shb = 2u * bj;
    int small_total = bcost, small_inter = bcost, ch = (int)bj, ch_sp = bsp, ch_en = ben;
#pragma unroll
    for (int j = 0; j < NL; j++) {
        const uint32_t f2 = g3_bfe(g3_tab_to_best<K>(j), shb, 2u);
        const bool c1 = bj != (uint32_t)j && s.sp[j] + (int)f2 <= bsp; /* :376-377 */
        const uint32_t from = f2 + (uint32_t)s.en[j];
        // :392 pop_count_between(from, starting_point(best)): 0 unless 0 <= from <= 127 and from < to <= from + 128
        const bool ok = from < 128u && (uint32_t)(bsp - 1 - (int)from) < 128u;
        const int tail = ok ? g3_ones_from(bv, from) - bfs : 0;
        const int total = inter[j] + (int)g3_absdiff(bj, (uint32_t)j) + tail;
        const bool acc = c1 && total <= small_total && inter[j] <= small_inter; /* :395, both <=: later lanes win ties */
        small_total = acc ? total : small_total;
        small_inter = acc ? inter[j] : small_inter;
        ch = acc ? j : ch, ch_sp = acc ? s.sp[j] : ch_sp, ch_en = acc ? s.en[j] : ch_en;
    }
    G3_MARK("commit_begin");
    // ---- _step commit (:411-433) ----
    const int ch_lane = ch - K;
    out.committed = true;
    out.to_lane = ch_lane;
    out.run = ch_en - (s.cur_col + g3_fwd(s.cur_lane, ch_lane));
    s.cost += small_inter; /* = switch_cost + hurdle_cost of the chosen lane (x = 1) */
    s.cur_lane = ch_lane;
    s.cur_col = ch_en;
    if (ch_en >= g3_dest(s.m, s.n, ch_lane)) s.finished = true;
    return out;
}

// ---- host: the rank table -----------------------------------------------------------------------------------------
#if 1
#include <algorithm>
#include <vector>
// Evaluates fma(indel, nsw, fma(mismatch, nh, match * len)) — the three roundings of the reference build —
// for every (len, nh, nsw) with nh + nsw < G3_TMAX, nsw <= 2K, and turns the doubles into integer ranks that compare (and
// tie) exactly as the doubles do.  Returns false when the scores do not have the structure the kernel's key relies on
// (mismatch_sig != indel_sig, or classes (len, nh + nsw) that interleave): the caller then keeps the FP64 kernel.
// This translation unit must be compiled with -ffp-contract=off (it is: the explicit fma() calls are the only fusions).
inline bool g3_build_table(const G3Sig& sig, int K, std::vector<uint2>& out) {
    if (K < 1 || K > 3) return false;
    if (!(sig.mismatch == sig.indel)) return false;
    const int NS = 2 * K + 1;
    struct Cls {
        double lo, hi;
        int len, T;
    };
    std::vector<Cls> cls;
    cls.reserve(G3_TABLE_ENTRIES);
    auto score = [&](int len, int nh, int nsw) {
        return __builtin_fma(sig.indel, (double)nsw, __builtin_fma(sig.mismatch, (double)nh, sig.match * (double)len));
    };
    for (int len = 0; len < G3_LENS; len++)
        for (int T = 0; T < G3_TMAX; T++) {
            Cls c;
            c.len = len, c.T = T;
            c.lo = __builtin_inf(), c.hi = -__builtin_inf();
            for (int nsw = 0; nsw < NS && nsw <= T; nsw++) {
                const double v = score(len, T - nsw, nsw);
                if (!(v == v)) return false;
                c.lo = v < c.lo ? v : c.lo, c.hi = v > c.hi ? v : c.hi;
            }
            cls.push_back(c);
        }
    std::sort(cls.begin(), cls.end(), [](const Cls& a, const Cls& b) { return a.lo < b.lo; });
    for (size_t i = 1; i < cls.size(); i++)
        if (!(cls[i - 1].hi < cls[i].lo)) return false; /* classes must be strictly separated */
    if (cls.size() > (1u << 14)) return false;
    out.assign(G3_TABLE_ENTRIES, make_uint2(0u, 0u));
    for (size_t r = 0; r < cls.size(); r++) {
        const Cls& c = cls[r];
        double vals[8];
        int cnt = 0;
        for (int nsw = 0; nsw < NS && nsw <= c.T; nsw++) vals[cnt++] = score(c.len, c.T - nsw, nsw);
        uint32_t nib = 0;
        for (int a = 0; a < cnt; a++) {
            int rank = 0; /* dense rank inside the class: number of distinct smaller values */
            for (int b = 0; b < cnt; b++) {
                bool smaller = vals[b] < vals[a], seen = false;
                for (int q = 0; q < b; q++) seen = seen || vals[q] == vals[b];
                if (smaller && !seen) rank++;
            }
            nib |= (uint32_t)rank << (4 * a);
        }
        out[(size_t)c.len * G3_TMAX + c.T] = make_uint2((uint32_t)r << 9, nib);
    }
    return true;
}
#endif
