// Greedy hurdle-matrix aligner for WIDE bands (k up to 31, unit penalties, GLOBAL) with EXACT PRUNING: a pass looks at the
// few band lanes that can matter instead of all 2k+1, and still commits exactly what hurdle_matrix<int_128bit> commits
// (GASMA/hurdle_matrix.h:285-434,568-597).  What makes that possible:
//
//  (1) The reference's per-lane highway cache has a closed form.  A lane is recomputed when its cached starting point lies
//      before this pass's start column (`if (starting_point < start_column)`, :293), and a recompute yields the first zero of the
//      lane's flipped vector at or after the start column.  By induction the cached starting point after any number of passes is
//                 sp_j = first_zero_j( E_j ),      E_j = max over the passes so far of start_column_j(pass),
//      whichever passes recomputed it (if the cached sp is >= a later start there is no zero between them, so a recompute would
//      have found the same sp).  The cached length is a function of sp (:299-308).  The stale num_switches (:294) is
//      |j - cur_lane(r)| for r = the FIRST pass whose start column lies beyond the last zero before sp (that pass recomputed to
//      this sp and no later pass did).  So a lane's cache is a function of the pass HISTORY {(cur_lane, cur_column)} and of the
//      lane's vector alone: a lane that was ignored for several passes can be evaluated later as if it had been there all along.
//  (2) `reaching_destination` (:305-308,334) is raised by a RECOMPUTED lane whose highway runs past its destination, i.e. whose
//      new sp lies beyond the lane's last hurdle LH_j before the destination: start_j > zl_j, zl_j = the last zero below LH_j —
//      a per-lane constant.  With the minimum of zl over all lanes one comparison clears a whole pass.
//  (3) Not reaching: the score (:328-330) of a lane is at most  match_sig * length, length <= the longest run of zeros of the
//      lane's flipped vector below its destination (MR_j, a per-lane constant) and <= destination - start; the switch term only
//      lowers it.  A lane whose bound is below the best score found so far cannot win the arg-max (:345-351, which is a
//      lexicographic maximum of (score, leap, lower lane) and so does not depend on the order lanes are looked at).
//      Reaching: the integer heuristic (:339-342) is at most -(switch + final switch) + (what the highway may stick out beyond
//      the destination), again a per-lane constant.
//  (4) _choose_best_highway (:368-401) can only accept a lane whose switch costs alone fit under the best lane's cost.
//
// The per-pair logic below is plain host/device code (PR_HD): tests/ run it on the CPU against the oracle, the kernel in
// asm_greedy_prune_kernel.h runs the same functions one thread per pair.
#pragma once
#include "asm_greedy3.h"

#define PR_HD G3_HD
#define PR_MAXK 31
#define PR_MAXL (2 * PR_MAXK + 1)
#define PR_HIST 32 /* passes remembered; a pair that needs more goes to the fallback list */

// last set bit of v strictly below position x (x in [0, 128]); -1 when there is none
PR_HD int pr_prev_one(G3V v, int x) {
    g3_u64 lo = v.lo, hi = v.hi;
    if (x <= 0) return -1;
    if (x < 64) lo &= (1ull << x) - 1ull, hi = 0ull;
    else if (x < 128) hi &= x == 64 ? 0ull : ((1ull << (x - 64)) - 1ull);
#if defined(__HIP_DEVICE_COMPILE__)
    if (hi) return 127 - __clzll((long long)hi);
    if (lo) return 63 - __clzll((long long)lo);
#else
    if (hi) return 127 - __builtin_clzll(hi);
    if (lo) return 63 - __builtin_clzll(lo);
#endif
    return -1;
}

// Lane vector of band lane `lane` (hurdle_matrix.h:441-455) and its flipped form
PR_HD void pr_lane_vectors(G3V A0, G3V A1, G3V B0, G3V B1, int lane, G3V& lo, G3V& lf) {
    const int a = lane < 0 ? -lane : lane;
    if (lane < 0) {
        const G3V x0 = g3_toward0(A0, a), x1 = g3_toward0(A1, a);
        lo.lo = (x0.lo ^ B0.lo) | (x1.lo ^ B1.lo), lo.hi = (x0.hi ^ B0.hi) | (x1.hi ^ B1.hi);
    } else {
        const G3V x0 = g3_toward0(B0, a), x1 = g3_toward0(B1, a);
        lo.lo = (x0.lo ^ A0.lo) | (x1.lo ^ A1.lo), lo.hi = (x0.hi ^ A0.hi) | (x1.hi ^ A1.hi);
    }
    lf = g3_flip1(lo);
}

// Per-lane constants of a pair, computed once by the set-up kernel
struct PrLaneInfo {
    int zl;  /* last zero of the flipped vector below its last hurdle at or before the destination; -1: none (always reaching) */
    int run; /* run class of the zeros below the destination: 0 none, 1 single zeros only, c >= 2: a run of >= 2^(c-1) (up to 6: >= 32) */
};
#define PR_RUN_CLASSES 7
PR_HD int pr_run_bound(int cls) { return cls == 0 ? 0 : (cls >= PR_RUN_CLASSES - 1 ? 128 : (1 << cls) - 1); } /* longest run <= */

PR_HD PrLaneInfo pr_lane_info(G3V lf, int dst) {
    PrLaneInfo f;
    const int dstc = dst > 0 ? dst : 0;
    // last hurdle at a position <= dst, and the last zero below it
    const int lh = pr_prev_one(lf, dstc + 1 > 128 ? 128 : dstc + 1);
    G3V nz;
    nz.lo = ~lf.lo, nz.hi = ~lf.hi;
    f.zl = lh < 0 ? -1 : pr_prev_one(nz, lh);
    /* lane 0 of a 128/128 pair without any hurdle: the reference's test is sp + 128 > 128 (no further hurdle counts as length
     * 128), which a highway starting at column 0 does not pass */
    if (lh < 0 && dst == 128) f.zl = 0;
    // runs of zeros inside [0, dst): of >= 2, 4, 8, 16, 32 by doubling
    G3V x = nz;
    if (dstc < 64) x.lo &= dstc == 0 ? 0ull : ((1ull << dstc) - 1ull), x.hi = 0ull;
    else if (dstc < 128) x.hi &= dstc == 64 ? 0ull : ((1ull << (dstc - 64)) - 1ull);
    int cls = (x.lo | x.hi) ? 1 : 0;
#pragma unroll
    for (int s = 1; s <= 16; s <<= 1) {
        G3V y;
        y.lo = x.lo & ((x.lo >> s) | (x.hi << (64 - s))), y.hi = x.hi & (x.hi >> s);
        x = y;
        if (x.lo | x.hi) cls++;
    }
    f.run = cls;
    return f;
}

// Per-pair summaries (what the set-up kernel leaves besides one zl byte per lane)
struct PrPairInfo {
    unsigned long long runs[PR_RUN_CLASSES - 1]; /* bit t of runs[c-1]: band lane t - K has run class >= c (c = 1..6) */
    unsigned char zl_lane[4];   /* the four lanes (t = lane + K) with the smallest zl, ascending by zl */
    signed char zl_val[4];      /* their zl */
    int zl_next;                /* the fifth smallest zl (127 when the band has fewer lanes) */
    int pad_;
};

// lanes of run class >= c (selects instead of an indexed load: the struct lives in registers)
PR_HD unsigned long long pr_runs(const PrPairInfo& pi, int c) {
    unsigned long long r = 0ull;
#pragma unroll
    for (int q = 1; q < PR_RUN_CLASSES; q++) r = c == q ? pi.runs[q - 1] : r;
    return r;
}

// A pair's state in registers; its pass history {(cur_lane, cur_col)} lives with the lane constants in the `Lanes` store
// (device: LDS columns), PR_HIST entries of cur_lane + 64 | cur_col << 8.
struct PrPair {
    int K, m, n, dest_lane;
    int zlmin;        /* min over the band's lanes of zl: no lane can raise reaching_destination while cur_col + |cur_lane| <= zlmin */
    PrPairInfo pi;    /* the set-up kernel's summaries */
    int np;           /* index of the current pass = passes committed so far */
    int cl, cc;       /* current lane and column */
    int cmin, cmax;   /* hull of the current lanes of all passes so far (incl. this one) */
    int cost;
    bool finished, overflow;
};

PR_HD uint32_t pr_hist_pack(int cl, int cc) { return (uint32_t)(cl + 64) | ((uint32_t)cc << 8); }
PR_HD int pr_hist_cl(uint32_t w) { return (int)(w & 255u) - 64; }
PR_HD int pr_hist_cc(uint32_t w) { return (int)(w >> 8); }

// E_j before the current pass: the largest start column of lane j over the passes before it (-1 for the first pass).
// Columns grow from pass to pass and a start column is at most |cur_lane| beyond its pass's column, so the walk back stops early.
template <class Lanes>
PR_HD int pr_e_before(const PrPair& s, const Lanes& lanes, int lane) {
    int e = -1;
    const int reach = s.cmax > -s.cmin ? s.cmax : -s.cmin;
    for (int q = s.np - 1; q >= 0; q--) {
        const uint32_t w = lanes.hist(q);
        const int cc = pr_hist_cc(w);
        if (cc + reach <= e) break;
        const int st = cc + g3_fwd(pr_hist_cl(w), lane);
        e = st > e ? st : e;
    }
    return e;
}

struct PrEval {
    int sp, en, nh, nsw;
    bool need;
};
// The reference's cache entry of lane j as the current pass sees it (after its own update of the lane): closed form (1) above
template <class Lanes>
PR_HD PrEval pr_eval_lane(const PrPair& st, const Lanes& lanes, int lane, G3V lo, G3V lf, int dst) {
    PrEval r;
    const int s = st.cc + g3_fwd(st.cl, lane), e = pr_e_before(st, lanes, lane);
    const int en0 = e > s ? e : s;
    G3V nz;
    nz.lo = ~lf.lo, nz.hi = ~lf.hi;
    const uint32_t spn = g3_umax(g3_umin(g3_next_one(nz, (uint32_t)en0), 128u), (uint32_t)en0);
    const uint32_t a1 = g3_next_one(lf, spn);
    const uint32_t dstc = (uint32_t)(dst > 0 ? dst : 0);
    const uint32_t lim = g3_umax(dstc, spn);
    r.sp = (int)spn;
    r.en = (int)g3_umin(a1, lim);
    r.nh = g3_ones_from(lo, (uint32_t)s) - g3_ones_from(lo, (uint32_t)r.en);
    const int sw = lane > st.cl ? lane - st.cl : st.cl - lane;
    int z = -1;
    if (e < 0) {
        r.need = true;
    } else if (e >= s) {
        r.need = false;
        z = pr_prev_one(nz, r.sp > 128 ? 128 : r.sp);
    } else {
        z = pr_prev_one(nz, s > 128 ? 128 : s);
        r.need = z >= e;
    }
    r.nsw = sw;
    if (!r.need) { /* the first pass whose start column lay beyond the last zero before sp recomputed the lane to this sp */
        for (int q = 0; q < st.np; q++) {
            const uint32_t w = lanes.hist(q);
            if (pr_hist_cc(w) + g3_fwd(pr_hist_cl(w), lane) > z) {
                const int c = pr_hist_cl(w);
                r.nsw = lane > c ? lane - c : c - lane;
                break;
            }
        }
    }
    return r;
}

// `need` alone (is lane j recomputed in the current pass?), for the reaching test
template <class Lanes>
PR_HD bool pr_need(const PrPair& st, const Lanes& lanes, int lane, G3V lf) {
    const int s = st.cc + g3_fwd(st.cl, lane), e = pr_e_before(st, lanes, lane);
    if (e < 0) return true;
    if (e >= s) return false;
    G3V nz;
    nz.lo = ~lf.lo, nz.hi = ~lf.hi;
    return pr_prev_one(nz, s > 128 ? 128 : s) >= e;
}

struct PrStats { /* work counters (tests and tools; the kernel passes nullptr) */
    long passes, evals, need_tests, ub_tests, cand_evals, raise_scans, reach_passes, reach_evals;
};

// `Lanes` supplies: get(lane, lo, lf), info(lane), hist(q) / set_hist(q, w), cache_get(slot) / cache_put(slot, w).
template <class Lanes>
PR_HD void pr_begin(PrPair& s, int K, uint32_t lens, const PrPairInfo& pi, Lanes& lanes) {
    int m = (int)(lens & 0xffffu), n = (int)(lens >> 16);
    m = m > 128 ? 128 : m, n = n > 128 ? 128 : n; /* hurdle_matrix.h:626-627 */
    s.K = K, s.m = m, s.n = n, s.dest_lane = n - m;
    s.zlmin = (int)pi.zl_val[0], s.pi = pi, s.np = 0, s.cl = 0, s.cc = 0, s.cmin = 0, s.cmax = 0, s.cost = 0, s.finished = false, s.overflow = false;
    lanes.set_hist(0, pr_hist_pack(0, 0));
}

struct PrStep { /* what a commit emits (CIGAR) */
    int from_lane, to_lane, run;
    bool committed;
};

// One pass of run()'s loop with pruning; same commits as g3_pass / the reference.
template <class Lanes>
PR_HD PrStep pr_pass(PrPair& s, const G3Sig& sig, Lanes& lanes, PrStats* st) {
    const int K = s.K;
    const int cl = s.cl, cc = s.cc;
    const int acl = cl < 0 ? -cl : cl;
    PrStep out;
    out.committed = false, out.from_lane = cl, out.to_lane = cl, out.run = 0;
    if (st) st->passes++;
    // ---- (2) reaching_destination: some recomputed lane's highway runs past its destination, i.e. its start column lies beyond
    // its zl.  The four lanes with the smallest zl are tried first; if none of them raises it and the fifth smallest zl is out of
    // reach, nobody does.  (Rarely: every lane, one by one.) ----
    bool reaching = false;
    if (cc + acl > s.zlmin) {
        if (st) st->raise_scans++;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int j = (int)s.pi.zl_lane[q] - K;
            if (!reaching && j <= K && cc + g3_fwd(cl, j) > (int)s.pi.zl_val[q]) {
                G3V lo, lf;
                lanes.get(j, lo, lf);
                if (st) st->need_tests++;
                reaching = pr_need(s, lanes, j, lf);
            }
        }
        if (!reaching && cc + acl > s.pi.zl_next) {
            for (int j = -K; j <= K && !reaching; j++) {
                if (cc + g3_fwd(cl, j) > lanes.zl(j)) {
                    G3V lo, lf;
                    lanes.get(j, lo, lf);
                    if (st) st->need_tests++;
                    reaching = pr_need(s, lanes, j, lf);
                }
            }
        }
    }
    if (st) st->reach_passes += reaching;
    // ---- (3) arg-max of (heuristic, leap, lower lane) over the lanes that can hold it ----
    double best_h = -__builtin_inf();
    int best_leap = 0, best = 0, bsp = 0, ben = 0, bcost = 0;
    bool have = false;
    uint32_t cached = 0u; /* lanes cl-3 .. cl+3 evaluated here, for (4) */
    auto look_at = [&](int j) { /* the lane's exact cache entry and score; keeps the lexicographic maximum */
        const int dst = g3_dest(s.m, s.n, j);
        const int sw = j > cl ? j - cl : cl - j;
        G3V lo, lf;
        lanes.get(j, lo, lf);
        const PrEval ev = pr_eval_lane(s, lanes, j, lo, lf, dst);
        if (st) st->evals++, st->reach_evals += reaching;
        if (sw <= 3) {
            lanes.cache_put(j - cl + 3, (uint32_t)ev.sp | ((uint32_t)ev.en << 8) | ((uint32_t)ev.nh << 16));
            cached |= 1u << (j - cl + 3);
        }
        const int len = ev.en - ev.sp;
        double heur;
        int leap = -sw;
        if (reaching) {
            const int fsw = j > s.dest_lane ? j - s.dest_lane : s.dest_lane - j;
            heur = (double)(-(sw + ev.nh) - fsw - (dst - ev.en));
            leap -= fsw;
        } else {
#if defined(__HIP_DEVICE_COMPILE__)
            heur = __fma_rn(sig.indel, (double)ev.nsw, __fma_rn(sig.mismatch, (double)ev.nh, __dmul_rn(sig.match, (double)len)));
#else
            heur = __builtin_fma(sig.indel, (double)ev.nsw, __builtin_fma(sig.mismatch, (double)ev.nh, sig.match * (double)len));
#endif
        }
        const bool better = !have || heur > best_h || (heur == best_h && (leap > best_leap || (leap == best_leap && j < best)));
        if (better) best_h = heur, best_leap = leap, best = j, bsp = ev.sp, ben = ev.en, bcost = sw + ev.nh, have = true;
    };
    look_at(cl);
    unsigned long long done = 1ull << (cl + K);
    auto look_at_mask = [&](unsigned long long todo, int run_bound) { /* not reaching: lanes of one run class */
        while (todo) {
#if defined(__HIP_DEVICE_COMPILE__)
            const int t = __ffsll((long long)todo) - 1;
#else
            const int t = __builtin_ctzll(todo);
#endif
            todo &= todo - 1ull;
            const int j = t - K;
            /* the lane's own bound: its length is also at most what is left up to its destination, and the best has grown */
            const int room0 = g3_dest(s.m, s.n, j) - (cc + g3_fwd(cl, j));
            const int room = room0 > 0 ? room0 : 0;
            const int len_ub = run_bound < room ? run_bound : room;
            const int hd = j < s.cmin ? s.cmin - j : (j > s.cmax ? j - s.cmax : 0);
#if defined(__HIP_DEVICE_COMPILE__)
            const double ub = __fma_rn(sig.indel, (double)hd, __dmul_rn(sig.match, (double)len_ub));
#else
            const double ub = __builtin_fma(sig.indel, (double)hd, sig.match * (double)len_ub);
#endif
            if (st) st->ub_tests++;
            if (ub < best_h) continue;
            done |= 1ull << t;
            look_at(j);
        }
    };
    auto span_mask = [&](int lo_lane, int hi_lane) { /* bits of the band lanes lo_lane .. hi_lane */
        int a = lo_lane + K, b = hi_lane + K;
        a = a < 0 ? 0 : a, b = b > 2 * K ? 2 * K : b;
        if (a > b) return 0ull;
        const unsigned long long upto_b = b >= 63 ? ~0ull : ((1ull << (b + 1)) - 1ull);
        return upto_b & ~((1ull << a) - 1ull);
    };
    if (!reaching) {
        /* score = fma(indel, switches, fma(mismatch, hurdles, match * length)) with mismatch, indel <= 0, and the roundings are
         * monotone: a lane's score is at most fma(indel, d, match * L) where L bounds its length (the lane's run class) and d is
         * its distance from the hull of all current lanes so far (every pass's |lane - cur_lane| is at least that).  Class by
         * class, longest runs first: the lanes of a class within the distance at which the bound still reaches the best so far. */
        for (int c = PR_RUN_CLASSES - 1; c >= 1; c--) {
            const double top = sig.match * (double)pr_run_bound(c);
            if (top < best_h) break; /* no lane of this class, nor of a lower one, can reach it */
            int r = 0; /* the largest distance at which a lane of this class could still reach the best */
            while (r < 2 * K) {
#if defined(__HIP_DEVICE_COMPILE__)
                const double ub = __fma_rn(sig.indel, (double)(r + 1), __dmul_rn(sig.match, (double)pr_run_bound(c)));
#else
                const double ub = __builtin_fma(sig.indel, (double)(r + 1), sig.match * (double)pr_run_bound(c));
#endif
                if (ub < best_h) break;
                r++;
            }
            const unsigned long long cls = pr_runs(s.pi, c) & ~pr_runs(s.pi, c + 1); /* exactly class c */
            look_at_mask(cls & span_mask(s.cmin - r, s.cmax + r) & ~done, pr_run_bound(c));
        }
    } else {
        /* heuristic = -(switch + hurdles) - final switch - (destination - end), end = sp + length.  The highway starts at the first
         * zero at or after E' = max(E, start); the columns from E' up to there are hurdles of the flipped vector and so of the
         * original one, all inside the counted range: hurdles >= sp - E'.  If the highway starts at or before the destination
         * its end does not pass it (:305-308): heuristic <= -(switch + final switch).  If it starts beyond, length is 0 and
         * heuristic <= -(switch + final switch) - (sp - E') + (sp - destination), E' <= cur_col + (largest |cur_lane| so far).
         * With B = -best: a lane can reach the best only if |j - cur| + |j - dest| - max(0, Emax - destination_j) <= B, which
         * confines j to an interval (destination_j = min(m + min(j, 0), n - max(j, 0))). */
        const int reachmax = s.cmax > -s.cmin ? s.cmax : -s.cmin;
        const int emax = cc + reachmax;
        for (int round = 0; round < 2 * K + 2; round++) {
            /* best_h is an integer-valued double here.  Lanes whose destination is just m (or n) — those between lane 0 and the
             * destination lane — gain emax - min(m, n) when the columns have run past it: widen the whole interval by that */
            const int mn = s.m < s.n ? s.m : s.n;
            const int B = (int)(-best_h) + (emax > mn ? emax - mn : 0);
            const int sum = cl + s.dest_lane;
            int lo_l = sum - B >= 0 ? (sum - B + 1) >> 1 : -((B - sum) >> 1);   /* ceil((sum - B) / 2) */
            int hi_l = sum + B >= 0 ? (sum + B) >> 1 : -((-(sum + B) + 1) >> 1); /* floor((sum + B) / 2) */
            const int far_r = B + sum - s.n + emax, far_l = sum + s.m - emax - B;
            hi_l = far_r > hi_l ? far_r : hi_l;
            lo_l = far_l < lo_l ? far_l : lo_l;
            const unsigned long long todo = span_mask(lo_l, hi_l) & ~done;
            if (st) st->ub_tests++;
            if (!todo) break;
            /* nearest to the current lane first: it is the likeliest to raise the bar */
            unsigned long long below = todo & ((1ull << (cl + K)) - 1ull), above = todo & ~((1ull << (cl + K)) - 1ull);
            int t;
            if (above && (!below || (
#if defined(__HIP_DEVICE_COMPILE__)
                             (__ffsll((long long)above) - 1) - (cl + K) <= (cl + K) - (63 - __clzll((long long)below))
#else
                             __builtin_ctzll(above) - (cl + K) <= (cl + K) - (63 - __builtin_clzll(below))
#endif
                                 )))
#if defined(__HIP_DEVICE_COMPILE__)
                t = __ffsll((long long)above) - 1;
#else
                t = __builtin_ctzll(above);
#endif
            else
#if defined(__HIP_DEVICE_COMPILE__)
                t = 63 - __clzll((long long)below);
#else
                t = 63 - __builtin_clzll(below);
#endif
            done |= 1ull << t;
            look_at(t - K);
        }
    }
    if (ben - bsp <= 0) { /* hurdle_matrix.h:358-361 */
        s.finished = true;
        return out;
    }
    // ---- (4) _choose_best_highway over the lanes whose switch costs alone fit under the best lane's cost ----
    G3V blo, blf;
    lanes.get(best, blo, blf);
    const int bfs = g3_ones_from(blo, (uint32_t)bsp);
    int small_total = bcost, small_inter = bcost, ch = best, ch_en = ben;
    /* sw + pen >= |cl - best|, and every lane outside [min, max] of the two adds twice its distance */
    const int lo_l = cl < best ? cl : best, hi_l = cl < best ? best : cl;
    const int span = hi_l - lo_l;
    if (span <= bcost) {
        const int ext = (bcost - span) >> 1;
        int j0 = lo_l - ext, j1 = hi_l + ext;
        j0 = j0 < -K ? -K : j0, j1 = j1 > K ? K : j1;
        for (int j = j0; j <= j1; j++) {
            if (j == best) continue;
            const int sw = j > cl ? j - cl : cl - j;
            const int pen = j > best ? j - best : best - j;
            int sp, en, nh;
            const int slot = j - cl + 3;
            if (slot >= 0 && slot < 7 && ((cached >> slot) & 1u)) {
                const uint32_t w = lanes.cache_get(slot);
                sp = (int)(w & 255u), en = (int)((w >> 8) & 255u), nh = (int)(w >> 16);
            } else {
                G3V lo, lf;
                lanes.get(j, lo, lf);
                const PrEval ev = pr_eval_lane(s, lanes, j, lo, lf, g3_dest(s.m, s.n, j));
                if (st) st->cand_evals++;
                sp = ev.sp, en = ev.en, nh = ev.nh;
            }
            const int f2 = g3_fwd(j, best);
            if (sp + f2 > bsp) continue; /* :376-377 */
            const int inter = sw + nh;
            const int from = f2 + en;
            const bool ok = (unsigned)from < 128u && (unsigned)(bsp - 1 - from) < 128u; /* utils.h:263-270 */
            const int tail = ok ? g3_ones_from(blo, (uint32_t)from) - bfs : 0;
            const int total = inter + pen + tail;
            if (total <= small_total && inter <= small_inter) small_total = total, small_inter = inter, ch = j, ch_en = en; /* :395 */
        }
    }
    // ---- commit (:411-433) ----
    out.committed = true, out.to_lane = ch, out.run = ch_en - (cc + g3_fwd(cl, ch));
    s.cost += small_inter;
    s.cl = ch, s.cc = ch_en;
    if (ch_en >= g3_dest(s.m, s.n, ch)) {
        s.finished = true;
        return out;
    }
    if (s.np + 1 >= PR_HIST) {
        s.overflow = true, s.finished = true;
        return out;
    }
    s.np++;
    s.cmin = ch < s.cmin ? ch : s.cmin, s.cmax = ch > s.cmax ? ch : s.cmax;
    lanes.set_hist(s.np, pr_hist_pack(ch, ch_en));
    return out;
}
