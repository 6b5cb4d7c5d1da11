// Greedy for medium and wide bands, T THREADS PER PAIR (T = 2, 4, 8 or 16 consecutive lanes of a wave), each thread holding
// LPT <= 8 consecutive band lanes in registers: 64 / T pairs per wavefront.
//
// Why: the thread-per-pair kernel (greedy_persist_kernel<K>) keeps all 2k+1 lane vectors of a pair in one thread's
// registers, which stops at k = 5 (11 lanes, 200 VGPRs); the wave-per-pair kernel (greedy_wave_kernel) gives every band lane
// a thread, so at k = 8 it works with 17 of 64 threads and the cost per pair does not depend on k at all (0.9 ms per 10^6
// pairs from k = 6 to k = 31 against 0.19 ms at k = 5, profiles/r01_dispatch_grid_c2.txt).  Here the band is cut into T
// slices of LPT lanes with T * LPT >= 2k + 1: e.g. k = 8 as T = 8, LPT = 3 (8 pairs per wave), k = 36 as T = 16, LPT = 5.
// (What it buys, measured, is at the end of this file: less than hoped.)
// The group agrees on the best lane with DPP butterflies that stay inside T lanes (quad_perm, row_half_mirror, row_mirror:
// no LDS, no barriers), and fetches the winner's highway with ds_bpermute.
//
// Same step structure and results as greedy_kernel<K> (hurdle_matrix.h:285-434,568-597): _update_highway_list in two
// passes (the `reaching_destination` flag of pass 1 changes every lane's score in pass 2), _choose_best_highway as an
// ordered fold over the few lanes that can still be accepted, _step commit, final hop.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "asm_bits.h"
#include "asm_kernels.h"

#define DPP_QUAD_XOR1 0xB1        /* quad_perm [1,0,3,2] */
#define DPP_QUAD_XOR2 0x4E        /* quad_perm [2,3,0,1] */
#define DPP_ROW_HALF_MIRROR 0x141 /* lane i <-> 7 - i inside each half row of 8 */
#define DPP_ROW_MIRROR 0x140      /* lane i <-> 15 - i inside each row of 16 */

// max over the T lanes of a group, returned in every lane of the group
template <int T>
ASM_DEV unsigned group_max_u32(unsigned v) {
#define GSTEP(ctrl)                                                                                   \
    {                                                                                                 \
        const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xf, 0xf, false);   \
        v = o > v ? o : v;                                                                            \
    }
    if (T >= 2) GSTEP(DPP_QUAD_XOR1)
    if (T >= 4) GSTEP(DPP_QUAD_XOR2)
    if (T >= 8) GSTEP(DPP_ROW_HALF_MIRROR)
    if (T >= 16) GSTEP(DPP_ROW_MIRROR)
#undef GSTEP
    return v;
}

// value of lane `src` (0..63, any lane of the wave) — the LDS crossbar, no LDS memory
ASM_DEV int lane_fetch(int v, int src) { return __builtin_amdgcn_ds_bpermute(src << 2, v); }

template <int T, int LPT>
__global__ __launch_bounds__(ASM_BLOCK, (LPT <= 3 ? 4 : LPT <= 5 ? 3 : 2)) void greedy_group_kernel(const uint4* __restrict__ planes,
                                                                 const uint32_t* __restrict__ lens, long n, int w4, int k,
                                                                 GreedyArgs args, OutMap out, CigarSink cig) {
    static_assert(T == 2 || T == 4 || T == 8 || T == 16, "a group is 2, 4, 8 or 16 lanes of one DPP row");
    static_assert(LPT >= 1 && LPT <= 8, "band lanes per thread");
    const int t = threadIdx.x & 63;
    const int r = t & (T - 1);        /* member of the group */
    const int gbase = t & ~(T - 1);   /* first wave lane of the group */
    const long group0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) / T;
    const long ngroups = ((long)gridDim.x * blockDim.x) / T;
    const int nl = 2 * k + 1;
    const int j0 = r * LPT;           /* this thread's band lanes are j0 .. j0+LPT-1, lane = j - k */
    const int x = args.x, o = args.o, e = args.e;
    const bool semi = args.semi != 0;

    for (long i = group0; i < n; i += ngroups) {
        const uint32_t ln = lens[i];
        int m = (int)(ln & 0xffffu), nn = (int)(ln >> 16);
        m = m > 128 ? 128 : m; /* hurdle_matrix.h:626-627 */
        nn = nn > 128 ? 128 : nn;
        const int dest_lane = nn - m;
        V128 lo_[LPT]; /* the flipped vector (lanes, hurdle_matrix.h:452-453) is rebuilt from lo_ where a highway is looked up */
        int sp[LPT], len[LPT], nsw[LPT];
        {   /* the planes are only needed to build the lane vectors (and once more for the final hop: re-read there) */
            const V128 A0 = v_from_uint4(planes[((long)0 * w4) * n + i]);
            const V128 A1 = v_from_uint4(planes[((long)1 * w4) * n + i]);
            const V128 B0 = v_from_uint4(planes[((long)2 * w4) * n + i]);
            const V128 B1 = v_from_uint4(planes[((long)3 * w4) * n + i]);
#pragma unroll
            for (int q = 0; q < LPT; q++) {
                lo_[q] = greedy_lane_vector(A0, A1, B0, B1, j0 + q - k);
                sp[q] = -1, len[q] = 0, nsw[q] = 128; /* hurdle_matrix.h:106-119 */
            }
        }
        int cur_lane = 0, cur_col = 0, cost = 0, ncig = 0;
        const long pair = out.index(i);
        bool done = false;
        for (int guard = 0; guard < 4 * 128 && !done; guard++) {
            // Straight-line lane work: everything is computed for every lane slot and selected, because the refresh is needed
            // for almost every lane in almost every step anyway and a divergent branch per lane costs more (exec-mask saves,
            // waits) than the few selects.  Slots beyond the band (j >= 2k+1) compute on garbage and are masked out of the
            // flag, the arg-max and the candidate set.
            // ---- _update_highway_list, pass 1: refresh the cached highways, hurdle counts ----
            int sw[LPT], nh[LPT];
            bool reach = false;
            const bool first_free = semi && guard == 0;
#pragma unroll
            for (int q = 0; q < LPT; q++) {
                const int lane = j0 + q - k;
                const bool valid = j0 + q < nl;
                const int start_col = cur_col + fwd_col(cur_lane, lane);
                const bool stale = sp[q] < start_col;
                const int dd = lane - cur_lane;
                int fz, nx;
                v_highway_from(v_flip_short_hurdles1(lo_[q]), start_col, fz, nx);
                const int nsp = start_col + fz;
                const int room = lane_destination(m, nn, lane) - nsp;
                const bool hits = nsp + nx > nsp + room; /* start_col + fz + nx > destination */
                const int nlen = hits ? (room > 0 ? room : 0) : nx;
                sp[q] = stale ? nsp : sp[q];
                len[q] = stale ? nlen : len[q];
                nsw[q] = stale ? (dd < 0 ? -dd : dd) : nsw[q];
                reach = reach || (stale && hits && valid);
                sw[q] = first_free ? 0 : lane_penalty(cur_lane, lane, o, e);
                nh[q] = v_pop_between(lo_[q], start_col, sp[q] + len[q]);
            }
            const bool reaching = group_max_u32<T>(reach ? 1u : 0u) != 0u;
            // ---- pass 2: scores; the thread's own best lane (first lane wins exact ties, hurdle_matrix.h:345-351) ----
            double bheur = -__builtin_inf();
            int bleap = -(1 << 20), bq = -1;
#pragma unroll
            for (int q = 0; q < LPT; q++) {
                const int lane = j0 + q - k;
                const int hc = x * nh[q];
                const int fsw = semi ? 0 : lane_penalty(lane, dest_lane, o, e);
                const double h_sig = greedy_significance(args, len[q], nh[q], nsw[q]);
                const double h_dst = (double)(-sw[q] - hc - fsw - x * (lane_destination(m, nn, lane) - sp[q] - len[q]));
                const double heur = reaching ? h_dst : h_sig;
                const int leap = reaching ? -sw[q] - fsw : -sw[q];
                const bool take = (j0 + q < nl) && (bq < 0 || heur > bheur || (heur == bheur && leap > bleap));
                bheur = take ? heur : bheur, bleap = take ? leap : bleap, bq = take ? q : bq;
            }
            // ---- arg-max over the group: order-preserving integer keys, three T-lane butterflies ----
            unsigned long long bkey = 0ull;
            if (bq >= 0) {
                const unsigned long long hb = (unsigned long long)__double_as_longlong(bheur + 0.0); /* -0.0 -> +0.0: equal values, equal keys */
                bkey = (hb >> 63) ? ~hb : (hb | 0x8000000000000000ull);
            }
            bq = bq < 0 ? 0 : bq;
            const unsigned khi = (unsigned)(bkey >> 32);
            const unsigned mhi = group_max_u32<T>(khi);
            bool cand = khi == mhi;
            const unsigned klo = cand ? (unsigned)bkey : 0u;
            const unsigned mlo = group_max_u32<T>(klo);
            cand = cand && klo == mlo && bkey != 0ull;
            const unsigned k3 = cand ? ((((unsigned)(bleap + 32768)) << 7) | (unsigned)(127 - (j0 + bq))) : 0u;
            const unsigned m3 = group_max_u32<T>(k3);
            const int bj = 127 - (int)(m3 & 127u); /* winning band lane, group-uniform */
            const int best = bj - k;
            const int owner = gbase + bj / LPT;
            // the owner's best lane IS the winner: every thread offers its own best lane's highway, all read the owner's
            int o_sp = sp[0], o_len = len[0], o_cost = sw[0] + x * nh[0];
            V128 o_vec = lo_[0];
#pragma unroll
            for (int q = 1; q < LPT; q++) {
                const bool is = bq == q;
                o_sp = is ? sp[q] : o_sp, o_len = is ? len[q] : o_len, o_cost = is ? sw[q] + x * nh[q] : o_cost;
                o_vec.lo = is ? lo_[q].lo : o_vec.lo, o_vec.hi = is ? lo_[q].hi : o_vec.hi;
            }
            const int best_sp = lane_fetch(o_sp, owner), best_len = lane_fetch(o_len, owner);
            const int best_cost = lane_fetch(o_cost, owner);
            if (best_len <= 0) break; /* hurdle_matrix.h:358-361 — uniform across the group */
            const V128 best_vec =
                v_make((u64)(unsigned)lane_fetch((int)(unsigned)o_vec.lo, owner) | ((u64)(unsigned)lane_fetch((int)(o_vec.lo >> 32), owner) << 32),
                       (u64)(unsigned)lane_fetch((int)(unsigned)o_vec.hi, owner) | ((u64)(unsigned)lane_fetch((int)(o_vec.hi >> 32), owner) << 32));
            // ---- _choose_best_highway: lanes that can still be accepted (thresholds only go down from best_cost) ----
            const int best_from_sp = v_ones_from(best_vec, best_sp);
            int ti[LPT]; /* total << 16 | inter of the lanes that can still be accepted */
            unsigned cmask = 0u;
#pragma unroll
            for (int q = 0; q < LPT; q++) {
                const int lane = j0 + q - k;
                const int fb = fwd_col(lane, best);
                const int inter = sw[q] + nh[q]; /* not multiplied by x (hurdle_matrix.h:388) */
                const int tail = x * v_pop_between_pre(best_vec, fb + sp[q] + len[q], best_sp, best_from_sp);
                const int total = inter + lane_penalty(lane, best, o, e) + (tail > 0 ? tail : 0);
                const bool can = (j0 + q < nl) && lane != best && !(sp[q] + fb > best_sp) && total <= best_cost && inter <= best_cost;
                ti[q] = (total << 16) | (inter & 0xffff);
                cmask |= can ? 1u << q : 0u;
            }
            // ordered fold in ascending lane order (both comparisons are <=: later lanes win ties, hurdle_matrix.h:393)
            int small_total = best_cost, small_inter = best_cost;
            int ch = bj, ch_sp = best_sp, ch_len = best_len, ch_cost = best_cost;
            for (;;) {
                const unsigned mine = cmask ? (unsigned)(127 - (j0 + (int)__builtin_ctz(cmask))) : 0u;
                const unsigned top = group_max_u32<T>(mine);
                if (top == 0u) break; /* group-uniform */
                const int cj = 127 - (int)top, cq = cj % LPT, cown = gbase + cj / LPT;
                int c_ti = ti[0], c_sp = sp[0], c_len = len[0], c_cost = sw[0] + x * nh[0];
#pragma unroll
                for (int q = 1; q < LPT; q++) {
                    const bool is = cq == q;
                    c_ti = is ? ti[q] : c_ti, c_sp = is ? sp[q] : c_sp, c_len = is ? len[q] : c_len;
                    c_cost = is ? sw[q] + x * nh[q] : c_cost;
                }
                c_ti = lane_fetch(c_ti, cown);
                const int c_tot = c_ti >> 16, c_itr = c_ti & 0xffff;
                if (c_tot <= small_total && c_itr <= small_inter) {
                    small_total = c_tot, small_inter = c_itr, ch = cj;
                    ch_sp = lane_fetch(c_sp, cown), ch_len = lane_fetch(c_len, cown), ch_cost = lane_fetch(c_cost, cown);
                }
                if (t == cown) cmask &= cmask - 1u;
            }
            // ---- _step commit (hurdle_matrix.h:411-433) ----
            cost += ch_cost;
            const int new_lane = ch - k, new_col = ch_sp + ch_len;
            if (cig.on() && r == 0) cig.step(pair, ncig, cur_lane, new_lane, new_col - (cur_col + fwd_col(cur_lane, new_lane)));
            cur_lane = new_lane;
            cur_col = new_col;
            done = cur_col >= lane_destination(m, nn, cur_lane);
        }
        if (r == 0) {
            // ---- final hop (hurdle_matrix.h:575-590); the destination lane may lie outside the band ----
            const int dest_col = lane_destination(m, nn, dest_lane);
            if (cur_lane != dest_lane || cur_col < dest_col) {
                const V128 dv = greedy_lane_vector(v_from_uint4(planes[((long)0 * w4) * n + i]), v_from_uint4(planes[((long)1 * w4) * n + i]),
                                                   v_from_uint4(planes[((long)2 * w4) * n + i]), v_from_uint4(planes[((long)3 * w4) * n + i]),
                                                   dest_lane);
                const int sw_f = semi ? 0 : lane_penalty(cur_lane, dest_lane, o, e);
                const int distance = v_pop_between(dv, cur_col + fwd_col(cur_lane, dest_lane), dest_col);
                const int hcf = x * distance;
                cost += sw_f + (hcf > 0 ? hcf : 0);
                if (cig.on()) cig.step(pair, ncig, cur_lane, dest_lane, distance);
            }
            if (cig.on()) cig.finish(pair, ncig);
            out.put(i, cost);
        }
    }
}

// Measured (MI355X, 10^6 C2 pairs, ms; profiles/r02_greedy_band_grid.txt): the group form beats the wave-per-pair kernel only
// where that one cannot be used at all —
//     k            6     8     12    16    24-31   32-39   40-50
//     (T, LPT)    4x4   8x3   8x4   16x3  16x4    16x5    16x6/7
//     group       0.47  0.63  0.80  1.07  1.39    1.78    2.2-2.6
//     wave / 2w   0.87  0.87  0.87  0.87  0.87    2.25    2.29
//     thread      0.22  0.35  0.48  (0.73 at k = 16: spills)
// — because a band lane costs ~200 instructions per step whichever thread runs it, the wave kernel already keeps 2k+1 of 64
// threads on lanes of their own with 41 VGPRs, and what the group form saves in idle threads it loses to registers (LPT lanes
// of state per thread: 3 waves per SIMD instead of 7) and to ds_bpermute latency.  So only (16, 5) is instantiated, for the
// 65..79 band lanes of 32 <= k <= 39; k <= 14 runs thread per pair, 15 <= k <= 31 wave per pair, k >= 40 two waves per pair.
static inline bool greedy_group_shape(int k, int& T, int& LPT) {
    const int nl = 2 * k + 1;
    static const int shapes[][2] = {{16, 5}};
    for (const auto& s : shapes)
        if (s[0] * s[1] >= nl) {
            T = s[0], LPT = s[1];
            return true;
        }
    return false;
}

template <int T, int LPT>
static inline hipError_t launch_greedy_group_tl(hipStream_t stream, const uint4* planes, const uint32_t* lens, int64_t n, int w4, int k,
                                                const GreedyArgs& ga, OutMap out, CigarSink cig, int num_cus) {
    int per_cu = 4;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, greedy_group_kernel<T, LPT>, ASM_BLOCK, 0);
    if (per_cu < 1) per_cu = 1;
    int64_t blocks = (int64_t)per_cu * num_cus;
    const int64_t need = (n * T + ASM_BLOCK - 1) / ASM_BLOCK;
    if (blocks > need) blocks = need;
    hipLaunchKernelGGL((greedy_group_kernel<T, LPT>), dim3((unsigned)blocks), dim3(ASM_BLOCK), 0, stream, planes, lens, (long)n, w4, k,
                       ga, out, cig);
    return hipGetLastError();
}

static inline hipError_t launch_greedy_group(hipStream_t stream, const uint4* planes, const uint32_t* lens, int64_t n, int w4, int k,
                                             const GreedyArgs& ga, OutMap out, CigarSink cig, int num_cus) {
    int T = 0, LPT = 0;
    if (!greedy_group_shape(k, T, LPT)) return hipErrorInvalidValue;
    if (T == 16 && LPT == 5) return launch_greedy_group_tl<16, 5>(stream, planes, lens, n, w4, k, ga, out, cig, num_cus);
    return hipErrorInvalidValue;
}
