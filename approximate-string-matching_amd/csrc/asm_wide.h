// General-parameter kernels: wide bands (k up to 50/63), arbitrary penalties, and the affine-gap NW.
//
// For a wide band the 2k+1 lanes no longer fit one thread's registers, so the mapping flips: ONE WORKGROUP
// PER READ PAIR, thread = band lane (the "wavefront per pair" shape: 61 lanes at k = 30 fill a 64-wide wave),
// neighbours exchange their furthest-reach values through a small LDS ring, and `__syncthreads_or` doubles
// as the per-generation barrier and the "some lane reached the end" vote.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "asm_bits.h"
#include "asm_kernels.h"

#define ASM_WIDE_MAX_K 63       /* 2k+1 <= 127 lanes -> one 128-thread workgroup                         */
#define ASM_WIDE_MAX_PENALTY 15 /* LEAP history ring depth 16; NW boundary values stay inside int16       */
#define ASM_WIDE_THREADS 128
#define ASM_WIDE_RING 16

// --------------------------------------------------------------------------------------------------------
// LEAP, any k <= 63, any penalties 1 <= ext <= o, x >= 1 (<= 15), len <= 64*W64.  LV::run, LV_BAG.cpp:127-245.
// --------------------------------------------------------------------------------------------------------
template <int W64>
__global__ __launch_bounds__(ASM_WIDE_THREADS) void leap_wide_kernel(const uint4* __restrict__ planes,
                                                                     const uint32_t* __restrict__ lens, long n,
                                                                     int w4, int k, int x, int o, int ext,
                                                                     int mode /* ASM_LEAP_* (LV::init's ED_modes) */, OutMap out) {
    // LOCAL and SEMI_FREE_BEGIN give every lane a start at generation 0, at the lane's distance from the main one (LV_BAG.cpp:
    // 102-104); LOCAL and SEMI_FREE_END accept any lane that reaches the end, without converge_ED's lane term and threshold (:220-238)
    const bool all_start = mode == 1 || mode == 2, converge_rule = mode == 0 || mode == 2;
    __shared__ int s_end[ASM_WIDE_RING][ASM_WIDE_THREADS + 2];
    __shared__ int s_ip[ASM_WIDE_RING][ASM_WIDE_THREADS + 2];
    __shared__ int s_dp[ASM_WIDE_RING][ASM_WIDE_THREADS + 2];
    const int t = threadIdx.x;
    const int nl = 2 * k + 1;
    const bool active = t < nl;
    const int d = t - k; /* lane l = t + 1, mid = k + 1 */
    const int diff = d < 0 ? -d : d;
    for (long i = blockIdx.x; i < n; i += gridDim.x) {
        const uint32_t ln = lens[i];
        const int m = (int)(ln & 0xffffu), nn = (int)(ln >> 16);
        const int len = m > nn ? m : nn;
        VW<W64> A0, A1, B0, B1;
        load_planes<W64>(planes, n, w4, i, A0, A1, B0, B1);
        const VW<W64> VA = vw_low_ones<W64>(m), VB = vw_low_ones<W64>(nn);
        VW<W64> mask;
        if (active) {
            mask = leap_lane_mask<W64>(A0, A1, B0, B1, VA, VB, d);
        } else {
#pragma unroll
            for (int q = 0; q < W64; q++) mask.w[q] = ~0ull;
        }
        // ring slots hold generations e mod RING; slot index t+1 leaves the two sentinel lanes at -2
        for (int g = 0; g < ASM_WIDE_RING; g++) {
            s_end[g][t + 1] = -2, s_ip[g][t + 1] = -2, s_dp[g][t + 1] = -2;
            if (t < 2) {
                const int edge = t == 0 ? 0 : ASM_WIDE_THREADS + 1;
                s_end[g][edge] = -2, s_ip[g][edge] = -2, s_dp[g][edge] = -2;
            }
        }
        int pass0 = 0;
        if (active && (d == 0 || all_start)) { /* e = 0, LV_BAG.cpp:131-147: start[l][0] = |l - mid| */
            const int from0 = diff > len ? len : diff; /* as a start of any later generation (count_ID_length, :9-23) */
            int r0 = vw_next_one<W64>(mask, from0);
            r0 = r0 > len ? len : r0;
            const int e0 = diff > len ? diff : r0;
            s_end[0][t + 1] = e0;
            pass0 = (e0 == len);
        }
        int result = -1;
        if (__syncthreads_or(pass0)) {
            result = 0;
        } else {
            for (int e = 1; e <= ASM_LEAP_AF_THRESHOLD; e++) {
                int pass = 0;
                int enew = -2, inew = -2, dnew = -2;
                if (active) {
                    const int top = d >= 0 ? 1 : 0, bot = d <= 0 ? 1 : 0;
                    const int so = (e - o) & (ASM_WIDE_RING - 1), se = (e - ext) & (ASM_WIDE_RING - 1),
                              sx = (e - x) & (ASM_WIDE_RING - 1);
                    /* lanes t-1 / t+1; beyond the band (sentinels and inactive threads) everything is -2 */
                    const int e_up = (e >= o && t > 0) ? s_end[so][t] : -2;
                    const int i_up = (e >= ext && t > 0) ? s_ip[se][t] : -2;
                    const int e_dn = (e >= o && t + 1 < nl) ? s_end[so][t + 2] : -2;
                    const int d_dn = (e >= ext && t + 1 < nl) ? s_dp[se][t + 2] : -2;
                    const int own = (e >= x) ? s_end[sx][t + 1] : -2;
                    if (e >= o && e_up >= 0 && e_up > i_up)
                        inew = e_up + top;
                    else if (e >= ext && i_up >= 0)
                        inew = i_up + top;
                    if (e >= o && e_dn >= 0 && e_dn > d_dn)
                        dnew = e_dn + bot;
                    else if (e >= ext && d_dn >= 0)
                        dnew = d_dn + bot;
                    int st = own >= 0 ? own + 1 : -2;
                    st = inew > st ? inew : st;
                    st = dnew > st ? dnew : st;
                    if (st >= 0) {
                        const int from = st > len ? len : st;
                        int r = vw_next_one<W64>(mask, from);
                        r = r > len ? len : r;
                        enew = st > len ? st : r;
                        if (enew == len) {
                            const int conv = e + (diff ? o + (diff - 1) * ext : 0);
                            if (!converge_rule || conv <= ASM_LEAP_AF_THRESHOLD) pass = 1;
                        }
                    }
                }
                const int sw = e & (ASM_WIDE_RING - 1);
                s_end[sw][t + 1] = enew, s_ip[sw][t + 1] = inew, s_dp[sw][t + 1] = dnew;
                if (__syncthreads_or(pass)) {
                    result = e;
                    break;
                }
            }
        }
        if (t == 0) out.put(i, result);
        __syncthreads(); /* ring is re-initialised by the next pair */
    }
}

static inline void launch_leap_wide(hipStream_t stream, const uint4* planes, const uint32_t* lens, int64_t n, int w4,
                                    int k, int x, int o, int e, OutMap out, int mode = 0) {
    int64_t blocks = n < 256 * 32 ? n : 256 * 32;
    const int maxw = w4 * 2;
    if (maxw <= 2)
        hipLaunchKernelGGL(leap_wide_kernel<2>, dim3((unsigned)blocks), dim3(ASM_WIDE_THREADS), 0, stream, planes, lens,
                           (long)n, w4, k, x, o, e, mode, out);
    else if (maxw <= 4)
        hipLaunchKernelGGL(leap_wide_kernel<4>, dim3((unsigned)blocks), dim3(ASM_WIDE_THREADS), 0, stream, planes, lens,
                           (long)n, w4, k, x, o, e, mode, out);
    else
        hipLaunchKernelGGL(leap_wide_kernel<8>, dim3((unsigned)blocks), dim3(ASM_WIDE_THREADS), 0, stream, planes, lens,
                           (long)n, w4, k, x, o, e, mode, out);
}

// --------------------------------------------------------------------------------------------------------
// Greedy, any k <= 50: workgroup per pair, thread = lane.  Same step structure as greedy_kernel<K>; the two
// lane loops of _update_highway_list become per-thread work plus a workgroup arg-max, and the order-dependent
// fold of _choose_best_highway (hurdle_matrix.h:382-399: a lane is accepted only if it is no worse than the
// LAST accepted one in both total and intermediate cost) is replayed in lane order from LDS by every thread.
// --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ASM_WIDE_THREADS) void greedy_wide_kernel(const uint4* __restrict__ planes,
                                                                       const uint32_t* __restrict__ lens, long n,
                                                                       int w4, int k, GreedyArgs args,
                                                                       OutMap out, CigarSink cig) {
    __shared__ double s_heur[ASM_WIDE_THREADS];
    __shared__ int s_leap[ASM_WIDE_THREADS];
    __shared__ int s_sp[ASM_WIDE_THREADS], s_len[ASM_WIDE_THREADS], s_cost[ASM_WIDE_THREADS];
    __shared__ int s_inter[ASM_WIDE_THREADS], s_total[ASM_WIDE_THREADS];
    __shared__ u64 s_vec[2];
    const int t = threadIdx.x;
    const int nl = 2 * k + 1;
    const bool active = t < nl;
    const int lane = t - k;
    const int x = args.x, o = args.o, e = args.e;
    const bool semi = args.semi != 0;
    for (long i = blockIdx.x; i < n; i += gridDim.x) {
        const V128 A0 = v_from_uint4(planes[((long)0 * w4) * n + i]);
        const V128 A1 = v_from_uint4(planes[((long)1 * w4) * n + i]);
        const V128 B0 = v_from_uint4(planes[((long)2 * w4) * n + i]);
        const V128 B1 = v_from_uint4(planes[((long)3 * w4) * n + i]);
        const uint32_t ln = lens[i];
        int m = (int)(ln & 0xffffu), nn = (int)(ln >> 16);
        m = m > 128 ? 128 : m;
        nn = nn > 128 ? 128 : nn;
        const int dest_lane = nn - m;
        const V128 lo_ = greedy_lane_vector(A0, A1, B0, B1, lane);
        const V128 lf_ = v_flip_short_hurdles1(lo_);
        int sp = -1, len = 0, nsw = 128;
        const int dst = lane_destination(m, nn, lane);
        int cur_lane = 0, cur_col = 0, cost = 0, ncig = 0;
        const long pair = out.index(i);
        for (int guard = 0; guard < 4 * 128; guard++) {
            int reach = 0, sw = 0, nh = 0;
            if (active) {
                const int start_col = cur_col + fwd_col(cur_lane, lane);
                if (sp < start_col) {
                    int dd = lane - cur_lane;
                    nsw = dd < 0 ? -dd : dd;
                    int fz, nx;
                    v_highway_from(lf_, start_col, fz, nx);
                    sp = start_col + fz;
                    len = nx;
                    if (start_col + fz + nx > dst) {
                        const int c = dst - (start_col + fz);
                        len = c > 0 ? c : 0;
                        reach = 1;
                    }
                }
                sw = (semi && guard == 0) ? 0 : lane_penalty(cur_lane, lane, o, e);
                nh = v_pop_between(lo_, start_col, sp + len);
            }
            const int reaching = __syncthreads_or(reach);
            const int hc = x * nh;
            if (active) {
                double heur = greedy_significance(args, len, nh, nsw);
                int leap = -sw;
                if (reaching) {
                    const int fsw = semi ? 0 : lane_penalty(lane, dest_lane, o, e);
                    heur = (double)(-sw - hc - fsw - x * (dst - sp - len));
                    leap -= fsw;
                }
                s_heur[t] = heur, s_leap[t] = leap;
                s_sp[t] = sp, s_len[t] = len, s_cost[t] = sw + hc;
            }
            __syncthreads();
            // arg-max in lane order, replayed by every thread from LDS (broadcast reads)
            double best_h = -__builtin_inf();
            int best_leap = 0, bt = k; /* reference default best lane 0 <-> thread k */
            for (int j = 0; j < nl; j++) {
                const double hj = s_heur[j];
                const int lj = s_leap[j];
                if (hj > best_h || (hj == best_h && lj > best_leap)) best_h = hj, best_leap = lj, bt = j;
            }
            const int best = bt - k, best_sp = s_sp[bt], best_len = s_len[bt], best_cost = s_cost[bt];
            if (best_len <= 0) break; /* uniform across the workgroup */
            if (t == bt) s_vec[0] = lo_.lo, s_vec[1] = lo_.hi;
            __syncthreads();
            const V128 best_vec = v_make(s_vec[0], s_vec[1]);
            const int best_from_sp = v_ones_from(best_vec, best_sp);
            int inter = 0x3fffffff, total = 0x3fffffff;
            if (active && lane != best && !(sp + fwd_col(lane, best) > best_sp)) {
                const int endp = sp + len;
                inter = sw + v_pop_between(lo_, cur_col + fwd_col(cur_lane, lane), endp);
                const int tail = x * v_pop_between_pre(best_vec, fwd_col(lane, best) + endp, best_sp, best_from_sp);
                total = inter + lane_penalty(lane, best, o, e) + (tail > 0 ? tail : 0);
            }
            s_inter[t] = inter, s_total[t] = total;
            __syncthreads();
            int small_inter = best_cost, small_total = best_cost, ct = bt;
            for (int j = 0; j < nl; j++) {
                const int tj = s_total[j], ij = s_inter[j];
                if (j != bt && tj <= small_total && ij <= small_inter) small_total = tj, small_inter = ij, ct = j;
            }
            const int ch = ct - k;
            cost += s_cost[ct];
            if (cig.on() && t == 0)
                cig.step(pair, ncig, cur_lane, ch, s_sp[ct] + s_len[ct] - (cur_col + fwd_col(cur_lane, ch)));
            cur_lane = ch;
            cur_col = s_sp[ct] + s_len[ct];
            const bool done = cur_col >= lane_destination(m, nn, ch);
            __syncthreads(); /* LDS arrays are rewritten by the next step */
            if (done) break;
        }
        if (t == 0) {
            const int dest_col = lane_destination(m, nn, dest_lane);
            if (cur_lane != dest_lane || cur_col < dest_col) {
                const V128 dv = greedy_lane_vector(A0, A1, B0, B1, dest_lane);
                const int sw_f = semi ? 0 : lane_penalty(cur_lane, dest_lane, o, e);
                const int distance = v_pop_between(dv, cur_col + fwd_col(cur_lane, dest_lane), dest_col);
                const int hc = x * distance;
                cost += sw_f + (hc > 0 ? hc : 0);
                if (cig.on()) cig.step(pair, ncig, cur_lane, dest_lane, distance);
            }
            if (cig.on()) cig.finish(pair, ncig);
            out.put(i, cost);
        }
        __syncthreads();
    }
}

static inline void launch_greedy_wide(hipStream_t stream, const uint4* planes, const uint32_t* lens, int64_t n, int w4,
                                      int k, const GreedyArgs& ga, OutMap out, CigarSink cig) {
    int64_t blocks = n < 256 * 32 ? n : 256 * 32;
    hipLaunchKernelGGL(greedy_wide_kernel, dim3((unsigned)blocks), dim3(ASM_WIDE_THREADS), 0, stream, planes, lens, (long)n,
                       w4, k, ga, out, cig);
}

// --------------------------------------------------------------------------------------------------------
// NW with affine gaps (Gotoh), arbitrary penalties: match 0, mismatch x, gap(L) = o + (L-1)*e — the score
// parasail's nw returns negated (benchmark_utils.h:139-142,288).  One thread per pair.  The DP matrix is
// swept in column blocks of 32: a block's H and F rows live in registers (fully unrolled), and only the
// block's right-hand boundary column (H, E per row, packed as two int16 in one dword) goes through LDS,
// laid out [row][thread] so a wave's accesses hit 64 different banks.
// --------------------------------------------------------------------------------------------------------
#define NW_CB 32
#define NW_BIG 30000

template <int W64, int MAXROWS>
__global__ __launch_bounds__(64) void nw_affine_kernel(const uint4* __restrict__ planes,
                                                       const uint32_t* __restrict__ lens, long n, int w4, int x,
                                                       int o, int e, OutMap out, const uint32_t* __restrict__ todo,
                                                       const uint32_t* __restrict__ todo_count) {
    __shared__ uint32_t s_bound[MAXROWS + 1][64];
    const int t = threadIdx.x;
    long i = (long)blockIdx.x * 64 + t;
    if (todo) { /* second pass of launch_nw_wfa: only the listed bucket slots */
        if (i >= (long)*todo_count) return;
        i = (long)todo[i];
    }
    if (i >= n) return;
    const uint32_t ln = lens[i];
    const int m = (int)(ln & 0xffffu), nn = (int)(ln >> 16);
    VW<W64> A0, A1, B0, B1;
    load_planes<W64>(planes, n, w4, i, A0, A1, B0, B1);
    int result = nn == 0 ? (m == 0 ? 0 : o + (m - 1) * e) : 0;
    for (int j0 = 0; j0 < nn; j0 += NW_CB) {
        // text codes of this column block
        uint32_t tb0 = 0, tb1 = 0;
#pragma unroll
        for (int q = 0; q < W64; q++) {
            if ((j0 >> 6) == q) {
                tb0 = (uint32_t)(B0.w[q] >> (j0 & 63));
                tb1 = (uint32_t)(B1.w[q] >> (j0 & 63));
            }
        }
        int H[NW_CB], F[NW_CB];
#pragma unroll
        for (int jj = 0; jj < NW_CB; jj++) {
            H[jj] = o + (j0 + jj) * e; /* H[0][j], j = j0+jj+1 */
            F[jj] = NW_BIG;
        }
        int diag = j0 == 0 ? 0 : o + (j0 - 1) * e; /* H[0][j0] */
        for (int r = 1; r <= m; r++) {
            // code of read character r-1, broadcast
            uint32_t a0 = 0, a1 = 0;
#pragma unroll
            for (int q = 0; q < W64; q++) {
                if (((r - 1) >> 6) == q) {
                    a0 = (uint32_t)(A0.w[q] >> ((r - 1) & 63)) & 1u;
                    a1 = (uint32_t)(A1.w[q] >> ((r - 1) & 63)) & 1u;
                }
            }
            const uint32_t mm = (tb0 ^ (0u - a0)) | (tb1 ^ (0u - a1));
            int hleft, eleft;
            if (j0 == 0) {
                hleft = o + (r - 1) * e; /* H[r][0] */
                eleft = NW_BIG;
            } else {
                const uint32_t pk = s_bound[r][t];
                hleft = (int)(pk & 0xffffu);
                eleft = (int)(pk >> 16);
            }
            const int next_diag = hleft;
#pragma unroll
            for (int jj = 0; jj < NW_CB; jj++) {
                const int up = H[jj];
                int f = F[jj] + e;
                f = up + o < f ? up + o : f;
                int ee = eleft + e;
                ee = hleft + o < ee ? hleft + o : ee;
                const int dg = diag + (((mm >> jj) & 1u) ? x : 0);
                int h = dg < f ? dg : f;
                h = ee < h ? ee : h;
                f = f > NW_BIG ? NW_BIG : f;
                ee = ee > NW_BIG ? NW_BIG : ee;
                diag = up;
                H[jj] = h, F[jj] = f;
                hleft = h, eleft = ee;
            }
            diag = next_diag;
            s_bound[r][t] = (uint32_t)hleft | ((uint32_t)eleft << 16);
        }
        if (nn > j0 && nn <= j0 + NW_CB) {
            const int want = nn - j0 - 1;
#pragma unroll
            for (int jj = 0; jj < NW_CB; jj++)
                if (jj == want) result = H[jj];
        }
    }
    out.put(i, result);
}

static inline void launch_nw_affine(hipStream_t stream, const uint4* planes, const uint32_t* lens, int64_t n, int w4,
                                    int maxlen, int x, int o, int e, OutMap out) {
    const dim3 g((unsigned)((n + 63) / 64)), t(64);
    if (maxlen <= 128)
        hipLaunchKernelGGL((nw_affine_kernel<2, 128>), g, t, 0, stream, planes, lens, (long)n, w4, x, o, e, out,
                           (const uint32_t*)nullptr, (const uint32_t*)nullptr);
    else if (maxlen <= 256)
        hipLaunchKernelGGL((nw_affine_kernel<4, 256>), g, t, 0, stream, planes, lens, (long)n, w4, x, o, e, out,
                           (const uint32_t*)nullptr, (const uint32_t*)nullptr);
    else
        hipLaunchKernelGGL((nw_affine_kernel<8, 512>), g, t, 0, stream, planes, lens, (long)n, w4, x, o, e, out,
                           (const uint32_t*)nullptr, (const uint32_t*)nullptr);
}
