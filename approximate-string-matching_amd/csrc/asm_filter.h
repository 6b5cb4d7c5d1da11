// Bit-parallel LEAP (SIMD_ED, Levenshtein mode, ED_GLOBAL) and the SHD pre-filter of GASMA/benchmark/LEAP_SIMD — the
// filtering stage in front of the aligners (driver LEAP_SIMD/main.cpp:95-101,186-195).  One thread per
// pair on the batch's 2-bit planes; vectors are W64 64-bit words (2: length <= 128, 4: <= 256 = _MAX_LENGTH_).
//
// Reference behaviour kept on purpose (oracle/asm_oracle_filter.c S1-S4):
//  S1  the reference's 256-bit shifts move each 128-bit half on its own (shift.cpp:33-61);
//  S2  SIMD_ED's verdict state (final_ED, final lane, converge_ED) flows from pair to pair: the kernels emit one *event*
//      per pair and, in sequential mode, two scans resolve the chain in batch order (simd_ed_*_kernel below);
//  S3  the mask-array SHD inside SIMD_ED amends nothing and masks the main lane with bits 0..254 (SHD.cpp:324-372);
//  S4  a lane that starts at or beyond the end reaches the end (SIMD_ED.cpp:57-60).
#pragma once
#include "asm_kernels.h"

#define ASM_FILTER_MAX_T 32    /* lanes 2T+1 <= 65 */
#define ASM_FILTER_REG_MAX_T 8 /* masks of all lanes held in registers up to here; computed per use beyond */
#define ASM_SHD_MAX_ERROR 16   /* MAX_ERROR_AVX (LEAP_SIMD/mask.h:21): rows of the reference's begin-mask table */

// events of one pair (S2)
#define EV_REJECTED (-1)  /* SHD said no: verdict fail, state untouched                                             */
#define EV_NOT_REACHED (-2) /* no lane reached the end within T generations: verdict = the carried state's             */
#define EV_EXACT (-3)     /* the main lane reached the end at e = 0: passes, sets (final_ED, lane) = (0, 0) only;
                             EV_EXACT - d: the same on a lane d off the main one (ED_modes LOCAL / SEMI_FREE_BEGIN)          */
/* >= 0: reached at generation fe on a lane fd away from the main one: fe | fd << 8                                     */

// bits move away from index 0 by s in [0, 63], each 128-bit half separately when W64 = 4 (S1)
template <int W64>
ASM_DEV VW<W64> avx_away0(const VW<W64>& v, int s) {
    VW<W64> r;
#pragma unroll
    for (int q = 0; q < W64; q++) {
        const u64 lo = (q & 1) ? v.w[q - 1] : 0ull; /* nothing crosses an even word boundary */
        r.w[q] = (v.w[q] << s) | (s ? (lo >> (64 - s)) : 0ull);
    }
    return r;
}

// hamming mask of lane d = l - mid in [-T, T] (SIMD_ED::calculate_masks, SIMD_ED.cpp:180-212): d < 0 shifts the reference,
// d > 0 the read
template <int W64>
ASM_DEV VW<W64> simd_lane_mask(const VW<W64>& A0, const VW<W64>& A1, const VW<W64>& B0, const VW<W64>& B1, int d) {
    VW<W64> r;
    const int s = d < 0 ? -d : d;
    if (d < 0) {
        const VW<W64> b0 = avx_away0<W64>(B0, s), b1 = avx_away0<W64>(B1, s);
#pragma unroll
        for (int q = 0; q < W64; q++) r.w[q] = (A0.w[q] ^ b0.w[q]) | (A1.w[q] ^ b1.w[q]);
    } else {
        const VW<W64> a0 = avx_away0<W64>(A0, s), a1 = avx_away0<W64>(A1, s);
#pragma unroll
        for (int q = 0; q < W64; q++) r.w[q] = (a0.w[q] ^ B0.w[q]) | (a1.w[q] ^ B1.w[q]);
    }
    return r;
}

// end position of the run of matches that starts at st (start + count_ID_length_avx, SIMD_ED.cpp:10-61): the next set bit
// of the lane mask, capped at len.  The reference shifts the mask down by st half by half (S1), so for st < 128 the bits
// 128 .. 127+st never arrive and read as matches.
template <int W64>
ASM_DEV int simd_extend(const VW<W64>& mask, int st, int len) {
    if (st >= len) return len; /* S4: st + (len - st) */
    VW<W64> v = mask;
    if (W64 == 4 && st < 128) {
        v.w[2] &= st >= 64 ? 0ull : (~0ull << st);
        if (st > 64) v.w[3] &= ~0ull << (st - 64);
    }
    const int r = vw_next_one<W64>(v, st);
    return r < len ? r : len;
}

// popcount_SHD_avx (popcount.cpp:44-76,78-110): a table lookup per nibble that is the number of runs of ones in the nibble,
// except that 0110 counts 2
template <int W64>
ASM_DEV int shd_popcount(const VW<W64>& v) {
    int s = 0;
#pragma unroll
    for (int q = 0; q < W64; q++) {
        const u64 w = v.w[q];
        const u64 x = w ^ 0x6666666666666666ull; /* a nibble of x is zero where w's is 0110 */
        const u64 is6 = ~(x | (x >> 1) | (x >> 2) | (x >> 3)) & 0x1111111111111111ull;
        s += __builtin_popcountll(w & ~((w << 1) & 0xeeeeeeeeeeeeeeeeull)) + __builtin_popcountll(is6);
    }
    return s;
}

template <int W64>
ASM_DEV VW<W64> vw_from(int cnt) { /* bits cnt.. set: MASK_AVX_BEG row cnt-1 (mask.cpp:149-166) */
    VW<W64> r = vw_low_ones<W64>(cnt);
#pragma unroll
    for (int q = 0; q < W64; q++) r.w[q] = ~r.w[q];
    return r;
}

// ---------------------------------------------------------------------------------------------------------------------
// SIMD_ED::run_levenshtein for one pair (SIMD_ED.cpp:269-352) -> event.  TT > 0: T = TT at compile time, all lane masks in
// registers, lanes unrolled.  TT = 0: T at run time, lane masks rebuilt per use, the previous generation's end positions in
// a thread-private LDS column.
// ---------------------------------------------------------------------------------------------------------------------
#define SIMD_ED_THREADS 128
template <int TT, int W64>
__global__ __launch_bounds__(SIMD_ED_THREADS) void simd_ed_kernel(const uint4* __restrict__ planes,
                                                                  const uint32_t* __restrict__ lens, long n, int w4, int t_rt,
                                                                  int shd_enable, int ed_mode /* ASM_LEAP_*; TT == 0 only */,
                                                                  OutMap events) {
    // init_levenshtein's ED_modes: LOCAL (1) and SEMI_FREE_BEGIN (2) keep every lane live from generation 0, starting at its
    // distance from the main lane (SIMD_ED.cpp:246-266: start[i][0] = ED, cur_ED[i] = 0); GLOBAL and SEMI_FREE_END let lane l
    // join at generation |l - mid|.  The compile-time-T instantiations serve GLOBAL / SEMI_FREE_END only.
    const bool all_start = TT == 0 && (ed_mode == 1 || ed_mode == 2);
    constexpr int NLC = TT > 0 ? 2 * TT + 1 : 1;
    extern __shared__ short s_end[]; /* TT = 0: [2][2T+3][threads] */
    const int T = TT > 0 ? TT : t_rt;
    const int NL = 2 * T + 1;
    const int t = threadIdx.x;
    const long i = (long)blockIdx.x * SIMD_ED_THREADS + t;
    if (i >= n) return;
    const int m = (int)(lens[i] & 0xffffu), nn_ref = (int)(lens[i] >> 16);
    const int len = m > 64 * W64 ? 64 * W64 : m; /* main.cpp:131-132: the read's length, at most _MAX_LENGTH_ */
    VW<W64> A0, A1, B0, B1;
    load_planes<W64>(planes, n, w4, i, A0, A1, B0, B1);
    {   /* strncpy(A/B, ., length): what lies beyond is never compared, but clear it so that shifts bring in zeros only */
        const VW<W64> lm = vw_low_ones<W64>(len), lb = vw_low_ones<W64>(nn_ref < len ? nn_ref : len);
#pragma unroll
        for (int q = 0; q < W64; q++) A0.w[q] &= lm.w[q], A1.w[q] &= lm.w[q], B0.w[q] &= lb.w[q], B1.w[q] &= lb.w[q];
    }
    VW<W64> hm[NLC];
    if (TT > 0) {
#pragma unroll
        for (int j = 0; j < NLC; j++) hm[j] = simd_lane_mask<W64>(A0, A1, B0, B1, j - TT);
    }
    if (shd_enable) { /* bit_vec_filter_avx(hamming_masks + 1, buffer_length, ED_t), S3 */
        const VW<W64> lm = vw_low_ones<W64>(len);
        VW<W64> diff = lm;
        if (TT > 0) {
#pragma unroll
            for (int j = 0; j < NLC; j++) {
                const int s = j < TT ? TT - j : j - TT;
                const VW<W64> tm = s ? vw_from<W64>(s) : vw_low_ones<W64>(255);
#pragma unroll
                for (int q = 0; q < W64; q++) diff.w[q] &= hm[j].w[q] & tm.w[q];
            }
        } else {
            for (int j = 0; j < NL; j++) {
                const int s = j < T ? T - j : j - T;
                const VW<W64> tm = s ? vw_from<W64>(s) : vw_low_ones<W64>(255);
                const VW<W64> h = simd_lane_mask<W64>(A0, A1, B0, B1, j - T);
#pragma unroll
                for (int q = 0; q < W64; q++) diff.w[q] &= h.w[q] & tm.w[q];
            }
        }
        if (shd_popcount<W64>(diff) > T) {
            events.put(i, EV_REJECTED);
            return;
        }
    }
    int ev = EV_NOT_REACHED;
    if (TT > 0) {
        int prev[NLC + 2]; /* end[.][e-1] with the two guard lanes; -2 = never written (SIMD_ED.cpp:236-241) */
#pragma unroll
        for (int j = 0; j < NLC + 2; j++) prev[j] = -2;
        prev[TT + 1] = simd_extend<W64>(hm[TT], 0, len);
        if (prev[TT + 1] == len) ev = EV_EXACT; /* SIMD_ED.cpp:291-296 */
        for (int e = 1; e <= TT && ev == EV_NOT_REACHED; e++) {
            int cur[NLC + 2];
            cur[0] = cur[NLC + 1] = -2;
#pragma unroll
            for (int j = 0; j < NLC; j++) {
                const int dist = j < TT ? TT - j : j - TT;
                int st = prev[j + 1] + 1; /* SIMD_ED.cpp:318-323 */
                const int up = prev[j] + (j >= TT ? 1 : 0), dn = prev[j + 2] + (j <= TT ? 1 : 0);
                st = up > st ? up : st;
                st = dn > st ? dn : st;
                int en = -2;
                if (dist <= e) { /* cur_ED[l] == e */
                    en = simd_extend<W64>(hm[j], st, len);
                    if (en == len && ev == EV_NOT_REACHED) ev = e | (dist << 8); /* the lowest lane wins (:334-340) */
                }
                cur[j + 1] = en;
            }
#pragma unroll
            for (int j = 0; j < NLC + 2; j++) prev[j] = cur[j];
        }
    } else {
        const int rows = NL + 2;
        short* const col = s_end + t;
#define END_AT(g, l) col[((g)*rows + (l)) * SIMD_ED_THREADS]
        for (int l = 0; l < rows; l++) END_AT(0, l) = -2, END_AT(1, l) = -2;
        for (int j = all_start ? 0 : T; j <= (all_start ? 2 * T : T) && ev == EV_NOT_REACHED; j++) { /* cur_ED == 0, ascending */
            const int dist = j < T ? T - j : j - T;
            const VW<W64> h = simd_lane_mask<W64>(A0, A1, B0, B1, j - T);
            const int e0 = simd_extend<W64>(h, dist, len);
            END_AT(0, j + 1) = (short)e0;
            if (e0 == len) ev = EV_EXACT - dist;
        }
        for (int e = 1; e <= T && ev == EV_NOT_REACHED; e++) {
            const int gp = (e - 1) & 1, gc = e & 1;
            for (int j = all_start ? 0 : T - e; j <= (all_start ? 2 * T : T + e); j++) { /* lanes with cur_ED == e, ascending */
                int st = (int)END_AT(gp, j + 1) + 1;
                const int up = (int)END_AT(gp, j) + (j >= T ? 1 : 0), dn = (int)END_AT(gp, j + 2) + (j <= T ? 1 : 0);
                st = up > st ? up : st;
                st = dn > st ? dn : st;
                const VW<W64> h = simd_lane_mask<W64>(A0, A1, B0, B1, j - T);
                const int en = simd_extend<W64>(h, st, len);
                END_AT(gc, j + 1) = (short)en;
                if (en == len) {
                    const int dist = j < T ? T - j : j - T;
                    ev = e | (dist << 8);
                    break;
                }
            }
        }
#undef END_AT
    }
    events.put(i, ev);
}

// ---------------------------------------------------------------------------------------------------------------------
// SIMD_ED in affine mode (init_affine / run_affine, SIMD_ED.cpp:435-616), CLEAN: every pair starts from the tables
// init_affine leaves.  (The reference as run keeps the I/D/end tables of the pair before — run_affine writes them only where
// its conditions hold and reset_affine touches none — so a pair's verdict depends on everything the object has seen; that
// chain runs through every table cell and has no parallel form.  Clean = the reference's verdict for the first pair after
// init_affine, which is how oracle/ref_harness_simd.cpp drives the compiled reference for the pin.)
// One thread per pair; the recurrence is LV::run's (leap_general_kernel) over SIMD_ED's own lane masks (simd_lane_mask, rebuilt
// per use) with the read's length as the target; generation rings of 2^a > max(x, o) and 2^b > ext generations in thread-
// private LDS columns [slot][lane row][thread], stored + 2 so that the initial zero fill is "never reached" (which also stands
// in for the e >= penalty guards, see leap_general_kernel).  Result: get_ED() = converge_ED, the smallest e + gap(lane
// distance) <= af_threshold over the lanes that reach the end in the first generation in which one does; 1000000 —
// reset_affine's value — for a pair that reaches the end at e = 0 (run_affine returns before converge_ED is written); -1 when
// the pair does not pass.  blockDim.x threads (64, 32 or 16: whatever lets the rings fit the LDS).
// ---------------------------------------------------------------------------------------------------------------------
template <int W64>
__global__ __launch_bounds__(64) void simd_ed_affine_kernel(const uint4* __restrict__ planes, const uint32_t* __restrict__ lens,
                                                            long n, int w4, int T, int af_t, int x, int o, int ext, int gm, int gi,
                                                            int mode /* ASM_LEAP_*: init_affine's ED_modes */, OutMap out) {
    // LOCAL (1) and SEMI_FREE_BEGIN (2): every lane starts at generation 0, at its distance from the main lane (SIMD_ED.cpp:476-478,
    // 497-516); LOCAL and SEMI_FREE_END (3): any lane that reaches the end passes and get_ED() is final_ED (:589-610,748-753)
    const bool all_start = mode == 1 || mode == 2, converge_rule = mode == 0 || mode == 2;
    extern __shared__ uint16_t s_afring[]; /* `end` [gm][rows][TH], I [gi][rows][TH], D [gi][rows][TH] */
    const int TH = (int)blockDim.x, t = threadIdx.x;
    const int rows = 2 * T + 3, mid = T + 1; /* lanes 1 .. 2T+1, guard rows 0 and 2T+2 (SIMD_ED.cpp:452-453) */
    const int slot = rows * TH;
    uint16_t* const r_en = s_afring + t;
    uint16_t* const r_ip = r_en + gm * slot;
    uint16_t* const r_dp = r_ip + gi * slot;
    {
        const int words = ((gm + 2 * gi) * slot + 1) / 2;
        uint32_t* const base = reinterpret_cast<uint32_t*>(s_afring);
        for (int w = t; w < words; w += TH) base[w] = 0u;
    }
    __syncthreads();
    const long i = (long)blockIdx.x * TH + t;
    if (i >= n) return;
    const int m = (int)(lens[i] & 0xffffu), nn_ref = (int)(lens[i] >> 16);
    const int len = m > 64 * W64 ? 64 * W64 : m;
    VW<W64> A0, A1, B0, B1;
    load_planes<W64>(planes, n, w4, i, A0, A1, B0, B1);
    {   /* strncpy(A/B, ., length); the reference string ends at its own length (a sequential-mode batch keeps Greedy's stale tail
           bits beyond it) */
        const VW<W64> lm = vw_low_ones<W64>(len), lb = vw_low_ones<W64>(nn_ref < len ? nn_ref : len);
#pragma unroll
        for (int q = 0; q < W64; q++) A0.w[q] &= lm.w[q], A1.w[q] &= lm.w[q], B0.w[q] &= lb.w[q], B1.w[q] &= lb.w[q];
    }
    int result = -2; /* -2: still running */
    {   /* e = 0: the main lane (ED_GLOBAL, SEMI_FREE_END) or every lane at its distance (:474-477,497-516) */
        const int l0 = all_start ? 1 : mid, l1 = all_start ? 2 * T + 1 : mid;
        for (int l = l0; l <= l1; l++) {
            const int dist = l < mid ? mid - l : l - mid;
            const int e0 = simd_extend<W64>(simd_lane_mask<W64>(A0, A1, B0, B1, l - mid), dist, len);
            r_en[l * TH] = (uint16_t)(e0 + 2);
            if (e0 == len) result = converge_rule ? 1000000 : 0; /* converge_ED is never written here; final_ED = 0 */
        }
    }
    for (int e = 1; e <= af_t; e++) {
        if (__ballot(result == -2) == 0ull) break;
        if (result != -2) continue;
        int dmax = e < o ? 0 : (e - o) / ext + 1; /* lanes a gap of this cost can have reached */
        dmax = (dmax > T || all_start) ? T : dmax; /* (every lane is live from the start in LOCAL / SEMI_FREE_BEGIN) */
        const uint16_t* const en_o = r_en + ((e - o) & (gm - 1)) * slot;
        const uint16_t* const en_x = r_en + ((e - x) & (gm - 1)) * slot;
        const uint16_t* const ip_e = r_ip + ((e - ext) & (gi - 1)) * slot;
        const uint16_t* const dp_e = r_dp + ((e - ext) & (gi - 1)) * slot;
        uint16_t* const en_w = r_en + (e & (gm - 1)) * slot;
        uint16_t* const ip_w = r_ip + (e & (gi - 1)) * slot;
        uint16_t* const dp_w = r_dp + (e & (gi - 1)) * slot;
        int conv = 1000000;
        for (int l = mid - dmax; l <= mid + dmax; l++) {
            const int top = l >= mid ? 1 : 0, bot = l <= mid ? 1 : 0;
            const int e_up = (int)en_o[(l - 1) * TH] - 2, i_up = (int)ip_e[(l - 1) * TH] - 2;
            const int e_dn = (int)en_o[(l + 1) * TH] - 2, d_dn = (int)dp_e[(l + 1) * TH] - 2;
            const int own = (int)en_x[l * TH] - 2;
            int inew = -2, dnew = -2;
            if (e_up >= 0 && e_up > i_up)
                inew = e_up + top; /* :533-538 */
            else if (i_up >= 0)
                inew = i_up + top; /* :539-544 */
            if (e_dn >= 0 && e_dn > d_dn)
                dnew = e_dn + bot; /* :546-547 */
            else if (d_dn >= 0)
                dnew = d_dn + bot; /* :548-549 */
            int st = own >= 0 ? own + 1 : -2; /* :551-558 */
            st = inew > st ? inew : st;
            st = dnew > st ? dnew : st;
            int enew = -2;
            if (st >= 0) {
                enew = simd_extend<W64>(simd_lane_mask<W64>(A0, A1, B0, B1, l - mid), st, len); /* :579-581 */
                if (enew == len) { /* :589-603 */
                    const int diff = l < mid ? mid - l : l - mid;
                    const int tc = converge_rule ? e + (diff ? o + (diff - 1) * ext : 0) : e; /* :604-608: final_ED, no threshold */
                    if ((!converge_rule || tc <= af_t) && tc < conv) conv = tc;
                }
            }
            en_w[l * TH] = (uint16_t)(enew + 2), ip_w[l * TH] = (uint16_t)(inew + 2), dp_w[l * TH] = (uint16_t)(dnew + 2);
        }
        if (conv != 1000000) result = conv; /* ED_pass: the generation loop ends (:609-610) */
    }
    out.put(i, result == -2 ? -1 : result);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same recurrence with FOUR THREADS PER PAIR, sixteen pairs per wave, for strings of at most 128 characters (one 128-bit
// half: none of the half-by-half shift effects of the 256-bit form) — the shape leap_quad_kernel gave LV::run (asm_wave.h):
// the live lanes of a generation are dealt round-robin to the quad, rings [slot][lane row][pair] of bytes (end + 2 <= 130)
// and the pair's four planes pair-major in LDS, each between two zero dwords so that the window of a shifted operand can start
// up to 32 positions before the string (avx_away0 fills with zeros) and run past its end; a wave-scope fence between
// generations.  Against one thread per pair the LDS a wave needs shrinks by four (gap 30, (2,3,1): 8 KB per wave instead of
// 61), which is what bounds the occupancy there, and a wave waits for the slowest of 16 pairs instead of 64.
// ---------------------------------------------------------------------------------------------------------------------
#define SIMD_QUAD_PD 6 /* zero dword, 4 plane dwords, zero dword */
#define SIMD_QUAD_PW (4 * SIMD_QUAD_PD + 1)
ASM_DEV uint32_t simd_quad_window(const uint32_t* plane, int pos) { /* pos is already offset by the leading zero dword */
    const int q = pos >> 5;
    return __builtin_amdgcn_alignbit(plane[q + 1], plane[q], (uint32_t)(pos & 31));
}
// simd_extend(simd_lane_mask(d), st, len) on the LDS planes: d < 0 shifts the reference away from 0, d > 0 the read
ASM_DEV int simd_quad_extend(const uint32_t* pl, int d, int st, int len) {
    if (st >= len) return len;
    const int s = d < 0 ? -d : d;
    int apos = st + 32 - (d > 0 ? s : 0), bpos = st + 32 - (d < 0 ? s : 0), p = st;
    for (;;) {
        const uint32_t diff = (simd_quad_window(pl, apos) ^ simd_quad_window(pl + 2 * SIMD_QUAD_PD, bpos)) |
                              (simd_quad_window(pl + SIMD_QUAD_PD, apos) ^ simd_quad_window(pl + 3 * SIMD_QUAD_PD, bpos));
        if (diff) {
            p += __builtin_ctz(diff);
            break;
        }
        p += 32, apos += 32, bpos += 32;
        if (p >= len) break;
    }
    return p < len ? p : len;
}
ASM_DEV int quad_min(int v) {
    int w = __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false); /* quad_perm [1,0,3,2] */
    v = w < v ? w : v;
    w = __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false); /* quad_perm [2,3,0,1] */
    return w < v ? w : v;
}

__global__ __launch_bounds__(64) void simd_ed_affine_quad_kernel(const uint4* __restrict__ planes, const uint32_t* __restrict__ lens,
                                                                 long n, int w4, int T, int af_t, int x, int o, int ext, int gm,
                                                                 int gi, int mode /* as simd_ed_affine_kernel */, OutMap out) {
    const bool all_start = mode == 1 || mode == 2, converge_rule = mode == 0 || mode == 2;
    constexpr int P = 16, PD = SIMD_QUAD_PD, PW = SIMD_QUAD_PW;
    typedef uint8_t EnT;
    extern __shared__ uint32_t s_afq[];
    const int t = threadIdx.x, pr = t >> 2, q = t & 3;
    const int rows = 2 * T + 3, mid = T + 1; /* lanes 1 .. 2T+1, guard rows 0 and 2T+2 (SIMD_ED.cpp:452-453) */
    const int slot = rows * P;
    const int ring_words = (int)(((size_t)(gm + 2 * gi) * slot * sizeof(EnT) + 3) / 4);
    uint32_t* const pl = s_afq + pr * PW;                               /* [P][4][PD] (+1) */
    EnT* const r_en = reinterpret_cast<EnT*>(s_afq + P * PW) + pr;    /* [gm][rows][P] */
    EnT* const r_ip = r_en + gm * slot;                                 /* [gi][rows][P] */
    EnT* const r_dp = r_ip + gi * slot;
    {
        uint32_t* const base = s_afq + P * PW;
        for (int w = t; w < ring_words; w += 64) base[w] = 0u;
    }
    const long i = (long)blockIdx.x * P + pr;
    const bool live = i < n;
    int len = 0;
    if (live) { /* thread q of the quad stages plane q: read plane 0/1, reference plane 0/1 */
        const uint32_t ln = lens[i];
        const int m = (int)(ln & 0xffffu), nn_ref = (int)(ln >> 16);
        len = m > 128 ? 128 : m;
        /* strncpy(A/B, ., length); the reference string ends at its own length (see simd_ed_affine_kernel) */
        const int keep = q < 2 ? len : (nn_ref < len ? nn_ref : len);
        const uint4 v = planes[((long)q * w4) * n + i];
        const uint32_t d4[4] = {v.x, v.y, v.z, v.w};
        uint32_t* dst = pl + q * PD;
        dst[0] = 0u, dst[5] = 0u;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int kb = keep - 32 * w;
            dst[1 + w] = kb >= 32 ? d4[w] : (kb <= 0 ? 0u : (d4[w] & ((1u << kb) - 1u)));
        }
    }
    leap_quad_fence();
    int result = live ? -2 : 0; /* -2: still running */
    if (live && !all_start) { /* e = 0: only the main lane has a start (ED_GLOBAL, SEMI_FREE_END; :474-477,497-516); the four
                                 threads compute the same value */
        const int e0 = simd_quad_extend(pl, 0, 0, len);
        if (q == 0) r_en[mid * P] = (EnT)(e0 + 2);
        if (e0 == len) result = converge_rule ? 1000000 : 0;
    }
    if (all_start) { /* LOCAL, SEMI_FREE_BEGIN: every lane starts at its distance from the main one; the lanes dealt to the quad */
        int exact = 0;
        if (live) {
            for (int l = 1 + q; l <= 2 * T + 1; l += 4) {
                const int dist = l < mid ? mid - l : l - mid;
                const int e0 = simd_quad_extend(pl, l - mid, dist, len);
                r_en[l * P] = (EnT)(e0 + 2);
                if (e0 == len) exact = 1;
            }
        }
        exact = quad_or(exact);
        if (live && exact) result = converge_rule ? 1000000 : 0;
    }
    leap_quad_fence();
    for (int e = 1; e <= af_t; e++) {
        if (__ballot(result == -2) == 0ull) break;
        int conv = 1000000;
        if (result == -2) {
            int dmax = e < o ? 0 : (e - o) / ext + 1; /* lanes a gap of this cost can have reached */
            dmax = (dmax > T || all_start) ? T : dmax;
            const EnT* const en_o = r_en + ((e - o) & (gm - 1)) * slot;
            const EnT* const en_x = r_en + ((e - x) & (gm - 1)) * slot;
            const EnT* const ip_e = r_ip + ((e - ext) & (gi - 1)) * slot;
            const EnT* const dp_e = r_dp + ((e - ext) & (gi - 1)) * slot;
            EnT* const en_w = r_en + (e & (gm - 1)) * slot;
            EnT* const ip_w = r_ip + (e & (gi - 1)) * slot;
            EnT* const dp_w = r_dp + (e & (gi - 1)) * slot;
            for (int l = mid - dmax + q; l <= mid + dmax; l += 4) {
                const int top = l >= mid ? 1 : 0, bot = l <= mid ? 1 : 0;
                const int e_up = (int)en_o[(l - 1) * P] - 2, i_up = (int)ip_e[(l - 1) * P] - 2;
                const int e_dn = (int)en_o[(l + 1) * P] - 2, d_dn = (int)dp_e[(l + 1) * P] - 2;
                const int own = (int)en_x[l * P] - 2;
                int inew = -2, dnew = -2;
                if (e_up >= 0 && e_up > i_up)
                    inew = e_up + top; /* :533-538 */
                else if (i_up >= 0)
                    inew = i_up + top; /* :539-544 */
                if (e_dn >= 0 && e_dn > d_dn)
                    dnew = e_dn + bot; /* :546-547 */
                else if (d_dn >= 0)
                    dnew = d_dn + bot; /* :548-549 */
                int st = own >= 0 ? own + 1 : -2; /* :551-558 */
                st = inew > st ? inew : st;
                st = dnew > st ? dnew : st;
                int enew = -2;
                if (st >= 0) {
                    enew = simd_quad_extend(pl, l - mid, st, len); /* :579-581 */
                    if (enew == len) { /* :589-603 */
                        const int diff = l < mid ? mid - l : l - mid;
                        const int tc = converge_rule ? e + (diff ? o + (diff - 1) * ext : 0) : e;
                        if ((!converge_rule || tc <= af_t) && tc < conv) conv = tc;
                    }
                }
                en_w[l * P] = (EnT)(enew + 2), ip_w[l * P] = (EnT)(inew + 2), dp_w[l * P] = (EnT)(dnew + 2);
            }
        }
        conv = quad_min(conv);
        if (conv != 1000000 && result == -2) result = conv; /* ED_pass: the generation loop ends (:609-610) */
        leap_quad_fence();
    }
    if (live && q == 0) out.put(i, result == -2 ? -1 : result);
}

static inline size_t simd_quad_lds(int T, int gm, int gi) {
    return (size_t)16 * SIMD_QUAD_PW * sizeof(uint32_t) + (((size_t)(gm + 2 * gi) * (2 * T + 3) * 16 + 3) & ~(size_t)3);
}

// init_affine's SHD_enable / SHD_threshold (SIMD_ED.cpp:435,445-446): run_affine starts with bit_vec_filter_avx(hamming_masks + 1,
// buffer_length, SHD_threshold) (:489-492), the mask-array filter of simd_ed_kernel (S3) over the FIRST 2*SHD_threshold+1 lane
// masks — lanes -gap .. -gap + 2*SHD_threshold, each cut by the begin mask of |j - SHD_threshold|: centred on the main lane only
// when SHD_threshold equals the gap threshold.  A rejected pair does not pass and nothing else changes, so this runs as a
// second, cheap kernel behind the affine one and overwrites the verdicts of the pairs it rejects.
template <int W64>
__global__ __launch_bounds__(ASM_BLOCK) void simd_affine_shd_kernel(const uint4* __restrict__ planes, const uint32_t* __restrict__ lens,
                                                                    long n, int w4, int gap_t, int shd_t, OutMap out) {
    const long i = (long)blockIdx.x * ASM_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int m = (int)(lens[i] & 0xffffu), nn_ref = (int)(lens[i] >> 16);
    const int len = m > 64 * W64 ? 64 * W64 : m;
    VW<W64> A0, A1, B0, B1;
    load_planes<W64>(planes, n, w4, i, A0, A1, B0, B1);
    const VW<W64> lm = vw_low_ones<W64>(len), lb = vw_low_ones<W64>(nn_ref < len ? nn_ref : len);
#pragma unroll
    for (int q = 0; q < W64; q++) A0.w[q] &= lm.w[q], A1.w[q] &= lm.w[q], B0.w[q] &= lb.w[q], B1.w[q] &= lb.w[q];
    VW<W64> diff = lm;
    for (int j = 0; j <= 2 * shd_t; j++) {
        const int s = j < shd_t ? shd_t - j : j - shd_t;
        const VW<W64> tm = s ? vw_from<W64>(s) : vw_low_ones<W64>(255);
        const VW<W64> h = simd_lane_mask<W64>(A0, A1, B0, B1, j - gap_t);
#pragma unroll
        for (int q = 0; q < W64; q++) diff.w[q] &= h.w[q] & tm.w[q];
    }
    if (shd_popcount<W64>(diff) > shd_t) out.put(i, -1);
}

// clean mode: every pair judged alone — never reached: fail; exact: 0; reached: final_ED + lane distance if <= T
__global__ __launch_bounds__(ASM_BLOCK) void simd_ed_clean_kernel(int32_t* __restrict__ ev_to_ed, long n, int T) {
    const long i = (long)blockIdx.x * ASM_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int ev = ev_to_ed[i];
    int ed = -1;
    if (ev <= EV_EXACT) ed = 0;
    if (ev >= 0) {
        const int conv = (ev & 0xff) + (ev >> 8);
        ed = conv <= T ? conv : -1;
    }
    ev_to_ed[i] = ed;
}

// ED_modes LOCAL and SEMI_FREE_END: a pair passes exactly when a lane reached the end, and get_ED() is final_ED
// (SIMD_ED.cpp:348-351 does not apply, :748-753) — no state of an earlier pair is read
__global__ __launch_bounds__(ASM_BLOCK) void simd_ed_final_kernel(int32_t* __restrict__ ev_to_ed, long n) {
    const long i = (long)blockIdx.x * ASM_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int ev = ev_to_ed[i];
    ev_to_ed[i] = ev <= EV_EXACT ? 0 : (ev >= 0 ? (ev & 0xff) : -1);
}

// sequential mode, step 1: key of the (final_ED, lane) chain — pairs that set it carry fe | fd << 8, the others -1
__global__ __launch_bounds__(ASM_BLOCK) void simd_ed_setter_kernel(const int32_t* __restrict__ ev, long n,
                                                                   int32_t* __restrict__ key) {
    const long i = (long)blockIdx.x * ASM_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int e = ev[i];
    key[i] = e <= EV_EXACT ? ((EV_EXACT - e) << 8) : (e >= 0 ? e : -1); /* exact: final_ED = 0 on a lane EV_EXACT - e off the main one */
}

// step 2 (after the last-setter scan of key): converge_ED is rewritten by pairs that run to the end of run_levenshtein
// (reached or not) from the state in force; exact and rejected pairs leave it alone
__global__ __launch_bounds__(ASM_BLOCK) void simd_ed_converge_kernel(const int32_t* __restrict__ ev,
                                                                     const int32_t* __restrict__ state_after, long n,
                                                                     int init_fe, int init_fd, int32_t* __restrict__ conv_key) {
    const long i = (long)blockIdx.x * ASM_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int e = ev[i];
    int k = -1;
    if (e >= 0 || e == EV_NOT_REACHED) {
        const int s = state_after[i];
        k = s >= 0 ? (s & 0xff) + (s >> 8) : init_fe + init_fd;
    }
    conv_key[i] = k;
}

// step 3 (after the last-setter scan of conv_key): the verdicts
__global__ __launch_bounds__(ASM_BLOCK) void simd_ed_verdict_kernel(int32_t* __restrict__ ev_to_ed,
                                                                    const int32_t* __restrict__ conv_after, long n, int T,
                                                                    int init_conv) {
    const long i = (long)blockIdx.x * ASM_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int e = ev_to_ed[i];
    int ed = -1;
    if (e <= EV_EXACT) { /* passes; get_ED() is the converge_ED left by the pairs before it (S2) */
        const int c = i > 0 ? conv_after[i - 1] : -1;
        ed = c >= 0 ? c : init_conv;
    } else if (e != EV_REJECTED) {
        const int c = conv_after[i];
        ed = c <= T ? c : -1;
    }
    ev_to_ed[i] = ed;
}

struct LastSetter { /* scan operator: the most recent non-negative key */
    __host__ __device__ int32_t operator()(int32_t a, int32_t b) const { return b >= 0 ? b : a; }
};

// ---------------------------------------------------------------------------------------------------------------------
// SHD on the planes: bit_vec_filter_avx(read0, read1, ref0, ref1, length, max_error) (SHD.cpp:241-322).
// ---------------------------------------------------------------------------------------------------------------------
// the four in-byte windows of flip_false_zero (SHD.cpp:95-122): in bits i..i+3 of every byte, every zero between the lowest
// and the highest set bit is set (MASK_SRS, mask.cpp:427-432); the windows are applied one after the other
ASM_DEV u64 srs_round_word(u64 w) {
    const u64 NIB = 0x0f0f0f0f0f0f0f0full;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const u64 nb = (w >> i) & NIB;
        const u64 below = nb | (nb << 1) | (nb << 2) | (nb << 3); /* a set bit at or below */
        const u64 above = nb | (nb >> 1) | (nb >> 2) | (nb >> 3); /* a set bit at or above */
        w |= (below & above & NIB) << i;
    }
    return w;
}

template <int W64>
ASM_DEV VW<W64> shd_flip_false_zero(VW<W64> v) { /* SHD.cpp:95-143 */
#pragma unroll
    for (int q = 0; q < W64; q++) v.w[q] = srs_round_word(v.w[q]);
    VW<W64> sv = avx_away0<W64>(v, 4); /* the windows that straddle a byte boundary */
#pragma unroll
    for (int q = 0; q < W64; q++) sv.w[q] = srs_round_word(sv.w[q]);
#pragma unroll
    for (int q = 0; q < W64; q++) { /* shift_left_avx(., 4): back towards index 0, half by half */
        const u64 hi = (q & 1) ? 0ull : sv.w[q + 1];
        v.w[q] |= (sv.w[q] >> 4) | (hi << 60);
    }
    return v;
}

template <int W64>
__global__ __launch_bounds__(ASM_BLOCK) void shd_kernel(const uint4* __restrict__ planes, const uint32_t* __restrict__ lens,
                                                        long n, int w4, int max_error, OutMap out) {
    const long i = (long)blockIdx.x * ASM_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int m = (int)(lens[i] & 0xffffu), nn_ref = (int)(lens[i] >> 16);
    const int len = m > 64 * W64 ? 64 * W64 : m;
    VW<W64> A0, A1, B0, B1;
    load_planes<W64>(planes, n, w4, i, A0, A1, B0, B1);
    const VW<W64> lm = vw_low_ones<W64>(len), lb = vw_low_ones<W64>(nn_ref < len ? nn_ref : len); /* see simd_ed_kernel */
#pragma unroll
    for (int q = 0; q < W64; q++) A0.w[q] &= lm.w[q], A1.w[q] &= lm.w[q], B0.w[q] &= lb.w[q], B1.w[q] &= lb.w[q];
    VW<W64> diff;
#pragma unroll
    for (int q = 0; q < W64; q++) diff.w[q] = (A0.w[q] ^ B0.w[q]) | (A1.w[q] ^ B1.w[q]);
    diff = shd_flip_false_zero<W64>(diff);
    for (int j = 1; j <= max_error; j++) {
        const VW<W64> tm = vw_from<W64>(j);
        VW<W64> t = simd_lane_mask<W64>(A0, A1, B0, B1, j); /* read shifted by j against the reference */
#pragma unroll
        for (int q = 0; q < W64; q++) t.w[q] &= tm.w[q] & lm.w[q];
        t = shd_flip_false_zero<W64>(t);
#pragma unroll
        for (int q = 0; q < W64; q++) diff.w[q] &= t.w[q];
        t = simd_lane_mask<W64>(A0, A1, B0, B1, -j); /* reference shifted by j against the read */
#pragma unroll
        for (int q = 0; q < W64; q++) t.w[q] &= tm.w[q] & lm.w[q];
        t = shd_flip_false_zero<W64>(t);
#pragma unroll
        for (int q = 0; q < W64; q++) diff.w[q] &= t.w[q];
    }
    out.put(i, shd_popcount<W64>(diff) > max_error ? 0 : 1);
}
