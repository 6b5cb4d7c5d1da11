/* TEST INFRASTRUCTURE ONLY — oracle, part 2: the bit-parallel LEAP (SIMD_ED, Levenshtein mode, ED_GLOBAL) and the SHD
 * pre-filter of GASMA/benchmark/LEAP_SIMD, restated in plain C on 4 x 64-bit words.  See asm_oracle.h for the rules that
 * apply to everything under oracle/.  Pinned against oracle/_ref/libasm_ref_simd.so (the real sources compiled in place)
 * by tests/test_oracle_vs_reference.py and tests/golden/.
 *
 * Behaviour kept because the reference has it (each is visible in the compiled reference's outputs):
 *  S1  shift_right_avx / shift_left_avx move each 128-bit half on its own: nothing crosses bit 127 <-> 128
 *      (shift.cpp:33-61, _mm256_slli_si256 / _mm256_srli_si256 are per-lane byte shifts).
 *  S2  SIMD_ED keeps final_ED, final_lane_idx and converge_ED from pair to pair (SIMD_ED.cpp:258-268 resets only
 *      ED_pass / cur_ED): a pair that never reaches the end gets the verdict of the last pair that did
 *      (SIMD_ED.cpp:348-351), an exact pair returns before converge_ED is written (:291-296) so get_ED() is the
 *      previous pair's value.  ORC_FILTER_SEQUENTIAL carries that state; ORC_FILTER_CLEAN judges every pair alone
 *      (never reached -> fail, exact -> 0).
 *  S3  bit_vec_filter_avx(masks, length, max_error) amends its *mask*, not the data (SHD.cpp:353: flip_false_zero is
 *      applied to temp_mask after temp_diff was taken), i.e. no amendment at all; and for the main lane it reads table
 *      row -1 of MASK_AVX_BEG (:349-350) — the 32 bytes in front of the table, which in the reference library as GCC lays
 *      it out are the last row of MASK_AVX_END: bits 0..254 set.  (Layout dependent; modelled as built here.)
 *  S4  count_ID_length_avx with start_pos >= buffer_length returns buffer_length - start_pos (<= 0) (SIMD_ED.cpp:57-60).
 */
#include "asm_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef struct {
    uint64_t w[4];
} v256;

static inline uint64_t shl64(uint64_t v, int n) { return n >= 64 ? 0 : v << n; } /* x86 vector shifts: count >= 64 gives 0 */
static inline uint64_t shr64(uint64_t v, int n) { return n >= 64 ? 0 : v >> n; }

/* shift_right_avx (shift.cpp:33-46): bits move towards higher indexes */
static v256 avx_shr(v256 v, int n) {
    if (n >= 128) {
        v.w[2] = v.w[0], v.w[3] = v.w[1], v.w[0] = v.w[1] = 0;
        n %= 128;
    }
    if (n >= 64) {
        v.w[1] = v.w[0], v.w[3] = v.w[2], v.w[0] = v.w[2] = 0;
        n %= 64;
    }
    v256 c = {{0, v.w[0], 0, v.w[2]}};
    v256 r;
    for (int q = 0; q < 4; q++) r.w[q] = shl64(v.w[q], n) | shr64(c.w[q], 64 - n);
    return r;
}

/* shift_left_avx (shift.cpp:48-61): bits move towards index 0 */
static v256 avx_shl(v256 v, int n) {
    if (n >= 128) {
        v.w[0] = v.w[2], v.w[1] = v.w[3], v.w[2] = v.w[3] = 0;
        n %= 128;
    }
    if (n >= 64) {
        v.w[0] = v.w[1], v.w[2] = v.w[3], v.w[1] = v.w[3] = 0;
        n %= 64;
    }
    v256 c = {{v.w[1], 0, v.w[3], 0}};
    v256 r;
    for (int q = 0; q < 4; q++) r.w[q] = shr64(v.w[q], n) | shl64(c.w[q], 64 - n);
    return r;
}

static inline v256 v_and(v256 a, v256 b) {
    v256 r;
    for (int q = 0; q < 4; q++) r.w[q] = a.w[q] & b.w[q];
    return r;
}
static inline v256 v_or(v256 a, v256 b) {
    v256 r;
    for (int q = 0; q < 4; q++) r.w[q] = a.w[q] | b.w[q];
    return r;
}
static inline v256 v_xor(v256 a, v256 b) {
    v256 r;
    for (int q = 0; q < 4; q++) r.w[q] = a.w[q] ^ b.w[q];
    return r;
}
static v256 low_ones(int len) { /* MASK_AVX_END row `len` (mask.cpp:168-425); all ones from 256 on */
    v256 r;
    for (int q = 0; q < 4; q++) {
        int rel = len - 64 * q;
        r.w[q] = rel <= 0 ? 0 : (rel >= 64 ? ~0ull : ((1ull << rel) - 1));
    }
    return r;
}
static v256 not_low_ones(int cnt) { /* MASK_AVX_BEG row cnt-1 (mask.cpp:149-166): the first cnt bits cleared */
    v256 r = low_ones(cnt);
    for (int q = 0; q < 4; q++) r.w[q] = ~r.w[q];
    return r;
}

/* 2-bit planes as avx_convert2bit leaves them (bit_convert.cpp:335-479): C = 01, G = 10, T = 11, any other byte 00;
 * bit p of plane b = bit b of the code of character p.  Characters at and beyond `len` are NUL padding (code 00). */
static void planes_of(const char* s, int len, v256* p0, v256* p1) {
    memset(p0, 0, sizeof *p0);
    memset(p1, 0, sizeof *p1);
    for (int p = 0; p < len && p < 256; p++) {
        int code = s[p] == 'C' ? 1 : (s[p] == 'G' ? 2 : (s[p] == 'T' ? 3 : 0));
        if (code & 1) p0->w[p >> 6] |= 1ull << (p & 63);
        if (code & 2) p1->w[p >> 6] |= 1ull << (p & 63);
    }
}

/* number of nibbles' runs of ones: POPCOUNT_SHD (popcount.cpp:44-76) summed over all 64 nibbles */
static int popcount_shd(v256 v) {
    static const uint8_t T[16] = {0, 1, 1, 1, 1, 2, 2, 1, 1, 2, 2, 2, 1, 2, 1, 1};
    int s = 0;
    for (int q = 0; q < 4; q++)
        for (int b = 0; b < 64; b += 4) s += T[(v.w[q] >> b) & 15];
    return s;
}

/* one round of the four in-byte windows of flip_false_zero (SHD.cpp:95-122): window bits i..i+3 of every byte are
 * looked up in MASK_SRS (mask.cpp:427-432: every zero between the lowest and highest set bit of the nibble is set) */
static v256 srs_round(v256 v) {
    static const uint8_t SRS[16] = {0x0, 0x1, 0x2, 0x3, 0x4, 0x7, 0x6, 0x7, 0x8, 0xf, 0xe, 0xf, 0xc, 0xf, 0xe, 0xf};
    for (int i = 0; i < 4; i++)
        for (int q = 0; q < 4; q++) {
            uint64_t add = 0;
            for (int b = 0; b < 8; b++) {
                unsigned byte = (unsigned)(v.w[q] >> (8 * b)) & 0xffu;
                add |= (uint64_t)((SRS[(byte >> i) & 15] << i) & 0xff) << (8 * b);
            }
            v.w[q] |= add;
        }
    return v;
}

static v256 flip_false_zero(v256 v) { /* SHD.cpp:95-143 */
    v = srs_round(v);
    v256 sv = srs_round(avx_shr(v, 4)); /* the windows that straddle a byte boundary */
    return v_or(avx_shl(sv, 4), v);
}

/* bit_vec_filter_avx(read planes, ref planes, length, max_error), SHD.cpp:241-322 */
static int shd_planes(v256 a0, v256 a1, v256 b0, v256 b1, int length, int max_error) {
    const v256 mask = low_ones(length >= 256 ? 256 : length);
    a0 = v_and(a0, mask), a1 = v_and(a1, mask), b0 = v_and(b0, mask), b1 = v_and(b1, mask);
    v256 diff = flip_false_zero(v_or(v_xor(a0, b0), v_xor(a1, b1)));
    for (int j = 1; j <= max_error; j++) {
        const v256 tm = v_and(not_low_ones(j), mask);
        v256 t = v_or(v_xor(avx_shr(a0, j), b0), v_xor(avx_shr(a1, j), b1));
        diff = v_and(diff, flip_false_zero(v_and(t, tm)));
        t = v_or(v_xor(avx_shr(b0, j), a0), v_xor(avx_shr(b1, j), a1));
        diff = v_and(diff, flip_false_zero(v_and(t, tm)));
    }
    return popcount_shd(diff) > max_error ? 0 : 1;
}

int orc_shd_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs, const uint32_t* ref_off,
                  int max_error, int32_t* pass) {
    if (max_error < 0 || max_error > 16) return -1; /* MAX_ERROR_AVX rows in MASK_AVX_BEG (mask.h:21) */
    for (int64_t i = 0; i < n; i++) {
        const int m = (int)(read_off[i + 1] - read_off[i]), nn = (int)(ref_off[i + 1] - ref_off[i]);
        const int length = m > 256 ? 256 : m;
        v256 a0, a1, b0, b1;
        planes_of(reads + read_off[i], length, &a0, &a1);
        planes_of(refs + ref_off[i], nn, &b0, &b1);
        pass[i] = shd_planes(a0, a1, b0, b1, length, max_error);
    }
    return 0;
}

/* count_ID_length_avx (SIMD_ED.cpp:10-61) */
static int count_id(v256 mask, int start, int len) {
    if (start >= len) return len - start; /* S4 */
    const v256 s = avx_shl(mask, start);
    int res = 0;
    for (int i = 0; i <= (len - start - 1) / 64; i++) {
        if (s.w[i] == 0) {
            res += 64;
        } else {
            res += __builtin_ctzll(s.w[i]);
            break;
        }
    }
    return res < len - start ? res : len - start;
}

#define SIMD_MAX_T 32

/* state[0..2] = final_ED, |final_lane_idx - mid_lane|, converge_ED carried into the first pair (sequential mode) */
int orc_simd_ed_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs, const uint32_t* ref_off,
                      int ed_t, int shd_enable, int mode, const int32_t* state, int32_t* ed, int32_t* ed_raw,
                      uint8_t* pass) {
    return orc_simd_ed_edmode_batch(n, reads, read_off, refs, ref_off, ed_t, shd_enable, mode, ORC_LEAP_GLOBAL, state, ed, ed_raw, pass);
}

/* ... with init_levenshtein's ED_modes argument (numbering of ORC_LEAP_*).  LOCAL and SEMI_FREE_BEGIN: every lane is live from
 * generation 0, starting at its distance from the main lane (SIMD_ED.cpp:246-266: start[i][0] = ED, cur_ED[i] = 0); GLOBAL and
 * SEMI_FREE_END: lane l joins at generation |l - mid|.  GLOBAL and SEMI_FREE_BEGIN end with converge_ED = final_ED + lane
 * distance <= ED_t (:348-351, the stale-state rule S2 included) and get_ED() = converge_ED; LOCAL and SEMI_FREE_END pass exactly
 * when a lane reaches the end and get_ED() = final_ED (:748-753) — no state of an earlier pair is ever read there. */
int orc_simd_ed_edmode_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs, const uint32_t* ref_off,
                             int ed_t, int shd_enable, int mode, int ed_mode, const int32_t* state, int32_t* ed, int32_t* ed_raw,
                             uint8_t* pass) {
    if (ed_t < 0 || ed_t > SIMD_MAX_T || (shd_enable && ed_t > 16)) return -1;
    if (ed_mode < ORC_LEAP_GLOBAL || ed_mode > ORC_LEAP_SEMI_FREE_END) return -1;
    const int all_start = ed_mode == ORC_LEAP_LOCAL || ed_mode == ORC_LEAP_SEMI_FREE_BEGIN;
    const int converge_rule = ed_mode == ORC_LEAP_GLOBAL || ed_mode == ORC_LEAP_SEMI_FREE_BEGIN;
    const int lanes = 2 * ed_t + 3, mid = ed_t + 1; /* SIMD_ED.cpp:222-223; lanes 0 and lanes-1 are guards */
    int fe = state ? state[0] : 0, fd = state ? state[1] : 0, conv = state ? state[2] : 0;
    v256 hm[2 * SIMD_MAX_T + 3];
    int end[2 * SIMD_MAX_T + 3][SIMD_MAX_T + 1];
    for (int64_t i = 0; i < n; i++) {
        const int m = (int)(read_off[i + 1] - read_off[i]), nn = (int)(ref_off[i + 1] - ref_off[i]);
        const int len = m > 256 ? 256 : m; /* main.cpp:131-132; SIMD_ED.cpp:140-151 */
        if (mode == ORC_FILTER_CLEAN) fe = ed_t + 1, fd = 0, conv = 0;
        v256 a0, a1, b0, b1;
        planes_of(reads + read_off[i], len, &a0, &a1);
        planes_of(refs + ref_off[i], nn < len ? nn : len, &b0, &b1); /* strncpy(B, ref, length), SIMD_ED.cpp:147 */
        for (int l = 1; l < lanes - 1; l++) { /* calculate_masks, SIMD_ED.cpp:180-212 */
            const int s = abs(l - mid);
            const v256 x0 = l > mid ? avx_shr(a0, s) : a0, x1 = l > mid ? avx_shr(a1, s) : a1;
            const v256 y0 = l < mid ? avx_shr(b0, s) : b0, y1 = l < mid ? avx_shr(b1, s) : b1;
            hm[l] = v_or(v_xor(x0, y0), v_xor(x1, y1));
        }
        int ok = 0, reached = 0;
        if (shd_enable) { /* bit_vec_filter_avx(hamming_masks + 1, buffer_length, ED_t): SHD.cpp:324-372, S3 */
            const v256 lm = low_ones(len >= 256 ? 256 : len);
            v256 diff = low_ones(256);
            for (int l = 1; l < lanes - 1; l++) {
                const int s = abs(l - mid);
                const v256 tm = v_and(s ? not_low_ones(s) : low_ones(255), lm);
                diff = v_and(diff, v_and(hm[l], tm));
            }
            if (popcount_shd(diff) > ed_t) { /* SIMD_ED.cpp:270-273: rejected, nothing else changes */
                pass[i] = 0, ed[i] = -1;
                if (ed_raw) ed_raw[i] = conv;
                continue;
            }
        }
        for (int l = 0; l < lanes; l++)
            for (int e = 0; e <= ed_t; e++) end[l][e] = -2; /* entries never written hold init_levenshtein's -2 */
        int exact0 = 0, exact_d = 0;
        for (int l = 1; l < lanes - 1 && !exact0; l++) { /* SIMD_ED.cpp:277-299: the lanes with cur_ED == 0, ascending */
            const int dist = abs(l - mid);
            if (dist != 0 && !all_start) continue;
            end[l][0] = dist + count_id(hm[l], dist, len);
            if (end[l][0] == len) exact0 = 1, exact_d = dist;
        }
        if (exact0) {
            fe = 0, fd = exact_d; /* returns with ED_pass = true; converge_ED is NOT rewritten (S2) */
            pass[i] = 1, ed[i] = converge_rule ? conv : 0;
            if (ed_raw) ed_raw[i] = converge_rule ? conv : 0;
            continue;
        }
        for (int e = 1; e <= ed_t && !reached; e++) { /* SIMD_ED.cpp:301-346 */
            for (int l = 1; l < lanes - 1; l++) {
                if (!all_start && abs(l - mid) > e) continue; /* cur_ED[l] == e */
                const int top = l >= mid, bot = l <= mid;
                int st = end[l][e - 1] + 1;
                if (end[l - 1][e - 1] + top > st) st = end[l - 1][e - 1] + top;
                if (end[l + 1][e - 1] + bot > st) st = end[l + 1][e - 1] + bot;
                end[l][e] = st + count_id(hm[l], st, len);
                if (end[l][e] == len) {
                    fe = e, fd = abs(l - mid), reached = 1;
                    break;
                }
            }
        }
        if (!converge_rule) { /* LOCAL, SEMI_FREE_END: ED_pass as the sweep left it, get_ED() = final_ED */
            ok = reached;
            pass[i] = (uint8_t)ok, ed[i] = ok ? fe : -1;
            if (ed_raw) ed_raw[i] = fe;
            continue;
        }
        conv = fe + fd; /* SIMD_ED.cpp:348-351 — with stale fe/fd when the end was never reached (S2) */
        ok = conv <= ed_t;
        pass[i] = (uint8_t)ok, ed[i] = ok ? conv : -1;
        if (ed_raw) ed_raw[i] = conv;
    }
    return 0;
}

/* SIMD_ED in affine mode (init_affine(gap_t, af_t, ED_GLOBAL, x, o, e) / load_reads / calculate_masks / reset_affine /
 * run_affine / check_pass / get_ED, SIMD_ED.cpp:435-616,744-753), CLEAN: every pair starts from the tables init_affine leaves
 * (-2 everywhere, start[mid][0] = 0).  The reference as run re-uses the tables of the pair before — run_affine writes I_pos,
 * D_pos and end only where its conditions hold, reset_affine touches none of them — so its verdicts depend on everything the
 * object has seen; equal to this function for the first pair after init_affine, which is how the pin drives it
 * (oracle/ref_harness_simd.cpp: init_affine before every pair).  SHD is off (init_affine's default).
 * ed[i] = get_ED() = converge_ED when the pair passes (1000000, reset_affine's value, for a pair that reaches the end at e = 0:
 * run_affine returns before converge_ED is written, :509-514), -1 when it does not. */
#define SIMD_AF_MAX 512
int orc_simd_ed_affine_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs, const uint32_t* ref_off,
                             int gap_t, int af_t, int x, int o, int ext, int32_t* ed, uint8_t* pass) {
    return orc_simd_ed_affine_shd_batch(n, reads, read_off, refs, ref_off, gap_t, af_t, x, o, ext, 0, 0, ed, pass);
}

/* The same with init_affine's last two arguments (SHD_enable, SHD_threshold; SIMD_ED.cpp:435,445-446): run_affine starts with
 * bit_vec_filter_avx(hamming_masks + 1, buffer_length, SHD_threshold) (:489-492) — the mask-array filter of SHD.cpp:334-372
 * over the FIRST 2*SHD_threshold+1 lane masks, j = 0 .. 2*SHD_threshold, each cut by the begin mask of |j - SHD_threshold|.
 * Those are the lanes -gap_t .. -gap_t + 2*SHD_threshold: centred on the main lane only when SHD_threshold == gap_t, and
 * beyond the object's array when it is larger (refused here).  A rejected pair does not pass; nothing else changes. */
int orc_simd_ed_affine_shd_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs, const uint32_t* ref_off,
                                 int gap_t, int af_t, int x, int o, int ext, int shd_enable, int shd_t, int32_t* ed, uint8_t* pass) {
    return orc_simd_ed_affine_mode_batch(n, reads, read_off, refs, ref_off, gap_t, af_t, x, o, ext, shd_enable, shd_t, ORC_LEAP_GLOBAL,
                                         ed, pass);
}

/* ... and with init_affine's ED_modes argument (own numbering as orc_leap_mode_batch: 0 GLOBAL, 1 LOCAL, 2 SEMI_FREE_BEGIN,
 * 3 SEMI_FREE_END).  LOCAL and SEMI_FREE_BEGIN give every lane a start at generation 0 (SIMD_ED.cpp:476-478) and generation 0
 * sweeps all of them in ascending order (:497-516); LOCAL and SEMI_FREE_END accept any lane that reaches the end, without
 * converge_ED's lane term and threshold (:589-610), and get_ED() returns final_ED there, converge_ED otherwise (:748-753) — so a
 * pair exact at generation 0 reads 0 in LOCAL / SEMI_FREE_END and reset_affine's 1000000 in the other two. */
int orc_simd_ed_affine_mode_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs, const uint32_t* ref_off,
                                  int gap_t, int af_t, int x, int o, int ext, int shd_enable, int shd_t, int mode, int32_t* ed,
                                  uint8_t* pass) {
    if (gap_t < 1 || gap_t > SIMD_MAX_T || af_t < 1 || af_t > SIMD_AF_MAX || x < 1 || o < 1 || ext < 1) return -1;
    if (shd_enable && (shd_t < 0 || shd_t > gap_t || shd_t > 16)) return -1;
    if (mode < ORC_LEAP_GLOBAL || mode > ORC_LEAP_SEMI_FREE_END) return -1;
    const int all_start = mode == ORC_LEAP_LOCAL || mode == ORC_LEAP_SEMI_FREE_BEGIN;
    const int converge_rule = mode == ORC_LEAP_GLOBAL || mode == ORC_LEAP_SEMI_FREE_BEGIN;
    const int lanes = 2 * gap_t + 3, mid = gap_t + 1; /* SIMD_ED.cpp:452-453 */
    v256 hm[2 * SIMD_MAX_T + 3];
    typedef int row_t[SIMD_AF_MAX + 1];
    row_t* st_ = (row_t*)malloc(sizeof(row_t) * lanes * 4);
    if (!st_) return -1;
    row_t *start = st_, *end = st_ + lanes, *ip = st_ + 2 * lanes, *dp = st_ + 3 * lanes;
    for (int64_t i = 0; i < n; i++) {
        const int m = (int)(read_off[i + 1] - read_off[i]), nn = (int)(ref_off[i + 1] - ref_off[i]);
        const int len = m > 256 ? 256 : m;
        v256 a0, a1, b0, b1;
        planes_of(reads + read_off[i], len, &a0, &a1);
        planes_of(refs + ref_off[i], nn < len ? nn : len, &b0, &b1);
        for (int l = 1; l < lanes - 1; l++) { /* calculate_masks, SIMD_ED.cpp:180-212 */
            const int s = abs(l - mid);
            const v256 x0 = l > mid ? avx_shr(a0, s) : a0, x1 = l > mid ? avx_shr(a1, s) : a1;
            const v256 y0 = l < mid ? avx_shr(b0, s) : b0, y1 = l < mid ? avx_shr(b1, s) : b1;
            hm[l] = v_or(v_xor(x0, y0), v_xor(x1, y1));
        }
        if (shd_enable) { /* :489-492; the filter as in orc_simd_ed_batch (S3), over masks hm[1 + j] */
            const v256 lm = low_ones(len >= 256 ? 256 : len);
            v256 diff = low_ones(256);
            for (int j = 0; j <= 2 * shd_t; j++) {
                const int s = abs(j - shd_t);
                const v256 tm = v_and(s ? not_low_ones(s) : low_ones(255), lm);
                diff = v_and(diff, v_and(hm[1 + j], tm));
            }
            if (popcount_shd(diff) > shd_t) {
                pass[i] = 0, ed[i] = -1;
                continue;
            }
        }
        for (int l = 0; l < lanes; l++) /* init_affine, :467-478 */
            for (int e = 0; e <= af_t; e++) start[l][e] = end[l][e] = ip[l][e] = dp[l][e] = -2;
        int ok = 0, conv = 1000000, fin = 0; /* reset_affine, :483-486 */
        int exact0 = 0;
        for (int l = 1; l < lanes - 1 && !exact0; l++) { /* :476-478 and :497-516 */
            const int dist = abs(l - mid);
            if (dist != 0 && !all_start) continue;
            start[l][0] = dist;
            end[l][0] = count_id(hm[l], dist, len) + dist;
            if (end[l][0] == len) exact0 = 1;
        }
        if (exact0) {
            pass[i] = 1, ed[i] = converge_rule ? conv : 0; /* get_ED(): converge_ED (never written here) or final_ED = 0 */
            continue;
        }
        for (int e = 1; e <= af_t && !ok; e++) { /* :518-614 */
            for (int l = 1; l < lanes - 1; l++) {
                const int top = l >= mid, bot = l <= mid;
                if (e >= o && end[l - 1][e - o] >= 0 && end[l - 1][e - o] > (e >= ext ? ip[l - 1][e - ext] : ip[l - 1][0]))
                    ip[l][e] = end[l - 1][e - o] + top;
                else if (e >= ext && ip[l - 1][e - ext] >= 0)
                    ip[l][e] = ip[l - 1][e - ext] + top;
                if (e >= o && end[l + 1][e - o] >= 0 && end[l + 1][e - o] > (e >= ext ? dp[l + 1][e - ext] : dp[l + 1][0]))
                    dp[l][e] = end[l + 1][e - o] + bot;
                else if (e >= ext && dp[l + 1][e - ext] >= 0)
                    dp[l][e] = dp[l + 1][e - ext] + bot;
                start[l][e] = -2;
                if (e >= x && end[l][e - x] >= 0) start[l][e] = end[l][e - x] + 1;
                if (ip[l][e] > start[l][e]) start[l][e] = ip[l][e];
                if (dp[l][e] > start[l][e]) start[l][e] = dp[l][e];
                if (start[l][e] >= 0) {
                    end[l][e] = start[l][e] + count_id(hm[l], start[l][e], len);
                    if (end[l][e] == len) {
                        const int diff = abs(mid - l);
                        const int tc = e + (diff ? o + (diff - 1) * ext : 0);
                        if (!converge_rule) ok = 1, fin = e;
                        else if (tc <= af_t && tc < conv) ok = 1, conv = tc;
                    }
                }
            }
        }
        pass[i] = (uint8_t)ok, ed[i] = ok ? (converge_rule ? conv : fin) : -1;
    }
    free(st_);
    return 0;
}
