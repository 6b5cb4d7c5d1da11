// 128-bit bit-vector primitives for the device kernels (gfx950).
//
// Device counterpart of the reference's int_128bit (GASMA/utils.h:49-271): same value semantics, including
// the corner cases the Greedy aligner relies on (shift counts outside [0,127] give 0, first_one of an empty
// vector is 128, pop_count_between of an empty or inverted range is 0).  Written for the CDNA4 scalar/vector
// integer pipes: a vector is two 64-bit halves held in VGPRs; `v_lshrrev_b64`/`v_lshlrev_b64`, `v_ffbl_b32`
// and `v_bcnt_u32_b32` do the work, selects replace branches so a wave never diverges inside a primitive.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ASM_DEV __device__ __forceinline__

typedef unsigned long long u64;

struct V128 {
    u64 lo, hi;
};

ASM_DEV V128 v_make(u64 lo, u64 hi) {
    V128 r;
    r.lo = lo;
    r.hi = hi;
    return r;
}
ASM_DEV V128 v_from_uint4(uint4 q) { return v_make((u64)q.x | ((u64)q.y << 32), (u64)q.z | ((u64)q.w << 32)); }
ASM_DEV V128 v_and(V128 a, V128 b) { return v_make(a.lo & b.lo, a.hi & b.hi); }
ASM_DEV V128 v_or(V128 a, V128 b) { return v_make(a.lo | b.lo, a.hi | b.hi); }
ASM_DEV V128 v_xor(V128 a, V128 b) { return v_make(a.lo ^ b.lo, a.hi ^ b.hi); }
ASM_DEV V128 v_not(V128 a) { return v_make(~a.lo, ~a.hi); }

// utils.h:143-153 "shift_left": bits move toward index 0; 0 for s outside [0,127].
// Two stages: a 128-bit funnel shift by s mod 64 (the hardware takes 64-bit shift counts mod 64, and "<< (63 - r) << 1"
// is a shift by 64 - r that is also right for r = 0), then a word move when s >= 64.
ASM_DEV V128 v_toward0(V128 v, int s) {
    const int r = s & 63;
    const u64 lo1 = (v.lo >> r) | ((v.hi << (63 - r)) << 1);
    const u64 hi1 = v.hi >> r;
    const bool big = (s & 64) != 0;
    const bool dead = (unsigned)s >= 128u;
    return v_make(dead ? 0ull : (big ? hi1 : lo1), (dead || big) ? 0ull : hi1);
}

// utils.h:131-141 "shift_right": bits move away from index 0; 0 for s outside [0,127].
ASM_DEV V128 v_away0(V128 v, int s) {
    const int r = s & 63;
    const u64 hi1 = (v.hi << r) | ((v.lo >> (63 - r)) >> 1);
    const u64 lo1 = v.lo << r;
    const bool big = (s & 64) != 0;
    const bool dead = (unsigned)s >= 128u;
    return v_make((dead || big) ? 0ull : lo1, dead ? 0ull : (big ? lo1 : hi1));
}

// utils.h:168-182: index of the lowest set bit, 128 when there is none.
ASM_DEV int v_first_one(V128 v) {
    int a = v.lo ? __builtin_ctzll(v.lo) : 64;
    int b = v.hi ? __builtin_ctzll(v.hi) : 64;
    return v.lo ? a : 64 + b;
}
// utils.h:187-191
ASM_DEV int v_first_zero(V128 v) { return v_first_one(v_not(v)); }

// first_one(l >> fz) for fz = first_zero(l), without the second shift: l has ones exactly below bit fz and a zero
// at fz, so l + 1 clears that run and sets bit fz, and (l + 1) & l keeps only the ones above fz.  128 when none.
ASM_DEV int v_next_one_after_zero_run(V128 l, int fz) {
    const u64 lo1 = l.lo + 1ull;
    const u64 hi1 = l.hi + (lo1 == 0ull ? 1ull : 0ull);
    const int p = v_first_one(v_make(lo1 & l.lo, hi1 & l.hi));
    return p == 128 ? 128 : p - fz;
}

// first set bit at index >= from, 128 when there is none (or from >= 128): only the word `from` falls into is shifted
ASM_DEV int v_next_one_from(V128 v, int from) {
    const bool low = from < 64;
    const u64 y = (low ? v.lo : v.hi) >> (from & 63);
    int res = (low && v.hi) ? 64 + __builtin_ctzll(v.hi) : 128;
    if (y) res = from + __builtin_ctzll(y);
    return from >= 128 ? 128 : res;
}

// The next highway of a (flipped) lane vector at or after column `start` (hurdle_matrix.h:299-303):
//   l = shift_left(v, start); fz = first_zero(l); nx = first_one(shift_left(l, fz))
// as two forward scans on the unshifted vector: a0 = first zero at or after start (128 when none: the zeros that the
// shift brings in at the top), a1 = first one after it.  start >= 128 is the empty shifted vector (fz = 0, nx = 128).
ASM_DEV void v_highway_from(V128 v, int start, int& fz, int& nx) {
    const int a0 = v_next_one_from(v_not(v), start);
    const int a1 = v_next_one_from(v, a0);
    fz = (unsigned)start >= 128u ? 0 : a0 - start;
    nx = a1 == 128 ? 128 : a1 - a0;
}

// The same two scans for a caller that keeps, per vector, the complement and the two "first one in the upper word" fall-backs
// (fb = 64 + ctz(hi), or 128 for an empty upper word) in registers: v_ffbl_b32 gives 0xFFFFFFFF for an empty word, and with
// saturating adds an empty shifted word turns into a candidate that loses every min() — 10 instructions per scan instead of 14,
// no compare-and-select chain (the wave-per-pair Greedy kernel runs both scans in every step, whatever its lanes need).
ASM_DEV unsigned v_ffbl_raw(unsigned x) {
    unsigned r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
ASM_DEV unsigned v_upper_fallback(V128 v) { return v.hi ? 64u + (unsigned)__builtin_ctzll(v.hi) : 128u; }
ASM_DEV int v_next_one_from_fb(V128 v, unsigned fb_hi, int from) { /* = v_next_one_from(v, from) for from >= 0 */
    const bool low = from < 64;
    const u64 y = (low ? v.lo : v.hi) >> (from & 63);
    const unsigned c = min(v_ffbl_raw((unsigned)y), __builtin_elementwise_add_sat(v_ffbl_raw((unsigned)(y >> 32)), 32u));
    const unsigned cand = __builtin_elementwise_add_sat((unsigned)from, c);
    return (int)min(cand, low ? fb_hi : 128u);
}
ASM_DEV void v_highway_from_fb(V128 v, V128 nv /* ~v */, unsigned fb_v, unsigned fb_nv, int start, int& fz, int& nx) {
    const int a0 = v_next_one_from_fb(nv, fb_nv, start);
    const int a1 = v_next_one_from_fb(v, fb_v, a0);
    fz = (unsigned)start >= 128u ? 0 : a0 - start;
    nx = a1 == 128 ? 128 : a1 - a0;
}

ASM_DEV int v_popcount(V128 v) { return __popcll(v.lo) + __popcll(v.hi); }

// ones at index >= s (any s >= 0; 0 from 128 on).  A popcount does not care where the surviving bits end up, so instead of
// a 128-bit funnel shift this picks the word that s falls into, shifts that one word (the hardware takes the count mod 64)
// and adds the whole upper word when s is in the lower one.
ASM_DEV int v_ones_from(V128 v, int s) {
    const bool low = s < 64;
    const u64 y = (low ? v.lo : v.hi) >> (s & 63);
    const int c = __popcll(y) + (low ? __popcll(v.hi) : 0);
    return s >= 128 ? 0 : c;
}

// utils.h:263-270 pop_count_between = popcount(shift_right(shift_left(v, from), from + 128 - to)): the ones in [from,to);
// 0 when `from` or from+128-to falls outside [0,127] (i.e. unless 0 <= from <= 127 and from < to <= from+128).
// `ones_from_to` = v_ones_from(v, to), which callers that count several ranges with one `to` compute once.
ASM_DEV int v_pop_between_pre(V128 v, int from, int to, int ones_from_to) {
    const bool ok = (unsigned)from < 128u && (unsigned)(to - from - 1) < 128u;
    return ok ? v_ones_from(v, from) - ones_from_to : 0;
}
ASM_DEV int v_pop_between(V128 v, int from, int to) { return v_pop_between_pre(v, from, to, v_ones_from(v, to)); }

// utils.h:200-216 with threshold 1: a set bit survives only next to another set bit.
ASM_DEV V128 v_flip_short_hurdles1(V128 v) {
    V128 a = v_toward0(v, 1), b = v_away0(v, 1);
    return v_and(v, v_or(a, b));
}

// utils.h:576-579
ASM_DEV int lane_penalty(int a, int b, int o, int e) {
    int d = a - b;
    d = d < 0 ? -d : d;
    return d == 0 ? 0 : o + e * (d - 1);
}

// utils.h:587-593: same sign (l1 * l2 >= 0) -> max(|l1| - |l2|, 0), opposite signs -> |l1|.  The sign test is an xor instead of
// the reference's product (a quarter-rate v_mul_lo): the two differ only when one lane is 0, and there both branches give the
// same value (l1 == 0: both 0; l2 == 0: both |l1|).
ASM_DEV int fwd_col(int l1, int l2) {
    int a1 = l1 < 0 ? -l1 : l1, a2 = l2 < 0 ? -l2 : l2;
    int same = a1 > a2 ? a1 - a2 : 0;
    return ((l1 ^ l2) < 0) ? a1 : same;
}

// hurdle_matrix.h:58-68
ASM_DEV int lane_destination(int m, int n, int lane) {
    int r;
    if (m >= n) {
        r = lane > 0 ? n - lane : (lane >= n - m ? n : m + lane);
    } else {
        r = lane < 0 ? m + lane : (lane <= n - m ? m : n - lane);
    }
    return r;
}

// mask with bits [0,len) set, len in [0,128]
ASM_DEV V128 v_low_ones(int len) {
    u64 lo = len >= 64 ? ~0ull : ((1ull << len) - 1ull);
    int h = len - 64;
    u64 hi = h <= 0 ? 0ull : (h >= 64 ? ~0ull : ((1ull << h) - 1ull));
    return v_make(lo, hi);
}

// Where a kernel's i-th pair writes its penalty: with length-bucketed batches the kernels run over a bucket's pairs
// in bucket order and `order` maps the bucket slot back to the caller's pair index (null = identity).
struct OutMap {
    int32_t* out;
    const uint32_t* order;
    ASM_DEV long index(long i) const { return order ? (long)order[i] : i; }
    ASM_DEV void put(long i, int v) const { out[index(i)] = v; }
};

// Optional CIGAR output of the Greedy kernels (hurdle_matrix::_update_CIGAR, GASMA/hurdle_matrix.h:238-251): per pair
// a row of `cap` uint16 entries (count << 3 | op, op 0 = 'M', 1 = 'I', 2 = 'D'; the NW traceback adds 3 = '=', 4 = 'X') and the number of entries produced
// (which may exceed cap: the row is then truncated and the caller sees nops > cap).  ops == null disables it.
struct CigarSink {
    uint16_t* ops;
    uint8_t* nops;
    int cap;
    ASM_DEV bool on() const { return ops != nullptr; }
    ASM_DEV void emit(long pair, int& cnt, int count, int op) const {
        if (cnt < cap) ops[pair * cap + cnt] = (uint16_t)((count << 3) | op);
        cnt++;
    }
    // lane switch, then the run of (mis)matches — the two appends of _update_CIGAR
    ASM_DEV void step(long pair, int& cnt, int from_lane, int to_lane, int run) const {
        if (to_lane < from_lane)
            emit(pair, cnt, from_lane - to_lane, 1);
        else if (to_lane > from_lane)
            emit(pair, cnt, to_lane - from_lane, 2);
        if (run > 0) emit(pair, cnt, run, 0);
    }
    ASM_DEV void finish(long pair, int cnt) const { nops[pair] = (uint8_t)(cnt > 255 ? 255 : cnt); }
};
