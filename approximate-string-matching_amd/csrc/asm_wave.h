// Wide-band kernels, ONE WAVEFRONT PER READ PAIR (k <= 31: the 2k+1 <= 63 band lanes map onto the 64 lanes of a
// wave; at the benchmark's k = 30 that is 61 of 64 lanes busy).  Neighbouring band lanes talk through DPP wave
// shifts, the wave agrees on "best lane" through DPP max-reductions on order-preserving integer keys, and
// wave-uniform decisions (termination, the chosen lane) live in scalar registers via __ballot / v_readlane.
// No LDS, no barriers — the workgroup-per-pair kernels of asm_wide.h remain as the fallback.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "asm_bits.h"
#include "asm_kernels.h"

#define ASM_WAVE_MAX_K 31

// DPP controls (gfx9 family): lane i reads lane i-1 / i+1 of the whole wave; lanes with no source keep `old`.
#define DPP_WAVE_SHL1 0x130 /* lane i <- lane i+1 */
#define DPP_WAVE_SHR1 0x138 /* lane i <- lane i-1 */
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_BCAST15 0x142
#define DPP_BCAST31 0x143

ASM_DEV int wave_from_below(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, DPP_WAVE_SHR1, 0xf, 0xf, false); }
ASM_DEV int wave_from_above(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, DPP_WAVE_SHL1, 0xf, 0xf, false); }

// max over the 64 lanes, returned wave-uniform (scan within rows of 16, then row broadcasts; identity 0)
ASM_DEV unsigned wave_max_u32(unsigned v) {
#define STEP(ctrl, rmask)                                                                              \
    {                                                                                                  \
        const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, false);  \
        v = o > v ? o : v;                                                                             \
    }
    STEP(DPP_ROW_SHR(1), 0xf)
    STEP(DPP_ROW_SHR(2), 0xf)
    STEP(DPP_ROW_SHR(4), 0xf)
    STEP(DPP_ROW_SHR(8), 0xf)
    STEP(DPP_BCAST15, 0xa)
    STEP(DPP_BCAST31, 0xc)
#undef STEP
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

ASM_DEV int lane_read(int v, int lane /* wave-uniform */) {
    return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(lane));
}

// --------------------------------------------------------------------------------------------------------
// Work distribution of the wave-per-pair kernels.  Their grid is sized to what is resident when the kernel runs alone; in
// asm_run_benchmark_async they run beside another aligner's kernel and only part of the grid is resident at first, so a
// static "wave w takes pairs w, w + W, ..." split leaves the late workgroups a full share to do after everybody else has
// finished.  Each wave therefore draws chunks of PAIR_QUEUE_CHUNK consecutive pairs from a counter in device memory (zeroed
// by the launcher on the same stream): one atomic per chunk, far below the rate a single address sustains.
// --------------------------------------------------------------------------------------------------------
#ifndef PAIR_QUEUE_CHUNK
#define PAIR_QUEUE_CHUNK 16
#endif
struct PairQueue {
    long chunk_end;
    ASM_DEV long grab(unsigned long long* queue) {
        unsigned long long b = 0;
        if ((threadIdx.x & 63) == 0) b = atomicAdd(queue, (unsigned long long)PAIR_QUEUE_CHUNK);
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
        return (long)(((unsigned long long)hi << 32) | lo);
    }
    ASM_DEV long first(unsigned long long* queue, long n) {
        const long b = grab(queue);
        chunk_end = b + PAIR_QUEUE_CHUNK;
        return b < n ? b : n;
    }
    ASM_DEV long next(unsigned long long* queue, long i, long n) {
        if (i + 1 < chunk_end) return i + 1;
        return first(queue, n);
    }
};

// --------------------------------------------------------------------------------------------------------
// count_ID_length (LV_BAG.cpp:9-23) on bit planes held in LDS, shared by the wide-band kernels below: a lane mask is not
// materialised — a 32-position window of the read is compared with the window of the reference the lane pairs it with, both
// cut out of the planes with v_alignbit_b32, until the first difference or where either string ends.
// (The thread-per-pair forms that first used these helpers, leap_band_kernel / leap_band_general_kernel, and the wave-per-pair
// leap_wave_kernel lost to the four-threads-per-pair kernel everywhere and were removed in round 4; DESIGN.md keeps their numbers.)
// --------------------------------------------------------------------------------------------------------
#define LEAP_BAND_THREADS 128

template <int PD, int TS = LEAP_BAND_THREADS> /* PD plane dwords per string in LDS, one zero dword of padding included; TS = column stride */
ASM_DEV uint32_t leap_band_window(const uint32_t* plane, int pos) {
    const int q = pos >> 5;
    return __builtin_amdgcn_alignbit(plane[(q + 1) * TS], plane[q * TS], (uint32_t)(pos & 31));
}

// first position p >= from at which lane d sees a mismatch (or either string has run out), as leap_lane_mask defines it
template <int PD, int TS = LEAP_BAND_THREADS, int PS = PD * TS> /* PS: dwords between planes */
ASM_DEV int leap_band_extend(const uint32_t* pl, int d, int from, int m, int nn) {
    const int s = d < 0 ? -d : d;
    int lim = d < 0 ? m + s : nn + s; /* d < 0: A[p-s] against B[p]; d >= 0: A[p] against B[p-s] */
    const int other = d < 0 ? nn : m;
    lim = lim < other ? lim : other;
    if (from < s || from >= lim) return from;
    int apos = d < 0 ? from - s : from, bpos = d < 0 ? from : from - s, p = from;
    for (;;) {
        const uint32_t diff = (leap_band_window<PD, TS>(pl, apos) ^ leap_band_window<PD, TS>(pl + 2 * PS, bpos)) |
                              (leap_band_window<PD, TS>(pl + PS, apos) ^ leap_band_window<PD, TS>(pl + 3 * PS, bpos));
        if (diff) {
            p += __builtin_ctz(diff);
            break;
        }
        p += 32, apos += 32, bpos += 32;
        if (p >= lim) break;
    }
    return p < lim ? p : lim;
}

// --------------------------------------------------------------------------------------------------------
// LEAP, wide band, FOUR THREADS PER PAIR (a quad), sixteen pairs per wave.  Within one generation the lanes are independent —
// `end`, I and D of generation e read generations e-o, e-x and e-ext only — so the live lanes of a pair are dealt round-robin to
// the four threads of its quad (lane l goes to thread l mod 4: the live range is centred, every thread gets a quarter of it).
// Against the thread-per-pair kernels above this divides the LDS a wave needs by four (rings and planes of 16 pairs instead
// of 64), which is what bounds their occupancy: at k = 30 the (2,3,1) rings let three waves share a CU there and seventeen
// here; a wave also waits for the slowest of 16 pairs rather than of 64.  Generations are kept in rings of thread-shared LDS
// columns [slot][lane row][pair] (bytes or shorts, stored +2 so that 0 = "never reached"); with unit penalties two `end`
// slots are enough (I and D are redundant at o == ext, see leap_unit_pair).  A wave's LDS operations complete in program
// order, so the only synchronisation between generations is a wave-scope fence that keeps the compiler from moving a read
// of another thread's value above the writes of the generation before.
// Same recurrences and results as leap_unit_kernel / leap_general_kernel (LV::run, LV_BAG.cpp:127-245).
// Measured and dropped: a persistent form with decoupled quads (every quad at its own generation, one lane per thread per
// trip of the wave's loop, a quad whose pair has passed takes the next pair from a device-memory queue) — it removes the wait
// for the slowest of the sixteen pairs (work grows with final_ED^2: ~1.6x the mean at C3), was bit-identical, and ran 1.31 ms
// against 0.91 per 10^6 C3 pairs (0.52 against 0.19 at C2): every refill stalls the whole wave for two dependent global
// round trips (lengths, then planes), 122 times per wave at 10^6 pairs.  It would need the next pair prefetched into registers.
// --------------------------------------------------------------------------------------------------------
#define LEAP_QUAD_THREADS 64
#define LEAP_QUAD_PAIRS 16
/* planes in LDS pair-major: the 4 x PD dwords of a pair are contiguous (a window is two adjacent dwords: one address add, no
 * multiply) and pairs are an odd number of dwords apart, so the sixteen quads start in different banks */
#define LEAP_QUAD_PAIR_DWORDS(PD) (4 * (PD) + 1)

ASM_DEV void leap_quad_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
ASM_DEV int quad_or(int v) {
    v |= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false); /* quad_perm [1,0,3,2] */
    v |= __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false); /* quad_perm [2,3,0,1] */
    return v;
}

// WAVES > 1 is the work-sorted form (as leap_unit_hint_kernel): with a per-pair work estimate at hand — the NW penalties, or
// the Greedy penalties where `_run_benchmark` has no NW —, the workgroup counting-sorts its 16 x WAVES pairs by it and each of
// its waves takes one slice of the order, so that the sixteen pairs a wave waits for need about the same number of generations
// (the work of a pair grows with final_ED^2).  A workgroup keeps its LDS and wave slots until its slowest wave is done, so
// sorting more pairs per workgroup does not pay (16 waves: 0.83 ms against 0.78 with 4 per 10^6 C3 pairs); for the long-running
// cases the launcher sorts the whole bucket instead (one 6-bit radix pass) and hands single-wave workgroups a permutation.  The hint only changes the schedule, never a result.  Each wave has its own LDS block.
template <int W32, typename EnT, bool UNIT, int WAVES>
__global__ __launch_bounds__(LEAP_QUAD_THREADS * WAVES) void leap_quad_kernel(const uint4* __restrict__ planes,
                                                                      const uint32_t* __restrict__ lens, long n, int w4, int k,
                                                                      int x, int o, int ext, int gm, int gi, OutMap out,
                                                                      const int32_t* __restrict__ hint,
                                                                      const uint32_t* __restrict__ perm) {
    constexpr int P = LEAP_QUAD_PAIRS, PD = W32 + 1, PW = LEAP_QUAD_PAIR_DWORDS(PD);
    extern __shared__ uint32_t s_band_all[];
    const int t = threadIdx.x & 63, pr = t >> 2, q = t & 3, wv = threadIdx.x >> 6;
    const int rows = 2 * k + 3; /* lane l at row l+1, guard rows 0 and 2k+2 */
    const int slot = rows * P;  /* elements per ring slot */
    const int ring_words = (int)(((size_t)(gm + 2 * gi) * slot * sizeof(EnT) + 3) / 4);
    uint32_t* const s_band = s_band_all + wv * (P * PW + ring_words);          /* this wave's block */
    uint32_t* const pl = s_band + pr * PW;                                     /* [P][4][PD] (+1) */
    EnT* const r_en = reinterpret_cast<EnT*>(s_band + P * PW) + pr;           /* [gm][rows][P] */
    EnT* const r_ip = r_en + gm * slot;                                        /* [gi][rows][P] (general penalties only) */
    EnT* const r_dp = r_ip + gi * slot;
    {
        uint32_t* const base = s_band + P * PW;
        for (int w = t; w < ring_words; w += LEAP_QUAD_THREADS) base[w] = 0u;
    }
    long i = ((long)blockIdx.x * WAVES + wv) * P + pr;
    bool live = i < n;
    if (WAVES == 1 && perm != nullptr && live) i = (long)perm[i]; /* globally work-sorted: slot -> pair (leap_sort_keys_kernel + radix sort) */
    if constexpr (WAVES > 1) {
        constexpr int BP = P * WAVES; /* pairs per workgroup */
        __shared__ uint16_t s_sorted[BP];
        __shared__ int s_bin[64];
        const int tt = threadIdx.x;
        const long base = (long)blockIdx.x * BP;
        const int cnt = (n - base) < BP ? (int)(n - base) : BP;
        if (tt < 64) s_bin[tt] = 0;
        __syncthreads();
        int key = -1;
        if (tt < cnt) {
            int hv = hint[out.index(base + tt)];
            if (!UNIT) hv /= (ext < x ? ext : x); /* penalties, not edits: bring them into the 64 bins */
            key = hv < 0 ? 63 : (hv > 63 ? 63 : hv);
            atomicAdd(&s_bin[key], 1);
        }
        __syncthreads();
        if (tt == 0) { /* exclusive scan of 64 bins */
            int run = 0;
            for (int b = 0; b < 64; b++) {
                const int c = s_bin[b];
                s_bin[b] = run;
                run += c;
            }
        }
        __syncthreads();
        if (key >= 0) s_sorted[atomicAdd(&s_bin[key], 1)] = (uint16_t)tt;
        __syncthreads();
        const int rank = wv * P + pr;
        live = rank < cnt;
        i = live ? base + s_sorted[rank] : 0;
    }
    int m = 0, nn = 0;
    if (live) { /* thread q of the quad stages plane q: read plane 0/1, reference plane 0/1 */
        const uint32_t ln = lens[i];
        m = (int)(ln & 0xffffu), nn = (int)(ln >> 16);
#pragma unroll
        for (int g = 0; g < (W32 + 3) / 4; g++) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (g < w4) v = planes[((long)q * w4 + g) * n + i];
            uint32_t* dst = pl + q * PD + 4 * g;
            dst[0] = v.x;
            if (4 * g + 1 < W32) dst[1] = v.y;
            if (4 * g + 2 < W32) dst[2] = v.z;
            if (4 * g + 3 < W32) dst[3] = v.w;
        }
        pl[q * PD + W32] = 0u;
    }
    const int len = m > nn ? m : nn; /* benchmark_utils.h:162 */
    leap_quad_fence();
    int result = live ? -1 : 0;
    if (live) { /* e = 0: main diagonal only (LV_BAG.cpp:102-104,131-147); the four threads compute the same value */
        int e0 = leap_band_extend<PD, 1, PD>(pl, 0, 0, m, nn);
        e0 = e0 > len ? len : e0;
        if (q == 0) r_en[(k + 1) * P] = (EnT)(e0 + 2);
        if (e0 == len) result = 0;
    }
    leap_quad_fence();
    for (int e = 1; e <= ASM_LEAP_AF_THRESHOLD; e++) {
        if (__ballot(result < 0) == 0ull) break;
        int pass = 0;
        if (result < 0) {
            if constexpr (UNIT) {
                const int lo = k - e > 0 ? k - e : 0, hi = k + e < 2 * k ? k + e : 2 * k;
                const EnT* const en_r = r_en + ((e - 1) & 1) * slot;
                EnT* const en_w = r_en + (e & 1) * slot;
                for (int l = lo + q; l <= hi; l += 4) {
                    const int up_old = (int)en_r[l * P] - 2, cur_old = (int)en_r[(l + 1) * P] - 2, dn_old = (int)en_r[(l + 2) * P] - 2;
                    const int d = l - k;
                    const int top = d >= 0 ? 1 : 0, bot = d <= 0 ? 1 : 0;
                    int st = cur_old + 1;                         /* LV_BAG.cpp:186-187 */
                    st = up_old + top > st ? up_old + top : st;   /* I_pos (redundant table at o = ext, see leap_unit_pair) */
                    st = dn_old + bot > st ? dn_old + bot : st;   /* D_pos */
                    int enew = -2;
                    if (st >= 0) {
                        const int from = st > len ? len : st;
                        int r = leap_band_extend<PD, 1, PD>(pl, d, from, m, nn); /* count_ID_length, :9-23 */
                        r = r > len ? len : r;
                        enew = st > len ? st : r;
                        const int diff = d < 0 ? -d : d;
                        if (enew == len && e + diff <= ASM_LEAP_AF_THRESHOLD) pass = 1; /* :220-238 */
                    }
                    en_w[(l + 1) * P] = (EnT)(enew + 2);
                }
            } else {
                int dmax = e < o ? 0 : (e - o) / ext + 1;
                dmax = dmax > k ? k : dmax;
                const EnT* const en_o = r_en + ((e - o) & (gm - 1)) * slot;
                const EnT* const en_x = r_en + ((e - x) & (gm - 1)) * slot;
                const EnT* const ip_e = r_ip + ((e - ext) & (gi - 1)) * slot;
                const EnT* const dp_e = r_dp + ((e - ext) & (gi - 1)) * slot;
                EnT* const en_w = r_en + (e & (gm - 1)) * slot;
                EnT* const ip_w = r_ip + (e & (gi - 1)) * slot;
                EnT* const dp_w = r_dp + (e & (gi - 1)) * slot;
                // The reference guards these reads with e >= o, e >= x, e >= ext (LV_BAG.cpp:166-187); here the slot of a generation
                // before 0 is the slot of one not written yet (ring depth > penalty), still zero from the initial fill
                for (int l = k - dmax + q; l <= k + dmax; l += 4) {
                    const int d = l - k;
                    const int top = d >= 0 ? 1 : 0, bot = d <= 0 ? 1 : 0;
                    const int e_up = (int)en_o[l * P] - 2;       /* lane l-1 sits at row l */
                    const int i_up = (int)ip_e[l * P] - 2;
                    const int e_dn = (int)en_o[(l + 2) * P] - 2; /* lane l+1 */
                    const int d_dn = (int)dp_e[(l + 2) * P] - 2;
                    const int own = (int)en_x[(l + 1) * P] - 2;
                    int inew = -2, dnew = -2;
                    if (e_up >= 0 && e_up > i_up)
                        inew = e_up + top; /* LV_BAG.cpp:166-167 */
                    else if (i_up >= 0)
                        inew = i_up + top; /* :172-176 */
                    if (e_dn >= 0 && e_dn > d_dn)
                        dnew = e_dn + bot; /* :179-180 */
                    else if (d_dn >= 0)
                        dnew = d_dn + bot; /* :181-182 */
                    int st = own >= 0 ? own + 1 : -2; /* :186-187 */
                    st = inew > st ? inew : st;
                    st = dnew > st ? dnew : st;
                    int enew = -2;
                    if (st >= 0) {
                        const int from = st > len ? len : st;
                        int r = leap_band_extend<PD, 1, PD>(pl, d, from, m, nn); /* count_ID_length, :9-23 */
                        r = r > len ? len : r;
                        enew = st > len ? st : r;
                        if (enew == len) { /* :220-238 */
                            const int diff = d < 0 ? -d : d;
                            const int conv = e + (diff ? o + (diff - 1) * ext : 0);
                            if (conv <= ASM_LEAP_AF_THRESHOLD) pass = 1;
                        }
                    }
                    en_w[(l + 1) * P] = (EnT)(enew + 2), ip_w[(l + 1) * P] = (EnT)(inew + 2), dp_w[(l + 1) * P] = (EnT)(dnew + 2);
                }
            }
        }
        pass = quad_or(pass);
        if (pass && result < 0) result = e; /* final_ED (LV_BAG.cpp:228,356-358) */
        leap_quad_fence();
    }
    if (live && q == 0) out.put(i, result);
}

static inline size_t leap_quad_lds(int w32, int k, int gm, int gi, size_t en_bytes) {
    return (size_t)LEAP_QUAD_PAIRS * LEAP_QUAD_PAIR_DWORDS(w32 + 1) * sizeof(uint32_t) +
           (((size_t)(gm + 2 * gi) * (2 * k + 3) * LEAP_QUAD_PAIRS * en_bytes + 3) & ~(size_t)3);
}

// keys of the global work sort: the hint (penalties) scaled into 64 bins, and the identity permutation to carry along
__global__ __launch_bounds__(256) void leap_sort_keys_kernel(const int32_t* __restrict__ hint, OutMap out, long n, int div,
                                                             uint8_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int hv = hint[out.index(j)] / div;
    keys[j] = (uint8_t)(hv < 0 ? 63 : (hv > 63 ? 63 : hv));
    idx[j] = (uint32_t)j;
}

template <int W32, typename EnT>
static inline hipError_t launch_leap_quad(hipStream_t stream, const uint4* planes, const uint32_t* lens, int64_t n, int w4, int k,
                                          bool unit, int x, int o, int e, int gm, int gi, OutMap out, const int32_t* hint,
                                          const uint32_t* perm) {
    const size_t lds1 = unit ? leap_quad_lds(W32, k, 2, 0, sizeof(EnT)) : leap_quad_lds(W32, k, gm, gi, sizeof(EnT));
    if (hint && !perm && 4 * lds1 + 512 <= 64 * 1024) { /* work-sorted inside workgroups of four waves (64 pairs) */
        const dim3 grid((unsigned)((n + 4 * LEAP_QUAD_PAIRS - 1) / (4 * LEAP_QUAD_PAIRS))), block(4 * LEAP_QUAD_THREADS);
        if (unit)
            hipLaunchKernelGGL((leap_quad_kernel<W32, EnT, true, 4>), grid, block, 4 * lds1, stream, planes, lens, (long)n, w4, k, 1, 1, 1,
                               2, 0, out, hint, nullptr);
        else
            hipLaunchKernelGGL((leap_quad_kernel<W32, EnT, false, 4>), grid, block, 4 * lds1, stream, planes, lens, (long)n, w4, k, x, o,
                               e, gm, gi, out, hint, nullptr);
        return hipGetLastError();
    }
    const dim3 grid((unsigned)((n + LEAP_QUAD_PAIRS - 1) / LEAP_QUAD_PAIRS)), block(LEAP_QUAD_THREADS);
    if (unit) {
        hipLaunchKernelGGL((leap_quad_kernel<W32, EnT, true, 1>), grid, block, lds1, stream, planes, lens, (long)n, w4, k, 1, 1, 1, 2, 0,
                           out, nullptr, perm);
    } else {
        hipLaunchKernelGGL((leap_quad_kernel<W32, EnT, false, 1>), grid, block, lds1, stream, planes, lens, (long)n, w4, k, x, o, e, gm,
                           gi, out, nullptr, perm);
    }
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------------
// Affine NW, banded wavefront, EIGHT THREADS PER PAIR — the second pass of launch_nw_wfa.  The pairs the thread-per-pair pass
// (nw_wfa_kernel, |d| <= 7, mismatch vectors in registers) could not settle are a percent or two of a batch: too few to fill
// the chip with one thread each (every thread would walk ~800 lane-steps alone on its SIMD), so here the lanes of a pair are
// dealt round-robin to eight threads, as leap_quad_kernel does, with the rings [slot][lane row][pair] and the planes in LDS
// and the band half-width K a run-time value.  Recurrence, exactness bound and todo list as nw_wfa_kernel.
// --------------------------------------------------------------------------------------------------------
#define NW_OCT_THREADS 64
#define NW_OCT_Q 8
#define NW_OCT_PAIRS (NW_OCT_THREADS / NW_OCT_Q)

// first i >= st on diagonal d (reference index i + d) at which the strings differ or either one ends (or i + d < 0)
template <int PD, int TS, int PS>
ASM_DEV int nw_diag_extend(const uint32_t* pl, int d, int st, int m, int nn) {
    const int lim = m < nn - d ? m : nn - d;
    if (st >= lim || st + d < 0) return st;
    int apos = st, bpos = st + d, p = st;
    for (;;) {
        const uint32_t diff = (leap_band_window<PD, TS>(pl, apos) ^ leap_band_window<PD, TS>(pl + 2 * PS, bpos)) |
                              (leap_band_window<PD, TS>(pl + PS, apos) ^ leap_band_window<PD, TS>(pl + 3 * PS, bpos));
        if (diff) {
            p += __builtin_ctz(diff);
            break;
        }
        p += 32, apos += 32, bpos += 32;
        if (p >= lim) break;
    }
    return p < lim ? p : lim;
}

template <int W32, typename EnT>
__global__ __launch_bounds__(NW_OCT_THREADS) void nw_oct_kernel(const uint4* __restrict__ planes, const uint32_t* __restrict__ lens,
                                                                long n, int w4, int K, int x, int o, int ext, int gm, int gi,
                                                                OutMap out, const uint32_t* __restrict__ in_list,
                                                                const uint32_t* __restrict__ in_count, uint32_t* __restrict__ todo,
                                                                uint32_t* __restrict__ todo_count) {
    constexpr int P = NW_OCT_PAIRS, Q = NW_OCT_Q, PD = W32 + 1, PW = LEAP_QUAD_PAIR_DWORDS(PD);
    extern __shared__ uint32_t s_band[];
    const int t = threadIdx.x, pr = t / Q, q = t % Q;
    const int rows = 2 * K + 3; /* lane l at row l+1, guard rows 0 and 2K+2 stay "never reached" */
    const int slot = rows * P;
    uint32_t* const pl = s_band + pr * PW;                                /* [P][4][PD] (+1), as leap_quad_kernel */
    EnT* const r_m = reinterpret_cast<EnT*>(s_band + P * PW) + pr;       /* [gm][rows][P] */
    EnT* const r_i = r_m + gm * slot;                                     /* [gi][rows][P] */
    EnT* const r_d = r_i + gi * slot;
    const long count = (long)*in_count;
    /* a fixed grid walks the list (its length is only known on the device) */
    for (long first = (long)blockIdx.x * P; first < count; first += (long)gridDim.x * P) {
    {
        const int words = (int)(((size_t)(gm + 2 * gi) * slot * sizeof(EnT) + 3) / 4);
        uint32_t* const base = s_band + P * PW;
        for (int w = t; w < words; w += NW_OCT_THREADS) base[w] = 0u;
    }
    const long slot_id = first + pr;
    const bool have = slot_id < count;
    const long i = have ? (long)in_list[slot_id] : 0;
    int m = 0, nn = 0;
    if (have) {
        const uint32_t ln = lens[i];
        m = (int)(ln & 0xffffu), nn = (int)(ln >> 16);
        if (q < 4) { /* threads 0..3 of the group stage one plane each */
#pragma unroll
            for (int g = 0; g < (W32 + 3) / 4; g++) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (g < w4) v = planes[((long)q * w4 + g) * n + i];
                uint32_t* dst = pl + q * PD + 4 * g;
                dst[0] = v.x;
                if (4 * g + 1 < W32) dst[1] = v.y;
                if (4 * g + 2 < W32) dst[2] = v.z;
                if (4 * g + 3 < W32) dst[3] = v.w;
            }
            pl[q * PD + W32] = 0u;
        }
    }
    const int df = nn - m, adf = df < 0 ? -df : df;
    leap_quad_fence();
    int result = have ? -1 : 0;
    bool unresolved = have && adf > K;
    if (unresolved) result = 0;
    const int bound = 2 * o + (2 * K - adf) * ext;
    if (result < 0) {
        const int e0 = nw_diag_extend<PD, 1, PD>(pl, 0, 0, m, nn);
        if (q == 0) r_m[(K + 1) * P] = (EnT)(e0 + 2);
        if (df == 0 && e0 >= m) result = 0;
    }
    leap_quad_fence();
    int sl_w = 0, sl_o = gm - o % gm, sl_x = gm - x % gm, sj_w = 0, sj_e = gi - ext % gi;
    sl_o = sl_o == gm ? 0 : sl_o, sl_x = sl_x == gm ? 0 : sl_x, sj_e = sj_e == gi ? 0 : sj_e;
    for (int s = 1;; s++) {
        if (result < 0 && s > bound) unresolved = true, result = 0;
        if (__ballot(result < 0) == 0ull) break;
        sl_w = sl_w + 1 == gm ? 0 : sl_w + 1, sl_o = sl_o + 1 == gm ? 0 : sl_o + 1, sl_x = sl_x + 1 == gm ? 0 : sl_x + 1;
        sj_w = sj_w + 1 == gi ? 0 : sj_w + 1, sj_e = sj_e + 1 == gi ? 0 : sj_e + 1;
        int done = 0;
        if (result < 0) {
            const EnT* const m_o = r_m + sl_o * slot;
            const EnT* const m_x = r_m + sl_x * slot;
            const EnT* const i_e = r_i + sj_e * slot;
            const EnT* const d_e = r_d + sj_e * slot;
            EnT* const m_w = r_m + sl_w * slot;
            EnT* const i_w = r_i + sj_w * slot;
            EnT* const d_w = r_d + sj_w * slot;
            int dmax = s < o ? 0 : (s - o) / ext + 1;
            dmax = dmax > K ? K : dmax;
            for (int l = K - dmax + q; l <= K + dmax; l += Q) {
                const int d = l - K;
                const int m_lo = (int)m_o[l * P] - 2, i_lo = (int)i_e[l * P] - 2;             /* lane l-1 sits at row l */
                const int m_hi = (int)m_o[(l + 2) * P] - 2, d_hi = (int)d_e[(l + 2) * P] - 2; /* lane l+1 */
                const int own = (int)m_x[(l + 1) * P] - 2;
                int inew = m_lo > i_lo ? m_lo : i_lo;      /* reference character consumed: i stays, j = i + d */
                inew = (inew >= 0 && inew + d <= nn) ? inew : -2;
                int dnew = m_hi > d_hi ? m_hi : d_hi;      /* read character consumed */
                dnew = (dnew >= 0 && dnew + 1 <= m) ? dnew + 1 : -2;
                int st = (own >= 0 && own + 1 <= m && own + 1 + d <= nn) ? own + 1 : -2;
                st = inew > st ? inew : st;
                st = dnew > st ? dnew : st;
                int mnew = -2;
                if (st >= 0) mnew = nw_diag_extend<PD, 1, PD>(pl, d, st, m, nn);
                if (d == df && mnew >= m) done = 1;
                m_w[(l + 1) * P] = (EnT)(mnew + 2), i_w[(l + 1) * P] = (EnT)(inew + 2), d_w[(l + 1) * P] = (EnT)(dnew + 2);
            }
        }
#pragma unroll
        for (int off = 1; off < Q; off <<= 1) done |= __shfl_xor(done, off);
        if (done && result < 0) result = s;
        leap_quad_fence();
    }
    if (have && q == 0) {
        if (unresolved)
            todo[atomicAdd(todo_count, 1u)] = (uint32_t)i;
        else
            out.put(i, result);
    }
    leap_quad_fence(); /* the next round refills rings and planes */
    }
}

static inline size_t nw_oct_lds(int w32, int K, int gm, int gi, size_t en_bytes) {
    return (size_t)NW_OCT_PAIRS * LEAP_QUAD_PAIR_DWORDS(w32 + 1) * sizeof(uint32_t) +
           (((size_t)(gm + 2 * gi) * (2 * K + 3) * NW_OCT_PAIRS * en_bytes + 3) & ~(size_t)3);
}

// --------------------------------------------------------------------------------------------------------
// Greedy, wave per pair.  Same step structure as greedy_persist_kernel<K> (hurdle_matrix.h:285-434,568-597); lane t of the
// wave is band lane t - k.
// Round 4 (C3: 13.5 -> see DESIGN.md): the kernel is VALU-issue bound and a wave executes every instruction whether one band
// lane needs it or sixty-one, so only the instruction COUNT matters.  (i) The lane vectors are built on four 32-bit words: a
// band lane is at most 31 positions off the main diagonal, so the 128-bit shift of _construct_hurdles is four v_alignbit_b32,
// and flip_short_hurdles' two shifts by one are eight more; (ii) the final hop reads the destination lane's vector out of the
// lane that holds it (v_readlane) and counts its hurdles in scalar registers instead of rebuilding the vector with one active
// thread; (iii) _choose_best_highway's per-lane tail count runs only when some lane passes the tests that need no count
// (switch + hurdles within the best lane's cost, start not behind the best lane's); (iv) fwd_col without the multiply.
// --------------------------------------------------------------------------------------------------------
ASM_DEV uint4 w_toward0_small(uint4 v, uint32_t s /* 0..31 */) { /* utils.h:143-153 on four words, shift below 32 */
    return make_uint4(__builtin_amdgcn_alignbit(v.y, v.x, s), __builtin_amdgcn_alignbit(v.z, v.y, s),
                      __builtin_amdgcn_alignbit(v.w, v.z, s), v.w >> s);
}
ASM_DEV V128 v_from_words(uint4 q) { return v_from_uint4(q); }

template <bool UNIT> /* UNIT: x = o = e = 1 at compile time */
__global__ __launch_bounds__(ASM_BLOCK) void greedy_wave_kernel(const uint4* __restrict__ planes,
                                                                const uint32_t* __restrict__ lens, long n, int w4,
                                                                int k, GreedyArgs args, OutMap out, CigarSink cig,
                                                                unsigned long long* __restrict__ queue,
                                                                const uint32_t* __restrict__ list /* or null: pairs 0..n-1 */,
                                                                const uint32_t* __restrict__ list_count) {
    const int t = threadIdx.x & 63;
    const int nl = 2 * k + 1;
    const bool active = t < nl;
    const int lane = t - k;
    const bool neg = lane < 0;
    const uint32_t sh = (uint32_t)(active ? (neg ? -lane : lane) : 0); /* <= 31 */
    const int x = UNIT ? 1 : args.x, o = UNIT ? 1 : args.o, e = UNIT ? 1 : args.e;
    const bool semi = UNIT ? false : args.semi != 0;
    const long n_work = list ? (long)*list_count : n; /* a list of pair slots, or all of them */
    PairQueue pq;
    for (long iq = pq.first(queue, n_work); iq < n_work; iq = pq.next(queue, iq, n_work)) {
        const long i = list ? (long)list[iq] : iq;
        const uint4 a0 = planes[((long)0 * w4) * n + i], a1 = planes[((long)1 * w4) * n + i];
        const uint4 b0 = planes[((long)2 * w4) * n + i], b1 = planes[((long)3 * w4) * n + i];
        const uint32_t ln = lens[i];
        int m = (int)(ln & 0xffffu), nn = (int)(ln >> 16);
        m = m > 128 ? 128 : m; /* hurdle_matrix.h:626-627 */
        nn = nn > 128 ? 128 : nn;
        const int dest_lane = nn - m;
        // ---- _construct_hurdles (hurdle_matrix.h:441-455): lane < 0 compares A[i + |lane|] with B[i], lane >= 0 B[i + lane] with A[i]
        V128 lo_, lf_;
        {
            const uint4 xs0 = w_toward0_small(neg ? a0 : b0, sh), xs1 = w_toward0_small(neg ? a1 : b1, sh);
            const uint4 y0 = neg ? b0 : a0, y1 = neg ? b1 : a1;
            const uint4 mw = make_uint4((xs0.x ^ y0.x) | (xs1.x ^ y1.x), (xs0.y ^ y0.y) | (xs1.y ^ y1.y), (xs0.z ^ y0.z) | (xs1.z ^ y1.z),
                                        (xs0.w ^ y0.w) | (xs1.w ^ y1.w));
            /* flip_short_hurdles(1) (utils.h:200-216): a hurdle survives only next to another one */
            const uint4 dn = w_toward0_small(mw, 1u);
            const uint4 up = make_uint4(mw.x << 1, __builtin_amdgcn_alignbit(mw.y, mw.x, 31u), __builtin_amdgcn_alignbit(mw.z, mw.y, 31u),
                                        __builtin_amdgcn_alignbit(mw.w, mw.z, 31u));
            lo_ = v_from_words(mw);
            lf_ = v_from_words(make_uint4(mw.x & (dn.x | up.x), mw.y & (dn.y | up.y), mw.z & (dn.z | up.z), mw.w & (dn.w | up.w)));
        }
        const int dst = lane_destination(m, nn, lane);
        const V128 nlf_ = v_not(lf_);
        const unsigned fb_lf = v_upper_fallback(lf_), fb_nlf = v_upper_fallback(nlf_);
        int sp = -1, len = 0, nsw = 128;
        int cur_lane = 0, cur_col = 0, cost = 0, ncig = 0;
        const long pair = out.index(i);
        for (int guard = 0; guard < 4 * 128; guard++) {
            // ---- _update_highway_list, one band lane per wave lane ----
            bool reach = false;
            int sw = 0, nh = 0;
            const int start_col = cur_col + fwd_col(cur_lane, lane);
            const int dd = lane - cur_lane;
            const int adist = dd < 0 ? -dd : dd;
            if (active) {
                if (sp < start_col) {
                    nsw = adist;
                    int fz, nx;
                    v_highway_from_fb(lf_, nlf_, fb_lf, fb_nlf, start_col, fz, nx);
                    sp = start_col + fz;
                    len = nx;
                    if (start_col + fz + nx > dst) {
                        const int c = dst - (start_col + fz);
                        len = c > 0 ? c : 0;
                        reach = true;
                    }
                }
                sw = (semi && guard == 0) ? 0 : (UNIT ? adist : lane_penalty(cur_lane, lane, o, e));
                nh = v_pop_between(lo_, start_col, sp + len);
            }
            const bool reaching = __ballot(reach) != 0ull;
            const int hc = x * nh;
            double heur = greedy_significance(args, len, nh, nsw);
            int leap = -sw;
            if (reaching) {
                const int fsw = semi ? 0 : lane_penalty(lane, dest_lane, o, e);
                heur = (double)(-sw - hc - fsw - x * (dst - sp - len));
                leap -= fsw;
            }
            // arg-max of (heur, leap), first lane wins exact ties (hurdle_matrix.h:345-351): order-preserving integer
            // keys and three 32-bit wave max-reductions
            heur = heur + 0.0; /* -0.0 -> +0.0 so that equal values have equal keys */
            const unsigned long long hb = (unsigned long long)__double_as_longlong(heur);
            // key = hb < 0 ? ~hb : hb | 2^63, word by word (the lower word is only needed when the upper words tie)
            const unsigned hhi = (unsigned)(hb >> 32);
            const unsigned sgn = (unsigned)((int)hhi >> 31); /* all ones for a negative score */
            unsigned khi = active ? ((hhi ^ sgn) | (~sgn & 0x80000000u)) : 0u;
            const unsigned mhi = wave_max_u32(khi);
            bool cand = active && khi == mhi;
            // Usually one lane alone holds the maximum already in the upper word (scores of different (length, hurdles) differ
            // by far more than 2^-20 relative; ties are lanes of one class): the other two reductions are skipped then
            const unsigned long long cm = __ballot(cand);
            int bt;
            if ((cm & (cm - 1ull)) == 0ull) { /* wave-uniform; cm != 0: some active lane holds the maximum */
                bt = __builtin_ctzll(cm);
            } else {
                const unsigned klo = cand ? ((unsigned)hb ^ sgn) : 0u;
                const unsigned mlo = wave_max_u32(klo);
                cand = cand && klo == mlo;
                const unsigned k3 = cand ? ((((unsigned)(leap + 32768)) << 6) | (unsigned)(63 - t)) : 0u;
                const unsigned m3 = wave_max_u32(k3);
                bt = 63 - (int)(m3 & 63u); /* wave-uniform winner */
            }
            const int best = bt - k;
            const int best_sp = lane_read(sp, bt), best_len = lane_read(len, bt);
            const int best_cost = lane_read(sw + hc, bt);
            if (best_len <= 0) break; /* hurdle_matrix.h:358-361 — uniform */
            // ---- _choose_best_highway ----
            // A lane can only be accepted with inter = sw + nh <= best_cost, and sw = lane_penalty(cur_lane, lane) > 0 for every
            // lane but cur_lane.  With best_cost == 0 that leaves cur_lane, and best_cost == 0 means the best lane has sw == 0,
            // i.e. it IS cur_lane — which the fold skips.  So the whole search is void then (wave-uniform: a scalar branch).
            // (Needs sw > 0 off cur_lane: not with a free gap-open, and not in SEMI_GLOBAL's first step.)
            int ct = bt;
            if (best_cost > 0 || o <= 0 || (semi && guard == 0)) {
                // first the tests that need nothing of the best lane's vector (:376-377 and the intermediate cost of :388,395):
                // when no lane passes them the fold accepts nobody, and the tail counts below are not needed
                const int fb = fwd_col(lane, best);
                const int inter0 = sw + nh;
                const bool may = active && lane != best && !(sp + fb > best_sp) && inter0 <= best_cost;
                if (__ballot(may) != 0ull) {
                    const V128 best_vec = v_make(
                        (u64)(unsigned)lane_read((int)(unsigned)lo_.lo, bt) | ((u64)(unsigned)lane_read((int)(lo_.lo >> 32), bt) << 32),
                        (u64)(unsigned)lane_read((int)(unsigned)lo_.hi, bt) | ((u64)(unsigned)lane_read((int)(lo_.hi >> 32), bt) << 32));
                    const int best_from_sp = v_ones_from(best_vec, best_sp);
                    int inter = 0x3fffffff, total = 0x3fffffff;
                    if (may) {
                        const int endp = sp + len;
                        inter = inter0;
                        const int tail = x * v_pop_between_pre(best_vec, fb + endp, best_sp, best_from_sp);
                        total = inter + lane_penalty(lane, best, o, e) + (tail > 0 ? tail : 0);
                    }
                    // The reference folds lanes in ascending order, accepting a lane only if it is no worse than the last
                    // accepted one in both total and intermediate cost.  Thresholds only ever go down from best_cost, so lanes
                    // above it can be dropped up front; the few that remain are folded in lane order with scalar code.
                    unsigned long long cmask = __ballot(total <= best_cost && inter <= best_cost);
                    int small_total = best_cost, small_inter = best_cost;
                    while (cmask) {
                        const int j = __builtin_ctzll(cmask);
                        cmask &= cmask - 1ull;
                        const int tj = lane_read(total, j), ij = lane_read(inter, j);
                        if (tj <= small_total && ij <= small_inter) small_total = tj, small_inter = ij, ct = j;
                    }
                }
            }
            // ---- _step commit (hurdle_matrix.h:411-433) ----
            cost += lane_read(sw + hc, ct);
            const int new_col = lane_read(sp, ct) + lane_read(len, ct);
            if (cig.on() && t == 0) cig.step(pair, ncig, cur_lane, ct - k, new_col - (cur_col + fwd_col(cur_lane, ct - k)));
            cur_lane = ct - k;
            cur_col = new_col;
            if (cur_col >= lane_destination(m, nn, cur_lane)) break;
        }
        // ---- final hop (hurdle_matrix.h:575-590); cur_lane, cur_col, cost and everything below are wave-uniform ----
        const int dest_col = lane_destination(m, nn, dest_lane);
        if (cur_lane != dest_lane || cur_col < dest_col) {
            V128 dv;
            const int dt = dest_lane + k;
            if (dt >= 0 && dt < nl) { /* the lane that holds the destination lane's vector */
                dv = v_make((u64)(unsigned)lane_read((int)(unsigned)lo_.lo, dt) | ((u64)(unsigned)lane_read((int)(lo_.lo >> 32), dt) << 32),
                            (u64)(unsigned)lane_read((int)(unsigned)lo_.hi, dt) | ((u64)(unsigned)lane_read((int)(lo_.hi >> 32), dt) << 32));
            } else { /* destination lane outside the band: undefined in the reference, built like a band lane */
                dv = greedy_lane_vector(v_from_uint4(a0), v_from_uint4(a1), v_from_uint4(b0), v_from_uint4(b1), dest_lane);
            }
            const int sw_f = semi ? 0 : lane_penalty(cur_lane, dest_lane, o, e);
            const int distance = v_pop_between(dv, cur_col + fwd_col(cur_lane, dest_lane), dest_col);
            const int hcf = x * distance;
            cost += sw_f + (hcf > 0 ? hcf : 0);
            if (cig.on() && t == 0) cig.step(pair, ncig, cur_lane, dest_lane, distance);
        }
        if (t == 0) {
            if (cig.on()) cig.finish(pair, ncig);
            out.put(i, cost);
        }
    }
}

// --------------------------------------------------------------------------------------------------------
// Greedy for 32 <= k <= 50 (65..101 band lanes): TWO wavefronts per pair, thread = band lane.  The lane work and the
// wave-level reductions are those of greedy_wave_kernel; each wave finds its own winner with DPP reductions and the two
// meet through LDS (three barriers per step).  Replaces the serial per-thread scans of greedy_wide_kernel (which stays
// as the ASM_WAVE=0 fallback).
// --------------------------------------------------------------------------------------------------------
#define GREEDY_WAVE2_THREADS 128
__global__ __launch_bounds__(GREEDY_WAVE2_THREADS) void greedy_wave2_kernel(const uint4* __restrict__ planes,
                                                                            const uint32_t* __restrict__ lens, long n, int w4,
                                                                            int k, GreedyArgs args, OutMap out, CigarSink cig) {
    __shared__ unsigned long long s_key[2], s_cmask[2];
    __shared__ int s_kleap[2], s_kbt[2];
    __shared__ u64 s_vec[2][2];
    __shared__ int s_sp[GREEDY_WAVE2_THREADS], s_len[GREEDY_WAVE2_THREADS], s_cost[GREEDY_WAVE2_THREADS];
    __shared__ int s_inter[GREEDY_WAVE2_THREADS], s_total[GREEDY_WAVE2_THREADS];
    const int t = threadIdx.x, w = t >> 6, tl = t & 63;
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
        m = m > 128 ? 128 : m; /* hurdle_matrix.h:626-627 */
        nn = nn > 128 ? 128 : nn;
        const int dest_lane = nn - m;
        const V128 lo_ = greedy_lane_vector(A0, A1, B0, B1, lane);
        const V128 lf_ = v_flip_short_hurdles1(lo_);
        int sp = -1, len = 0, nsw = 128;
        const int dst = lane_destination(m, nn, lane);
        int cur_lane = 0, cur_col = 0, cost = 0, ncig = 0;
        const long pair = out.index(i);
        for (int guard = 0; guard < 4 * 128; guard++) {
            // ---- _update_highway_list, one band lane per thread ----
            int reach = 0, sw = 0, nh = 0;
            if (active) {
                const int start_col = cur_col + fwd_col(cur_lane, lane);
                if (sp < start_col) {
                    const int dd = lane - cur_lane;
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
            double heur = greedy_significance(args, len, nh, nsw);
            int leap = -sw;
            if (reaching) {
                const int fsw = semi ? 0 : lane_penalty(lane, dest_lane, o, e);
                heur = (double)(-sw - hc - fsw - x * (dst - sp - len));
                leap -= fsw;
            }
            // arg-max of (heur, leap) inside the wave, first lane wins exact ties (hurdle_matrix.h:345-351)
            heur = heur + 0.0; /* -0.0 -> +0.0 so that equal values have equal keys */
            const unsigned long long hb = (unsigned long long)__double_as_longlong(heur);
            const unsigned long long key = (hb >> 63) ? ~hb : (hb | 0x8000000000000000ull);
            const unsigned khi = active ? (unsigned)(key >> 32) : 0u;
            const unsigned mhi = wave_max_u32(khi);
            bool cand = active && khi == mhi;
            const unsigned klo = cand ? (unsigned)key : 0u;
            const unsigned mlo = wave_max_u32(klo);
            cand = cand && klo == mlo;
            const unsigned k3 = cand ? ((((unsigned)(leap + 32768)) << 6) | (unsigned)(63 - tl)) : 0u;
            const unsigned m3 = wave_max_u32(k3);
            const int btl = 63 - (int)(m3 & 63u);
            if (tl == btl) { /* this wave's winner (wave 1 always has active lanes: nl >= 65) */
                s_key[w] = active ? key : 0ull;
                s_kleap[w] = leap;
                s_kbt[w] = t;
                s_vec[w][0] = lo_.lo, s_vec[w][1] = lo_.hi;
            }
            s_sp[t] = sp, s_len[t] = len, s_cost[t] = sw + hc;
            __syncthreads();
            const bool take1 = s_key[1] > s_key[0] || (s_key[1] == s_key[0] && s_kleap[1] > s_kleap[0]); /* ties: lower lane */
            const int bt = take1 ? s_kbt[1] : s_kbt[0];
            const int best = bt - k, best_sp = s_sp[bt], best_len = s_len[bt], best_cost = s_cost[bt];
            if (best_len <= 0) break; /* hurdle_matrix.h:358-361 — uniform across the workgroup */
            // ---- _choose_best_highway ----
            const V128 best_vec = v_make(s_vec[take1 ? 1 : 0][0], s_vec[take1 ? 1 : 0][1]);
            const int best_from_sp = v_ones_from(best_vec, best_sp);
            int inter = 0x3fffffff, total = 0x3fffffff;
            if (active && lane != best && !(sp + fwd_col(lane, best) > best_sp)) {
                const int endp = sp + len;
                inter = sw + nh;
                const int tail = x * v_pop_between_pre(best_vec, fwd_col(lane, best) + endp, best_sp, best_from_sp);
                total = inter + lane_penalty(lane, best, o, e) + (tail > 0 ? tail : 0);
            }
            // lanes that can still be accepted (thresholds only go down from best_cost), folded in lane order
            const unsigned long long cm = __ballot(total <= best_cost && inter <= best_cost);
            if (tl == 0) s_cmask[w] = cm;
            s_inter[t] = inter, s_total[t] = total;
            __syncthreads();
            int small_total = best_cost, small_inter = best_cost, ct = bt;
#pragma unroll
            for (int ww = 0; ww < 2; ww++) {
                unsigned long long cmask = s_cmask[ww];
                while (cmask) {
                    const int j = 64 * ww + __builtin_ctzll(cmask);
                    cmask &= cmask - 1ull;
                    const int tj = s_total[j], ij = s_inter[j];
                    if (tj <= small_total && ij <= small_inter) small_total = tj, small_inter = ij, ct = j;
                }
            }
            // ---- _step commit (hurdle_matrix.h:411-433) ----
            cost += s_cost[ct];
            const int new_col = s_sp[ct] + s_len[ct];
            if (cig.on() && t == 0) cig.step(pair, ncig, cur_lane, ct - k, new_col - (cur_col + fwd_col(cur_lane, ct - k)));
            cur_lane = ct - k;
            cur_col = new_col;
            const bool done = cur_col >= lane_destination(m, nn, cur_lane);
            __syncthreads(); /* the LDS arrays are rewritten by the next step */
            if (done) break;
        }
        if (t == 0) {
            // ---- final hop (hurdle_matrix.h:575-590) ----
            const int dest_col = lane_destination(m, nn, dest_lane);
            if (cur_lane != dest_lane || cur_col < dest_col) {
                const V128 dv = greedy_lane_vector(A0, A1, B0, B1, dest_lane);
                const int sw_f = semi ? 0 : lane_penalty(cur_lane, dest_lane, o, e);
                const int distance = v_pop_between(dv, cur_col + fwd_col(cur_lane, dest_lane), dest_col);
                const int hcf = x * distance;
                cost += sw_f + (hcf > 0 ? hcf : 0);
                if (cig.on()) cig.step(pair, ncig, cur_lane, dest_lane, distance);
            }
            if (cig.on()) cig.finish(pair, ncig);
            out.put(i, cost);
        }
        __syncthreads();
    }
}

static inline void launch_greedy_wave2(hipStream_t stream, const uint4* planes, const uint32_t* lens, int64_t n, int w4, int k,
                                       const GreedyArgs& ga, OutMap out, CigarSink cig, int num_cus) {
    int64_t blocks = (int64_t)num_cus * 16; /* two waves each; the kernel strides over the pairs */
    if (blocks > n) blocks = n;
    hipLaunchKernelGGL(greedy_wave2_kernel, dim3((unsigned)blocks), dim3(GREEDY_WAVE2_THREADS), 0, stream, planes, lens, (long)n,
                       w4, k, ga, out, cig);
}

template <typename Kern, typename... Args>
static inline void launch_wave_per_pair(hipStream_t stream, Kern kern, int64_t n, int num_cus, Args... args) {
    int per_cu = 8;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, ASM_BLOCK, 0);
    if (per_cu < 1) per_cu = 1;
    int64_t blocks = (int64_t)per_cu * num_cus;
    const int64_t need = (n + 3) / 4; /* four waves (pairs in flight) per workgroup */
    if (blocks > need) blocks = need;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(ASM_BLOCK), 0, stream, args...);
}
