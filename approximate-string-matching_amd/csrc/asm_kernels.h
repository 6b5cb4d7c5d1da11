// HIP kernels of the batched pair aligner for MI355X (gfx950, wave64).
//
// Data layout in HBM: a batch of n pairs is held as
//   planes : uint4[4][w4][n]   plane p in {A.bit0, A.bit1, B.bit0, B.bit1}, granule g = positions
//                              [128g, 128g+128), pair i at planes[(p*w4+g)*n + i]  (bit q of a granule <->
//                              character 128g+q; codes A=00 C=01 G=10 T=11, anything else 00 — the code of
//                              GASMA/bit_convert.cpp:340-355)
//   lens   : uint32[n]         m | n << 16
// so that thread i of a wave reads 16 contiguous bytes next to thread i+1's: every load instruction of a
// thread-per-pair kernel is one fully coalesced 1 KiB request.
//
// Work decomposition: ONE THREAD PER READ PAIR for the narrow band (k <= 7, the benchmark's k = 3): the
// 2k+1 lanes of the band live in registers, the lane loops are fully unrolled, no cross-lane traffic, and
// every VALU lane does useful work (a wave-per-pair mapping would idle 57 of 64 lanes at k = 3).  Wide bands
// (k up to 50) use the block-per-pair kernels further down (thread = band lane).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "asm_bits.h"
#include "asm_gen.h"

#define ASM_BLOCK 256

// --------------------------------------------------------------------------------------------------------
// Pack: ASCII -> bit planes.  Device counterpart of sse3_convert2bit1 (GASMA/bit_convert.cpp:248-369), minus
// its in-place byte permutation (whose only observable effect, the stale tails of later pairs, is supplied
// through `tails` in sequential mode — see asm_tails.h).
//
// A workgroup owns 256 consecutive pairs.  Their ASCII is one contiguous byte range of the batch, so it is
// staged into LDS with fully coalesced 16 B/lane loads (a thread-per-pair gather straight from HBM would touch
// ~50 cache lines per load instruction); then every thread converts its own string out of LDS, four characters
// per dword with SWAR byte compares (exactly 'C','G','T' set bits; every other byte is code 00).  The staging
// buffer is XOR-swizzled per 128-byte row so that strings whose length is a multiple of 128 B do not all hit
// one LDS bank.  Long batches are staged in rounds of whole pairs that fit the buffer.
// --------------------------------------------------------------------------------------------------------
#define PACK_SB (48 * 1024)
#ifndef PACK_BLOCK
#define PACK_BLOCK 256 /* pairs (threads) per pack workgroup; 128 and 64 measured: C2 59.6 and 60.8 us against 58.7, C5 1.93 and 1.71 ms against 1.28 */
#endif

// Length buckets of a batch (mixed-length batches are grouped by the number of 128-position granules their longer
// string needs, so that every kernel launch works on pairs of one width class).  Bucket b holds the pairs at slots
// [start[b], start[b+1]) of the bucketed order; its planes are a uint4[4][w4[b]][size] block at plane_off[b].
struct PackBuckets {
    int nb;
    int w4[4];
    long start[5];
    long plane_off[4];
};

ASM_DEV uint32_t swar_zero_bytes(uint32_t t) { /* bit 7 of each byte set iff that byte of t is zero (exact) */
    return ~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t | 0x7f7f7f7fu);
}

// One thread converts its string out of the (swizzled) LDS staging buffer and stores its plane granules.
template <int W4>
ASM_DEV void pack_convert(const uint32_t* sb, uint32_t b0, int len, int w4, int s, const uint4* __restrict__ tails,
                          long n, long pair, uint4* __restrict__ bplanes, long bn, long local) {
    const int a0 = (int)(b0 >> 2);
    const uint32_t sh = (b0 & 3u) * 8u;
#pragma unroll
    for (int g = 0; g < W4; g++) {
        if (g < w4) {
            uint32_t q0[4] = {0u, 0u, 0u, 0u}, q1[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const int cbase = g * 128 + w * 32;
                if (cbase < len) {
                    uint32_t d[9];
#pragma unroll
                    for (int q = 0; q < 9; q++) {
                        const int a = a0 + (cbase >> 2) + q;
                        d[q] = sb[a ^ (((a >> 5) & 7) << 2)];
                    }
                    uint32_t g0 = 0u, g1 = 0u; /* flag bytes of the previous (even) dword, gathered */
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const uint32_t ch = __builtin_amdgcn_alignbit(d[q + 1], d[q], sh);
                        // Four characters at once.  Bits 1-2 of 'A','C','T','G' are 0,1,2,3: use them as a v_perm_b32
                        // selector into the table "ACTG" to get the one base each byte could be, and accept the byte
                        // only if it IS that base (exactly 'C','G','T' set plane bits; any other byte, NUL included,
                        // is code 00 as in bit_convert.cpp:340-355).  plane1 (G|T) = bit 2, plane0 (C|T) = bit 1 ^ bit 2.
                        const uint32_t half = ch >> 1;
                        const uint32_t canon = __builtin_amdgcn_perm(0u, 0x47544341u, half & 0x03030303u);
                        const uint32_t ok = swar_zero_bytes(canon ^ ch) >> 7; /* 0x01 in every byte that is a real base */
                        const uint32_t f0 = ((ch ^ half) >> 1) & ok, f1 = (ch >> 2) & ok;
                        /* gather the four byte flags into a nibble with one v_dot4_u32_u8 (weights 1,2,4,8; 16..128 for the odd
                         * dword, accumulated onto the even one): a byte of the plane word per two dwords */
                        if ((q & 1) == 0) {
                            g0 = __builtin_amdgcn_udot4(f0, 0x08040201u, 0u, false);
                            g1 = __builtin_amdgcn_udot4(f1, 0x08040201u, 0u, false);
                        } else {
                            q0[w] |= __builtin_amdgcn_udot4(f0, 0x80402010u, g0, false) << (8 * (q >> 1));
                            q1[w] |= __builtin_amdgcn_udot4(f1, 0x80402010u, g1, false) << (8 * (q >> 1));
                        }
                    }
                    // characters beyond the string's end (the next pair's bytes in the staging buffer) are dropped here,
                    // once per 32 positions, instead of being masked out of every dword
                    const int keep = len - cbase; /* > 0 */
                    const uint32_t km = keep >= 32 ? ~0u : ((1u << keep) - 1u);
                    q0[w] &= km, q1[w] &= km;
                }
            }
            uint4 v0 = make_uint4(q0[0], q0[1], q0[2], q0[3]);
            uint4 v1 = make_uint4(q1[0], q1[1], q1[2], q1[3]);
            if (tails != nullptr && g == 0) {
                const uint4 t0 = tails[(long)(2 * s) * n + pair], t1 = tails[(long)(2 * s + 1) * n + pair];
                v0.x |= t0.x, v0.y |= t0.y, v0.z |= t0.z, v0.w |= t0.w;
                v1.x |= t1.x, v1.y |= t1.y, v1.z |= t1.z, v1.w |= t1.w;
            }
            bplanes[((long)(2 * s) * w4 + g) * bn + local] = v0;
            bplanes[((long)(2 * s + 1) * w4 + g) * bn + local] = v1;
        }
    }
}


// The same conversion for strings that hold nothing but A, C, G, T — what the aligners are fed almost always.  Then bits 1-2 of
// a character ARE its code (A 00, C 01, T 10, G 11 -> plane0 = bit1 ^ bit2, plane1 = bit2) and the per-byte "is it exactly the
// base it could be" test of pack_convert (a SWAR zero-byte test and two masks per dword: half of its instructions) shrinks to
// accumulating canon ^ ch over the string.  Returns that accumulated difference: non-zero = some byte was not a base (or the
// string's last dwords ran into bytes that are not), and the caller converts the string again with pack_convert, whose stores
// overwrite these.  The staging buffer is padded with 'A's behind the last string so that running past a string's end into
// its neighbour — or into the padding — never raises the flag by itself.
template <int W4>
ASM_DEV uint32_t pack_convert_acgt(const uint32_t* sb, uint32_t b0, int len, int w4, int s, const uint4* __restrict__ tails,
                                   long n, long pair, uint4* __restrict__ bplanes, long bn, long local) {
    const int a0 = (int)(b0 >> 2);
    const uint32_t sh = (b0 & 3u) * 8u;
    uint32_t bad = 0u;
#pragma unroll
    for (int g = 0; g < W4; g++) {
        if (g < w4) {
            uint32_t q0[4] = {0u, 0u, 0u, 0u}, q1[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const int cbase = g * 128 + w * 32;
                if (cbase < len) {
                    uint32_t d[9];
#pragma unroll
                    for (int q = 0; q < 9; q++) {
                        const int a = a0 + (cbase >> 2) + q;
                        d[q] = sb[a ^ (((a >> 5) & 7) << 2)];
                    }
                    uint32_t g0 = 0u, g1 = 0u;
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const uint32_t ch = __builtin_amdgcn_alignbit(d[q + 1], d[q], sh);
                        const uint32_t half = ch >> 1, quarter = ch >> 2;
                        const uint32_t canon = __builtin_amdgcn_perm(0u, 0x47544341u, half & 0x03030303u);
                        bad |= canon ^ ch;
                        const uint32_t f0 = (half ^ quarter) & 0x01010101u, f1 = quarter & 0x01010101u;
                        if ((q & 1) == 0) {
                            g0 = __builtin_amdgcn_udot4(f0, 0x08040201u, 0u, false);
                            g1 = __builtin_amdgcn_udot4(f1, 0x08040201u, 0u, false);
                        } else {
                            q0[w] |= __builtin_amdgcn_udot4(f0, 0x80402010u, g0, false) << (8 * (q >> 1));
                            q1[w] |= __builtin_amdgcn_udot4(f1, 0x80402010u, g1, false) << (8 * (q >> 1));
                        }
                    }
                    const int keep = len - cbase; /* > 0 */
                    const uint32_t km = keep >= 32 ? ~0u : ((1u << keep) - 1u);
                    q0[w] &= km, q1[w] &= km;
                }
            }
            uint4 v0 = make_uint4(q0[0], q0[1], q0[2], q0[3]);
            uint4 v1 = make_uint4(q1[0], q1[1], q1[2], q1[3]);
            if (tails != nullptr && g == 0) {
                const uint4 t0 = tails[(long)(2 * s) * n + pair], t1 = tails[(long)(2 * s + 1) * n + pair];
                v0.x |= t0.x, v0.y |= t0.y, v0.z |= t0.z, v0.w |= t0.w;
                v1.x |= t1.x, v1.y |= t1.y, v1.z |= t1.z, v1.w |= t1.w;
            }
            bplanes[((long)(2 * s) * w4 + g) * bn + local] = v0;
            bplanes[((long)(2 * s + 1) * w4 + g) * bn + local] = v1;
        }
    }
    return bad;
}

// 64 bytes of 'A' behind the staged vectors (`vecs` 16-byte vectors were staged; swizzled like everything else).  Written by
// the threads that do not store the last vectors' neighbours: no overlap with the staging stores, same barrier.
ASM_DEV void pack_pad(uint4* s_buf, int vecs, int t) {
    if (t < 4) {
        const int a = 4 * (vecs + t);
        s_buf[(a ^ (((a >> 5) & 7) << 2)) >> 2] = make_uint4(0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u);
    }
}

template <int W4>
ASM_DEV void pack_convert_any(const uint32_t* sb, uint32_t b0, int len, int w4, int s, const uint4* __restrict__ tails, long n,
                              long pair, uint4* __restrict__ bplanes, long bn, long local) {
    if (pack_convert_acgt<W4>(sb, b0, len, w4, s, tails, n, pair, bplanes, bn, local) != 0u)
        pack_convert<W4>(sb, b0, len, w4, s, tails, n, pair, bplanes, bn, local);
}

template <int W4, int NV> /* NV = staging vectors (16 B) per thread the fast path may hold in registers */
__global__ __launch_bounds__(PACK_BLOCK) void pack_kernel(const char* __restrict__ reads,
                                                         const uint32_t* __restrict__ read_off,
                                                         const char* __restrict__ refs,
                                                         const uint32_t* __restrict__ ref_off,
                                                         const uint4* __restrict__ tails, /* [4][n] or null */
                                                         uint4* __restrict__ planes, uint32_t* __restrict__ lens,
                                                         long n, PackBuckets pb,
                                                         const uint32_t* __restrict__ pos /* pair -> slot, or null */,
                                                         uint32_t stage_bytes /* dynamic LDS staging size, <= PACK_SB */) {
    // staging buffer sized by the host from the batch's longest string: short reads leave room for more resident
    // workgroups per CU (5 at 100 bp instead of 3), which is what hides the HBM latency of the staging loads
    extern __shared__ uint4 s_buf[];
    __shared__ uint32_t s_off[2][PACK_BLOCK + 1];
    const int t = threadIdx.x;
    const long p0 = (long)blockIdx.x * PACK_BLOCK;
    const int np = (n - p0) < PACK_BLOCK ? (int)(n - p0) : PACK_BLOCK;
    const uint32_t* sb = reinterpret_cast<const uint32_t*>(s_buf);
    // where this thread's pair lives in the bucketed layout
    const long slot = (t < np) ? (pos ? (long)pos[p0 + t] : p0 + t) : 0;
    int bk = 0;
#pragma unroll
    for (int q = 1; q < 4; q++)
        if (q < pb.nb && slot >= pb.start[q]) bk = q;
    const int w4 = pb.w4[bk];
    const long bn = pb.start[bk + 1] - pb.start[bk], local = slot - pb.start[bk];
    uint4* bplanes = planes + pb.plane_off[bk];

    // both offset tables up front: one global round trip instead of two
    s_off[0][t] = read_off[p0 + (t < np ? t : np)];
    s_off[1][t] = ref_off[p0 + (t < np ? t : np)];
    if (t == 0) s_off[0][PACK_BLOCK] = read_off[p0 + np], s_off[1][PACK_BLOCK] = ref_off[p0 + np];
    __syncthreads();
    const uint32_t oA0 = s_off[0][t], oA1 = s_off[0][t + 1], oB0 = s_off[1][t], oB1 = s_off[1][t + 1];
    const uint32_t baseA = s_off[0][0] & ~15u, baseB = s_off[1][0] & ~15u;
    const uint32_t bytesA = s_off[0][PACK_BLOCK] - baseA, bytesB = s_off[1][PACK_BLOCK] - baseB;
    if (t < np) lens[slot] = (oA1 - oA0) | ((oB1 - oB0) << 16);

    if (bytesA <= stage_bytes && bytesB <= stage_bytes && bytesB <= (uint32_t)(NV * PACK_BLOCK * 16)) {
        // Fast path (every string of the block fits the buffer): the refs' bytes are fetched into registers while the
        // reads are being converted, so their HBM latency hides behind the SWAR work.
        const int nvA = (int)((bytesA + 15u) >> 4), nvB = (int)((bytesB + 15u) >> 4);
        const uint4* srcA = reinterpret_cast<const uint4*>(reads + baseA);
        const uint4* srcB = reinterpret_cast<const uint4*>(refs + baseB);
        for (int v = t; v < nvA; v += PACK_BLOCK) {
            const int a = 4 * v;
            s_buf[(a ^ (((a >> 5) & 7) << 2)) >> 2] = srcA[v];
        }
        pack_pad(s_buf, nvA, t);
        uint4 rb[NV];
#pragma unroll
        for (int q = 0; q < NV; q++) {
            const int v = t + q * PACK_BLOCK;
            rb[q] = v < nvB ? srcB[v] : make_uint4(0u, 0u, 0u, 0u);
        }
        __syncthreads();
        if (t < np) pack_convert_any<W4>(sb, oA0 - baseA, (int)(oA1 - oA0), w4, 0, tails, n, p0 + t, bplanes, bn, local);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NV; q++) {
            const int v = t + q * PACK_BLOCK;
            if (v < nvB) {
                const int a = 4 * v;
                s_buf[(a ^ (((a >> 5) & 7) << 2)) >> 2] = rb[q];
            }
        }
        pack_pad(s_buf, nvB, t);
        __syncthreads();
        if (t < np) pack_convert_any<W4>(sb, oB0 - baseB, (int)(oB1 - oB0), w4, 1, tails, n, p0 + t, bplanes, bn, local);
        return;
    }
    // General path: stage whole pairs in rounds that fit the buffer.
#pragma unroll 1
    for (int s = 0; s < 2; s++) {
        const char* str = s ? refs : reads;
        const uint32_t o0 = s ? oB0 : oA0, o1 = s ? oB1 : oA1;
        const int len = (int)(o1 - o0);
        int ps = 0;
        while (ps < np) { /* uniform across the workgroup */
            __syncthreads();
            const uint32_t base = s_off[s][ps] & ~15u;
            const int fits = (t >= ps && t < np && (o1 - base) <= stage_bytes) ? 1 : 0;
            const int pe = ps + __syncthreads_count(fits); /* offsets are monotone: the fitting pairs are [ps, pe) */
            const uint32_t hi = s_off[s][pe];
            const int nvec = (int)((hi - base + 15u) >> 4);
            const uint4* src = reinterpret_cast<const uint4*>(str + base);
            for (int v = t; v < nvec; v += PACK_BLOCK) {
                const int a = 4 * v;
                s_buf[(a ^ (((a >> 5) & 7) << 2)) >> 2] = src[v];
            }
            pack_pad(s_buf, nvec, t);
            __syncthreads();
            if (t >= ps && t < pe) pack_convert_any<W4>(sb, o0 - base, len, w4, s, tails, n, p0 + t, bplanes, bn, local);
            ps = pe;
        }
    }
}

// width class of every pair (granules of the longer string, minus one) and the class histogram
__global__ __launch_bounds__(ASM_BLOCK) void classify_kernel(const uint32_t* __restrict__ read_off,
                                                             const uint32_t* __restrict__ ref_off, long n,
                                                             uint8_t* __restrict__ cls, uint32_t* __restrict__ idx,
                                                             unsigned int* __restrict__ counts /* [4] */) {
    __shared__ unsigned int s_cnt[4];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint32_t m = read_off[i + 1] - read_off[i], nn = ref_off[i + 1] - ref_off[i];
        uint32_t L = m > nn ? m : nn;
        L = L < 1u ? 1u : L;
        uint32_t c = (L + 127u) / 128u - 1u;
        c = c > 3u ? 3u : c;
        cls[i] = (uint8_t)c;
        idx[i] = (uint32_t)i;
        atomicAdd(&s_cnt[c], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 4 && s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]);
}

__global__ __launch_bounds__(ASM_BLOCK) void invert_order_kernel(const uint32_t* __restrict__ order, long n,
                                                                 uint32_t* __restrict__ pos) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) pos[order[j]] = (uint32_t)j;
}

// --------------------------------------------------------------------------------------------------------
// Greedy hurdle-matrix aligner, GLOBAL mode, band lanes -K..K in registers, one thread per pair.
// Follows hurdle_matrix<int_128bit>: _construct_hurdles (hurdle_matrix.h:441-455), _update_highway_list
// (:285-362), _choose_best_highway (:368-401), _step (:407-434), run (:568-597).  The per-lane cache
// {starting_point,length,num_switches} persists across steps exactly as highway_info does (:26-44).
// --------------------------------------------------------------------------------------------------------
struct GreedyArgs {
    int x, o, e;
    int semi; /* SEMI_GLOBAL (hurdle_matrix.h:313-316,335-338,577-580): no switch cost into the first highway, into the
                 destination lane, or for the final hop */
    double sig_match, sig_mismatch, sig_indel; /* hurdle_matrix.h:536-538, computed on the host with libm */
};

// hurdle_matrix.h:328-330.  With the default probabilities mismatch_sig == indel_sig bit for bit, so lanes that
// trade a hurdle for a lane switch tie up to rounding, and the rounding sequence decides the arg-max.  The
// reference built with its own flags (-O3 -march=native: GCC contracts a*b+c on FMA hardware) evaluates
// fma(indel, nsw, fma(mismatch, nh, match*len)); the same three roundings are issued here explicitly.
ASM_DEV double greedy_significance(const GreedyArgs& a, int len, int nh, int nsw) {
    return __fma_rn(a.sig_indel, (double)nsw, __fma_rn(a.sig_mismatch, (double)nh, __dmul_rn(a.sig_match, (double)len)));
}

ASM_DEV V128 greedy_lane_vector(V128 A0, V128 A1, V128 B0, V128 B1, int lane) {
    const int s = lane < 0 ? -lane : lane;
    V128 m0, m1;
    if (lane < 0) {
        m0 = v_xor(v_toward0(A0, s), B0);
        m1 = v_xor(v_toward0(A1, s), B1);
    } else {
        m0 = v_xor(v_toward0(B0, s), A0);
        m1 = v_xor(v_toward0(B1, s), A1);
    }
    return v_or(m0, m1);
}

// Wave-local work queue for the persistent kernels: every wave owns a contiguous slice [q_next, q_end) of the batch
// (static split over the resident waves: ~n/3000 pairs each, so slices are balanced to a few percent) and its lanes pull
// the next pair with wave-level bit tricks only — no memory atomics (a single global queue head saturates at ~88
// dequeues/us on this chip, far below the refill rate of these kernels).
// Measured dead end (C2, 10^6 pairs, 3072 waves, ~120 us): drawing chunks of pairs from counters in device memory instead —
// one counter, or eight per-XCD-group counters 128 B or 8 KB apart, with a static first chunk and stealing between groups —
// costs ~7 ns of kernel time per draw whatever the layout (chunks of 16 / 64 / 256 pairs: 0.56 / 0.19 / 0.13 ms against 0.12):
// the kernel hands out 8 pairs per nanosecond, and a wave stalls for two memory-side round trips per draw.  The wave-per-pair
// kernels, where a chunk of 16 pairs lasts a wave ~100 us, do use such a queue (PairQueue, asm_wave.h).
struct WaveQueue {
    long next, end;
    ASM_DEV void init(long n) {
        const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
        next = n * wave / nwaves;
        end = n * (wave + 1) / nwaves;
    }
    // lanes with need=true receive consecutive indices; returns -1 when the slice is used up
    ASM_DEV long pull(bool need) {
        const unsigned long long mask = __ballot(need);
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        const long cnt = (long)__popcll(mask);
        const long avail = end - next;
        const long idx = (need && rank < avail) ? next + rank : -1;
        next += cnt < avail ? cnt : avail;
        return idx;
    }
};

// --------------------------------------------------------------------------------------------------------
// Persistent, lane-refilling thread-per-pair Greedy kernel.  Pairs need 1..7 steps each (mean ~2 at err 0.10); with one
// pair per thread a wave runs as long as its slowest pair and the other lanes idle.  Here a wave keeps all 64
// lanes busy: whenever a lane's pair terminates it writes the result and pulls the next pair index from a global
// wave-local queue (WaveQueue above), so every pass of the step body works on 64 live pairs.  The grid is sized to
// what is resident (CUs x occupancy), not to n.
// --------------------------------------------------------------------------------------------------------
// K = 8, 9 (and 10 with unit penalties) fit 256 registers without scratch when asked to; left alone the allocator spreads into
// the AGPR half and the kernel runs at one wave per SIMD instead of two.
#ifndef GREEDY_PERSIST_MIN_WAVES
#define GREEDY_PERSIST_MIN_WAVES(K, UNIT) (((K) == 8 || (K) == 9 || ((K) == 10 && (UNIT))) ? 2 : 1)
#endif
template <int K, bool UNIT> /* UNIT: x = o = e = 1 known at compile time (the benchmark's penalties): the multiplies fold away */
__global__ __launch_bounds__(ASM_BLOCK, GREEDY_PERSIST_MIN_WAVES(K, UNIT)) void greedy_persist_kernel(const uint4* __restrict__ planes,
                                                                   const uint32_t* __restrict__ lens, long n, int w4,
                                                                   GreedyArgs args, OutMap out,
                                                                   CigarSink cig, int refill_min) {
    constexpr int NL = 2 * K + 1;
#ifndef ASM_PERSIST_KEEP_LF
#define ASM_PERSIST_KEEP_LF 5 /* largest K whose flipped lane vectors stay in registers; above it they are rebuilt per look-up */
#endif
    constexpr bool KEEP_LF = K <= ASM_PERSIST_KEEP_LF;
    constexpr int NLF = KEEP_LF ? NL : 1;
    const int x = UNIT ? 1 : args.x, o = UNIT ? 1 : args.o, e = UNIT ? 1 : args.e;
    const bool semi = UNIT ? false : args.semi != 0; /* the UNIT instantiation is GLOBAL only */
    V128 lo_[NL], lf_[NLF];
    int sp[NL], len[NL], nsw[NL], dst[NL], sw[NL], nh[NL];
    V128 dest_vec = v_make(0, 0);
    int m = 0, nn = 0, dest_lane = 0, cur_lane = 0, cur_col = 0, cost = 0, guard = 0, ncig = 0;
    long idx = -1, pair = 0;
    bool active = false, finished = true, exhausted = false;
    WaveQueue wq;
    wq.init(n);
#pragma unroll
    for (int j = 0; j < NL; j++) {
        lo_[j] = v_make(0, 0);
        if (KEEP_LF) lf_[j] = v_make(0, 0);
        sp[j] = -1, len[j] = 0, nsw[j] = 128, dst[j] = 0, sw[j] = nh[j] = 0;
    }
#ifdef GREEDY_DIAG
    unsigned long long dg_refill = 0, dg_step = 0, dg_iters = 0, dg_refills = 0, dg_lanes = 0, dg_t0 = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
#ifdef GREEDY_DIAG
        const unsigned long long dg_a = __builtin_amdgcn_s_memtime();
#endif
        const bool need = finished && !exhausted;
        // Refill lazily: the setup below costs about as much as a step, so wait until `refill_min` lanes are idle
        // (or nothing else is left to do) before paying for it.
        const unsigned long long need_mask = __ballot(need);
        if (need_mask != 0ull && (__popcll(need_mask) >= refill_min || __ballot(active && !finished) == 0ull)) {
            if (need && active) {
                // ---- final hop (hurdle_matrix.h:575-590) ----
                const int dest_col = lane_destination(m, nn, dest_lane);
                if (cur_lane != dest_lane || cur_col < dest_col) {
                    const int sw_f = semi ? 0 : lane_penalty(cur_lane, dest_lane, o, e);
                    const int distance = v_pop_between(dest_vec, cur_col + fwd_col(cur_lane, dest_lane), dest_col);
                    const int hc = x * distance;
                    cost += sw_f + (hc > 0 ? hc : 0);
                    if (cig.on()) cig.step(pair, ncig, cur_lane, dest_lane, distance);
                }
                if (cig.on()) cig.finish(pair, ncig);
                out.put(idx, cost);
            }
            const long got = wq.pull(need);
            if (need) {
                idx = got;
                active = got >= 0;
                exhausted = !active;
            }
            if (need && active) {
                const V128 A0 = v_from_uint4(planes[((long)0 * w4) * n + idx]);
                const V128 A1 = v_from_uint4(planes[((long)1 * w4) * n + idx]);
                const V128 B0 = v_from_uint4(planes[((long)2 * w4) * n + idx]);
                const V128 B1 = v_from_uint4(planes[((long)3 * w4) * n + idx]);
                const uint32_t ln = lens[idx];
                m = (int)(ln & 0xffffu), nn = (int)(ln >> 16);
                m = m > 128 ? 128 : m; /* hurdle_matrix.h:626-627 */
                nn = nn > 128 ? 128 : nn;
                dest_lane = nn - m; /* hurdle_matrix.h:649 */
#pragma unroll
                for (int j = 0; j < NL; j++) {
                    const int lane = j - K;
                    lo_[j] = greedy_lane_vector(A0, A1, B0, B1, lane);
                    if (KEEP_LF) lf_[j] = v_flip_short_hurdles1(lo_[j]);
                    sp[j] = -1; /* hurdle_matrix.h:106-119 */
                    len[j] = 0;
                    nsw[j] = 128;
                    dst[j] = lane_destination(m, nn, lane);
                }
                dest_vec = greedy_lane_vector(A0, A1, B0, B1, dest_lane); /* may lie outside the band (G13) */
                cur_lane = 0, cur_col = 0, cost = 0, guard = 0, ncig = 0;
                pair = out.index(idx);
                finished = false;
            }
        }
#ifdef GREEDY_DIAG
        const unsigned long long dg_b = __builtin_amdgcn_s_memtime();
        dg_refill += dg_b - dg_a;
        dg_iters++;
        dg_lanes += __popcll(__ballot(active && !finished));
#endif
        if (__ballot(active && !finished) == 0ull) break; /* wave-uniform: every lane has drained the queue */
        if (active && !finished) {
            // ---- _update_highway_list ----
            bool reaching = false;
#pragma unroll
            for (int j = 0; j < NL; j++) {
                const int lane = j - K;
                const int start_col = cur_col + fwd_col(cur_lane, lane);
                if (sp[j] < start_col) {
                    int d = lane - cur_lane;
                    nsw[j] = d < 0 ? -d : d;
                    int fz, nx;
                    v_highway_from(KEEP_LF ? lf_[j] : v_flip_short_hurdles1(lo_[j]), start_col, fz, nx);
                    sp[j] = start_col + fz;
                    len[j] = nx;
                    if (start_col + fz + nx > dst[j]) {
                        const int c = dst[j] - (start_col + fz);
                        len[j] = c > 0 ? c : 0;
                        reaching = true;
                    }
                }
                sw[j] = (semi && guard == 0) ? 0 : lane_penalty(cur_lane, lane, o, e);
                nh[j] = v_pop_between(lo_[j], start_col, sp[j] + len[j]);
            }
            double best_h = -__builtin_inf();
            int best_leap = 0; /* -numeric_limits<int>::infinity() == 0 (hurdle_matrix.h:287) */
            int best = 0, best_sp = sp[K], best_len = len[K], best_cost = x * nh[K] + sw[K];
            V128 best_vec = lo_[K];
#pragma unroll
            for (int j = 0; j < NL; j++) {
                const int lane = j - K;
                const int hc = x * nh[j];
                double heur = greedy_significance(args, len[j], nh[j], nsw[j]);
                int leap = -sw[j];
                if (reaching) {
                    const int fsw = semi ? 0 : lane_penalty(lane, dest_lane, o, e);
                    heur = (double)(-sw[j] - hc - fsw - x * (dst[j] - sp[j] - len[j]));
                    leap -= fsw;
                }
                if (heur > best_h || (heur == best_h && leap > best_leap)) {
                    best_h = heur, best_leap = leap, best = lane;
                    best_sp = sp[j], best_len = len[j], best_cost = hc + sw[j], best_vec = lo_[j];
                }
            }
            if (best_len <= 0 || ++guard > 4 * 128) { /* hurdle_matrix.h:358-361 */
                finished = true;
            } else {
                // ---- _choose_best_highway ----
                const int best_from_sp = v_ones_from(best_vec, best_sp);
                int small_inter = best_cost, small_total = best_cost;
                int ch = best, ch_sp = best_sp, ch_len = best_len, ch_cost = best_cost;
#pragma unroll
                for (int j = 0; j < NL; j++) {
                    const int lane = j - K;
                    if (lane != best && !(sp[j] + fwd_col(lane, best) > best_sp)) {
                        const int endp = sp[j] + len[j];
                        const int inter = sw[j] + nh[j]; /* the same range loop 1 counted: [cur_col + fwd, sp + len) */
                        const int tail = x * v_pop_between_pre(best_vec, fwd_col(lane, best) + endp, best_sp, best_from_sp);
                        const int total = inter + lane_penalty(lane, best, o, e) + (tail > 0 ? tail : 0);
                        if (total <= small_total && inter <= small_inter) {
                            small_total = total, small_inter = inter;
                            ch = lane, ch_sp = sp[j], ch_len = len[j], ch_cost = sw[j] + x * nh[j];
                        }
                    }
                }
                // ---- _step commit (hurdle_matrix.h:411-433) ----
                cost += ch_cost;
                if (cig.on()) cig.step(pair, ncig, cur_lane, ch, ch_sp + ch_len - (cur_col + fwd_col(cur_lane, ch)));
                cur_lane = ch;
                cur_col = ch_sp + ch_len;
                if (cur_col >= lane_destination(m, nn, ch)) finished = true;
            }
        }
#ifdef GREEDY_DIAG
        dg_step += __builtin_amdgcn_s_memtime() - dg_b;
#endif
    }
#ifdef GREEDY_DIAG
    if ((threadIdx.x & 63) == 0 && cig.nops != nullptr && cig.ops == nullptr) { /* diag build: cig.nops doubles as the debug buffer */
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(cig.nops) + 8 * (((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
        dbg[0] = dg_refill, dbg[1] = dg_step, dbg[2] = dg_iters, dbg[3] = dg_lanes, dbg[4] = __builtin_amdgcn_s_memtime() - dg_t0;
    }
#endif
}

// --------------------------------------------------------------------------------------------------------
// LEAP (banded affine Landau-Vishkin, "BAG"), unit penalties x = o = e = 1, lanes in registers, one thread
// per pair.  Follows LV::run (LEAP_SIMD/LV_BAG.cpp:127-245) with init(k,200,ED_GLOBAL,1,1,1); the scalar
// character loop count_ID_length (:9-23) becomes a count-trailing-zeros on the lane's mismatch bit-vector.
// Only generation e-1 is live, so three registers per lane replace the four [2k+3][201]
// tables.  W64 = number of 64-bit words per vector (2: len <= 128, 4: len <= 256).
// --------------------------------------------------------------------------------------------------------
template <int W64>
struct VW {
    u64 w[W64];
};

// first set bit at or after `from` in a W64-word vector, or W64*64 when none.  `from` in [0, W64*64].
// Only the word `from` falls into needs its low bits dropped (one 64-bit shift, taken mod 64 by the hardware); the words
// above it count from their own bit 0, the words below it not at all.
template <int W64>
ASM_DEV int vw_next_one(const VW<W64>& v, int from) {
    const int q = from >> 6;
    u64 x = v.w[W64 - 1];
#pragma unroll
    for (int qq = W64 - 2; qq >= 0; qq--) x = q == qq ? v.w[qq] : x;
    const u64 y = x >> (from & 63);
    int res = W64 * 64;
#pragma unroll
    for (int qq = W64 - 1; qq >= 1; qq--) /* descending: the lowest non-empty word above `from` wins */
        if (qq > q && v.w[qq]) res = qq * 64 + __builtin_ctzll(v.w[qq]);
    if (y) res = from + __builtin_ctzll(y);
    return from >= W64 * 64 ? W64 * 64 : res;
}

// The same scan for a caller that keeps, per vector, "first set bit in the words above word q" (W64 * 64 when none) for every q
// but the last: v_ffbl_b32 gives 0xFFFFFFFF for an empty word, and with saturating adds an empty shifted word turns into a candidate
// that loses the final min — no zero tests, no compare-and-select chain behind the shift (asm_bits.h, v_next_one_from_fb).
/* Up to three words per vector the fall-backs cost no registers the compiler was not already spending (it hoists the upper
 * words' ctz out of the generation loop either way: 59 and 90 VGPRs before and after at two and three words).  From four words
 * on they do — 120 -> 146 and 165 -> 209 VGPRs at four and six words, a wave per SIMD less — and the wider classes of C5 lost
 * what the shorter scan gained: those keep vw_next_one. */
#define VW_SCAN_FB(W64) ((W64) <= 3)
template <int W64>
struct VWAbove {
    unsigned fb[W64 > 1 ? W64 - 1 : 1];
};
template <int W64>
ASM_DEV VWAbove<W64> vw_above(const VW<W64>& v) {
    VWAbove<W64> r;
    unsigned run = W64 * 64u;
#pragma unroll
    for (int q = W64 - 2; q >= 0; q--) {
        run = v.w[q + 1] ? (unsigned)(q + 1) * 64u + (unsigned)__builtin_ctzll(v.w[q + 1]) : run;
        r.fb[q] = run;
    }
    return r;
}
template <int W64>
ASM_DEV int vw_next_one_fb(const VW<W64>& v, const VWAbove<W64>& ab, int from) { /* = vw_next_one(v, from) for from >= 0 */
    u64 x = v.w[W64 - 1];
    unsigned f = W64 * 64u;
#pragma unroll
    for (int q = W64 - 2; q >= 0; q--) { /* the nested tests leave the word `from` falls into, and what lies above it */
        const bool below = from < (q + 1) * 64;
        x = below ? v.w[q] : x;
        f = below ? ab.fb[q] : f;
    }
    const u64 y = x >> (from & 63);
    const unsigned c = min(v_ffbl_raw((unsigned)y), __builtin_elementwise_add_sat(v_ffbl_raw((unsigned)(y >> 32)), 32u));
    return (int)min(__builtin_elementwise_add_sat((unsigned)from, c), f);
}

// bit p of the result = bit (p - s) of v (bits move away from index 0), s in [0, 63]
template <int W64>
ASM_DEV VW<W64> vw_away0_small(const VW<W64>& v, int s) {
    VW<W64> r;
#pragma unroll
    for (int q = 0; q < W64; q++) {
        u64 lo = q > 0 ? v.w[q - 1] : 0ull;
        r.w[q] = (v.w[q] << s) | (s ? (lo >> (64 - s)) : 0ull);
    }
    return r;
}

template <int W64>
ASM_DEV VW<W64> vw_low_ones(int len) {
    VW<W64> r;
#pragma unroll
    for (int q = 0; q < W64; q++) {
        const int rel = len - q * 64;
        r.w[q] = rel <= 0 ? 0ull : (rel >= 64 ? ~0ull : ((1ull << rel) - 1ull));
    }
    return r;
}

template <int W64>
ASM_DEV void load_planes(const uint4* __restrict__ planes, long n, int w4, long i, VW<W64>& A0, VW<W64>& A1,
                         VW<W64>& B0, VW<W64>& B1) {
#pragma unroll
    for (int g = 0; g < (W64 + 1) / 2; g++) { /* an odd W64 takes only the low half of its last granule */
        uint4 qa0 = make_uint4(0u, 0u, 0u, 0u), qa1 = qa0, qb0 = qa0, qb1 = qa0;
        if (g < w4) { /* a vector wider than the batch's granule count has empty upper words */
            qa0 = planes[((long)0 * w4 + g) * n + i];
            qa1 = planes[((long)1 * w4 + g) * n + i];
            qb0 = planes[((long)2 * w4 + g) * n + i];
            qb1 = planes[((long)3 * w4 + g) * n + i];
        }
        A0.w[2 * g] = (u64)qa0.x | ((u64)qa0.y << 32), A1.w[2 * g] = (u64)qa1.x | ((u64)qa1.y << 32);
        B0.w[2 * g] = (u64)qb0.x | ((u64)qb0.y << 32), B1.w[2 * g] = (u64)qb1.x | ((u64)qb1.y << 32);
        if (2 * g + 1 < W64) {
            A0.w[2 * g + 1] = (u64)qa0.z | ((u64)qa0.w << 32), A1.w[2 * g + 1] = (u64)qa1.z | ((u64)qa1.w << 32);
            B0.w[2 * g + 1] = (u64)qb0.z | ((u64)qb0.w << 32), B1.w[2 * g + 1] = (u64)qb1.z | ((u64)qb1.w << 32);
        }
    }
}

// Mismatch vector of LEAP lane d = l - mid (LV_BAG.cpp:13-18): position p = max(read idx, ref idx);
// d < 0 compares A[p-|d|] with B[p], d > 0 compares A[p] with B[p-d].  Positions where either string has
// run out (the NUL padding of LV::load_reads, LV_BAG.cpp:116-117) and all positions >= len are mismatches.
template <int W64>
ASM_DEV VW<W64> leap_lane_mask(const VW<W64>& A0, const VW<W64>& A1, const VW<W64>& B0, const VW<W64>& B1,
                               const VW<W64>& VA, const VW<W64>& VB, int d) {
    VW<W64> r;
    const int s = d < 0 ? -d : d;
    if (d < 0) {
        VW<W64> a0 = vw_away0_small<W64>(A0, s), a1 = vw_away0_small<W64>(A1, s), va = vw_away0_small<W64>(VA, s);
#pragma unroll
        for (int q = 0; q < W64; q++) r.w[q] = (a0.w[q] ^ B0.w[q]) | (a1.w[q] ^ B1.w[q]) | ~(va.w[q] & VB.w[q]);
    } else {
        VW<W64> b0 = vw_away0_small<W64>(B0, s), b1 = vw_away0_small<W64>(B1, s), vb = vw_away0_small<W64>(VB, s);
#pragma unroll
        for (int q = 0; q < W64; q++) r.w[q] = (A0.w[q] ^ b0.w[q]) | (A1.w[q] ^ b1.w[q]) | ~(VA.w[q] & vb.w[q]);
    }
    return r;
}

template <int K, int W64>
ASM_DEV int leap_unit_pair(const uint4* __restrict__ planes, const uint32_t* __restrict__ lens, long n, int w4, long i) {
    constexpr int NL = 2 * K + 1;
    const uint32_t ln = lens[i];
    const int m = (int)(ln & 0xffffu), nn = (int)(ln >> 16);
    const int len = m > nn ? m : nn; /* benchmark_utils.h:162 */
    VW<W64> A0, A1, B0, B1;
    load_planes<W64>(planes, n, w4, i, A0, A1, B0, B1);
    const VW<W64> VA = vw_low_ones<W64>(m), VB = vw_low_ones<W64>(nn);

    VW<W64> mask[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) mask[j] = leap_lane_mask<W64>(A0, A1, B0, B1, VA, VB, j - K);

    // Generation e-1 state per lane: `end` only (-2 = never reached, LV_BAG.cpp:95-101).  With o == ext (here 1 == 1) the
    // I and D tables carry no information of their own: I[l][e] is taken from end[l-1][e-o] when that is > I[l-1][e-ext],
    // else from I[l-1][e-ext] (LV_BAG.cpp:166-176) — the same generation e-1 on both sides — and end[.][g] >= I[.][g]
    // whenever I[.][g] >= 0, because a lane's start is max(end+1, I, D) (:186-201) and its end is never before its start.
    // So I[l][e] = end[l-1][e-1] + top if that end is >= 0, else -2; likewise D from lane l+1.  And since -2 + {0,1} and
    // -2 + 1 stay negative, start = max(end+1, end_up+top, end_dn+bot) needs no selects: it is negative exactly when all
    // three sources are -2.
    int en[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) en[j] = -2;
    VWAbove<W64> above[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) above[j] = vw_above<W64>(mask[j]);

    int result = -1;
    // e = 0: only the main diagonal is live in ED_GLOBAL (LV_BAG.cpp:102-104,131-147)
    {
        int e0 = vw_next_one<W64>(mask[K], 0);
        e0 = e0 > len ? len : e0;
        en[K] = e0;
        if (e0 == len) result = 0;
    }
    for (int e = 1; e <= ASM_LEAP_AF_THRESHOLD && result < 0; e++) {
        int en2[NL];
        bool pass = false;
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int d = j - K;
            const int top = d >= 0 ? 1 : 0, bot = d <= 0 ? 1 : 0;
            const int e_up = j > 0 ? en[j - 1] : -2;
            const int e_dn = j < NL - 1 ? en[j + 1] : -2;
            int st = en[j] + 1;                        /* :186-187 */
            st = e_up + top > st ? e_up + top : st;    /* I_pos, :166-176,193-194 */
            st = e_dn + bot > st ? e_dn + bot : st;    /* D_pos, :179-182,200-201 */
            int enew = -2;
            if (st >= 0) {
                const int from = st > len ? len : st;
                /* count_ID_length (:9-23) as a saturating scan: first mismatch at or after `from`, capped at len; enew = max(t, st)
                 * is the reference's "a start beyond the end stays where it is" (t >= from = st whenever st <= len) */
                int t = VW_SCAN_FB(W64) ? vw_next_one_fb<W64>(mask[j], above[j], from) : vw_next_one<W64>(mask[j], from);
                t = t > len ? len : t;
                enew = t > st ? t : st;
                if (enew == len) { /* :220-238 */
                    const int diff = d < 0 ? -d : d;
                    const int conv = e + diff; /* o + (diff-1)*ext with o = ext = 1 */
                    if (conv <= ASM_LEAP_AF_THRESHOLD) pass = true;
                }
            }
            en2[j] = enew;
        }
#pragma unroll
        for (int j = 0; j < NL; j++) en[j] = en2[j];
        if (pass) result = e; /* final_ED (LV_BAG.cpp:228,356-358), not converge_ED */
    }
    return result;
}

template <int K, int W64>
__global__ __launch_bounds__(ASM_BLOCK) void leap_unit_kernel(const uint4* __restrict__ planes,
                                                              const uint32_t* __restrict__ lens, long n, int w4,
                                                              OutMap out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out.put(i, leap_unit_pair<K, W64>(planes, lens, n, w4, i));
}

// Work-sorted form.  A pair needs final_ED + 1 generations, and a wave runs as long as its slowest pair: with pairs in
// input order a wave of 64 runs ~16 generations for a mean of ~8.  When a per-pair work estimate is at hand — the NW
// penalty of the same pair, computed one kernel earlier in `_run_benchmark`, is within one generation of LEAP's count
// for ~98 % of pairs — each workgroup counting-sorts its 256 pairs by that hint in LDS and thread t takes the pair of
// rank t, so each of the four waves works on one quartile of the workgroup's pairs, i.e. on 64 pairs that need
// (nearly) the same number of generations.  The hint only changes the schedule, never a result.
#define LEAP_HINT_PAIRS 256
template <int K, int W64>
__global__ __launch_bounds__(ASM_BLOCK) void leap_unit_hint_kernel(const uint4* __restrict__ planes,
                                                                   const uint32_t* __restrict__ lens, long n, int w4,
                                                                   OutMap out, const int32_t* __restrict__ hint) {
    __shared__ uint16_t s_sorted[LEAP_HINT_PAIRS];
    __shared__ int s_bin[64];
    const int t = threadIdx.x;
    const long base = (long)blockIdx.x * LEAP_HINT_PAIRS;
    const int cnt = (n - base) < LEAP_HINT_PAIRS ? (int)(n - base) : LEAP_HINT_PAIRS;
    if (t < 64) s_bin[t] = 0;
    __syncthreads();
    int key[LEAP_HINT_PAIRS / ASM_BLOCK];
#pragma unroll
    for (int q = 0; q < LEAP_HINT_PAIRS / ASM_BLOCK; q++) {
        const int local = t + q * ASM_BLOCK;
        key[q] = -1;
        if (local < cnt) {
            int h = hint[out.index(base + local)];
            h = h < 0 ? 63 : (h > 63 ? 63 : h);
            key[q] = h;
            atomicAdd(&s_bin[h], 1);
        }
    }
    __syncthreads();
    if (t == 0) { /* exclusive scan of 64 bins */
        int run = 0;
        for (int b = 0; b < 64; b++) {
            const int c = s_bin[b];
            s_bin[b] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < LEAP_HINT_PAIRS / ASM_BLOCK; q++) {
        if (key[q] >= 0) {
            const int slot = atomicAdd(&s_bin[key[q]], 1);
            s_sorted[slot] = (uint16_t)(t + q * ASM_BLOCK);
        }
    }
    __syncthreads();
    for (int q = 0; q < LEAP_HINT_PAIRS / ASM_BLOCK; q++) {
        const int p = t + q * ASM_BLOCK;
        if (p < cnt) {
            const long i = base + s_sorted[p];
            out.put(i, leap_unit_pair<K, W64>(planes, lens, n, w4, i));
        }
    }
}

// --------------------------------------------------------------------------------------------------------
// LEAP with general penalties (x, o, ext), narrow band, one thread per pair.  The recurrence reads generations e-o,
// e-x (end) and e-ext (I, D) (LV_BAG.cpp:166-187), so rings of GM = 2^a > max(x, o) generations of `end` and
// GI = 2^b > ext generations of I and D are kept — per thread, as int16 (positions <= 512), in dynamic LDS laid out
// [ring][slot][lane][thread]: thread-private columns, so no barriers and no bank conflicts; slot offsets are wave-uniform
// (scalar).  All slots are filled with "never reached" before generation 0, which also stands in for the e >= penalty guards
// of the reference (see the loop).
// --------------------------------------------------------------------------------------------------------
#define LEAP_GEN_THREADS 128
struct RingGeometry { /* power-of-two ring depths for (x, o, e) and the LDS bytes a block of T threads with NL lanes needs */
    int gm, gi;
    __host__ __device__ static int pow2_above(int v) {
        int g = 2;
        while (g <= v) g <<= 1;
        return g;
    }
    __host__ RingGeometry(int x, int o, int e) : gm(pow2_above(x > o ? x : o)), gi(pow2_above(e)) {}
    __host__ size_t lds_bytes(int nl, int threads, size_t elem = sizeof(short)) const { return ((size_t)(gm + 2 * gi) * nl * threads * elem + 3) & ~(size_t)3; }
};

template <int K, int W64, typename EnT> /* EnT: ring element, values stored + 2 (0 = never reached) */
__global__ __launch_bounds__(LEAP_GEN_THREADS) void leap_general_kernel(const uint4* __restrict__ planes,
                                                                        const uint32_t* __restrict__ lens, long n, int w4,
                                                                        int x, int o, int ext, int gm, int gi, OutMap out) {
    constexpr int NL = 2 * K + 1, T = LEAP_GEN_THREADS, SLOT = NL * T;
    extern __shared__ uint32_t s_ring32[];
    const int t = threadIdx.x;
    EnT* const r_en = reinterpret_cast<EnT*>(s_ring32) + t;
    EnT* const r_ip = r_en + gm * SLOT;
    EnT* const r_dp = r_ip + gi * SLOT;
    const long i = (long)blockIdx.x * T + t;
    if (i >= n) return;
    const uint32_t ln = lens[i];
    const int m = (int)(ln & 0xffffu), nn = (int)(ln >> 16);
    const int len = m > nn ? m : nn; /* benchmark_utils.h:162 */
    VW<W64> A0, A1, B0, B1;
    load_planes<W64>(planes, n, w4, i, A0, A1, B0, B1);
    const VW<W64> VA = vw_low_ones<W64>(m), VB = vw_low_ones<W64>(nn);
    VW<W64> mask[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) mask[j] = leap_lane_mask<W64>(A0, A1, B0, B1, VA, VB, j - K);
    VWAbove<W64> above[NL]; /* the saturating scan's fall-backs (vw_next_one_fb) */
#pragma unroll
    for (int j = 0; j < NL; j++) above[j] = vw_above<W64>(mask[j]);

    int result = -1;
    // e = 0: only the main diagonal is live (LV_BAG.cpp:102-104,131-147)
#pragma unroll
    for (int j = 0; j < NL; j++) { /* every slot starts as "never reached" (see the note on guards below) */
        for (int q = 0; q < gm; q++) r_en[q * SLOT + j * T] = (EnT)0;
        for (int q = 0; q < gi; q++) r_ip[q * SLOT + j * T] = (EnT)0, r_dp[q * SLOT + j * T] = (EnT)0;
    }
    {
        int e0 = vw_next_one<W64>(mask[K], 0);
        e0 = e0 > len ? len : e0;
        r_en[K * T] = (EnT)(e0 + 2);
        if (e0 == len) result = 0;
    }
    for (int e = 1; e <= ASM_LEAP_AF_THRESHOLD && result < 0; e++) {
        const EnT* const en_o = r_en + ((e - o) & (gm - 1)) * SLOT;
        const EnT* const en_x = r_en + ((e - x) & (gm - 1)) * SLOT;
        const EnT* const ip_e = r_ip + ((e - ext) & (gi - 1)) * SLOT;
        const EnT* const dp_e = r_dp + ((e - ext) & (gi - 1)) * SLOT;
        EnT* const en_w = r_en + (e & (gm - 1)) * SLOT;
        EnT* const ip_w = r_ip + (e & (gi - 1)) * SLOT;
        EnT* const dp_w = r_dp + (e & (gi - 1)) * SLOT;
        // The reference guards these reads with e >= o, e >= x, e >= ext (LV_BAG.cpp:166-187).  Here the guards are implied:
        // for e < o the slot of generation e - o is that of a generation not written yet (ring depth > o), still "never
        // reached" from the initial fill; unguarded, the five reads issue back to back instead of a branch and a wait each.
        bool pass = false;
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int d = j - K;
            const int top = d >= 0 ? 1 : 0, bot = d <= 0 ? 1 : 0;
            const int e_up = j > 0 ? (int)en_o[(j > 0 ? j - 1 : 0) * T] - 2 : -2;
            const int i_up = j > 0 ? (int)ip_e[(j > 0 ? j - 1 : 0) * T] - 2 : -2;
            const int e_dn = j < NL - 1 ? (int)en_o[(j < NL - 1 ? j + 1 : j) * T] - 2 : -2;
            const int d_dn = j < NL - 1 ? (int)dp_e[(j < NL - 1 ? j + 1 : j) * T] - 2 : -2;
            const int own = (int)en_x[j * T] - 2;
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
                int r = VW_SCAN_FB(W64) ? vw_next_one_fb<W64>(mask[j], above[j], from) : vw_next_one<W64>(mask[j], from); /* count_ID_length, :9-23 */
                r = r > len ? len : r;
                enew = r > st ? r : st; /* st beyond the end stays st (r >= from = st otherwise) */
                if (enew == len) { /* :220-238 */
                    const int diff = d < 0 ? -d : d;
                    const int conv = e + (diff ? o + (diff - 1) * ext : 0);
                    if (conv <= ASM_LEAP_AF_THRESHOLD) pass = true;
                }
            }
            en_w[j * T] = (EnT)(enew + 2), ip_w[j * T] = (EnT)(inew + 2), dp_w[j * T] = (EnT)(dnew + 2);
        }
        if (pass) result = e; /* final_ED (LV_BAG.cpp:228,356-358) */
    }
    out.put(i, result);
}

// --------------------------------------------------------------------------------------------------------
// NW with affine gaps as a banded wavefront (furthest-reaching offsets per diagonal and score), one thread per pair —
// the same LDS rings as leap_general_kernel.  Diagonal d = j - i (i: read characters consumed, j: reference characters),
// lanes d in [-K, K].  M[s][d] = furthest i with cost s ending in a match/mismatch state; I consumes the reference
// (d-1 -> d, i unchanged), D consumes the read (d+1 -> d, i+1).  gap(L) = o + (L-1)*e, mismatch x — the model of
// nw_affine_kernel and of the oracle (benchmark_utils.h:139-142,288).  The answer is the first s whose M on diagonal
// n-m reaches i = m.
// Exactness of the band: a path that leaves it touches a diagonal |D| = K+1, which costs at least gap(K+1) to reach
// and gap(K+1-|n-m|) to come back from: bound = 2o + (2K - |n-m|)e.  A banded result s <= bound is therefore the global
// optimum; pairs that do not finish by then (or with |n-m| > K) are appended to `todo` for the full-matrix kernel.
// --------------------------------------------------------------------------------------------------------
#define NW_WFA_K 7
#define NW_WFA_K2 15 /* second pass over the pairs the first band could not settle (maxlen <= 128 only: 31 masks of 4 dwords) */
struct WfaRings { /* ring depths of the banded wavefront: `M` needs scores s-o and s-x, I and D need s-ext; slot = score mod depth (scalar) */
    int gm, gi;
    __host__ WfaRings(int x, int o, int e) : gm((x > o ? x : o) + 1), gi(e + 1) {}
    __host__ size_t lds_bytes(int nl, int threads, size_t elem) const { return ((size_t)(gm + 2 * gi) * nl * threads * elem + 3) & ~(size_t)3; }
};
// EnT: bytes when every position + 2 fits (values are stored + 2, 0 = never reached), else shorts.  LISTED: the pairs come from
// `in_list[0 .. *in_count)` (the previous pass's todo list) instead of 0..n.
template <int K, int W64, typename EnT, bool LISTED>
__global__ __launch_bounds__(LEAP_GEN_THREADS) void nw_wfa_kernel(const uint4* __restrict__ planes,
                                                                  const uint32_t* __restrict__ lens, long n, int w4, int x,
                                                                  int o, int ext, int gm, int gi, OutMap out,
                                                                  const uint32_t* __restrict__ in_list, const uint32_t* __restrict__ in_count,
                                                                  uint32_t* __restrict__ todo, uint32_t* __restrict__ todo_count) {
    constexpr int NL = 2 * K + 1, T = LEAP_GEN_THREADS, SLOT = NL * T;
    extern __shared__ uint32_t s_ring32[];
    EnT* const s_ring = reinterpret_cast<EnT*>(s_ring32);
    const int t = threadIdx.x;
    EnT* const r_m = s_ring + t;
    EnT* const r_i = r_m + gm * SLOT;
    EnT* const r_d = r_i + gi * SLOT;
    const long slot_id = (long)blockIdx.x * T + t;
    long i = slot_id;
    bool have = slot_id < n;
    if constexpr (LISTED) {
        have = slot_id < (long)*in_count;
        i = have ? (long)in_list[slot_id] : 0;
    }
    int result = 0;
    bool unresolved = false;
    if (have) {
        const uint32_t ln = lens[i];
        const int m = (int)(ln & 0xffffu), nn = (int)(ln >> 16);
        const int df = nn - m, adf = df < 0 ? -df : df;
        if (adf > K) {
            unresolved = true;
        } else {
            VW<W64> A0, A1, B0, B1;
            load_planes<W64>(planes, n, w4, i, A0, A1, B0, B1);
            // mismatch-or-outside vector per diagonal, indexed by i: bit p = (A[p] != B[p+d]) or p >= min(m, n-d) or p+d < 0
            VW<W64> mask[NL];
#pragma unroll
            for (int j = 0; j < NL; j++) {
                const int d = j - K, sft = d < 0 ? -d : d;
                const int lim = m < nn - d ? m : nn - d;
                const VW<W64> valid = vw_low_ones<W64>(lim);
                if (d >= 0) { /* bit p of b = B[p+d]: towards index 0 */
#pragma unroll
                    for (int q = 0; q < W64; q++) {
                        const u64 h0 = q + 1 < W64 ? B0.w[q + 1] : 0ull, h1 = q + 1 < W64 ? B1.w[q + 1] : 0ull;
                        const u64 b0 = (B0.w[q] >> sft) | (sft ? (h0 << (64 - sft)) : 0ull);
                        const u64 b1 = (B1.w[q] >> sft) | (sft ? (h1 << (64 - sft)) : 0ull);
                        mask[j].w[q] = (A0.w[q] ^ b0) | (A1.w[q] ^ b1) | ~valid.w[q];
                    }
                } else {
                    const VW<W64> b0 = vw_away0_small<W64>(B0, sft), b1 = vw_away0_small<W64>(B1, sft);
                    const VW<W64> low = vw_low_ones<W64>(sft);
#pragma unroll
                    for (int q = 0; q < W64; q++) mask[j].w[q] = (A0.w[q] ^ b0.w[q]) | (A1.w[q] ^ b1.w[q]) | ~valid.w[q] | low.w[q];
                }
            }
            const int bound = 2 * o + (2 * K - adf) * ext;
            result = -1;
            // every ring slot starts as "never reached": score s only sweeps the diagonals a gap of that cost can reach,
            // |d| <= (s - o) / ext + 1 (a bound that never shrinks), and their neighbours are read before they are ever written
            for (int q = 0; q < gm; q++)
#pragma unroll
                for (int j = 0; j < NL; j++) r_m[q * SLOT + j * T] = (EnT)0;
            for (int q = 0; q < gi; q++)
#pragma unroll
                for (int j = 0; j < NL; j++) r_i[q * SLOT + j * T] = (EnT)0, r_d[q * SLOT + j * T] = (EnT)0;
            {
                const int e0 = vw_next_one<W64>(mask[K], 0); /* <= min(m, n) */
                r_m[K * T] = (EnT)(e0 + 2);
                if (df == 0 && e0 >= m) result = 0;
            }
            // ring slots by score: scalar counters instead of a modulo (s, s-o, s-x, s-ext advance together)
            int sl_w = 0, sl_o = gm - o % gm, sl_x = gm - x % gm; /* slot of s, s-o, s-x in the M ring at s = 0 */
            int sj_w = 0, sj_e = gi - ext % gi;                    /* slot of s, s-ext in the I/D rings */
            sl_o = sl_o == gm ? 0 : sl_o, sl_x = sl_x == gm ? 0 : sl_x, sj_e = sj_e == gi ? 0 : sj_e;
            using LdsEl = __attribute__((address_space(3))) EnT;
            const uint32_t a_m = (uint32_t)(uintptr_t)(LdsEl*)r_m, a_i = (uint32_t)(uintptr_t)(LdsEl*)r_i, a_d = (uint32_t)(uintptr_t)(LdsEl*)r_d;
            const int m2 = m + 2;
            for (int s = 1; s <= bound && result < 0; s++) {
                sl_w = sl_w + 1 == gm ? 0 : sl_w + 1, sl_o = sl_o + 1 == gm ? 0 : sl_o + 1, sl_x = sl_x + 1 == gm ? 0 : sl_x + 1;
                sj_w = sj_w + 1 == gi ? 0 : sj_w + 1, sj_e = sj_e + 1 == gi ? 0 : sj_e + 1;
                // The seven slot addresses of this score, each pinned in a VGPR: a DS instruction takes one VGPR and an
                // immediate, and left alone the compiler re-adds the (scalar) slot offset in front of every access.
                uint32_t p_mo = a_m + (uint32_t)(sl_o * SLOT * (int)sizeof(EnT)), p_mx = a_m + (uint32_t)(sl_x * SLOT * (int)sizeof(EnT));
                uint32_t p_ie = a_i + (uint32_t)(sj_e * SLOT * (int)sizeof(EnT)), p_de = a_d + (uint32_t)(sj_e * SLOT * (int)sizeof(EnT));
                uint32_t p_mw = a_m + (uint32_t)(sl_w * SLOT * (int)sizeof(EnT)), p_iw = a_i + (uint32_t)(sj_w * SLOT * (int)sizeof(EnT));
                uint32_t p_dw = a_d + (uint32_t)(sj_w * SLOT * (int)sizeof(EnT));
                asm volatile("" : "+v"(p_mo), "+v"(p_mx), "+v"(p_ie), "+v"(p_de), "+v"(p_mw), "+v"(p_iw), "+v"(p_dw));
                const LdsEl* const m_o = (const LdsEl*)(uintptr_t)p_mo;
                const LdsEl* const m_x = (const LdsEl*)(uintptr_t)p_mx;
                const LdsEl* const i_e = (const LdsEl*)(uintptr_t)p_ie;
                const LdsEl* const d_e = (const LdsEl*)(uintptr_t)p_de;
                LdsEl* const m_w = (LdsEl*)(uintptr_t)p_mw;
                LdsEl* const i_w = (LdsEl*)(uintptr_t)p_iw;
                LdsEl* const d_w = (LdsEl*)(uintptr_t)p_dw;
                // No "s >= o" style guards on the reads: for s < o the slot of s - o is the slot of a score that has not been
                // written yet (ring depth > o), which still holds the initial "never reached"; likewise s - x and s - ext.
                // Unguarded, the five reads of a lane issue back to back instead of one scalar branch and one wait each.
                const int dmax = s < o ? 0 : (s - o) / ext + 1; /* wave-uniform: every thread is at the same score */
                int done = 0;
#pragma unroll
                for (int j = 0; j < NL; j++) {
                    const int d = j - K;
                    if ((d < 0 ? -d : d) > dmax) continue;
                    // all values in the stored form u = position + 2 (0 = never reached, valid from 2)
                    const int m_lo = j > 0 ? (int)m_o[(j > 0 ? j - 1 : 0) * T] : 0;
                    const int i_lo = j > 0 ? (int)i_e[(j > 0 ? j - 1 : 0) * T] : 0;
                    const int m_hi = j < NL - 1 ? (int)m_o[(j < NL - 1 ? j + 1 : j) * T] : 0;
                    const int d_hi = j < NL - 1 ? (int)d_e[(j < NL - 1 ? j + 1 : j) * T] : 0;
                    const int own = (int)m_x[j * T];
                    int inew = m_lo > i_lo ? m_lo : i_lo;      /* reference character consumed: i stays, j = i + d */
                    inew = (inew >= 2 && inew + (d - 2) <= nn) ? inew : 0;
                    int dnew = m_hi > d_hi ? m_hi : d_hi;      /* read character consumed: i + 1 */
                    dnew = (dnew >= 2 && dnew < m2) ? dnew + 1 : 0;
                    int st = (own >= 2 && own < m2 && own + (d - 1) <= nn) ? own + 1 : 0;
                    st = inew > st ? inew : st;
                    st = dnew > st ? dnew : st;
                    int mnew = 0;
                    if (st >= 2) mnew = vw_next_one<W64>(mask[j], st - 2) + 2;
                    if (d == df && mnew >= m2) done = 1;
                    m_w[j * T] = (EnT)mnew, i_w[j * T] = (EnT)inew, d_w[j * T] = (EnT)dnew;
                }
                if (done) result = s;
            }
            if (result < 0) unresolved = true;
        }
    }
    // wave-aggregated append of the unresolved pairs
    const u64 bal = __ballot(unresolved);
    if (bal) {
        const int lane = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        uint32_t base = 0;
        const int leader = __builtin_ctzll(bal);
        if ((t & 63) == leader) base = atomicAdd(todo_count, (uint32_t)__builtin_popcountll(bal));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
        if (unresolved) todo[base + lane] = (uint32_t)i;
    }
    if (have && !unresolved) out.put(i, result);
}

// --------------------------------------------------------------------------------------------------------
// NW for unit penalties (x = o = e = 1): global edit distance by the Myers/Hyyro bit-parallel recurrence —
// the column of vertical deltas of the DP matrix lives in bit-vectors, one thread per pair, the read is the
// vertical string.  Gives exactly the penalty parasail's NW returns for these scores (benchmark_utils.h:
// 139-142,288 with matrix (0,-1), open = extend = 1, i.e. Levenshtein distance).
// --------------------------------------------------------------------------------------------------------
// Full-height bit-parallel column sweep: W64 64-bit words hold all m vertical deltas.
template <int W64>
ASM_DEV int nw_unit_full(const VW<W64>& A0, const VW<W64>& A1, const VW<W64>& B0, const VW<W64>& B1, int m, int nn) {
    if (m == 0) return nn;
    const VW<W64> VA = vw_low_ones<W64>(m);
    VW<W64> Pv, Mv;
#pragma unroll
    for (int q = 0; q < W64; q++) Pv.w[q] = ~0ull, Mv.w[q] = 0ull;
    int score = m;
    const int top_word = (m - 1) >> 6;
    const u64 top_bit = 1ull << ((m - 1) & 63);
    for (int j = 0; j < nn; j++) {
        // code of text character j, broadcast to all bits
        u64 t0 = 0, t1 = 0;
#pragma unroll
        for (int q = 0; q < W64; q++) {
            if ((j >> 6) == q) {
                t0 = (B0.w[q] >> (j & 63)) & 1ull;
                t1 = (B1.w[q] >> (j & 63)) & 1ull;
            }
        }
        const u64 T0 = 0ull - t0, T1 = 0ull - t1;
        u64 carry = 0, ph_in = 1ull, mh_in = 0ull; /* ph_in = 1: D[0][j] = j (global alignment) */
        int delta = 0;
#pragma unroll
        for (int q = 0; q < W64; q++) {
            const u64 Eq = ~(A0.w[q] ^ T0) & ~(A1.w[q] ^ T1) & VA.w[q];
            const u64 pv = Pv.w[q], mv = Mv.w[q];
            const u64 Xv = Eq | mv;
            const u64 x1 = Eq & pv;
            const u64 s1 = x1 + pv;
            const u64 c1 = s1 < x1 ? 1ull : 0ull;
            const u64 s2 = s1 + carry;
            const u64 c2 = s2 < s1 ? 1ull : 0ull;
            carry = c1 | c2;
            const u64 Xh = (s2 ^ pv) | Eq;
            u64 Ph = mv | ~(Xh | pv);
            u64 Mh = pv & Xh;
            if (q == top_word) {
                delta = (Ph & top_bit) ? 1 : ((Mh & top_bit) ? -1 : 0);
            }
            const u64 ph_out = Ph >> 63, mh_out = Mh >> 63;
            Ph = (Ph << 1) | ph_in;
            Mh = (Mh << 1) | mh_in;
            ph_in = ph_out, mh_in = mh_out;
            Pv.w[q] = Mh | ~(Xv | Ph);
            Mv.w[q] = Ph & Xv;
        }
        score += delta;
    }
    return score;
}

template <int W64>
__global__ __launch_bounds__(ASM_BLOCK) void nw_unit_kernel(const uint4* __restrict__ planes,
                                                            const uint32_t* __restrict__ lens, long n, int w4,
                                                            OutMap out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t ln = lens[i];
    const int m = (int)(ln & 0xffffu), nn = (int)(ln >> 16);
    VW<W64> A0, A1, B0, B1;
    load_planes<W64>(planes, n, w4, i, A0, A1, B0, B1);
    out.put(i, nw_unit_full<W64>(A0, A1, B0, B1, m, nn));
}

// --------------------------------------------------------------------------------------------------------
// NW for unit penalties, BANDED: the same bit-parallel recurrence restricted to a 32-row window that slides down
// the main diagonal one row per column (rows j-15 .. j+16 of column j), so a column costs ~30 VALU instructions on
// one dword instead of ~25 per 32 rows of the full height.  Cells outside the band are taken as "one more than
// their in-band neighbour" (vertical delta +1 for the row entering at the bottom, horizontal delta +1 for the row
// leaving at the top), which makes every in-band value an upper bound of the true DP value and exact whenever an
// optimal path stays inside the band.  Hence: a banded result r <= 15 IS the edit distance (then d <= r <= 15 and
// every optimal path has |i-j| <= d); any other outcome (r > 15, or the end cell outside the band) is recomputed
// by the full-height sweep in the same kernel.  At the benchmark's error rates no pair needs the recompute.
// Rows beyond the read's end hold arbitrary plane bits: they only feed cells below row m, never D[m][n].
// --------------------------------------------------------------------------------------------------------
// The window is W = 32 or 64 rows tall (one dword / one 64-bit pair per vector): rows j-W/2+1 .. j+W/2 of column j,
// proven exact for results up to W/2 - 1.  A kernel tries the narrow window first where the batch is short, then the
// 64-row window, then the full-height sweep.
template <int W>
struct BandWord;
template <>
struct BandWord<32> {
    typedef uint32_t T;
};
template <>
struct BandWord<64> {
    typedef u64 T;
};

struct NoColumnSink {
    static constexpr bool kNeedsColumns = false;
    template <typename WT>
    ASM_DEV void operator()(int, WT, WT) const {}
};

// `sink(j, VP, VN)` sees the vertical delta vectors of every finished column j (1-based, in that column's window
// coordinates); the traceback of asm_cover.h stores them.
// all-ones / all-zeros word from bit r of the text block (one v_bfe_i32 for the 32-bit window)
template <int W>
ASM_DEV typename BandWord<W>::T band_text_bit(typename BandWord<W>::T b, int r);
template <>
ASM_DEV uint32_t band_text_bit<32>(uint32_t b, int r) {
    return (uint32_t)__builtin_amdgcn_sbfe((int)b, r, 1);
}
template <>
ASM_DEV u64 band_text_bit<64>(u64 b, int r) {
    return 0ull - ((b >> r) & 1ull);
}

template <int ND, int W, typename Sink = NoColumnSink> /* ND = plane dwords per string (4 * w4); A arrays carry two zero dwords of padding */
ASM_DEV int nw_band(const uint32_t (&A0)[ND + 2], const uint32_t (&A1)[ND + 2], const uint32_t (&B0)[ND],
                    const uint32_t (&B1)[ND], int m, int nn, const Sink& sink = Sink()) {
    typedef typename BandWord<W>::T WT;
    constexpr int C = W / 2;          /* window top row of column j is max(1, j - C + 1) */
    constexpr int NBLK = ND * 32 / W; /* W-column blocks */
    constexpr WT TOP = (WT)1 << (W - 1);
#define BLK(ARR, q) (W == 32 ? (WT)ARR[(q)] : (WT)((u64)ARR[2 * (q)] | ((u64)ARR[2 * (q) + 1] << 32)))
    WT VP = ~(WT)0, VN = 0; /* column 0: D[i][0] = i */
    int S = W;              /* D[bottom row of the window][column] */
    WT lo0 = BLK(A0, 0), lo1 = BLK(A1, 0), hi0 = 0, hi1 = 0;

#define NW_BAND_COLUMN(SLIDE, BW0, BW1, R)                                                           \
    {                                                                                                 \
        if (SLIDE) {                                                                                  \
            lo0 = (lo0 >> 1) | (hi0 << (W - 1)), hi0 >>= 1;                                           \
            lo1 = (lo1 >> 1) | (hi1 << (W - 1)), hi1 >>= 1;                                           \
            VP = (VP >> 1) | TOP, VN >>= 1;                                                           \
        }                                                                                             \
        const WT T0 = (WT)0 - (((BW0) >> (R)) & (WT)1);                                               \
        const WT T1 = (WT)0 - (((BW1) >> (R)) & (WT)1);                                               \
        const WT Eq = ~((lo0 ^ T0) | (lo1 ^ T1));                                                     \
        const WT D0 = ((((Eq & VP) + VP) ^ VP) | Eq) | VN;                                            \
        const WT HP = VN | ~(D0 | VP);                                                                \
        const WT HN = VP & D0;                                                                        \
        if (SLIDE)                                                                                    \
            S += 1 - (int)(D0 >> (W - 1));                                                            \
        else                                                                                          \
            S += (int)(HP >> (W - 1)) - (int)(HN >> (W - 1));                                         \
        const WT X = (HP << 1) | (WT)1;                                                               \
        VP = (HN << 1) | ~(D0 | X);                                                                   \
        VN = D0 & X;                                                                                  \
    }

    // columns 1..C: the window still sits on rows 1..W
    {
        const WT b0 = BLK(B0, 0), b1 = BLK(B1, 0);
        const int c1 = nn < C ? nn : C;
        for (int r = 0; r < c1; r++) {
            NW_BAND_COLUMN(false, b0, b1, r)
            sink(r + 1, VP, VN);
        }
    }
    // columns C+1..n: slide one row per column; the reservoir's upper word is refilled every W slides.
    if (Sink::kNeedsColumns) {
        // plain form: every column's (VP, VN) in that column's own window coordinates, handed to the sink
#pragma unroll
        for (int bq = 0; bq < NBLK; bq++) {
            const WT b0 = BLK(B0, bq), b1 = BLK(B1, bq);
            const int r0 = bq == 0 ? C : 0;
            int rend = nn - W * bq;
            rend = rend > W ? W : rend;
            for (int r = r0; r < rend; r++) {
                if (r == C) hi0 = BLK(A0, bq + 1), hi1 = BLK(A1, bq + 1); /* wave-uniform */
                NW_BAND_COLUMN(true, b0, b1, r)
                sink(W * bq + r + 1, VP, VN);
            }
        }
    } else if (nn > C) {
        // Fused form (penalty only).  Algebraically the same recurrence: instead of producing a column's vertical deltas
        // in its own window and shifting them for the next column, produce them directly in the NEXT column's window:
        //   VPin' = HN | ~((D0 >> 1) | HP) | TOP ,  VNin' = HP & (D0 >> 1)
        // (the "+1 for the row entering at the bottom" is the TOP bit; the "+1 above the window" is the zero shifted
        // into D0 >> 1).  Three instructions fewer per column.  The bottom-row diagonal deltas are shifted into an
        // accumulator and counted once per block instead of being added column by column.
        WT VPin = (VP >> 1) | TOP, VNin = VN >> 1;
#pragma unroll
        for (int bq = 0; bq < NBLK; bq++) {
            const WT b0 = BLK(B0, bq), b1 = BLK(B1, bq);
            const int r0 = bq == 0 ? C : 0;
            int rend = nn - W * bq;
            rend = rend > W ? W : rend;
            WT acc = 0;
            // the block's columns in two runs: the pattern window of column 32*bq + r starts at row 32*bq + r - (C-1), i.e.
            // in pattern word bq-1 for r < C-1 and in word bq from there on
#pragma unroll
            for (int half = 0; half < 2; half++) {
                constexpr int CB = W == 32 ? C - 1 : C;
                const int ra = half == 0 ? r0 : (r0 > CB ? r0 : CB);
                const int rb = half == 0 ? (rend < CB ? rend : CB) : rend;
                if (W != 32 && half == 1) hi0 = BLK(A0, bq + 1), hi1 = BLK(A1, bq + 1);
                // W = 32: the window is cut straight out of two adjacent pattern words with one v_alignbit_b32 (wave-uniform
                // shift), no sliding state to update
                const uint32_t p0l = W == 32 ? (half == 0 ? (bq > 0 ? A0[bq > 0 ? bq - 1 : 0] : 0u) : A0[bq]) : 0u;
                const uint32_t p0h = W == 32 ? (half == 0 ? A0[bq] : A0[bq + 1]) : 0u;
                const uint32_t p1l = W == 32 ? (half == 0 ? (bq > 0 ? A1[bq > 0 ? bq - 1 : 0] : 0u) : A1[bq]) : 0u;
                const uint32_t p1h = W == 32 ? (half == 0 ? A1[bq] : A1[bq + 1]) : 0u;
                const int shb = half == 0 ? W - CB : -CB;
                for (int r = ra; r < rb; r++) {
                    if (W == 32) {
                        lo0 = (WT)__builtin_amdgcn_alignbit(p0h, p0l, (uint32_t)(r + shb));
                        lo1 = (WT)__builtin_amdgcn_alignbit(p1h, p1l, (uint32_t)(r + shb));
                    } else {
                        lo0 = (lo0 >> 1) | (hi0 << (W - 1)), hi0 >>= 1;
                        lo1 = (lo1 >> 1) | (hi1 << (W - 1)), hi1 >>= 1;
                    }
                    const WT T0 = band_text_bit<W>(b0, r), T1 = band_text_bit<W>(b1, r);
                    const WT Eq = ~((lo0 ^ T0) | (lo1 ^ T1));
                    const WT D0 = ((((Eq & VPin) + VPin) ^ VPin) | Eq) | VNin;
                    const WT HP = VNin | ~(D0 | VPin);
                    const WT HN = VPin & D0;
                    acc = (acc << 1) | (D0 >> (W - 1));
                    const WT D0s = D0 >> 1;
                    VPin = HN | ~(D0s | HP) | TOP;
                    VNin = HP & D0s;
                }
            }
            if (rend > r0) S += (rend - r0) - (W == 32 ? __popc((uint32_t)acc) : __popcll((u64)acc));
        }
        // back to the last column's own window: VP = VPin << 1 (its bit 0 is always 0); VN = VNin << 1 | D0[0] — only
        // the bits above row m are needed below, and bit 0 never is
        VP = VPin << 1;
        VN = VNin << 1;
    }
#undef NW_BAND_COLUMN
#undef BLK
    const int top = nn > C - 1 ? nn - (C - 1) : 1; /* window top row of the last column */
    const int bstar = m - top;                     /* bit of row m */
    if (bstar < 0 || bstar > W - 1) return -1;
    const WT above = bstar == W - 1 ? (WT)0 : (~(WT)0 << (bstar + 1));
    const int up = W == 32 ? __popc((uint32_t)(VP & above)) : __popcll((u64)(VP & above));
    const int dn = W == 32 ? __popc((uint32_t)(VN & above)) : __popcll((u64)(VN & above));
    const int result = S - up + dn;
    return result <= C - 1 ? result : -1; /* proven exact only up to W/2 - 1 */
}

template <int ND, int FIRSTW>
ASM_DEV int nw_banded_pair(const uint4* __restrict__ planes, const uint32_t* __restrict__ lens, long n, int w4, long i) {
    const uint32_t ln = lens[i];
    const int m = (int)(ln & 0xffffu), nn = (int)(ln >> 16);
    uint32_t A0[ND + 2], A1[ND + 2], B0[ND], B1[ND];
#pragma unroll
    for (int g = 0; g < ND / 4; g++) {
        uint4 q;
        q = planes[((long)0 * w4 + g) * n + i];
        A0[4 * g] = q.x, A0[4 * g + 1] = q.y, A0[4 * g + 2] = q.z, A0[4 * g + 3] = q.w;
        q = planes[((long)1 * w4 + g) * n + i];
        A1[4 * g] = q.x, A1[4 * g + 1] = q.y, A1[4 * g + 2] = q.z, A1[4 * g + 3] = q.w;
        q = planes[((long)2 * w4 + g) * n + i];
        B0[4 * g] = q.x, B0[4 * g + 1] = q.y, B0[4 * g + 2] = q.z, B0[4 * g + 3] = q.w;
        q = planes[((long)3 * w4 + g) * n + i];
        B1[4 * g] = q.x, B1[4 * g + 1] = q.y, B1[4 * g + 2] = q.z, B1[4 * g + 3] = q.w;
    }
    A0[ND] = A1[ND] = A0[ND + 1] = A1[ND + 1] = 0u;

    int result = -1;
    if (FIRSTW == 32) result = nw_band<ND, 32>(A0, A1, B0, B1, m, nn);
    if (result < 0) result = nw_band<ND, 64>(A0, A1, B0, B1, m, nn);
    if (result < 0) {
        // outside both proven bands: full-height sweep for this lane
        constexpr int W64 = ND / 2;
        VW<W64> a0, a1, b0, b1;
#pragma unroll
        for (int q = 0; q < W64; q++) {
            a0.w[q] = (u64)A0[2 * q] | ((u64)A0[2 * q + 1] << 32);
            a1.w[q] = (u64)A1[2 * q] | ((u64)A1[2 * q + 1] << 32);
            b0.w[q] = (u64)B0[2 * q] | ((u64)B0[2 * q + 1] << 32);
            b1.w[q] = (u64)B1[2 * q] | ((u64)B1[2 * q + 1] << 32);
        }
        result = nw_unit_full<W64>(a0, a1, b0, b1, m, nn);
    }
    return result;
}
// Mixed-length buckets (config C5): the banded sweep costs one step per reference character, and a wave runs as long as its
// longest pair — with the pairs of a width class in input order a wave of 64 holds lengths from all over the class's 128-base
// range and two thirds... three quarters of its lane-steps do work (lane utilisation 0.76 measured).  BYLEN: each workgroup
// counting-sorts its window of NW_SORT_PAIRS slots by reference length in LDS (128 bins: one per base of the class's range) and
// thread t takes the slot of rank t (t + 256, ... for wider windows): a wave then works on a quarter of the window's lengths.
// Loads stay inside the window (4 KB per plane granule: what a wave leaves of a sector its neighbours take from the cache),
// results go where they always went.
#define NW_SORT_PAIRS 256 /* slots a workgroup sorts by length: one per thread, a wave gets a quarter of the window's lengths.
                             Measured at C5, 10^7 pairs: no sort 2.31 ms, 256 slots 2.07, 1024 slots (four per thread) 2.21 — the
                             wider window is sorted better but its scattered 16-byte loads no longer share their sectors in cache */
template <int ND, int FIRSTW, bool BYLEN> /* plane dwords per string: 4 * w4; FIRSTW = 32 or 64: the first window tried */
__global__ __launch_bounds__(ASM_BLOCK) void nw_banded_kernel(const uint4* __restrict__ planes,
                                                              const uint32_t* __restrict__ lens, long n, int w4,
                                                              OutMap out) {
    if (!BYLEN) {
        const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
        if (i < n) out.put(i, nw_banded_pair<ND, FIRSTW>(planes, lens, n, w4, i));
        return;
    }
    __shared__ uint16_t s_sorted[NW_SORT_PAIRS];
    __shared__ int s_bin[128];
    const int t = threadIdx.x;
    const long base = (long)blockIdx.x * NW_SORT_PAIRS;
    const int cnt = (n - base) < NW_SORT_PAIRS ? (int)(n - base) : NW_SORT_PAIRS;
    if (t < 128) s_bin[t] = 0;
    __syncthreads();
    int key[NW_SORT_PAIRS / ASM_BLOCK];
#pragma unroll
    for (int q = 0; q < NW_SORT_PAIRS / ASM_BLOCK; q++) {
        const int local = t + q * ASM_BLOCK;
        key[q] = -1;
        if (local < cnt) {
            const int cols = (int)(lens[base + local] >> 16) - 128 * (w4 - 1) - 1; /* 0 .. 127 inside the class */
            key[q] = cols < 0 ? 0 : (cols > 127 ? 127 : cols);
            atomicAdd(&s_bin[key[q]], 1);
        }
    }
    __syncthreads();
    if (t < 64) { /* exclusive scan of 128 bins by one wave: two bins per lane, wave prefix by DPP-free shuffles */
        const int a = s_bin[2 * t], c = s_bin[2 * t + 1];
        int v = a + c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(v, d, 64);
            if (t >= d) v += up;
        }
        s_bin[2 * t] = v - a - c;
        s_bin[2 * t + 1] = v - c;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NW_SORT_PAIRS / ASM_BLOCK; q++)
        if (key[q] >= 0) s_sorted[atomicAdd(&s_bin[key[q]], 1)] = (uint16_t)(t + q * ASM_BLOCK);
    __syncthreads();
#pragma unroll 1
    for (int q = 0; q < NW_SORT_PAIRS / ASM_BLOCK; q++) {
        const int rank = t + q * ASM_BLOCK;
        if (rank < cnt) {
            const long i = base + s_sorted[rank];
            out.put(i, nw_banded_pair<ND, FIRSTW>(planes, lens, n, w4, i));
        }
    }
}

// --------------------------------------------------------------------------------------------------------
// Seed-hit batches, the shape of the reference's read mapper (GASMA/mapper/main.cpp:77-86): for a hit of read i at
// reference position p the aligner sees reference[start, start + len_i + 1) with start = p ? p - 1 : 0 (clipped at
// the reference's end).  The reference text stays resident in HBM; windows are gathered on the device.
// --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ASM_BLOCK) void hit_window_lengths_kernel(const uint32_t* __restrict__ read_off,
                                                                       const unsigned long long* __restrict__ hit_pos,
                                                                       unsigned long long ref_len, long n,
                                                                       uint32_t* __restrict__ win_len) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) {
        win_len[n] = 0u; /* so that an exclusive scan over n+1 entries yields the total */
        return;
    }
    const unsigned long long p = hit_pos[i];
    const unsigned long long start = p ? p - 1ull : 0ull;
    const unsigned long long want = (unsigned long long)(read_off[i + 1] - read_off[i]) + 1ull;
    const unsigned long long room = start < ref_len ? ref_len - start : 0ull;
    win_len[i] = (uint32_t)(want < room ? want : room);
}

__global__ __launch_bounds__(ASM_BLOCK) void hit_window_gather_kernel(const char* __restrict__ reference,
                                                                      const unsigned long long* __restrict__ hit_pos,
                                                                      const uint32_t* __restrict__ ref_off, long n,
                                                                      char* __restrict__ refs) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    for (long i = wave; i < n; i += nwaves) { /* one wave per hit: 64 contiguous bytes per load */
        const unsigned long long p = hit_pos[i];
        const unsigned long long start = p ? p - 1ull : 0ull;
        const uint32_t o = ref_off[i], len = ref_off[i + 1] - o;
        for (uint32_t q = (uint32_t)lane; q < len; q += 64u) refs[o + q] = reference[start + q];
    }
}

// accuracy counters (benchmark_utils.h:249-255).  A single hot word saturates at ~88 atomics/us on this chip
// (MI355X_MICROARCH.md "dequeue"), so: wave shuffle -> LDS -> ONE atomic per workgroup, and a small grid.
ASM_DEV unsigned int block_sum_256(unsigned int v, unsigned int* s_part /* [4] */) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned int r = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(ASM_BLOCK) void count_equal_kernel(const int32_t* __restrict__ a,
                                                                const int32_t* __restrict__ b, long n,
                                                                unsigned long long* __restrict__ count) {
    __shared__ unsigned int s_part[4];
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    unsigned int local = 0;
    for (; i < n; i += stride) local += (a[i] == b[i]) ? 1u : 0u;
    const unsigned int tot = block_sum_256(local, s_part);
    if (threadIdx.x == 0 && tot) atomicAdd(count, (unsigned long long)tot);
}

// All of `_run_benchmark`'s counters in one pass: counters = {total_tests, nw_correct, LEAP_correct,
// greedy_correct}; the correct answer is answers[i] when given and != INT32_MIN, else the NW penalty
// (benchmark_utils.h:249-252).  leap / greedy may be null (aligner not run).
__global__ __launch_bounds__(ASM_BLOCK) void accuracy_kernel(const int32_t* __restrict__ nw,
                                                             const int32_t* __restrict__ leap,
                                                             const int32_t* __restrict__ greedy,
                                                             const int32_t* __restrict__ answers, long n,
                                                             unsigned long long* __restrict__ counters) {
    __shared__ unsigned int s_part[4];
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    unsigned int c_nw = 0, c_leap = 0, c_greedy = 0;
    // four pairs per thread and iteration: 16-byte loads, and enough independent loads in flight to cover HBM latency.
    // Without NW (nw == nullptr) the correct answer of a pair is its entry of the answers file, or nothing (INT32_MIN never
    // equals a penalty): total_tests still counts every pair.
    const long n4 = (nw != nullptr || answers != nullptr) ? n >> 2 : 0;
    const long n_cmp = (nw != nullptr || answers != nullptr) ? n : 0;
    const int4* nw4 = reinterpret_cast<const int4*>(nw);
    const int4* leap4 = reinterpret_cast<const int4*>(leap);
    const int4* greedy4 = reinterpret_cast<const int4*>(greedy);
    const int4* ans4 = reinterpret_cast<const int4*>(answers);
    // ... and ACC_AHEAD such groups in flight per thread: with one group per iteration the kernel is a chain of HBM round trips
    // (eight of them at 10^6 pairs), which is what it costs inside a step, where it sits on the NW -> LEAP -> counters chain.
    constexpr int ACC_AHEAD = 4;
    for (long q0 = i; q0 < n4; q0 += ACC_AHEAD * stride) {
        int4 p[ACC_AHEAD], a[ACC_AHEAD], l[ACC_AHEAD], g[ACC_AHEAD];
#pragma unroll
        for (int u = 0; u < ACC_AHEAD; u++) {
            const long q = q0 + u * stride;
            const long qq = q < n4 ? q : q0; /* a group beyond the end re-reads the first one and is not counted */
            p[u] = a[u] = l[u] = g[u] = make_int4(INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN);
            if (nw != nullptr) p[u] = nw4[qq];
            if (answers != nullptr) a[u] = ans4[qq];
            if (leap != nullptr) l[u] = leap4[qq];
            if (greedy != nullptr) g[u] = greedy4[qq];
        }
#pragma unroll
        for (int u = 0; u < ACC_AHEAD; u++) {
            if (q0 + u * stride >= n4) break;
            int4 want = p[u];
            if (answers != nullptr) {
                want.x = a[u].x != INT32_MIN ? a[u].x : p[u].x, want.y = a[u].y != INT32_MIN ? a[u].y : p[u].y;
                want.z = a[u].z != INT32_MIN ? a[u].z : p[u].z, want.w = a[u].w != INT32_MIN ? a[u].w : p[u].w;
            }
            if (nw != nullptr) c_nw += (p[u].x == want.x) + (p[u].y == want.y) + (p[u].z == want.z) + (p[u].w == want.w);
            if (leap != nullptr) c_leap += (l[u].x == want.x) + (l[u].y == want.y) + (l[u].z == want.z) + (l[u].w == want.w);
            if (greedy != nullptr) c_greedy += (g[u].x == want.x) + (g[u].y == want.y) + (g[u].z == want.z) + (g[u].w == want.w);
        }
    }
    for (long r = (n4 << 2) + i; r < n_cmp; r += stride) { /* the last n mod 4 pairs */
        const int32_t p = nw != nullptr ? nw[r] : INT32_MIN;
        int32_t want = p;
        if (answers != nullptr && answers[r] != INT32_MIN) want = answers[r];
        if (nw != nullptr) c_nw += (p == want) ? 1u : 0u;
        if (leap != nullptr) c_leap += (leap[r] == want) ? 1u : 0u;
        if (greedy != nullptr) c_greedy += (greedy[r] == want) ? 1u : 0u;
    }
    const unsigned int t_nw = block_sum_256(c_nw, s_part);
    const unsigned int t_leap = block_sum_256(c_leap, s_part);
    const unsigned int t_greedy = block_sum_256(c_greedy, s_part);
    if (threadIdx.x == 0) {
        if (blockIdx.x == 0) atomicAdd(&counters[0], (unsigned long long)n);
        if (t_nw) atomicAdd(&counters[1], (unsigned long long)t_nw);
        if (t_leap) atomicAdd(&counters[2], (unsigned long long)t_leap);
        if (t_greedy) atomicAdd(&counters[3], (unsigned long long)t_greedy);
    }
}

// --------------------------------------------------------------------------------------------------------
// Device generator (asm_batch_generate): same inline code as the host generator (asm_gen.h).
// --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ASM_BLOCK) void gen_lengths_kernel(asm_gen_config cfg, long first, long n,
                                                                uint32_t* __restrict__ mlen,
                                                                uint32_t* __restrict__ nlen) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int m, nn;
    asm_gen_lengths(&cfg, (uint64_t)(first + i), &m, &nn);
    mlen[i] = (uint32_t)m;
    nlen[i] = (uint32_t)nn;
}

__global__ __launch_bounds__(64) void gen_fill_kernel(asm_gen_config cfg, long first, long n,
                                                      const uint32_t* __restrict__ read_off,
                                                      const uint32_t* __restrict__ ref_off,
                                                      char* __restrict__ reads, char* __restrict__ refs) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    char rd[ASM_MAX_LENGTH + 8];
    char tx[ASM_GEN_MAX_TEXT];
    int m, nn;
    asm_gen_pair(&cfg, (uint64_t)(first + i), rd, tx, &m, &nn);
    char* r = reads + read_off[i];
    char* t = refs + ref_off[i];
    for (int j = 0; j < m; j++) r[j] = rd[j];
    for (int j = 0; j < nn; j++) t[j] = tx[j];
}
