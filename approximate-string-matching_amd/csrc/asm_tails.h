// Stale-tail resolver for ASM_GREEDY_SEQUENTIAL — device side.
//
// hurdle_matrix keeps two persistent 128-byte buffers (GASMA/hurdle_matrix.h:136-137).  reset() copies only the
// first m / n characters in (:630-631) and sse3_convert2bit1 then permutes each whole buffer in place
// (GASMA/bit_convert.cpp:265-330: after[q] = before[SRC[q]], SRC[q] = 8*(q mod 16) + P[q div 16], P = {0,2,1,3,4,6,5,7}).
// So the bytes beyond a string's end that the conversion of pair t sees are scrambled characters of earlier pairs.
//
// Parallel formulation.  Follow one buffer slot forward in time: a byte sitting in slot s when pair t is converted
// sits in slot SRC^-1[s] when pair t+1 is converted — unless pair t+1's string is long enough to overwrite it.  So
// each of the 128 slots of a side starts a *trajectory* s, SRC^-1[s], SRC^-1[SRC^-1[s]], ... that carries "the code
// of the last character written on this trajectory".  SRC has order 10 (cycles of length 1, 2, 5, 10), so after any
// multiple of 10 pairs every trajectory is back in its starting slot; with chunks of 640 pairs the trajectories of
// consecutive chunks line up by thread index and the carried state is just a 2-bit code:
//   pass 1  tails_chunk_kernel : per chunk and trajectory, the code of the last write inside the chunk (or none)
//   pass 2  tails_carry_kernel : exclusive "last write wins" prefix over chunks (initial buffers = NUL, code 00)
//   pass 3  tails_emit_kernel  : replay each chunk from its carry-in; wherever the trajectory's slot lies beyond the
//                                pair's string, that pair's conversion sees the carried code: set the bits in `tails`
// Character codes are read from the clean-mode bit planes (bit q of a plane <-> character q), so the passes touch no
// ASCII.  The pack kernel ORs `tails` into granule 0.  Initial buffer content is pinned to zero (the reference's is
// indeterminate heap memory).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "asm_host.h" /* TAIL_NONE, tail_slot_after: the host side of the chain */

#define TAIL_CHUNK 640  /* pairs per chunk: a multiple of 10 (the order of SRC); 10^6 pairs = 1563 workgroups, one round of the chip */
#define TAIL_SUB 256    /* pairs whose tail planes are accumulated in LDS at a time */
#define TAIL_BATCH 16   /* pairs whose plane words are fetched ahead (the walk is a chain of dependent round trips to HBM) */

// SRC^-1: SRC[q] = 8*(q & 15) + P[q >> 4] and P is an involution, so q = (P[y & 7] << 4) | (y >> 3).
__device__ __forceinline__ int tail_src_inv(int y) {
    const int v = y & 7;
    const int low2 = v & 3;
    const int pv = (low2 == 1 || low2 == 2) ? (v ^ 3) : v; /* P swaps 1<->2 and 5<->6 */
    return (pv << 4) | (y >> 3);
}

struct TailLane {
    int side, slot;
    const uint32_t *p0, *p1; /* this side's two bit planes, granule 0, viewed as dwords (4 per pair) */
};

__device__ __forceinline__ TailLane tail_lane_init(const uint4* planes, long n, int w4) {
    TailLane tl;
    tl.side = threadIdx.x >> 7;
    tl.slot = threadIdx.x & 127;
    tl.p0 = reinterpret_cast<const uint32_t*>(planes + ((long)(2 * tl.side) * w4) * n);
    tl.p1 = reinterpret_cast<const uint32_t*>(planes + ((long)(2 * tl.side + 1) * w4) * n);
    return tl;
}

// One batch of up to TAIL_BATCH pairs starting at t: all loads first (the slot sequence is data independent), then
// the sequential replay.  EMIT=false: only track the carried code.  EMIT=true: also set tail bits in LDS.
template <bool EMIT>
__device__ __forceinline__ void tail_batch(TailLane& tl, const uint32_t* __restrict__ lens, long t, int cnt,
                                           uint32_t& code, uint32_t* s_tail, long sub_base) {
    int slots[TAIL_BATCH];
    uint32_t L[TAIL_BATCH], w0[TAIL_BATCH], w1[TAIL_BATCH];
    int s = tl.slot;
#pragma unroll
    for (int i = 0; i < TAIL_BATCH; i++) {
        slots[i] = s;
        s = tail_src_inv(s);
        const long tt = t + (i < cnt ? i : cnt - 1);
        const uint32_t ln = lens[tt];
        uint32_t len = tl.side ? (ln >> 16) : (ln & 0xffffu);
        L[i] = len > 128u ? 128u : len; /* hurdle_matrix.h:626-627 */
        w0[i] = tl.p0[tt * 4 + (slots[i] >> 5)];
        w1[i] = tl.p1[tt * 4 + (slots[i] >> 5)];
    }
#pragma unroll
    for (int i = 0; i < TAIL_BATCH; i++) {
        if (i < cnt) {
            const int q = slots[i];
            if ((uint32_t)q < L[i]) {
                code = ((w0[i] >> (q & 31)) & 1u) | (((w1[i] >> (q & 31)) & 1u) << 1); /* this pair overwrites the slot */
            } else if (EMIT && code != 0u) {
                /* slot q lies beyond pair (t+i)'s string: its conversion sees the carried character */
                uint32_t* row = s_tail + ((size_t)(t + i - sub_base) * 4 + 2 * tl.side) * 4;
                if (code & 1u) atomicOr(&row[q >> 5], 1u << (q & 31));
                if (code & 2u) atomicOr(&row[4 + (q >> 5)], 1u << (q & 31));
            }
            tl.slot = tail_src_inv(q);
        }
    }
}

__global__ __launch_bounds__(256) void tails_chunk_kernel(const uint4* __restrict__ planes,
                                                          const uint32_t* __restrict__ lens, long n, int w4,
                                                          uint8_t* __restrict__ chunk_last /* [nchunks][256] */) {
    TailLane tl = tail_lane_init(planes, n, w4);
    const long t0 = (long)blockIdx.x * TAIL_CHUNK;
    const long t1 = t0 + TAIL_CHUNK < n ? t0 + TAIL_CHUNK : n;
    uint32_t code = TAIL_NONE;
    for (long t = t0; t < t1; t += TAIL_BATCH) {
        const int cnt = (t1 - t) < TAIL_BATCH ? (int)(t1 - t) : TAIL_BATCH;
        tail_batch<false>(tl, lens, t, cnt, code, nullptr, 0);
    }
    chunk_last[(long)blockIdx.x * 256 + threadIdx.x] = (uint8_t)code;
}

// `init` ([256] = [side][slot]): the codes the two buffers hold before the first pair of this batch — zeros (NUL
// bytes) for a batch that starts a file, the state after the previous shard/chunk otherwise (asm_tail_state_advance).
// `summary` (optional, [256]): per trajectory, indexed by its slot before the first pair, the code of the last character the
// batch wrote on it, or TAIL_NONE when the batch never touched it — the batch's whole effect on the buffers.
struct TailState { /* passed by value: no host buffer has to outlive the enqueue */
    uint8_t code[256];
};
// `init` ([256] = [side][slot]): the codes the two buffers hold before the first pair of this batch — zeros (NUL
// bytes) for a batch that starts a file, the state after the previous shard/chunk otherwise (asm_tail_state_advance).
// `summary` (optional, [256]): per trajectory, indexed by its slot before the first pair, the code of the last character the
// batch wrote on it, or TAIL_NONE when the batch never touched it — the batch's whole effect on the buffers.
// One workgroup of 1024 threads = 256 trajectories x 4 segments of the chunk sequence: every thread first finds the last
// write inside its segment, the four meet in LDS, then every thread replays its segment from the right carry-in — two walks of
// nchunks / 4 entries with 16 independent loads in flight, instead of one dependent walk over all chunks.
#define TAIL_CARRY_SEGS 4
__global__ __launch_bounds__(256 * TAIL_CARRY_SEGS) void tails_carry_kernel(const uint8_t* __restrict__ chunk_last,
                                                                            uint8_t* __restrict__ carry_in, long nchunks,
                                                                            TailState init, uint8_t* __restrict__ summary) {
    __shared__ uint8_t s_seg[TAIL_CARRY_SEGS][256];
    const int tr = threadIdx.x & 255, seg = threadIdx.x >> 8;
    const long per = (nchunks + TAIL_CARRY_SEGS - 1) / TAIL_CARRY_SEGS;
    const long c0 = seg * per, c1 = c0 + per < nchunks ? c0 + per : nchunks;
    uint32_t last = TAIL_NONE;
    for (long c = c0; c < c1; c += 16) {
        uint32_t v[16];
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = c + q < c1 ? chunk_last[(c + q) * 256 + tr] : TAIL_NONE;
#pragma unroll
        for (int q = 0; q < 16; q++)
            if (v[q] != TAIL_NONE) last = v[q];
    }
    s_seg[seg][tr] = (uint8_t)last;
    __syncthreads();
    uint32_t cur = (uint32_t)init.code[tr]; /* a file starts with NUL bytes: code 00 */
    for (int q = 0; q < seg; q++)
        if (s_seg[q][tr] != TAIL_NONE) cur = s_seg[q][tr];
    for (long c = c0; c < c1; c += 16) {
        uint32_t v[16];
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = c + q < c1 ? chunk_last[(c + q) * 256 + tr] : TAIL_NONE;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            if (c + q < c1) carry_in[(c + q) * 256 + tr] = (uint8_t)cur;
            if (v[q] != TAIL_NONE) cur = v[q];
        }
    }
    if (summary && seg == TAIL_CARRY_SEGS - 1) {
        uint32_t all = TAIL_NONE;
        for (int q = 0; q < TAIL_CARRY_SEGS; q++)
            if (s_seg[q][tr] != TAIL_NONE) all = s_seg[q][tr];
        summary[tr] = (uint8_t)all;
    }
}

__global__ __launch_bounds__(256) void tails_emit_kernel(const uint4* __restrict__ planes,
                                                         const uint32_t* __restrict__ lens, long n, int w4,
                                                         const uint8_t* __restrict__ carry_in,
                                                         uint4* __restrict__ tails /* [4][n] */) {
    __shared__ uint32_t s_tail[TAIL_SUB * 16]; /* [pair][plane A0,A1,B0,B1][4 dwords] */
    TailLane tl = tail_lane_init(planes, n, w4);
    const long t0 = (long)blockIdx.x * TAIL_CHUNK;
    const long t1 = t0 + TAIL_CHUNK < n ? t0 + TAIL_CHUNK : n;
    uint32_t code = carry_in[(long)blockIdx.x * 256 + threadIdx.x];
    for (long sb = t0; sb < t1; sb += TAIL_SUB) {
        const long se = sb + TAIL_SUB < t1 ? sb + TAIL_SUB : t1;
        for (int i = threadIdx.x; i < TAIL_SUB * 16; i += 256) s_tail[i] = 0u;
        __syncthreads();
        for (long t = sb; t < se; t += TAIL_BATCH) {
            const int cnt = (se - t) < TAIL_BATCH ? (int)(se - t) : TAIL_BATCH;
            tail_batch<true>(tl, lens, t, cnt, code, s_tail, sb);
        }
        __syncthreads();
        // flush: thread i owns pair sb+i
        const long t = sb + threadIdx.x;
        if (t < se) {
            const uint32_t* row = s_tail + (size_t)threadIdx.x * 16;
#pragma unroll
            for (int p = 0; p < 4; p++)
                tails[(long)p * n + t] = make_uint4(row[4 * p], row[4 * p + 1], row[4 * p + 2], row[4 * p + 3]);
        }
        __syncthreads();
    }
}
