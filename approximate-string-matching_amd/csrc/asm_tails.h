// Stale-tail resolver for ASM_GREEDY_SEQUENTIAL — device side.
//
// hurdle_matrix keeps two persistent 128-byte buffers (GASMA/hurdle_matrix.h:136-137).  reset() copies only the
// first m / n characters in (:630-631) and sse3_convert2bit1 then permutes each whole buffer in place
// (GASMA/bit_convert.cpp:265-330: after[q] = before[SRC[q]], SRC[q] = 8*(q mod 16) + P[q div 16], P = {0,2,1,3,4,6,5,7}).
// So the bytes beyond a string's end that the conversion of pair t sees are scrambled characters of earlier pairs.
//
// Parallel formulation (round 4: bit-parallel; rounds 2-3 walked one trajectory per thread and set the tail bits with LDS
// atomics, 0.78 ms per 10^6 pairs in the three kernels).  A buffer's state is two 128-bit planes S0, S1: bit q of Sp = bit p
// of the 2-bit code of the character sitting in slot q.  Pair t with (clamped) length L and clean-mode planes A0, A1:
//     tails_t = S & ~[0, L)                  what the conversion of pair t sees beyond the string
//     S      <- permute((A & [0, L)) | (S & ~[0, L)))       the copy, then the in-place permutation
// and `permute` is a 16 x 8 bit-matrix transpose (asm_host.h: tail_permute, ~44 integer instructions per plane).  One thread
// owns a whole buffer for a chunk of consecutive pairs; three passes make the chunks independent:
//   pass 1  tails_chunk_kernel : per chunk, its effect on the buffer from an unknown start: W = slots written, C = their codes;
//                                then an exclusive scan of (W, C) over the workgroup's 128 chunks in LDS
//   pass 2  tails_carry_kernel : the same scan over the workgroups' totals (one workgroup; initial buffers from `init`)
//   pass 3  tails_emit_kernel  : replay each chunk from its carry-in, writing tails_t
// SRC has order 10, so with chunks of a multiple of 10 pairs the permutations inside a chunk compose to the identity and two
// chunks' effects compose slot by slot: "the later write wins", (W2, C2) o (W1, C1) = (W1 | W2, C2 where W2 else C1).
// Character codes are read from the clean-mode bit planes (bit q of a plane <-> character q), so the passes touch no
// ASCII.  The pack kernel ORs `tails` into granule 0.  Initial buffer content is pinned to zero (the reference's is
// indeterminate heap memory).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "asm_host.h" /* TAIL_NONE, TailBits, tail_permute, tail_prefix: shared with the host side of the chain */

#define TAIL_CHUNK 40   /* pairs per chunk: a multiple of 10 (the order of SRC); 10^6 pairs = 50 000 threads per pass */
#define TAIL_GROUP 128  /* chunks per workgroup: thread = (side, chunk) */
#define TAIL_AHEAD 4    /* pairs whose lengths and planes are fetched before the first of them is walked */

using asm_host::TailBits;

struct TailOp { /* the effect of a run of pairs on one buffer, in the slot frame of the run's first pair */
    TailBits w, c0, c1;
};

__device__ __forceinline__ TailBits tail_select(TailBits m, TailBits x, TailBits y) { /* x where m, else y */
    TailBits o;
#pragma unroll
    for (int d = 0; d < 4; d++) o.w[d] = asm_host::tail_bfi(m.w[d], x.w[d], y.w[d]);
    return o;
}
__device__ __forceinline__ TailBits tail_from(uint4 v) { return TailBits{{v.x, v.y, v.z, v.w}}; }
__device__ __forceinline__ uint4 tail_to(TailBits b) { return make_uint4(b.w[0], b.w[1], b.w[2], b.w[3]); }
__device__ __forceinline__ TailOp tail_then(const TailOp& first, const TailOp& later) {
    TailOp o;
#pragma unroll
    for (int d = 0; d < 4; d++) o.w.w[d] = first.w.w[d] | later.w.w[d];
    o.c0 = tail_select(later.w, later.c0, first.c0);
    o.c1 = tail_select(later.w, later.c1, first.c1);
    return o;
}

// One pair's step on one buffer: emit what the conversion sees beyond the string, copy the string in, permute.
template <bool TRACK_W, bool EMIT>
__device__ __forceinline__ void tail_step(uint32_t len2, uint4 a0, uint4 a1, int shift, TailBits& s0, TailBits& s1, TailBits& w,
                                          uint4* __restrict__ out0, uint4* __restrict__ out1) {
    uint32_t L = (len2 >> shift) & 0xffffu;
    L = L > 128u ? 128u : L; /* hurdle_matrix.h:626-627 */
    const TailBits m = asm_host::tail_prefix(L);
    if (EMIT) {
        const TailBits zero{{0u, 0u, 0u, 0u}};
        *out0 = tail_to(tail_select(m, zero, s0));
        *out1 = tail_to(tail_select(m, zero, s1));
    }
    s0 = asm_host::tail_permute(tail_select(m, tail_from(a0), s0));
    s1 = asm_host::tail_permute(tail_select(m, tail_from(a1), s1));
    if (TRACK_W) {
        w.w[0] |= m.w[0], w.w[1] |= m.w[1], w.w[2] |= m.w[2], w.w[3] |= m.w[3];
        w = asm_host::tail_permute(w);
    }
}

// The walk of one buffer over pairs [t0, t0 + cnt) (cnt <= TAIL_CHUNK).  TRACK_W: also collect the written slots (pass 1).
// EMIT: write tails_t (pass 3).  TAIL_AHEAD pairs are fetched before the first of them is walked (the fetches do not depend on
// the state); the pairs left over, and the cnt mod 10 permutations that bring pass 1's planes back into the slot frame of the
// chunk's first pair, are only ever the batch's last chunk's.
template <bool TRACK_W, bool EMIT>
__device__ __forceinline__ void tail_walk(const uint4* __restrict__ planes, const uint32_t* __restrict__ lens, long n, int side,
                                          long t0, int cnt, TailBits& s0, TailBits& s1, TailBits& w, uint4* __restrict__ tails) {
    const uint4* __restrict__ p0 = planes + (long)(2 * side) * n + t0;
    const uint4* __restrict__ p1 = planes + (long)(2 * side + 1) * n + t0;
    const uint32_t* __restrict__ pl = lens + t0;
    uint4* __restrict__ o0 = tails + (EMIT ? (long)(2 * side) * n + t0 : 0);
    uint4* __restrict__ o1 = tails + (EMIT ? (long)(2 * side + 1) * n + t0 : 0);
    const int shift = side ? 16 : 0;
    int i = 0;
    for (; i + TAIL_AHEAD <= cnt; i += TAIL_AHEAD) {
        uint32_t ln[TAIL_AHEAD];
        uint4 a0[TAIL_AHEAD], a1[TAIL_AHEAD];
#pragma unroll
        for (int j = 0; j < TAIL_AHEAD; j++) ln[j] = pl[i + j], a0[j] = p0[i + j], a1[j] = p1[i + j];
#pragma unroll
        for (int j = 0; j < TAIL_AHEAD; j++) tail_step<TRACK_W, EMIT>(ln[j], a0[j], a1[j], shift, s0, s1, w, o0 + i + j, o1 + i + j);
    }
    for (; i < cnt; i++) tail_step<TRACK_W, EMIT>(pl[i], p0[i], p1[i], shift, s0, s1, w, o0 + i, o1 + i);
    for (int r = cnt % 10; !EMIT && r != 0 && r < 10; r++) {
        s0 = asm_host::tail_permute(s0), s1 = asm_host::tail_permute(s1);
        if (TRACK_W) w = asm_host::tail_permute(w);
    }
}

// Scratch layout (asm_capi.hip: batch_resolve_tails): loc[nchunks_padded][2] = a chunk's exclusive prefix inside its workgroup,
// grp[ngroups][2] = a workgroup's total, gcarry[ngroups][2][2] = the buffers' planes S0, S1 before a workgroup's first pair.
__global__ __launch_bounds__(2 * TAIL_GROUP) void tails_chunk_kernel(const uint4* __restrict__ planes,
                                                                     const uint32_t* __restrict__ lens, long n,
                                                                     TailOp* __restrict__ loc, TailOp* __restrict__ grp) {
    __shared__ TailOp s_op[2][TAIL_GROUP + 1]; /* [0] = "nothing written", chunk ci at [ci + 1] */
    const int side = threadIdx.x / TAIL_GROUP, ci = threadIdx.x % TAIL_GROUP;
    const long chunk = (long)blockIdx.x * TAIL_GROUP + ci;
    const long t0 = chunk * TAIL_CHUNK;
    const int cnt = t0 >= n ? 0 : (int)((t0 + TAIL_CHUNK < n ? t0 + TAIL_CHUNK : n) - t0);
    TailOp mine{};
    if (ci == 0) s_op[side][0] = mine;
    if (cnt > 0) tail_walk<true, false>(planes, lens, n, side, t0, cnt, mine.c0, mine.c1, mine.w, nullptr);
    /* (the walk started from zero planes: C is zero wherever W is not) */
    s_op[side][ci + 1] = mine;
    __syncthreads();
    for (int off = 1; off < TAIL_GROUP; off <<= 1) { /* inclusive scan over the workgroup's chunks, per side */
        const TailOp prev = s_op[side][ci >= off ? ci + 1 - off : 0];
        __syncthreads();
        mine = tail_then(prev, mine), s_op[side][ci + 1] = mine;
        __syncthreads();
    }
    loc[chunk * 2 + side] = s_op[side][ci];
    if (ci == TAIL_GROUP - 1) grp[(long)blockIdx.x * 2 + side] = mine;
}

// `init` ([256] = [side][slot]): the codes the two buffers hold before the first pair of this batch — zeros (NUL
// bytes) for a batch that starts a file, the state after the previous shard/chunk otherwise (asm_tail_state_advance).
// `summary` (optional, [256]): per trajectory, indexed by its slot before the first pair, the code of the last character the
// batch wrote on it, or TAIL_NONE when the batch never touched it — the batch's whole effect on the buffers.
struct TailState { /* passed by value: no host buffer has to outlive the enqueue */
    uint8_t code[256];
};
// One workgroup: thread = (segment of the workgroup sequence, side, dword of the planes) — the scan is bitwise, so the eight
// (side, dword) columns are independent.  Every thread folds its segment, the segments meet in LDS, then every thread replays
// its segment from the right carry-in.
#define TAIL_CARRY_SEGS 32
__global__ __launch_bounds__(8 * TAIL_CARRY_SEGS) void tails_carry_kernel(const uint32_t* __restrict__ grp /* TailOp[ngroups][2] */,
                                                                          uint32_t* __restrict__ gcarry /* [ngroups][2][2][4] */,
                                                                          long ngroups, TailState init, uint8_t* __restrict__ summary) {
    __shared__ uint32_t s_seg[TAIL_CARRY_SEGS][8][3];
    __shared__ uint32_t s_init[2][2][4], s_tot[2][3][4];
    {   /* the initial buffers as planes: wave v holds slots 64 * (v & 1) .. of side v >> 1 */
        const uint32_t code = init.code[threadIdx.x];
        const unsigned long long b0 = __ballot(code & 1u), b1 = __ballot(code & 2u);
        if ((threadIdx.x & 63) == 0) {
            const int sd = threadIdx.x >> 7, half = (threadIdx.x >> 6) & 1;
            s_init[sd][0][2 * half] = (uint32_t)b0, s_init[sd][0][2 * half + 1] = (uint32_t)(b0 >> 32);
            s_init[sd][1][2 * half] = (uint32_t)b1, s_init[sd][1][2 * half + 1] = (uint32_t)(b1 >> 32);
        }
    }
    const int col = threadIdx.x & 7, seg = threadIdx.x >> 3; /* col = side * 4 + dword */
    const int side = col >> 2, d = col & 3;
    const long per = (ngroups + TAIL_CARRY_SEGS - 1) / TAIL_CARRY_SEGS;
    const long g0 = seg * per, g1 = g0 + per < ngroups ? g0 + per : ngroups;
    uint32_t w = 0u, c0 = 0u, c1 = 0u;
    for (long g = g0; g < g1; g++) {
        const uint32_t* op = grp + (g * 2 + side) * 12;
        const uint32_t gw = op[d];
        w |= gw, c0 = asm_host::tail_bfi(gw, op[4 + d], c0), c1 = asm_host::tail_bfi(gw, op[8 + d], c1);
    }
    s_seg[seg][col][0] = w, s_seg[seg][col][1] = c0, s_seg[seg][col][2] = c1;
    __syncthreads();
    uint32_t cur0 = s_init[side][0][d], cur1 = s_init[side][1][d];
    for (int q = 0; q < seg; q++) {
        const uint32_t gw = s_seg[q][col][0];
        cur0 = asm_host::tail_bfi(gw, s_seg[q][col][1], cur0), cur1 = asm_host::tail_bfi(gw, s_seg[q][col][2], cur1);
    }
    for (long g = g0; g < g1; g++) {
        uint32_t* out = gcarry + (g * 2 + side) * 8;
        out[d] = cur0, out[4 + d] = cur1;
        const uint32_t* op = grp + (g * 2 + side) * 12;
        const uint32_t gw = op[d];
        cur0 = asm_host::tail_bfi(gw, op[4 + d], cur0), cur1 = asm_host::tail_bfi(gw, op[8 + d], cur1);
    }
    if (summary) {
        if (seg == 0) {
            uint32_t tw = 0u, t0 = 0u, t1 = 0u;
            for (int q = 0; q < TAIL_CARRY_SEGS; q++) {
                const uint32_t gw = s_seg[q][col][0];
                tw |= gw, t0 = asm_host::tail_bfi(gw, s_seg[q][col][1], t0), t1 = asm_host::tail_bfi(gw, s_seg[q][col][2], t1);
            }
            s_tot[side][0][d] = tw, s_tot[side][1][d] = t0, s_tot[side][2][d] = t1;
        }
        __syncthreads();
        const int sd = threadIdx.x >> 7, slot = threadIdx.x & 127;
        const uint32_t bit = 1u << (slot & 31);
        const uint32_t code = ((s_tot[sd][1][slot >> 5] & bit) ? 1u : 0u) | ((s_tot[sd][2][slot >> 5] & bit) ? 2u : 0u);
        summary[threadIdx.x] = (uint8_t)((s_tot[sd][0][slot >> 5] & bit) ? code : TAIL_NONE);
    }
}

__global__ __launch_bounds__(2 * TAIL_GROUP) void tails_emit_kernel(const uint4* __restrict__ planes,
                                                                    const uint32_t* __restrict__ lens, long n,
                                                                    const TailOp* __restrict__ loc,
                                                                    const uint4* __restrict__ gcarry /* [ngroups][2][2] */,
                                                                    uint4* __restrict__ tails /* [4][n] */) {
    const int side = threadIdx.x / TAIL_GROUP, ci = threadIdx.x % TAIL_GROUP;
    const long chunk = (long)blockIdx.x * TAIL_GROUP + ci;
    const long t0 = chunk * TAIL_CHUNK;
    if (t0 >= n) return;
    const int cnt = (int)((t0 + TAIL_CHUNK < n ? t0 + TAIL_CHUNK : n) - t0);
    const TailOp before = loc[chunk * 2 + side];
    TailBits s0 = tail_select(before.w, before.c0, tail_from(gcarry[((long)blockIdx.x * 2 + side) * 2]));
    TailBits s1 = tail_select(before.w, before.c1, tail_from(gcarry[((long)blockIdx.x * 2 + side) * 2 + 1]));
    TailBits unused{};
    tail_walk<false, true>(planes, lens, n, side, t0, cnt, s0, s1, unused, tails);
}
