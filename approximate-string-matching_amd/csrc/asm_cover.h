// Coverage counter of the benchmark harness on the device:
//   greedy_coverage += covers(LCM(s1, greedy CIGAR, threshold 1), LCM(s1, NW CIGAR, threshold 3))
// (GASMA/benchmark/benchmark_utils.h:214-225,256-258; long_consecutive_matching_substring and covers,
// GASMA/benchmark/benchmark_coverage.h:26-67,73-91).
//
// The NW side needs an alignment, not just a score.  Unit penalties only (the harness configuration):
//   1. nw_trace_forward_kernel — the banded bit-parallel sweep of nw_band<>, storing every column's vertical delta
//      vectors (VP, VN) in a scratch array laid out [column][pair] (coalesced);
//   2. nw_trace_cover_kernel   — walks back from (m, n): for the column it stands in it rebuilds D0 / HP from the
//      stored vectors of the column to its left, reads the three delta bits it needs, and follows the oracle's
//      documented preference (diagonal, then a gap in the read 'D', then a gap in the reference 'I'; inside a gap, keep
//      extending while the neighbouring delta is +1).  parasail's own preference is internal to that library and
//      absent from the reference tree, so this tie-break is OURS; it is the one the oracle uses.
//      While walking it run-length encodes the CIGAR ('=' 'X' 'I' 'D', emitted back to front) and collects the read
//      positions of '=' runs of length >= 3 (LCM2).  LCM1 comes from the Greedy CIGAR row of the same pair (every 'M'
//      run: Greedy writes matches and mismatches alike as M), and covers() is the leftmost subsequence test on the
//      read's characters at those positions.
// The banded pass answers a pair only when its distance leaves a margin of two diagonals to the band edge (the traceback
// inspects neighbours of the optimal path).  Every other pair — and EVERY pair when (x, o, e) != (1, 1, 1) — goes through
//   3. nw_trace_affine_kernel  — the full Gotoh matrix (any penalties, any distance) with four direction bits per cell
//      (H came from the diagonal / H equals E / E extends / F extends) in a scratch array [row][8 cells][pair], then the
//      oracle's traceback rule by rule: in H prefer the diagonal, then E ('D'), then F ('I'); inside a gap prefer extending.
// so that no pair the harness covers (benchmark_utils.h:214-225,256-258) is left "not determined".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "asm_bits.h"
#include "asm_kernels.h"

#define COVER_YES 1
#define COVER_NO 0
#define COVER_UNKNOWN 2

template <int W>
struct TraceCell; /* (VP, VN) of one column */
template <>
struct TraceCell<32> {
    typedef uint2 T;
    static ASM_DEV T make(uint32_t vp, uint32_t vn) { return make_uint2(vp, vn); }
    static ASM_DEV uint32_t vp(T c) { return c.x; }
    static ASM_DEV uint32_t vn(T c) { return c.y; }
};
template <>
struct TraceCell<64> {
    typedef uint4 T;
    static ASM_DEV T make(u64 vp, u64 vn) {
        return make_uint4((uint32_t)vp, (uint32_t)(vp >> 32), (uint32_t)vn, (uint32_t)(vn >> 32));
    }
    static ASM_DEV u64 vp(T c) { return (u64)c.x | ((u64)c.y << 32); }
    static ASM_DEV u64 vn(T c) { return (u64)c.z | ((u64)c.w << 32); }
};

template <int W>
struct TraceSink {
    static constexpr bool kNeedsColumns = true;
    typename TraceCell<W>::T* trace;
    long n, i;
    template <typename WT>
    ASM_DEV void operator()(int j, WT vp, WT vn) const {
        trace[(long)(j - 1) * n + i] = TraceCell<W>::make(vp, vn);
    }
};

template <int ND, int W>
__global__ __launch_bounds__(ASM_BLOCK) void nw_trace_forward_kernel(const uint4* __restrict__ planes,
                                                                     const uint32_t* __restrict__ lens, long cnt,
                                                                     long n /* plane stride: pairs of the bucket */, int w4,
                                                                     typename TraceCell<W>::T* __restrict__ trace,
                                                                     int32_t* __restrict__ band_result) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
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
    TraceSink<W> sink{trace, cnt, i};
    band_result[i] = nw_band<ND, W, TraceSink<W>>(A0, A1, B0, B1, m, nn, sink);
}

// bit p of a multi-word vector (p >= 0; 0 beyond the vector)
template <int W64>
ASM_DEV uint32_t vw_bit(const VW<W64>& v, int p) {
    u64 word = 0;
#pragma unroll
    for (int q = 0; q < W64; q++)
        if ((p >> 6) == q) word = v.w[q];
    return (uint32_t)(word >> (p & 63)) & 1u;
}

// 64 bits of v starting at bit `off` (zeros beyond the vector)
template <int W64>
ASM_DEV u64 vw_window(const VW<W64>& v, int off) {
    u64 lo = 0, hi = 0;
#pragma unroll
    for (int q = 0; q < W64; q++) {
        if ((off >> 6) == q) lo = v.w[q];
        if ((off >> 6) + 1 == q) hi = v.w[q];
    }
    const int sh = off & 63;
    return (lo >> sh) | (sh ? (hi << (64 - sh)) : 0ull);
}

template <int W64>
ASM_DEV void vw_or_range(VW<W64>& v, int lo, int len) { /* set bits [lo, lo+len) */
#pragma unroll
    for (int q = 0; q < W64; q++) {
        const int a = lo - q * 64, b = lo + len - q * 64;
        const u64 ma = a <= 0 ? ~0ull : (a >= 64 ? 0ull : (~0ull << a));
        const u64 mb = b <= 0 ? 0ull : (b >= 64 ? ~0ull : ((1ull << b) - 1ull));
        v.w[q] |= ma & mb;
    }
}

struct CoverArgs {
    const uint16_t* g_ops; /* Greedy CIGAR rows [n][g_cap], indexed by input pair index */
    const uint8_t* g_nops;
    int g_cap;
    uint8_t* cover;    /* [n] by input pair index: COVER_YES / COVER_NO / COVER_UNKNOWN */
    uint16_t* nw_ops;  /* optional NW CIGAR rows [n][nw_cap], entries in traceback (reverse) order */
    uint8_t* nw_nops;
    int nw_cap;
    unsigned long long* counters; /* [0] += covered, [1] += undetermined */
};

// covers(LCM(read, Greedy CIGAR, 1), LCM(read, NW CIGAR, 3)) given the read positions of the NW side's long '=' runs
// (benchmark_coverage.h:26-67,73-91)
template <int W64>
ASM_DEV int cover_verdict(const VW<W64>& A0, const VW<W64>& A1, int m, const VW<W64>& lcm2, const CoverArgs& ca, long pair) {
    // LCM1: read positions under Greedy's 'M' runs (benchmark_coverage.h:37-62 with threshold 1)
    VW<W64> lcm1;
#pragma unroll
    for (int q = 0; q < W64; q++) lcm1.w[q] = 0ull;
    const int gn = ca.g_nops[pair] < ca.g_cap ? ca.g_nops[pair] : ca.g_cap;
    int ridx = 0;
    for (int t = 0; t < gn; t++) {
        const uint16_t e = ca.g_ops[pair * ca.g_cap + t];
        const int cnt = e >> 3, op = e & 7;
        if (op == 0) {
            const int lim = ridx + cnt > m ? (m > ridx ? m - ridx : 0) : cnt; /* never read past the string */
            vw_or_range<W64>(lcm1, ridx, lim);
            ridx += cnt;
        } else if (op == 1) {
            ridx += cnt;
        }
    }
    int n1 = 0, n2 = 0;
    bool subset = true;
#pragma unroll
    for (int q = 0; q < W64; q++) {
        n1 += __popcll(lcm1.w[q]), n2 += __popcll(lcm2.w[q]);
        subset = subset && ((lcm2.w[q] & ~lcm1.w[q]) == 0ull);
    }
    bool cov;
    if (n1 < n2) {
        cov = false; /* benchmark_coverage.h:78-80 */
    } else if (subset) {
        cov = true; /* the same read positions serve as the embedding */
    } else {
        // leftmost subsequence test (benchmark_coverage.h:81-90) on per-base candidate vectors
        VW<W64> pc[4];
#pragma unroll
        for (int q = 0; q < W64; q++) {
            pc[0].w[q] = lcm1.w[q] & ~A0.w[q] & ~A1.w[q];
            pc[1].w[q] = lcm1.w[q] & A0.w[q] & ~A1.w[q];
            pc[2].w[q] = lcm1.w[q] & ~A0.w[q] & A1.w[q];
            pc[3].w[q] = lcm1.w[q] & A0.w[q] & A1.w[q];
        }
        cov = true;
        int cursor = 0, p2 = vw_next_one<W64>(lcm2, 0);
        while (p2 < W64 * 64) {
            const uint32_t code = vw_bit<W64>(A0, p2) | (vw_bit<W64>(A1, p2) << 1);
            VW<W64> cand;
#pragma unroll
            for (int q = 0; q < W64; q++)
                cand.w[q] = code == 0 ? pc[0].w[q] : code == 1 ? pc[1].w[q] : code == 2 ? pc[2].w[q] : pc[3].w[q];
            const int p1 = vw_next_one<W64>(cand, cursor);
            if (p1 >= W64 * 64) {
                cov = false;
                break;
            }
            cursor = p1 + 1;
            p2 = vw_next_one<W64>(lcm2, p2 + 1);
        }
    }
    return cov ? COVER_YES : COVER_NO;
}

template <int W64, int W>
__global__ __launch_bounds__(ASM_BLOCK) void nw_trace_cover_kernel(const uint4* __restrict__ planes,
                                                                   const uint32_t* __restrict__ lens, long cnt,
                                                                   long n /* plane stride */, int w4,
                                                                   const typename TraceCell<W>::T* __restrict__ trace,
                                                                   const int32_t* __restrict__ band_result,
                                                                   const uint32_t* __restrict__ order, long slice_lo,
                                                                   CoverArgs ca, uint32_t* __restrict__ todo /* slice slots the band
                                                                   could not answer, for the full-matrix pass; null: flag them 2 */,
                                                                   uint32_t* __restrict__ todo_count) {
    typedef typename BandWord<W>::T WT;
    constexpr int C = W / 2;
    __shared__ unsigned int s_part[4];
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned int covered = 0, unknown = 0;
    if (i < cnt) {
        const long pair = order ? (long)order[i] : slice_lo + i;
        const uint32_t ln = lens[i];
        const int m = (int)(ln & 0xffffu), nn = (int)(ln >> 16);
        VW<W64> A0, A1, B0, B1;
        load_planes<W64>(planes, n, w4, i, A0, A1, B0, B1);
        const int r = band_result[i];
        bool ok = r >= 0 && r <= C - 3;
        VW<W64> lcm2;
#pragma unroll
        for (int q = 0; q < W64; q++) lcm2.w[q] = 0ull;
        int nops = 0, cur_op = -1, cur_cnt = 0;
        // ---------------- traceback ----------------
        if (ok) {
            int ii = m, j = nn, state = 0; /* 0 = H, 1 = E ('D'), 2 = F ('I') */
            int curcol = 0, top = 1;
            WT D0 = 0, HP = 0, VPc = 0;
            auto flush = [&]() {
                if (cur_cnt > 0) {
                    if (cur_op == 3 && cur_cnt >= 3) vw_or_range<W64>(lcm2, ii, cur_cnt); /* '=' run: read [ii, ii+cnt) */
                    if (ca.nw_ops && nops < ca.nw_cap) ca.nw_ops[pair * ca.nw_cap + nops] = (uint16_t)((cur_cnt << 3) | cur_op);
                    nops++;
                }
                cur_cnt = 0;
            };
            auto emit = [&](int op) {
                if (op != cur_op) {
                    flush();
                    cur_op = op;
                }
                cur_cnt++;
            };
            // rebuild D0 / HP of column `col` from the stored vectors of column col-1, exactly as nw_band<> computed them
            auto rebuild = [&](int col) {
                WT vp_prev = ~(WT)0, vn_prev = 0;
                if (col >= 2) {
                    const typename TraceCell<W>::T c = trace[(long)(col - 2) * cnt + i];
                    vp_prev = (WT)TraceCell<W>::vp(c), vn_prev = (WT)TraceCell<W>::vn(c);
                }
                const typename TraceCell<W>::T cc = trace[(long)(col - 1) * cnt + i];
                VPc = (WT)TraceCell<W>::vp(cc);
                const bool slide = col > C;
                const WT vpin = slide ? (WT)((vp_prev >> 1) | ((WT)1 << (W - 1))) : vp_prev;
                const WT vnin = slide ? (WT)(vn_prev >> 1) : vn_prev;
                top = col > C - 1 ? col - (C - 1) : 1;
                const WT a0w = (WT)vw_window<W64>(A0, top - 1), a1w = (WT)vw_window<W64>(A1, top - 1);
                const WT T0 = (WT)0 - (WT)vw_bit<W64>(B0, col - 1), T1 = (WT)0 - (WT)vw_bit<W64>(B1, col - 1);
                const WT Eq = ~((a0w ^ T0) | (a1w ^ T1));
                D0 = ((((Eq & vpin) + vpin) ^ vpin) | Eq) | vnin;
                HP = vnin | ~(D0 | vpin);
                curcol = col;
            };
            for (int guard = 0; guard < 2 * ASM_MAX_LENGTH + 4 && (ii > 0 || j > 0) && ok; guard++) {
                if (j > 0 && curcol != j) rebuild(j);
                const int b = ii - top; /* bit of row ii in the current column's window */
                if (state == 0) {
                    if (ii > 0 && j > 0) {
                        if (b < 0 || b > W - 1) {
                            ok = false;
                            break;
                        }
                        const bool match = ((vw_bit<W64>(A0, ii - 1) ^ vw_bit<W64>(B0, j - 1)) |
                                            (vw_bit<W64>(A1, ii - 1) ^ vw_bit<W64>(B1, j - 1))) == 0u;
                        const bool d0 = (D0 >> b) & 1, hp = (HP >> b) & 1;
                        if (match || !d0) {
                            emit(match ? 3 : 4);
                            ii--, j--;
                        } else if (hp) {
                            state = 1;
                        } else {
                            state = 2;
                        }
                    } else {
                        state = ii == 0 ? 1 : 2; /* first row: only gaps in the read; first column: only gaps in the ref */
                    }
                } else if (state == 1) {
                    emit(2); /* 'D' */
                    j--;
                    // keep extending while the horizontal delta of the cell now under the cursor is +1
                    bool cont = j >= 1;
                    if (cont && ii > 0) {
                        rebuild(j); /* HP of the column now under the cursor */
                        const int b2 = ii - top;
                        if (b2 < 0 || b2 > W - 1) {
                            ok = false;
                            break;
                        }
                        cont = (HP >> b2) & 1;
                    }
                    if (!cont) state = 0;
                } else {
                    emit(1); /* 'I' */
                    // keep extending while the vertical delta of row ii-1 in this column is +1
                    bool cont = ii > 1;
                    if (cont && j > 0) {
                        const int b2 = ii - 1 - top;
                        if (b2 < 0 || b2 > W - 1) {
                            ok = false;
                            break;
                        }
                        cont = (VPc >> b2) & 1;
                    }
                    ii--;
                    if (!cont) state = 0;
                }
            }
            if (ii > 0 || j > 0) ok = false;
            flush();
        }
        // ---------------- coverage ----------------
        int flag = COVER_UNKNOWN;
        if (ok) flag = cover_verdict<W64>(A0, A1, m, lcm2, ca, pair);
        ca.cover[pair] = (uint8_t)flag;
        if (ca.nw_nops) ca.nw_nops[pair] = (uint8_t)(ok ? (nops > 255 ? 255 : nops) : 0);
        covered = flag == COVER_YES, unknown = flag == COVER_UNKNOWN;
    }
    if (todo != nullptr) { /* wave-aggregated append: one atomic per wave */
        const u64 bal = __ballot(unknown != 0u);
        if (bal) {
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            const int leader = __builtin_ctzll(bal);
            uint32_t base = 0;
            if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(todo_count, (uint32_t)__builtin_popcountll(bal));
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
            if (unknown) todo[base + rank] = (uint32_t)i;
        }
        unknown = 0; /* the full-matrix pass answers them */
    }
    const unsigned int tc = block_sum_256(covered, s_part), tu = block_sum_256(unknown, s_part);
    if (threadIdx.x == 0) {
        if (tc) atomicAdd(&ca.counters[0], (unsigned long long)tc);
        if (tu) atomicAdd(&ca.counters[1], (unsigned long long)tu);
    }
}

// --------------------------------------------------------------------------------------------------------
// Full-matrix Gotoh with traceback and coverage verdict, any (x, o, e), any distance.  One thread per pair; the forward sweep is
// nw_affine_kernel's (column blocks of 32 in registers, block boundary column through LDS) and additionally records, per
// cell, the four facts the oracle's traceback asks for (oracle/asm_oracle.c, orc_nw_cigar_batch):
//   bit 0  H == H[i-1][j-1] + sub      bit 1  H == E      bit 2  E == E[i][j-1] + e      bit 3  F == F[i-1][j] + e
// packed 8 cells per dword in scratch[((i-1) * cols8 + (j-1)/8) * stride + slot] (coalesced across the pairs of a wave).
// --------------------------------------------------------------------------------------------------------
template <int W64, int MAXROWS>
__global__ __launch_bounds__(64) void nw_trace_affine_kernel(const uint4* __restrict__ planes, const uint32_t* __restrict__ lens,
                                                             long cnt /* slots of this launch */, long n /* plane stride */, int w4,
                                                             int x, int o, int e, uint32_t* __restrict__ scratch, int cols8,
                                                             const uint32_t* __restrict__ todo /* cnt bucket slots, or null: slots 0..cnt-1 */,
                                                             const uint32_t* __restrict__ order, long slice_lo, CoverArgs ca) {
    __shared__ uint32_t s_bound[MAXROWS + 1][64];
    __shared__ unsigned int s_cov[1];
    const int t = threadIdx.x;
    const long slot = (long)blockIdx.x * 64 + t;
    const long stride = cnt;
    if (t == 0) s_cov[0] = 0u;
    __syncthreads();
    unsigned int covered = 0u;
    if (slot < stride) {
        const long i = todo ? (long)todo[slot] : slot;
        const long pair = order ? (long)order[i] : slice_lo + i;
        const uint32_t ln = lens[i];
        const int m = (int)(ln & 0xffffu), nn = (int)(ln >> 16);
        VW<W64> A0, A1, B0, B1;
        load_planes<W64>(planes, n, w4, i, A0, A1, B0, B1);
        // ---------------- forward ----------------
        for (int j0 = 0; j0 < nn; j0 += NW_CB) {
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
                uint32_t bits[NW_CB / 8];
#pragma unroll
                for (int q = 0; q < NW_CB / 8; q++) bits[q] = 0u;
#pragma unroll
                for (int jj = 0; jj < NW_CB; jj++) {
                    const int up = H[jj];
                    const int fext = F[jj] + e;
                    int f = up + o < fext ? up + o : fext;
                    const int eext = eleft + e;
                    int ee = hleft + o < eext ? hleft + o : eext;
                    const int dg = diag + (((mm >> jj) & 1u) ? x : 0);
                    int h = dg < f ? dg : f;
                    h = ee < h ? ee : h;
                    const uint32_t nib = (h == dg ? 1u : 0u) | (h == ee ? 2u : 0u) | (ee == eext ? 4u : 0u) | (f == fext ? 8u : 0u);
                    bits[jj >> 3] |= nib << (4 * (jj & 7));
                    f = f > NW_BIG ? NW_BIG : f;
                    ee = ee > NW_BIG ? NW_BIG : ee;
                    diag = up;
                    H[jj] = h, F[jj] = f;
                    hleft = h, eleft = ee;
                }
                diag = next_diag;
                s_bound[r][t] = (uint32_t)hleft | ((uint32_t)eleft << 16);
#pragma unroll
                for (int q = 0; q < NW_CB / 8; q++)
                    if (j0 + 8 * q < nn) scratch[((long)(r - 1) * cols8 + (j0 >> 3) + q) * stride + slot] = bits[q];
            }
        }
        // ---------------- traceback (oracle/asm_oracle.c orc_nw_cigar_batch, rule by rule) ----------------
        VW<W64> lcm2;
#pragma unroll
        for (int q = 0; q < W64; q++) lcm2.w[q] = 0ull;
        int nops = 0, cur_op = -1, cur_cnt = 0;
        int ii = m, j = nn, state = 0; /* 0 = H, 1 = E ('D'), 2 = F ('I') */
        auto flush = [&]() {
            if (cur_cnt > 0) {
                if (cur_op == 3 && cur_cnt >= 3) vw_or_range<W64>(lcm2, ii, cur_cnt); /* '=' run: read [ii, ii+cnt) */
                if (ca.nw_ops && nops < ca.nw_cap) ca.nw_ops[pair * ca.nw_cap + nops] = (uint16_t)((cur_cnt << 3) | cur_op);
                nops++;
            }
            cur_cnt = 0;
        };
        auto emit = [&](int op) {
            if (op != cur_op) {
                flush();
                cur_op = op;
            }
            cur_cnt++;
        };
        for (int guard = 0; guard < 2 * ASM_MAX_LENGTH + 4 && (ii > 0 || j > 0); guard++) {
            uint32_t nib = 0u;
            if (ii > 0 && j > 0) nib = (scratch[((long)(ii - 1) * cols8 + ((j - 1) >> 3)) * stride + slot] >> (4 * ((j - 1) & 7))) & 15u;
            if (state == 0) {
                if (ii > 0 && j > 0 && (nib & 1u)) {
                    const bool match = ((vw_bit<W64>(A0, ii - 1) ^ vw_bit<W64>(B0, j - 1)) |
                                        (vw_bit<W64>(A1, ii - 1) ^ vw_bit<W64>(B1, j - 1))) == 0u;
                    emit(match ? 3 : 4);
                    ii--, j--;
                } else if (j > 0 && (ii == 0 || (nib & 2u))) {
                    state = 1; /* H == E; on row 0 H is E by construction */
                } else {
                    state = 2;
                }
            } else if (state == 1) {
                emit(2); /* 'D' */
                /* E[i][j] == E[i][j-1] + e, j > 1: on row 0 E[0][j] = o + (j-1) e, so always */
                const bool ext = j > 1 && (ii == 0 || (nib & 4u));
                if (!ext) state = 0;
                j--;
            } else {
                emit(1); /* 'I' */
                const bool ext = ii > 1 && (j == 0 || (nib & 8u));
                if (!ext) state = 0;
                ii--;
            }
        }
        flush();
        const int flag = cover_verdict<W64>(A0, A1, m, lcm2, ca, pair);
        ca.cover[pair] = (uint8_t)flag;
        if (ca.nw_nops) ca.nw_nops[pair] = (uint8_t)(nops > 255 ? 255 : nops);
        covered = flag == COVER_YES;
    }
    if (covered) atomicAdd(&s_cov[0], 1u);
    __syncthreads();
    if (t == 0 && s_cov[0]) atomicAdd(&ca.counters[0], (unsigned long long)s_cov[0]);
}
