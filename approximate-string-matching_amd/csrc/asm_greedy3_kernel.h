// The persistent, lane-refilling Greedy kernel around the straight-line pass of asm_greedy3.h (narrow band, unit penalties,
// GLOBAL: the benchmark's configuration).  Included after asm_kernels.h (WaveQueue, OutMap, CigarSink).
#pragma once
#include "asm_greedy3.h"
#include "asm_kernels.h"


#ifndef G3_THREADS
#define G3_THREADS 512
#endif
/* G3_THREADS: one workgroup per CU: 8 waves = 2 per SIMD; LDS = table + 16 B per lane vector per thread */

template <int NT>
struct G3LdsStore {
    G3V* base; /* this thread's column of the [lane][thread] array */
    __device__ __forceinline__ void put(int j, G3V v) { base[j * NT] = v; }
    __device__ __forceinline__ G3V get(int j) const { return base[j * NT]; }
};
struct G3LdsTable {
    const uint2* t;
    __device__ __forceinline__ uint2 get(uint32_t i) const { return t[i]; }
};

/* costs of the workgroup's slice staged in LDS and written out in one coalesced sweep at the end (when the slice fits and the
 * results go to their own index): a refilling lane's scattered 4-byte store — 2.5 x write amplification at C2, 64-bit address
 * arithmetic inside the half-empty refill block — becomes one ds_write_b32 */
#ifndef G3_STAGE_ENTRIES
#define G3_STAGE_ENTRIES 4096
#endif
typedef uint16_t g3_stage_t; /* 0xFFFF: "this cost did not fit and went straight to memory" (never at 128 characters in practice) */
constexpr size_t g3_lds_bytes(int K, int NT) {
    return (size_t)G3_TABLE_ENTRIES * 8 + (size_t)(2 * K + 1) * NT * 16 + (size_t)G3_STAGE_ENTRIES * sizeof(g3_stage_t);
}

__device__ __forceinline__ G3V g3_from_uint4(uint4 q) {
    G3V r;
    r.lo = (g3_u64)q.x | ((g3_u64)q.y << 32), r.hi = (g3_u64)q.z | ((g3_u64)q.w << 32);
    return r;
}

// Same contract as greedy_persist_kernel<K, true> (asm_kernels.h): wave-local static slice of the batch, lanes pull the next
// pair when theirs is done (WaveQueue), one int32 cost per pair through OutMap, optional CIGAR rows.
template <int K, int NT>
__global__ __launch_bounds__(NT) void greedy_fast_kernel(const uint4* __restrict__ planes, const uint32_t* __restrict__ lens,
                                                         long n, int w4, G3Sig sig, const uint2* __restrict__ table_g,
                                                         OutMap out, CigarSink cig, int refill_min) {
    constexpr int NL = 2 * K + 1;
    extern __shared__ uint4 g3_smem[];
    uint2* const tab = reinterpret_cast<uint2*>(g3_smem);
    G3V* const vecs = reinterpret_cast<G3V*>(g3_smem + G3_TABLE_ENTRIES / 2);
    g3_stage_t* const stage = reinterpret_cast<g3_stage_t*>(vecs + (size_t)NL * NT);
    for (int i = threadIdx.x; i < G3_TABLE_ENTRIES / 2; i += NT) g3_smem[i] = reinterpret_cast<const uint4*>(table_g)[i];
    __syncthreads();
    const G3LdsTable table{tab};
    G3LdsStore<NT> store{vecs + threadIdx.x};

    G3State<K> s;
#pragma unroll
    for (int j = 0; j < NL; j++) {
        s.lo[j].lo = s.lo[j].hi = s.lf[j].lo = s.lf[j].hi = 0ull;
        s.sp[j] = -1, s.en[j] = -1, s.nsw[j] = 0, s.dst[j] = 0;
    }
    s.m = s.n = s.dest_lane = s.cur_lane = s.cur_col = s.cost = s.guard = 0;
    s.finished = true;
    long idx = -1, pair = 0;
    int ncig = 0;
    bool active = false, exhausted = false;
    // The workgroup owns a contiguous slice of the batch and ALL its waves draw pairs from one counter in LDS.  (With a static
    // slice per wave the waves of a SIMD finish one after the other — the arbiter serves the oldest first, and this kernel is a
    // dense stream of 4-cycle vector operations, so the younger wave gets about half the older one's issue rate: 144 k against
    // 217 k cycles at two waves per SIMD — and the kernel lasts as long as the slower wave.  A shared counter lets the faster
    // wave take more pairs; one LDS atomic per refill, every ~7 k cycles.)
    // (Measured and dropped: handing the last 8-25 % of the batch out dynamically, in chunks of 128 pairs from a counter in device
    // memory, so that workgroups which are ahead take more — 0.145 ms against 0.113: the atomics' round trips stall the refills.)
    __shared__ unsigned int g3_next, g3_end;
    const long slice_lo = n * (long)blockIdx.x / (long)gridDim.x, slice_hi = n * ((long)blockIdx.x + 1) / (long)gridDim.x;
    const bool staged = out.order == nullptr && slice_hi - slice_lo <= (long)G3_STAGE_ENTRIES; /* workgroup-uniform */
    if (threadIdx.x == 0) g3_next = (unsigned int)slice_lo, g3_end = (unsigned int)slice_hi;
    __syncthreads();
#ifdef GREEDY_DIAG
    unsigned long long dg_refill = 0, dg_step = 0, dg_iters = 0, dg_lanes = 0, dg_t0 = __builtin_amdgcn_s_memtime();
#endif
    // (Measured and removed in round 4: DRAIN COMPACTION — a wave down to 64 / (waves per workgroup) live pairs parked them in LDS
    // and ended, the wave that parked last adopted all of them.  Fewer wave-iterations, bit-identical, and slower in every form
    // of the step: 125 us against 112 stand-alone, 0.234 against 0.230 ms per overlapped step.  Last present in commit 312851a.)
    for (;;) {
#ifdef GREEDY_DIAG
        const unsigned long long dg_a = __builtin_amdgcn_s_memtime();
#endif
        const bool need = s.finished && !exhausted;
        const unsigned long long need_mask = __ballot(need);
        const bool dry = __ballot(exhausted) != 0ull; /* some lane of this wave found the counter used up: no more pairs to wait for */
        if (need_mask != 0ull && (__popcll(need_mask) >= refill_min || dry || __ballot(active && !s.finished) == 0ull)) {
            if (need && active) {
                // ---- final hop (hurdle_matrix.h:575-590) ----
                const int dest_col = g3_dest(s.m, s.n, s.dest_lane);
                if (s.cur_lane != s.dest_lane || s.cur_col < dest_col) {
                    G3V dv;
                    if (s.dest_lane >= -K && s.dest_lane <= K) {
                        dv = store.get(s.dest_lane + K);
                    } else { /* destination lane outside the band: undefined in the reference, built like a band lane */
                        const G3V A0 = g3_from_uint4(planes[((long)0 * w4) * n + idx]), A1 = g3_from_uint4(planes[((long)1 * w4) * n + idx]);
                        const G3V B0 = g3_from_uint4(planes[((long)2 * w4) * n + idx]), B1 = g3_from_uint4(planes[((long)3 * w4) * n + idx]);
                        const int a = s.dest_lane < 0 ? -s.dest_lane : s.dest_lane;
                        if (s.dest_lane < 0) {
                            const G3V x0 = g3_toward0(A0, a), x1 = g3_toward0(A1, a);
                            dv.lo = (x0.lo ^ B0.lo) | (x1.lo ^ B1.lo), dv.hi = (x0.hi ^ B0.hi) | (x1.hi ^ B1.hi);
                        } else {
                            const G3V x0 = g3_toward0(B0, a), x1 = g3_toward0(B1, a);
                            dv.lo = (x0.lo ^ A0.lo) | (x1.lo ^ A1.lo), dv.hi = (x0.hi ^ A0.hi) | (x1.hi ^ A1.hi);
                        }
                    }
                    const int d = s.cur_lane - s.dest_lane;
                    const int from = s.cur_col + g3_fwd(s.cur_lane, s.dest_lane);
                    const bool ok = (unsigned)from < 128u && (unsigned)(dest_col - from - 1) < 128u; /* utils.h:263-270 */
                    const int distance = ok ? g3_ones_from(dv, (uint32_t)from) - g3_ones_from(dv, (uint32_t)dest_col) : 0;
                    s.cost += (d < 0 ? -d : d) + distance;
                    if (cig.on()) cig.step(pair, ncig, s.cur_lane, s.dest_lane, distance); /* the hurdle count (:589) */
                }
                if (cig.on()) cig.finish(pair, ncig);
                if (staged) stage[idx - slice_lo] = (g3_stage_t)(s.cost < 0xFFFF ? s.cost : 0xFFFF);
                if (!staged || s.cost >= 0xFFFF) out.put(idx, s.cost);
            }
            long got = -1;
            { /* consecutive pairs for the lanes that need one: rank inside the wave + the workgroup's counter */
                const unsigned long long mask = __ballot(need);
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                unsigned int base = 0u;
                if ((threadIdx.x & 63) == 0) base = atomicAdd(&g3_next, (unsigned int)__popcll(mask));
                base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
                const unsigned int mine = base + (unsigned int)rank;
                if (need && mine < g3_end) got = (long)mine;
            }
            if (need) {
                idx = got;
                active = got >= 0;
                exhausted = !active;
            }
            if (need && active) {
                const G3V A0 = g3_from_uint4(planes[((long)0 * w4) * n + idx]), A1 = g3_from_uint4(planes[((long)1 * w4) * n + idx]);
                const G3V B0 = g3_from_uint4(planes[((long)2 * w4) * n + idx]), B1 = g3_from_uint4(planes[((long)3 * w4) * n + idx]);
                g3_setup<K>(s, A0, A1, B0, B1, lens[idx], store);
                ncig = 0;
                pair = out.index(idx);
            }
        }
#ifdef GREEDY_DIAG
        const unsigned long long dg_b = __builtin_amdgcn_s_memtime();
        dg_refill += dg_b - dg_a;
        dg_iters++;
        dg_lanes += __popcll(__ballot(active && !s.finished));
#endif
        if (__ballot(active && !s.finished) == 0ull) break; /* wave-uniform: the slice is used up and every pair is done */
        if (active && !s.finished) {
            const G3Step st = g3_pass<K>(s, table, sig, store);
            if (cig.on() && st.committed) cig.step(pair, ncig, st.from_lane, st.to_lane, st.run);
        }
#ifdef GREEDY_DIAG
        dg_step += __builtin_amdgcn_s_memtime() - dg_b;
#endif
    }
    if (staged) { /* every wave leaves the loop through its break: the whole workgroup meets here */
        __syncthreads();
        for (long q = threadIdx.x; q < slice_hi - slice_lo; q += NT) if (stage[q] != 0xFFFFu) out.out[slice_lo + q] = (int32_t)stage[q];
    }
#ifdef GREEDY_DIAG
    if ((threadIdx.x & 63) == 0 && cig.nops != nullptr && cig.ops == nullptr) { /* diag build: cig.nops doubles as the debug buffer */
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(cig.nops) + 8 * (((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
        dbg[0] = dg_refill, dbg[1] = dg_step, dbg[2] = dg_iters, dbg[3] = dg_lanes, dbg[4] = __builtin_amdgcn_s_memtime() - dg_t0;
    }
#endif
}
