// The persistent, lane-refilling Greedy kernel around the straight-line pass of asm_greedy3.h (narrow band, unit penalties,
// GLOBAL: the benchmark's configuration).  Included after asm_kernels.h (WaveQueue, OutMap, CigarSink).
#pragma once
#include "asm_greedy3.h"
#include "asm_kernels.h"


#define G3_THREADS 512 /* one workgroup per CU: 8 waves = 2 per SIMD; LDS = table + 16 B per lane vector per thread */

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

constexpr size_t g3_lds_bytes(int K, int NT) { return (size_t)G3_TABLE_ENTRIES * 8 + (size_t)(2 * K + 1) * NT * 16; }

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
                                                         OutMap out, CigarSink cig, int refill_min, int park) {
    constexpr int NL = 2 * K + 1;
    extern __shared__ uint4 g3_smem[];
    uint2* const tab = reinterpret_cast<uint2*>(g3_smem);
    G3V* const vecs = reinterpret_cast<G3V*>(g3_smem + G3_TABLE_ENTRIES / 2);
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
    if (threadIdx.x == 0) {
        const long lo = n * (long)blockIdx.x / (long)gridDim.x, hi = n * ((long)blockIdx.x + 1) / (long)gridDim.x;
        g3_next = (unsigned int)lo, g3_end = (unsigned int)hi;
    }
    __syncthreads();
#ifdef GREEDY_DIAG
    unsigned long long dg_refill = 0, dg_step = 0, dg_iters = 0, dg_lanes = 0, dg_t0 = __builtin_amdgcn_s_memtime();
#endif
    // DRAIN COMPACTION (park != 0).  When the workgroup's counter has run dry a wave is left with its last pairs, fewer in every
    // iteration (a pair of seven passes takes seven iterations whoever else is done), and an iteration costs a wave the same
    // issue slots with 3 live lanes as with 64.  So a wave that is down to 64 / (waves per workgroup) live pairs PARKS them —
    // sixteen dwords of state per pair in LDS; the lane vectors stay where they are, in the wave's LDS columns — and ends; the
    // wave that parks last takes all parked pairs of the workgroup (at most 64) into its lanes and runs one shared tail.  No
    // wave ever waits for another: who is last is decided by the counter every wave bumps after its entries are written.
    constexpr int NWAVES = NT / 64, PARK_T = 64 / NWAVES;
    __shared__ unsigned int g3_parked, g3_done;
    __shared__ uint32_t g3_park[64 * 16];
    if (threadIdx.x == 0) g3_parked = 0u, g3_done = 0u;
    __syncthreads();
    bool adopter = false;
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
                out.put(idx, s.cost);
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
        if (park != 0 && !adopter) {
            const bool live = active && !s.finished;
            const unsigned long long live_mask = __ballot(live);
            const bool flushed = __ballot(s.finished && !exhausted) == 0ull; /* every finished pair has been written out */
            if (__ballot(exhausted) != 0ull && flushed && __popcll(live_mask) <= PARK_T) {
                unsigned int base = 0u;
                if ((threadIdx.x & 63) == 0 && live_mask != 0ull) base = atomicAdd(&g3_parked, (unsigned int)__popcll(live_mask));
                base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
                if (live) {
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(live_mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)live_mask, 0u));
                    uint32_t* e = g3_park + 16 * (base + (unsigned int)rank);
#pragma unroll
                    for (int j = 0; j < NL; j++) /* sp, en in -1 .. 128; nsw 0 .. 2K; dst may be below 0 for strings shorter than the band */
                        e[j] = (uint32_t)(s.sp[j] + 1) | ((uint32_t)(s.en[j] + 1) << 8) | ((uint32_t)s.nsw[j] << 16) | ((uint32_t)(s.dst[j] + 16) << 24);
                    e[NL + 0] = (uint32_t)s.m | ((uint32_t)s.n << 8) | ((uint32_t)s.cur_col << 16) | ((uint32_t)(s.cur_lane + 16) << 24);
                    e[NL + 1] = (uint32_t)(s.dest_lane + 256) | ((uint32_t)threadIdx.x << 16);
                    e[NL + 2] = (uint32_t)s.cost;
                    e[NL + 3] = (uint32_t)s.guard;
                    e[NL + 4] = (uint32_t)idx;
                    e[NL + 5] = (uint32_t)ncig;
                }
                __threadfence_block();
                unsigned int order = 0u;
                if ((threadIdx.x & 63) == 0) order = atomicAdd(&g3_done, 1u);
                order = (unsigned int)__builtin_amdgcn_readfirstlane((int)order);
                if (order != (unsigned int)(NWAVES - 1)) return; /* not the last to park: this wave is done */
                __threadfence_block();
                const unsigned int total = *(volatile unsigned int*)&g3_parked; /* <= NWAVES * PARK_T <= 64 */
                adopter = true;
                const int t = (int)(threadIdx.x & 63);
                active = (unsigned int)t < total;
                exhausted = !active; /* an adopted pair still goes through the refill block when it is done: final hop and output */
                s.finished = !active;
                if (active) {
                    const uint32_t* e = g3_park + 16 * t;
#pragma unroll
                    for (int j = 0; j < NL; j++) {
                        const uint32_t w = e[j];
                        s.sp[j] = (int)(w & 255u) - 1, s.en[j] = (int)((w >> 8) & 255u) - 1, s.nsw[j] = (int)((w >> 16) & 255u);
                        s.dst[j] = (int)(w >> 24) - 16;
                    }
                    const uint32_t w0 = e[NL + 0], w1 = e[NL + 1];
                    s.m = (int)(w0 & 255u), s.n = (int)((w0 >> 8) & 255u), s.cur_col = (int)((w0 >> 16) & 255u);
                    s.cur_lane = (int)(w0 >> 24) - 16;
                    s.dest_lane = (int)(w1 & 0xffffu) - 256;
                    const G3V* src = vecs + (w1 >> 16); /* the parking thread's column of lane vectors */
                    s.cost = (int)e[NL + 2], s.guard = (int)e[NL + 3];
                    idx = (long)e[NL + 4];
                    ncig = (int)e[NL + 5];
                    pair = out.index(idx);
#pragma unroll
                    for (int j = 0; j < NL; j++) s.lo[j] = src[j * NT];
                }
                /* every lane has read its source column (one wave, program order) before any lane overwrites a column */
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (active) {
#pragma unroll
                    for (int j = 0; j < NL; j++) {
                        store.put(j, s.lo[j]);
                        s.lf[j] = g3_flip1(s.lo[j]);
                    }
                }
            }
        }
        if (__ballot(active && !s.finished) == 0ull) break; /* wave-uniform: the slice is used up and every pair is done */
        if (active && !s.finished) {
            const G3Step st = g3_pass<K>(s, table, sig, store);
            if (cig.on() && st.committed) cig.step(pair, ncig, st.from_lane, st.to_lane, st.run);
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
