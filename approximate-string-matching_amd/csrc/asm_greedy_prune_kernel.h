// Device side of the pruned wide-band Greedy (asm_greedy_prune.h has the algorithm and why it is exact):
//   prune_setup_kernel   one WAVE per pair, thread = band lane: the lane's vector, its constants (zl, longest zero run), the
//                        running maxima of the runs towards both band edges (wave prefix scans) and the pair's minimum zl
//                        -> 64 dwords + 1 byte per pair in HBM
//   greedy_prune_kernel  one THREAD per pair, persistent with lane refill (WaveQueue, as the narrow-band kernels): a pass
//                        evaluates the handful of lanes pr_pass finds worth looking at, rebuilding their vectors from the
//                        pair's bit planes (registers); per-lane constants, pass history and a small evaluation cache sit in
//                        thread-private LDS columns.  Pairs that outrun the history go to a list for the wave-per-pair kernel.
// Included after asm_kernels.h (WaveQueue, OutMap, CigarSink).
#pragma once
#include "asm_greedy_prune.h"
#include "asm_greedy3_kernel.h"
#include "asm_kernels.h"

#include "asm_wave.h" /* wave_max_u32 */

__global__ __launch_bounds__(ASM_BLOCK) void prune_setup_kernel(const uint4* __restrict__ planes, const uint32_t* __restrict__ lens,
                                                                long n, int w4, int K, signed char* __restrict__ zls /* [n][64] */,
                                                                PrPairInfo* __restrict__ pinfo) {
    const int t = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const bool active = t < 2 * K + 1;
    const int lane = active ? t - K : K;
    for (long i = wave; i < n; i += nwaves) {
        const G3V A0 = g3_from_uint4(planes[((long)0 * w4) * n + i]), A1 = g3_from_uint4(planes[((long)1 * w4) * n + i]);
        const G3V B0 = g3_from_uint4(planes[((long)2 * w4) * n + i]), B1 = g3_from_uint4(planes[((long)3 * w4) * n + i]);
        const uint32_t ln = lens[i];
        int m = (int)(ln & 0xffffu), nn = (int)(ln >> 16);
        m = m > 128 ? 128 : m, nn = nn > 128 ? 128 : nn;
        G3V lo, lf;
        pr_lane_vectors(A0, A1, B0, B1, lane, lo, lf);
        const PrLaneInfo f = pr_lane_info(lf, g3_dest(m, nn, lane));
        PrPairInfo pi;
#pragma unroll
        for (int c = 1; c < PR_RUN_CLASSES; c++) pi.runs[c - 1] = __ballot(active && f.run >= c);
        // the four lanes with the smallest zl (ties: the lower lane) and the fifth smallest value: five wave minima of zl << 8 | t
        unsigned key = active ? (((unsigned)(f.zl + 1) << 8) | (unsigned)t) : 0xffffu;
#pragma unroll
        for (int q = 0; q < 5; q++) {
            const unsigned mn = 0xffffu - wave_max_u32(0xffffu - key);
            const int zl = mn >= 0xff00u ? 127 : (int)(mn >> 8) - 1;
            if (q < 4) pi.zl_lane[q] = (unsigned char)(mn >= 0xff00u ? 127 : (mn & 255u)), pi.zl_val[q] = (signed char)zl;
            else pi.zl_next = zl;
            if (key == mn) key = 0xffffu;
        }
        pi.pad_ = 0;
        zls[i * 64 + t] = (signed char)(active ? f.zl : 127);
        if (t == 0) pinfo[i] = pi;
    }
}

// Thread-private columns in LDS: [entry][thread] dwords; entries = 16 (64 lane zl bytes), PR_HIST / 2 (history, 16 bits per
// pass), 8 cache slots
#define PR_LDS_DWORDS (16 + PR_HIST / 2 + 8)
template <int NT>
struct PrDeviceLanes {
    G3V A0, A1, B0, B1;
    uint32_t* col; /* this thread's column */
    int K;
    __device__ __forceinline__ void get(int lane, G3V& lo, G3V& lf) const { pr_lane_vectors(A0, A1, B0, B1, lane, lo, lf); }
    __device__ __forceinline__ int zl(int lane) const {
        const int t = lane + K;
        return (int)(signed char)((col[(t >> 2) * NT] >> (8 * (t & 3))) & 255u);
    }
    __device__ __forceinline__ uint32_t hist(int q) const {
        const uint32_t w = col[(16 + (q >> 1)) * NT];
        return (q & 1) ? (w >> 16) : (w & 0xffffu);
    }
    __device__ __forceinline__ void set_hist(int q, uint32_t v) {
        uint32_t* p = &col[(16 + (q >> 1)) * NT];
        const uint32_t w = *p;
        *p = (q & 1) ? ((w & 0xffffu) | (v << 16)) : ((w & 0xffff0000u) | (v & 0xffffu));
    }
    __device__ __forceinline__ uint32_t cache_get(int slot) const { return col[(16 + PR_HIST / 2 + slot) * NT]; }
    __device__ __forceinline__ void cache_put(int slot, uint32_t v) { col[(16 + PR_HIST / 2 + slot) * NT] = v; }
};

constexpr size_t pr_lds_bytes(int NT) { return (size_t)NT * PR_LDS_DWORDS * 4; }

template <int NT>
__global__ __launch_bounds__(NT) void greedy_prune_kernel(const uint4* __restrict__ planes, const uint32_t* __restrict__ lens, long n,
                                                          int w4, int K, G3Sig sig, const signed char* __restrict__ zls,
                                                          const PrPairInfo* __restrict__ pinfo, OutMap out, CigarSink cig,
                                                          int refill_min, uint32_t* __restrict__ todo, uint32_t* __restrict__ todo_count) {
    extern __shared__ uint32_t pr_smem[];
    PrDeviceLanes<NT> L;
    L.col = pr_smem + threadIdx.x;
    L.K = K;
    L.A0.lo = L.A0.hi = L.A1.lo = L.A1.hi = L.B0.lo = L.B0.hi = L.B1.lo = L.B1.hi = 0ull;
    PrPair s;
    s.K = K, s.m = s.n = s.dest_lane = s.zlmin = s.np = s.cl = s.cc = s.cmin = s.cmax = s.cost = 0;
    s.finished = true, s.overflow = false;
    long idx = -1, pair = 0;
    int ncig = 0;
    bool active = false, exhausted = false;
    WaveQueue wq;
    wq.init(n);
    for (;;) {
        const bool need = s.finished && !exhausted;
        const unsigned long long need_mask = __ballot(need);
        if (need_mask != 0ull && (__popcll(need_mask) >= refill_min || __ballot(active && !s.finished) == 0ull)) {
            if (need && active) {
                if (s.overflow) { /* more passes than the history holds: the wave-per-pair kernel does this pair */
                    todo[atomicAdd(todo_count, 1u)] = (uint32_t)idx;
                } else {
                    // ---- final hop (hurdle_matrix.h:575-590) ----
                    const int dest_col = g3_dest(s.m, s.n, s.dest_lane);
                    if (s.cl != s.dest_lane || s.cc < dest_col) {
                        G3V dv, dflip;
                        pr_lane_vectors(L.A0, L.A1, L.B0, L.B1, s.dest_lane, dv, dflip);
                        const int d = s.cl - s.dest_lane;
                        const int from = s.cc + g3_fwd(s.cl, s.dest_lane);
                        const bool ok = (unsigned)from < 128u && (unsigned)(dest_col - from - 1) < 128u; /* utils.h:263-270 */
                        const int distance = ok ? g3_ones_from(dv, (uint32_t)from) - g3_ones_from(dv, (uint32_t)dest_col) : 0;
                        s.cost += (d < 0 ? -d : d) + distance;
                        if (cig.on()) cig.step(pair, ncig, s.cl, s.dest_lane, distance); /* the hurdle count (:589) */
                    }
                    if (cig.on()) cig.finish(pair, ncig);
                    out.put(idx, s.cost);
                }
            }
            const long got = wq.pull(need);
            if (need) {
                idx = got;
                active = got >= 0;
                exhausted = !active;
            }
            if (need && active) {
                L.A0 = g3_from_uint4(planes[((long)0 * w4) * n + idx]), L.A1 = g3_from_uint4(planes[((long)1 * w4) * n + idx]);
                L.B0 = g3_from_uint4(planes[((long)2 * w4) * n + idx]), L.B1 = g3_from_uint4(planes[((long)3 * w4) * n + idx]);
                const uint4* row = reinterpret_cast<const uint4*>(zls + idx * 64);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint4 v = row[q];
                    L.col[(4 * q + 0) * NT] = v.x, L.col[(4 * q + 1) * NT] = v.y, L.col[(4 * q + 2) * NT] = v.z, L.col[(4 * q + 3) * NT] = v.w;
                }
                pr_begin(s, K, lens[idx], pinfo[idx], L);
                ncig = 0;
                pair = out.index(idx);
            }
        }
        if (__ballot(active && !s.finished) == 0ull) break; /* wave-uniform: the slice is used up and every pair is done */
        if (active && !s.finished) {
            const PrStep st = pr_pass(s, sig, L, (PrStats*)nullptr);
            if (cig.on() && st.committed) cig.step(pair, ncig, st.from_lane, st.to_lane, st.run);
        }
    }
}
