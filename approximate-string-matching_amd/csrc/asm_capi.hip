// C ABI of libasm_mi355x.so (declared in include/asm_mi355x.h): handle/stream management, device-resident
// batches, kernel dispatch.  Host side is C++17 compiled by hipcc; nothing here computes alignments on the
// CPU — without a HIP device every compute entry point returns ASM_ENODEVICE.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <fcntl.h>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <unordered_map>
#include <vector>

#include "../../include/asm_mi355x.h"
#include "asm_host.h"
#include "asm_kernels.h"
#include "asm_greedy3_kernel.h"
#include "asm_wide.h"
#include "asm_wave.h"
#include "asm_group.h"
#include "asm_cover.h"
#include "asm_tails.h"
#include "asm_filter.h"
#include "asm_ingest.h"

struct asm_handle {
    unsigned long long serial = 0;        /* unique over the life of the process: a batch names its owner by (pointer, serial), so a
                                             new handle that happens to get a destroyed one's address is not taken for it */
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    unsigned long long* d_pair_queue = nullptr; /* chunk counter of the wave-per-pair Greedy kernel (PairQueue, asm_wave.h) */
    hipStream_t side_stream = nullptr;    /* asm_run_benchmark_async runs Greedy beside the NW -> LEAP chain */
    hipStream_t pack_stream = nullptr;    /* ... and, with repack = 2, packs for this call while the previous call still aligns */
    hipEvent_t ev_packed = nullptr;
    hipEvent_t ev_nw = nullptr;
    hipStream_t acc_stream = nullptr;     /* repack = 3: the counters of a call, behind both of its chains */
    hipEvent_t ev_leap = nullptr, ev_tail = nullptr;
    bool tail_set = false;
    hipEvent_t ev_out[2] = {nullptr, nullptr}; /* repack = 3: the counters of the call before last (same output arrays) are done */
    unsigned calls3 = 0;
    const void* last3_out[3] = {nullptr, nullptr, nullptr}; /* the output arrays of the previous overlapped call (misuse guard) */
    bool last3_valid = false;
    bool pipe_prev = false;               /* the previous asm_run_benchmark_async call was a pipelined one (repack 2 or 3) */
    hipEvent_t ev_gate = nullptr;         /* repack = 2: the next call's pack starts behind this point of the current call */
    bool gate_set = false;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_switch = nullptr;
    bool overlap = true;                  /* ASM_OVERLAP=0: everything on one stream */
    std::string err;
    int num_cus = 256;
    int refill_greedy = 8;                /* idle lanes a wave of the persistent Greedy kernels collects before it refills (1 … 32 measured: flat) */
    bool greedy_fast = true;              /* Greedy, k <= 3, unit penalties, GLOBAL: the straight-line pass with integer rank keys
                                             (asm_greedy3.h; ASM_GREEDY_FAST=0: the FP64 kernel) */
    /* its rank tables (66 KB each), one per (significance constants, K) seen so far.  A table is written once, before its first
     * kernel, and never again — a call with other parameters gets another table, so kernels of earlier, still running calls on
     * other streams keep reading theirs.  The newest is tried first. */
    struct G3Table {
        G3Sig sig;
        int k;
        bool ok;
        uint2* d;
    };
    std::vector<G3Table> g3_tables;
    const uint2* d_g3_table = nullptr;    /* the table of the current call (set by g3_prepare) */
    bool leap_hint = true;                /* LEAP scheduled by a work hint when one is given (ASM_LEAP_HINT=0 disables) */
    bool bucketing = true;                /* group mixed-length batches by width class (ASM_BUCKET=0 disables) */
    bool wave_kernels = true;             /* wave-per-pair kernels for 6 <= k <= 31 (ASM_WAVE=0: workgroup-per-pair LDS kernels) */
    bool nw_bylen = true;                 /* unit-cost NW on mixed-length batches: workgroup-local sort by length (ASM_NW_BYLEN=0) */
    bool nw_banded = true;                /* banded bit-parallel NW with in-kernel full-height recompute (ASM_NW_BANDED=0) */
    bool nw_wfa = true;                   /* affine NW: banded wavefront first, full matrix for the rest (ASM_NW_WFA=0: full matrix only) */
    std::vector<hipEvent_t> prof_ev;      /* asm_profile_enable: 8 events per recorded asm_run_benchmark_async call */
    std::vector<unsigned> prof_mask;      /* which of a call's four kernels were launched */
    int prof_cap = 0;
    unsigned prof_select = 0xfu;          /* which kernels are bracketed: bit 0 pack, 1 NW, 2 LEAP, 3 Greedy */
    void* d_sort = nullptr;               /* wide-band LEAP: keys, permutation and radix-sort scratch of the global work sort */
    size_t sort_cap = 0;
    bool leap_sort = true;                /* ASM_LEAP_SORT=0: work-sort inside workgroups only */
    uint32_t* d_todo = nullptr;           /* affine NW: [0] = count, [1..] = bucket slots the wavefront band could not settle */
    size_t todo_cap = 0;
    /* Device memory of batches is recycled instead of freed: a streamed file, or the reference-shaped per-pair objects, create
     * and drop a batch per call, and hipMalloc / hipFree cost more than the kernels (hipFree also drains the device).  Blocks
     * are handed out again in stream order (everything a handle enqueues is ordered on its stream), so no wait is needed. */
    std::unordered_map<void*, size_t> pool_live;
    std::multimap<size_t, void*> pool_idle;
    size_t pool_idle_bytes = 0;
    bool pooling = true;                  /* ASM_POOL=0: plain hipMalloc / hipFree */
    /* pinned host memory of asm_stream_seq_file, kept between calls (pinning 200 MB costs as much as streaming it) */
    char* pin_raw[3] = {nullptr, nullptr, nullptr};
    size_t pin_raw_cap = 0;
    int32_t* pin_pen[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    int64_t pin_pen_cap[2][3] = {{0, 0, 0}, {0, 0, 0}};
};

static size_t pool_round(size_t bytes) { /* eight size classes per octave, at least 4 KiB */
    if (bytes <= 4096) return 4096;
    int top = 63 - __builtin_clzll((unsigned long long)bytes);
    const size_t step = (size_t)1 << (top - 3);
    return (bytes + step - 1) & ~(step - 1);
}

static void pool_release_idle(asm_handle* h) {
    for (auto& kv : h->pool_idle) (void)hipFree(kv.second);
    h->pool_idle.clear();
    h->pool_idle_bytes = 0;
}

static hipError_t pool_alloc(asm_handle* h, void** p, size_t bytes) {
    if (!h->pooling) return hipMalloc(p, bytes ? bytes : 1);
    const size_t want = pool_round(bytes);
    auto it = h->pool_idle.lower_bound(want);
    if (it != h->pool_idle.end() && it->first <= want + want / 4) { /* a block of this class or slightly above */
        *p = it->second;
        h->pool_live[*p] = it->first;
        h->pool_idle_bytes -= it->first;
        h->pool_idle.erase(it);
        return hipSuccess;
    }
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) { /* out of memory: give the idle blocks back and try once more */
        (void)hipGetLastError();
        pool_release_idle(h);
        e = hipMalloc(p, want);
    }
    if (e == hipSuccess) h->pool_live[*p] = want;
    return e;
}

/* hipMalloc for the scratch blocks that do not come from the pool (traceback cells, sort scratch, todo lists, ...): when the
 * device is full, the idle blocks of the pool (up to 24 GiB) are what is in the way — give them back and try once more. */
static hipError_t big_malloc(asm_handle* h, void** p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess && h && !h->pool_idle.empty()) {
        (void)hipGetLastError();
        (void)hipDeviceSynchronize(); /* idle blocks may still be read by queued work */
        pool_release_idle(h);
        e = hipMalloc(p, bytes);
    }
    return e;
}

static void pool_free(asm_handle* h, void* p) {
    if (!p) return;
    if (!h || !h->pooling) {
        (void)hipFree(p);
        return;
    }
    auto it = h->pool_live.find(p);
    if (it == h->pool_live.end()) { /* not ours (allocated before pooling was switched, or by another handle) */
        (void)hipFree(p);
        return;
    }
    const size_t sz = it->second;
    h->pool_live.erase(it);
    if (h->pool_idle_bytes + sz > ((size_t)24 << 30)) { /* keep at most 24 GiB idle */
        (void)hipFree(p);
        return;
    }
    h->pool_idle.emplace(sz, p);
    h->pool_idle_bytes += sz;
}

/* One width class of a batch: pairs whose longer string needs `w4` granules of 128 positions. */
struct asm_bucket {
    int64_t n = 0;
    int w4 = 1;
    int maxlen = 0;
    uint4* planes = nullptr;    /* uint4[4][w4][n], inside asm_batch::d_planes */
    uint32_t* lens = nullptr;   /* uint32[n],       inside asm_batch::d_lens   */
    uint32_t* order = nullptr;  /* bucket slot -> pair index; null when the batch is one bucket in input order */
    bool mixed = false;         /* one of several width classes of a mixed-length batch */
};

struct asm_batch {
    asm_handle* owner = nullptr; /* whose pool the device blocks come from */
    unsigned long long owner_serial = 0;
    int64_t n = 0;
    int maxlen = 0;
    int greedy_mode = ASM_GREEDY_CLEAN;
    size_t reads_bytes = 0, refs_bytes = 0;
    char* d_reads = nullptr;
    char* d_refs = nullptr;
    uint32_t* d_read_off = nullptr;
    uint32_t* d_ref_off = nullptr;
    uint4* d_planes = nullptr;  /* all buckets back to back */
    uint32_t* d_lens = nullptr; /* in bucketed order */
    uint32_t* d_order = nullptr; /* bucketed slot -> pair index (null: identity) */
    uint32_t* d_pos = nullptr;   /* pair index -> bucketed slot (null: identity) */
    /* pipelined repack (asm_run_benchmark_async, repack = 2): a second set of planes/lens that the next call's pack fills
     * while this call's aligners still read the current one */
    uint4* d_planes_alt = nullptr;
    uint32_t* d_lens_alt = nullptr;
    size_t planes_total = 0;     /* uint4 entries in d_planes */
    hipEvent_t ev_consumed[2] = {nullptr, nullptr}; /* aligners of the call that used buffer q are done */
    int cur = 0;
    uint4* d_tails = nullptr;    /* sequential mode: stale-tail planes, uint4[4][n] in input order */
    uint4* d_tail_g0 = nullptr;  /* tail resolver scratch (asm_tails.h), allocated on first use: clean granule 0 in input order, */
    uint32_t* d_tail_l0 = nullptr; /* its lengths, */
    uint8_t* d_tail_chunks = nullptr; /* per-chunk prefixes, per-workgroup totals and carries + the 256-byte summary */
    int nb = 1;
    asm_bucket bk[4];
    PackBuckets pb;
};

struct asm_reference {
    char* d_text = nullptr;
    size_t len = 0;
};

static thread_local std::string g_err;
/* handles that exist: a batch remembers the handle whose pool its device blocks came from, and may be freed through another
 * handle, or after its own is gone */
static std::mutex g_live_mu;
static std::vector<asm_handle*> g_live_handles;
static unsigned long long g_next_serial = 1; /* under g_live_mu */
/* call with g_live_mu held: is the handle that created a batch still the one living at that address? */
static bool owner_is_live_locked(const asm_handle* h, unsigned long long serial) {
    for (asm_handle* q : g_live_handles)
        if (q == h) return q->serial == serial;
    return false;
}

static int fail(asm_handle* h, int code, const std::string& msg) {
    g_err = msg;
    if (h) h->err = msg;
    return code;
}

#define HIPCHK(h, call)                                                                           \
    do {                                                                                          \
        hipError_t _e = (call);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            return fail(h, _e == hipErrorOutOfMemory ? ASM_ENOMEM : ASM_ENODEVICE,                \
                        std::string(#call) + ": " + hipGetErrorString(_e));                       \
        }                                                                                         \
    } while (0)

static int grid_for(int64_t n) { return (int)((n + ASM_BLOCK - 1) / ASM_BLOCK); }

// Persistent launch: grid = what is resident (CUs x occupancy); each wave owns a static slice of the batch.
template <typename Kern, typename... Args>
static hipError_t launch_persistent(asm_handle* h, Kern kern, int64_t n, Args... args) {
    int per_cu = 1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, ASM_BLOCK, 0);
    if (e != hipSuccess) return e;
    if (per_cu < 1) per_cu = 1;
    int64_t blocks = (int64_t)per_cu * h->num_cus;
    const int64_t need = (n + ASM_BLOCK - 1) / ASM_BLOCK;
    if (blocks > need) blocks = need;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(ASM_BLOCK), 0, h->stream, args...);
    return hipGetLastError();
}

#ifdef GREEDY_DIAG
static void* g_diag_buf = nullptr; /* diagnostic build only (never the shipped library): per-wave cycle stamps of the Greedy kernel */
extern "C" void asm_diag_set_buffer(void* d) { g_diag_buf = d; }
#endif
/* Rank table of the straight-line Greedy kernel for these significance constants (asm_greedy3.h): built on the host with the
 * reference build's three roundings, uploaded once per (constants, k).  False when the constants do not have the structure
 * the integer keys need (then the FP64 kernel runs). */
static bool g3_prepare(asm_handle* h, const GreedyArgs& ga, int K) {
    const G3Sig sig = {ga.sig_match, ga.sig_mismatch, ga.sig_indel};
    for (size_t q = h->g3_tables.size(); q-- > 0;) {
        const asm_handle::G3Table& t = h->g3_tables[q];
        if (t.k == K && memcmp(&sig, &t.sig, sizeof(sig)) == 0) {
            h->d_g3_table = t.d;
            return t.ok;
        }
    }
    /* a new parameter set: builds the table on the host and BLOCKS until it is on the device (once per parameter set) */
    asm_handle::G3Table t = {sig, K, false, nullptr};
    std::vector<uint2> tab;
    if (g3_build_table(sig, K, tab)) {
        if (h->g3_tables.size() >= 16) { /* a caller cycling through parameter sets: drop the oldest, after everything has drained */
            (void)hipDeviceSynchronize();
            if (h->g3_tables.front().d) (void)hipFree(h->g3_tables.front().d);
            h->g3_tables.erase(h->g3_tables.begin());
        }
        if (big_malloc(h, (void**)&t.d, sizeof(uint2) * G3_TABLE_ENTRIES) == hipSuccess &&
            hipMemcpyAsync(t.d, tab.data(), sizeof(uint2) * G3_TABLE_ENTRIES, hipMemcpyHostToDevice, h->stream) == hipSuccess &&
            hipStreamSynchronize(h->stream) == hipSuccess) { /* pageable source: staged before the call returns */
            t.ok = true;
        } else {
            (void)hipGetLastError();
            if (t.d) (void)hipFree(t.d);
            t.d = nullptr;
        }
    }
    h->g3_tables.push_back(t);
    h->d_g3_table = t.d;
    return t.ok;
}

template <int K>
static hipError_t launch_greedy_fast(asm_handle* h, const asm_bucket& b, const GreedyArgs& ga, OutMap out, CigarSink cig) {
    /* One 512-thread workgroup per CU = two waves per SIMD.  The waves of a SIMD are served oldest first and this kernel is a
     * dense stream of 4-cycle vector operations, so a third wave adds little issue rate and a third more lanes to drain at the
     * end: stand-alone 1 / 2 / 3 waves take 130 / 118 / 107 us, inside asm_run_benchmark_async's overlapped step 0.234 / 0.227 /
     * 0.244 ms per step (1.5 and 2.5 waves: 0.237, —).  The other instantiations were removed in round 4 (last in 312851a). */
    constexpr int NT = G3_THREADS;
    auto kern = greedy_fast_kernel<K, NT>;
    const size_t lds = g3_lds_bytes(K, NT);
    { /* more than 64 KB of dynamic LDS has to be asked for */
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    int64_t blocks = h->num_cus; /* one workgroup per CU (LDS) */
    const int64_t need = (b.n + NT - 1) / NT;
    if (blocks > need) blocks = need;
    const G3Sig sig = {ga.sig_match, ga.sig_mismatch, ga.sig_indel};
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NT), lds, h->stream, (const uint4*)b.planes, (const uint32_t*)b.lens,
                       (long)b.n, b.w4, sig, h->d_g3_table, out, cig, h->refill_greedy);
    return hipGetLastError();
}

/* Thread-per-pair Greedy, k <= 16: the straight-line integer-key kernel for the benchmark's own configuration (k <= 3, unit
 * penalties, GLOBAL), the FP64 lane-refilling kernel for everything else (unit and general penalties, GLOBAL and SEMI_GLOBAL:
 * at k = 8 with (2,3,1) 0.27 ms per 10^6 pairs against 0.87 for the wave-per-pair kernel). */
template <int K>
static hipError_t launch_greedy(asm_handle* h, const asm_bucket& b, const GreedyArgs& ga, OutMap out, CigarSink cig) {
#ifdef GREEDY_DIAG
    if (cig.ops == nullptr && g_diag_buf) cig.nops = (uint8_t*)g_diag_buf;
#endif
    const bool unit = ga.x == 1 && ga.o == 1 && ga.e == 1 && !ga.semi;
    if constexpr (K <= 3) {
        if (h->greedy_fast && unit && g3_prepare(h, ga, K)) return launch_greedy_fast<K>(h, b, ga, out, cig);
    }
    if (unit)
        return launch_persistent(h, greedy_persist_kernel<K, true>, b.n, (const uint4*)b.planes, (const uint32_t*)b.lens,
                                 (long)b.n, b.w4, ga, out, cig, h->refill_greedy);
    return launch_persistent(h, greedy_persist_kernel<K, false>, b.n, (const uint4*)b.planes, (const uint32_t*)b.lens,
                             (long)b.n, b.w4, ga, out, cig, h->refill_greedy);
}

#define LEAP_UNIT_WIDE_K 5 /* thread-per-pair unit-cost LEAP at every string length up to this band */
#define LEAP_UNIT_MAX_K 10 /* ... and for strings of one granule (<= 128 characters) up to this one (k = 12: 0.148 ms against
                              0.180 at C2, but 0.120 against 0.104 at C4's 2.5 % errors, where four threads per pair have little to do) */
template <int K, int W64>
static hipError_t launch_leap_unit_w(asm_handle* h, const asm_bucket& b, OutMap out, const int32_t* hint) {
    if (hint && h->leap_hint) {
        const unsigned blocks = (unsigned)((b.n + LEAP_HINT_PAIRS - 1) / LEAP_HINT_PAIRS);
        hipLaunchKernelGGL((leap_unit_hint_kernel<K, W64>), dim3(blocks), dim3(ASM_BLOCK), 0, h->stream, b.planes, b.lens,
                           (long)b.n, b.w4, out, hint);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((leap_unit_kernel<K, W64>), dim3(grid_for(b.n)), dim3(ASM_BLOCK), 0, h->stream, b.planes, b.lens,
                       (long)b.n, b.w4, out);
    return hipGetLastError();
}

template <int K, int W64>
static hipError_t launch_leap_general_w(asm_handle* h, const asm_bucket& b, const asm_params* p, OutMap out) {
    const dim3 grid((unsigned)((b.n + LEAP_GEN_THREADS - 1) / LEAP_GEN_THREADS)), block(LEAP_GEN_THREADS);
    const RingGeometry rg(p->x, p->o, p->e);
    /* bytes (every stored position + 2 fits) halve the LDS and double the waves per CU, but four threads then write into one
     * dword: worth it only where shorts would leave the CU underfilled */
    const bool bytes = b.maxlen + 4 <= 255 && rg.lds_bytes(2 * K + 1, LEAP_GEN_THREADS, 2) > 20 * 1024;
    if (bytes)
        hipLaunchKernelGGL((leap_general_kernel<K, W64, uint8_t>), grid, block, rg.lds_bytes(2 * K + 1, LEAP_GEN_THREADS, 1), h->stream,
                           b.planes, b.lens, (long)b.n, b.w4, (int)p->x, (int)p->o, (int)p->e, rg.gm, rg.gi, out);
    else
        hipLaunchKernelGGL((leap_general_kernel<K, W64, uint16_t>), grid, block, rg.lds_bytes(2 * K + 1, LEAP_GEN_THREADS, 2), h->stream,
                           b.planes, b.lens, (long)b.n, b.w4, (int)p->x, (int)p->o, (int)p->e, rg.gm, rg.gi, out);
    return hipGetLastError();
}

/* Affine NW: banded wavefront pass (|d| <= 7), a second one with |d| <= 15 over the pairs the first could not settle
 * (strings up to 128 only), then the full-matrix kernel over what is left.  Two todo lists of n + 1 words each. */
template <int K, int W64, typename EnT>
static void launch_nw_wfa_pass(asm_handle* h, const asm_bucket& b, const asm_params* p, OutMap out, const uint32_t* in_list,
                               const uint32_t* in_count, uint32_t* todo, uint32_t* todo_count) {
    const WfaRings rg(p->x, p->o, p->e);
    const dim3 grid((unsigned)((b.n + LEAP_GEN_THREADS - 1) / LEAP_GEN_THREADS)), block(LEAP_GEN_THREADS);
    hipLaunchKernelGGL((nw_wfa_kernel<K, W64, EnT, false>), grid, block, rg.lds_bytes(2 * K + 1, LEAP_GEN_THREADS, sizeof(EnT)),
                       h->stream, b.planes, b.lens, (long)b.n, b.w4, (int)p->x, (int)p->o, (int)p->e, rg.gm, rg.gi, out, in_list,
                       in_count, todo, todo_count);
}

template <int W64, int MAXROWS>
static int launch_nw_wfa(asm_handle* h, const asm_bucket& b, const asm_params* p, OutMap out) {
    const size_t words = 2 * ((size_t)b.n + 1);
    if (h->todo_cap < words) {
        if (h->d_todo) (void)hipFree(h->d_todo);
        h->d_todo = nullptr, h->todo_cap = 0;
        HIPCHK(h, big_malloc(h, (void**)&h->d_todo, sizeof(uint32_t) * words));
        h->todo_cap = words;
    }
    uint32_t* const list_a = h->d_todo;                 /* [0] = count, [1..] = pair slots */
    uint32_t* const list_b = h->d_todo + (size_t)b.n + 1;
    HIPCHK(h, hipMemsetAsync(list_a, 0, sizeof(uint32_t), h->stream));
    HIPCHK(h, hipMemsetAsync(list_b, 0, sizeof(uint32_t), h->stream));
    const WfaRings rg(p->x, p->o, p->e);
    bool bytes = false; /* strings up to 128: every stored position + 2 fits a byte, half the LDS */
    if constexpr (W64 == 2) {
        bytes = true;
        launch_nw_wfa_pass<NW_WFA_K, 2, uint8_t>(h, b, p, out, nullptr, nullptr, list_a + 1, list_a);
    }
    if (!bytes) launch_nw_wfa_pass<NW_WFA_K, W64, uint16_t>(h, b, p, out, nullptr, nullptr, list_a + 1, list_a);
    const uint32_t* rest = list_a;
    const size_t en = b.maxlen + 2 <= 255 ? 1 : 2;
    if (nw_oct_lds(2 * W64, NW_WFA_K2, rg.gm, rg.gi, en) <= 64 * 1024) {
        /* the unsettled percent or two: eight threads per pair, |d| <= 15 (asm_wave.h) */
        const dim3 grid((unsigned)std::min<int64_t>((b.n + NW_OCT_PAIRS - 1) / NW_OCT_PAIRS, (int64_t)h->num_cus * 16));
        if (en == 1)
            hipLaunchKernelGGL((nw_oct_kernel<2 * W64, uint8_t>), grid, dim3(NW_OCT_THREADS), nw_oct_lds(2 * W64, NW_WFA_K2, rg.gm, rg.gi, 1),
                               h->stream, b.planes, b.lens, (long)b.n, b.w4, NW_WFA_K2, (int)p->x, (int)p->o, (int)p->e, rg.gm, rg.gi, out,
                               (const uint32_t*)(list_a + 1), (const uint32_t*)list_a, list_b + 1, list_b);
        else
            hipLaunchKernelGGL((nw_oct_kernel<2 * W64, uint16_t>), grid, dim3(NW_OCT_THREADS), nw_oct_lds(2 * W64, NW_WFA_K2, rg.gm, rg.gi, 2),
                               h->stream, b.planes, b.lens, (long)b.n, b.w4, NW_WFA_K2, (int)p->x, (int)p->o, (int)p->e, rg.gm, rg.gi, out,
                               (const uint32_t*)(list_a + 1), (const uint32_t*)list_a, list_b + 1, list_b);
        rest = list_b;
    }
    hipLaunchKernelGGL((nw_affine_kernel<W64, MAXROWS>), dim3((unsigned)((b.n + 63) / 64)), dim3(64), 0, h->stream, b.planes,
                       b.lens, (long)b.n, b.w4, (int)p->x, (int)p->o, (int)p->e, out, rest + 1, rest);
    HIPCHK(h, hipGetLastError());
    return ASM_OK;
}

template <int K>
static hipError_t launch_leap_general(asm_handle* h, const asm_bucket& b, const asm_params* p, OutMap out) {
    if (b.maxlen <= 128) return launch_leap_general_w<K, 2>(h, b, p, out);
    if constexpr (K <= 5) return launch_leap_general_w<K, 4>(h, b, p, out);
    return hipErrorInvalidValue; /* not reached: bands 6..8 come here for strings of one granule only (align_bucket) */
}

template <int K>
static hipError_t launch_leap_unit(asm_handle* h, const asm_bucket& b, OutMap out, const int32_t* hint) {
    if (b.maxlen <= 128) return launch_leap_unit_w<K, 2>(h, b, out, hint);
    if constexpr (K <= LEAP_UNIT_WIDE_K) { /* the wider bands (6..8 lanes each side) are kept in registers for one granule only */
        if (b.maxlen <= 192) return launch_leap_unit_w<K, 3>(h, b, out, hint);
        if (b.maxlen <= 256) return launch_leap_unit_w<K, 4>(h, b, out, hint);
        if (b.maxlen <= 320) return launch_leap_unit_w<K, 5>(h, b, out, hint); /* C5's longest class (257-300) on five words */
        return launch_leap_unit_w<K, 6>(h, b, out, hint);
    }
    return hipErrorInvalidValue; /* not reached: align_bucket sends such buckets to the four-threads-per-pair kernel */
}

static int ensure_todo(asm_handle* h, size_t n) {
    if (h->todo_cap < n + 1) {
        if (h->d_todo) (void)hipFree(h->d_todo);
        h->d_todo = nullptr, h->todo_cap = 0;
        HIPCHK(h, big_malloc(h, (void**)&h->d_todo, sizeof(uint32_t) * (n + 1)));
        h->todo_cap = n + 1;
    }
    return ASM_OK;
}

/* Full-matrix Gotoh + traceback + coverage verdict (nw_trace_affine_kernel) over `count` slots of a bucket slice: the slots
 * listed in d_list (bucket-slice indices), or slots 0..count-1 when d_list is null.  Scratch: 4 direction bits per cell. */
static int cover_full_matrix(asm_handle* h, const asm_bucket& b, const asm_params* p, const uint4* planes, const uint32_t* lens,
                             const uint32_t* order, long slice_lo, const uint32_t* d_list, int64_t count, const CoverArgs& ca) {
    if (count <= 0) return ASM_OK;
    const int rows = b.maxlen > 0 ? b.maxlen : 1, cols8 = (rows + 7) / 8;
    const size_t per_pair = (size_t)rows * (size_t)cols8 * sizeof(uint32_t);
    int64_t chunk = (int64_t)((size_t)(6ull << 30) / per_pair);
    chunk = chunk < 64 ? 64 : chunk;
    chunk = chunk > count ? count : chunk;
    uint32_t* d_scratch = nullptr;
    HIPCHK(h, big_malloc(h, (void**)&d_scratch, per_pair * (size_t)chunk));
    int rc = ASM_OK;
    for (int64_t lo = 0; lo < count && !rc; lo += chunk) {
        const int64_t c = count - lo < chunk ? count - lo : chunk;
        const dim3 grid((unsigned)((c + 63) / 64)), block(64);
        /* without a list the chunk is a contiguous run of slots: shift the base pointers like cover_bucket does */
        const uint4* pl = d_list ? planes : planes + lo;
        const uint32_t* ln = d_list ? lens : lens + lo;
        const uint32_t* od = (d_list || !order) ? order : order + lo;
        const long base = d_list ? slice_lo : slice_lo + lo;
        const uint32_t* list = d_list ? d_list + lo : nullptr;
#define AFFINE_TRACE(W64, ROWS)                                                                                                  \
    hipLaunchKernelGGL((nw_trace_affine_kernel<W64, ROWS>), grid, block, 0, h->stream, pl, ln, (long)c, (long)b.n, b.w4, (int)p->x, \
                       (int)p->o, (int)p->e, d_scratch, cols8, list, od, base, ca)
        if (b.maxlen <= 128) AFFINE_TRACE(2, 128);
        else if (b.maxlen <= 256) AFFINE_TRACE(4, 256);
        else AFFINE_TRACE(8, 512);
#undef AFFINE_TRACE
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
            rc = fail(h, ASM_ENODEVICE, "asm_coverage: full-matrix traceback kernel failed");
    }
    (void)hipFree(d_scratch);
    return rc;
}

/* forward sweep + traceback/coverage for one width class with window W; pairs the band cannot answer go through the
 * full-matrix pass */
template <int ND, int W>
static int cover_bucket(asm_handle* h, const asm_bucket& b, const asm_params* p, const CoverArgs& ca) {
    typedef typename TraceCell<W>::T Cell;
    if (b.n == 0) return ASM_OK;
    // scratch: [column][pair]; processed in slices of pairs so that it stays below ~6 GiB
    const int64_t cols = b.maxlen > 0 ? b.maxlen : 1;
    int64_t slice = (int64_t)(6ll << 30) / (cols * (int64_t)sizeof(Cell));
    slice = slice > b.n ? b.n : (slice < 4096 ? 4096 : slice);
    int rc = ensure_todo(h, (size_t)slice);
    if (rc) return rc;
    Cell* d_trace = nullptr;
    int32_t* d_band = nullptr;
    HIPCHK(h, big_malloc(h, (void**)&d_trace, (size_t)cols * (size_t)slice * sizeof(Cell)));
    if (big_malloc(h, (void**)&d_band, sizeof(int32_t) * (size_t)slice) != hipSuccess) {
        (void)hipFree(d_trace);
        return fail(h, ASM_ENOMEM, "asm_coverage: hipMalloc failed");
    }
    for (int64_t lo = 0; lo < b.n && !rc; lo += slice) {
        const int64_t cnt = b.n - lo < slice ? b.n - lo : slice;
        // a slice is addressed as a sub-batch: plane rows keep the bucket's stride, so pass shifted base pointers
        const uint4* planes = b.planes + lo;
        const uint32_t* lens = b.lens + lo;
        const uint32_t* order = b.order ? b.order + lo : nullptr;
        uint32_t todo_count = 0;
        if (hipMemsetAsync(h->d_todo, 0, sizeof(uint32_t), h->stream) != hipSuccess) rc = fail(h, ASM_ENODEVICE, "asm_coverage: memset failed");
        // the kernels index planes as [(p*w4+g)*n + i] with n = pairs of the BUCKET: use dedicated strided variants
        hipLaunchKernelGGL((nw_trace_forward_kernel<ND, W>), dim3(grid_for(cnt)), dim3(ASM_BLOCK), 0, h->stream, planes, lens,
                           (long)cnt, (long)b.n, b.w4, d_trace, d_band);
        hipLaunchKernelGGL((nw_trace_cover_kernel<ND / 2, W>), dim3(grid_for(cnt)), dim3(ASM_BLOCK), 0, h->stream, planes, lens,
                           (long)cnt, (long)b.n, b.w4, (const Cell*)d_trace, (const int32_t*)d_band, order, lo, ca, h->d_todo + 1,
                           h->d_todo);
        if (!rc && (hipGetLastError() != hipSuccess ||
                    hipMemcpyAsync(&todo_count, h->d_todo, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                    hipStreamSynchronize(h->stream) != hipSuccess))
            rc = fail(h, ASM_ENODEVICE, "asm_coverage: kernel failed");
        if (!rc && todo_count) rc = cover_full_matrix(h, b, p, planes, lens, order, lo, h->d_todo + 1, (int64_t)todo_count, ca);
    }
    (void)hipFree(d_trace);
    (void)hipFree(d_band);
    return rc;
}

/* SIMD_ED events of one bucket (asm_filter.h) */
template <int W64>
static int launch_simd_ed_events(asm_handle* h, const asm_bucket& b, int T, int shd, OutMap ev, int ed_mode = 0) {
    const dim3 grid((unsigned)((b.n + SIMD_ED_THREADS - 1) / SIMD_ED_THREADS)), block(SIMD_ED_THREADS);
#define SIMD_ED_CASE(TT)                                                                                                \
    case TT:                                                                                                            \
        hipLaunchKernelGGL((simd_ed_kernel<TT, W64>), grid, block, 0, h->stream, b.planes, b.lens, (long)b.n, b.w4, T, shd, 0, ev); \
        break;
    switch ((T <= ASM_FILTER_REG_MAX_T && ed_mode != 1 && ed_mode != 2) ? T : 0) { /* every lane live from generation 0: run-time form */
        SIMD_ED_CASE(1)
        SIMD_ED_CASE(2)
        SIMD_ED_CASE(3)
        SIMD_ED_CASE(4)
        SIMD_ED_CASE(5)
        SIMD_ED_CASE(6)
        SIMD_ED_CASE(7)
        SIMD_ED_CASE(8)
        default: {
            const size_t lds = (size_t)2 * (2 * T + 3) * SIMD_ED_THREADS * sizeof(short);
            hipLaunchKernelGGL((simd_ed_kernel<0, W64>), grid, block, lds, h->stream, b.planes, b.lens, (long)b.n, b.w4, T, shd,
                               ed_mode, ev);
        }
    }
#undef SIMD_ED_CASE
    HIPCHK(h, hipGetLastError());
    return ASM_OK;
}

extern "C" {

const char* asm_version(void) { return "asm_mi355x 0.1 (gfx950)"; }

void asm_default_params(asm_params* p) {
    if (!p) return;
    p->k = 3; /* benchmark.cpp:22 */
    p->x = p->o = p->e = 1;
    p->p_match = 0.80; /* hurdle_matrix.h:557-559 */
    p->p_mismatch = 0.20 / 3;
    p->p_indel = 0.40 / 3;
    p->alignment_type = ASM_ALIGN_GLOBAL; /* hurdle_matrix.h:553 default; the harness never passes another */
    p->leap_mode = ASM_LEAP_GLOBAL;
}

int asm_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

int asm_create(asm_handle** out, int device) {
    if (!out) return fail(nullptr, ASM_EINVAL, "asm_create: out is NULL");
    *out = nullptr;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0)
        return fail(nullptr, ASM_ENODEVICE, "asm_create: no HIP device is visible (this library has no CPU path)");
    if (device < 0 || device >= c) return fail(nullptr, ASM_EINVAL, "asm_create: device index out of range");
    asm_handle* h = new asm_handle;
    h->device = device;
    HIPCHK(h, hipSetDevice(device));
    HIPCHK(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    HIPCHK(h, hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
    HIPCHK(h, hipMalloc((void**)&h->d_pair_queue, sizeof(unsigned long long)));
    HIPCHK(h, hipStreamCreateWithFlags(&h->pack_stream, hipStreamNonBlocking));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_packed, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_gate, hipEventDisableTiming));
    HIPCHK(h, hipStreamCreateWithFlags(&h->acc_stream, hipStreamNonBlocking));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_nw, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_leap, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_tail, hipEventDisableTiming));
    for (hipEvent_t& ev : h->ev_out) HIPCHK(h, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_switch, hipEventDisableTiming));
    hipDeviceProp_t prop;
    HIPCHK(h, hipGetDeviceProperties(&prop, device));
    h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    /* The switches the library keeps (read once, here): one fallback per kernel family, the allocator, and the stream layout.
     * Everything else that used to be switchable was an A/B whose outcome is recorded in DESIGN.md. */
    const char* env;
    if ((env = getenv("ASM_OVERLAP"))) h->overlap = env[0] != '0';         /* 0: asm_run_benchmark_async on one stream */
    if ((env = getenv("ASM_LEAP_HINT"))) h->leap_hint = env[0] != '0';     /* 0: LEAP in input order (no work sort) */
    if ((env = getenv("ASM_LEAP_SORT"))) h->leap_sort = env[0] != '0';     /* 0: wide-band LEAP sorts inside workgroups only */
    if ((env = getenv("ASM_BUCKET"))) h->bucketing = env[0] != '0';        /* 0: mixed-length batches in one width class */
    if ((env = getenv("ASM_WAVE"))) h->wave_kernels = env[0] != '0';       /* 0: workgroup-per-pair fallbacks (asm_wide.h) */
    if ((env = getenv("ASM_POOL"))) h->pooling = env[0] != '0';            /* 0: plain hipMalloc / hipFree */
    if ((env = getenv("ASM_NW_BANDED"))) h->nw_banded = env[0] != '0';     /* 0: full-height bit-parallel NW */
    if ((env = getenv("ASM_NW_BYLEN"))) h->nw_bylen = env[0] != '0';       /* 0: mixed-length NW without the length sort */
    if ((env = getenv("ASM_NW_WFA"))) h->nw_wfa = env[0] != '0';           /* 0: affine NW by the full matrix only */
    if ((env = getenv("ASM_GREEDY_FAST"))) h->greedy_fast = env[0] != '0'; /* 0: FP64 Greedy kernel at k <= 3 */
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        h->serial = g_next_serial++;
        g_live_handles.push_back(h);
    }
    *out = h;
    return ASM_OK;
}

int asm_destroy(asm_handle* h) {
    if (!h) return ASM_OK;
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        for (size_t i = 0; i < g_live_handles.size(); i++)
            if (g_live_handles[i] == h) {
                g_live_handles.erase(g_live_handles.begin() + (long)i);
                break;
            }
    }
    (void)hipSetDevice(h->device);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    if (h->d_pair_queue) (void)hipFree(h->d_pair_queue);
    if (h->d_sort) (void)hipFree(h->d_sort);
    if (h->pack_stream) (void)hipStreamDestroy(h->pack_stream);
    if (h->ev_packed) (void)hipEventDestroy(h->ev_packed);
    if (h->ev_gate) (void)hipEventDestroy(h->ev_gate);
    if (h->ev_leap) (void)hipEventDestroy(h->ev_leap);
    if (h->ev_tail) (void)hipEventDestroy(h->ev_tail);
    for (hipEvent_t ev : h->ev_out)
        if (ev) (void)hipEventDestroy(ev);
    if (h->acc_stream) (void)hipStreamDestroy(h->acc_stream);
    if (h->ev_nw) (void)hipEventDestroy(h->ev_nw);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->ev_switch) (void)hipEventDestroy(h->ev_switch);
    if (h->d_todo) (void)hipFree(h->d_todo);
    for (auto& t : h->g3_tables)
        if (t.d) (void)hipFree(t.d);
    for (hipEvent_t ev : h->prof_ev) (void)hipEventDestroy(ev);
    (void)hipDeviceSynchronize();
    for (char* q : h->pin_raw)
        if (q) (void)hipHostFree(q);
    for (auto& row : h->pin_pen)
        for (int32_t* q : row)
            if (q) (void)hipHostFree(q);
    pool_release_idle(h);
    for (auto& kv : h->pool_live) (void)hipFree(kv.first); /* batches the caller never freed */
    delete h;
    return ASM_OK;
}

const char* asm_last_error(const asm_handle* h) { return h ? h->err.c_str() : g_err.c_str(); }

/* The pool recycles freed blocks in the order of the handle's stream.  When that stream changes, work queued on the old one may
 * still use blocks that are live now and go back to the pool later (or are idle already): everything enqueued under the new
 * stream is therefore ordered behind everything enqueued so far on the old one — an event, no host wait. */
static int switch_stream(asm_handle* h, hipStream_t next) {
    if (next == h->stream) return ASM_OK;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipEventRecord(h->ev_switch, h->stream));
    HIPCHK(h, hipStreamWaitEvent(next, h->ev_switch, 0));
    if (h->tail_set) HIPCHK(h, hipStreamWaitEvent(next, h->ev_tail, 0)); /* overlapped calls end on the library's own streams */
    h->stream = next;
    return ASM_OK;
}

int asm_set_stream(asm_handle* h, void* hip_stream) {
    if (!h) return fail(nullptr, ASM_EINVAL, "asm_set_stream: NULL handle");
    /* NULL is HIP's legacy default stream (what torch.cuda.current_stream().cuda_stream is outside a stream context): work is
     * then ordered with everything else the caller enqueues there.  The handle's own stream is non-blocking — it never
     * synchronises with the legacy stream — so it is only restored on request (asm_reset_stream). */
    return switch_stream(h, (hipStream_t)hip_stream);
}

int asm_reset_stream(asm_handle* h) {
    if (!h) return fail(nullptr, ASM_EINVAL, "asm_reset_stream: NULL handle");
    return switch_stream(h, h->own_stream);
}

int asm_synchronize(asm_handle* h) {
    if (!h) return fail(nullptr, ASM_EINVAL, "asm_synchronize: NULL handle");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->tail_set) HIPCHK(h, hipStreamSynchronize(h->acc_stream)); /* calls with repack = 3 end on the library's own streams */
    return ASM_OK;
}

int asm_pipeline_join_async(asm_handle* h) {
    if (!h) return fail(nullptr, ASM_EINVAL, "asm_pipeline_join_async: NULL handle");
    HIPCHK(h, hipSetDevice(h->device));
    if (h->tail_set) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_tail, 0));
    h->pipe_prev = false; /* the next pipelined call starts behind this point of the caller's stream: any output set may follow */
    h->last3_valid = false;
    return ASM_OK;
}

/* ---------------------------------------------------------------------------------------------------- */
static int check_gen(const asm_gen_config* cfg) {
    std::string err;
    const int rc = asm_host::check_gen(cfg, err);
    return rc ? fail(nullptr, rc, err) : ASM_OK;
}

int asm_generate_pairs(const asm_gen_config* cfg, int64_t first, int64_t n, uint32_t* read_off, uint32_t* ref_off,
                       char* reads, size_t reads_cap, char* refs, size_t refs_cap) {
    std::string err;
    const int rc = asm_host::generate_pairs(cfg, first, n, read_off, ref_off, reads, reads_cap, refs, refs_cap, err);
    return rc ? fail(nullptr, rc, err) : ASM_OK;
}

/* ---------------------------------------------------------------------------------------------------- */

static void batch_release(asm_batch* b) {
    if (!b) return;
    pool_free(b->owner, b->d_reads);
    pool_free(b->owner, b->d_refs);
    pool_free(b->owner, b->d_read_off);
    pool_free(b->owner, b->d_ref_off);
    pool_free(b->owner, b->d_planes);
    pool_free(b->owner, b->d_lens);
    pool_free(b->owner, b->d_planes_alt);
    pool_free(b->owner, b->d_lens_alt);
    for (hipEvent_t ev : b->ev_consumed)
        if (ev) (void)hipEventDestroy(ev);
    pool_free(b->owner, b->d_order);
    pool_free(b->owner, b->d_pos);
    pool_free(b->owner, b->d_tails);
    pool_free(b->owner, b->d_tail_g0);
    pool_free(b->owner, b->d_tail_l0);
    pool_free(b->owner, b->d_tail_chunks);
    delete b;
}

static hipError_t launch_pack(asm_handle* h, const asm_batch* b, const uint4* tails, uint4* planes, uint32_t* lens,
                              const PackBuckets& pb, const uint32_t* pos) {
    const dim3 grid((unsigned)((b->n + PACK_BLOCK - 1) / PACK_BLOCK)), block(PACK_BLOCK);
    int wmax = 1;
    for (int q = 0; q < pb.nb; q++) wmax = pb.w4[q] > wmax ? pb.w4[q] : wmax;
    // LDS staging: room for 256 strings of the batch's longest length (+ alignment slack), at least one string
    size_t stage = (size_t)PACK_BLOCK * (size_t)(b->maxlen + 4) + 64;
    stage = (stage + 1023) & ~(size_t)1023;
    if (stage > PACK_SB) stage = PACK_SB;
    if (stage < 2048) stage = 2048;
    const size_t lds = stage + 256; /* + the over-read slack of pack_convert and the 64 bytes of padding behind the staged data
                                       (pack_pad; the swizzle stays inside a 128-byte row) */
#define PACK_LAUNCH(W, NV)                                                                                               \
    hipLaunchKernelGGL((pack_kernel<W, NV>), grid, block, lds, h->stream, b->d_reads, b->d_read_off, b->d_refs,          \
                       b->d_ref_off, tails, planes, lens, (long)b->n, pb, pos, (uint32_t)stage)
    /* NV = staging vectors a thread may hold in registers: 7 covers 256 strings of up to 108 characters (28 KB), which only a
     * one-granule batch can be; everything longer takes 12 */
    switch (wmax) {
        case 1:
            if (stage <= 7 * PACK_BLOCK * 16) PACK_LAUNCH(1, 7);
            else PACK_LAUNCH(1, 12);
            break;
        case 2: PACK_LAUNCH(2, 12); break;
        case 3: PACK_LAUNCH(3, 12); break;
        default: PACK_LAUNCH(4, 12); break;
    }
#undef PACK_LAUNCH
    return hipGetLastError();
}

int asm_batch_pack_async(asm_handle* h, asm_batch* b) {
    if (!h || !b) return fail(h, ASM_EINVAL, "asm_batch_pack_async: NULL argument");
    if (b->n == 0) return ASM_OK;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, launch_pack(h, b, b->d_tails, b->d_planes, b->d_lens, b->pb, b->d_pos));
    return ASM_OK;
}

/* ASM_GREEDY_SEQUENTIAL: derive the stale buffer tails on the device (asm_tails.h).  The resolver walks the pairs
 * in INPUT order and only needs granule 0, so it works on a temporary clean-mode, unbucketed, one-granule packing.
 * init256 (host, optional): buffer codes before the batch's first pair; summary256 (host, optional): the batch's effect on
 * the buffers (see tails_carry_kernel); emit = false stops after the summary (b->d_tails untouched). */
static int batch_resolve_tails(asm_handle* h, asm_batch* b, const uint8_t* init256, uint8_t* summary256, bool emit) {
    if (summary256) memset(summary256, TAIL_NONE, 256);
    if (b->n == 0) return ASM_OK;
    const long nchunks = (b->n + TAIL_CHUNK - 1) / TAIL_CHUNK;
    const long ngroups = (nchunks + TAIL_GROUP - 1) / TAIL_GROUP;
    TailState init;
    if (init256) memcpy(init.code, init256, 256);
    else memset(init.code, 0, 256);
    /* scratch lives with the batch: a streamed file re-resolves every chunk, a bench step every iteration */
    if (emit && !b->d_tails) HIPCHK(h, pool_alloc(h, (void**)&b->d_tails, sizeof(uint4) * 4 * (size_t)b->n));
    const size_t loc_bytes = (size_t)ngroups * TAIL_GROUP * 2 * sizeof(TailOp), grp_bytes = (size_t)ngroups * 2 * sizeof(TailOp);
    const size_t carry_bytes = (size_t)ngroups * 2 * 2 * sizeof(uint4);
    if (!b->d_tail_g0) {
        HIPCHK(h, pool_alloc(h, (void**)&b->d_tail_g0, sizeof(uint4) * 4 * (size_t)b->n));
        HIPCHK(h, pool_alloc(h, (void**)&b->d_tail_l0, sizeof(uint32_t) * (size_t)b->n));
        HIPCHK(h, pool_alloc(h, (void**)&b->d_tail_chunks, loc_bytes + grp_bytes + carry_bytes + 256)); /* loc, grp, gcarry, summary[256] */
    }
    TailOp* const d_loc = (TailOp*)b->d_tail_chunks;
    TailOp* const d_grp = (TailOp*)(b->d_tail_chunks + loc_bytes);
    uint4* const d_carry = (uint4*)(b->d_tail_chunks + loc_bytes + grp_bytes);
    uint8_t* const d_sum = b->d_tail_chunks + loc_bytes + grp_bytes + carry_bytes;
    PackBuckets one{};
    one.nb = 1, one.w4[0] = 1, one.start[0] = 0, one.start[1] = b->n, one.plane_off[0] = 0;
    HIPCHK(h, launch_pack(h, b, nullptr, b->d_tail_g0, b->d_tail_l0, one, nullptr));
    hipLaunchKernelGGL(tails_chunk_kernel, dim3((unsigned)ngroups), dim3(2 * TAIL_GROUP), 0, h->stream, b->d_tail_g0, b->d_tail_l0,
                       (long)b->n, d_loc, d_grp);
    hipLaunchKernelGGL(tails_carry_kernel, dim3(1), dim3(8 * TAIL_CARRY_SEGS), 0, h->stream, (const uint32_t*)d_grp, (uint32_t*)d_carry,
                       ngroups, init, summary256 ? d_sum : (uint8_t*)nullptr);
    if (emit)
        hipLaunchKernelGGL(tails_emit_kernel, dim3((unsigned)ngroups), dim3(2 * TAIL_GROUP), 0, h->stream, b->d_tail_g0, b->d_tail_l0,
                           (long)b->n, (const TailOp*)d_loc, (const uint4*)d_carry, b->d_tails);
    HIPCHK(h, hipGetLastError());
    if (summary256) {
        HIPCHK(h, hipMemcpyAsync(summary256, d_sum, 256, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return ASM_OK;
}

/* Common tail of upload/generate once ASCII + offsets are resident and b->maxlen is known: group the pairs into width
 * classes (only when the batch really mixes classes), allocate the packed form, resolve tails, pack. */
static int batch_finish(asm_handle* h, asm_batch* b) {
    const int64_t n = b->n;
    const int wmax = b->maxlen <= 128 ? 1 : (b->maxlen + 127) / 128;
    unsigned int counts[4] = {0, 0, 0, 0};
    uint8_t *d_cls = nullptr, *d_cls2 = nullptr;
    uint32_t* d_idx = nullptr;
    unsigned int* d_counts = nullptr;
    void* d_tmp = nullptr;
    int rc = ASM_OK;
    do {
#define TRY(call)                                                        \
    if ((call) != hipSuccess) {                                          \
        rc = fail(h, ASM_ENODEVICE, std::string(#call) + " failed");     \
        break;                                                           \
    }
        bool bucketed = false;
        if (wmax > 1 && n >= 4096 && h->bucketing) {
            TRY(pool_alloc(h, (void**)&d_cls, (size_t)n));
            TRY(pool_alloc(h, (void**)&d_cls2, (size_t)n));
            TRY(pool_alloc(h, (void**)&d_idx, sizeof(uint32_t) * (size_t)n));
            TRY(pool_alloc(h, (void**)&d_counts, 16));
            TRY(hipMemsetAsync(d_counts, 0, 16, h->stream));
            hipLaunchKernelGGL(classify_kernel, dim3(grid_for(n)), dim3(ASM_BLOCK), 0, h->stream, b->d_read_off, b->d_ref_off,
                               (long)n, d_cls, d_idx, d_counts);
            TRY(hipGetLastError());
            TRY(hipMemcpyAsync(counts, d_counts, 16, hipMemcpyDeviceToHost, h->stream));
            TRY(hipStreamSynchronize(h->stream));
            int classes = 0;
            for (int c = 0; c < 4; c++) classes += counts[c] ? 1 : 0;
            bucketed = classes > 1;
            if (bucketed) {
                TRY(pool_alloc(h, (void**)&b->d_order, sizeof(uint32_t) * (size_t)n));
                TRY(pool_alloc(h, (void**)&b->d_pos, sizeof(uint32_t) * (size_t)n));
                size_t tmp_bytes = 0;
                TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_cls, d_cls2, d_idx, b->d_order, (int)n, 0, 2,
                                                       h->stream));
                TRY(pool_alloc(h, &d_tmp, tmp_bytes + 16));
                TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_cls, d_cls2, d_idx, b->d_order, (int)n, 0, 2,
                                                       h->stream)); /* stable: input order is kept inside a class */
                hipLaunchKernelGGL(invert_order_kernel, dim3(grid_for(n)), dim3(ASM_BLOCK), 0, h->stream, b->d_order, (long)n,
                                   b->d_pos);
                TRY(hipGetLastError());
            }
        }
        // bucket table
        b->nb = 0;
        size_t plane_total = 0;
        int64_t slot = 0;
        b->pb = PackBuckets{};
        if (!bucketed) {
            counts[0] = counts[1] = counts[2] = counts[3] = 0;
            counts[wmax - 1] = (unsigned int)n;
        }
        for (int c = 0; c < 4; c++) {
            if (!counts[c] && !(n == 0 && c == 0)) continue;
            asm_bucket& k = b->bk[b->nb];
            k.n = counts[c];
            k.w4 = c + 1;
            k.maxlen = b->maxlen < 128 * (c + 1) ? b->maxlen : 128 * (c + 1);
            b->pb.w4[b->nb] = k.w4;
            b->pb.start[b->nb] = slot;
            b->pb.plane_off[b->nb] = (long)plane_total;
            plane_total += (size_t)4 * (size_t)k.w4 * (size_t)k.n;
            slot += k.n;
            b->nb++;
        }
        b->pb.nb = b->nb;
        b->pb.start[b->nb] = slot;
        b->planes_total = plane_total ? plane_total : 1;
        TRY(pool_alloc(h, (void**)&b->d_planes, sizeof(uint4) * (plane_total ? plane_total : 1)));
        TRY(pool_alloc(h, (void**)&b->d_lens, sizeof(uint32_t) * (size_t)(n > 0 ? n : 1)));
        for (int q = 0; q < b->nb; q++) {
            b->bk[q].planes = b->d_planes + b->pb.plane_off[q];
            b->bk[q].lens = b->d_lens + b->pb.start[q];
            b->bk[q].order = b->d_order ? b->d_order + b->pb.start[q] : nullptr;
            b->bk[q].mixed = bucketed;
        }
#undef TRY
        if (b->greedy_mode == ASM_GREEDY_SEQUENTIAL) rc = batch_resolve_tails(h, b, nullptr, nullptr, true);
        if (rc) break;
        rc = asm_batch_pack_async(h, b);
        if (rc) break;
        if (hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(h, ASM_ENODEVICE, "batch: stream synchronize failed");
    } while (0);
    pool_free(h, d_cls);
    pool_free(h, d_cls2);
    pool_free(h, d_idx);
    pool_free(h, d_counts);
    pool_free(h, d_tmp);
    return rc;
}

int asm_batch_upload(asm_handle* h, int64_t n, const char* reads, const uint32_t* read_off, const char* refs,
                     const uint32_t* ref_off, int greedy_mode, asm_batch** out) {
    if (!h || !out || n < 0 || !read_off || !ref_off || (n > 0 && (!reads || !refs)))
        return fail(h, ASM_EINVAL, "asm_batch_upload: bad arguments");
    if (greedy_mode != ASM_GREEDY_CLEAN && greedy_mode != ASM_GREEDY_SEQUENTIAL)
        return fail(h, ASM_EINVAL, "asm_batch_upload: unknown greedy_mode");
    *out = nullptr;
    HIPCHK(h, hipSetDevice(h->device));
    int maxlen = 0;
    for (int64_t i = 0; i < n; i++) {
        if (read_off[i + 1] < read_off[i] || ref_off[i + 1] < ref_off[i])
            return fail(h, ASM_EINVAL, "asm_batch_upload: offsets must be non-decreasing");
        int m = (int)(read_off[i + 1] - read_off[i]), nn = (int)(ref_off[i + 1] - ref_off[i]);
        if (m > maxlen) maxlen = m;
        if (nn > maxlen) maxlen = nn;
    }
    if (maxlen > ASM_MAX_LENGTH)
        return fail(h, ASM_EUNSUPPORTED, "asm_batch_upload: a sequence is longer than ASM_MAX_LENGTH");
    asm_batch* b = new asm_batch;
    b->owner = h, b->owner_serial = h->serial;
    b->n = n;
    b->maxlen = maxlen;
    b->greedy_mode = greedy_mode;
    b->reads_bytes = n ? read_off[n] : 0;
    b->refs_bytes = n ? ref_off[n] : 0;
    int rc = ASM_OK;
    do {
#define TRY(call)                                          \
    if ((call) != hipSuccess) {                            \
        rc = fail(h, ASM_ENOMEM, std::string(#call) + " failed"); \
        break;                                             \
    }
        TRY(pool_alloc(h, (void**)&b->d_reads, b->reads_bytes + 16));
        TRY(pool_alloc(h, (void**)&b->d_refs, b->refs_bytes + 16));
        TRY(pool_alloc(h, (void**)&b->d_read_off, sizeof(uint32_t) * (size_t)(n + 1)));
        TRY(pool_alloc(h, (void**)&b->d_ref_off, sizeof(uint32_t) * (size_t)(n + 1)));
        TRY(hipMemcpyAsync(b->d_reads, reads, b->reads_bytes, hipMemcpyHostToDevice, h->stream));
        TRY(hipMemcpyAsync(b->d_refs, refs, b->refs_bytes, hipMemcpyHostToDevice, h->stream));
        TRY(hipMemcpyAsync(b->d_read_off, read_off, sizeof(uint32_t) * (size_t)(n + 1), hipMemcpyHostToDevice, h->stream));
        TRY(hipMemcpyAsync(b->d_ref_off, ref_off, sizeof(uint32_t) * (size_t)(n + 1), hipMemcpyHostToDevice, h->stream));
#undef TRY
        rc = batch_finish(h, b);
    } while (0);
    if (rc) {
        batch_release(b);
        return rc;
    }
    *out = b;
    return ASM_OK;
}

int asm_batch_generate(asm_handle* h, const asm_gen_config* cfg, int64_t first, int64_t n, int greedy_mode,
                       asm_batch** out) {
    if (!h || !out || n < 0 || first < 0) return fail(h, ASM_EINVAL, "asm_batch_generate: bad arguments");
    int rc = check_gen(cfg);
    if (rc) return fail(h, rc, g_err);
    if (greedy_mode != ASM_GREEDY_CLEAN && greedy_mode != ASM_GREEDY_SEQUENTIAL)
        return fail(h, ASM_EINVAL, "asm_batch_generate: unknown greedy_mode");
    *out = nullptr;
    HIPCHK(h, hipSetDevice(h->device));
    asm_batch* b = new asm_batch;
    b->owner = h, b->owner_serial = h->serial;
    b->n = n;
    b->greedy_mode = greedy_mode;
    uint32_t *d_m = nullptr, *d_n = nullptr, *d_max = nullptr;
    void* d_tmp = nullptr;
    rc = ASM_OK;
    do {
#define TRY(call)                                                  \
    if ((call) != hipSuccess) {                                    \
        rc = fail(h, ASM_ENODEVICE, std::string(#call) + " failed"); \
        break;                                                     \
    }
        const size_t cnt = (size_t)n + 1;
        TRY(pool_alloc(h, (void**)&d_m, sizeof(uint32_t) * cnt));
        TRY(pool_alloc(h, (void**)&d_n, sizeof(uint32_t) * cnt));
        TRY(pool_alloc(h, (void**)&d_max, sizeof(uint32_t) * 2));
        TRY(pool_alloc(h, (void**)&b->d_read_off, sizeof(uint32_t) * cnt));
        TRY(pool_alloc(h, (void**)&b->d_ref_off, sizeof(uint32_t) * cnt));
        TRY(hipMemsetAsync(d_m, 0, sizeof(uint32_t) * cnt, h->stream));
        TRY(hipMemsetAsync(d_n, 0, sizeof(uint32_t) * cnt, h->stream));
        if (n > 0) {
            hipLaunchKernelGGL(gen_lengths_kernel, dim3(grid_for(n)), dim3(ASM_BLOCK), 0, h->stream, *cfg, (long)first,
                               (long)n, d_m, d_n);
            TRY(hipGetLastError());
        }
        size_t tmp_bytes = 0, t2 = 0;
        TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_m, b->d_read_off, (int)cnt, h->stream));
        TRY(hipcub::DeviceReduce::Max(nullptr, t2, d_n, d_max, (int)cnt, h->stream));
        if (t2 > tmp_bytes) tmp_bytes = t2;
        TRY(pool_alloc(h, &d_tmp, tmp_bytes + 16));
        TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_m, b->d_read_off, (int)cnt, h->stream));
        TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_n, b->d_ref_off, (int)cnt, h->stream));
        TRY(hipcub::DeviceReduce::Max(d_tmp, tmp_bytes, d_m, d_max, (int)cnt, h->stream));
        TRY(hipcub::DeviceReduce::Max(d_tmp, tmp_bytes, d_n, d_max + 1, (int)cnt, h->stream));
        uint32_t tot[2] = {0, 0}, mx[2] = {0, 0};
        TRY(hipMemcpyAsync(&tot[0], b->d_read_off + n, 4, hipMemcpyDeviceToHost, h->stream));
        TRY(hipMemcpyAsync(&tot[1], b->d_ref_off + n, 4, hipMemcpyDeviceToHost, h->stream));
        TRY(hipMemcpyAsync(mx, d_max, 8, hipMemcpyDeviceToHost, h->stream));
        TRY(hipStreamSynchronize(h->stream));
        /* 32-bit offsets: the host-side bound (n * worst length) must stay below 4 GiB */
        {
            double bound = (double)n * (double)(cfg->len_hi * 2 + 2);
            if (bound >= 4294967295.0 && (double)n * (double)(mx[0] > mx[1] ? mx[0] : mx[1]) >= 4294967295.0) {
                rc = fail(h, ASM_EUNSUPPORTED, "asm_batch_generate: batch exceeds 4 GiB of text; split it");
                break;
            }
        }
        b->reads_bytes = tot[0];
        b->refs_bytes = tot[1];
        b->maxlen = (int)(mx[0] > mx[1] ? mx[0] : mx[1]);
        if (b->maxlen > ASM_MAX_LENGTH) {
            rc = fail(h, ASM_EUNSUPPORTED, "asm_batch_generate: a generated sequence exceeds ASM_MAX_LENGTH");
            break;
        }
        TRY(pool_alloc(h, (void**)&b->d_reads, b->reads_bytes + 16));
        TRY(pool_alloc(h, (void**)&b->d_refs, b->refs_bytes + 16));
        if (n > 0) {
            hipLaunchKernelGGL(gen_fill_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, h->stream, *cfg, (long)first,
                               (long)n, b->d_read_off, b->d_ref_off, b->d_reads, b->d_refs);
            TRY(hipGetLastError());
        }
#undef TRY
        rc = batch_finish(h, b);
    } while (0);
    pool_free(h, d_m);
    pool_free(h, d_n);
    pool_free(h, d_max);
    pool_free(h, d_tmp);
    if (rc) {
        batch_release(b);
        return rc;
    }
    *out = b;
    return ASM_OK;
}

int asm_reference_upload(asm_handle* h, const char* text, size_t len, asm_reference** out) {
    if (!h || !out || (!text && len)) return fail(h, ASM_EINVAL, "asm_reference_upload: bad argument");
    *out = nullptr;
    HIPCHK(h, hipSetDevice(h->device));
    asm_reference* r = new asm_reference;
    r->len = len;
    if (big_malloc(h, (void**)&r->d_text, len + 16) != hipSuccess) {
        delete r;
        return fail(h, ASM_ENOMEM, "asm_reference_upload: hipMalloc failed");
    }
    if (len && hipMemcpy(r->d_text, text, len, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(r->d_text);
        delete r;
        return fail(h, ASM_ENODEVICE, "asm_reference_upload: copy failed");
    }
    *out = r;
    return ASM_OK;
}

int asm_reference_free(asm_handle* h, asm_reference* r) {
    if (h) (void)hipSetDevice(h->device);
    if (r) {
        (void)hipFree(r->d_text);
        delete r;
    }
    return ASM_OK;
}

int asm_batch_from_hits(asm_handle* h, const asm_reference* ref, int64_t n, const char* reads, const uint32_t* read_off,
                        const uint64_t* hit_pos, int greedy_mode, asm_batch** out) {
    if (!h || !ref || !out || n < 0 || !read_off || (n > 0 && (!reads || !hit_pos)))
        return fail(h, ASM_EINVAL, "asm_batch_from_hits: bad arguments");
    if (greedy_mode != ASM_GREEDY_CLEAN && greedy_mode != ASM_GREEDY_SEQUENTIAL)
        return fail(h, ASM_EINVAL, "asm_batch_from_hits: unknown greedy_mode");
    *out = nullptr;
    HIPCHK(h, hipSetDevice(h->device));
    int maxlen = 0;
    for (int64_t i = 0; i < n; i++) {
        if (read_off[i + 1] < read_off[i]) return fail(h, ASM_EINVAL, "asm_batch_from_hits: offsets must be non-decreasing");
        const int m = (int)(read_off[i + 1] - read_off[i]);
        maxlen = m > maxlen ? m : maxlen;
    }
    if (maxlen + 1 > ASM_MAX_LENGTH) return fail(h, ASM_EUNSUPPORTED, "asm_batch_from_hits: a read is longer than ASM_MAX_LENGTH - 1");
    asm_batch* b = new asm_batch;
    b->owner = h, b->owner_serial = h->serial;
    b->n = n;
    b->maxlen = maxlen + 1; /* the window is one base longer than the read (mapper/main.cpp:80) */
    b->greedy_mode = greedy_mode;
    b->reads_bytes = n ? read_off[n] : 0;
    unsigned long long* d_pos = nullptr;
    uint32_t* d_wl = nullptr;
    void* d_tmp = nullptr;
    int rc = ASM_OK;
    do {
#define TRY(call)                                                        \
    if ((call) != hipSuccess) {                                          \
        rc = fail(h, ASM_ENODEVICE, std::string(#call) + " failed");     \
        break;                                                           \
    }
        const size_t cnt = (size_t)n + 1;
        TRY(pool_alloc(h, (void**)&b->d_reads, b->reads_bytes + 16));
        TRY(pool_alloc(h, (void**)&b->d_read_off, sizeof(uint32_t) * cnt));
        TRY(pool_alloc(h, (void**)&b->d_ref_off, sizeof(uint32_t) * cnt));
        TRY(pool_alloc(h, (void**)&d_pos, sizeof(unsigned long long) * cnt));
        TRY(pool_alloc(h, (void**)&d_wl, sizeof(uint32_t) * cnt));
        TRY(hipMemcpyAsync(b->d_reads, reads, b->reads_bytes, hipMemcpyHostToDevice, h->stream));
        TRY(hipMemcpyAsync(b->d_read_off, read_off, sizeof(uint32_t) * cnt, hipMemcpyHostToDevice, h->stream));
        if (n) TRY(hipMemcpyAsync(d_pos, hit_pos, sizeof(unsigned long long) * (size_t)n, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(hit_window_lengths_kernel, dim3(grid_for(n + 1)), dim3(ASM_BLOCK), 0, h->stream, b->d_read_off,
                           d_pos, (unsigned long long)ref->len, (long)n, d_wl);
        TRY(hipGetLastError());
        size_t tmp_bytes = 0;
        TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_wl, b->d_ref_off, (int)cnt, h->stream));
        TRY(pool_alloc(h, &d_tmp, tmp_bytes + 16));
        TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_wl, b->d_ref_off, (int)cnt, h->stream));
        uint32_t total = 0;
        TRY(hipMemcpyAsync(&total, b->d_ref_off + n, 4, hipMemcpyDeviceToHost, h->stream));
        TRY(hipStreamSynchronize(h->stream));
        b->refs_bytes = total;
        TRY(pool_alloc(h, (void**)&b->d_refs, b->refs_bytes + 16));
        if (n) {
            int64_t blocks = (n + 3) / 4;
            blocks = blocks > 256 * 16 ? 256 * 16 : blocks;
            hipLaunchKernelGGL(hit_window_gather_kernel, dim3((unsigned)blocks), dim3(ASM_BLOCK), 0, h->stream, ref->d_text,
                               d_pos, b->d_ref_off, (long)n, b->d_refs);
            TRY(hipGetLastError());
        }
#undef TRY
        rc = batch_finish(h, b);
    } while (0);
    pool_free(h, d_pos);
    pool_free(h, d_wl);
    pool_free(h, d_tmp);
    if (rc) {
        batch_release(b);
        return rc;
    }
    *out = b;
    return ASM_OK;
}

/* ---- Greedy sequential mode across batches: shards of one file on several GPUs, or chunks of a streamed file ---- */
int asm_batch_tail_summary(asm_handle* h, const asm_batch* b, uint8_t* summary) {
    if (!h || !b || !summary) return fail(h, ASM_EINVAL, "asm_batch_tail_summary: NULL argument");
    HIPCHK(h, hipSetDevice(h->device));
    return batch_resolve_tails(h, const_cast<asm_batch*>(b), nullptr, summary, false);
}

int asm_tail_state_advance(uint8_t* state, const uint8_t* summary, int64_t n_pairs) {
    std::string err;
    const int rc = asm_host::tail_state_advance(state, summary, n_pairs, err);
    return rc ? fail(nullptr, rc, err) : ASM_OK;
}

int asm_batch_resolve_tails(asm_handle* h, asm_batch* b, const uint8_t* state) {
    if (!h || !b) return fail(h, ASM_EINVAL, "asm_batch_resolve_tails: NULL argument");
    if (state)
        for (int q = 0; q < 256; q++)
            if (state[q] > 3) return fail(h, ASM_EINVAL, "asm_batch_resolve_tails: state entries are 2-bit codes (0..3)");
    HIPCHK(h, hipSetDevice(h->device));
    b->greedy_mode = ASM_GREEDY_SEQUENTIAL;
    int rc = batch_resolve_tails(h, b, state, nullptr, true);
    if (!rc) rc = asm_batch_pack_async(h, b);
    return rc;
}

int asm_batch_free(asm_handle* h, asm_batch* b) {
    if (!b) return ASM_OK;
    /* The device blocks belong to the pool of the handle that created the batch, whichever handle is named here (h may be NULL):
     * they go back THERE.  The owner is named by (address, serial): if that handle is gone — asm_destroy has released every
     * block of its pool, this batch's included — only the record is left, also when a NEW handle lives at the old address.
     * Look-up and release happen under one lock, so the owner cannot be destroyed in between. */
    (void)h;
    std::lock_guard<std::mutex> lk(g_live_mu);
    if (!owner_is_live_locked(b->owner, b->owner_serial)) {
        for (hipEvent_t ev : b->ev_consumed) /* events belong to the device context, not to the handle */
            if (ev) (void)hipEventDestroy(ev);
        delete b;
        return ASM_OK;
    }
    (void)hipSetDevice(b->owner->device);
    batch_release(b);
    return ASM_OK;
}

int64_t asm_batch_size(const asm_batch* b) { return b ? b->n : 0; }
int asm_batch_max_length(const asm_batch* b) { return b ? b->maxlen : 0; }
int64_t asm_batch_text_bytes(const asm_batch* b) { return b ? (int64_t)(b->reads_bytes + b->refs_bytes) : 0; }

int asm_batch_download(asm_handle* h, const asm_batch* b, uint32_t* read_off, uint32_t* ref_off, char* reads,
                       size_t reads_cap, char* refs, size_t refs_cap) {
    if (!h || !b) return fail(h, ASM_EINVAL, "asm_batch_download: NULL argument");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (read_off) HIPCHK(h, hipMemcpy(read_off, b->d_read_off, sizeof(uint32_t) * (size_t)(b->n + 1), hipMemcpyDeviceToHost));
    if (ref_off) HIPCHK(h, hipMemcpy(ref_off, b->d_ref_off, sizeof(uint32_t) * (size_t)(b->n + 1), hipMemcpyDeviceToHost));
    if (reads) {
        if (reads_cap < b->reads_bytes) return fail(h, ASM_EINVAL, "asm_batch_download: reads buffer too small");
        HIPCHK(h, hipMemcpy(reads, b->d_reads, b->reads_bytes, hipMemcpyDeviceToHost));
    }
    if (refs) {
        if (refs_cap < b->refs_bytes) return fail(h, ASM_EINVAL, "asm_batch_download: refs buffer too small");
        HIPCHK(h, hipMemcpy(refs, b->d_refs, b->refs_bytes, hipMemcpyDeviceToHost));
    }
    return ASM_OK;
}

/* ---------------------------------------------------------------------------------------------------- */
static int check_params(asm_handle* h, int aligner, const asm_params* p, int maxlen = ASM_MAX_LENGTH) {
    if (!p) return fail(h, ASM_EINVAL, "params is NULL");
    if (p->x < 0 || p->o < 0 || p->e < 0) return fail(h, ASM_EINVAL, "penalties must be non-negative");
    if (aligner == ASM_GREEDY) {
        if (p->k < 0 || p->k > ASM_GREEDY_MAX_K) return fail(h, ASM_EINVAL, "Greedy: k must be in [0, 50] (MAX_K, hurdle_matrix.h:8)");
        if (!(p->p_match > 0 && p->p_mismatch > 0 && p->p_indel > 0)) return fail(h, ASM_EINVAL, "Greedy: probabilities must be positive");
        if (p->alignment_type != ASM_ALIGN_GLOBAL && p->alignment_type != ASM_ALIGN_SEMI_GLOBAL)
            return fail(h, ASM_EUNSUPPORTED, "Greedy: alignment type must be GLOBAL or SEMI_GLOBAL (LOCAL is unsupported in the reference too, hurdle_matrix.h:467)");
    } else if (aligner == ASM_LEAP) {
        if (p->k < 0 || p->k > ASM_WIDE_MAX_K) return fail(h, ASM_EINVAL, "LEAP: k out of range");
        /* the recurrence reads generation e-o / e-ext / e-x: zero penalties would be same-generation reads,
         * and LV_BAG.cpp:165 assumes open >= extend */
        if (p->x < 1 || p->o < 1 || p->e < 1 || p->o < p->e)
            return fail(h, ASM_EINVAL, "LEAP: need x >= 1, o >= e >= 1 (LV_BAG.cpp:165-166)");
        if (p->x > ASM_WIDE_MAX_PENALTY || p->o > ASM_WIDE_MAX_PENALTY)
            return fail(h, ASM_EUNSUPPORTED, "LEAP: penalties above the compiled history depth");
        if (p->leap_mode < ASM_LEAP_GLOBAL || p->leap_mode > ASM_LEAP_SEMI_FREE_END)
            return fail(h, ASM_EINVAL, "LEAP: leap_mode must be one of ASM_LEAP_GLOBAL/LOCAL/SEMI_FREE_BEGIN/SEMI_FREE_END");
        if (p->leap_mode != ASM_LEAP_GLOBAL && maxlen > 512)
            return fail(h, ASM_EUNSUPPORTED, "LEAP: the non-GLOBAL modes are built for strings up to 512 characters");
    } else if (aligner == ASM_NW) {
        /* nw_affine_kernel keeps H/E/F below NW_BIG (int16 halves of one dword at the block boundary): every cell is at most
         * gap(i) + gap(j), and E/F one gap-open above that */
        const long worst = 2L * ((long)p->o + (long)(maxlen > 0 ? maxlen - 1 : 0) * (long)p->e) + (long)p->o;
        if (p->x > 10000 || p->o > 10000 || p->e > 10000 || worst >= (long)NW_BIG)
            return fail(h, ASM_EUNSUPPORTED, "NW: penalties too large for this batch's longest sequence "
                                             "(need 2*(o + (maxlen-1)*e) + o < 30000 and x, o, e <= 10000)");
    } else {
        return fail(h, ASM_EINVAL, "unknown aligner id");
    }
    return ASM_OK;
}

/* one aligner over one width class */
static int align_bucket(asm_handle* h, const asm_bucket& b, int aligner, const asm_params* p, OutMap out,
                        CigarSink cig = CigarSink{nullptr, nullptr, 0}, const int32_t* hint = nullptr) {
    if (b.n == 0) return ASM_OK;
    const bool unit = (p->x == 1 && p->o == 1 && p->e == 1);
    const uint4* planes = b.planes;
    const uint32_t* lens = b.lens;
    if (aligner == ASM_GREEDY) {
        GreedyArgs ga;
        ga.x = p->x, ga.o = p->o, ga.e = p->e;
        ga.semi = p->alignment_type == ASM_ALIGN_SEMI_GLOBAL ? 1 : 0;
        ga.sig_match = log(p->p_match / 0.25); /* hurdle_matrix.h:536-538 */
        ga.sig_mismatch = log(p->p_mismatch / 0.25);
        ga.sig_indel = log(p->p_indel / 2 / 0.25);
        switch (p->k) {
            case 1: HIPCHK(h, launch_greedy<1>(h, b, ga, out, cig)); break;
            case 2: HIPCHK(h, launch_greedy<2>(h, b, ga, out, cig)); break;
            case 3: HIPCHK(h, launch_greedy<3>(h, b, ga, out, cig)); break;
            case 4: HIPCHK(h, launch_greedy<4>(h, b, ga, out, cig)); break;
            case 5: HIPCHK(h, launch_greedy<5>(h, b, ga, out, cig)); break;
#define GREEDY_WIDE_CASE(KK) \
            case KK: HIPCHK(h, launch_greedy<KK>(h, b, ga, out, cig)); break;
            GREEDY_WIDE_CASE(6) GREEDY_WIDE_CASE(7) GREEDY_WIDE_CASE(8) GREEDY_WIDE_CASE(9) GREEDY_WIDE_CASE(10) GREEDY_WIDE_CASE(11)
            GREEDY_WIDE_CASE(12) GREEDY_WIDE_CASE(13) GREEDY_WIDE_CASE(14) GREEDY_WIDE_CASE(15) GREEDY_WIDE_CASE(16)
            /* K = 17, 18 still win at 100 bp (0.71, 0.79 ms against 0.90) but lose at 150 bp, err 0.20 (1.67, 1.83 against 1.60) */
#undef GREEDY_WIDE_CASE
            default:
                if (h->wave_kernels && p->k >= 32 && p->k <= 39 && (long)p->o + 110L * p->e < 16000L) {
                    /* 65..79 band lanes: sixteen threads per pair, five lanes each (asm_group.h): 1.78 ms per 10^6 C2 pairs
                     * against 2.25 ms for the two-wavefront kernel */
                    HIPCHK(h, launch_greedy_group(h->stream, planes, lens, b.n, b.w4, (int)p->k, ga, out, cig, h->num_cus));
                } else if (p->k <= ASM_WAVE_MAX_K && h->wave_kernels && unit && !ga.semi) {
                    HIPCHK(h, hipMemsetAsync(h->d_pair_queue, 0, sizeof(unsigned long long), h->stream));
                    launch_wave_per_pair(h->stream, greedy_wave_kernel<true>, b.n, h->num_cus, planes, lens, (long)b.n, b.w4,
                                         (int)p->k, ga, out, cig, h->d_pair_queue, (const uint32_t*)nullptr, (const uint32_t*)nullptr);
                } else if (p->k <= ASM_WAVE_MAX_K && h->wave_kernels && (long)p->o + 62L * p->e < 16000L) {
                    HIPCHK(h, hipMemsetAsync(h->d_pair_queue, 0, sizeof(unsigned long long), h->stream));
                    launch_wave_per_pair(h->stream, greedy_wave_kernel<false>, b.n, h->num_cus, planes, lens, (long)b.n, b.w4,
                                         (int)p->k, ga, out, cig, h->d_pair_queue, (const uint32_t*)nullptr, (const uint32_t*)nullptr);
                } else if (h->wave_kernels && (long)p->o + 100L * p->e < 16000L)
                    launch_greedy_wave2(h->stream, planes, lens, b.n, b.w4, (int)p->k, ga, out, cig, h->num_cus);
                else
                    launch_greedy_wide(h->stream, planes, lens, b.n, b.w4, p->k, ga, out, cig);
                break;
        }
    } else if (aligner == ASM_LEAP) {
        if (p->leap_mode != ASM_LEAP_GLOBAL) {
            /* LV's other ED_modes have no caller in the reference: one kernel serves them, the workgroup-per-pair form that takes
             * any band and any penalties (asm_wide.h) */
            launch_leap_wide(h->stream, planes, lens, b.n, b.w4, p->k, p->x, p->o, p->e, out, (int)p->leap_mode);
        } else if (unit && p->k >= 1 && ((p->k <= LEAP_UNIT_WIDE_K && b.maxlen <= 384) || (p->k <= LEAP_UNIT_MAX_K && b.maxlen <= 128))) {
            /* thread per pair, band lanes in registers: k <= 5 at any length; k = 6..10 for strings of one granule, where the
             * four-threads-per-pair kernel is 1.6-2 x slower (C2, 10^6 pairs, k = 6 / 8 / 10: 0.085 / 0.089 / 0.111 ms against
             * 0.174 / 0.181 / 0.180) */
            switch (p->k) {
                case 1: HIPCHK(h, launch_leap_unit<1>(h, b, out, hint)); break;
                case 2: HIPCHK(h, launch_leap_unit<2>(h, b, out, hint)); break;
                case 3: HIPCHK(h, launch_leap_unit<3>(h, b, out, hint)); break;
                case 4: HIPCHK(h, launch_leap_unit<4>(h, b, out, hint)); break;
                case 5: HIPCHK(h, launch_leap_unit<5>(h, b, out, hint)); break;
                case 6: HIPCHK(h, launch_leap_unit<6>(h, b, out, hint)); break;
                case 7: HIPCHK(h, launch_leap_unit<7>(h, b, out, hint)); break;
                case 8: HIPCHK(h, launch_leap_unit<8>(h, b, out, hint)); break;
                case 9: HIPCHK(h, launch_leap_unit<9>(h, b, out, hint)); break;
                default: HIPCHK(h, launch_leap_unit<10>(h, b, out, hint)); break;
            }
        } else if (!unit && p->k >= 1 && ((p->k <= 5 && b.maxlen <= 256) || (p->k <= 8 && b.maxlen <= 128)) && h->wave_kernels &&
                   RingGeometry(p->x, p->o, p->e).lds_bytes(2 * p->k + 1, LEAP_GEN_THREADS) <= 64 * 1024) {
            /* general penalties, narrow band: thread per pair with an LDS generation ring */
            switch (p->k) {
                case 1: HIPCHK(h, launch_leap_general<1>(h, b, p, out)); break;
                case 2: HIPCHK(h, launch_leap_general<2>(h, b, p, out)); break;
                case 3: HIPCHK(h, launch_leap_general<3>(h, b, p, out)); break;
                case 4: HIPCHK(h, launch_leap_general<4>(h, b, p, out)); break;
                case 5: HIPCHK(h, launch_leap_general<5>(h, b, p, out)); break;
                case 6: HIPCHK(h, launch_leap_general<6>(h, b, p, out)); break;
                case 7: HIPCHK(h, launch_leap_general<7>(h, b, p, out)); break;
                default: HIPCHK(h, launch_leap_general<8>(h, b, p, out)); break;
            }
        } else if (h->wave_kernels && b.maxlen <= 512 &&
                   leap_quad_lds((b.maxlen + 31) / 32, (int)p->k, unit ? 2 : RingGeometry(p->x, p->o, p->e).gm,
                                 unit ? 0 : RingGeometry(p->x, p->o, p->e).gi, b.maxlen + 2 <= 255 ? 1 : 2) <= 64 * 1024) {
            /* wide band: four threads per pair, generation rings and planes in LDS (asm_wave.h) */
            const RingGeometry rg(p->x, p->o, p->e);
            const int w32 = (b.maxlen + 31) / 32;
            const int32_t* const qhint = h->leap_hint ? hint : nullptr;
            /* long-running cases (long strings or general penalties): sort the whole bucket by the hint — one 6-bit radix pass —
             * and let every single-wave workgroup take sixteen neighbours of the order; short ones sort inside the workgroup */
            const uint32_t* qperm = nullptr;
            if (qhint && h->leap_sort && (b.maxlen > 128 || !unit)) {
                const size_t n = (size_t)b.n;
                size_t tmp_bytes = 0;
                HIPCHK(h, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (uint8_t*)nullptr, (uint8_t*)nullptr, (uint32_t*)nullptr,
                                                             (uint32_t*)nullptr, (int)n, 0, 6, h->stream));
                const size_t need = 2 * n + 8 * n + tmp_bytes + 256; /* keys, keys', idx, perm, temp */
                if (h->sort_cap < need) {
                    if (h->d_sort) (void)hipFree(h->d_sort);
                    h->d_sort = nullptr, h->sort_cap = 0;
                    HIPCHK(h, big_malloc(h, (void**)&h->d_sort, need));
                    h->sort_cap = need;
                }
                uint32_t* const d_idx = (uint32_t*)h->d_sort;
                uint32_t* const d_perm = d_idx + n;
                uint8_t* const d_keys = (uint8_t*)(d_perm + n);
                uint8_t* const d_keys2 = d_keys + n;
                void* const d_tmp = (void*)(((uintptr_t)(d_keys2 + n) + 255) & ~(uintptr_t)255);
                const int div = unit ? 1 : (int)(p->e < p->x ? p->e : p->x);
                hipLaunchKernelGGL(leap_sort_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, qhint, out, (long)n,
                                   div, d_keys, d_idx);
                HIPCHK(h, hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_keys, d_keys2, d_idx, d_perm, (int)n, 0, 6, h->stream));
                qperm = d_perm;
            }
#define LEAP_QUAD(W)                                                                                                          \
    HIPCHK(h, (b.maxlen + 2 <= 255 ? launch_leap_quad<W, uint8_t>(h->stream, planes, lens, b.n, b.w4, (int)p->k, unit, (int)p->x, \
                                                                  (int)p->o, (int)p->e, rg.gm, rg.gi, out, qhint, qperm)        \
                                   : launch_leap_quad<W, uint16_t>(h->stream, planes, lens, b.n, b.w4, (int)p->k, unit, (int)p->x, \
                                                                   (int)p->o, (int)p->e, rg.gm, rg.gi, out, qhint, qperm)))
            if (w32 <= 4) LEAP_QUAD(4);
            else if (w32 <= 5) LEAP_QUAD(5);
            else if (w32 <= 6) LEAP_QUAD(6);
            else if (w32 <= 8) LEAP_QUAD(8);
            else if (w32 <= 12) LEAP_QUAD(12);
            else LEAP_QUAD(16);
#undef LEAP_QUAD
        } else {
            launch_leap_wide(h->stream, planes, lens, b.n, b.w4, p->k, p->x, p->o, p->e, out);
        }
    } else {
        if (unit) {
            const dim3 g(grid_for(b.n)), t(ASM_BLOCK);
            const long n = (long)b.n;
            if (!h->nw_banded) {
                if (b.w4 == 1)
                    hipLaunchKernelGGL(nw_unit_kernel<2>, g, t, 0, h->stream, planes, lens, n, b.w4, out);
                else if (b.w4 == 2)
                    hipLaunchKernelGGL(nw_unit_kernel<4>, g, t, 0, h->stream, planes, lens, n, b.w4, out);
                else if (b.w4 == 3)
                    hipLaunchKernelGGL(nw_unit_kernel<6>, g, t, 0, h->stream, planes, lens, n, b.w4, out);
                else
                    hipLaunchKernelGGL(nw_unit_kernel<8>, g, t, 0, h->stream, planes, lens, n, b.w4, out);
            } else if (b.mixed && h->nw_bylen) { /* a width class of a mixed-length batch: workgroup-local sort by length */
                /* (measured and dropped in round 4: ONE wave per 256-pair sort window working through its four length quartiles in
                 * turn, so that every wave of the grid gets the same mix — C5, 10^7 pairs: 2.16-2.21 ms against 2.12 for this form;
                 * the dispatcher does not pile the long quartiles on one SIMD) */
                const dim3 g((unsigned)((b.n + NW_SORT_PAIRS - 1) / NW_SORT_PAIRS));
                if (b.w4 == 1)
                    hipLaunchKernelGGL((nw_banded_kernel<4, 32, true>), g, t, 0, h->stream, planes, lens, n, b.w4, out);
                else if (b.w4 == 2)
                    hipLaunchKernelGGL((nw_banded_kernel<8, 64, true>), g, t, 0, h->stream, planes, lens, n, b.w4, out);
                else if (b.w4 == 3)
                    hipLaunchKernelGGL((nw_banded_kernel<12, 64, true>), g, t, 0, h->stream, planes, lens, n, b.w4, out);
                else
                    hipLaunchKernelGGL((nw_banded_kernel<16, 64, true>), g, t, 0, h->stream, planes, lens, n, b.w4, out);
            } else if (b.w4 == 1)
                hipLaunchKernelGGL((nw_banded_kernel<4, 32, false>), g, t, 0, h->stream, planes, lens, n, b.w4, out);
            else if (b.w4 == 2)
                hipLaunchKernelGGL((nw_banded_kernel<8, 64, false>), g, t, 0, h->stream, planes, lens, n, b.w4, out);
            else if (b.w4 == 3)
                hipLaunchKernelGGL((nw_banded_kernel<12, 64, false>), g, t, 0, h->stream, planes, lens, n, b.w4, out);
            else
                hipLaunchKernelGGL((nw_banded_kernel<16, 64, false>), g, t, 0, h->stream, planes, lens, n, b.w4, out);
        } else {
            /* a zero penalty makes the wavefront read the generation it is writing (ring slot s - 0): plain Gotoh handles it */
            const bool positive = p->x >= 1 && p->o >= 1 && p->e >= 1;
            if (h->nw_wfa && positive && b.maxlen <= 256 &&
                WfaRings(p->x, p->o, p->e).lds_bytes(2 * NW_WFA_K + 1, LEAP_GEN_THREADS, 2) <= 64 * 1024) {
                const int rc = b.maxlen <= 128 ? launch_nw_wfa<2, 128>(h, b, p, out) : launch_nw_wfa<4, 256>(h, b, p, out);
                if (rc != ASM_OK) return rc;
            } else {
                launch_nw_affine(h->stream, planes, lens, b.n, b.w4, b.maxlen, p->x, p->o, p->e, out);
            }
        }
    }
    HIPCHK(h, hipGetLastError());
    return ASM_OK;
}

int asm_align_batch_hinted_async(asm_handle* h, const asm_batch* b, int aligner, const asm_params* p,
                                 const int32_t* d_work_hint, int32_t* d_penalties) {
    if (!h || !b || !d_penalties) return fail(h, ASM_EINVAL, "asm_align_batch_hinted_async: NULL argument");
    int rc = check_params(h, aligner, p, b->maxlen);
    if (rc) return rc;
    if (b->n == 0) return ASM_OK;
    HIPCHK(h, hipSetDevice(h->device));
    for (int q = 0; q < b->nb && !rc; q++) {
        OutMap out;
        out.out = d_penalties;
        out.order = b->bk[q].order;
        rc = align_bucket(h, b->bk[q], aligner, p, out, CigarSink{nullptr, nullptr, 0}, d_work_hint);
    }
    return rc;
}

int asm_align_batch_async(asm_handle* h, const asm_batch* b, int aligner, const asm_params* p, int32_t* d_penalties) {
    if (!h || !b || !d_penalties) return fail(h, ASM_EINVAL, "asm_align_batch_async: NULL argument");
    int rc = check_params(h, aligner, p, b->maxlen);
    if (rc) return rc;
    if (b->n == 0) return ASM_OK;
    HIPCHK(h, hipSetDevice(h->device));
    for (int q = 0; q < b->nb && !rc; q++) {
        OutMap out;
        out.out = d_penalties;
        out.order = b->bk[q].order;
        rc = align_bucket(h, b->bk[q], aligner, p, out);
    }
    return rc;
}

int asm_greedy_cigar_batch_async(asm_handle* h, const asm_batch* b, const asm_params* p, int32_t* d_penalties,
                                 uint16_t* d_ops, int cap, uint8_t* d_nops) {
    if (!h || !b || !d_penalties || !d_ops || !d_nops || cap < 1)
        return fail(h, ASM_EINVAL, "asm_greedy_cigar_batch_async: bad argument");
    int rc = check_params(h, ASM_GREEDY, p);
    if (rc) return rc;
    if (b->n == 0) return ASM_OK;
    HIPCHK(h, hipSetDevice(h->device));
    for (int q = 0; q < b->nb && !rc; q++) {
        OutMap out;
        out.out = d_penalties;
        out.order = b->bk[q].order;
        rc = align_bucket(h, b->bk[q], ASM_GREEDY, p, out, CigarSink{d_ops, d_nops, cap});
    }
    return rc;
}

int asm_cigar_format(const uint16_t* ops, int nops, int cap, char* out, size_t out_cap) {
    return asm_host::cigar_format(ops, nops, cap, out, out_cap);
}

int asm_coverage(asm_handle* h, const asm_batch* b, const asm_params* p, const uint16_t* d_greedy_ops, int greedy_cap,
                 const uint8_t* d_greedy_nops, int window, uint8_t* d_cover, uint16_t* d_nw_ops, int nw_cap,
                 uint8_t* d_nw_nops, unsigned long long* d_counters) {
    if (!h || !b || !p || !d_greedy_ops || !d_greedy_nops || !d_cover || !d_counters || greedy_cap < 1)
        return fail(h, ASM_EINVAL, "asm_coverage: bad argument");
    int prc = check_params(h, ASM_NW, p, b->maxlen);
    if (prc) return prc;
    if (window != 32 && window != 64) return fail(h, ASM_EINVAL, "asm_coverage: window must be 32 or 64");
    if (d_nw_ops && (!d_nw_nops || nw_cap < 1)) return fail(h, ASM_EINVAL, "asm_coverage: bad NW CIGAR buffers");
    if (b->n == 0) return ASM_OK;
    HIPCHK(h, hipSetDevice(h->device));
    CoverArgs ca;
    ca.g_ops = d_greedy_ops, ca.g_nops = d_greedy_nops, ca.g_cap = greedy_cap;
    ca.cover = d_cover, ca.nw_ops = d_nw_ops, ca.nw_nops = d_nw_nops, ca.nw_cap = nw_cap, ca.counters = d_counters;
    int rc = ASM_OK;
    const bool unit = p->x == 1 && p->o == 1 && p->e == 1;
    for (int q = 0; q < b->nb && !rc; q++) {
        const asm_bucket& k = b->bk[q];
        if (!unit) { /* general penalties: every pair through the full matrix */
            rc = cover_full_matrix(h, k, p, k.planes, k.lens, k.order, 0, nullptr, k.n, ca);
            continue;
        }
#define COVER(ND) rc = window == 32 ? cover_bucket<ND, 32>(h, k, p, ca) : cover_bucket<ND, 64>(h, k, p, ca)
        switch (k.w4) {
            case 1: COVER(4); break;
            case 2: COVER(8); break;
            case 3: COVER(12); break;
            default: COVER(16); break;
        }
#undef COVER
    }
    return rc;
}

int asm_align_batch(asm_handle* h, int aligner, int64_t n, const char* reads, const uint32_t* read_off, const char* refs,
                    const uint32_t* ref_off, const asm_params* p, int greedy_mode, int32_t* penalties) {
    if (!h || !penalties) return fail(h, ASM_EINVAL, "asm_align_batch: NULL argument");
    int rc = check_params(h, aligner, p);
    if (rc) return rc;
    asm_batch* b = nullptr;
    rc = asm_batch_upload(h, n, reads, read_off, refs, ref_off, greedy_mode, &b);
    if (rc) return rc;
    int32_t* d_out = nullptr;
    if (pool_alloc(h, (void**)&d_out, sizeof(int32_t) * (size_t)(n > 0 ? n : 1)) != hipSuccess) {
        batch_release(b);
        return fail(h, ASM_ENOMEM, "asm_align_batch: hipMalloc failed");
    }
    rc = asm_align_batch_async(h, b, aligner, p, d_out);
    if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(h, ASM_ENODEVICE, "asm_align_batch: kernel failed");
    if (!rc && n > 0 && hipMemcpy(penalties, d_out, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(h, ASM_ENODEVICE, "asm_align_batch: copy back failed");
    pool_free(h, d_out);
    batch_release(b);
    return rc;
}

int asm_simd_ed_batch_async(asm_handle* h, const asm_batch* b, int ed_threshold, int shd_enable, int mode,
                            int32_t* state, int32_t* d_ed) {
    return asm_simd_ed_mode_batch_async(h, b, ed_threshold, shd_enable, mode, ASM_LEAP_GLOBAL, state, d_ed);
}

int asm_simd_ed_mode_batch_async(asm_handle* h, const asm_batch* b, int ed_threshold, int shd_enable, int mode, int ed_mode,
                                 int32_t* state, int32_t* d_ed) {
    if (!h || !b || !d_ed) return fail(h, ASM_EINVAL, "asm_simd_ed_batch_async: NULL argument");
    if (ed_mode < ASM_LEAP_GLOBAL || ed_mode > ASM_LEAP_SEMI_FREE_END)
        return fail(h, ASM_EINVAL, "asm_simd_ed_mode_batch_async: ed_mode must be one of ASM_LEAP_GLOBAL/LOCAL/SEMI_FREE_BEGIN/SEMI_FREE_END");
    if (ed_threshold < 1 || ed_threshold > ASM_FILTER_MAX_T)
        return fail(h, ASM_EINVAL, "asm_simd_ed_batch_async: ED threshold must be in [1, 32]");
    if (shd_enable && ed_threshold > ASM_SHD_MAX_ERROR)
        return fail(h, ASM_EINVAL, "asm_simd_ed_batch_async: SHD needs an ED threshold <= 16 (MAX_ERROR_AVX)");
    if (mode != ASM_FILTER_SEQUENTIAL && mode != ASM_FILTER_CLEAN)
        return fail(h, ASM_EINVAL, "asm_simd_ed_batch_async: unknown mode");
    if (mode == ASM_FILTER_SEQUENTIAL && state && (state[0] < 0 || state[0] > 255 || state[1] < 0 || state[1] > 255 || state[2] < 0))
        return fail(h, ASM_EINVAL, "asm_simd_ed_batch_async: state out of range");
    if (b->n == 0) return ASM_OK;
    HIPCHK(h, hipSetDevice(h->device));
    int rc = ASM_OK;
    for (int q = 0; q < b->nb && !rc; q++) {
        OutMap ev;
        ev.out = d_ed;
        ev.order = b->bk[q].order;
        rc = b->bk[q].maxlen <= 128 ? launch_simd_ed_events<2>(h, b->bk[q], ed_threshold, shd_enable ? 1 : 0, ev, ed_mode)
                                    : launch_simd_ed_events<4>(h, b->bk[q], ed_threshold, shd_enable ? 1 : 0, ev, ed_mode);
    }
    if (rc) return rc;
    const long n = (long)b->n;
    const dim3 g(grid_for(b->n)), t(ASM_BLOCK);
    if (ed_mode == ASM_LEAP_LOCAL || ed_mode == ASM_LEAP_SEMI_FREE_END) { /* no converge_ED, no carried state: SEQUENTIAL = CLEAN */
        hipLaunchKernelGGL(simd_ed_final_kernel, g, t, 0, h->stream, d_ed, n);
        HIPCHK(h, hipGetLastError());
        return ASM_OK;
    }
    if (mode == ASM_FILTER_CLEAN) {
        hipLaunchKernelGGL(simd_ed_clean_kernel, g, t, 0, h->stream, d_ed, n, ed_threshold);
        HIPCHK(h, hipGetLastError());
        return ASM_OK;
    }
    /* sequential: resolve the verdict chain in batch order with two last-setter scans */
    const int init_fe = state ? state[0] : 0, init_fd = state ? state[1] : 0, init_conv = state ? state[2] : 0;
    int32_t *d_a = nullptr, *d_b = nullptr;
    int32_t last_setter = -1, last_conv = -1;
    void* d_tmp = nullptr;
    size_t tmp_bytes = 0;
    do {
#define TRY(call)                                                  \
    if ((call) != hipSuccess) {                                    \
        rc = fail(h, ASM_ENODEVICE, std::string(#call) + " failed"); \
        break;                                                     \
    }
        TRY(big_malloc(h, (void**)&d_a, sizeof(int32_t) * (size_t)n));
        TRY(big_malloc(h, (void**)&d_b, sizeof(int32_t) * (size_t)n));
        TRY(hipcub::DeviceScan::InclusiveScan(nullptr, tmp_bytes, d_a, d_b, LastSetter(), (int)n, h->stream));
        TRY(big_malloc(h, &d_tmp, tmp_bytes + 16));
        hipLaunchKernelGGL(simd_ed_setter_kernel, g, t, 0, h->stream, (const int32_t*)d_ed, n, d_a);
        TRY(hipcub::DeviceScan::InclusiveScan(d_tmp, tmp_bytes, d_a, d_b, LastSetter(), (int)n, h->stream));
        TRY(hipMemcpyAsync(&last_setter, d_b + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        hipLaunchKernelGGL(simd_ed_converge_kernel, g, t, 0, h->stream, (const int32_t*)d_ed, (const int32_t*)d_b, n, init_fe,
                           init_fd, d_a);
        TRY(hipcub::DeviceScan::InclusiveScan(d_tmp, tmp_bytes, d_a, d_b, LastSetter(), (int)n, h->stream));
        TRY(hipMemcpyAsync(&last_conv, d_b + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        hipLaunchKernelGGL(simd_ed_verdict_kernel, g, t, 0, h->stream, d_ed, (const int32_t*)d_b, n, ed_threshold, init_conv);
        TRY(hipGetLastError());
        TRY(hipStreamSynchronize(h->stream)); /* the scratch below is freed on return */
        if (state) { /* the state the next batch of the same stream of pairs starts from */
            if (last_setter >= 0) state[0] = last_setter & 0xff, state[1] = last_setter >> 8;
            if (last_conv >= 0) state[2] = last_conv;
        }
#undef TRY
    } while (0);
    if (d_a) (void)hipFree(d_a);
    if (d_b) (void)hipFree(d_b);
    if (d_tmp) (void)hipFree(d_tmp);
    return rc;
}

static int simd_ed_affine_launch(asm_handle* h, const asm_batch* b, int gap_threshold, int af_threshold, int x, int o, int e, int mode,
                                 int32_t* d_ed);
int asm_simd_ed_affine_batch_async(asm_handle* h, const asm_batch* b, int gap_threshold, int af_threshold, int x, int o, int e,
                                   int32_t* d_ed) {
    return simd_ed_affine_launch(h, b, gap_threshold, af_threshold, x, o, e, ASM_LEAP_GLOBAL, d_ed);
}
static int simd_ed_affine_launch(asm_handle* h, const asm_batch* b, int gap_threshold, int af_threshold, int x, int o, int e, int mode,
                                 int32_t* d_ed) {
    if (!h || !b || !d_ed) return fail(h, ASM_EINVAL, "asm_simd_ed_affine_batch_async: NULL argument");
    if (mode < ASM_LEAP_GLOBAL || mode > ASM_LEAP_SEMI_FREE_END)
        return fail(h, ASM_EINVAL, "asm_simd_ed_affine_mode_batch_async: mode must be one of ASM_LEAP_GLOBAL/LOCAL/SEMI_FREE_BEGIN/SEMI_FREE_END");
    if (gap_threshold < 1 || gap_threshold > ASM_FILTER_MAX_T)
        return fail(h, ASM_EINVAL, "asm_simd_ed_affine_batch_async: gap threshold must be in [1, 32]");
    if (af_threshold < 1 || af_threshold > 512)
        return fail(h, ASM_EINVAL, "asm_simd_ed_affine_batch_async: affine threshold must be in [1, 512]");
    if (x < 1 || x > 15 || o < 1 || o > 15 || e < 1 || e > o)
        return fail(h, ASM_EINVAL, "asm_simd_ed_affine_batch_async: penalties must satisfy 1 <= e <= o <= 15, 1 <= x <= 15");
    if (b->n == 0) return ASM_OK;
    HIPCHK(h, hipSetDevice(h->device));
    const RingGeometry rg(x, o, e);
    const int rows = 2 * gap_threshold + 3;
    int threads = 64;
    while (threads > 16 && (size_t)(rg.gm + 2 * rg.gi) * rows * threads * sizeof(uint16_t) > 150 * 1024) threads >>= 1;
    const size_t lds = (((size_t)(rg.gm + 2 * rg.gi) * rows * threads * sizeof(uint16_t)) + 3) & ~(size_t)3;
    if (lds > 150 * 1024) return fail(h, ASM_EINVAL, "asm_simd_ed_affine_batch_async: generation rings exceed the LDS");
    for (int q = 0; q < b->nb; q++) {
        const asm_bucket& k = b->bk[q];
        OutMap out;
        out.out = d_ed;
        out.order = k.order;
        const dim3 grid((unsigned)((k.n + threads - 1) / threads)), block((unsigned)threads);
        if (k.maxlen <= 128) { /* four threads per pair (thread per pair measured 0.28/0.71/3.5 ms per 10^6 C2 pairs at gap 3/8/30 against 0.29/0.46/0.79) */
            const size_t qlds = simd_quad_lds(gap_threshold, rg.gm, rg.gi);
            hipLaunchKernelGGL(simd_ed_affine_quad_kernel, dim3((unsigned)((k.n + 15) / 16)), dim3(64), qlds, h->stream, k.planes, k.lens,
                               (long)k.n, k.w4, gap_threshold, af_threshold, x, o, e, rg.gm, rg.gi, mode, out);
        } else if (k.maxlen <= 128) {
            if (lds > 64 * 1024)
                HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&simd_ed_affine_kernel<2>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(simd_ed_affine_kernel<2>, grid, block, lds, h->stream, k.planes, k.lens, (long)k.n, k.w4, gap_threshold,
                               af_threshold, x, o, e, rg.gm, rg.gi, mode, out);
        } else {
            if (lds > 64 * 1024)
                HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&simd_ed_affine_kernel<4>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(simd_ed_affine_kernel<4>, grid, block, lds, h->stream, k.planes, k.lens, (long)k.n, k.w4, gap_threshold,
                               af_threshold, x, o, e, rg.gm, rg.gi, mode, out);
        }
        HIPCHK(h, hipGetLastError());
    }
    return ASM_OK;
}

int asm_simd_ed_affine_shd_batch_async(asm_handle* h, const asm_batch* b, int gap_threshold, int af_threshold, int x, int o, int e,
                                       int shd_threshold, int32_t* d_ed) {
    if (shd_threshold < 0) return fail(h, ASM_EINVAL, "asm_simd_ed_affine_shd_batch_async: SHD threshold must be in [0, min(16, gap threshold)]");
    return asm_simd_ed_affine_mode_batch_async(h, b, gap_threshold, af_threshold, x, o, e, shd_threshold, ASM_LEAP_GLOBAL, d_ed);
}

int asm_simd_ed_affine_mode_batch_async(asm_handle* h, const asm_batch* b, int gap_threshold, int af_threshold, int x, int o, int e,
                                        int shd_threshold, int mode, int32_t* d_ed) {
    if (shd_threshold < 0) return simd_ed_affine_launch(h, b, gap_threshold, af_threshold, x, o, e, mode, d_ed); /* SHD off */
    if (shd_threshold > ASM_SHD_MAX_ERROR || shd_threshold > gap_threshold)
        return fail(h, ASM_EINVAL, "asm_simd_ed_affine_shd_batch_async: SHD threshold must be in [0, min(16, gap threshold)] "
                                   "(the reference reads 2*SHD_threshold+1 of its 2*gap_threshold+1 lane masks)");
    const int rc = simd_ed_affine_launch(h, b, gap_threshold, af_threshold, x, o, e, mode, d_ed);
    if (rc != ASM_OK || b->n == 0) return rc;
    for (int q = 0; q < b->nb; q++) {
        const asm_bucket& k = b->bk[q];
        OutMap out;
        out.out = d_ed;
        out.order = k.order;
        const dim3 g(grid_for(k.n)), t(ASM_BLOCK);
        if (k.maxlen <= 128)
            hipLaunchKernelGGL(simd_affine_shd_kernel<2>, g, t, 0, h->stream, k.planes, k.lens, (long)k.n, k.w4, gap_threshold,
                               shd_threshold, out);
        else
            hipLaunchKernelGGL(simd_affine_shd_kernel<4>, g, t, 0, h->stream, k.planes, k.lens, (long)k.n, k.w4, gap_threshold,
                               shd_threshold, out);
    }
    HIPCHK(h, hipGetLastError());
    return ASM_OK;
}

int asm_shd_filter_batch_async(asm_handle* h, const asm_batch* b, int max_error, int32_t* d_pass) {
    if (!h || !b || !d_pass) return fail(h, ASM_EINVAL, "asm_shd_filter_batch_async: NULL argument");
    if (max_error < 0 || max_error > ASM_SHD_MAX_ERROR)
        return fail(h, ASM_EINVAL, "asm_shd_filter_batch_async: max_error must be in [0, 16] (MAX_ERROR_AVX)");
    if (b->n == 0) return ASM_OK;
    HIPCHK(h, hipSetDevice(h->device));
    for (int q = 0; q < b->nb; q++) {
        const asm_bucket& k = b->bk[q];
        OutMap out;
        out.out = d_pass;
        out.order = k.order;
        const dim3 g(grid_for(k.n)), t(ASM_BLOCK);
        if (k.maxlen <= 128)
            hipLaunchKernelGGL(shd_kernel<2>, g, t, 0, h->stream, k.planes, k.lens, (long)k.n, k.w4, max_error, out);
        else
            hipLaunchKernelGGL(shd_kernel<4>, g, t, 0, h->stream, k.planes, k.lens, (long)k.n, k.w4, max_error, out);
    }
    HIPCHK(h, hipGetLastError());
    return ASM_OK;
}

int asm_count_equal_async(asm_handle* h, const int32_t* d_a, const int32_t* d_b, int64_t n, unsigned long long* d_count) {
    if (!h || !d_a || !d_b || !d_count) return fail(h, ASM_EINVAL, "asm_count_equal_async: NULL argument");
    if (n <= 0) return ASM_OK;
    HIPCHK(h, hipSetDevice(h->device));
    int64_t blocks = (n + ASM_BLOCK - 1) / ASM_BLOCK;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(count_equal_kernel, dim3((unsigned)blocks), dim3(ASM_BLOCK), 0, h->stream, d_a, d_b, (long)n, d_count);
    HIPCHK(h, hipGetLastError());
    return ASM_OK;
}

int asm_accuracy_async(asm_handle* h, const int32_t* d_nw, const int32_t* d_leap, const int32_t* d_greedy,
                       const int32_t* d_answers, int64_t n, unsigned long long* d_counters) {
    if (!h || !d_counters) return fail(h, ASM_EINVAL, "asm_accuracy_async: NULL argument");
    if (n <= 0) return ASM_OK;
    if ((((uintptr_t)d_nw | (uintptr_t)d_leap | (uintptr_t)d_greedy | (uintptr_t)d_answers) & 15u) != 0)
        return fail(h, ASM_EINVAL, "asm_accuracy_async: penalty arrays must be 16-byte aligned");
    HIPCHK(h, hipSetDevice(h->device));
    int64_t blocks = (n / 4 + ASM_BLOCK - 1) / ASM_BLOCK;
    /* every workgroup ends with three atomics on one cache line, and a hot line takes only ~90 atomics/us: 977 workgroups
     * made this kernel 38 us at 10^6 pairs (rocprof, round 1); 128 grid-striding workgroups keep the loads wide enough */
    blocks = blocks < 1 ? 1 : (blocks > 128 ? 128 : blocks);
    if (!d_nw && !d_answers) blocks = 1; /* nothing to compare with: total_tests only */
    hipLaunchKernelGGL(accuracy_kernel, dim3((unsigned)blocks), dim3(ASM_BLOCK), 0, h->stream, d_nw, d_leap, d_greedy,
                       d_answers, (long)n, d_counters);
    HIPCHK(h, hipGetLastError());
    return ASM_OK;
}

int asm_profile_enable(asm_handle* h, int max_calls, unsigned kernel_mask) {
    if (!h || max_calls < 0) return fail(h, ASM_EINVAL, "asm_profile_enable: bad argument");
    h->prof_select = kernel_mask & 0xfu;
    HIPCHK(h, hipSetDevice(h->device));
    for (hipEvent_t ev : h->prof_ev) (void)hipEventDestroy(ev);
    h->prof_ev.clear();
    h->prof_mask.clear();
    h->prof_cap = 0;
    for (int i = 0; i < 8 * max_calls; i++) {
        hipEvent_t ev;
        HIPCHK(h, hipEventCreate(&ev));
        h->prof_ev.push_back(ev);
    }
    h->prof_cap = max_calls;
    return ASM_OK;
}

int asm_profile_read(asm_handle* h, float* ms, int cap_calls, int* n_calls) {
    if (!h || !n_calls || (cap_calls > 0 && !ms)) return fail(h, ASM_EINVAL, "asm_profile_read: NULL argument");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->side_stream) HIPCHK(h, hipStreamSynchronize(h->side_stream));
    if (h->pack_stream) HIPCHK(h, hipStreamSynchronize(h->pack_stream));
    if (h->acc_stream) HIPCHK(h, hipStreamSynchronize(h->acc_stream));
    const int n = (int)h->prof_mask.size();
    *n_calls = n;
    for (int c = 0; c < n && c < cap_calls; c++)
        for (int q = 0; q < 4; q++) {
            float v = -1.0f;
            if (h->prof_mask[(size_t)c] & (1u << q))
                HIPCHK(h, hipEventElapsedTime(&v, h->prof_ev[(size_t)(8 * c + 2 * q)], h->prof_ev[(size_t)(8 * c + 2 * q + 1)]));
            ms[4 * c + q] = v;
        }
    return ASM_OK;
}

int asm_run_benchmark_async(asm_handle* h, asm_batch* b, const asm_params* p, int repack, int32_t* d_nw,
                            int32_t* d_leap, int32_t* d_greedy, const int32_t* d_answers,
                            unsigned long long* d_counters) {
    if (!h || !b || !p) return fail(h, ASM_EINVAL, "asm_run_benchmark_async: NULL argument");
    int rc = ASM_OK;
    // Without NW in the mask, wide-band LEAP (four threads per pair, asm_wave.h) is scheduled by the Greedy penalties:
    // Greedy first, on the same stream — at wide bands both kernels are VALU-bound and side by side they gain nothing (C3:
    // 23.8 ms against 23.6 in a row), while the work-sorted LEAP saves a quarter of its time.
    const bool greedy_first_shape = d_greedy && d_leap && !d_nw && p->k > 5 && h->leap_hint && h->wave_kernels;
    /* A call that asks for overlapped steps but has the Greedy-first shape (or no pairs) runs as a pipelined-pack call: it is
     * ordered like one (behind every earlier overlapped call), and the bookkeeping of the overlapped form starts afresh. */
    if (repack == 3 && (b->n <= 0 || greedy_first_shape)) repack = 2;
    // optional per-kernel timing inside the caller's timed region: events on the stream each kernel is launched on
    hipEvent_t* pe = nullptr;
    unsigned pmask = 0u;
    if ((int)h->prof_mask.size() < h->prof_cap) pe = &h->prof_ev[8 * h->prof_mask.size()];
#define PROF(q, which, stream_)                                              \
    if (pe && !rc && (h->prof_select & (1u << (q)))) {                       \
        HIPCHK(h, hipEventRecord(pe[2 * (q) + (which)], (stream_)));         \
        pmask |= 1u << (q);                                                  \
    }
    hipStream_t main_stream = h->stream;
    /* the contract of repack = 3: consecutive overlapped calls write different arrays (the previous call's counters may still be
     * reading its own).  A caller that forgets is told so instead of getting a race. */
    if (repack == 3 && h->last3_valid && ((d_nw && d_nw == h->last3_out[0]) || (d_leap && d_leap == h->last3_out[1]) ||
                                          (d_greedy && d_greedy == h->last3_out[2])))
        return fail(h, ASM_EINVAL, "asm_run_benchmark_async: repack = 3 needs output arrays that alternate between two sets (these "
                                   "were the previous call's); asm_pipeline_join_async first to reuse them");
    if (repack != 3) h->last3_valid = false, h->calls3 = 0;
    if (repack != 3 && h->tail_set) { /* earlier overlapped calls: everything of theirs before anything of this one */
        HIPCHK(h, hipSetDevice(h->device));
        HIPCHK(h, hipStreamWaitEvent(main_stream, h->ev_tail, 0));
        h->tail_set = false;
    }
    bool pipelined = false;
    if ((repack == 2 || repack == 3) && b->n > 0) {
        /* Pipelined repack: pack fills the OTHER set of planes on its own stream, so it runs beside the aligners of the previous
         * call (which read the current set) instead of behind them; this call's aligners wait for it.  The set it fills was last
         * read two calls ago (ev_consumed).  The caller guarantees that nothing enqueued since the previous call changes what
         * pack reads (the resident ASCII, the tails). */
        HIPCHK(h, hipSetDevice(h->device));
        if (!b->d_planes_alt) {
            HIPCHK(h, pool_alloc(h, (void**)&b->d_planes_alt, sizeof(uint4) * b->planes_total));
            HIPCHK(h, pool_alloc(h, (void**)&b->d_lens_alt, sizeof(uint32_t) * (size_t)b->n));
            for (hipEvent_t& ev : b->ev_consumed) HIPCHK(h, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            HIPCHK(h, hipEventRecord(b->ev_consumed[0], main_stream)); /* everything enqueued so far (the batch's creation) */
            HIPCHK(h, hipEventRecord(b->ev_consumed[1], main_stream));
        }
        const int nxt = b->cur ^ 1;
        std::swap(b->d_planes, b->d_planes_alt);
        std::swap(b->d_lens, b->d_lens_alt);
        b->cur = nxt;
        for (int q = 0; q < b->nb; q++) {
            b->bk[q].planes = b->d_planes + b->pb.plane_off[q];
            b->bk[q].lens = b->d_lens + b->pb.start[q];
        }
        HIPCHK(h, hipStreamWaitEvent(h->pack_stream, b->ev_consumed[nxt], 0));
        /* Overlapped calls on DIFFERENT batches (a caller rotating over several resident batches): the plane set of this batch
         * was consumed long ago, so nothing above holds the pack back, and the pack chain would run as many calls ahead as the
         * host has enqueued — thousands of short pack workgroups dispatched beside every persistent Greedy kernel (0.258 ms per
         * step against 0.218 on one batch).  Pace it as one batch paces itself: the pack of call c behind the counters of call
         * c - 2 (the event that also frees that call's output arrays). */
        if (repack == 3 && h->calls3 >= 2) HIPCHK(h, hipStreamWaitEvent(h->pack_stream, h->ev_out[h->calls3 & 1u], 0));
        if (!h->pipe_prev) { /* first of a run of pipelined calls: behind whatever the caller's stream holds so far */
            HIPCHK(h, hipEventRecord(h->ev_fork, main_stream));
            HIPCHK(h, hipStreamWaitEvent(h->pack_stream, h->ev_fork, 0));
        }
        /* ... and not before the previous call's persistent Greedy kernel is resident everywhere: pack's thousands of short
         * workgroups, dispatched at the same moment, keep Greedy's 122 KB-LDS workgroups off the CUs (0.280 ms/step); behind
         * the previous NW they find Greedy running and take the slots NW left (0.238).  With overlapped calls (repack = 3) the
         * pack chain runs a call ahead and meets no Greedy launch: no gate there (0.229 against 0.251 gated) */
        if (h->gate_set && repack == 2) HIPCHK(h, hipStreamWaitEvent(h->pack_stream, h->ev_gate, 0));
        PROF(0, 0, h->pack_stream)
        h->stream = h->pack_stream;
        rc = asm_batch_pack_async(h, b);
        h->stream = main_stream;
        PROF(0, 1, h->pack_stream)
        if (!rc) {
            HIPCHK(h, hipEventRecord(h->ev_packed, h->pack_stream));
            HIPCHK(h, hipStreamWaitEvent(main_stream, h->ev_packed, 0));
        }
        pipelined = true;
    } else if (repack) {
        PROF(0, 0, main_stream)
        rc = asm_batch_pack_async(h, b);
        PROF(0, 1, main_stream)
    }
    // Greedy depends only on the packed planes, NW -> LEAP form their own chain (LEAP is scheduled by the NW penalties):
    // run Greedy on a side stream so that the two chains fill each other's launch gaps and tail waves.
    const bool greedy_first = greedy_first_shape && !rc;
    if (greedy_first) {
        PROF(3, 0, main_stream)
        rc = asm_align_batch_async(h, b, ASM_GREEDY, p, d_greedy);
        PROF(3, 1, main_stream)
    }
    if (repack == 3 && pipelined && !rc) {
        h->last3_out[0] = d_nw, h->last3_out[1] = d_leap, h->last3_out[2] = d_greedy;
        h->last3_valid = true;
        /* OVERLAPPED calls: nothing of this call waits for the previous call's Greedy, and the caller's stream is not joined
         * here (asm_pipeline_join_async does that).  Three chains run through consecutive calls — NW -> LEAP -> counters -> NW
         * -> ... on a stream of the library, Greedy -> Greedy on the side stream, pack -> pack on the pack stream; a call's
         * counters wait for its Greedy, and their event also frees the call's plane set for the pack two calls later (and, with
         * the caller alternating output arrays, says those arrays may be written again). */
        /* the output arrays were last written two calls ago (the caller alternates): behind that call's counters.  For calls
         * on ONE batch ev_packed already implies it; calls on different batches have nothing else that orders them. */
        hipEvent_t out_free = h->ev_out[h->calls3 & 1u];
        if (h->calls3 >= 2) {
            HIPCHK(h, hipStreamWaitEvent(main_stream, out_free, 0));
            if (d_greedy) HIPCHK(h, hipStreamWaitEvent(h->side_stream, out_free, 0));
        }
        if (d_greedy) {
            /* one side stream: Greedy kernels of consecutive calls in a row.  Alternating two streams, so that the next call's
             * workgroups move in as the previous call's leave, was measured at 0.273 ms/step against 0.233 — two persistent
             * kernels that each want a CU's LDS keep each other out */
            hipStream_t side = h->side_stream;
            HIPCHK(h, hipStreamWaitEvent(side, h->ev_packed, 0));
            PROF(3, 0, side)
            h->stream = side;
            rc = asm_align_batch_async(h, b, ASM_GREEDY, p, d_greedy);
            h->stream = main_stream;
            PROF(3, 1, side)
            if (!rc) HIPCHK(h, hipEventRecord(h->ev_join, side));
        }
        /* NW -> LEAP -> counters of a call, then the next call's NW, in a row on ONE stream (when all three are asked for).
         * Rounds 3-4 launched NW on the caller's stream and LEAP + counters on a stream of the library; under torch the two
         * shared a hardware queue (the runtime spreads streams over four, the handle's idle own stream holds one), which
         * serialised them in exactly this order — and that order is the fast one: with GPU_MAX_HW_QUEUES=8, where the next NW
         * really starts beside this call's LEAP and counters, a step takes 0.219 ms against 0.204.  Saying so explicitly makes
         * the step independent of how the host's streams happen to map to queues (same box: 0.206 at four queues, 0.208 at
         * eight). */
        hipStream_t chain = (d_nw && d_leap) ? h->acc_stream : main_stream;
        if (chain != main_stream) HIPCHK(h, hipStreamWaitEvent(chain, h->ev_packed, 0));
        if (!rc && d_nw) {
            PROF(1, 0, chain)
            h->stream = chain;
            rc = asm_align_batch_async(h, b, ASM_NW, p, d_nw);
            h->stream = main_stream;
            PROF(1, 1, chain)
        }
        hipStream_t ls = (d_nw && d_leap) ? h->acc_stream : main_stream;
        if (!rc && d_leap) {
            if (ls != chain) {
                HIPCHK(h, hipEventRecord(h->ev_nw, chain));
                HIPCHK(h, hipStreamWaitEvent(ls, h->ev_nw, 0));
            }
            PROF(2, 0, ls)
            h->stream = ls;
            rc = asm_align_batch_hinted_async(h, b, ASM_LEAP, p, d_nw, d_leap);
            h->stream = main_stream;
            PROF(2, 1, ls)
        }
        if (pe) h->prof_mask.push_back(pmask);
        if (rc) return rc;
        HIPCHK(h, hipEventRecord(h->ev_leap, ls));
        HIPCHK(h, hipStreamWaitEvent(h->acc_stream, h->ev_leap, 0));
        if (d_greedy) HIPCHK(h, hipStreamWaitEvent(h->acc_stream, h->ev_join, 0));
        if (d_counters) {
            h->stream = h->acc_stream;
            rc = asm_accuracy_async(h, d_nw, d_leap, d_greedy, d_answers, b->n, d_counters);
            h->stream = main_stream;
            if (rc) return rc;
        }
        HIPCHK(h, hipEventRecord(b->ev_consumed[b->cur], h->acc_stream));
        HIPCHK(h, hipEventRecord(out_free, h->acc_stream));
        h->calls3++;
        HIPCHK(h, hipEventRecord(h->ev_tail, h->acc_stream));
        h->tail_set = true;
        h->pipe_prev = true;
        return ASM_OK;
    }
    const bool fork = h->overlap && d_greedy && (d_nw || d_leap) && !rc && !greedy_first;
    if (fork) {
        HIPCHK(h, hipEventRecord(h->ev_fork, main_stream));
        HIPCHK(h, hipStreamWaitEvent(h->side_stream, h->ev_fork, 0));
        PROF(3, 0, h->side_stream)
        h->stream = h->side_stream;
        rc = asm_align_batch_async(h, b, ASM_GREEDY, p, d_greedy);
        h->stream = main_stream;
        PROF(3, 1, h->side_stream)
        if (!rc) HIPCHK(h, hipEventRecord(h->ev_join, h->side_stream));
    }
    if (!rc && d_nw) {
        PROF(1, 0, main_stream)
        rc = asm_align_batch_async(h, b, ASM_NW, p, d_nw);
        PROF(1, 1, main_stream)
        if (!rc && repack == 2) {
            HIPCHK(h, hipEventRecord(h->ev_gate, main_stream));
            h->gate_set = true;
        }
    }
    /* LEAP is scheduled by the NW penalties just computed (same work, sorted inside each workgroup) */
    if (!rc && d_leap) {
        PROF(2, 0, main_stream)
        rc = asm_align_batch_hinted_async(h, b, ASM_LEAP, p, greedy_first ? d_greedy : d_nw, d_leap);
        PROF(2, 1, main_stream)
    }
    if (fork) {
        if (!rc) HIPCHK(h, hipStreamWaitEvent(main_stream, h->ev_join, 0));
    } else if (!rc && d_greedy && !greedy_first) {
        PROF(3, 0, main_stream)
        rc = asm_align_batch_async(h, b, ASM_GREEDY, p, d_greedy);
        PROF(3, 1, main_stream)
    }
#undef PROF
    if (pe) h->prof_mask.push_back(pmask);
    if (pipelined && !rc) HIPCHK(h, hipEventRecord(b->ev_consumed[b->cur], main_stream)); /* Greedy's stream has joined above */
    if (!rc && d_counters) rc = asm_accuracy_async(h, d_nw, d_leap, d_greedy, d_answers, b->n, d_counters);
    h->pipe_prev = pipelined;
    return rc;
}

/* ---------------------------------------------------------------------------------------------------- */
/* Streaming ingest: a `>read\n<ref\n` file (benchmark_utils.h:325-352) through the aligners in chunks.
 *   reader threads   pread() the next chunk into pinned host memory (three buffers in rotation) and count its newlines, so
 *                    that the chunk ends on a pair boundary; what follows the boundary is carried into the next chunk
 *   copy stream      raw bytes -> HBM (two device buffers), overlapped with the compute stream's work on the chunk before
 *   compute stream   parse on the device (asm_ingest.h), pack, [sequential mode: chain the stale tails from the state the
 *                    chunks before left behind], NW / LEAP / Greedy, counters, penalties back into pinned staging
 *   caller's thread  hands results of chunk c-2 to the caller's arrays while chunk c-1 computes and chunk c is copied */
namespace {
using asm_host::NlScan;
using asm_host::scan_newlines;

/* A batch out of raw text already in HBM (n pairs = 2n lines, every line ending in '\n'). */
static int batch_from_device_text(asm_handle* h, const char* d_raw, size_t nbytes, int64_t n, int greedy_mode, asm_batch** out) {
    *out = nullptr;
    asm_batch* b = new asm_batch;
    b->owner = h, b->owner_serial = h->serial;
    b->n = n;
    b->greedy_mode = greedy_mode;
    uint32_t *d_tile = nullptr, *d_tbase = nullptr, *d_nl = nullptr, *d_m = nullptr, *d_n = nullptr, *d_max = nullptr;
    unsigned long long *d_sa = nullptr, *d_sb = nullptr;
    void* d_tmp = nullptr;
    int rc = ASM_OK;
    do {
#define TRY(call)                                                        \
    if ((call) != hipSuccess) {                                          \
        rc = fail(h, ASM_ENODEVICE, std::string(#call) + " failed");     \
        break;                                                           \
    }
        const size_t cnt = (size_t)n + 1;
        const long ntiles = (long)((nbytes + SEQ_TILE - 1) / SEQ_TILE);
        TRY(pool_alloc(h, (void**)&b->d_read_off, sizeof(uint32_t) * cnt));
        TRY(pool_alloc(h, (void**)&b->d_ref_off, sizeof(uint32_t) * cnt));
        TRY(pool_alloc(h, (void**)&d_m, sizeof(uint32_t) * cnt));
        TRY(pool_alloc(h, (void**)&d_n, sizeof(uint32_t) * cnt));
        TRY(pool_alloc(h, (void**)&d_sa, sizeof(unsigned long long) * cnt));
        TRY(pool_alloc(h, (void**)&d_sb, sizeof(unsigned long long) * cnt));
        TRY(pool_alloc(h, (void**)&d_nl, sizeof(uint32_t) * (2 * (size_t)n + 2)));
        TRY(pool_alloc(h, (void**)&d_tile, sizeof(uint32_t) * ((size_t)ntiles + 1)));
        TRY(pool_alloc(h, (void**)&d_tbase, sizeof(uint32_t) * ((size_t)ntiles + 1)));
        TRY(pool_alloc(h, (void**)&d_max, 16));
        TRY(hipMemsetAsync(d_max, 0, 16, h->stream));
        size_t tmp_bytes = 0, t2 = 0;
        TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_tile, d_tbase, (int)ntiles, h->stream));
        TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, t2, d_m, b->d_read_off, (int)cnt, h->stream));
        tmp_bytes = t2 > tmp_bytes ? t2 : tmp_bytes;
        TRY(pool_alloc(h, &d_tmp, tmp_bytes + 16));
        if (n > 0) {
            hipLaunchKernelGGL(seq_count_kernel, dim3((unsigned)ntiles), dim3(256), 0, h->stream, d_raw, (long)nbytes, d_tile);
            TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_tile, d_tbase, (int)ntiles, h->stream));
            hipLaunchKernelGGL(seq_index_kernel, dim3((unsigned)ntiles), dim3(256), 0, h->stream, d_raw, (long)nbytes,
                               (const uint32_t*)d_tbase, d_nl, (long)(2 * n));
        }
        hipLaunchKernelGGL(seq_lengths_kernel, dim3(grid_for(n + 1)), dim3(ASM_BLOCK), 0, h->stream, (const uint32_t*)d_nl, (long)n,
                           d_m, d_n, d_sa, d_sb);
        TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_m, b->d_read_off, (int)cnt, h->stream));
        TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_n, b->d_ref_off, (int)cnt, h->stream));
        if (n > 0) {
            int64_t blocks = (n + ASM_BLOCK - 1) / ASM_BLOCK;
            blocks = blocks > 1024 ? 1024 : blocks;
            hipLaunchKernelGGL(seq_max_kernel, dim3((unsigned)blocks), dim3(ASM_BLOCK), 0, h->stream, (const uint32_t*)d_m,
                               (const uint32_t*)d_n, (long)n, d_max);
        }
        TRY(hipGetLastError());
        uint32_t tot[2] = {0, 0}, mx = 0;
        TRY(hipMemcpyAsync(&tot[0], b->d_read_off + n, 4, hipMemcpyDeviceToHost, h->stream));
        TRY(hipMemcpyAsync(&tot[1], b->d_ref_off + n, 4, hipMemcpyDeviceToHost, h->stream));
        TRY(hipMemcpyAsync(&mx, d_max, 4, hipMemcpyDeviceToHost, h->stream));
        TRY(hipStreamSynchronize(h->stream));
        if ((int)mx > ASM_MAX_LENGTH) {
            rc = fail(h, ASM_EUNSUPPORTED, "streamed file: a sequence is longer than ASM_MAX_LENGTH");
            break;
        }
        b->reads_bytes = tot[0], b->refs_bytes = tot[1], b->maxlen = (int)mx;
        TRY(pool_alloc(h, (void**)&b->d_reads, b->reads_bytes + 16));
        TRY(pool_alloc(h, (void**)&b->d_refs, b->refs_bytes + 16));
        if (n > 0) {
            int64_t blocks = (n + 3) / 4;
            blocks = blocks > 256 * 16 ? 256 * 16 : blocks;
            hipLaunchKernelGGL(seq_gather_kernel, dim3((unsigned)blocks), dim3(ASM_BLOCK), 0, h->stream, d_raw,
                               (const unsigned long long*)d_sa, (const uint32_t*)b->d_read_off, (long)n, b->d_reads);
            hipLaunchKernelGGL(seq_gather_kernel, dim3((unsigned)blocks), dim3(ASM_BLOCK), 0, h->stream, d_raw,
                               (const unsigned long long*)d_sb, (const uint32_t*)b->d_ref_off, (long)n, b->d_refs);
            TRY(hipGetLastError());
        }
#undef TRY
        rc = batch_finish(h, b);
    } while (0);
    pool_free(h, d_tile), pool_free(h, d_tbase), pool_free(h, d_nl), pool_free(h, d_m), pool_free(h, d_n);
    pool_free(h, d_sa), pool_free(h, d_sb), pool_free(h, d_max), pool_free(h, d_tmp);
    if (rc) {
        batch_release(b);
        return rc;
    }
    *out = b;
    return ASM_OK;
}

}  // namespace

int asm_batch_from_text(asm_handle* h, const char* text, size_t nbytes, int greedy_mode, asm_batch** out) {
    if (!h || !out || (!text && nbytes)) return fail(h, ASM_EINVAL, "asm_batch_from_text: bad argument");
    if (greedy_mode != ASM_GREEDY_CLEAN && greedy_mode != ASM_GREEDY_SEQUENTIAL)
        return fail(h, ASM_EINVAL, "asm_batch_from_text: unknown greedy_mode");
    if (nbytes >= 0xfffffff0ull) return fail(h, ASM_EUNSUPPORTED, "asm_batch_from_text: more than 4 GiB of text; split it");
    HIPCHK(h, hipSetDevice(h->device));
    const bool open_line = nbytes && text[nbytes - 1] != '\n';
    NlScan sc = scan_newlines(text, nbytes, 8);
    int64_t lines = sc.count + (open_line ? 1 : 0);
    const bool odd = (lines & 1) != 0; /* a read without its reference line: the reference gets an empty string there */
    const size_t total = nbytes + (open_line ? 1 : 0) + (odd ? 1 : 0);
    lines += odd ? 1 : 0;
    char* d_raw = nullptr;
    HIPCHK(h, pool_alloc(h, (void**)&d_raw, total + 32));
    int rc = ASM_OK;
    if (nbytes && hipMemcpyAsync(d_raw, text, nbytes, hipMemcpyHostToDevice, h->stream) != hipSuccess) rc = fail(h, ASM_ENODEVICE, "asm_batch_from_text: copy failed");
    if (!rc && total > nbytes && hipMemsetAsync(d_raw + nbytes, '\n', total - nbytes, h->stream) != hipSuccess) rc = fail(h, ASM_ENODEVICE, "asm_batch_from_text: memset failed");
    if (!rc) rc = batch_from_device_text(h, d_raw, total, lines / 2, greedy_mode, out);
    pool_free(h, d_raw);
    return rc;
}

int asm_stream_seq_file(asm_handle* h, const char* path, const asm_params* p, int greedy_mode, int aligner_mask,
                        int64_t chunk_bytes, int64_t max_pairs, int32_t* nw, int32_t* leap, int32_t* greedy, int64_t out_cap,
                        const int32_t* answers, int64_t n_answers, asm_stream_stats* stats) {
    if (!h || !path || !p || !stats) return fail(h, ASM_EINVAL, "asm_stream_seq_file: NULL argument");
    if (greedy_mode != ASM_GREEDY_CLEAN && greedy_mode != ASM_GREEDY_SEQUENTIAL)
        return fail(h, ASM_EINVAL, "asm_stream_seq_file: unknown greedy_mode");
    memset(stats, 0, sizeof *stats);
    const bool do_nw = (aligner_mask & 1) != 0, do_leap = (aligner_mask & 2) != 0, do_greedy = (aligner_mask & 4) != 0;
    if (!(do_nw || do_leap || do_greedy)) return fail(h, ASM_EINVAL, "asm_stream_seq_file: empty aligner mask");
    HIPCHK(h, hipSetDevice(h->device));
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(h, ASM_EINVAL, std::string("asm_stream_seq_file: cannot open ") + path); /* benchmark_utils.h:350 */
    struct stat st;
    if (fstat(fd, &st) != 0) {
        close(fd);
        return fail(h, ASM_EINVAL, "asm_stream_seq_file: fstat failed");
    }
    const size_t file_bytes = (size_t)st.st_size;
    size_t chunk = chunk_bytes > 0 ? (size_t)chunk_bytes : ((size_t)64 << 20);
    chunk = chunk < 4096 ? 4096 : chunk;
    if (chunk > ((size_t)1 << 30)) chunk = (size_t)1 << 30;
    const size_t slot_cap = chunk + ((size_t)4 << 20); /* + room for the carried tail and the EOF padding */
    /* reader workers: one thread's page-cache copy (~5 GB/s) is far below PCIe; a GPU box gives a job 16 CPUs per GPU */
    int reader_threads = (int)std::thread::hardware_concurrency();
    reader_threads = reader_threads > 16 ? 16 : (reader_threads < 2 ? 2 : reader_threads);
    if (const char* env = getenv("ASM_READER_THREADS")) reader_threads = atoi(env) > 0 ? atoi(env) : reader_threads;
    const auto t_begin = std::chrono::steady_clock::now();

    hipEvent_t ev_shipped[3] = {nullptr, nullptr, nullptr}; /* the H2D copy out of host slot q is over */
    /* the reader side lives in asm_host.h (no HIP there: the same code runs under ThreadSanitizer in host/asm_host_check.cpp);
     * the one thing it needs from the device is "has the copy out of this slot finished" */
    /* large chunks ramp up from a sixteenth (asm_host::SeqReader): the transfer of the first chunk starts after 1/16 of a chunk
     * has been read instead of after a whole one */
    const size_t first_chunk = chunk >= ((size_t)32 << 20) ? chunk / 16 : chunk;
    asm_host::SeqReader rd(fd, file_bytes, chunk, reader_threads, max_pairs, [&](int q) {
        (void)hipSetDevice(h->device);
        (void)hipEventSynchronize(ev_shipped[q]);
    }, first_chunk);
    char* d_raw[2] = {nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    int32_t* h_pen[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    int32_t* d_pen[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    int32_t* d_ans[2] = {nullptr, nullptr};
    unsigned long long* d_cnt = nullptr;
    int64_t pen_cap = 0;
    int rc = ASM_OK;
    auto cleanup = [&]() {
        rd.stop();
        (void)hipDeviceSynchronize();
        for (hipEvent_t ev : ev_shipped)
            if (ev) (void)hipEventDestroy(ev);
        for (int q = 0; q < 2; q++) {
            pool_free(h, d_raw[q]);
            if (ev_h2d[q]) (void)hipEventDestroy(ev_h2d[q]);
            if (ev_done[q]) (void)hipEventDestroy(ev_done[q]);
            for (int a = 0; a < 3; a++) pool_free(h, d_pen[q][a]);
            pool_free(h, d_ans[q]);
        }
        pool_free(h, d_cnt);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        close(fd);
    };
#define STREAM_TRY(call)                                                                   \
    if (!rc && (call) != hipSuccess) rc = fail(h, ASM_ENODEVICE, std::string(#call) + " failed")

    if (h->pin_raw_cap < slot_cap) { /* (re)pin */
        for (char*& q : h->pin_raw) {
            if (q) (void)hipHostFree(q);
            q = nullptr;
        }
        h->pin_raw_cap = 0;
        for (char*& q : h->pin_raw) STREAM_TRY(hipHostMalloc((void**)&q, slot_cap + 64, hipHostMallocDefault));
        if (!rc) h->pin_raw_cap = slot_cap;
    }
    for (int q = 0; q < 3; q++) {
        rd.slot[q].buf = h->pin_raw[q];
        rd.slot[q].cap = slot_cap;
        STREAM_TRY(hipEventCreateWithFlags(&ev_shipped[q], hipEventDisableTiming));
    }
    for (int qq = 0; qq < 2; qq++)
        for (int a = 0; a < 3; a++) h_pen[qq][a] = h->pin_pen[qq][a];
    pen_cap = 0; /* device staging is (re)allocated with the first chunk; the pinned side is reused when large enough */
    STREAM_TRY(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    for (int q = 0; q < 2; q++) {
        STREAM_TRY(pool_alloc(h, (void**)&d_raw[q], slot_cap + 64));
        STREAM_TRY(hipEventCreateWithFlags(&ev_h2d[q], hipEventDisableTiming));
        STREAM_TRY(hipEventCreateWithFlags(&ev_done[q], hipEventDisableTiming));
    }
    STREAM_TRY(pool_alloc(h, (void**)&d_cnt, 32));
    STREAM_TRY(hipMemsetAsync(d_cnt, 0, 32, h->stream));
    /* d_raw comes from the pool, whose blocks are recycled in the order of the HANDLE's stream: kernels still queued there may
     * read the block's previous life.  The copy stream is non-blocking and would not wait for them by itself. */
    if (!rc) {
        hipEvent_t ev_pool = nullptr;
        STREAM_TRY(hipEventCreateWithFlags(&ev_pool, hipEventDisableTiming));
        STREAM_TRY(hipEventRecord(ev_pool, h->stream));
        STREAM_TRY(hipStreamWaitEvent(copy_stream, ev_pool, 0));
        if (ev_pool) (void)hipEventDestroy(ev_pool);
    }
    if (rc) {
        cleanup();
        return rc;
    }

    rd.start(); /* the reader thread: fills the three slots in rotation (asm_host::SeqReader) */

    /* ---- caller's thread: ship, compute, harvest ---- */
    uint8_t tail_state[256];
    memset(tail_state, 0, sizeof tail_state);
    int64_t chunk_first[2] = {0, 0}, chunk_pairs[2] = {0, 0};
    asm_batch* chunk_batch[2] = {nullptr, nullptr};
    int64_t done_pairs = 0;
    int maxlen = 0;
    auto harvest = [&](int q) { /* results of the chunk that used device buffer q */
        if (chunk_pairs[q] <= 0 && !chunk_batch[q]) return;
        if (hipEventSynchronize(ev_done[q]) != hipSuccess && !rc) rc = fail(h, ASM_ENODEVICE, "asm_stream_seq_file: chunk failed");
        int32_t* dst[3] = {nw, leap, greedy};
        for (int a = 0; a < 3; a++) {
            if (!dst[a] || !h_pen[q][a]) continue;
            const int64_t room = out_cap - chunk_first[q];
            const int64_t cnt = chunk_pairs[q] < room ? chunk_pairs[q] : (room > 0 ? room : 0);
            if (cnt > 0) memcpy(dst[a] + chunk_first[q], h_pen[q][a], sizeof(int32_t) * (size_t)cnt);
        }
        if (chunk_batch[q]) batch_release(chunk_batch[q]);
        chunk_batch[q] = nullptr;
        chunk_pairs[q] = 0;
    };
    /* The loop runs two stages per iteration, one chunk apart: SHIP chunk c (host buffer -> HBM on the copy stream) and only then
     * PROCESS chunk c-1 (parse, pack, aligners, results back).  Processing blocks this thread twice (the parser's totals and the
     * packed batch), so with the stages the other way round the transfer of the next chunk could not start before the current
     * one was packed, and the copy engine idled through every parse (round 2: 1.1e8 pairs/s; the same code in this order:
     * DESIGN.md section 4b). */
    bool last = false;
    int64_t chunks = 0;
    size_t bytes_total = 0;
    int64_t pend_pairs[2] = {0, 0};
    size_t pend_bytes[2] = {0, 0};
    bool pend_valid[2] = {false, false};
    auto process = [&](int q) { /* the chunk whose text sits in d_raw[q] */
        if (!pend_valid[q]) return;
        pend_valid[q] = false;
        const int64_t n = pend_pairs[q];
        const size_t shipped = pend_bytes[q];
        if (n <= 0 || rc) return;
        harvest(q); /* the chunk two back used the same result staging */
        if (n > pen_cap) { /* staging sized from the first chunk for a full one (its pairs per byte, times the slot's bytes, plus an
                              eighth: chunks ramp up to `chunk`); pen_cap = 0 until the first chunk — not "is the NW buffer there",
                              which a mask without NW never satisfies */
            harvest(q ^ 1);
            const int64_t full = (int64_t)((double)n / (double)(shipped ? shipped : 1) * (double)slot_cap) + 1;
            const int64_t cap = (full > n ? full : n) + (full > n ? full : n) / 8 + 1024;
            for (int qq = 0; qq < 2 && !rc; qq++)
                for (int a = 0; a < 3 && !rc; a++) {
                    if (!((aligner_mask >> a) & 1)) continue;
                    pool_free(h, d_pen[qq][a]);
                    d_pen[qq][a] = nullptr;
                    if (!h->pin_pen[qq][a] || h->pin_pen_cap[qq][a] < cap) {
                        if (h->pin_pen[qq][a]) (void)hipHostFree(h->pin_pen[qq][a]);
                        h->pin_pen[qq][a] = nullptr, h->pin_pen_cap[qq][a] = 0;
                        STREAM_TRY(hipHostMalloc((void**)&h->pin_pen[qq][a], sizeof(int32_t) * (size_t)cap, hipHostMallocDefault));
                        if (!rc) h->pin_pen_cap[qq][a] = cap;
                    }
                    h_pen[qq][a] = h->pin_pen[qq][a];
                    STREAM_TRY(pool_alloc(h, (void**)&d_pen[qq][a], sizeof(int32_t) * (size_t)cap));
                }
            for (int qq = 0; qq < 2 && !rc && answers; qq++) {
                pool_free(h, d_ans[qq]);
                d_ans[qq] = nullptr;
                STREAM_TRY(pool_alloc(h, (void**)&d_ans[qq], sizeof(int32_t) * (size_t)cap));
            }
            pen_cap = cap;
        }
        if (rc) return;
        STREAM_TRY(hipStreamWaitEvent(h->stream, ev_h2d[q], 0));
        asm_batch* b = nullptr;
        if (!rc) rc = batch_from_device_text(h, d_raw[q], shipped, n, ASM_GREEDY_CLEAN, &b);
        if (rc) return;
        if (greedy_mode == ASM_GREEDY_SEQUENTIAL && do_greedy) { /* the chain of hurdle_matrix.h:136-137 across chunk boundaries */
            uint8_t summary[256];
            rc = batch_resolve_tails(h, b, nullptr, summary, false);
            if (!rc) {
                b->greedy_mode = ASM_GREEDY_SEQUENTIAL;
                rc = batch_resolve_tails(h, b, tail_state, nullptr, true);
            }
            if (!rc) rc = asm_batch_pack_async(h, b);
            if (!rc) rc = asm_tail_state_advance(tail_state, summary, n);
        }
        const int32_t* ans = nullptr;
        if (!rc && answers && done_pairs < n_answers) { /* read_answer_file (benchmark_utils.h:358-368): one integer per pair */
            const int64_t have = n_answers - done_pairs < n ? n_answers - done_pairs : n;
            std::vector<int32_t> pad((size_t)n, INT32_MIN);
            memcpy(pad.data(), answers + done_pairs, sizeof(int32_t) * (size_t)have);
            STREAM_TRY(hipMemcpyAsync(d_ans[q], pad.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, h->stream));
            STREAM_TRY(hipStreamSynchronize(h->stream)); /* `pad` is pageable and about to go out of scope */
            ans = d_ans[q];
        }
        if (!rc)
            rc = asm_run_benchmark_async(h, b, p, 0, do_nw ? d_pen[q][0] : nullptr, do_leap ? d_pen[q][1] : nullptr,
                                         do_greedy ? d_pen[q][2] : nullptr, ans, d_cnt);
        for (int a = 0; a < 3 && !rc; a++)
            if (d_pen[q][a]) STREAM_TRY(hipMemcpyAsync(h_pen[q][a], d_pen[q][a], sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
        STREAM_TRY(hipEventRecord(ev_done[q], h->stream));
        if (rc) {
            if (b) batch_release(b);
            return;
        }
        chunk_batch[q] = b, chunk_first[q] = done_pairs, chunk_pairs[q] = n;
        maxlen = b->maxlen > maxlen ? b->maxlen : maxlen;
        done_pairs += n, bytes_total += shipped, chunks++;
    };
    for (int c = 0; !last && !rc; c++) {
        asm_host::SeqSlot* sp = rd.wait_ready(c);
        if (!sp) {
            rc = fail(h, ASM_EINVAL, "asm_stream_seq_file: read failed (or one pair is longer than a chunk)");
            break;
        }
        asm_host::SeqSlot& s = *sp;
        const int q = c & 1;
        /* SHIP chunk c.  d_raw[q] held chunk c-2, which was processed (and its text gathered into the batch's own arrays, with
         * this thread waiting for that) in the iteration before this one. */
        last = s.last;
        bool shipping = false;
        if (s.pairs > 0) {
            STREAM_TRY(hipMemcpyAsync(d_raw[q], s.buf, s.bytes, hipMemcpyHostToDevice, copy_stream));
            STREAM_TRY(hipEventRecord(ev_shipped[c % 3], copy_stream));
            shipping = !rc;
            STREAM_TRY(hipEventRecord(ev_h2d[q], copy_stream));
        }
        pend_pairs[q] = s.pairs, pend_bytes[q] = s.bytes, pend_valid[q] = true;
        rd.consumed(c, shipping); /* the reader may refill the slot once ev_shipped has fired */
        /* PROCESS chunk c-1 while chunk c is on its way */
        process(q ^ 1);
    }
    if (!rc && !rd.failed()) process(0), process(1); /* the last chunk shipped (only one of the two is pending) */
    harvest(0), harvest(1);
    if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(h, ASM_ENODEVICE, "asm_stream_seq_file: stream synchronize failed");
    if (!rc && hipMemcpy(stats->counters, d_cnt, 32, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(h, ASM_ENODEVICE, "asm_stream_seq_file: counters copy failed");
#undef STREAM_TRY
    cleanup();
    stats->pairs = done_pairs, stats->chunks = chunks, stats->bytes = (int64_t)bytes_total, stats->max_length = maxlen;
    stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    stats->seconds_read = rd.read_seconds();
    return rc;
}

/* ---------------------------------------------------------------------------------------------------- */
int asm_device_malloc(asm_handle* h, size_t bytes, void** d_ptr) {
    if (!h || !d_ptr) return fail(h, ASM_EINVAL, "asm_device_malloc: NULL argument");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, pool_alloc(h, d_ptr, bytes ? bytes : 1));
    return ASM_OK;
}
int asm_device_free(asm_handle* h, void* d_ptr) {
    if (!h) return fail(h, ASM_EINVAL, "asm_device_free: NULL handle");
    HIPCHK(h, hipSetDevice(h->device));
    pool_free(h, d_ptr); /* recycled in stream order: work already enqueued on the handle's stream may still use it */
    return ASM_OK;
}
int asm_memcpy_d2h(asm_handle* h, void* dst, const void* d_src, size_t bytes) {
    if (!h) return fail(h, ASM_EINVAL, "asm_memcpy_d2h: NULL handle");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return ASM_OK;
}
int asm_memcpy_h2d(asm_handle* h, void* d_dst, const void* src, size_t bytes) {
    if (!h) return fail(h, ASM_EINVAL, "asm_memcpy_h2d: NULL handle");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
    return ASM_OK;
}
int asm_memset_async(asm_handle* h, void* d_ptr, int value, size_t bytes) {
    if (!h) return fail(h, ASM_EINVAL, "asm_memset_async: NULL handle");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemsetAsync(d_ptr, value, bytes, h->stream));
    return ASM_OK;
}

struct asm_timer {
    hipEvent_t a, b;
};
int asm_timer_create(asm_handle* h, void** timer) {
    if (!h || !timer) return fail(h, ASM_EINVAL, "asm_timer_create: NULL argument");
    HIPCHK(h, hipSetDevice(h->device));
    asm_timer* t = new asm_timer;
    HIPCHK(h, hipEventCreate(&t->a));
    HIPCHK(h, hipEventCreate(&t->b));
    *timer = t;
    return ASM_OK;
}
int asm_timer_start(asm_handle* h, void* timer) {
    if (!h || !timer) return fail(h, ASM_EINVAL, "asm_timer_start: NULL argument");
    HIPCHK(h, hipEventRecord(((asm_timer*)timer)->a, h->stream));
    return ASM_OK;
}
int asm_timer_stop(asm_handle* h, void* timer) {
    if (!h || !timer) return fail(h, ASM_EINVAL, "asm_timer_stop: NULL argument");
    HIPCHK(h, hipEventRecord(((asm_timer*)timer)->b, h->stream));
    return ASM_OK;
}
int asm_timer_elapsed_ms(asm_handle* h, void* timer, float* ms) {
    if (!h || !timer || !ms) return fail(h, ASM_EINVAL, "asm_timer_elapsed_ms: NULL argument");
    asm_timer* t = (asm_timer*)timer;
    HIPCHK(h, hipEventSynchronize(t->b));
    HIPCHK(h, hipEventElapsedTime(ms, t->a, t->b));
    return ASM_OK;
}
int asm_timer_destroy(asm_handle* h, void* timer) {
    if (!timer) return ASM_OK;
    asm_timer* t = (asm_timer*)timer;
    (void)hipEventDestroy(t->a);
    (void)hipEventDestroy(t->b);
    delete t;
    (void)h;
    return ASM_OK;
}

} /* extern "C" */
