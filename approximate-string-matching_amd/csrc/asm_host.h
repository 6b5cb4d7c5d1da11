// The device-free part of the C ABI's host side: everything in libasm_mi355x.so that runs on the CPU and never calls HIP —
// the seeded generator's host loop (asm_generate_pairs), the stale-tail state arithmetic (asm_tail_state_advance), the CIGAR
// formatter (asm_cigar_format), and the reader side of asm_stream_seq_file: newline scanning, the persistent reader pool, and
// the three-slot hand-over between the reader thread and the caller's thread (SeqReader).
//
// Kept in a header without any HIP include so that the SAME code is compiled twice: into the product by hipcc (asm_capi.hip),
// and by plain g++ under -fsanitize=thread / address,undefined into host/asm_host_check.cpp (`make -C oracle asan`,
// tests/test_sanitizers.py).  GPU AddressSanitizer is not available on the target pool; this is the part of the library a CPU
// sanitizer can see.
#pragma once
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/types.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/asm_mi355x.h"
#include "asm_gen.h"

namespace asm_host {

// ---- generator (benchmark_dataset.h:85-253 restated over a counter-based stream, asm_gen.h) ---------------------------------
inline int check_gen(const asm_gen_config* cfg, std::string& err) {
    if (!cfg) return err = "generator: cfg is NULL", ASM_EINVAL;
    if (cfg->len_lo < 1 || cfg->len_hi < cfg->len_lo || cfg->len_hi > ASM_MAX_LENGTH)
        return err = "generator: need 1 <= len_lo <= len_hi <= ASM_MAX_LENGTH", ASM_EINVAL;
    if (cfg->kind == ASM_GEN_EXACT_ERRORS || cfg->kind == ASM_GEN_UP_TO_ERRORS) {
        /* benchmark_dataset.h:192-204 */
        if (!(cfg->err >= 0.f && cfg->err <= 0.7f)) return err = "generator: err must be in [0, 0.7]", ASM_EINVAL;
        if (!(cfg->mismatch_rate >= 0.f && cfg->mismatch_rate <= 1.f)) return err = "generator: mismatch_rate must be in [0, 1]", ASM_EINVAL;
    } else if (cfg->kind == ASM_GEN_PER_BASE) {
        if (!(cfg->p_sub >= 0.f && cfg->p_ins >= 0.f && cfg->p_del >= 0.f && cfg->p_sub + cfg->p_del <= 1.f && cfg->p_ins <= 1.f))
            return err = "generator: per-base rates out of range", ASM_EINVAL;
    } else {
        return err = "generator: unknown kind", ASM_EINVAL;
    }
    return ASM_OK;
}

inline int generate_pairs(const asm_gen_config* cfg, int64_t first, int64_t n, uint32_t* read_off, uint32_t* ref_off, char* reads,
                          size_t reads_cap, char* refs, size_t refs_cap, std::string& err) {
    int rc = check_gen(cfg, err);
    if (rc) return rc;
    if (n < 0 || first < 0 || !read_off || !ref_off) return err = "asm_generate_pairs: bad arguments", ASM_EINVAL;
    uint64_t ra = 0, rb = 0;
    for (int64_t i = 0; i < n; i++) {
        int m, nn;
        asm_gen_lengths(cfg, (uint64_t)(first + i), &m, &nn);
        read_off[i] = (uint32_t)ra;
        ref_off[i] = (uint32_t)rb;
        ra += (uint64_t)m;
        rb += (uint64_t)nn;
        if (ra > 0xffffffffull || rb > 0xffffffffull) return err = "asm_generate_pairs: batch exceeds 4 GiB of text; split it", ASM_EUNSUPPORTED;
    }
    read_off[n] = (uint32_t)ra;
    ref_off[n] = (uint32_t)rb;
    if (!reads || !refs) return ASM_OK; /* sizing pass */
    if (reads_cap < ra || refs_cap < rb) return err = "asm_generate_pairs: output buffers too small", ASM_EINVAL;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        char rd[ASM_MAX_LENGTH + 8], tx[ASM_GEN_MAX_TEXT];
        int m, nn;
        asm_gen_pair(cfg, (uint64_t)(first + i), rd, tx, &m, &nn);
        memcpy(reads + read_off[i], rd, (size_t)m);
        memcpy(refs + ref_off[i], tx, (size_t)nn);
    }
    return ASM_OK;
}

// ---- Greedy's stale tails: the host side of the chain across batches (hurdle_matrix.h:136-137,630-631) -----------------------
#define TAIL_NONE 0xFFu
// Where the trajectory that starts in `slot` sits after n pairs: every conversion leaves buf'[q] = buf[SRC[q]],
// SRC[q] = 8 * (q & 15) + P[q >> 4] (bit_convert.cpp:265-330), and SRC has order 10.
static inline int tail_slot_after(int slot, long long n) {
    for (int i = 0, r = (int)(n % 10); i < r; i++) {
        const int v = slot & 7, low2 = v & 3;
        const int pv = (low2 == 1 || low2 == 2) ? (v ^ 3) : v;
        slot = (pv << 4) | (slot >> 3);
    }
    return slot;
}

// The same step on all 128 slots of a buffer at once, one bit per slot (bit q of w[q >> 5] = slot q): what the resolver's
// device passes run per pair and plane (csrc/asm_tails.h).  after[q] = before[SRC[q]] reads the buffer as 16 bytes of 8 bits and
// leaves 8 halfwords of 16 bits: halfword j, bit r = byte r, bit P[j] — two 8x8 bit transposes (bytes 0-7, bytes 8-15), then
// byte (P[j]) of the first next to byte (P[j]) of the second.  constexpr: clang compiles these for host and device alike, and
// the static_assert below checks the permutation against its definition for every slot while the library is being compiled.
struct TailBits {
    uint32_t w[4];
};
constexpr uint32_t tail_bfi(uint32_t m, uint32_t x, uint32_t y) { return (x & m) | (y & ~m); } /* v_bfi_b32 */
constexpr void tail_transpose8(uint32_t& lo, uint32_t& hi) { /* bytes 0-3 in lo, 4-7 in hi: byte c, bit r <- byte r, bit c */
    lo = tail_bfi(0xAA55AA55u, lo, tail_bfi(0x00AA00AAu, lo >> 7, lo << 7));
    hi = tail_bfi(0xAA55AA55u, hi, tail_bfi(0x00AA00AAu, hi >> 7, hi << 7));
    lo = tail_bfi(0xCCCC3333u, lo, tail_bfi(0x0000CCCCu, lo >> 14, lo << 14));
    hi = tail_bfi(0xCCCC3333u, hi, tail_bfi(0x0000CCCCu, hi >> 14, hi << 14));
    const uint32_t l2 = tail_bfi(0x0F0F0F0Fu, lo, hi << 4);
    hi = tail_bfi(0xF0F0F0F0u, hi, lo >> 4);
    lo = l2;
}
constexpr TailBits tail_permute(TailBits v) {
    uint32_t a0 = v.w[0], a1 = v.w[1], b0 = v.w[2], b1 = v.w[3];
    tail_transpose8(a0, a1);
    tail_transpose8(b0, b1);
    TailBits o{};
    o.w[0] = tail_bfi(0x00FF00FFu, a0, b0 << 8);      /* halfwords 0, 1: bytes P[0] = 0 and P[1] = 2 of both transposes */
    o.w[1] = tail_bfi(0x00FF00FFu, a0 >> 8, b0);      /* halfwords 2, 3: bytes 1, 3 */
    o.w[2] = tail_bfi(0x00FF00FFu, a1, b1 << 8);      /* halfwords 4, 5: bytes 4, 6 */
    o.w[3] = tail_bfi(0x00FF00FFu, a1 >> 8, b1);      /* halfwords 6, 7: bytes 5, 7 */
    return o;
}
constexpr uint32_t tail_prefix_word(uint32_t len, uint32_t lo) { /* the bits of [0, len) that fall into slots [lo, lo + 32) */
    return len >= lo + 32u ? 0xFFFFFFFFu : (len > lo ? (1u << ((len - lo) & 31u)) - 1u : 0u);
}
constexpr TailBits tail_prefix(uint32_t len) { /* slots [0, len), len <= 128 */
    return TailBits{{tail_prefix_word(len, 0u), tail_prefix_word(len, 32u), tail_prefix_word(len, 64u), tail_prefix_word(len, 96u)}};
}
constexpr bool tail_permute_selfcheck() {
    for (int q = 0; q < 128; q++) { /* a byte in slot s moves to tail_slot_after(s, 1) */
        TailBits one{};
        one.w[q >> 5] = 1u << (q & 31);
        const TailBits got = tail_permute(one);
        const int v = q & 7, low2 = v & 3, pv = (low2 == 1 || low2 == 2) ? (v ^ 3) : v, to = (pv << 4) | (q >> 3);
        for (int d = 0; d < 4; d++)
            if (got.w[d] != ((to >> 5) == d ? 1u << (to & 31) : 0u)) return false;
    }
    return true;
}
static_assert(tail_permute_selfcheck(), "tail_permute does not move slot s to SRC^-1[s]");

inline int tail_state_advance(uint8_t* state, const uint8_t* summary, int64_t n_pairs, std::string& err) {
    if (!state || !summary || n_pairs < 0) return err = "asm_tail_state_advance: bad argument", ASM_EINVAL;
    uint8_t next[256];
    for (int side = 0; side < 2; side++)
        for (int s = 0; s < 128; s++) {
            const uint8_t w = summary[side * 128 + s];
            if (w != TAIL_NONE && w > 3) return err = "asm_tail_state_advance: summary entries are 0..3 or 0xFF", ASM_EINVAL;
            next[side * 128 + tail_slot_after(s, (long long)n_pairs)] = w != TAIL_NONE ? w : state[side * 128 + s];
        }
    memcpy(state, next, 256);
    return ASM_OK;
}

// ---- CIGAR rows -> text (hurdle_matrix::_update_CIGAR, hurdle_matrix.h:238-251; '=' and 'X' come from the NW traceback) -----
inline int cigar_format(const uint16_t* ops, int nops, int cap, char* out, size_t out_cap) {
    if (!ops || !out || out_cap == 0) return ASM_EINVAL;
    size_t len = 0;
    out[0] = 0;
    const int cnt = nops < cap ? nops : cap;
    for (int i = 0; i < cnt; i++) {
        const char op = "MID=X???"[ops[i] & 7];
        const int w = snprintf(out + len, out_cap - len, "%d%c", (int)(ops[i] >> 3), op);
        if (w < 0 || len + (size_t)w >= out_cap) return ASM_EINVAL;
        len += (size_t)w;
    }
    return nops > cap ? ASM_EUNSUPPORTED : ASM_OK; /* truncated row */
}

// ---- `>read\n<ref\n` files (benchmark_utils.h:325-352): newline scanning and the reader side of the streaming ingest ---------
/* newlines in [p, p+len): count and the positions (relative to p) of the last two */
struct NlScan {
    int64_t count = 0;
    int64_t last = -1, prev = -1;
};

inline void nl_merge(NlScan& tot, const NlScan& r) { /* append a later segment's summary (positions on one common base) */
    if (!r.count) return;
    tot.count += r.count;
    if (r.count >= 2) tot.prev = r.prev;
    else tot.prev = tot.last; /* the segment's only newline: the one before it is the running last */
    tot.last = r.last;
}

inline NlScan scan_range(const char* base, size_t a, size_t b) { /* newlines of base[a, b), positions relative to base */
    NlScan r;
    const char* q = base + a;
    const char* end = base + (a < b ? b : a);
    while (q < end) {
        const char* hit = (const char*)memchr(q, '\n', (size_t)(end - q));
        if (!hit) break;
        r.count++, r.prev = r.last, r.last = (int64_t)(hit - base);
        q = hit + 1;
    }
    return r;
}

inline NlScan scan_newlines(const char* p, size_t len, int threads) {
    if (threads < 1) threads = 1;
    std::vector<NlScan> part((size_t)threads);
    std::vector<std::thread> pool;
    const size_t step = (len + (size_t)threads - 1) / (size_t)threads;
    auto work = [&](int t) {
        const size_t a = (size_t)t * step, b = a + step < len ? a + step : len;
        part[(size_t)t] = scan_range(p, a, b);
    };
    for (int t = 1; t < threads; t++) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
    NlScan tot;
    for (const NlScan& r : part) nl_merge(tot, r);
    return tot;
}

/* Worker threads that live as long as one asm_stream_seq_file call: a chunk is read AND scanned for newlines by the same
 * workers in one go (each its own slice: pread into the pinned buffer, then memchr over the bytes it has just written).
 * Round 2 started 2 x 8 threads per chunk — half a millisecond of every 64 MB chunk — and passed over the data twice. */
class StreamWorkers {
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    std::function<void(int)> job_;
    int generation_ = 0, pending_ = 0;
    bool quit_ = false;

public:
    explicit StreamWorkers(int n) {
        for (int t = 0; t < n; t++)
            threads_.emplace_back([this, t]() {
                int seen = 0;
                for (;;) {
                    std::function<void(int)> job;
                    {
                        std::unique_lock<std::mutex> lk(mu_);
                        cv_work_.wait(lk, [&] { return quit_ || generation_ != seen; });
                        if (quit_) return;
                        seen = generation_;
                        job = job_;
                    }
                    job(t);
                    {
                        std::lock_guard<std::mutex> lk(mu_);
                        if (--pending_ == 0) cv_done_.notify_all();
                    }
                }
            });
    }
    int size() const { return (int)threads_.size(); }
    void run(const std::function<void(int)>& job) { /* job(t) on every worker t; returns when all are done */
        std::unique_lock<std::mutex> lk(mu_);
        job_ = job;
        pending_ = (int)threads_.size();
        generation_++;
        cv_work_.notify_all();
        cv_done_.wait(lk, [&] { return pending_ == 0; });
    }
    ~StreamWorkers() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
        }
        cv_work_.notify_all();
        for (auto& th : threads_) th.join();
    }
};

/* buf[0, head) is already there (the carry of the chunk before); reads `len` file bytes behind it and returns the newline
 * summary of buf[0, head + len) */
inline NlScan read_and_scan(StreamWorkers& pool, int fd, char* buf, size_t head, size_t len, off_t off, std::atomic<bool>& failed) {
    const int threads = pool.size();
    std::vector<NlScan> part((size_t)threads + 1);
    const size_t step = ((len + (size_t)threads - 1) / (size_t)threads + 4095) & ~(size_t)4095;
    part[0] = scan_range(buf, 0, head);
    pool.run([&](int t) {
        /* read and scan in blocks of 1 MB: the scan then finds the bytes the copy has just written still in the core's cache
         * (scanning an 8 MB slice after reading all of it fetched every byte from DRAM a second time) */
        const size_t a0 = (size_t)t * step, b = a0 + step < len ? a0 + step : len;
        NlScan mine;
        for (size_t a = a0; a < b;) {
            const size_t blk_end = a + ((size_t)1 << 20) < b ? a + ((size_t)1 << 20) : b;
            const size_t blk_a = a;
            while (a < blk_end) {
                const ssize_t got = pread(fd, buf + head + a, blk_end - a, off + (off_t)a);
                if (got <= 0) {
                    failed = true;
                    return;
                }
                a += (size_t)got;
            }
            nl_merge(mine, scan_range(buf, head + blk_a, head + blk_end));
        }
        part[(size_t)t + 1] = mine;
    });
    NlScan tot;
    for (const NlScan& r : part) nl_merge(tot, r);
    return tot;
}

struct SeqSlot { /* one (pinned) host buffer */
    char* buf = nullptr;
    size_t cap = 0;
    size_t bytes = 0;     /* raw bytes to ship: whole pairs only */
    int64_t pairs = 0;
    bool last = false;
    bool ready = false;     /* filled by the reader, not yet taken by the consumer */
    bool in_flight = false; /* the consumer has started an asynchronous copy out of it; wait_shipped(slot) tells when it is over */
};

/* The reader thread of asm_stream_seq_file and its hand-over to the caller's thread.  Three slots in rotation: the reader fills
 * slot c % 3 with chunk c — the carry of the chunk before, then `chunk` more file bytes, cut behind the last complete pair —
 * and marks it ready; the consumer takes the chunks in order (wait_ready), starts its copy out of the buffer and gives the
 * slot back (consumed), after which the reader may refill it once wait_shipped(slot) says the copy is over. */
class SeqReader {
    const int fd_;
    const size_t file_bytes_, chunk_, first_chunk_;
    const int reader_threads_;
    const int64_t max_pairs_;
    const std::function<void(int)> wait_shipped_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::atomic<bool> failed_{false}, stop_{false};
    double read_seconds_ = 0;
    std::thread reader_;

    void loop() {
        StreamWorkers workers(reader_threads_);
        std::vector<char> carry;
        size_t file_off = 0;
        int64_t pairs_left = max_pairs_ > 0 ? max_pairs_ : INT64_MAX;
        bool eof = false;
        for (int c = 0; !eof && !stop_; c++) {
            SeqSlot& s = slot[c % 3];
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || !s.ready; });
                if (stop_) return;
            }
            if (s.in_flight) { /* the copy out of this buffer (three chunks ago) must be over before it is overwritten */
                wait_shipped_(c % 3);
                s.in_flight = false;
            }
            const auto t0 = std::chrono::steady_clock::now();
            size_t have = carry.size();
            if (have) memcpy(s.buf, carry.data(), have);
            carry.clear();
            /* chunk c takes first_chunk << c file bytes until that reaches `chunk`: the consumer's pipeline (ship, parse, align) starts
             * after the FIRST chunk is in memory, so a small first chunk shortens the fill of the pipeline and the large later ones
             * keep the per-chunk costs rare */
            size_t want = c < 30 && (first_chunk_ << c) < chunk_ ? first_chunk_ << c : chunk_;
            if (file_off + want > file_bytes_) want = file_bytes_ - file_off;
            if (have + want > s.cap - 8) want = s.cap - 8 - have;
            NlScan sc = read_and_scan(workers, fd_, s.buf, have, want, (off_t)file_off, failed_);
            file_off += want;
            have += want;
            eof = file_off >= file_bytes_;
            if (eof && have && s.buf[have - 1] != '\n') { /* a last line without its newline */
                s.buf[have++] = '\n';
                sc.prev = sc.last, sc.last = (int64_t)have - 1, sc.count++;
            }
            if (eof && (sc.count & 1)) { /* a read line without its reference line: an empty reference */
                s.buf[have++] = '\n';
                sc.prev = sc.last, sc.last = (int64_t)have - 1, sc.count++;
            }
            int64_t lines = sc.count & ~(int64_t)1;
            size_t boundary = lines == 0 ? 0 : (size_t)((lines == sc.count ? sc.last : sc.prev) + 1);
            if (lines / 2 > pairs_left) { /* max_pairs cuts inside this chunk: find the boundary of the pairs_left-th pair */
                const int64_t need = 2 * pairs_left;
                const char* q = s.buf;
                for (int64_t l = 0; l < need; l++) q = (const char*)memchr(q, '\n', (size_t)(s.buf + have - q)) + 1;
                boundary = (size_t)(q - s.buf), lines = need;
                eof = true;
            }
            if (!eof) {
                /* (the two bytes the reader may append above stay inside the slot: reads stop at cap - 8 and every slot is
                 * allocated with cap + 64) */
                if (boundary == 0 && have >= s.cap - 8) failed_ = true; /* one pair longer than a whole chunk */
                carry.assign(s.buf + boundary, s.buf + have);
            }
            pairs_left -= lines / 2;
            if (pairs_left <= 0) eof = true;
            read_seconds_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            {
                std::lock_guard<std::mutex> lk(mu_);
                s.bytes = boundary, s.pairs = lines / 2, s.last = eof, s.ready = true;
            }
            cv_.notify_all();
            if (failed_) return;
        }
    }

public:
    SeqSlot slot[3]; /* the caller sets buf and cap (usable bytes; allocate cap + 64) before start() */

    /* first_chunk: file bytes of chunk 0 (0 or >= chunk: every chunk takes `chunk`) */
    SeqReader(int fd, size_t file_bytes, size_t chunk, int reader_threads, int64_t max_pairs, std::function<void(int)> wait_shipped,
              size_t first_chunk = 0)
        : fd_(fd), file_bytes_(file_bytes), chunk_(chunk), first_chunk_(first_chunk > 0 && first_chunk < chunk ? first_chunk : chunk),
          reader_threads_(reader_threads), max_pairs_(max_pairs), wait_shipped_(std::move(wait_shipped)) {}
    ~SeqReader() { stop(); }
    void start() { reader_ = std::thread([this] { loop(); }); }
    /* consumer: chunk c (in order, c = 0, 1, ...); nullptr when reading failed (or one pair is longer than a chunk) */
    SeqSlot* wait_ready(int c) {
        SeqSlot& s = slot[c % 3];
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return s.ready || failed_.load(); });
        }
        return failed_ ? nullptr : &s;
    }
    /* consumer: done with chunk c's slot, apart from an asynchronous copy out of it when `in_flight` */
    void consumed(int c, bool in_flight) {
        SeqSlot& s = slot[c % 3];
        s.in_flight = in_flight;
        {
            std::lock_guard<std::mutex> lk(mu_);
            s.ready = false; /* the reader may refill it once wait_shipped has returned */
        }
        cv_.notify_all();
    }
    void stop() {
        {   /* under the mutex: the reader evaluates its wait predicate under it, and a store between its test and its block
               would otherwise be a lost wake-up */
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        if (reader_.joinable()) reader_.join();
    }
    bool failed() const { return failed_.load(); }
    double read_seconds() const { return read_seconds_; } /* after stop() */
};

}  // namespace asm_host
