// TEST PROGRAM for CPU sanitizers (`make -C oracle asan`; tests/test_sanitizers.py runs the two builds): the device-free host
// side of the C ABI — csrc/asm_host.h, the very code libasm_mi355x.so compiles — exercised without a GPU:
//   * the generator's host loop and the `>read\n<ref\n` text it stands for,
//   * the streaming reader (asm_stream_seq_file's reader pool + three-slot hand-over) against a consumer that "ships" each
//     chunk asynchronously on a thread of its own, the way the copy stream does: every byte of the file must come out once, in
//     order, cut at pair boundaries, for several chunk sizes, reader counts and max_pairs cuts, incl. files that end without
//     a newline or on a read line,
//   * the stale-tail state arithmetic and the CIGAR formatter at their edges.
// Built twice: -fsanitize=thread (races in the hand-over / the pool) and -fsanitize=address,undefined (buffer edges).
// Usage: asm_host_check <scratch directory>.  Prints "host check ok" and exits 0, or says what differed and exits 1.
#include <sys/stat.h>

#include <cstdlib>
#include <deque>
#include <future>
#include <string>
#include <vector>

#include "../csrc/asm_host.h"

using namespace asm_host;

static int g_fail = 0;
#define EXPECT(cond, ...)                                      \
    do {                                                       \
        if (!(cond)) {                                         \
            fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);                      \
            fprintf(stderr, "\n");                             \
            g_fail++;                                          \
        }                                                      \
    } while (0)

static std::string make_text(const asm_gen_config& cfg, int64_t first, int64_t n, int64_t* pairs_out) {
    std::string err;
    std::vector<uint32_t> ro((size_t)n + 1), fo((size_t)n + 1);
    int rc = generate_pairs(&cfg, first, n, ro.data(), fo.data(), nullptr, 0, nullptr, 0, err);
    EXPECT(rc == ASM_OK, "sizing pass: %s", err.c_str());
    std::vector<char> reads(ro.back() + 1), refs(fo.back() + 1);
    rc = generate_pairs(&cfg, first, n, ro.data(), fo.data(), reads.data(), reads.size(), refs.data(), refs.size(), err);
    EXPECT(rc == ASM_OK, "fill pass: %s", err.c_str());
    std::string text;
    for (int64_t i = 0; i < n; i++) {
        text += '>';
        text.append(reads.data() + ro[(size_t)i], ro[(size_t)i + 1] - ro[(size_t)i]);
        text += "\n<";
        text.append(refs.data() + fo[(size_t)i], fo[(size_t)i + 1] - fo[(size_t)i]);
        text += '\n';
    }
    *pairs_out = n;
    return text;
}

/* Streams `text` (written to `path`) through SeqReader; returns what the consumer received, concatenated. */
static std::string stream_file(const std::string& path, const std::string& text, size_t chunk, int readers, int64_t max_pairs,
                               int64_t* pairs_seen, int* chunks_seen, bool* failed, size_t slack = (size_t)4 << 10, size_t first_chunk = 0) {
    FILE* f = fopen(path.c_str(), "wb");
    fwrite(text.data(), 1, text.size(), f);
    fclose(f);
    const int fd = open(path.c_str(), O_RDONLY);
    const size_t cap = chunk + slack;
    std::vector<std::vector<char>> bufs(3, std::vector<char>(cap + 64));
    std::future<void> copy[3]; /* the "copy stream": an asynchronous reader of the slot's buffer */
    SeqReader rd(fd, text.size(), chunk, readers, max_pairs, [&](int q) {
        if (copy[q].valid()) copy[q].wait();
    }, first_chunk);
    for (int q = 0; q < 3; q++) rd.slot[q].buf = bufs[(size_t)q].data(), rd.slot[q].cap = cap;
    rd.start();
    std::deque<std::string> parts; /* a deque: elements stay where they are while asynchronous copies write into them */
    *pairs_seen = 0, *chunks_seen = 0, *failed = false;
    bool last = false;
    for (int c = 0; !last; c++) {
        SeqSlot* s = rd.wait_ready(c);
        if (!s) {
            *failed = true;
            break;
        }
        last = s->last;
        parts.emplace_back();
        std::string* dst = &parts.back();
        const char* src = s->buf;
        const size_t bytes = s->bytes;
        *pairs_seen += s->pairs, *chunks_seen += 1;
        /* the slot's previous copy (three chunks ago) was awaited by the reader before it refilled the buffer */
        copy[c % 3] = std::async(std::launch::async, [dst, src, bytes] { dst->assign(src, bytes); });
        rd.consumed(c, true);
        if (c % 2) std::this_thread::yield();
    }
    for (auto& fu : copy)
        if (fu.valid()) fu.wait();
    rd.stop();
    close(fd);
    std::string out;
    for (const std::string& p : parts) out += p;
    return out;
}

static std::string first_pairs(const std::string& text, int64_t pairs) { /* the text of the first `pairs` pairs */
    size_t pos = 0;
    for (int64_t l = 0; l < 2 * pairs; l++) {
        const size_t nl = text.find('\n', pos);
        if (nl == std::string::npos) return text;
        pos = nl + 1;
    }
    return text.substr(0, pos);
}

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const std::string path = dir + "/asm_host_check.seq";
    asm_gen_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.seed = 7, cfg.kind = ASM_GEN_EXACT_ERRORS, cfg.len_lo = 64, cfg.len_hi = 300, cfg.err = 0.10f, cfg.mismatch_rate = 0.96f;
    int64_t n = 0;
    const std::string text = make_text(cfg, 123, 6000, &n);
    {   /* the same pairs whichever slice of the stream asks for them */
        int64_t n2 = 0;
        const std::string tail = make_text(cfg, 123 + 5000, 1000, &n2);
        EXPECT(text.size() > tail.size() && text.compare(text.size() - tail.size(), tail.size(), tail) == 0, "generator: slices disagree");
        std::string err;
        asm_gen_config bad = cfg;
        bad.err = 0.9f;
        uint32_t off[2];
        EXPECT(generate_pairs(&bad, 0, 1, off, off, nullptr, 0, nullptr, 0, err) == ASM_EINVAL, "generator accepted err 0.9");
    }
    EXPECT(scan_newlines(text.data(), text.size(), 5).count == 2 * n, "scan_newlines count");

    struct Case {
        size_t chunk;
        int readers;
        int64_t max_pairs;
    };
    const Case cases[] = {{4096, 1, 0}, {4096, 4, 0}, {65536, 3, 0}, {1 << 20, 8, 0}, {8192, 2, 777}, {4 << 20, 4, 0}, {50000, 5, 5999}};
    for (const Case& c : cases) {
        int64_t pairs = 0;
        int chunks = 0;
        bool failed = false;
        const std::string got = stream_file(path, text, c.chunk, c.readers, c.max_pairs, &pairs, &chunks, &failed);
        const int64_t want_pairs = c.max_pairs > 0 && c.max_pairs < n ? c.max_pairs : n;
        EXPECT(!failed, "stream failed (chunk %zu)", c.chunk);
        EXPECT(pairs == want_pairs, "chunk %zu readers %d: %lld pairs, want %lld", c.chunk, c.readers, (long long)pairs, (long long)want_pairs);
        EXPECT(got == first_pairs(text, want_pairs), "chunk %zu readers %d: bytes differ (%zu against %zu)", c.chunk, c.readers, got.size(),
               first_pairs(text, want_pairs).size());
    }
    {   /* chunks that ramp up from a small first one: the same bytes, more chunks than the full size alone would give */
        int64_t pairs = 0;
        int chunks_flat = 0, chunks_ramp = 0;
        bool failed = false;
        std::string got = stream_file(path, text, 1 << 19, 3, 0, &pairs, &chunks_flat, &failed);
        EXPECT(!failed && pairs == n && got == text, "flat 512 KB chunks");
        got = stream_file(path, text, 1 << 19, 3, 0, &pairs, &chunks_ramp, &failed, (size_t)4 << 10, 1 << 15);
        EXPECT(!failed && pairs == n && got == text, "ramped chunks: %lld pairs", (long long)pairs);
        EXPECT(chunks_ramp > chunks_flat, "ramp: %d chunks against %d", chunks_ramp, chunks_flat);
        got = stream_file(path, text, 1 << 19, 2, 4321, &pairs, &chunks_ramp, &failed, (size_t)4 << 10, 600); /* a first chunk of barely a pair */
        EXPECT(!failed && pairs == 4321 && got == first_pairs(text, 4321), "ramp from 600 bytes: %lld pairs", (long long)pairs);
    }
    {   /* a last line without its newline; a file that ends on a read line (the reference gets an empty string there) */
        int64_t pairs = 0;
        int chunks = 0;
        bool failed = false;
        std::string open_end = text.substr(0, text.size() - 1);
        std::string got = stream_file(path, open_end, 30000, 3, 0, &pairs, &chunks, &failed);
        EXPECT(!failed && pairs == n && got == text, "missing final newline: %lld pairs", (long long)pairs);
        std::string odd = text + ">ACGT\n";
        got = stream_file(path, odd, 30000, 3, 0, &pairs, &chunks, &failed);
        EXPECT(!failed && pairs == n + 1 && got == odd + "\n", "odd line count: %lld pairs", (long long)pairs);
        /* a pair longer than a whole slot is refused, not cut */
        got = stream_file(path, text, 256, 2, 0, &pairs, &chunks, &failed, 16);
        EXPECT(failed, "a chunk smaller than one pair must fail");
    }
    {   /* stale-tail state: advancing over a + b untouched pairs = advancing over a, then b; a write lands where its slot goes */
        std::string err;
        uint8_t none[256], st1[256], st2[256];
        memset(none, TAIL_NONE, sizeof none);
        for (int q = 0; q < 256; q++) st1[q] = st2[q] = (uint8_t)(q & 3);
        EXPECT(tail_state_advance(st1, none, 7, err) == ASM_OK && tail_state_advance(st1, none, 14, err) == ASM_OK, "advance");
        EXPECT(tail_state_advance(st2, none, 21, err) == ASM_OK, "advance");
        EXPECT(memcmp(st1, st2, 256) == 0, "tail_state_advance is not additive over untouched pairs");
        for (int s = 0; s < 128; s++) EXPECT(tail_slot_after(tail_slot_after(s, 3), 7) == s, "SRC does not have order 10 at slot %d", s);
        uint8_t sum[256];
        memset(sum, TAIL_NONE, sizeof sum);
        sum[5] = 2, sum[128 + 127] = 1;
        EXPECT(tail_state_advance(st2, sum, 1, err) == ASM_OK, "advance with writes");
        EXPECT(st2[tail_slot_after(5, 1)] == 2 && st2[128 + tail_slot_after(127, 1)] == 1, "a written code did not land on its trajectory");
        sum[9] = 17;
        EXPECT(tail_state_advance(st2, sum, 1, err) == ASM_EINVAL, "summary entry 17 accepted");
        EXPECT(tail_state_advance(nullptr, sum, 1, err) == ASM_EINVAL, "NULL state accepted");
    }
    {   /* the resolver's bit-plane step (tail_permute, tail_prefix: what csrc/asm_tails.h runs per pair) against the byte buffer
           it stands for: copy the first L codes in, then after[q] = before[8 * (q & 15) + P[q >> 4]] (bit_convert.cpp:265-330) */
        static const int P[8] = {0, 2, 1, 3, 4, 6, 5, 7};
        uint8_t buf[128] = {0};
        TailBits s0{}, s1{};
        uint64_t rng = 88172645463325252ull;
        auto next = [&rng] { return rng ^= rng << 13, rng ^= rng >> 7, rng ^= rng << 17, (uint32_t)(rng >> 11); };
        int bad = 0;
        for (int t = 0; t < 2000 && !bad; t++) {
            const uint32_t L = t % 7 == 0 ? (t % 14 ? 128u : 0u) : next() % 129u;
            TailBits a0{}, a1{};
            for (uint32_t q = 0; q < L; q++) {
                const uint8_t code = (uint8_t)(next() & 3u);
                buf[q] = code;
                a0.w[q >> 5] |= (uint32_t)(code & 1u) << (q & 31), a1.w[q >> 5] |= (uint32_t)(code >> 1) << (q & 31);
            }
            const TailBits m = tail_prefix(L);
            for (uint32_t q = 0; q < 128; q++) { /* what the conversion sees: the string, then the stale codes */
                const uint32_t want = buf[q], bit = 1u << (q & 31);
                const uint32_t t0 = q < L ? a0.w[q >> 5] : (s0.w[q >> 5] & ~m.w[q >> 5]), t1 = q < L ? a1.w[q >> 5] : (s1.w[q >> 5] & ~m.w[q >> 5]);
                if ((((t0 & bit) ? 1u : 0u) | ((t1 & bit) ? 2u : 0u)) != want) bad++;
            }
            uint8_t after[128];
            for (int q = 0; q < 128; q++) after[q] = buf[8 * (q & 15) + P[q >> 4]];
            memcpy(buf, after, 128);
            TailBits n0{}, n1{};
            for (int d = 0; d < 4; d++) n0.w[d] = tail_bfi(m.w[d], a0.w[d], s0.w[d]), n1.w[d] = tail_bfi(m.w[d], a1.w[d], s1.w[d]);
            s0 = tail_permute(n0), s1 = tail_permute(n1);
        }
        EXPECT(bad == 0, "bit-plane buffer step differs from the byte buffer");
        TailBits x{{0x12345678u, 0x9abcdef0u, 0x0fedcba9u, 0x87654321u}}, y = x;
        for (int r = 0; r < 10; r++) y = tail_permute(y);
        EXPECT(memcmp(&x, &y, sizeof x) == 0, "tail_permute does not have order 10");
    }
    {   /* CIGAR rows: formatting, truncated rows, output buffers of every size down to one byte */
        const uint16_t ops[5] = {(uint16_t)(22 << 3 | 0), (uint16_t)(1 << 3 | 2), (uint16_t)(50 << 3 | 0), (uint16_t)(1 << 3 | 1), (uint16_t)(128 << 3 | 4)};
        char out[64];
        EXPECT(cigar_format(ops, 5, 8, out, sizeof out) == ASM_OK && std::string(out) == "22M1D50M1I128X", "cigar text: %s", out);
        EXPECT(cigar_format(ops, 9, 5, out, sizeof out) == ASM_EUNSUPPORTED && std::string(out) == "22M1D50M1I128X", "truncated row");
        EXPECT(cigar_format(ops, 0, 5, out, sizeof out) == ASM_OK && out[0] == 0, "empty row");
        for (size_t cap = 1; cap <= 15; cap++) {
            std::vector<char> small(cap); /* exactly `cap` bytes on the heap: an overrun is an ASan report */
            const int rc = cigar_format(ops, 5, 8, small.data(), cap);
            EXPECT(rc == (cap >= 15 ? ASM_OK : ASM_EINVAL), "out_cap %zu: rc %d", cap, rc);
        }
        EXPECT(cigar_format(nullptr, 1, 1, out, sizeof out) == ASM_EINVAL && cigar_format(ops, 1, 1, out, 0) == ASM_EINVAL, "bad arguments");
    }
    remove(path.c_str());
    if (g_fail) {
        fprintf(stderr, "%d host checks failed\n", g_fail);
        return 1;
    }
    printf("host check ok\n");
    return 0;
}
