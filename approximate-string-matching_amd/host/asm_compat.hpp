// Header-only C++ host side over the C ABI (include/asm_mi355x.h), mirroring the reference's interface for the
// hot path so that code written against the reference reads the same:
//   hurdle_matrix<T>  reset/run/get_cost/get_CIGAR GASMA/hurdle_matrix.h:552-562,568,613,625,667,677
//   LV                init/load_reads/reset/run/check_pass/get_ED   GASMA/benchmark/LEAP_SIMD/LV_BAG.h:40-54
//   benchmark         read_string_file/read_answer_file/run/print   GASMA/benchmark/benchmark_utils.h:263-402
//   Dataset           output()                                       GASMA/benchmark/benchmark_dataset.h:189-253
//   SIMD_ED           init_levenshtein/load_reads/calculate_masks/reset/run/check_pass/get_ED   LEAP_SIMD/SIMD_ED.h:47-70
//   leap_simd_filter  the stdin filter driver                        GASMA/benchmark/LEAP_SIMD/main.cpp:31-300
// Same names, argument meaning and (absence of an) error channel as the reference, except that failures of
// the device library throw std::runtime_error instead of being ignored.  Everything computes on the GPU through
// libasm_mi355x.so; there is no CPU path.  The per-pair classes launch one-pair batches (kept for interface
// compatibility; slow by construction) — the performance path is `benchmark`, which runs the whole file as one batch.
// ASM_COMPAT_NO_HARNESS / ASM_COMPAT_NO_DATASET leave `benchmark` / `Dataset` out: for builds in which the reference's OWN
// benchmark_utils.h / benchmark_dataset.h define those classes over the per-pair objects of this header (oracle/Makefile,
// target ref_harness_on_shim).
#pragma once
#include <climits>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <sys/stat.h>
#include <string>
#include <vector>

#include "asm_mi355x.h"

namespace asm_amd {

enum alignment_type_t { GLOBAL, SEMI_GLOBAL, LOCAL };                            // GASMA/utils.h:554-558
enum ED_modes { ED_LOCAL, ED_GLOBAL, ED_SEMI_FREE_BEGIN, ED_SEMI_FREE_END };     // LEAP_SIMD/LV_BAG.h:38
struct int_128bit {};                                                            // tag only: vectors live on the GPU

inline void check(asm_handle* h, int rc) {
    if (rc != ASM_OK) throw std::runtime_error(std::string("asm_mi355x: ") + asm_last_error(h));
}

// One handle per process and GPU, shared by the compat objects.
inline asm_handle* shared_handle(int device = 0) {
    static asm_handle* h = nullptr;
    if (!h) check(nullptr, asm_create(&h, device));
    return h;
}

inline int align_one(int aligner, const char* read, int m, const char* ref, int n, const asm_params& p, int mode) {
    uint32_t ro[2] = {0u, (uint32_t)m}, fo[2] = {0u, (uint32_t)n};
    int32_t out = 0;
    asm_handle* h = shared_handle();
    check(h, asm_align_batch(h, aligner, 1, read, ro, ref, fo, &p, mode, &out));
    return out;
}

template <typename T = int_128bit>
class hurdle_matrix {
    asm_params p_;
    std::string read_, ref_, cigar_;
    int cost_ = 0;
    // The reference object's two 128-byte buffers live on from reset() to reset() and every conversion permutes them in place
    // (hurdle_matrix.h:136-137,630-631, bit_convert.cpp:265-330): a pair sees the stale tails of the pairs this object aligned
    // before.  Only the 2-bit codes matter; this is that state (zeros where the reference's fresh heap memory is indeterminate).
    uint8_t tails_[256] = {0};
    long pairs_run_ = 0;

public:
    explicit hurdle_matrix(alignment_type_t type = GLOBAL, int x = 1, int o = 1, int e = 1, double match_prob = 0.80,
                           double mismatch_prob = 0.20 / 3, double indel_prob = 0.40 / 3) {
        if (type == LOCAL) throw std::runtime_error("hurdle_matrix: LOCAL is unsupported (in the reference too, hurdle_matrix.h:467)");
        asm_default_params(&p_);
        p_.alignment_type = type == SEMI_GLOBAL ? ASM_ALIGN_SEMI_GLOBAL : ASM_ALIGN_GLOBAL;
        p_.x = x, p_.o = o, p_.e = e;
        p_.p_match = match_prob, p_.p_mismatch = mismatch_prob, p_.p_indel = indel_prob;
    }
    // hurdle_matrix.h:473-539: the constructor that takes the pair (GASMA/main.cpp:5-9 uses it); note ITS default
    // probabilities (0.95, 0.02, 0.03), which are not the penalty-only constructor's
    hurdle_matrix(const char* read, const char* ref, int error, alignment_type_t type = GLOBAL, int x = 1, int o = 1, int e = 1,
                  double match_prob = 0.95, double mismatch_prob = 0.02, double indel_prob = 0.03)
        : hurdle_matrix(type, x, o, e, match_prob, mismatch_prob, indel_prob) {
        reset(read, ref, error);
    }
    void reset(const char* read, const int read_len, const char* ref, const int ref_len, int error) {
        read_.assign(read, (size_t)read_len);
        ref_.assign(ref, (size_t)ref_len);
        p_.k = error;
    }
    // hurdle_matrix.h:602-607: "lane %d:" and the lane's 128 hurdle bits (short hurdles flipped, :452-453), position 0 first
    // (utils.h:37-46,78-82).  Display only — nothing is aligned here: the bits are character comparisons of the two strings
    // as the conversion sees them (first 128 bases, codes of bit_convert.cpp:340-355, a lone object's buffer tails are zero).
    void print() const {
        auto code = [](const std::string& s, int i) {
            if (i < 0 || i >= 128 || i >= (int)s.size()) return 0;
            const char c = s[(size_t)i];
            return c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 0;
        };
        for (int lane = -p_.k; lane <= p_.k; lane++) {
            bool raw[130];
            raw[0] = raw[129] = false; /* zeros shifted in at both ends (utils.h:200-216) */
            for (int i = 0; i < 128; i++)
                raw[i + 1] = lane < 0 ? code(read_, i - lane) != code(ref_, i) : code(ref_, i + lane) != code(read_, i);
            printf("lane %d:", lane);
            for (int i = 1; i <= 128; i++) putchar(raw[i] && (raw[i - 1] || raw[i + 1]) ? '1' : '0');
            printf("\n");
        }
    }
    void reset(const char* read, const char* ref, int error) { reset(read, (int)strlen(read), ref, (int)strlen(ref), error); }
    // cost and CIGAR in one device call (hurdle_matrix.h:568-597,613,677).  The object's first pair sees clean tails; every later
    // one the tails the earlier pairs of THIS object left behind, exactly as a reference object reused for a whole file does
    // (benchmark_utils.h:191): a one-pair batch resolved against the carried state (asm_batch_resolve_tails), whose own effect
    // on the buffers (asm_batch_tail_summary) is folded into the state afterwards.
    void run() {
        asm_handle* h = shared_handle();
        uint32_t ro[2] = {0u, (uint32_t)read_.size()}, fo[2] = {0u, (uint32_t)ref_.size()};
        const int cap = 192;
        asm_batch* b = nullptr;
        void *d_cost = nullptr, *d_ops = nullptr, *d_nops = nullptr;
        check(h, asm_batch_upload(h, 1, read_.data(), ro, ref_.data(), fo, ASM_GREEDY_CLEAN, &b));
        uint8_t summary[256];
        int rc2 = asm_batch_tail_summary(h, b, summary);
        if (rc2 == ASM_OK && pairs_run_ > 0) rc2 = asm_batch_resolve_tails(h, b, tails_);
        if (rc2 != ASM_OK) {
            asm_batch_free(h, b);
            check(h, rc2);
        }
        check(h, asm_device_malloc(h, sizeof(int32_t), &d_cost));
        check(h, asm_device_malloc(h, sizeof(uint16_t) * cap, &d_ops));
        check(h, asm_device_malloc(h, 1, &d_nops));
        const int rc = asm_greedy_cigar_batch_async(h, b, &p_, (int32_t*)d_cost, (uint16_t*)d_ops, cap, (uint8_t*)d_nops);
        uint16_t ops[192];
        uint8_t nops = 0;
        int32_t cost = 0;
        if (rc == ASM_OK) {
            check(h, asm_memcpy_d2h(h, &cost, d_cost, sizeof cost));
            check(h, asm_memcpy_d2h(h, ops, d_ops, sizeof ops));
            check(h, asm_memcpy_d2h(h, &nops, d_nops, 1));
        }
        asm_device_free(h, d_cost), asm_device_free(h, d_ops), asm_device_free(h, d_nops);
        asm_batch_free(h, b);
        check(h, rc);
        check(h, asm_tail_state_advance(tails_, summary, 1));
        pairs_run_++;
        cost_ = cost;
        char text[1024];
        check(h, asm_cigar_format(ops, nops, cap, text, sizeof text));
        cigar_ = text;
    }
    int get_cost() const { return cost_; }
    std::string get_CIGAR() const { return cigar_; }
};

class LV {
    asm_params p_;
    std::string read_, ref_;
    int ed_ = -1;

public:
    LV() { asm_default_params(&p_); }
    void init(int gap_threshold, int af_threshold, ED_modes mode, int ms_penalty, int gap_open_penalty, int gap_ext_penalty) {
        if (af_threshold != ASM_LEAP_AF_THRESHOLD)
            throw std::runtime_error("LV::init: the accelerated path has af_threshold 200 (benchmark_utils.h:289)");
        p_.k = gap_threshold, p_.x = ms_penalty, p_.o = gap_open_penalty, p_.e = gap_ext_penalty;
        p_.leap_mode = mode == ED_GLOBAL ? ASM_LEAP_GLOBAL  // LV_BAG.h:38 -> asm_params.leap_mode
                       : mode == ED_LOCAL ? ASM_LEAP_LOCAL
                       : mode == ED_SEMI_FREE_BEGIN ? ASM_LEAP_SEMI_FREE_BEGIN : ASM_LEAP_SEMI_FREE_END;
    }
    void load_reads(char* read, char* ref, int length) {  // NUL-terminated inputs, as the harness passes them
        read_.assign(read, strnlen(read, (size_t)length));
        ref_.assign(ref, strnlen(ref, (size_t)length));
    }
    void reset() { ed_ = -1; }
    void run() { ed_ = align_one(ASM_LEAP, read_.data(), (int)read_.size(), ref_.data(), (int)ref_.size(), p_, ASM_GREEDY_CLEAN); }
    bool check_pass() const { return ed_ >= 0; }
    int get_ED() const { return ed_; }
    // LV_BAG.cpp:251-354,360-383.  The harness calls both inside its timed LEAP section and never reads the string
    // (benchmark_utils.h:169-174,256: coverage uses the Greedy and NW CIGARs only); the reference's get_CIGAR prints
    // ED_info[0].id_length for every operation, i.e. it does not describe the alignment.  Here: nothing to do, empty string.
    void backtrack() {}
    std::string get_CIGAR() const { return std::string(); }
};

// Bit-parallel LEAP in Levenshtein mode (LEAP_SIMD/SIMD_ED.h:47-70).  Like the reference object it keeps its verdict
// state from pair to pair (SIMD_ED.cpp:258-268,348-351); unlike the reference's, the state starts defined (zeros).
class SIMD_ED {
    int ed_t_ = 0;
    bool shd_ = true;
    bool affine_ = false;
    int gap_t_ = 0, af_t_ = 0, x_ = 1, o_ = 1, e_ = 1;
    bool af_shd_ = false;
    int af_shd_t_ = 0, af_mode_ = 0, lev_mode_ = 0;
    int32_t state_[3] = {0, 0, 0};
    std::string read_, ref_;
    int ed_ = -1;

public:
    // SIMD_ED.h:50 / SIMD_ED.cpp:435-616.  CLEAN form: run() judges the pair from the tables init_affine leaves, whatever ran
    // before (the reference object re-uses the tables of the pair before; see asm_simd_ed_affine_batch_async).
    void init_affine(int gap_threshold, int AF_threshold, ED_modes mode, int ms_penalty, int gap_open_penalty, int gap_ext_penalty,
                     bool SHD_enable = false, int SHD_threshold = 10) {
        af_mode_ = mode == ED_GLOBAL ? ASM_LEAP_GLOBAL : mode == ED_LOCAL ? ASM_LEAP_LOCAL
                   : mode == ED_SEMI_FREE_BEGIN ? ASM_LEAP_SEMI_FREE_BEGIN : ASM_LEAP_SEMI_FREE_END;
        affine_ = true, gap_t_ = gap_threshold, af_t_ = AF_threshold, x_ = ms_penalty, o_ = gap_open_penalty, e_ = gap_ext_penalty;
        af_shd_ = SHD_enable, af_shd_t_ = SHD_threshold; /* run_affine's SHD over the first 2*SHD_threshold+1 lane masks (:489-492) */
    }
    void init_levenshtein(int ED_threshold, ED_modes mode = ED_GLOBAL, bool SHD_enable = true) {
        lev_mode_ = mode == ED_GLOBAL ? ASM_LEAP_GLOBAL : mode == ED_LOCAL ? ASM_LEAP_LOCAL
                    : mode == ED_SEMI_FREE_BEGIN ? ASM_LEAP_SEMI_FREE_BEGIN : ASM_LEAP_SEMI_FREE_END;
        ed_t_ = ED_threshold, shd_ = SHD_enable, affine_ = false;
    }
    void load_reads(char* read, char* ref, int length) {  // strncpy semantics: NUL-terminated, at most `length` characters
        read_.assign(read, strnlen(read, (size_t)length));
        ref_.assign(ref, strnlen(ref, (size_t)length));
        read_.resize((size_t)length, '\0');  // buffer_length = length whatever the strings hold (SIMD_ED.cpp:140-151)
    }
    void calculate_masks() {}
    void reset() { ed_ = -1; }
    void run() {
        asm_handle* h = shared_handle();
        uint32_t ro[2] = {0u, (uint32_t)read_.size()}, fo[2] = {0u, (uint32_t)ref_.size()};
        asm_batch* b = nullptr;
        void* d = nullptr;
        int32_t out = -1;
        check(h, asm_batch_upload(h, 1, read_.data(), ro, ref_.data(), fo, ASM_GREEDY_CLEAN, &b));
        check(h, asm_device_malloc(h, sizeof(int32_t), &d));
        const int rc = affine_ ? asm_simd_ed_affine_mode_batch_async(h, b, gap_t_, af_t_, x_, o_, e_, af_shd_ ? af_shd_t_ : -1, af_mode_,
                                                                     (int32_t*)d)
                               : asm_simd_ed_mode_batch_async(h, b, ed_t_, shd_ ? 1 : 0, ASM_FILTER_SEQUENTIAL, lev_mode_, state_, (int32_t*)d);
        if (rc == ASM_OK) check(h, asm_memcpy_d2h(h, &out, d, sizeof(int32_t)));
        asm_device_free(h, d);
        asm_batch_free(h, b);
        check(h, rc);
        ed_ = out;
    }
    bool check_pass() const { return ed_ >= 0; }
    int get_ED() const { return ed_; }
};

// The stdin filter driver (LEAP_SIMD/main.cpp:31-300): pairs of lines (read, reference) until "end_of_file", processed in
// chunks of BATCH_RUN pairs — each chunk one GPU batch, the verdict state chained from chunk to chunk — then
// passNum / totalNum / total_time (GPU time of the filter calls instead of CPU user time).
inline int leap_simd_filter(FILE* in, int error, bool use_shd, int64_t batch_run = 1000000) {
    asm_handle* h = shared_handle();
    int32_t state[3] = {0, 0, 0};
    unsigned long long pass_num = 0, total_num = 0;
    double seconds = 0;
    char* line = nullptr;
    size_t cap = 0;
    bool stop = false;
    void* tm = nullptr;
    check(h, asm_timer_create(h, &tm));
    while (!stop) {
        std::vector<char> reads, refs;
        std::vector<uint32_t> ro{0}, fo{0};
        for (int64_t i = 0; i < batch_run; i++) {
            ssize_t got = getline(&line, &cap, in);
            if (got <= 0) { stop = true; break; }
            if (line[got - 1] == '\n') got--;
            if (got == 11 && !strncmp(line, "end_of_file", 11)) { stop = true; break; }
            reads.insert(reads.end(), line, line + got);
            ro.push_back((uint32_t)reads.size());
            got = getline(&line, &cap, in);
            if (got < 0) got = 0;
            if (got > 0 && line[got - 1] == '\n') got--;
            refs.insert(refs.end(), line, line + got);
            fo.push_back((uint32_t)refs.size());
        }
        const int64_t n = (int64_t)ro.size() - 1;
        if (n == 0) break;
        asm_batch* b = nullptr;
        void* d = nullptr;
        float ms = 0;
        check(h, asm_batch_upload(h, n, reads.data(), ro.data(), refs.data(), fo.data(), ASM_GREEDY_CLEAN, &b));
        check(h, asm_device_malloc(h, sizeof(int32_t) * (size_t)n, &d));
        check(h, asm_timer_start(h, tm));
        check(h, asm_simd_ed_batch_async(h, b, error, use_shd ? 1 : 0, ASM_FILTER_SEQUENTIAL, state, (int32_t*)d));
        check(h, asm_timer_stop(h, tm));
        check(h, asm_timer_elapsed_ms(h, tm, &ms));
        seconds += ms * 1e-3;
        std::vector<int32_t> ed((size_t)n);
        check(h, asm_memcpy_d2h(h, ed.data(), d, sizeof(int32_t) * (size_t)n));
        for (int32_t v : ed) pass_num += v >= 0;
        total_num += (unsigned long long)n;
        asm_device_free(h, d);
        asm_batch_free(h, b);
    }
    free(line);
    asm_timer_destroy(h, tm);
    fprintf(stderr, "end_of_file\n");
    printf("passNum:\t%llu\n", pass_num);
    printf("totalNum:\t%llu\n", total_num);
    printf("total_time: %f\n", seconds);
    return 0;
}

#ifndef ASM_COMPAT_NO_DATASET
// Seeded counterpart of `Dataset` (writes the same ">read\n<ref\n" file; the seed replaces time()).
class Dataset {
    asm_gen_config cfg_{};
    int num_reads_;

public:
    Dataset(int num_reads, int length, float error_rate, float mismatch_rate, bool exact_error_rate = true, uint64_t seed = 1) {
        cfg_.seed = seed, cfg_.kind = exact_error_rate ? ASM_GEN_EXACT_ERRORS : ASM_GEN_UP_TO_ERRORS, cfg_.len_lo = cfg_.len_hi = length;
        cfg_.err = error_rate, cfg_.mismatch_rate = mismatch_rate;
        num_reads_ = num_reads;
    }
    const asm_gen_config& config() const { return cfg_; }
    void output(const char* path) const {
        std::vector<uint32_t> ro((size_t)num_reads_ + 1), fo((size_t)num_reads_ + 1);
        check(nullptr, asm_generate_pairs(&cfg_, 0, num_reads_, ro.data(), fo.data(), nullptr, 0, nullptr, 0));
        std::vector<char> reads(ro.back() + 1), refs(fo.back() + 1);
        check(nullptr, asm_generate_pairs(&cfg_, 0, num_reads_, ro.data(), fo.data(), reads.data(), reads.size(), refs.data(), refs.size()));
        FILE* f = fopen(path, "w");
        if (!f) throw std::runtime_error(std::string("Dataset: cannot write ") + path);
        for (int i = 0; i < num_reads_; i++) {
            fprintf(f, ">%.*s\n<%.*s\n", (int)(ro[i + 1] - ro[i]), reads.data() + ro[i], (int)(fo[i + 1] - fo[i]), refs.data() + fo[i]);
        }
        fclose(f);
    }
    std::string output() const {  // benchmark_dataset.h:242-253 file name, plus the seed
        std::string name = "simulated_" + std::to_string(num_reads_) + "_" + std::to_string(cfg_.len_lo) + "_" +
                           std::to_string(cfg_.err) + (cfg_.kind == ASM_GEN_UP_TO_ERRORS ? "_lt_eq_seed" : "_eq_seed") +
                           std::to_string(cfg_.seed) + ".seq";
        output(name.c_str());
        return name;
    }
};

#endif /* ASM_COMPAT_NO_DATASET */

#ifndef ASM_COMPAT_NO_HARNESS
class benchmark {
    asm_params p_;
    int max_tests_;
    int greedy_mode_;
    std::vector<char> reads_, refs_;
    std::vector<uint32_t> ro_{0}, fo_{0};
    std::vector<int32_t> answers_;
    unsigned long long counters_[4] = {0, 0, 0, 0};
    unsigned long long cover_[2] = {0, 0}; /* covered, not determined */
    bool have_cover_ = false;
    float ms_[3] = {0, 0, 0};
    std::vector<int32_t> pen_[3];
    bool streamed_ = false;
    double stream_seconds_ = 0, stream_read_seconds_ = 0;
    int64_t stream_chunks_ = 0, stream_bytes_ = 0;

public:
    // benchmark_utils.h:263-269; greedy_mode: ASM_GREEDY_SEQUENTIAL reproduces the reference's run order dependence
    benchmark(int x, int o, int e, int k, int max_test_num, bool /*use_SIMD*/ = true, int greedy_mode = ASM_GREEDY_SEQUENTIAL)
        : max_tests_(max_test_num), greedy_mode_(greedy_mode) {
        asm_default_params(&p_);
        p_.x = x, p_.o = o, p_.e = e, p_.k = k;
    }
    // benchmark_utils.h:325-352: the first character of every line is skipped blindly
    void read_string_file(const char* path, bool skip_first_char = true) {
        std::ifstream in(path);
        if (!in.is_open()) {
            printf("Unable to open data file: %s\n", path);
            return;
        }
        std::string a, b;
        int i = 0;
        for (; i < max_tests_; i++) {
            if (!std::getline(in, a)) break;
            std::getline(in, b);
            const size_t s = skip_first_char ? 1 : 0;
            if (a.size() >= s) reads_.insert(reads_.end(), a.begin() + s, a.end());
            if (b.size() >= s) refs_.insert(refs_.end(), b.begin() + s, b.end());
            ro_.push_back((uint32_t)reads_.size());
            fo_.push_back((uint32_t)refs_.size());
        }
        max_tests_ = i;
        printf("Processed data file: %s\n", path);
    }
    // benchmark_utils.h:358-368
    void read_answer_file(const char* path) {
        std::ifstream in(path);
        std::string line;
        answers_.assign((size_t)max_tests_, INT32_MIN);
        for (int i = 0; i < max_tests_ && std::getline(in, line); i++) answers_[(size_t)i] = atoi(line.c_str());
    }
    // benchmark_utils.h:373-385 — the whole file as ONE batch on the GPU
    void run() {
        asm_handle* h = shared_handle();
        const int64_t n = (int64_t)ro_.size() - 1;
        asm_batch* b = nullptr;
        check(h, asm_batch_upload(h, n, reads_.data(), ro_.data(), refs_.data(), fo_.data(), greedy_mode_, &b));
        void *d_pen[3], *d_cnt = nullptr, *d_ans = nullptr, *tm = nullptr;
        for (auto& d : d_pen) check(h, asm_device_malloc(h, sizeof(int32_t) * (size_t)(n > 0 ? n : 1), &d));
        check(h, asm_device_malloc(h, 32, &d_cnt));
        check(h, asm_memset_async(h, d_cnt, 0, 32));
        if (!answers_.empty()) {
            check(h, asm_device_malloc(h, sizeof(int32_t) * answers_.size(), &d_ans));
            check(h, asm_memcpy_h2d(h, d_ans, answers_.data(), sizeof(int32_t) * answers_.size()));
        }
        check(h, asm_timer_create(h, &tm));
        for (int a = 0; a < 3; a++) {
            check(h, asm_timer_start(h, tm));
            check(h, asm_align_batch_async(h, b, a, &p_, (int32_t*)d_pen[a]));
            check(h, asm_timer_stop(h, tm));
            check(h, asm_timer_elapsed_ms(h, tm, &ms_[a]));
        }
        check(h, asm_accuracy_async(h, (int32_t*)d_pen[0], (int32_t*)d_pen[1], (int32_t*)d_pen[2], (int32_t*)d_ans, n,
                                    (unsigned long long*)d_cnt));
        check(h, asm_memcpy_d2h(h, counters_, d_cnt, 32));
        // [Coverage] (benchmark_utils.h:256-258): Greedy CIGAR + NW traceback on the device, any penalties
        if (n > 0) {
            const int cap = 96;
            void *d_ops = nullptr, *d_nops = nullptr, *d_cov = nullptr, *d_cc = nullptr;
            check(h, asm_device_malloc(h, sizeof(uint16_t) * (size_t)cap * (size_t)n, &d_ops));
            check(h, asm_device_malloc(h, (size_t)n, &d_nops));
            check(h, asm_device_malloc(h, (size_t)n, &d_cov));
            check(h, asm_device_malloc(h, 16, &d_cc));
            check(h, asm_memset_async(h, d_cc, 0, 16));
            check(h, asm_greedy_cigar_batch_async(h, b, &p_, (int32_t*)d_pen[2], (uint16_t*)d_ops, cap, (uint8_t*)d_nops));
            check(h, asm_coverage(h, b, &p_, (uint16_t*)d_ops, cap, (uint8_t*)d_nops, 64, (uint8_t*)d_cov, nullptr, 0, nullptr,
                                  (unsigned long long*)d_cc));
            check(h, asm_memcpy_d2h(h, cover_, d_cc, 16));
            have_cover_ = true;
            asm_device_free(h, d_ops), asm_device_free(h, d_nops), asm_device_free(h, d_cov), asm_device_free(h, d_cc);
        }
        for (int a = 0; a < 3; a++) {
            pen_[a].resize((size_t)n);
            if (n) check(h, asm_memcpy_d2h(h, pen_[a].data(), d_pen[a], sizeof(int32_t) * (size_t)n));
            asm_device_free(h, d_pen[a]);
        }
        asm_device_free(h, d_cnt);
        if (d_ans) asm_device_free(h, d_ans);
        asm_timer_destroy(h, tm);
        asm_batch_free(h, b);
        printf("...complete.\n");
    }
    // The same work for a file of any size, streamed (asm_stream_seq_file): reader threads -> pinned buffers -> HBM while the
    // chunk before is parsed, packed and aligned; end-to-end pairs/s include reading the file.  No [Coverage] line here.
    void run_streamed(const char* path, int64_t chunk_bytes = 0) {
        asm_handle* h = shared_handle();
        asm_stream_stats st;
        std::vector<int32_t> out[3];
        struct stat sb;
        const int64_t cap = (stat(path, &sb) == 0 ? (int64_t)sb.st_size / 4 : 0) + 16;
        const int64_t room = max_tests_ > 0 && max_tests_ < cap ? max_tests_ : cap;
        for (auto& v : out) v.resize((size_t)room);
        check(h, asm_stream_seq_file(h, path, &p_, greedy_mode_, 7, chunk_bytes, max_tests_, out[0].data(), out[1].data(), out[2].data(),
                                     room, answers_.empty() ? nullptr : answers_.data(), (int64_t)answers_.size(), &st));
        for (int a = 0; a < 3; a++) {
            out[a].resize((size_t)st.pairs);
            pen_[a].swap(out[a]);
        }
        memcpy(counters_, st.counters, sizeof counters_);
        have_cover_ = false;
        streamed_ = true;
        stream_seconds_ = st.seconds, stream_read_seconds_ = st.seconds_read, stream_chunks_ = st.chunks, stream_bytes_ = st.bytes;
        printf("...complete.\n");
    }
    const std::vector<int32_t>& penalties(int aligner) const { return pen_[aligner]; }
    // benchmark_utils.h:390-402 — same block; [Time] is GPU kernel time (HIP events) instead of CPU user time
    void print() const {
        const double total = (double)counters_[0];
        printf("===================== Benchmark Results =====================\n");
        printf("Total number of alignments: %d\n[Time]\n", (int)counters_[0]);
        printf("=> Needleman-Wunsch | %.3f s\n", ms_[0] * 1e-3);
        printf("=> LEAP             | %.3f s\n", ms_[1] * 1e-3);
        printf("=> Greedy           | %.3f s\n", ms_[2] * 1e-3);
        printf("[Accuracy] (percentage of alignments matching optimal penalty)\n");
        printf("=> Needleman-Wunsch | %.3f %%\n", (double)counters_[1] / total * 100);
        printf("=> LEAP             | %.3f %%\n", (double)counters_[2] / total * 100);
        printf("=> Greedy           | %.3f %%\n", (double)counters_[3] / total * 100);
        printf("[Coverage] (percentage of alignments covering all long consecutive matches)\n");
        if (have_cover_) {
            printf("=> Greedy           | %.3f %%\n", (double)cover_[0] / (total - (double)cover_[1]) * 100);
            if (cover_[1])
                printf("   (%llu pairs beyond the traceback band are excluded; NW tie-break is this library's, not parasail's)\n",
                       cover_[1]);
        } else {
            printf("=> Greedy           | not computed (streamed run)\n");
        }
        if (streamed_)
            printf("[Streamed] %lld pairs, %lld chunks, %.1f MB in %.3f s = %.3e pairs/s end to end (reader threads busy %.3f s)\n",
                   (long long)counters_[0], (long long)stream_chunks_, stream_bytes_ / 1e6, stream_seconds_,
                   (double)counters_[0] / stream_seconds_, stream_read_seconds_);
        else
            printf("[GPU kernel time] NW %.3f ms | LEAP %.3f ms | Greedy %.3f ms\n", ms_[0], ms_[1], ms_[2]);
    }
};
#endif /* ASM_COMPAT_NO_HARNESS */

}  // namespace asm_amd
