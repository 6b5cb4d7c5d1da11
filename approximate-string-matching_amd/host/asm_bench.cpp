// asm-bench — the counterpart of the reference's `hurdle-matrix-benchmark` (GASMA/benchmark/benchmark.cpp:12-32):
// generate (or read) a ">read\n<ref\n" file, run NW + LEAP + Greedy over it, print the results block.
//   asm-bench [--file path | --n N --len L --err E --seed S] [--k K --x X --o O --e E] [--mode sequential|clean]
//             [--answers path] [--stream [--chunk-mb M]]      --stream: the file goes through asm_stream_seq_file
//   asm-bench --gpus N [--n PAIRS_PER_GPU --len L --err E --seed S --steps K]   the hot path on N GPUs of this node: one host
//                                                                        thread and one handle per GPU, rank r generates and
//                                                                        aligns pairs [r*n, (r+1)*n) of the seeded stream, one
//                                                                        RCCL all-reduce of the four counters over xGMI
//   asm-bench --pair READ REF [--k K]                                     the per-pair classes of the reference on one pair
//                                                                        (hurdle_matrix reset/run/get_cost/get_CIGAR, LV, SIMD_ED)
//   asm-bench --leap-simd ERROR [--shd 0|1] [--batch-run N] < pairs      the LEAP_SIMD stdin filter driver
//                                                                        (GASMA/benchmark/LEAP_SIMD/main.cpp:52-101)
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "asm_compat.hpp"

// The multi-GPU form of the harness loop: pairs are independent, so rank r owns a contiguous shard of the seeded stream,
// generated on its own device; the only exchange is the sum of {total_tests, nw_correct, LEAP_correct, greedy_correct}
// (benchmark_utils.h:238,249-255) — 32 bytes, one ncclAllReduce over the direct xGMI links.
// A rank that fails before the collective must not leave the others blocked in it: every rank arrives at a host-side barrier
// first, and the all-reduce is enqueued only if nobody has failed by then.
struct RankBarrier {
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0, generation = 0, parties;
    explicit RankBarrier(int n) : parties(n) {}
    void arrive_and_wait() {
        std::unique_lock<std::mutex> lk(mu);
        const int gen = generation;
        if (++waiting == parties) {
            waiting = 0, generation++;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen; });
        }
    }
};

static int run_multi_gpu(int gpus, long n, int len, float err, uint64_t seed, const asm_params& p, int mode, int steps) {
    int have = asm_device_count();
    if (have < gpus) {
        fprintf(stderr, "--gpus %d but only %d GPU(s) visible\n", gpus, have);
        return 1;
    }
    std::vector<ncclComm_t> comm((size_t)gpus);
    std::vector<int> devs((size_t)gpus);
    for (int r = 0; r < gpus; r++) devs[(size_t)r] = r;
    if (ncclCommInitAll(comm.data(), gpus, devs.data()) != ncclSuccess) {
        fprintf(stderr, "ncclCommInitAll failed\n");
        return 1;
    }
    std::vector<double> ms((size_t)gpus, 0.0);
    std::vector<unsigned long long> sums((size_t)gpus * 4, 0ull);
    std::vector<int> rcs((size_t)gpus, 0);
    std::vector<std::thread> ranks;
    std::atomic<int> failed(0);
    RankBarrier barrier(gpus);
    std::vector<uint8_t> summaries((size_t)gpus * 256, 0); /* sequential mode: what each shard does to the reference's buffers */
    const char* inject = getenv("ASM_BENCH_FAIL_RANK"); /* test hook: this rank reports a failure before the collective */
    const int fail_rank = inject ? atoi(inject) : -1;
    asm_gen_config cfg{};
    cfg.seed = seed, cfg.kind = ASM_GEN_EXACT_ERRORS, cfg.len_lo = cfg.len_hi = len, cfg.err = err, cfg.mismatch_rate = 0.96f;
    for (int r = 0; r < gpus; r++)
        ranks.emplace_back([&, r]() {
            asm_handle* h = nullptr;
            asm_batch* b = nullptr;
            void *d_pen[3] = {nullptr, nullptr, nullptr}, *d_cnt = nullptr;
            hipStream_t stream = nullptr;
            int rc = asm_create(&h, r);
            if (!rc && hipSetDevice(r) != hipSuccess) rc = ASM_ENODEVICE;
            if (!rc && hipStreamCreate(&stream) != hipSuccess) rc = ASM_ENODEVICE;
            if (!rc) rc = asm_set_stream(h, stream); /* the library's kernels and the collective on ONE stream */
            if (!rc) rc = asm_batch_generate(h, &cfg, (int64_t)r * n, n, mode, &b);
            /* Greedy's sequential mode is the reference run over the WHOLE stream: the shards are chained exactly as bench.py and
             * the Python ranks do it — every shard's 256-byte summary is exchanged (here through host memory, the ranks being
             * threads), each rank folds the shards before its own and resolves its tails from that state (include/asm_mi355x.h) */
            if (mode == ASM_GREEDY_SEQUENTIAL) {
                if (!rc) rc = asm_batch_tail_summary(h, b, &summaries[(size_t)r * 256]);
                if (rc) failed = 1;
                barrier.arrive_and_wait();
                if (!rc && !failed) {
                    uint8_t state[256];
                    memset(state, 0, sizeof state);
                    for (int q = 0; q < r && !rc; q++) rc = asm_tail_state_advance(state, &summaries[(size_t)q * 256], n);
                    if (!rc) rc = asm_batch_resolve_tails(h, b, state);
                }
            }
            for (auto& d : d_pen)
                if (!rc) rc = asm_device_malloc(h, sizeof(int32_t) * (size_t)n, &d);
            if (!rc) rc = asm_device_malloc(h, 32, &d_cnt);
            if (!rc) rc = asm_memset_async(h, d_cnt, 0, 32);
            if (!rc) rc = asm_run_benchmark_async(h, b, &p, 1, (int32_t*)d_pen[0], (int32_t*)d_pen[1], (int32_t*)d_pen[2], nullptr, nullptr); /* warm-up */
            if (!rc) rc = asm_synchronize(h);
            const auto t0 = std::chrono::steady_clock::now();
            for (int s = 0; s < steps && !rc; s++)
                rc = asm_run_benchmark_async(h, b, &p, 1, (int32_t*)d_pen[0], (int32_t*)d_pen[1], (int32_t*)d_pen[2], nullptr,
                                             (unsigned long long*)d_cnt);
            if (r == fail_rank) rc = ASM_EINVAL;
            if (rc) failed = 1;
            barrier.arrive_and_wait(); /* nobody enters the collective unless everybody can */
            if (!rc && failed) rc = ASM_ENODEVICE;
            if (!rc && ncclAllReduce(d_cnt, d_cnt, 4, ncclUint64, ncclSum, comm[(size_t)r], stream) != ncclSuccess) rc = ASM_ENODEVICE;
            if (!rc) rc = asm_synchronize(h);
            ms[(size_t)r] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (!rc) rc = asm_memcpy_d2h(h, &sums[(size_t)r * 4], d_cnt, 32);
            for (auto d : d_pen)
                if (d) asm_device_free(h, d);
            if (d_cnt) asm_device_free(h, d_cnt);
            if (b) asm_batch_free(h, b);
            if (h) asm_destroy(h);
            if (stream) (void)hipStreamDestroy(stream);
            rcs[(size_t)r] = rc;
        });
    for (auto& t : ranks) t.join();
    for (int r = 0; r < gpus; r++) {
        if (failed) ncclCommAbort(comm[(size_t)r]);
        else ncclCommDestroy(comm[(size_t)r]);
    }
    double worst = 0;
    for (int r = 0; r < gpus; r++) {
        if (rcs[(size_t)r]) {
            fprintf(stderr, "rank %d failed (%d)\n", r, rcs[(size_t)r]);
            return 1;
        }
        worst = ms[(size_t)r] > worst ? ms[(size_t)r] : worst;
    }
    const unsigned long long* c = &sums[0]; /* every rank holds the same sums after the all-reduce */
    printf("===================== Benchmark Results =====================\n");
    printf("Total number of alignments: %llu  (%d GPUs x %ld pairs x %d steps)\n", c[0], gpus, n, steps);
    printf("[Accuracy] (percentage of alignments matching optimal penalty)\n");
    printf("=> Needleman-Wunsch | %.3f %%\n=> LEAP             | %.3f %%\n=> Greedy           | %.3f %%\n", 100.0 * c[1] / c[0],
           100.0 * c[2] / c[0], 100.0 * c[3] / c[0]);
    printf("[Throughput] %.3f ms per step on the slowest rank, %.3e pairs/s through NW + LEAP + Greedy over %d GPUs\n", worst / steps,
           (double)c[0] / (worst * 1e-3), gpus);
    for (int r = 1; r < gpus; r++)
        if (memcmp(&sums[(size_t)r * 4], c, 32) != 0) {
            fprintf(stderr, "rank %d holds different sums after the all-reduce\n", r);
            return 1;
        }
    return 0;
}

int main(int argc, char** argv) {
    using namespace asm_amd;
    std::string file, answers, mode = "sequential";
    int leap_simd = -1, shd = 1;
    std::string pair_read, pair_ref;
    bool pair_mode = false;
    long batch_run = 1000000;
    bool stream = false;
    long chunk_mb = 0;
    int gpus = 0, steps = 20;
    int n = 1000000, len = 100, k = 3, x = 1, o = 1, e = 1;
    float err = 0.10f;
    uint64_t seed = 2;
    for (int i = 1; i < argc; i++) {
        auto arg = [&](const char* name) { return !strcmp(argv[i], name) && i + 1 < argc; };
        if (arg("--file")) file = argv[++i];
        else if (arg("--answers")) answers = argv[++i];
        else if (arg("--n")) n = atoi(argv[++i]);
        else if (arg("--len")) len = atoi(argv[++i]);
        else if (arg("--err")) err = (float)atof(argv[++i]);
        else if (arg("--seed")) seed = strtoull(argv[++i], nullptr, 10);
        else if (arg("--k")) k = atoi(argv[++i]);
        else if (arg("--x")) x = atoi(argv[++i]);
        else if (arg("--o")) o = atoi(argv[++i]);
        else if (arg("--e")) e = atoi(argv[++i]);
        else if (arg("--mode")) mode = argv[++i];
        else if (!strcmp(argv[i], "--pair") && i + 2 < argc) pair_mode = true, pair_read = argv[++i], pair_ref = argv[++i];
        else if (arg("--leap-simd")) leap_simd = atoi(argv[++i]);
        else if (arg("--shd")) shd = atoi(argv[++i]);
        else if (arg("--batch-run")) batch_run = atol(argv[++i]);
        else if (!strcmp(argv[i], "--stream")) stream = true;
        else if (arg("--chunk-mb")) chunk_mb = atol(argv[++i]);
        else if (arg("--gpus")) gpus = atoi(argv[++i]);
        else if (arg("--steps")) steps = atoi(argv[++i]);
        else {
            fprintf(stderr, "unknown argument %s\n", argv[i]);
            return 2;
        }
    }
    try {
        if (leap_simd >= 0) return leap_simd_filter(stdin, leap_simd, shd != 0, batch_run);
        if (gpus > 0) {
            asm_params p;
            asm_default_params(&p);
            p.k = k, p.x = x, p.o = o, p.e = e;
            return run_multi_gpu(gpus, n, len, err, seed, p, mode == "sequential" ? ASM_GREEDY_SEQUENTIAL : ASM_GREEDY_CLEAN, steps);
        }
        if (pair_mode) {  // GASMA/main.cpp:5-17 in the reference's own vocabulary
            hurdle_matrix<int_128bit> greedy(GLOBAL, x, o, e);
            greedy.reset(pair_read.c_str(), pair_ref.c_str(), k);
            greedy.run();
            printf("greedy cost %d CIGAR %s\n", greedy.get_cost(), greedy.get_CIGAR().c_str());
            LV lv;
            lv.init(k, 200, ED_GLOBAL, x, o, e);
            const int length = (int)(pair_read.size() > pair_ref.size() ? pair_read.size() : pair_ref.size());
            lv.load_reads((char*)pair_read.c_str(), (char*)pair_ref.c_str(), length);
            lv.reset();
            lv.run();
            printf("leap pass %d ED %d\n", lv.check_pass() ? 1 : 0, lv.get_ED());
            SIMD_ED sed;
            sed.init_levenshtein(k, ED_GLOBAL, true);
            sed.load_reads((char*)pair_read.c_str(), (char*)pair_ref.c_str(), (int)pair_read.size());
            sed.calculate_masks();
            sed.reset();
            sed.run();
            printf("simd_ed pass %d ED %d\n", sed.check_pass() ? 1 : 0, sed.get_ED());
            SIMD_ED saf;  // the affine mode of the same class (SIMD_ED.h:50), every pair from clean tables
            saf.init_affine(k, 200, ED_GLOBAL, x, o, e);
            saf.load_reads((char*)pair_read.c_str(), (char*)pair_ref.c_str(), (int)pair_read.size());
            saf.calculate_masks();
            saf.reset();
            saf.run();
            printf("simd_ed affine pass %d ED %d\n", saf.check_pass() ? 1 : 0, saf.get_ED());
            return 0;
        }
        if (file.empty()) {
            Dataset dataset(n, len, err, 0.96f, true, seed);  // benchmark.cpp:19
            file = dataset.output();
        }
        benchmark bench(x, o, e, k, n, true, mode == "clean" ? ASM_GREEDY_CLEAN : ASM_GREEDY_SEQUENTIAL);  // benchmark.cpp:22
        if (stream) {
            if (!answers.empty()) bench.read_answer_file(answers.c_str());
            bench.run_streamed(file.c_str(), (int64_t)chunk_mb << 20);
        } else {
            bench.read_string_file(file.c_str());
            if (!answers.empty()) bench.read_answer_file(answers.c_str());
            bench.run();
        }
        bench.print();
    } catch (const std::exception& ex) {
        fprintf(stderr, "%s\n", ex.what());
        return 1;
    }
    return 0;
}
