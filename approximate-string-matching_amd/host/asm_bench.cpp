// asm-bench — the counterpart of the reference's `hurdle-matrix-benchmark` (GASMA/benchmark/benchmark.cpp:12-32):
// generate (or read) a ">read\n<ref\n" file, run NW + LEAP + Greedy over it, print the results block.
//   asm-bench [--file path | --n N --len L --err E --seed S] [--k K --x X --o O --e E] [--mode sequential|clean]
//             [--answers path] [--stream [--chunk-mb M]]      --stream: the file goes through asm_stream_seq_file
//   asm-bench --pair READ REF [--k K]                                     the per-pair classes of the reference on one pair
//                                                                        (hurdle_matrix reset/run/get_cost/get_CIGAR, LV, SIMD_ED)
//   asm-bench --leap-simd ERROR [--shd 0|1] [--batch-run N] < pairs      the LEAP_SIMD stdin filter driver
//                                                                        (GASMA/benchmark/LEAP_SIMD/main.cpp:52-101)
#include <cstdlib>
#include <cstring>
#include <string>

#include "asm_compat.hpp"

int main(int argc, char** argv) {
    using namespace asm_amd;
    std::string file, answers, mode = "sequential";
    int leap_simd = -1, shd = 1;
    std::string pair_read, pair_ref;
    bool pair_mode = false;
    long batch_run = 1000000;
    bool stream = false;
    long chunk_mb = 0;
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
        else {
            fprintf(stderr, "unknown argument %s\n", argv[i]);
            return 2;
        }
    }
    try {
        if (leap_simd >= 0) return leap_simd_filter(stdin, leap_simd, shd != 0, batch_run);
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
