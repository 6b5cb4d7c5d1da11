// asm-bench — the counterpart of the reference's `hurdle-matrix-benchmark` (GASMA/benchmark/benchmark.cpp:12-32):
// generate (or read) a ">read\n<ref\n" file, run NW + LEAP + Greedy over it, print the results block.
//   asm-bench [--file path | --n N --len L --err E --seed S] [--k K --x X --o O --e E] [--mode sequential|clean]
//             [--answers path]
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
    long batch_run = 1000000;
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
        else if (arg("--leap-simd")) leap_simd = atoi(argv[++i]);
        else if (arg("--shd")) shd = atoi(argv[++i]);
        else if (arg("--batch-run")) batch_run = atol(argv[++i]);
        else {
            fprintf(stderr, "unknown argument %s\n", argv[i]);
            return 2;
        }
    }
    try {
        if (leap_simd >= 0) return leap_simd_filter(stdin, leap_simd, shd != 0, batch_run);
        if (file.empty()) {
            Dataset dataset(n, len, err, 0.96f, true, seed);  // benchmark.cpp:19
            file = dataset.output();
        }
        benchmark bench(x, o, e, k, n, true, mode == "clean" ? ASM_GREEDY_CLEAN : ASM_GREEDY_SEQUENTIAL);  // benchmark.cpp:22
        bench.read_string_file(file.c_str());
        if (!answers.empty()) bench.read_answer_file(answers.c_str());
        bench.run();
        bench.print();
    } catch (const std::exception& ex) {
        fprintf(stderr, "%s\n", ex.what());
        return 1;
    }
    return 0;
}
