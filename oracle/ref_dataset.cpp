// TEST INFRASTRUCTURE ONLY.  Driver around the reference's own input generator, `Dataset`
// (GASMA/benchmark/benchmark_dataset.h, #included in place — nothing copied), built into oracle/_ref/ref_dataset where
// /root/reference exists.  usage: ref_dataset <seed> <num_reads> <length> <error_rate> <out.seq> [exact=1]
// Dataset seeds libc rand() from time() (benchmark_dataset.h:190,223); this program supplies its own time() so that a run is
// reproducible and can be compared byte for byte with oracle/asm_oracle_dataset.c (tests/test_oracle_vs_reference.py).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>

static time_t g_fake_time = 1;
extern "C" time_t time(time_t* out) {
    if (out) *out = g_fake_time;
    return g_fake_time;
}

#include "benchmark_dataset.h"

int main(int argc, char** argv) {
    if (argc != 6 && argc != 7) {
        fprintf(stderr, "usage: %s <seed> <num_reads> <length> <error_rate> <out.seq> [exact=1]\n", argv[0]);
        return 2;
    }
    g_fake_time = (time_t)atoll(argv[1]);
    const bool exact = argc == 7 ? atoi(argv[6]) != 0 : true; /* 0: the "lt_eq" files (benchmark_dataset.h:153-156,246-250) */
    Dataset d(atoi(argv[2]), atoi(argv[3]), (float)atof(argv[4]), 0.96, exact, true); /* benchmark.cpp:19 */
    d.output(argv[5]);
    return 0;
}
