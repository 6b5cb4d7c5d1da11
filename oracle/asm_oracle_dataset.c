/* TEST INFRASTRUCTURE ONLY — the reference's input distribution, exactly.
 *
 * The reference's `Dataset` (GASMA/benchmark/benchmark_dataset.h:85-187,212-240) draws everything from libc rand(): every
 * pattern character, every edit kind, position and base.  glibc's rand() is a lagged additive generator, r[i] = r[i-3] +
 * r[i-31] (mod 2^32), output r[i] >> 1 — so the "random" pattern characters (the top two bits of consecutive outputs) obey
 * c[i] = c[i-3] + c[i-31] + carry (mod 4): they are NOT independent, and the accuracy percentages the reference's README
 * publishes (README.md:16-67) are statistics of THAT stream.  The product's generator (csrc/asm_gen.h) is a counter-based
 * iid generator (random access, shardable); on its pairs LEAP/Greedy agree with NW ~0.2 points less often at err >= 0.15
 * (4-6 sigma at 10^6 pairs) while every other statistic matches.  To compare with the README at full statistical power the
 * tests therefore feed the aligners pairs drawn the reference's way: this file restates Dataset's procedure over an
 * emulation of glibc's TYPE_3 rand() (the public algorithm of random_r.c; tests check it against the running libc).
 * Sequential by nature (the number of draws per pair depends on the draws), which is why the product does not use it. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "asm_oracle.h"

typedef struct {
    uint32_t r[34];
    int k; /* index of the next output modulo 34 */
} glibc_rand;

static void glibc_srand(glibc_rand* g, unsigned int seed) {
    int32_t st[34 + 310];
    if (seed == 0) seed = 1;
    st[0] = (int32_t)seed;
    for (int i = 1; i < 31; i++) { /* 16807 * st[i-1] mod 2^31-1 without overflow (random_r.c: __srandom_r) */
        const long hi = st[i - 1] / 127773, lo = st[i - 1] % 127773;
        long word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        st[i] = (int32_t)word;
    }
    /* the state is the 31 words; the generator then discards its first 310 outputs */
    uint32_t ring[31];
    for (int i = 0; i < 31; i++) ring[i] = (uint32_t)st[i];
    int f = 3, b = 0; /* front = rear + 3 */
    for (int i = 0; i < 310; i++) {
        ring[f] += ring[b];
        f = (f + 1) % 31, b = (b + 1) % 31;
    }
    memset(g, 0, sizeof *g);
    for (int i = 0; i < 31; i++) g->r[i] = ring[i];
    g->k = f;       /* reuse fields: r[0..30] ring, r[31] unused */
    g->r[32] = (uint32_t)b;
}

static int glibc_rand_next(glibc_rand* g) {
    int f = g->k, b = (int)g->r[32];
    g->r[f] += g->r[b];
    const uint32_t out = g->r[f] >> 1;
    f = (f + 1) % 31, b = (b + 1) % 31;
    g->k = f, g->r[32] = (uint32_t)b;
    return (int)out;
}

#define ORC_RAND_MAX 2147483647

/* benchmark_dataset.h:85-96 */
static uint64_t rand_iid(glibc_rand* g, uint64_t min, uint64_t max) {
    for (;;) {
        const int n_rand = glibc_rand_next(g);
        const uint64_t range = max - min;
        const uint64_t rem = ORC_RAND_MAX % range;
        const uint64_t sample = ORC_RAND_MAX / range;
        if ((uint64_t)n_rand < ORC_RAND_MAX - rem) return min + (uint64_t)n_rand / sample;
    }
}

/* Dataset(num_reads, length, error_rate, mismatch_rate, exact).output() with srand(seed) (the reference seeds from
 * time()): pairs in the batch layout.  reads must hold n*length bytes, refs n*(length + ceil(length*err) + 1). */
int orc_reference_dataset_ex(int64_t n, int length, float error_rate, float mismatch_rate, int exact, unsigned int seed,
                             char* reads, uint32_t* read_off, char* refs, uint32_t* ref_off) {
    static const char alphabet[4] = {'A', 'C', 'G', 'T'};
    if (n < 0 || length < 1 || length > 400) return -1;
    glibc_rand g;
    glibc_srand(&g, seed);
    const uint64_t max_errors = (uint64_t)ceil((float)length * error_rate); /* :154 — a float product */
    if (!exact && max_errors == 0) return -1; /* rand_iid(0, 0) divides by zero in the reference (:156) */
    char* text = (char*)malloc((size_t)length + max_errors + 2);
    if (!text) return -2;
    uint64_t ra = 0, rb = 0;
    for (int64_t p = 0; p < n; p++) {
        char* pattern = reads + ra;
        for (int i = 0; i < length; i++) pattern[i] = alphabet[rand_iid(&g, 0, 4)]; /* :100-109 */
        /* :153-156: exactly ceil(L * err) operations, or ("lt_eq" files, exact_error_rate = false) uniformly 0 .. ceil - 1 */
        const uint64_t num_errors = exact ? max_errors : rand_iid(&g, 0, max_errors);
        memcpy(text, pattern, (size_t)length);
        uint64_t len = (uint64_t)length;
        for (uint64_t q = 0; q < num_errors; q++) { /* :161-181 */
            const float random = ((float)glibc_rand_next(&g)) / (float)ORC_RAND_MAX;
            if (random <= mismatch_rate) { /* :113-120 */
                const int position = (int)rand_iid(&g, 0, len);
                text[position] = alphabet[rand_iid(&g, 0, 4)];
            } else if (rand_iid(&g, 1, 3) == 1) { /* deletion :121-132 */
                const int position = (int)rand_iid(&g, 0, len);
                memmove(text + position, text + position + 1, (size_t)(len - 1 - (uint64_t)position));
                len--;
            } else { /* insertion :133-146 */
                const int position = (int)rand_iid(&g, 0, len);
                memmove(text + position + 1, text + position, (size_t)(len - (uint64_t)position));
                len++;
                text[position] = alphabet[rand_iid(&g, 0, 4)];
            }
        }
        memcpy(refs + rb, text, (size_t)len);
        read_off[p] = (uint32_t)ra, ref_off[p] = (uint32_t)rb;
        ra += (uint64_t)length, rb += len;
    }
    read_off[n] = (uint32_t)ra, ref_off[n] = (uint32_t)rb;
    free(text);
    return 0;
}

int orc_reference_dataset(int64_t n, int length, float error_rate, float mismatch_rate, unsigned int seed, char* reads,
                          uint32_t* read_off, char* refs, uint32_t* ref_off) {
    return orc_reference_dataset_ex(n, length, error_rate, mismatch_rate, 1, seed, reads, read_off, refs, ref_off);
}

/* the first `count` outputs of the emulated rand() after srand(seed) — so that a test can compare with the running libc */
void orc_glibc_rand_stream(unsigned int seed, int count, int32_t* out) {
    glibc_rand g;
    glibc_srand(&g, seed);
    for (int i = 0; i < count; i++) out[i] = glibc_rand_next(&g);
}
