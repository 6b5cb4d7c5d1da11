// Seeded pair generator shared by the host generator (asm_generate_pairs) and the device generator kernel
// (asm_batch_generate): the same inline code runs on both sides, so the two are bit-identical by construction
// (tests/test_generator.py checks it on the GPU).
//
// Distribution restated from the reference's `Dataset` (GASMA/benchmark/benchmark_dataset.h:85-187;
// SURVEY.md App. D): pattern = iid uniform ACGT; exactly ceil(L*err) edit operations, each a substitution
// with probability mismatch_rate (uniform position, uniform base — may equal the old one), else a deletion
// or an insertion (50/50) at a uniform position of the current text.  The reference seeds libc rand() from
// time(); here every pair owns a counter-based splitmix64 state derived from (seed, pair index), so any
// slice of the stream can be produced independently (shards, host vs device).
#pragma once
#include <stdint.h>

#include "../../include/asm_mi355x.h"

#if defined(__HIPCC__)
#define ASM_HD __host__ __device__ __forceinline__
#else
#define ASM_HD inline
#endif

#define ASM_GEN_MAX_TEXT 1024 /* scratch size for one mutated text */

struct asm_rng {
    uint64_t s;
    ASM_HD uint64_t next() {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    ASM_HD uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
    ASM_HD float unit() { return (float)(next() >> 40) * (1.0f / 16777216.0f); } /* [0,1) */
};

// Two independent streams per pair: `types` decides lengths and edit kinds (so the sizing pass needs only
// this one), `values` decides bases and positions.
ASM_HD void asm_rng_for_pair(uint64_t seed, uint64_t pair, asm_rng* types, asm_rng* values) {
    asm_rng r;
    r.s = seed * 0xD1342543DE82EF95ull + pair * 0x2545F4914F6CDD1Dull + 0x1234567ull;
    types->s = r.next();
    values->s = r.next();
}

ASM_HD int asm_gen_num_errors(int L, float err) {
    // benchmark_dataset.h:154: ceil(uint64 * float) — a float product, widened for ceil
    float prod = (float)L * err;
    int c = (int)prod;
    return (float)c < prod ? c + 1 : c;
}

// Number of edit operations of a pair: ceil(L*err), or (ASM_GEN_UP_TO_ERRORS: rand_iid(0, ceil(L*err)), benchmark_dataset.h:156)
// uniform below it.  Drawn from the `types` stream right after the length, by both passes.
ASM_HD int asm_gen_pair_errors(const asm_gen_config* cfg, int L, asm_rng* t) {
    const int ne = asm_gen_num_errors(L, cfg->err);
    if (cfg->kind != ASM_GEN_UP_TO_ERRORS) return ne;
    return ne > 0 ? (int)t->below((uint32_t)ne) : 0;
}

// Lengths only: *m = read length, *n = ref length.
ASM_HD void asm_gen_lengths(const asm_gen_config* cfg, uint64_t pair, int* m, int* n) {
    asm_rng t, v;
    asm_rng_for_pair(cfg->seed, pair, &t, &v);
    int L = cfg->len_lo + (int)t.below((uint32_t)(cfg->len_hi - cfg->len_lo + 1));
    int len = L;
    if (cfg->kind == ASM_GEN_EXACT_ERRORS || cfg->kind == ASM_GEN_UP_TO_ERRORS) {
        int ne = asm_gen_pair_errors(cfg, L, &t);
        for (int i = 0; i < ne; i++) {
            float u = t.unit();
            if (u <= cfg->mismatch_rate) continue;
            uint32_t kind = t.below(2);
            if (kind == 0) {
                if (len > 0) len--;
            } else {
                len++;
            }
        }
    } else {
        len = 0;
        for (int i = 0; i < L; i++) {
            float u = t.unit();
            if (!(u < cfg->p_del)) len++;
            float w = t.unit();
            if (w < cfg->p_ins) len++;
        }
    }
    *m = L;
    *n = len;
}

// Full generation into caller buffers (read: >= len_hi bytes, text: ASM_GEN_MAX_TEXT bytes).
ASM_HD void asm_gen_pair(const asm_gen_config* cfg, uint64_t pair, char* read, char* text, int* m, int* n) {
    const char alphabet[4] = {'A', 'C', 'G', 'T'};
    asm_rng t, v;
    asm_rng_for_pair(cfg->seed, pair, &t, &v);
    int L = cfg->len_lo + (int)t.below((uint32_t)(cfg->len_hi - cfg->len_lo + 1));
    for (int i = 0; i < L; i++) read[i] = alphabet[v.below(4)];
    int len = 0;
    if (cfg->kind == ASM_GEN_EXACT_ERRORS || cfg->kind == ASM_GEN_UP_TO_ERRORS) {
        for (int i = 0; i < L; i++) text[i] = read[i];
        len = L;
        int ne = asm_gen_pair_errors(cfg, L, &t);
        for (int i = 0; i < ne; i++) {
            float u = t.unit();
            if (u <= cfg->mismatch_rate) { /* benchmark_dataset.h:113-120 */
                uint32_t pos = v.below((uint32_t)(len > 0 ? len : 1));
                char c = alphabet[v.below(4)];
                if (len > 0) text[pos] = c;
                continue;
            }
            uint32_t kind = t.below(2);
            if (kind == 0) { /* deletion, benchmark_dataset.h:121-132 */
                uint32_t pos = v.below((uint32_t)(len > 0 ? len : 1));
                if (len > 0) {
                    for (int j = (int)pos; j < len - 1; j++) text[j] = text[j + 1];
                    len--;
                }
            } else { /* insertion, benchmark_dataset.h:133-146 */
                uint32_t pos = v.below((uint32_t)(len > 0 ? len : 1));
                char c = alphabet[v.below(4)];
                if (len + 1 < ASM_GEN_MAX_TEXT) {
                    for (int j = len; j > (int)pos; j--) text[j] = text[j - 1];
                    text[pos] = c;
                    len++;
                }
            }
        }
    } else {
        for (int i = 0; i < L; i++) {
            float u = t.unit();
            char c = read[i];
            if (u < cfg->p_del) {
                /* base dropped */
            } else if (u < cfg->p_del + cfg->p_sub) {
                int code = (c == 'C') ? 1 : (c == 'G') ? 2 : (c == 'T') ? 3 : 0;
                text[len++] = alphabet[(code + 1 + (int)v.below(3)) & 3];
            } else {
                text[len++] = c;
            }
            float w = t.unit();
            if (w < cfg->p_ins && len + 1 < ASM_GEN_MAX_TEXT) text[len++] = alphabet[v.below(4)];
        }
    }
    *m = L;
    *n = len;
}
