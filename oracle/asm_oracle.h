/* TEST INFRASTRUCTURE ONLY — the oracle.
 *
 * A plain-C CPU restatement of the three aligners on the reference's benchmark hot path, written from the
 * behavioural specification and the reference sources read as text.  It exists so that the
 * HIP kernels can be checked on machines where /root/reference does not exist (the GPU box).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product library never does.
 *
 * Pinning: tests/test_oracle_vs_reference.py diffs every function here against oracle/_ref/libasm_ref.so
 * (the real reference compiled in place, this container only) and tests/golden/ holds vectors generated
 * from that library by tests/golden/make_golden.py.  NW: PARITY UNPINNED by reference vectors (parasail is absent
 * from the reference tree): its parity is by definition (Gotoh global affine distance),
 * cross-checked against plain Levenshtein DP for unit costs, and statistically against every accuracy and coverage line of
 * the reference's README on pairs drawn the reference's way (asm_oracle_dataset.c; tests/test_oracle_golden.py).
 *
 * Batch layout used by every entry point: `reads`/`refs` are concatenated ASCII strings (no terminators),
 * `*_off` are n+1 prefix offsets (uint32), pair i is reads[read_off[i] .. read_off[i+1]).
 */
#ifndef ASM_ORACLE_H
#define ASM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_GREEDY_SEQUENTIAL 0 /* reference as run: stale buffer tails flow from pair to pair (F4)     */
#define ORC_GREEDY_CLEAN 1      /* buffers zeroed before every pair: order independent                   */

/* Greedy hurdle-matrix aligner, GLOBAL mode.
 * Follows GASMA/hurdle_matrix.h:285-455,568-597,625-665 and GASMA/utils.h:131-153,168-216,263-270,576-593.
 * probs = {p_match, p_mismatch, p_indel} (hurdle_matrix.h:552-559 defaults 0.80, 0.20/3, 0.40/3).
 * cigars may be NULL; otherwise n*cigar_stride bytes, NUL terminated.
 * steps (optional): number of _step() commits per pair (work statistic).
 * Pairs whose destination lane n-m lies outside [-k,k] are undefined in the reference; here the
 * destination lane is built like an in-band lane (documented own behaviour, excluded from parity counts). */
int orc_greedy_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs,
                     const uint32_t* ref_off, int k, int x, int o, int e, const double* probs, int mode,
                     int32_t* costs, char* cigars, int cigar_stride, int32_t* steps);

/* Same with the constructor's alignment type (hurdle_matrix.h:477,553): ORC_ALIGN_SEMI_GLOBAL zeroes the switch cost into
 * the first highway, into the destination lane and of the final hop (hurdle_matrix.h:313-316,335-338,577-580).  LOCAL is
 * declared but unsupported in the reference (:467). */
#define ORC_ALIGN_GLOBAL 0
#define ORC_ALIGN_SEMI_GLOBAL 1
int orc_greedy_batch_typed(int64_t n, const char* reads, const uint32_t* read_off, const char* refs,
                           const uint32_t* ref_off, int k, int x, int o, int e, const double* probs, int mode,
                           int alignment_type, int32_t* costs, char* cigars, int cigar_stride, int32_t* steps);

/* The 128-byte A and B buffers each pair's conversion sees (A at views[i*256], B at views[i*256+128]).
 * Model of the in-place permutation of GASMA/bit_convert.cpp:265-330 (SRC table). */
int orc_greedy_views(int64_t n, const char* reads, const uint32_t* read_off, const char* refs,
                     const uint32_t* ref_off, int mode, uint8_t* views);

/* Sequential mode in pieces: content of the A and B buffers before the first pair of the next orc_greedy_* call (NULL = zeros,
 * the default) and after the last pair of the previous one.  Not thread safe; test use only. */
void orc_greedy_set_initial_buffers(const uint8_t* ab /* [256] */);
void orc_greedy_get_final_buffers(uint8_t* ab /* [256] */);

/* LEAP (banded affine Landau-Vishkin "BAG") penalty = final_ED, as benchmarked.
 * Follows GASMA/benchmark/LEAP_SIMD/LV_BAG.cpp:9-23,65-245,356-358 with init(k,200,ED_GLOBAL,x,o,e).
 * eds[i] = -1 when no lane passes within af_threshold=200 (reference returns a stale value there). */
int orc_leap_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs,
                   const uint32_t* ref_off, int k, int x, int o, int e, int32_t* eds);
/* LV::init's ED_modes (LV_BAG.h:38; own numbering, GLOBAL = 0 = what the harness uses): which lanes start at generation 0 and
 * whether reaching the end is judged by converge_ED (LV_BAG.cpp:102-104,220-238) */
#define ORC_LEAP_GLOBAL 0
#define ORC_LEAP_LOCAL 1
#define ORC_LEAP_SEMI_FREE_BEGIN 2
#define ORC_LEAP_SEMI_FREE_END 3
int orc_leap_mode_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs,
                        const uint32_t* ref_off, int k, int x, int o, int e, int mode, int32_t* eds);

/* NW: global affine-gap distance, match 0, mismatch x, gap(L) = o + (L-1)*e.
 * Semantics of the call site GASMA/benchmark/benchmark_utils.h:139-142,288 (parasail, absent): penalty = -score. */
int orc_nw_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs,
                 const uint32_t* ref_off, int x, int o, int e, int32_t* penalties);

/* Plain Levenshtein distance (independent cross-check for x=o=e=1). */
int orc_levenshtein_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs,
                          const uint32_t* ref_off, int32_t* dist);

/* NW with traceback: CIGAR with '=', 'X', 'I', 'D' (own documented tie-break: diagonal, then I, then D
 * when walking back from (m,n)); parasail's preference is internal and unpinned. */
int orc_nw_cigar_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs,
                       const uint32_t* ref_off, int x, int o, int e, int32_t* penalties, char* cigars,
                       int cigar_stride);

/* Coverage metric: long_consecutive_matching_substring + covers
 * (GASMA/benchmark/benchmark_coverage.h:26-67,73-91; call site benchmark_utils.h:214-225,256).
 * out[i] = 1 when LCM(read,ref,cigar1,thr1) covers LCM(read,ref,cigar2,thr2). */
int orc_coverage_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs,
                       const uint32_t* ref_off, const char* cigars1, int stride1, int thr1,
                       const char* cigars2, int stride2, int thr2, uint8_t* out);

/* ---- bit-parallel LEAP (SIMD_ED) and SHD pre-filter: asm_oracle_filter.c ------------------------------------------ */
#define ORC_FILTER_SEQUENTIAL 0 /* reference as run: the verdict state flows from pair to pair (S2)      */
#define ORC_FILTER_CLEAN 1      /* every pair judged alone                                               */

/* SIMD_ED::init_levenshtein(ed_t, ED_GLOBAL, shd_enable) then, per pair, load_reads(read, ref, min(m,256)) /
 * calculate_masks / reset / run (GASMA/benchmark/LEAP_SIMD/SIMD_ED.cpp:10-61,140-212,214-268,269-352; driver
 * LEAP_SIMD/main.cpp:95-101,186-195).  pass[i] = check_pass(); ed[i] = get_ED() (= converge_ED, :748-753) when it
 * passes, else -1; ed_raw (optional) = get_ED() whatever the verdict.  state = {final_ED, lane distance, converge_ED}
 * before the first pair (the reference leaves them uninitialised).  shd_enable needs ed_t <= 16 (MAX_ERROR_AVX). */
int orc_simd_ed_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs, const uint32_t* ref_off,
                      int ed_t, int shd_enable, int mode, const int32_t* state, int32_t* ed, int32_t* ed_raw,
                      uint8_t* pass);
/* ... with init_levenshtein's ED_modes (numbering of ORC_LEAP_*, declared below) */
int orc_simd_ed_edmode_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs, const uint32_t* ref_off,
                             int ed_t, int shd_enable, int mode, int ed_mode, const int32_t* state, int32_t* ed, int32_t* ed_raw,
                             uint8_t* pass);

/* SHD on the pair's 2-bit planes: bit_vec_filter_avx(read0, read1, ref0, ref1, min(m,256), max_error)
 * (GASMA/benchmark/LEAP_SIMD/SHD.cpp:95-143,241-322; popcount.cpp:44-76,78-110).  pass[i] in {0,1}. */
/* SIMD_ED affine mode, clean (every pair from init_affine's tables): ed[i] = get_ED() if the pair passes, else -1 */
int orc_simd_ed_affine_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs, const uint32_t* ref_off,
                             int gap_t, int af_t, int x, int o, int ext, int32_t* ed, uint8_t* pass);
/* ... with init_affine's SHD_enable / SHD_threshold (shd_t <= gap_t, <= 16): the mask-array SHD in front of run_affine */
int orc_simd_ed_affine_shd_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs, const uint32_t* ref_off,
                                 int gap_t, int af_t, int x, int o, int ext, int shd_enable, int shd_t, int32_t* ed, uint8_t* pass);
/* ... and with init_affine's ED_modes (numbering of ORC_LEAP_*) */
int orc_simd_ed_affine_mode_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs, const uint32_t* ref_off,
                                  int gap_t, int af_t, int x, int o, int ext, int shd_enable, int shd_t, int mode, int32_t* ed,
                                  uint8_t* pass);

int orc_shd_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs, const uint32_t* ref_off,
                  int max_error, int32_t* pass);

/* ---- the reference's input distribution: asm_oracle_dataset.c ----------------------------------------------------------
 * Dataset(n, length, error_rate, mismatch_rate, exact = true).output() (GASMA/benchmark/benchmark_dataset.h:85-187,212-240) over
 * an emulation of glibc's rand() seeded with srand(seed) — the stream the README's accuracy lines are statistics of.
 * reads: n*length bytes; refs: n*(length + ceil(length*error_rate) + 1) bytes; offsets n+1 each. */
int orc_reference_dataset(int64_t n, int length, float error_rate, float mismatch_rate, unsigned int seed, char* reads,
                          uint32_t* read_off, char* refs, uint32_t* ref_off);
/* the same with Dataset's exact_error_rate flag: 0 = the "lt_eq" files (number of edits uniform in 0 .. ceil(L*err) - 1,
 * benchmark_dataset.h:153-156) that the result blocks of GASMA/benchmark/README.md were measured on */
int orc_reference_dataset_ex(int64_t n, int length, float error_rate, float mismatch_rate, int exact, unsigned int seed,
                             char* reads, uint32_t* read_off, char* refs, uint32_t* ref_off);
void orc_glibc_rand_stream(unsigned int seed, int count, int32_t* out);

/* number of OpenMP threads the batch entry points will use (LEAP/NW/Greedy-clean are parallel over pairs;
 * Greedy-sequential resolves views serially first, then runs pairs in parallel). */
int orc_set_threads(int nthreads);

#ifdef __cplusplus
}
#endif
#endif
