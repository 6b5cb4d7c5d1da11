// The part of parasail's C API (github.com/jeffdaily/parasail, parasail.h; no version is pinned by the reference, its
// submodule directory is empty) that GASMA/benchmark/benchmark_utils.h uses — :17,53,109-123,135-149,288 — over the MI355X
// library, so that the reference's own harness class compiles and runs on it unmodified:
//
//   parasail_matrix_create("ACGT", match, mismatch)                          :288
//   parasail_nw_trace / parasail_nw_trace_striped_sse41_128_16(s1, n1, s2, n2, open, extend, matrix)   :113,139
//   result->score                                                             :116,142   (= -penalty)
//   parasail_result_get_cigar(result, s1, n1, s2, n2, matrix)                 :114,140
//   parasail_cigar_decode(cigar)  -> malloc'ed "12=1X3I..." string             :115,141
//   parasail_result_free / parasail_cigar_free / parasail_matrix_free         :122-123,146-147
//
// Semantics: global alignment, end gaps penalised, a gap of length L costs open + (L-1) * extend, substitution
// score `match` (must be 0 here) or `mismatch` (<= 0): the score is minus the penalty asm_align_batch(ASM_NW) returns.  The
// traceback is this library's (asm_coverage's NW CIGAR rows: '=', 'X', 'I' = a read character without partner, 'D' = a reference
// character without partner); parasail's own tie-breaking between equally good alignments is not pinned by anything in the
// reference tree, so where several optimal alignments exist the string may differ from parasail's while the score cannot.
// One pair per call, one-pair device batches: an interface shim, slow by construction — the batch path is asm_run_benchmark_async.
#pragma once
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#define ASM_COMPAT_PARASAIL_SHIM 1
#include "../../asm_compat.hpp"

typedef struct parasail_matrix {
    int match, mismatch;
} parasail_matrix_t;

typedef struct parasail_result {
    int score;
    int end_query, end_ref;
    char* cigar_text_; /* the decoded CIGAR of this pair (owned by the result) */
} parasail_result_t;

typedef struct parasail_cigar {
    char* text_;
    int len;
    int beg_query, beg_ref;
} parasail_cigar_t;

inline parasail_matrix_t* parasail_matrix_create(const char* alphabet, const int match, const int mismatch) {
    if (!alphabet || std::strcmp(alphabet, "ACGT") != 0)
        throw std::runtime_error("parasail shim: the alphabet must be \"ACGT\" (benchmark_utils.h:288)");
    if (match != 0 || mismatch > 0)
        throw std::runtime_error("parasail shim: scores must be match 0, mismatch <= 0 (penalty form, benchmark_utils.h:288)");
    parasail_matrix_t* m = (parasail_matrix_t*)std::malloc(sizeof *m);
    m->match = match, m->mismatch = mismatch;
    return m;
}
inline void parasail_matrix_free(parasail_matrix_t* m) { std::free(m); }

inline parasail_result_t* parasail_nw_trace(const char* s1, const int s1Len, const char* s2, const int s2Len, const int open,
                                            const int extend, const parasail_matrix_t* matrix) {
    using namespace asm_amd;
    asm_handle* h = shared_handle();
    asm_params p;
    asm_default_params(&p);
    p.x = -matrix->mismatch, p.o = open, p.e = extend;
    uint32_t ro[2] = {0u, (uint32_t)s1Len}, fo[2] = {0u, (uint32_t)s2Len};
    const int gcap = 192, ncap = 1024;
    asm_batch* b = nullptr;
    void *d_pen = nullptr, *d_gops = nullptr, *d_gn = nullptr, *d_cov = nullptr, *d_nops = nullptr, *d_nn = nullptr, *d_cc = nullptr;
    check(h, asm_batch_upload(h, 1, s1, ro, s2, fo, ASM_GREEDY_CLEAN, &b));
    int rc = asm_device_malloc(h, 2 * sizeof(int32_t), &d_pen);
    if (!rc) rc = asm_device_malloc(h, sizeof(uint16_t) * gcap, &d_gops);
    if (!rc) rc = asm_device_malloc(h, 4, &d_gn);
    if (!rc) rc = asm_device_malloc(h, 4, &d_cov);
    if (!rc) rc = asm_device_malloc(h, sizeof(uint16_t) * ncap, &d_nops);
    if (!rc) rc = asm_device_malloc(h, 4, &d_nn);
    if (!rc) rc = asm_device_malloc(h, 16, &d_cc);
    if (!rc) rc = asm_memset_async(h, d_cc, 0, 16);
    int32_t pen = 0;
    uint8_t nn = 0;
    static thread_local uint16_t ops[1024];
    if (!rc) rc = asm_align_batch_async(h, b, ASM_NW, &p, (int32_t*)d_pen);
    /* the traceback rides on asm_coverage, which wants the Greedy CIGAR of the pair as its other input */
    if (!rc) rc = asm_greedy_cigar_batch_async(h, b, &p, (int32_t*)d_pen + 1, (uint16_t*)d_gops, gcap, (uint8_t*)d_gn);
    if (!rc) rc = asm_coverage(h, b, &p, (uint16_t*)d_gops, gcap, (uint8_t*)d_gn, 64, (uint8_t*)d_cov, (uint16_t*)d_nops, ncap, (uint8_t*)d_nn,
                               (unsigned long long*)d_cc);
    if (!rc) rc = asm_memcpy_d2h(h, &pen, d_pen, sizeof pen);
    if (!rc) rc = asm_memcpy_d2h(h, &nn, d_nn, 1);
    if (!rc) rc = asm_memcpy_d2h(h, ops, d_nops, sizeof(uint16_t) * ncap);
    asm_device_free(h, d_pen), asm_device_free(h, d_gops), asm_device_free(h, d_gn), asm_device_free(h, d_cov);
    asm_device_free(h, d_nops), asm_device_free(h, d_nn), asm_device_free(h, d_cc);
    asm_batch_free(h, b);
    check(h, rc);
    if (nn == 255) throw std::runtime_error("parasail shim: CIGAR with more than 254 runs");
    for (int a = 0, z = (int)nn - 1; a < z; a++, z--) { /* the device emits the rows in traceback order: last operation first */
        const uint16_t tmp = ops[a];
        ops[a] = ops[z], ops[z] = tmp;
    }
    char text[8192];
    check(h, asm_cigar_format(ops, nn, ncap, text, sizeof text));
    parasail_result_t* r = (parasail_result_t*)std::malloc(sizeof *r);
    r->score = -pen, r->end_query = s1Len - 1, r->end_ref = s2Len - 1;
    r->cigar_text_ = strdup(text);
    return r;
}
/* the vectorised entry point the harness uses by default (:139): same contract, the device does the work either way */
inline parasail_result_t* parasail_nw_trace_striped_sse41_128_16(const char* s1, const int s1Len, const char* s2, const int s2Len,
                                                                 const int open, const int extend, const parasail_matrix_t* matrix) {
    return parasail_nw_trace(s1, s1Len, s2, s2Len, open, extend, matrix);
}
inline void parasail_result_free(parasail_result_t* r) {
    if (r) std::free(r->cigar_text_);
    std::free(r);
}

inline parasail_cigar_t* parasail_result_get_cigar(parasail_result_t* result, const char* /*s1*/, int /*s1Len*/, const char* /*s2*/,
                                                   int /*s2Len*/, const parasail_matrix_t* /*matrix*/) {
    parasail_cigar_t* c = (parasail_cigar_t*)std::malloc(sizeof *c);
    c->text_ = strdup(result->cigar_text_ ? result->cigar_text_ : "");
    c->len = (int)std::strlen(c->text_), c->beg_query = 0, c->beg_ref = 0;
    return c;
}
/* parasail hands out a malloc'ed string the caller owns (the harness copies it into a std::string and leaks it, as it does with
 * the real library) */
inline char* parasail_cigar_decode(parasail_cigar_t* cigar) { return strdup(cigar->text_); }
inline void parasail_cigar_free(parasail_cigar_t* c) {
    if (c) std::free(c->text_);
    std::free(c);
}
