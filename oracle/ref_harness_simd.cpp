// TEST INFRASTRUCTURE ONLY — never linked into, imported by or shipped with the product.
//
// Driver around the *real* bit-parallel LEAP and SHD sources of the reference (compiled in place from
// /root/reference/GASMA/benchmark/LEAP_SIMD by oracle/Makefile into oracle/_ref/libasm_ref_simd.so; nothing is copied).
// A second library, because LEAP_SIMD/bit_convert.cpp and GASMA/bit_convert.cpp define the same symbols.
//
// Call sequences (LEAP_SIMD/main.cpp:95-101,186-195 — the stdin filter driver):
//   SIMD_ED::init_levenshtein(error, ED_GLOBAL, SHD on/off) once, then per pair
//   load_reads / calculate_masks / reset / run / check_pass (+ get_ED)
//   bit_vec_filter_avx(read planes, ref planes, length, max_error)          (SHD.h:17-18, SHD.cpp:241-322)
#include <cstdint>
#include <cstring>
#include <string>

#include "SIMD_ED.h"  // /root/reference/GASMA/benchmark/LEAP_SIMD (via -I)
#include "SHD.h"

extern "C" {

// ed[i] = get_ED() as the reference returns it (also when check_pass() is false), pass[i] = check_pass().
// SIMD_ED keeps final_ED / final_lane_idx / converge_ED across pairs and never initialises them (SIMD_ED.cpp:63-79,
// 258-268), so the driver first runs one warm-up pair ("AAAA" vs "AAAT": reaches the end at e = 1 on the main lane),
// which pins that state before the batch starts: lane mid-1 is swept first and reaches the end at e = 1 through its
// diagonal neighbour, so final_ED = 1, lane distance 1, converge_ED = 2.
int ref_simd_ed_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs, const uint32_t* ref_off,
                      int ed_t, int shd_enable, int32_t* ed, uint8_t* pass) {
    SIMD_ED* obj = new SIMD_ED;
    obj->init_levenshtein(ed_t, ED_GLOBAL, shd_enable != 0);
    {
        char a[8] = "AAAA", b[8] = "AAAT";
        obj->load_reads(a, b, 4);
        obj->calculate_masks();
        obj->reset();
        obj->run();
    }
    std::string s1, s2;
    for (int64_t i = 0; i < n; i++) {
        int m = (int)(read_off[i + 1] - read_off[i]);
        int nn = (int)(ref_off[i + 1] - ref_off[i]);
        s1.assign(reads + read_off[i], m);
        s2.assign(refs + ref_off[i], nn);
        int length = m > 256 ? 256 : m; /* main.cpp:131-132 */
        obj->load_reads((char*)s1.c_str(), (char*)s2.c_str(), length);
        obj->calculate_masks();
        obj->reset();
        obj->run();
        pass[i] = obj->check_pass() ? 1 : 0;
        ed[i] = obj->get_ED();
    }
    delete obj;
    return 0;
}

// The same driver with init_levenshtein's ED_modes argument (oracle numbering: 0 GLOBAL, 1 LOCAL, 2 SEMI_FREE_BEGIN,
// 3 SEMI_FREE_END) and a caller-chosen warm-up pair (it must reach the end at a generation >= 1 in the mode at hand, so that
// final_ED, final_lane_idx and converge_ED are all written before the batch starts).
int ref_simd_ed_edmode_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs, const uint32_t* ref_off,
                             int ed_t, int shd_enable, int mode, const char* warm_read, const char* warm_ref, int32_t* ed,
                             uint8_t* pass) {
    static const ED_modes map[4] = {ED_GLOBAL, ED_LOCAL, ED_SEMI_FREE_BEGIN, ED_SEMI_FREE_END};
    if (mode < 0 || mode > 3) return -1;
    SIMD_ED* obj = new SIMD_ED;
    obj->init_levenshtein(ed_t, map[mode], shd_enable != 0);
    {
        std::string a(warm_read), b(warm_ref);
        obj->load_reads((char*)a.c_str(), (char*)b.c_str(), (int)a.size());
        obj->calculate_masks();
        obj->reset();
        obj->run();
    }
    std::string s1, s2;
    for (int64_t i = 0; i < n; i++) {
        int m = (int)(read_off[i + 1] - read_off[i]);
        int nn = (int)(ref_off[i + 1] - ref_off[i]);
        s1.assign(reads + read_off[i], m);
        s2.assign(refs + ref_off[i], nn);
        int length = m > 256 ? 256 : m;
        obj->load_reads((char*)s1.c_str(), (char*)s2.c_str(), length);
        obj->calculate_masks();
        obj->reset();
        obj->run();
        pass[i] = obj->check_pass() ? 1 : 0;
        ed[i] = obj->get_ED();
    }
    delete obj;
    return 0;
}

// Affine mode, CLEAN: init_affine before every pair (it destroys and rebuilds the tables: -2 everywhere, start[mid][0] = 0), so
// that no pair sees the tables of the one before; ed[i] = get_ED(), pass[i] = check_pass().
int ref_simd_ed_affine_shd_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs, const uint32_t* ref_off,
                                 int gap_t, int af_t, int x, int o, int e, int shd_enable, int shd_t, int32_t* ed, uint8_t* pass);
int ref_simd_ed_affine_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs, const uint32_t* ref_off,
                             int gap_t, int af_t, int x, int o, int e, int32_t* ed, uint8_t* pass) {
    return ref_simd_ed_affine_shd_batch(n, reads, read_off, refs, ref_off, gap_t, af_t, x, o, e, 0, 0, ed, pass);
}
int ref_simd_ed_affine_mode_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs, const uint32_t* ref_off,
                                  int gap_t, int af_t, int x, int o, int e, int shd_enable, int shd_t, int mode, int32_t* ed,
                                  uint8_t* pass);
// ... with init_affine's SHD_enable / SHD_threshold (SIMD_ED.h:50)
int ref_simd_ed_affine_shd_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs, const uint32_t* ref_off,
                                 int gap_t, int af_t, int x, int o, int e, int shd_enable, int shd_t, int32_t* ed, uint8_t* pass) {
    return ref_simd_ed_affine_mode_batch(n, reads, read_off, refs, ref_off, gap_t, af_t, x, o, e, shd_enable, shd_t, 0, ed, pass);
}
// ... and its ED_modes (mode in the oracle's numbering: 0 GLOBAL, 1 LOCAL, 2 SEMI_FREE_BEGIN, 3 SEMI_FREE_END)
int ref_simd_ed_affine_mode_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs, const uint32_t* ref_off,
                                  int gap_t, int af_t, int x, int o, int e, int shd_enable, int shd_t, int mode, int32_t* ed,
                                  uint8_t* pass) {
    static const ED_modes map[4] = {ED_GLOBAL, ED_LOCAL, ED_SEMI_FREE_BEGIN, ED_SEMI_FREE_END};
    if (mode < 0 || mode > 3) return -1;
    const ED_modes ref_mode = map[mode];
    SIMD_ED* obj = new SIMD_ED;
    std::string s1, s2;
    for (int64_t i = 0; i < n; i++) {
        int m = (int)(read_off[i + 1] - read_off[i]);
        int nn = (int)(ref_off[i + 1] - ref_off[i]);
        s1.assign(reads + read_off[i], m);
        s2.assign(refs + ref_off[i], nn);
        int length = m > 256 ? 256 : m;
        obj->init_affine(gap_t, af_t, ref_mode, x, o, e, shd_enable != 0, shd_t);
        obj->load_reads((char*)s1.c_str(), (char*)s2.c_str(), length);
        obj->calculate_masks();
        obj->reset(); /* -> reset_affine / run_affine: affine_mode is set by init_affine */
        obj->run();
        pass[i] = obj->check_pass() ? 1 : 0;
        ed[i] = obj->get_ED();
    }
    delete obj;
    return 0;
}

// SHD on the 2-bit planes of a pair: strings NUL-padded to 256 characters, converted with avx_convert2bit.
int ref_shd_batch(int64_t n, const char* reads, const uint32_t* read_off, const char* refs, const uint32_t* ref_off,
                  int max_error, int32_t* pass) {
    alignas(32) char a[256], b[256];
    alignas(32) uint8_t a0[32], a1[32], b0[32], b1[32];
    for (int64_t i = 0; i < n; i++) {
        int m = (int)(read_off[i + 1] - read_off[i]);
        int nn = (int)(ref_off[i + 1] - ref_off[i]);
        int length = m > 256 ? 256 : m;
        memset(a, 0, 256);
        memset(b, 0, 256);
        memcpy(a, reads + read_off[i], length);
        memcpy(b, refs + ref_off[i], nn > 256 ? 256 : nn);
        avx_convert2bit(a, a0, a1);
        avx_convert2bit(b, b0, b1);
        pass[i] = bit_vec_filter_avx(_mm256_load_si256((__m256i*)a0), _mm256_load_si256((__m256i*)a1),
                                     _mm256_load_si256((__m256i*)b0), _mm256_load_si256((__m256i*)b1), length, max_error);
    }
    return 0;
}

}  // extern "C"
