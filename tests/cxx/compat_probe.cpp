// Test program (built by tests/test_gpu_parity.py on the GPU box): the per-pair objects of host/asm_compat.hpp and the
// parasail-API shim, driven the way the reference's harness drives them (benchmark_utils.h:130-201), one line per pair:
//   <nw penalty> <nw cigar> <leap ED> <greedy cost> <greedy cigar>
// Input: a `>read\n<ref\n` file.  One hurdle_matrix object for the whole file (the reference's stale-buffer chain).
#include <fstream>
#include <iostream>
#include <string>

#include "parasail/parasail.h"
#include "asm_compat.hpp"

int main(int argc, char** argv) {
    using namespace asm_amd;
    if (argc < 3) return 2;
    const int k = atoi(argv[2]);
    std::ifstream in(argv[1]);
    std::string a, b;
    parasail_matrix_t* mat = parasail_matrix_create("ACGT", 0, -1);
    hurdle_matrix<int_128bit> hm(GLOBAL, 1, 1, 1);
    LV lv;
    lv.init(k, 200, ED_GLOBAL, 1, 1, 1);
    while (std::getline(in, a) && std::getline(in, b)) {
        a.erase(0, 1), b.erase(0, 1);
        parasail_result_t* r = parasail_nw_trace_striped_sse41_128_16(a.c_str(), (int)a.size(), b.c_str(), (int)b.size(), 1, 1, mat);
        parasail_cigar_t* c = parasail_result_get_cigar(r, a.c_str(), (int)a.size(), b.c_str(), (int)b.size(), mat);
        char* text = parasail_cigar_decode(c);
        lv.load_reads((char*)a.c_str(), (char*)b.c_str(), (int)std::max(a.size(), b.size()));
        lv.reset();
        lv.run();
        hm.reset(a.c_str(), (int)a.size(), b.c_str(), (int)b.size(), k);
        hm.run();
        std::cout << -r->score << " " << (text[0] ? text : "-") << " " << lv.get_ED() << " " << hm.get_cost() << " "
                  << (hm.get_CIGAR().empty() ? "-" : hm.get_CIGAR()) << "\n";
        free(text);
        parasail_cigar_free(c);
        parasail_result_free(r);
    }
    parasail_matrix_free(mat);
    return 0;
}
