// Forwarding header: code written against the reference's GASMA/benchmark/benchmark_utils.h (class benchmark, :47-414, and
// through it hurdle_matrix / LV) compiles against the MI355X library unchanged when this directory comes first on the quote
// include path (`-iquote host/compat -I include`): same un-namespaced names, same constructor and member signatures.
// INTEGRATION.md shows the build line; tests/test_cabi_and_host.py compiles the reference's own benchmark.cpp this way.
#pragma once
#include "../asm_compat.hpp"
using namespace asm_amd;
