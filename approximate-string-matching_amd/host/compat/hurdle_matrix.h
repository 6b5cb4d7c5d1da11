// Forwarding header for the reference's GASMA/hurdle_matrix.h (hurdle_matrix<T>, :20-685; int_128bit and alignment_type_t of
// GASMA/utils.h come with it): see benchmark_utils.h.  GASMA/main.cpp compiles against it unchanged.
#pragma once
#if __has_include("../asm_compat.hpp")
#include "../asm_compat.hpp"
#else /* reached through a compiler VFS overlay under the reference's file name: found by -I <package>/host */
#include "asm_compat.hpp"
#endif
using namespace asm_amd;
