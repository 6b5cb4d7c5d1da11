// Forwarding header for the reference's GASMA/benchmark/LEAP_SIMD/LV_BAG.h (class LV, :40-54; enum ED_modes, :38): code that
// includes it gets the device-backed `LV` of host/asm_compat.hpp under the same un-namespaced name.  Like the original (:21) it
// puts `using namespace std` into the including file — the reference's benchmark_utils.h relies on that for its bare `string`.
#pragma once
#include <iostream>
#include <string>

/* beside this file's real location; when the header is reached through a compiler VFS overlay under the reference's file name
 * (oracle/Makefile, ref_harness_on_shim) relative paths resolve in the reference tree, and -I <package>/host finds it instead */
#if __has_include("../../asm_compat.hpp")
#include "../../asm_compat.hpp"
#else
#include "asm_compat.hpp"
#endif
using namespace std;
using namespace asm_amd;
