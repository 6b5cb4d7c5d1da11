// Forwarding header for the reference's GASMA/benchmark/benchmark_dataset.h (class Dataset, :61-253): see benchmark_utils.h.
#pragma once
#include "../asm_compat.hpp"
using namespace asm_amd;
