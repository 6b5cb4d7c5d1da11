// Micro-benchmark: issue cost of the integer/bit instructions the aligner kernels are made of (gfx950).
// Each kernel runs 8 independent chains of one operation so the result is throughput, not latency.
// build: hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
#define ITER 4096
#define CHAINS 8

#define KERNEL(name, T, INIT, OP)                                                        \
    __global__ __launch_bounds__(256) void name(T* out, int s, T seed) {                 \
        T v[CHAINS];                                                                     \
        for (int c = 0; c < CHAINS; c++) v[c] = INIT;                                    \
        for (int it = 0; it < ITER; it++) {                                              \
            _Pragma("unroll") for (int c = 0; c < CHAINS; c++) { OP; }                   \
        }                                                                                \
        T acc = 0;                                                                       \
        for (int c = 0; c < CHAINS; c++) acc += v[c];                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                \
    }

KERNEL(k_add32, uint32_t, seed + threadIdx.x + c, v[c] = v[c] + (uint32_t)s)
KERNEL(k_xor32, uint32_t, seed + threadIdx.x + c, v[c] = (v[c] ^ (uint32_t)s) + 1u)
KERNEL(k_shl32, uint32_t, seed + threadIdx.x + c, v[c] = (v[c] << (s & 31)) + 1u)
KERNEL(k_shl64, u64, seed + threadIdx.x + c, v[c] = (v[c] << (s & 63)) + 1ull)
KERNEL(k_shr64v, u64, seed + threadIdx.x + c, v[c] = (v[c] >> (v[c] & 7)) + 0x100000001ull)
KERNEL(k_add64, u64, seed + threadIdx.x + c, v[c] = v[c] + (u64)s * 0x100000001ull)
KERNEL(k_ctz32, uint32_t, seed + threadIdx.x + c, v[c] = (uint32_t)__builtin_ctz(v[c] | 0x80000000u) + v[c])
KERNEL(k_ctz64, u64, seed + threadIdx.x + c, v[c] = (u64)__builtin_ctzll(v[c] | (1ull << 63)) + v[c])
KERNEL(k_pop32, uint32_t, seed + threadIdx.x + c, v[c] = (uint32_t)__popc(v[c]) + v[c])
KERNEL(k_pop64, u64, seed + threadIdx.x + c, v[c] = (u64)__popcll(v[c]) + v[c])
KERNEL(k_mul32, uint32_t, seed + threadIdx.x + c, v[c] = v[c] * 0x00204081u + 1u)
KERNEL(k_alignbit, uint32_t, seed + threadIdx.x + c, v[c] = __builtin_amdgcn_alignbit(v[c], (uint32_t)seed, (uint32_t)s) + 1u)
KERNEL(k_cndmask, uint32_t, seed + threadIdx.x + c, v[c] = (v[c] > (uint32_t)s ? v[c] - 3u : v[c] + 7u))
KERNEL(k_min32, uint32_t, seed + threadIdx.x + c, v[c] = min(v[c] + 5u, (uint32_t)s * 977u + c))
KERNEL(k_fma64, double, (double)(seed + threadIdx.x + c), v[c] = __fma_rn(v[c], 1.0000001, (double)s))
KERNEL(k_cvt64, double, (double)(seed + threadIdx.x + c), v[c] = (double)((int)v[c] + s))
KERNEL(k_bfe, uint32_t, seed + threadIdx.x + c, v[c] = ((v[c] >> (s & 31)) & 1u) + v[c] + 1u)

template <typename T, typename K>
void run(const char* name, K kern, int ops_per_iter_per_chain) {
    T* d;
    const int blocks = 256 * 8, threads = 256;
    hipMalloc(&d, sizeof(T) * blocks * threads);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    kern<<<blocks, threads>>>(d, 3, (T)5);
    hipDeviceSynchronize();
    hipEventRecord(a);
    kern<<<blocks, threads>>>(d, 3, (T)5);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    // wave-instructions executed per SIMD: blocks*4 waves / 1024 SIMDs, each ITER*CHAINS "ops"
    double waves_per_simd = blocks * 4.0 / 1024.0;
    double ops = waves_per_simd * ITER * CHAINS;
    double cycles = ms * 1e-3 * 2.4e9;
    printf("%-10s %8.3f ms  %6.2f cycles per source-level op per SIMD (at 2.4 GHz, %d ISA ops expected)\n", name, ms,
           cycles / ops, ops_per_iter_per_chain);
    hipFree(d);
}

int main() {
    run<uint32_t>("add32", k_add32, 1);
    run<uint32_t>("xor+add32", k_xor32, 2);
    run<uint32_t>("shl+add32", k_shl32, 2);
    run<u64>("shl64+add", k_shl64, 3);
    run<u64>("shr64v+add", k_shr64v, 4);
    run<u64>("add64", k_add64, 2);
    run<uint32_t>("ctz32+or+add", k_ctz32, 3);
    run<u64>("ctz64+..", k_ctz64, 8);
    run<uint32_t>("pop32+add", k_pop32, 2);
    run<u64>("pop64+add", k_pop64, 4);
    run<uint32_t>("mul32+add", k_mul32, 2);
    run<uint32_t>("alignbit+add", k_alignbit, 2);
    run<uint32_t>("cmp+cnd(+2)", k_cndmask, 4);
    run<uint32_t>("min+add", k_min32, 2);
    run<double>("fma64", k_fma64, 1);
    run<double>("cvt64 x2+add", k_cvt64, 3);
    run<uint32_t>("bfe+add+add", k_bfe, 3);
    return 0;
}
