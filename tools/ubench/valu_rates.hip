// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the integer/bit instructions the aligner
// kernels are made of, on gfx950.  Each kernel is a loop of 16 independent single-instruction chains written in inline
// asm, so the number is throughput of exactly that instruction with 8 waves resident per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define ITER 2048
#define CH 16

#define BENCH32(name, ASM)                                                                  \
    __global__ __launch_bounds__(256) void name(uint32_t* out, uint32_t a, uint32_t b) {    \
        uint32_t v[CH];                                                                     \
        for (int c = 0; c < CH; c++) v[c] = a + threadIdx.x * 3 + c;                        \
        uint32_t w = b + threadIdx.x;                                                       \
        for (int it = 0; it < ITER; it++) {                                                 \
            _Pragma("unroll") for (int c = 0; c < CH; c++) asm volatile(ASM : "+v"(v[c]) : "v"(w), "s"(b)); \
        }                                                                                   \
        uint32_t acc = 0;                                                                   \
        for (int c = 0; c < CH; c++) acc ^= v[c];                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                   \
    }

#define BENCH64(name, ASM)                                                                  \
    __global__ __launch_bounds__(256) void name(uint32_t* out, uint32_t a, uint32_t b) {    \
        unsigned long long v[CH];                                                           \
        for (int c = 0; c < CH; c++) v[c] = ((unsigned long long)a << 20) + threadIdx.x * 3 + c; \
        unsigned long long w = ((unsigned long long)b << 33) + threadIdx.x;                 \
        uint32_t w32 = (b + threadIdx.x) & 31;                                              \
        for (int it = 0; it < ITER; it++) {                                                 \
            _Pragma("unroll") for (int c = 0; c < CH; c++) asm volatile(ASM : "+v"(v[c]) : "v"(w), "s"(b), "v"(w32)); \
        }                                                                                   \
        unsigned long long acc = 0;                                                         \
        for (int c = 0; c < CH; c++) acc ^= v[c];                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)acc ^ (uint32_t)(acc >> 32); \
    }

BENCH32(k_xor, "v_xor_b32 %0, %0, %1")
BENCH32(k_add, "v_add_u32 %0, %0, %1")
BENCH32(k_shl, "v_lshlrev_b32 %0, %2, %0")
BENCH32(k_and_or, "v_and_or_b32 %0, %0, %1, %1")
BENCH32(k_or3, "v_or3_b32 %0, %0, %1, %1")
BENCH32(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %1")
BENCH32(k_add3, "v_add3_u32 %0, %0, %1, %1")
BENCH32(k_xad, "v_xad_u32 %0, %0, %1, %1")
BENCH32(k_bfe, "v_bfe_u32 %0, %0, 3, 5")
BENCH32(k_bfi, "v_bfi_b32 %0, %1, %0, %1")
BENCH32(k_alignbit, "v_alignbit_b32 %0, %0, %1, %2")
BENCH32(k_ffbl, "v_ffbl_b32 %0, %0")
BENCH32(k_ffbh, "v_ffbh_u32 %0, %0")
BENCH32(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
BENCH32(k_min, "v_min_u32 %0, %0, %1")
BENCH32(k_med3, "v_med3_i32 %0, %0, %1, %1")
BENCH32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
BENCH32(k_mul24, "v_mul_u32_u24 %0, %0, %1")
BENCH32(k_mad24, "v_mad_u32_u24 %0, %0, %1, %1")
BENCH32(k_cmp_cnd, "v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc")
BENCH32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
BENCH32(k_not, "v_not_b32 %0, %0")
BENCH32(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
BENCH32(k_perm, "v_perm_b32 %0, %0, %1, %1")
BENCH32(k_dot4, "v_dot4_u32_u8 %0, %0, %1, %0")
BENCH32(k_bitop3, "v_bitop3_b32 %0, %0, %1, %1 bitop3:0x48")
BENCH32(k_sad, "v_sad_u8 %0, %0, %1, %0")
BENCH32(k_dpp_mov, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
BENCH32(k_dpp_wave, "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf")
BENCH32(k_readlane, "v_readlane_b32 s20, %0, 3\n\tv_xor_b32 %0, s20, %0")
BENCH64(k_shl64, "v_lshlrev_b64 %0, %2, %0")
BENCH64(k_shr64, "v_lshrrev_b64 %0, %2, %0")
BENCH64(k_shr64v, "v_lshrrev_b64 %0, %3, %0")
BENCH64(k_add64, "v_lshl_add_u64 %0, %0, 0, %1")
BENCH64(k_fma64, "v_fma_f64 %0, %0, %1, %1")
BENCH64(k_mul64, "v_mul_f64 %0, %0, %1")
BENCH64(k_cvt, "v_cvt_f64_i32 %0, %3")

template <typename K>
void run(const char* name, K kern, int instr_per_op) {
    uint32_t* d;
    const int blocks = 256 * 8, threads = 256; /* 8 waves per SIMD */
    hipMalloc(&d, 4 * blocks * threads);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    kern<<<blocks, threads>>>(d, 5, 3);
    hipDeviceSynchronize();
    hipEventRecord(a);
    kern<<<blocks, threads>>>(d, 5, 3);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double ops = (blocks * 4.0 / 1024.0) * ITER * CH * instr_per_op; /* wave-instructions per SIMD */
    printf("%-12s %8.3f ms  %6.2f cycles/instr/SIMD @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / ops);
    hipFree(d);
}

int main() {
#define R(k) run(#k, k, 1)
    R(k_xor); R(k_add); R(k_shl); R(k_and_or); R(k_or3); R(k_lshl_or); R(k_add3); R(k_xad); R(k_bfe); R(k_bfi);
    R(k_alignbit); R(k_ffbl); R(k_ffbh); R(k_bcnt); R(k_min); R(k_med3); R(k_mul_lo); R(k_mul24); R(k_mad24);
    run("k_cmp_cnd", k_cmp_cnd, 2); R(k_cndmask); R(k_not); R(k_mbcnt); R(k_perm); R(k_dot4); R(k_bitop3); R(k_sad); R(k_dpp_mov); R(k_dpp_wave);
    run("k_readlane", k_readlane, 2);
    R(k_shl64); R(k_shr64); R(k_shr64v); R(k_add64); R(k_fma64); R(k_mul64); R(k_cvt);
    return 0;
}
