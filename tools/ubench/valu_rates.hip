// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the integer/bit instructions the aligner
// kernels are made of, on gfx950 — with FP32 control rows and the shader clock measured in the kernel itself.
//
// Each kernel is a loop of 16 independent single-instruction chains written in inline asm, 8 waves resident per SIMD
// (2048 blocks x 4 waves = 8 x 1024 SIMDs), so the number is the throughput of exactly that instruction.  Cycles come from
// s_memtime (shader cycles) around the loop; the clock the chip actually holds is delta(s_memtime) / delta(s_memrealtime)
// x 100 MHz (MI355X_MICROARCH.md, DVFS item 6) — nothing is priced at a nominal clock.  Control rows: v_fma_f32 /
// v_add_f32 / v_fmac_f32 / v_pk_fma_f32, whose architectural rate the guide gives as 2 cycles per wave64 instruction on a
// SIMD-32 with several waves resident.
// build: hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip ; run on the GPU box; last line is JSON.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#define ITER 32768
#define CH 16

struct Stamp {
    unsigned long long cyc, real, r0, r1;
};

// All waves of the grid start their loops together: every block checks in on a counter and spins (bounded: 3 ms of the
// 100 MHz clock, then it goes ahead anyway and the row is reported as not synchronised) until the whole grid has arrived, so
// the stamped loops of a SIMD's resident waves really run side by side.
#define STAMP_BEGIN                                                                                              \
    if (threadIdx.x == 0) {                                                                                      \
        __hip_atomic_fetch_add(gate, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                            \
        const unsigned long long t_in = __builtin_amdgcn_s_memrealtime();                                        \
        while (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x &&                \
               __builtin_amdgcn_s_memrealtime() - t_in < 300000ull)                                              \
            __builtin_amdgcn_s_sleep(8);                                                                         \
    }                                                                                                            \
    __syncthreads();                                                                                             \
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();                                                  \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#define STAMP_END                                                                                  \
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();                                    \
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                \
    if ((threadIdx.x & 63) == 0) {                                                                 \
        Stamp s;                                                                                   \
        s.cyc = c1 - c0, s.real = r1 - r0, s.r0 = r0, s.r1 = r1;                                   \
        stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s;                                  \
    }

#define BENCH32(name, ASM)                                                                                   \
    __global__ __launch_bounds__(256) void name(uint32_t* out, Stamp* stamps, unsigned* gate, uint32_t a, uint32_t b) { \
        uint32_t v[CH];                                                                                      \
        for (int c = 0; c < CH; c++) v[c] = a + threadIdx.x * 3 + c;                                         \
        uint32_t w = b + threadIdx.x;                                                                        \
        STAMP_BEGIN                                                                                          \
        for (int it = 0; it < ITER; it++) {                                                                  \
            _Pragma("unroll") for (int c = 0; c < CH; c++) asm volatile(ASM : "+v"(v[c]) : "v"(w), "s"(b));  \
        }                                                                                                    \
        STAMP_END                                                                                            \
        uint32_t acc = 0;                                                                                    \
        for (int c = 0; c < CH; c++) acc ^= v[c];                                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                                    \
    }

#define BENCH64(name, ASM)                                                                                   \
    __global__ __launch_bounds__(256) void name(uint32_t* out, Stamp* stamps, unsigned* gate, uint32_t a, uint32_t b) { \
        unsigned long long v[CH];                                                                            \
        for (int c = 0; c < CH; c++) v[c] = ((unsigned long long)a << 20) + threadIdx.x * 3 + c;             \
        unsigned long long w = ((unsigned long long)b << 33) + threadIdx.x;                                  \
        uint32_t w32 = (b + threadIdx.x) & 31;                                                               \
        STAMP_BEGIN                                                                                          \
        for (int it = 0; it < ITER; it++) {                                                                  \
            _Pragma("unroll") for (int c = 0; c < CH; c++) asm volatile(ASM : "+v"(v[c]) : "v"(w), "s"(b), "v"(w32)); \
        }                                                                                                    \
        STAMP_END                                                                                            \
        unsigned long long acc = 0;                                                                          \
        for (int c = 0; c < CH; c++) acc ^= v[c];                                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)acc ^ (uint32_t)(acc >> 32);                  \
    }

// FP32 controls
BENCH32(c_fma_f32, "v_fma_f32 %0, %0, %1, %1")
BENCH32(c_fmac_f32, "v_fmac_f32 %0, %1, %1")
BENCH32(c_add_f32, "v_add_f32 %0, %0, %1")
BENCH32(c_mul_f32, "v_mul_f32 %0, %0, %1")
BENCH64(c_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %1")
BENCH64(c_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
BENCH32(c_mov, "v_mov_b32 %0, %1")
// integer / bit instructions of the aligner kernels
BENCH32(k_xor, "v_xor_b32 %0, %0, %1")
BENCH32(k_add, "v_add_u32 %0, %0, %1")
BENCH32(k_and, "v_and_b32 %0, %0, %1")
BENCH32(k_shl, "v_lshlrev_b32 %0, %2, %0")
BENCH32(k_shr_v, "v_lshrrev_b32 %0, %1, %0")
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
BENCH32(k_max_i, "v_max_i32 %0, %0, %1")
BENCH32(k_med3, "v_med3_i32 %0, %0, %1, %1")
BENCH32(k_max3, "v_max3_i32 %0, %0, %1, %1")
BENCH32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
BENCH32(k_mul24, "v_mul_u32_u24 %0, %0, %1")
BENCH32(k_mad24, "v_mad_u32_u24 %0, %0, %1, %1")
BENCH32(k_cmp_cnd, "v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc")
BENCH32(k_cmp_s, "v_cmp_lt_u32 s[20:21], %0, %1\n\tv_cndmask_b32 %0, %0, %1, s[20:21]")
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

static std::string g_json;
static std::vector<double> g_clocks;

template <typename K>
void run(const char* name, K kern, int instr_per_op, int waves_per_simd = 8) {
    uint32_t* d;
    Stamp* st;
    const int blocks = 256 * waves_per_simd, threads = 256; /* 4 waves per block, one per SIMD */
    const int nw = blocks * 4;
    hipMalloc(&d, 4 * (size_t)blocks * threads);
    hipMalloc(&st, sizeof(Stamp) * nw);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    unsigned* gate;
    hipMalloc(&gate, 4);
    hipMemset(gate, 0, 4);
    kern<<<blocks, threads>>>(d, st, gate, 5, 3);
    hipDeviceSynchronize();
    hipMemset(gate, 0, 4);
    hipEventRecord(a);
    kern<<<blocks, threads>>>(d, st, gate, 5, 3);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    std::vector<Stamp> h(nw);
    hipMemcpy(h.data(), st, sizeof(Stamp) * nw, hipMemcpyDeviceToHost);
    std::vector<double> cyc(nw), clk(nw);
    unsigned long long first = ~0ull, last = 0ull;
    for (int i = 0; i < nw; i++) {
        cyc[i] = (double)h[i].cyc, clk[i] = h[i].real ? (double)h[i].cyc / (double)h[i].real * 1e8 : 0.0;
        first = h[i].r0 < first ? h[i].r0 : first, last = h[i].r1 > last ? h[i].r1 : last;
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    const double wave_cycles = cyc[nw / 2], sclk = clk[nw / 2];
    const double per_wave = (double)ITER * CH * instr_per_op;
    // Lower bound: all waves_per_simd waves of a SIMD run their loops side by side for a wave's whole loop time.
    // Upper bound: the SIMD needs the whole span from the first wave's start to the last wave's end for them (waves that
    // were dispatched late, e.g. when the grid exactly fills the chip and a CU was full, stretch the span).
    const double cpi_stamp = wave_cycles / (waves_per_simd * per_wave);
    const double span_cycles = (double)(last - first) * 1e-8 * sclk;
    const double cpi_span = span_cycles / (waves_per_simd * per_wave);
    printf("%-14s w/SIMD %d  %8.3f ms  sclk %.3f GHz  cyc/inst/SIMD: %5.2f (wave loop)  %5.2f (first start..last end)  overlap %.2f\n",
           name, waves_per_simd, ms, sclk * 1e-9, cpi_stamp, cpi_span, wave_cycles / span_cycles);
    char buf[320];
    snprintf(buf, sizeof buf, "%s\"%s@%d\": {\"cycles\": %.3f, \"cycles_span\": %.3f, \"overlap\": %.3f, \"sclk_ghz\": %.4f, \"ms\": %.4f}",
             g_json.empty() ? "" : ", ", name, waves_per_simd, cpi_stamp, cpi_span, wave_cycles / span_cycles, sclk * 1e-9, ms);
    g_json += buf;
    g_clocks.push_back(sclk);
    hipFree(d);
    hipFree(st);
    hipFree(gate);
}

int main() {
#define R(k) run(#k, k, 1)
    // controls first, at 8, 4, 2 and 1 waves per SIMD
    for (int w : {8, 6, 4, 3, 2, 1}) {
        run("c_fma_f32", c_fma_f32, 1, w);
        run("c_add_f32", c_add_f32, 1, w);
        run("c_pk_fma_f32", c_pk_fma_f32, 1, w);
        run("k_xor", k_xor, 1, w);
        run("k_alignbit", k_alignbit, 1, w);
        run("k_shr64v", k_shr64v, 1, w);
    }
    R(c_fmac_f32); R(c_mul_f32); R(c_pk_add_f32); R(c_mov);
    R(k_add); R(k_and); R(k_shl); R(k_shr_v); R(k_and_or); R(k_or3); R(k_lshl_or); R(k_add3); R(k_xad); R(k_bfe); R(k_bfi);
    R(k_ffbl); R(k_ffbh); R(k_bcnt); R(k_min); R(k_max_i); R(k_med3); R(k_max3); R(k_mul_lo); R(k_mul24); R(k_mad24);
    run("k_cmp_cnd", k_cmp_cnd, 2); run("k_cmp_s", k_cmp_s, 2); R(k_not); R(k_mbcnt); R(k_perm); R(k_dot4); R(k_bitop3); R(k_sad);
    R(k_dpp_mov); R(k_dpp_wave);
    run("k_readlane", k_readlane, 2);
    R(k_shl64); R(k_shr64); R(k_add64); R(k_fma64); R(k_mul64); R(k_cvt);
    std::sort(g_clocks.begin(), g_clocks.end());
    printf("JSON {\"sclk_hz_median\": %.0f, \"rows\": {%s}}\n", g_clocks[g_clocks.size() / 2], g_json.c_str());
    return 0;
}
