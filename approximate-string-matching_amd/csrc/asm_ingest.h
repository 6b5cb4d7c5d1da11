// Device-side parser of the harness's input format (benchmark_utils.h:325-352): a text of lines, line 2i = one marker
// character + read i, line 2i+1 = one marker character + reference i; the first character of every line is skipped blindly
// (`line.substr(1)`, :337,:343).  The raw bytes of a chunk of the file are copied to the GPU as they are (pinned, at PCIe
// rate) and indexed there:
//   seq_count_kernel   newlines per 4 KiB tile
//   (exclusive scan of the tile counts)
//   seq_index_kernel   byte position of every newline, in line order
//   seq_lengths_kernel read / reference length of every pair (scanned into the batch's offset arrays)
//   seq_gather_kernel  the strings, without markers and newlines, into the batch's two concatenated arrays
// after which the batch is packed like any other.  The host only counts newlines (to cut chunks at pair boundaries).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "asm_bits.h"

#define SEQ_TILE 4096 /* bytes per workgroup: 256 threads x 16 */

ASM_DEV uint32_t seq_newline_mask(uint4 v, long base, long nbytes) { /* bit q: byte base+q is '\n' and inside the chunk */
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t mask = 0u;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const uint32_t byte = (w[q >> 2] >> (8 * (q & 3))) & 0xffu;
        if (byte == 0x0au && base + q < nbytes) mask |= 1u << q;
    }
    return mask;
}

__global__ __launch_bounds__(256) void seq_count_kernel(const char* __restrict__ raw, long nbytes, uint32_t* __restrict__ tile_cnt) {
    __shared__ unsigned int s_part[4];
    const long base = (long)blockIdx.x * SEQ_TILE + (long)threadIdx.x * 16;
    uint32_t c = 0u;
    if (base < nbytes) c = (uint32_t)__popc(seq_newline_mask(*reinterpret_cast<const uint4*>(raw + base), base, nbytes)); /* buffer is padded to 16 */
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

__global__ __launch_bounds__(256) void seq_index_kernel(const char* __restrict__ raw, long nbytes, const uint32_t* __restrict__ tile_base,
                                                        uint32_t* __restrict__ nl_pos, long nlines) {
    __shared__ uint32_t s_wave[4];
    const long base = (long)blockIdx.x * SEQ_TILE + (long)threadIdx.x * 16;
    uint32_t mask = 0u;
    if (base < nbytes) mask = seq_newline_mask(*reinterpret_cast<const uint4*>(raw + base), base, nbytes);
    const uint32_t c = (uint32_t)__popc(mask);
    // exclusive prefix of c over the 256 threads: wave scan, then the four wave totals
    uint32_t incl = c;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if ((int)(threadIdx.x & 63) >= off) incl += up;
    }
    if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = tile_base[blockIdx.x] + incl - c;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) before += s_wave[w];
    uint32_t m = mask;
    while (m) {
        const int q = __builtin_ctz(m);
        m &= m - 1u;
        if ((long)before < nlines) nl_pos[before] = (uint32_t)(base + q);
        before++;
    }
}

// pair i: read = line 2i, reference = line 2i+1; a line's string is what follows its first character
__global__ __launch_bounds__(256) void seq_lengths_kernel(const uint32_t* __restrict__ nl_pos, long n, uint32_t* __restrict__ m_len,
                                                          uint32_t* __restrict__ n_len, unsigned long long* __restrict__ start_a,
                                                          unsigned long long* __restrict__ start_b) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) {
        m_len[n] = n_len[n] = 0u; /* so that an exclusive scan over n+1 entries yields the totals */
        return;
    }
    const long a0 = i == 0 ? 0 : (long)nl_pos[2 * i - 1] + 1, a1 = (long)nl_pos[2 * i];
    const long b0 = a1 + 1, b1 = (long)nl_pos[2 * i + 1];
    m_len[i] = (uint32_t)(a1 - a0 > 1 ? a1 - a0 - 1 : 0);
    n_len[i] = (uint32_t)(b1 - b0 > 1 ? b1 - b0 - 1 : 0);
    start_a[i] = (unsigned long long)(a0 + 1);
    start_b[i] = (unsigned long long)(b0 + 1);
}

__global__ __launch_bounds__(256) void seq_gather_kernel(const char* __restrict__ raw, const unsigned long long* __restrict__ start,
                                                         const uint32_t* __restrict__ off, long n, char* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    for (long i = wave; i < n; i += nwaves) { /* one wave per string: 64 contiguous bytes per load */
        const unsigned long long s = start[i];
        const uint32_t o = off[i], len = off[i + 1] - o;
        for (uint32_t q = (uint32_t)lane; q < len; q += 64u) out[o + q] = raw[s + q];
    }
}

__global__ __launch_bounds__(256) void seq_max_kernel(const uint32_t* __restrict__ m_len, const uint32_t* __restrict__ n_len, long n,
                                                      uint32_t* __restrict__ out_max) {
    uint32_t v = 0u;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        v = m_len[i] > v ? m_len[i] : v;
        v = n_len[i] > v ? n_len[i] : v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_down(v, off, 64);
        v = o > v ? o : v;
    }
    if ((threadIdx.x & 63) == 0 && v) atomicMax(out_max, v);
}
