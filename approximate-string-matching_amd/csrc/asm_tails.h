// Stale-tail model of the reference's Greedy buffers — host pre-pass, sequential mode only.
//
// hurdle_matrix keeps two persistent 128-byte buffers (GASMA/hurdle_matrix.h:136-137).  reset() copies only
// the first m / n characters in (:630-631) and sse3_convert2bit1 then permutes each whole buffer in place
// (GASMA/bit_convert.cpp:265-330: after[q] = before[8*(q mod 16) + P[q div 16]], P = {0,2,1,3,4,6,5,7}).
// So the bytes beyond a string's end that the conversion of pair t sees are scrambled characters of earlier
// pairs.  This pass replays that chain in batch order (it is a strict sequential dependency through the
// buffers, carried before the batch is sharded) and emits, per pair, the bit planes of those tail bytes only;
// the pack kernel ORs them into granule 0.  Initial buffer content is pinned to zero (the reference's is
// indeterminate heap memory).
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

static inline void asm_resolve_tails_host(int64_t n, const char* reads, const uint32_t* read_off, const char* refs,
                                          const uint32_t* ref_off, uint32_t* tails /* [4][n][4] */) {
    static const int P[8] = {0, 2, 1, 3, 4, 6, 5, 7};
    int src[128];
    for (int q = 0; q < 128; q++) src[q] = 8 * (q & 15) + P[q >> 4];
    uint8_t buf[2][128], tmp[128];
    memset(buf, 0, sizeof(buf));
    for (int64_t i = 0; i < n; i++) {
        const char* str[2] = {reads + read_off[i], refs + ref_off[i]};
        int len[2] = {(int)(read_off[i + 1] - read_off[i]), (int)(ref_off[i + 1] - ref_off[i])};
        for (int s = 0; s < 2; s++) {
            if (len[s] > 128) len[s] = 128;
            memcpy(buf[s], str[s], (size_t)len[s]);
            uint32_t p0[4] = {0, 0, 0, 0}, p1[4] = {0, 0, 0, 0};
            for (int q = len[s]; q < 128; q++) {
                const uint8_t c = buf[s][q];
                if (c == 'C' || c == 'T') p0[q >> 5] |= 1u << (q & 31);
                if (c == 'G' || c == 'T') p1[q >> 5] |= 1u << (q & 31);
            }
            memcpy(tails + ((size_t)(2 * s + 0) * n + i) * 4, p0, 16);
            memcpy(tails + ((size_t)(2 * s + 1) * n + i) * 4, p1, 16);
            for (int q = 0; q < 128; q++) tmp[q] = buf[s][src[q]];
            memcpy(buf[s], tmp, 128);
        }
    }
}
