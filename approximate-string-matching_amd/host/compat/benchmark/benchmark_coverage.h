// Host string form of the reference's coverage helpers (GASMA/benchmark/benchmark_coverage.h:26-67 and :73-91) for callers that
// use them directly on one pair, as GASMA/main.cpp:13-16 does.  The harness's [Coverage] counter does not go through these: it
// is computed for the whole batch on the device (asm_coverage, csrc/asm_cover.h).
#pragma once
#include <cstddef>
#include <string>

// Characters of s1 inside runs of '=' / 'M' that are at least `threshold` long; 'X' skips both strings, 'I' the read, 'D' the
// reference; any other operation letter is ignored (benchmark_coverage.h:40-64).
inline std::string long_consecutive_matching_substring(const char* s1, const char* /*s2*/, const std::string& CIGAR, int threshold = 3) {
    std::string lcm;
    std::size_t i = 0, p = 0;
    while (p < CIGAR.size()) {
        while (p < CIGAR.size() && (CIGAR[p] == ' ' || CIGAR[p] == '\t' || CIGAR[p] == '\n')) p++;
        bool neg = false, digits = false;
        if (p < CIGAR.size() && (CIGAR[p] == '-' || CIGAR[p] == '+')) neg = CIGAR[p++] == '-';
        long len = 0;
        while (p < CIGAR.size() && CIGAR[p] >= '0' && CIGAR[p] <= '9') len = len * 10 + (CIGAR[p++] - '0'), digits = true;
        if (!digits) break; /* the reference's stream extraction fails here and its loop ends */
        while (p < CIGAR.size() && (CIGAR[p] == ' ' || CIGAR[p] == '\t' || CIGAR[p] == '\n')) p++;
        if (p >= CIGAR.size()) break;
        const char op = CIGAR[p++];
        if (neg) len = -len;
        if (op == 'X' || op == 'I') {
            i += (std::size_t)len;
        } else if (op == '=' || op == 'M') {
            for (long q = 0; q < len; q++, i++)
                if (len >= threshold) lcm += s1[i];
        }
    }
    return lcm;
}

// s2 is a subsequence of s1 (benchmark_coverage.h:73-91)
inline bool covers(const std::string& s1, const std::string& s2) {
    std::size_t i = 0;
    for (char c : s2) {
        while (i < s1.size() && s1[i] != c) i++;
        if (i >= s1.size()) return false;
        i++;
    }
    return true;
}
