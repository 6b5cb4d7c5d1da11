/* asm_mi355x.h — C ABI of the MI355X-native batched pair aligner (libasm_mi355x.so).
 *
 * Drop-in boundary for ONE path of GZHoffie/approximate-string-matching: the per-pair work of its benchmark
 * harness, `benchmark::_run_benchmark` (GASMA/benchmark/benchmark_utils.h:231-259), i.e.
 *   NW     `_run_nw_sse`  benchmark_utils.h:130-150  (parasail_nw_trace_striped_sse41_128_16; penalty = -score)
 *   LEAP   `_run_LEAP`    benchmark_utils.h:156-179  (LV::load_reads/reset/run/get_ED, LEAP_SIMD/LV_BAG.cpp:110-245,356)
 *   Greedy `_run_greedy`  benchmark_utils.h:185-201  (hurdle_matrix<int_128bit>::reset/run/get_cost, hurdle_matrix.h:568,625,677)
 * plus the accuracy counters of benchmark_utils.h:249-255, the coverage metric (:214-225,256), the input definition of
 * benchmark_dataset.h, the mapper's per-hit call shape (GASMA/mapper/main.cpp:77-96) and the filtering stage in front of the
 * aligners (bit-parallel LEAP `SIMD_ED` and SHD, GASMA/benchmark/LEAP_SIMD/main.cpp:95-101,186-195).
 * The reference runs these one pair at a time on one CPU thread; this library runs a whole batch of pairs
 * per call on one GPU.  Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * Batch layout: `reads`/`refs` = concatenated ASCII (no terminators); `read_off`/`ref_off` = n+1 prefix
 * offsets (uint32); pair i = reads[read_off[i] .. read_off[i+1]) vs refs[ref_off[i] .. ref_off[i+1]).
 *
 * Threading: one handle per GPU; calls on one handle must be serialised by the caller.  Calls are
 * synchronous at return unless the name ends in `_async` (those only enqueue on the handle's stream).
 * Every function returns 0 on success or a negative ASM_E* code; asm_last_error() gives the message.
 * There is NO CPU fallback: without a usable HIP device every compute entry point fails with ASM_ENODEVICE.
 */
#ifndef ASM_MI355X_H
#define ASM_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASM_OK 0
#define ASM_EINVAL (-1)    /* bad argument (k out of range, o < e for LEAP, NULL pointer, ...)            */
#define ASM_ENODEVICE (-2) /* no HIP device / device call failed                                          */
#define ASM_ENOMEM (-3)
#define ASM_EUNSUPPORTED (-4) /* e.g. sequence longer than the kernels' compiled limit                    */

/* aligner ids — the three aligners `_run_benchmark` runs per pair */
#define ASM_NW 0
#define ASM_LEAP 1
#define ASM_GREEDY 2

/* Greedy buffer-tail mode (hurdle_matrix.h:136-137,625-631 + bit_convert.cpp:265-330) */
#define ASM_GREEDY_SEQUENTIAL 0 /* reference as run: stale bytes of earlier pairs flow into later ones   */
#define ASM_GREEDY_CLEAN 1      /* tail bytes are NUL for every pair (order independent)                  */

/* reference limits reproduced here */
#define ASM_GREEDY_MAX_LENGTH 128 /* GASMA/utils.h:23-25 — Greedy aligns the first 128 bases            */
#define ASM_GREEDY_MAX_K 50       /* GASMA/hurdle_matrix.h:8                                             */
#define ASM_LEAP_MAX_LENGTH 256   /* LEAP_SIMD/LV_BAG.h:17-18; longer pairs are undefined in the reference */
#define ASM_LEAP_AF_THRESHOLD 200 /* benchmark_utils.h:289                                               */
#define ASM_MAX_LENGTH 512        /* longest sequence any kernel here accepts                            */

typedef struct asm_handle asm_handle; /* one per GPU: device id, stream, scratch                          */
typedef struct asm_batch asm_batch;   /* a device-resident batch of read pairs (ASCII + packed bit planes) */

/* Scoring and aligner parameters — the constructor arguments of `benchmark` (benchmark_utils.h:263-289)
 * and of hurdle_matrix (hurdle_matrix.h:552-559). */
typedef struct asm_params {
    int32_t k;           /* band half-width (lanes -k..k)                                                 */
    int32_t x, o, e;     /* mismatch, gap-open (first gap base), gap-extend penalties.  NW: all >= 0, and small enough for
                            the batch: 2*(o + (maxlen-1)*e) + o < 30000 (else ASM_EUNSUPPORTED).  LEAP: x >= 1,
                            15 >= o >= e >= 1 (LV_BAG.cpp:165-166).  Greedy: >= 0                                       */
    double p_match;      /* Greedy significance model; defaults 0.80, 0.20/3, 0.40/3                      */
    double p_mismatch;
    double p_indel;
    int32_t alignment_type; /* Greedy: ASM_ALIGN_GLOBAL (default, what the harness uses) or ASM_ALIGN_SEMI_GLOBAL —
                               hurdle_matrix's alignment_type_t (hurdle_matrix.h:477,553; utils.h:554-558); LOCAL is
                               declared but unsupported in the reference (hurdle_matrix.h:467).  NW and LEAP ignore it. */
    int32_t leap_mode;      /* LEAP: LV::init's ED_modes (LEAP_SIMD/LV_BAG.h:38,65) — ASM_LEAP_GLOBAL (default; what the harness
                               passes, benchmark_utils.h:289), ASM_LEAP_LOCAL, ASM_LEAP_SEMI_FREE_BEGIN, ASM_LEAP_SEMI_FREE_END:
                               LOCAL and SEMI_FREE_BEGIN start every lane at generation 0 (LV_BAG.cpp:102-104), LOCAL and
                               SEMI_FREE_END accept any lane that reaches the end (:220-238).  NW and Greedy ignore it.  (This
                               field was `reserved_`, always 0 = GLOBAL.) */
} asm_params;
#define ASM_ALIGN_GLOBAL 0
#define ASM_ALIGN_SEMI_GLOBAL 1
#define ASM_LEAP_GLOBAL 0
#define ASM_LEAP_LOCAL 1
#define ASM_LEAP_SEMI_FREE_BEGIN 2
#define ASM_LEAP_SEMI_FREE_END 3

/* ---- library / handle ------------------------------------------------------------------------------ */
const char* asm_version(void);
void asm_default_params(asm_params* p);          /* x=o=e=1, k=3, default probabilities (benchmark.cpp:22) */
int asm_device_count(void);                       /* number of HIP devices (0 without a GPU; never fails)   */
int asm_create(asm_handle** out, int device);     /* binds the handle to HIP device `device`               */
int asm_destroy(asm_handle* h);
const char* asm_last_error(const asm_handle* h);  /* h may be NULL: last error of the calling thread       */
/* Launch everything on a caller-owned hipStream_t (e.g. torch's current stream).  NULL is HIP's legacy default stream —
 * which is what torch hands out outside a `torch.cuda.stream(...)` context — NOT "no stream": the handle's own stream is
 * non-blocking and never synchronises with the legacy one, so a caller that mixes its own work (a collective, a torch op
 * on d_penalties) with this library's must put both on one stream through this call.  asm_reset_stream goes back to the
 * handle's own stream. */
int asm_set_stream(asm_handle* h, void* hip_stream);
int asm_reset_stream(asm_handle* h);
int asm_synchronize(asm_handle* h);

/* ---- input definition: seeded restatement of `Dataset` (benchmark_dataset.h:61-253) ---- */
#define ASM_GEN_EXACT_ERRORS 0 /* exactly ceil(L*err) edit operations per pair (Dataset exact=true)        */
#define ASM_GEN_PER_BASE 1     /* independent per-base substitution / insertion / deletion (SRR611076-shaped) */
#define ASM_GEN_UP_TO_ERRORS 2 /* Dataset exact=false, the "lt_eq" files (benchmark_dataset.h:153-156,246-250): the number of
                                  edit operations is uniform in 0 .. ceil(L*err) - 1; everything else as ASM_GEN_EXACT_ERRORS */
typedef struct asm_gen_config {
    uint64_t seed;
    int32_t kind;
    int32_t len_lo, len_hi; /* read length uniform in [len_lo, len_hi]                                     */
    float err;              /* ASM_GEN_EXACT_ERRORS: error rate                                           */
    float mismatch_rate;    /* ASM_GEN_EXACT_ERRORS: P(edit is a substitution); Dataset uses 0.96         */
    float p_sub, p_ins, p_del; /* ASM_GEN_PER_BASE rates per base                                         */
} asm_gen_config;
/* Host generator.  Pairs [first, first+n) of the seeded stream (every pair has its own counter-based RNG
 * state, so any slice can be generated independently — this is how shards get their part).
 * With reads == NULL only the offsets are produced (sizing pass). */
int asm_generate_pairs(const asm_gen_config* cfg, int64_t first, int64_t n, uint32_t* read_off,
                       uint32_t* ref_off, char* reads, size_t reads_cap, char* refs, size_t refs_cap);

/* ---- device-resident batches ------------------------------------------------------------------------- */
/* Copies the ASCII batch to the GPU and packs it (pack kernel: ASCII -> 2 bit planes per string, the
 * device counterpart of sse3_convert2bit1, bit_convert.cpp:248-369).  greedy_mode selects what the bit
 * planes hold beyond each string's end (the bytes Greedy's conversion would see). */
int asm_batch_upload(asm_handle* h, int64_t n, const char* reads, const uint32_t* read_off,
                     const char* refs, const uint32_t* ref_off, int greedy_mode, asm_batch** out);
/* Generates pairs [first, first+n) of the seeded stream directly in HBM (device generator; bit-identical
 * to asm_generate_pairs) and packs them: no PCIe traffic. */
int asm_batch_generate(asm_handle* h, const asm_gen_config* cfg, int64_t first, int64_t n, int greedy_mode,
                       asm_batch** out);
/* Seed-hit batches — the shape in which the reference's read mapper calls the Greedy aligner (GASMA/mapper/main.cpp:
 * 67-96): the reference text is uploaded once and stays in HBM; for hit i of read i at 0-based reference position
 * hit_pos[i] the pair is (read i, reference[start, start + len_i + 1)) with start = hit_pos ? hit_pos - 1 : 0, clipped at
 * the reference's end (mapper/main.cpp:79-80).  Windows are gathered on the device.  MAPQ = 60 + Greedy cost (:95). */
typedef struct asm_reference asm_reference;
int asm_reference_upload(asm_handle* h, const char* text, size_t len, asm_reference** out);
int asm_reference_free(asm_handle* h, asm_reference* r);
int asm_batch_from_hits(asm_handle* h, const asm_reference* ref, int64_t n, const char* reads, const uint32_t* read_off,
                        const uint64_t* hit_pos, int greedy_mode, asm_batch** out);
/* Greedy's sequential mode ACROSS batches — shards of one file on several GPUs, or chunks of a streamed file.  The
 * reference's two 128-byte buffers (hurdle_matrix.h:136-137) live on from pair to pair: reset() overwrites their first m / n
 * bytes (:630-631) and every conversion permutes them in place (bit_convert.cpp:265-330), so what pair t sees beyond its
 * strings depends on every earlier pair of the file.  Only the 2-bit codes matter, so the chain's state is 256 codes,
 * state[side*128 + slot], side 0 = read buffer, 1 = reference buffer; a file starts from zeros (NUL bytes).
 *   asm_batch_tail_summary  — what one batch does to the buffers: summary[side*128 + s] = code of the last character the
 *                             batch writes on the trajectory that sits in slot s before its first pair, 0xFF = untouched.
 *                             Works on a batch of either mode; needs no carry-in, so all shards compute it in parallel.
 *   asm_tail_state_advance  — host only, no device: state after a batch of n_pairs = its summary applied to the state
 *                             before it (the permutation has order 10: n_pairs mod 10 decides where trajectories end).
 *   asm_batch_resolve_tails — (re)derives the batch's tails from the given state before its first pair (NULL = zeros),
 *                             switches the batch to ASM_GREEDY_SEQUENTIAL and repacks it.  Enqueue only (the state is
 *                             copied at the call).
 * Shard r of a file: summaries of all shards are exchanged (one all-gather of 256 bytes + the shard sizes), every rank folds
 * the summaries of the shards before its own with asm_tail_state_advance and resolves — the N-GPU result then equals the
 * reference run over the whole file. */
int asm_batch_tail_summary(asm_handle* h, const asm_batch* b, uint8_t* summary /* [256] */);
int asm_tail_state_advance(uint8_t* state /* [256], in/out */, const uint8_t* summary /* [256] */, int64_t n_pairs);
int asm_batch_resolve_tails(asm_handle* h, asm_batch* b, const uint8_t* state /* [256] or NULL */);
int asm_batch_free(asm_handle* h, asm_batch* b); /* the device blocks go back to the pool of the handle that MADE the batch,
                                                    whichever live handle (or NULL) is named here; once that handle has been
                                                    destroyed its device memory went with it and only the record is freed */
int64_t asm_batch_size(const asm_batch* b);
int asm_batch_max_length(const asm_batch* b);
/* bytes of resident ASCII (reads + references) */
int64_t asm_batch_text_bytes(const asm_batch* b);
/* Copies the batch's ASCII form back to the host (buffers sized by the caller from the offsets). */
int asm_batch_download(asm_handle* h, const asm_batch* b, uint32_t* read_off, uint32_t* ref_off,
                       char* reads, size_t reads_cap, char* refs, size_t refs_cap);
/* Re-runs only the pack kernel on the batch's resident ASCII (enqueue only). */
int asm_batch_pack_async(asm_handle* h, asm_batch* b);

/* ---- the hot path -------------------------------------------------------------------------------------- */
/* One aligner over a resident batch; d_penalties = DEVICE pointer to n int32 (enqueue only).
 * Replaces, per pair: ASM_NW -> -parasail score (benchmark_utils.h:139-142); ASM_LEAP -> LV::get_ED()
 * (LV_BAG.cpp:356; -1 where no lane passes within 200); ASM_GREEDY -> hurdle_matrix::get_cost()
 * (hurdle_matrix.h:677). */
int asm_align_batch_async(asm_handle* h, const asm_batch* b, int aligner, const asm_params* p,
                          int32_t* d_penalties);
/* Same as asm_align_batch_async with an optional per-pair work estimate (device int32[n], input order; NULL = none).
 * The estimate only steers scheduling — LEAP sorts its pairs by it (inside each workgroup, or over the whole width class with
 * one radix pass for long strings and general penalties) so that the pairs a wave waits for need about the same number of
 * generations — and never changes a result.  `_run_benchmark` passes the NW penalties of the same pairs, or, where NW is
 * not run, the Greedy penalties. */
int asm_align_batch_hinted_async(asm_handle* h, const asm_batch* b, int aligner, const asm_params* p,
                                 const int32_t* d_work_hint, int32_t* d_penalties);
/* Greedy with its CIGAR (hurdle_matrix::get_CIGAR, hurdle_matrix.h:613; built by _update_CIGAR :238-251 — lane switches
 * as nI / nD, runs of matches AND mismatches as nM, and the final hop's run is the hurdle count, :589).  d_ops = device
 * uint16[n][cap], entry = count << 3 | op with op 0 'M', 1 'I', 2 'D' (3 '=' and 4 'X' appear in NW CIGARs only); d_nops = device uint8[n] = entries produced (a
 * value above cap means the row was truncated).  Enqueue only. */
int asm_greedy_cigar_batch_async(asm_handle* h, const asm_batch* b, const asm_params* p, int32_t* d_penalties,
                                 uint16_t* d_ops, int cap, uint8_t* d_nops);
/* Host helper: formats one encoded row as the reference's string ("22M1D50M1D28M"). */
int asm_cigar_format(const uint16_t* ops, int nops, int cap, char* out, size_t out_cap);
/* The harness's coverage counter (benchmark_utils.h:214-225,256-258; benchmark_coverage.h:26-91): for every pair,
 * does LCM(read, Greedy CIGAR, threshold 1) cover LCM(read, NW CIGAR, threshold 3)?  Needs the Greedy CIGAR rows of
 * asm_greedy_cigar_batch_async.  The NW alignment is traced back on the device with this library's own documented
 * preference (in H the diagonal, then a gap in the read 'D', then a gap in the reference 'I'; inside a gap, extending it —
 * parasail's is internal and unpinned), for any penalties p->x, p->o, p->e the NW aligner accepts.  Unit penalties run a
 * banded bit-parallel pass with `window` = 32 or 64 rows first; pairs it cannot answer (distance above window/2 - 3) and
 * every pair under other penalties go through a full Gotoh matrix with stored directions, so every pair gets an answer:
 * d_cover[i] = 1 covers / 0 does not (2 = not determined is no longer produced); d_counters[0] += covered,
 * d_counters[1] += not determined (stays 0).  d_nw_ops (optional) receives the NW CIGAR rows in traceback (reverse) order,
 * entries as above.  Synchronous. */
int asm_coverage(asm_handle* h, const asm_batch* b, const asm_params* p, const uint16_t* d_greedy_ops, int greedy_cap,
                 const uint8_t* d_greedy_nops, int window, uint8_t* d_cover, uint16_t* d_nw_ops, int nw_cap,
                 uint8_t* d_nw_nops, unsigned long long* d_counters);
/* Convenience: host in, host out (upload + pack + align + copy back).  The reference-shaped call:
 * align(read, ref, k) for every pair of the batch. */
int asm_align_batch(asm_handle* h, int aligner, int64_t n, const char* reads, const uint32_t* read_off,
                    const char* refs, const uint32_t* ref_off, const asm_params* p, int greedy_mode,
                    int32_t* penalties);
/* ---- input side: the harness's file format (benchmark_utils.h:325-352) ---------------------------------------------------
 * Line 2i = one marker character + read i, line 2i+1 = one marker character + reference i ('>' and '<' as Dataset writes
 * them, benchmark_dataset.h:229,234; the first character of every line is skipped blindly, :337,:343).
 * asm_batch_from_text: a batch out of such a text held in host memory; the raw bytes go to the GPU as they are and are
 * parsed there (newline index, offsets, gather: csrc/asm_ingest.h). */
int asm_batch_from_text(asm_handle* h, const char* text, size_t nbytes, int greedy_mode, asm_batch** out);
/* asm_stream_seq_file: `read_string_file` + `run` for a file of any size, in chunks of about chunk_bytes (0 = 64 MiB; from 32 MiB
 * on, the first chunk is a sixteenth of that and the chunks double up to it, so that the first transfer starts early): reader
 * threads fill pinned buffers (three in rotation) and cut them at pair boundaries, the raw bytes are copied to HBM on a copy
 * stream while the chunk before is parsed, packed and aligned on the handle's stream, and results of the chunk before that are
 * handed over.  aligner_mask: bit 0 NW, 1 LEAP, 2 Greedy.  nw / leap / greedy: host arrays of out_cap entries (or NULL),
 * pair i of the file at index i.  answers (optional, host, n_answers entries): read_answer_file's integers
 * (benchmark_utils.h:358-368; INT32_MIN = "use the NW penalty").  With ASM_GREEDY_SEQUENTIAL the stale-tail chain of
 * hurdle_matrix.h:136-137 runs through the chunk boundaries, so the result equals the reference run over the whole file.
 * max_pairs > 0 stops after that many pairs (benchmark's max_test_num, benchmark_utils.h:331).  Synchronous. */
typedef struct asm_stream_stats {
    int64_t pairs, chunks, bytes;       /* pairs aligned, chunks shipped, file bytes shipped to the GPU                   */
    unsigned long long counters[4];     /* total_tests, nw_correct, LEAP_correct, greedy_correct (benchmark_utils.h:249-255) */
    double seconds;                     /* wall clock of the whole call                                                  */
    double seconds_read;                /* of which the reader threads were busy (file -> pinned memory, newline count)    */
    int32_t max_length, reserved_;
} asm_stream_stats;
int asm_stream_seq_file(asm_handle* h, const char* path, const asm_params* p, int greedy_mode, int aligner_mask,
                        int64_t chunk_bytes, int64_t max_pairs, int32_t* nw, int32_t* leap, int32_t* greedy, int64_t out_cap,
                        const int32_t* answers, int64_t n_answers, asm_stream_stats* stats);

/* ---- filtering stage in front of the aligners: bit-parallel LEAP (SIMD_ED) and SHD ------------------------------
 * Replaces, per pair of the batch, the stdin filter driver's sequence (GASMA/benchmark/LEAP_SIMD/main.cpp:95-101,
 * 186-195): SIMD_ED::init_levenshtein(ed_threshold, ED_GLOBAL, shd_enable) once, then load_reads(read, ref,
 * min(m, 256)) / calculate_masks() / reset() / run() / check_pass() / get_ED()  (SIMD_ED.h:47-70, SIMD_ED.cpp:269-352,
 * 748-753).  d_ed[i] = get_ED() (converge_ED) when check_pass() holds, -1 otherwise.
 * mode: the reference keeps its verdict state from pair to pair (a pair that never reaches the end inherits the verdict
 * of the last pair that did; an exact pair reports the previous converge_ED) — ASM_FILTER_SEQUENTIAL reproduces that in
 * batch order, starting from state = {final_ED, lane distance, converge_ED} (NULL: zeros; the reference leaves them
 * uninitialised) and writes the state after the last pair back into it, so that a file processed in chunks (BATCH_RUN,
 * main.cpp:20) chains exactly; ASM_FILTER_CLEAN judges every pair on its own (never reached: -1; exact: 0).
 * ed_threshold in [1, 32]; with shd_enable at most 16 (MAX_ERROR_AVX, LEAP_SIMD/mask.h:21).  Enqueue only in clean
 * mode; sequential mode returns after its scans have run. */
#define ASM_FILTER_SEQUENTIAL 0
#define ASM_FILTER_CLEAN 1
int asm_simd_ed_batch_async(asm_handle* h, const asm_batch* b, int ed_threshold, int shd_enable, int mode,
                            int32_t* state, int32_t* d_ed);
/* The same with init_levenshtein's ED_modes argument (SIMD_ED.h:41,49; ed_mode = ASM_LEAP_GLOBAL / LOCAL / SEMI_FREE_BEGIN /
 * SEMI_FREE_END): LOCAL and SEMI_FREE_BEGIN keep every lane live from generation 0 (SIMD_ED.cpp:246-266); LOCAL and
 * SEMI_FREE_END pass exactly when a lane reaches the end, d_ed[i] = final_ED, and neither read nor change `state`
 * (:348-351,748-753 — `mode` makes no difference there); SEMI_FREE_BEGIN follows GLOBAL's rules, carried state included. */
int asm_simd_ed_mode_batch_async(asm_handle* h, const asm_batch* b, int ed_threshold, int shd_enable, int mode, int ed_mode,
                                 int32_t* state, int32_t* d_ed);
/* SIMD_ED in affine mode (LEAP_SIMD/SIMD_ED.h:50, SIMD_ED.cpp:435-616): init_affine(gap_threshold, af_threshold, ED_GLOBAL, x,
 * o, e) then, per pair, load_reads(read, ref, min(m, 256)) / calculate_masks() / reset() / run() / check_pass() / get_ED(), in
 * CLEAN form: every pair starts from the tables init_affine leaves.  (The reference object keeps its I/D/end tables from pair to
 * pair and never clears them, so as run a verdict depends on all pairs seen before; this entry point gives the verdict of the
 * first pair after init_affine.)  d_ed[i] = get_ED() = converge_ED when the pair passes — 1000000 for a pair whose main lane
 * reaches the end at e = 0, as the reference returns it —, -1 when it does not.  gap_threshold in [1, 32], af_threshold in
 * [1, 512], 1 <= e <= o <= 15, 1 <= x <= 15; SHD off (init_affine's default).  Enqueue only. */
int asm_simd_ed_affine_batch_async(asm_handle* h, const asm_batch* b, int gap_threshold, int af_threshold, int x, int o, int e,
                                   int32_t* d_ed);
/* The same with init_affine's SHD_enable = true and SHD_threshold (LEAP_SIMD/SIMD_ED.h:50, SIMD_ED.cpp:445-446,489-492):
 * run_affine starts with bit_vec_filter_avx(hamming_masks + 1, buffer_length, SHD_threshold) — the mask-array SHD
 * (SHD.cpp:334-372) over the first 2*SHD_threshold+1 lane masks, i.e. lanes -gap .. -gap + 2*SHD_threshold (centred on the main
 * lane only when the two thresholds are equal) — and a rejected pair does not pass (d_ed[i] = -1).  shd_threshold in
 * [0, min(16, gap_threshold)]: the reference object holds 2*gap_threshold+1 masks and would read past them.  Enqueue only. */
int asm_simd_ed_affine_shd_batch_async(asm_handle* h, const asm_batch* b, int gap_threshold, int af_threshold, int x, int o, int e,
                                       int shd_threshold, int32_t* d_ed);
/* ... and with init_affine's ED_modes argument (SIMD_ED.h:41,50): mode = ASM_LEAP_GLOBAL / LOCAL / SEMI_FREE_BEGIN / SEMI_FREE_END
 * (the enum is LV's).  LOCAL and SEMI_FREE_BEGIN start every lane at generation 0 (SIMD_ED.cpp:476-478,497-516); LOCAL and
 * SEMI_FREE_END accept any lane that reaches the end and d_ed[i] is final_ED there (:589-610,748-753) — 0, not 1000000, for a
 * pair exact at generation 0.  shd_threshold < 0: SHD off.  Enqueue only. */
int asm_simd_ed_affine_mode_batch_async(asm_handle* h, const asm_batch* b, int gap_threshold, int af_threshold, int x, int o, int e,
                                        int shd_threshold, int mode, int32_t* d_ed);
/* bit_vec_filter_avx(read planes, ref planes, min(m, 256), max_error) (LEAP_SIMD/SHD.h:17-18, SHD.cpp:241-322):
 * d_pass[i] = 1 when the pair survives the shifted-Hamming-distance filter, 0 when it is rejected.  max_error in
 * [0, 16].  Enqueue only. */
int asm_shd_filter_batch_async(asm_handle* h, const asm_batch* b, int max_error, int32_t* d_pass);
/* accuracy counter of benchmark_utils.h:253-255: *d_count += #{i : a[i] == b[i]} (device pointers; enqueue
 * only; the caller zeroes *d_count). */
int asm_count_equal_async(asm_handle* h, const int32_t* d_a, const int32_t* d_b, int64_t n,
                          unsigned long long* d_count);

/* All counters of `_run_benchmark` (benchmark_utils.h:238,249-255) in one pass over device penalty arrays:
 * d_counters[0..3] += {total_tests, nw_correct, LEAP_correct, greedy_correct}; the correct answer of pair i is
 * d_answers[i] when d_answers is given and != INT32_MIN (read_answer_file, benchmark_utils.h:358-368), else the NW
 * penalty.  d_nw / d_leap / d_greedy / d_answers may be NULL; without d_nw a pair has a correct answer only where the
 * answers file gives one, and total_tests counts every pair regardless.  Enqueue only. */
int asm_accuracy_async(asm_handle* h, const int32_t* d_nw, const int32_t* d_leap, const int32_t* d_greedy,
                       const int32_t* d_answers, int64_t n, unsigned long long* d_counters);
/* `_run_benchmark` for a whole resident batch in one call (benchmark_utils.h:231-259): optional re-pack of the
 * resident ASCII, then every aligner whose output pointer is non-NULL, then the counters.  Enqueue only.
 * Order inside the call: Greedy beside the chain NW -> LEAP (LEAP scheduled by the NW penalties); with d_nw NULL and a wide
 * band, Greedy first and LEAP scheduled by its penalties.
 * repack: 0 = use the planes as they are; 1 = pack first, in stream order; 2 = pack first, PIPELINED: the pack of this call
 * fills a second set of planes on its own stream and so overlaps the aligners of the previous call (pack waits on memory for
 * half of its time, the aligners are instruction-bound); this call's aligners wait for it.  With 2 the caller guarantees that
 * nothing it enqueued since the previous call on this batch changes what pack reads.
 * 3 = OVERLAPPED CALLS, the form a caller with a stream of batches (or of passes) wants: as 2, and in addition nothing of this
 * call waits for the previous call's Greedy kernel — consecutive calls form three chains on streams of the library (NW -> LEAP
 * -> counters -> NW ..., Greedy -> Greedy, pack -> pack; a call's counters also wait for its Greedy).  The caller's stream is
 * NOT joined: the
 * outputs and counters of all calls so far are complete on it after asm_pipeline_join_async (asm_synchronize also waits for
 * them).  The caller ALTERNATES between two sets of output arrays from call to call (a set is written again two calls later,
 * when the library has seen its counters finish), and calls asm_pipeline_join_async before it touches the batch or the outputs
 * in any other way.  A later call with repack != 3 joins by itself.  Passing the previous overlapped call's arrays again without a
 * join in between is refused (ASM_EINVAL).  Consecutive calls may name DIFFERENT batches (a caller rotating over several resident
 * batches): they are paced like calls on one batch — the pack of call c starts behind the counters of call c - 2.  A call that
 * cannot overlap (the Greedy-first shape above, an empty batch) runs as repack = 2, ordered behind every earlier call. */
int asm_run_benchmark_async(asm_handle* h, asm_batch* b, const asm_params* p, int repack, int32_t* d_nw,
                            int32_t* d_leap, int32_t* d_greedy, const int32_t* d_answers,
                            unsigned long long* d_counters);

/* Joins the overlapped calls (repack = 3) into the handle's stream: everything they enqueued is ordered before whatever is
 * enqueued on that stream next.  Enqueue only; a no-op when there are none. */
int asm_pipeline_join_async(asm_handle* h);

/* Per-kernel timing INSIDE a caller's timed region: after asm_profile_enable(h, max_calls, kernel_mask) the next max_calls
 * calls of asm_run_benchmark_async bracket the selected kernels (bit 0 pack, 1 NW, 2 LEAP, 3 Greedy) with HIP events
 * recorded on the stream that kernel is launched on (Greedy: the handle's side stream); asm_profile_read synchronises and
 * returns ms[call][4] (-1 where a kernel was not launched or not selected).  An event record keeps the next kernel of its
 * stream from starting early, so bracket only what is needed: all four cost ~8 % of the C2 step, Greedy alone (its own
 * stream) ~1 %.  asm_profile_enable(h, 0, 0) switches it off. */
int asm_profile_enable(asm_handle* h, int max_calls, unsigned kernel_mask);
int asm_profile_read(asm_handle* h, float* ms, int cap_calls, int* n_calls);

/* ---- plain device memory helpers (so that non-torch hosts can drive the async API) --------------------- */
int asm_device_malloc(asm_handle* h, size_t bytes, void** d_ptr);
int asm_device_free(asm_handle* h, void* d_ptr);
int asm_memcpy_d2h(asm_handle* h, void* dst, const void* d_src, size_t bytes);
int asm_memcpy_h2d(asm_handle* h, void* d_dst, const void* src, size_t bytes);
int asm_memset_async(asm_handle* h, void* d_ptr, int value, size_t bytes);

/* ---- timing helper: HIP events on the handle's stream (bench.py measures kernels with these) ----------- */
int asm_timer_create(asm_handle* h, void** timer);
int asm_timer_start(asm_handle* h, void* timer);
int asm_timer_stop(asm_handle* h, void* timer);
int asm_timer_elapsed_ms(asm_handle* h, void* timer, float* ms); /* synchronises on the stop event */
int asm_timer_destroy(asm_handle* h, void* timer);

#ifdef __cplusplus
}
#endif
#endif /* ASM_MI355X_H */
