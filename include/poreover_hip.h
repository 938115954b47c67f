/* poreover_hip.h — C-ABI of libporeover_hip.so, the MI355X (gfx950) decoding engine.
 *
 * This is the drop-in boundary for PoreOver's native decode path.  In the reference the
 * boundary is the set of Python-callable functions that Cython exports from
 * poreover/decoding/decoding_cpp.pyx, poreover/decoding/decoding_cy.pyx and
 * poreover/align/align.pyx; each entry point below names the reference interface it
 * replaces.  The reference decodes ONE read / pair per call and gets its parallelism from
 * multiprocessing.Pool (decode.py:158-162, pair_decode.py:292-297); here every entry point is
 * BATCHED — one launch decodes n independent reads / pairs — which is the natural shape for a
 * GPU.  The single-item Python wrappers with the reference's exact signatures
 * (poreover_amd/decoding/decoding_cpp.py) call these with n == 1.
 *
 * Conventions
 *   - plain C types only; every pointer is a DEVICE pointer unless the name ends in _h
 *   - the device-pointer calls enqueue their kernels on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream) and return without waiting for them; scratch comes from `ws`, and none allocates or frees device memory —
 *     except that the first pair beam search (method row_col with an envelope, W <= 12) of a process on a device creates,
 *     per tree model, a pool of value-store slices that the library keeps (5.3 GB for the default model).
 *     The calls whose geometry depends on batch maxima they are not given (po_pair_decode_batch, po_beam2d_batch,
 *     po_beam1d_batch, the lattice / alignment calls) first read the offset tables back (one small D2H copy and a
 *     stream synchronise) — enqueue the inputs before calling; po_viterbi_batch (CTC kinds) and po_ingest_batch do
 *     not.  The *_h forms and po_pipeline_pair_decode are synchronous.
 *   - a workspace may be reused from call to call and its contents need not be preserved (beam2d_kernel keeps
 *     per-workgroup epoch counters in it and clears what it finds untagged: first use)
 *   - po_last_error() is per host thread and refers to the last failing call of that thread; the profiling aid at
 *     the end of this header (po_profile_*) is process-wide and meant for one measuring thread
 *   - y:       concatenated C-contiguous (T_i, C) float64 natural-log probabilities
 *     y_off:   int64[n+1] ROW offsets into y (read i owns rows [y_off[i], y_off[i+1]))
 *   - env:     concatenated (U_i, 2) int32 half-open column ranges, row-aligned with y1
 *   - seq:     output characters; read/pair i writes at seq + seq_off[i], at most
 *              seq_off[i+1]-seq_off[i] bytes (no terminator); seq_len[i] gets the length
 *   - status:  int32[n], 0 on success or a PO_E_* code for that item (the launch itself
 *              returns 0 unless the arguments are unusable)
 *   - model:   PO_MODEL_CTC ('ctc'), PO_MODEL_MERGE ('ctc_merge_repeats'),
 *              PO_MODEL_FLIPFLOP ('ctc_flipflop')       (decode.py:172, pair_decode.py:147)
 *   - alphabet: HOST string of A <= 4 symbols (the reference's alphabet_ argument, default
 *              "ACGT"); C == A + 1 for the CTC models (blank last), C == 2A for flip-flop
 */
#ifndef POREOVER_HIP_H
#define POREOVER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PO_MODEL_CTC 0
#define PO_MODEL_MERGE 1
#define PO_MODEL_FLIPFLOP 2

#define PO_METHOD_ROW 0
#define PO_METHOD_ROW_COL 1
#define PO_METHOD_GRID 2
#define PO_METHOD_GRID_NOENV 3 /* po_beam2d_workspace_bytes only: grid without an envelope (env == NULL, method != row) */

#define PO_KIND_POREOVER 0
#define PO_KIND_BONITO 1
#define PO_KIND_FLIPFLOP 2

#define PO_OK 0
#define PO_E_CAP (-1)        /* an output or workspace buffer is too small                    */
#define PO_E_ARG (-2)        /* unusable argument                                             */
#define PO_E_ENVELOPE (-3)   /* envelope on which the reference itself is undefined           */
#define PO_E_NOMEM (-4)      /* per-item node arena / band capacity exceeded                  */
#define PO_E_DIVERGE (-5)    /* input on which the reference never terminates                 */
#define PO_E_UNSUPPORTED (-6)/* valid for the reference, not yet handled by this engine       */
#define PO_E_HIP (-7)        /* HIP runtime error (see po_last_error)                         */
#define PO_SKIP_LENGTH (-10) /* pair skipped: |len1 - len2| > 1000   (pair_decode.py:372-375) */
#define PO_SKIP_IDENTITY (-11)/* pair skipped: identity < 0.5        (pair_decode.py:395-398) */

/* ---- library / device ----------------------------------------------------------------- */
int po_version(void);
int po_device_count(void);
int po_set_device(int device);
const char* po_last_error(void);
/* name, compute units and clock of `device`; returns 0 on success */
int po_device_info(int device, char* name, int name_cap, int* compute_units, int* clock_khz,
                   size_t* total_mem);

/* test / tuning hook: which kernel serves the pair beam search.  PO_ROUTE_AUTO = the engine's choice (DESIGN.md §3.3:
 * beam2d_reg_kernel for row_col with an envelope, every tree model, W <= 12; beam2d_kernel elsewhere); PO_ROUTE_REG = the
 * same, named; PO_ROUTE_LEGACY = always beam2d_kernel; defer_odd != 0: the register-state kernel hands every odd pair to
 * beam2d_kernel (exercises the hand-over).  Process-wide; results are identical on every route.
 * (Values 1 and 3 named the two-pairs-per-wave and LDS-ring kernels of rounds 1 - 4, retired in round 5: PO_E_ARG.) */
#define PO_ROUTE_AUTO 0
#define PO_ROUTE_LEGACY 2
#define PO_ROUTE_REG 4   /* beam2d_reg_kernel: element state in registers, values in the (tag-free) HBM store */
/* defer_odd bits 1 and 2 (values 2, 4) are further test hooks of that hand-over: the kernel runs with a dozen row groups /
 * with a tree arena of a few nodes, so that pairs run out of them and are handed on (tests/test_gpu_parity_2d.py). */
int po_set_pair_route(int route, int defer_odd);
/* How the register-state pair kernel computes the window of a NEW element (the children of a node that entered the beam:
 * BeamSearch.h:342-375 -> update_prob over the whole window, PrefixTree.h:518-531).
 *   PO_CHAIN_SERIAL (default): the reference's serial logaddexp chain, operation for operation.
 *   PO_CHAIN_CLOSED_FORM: x_t = B_t + log(sum_{s<=t} exp(p_{s-1} + y_s - B_s)), B = running sum of the stay (blank) column —
 *        one exp per (element, time), a prefix sum in the probability domain, one log; time-parallel (8 lanes per chain).
 *        The VALUES differ from the serial chain's by ~ 1e-12 (they are closer to the exact value than the chain's:
 *        scripts/check_chain_scan.cpp); decoded strings are held to north_star's tolerance (<= 0.1 % edit distance; measured:
 *        0 differing pairs of the bench's 10 000 and of the parity suite).  A chain whose finite terms span more than 600 nats
 *        sends its step to the kernel's general scan (the serial chain).  Round 6 built it as the one remaining lever on the
 *        headline — and measured it SLOWER than the serial chain (65 vs 52 ms per 10 000 pairs: profiles/r06_ab_chain_scan.txt,
 *        DESIGN.md 3.3 says why), so it is opt-in (also: environment variable PO_CHAIN_CLOSED as the initial value).
 *   PO_CHAIN_CLOSED_GUARD3: test hook — the closed form with its 600-nat guard at 3 nats, so that most steps take the hand-over.
 * Applies to the one-value tree model (ctc) on the register-state route.  Process-wide. */
#define PO_CHAIN_SERIAL 0
#define PO_CHAIN_CLOSED_FORM 1
#define PO_CHAIN_CLOSED_GUARD3 2
int po_set_chain_mode(int mode);
int po_get_chain_mode(void);
/* The register-state pair kernel keeps its value stores and tree arenas in a slice POOL the library owns: one per device, tree
 * model and lane layout (beam_width <= 6 / 7..12), as many slices as the device holds pair waves (<ctc, W <= 6>: 4 096 x 1.28 MB
 * = 5.3 GB), made by the first workspace-size query that selects the route and kept for the life of the process.
 * po_reg_pool_prewarm makes the pool of (model, beam_width) on the current device ahead of time (PO_E_NOMEM if it cannot be
 * allocated: beam2d_kernel then serves the route, with results identical); po_reg_pool_release waits for the device and frees
 * every pool of the current device — a long-lived process that has used several models and widths holds tens of GB otherwise. */
int po_reg_pool_prewarm(int model, int beam_width);
int po_reg_pool_release(void);
/* test hook: pairs the register-state kernel or its pre-pass handed to beam2d_kernel on the current device since the last
 * reset (windows beyond its store geometry or its packed walk records, row groups or arena exhausted, non-monotone envelopes,
 * the defer_odd hooks); reset != 0 clears the count.  Synchronises the device; -1 on a HIP error. */
long long po_debug_deferred_pairs(int reset);
/* test / tuning hook: legacy != 0 -> the banded aligner (align.pyx:100-178) runs the row-at-a-time kernel that stores the
 * score table instead of the skewed-wavefront kernel.  Process-wide; results are identical. */
int po_set_align_route(int legacy);

/* ---- trace ingest ------------------------------------------------------------------------------
 * replaces decode.logit_to_log_likelihood (decode.py:34-39), the uint8 trace scaling of
 * model_from_trace (decode.py:89-93,99-103), the Bonito column order (decode.py:79) and
 * transducer.reverse_complement (transducer.py:68-70,104-106) — one streaming pass on the device.
 *   mode PO_INGEST_LOGITS_F32: src float32 (rows, C) logits -> x - logsumexp(x), in float32 like the
 *        reference, widened to float64;  PO_INGEST_TRACE_U8: src uint8 -> log((x+1e-7)/(255+1e-7));
 *        PO_INGEST_F64: src float64, copied.   perm (host, C ints or NULL): out[:, c] = value[:, perm[c]].
 *   reverse != 0: every item [row_off[i], row_off[i+1]) is time-reversed.   out: float64 (rows, C). */
#define PO_INGEST_LOGITS_F32 0
#define PO_INGEST_TRACE_U8 1
#define PO_INGEST_F64 2
int po_ingest_batch(const void* src, const int64_t* row_off, int n, int C, int mode, const int* perm_h,
                    int reverse, double* out, void* stream);

/* ---- transducer.argmax_decode / viterbi_decode ------------------------------------------
 * replaces transducer.py:27-33 (argmax), :72-73 (poreover), :83-89 (bonito), :35-59 +
 * :94-103 (flip-flop Viterbi) and pair_decode.get_sequence_mapping (pair_decode.py:114-142).
 *   path    int8[total_rows]   per-frame state                          (may be NULL)
 *   map     int32[total_rows]  frame index of each emitted base, written at map + y_off[i]
 *                              for read i, seq_len[i] entries            (may be NULL)
 * seq capacity per read must be >= T_i.  */
size_t po_viterbi_workspace_bytes(int n, int64_t total_rows, int C, int kind);
int po_viterbi_batch(const double* y, const int64_t* y_off, int n, int C, const char* alphabet, int kind,
                     int8_t* path,
                     char* seq, const int64_t* seq_off, int32_t* seq_len, int32_t* map,
                     int32_t* status, void* ws, size_t ws_bytes, void* stream);

/* ---- decoding_cpp.cpp_beam_search ---------------------------------------------------------
 * replaces decoding_cpp.pyx:88-103 -> BeamSearch.h:400-408 beam_search(y, t_max, alphabet,
 * beam_width, model) -> beam_search_<Tree,Beam> (BeamSearch.h:18-58).  */
size_t po_beam1d_workspace_bytes(int n, int64_t total_rows, int64_t max_rows, int C, int beam_width,
                                 int model);
int po_beam1d_batch(const double* y, const int64_t* y_off, int n, int C, const char* alphabet, int beam_width,
                    int model, char* seq, const int64_t* seq_off, int32_t* seq_len, int32_t* status, void* ws,
                    size_t ws_bytes, void* stream);

/* ---- decoding_cpp.cpp_beam_search_2d ------------------------------------------------------
 * replaces decoding_cpp.pyx:107-139 -> BeamSearch.h:411-458 beam_search(y1, y2, U, V, alphabet,
 * envelope_ranges, beam_width, model, method): method row_col = :262-397 (CLI default,
 * __main__.py:89), row = :110-172 / :175-260, grid = BeamSearch2.h:33-184 (hidden upstream option).
 * env rows are [lo, hi) per row of y1; env == NULL: "row" runs its no-envelope form and every other
 * method runs grid without an envelope, as the reference's dispatcher does (BeamSearch.h:441-458) —
 * size the workspace with PO_METHOD_GRID_NOENV in that case.  grid needs row starts that do not move
 * backwards (PO_E_UNSUPPORTED otherwise) and keeps 2 x V x beam_width nodes of cell beams per pair in
 * flight; without an envelope every read-1 time of a node stays readable, so it only fits short reads
 * (PO_E_NOMEM otherwise).
 * The size query is where the register-state route's slice pool is made (first query per device, tree model and lane layout
 * that selects the route: po_reg_pool_prewarm above) — so that the launch that follows allocates nothing.  */
size_t po_beam2d_workspace_bytes(int n, int64_t total_rows1, int64_t total_rows2, int64_t max_rows1,
                                 int64_t max_rows2, int C, int beam_width, int model, int method);
int po_beam2d_batch(const double* y1, const int64_t* y1_off, const double* y2, const int64_t* y2_off,
                    const int32_t* env, int n, int C, const char* alphabet, int beam_width, int model,
                    int method, char* seq, const int64_t* seq_off, int32_t* seq_len, int32_t* status,
                    void* ws, size_t ws_bytes, void* stream);

/* ---- decoding_cpp.cpp_forward ---------------------------------------------------------------
 * replaces decoding_cpp.pyx:49-65 -> forward(y, t_max, label, alphabet, model) (PrefixTree.h:751-759):
 * log P(label | y) under the model's tree recurrence.  labels: concatenated characters (device),
 * label_off int64[n+1]; characters outside the alphabet count as its first symbol, as upstream. */
size_t po_forward_workspace_bytes(int n, int64_t max_rows, int model);
int po_forward_batch(const double* y, const int64_t* y_off, int n, int C, const char* alphabet, int model,
                     const char* labels, const int64_t* label_off, double* logp, int32_t* status, void* ws,
                     size_t ws_bytes, void* stream);

/* ---- decoding_cpp.cpp_viterbi_acceptor -------------------------------------------------------
 * replaces decoding_cpp.pyx:69-84 -> viterbi_acceptor_poreover (Forward.h:14-121): best alignment
 * path of a known label, band +-band_size around the diagonal.  path int32[total_rows] (blank = A). */
size_t po_viterbi_acceptor_workspace_bytes(int n, int64_t max_rows, int64_t max_label);
int po_viterbi_acceptor_batch(const double* y, const int64_t* y_off, int n, int C, const char* alphabet,
                              int band_size, const char* labels, const int64_t* label_off, int32_t* path,
                              int32_t* status, void* ws, size_t ws_bytes, void* stream);

/* ---- prefix_search.prefix_search_log_cy -----------------------------------------------------
 * replaces prefix_search.py:176-238 (with decoding_cy.forward_vec_log, decoding_cy.pyx:127-156):
 * Graves prefix search of each item; an item is any row range [y_off[i], y_off[i+1]) of y, so the
 * windows of `decode --algorithm prefix` (decode.py:182-188) are just finer offsets into one matrix.
 * logp: log-probability of the returned label.  CTC model ('poreover') only, as upstream. */
size_t po_prefix_search_workspace_bytes(int n, int64_t max_rows);
int po_prefix_search_batch(const double* y, const int64_t* y_off, int n, int C, const char* alphabet, char* seq,
                           const int64_t* seq_off, int32_t* seq_len, double* logp, int32_t* status, void* ws,
                           size_t ws_bytes, void* stream);

/* ---- align.global_pair / align.global_pair_banded -------------------------------------------
 * replaces align/align.pyx:29-98 (band_width <= 0: full Needleman-Wunsch) and :100-178 (banded, as
 * written upstream), match 2 / mismatch -1 / gap -1.  seqs: concatenated characters, pair i =
 * [seq_off[2i], seq_off[2i+1]) and [seq_off[2i+1], seq_off[2i+2]).  Outputs: the two alignment rows
 * (same length ncol[i], '-' for gaps) at aln1/aln2 + aln_off[i]; capacity len1 + len2 + 8 suffices. */
size_t po_align_workspace_bytes(int n, int64_t max_len1, int64_t max_len2, int band_width);
int po_align_batch(const char* seqs, const int64_t* seq_off, int n, int band_width, char* aln1, char* aln2,
                   const int64_t* aln_off, int32_t* ncol, int32_t* status, void* ws, size_t ws_bytes, void* stream);

/* the third item align.global_pair returns (align.pyx:34-52,98): its dense DP matrix, (len1 + 1) x (len2 + 1) int32 row-major
 * at dp + dp_off[i] (dp_off[n] = total cells), with the reference's score arguments.  Enqueue-only, no workspace. */
int po_nw_matrix_batch(const char* seqs, const int64_t* seq_off, int n, int match, int mismatch, int gap_cost, int32_t* dp,
                       const int64_t* dp_off, int32_t* status, void* stream);

/* ---- envelope.get_alignment_columns + build_envelope -----------------------------------------
 * replaces decoding/envelope.py:26-87: alignment rows + the frame index of every base of both reads
 * (map*, get_sequence_mapping) + signal lengths U, V -> per-row column range [lo, hi) of read 2,
 * padded and fixed up as upstream.  env rows of pair i at env + 2 * env_off[i] (U[i] rows). */
size_t po_envelope_workspace_bytes(int n, int64_t max_ncol);
int po_envelope_batch(const char* aln1, const char* aln2, const int64_t* aln_off, const int32_t* ncol, int n,
                      const int32_t* map1, const int64_t* map1_off, const int32_t* map2, const int64_t* map2_off,
                      const int32_t* U, const int32_t* V, int padding, int32_t* env, const int64_t* env_off,
                      int32_t* status, void* ws, size_t ws_bytes, void* stream);

/* ---- decoding_cpp.cpp_pair_gamma_log_envelope / decoding_cy.pair_gamma_log -------------------
 * replaces decoding_cpp.pyx:168-188 -> pair_gamma_log_envelope (Gamma.h:15-98) and the dense
 * decoding_cy.pair_gamma_log (decoding_cy.pyx:177-220): gamma(0,0) = log P(both reads emit the same
 * label).  env: (U_i + 1) rows per pair with INCLUSIVE column ends (Gamma.h:26-30), rows of pair i at
 * env + 2 * env_off[i]; env == NULL: dense.  flavor 0 = Gamma.h arithmetic (logaddexp, -inf),
 * 1 = decoding_cy arithmetic (log(exp+exp), LOG_0 = -9999), 2 = decoding_cy.pair_gamma_log_envelope
 * (decoding_cy.pyx:224-271; envelope only: log(exp+exp), -inf defaults, cells [start, min(end, V-1)] computed).
 * dense_out (optional): the full (U+1) x (V+1) gamma matrices at dense_out + dense_off[i] (-inf outside an envelope).  max_cells = largest per-pair number of
 * stored cells (sum over rows of end - start + 1), as used to size the workspace. */
size_t po_pair_gamma_workspace_bytes(int n, int64_t max_cells, int64_t max_rows1, int64_t max_rows2);
int po_pair_gamma_batch(const double* y1, const int64_t* y1_off, const double* y2, const int64_t* y2_off,
                        const int32_t* env, const int64_t* env_off, int n, int C, int flavor, int64_t max_cells,
                        double* gamma00, double* dense_out, const int64_t* dense_off, int32_t* status, void* ws,
                        size_t ws_bytes, void* stream);

/* ---- pair_decode.pair_decode_helper stage chain -------------------------------------------
 * replaces pair_decode.py:305-529 for the default route (--method envelope --algorithm beam
 * --single viterbi): 1-D Viterbi of both reads (:360-362) -> length skip (:372-375) ->
 * get_sequence_mapping (:377-382) -> align.global_pair_banded / global_pair
 * (align.pyx:100-178 / :29-98; :385-389) -> identity skip (:391-398) ->
 * envelope.get_alignment_columns + build_envelope (envelope.py:26-87; :500-501) ->
 * cpp_beam_search_2d (:166-173,511).  All stages run on the device.
 *   seq1d / seq1d_off / len1 / len2: the two 1-D basecalls (read 1 at seq1d + seq1d_off[2i],
 *                                    read 2 at seq1d + seq1d_off[2i+1])
 *   identity  float64[n]  matches / alignment columns
 *   env_out   int32[2*total_rows1]  the envelope that was used (may be NULL) */
typedef struct {
    int beam_width;        /* --beam_width 5                     */
    int model;             /* PO_MODEL_*                         */
    int method;            /* --beam_search_method row_col       */
    int padding;           /* --padding 5                        */
    int full_alignment;    /* --alignment full (0 = banded, 500) */
    int diagonal_envelope; /* --diagonal_envelope                */
    int diagonal_width;    /* --diagonal_width 50                */
} po_pair_options;
size_t po_pair_decode_workspace_bytes(int n, int64_t total_rows1, int64_t total_rows2, int64_t max_rows1,
                                      int64_t max_rows2, int C, const po_pair_options* opt);
int po_pair_decode_batch(const double* y1, const int64_t* y1_off, const double* y2, const int64_t* y2_off,
                         int n, int C, const po_pair_options* opt, char* seq1d, const int64_t* seq1d_off,
                         int32_t* len1, int32_t* len2, double* identity, int32_t* env_out, char* seq,
                         const int64_t* seq_off, int32_t* seq_len, int32_t* status, void* ws,
                         size_t ws_bytes, void* stream);

/* ---- host-buffer conveniences (numpy callers): allocate, copy in, launch, copy out, free ----
 * Same semantics as the device-pointer forms with every pointer a HOST pointer; synchronous. */
int po_ingest_batch_h(const void* src_h, const int64_t* row_off_h, int n, int C, int mode, const int* perm_h,
                      int reverse, double* out_h);
/* `poreover decode` for a batch of traces in one call (decode.py:114-192: model_from_trace + viterbi_decode /
 * cpp_beam_search): src_h = the basecaller's own output (float32 logits, uint8 trace or float64 log-probabilities, rows of
 * all reads back to back, row_off_h[0] == 0), ingest on the device as po_ingest_batch, then Viterbi (beam_width <= 0,
 * `kind`) or the 1-D beam search (`model`, beam_width) — no host log-softmax, 4 or 1 bytes per value over PCIe. */
int po_decode_1d_batch_h(const void* src_h, const int64_t* row_off_h, int n, int C, int in_mode, const int* perm_h, int reverse,
                         const char* alphabet, int kind, int beam_width, int model, char* seq_h, const int64_t* seq_off_h,
                         int32_t* seq_len_h, int32_t* status_h);
int po_viterbi_batch_h(const double* y_h, const int64_t* y_off_h, int n, int C, const char* alphabet, int kind,
                       int8_t* path_h,
                       char* seq_h, const int64_t* seq_off_h, int32_t* seq_len_h, int32_t* map_h,
                       int32_t* status_h);
int po_beam1d_batch_h(const double* y_h, const int64_t* y_off_h, int n, int C, const char* alphabet,
                      int beam_width, int model,
                      char* seq_h, const int64_t* seq_off_h, int32_t* seq_len_h, int32_t* status_h);
int po_forward_batch_h(const double* y_h, const int64_t* y_off_h, int n, int C, const char* alphabet, int model,
                       const char* labels_h, const int64_t* label_off_h, double* logp_h, int32_t* status_h);
int po_viterbi_acceptor_batch_h(const double* y_h, const int64_t* y_off_h, int n, int C, const char* alphabet,
                                int band_size, const char* labels_h, const int64_t* label_off_h, int32_t* path_h,
                                int32_t* status_h);
/* decoding_cy.viterbi_acceptor (decoding_cy.pyx:60-123), the Cython twin of the acceptor, reproduced as written
 * (dense, '>' tie rule, its own band expression; band_size 0 = whole matrix).  A label character outside the
 * alphabet is PO_E_ARG (KeyError upstream).  (Device-pointer form: po_viterbi_acceptor_batch with
 * band_size = -(band + 1).) */
int po_viterbi_acceptor_cy_batch_h(const double* y_h, const int64_t* y_off_h, int n, int C, const char* alphabet,
                                   int band_size, const char* labels_h, const int64_t* label_off_h, int32_t* path_h,
                                   int32_t* status_h);
int po_prefix_search_batch_h(const double* y_h, const int64_t* y_off_h, int n, int C, const char* alphabet,
                             char* seq_h, const int64_t* seq_off_h, int32_t* seq_len_h, double* logp_h,
                             int32_t* status_h);
int po_align_batch_h(const char* seqs_h, const int64_t* seq_off_h, int n, int band_width, char* aln1_h, char* aln2_h,
                     const int64_t* aln_off_h, int32_t* ncol_h, int32_t* status_h);
int po_nw_matrix_batch_h(const char* seqs_h, const int64_t* seq_off_h, int n, int match, int mismatch, int gap_cost,
                         int32_t* dp_h, const int64_t* dp_off_h, int32_t* status_h);
/* the same with the reference's score arguments (align.pyx:29,100: match, mismatch, gap_cost) */
int po_align_scores_batch_h(const char* seqs_h, const int64_t* seq_off_h, int n, int band_width, int match, int mismatch,
                            int gap_cost, char* aln1_h, char* aln2_h, const int64_t* aln_off_h, int32_t* ncol_h,
                            int32_t* status_h);
int po_envelope_batch_h(const char* aln1_h, const char* aln2_h, const int64_t* aln_off_h, const int32_t* ncol_h, int n,
                        const int32_t* map1_h, const int64_t* map1_off_h, const int32_t* map2_h,
                        const int64_t* map2_off_h, const int32_t* U_h, const int32_t* V_h, int padding,
                        int32_t* env_h, const int64_t* env_off_h, int32_t* status_h);
/* prefix_search.pair_prefix_search_log / _cy (prefix_search.py:247-385) on small boxes: dense gamma
 * (pair_gamma_log of the same flavour) computed on the device, then the pair prefix search.
 * flavor 0: prefix_search.py arithmetic, 1: decoding_cy.  seq at seq_h + seq_off_h[i], capacity
 * seq_off_h[i+1] - seq_off_h[i] (max(U, V) + 1 always suffices). */
int po_pair_prefix_search_batch_h(const double* y1_h, const int64_t* y1_off_h, const double* y2_h, const int64_t* y2_off_h,
                                  int n, int C, const char* alphabet, int flavor, char* seq_h, const int64_t* seq_off_h,
                                  int32_t* seq_len_h, double* logp_h, int32_t* status_h);
/* The pair prefix search WITH AN ENVELOPE — the working form of decoding_cpp.cpp_pair_prefix_search_log
 * (decoding_cpp.pyx:143-164 -> pair_prefix_search_log, PairPrefixSearch.cpp:79-229; upstream passes its gamma matrices by
 * value and crashes): gamma from the envelope DP of Gamma.h:15-98 (env_h: U_i + 1 rows with INCLUSIVE column ends, rows of
 * pair i at env_h + 2 * env_off_h[i]; -inf outside the stored ranges), the search itself as in the Python paths
 * (prefix_search.py:247-385; flavor selects the arithmetic of the forward rows).  env_h == NULL: the dense search above. */
int po_pair_prefix_search_env_batch_h(const double* y1_h, const int64_t* y1_off_h, const double* y2_h, const int64_t* y2_off_h,
                                      const int32_t* env_h, const int64_t* env_off_h, int n, int C, const char* alphabet,
                                      int flavor, char* seq_h, const int64_t* seq_off_h, int32_t* seq_len_h, double* logp_h,
                                      int32_t* status_h);
/* decoding_cy.forward_vec_log (decoding_cy.pyx:127-156; flavor 1) / prefix_search.forward_vec_log
 * (prefix_search.py:81-96; flavor 0): one row of the CTC forward matrix for symbol s (-1: blank) and label
 * length i, for every item; previous_h (same layout as out_h: one double per frame, items back to back) is the row
 * of the label without its last symbol and may be NULL for i == 0. */
int po_forward_vec_batch_h(const double* y_h, const int64_t* y_off_h, int n, int C, int s, int i, int flavor,
                           const double* previous_h, double* out_h);
int po_pair_gamma_batch_h(const double* y1_h, const int64_t* y1_off_h, const double* y2_h, const int64_t* y2_off_h,
                          const int32_t* env_h, const int64_t* env_off_h, int n, int C, int flavor, double* gamma00_h,
                          double* dense_out_h, const int64_t* dense_off_h, int32_t* status_h);
int po_beam2d_batch_h(const double* y1_h, const int64_t* y1_off_h, const double* y2_h,
                      const int64_t* y2_off_h, const int32_t* env_h, int n, int C, const char* alphabet,
                      int beam_width, int model, int method, char* seq_h, const int64_t* seq_off_h, int32_t* seq_len_h,
                      int32_t* status_h);
/* pair decode with the 1-D stage supplied by the caller (pair-decode --single beam, pair_decode.py:363-370:
 * cpp_beam_search + cpp_viterbi_acceptor): seq1d / len1 / len2 and the frame maps are INPUTS.
 * map1_h / map2_h: int32, frame index of every base, read i at map + y*_off[i]. */
int po_pair_decode_from_1d_batch_h(const double* y1_h, const int64_t* y1_off_h, const double* y2_h,
                                   const int64_t* y2_off_h, int n, int C, const po_pair_options* opt,
                                   const char* seq1d_h, const int64_t* seq1d_off_h, const int32_t* len1_h,
                                   const int32_t* len2_h, const int32_t* map1_h, const int32_t* map2_h,
                                   double* identity_h, int32_t* env_out_h, char* seq_h, const int64_t* seq_off_h,
                                   int32_t* seq_len_h, int32_t* status_h);
int po_pair_decode_batch_h(const double* y1_h, const int64_t* y1_off_h, const double* y2_h,
                           const int64_t* y2_off_h, int n, int C, const po_pair_options* opt, char* seq1d_h,
                           const int64_t* seq1d_off_h, int32_t* len1_h, int32_t* len2_h, double* identity_h,
                           int32_t* env_out_h, char* seq_h, const int64_t* seq_off_h, int32_t* seq_len_h,
                           int32_t* status_h);

/* ---- host-to-strings pipeline of the pair decoder ---------------------------------------------------
 * replaces the reference's fan-out of pair_decode_helper over worker processes (pair_decode.py:292-297)
 * together with the trace loading each worker does: decode.load_logits / logit_to_log_likelihood
 * (decode.py:34-51), the Bonito column order (decode.py:79), the uint8 trace scaling (decode.py:89-93) and
 * transducer.reverse_complement of read 2 (transducer.py:68-70,104-106; pair_decode.py:323-329).
 * Inputs are HOST arrays as the basecaller wrote them: y1_h[i] / y2_h[i] point to C-contiguous (rows1[i], C) /
 * (rows2[i], C) matrices of float32 logits (in_mode PO_INGEST_LOGITS_F32), uint8 traces (PO_INGEST_TRACE_U8) or
 * float64 log-probabilities (PO_INGEST_F64); perm1 / perm2 (C ints or NULL): column order of read 1 / read 2
 * (out[:, c] = in[:, perm[c]]), reverse2: read 2 is time-reversed.  The pairs are decoded in waves of at most
 * wave_pairs pairs / wave_rows frames through three slots {stream, pinned staging, device buffers, workspace}:
 * waves k + 1 and k + 2 are packed and uploaded while wave k decodes, device memory is bounded by three waves whatever
 * n is, and nothing is allocated per call once the buffers have grown to the wave size.  wave_pairs 0: a job of at
 * most 4 096 pairs is one wave, a larger one is cut into even waves of about 2 500 pairs.
 * Outputs as po_pair_decode_batch_h (all host); env_out_h may be NULL (rows of pair i at 2 * sum(rows1[:i])).
 * A pipeline belongs to one host thread and one device.  */
typedef struct po_pipeline po_pipeline;
po_pipeline* po_pipeline_create(int device, int wave_pairs, int64_t wave_rows, int host_threads); /* 0 = defaults */
void po_pipeline_destroy(po_pipeline* p);
int po_pipeline_pair_decode(po_pipeline* p, const void* const* y1_h, const int64_t* rows1, const void* const* y2_h,
                            const int64_t* rows2, int n, int C, int in_mode, const int* perm1, const int* perm2,
                            int reverse2, const po_pair_options* opt, char* seq1d_h, const int64_t* seq1d_off_h,
                            int32_t* len1_h, int32_t* len2_h, double* identity_h, int32_t* env_out_h, char* seq_h,
                            const int64_t* seq_off_h, int32_t* seq_len_h, int32_t* status_h);
/* host milliseconds of the last call: packing, waiting for the device, whole call; number of waves */
int po_pipeline_stats(po_pipeline* p, double* pack_ms, double* wait_ms, double* total_ms, int* waves);

/* ---- several devices, one process ------------------------------------------------------------
 * Replaces the reference's fan-out over worker processes on a multi-GPU node (pair_decode.py:292-297: a
 * multiprocessing.Pool over pair_decode_helper): one pipeline per entry of `devices` (an index may repeat), each driven
 * by its own host thread inside the call, all taking waves of the batch from one planner — uneven pairs balance
 * themselves — and writing their results straight into the caller's arrays at the pairs' own indices (input order; no
 * gather, no copy between processes, no collective).  Arguments and results as po_pipeline_pair_decode; synchronous.
 * po_multi_stats: pairs decoded and pack / wait / total milliseconds of pipeline i in the last call. */
typedef struct po_multi po_multi;
po_multi* po_multi_create(const int* devices, int ndev, int wave_pairs, int64_t wave_rows, int host_threads);
void po_multi_destroy(po_multi* m);
int po_multi_devices(po_multi* m);
int po_multi_pair_decode(po_multi* m, const void* const* y1_h, const int64_t* rows1, const void* const* y2_h,
                         const int64_t* rows2, int n, int C, int in_mode, const int* perm1, const int* perm2,
                         int reverse2, const po_pair_options* opt, char* seq1d_h, const int64_t* seq1d_off_h,
                         int32_t* len1_h, int32_t* len2_h, double* identity_h, int32_t* env_out_h, char* seq_h,
                         const int64_t* seq_off_h, int32_t* seq_len_h, int32_t* status_h);
int po_multi_stats(po_multi* m, int i, int* pairs, double* pack_ms, double* wait_ms, double* total_ms, int* waves);
/* the waves such a call is cut into, in the order they are handed out (touches no device): first[k], count[k] for
 * k < returned number of waves (at most cap are written); wave_pairs / wave_rows 0 = the pipeline's defaults */
int po_wave_plan(const int64_t* rows1, const int64_t* rows2, int n, int wave_pairs, int64_t wave_rows, int ndev, int* first,
                 int* count, int cap);

/* ---- timing aid for bench.py: HIP events on the stream the kernels run on ----------------- */
void* po_event_create(void);
int po_event_record(void* ev, void* stream);
int po_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on `stop` */
void po_event_destroy(void* ev);
/* accumulated device time (ms) and launch count of the named kernel family since the last
 * reset, measured with HIP events around each launch when profiling is enabled */
#define PO_K_VITERBI 0
#define PO_K_BEAM1D 1
#define PO_K_BEAM2D 2
#define PO_K_ALIGN 3
#define PO_K_ENVELOPE 4
#define PO_K_BEAM2D_MAIN 5 /* the pair beam search kernel alone (PO_K_BEAM2D = the stage: + pre-pass, walk, queue reset) */
#define PO_K_COUNT 6
void po_profile_enable(int on);
void po_profile_reset(void);
int po_profile_get(int kernel, double* total_ms, int64_t* launches);
/* Compute-side pricing of the pair beam search (its roofline is the f64 logaddexp stream, not HBM):
 * po_profile_update_counter - two device counters (uint64[2], or NULL to stop) the pair beam kernels add
 *   update_prob evaluations to (PrefixTree.h:478-704; ctc: one logaddexp each, merge-repeats two, flip-flop
 *   three or four): [0] those the reference's schedule makes for the same input (every element over its
 *   full windows in every step, every catch-up step), [1] those the kernels executed (they leave out the
 *   ones whose result is provably already stored);
 * po_lae_peak - measured peak rate of the engine's logaddexp on this device (micro-benchmark, all lanes
 *   busy, 4 independent chains per lane). */
int po_profile_update_counter(uint64_t* device_counter);
int po_lae_peak(int iters, double* lae_per_s, void* stream);

#ifdef __cplusplus
}
#endif
#endif
