// C-ABI of libporeover_hip.so (include/poreover_hip.h): argument checks, workspace carving,
// kernel launches, host-buffer conveniences and the HIP-event profiling aid used by bench.py.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/poreover_hip.h"

extern "C" {
int po_launch_viterbi(const double*, const int64_t*, int, int, int, uint32_t, int, int8_t*, char*, const int64_t*, int32_t*,
                      int32_t*, int32_t*, int8_t*, int8_t*, hipStream_t);
int po_launch_beam1d(const double*, const int64_t*, int, int, int, uint32_t, int, int, int*, int*, char*, const int64_t*,
                     int32_t*, int32_t*, hipStream_t);
int64_t po_beam1d_arena_nodes(int, int64_t, int);
size_t po_beam2d_ws_bytes_impl(int, int64_t, int64_t, int64_t, int64_t, int, int, int, int);
int po_launch_beam2d(const double*, const int64_t*, const double*, const int64_t*, const int32_t*, int, int,
                     int, uint32_t, int, int, int, char*, const int64_t*, int32_t*, int32_t*, void*, size_t, hipStream_t);
size_t po_pair_ws_bytes_impl(int, int64_t, int64_t, int64_t, int64_t, int, const po_pair_options*);
size_t po_lattice_ws_bytes(int, int64_t, int64_t, int, int);
size_t po_prefix_ws_bytes(int, int64_t);
size_t po_gamma_ws_bytes(int, int64_t, int64_t, int64_t);
size_t po_pair_prefix_ws_bytes(int, int64_t);
int po_launch_forward_vec(const double*, const int64_t*, int, int, int, int, int, const double*, double*, hipStream_t);
int po_launch_pair_prefix_search(const double*, const int64_t*, const double*, const int64_t*, const double*, const int64_t*, int, int,
                                 int, uint32_t, int, int64_t, char*, const int64_t*, int32_t*, double*, int32_t*, void*, size_t,
                                 hipStream_t);
void po_b2_set_update_counter(unsigned long long*);
void po_b2_set_mark(void (*)(int, hipStream_t));
int po_launch_lae_peak(int, double*, hipStream_t);
int po_launch_pair_decode_from_1d(const double*, const int64_t*, const double*, const int64_t*, int, int, const po_pair_options*,
                                  int64_t, int64_t, int64_t, int64_t, const int32_t*, const int32_t*, char*, const int64_t*,
                                  int32_t*, int32_t*, double*, int32_t*, char*, const int64_t*, int32_t*, int32_t*, void*, size_t,
                                  hipStream_t);
int po_launch_gamma(const double*, const int64_t*, const double*, const int64_t*, const int32_t*, const int64_t*, int, int, int,
                    int64_t, int64_t, int64_t, double*, double*, const int64_t*, int32_t*, void*, size_t, hipStream_t);
int po_launch_ingest(const void*, const int64_t*, int, int, int, const int*, int, int64_t, double*, hipStream_t);
size_t po_align_ws_bytes(int, int64_t, int64_t, int);
int po_launch_align(const char*, const int64_t*, int, int, int64_t, int64_t, char*, char*, const int64_t*, int32_t*, int32_t*,
                    void*, size_t, hipStream_t);
int po_launch_nw_matrix(const char*, const int64_t*, int, int, int, int, int32_t*, const int64_t*, int32_t*, hipStream_t);
int po_launch_align_scores(const char*, const int64_t*, int, int, int, int, int, int64_t, int64_t, char*, char*, const int64_t*,
                           int32_t*, int32_t*, void*, size_t, hipStream_t);
size_t po_envelope_ws_bytes(int, int64_t);
int po_launch_envelope(const char*, const char*, const int64_t*, const int32_t*, int, const int32_t*, const int64_t*,
                       const int32_t*, const int64_t*, const int32_t*, const int32_t*, int, int64_t, int32_t*,
                       const int64_t*, int32_t*, void*, size_t, hipStream_t);
int po_launch_prefix_search(const double*, const int64_t*, int, int, int, uint32_t, int64_t, char*, const int64_t*, int32_t*,
                            double*, int32_t*, void*, size_t, hipStream_t);
int po_launch_forward(const double*, const int64_t*, int, int, int, uint32_t, int, const char*, const int64_t*, int64_t,
                      double*, int32_t*, void*, size_t, hipStream_t);
int po_launch_acceptor(const double*, const int64_t*, int, int, int, uint32_t, int, const char*, const int64_t*, int64_t,
                       int64_t, int32_t*, int32_t*, void*, size_t, hipStream_t);
int po_launch_pair_decode(const double*, const int64_t*, const double*, const int64_t*, int, int,
                          const po_pair_options*, char*, const int64_t*, int32_t*, int32_t*, double*, int32_t*,
                          char*, const int64_t*, int32_t*, int32_t*, void*, size_t, hipStream_t);
}

namespace {
thread_local std::string g_err;
int fail_hip(hipError_t e, const char* what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return PO_E_HIP;
}
#define HIPCHK(x)                                      \
    do {                                               \
        hipError_t e_ = (x);                           \
        if (e_ != hipSuccess) return fail_hip(e_, #x); \
    } while (0)
inline size_t al256(size_t b) { return (b + 255) & ~size_t(255); }
// alphabet string -> (A, packed bytes); NULL means "ACGT"
inline int pack_alphabet(const char* a, uint32_t* packed) {
    if (!a) a = "ACGT";
    const size_t n = std::strlen(a);
    if (n < 1 || n > 4) return -1;
    uint32_t p = 0;
    for (size_t i = 0; i < n; ++i) p |= (uint32_t)(unsigned char)a[i] << (8 * i);
    *packed = p;
    return (int)n;
}

// ---- per-launch event timing (bench.py's roofline leg) ----
struct ProfRec { hipEvent_t a, b; int kernel; };
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
double g_prof_ms[PO_K_COUNT];
int64_t g_prof_n[PO_K_COUNT];

struct ProfScope {
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t s;
    int k;
    bool on;
    ProfScope(int kernel, hipStream_t stream) : s(stream), k(kernel), on(g_prof_on) {
        if (on) {
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            (void)hipEventRecord(a, s);
        }
    }
    ~ProfScope() {
        if (on) {
            (void)hipEventRecord(b, s);
            std::lock_guard<std::mutex> lk(g_prof_mu);
            g_prof.push_back({a, b, k});
        }
    }
};
// brackets the main pair beam kernel (called from po_beam2d.hip around its launch)
hipEvent_t g_mark_a = nullptr;
void b2_mark(int begin, hipStream_t s) {
    if (!g_prof_on) return;
    if (begin) {
        (void)hipEventCreate(&g_mark_a);
        (void)hipEventRecord(g_mark_a, s);
    } else if (g_mark_a) {
        hipEvent_t b = nullptr;
        (void)hipEventCreate(&b);
        (void)hipEventRecord(b, s);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back({g_mark_a, b, PO_K_BEAM2D_MAIN});
        g_mark_a = nullptr;
    }
}
void prof_drain() {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) {
        float ms = 0;
        (void)hipEventSynchronize(r.b);
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { g_prof_ms[r.kernel] += ms; g_prof_n[r.kernel]++; }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
}
}  // namespace

extern "C" {

int po_version(void) { return 100; }

int po_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int po_set_device(int device) {
    g_err.clear();
    HIPCHK(hipSetDevice(device));
    return PO_OK;
}

const char* po_last_error(void) { return g_err.c_str(); }
// internal: lets the other translation units (po_stream.hip) leave a message for po_last_error
void po_set_error(const char* msg) { g_err = msg ? msg : ""; }

int po_device_info(int device, char* name, int name_cap, int* cus, int* clock_khz, size_t* total_mem) {
    g_err.clear();
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, device));
    if (name && name_cap > 0) { std::strncpy(name, p.name, name_cap - 1); name[name_cap - 1] = 0; }
    if (cus) *cus = p.multiProcessorCount;
    if (clock_khz) *clock_khz = p.clockRate;
    if (total_mem) *total_mem = p.totalGlobalMem;
    return PO_OK;
}

// -------------------------------------------------------------------------------- ingest
int po_ingest_batch(const void* src, const int64_t* row_off, int n, int C, int mode, const int* perm_h, int reverse,
                    double* out, void* stream) {
    g_err.clear();
    if (n < 0 || !src || !row_off || !out) { g_err = "po_ingest_batch: null argument"; return PO_E_ARG; }
    if (n == 0) return PO_OK;
    int64_t ends[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(&ends[0], row_off, sizeof(int64_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipMemcpyAsync(&ends[1], row_off + n, sizeof(int64_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (ends[0] != 0) { g_err = "po_ingest_batch: row_off[0] must be 0"; return PO_E_ARG; }
    int rc = po_launch_ingest(src, row_off, n, C, mode, perm_h, reverse, ends[1], out, (hipStream_t)stream);
    if (rc != PO_OK) { g_err = "po_ingest_batch: bad C / mode / permutation"; return rc; }
    HIPCHK(hipGetLastError());
    return PO_OK;
}

// -------------------------------------------------------------------------------- viterbi
size_t po_viterbi_workspace_bytes(int n, int64_t total_rows, int C, int kind) {
    g_err.clear();
    if (kind != PO_KIND_FLIPFLOP) return 256;
    return al256((size_t)total_rows * 8) + al256((size_t)total_rows) + 256;  // ptr[T][8] + path[T]
}

int po_viterbi_batch(const double* y, const int64_t* y_off, int n, int C, const char* alphabet, int kind,
                     int8_t* path, char* seq,
                     const int64_t* seq_off, int32_t* seq_len, int32_t* map, int32_t* status, void* ws,
                     size_t ws_bytes, void* stream) {
    g_err.clear();
    if (n < 0 || !y || !y_off || !seq || !seq_off || !seq_len || !status) { g_err = "po_viterbi_batch: null argument"; return PO_E_ARG; }
    uint32_t ap = 0;
    const int A = pack_alphabet(alphabet, &ap);
    if (A < 0) { g_err = "po_viterbi_batch: alphabet must have 1..4 symbols"; return PO_E_ARG; }
    int8_t *ff_ptr = nullptr, *ff_path = nullptr;
    if (kind == PO_KIND_FLIPFLOP) {
        if (!ws) { g_err = "po_viterbi_batch: flip-flop needs a workspace"; return PO_E_CAP; }
        int64_t ends[2] = {0, 0};  // total rows, to split and bounds-check the workspace
        HIPCHK(hipMemcpyAsync(&ends[0], y_off, sizeof(int64_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
        HIPCHK(hipMemcpyAsync(&ends[1], y_off + n, sizeof(int64_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
        HIPCHK(hipStreamSynchronize((hipStream_t)stream));
        const size_t rows = (size_t)(ends[1] - ends[0]);
        if (ws_bytes < al256(rows * 8) + al256(rows)) { g_err = "po_viterbi_batch: workspace too small"; return PO_E_CAP; }
        ff_ptr = (int8_t*)ws;
        ff_path = ff_ptr + al256(rows * 8);
    }
    ProfScope ps(PO_K_VITERBI, (hipStream_t)stream);
    int rc = po_launch_viterbi(y, y_off, n, C, A, ap, kind, path, seq, seq_off, seq_len, map, status, ff_ptr, ff_path,
                               (hipStream_t)stream);
    if (rc != PO_OK) { g_err = "po_viterbi_batch: unsupported C/kind"; return rc; }
    HIPCHK(hipGetLastError());
    return PO_OK;
}

// -------------------------------------------------------------------------------- beam 1-D
size_t po_beam1d_workspace_bytes(int n, int64_t total_rows, int64_t max_rows, int C, int W, int model) {
    g_err.clear();
    (void)max_rows; (void)C; (void)model;
    return 2 * al256(sizeof(int) * (size_t)po_beam1d_arena_nodes(n, total_rows, W)) + 256;
}

// The arena is sized from the rows the CALLER declared when sizing the workspace; the kernel
// bounds-checks every allocation against its own per-read share, so a short workspace yields
// PO_E_CAP here rather than a stray write.
int po_beam1d_batch(const double* y, const int64_t* y_off, int n, int C, const char* alphabet, int W, int model,
                    char* seq,
                    const int64_t* seq_off, int32_t* seq_len, int32_t* status, void* ws, size_t ws_bytes,
                    void* stream) {
    g_err.clear();
    if (n < 0 || !y || !y_off || !seq || !seq_off || !seq_len || !status || !ws) { g_err = "po_beam1d_batch: null argument"; return PO_E_ARG; }
    if (ws_bytes < 512) { g_err = "po_beam1d_batch: workspace too small"; return PO_E_CAP; }
    uint32_t ap = 0;
    const int A = pack_alphabet(alphabet, &ap);
    if (A < 0) { g_err = "po_beam1d_batch: alphabet must have 1..4 symbols"; return PO_E_ARG; }
    const size_t half = ((ws_bytes - 256) / 2) / 256 * 256;
    int* apl = (int*)ws;
    int* afc = (int*)((char*)ws + half);
    // total rows, for the capacity check (one small D2H; the rest of the call stays asynchronous)
    int64_t ends[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(&ends[0], y_off, sizeof(int64_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipMemcpyAsync(&ends[1], y_off + n, sizeof(int64_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (sizeof(int) * (size_t)po_beam1d_arena_nodes(n, ends[1] - ends[0], W) > half) { g_err = "po_beam1d_batch: workspace too small"; return PO_E_CAP; }
    ProfScope ps(PO_K_BEAM1D, (hipStream_t)stream);
    int rc = po_launch_beam1d(y, y_off, n, C, A, ap, W, model, apl, afc, seq, seq_off, seq_len, status, (hipStream_t)stream);
    if (rc != PO_OK) { g_err = "po_beam1d_batch: unsupported C/model/beam_width"; return rc; }
    HIPCHK(hipGetLastError());
    return PO_OK;
}

// -------------------------------------------------------------------------------- beam 2-D
size_t po_beam2d_workspace_bytes(int n, int64_t tr1, int64_t tr2, int64_t mr1, int64_t mr2, int C, int W,
                                 int model, int method) {
    g_err.clear();
    return po_beam2d_ws_bytes_impl(n, tr1, tr2, mr1, mr2, C, W, model, method);
}

int po_beam2d_batch(const double* y1, const int64_t* y1_off, const double* y2, const int64_t* y2_off,
                    const int32_t* env, int n, int C, const char* alphabet, int W, int model, int method,
                    char* seq, const int64_t* seq_off, int32_t* seq_len, int32_t* status, void* ws,
                    size_t ws_bytes, void* stream) {
    g_err.clear();
    if (n < 0 || !y1 || !y1_off || !y2 || !y2_off || !seq || !seq_off || !seq_len || !status || !ws) { g_err = "po_beam2d_batch: null argument"; return PO_E_ARG; }
    uint32_t ap = 0;
    const int A = pack_alphabet(alphabet, &ap);
    if (A < 0) { g_err = "po_beam2d_batch: alphabet must have 1..4 symbols"; return PO_E_ARG; }
    ProfScope ps(PO_K_BEAM2D, (hipStream_t)stream);
    int rc = po_launch_beam2d(y1, y1_off, y2, y2_off, env, n, C, A, ap, W, model, method, seq, seq_off, seq_len, status,
                              ws, ws_bytes, (hipStream_t)stream);
    if (rc != PO_OK) { g_err = "po_beam2d_batch: launch refused"; return rc; }
    HIPCHK(hipGetLastError());
    return PO_OK;
}

// -------------------------------------------------------------------------------- forward / acceptor
namespace {
// max rows / max label length of a batch, read back from the device offset tables
int batch_maxima(const int64_t* y_off, const int64_t* label_off, int n, hipStream_t s, int64_t* mr, int64_t* ml) {
    std::vector<int64_t> h(2 * (size_t)(n + 1));
    HIPCHK(hipMemcpyAsync(h.data(), y_off, sizeof(int64_t) * (n + 1), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h.data() + n + 1, label_off, sizeof(int64_t) * (n + 1), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    *mr = 0; *ml = 0;
    for (int i = 0; i < n; ++i) {
        *mr = std::max<int64_t>(*mr, h[i + 1] - h[i]);
        *ml = std::max<int64_t>(*ml, h[n + 1 + i + 1] - h[n + 1 + i]);
    }
    return PO_OK;
}
}  // namespace

size_t po_forward_workspace_bytes(int n, int64_t max_rows, int model) {
    g_err.clear(); return po_lattice_ws_bytes(n, max_rows, 0, model, 0); }

int po_forward_batch(const double* y, const int64_t* y_off, int n, int C, const char* alphabet, int model,
                     const char* labels, const int64_t* label_off, double* logp, int32_t* status, void* ws,
                     size_t ws_bytes, void* stream) {
    g_err.clear();
    if (n < 0 || !y || !y_off || !labels || !label_off || !logp || !status || !ws) { g_err = "po_forward_batch: null argument"; return PO_E_ARG; }
    uint32_t ap = 0;
    const int A = pack_alphabet(alphabet, &ap);
    if (A < 0) { g_err = "po_forward_batch: alphabet must have 1..4 symbols"; return PO_E_ARG; }
    if (n == 0) return PO_OK;
    int64_t mr = 0, ml = 0;
    int rc = batch_maxima(y_off, label_off, n, (hipStream_t)stream, &mr, &ml);
    if (rc != PO_OK) return rc;
    rc = po_launch_forward(y, y_off, n, C, A, ap, model, labels, label_off, mr, logp, status, ws, ws_bytes, (hipStream_t)stream);
    if (rc != PO_OK) { g_err = "po_forward_batch: unsupported C/model or workspace too small"; return rc; }
    HIPCHK(hipGetLastError());
    return PO_OK;
}

size_t po_viterbi_acceptor_workspace_bytes(int n, int64_t max_rows, int64_t max_label) {
    g_err.clear();
    return po_lattice_ws_bytes(n, max_rows, max_label, PO_MODEL_CTC, 1);
}

int po_viterbi_acceptor_batch(const double* y, const int64_t* y_off, int n, int C, const char* alphabet, int band_size,
                              const char* labels, const int64_t* label_off, int32_t* path, int32_t* status, void* ws,
                              size_t ws_bytes, void* stream) {
    g_err.clear();
    if (n < 0 || !y || !y_off || !labels || !label_off || !path || !status || !ws) { g_err = "po_viterbi_acceptor_batch: null argument"; return PO_E_ARG; }
    uint32_t ap = 0;
    const int A = pack_alphabet(alphabet, &ap);
    if (A < 0) { g_err = "po_viterbi_acceptor_batch: alphabet must have 1..4 symbols"; return PO_E_ARG; }
    if (n == 0) return PO_OK;
    int64_t mr = 0, ml = 0;
    int rc = batch_maxima(y_off, label_off, n, (hipStream_t)stream, &mr, &ml);
    if (rc != PO_OK) return rc;
    rc = po_launch_acceptor(y, y_off, n, C, A, ap, band_size, labels, label_off, mr, ml, path, status, ws, ws_bytes,
                            (hipStream_t)stream);
    if (rc != PO_OK) { g_err = "po_viterbi_acceptor_batch: unsupported C or workspace too small"; return rc; }
    HIPCHK(hipGetLastError());
    return PO_OK;
}

// -------------------------------------------------------------------------------- prefix search
size_t po_prefix_search_workspace_bytes(int n, int64_t max_rows) {
    g_err.clear(); return po_prefix_ws_bytes(n, max_rows); }

int po_prefix_search_batch(const double* y, const int64_t* y_off, int n, int C, const char* alphabet, char* seq,
                           const int64_t* seq_off, int32_t* seq_len, double* logp, int32_t* status, void* ws,
                           size_t ws_bytes, void* stream) {
    g_err.clear();
    if (n < 0 || !y || !y_off || !seq || !seq_off || !seq_len || !logp || !status || !ws) { g_err = "po_prefix_search_batch: null argument"; return PO_E_ARG; }
    uint32_t ap = 0;
    const int A = pack_alphabet(alphabet, &ap);
    if (A < 0) { g_err = "po_prefix_search_batch: alphabet must have 1..4 symbols"; return PO_E_ARG; }
    if (n == 0) return PO_OK;
    std::vector<int64_t> h((size_t)n + 1);
    HIPCHK(hipMemcpyAsync(h.data(), y_off, sizeof(int64_t) * (n + 1), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    int64_t mr = 0;
    for (int i = 0; i < n; ++i) mr = std::max<int64_t>(mr, h[i + 1] - h[i]);
    int rc = po_launch_prefix_search(y, y_off, n, C, A, ap, mr, seq, seq_off, seq_len, logp, status, ws, ws_bytes,
                                     (hipStream_t)stream);
    if (rc != PO_OK) { g_err = "po_prefix_search_batch: unsupported C / window longer than the LDS rows / workspace too small"; return rc; }
    HIPCHK(hipGetLastError());
    return PO_OK;
}

// -------------------------------------------------------------------------------- align / envelope
size_t po_align_workspace_bytes(int n, int64_t ml1, int64_t ml2, int band) {
    g_err.clear(); return po_align_ws_bytes(n, ml1, ml2, band); }

int po_align_batch(const char* seqs, const int64_t* seq_off, int n, int band_width, char* aln1, char* aln2,
                   const int64_t* aln_off, int32_t* ncol, int32_t* status, void* ws, size_t ws_bytes, void* stream) {
    g_err.clear();
    if (n < 0 || !seqs || !seq_off || !aln1 || !aln2 || !aln_off || !ncol || !status || !ws) { g_err = "po_align_batch: null argument"; return PO_E_ARG; }
    if (n == 0) return PO_OK;
    std::vector<int64_t> h(2 * (size_t)n + 1);
    HIPCHK(hipMemcpyAsync(h.data(), seq_off, sizeof(int64_t) * (2 * n + 1), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    int64_t m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) { m1 = std::max<int64_t>(m1, h[2 * i + 1] - h[2 * i]); m2 = std::max<int64_t>(m2, h[2 * i + 2] - h[2 * i + 1]); }
    int rc = po_launch_align(seqs, seq_off, n, band_width, m1, m2, aln1, aln2, aln_off, ncol, status, ws, ws_bytes, (hipStream_t)stream);
    if (rc != PO_OK) { g_err = "po_align_batch: workspace too small"; return rc; }
    HIPCHK(hipGetLastError());
    return PO_OK;
}

size_t po_envelope_workspace_bytes(int n, int64_t max_ncol) {
    g_err.clear(); return po_envelope_ws_bytes(n, max_ncol); }

int po_envelope_batch(const char* aln1, const char* aln2, const int64_t* aln_off, const int32_t* ncol, int n,
                      const int32_t* map1, const int64_t* map1_off, const int32_t* map2, const int64_t* map2_off,
                      const int32_t* U, const int32_t* V, int padding, int32_t* env, const int64_t* env_off,
                      int32_t* status, void* ws, size_t ws_bytes, void* stream) {
    g_err.clear();
    if (n < 0 || !aln1 || !aln2 || !aln_off || !ncol || !map1 || !map1_off || !map2 || !map2_off || !U || !V || !env ||
        !env_off || !status || !ws) { g_err = "po_envelope_batch: null argument"; return PO_E_ARG; }
    if (n == 0) return PO_OK;
    std::vector<int32_t> h((size_t)n);
    HIPCHK(hipMemcpyAsync(h.data(), ncol, sizeof(int32_t) * n, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    int64_t mc = 0;
    for (int i = 0; i < n; ++i) mc = std::max<int64_t>(mc, h[i]);
    int rc = po_launch_envelope(aln1, aln2, aln_off, ncol, n, map1, map1_off, map2, map2_off, U, V, padding, mc, env,
                                env_off, status, ws, ws_bytes, (hipStream_t)stream);
    if (rc != PO_OK) { g_err = "po_envelope_batch: workspace too small"; return rc; }
    HIPCHK(hipGetLastError());
    return PO_OK;
}

// -------------------------------------------------------------------------------- pair gamma
size_t po_pair_gamma_workspace_bytes(int n, int64_t max_cells, int64_t mr1, int64_t mr2) {
    g_err.clear(); return po_gamma_ws_bytes(n, max_cells, mr1, mr2); }

int po_pair_gamma_batch(const double* y1, const int64_t* y1_off, const double* y2, const int64_t* y2_off,
                        const int32_t* env, const int64_t* env_off, int n, int C, int flavor, int64_t max_cells,
                        double* gamma00, double* dense_out, const int64_t* dense_off, int32_t* status, void* ws,
                        size_t ws_bytes, void* stream) {
    g_err.clear();
    if (n < 0 || !y1 || !y1_off || !y2 || !y2_off || !gamma00 || !status || !ws || (env && !env_off) ||
        (dense_out && !dense_off)) { g_err = "po_pair_gamma_batch: null argument"; return PO_E_ARG; }
    if (n == 0) return PO_OK;
    std::vector<int64_t> h(2 * (size_t)(n + 1));
    HIPCHK(hipMemcpyAsync(h.data(), y1_off, sizeof(int64_t) * (n + 1), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipMemcpyAsync(h.data() + n + 1, y2_off, sizeof(int64_t) * (n + 1), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    int64_t m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) { m1 = std::max<int64_t>(m1, h[i + 1] - h[i]); m2 = std::max<int64_t>(m2, h[n + 1 + i + 1] - h[n + 1 + i]); }
    int rc = po_launch_gamma(y1, y1_off, y2, y2_off, env, env_off, n, C, flavor, max_cells, m1, m2, gamma00, dense_out,
                             dense_off, status, ws, ws_bytes, (hipStream_t)stream);
    if (rc != PO_OK) { g_err = "po_pair_gamma_batch: bad C or workspace too small"; return rc; }
    HIPCHK(hipGetLastError());
    return PO_OK;
}

// -------------------------------------------------------------------------------- pair decode
size_t po_pair_decode_workspace_bytes(int n, int64_t tr1, int64_t tr2, int64_t mr1, int64_t mr2, int C,
                                      const po_pair_options* opt) {
    g_err.clear();
    return po_pair_ws_bytes_impl(n, tr1, tr2, mr1, mr2, C, opt);
}

int po_pair_decode_batch(const double* y1, const int64_t* y1_off, const double* y2, const int64_t* y2_off, int n,
                         int C, const po_pair_options* opt, char* seq1d, const int64_t* seq1d_off, int32_t* len1,
                         int32_t* len2, double* identity, int32_t* env_out, char* seq, const int64_t* seq_off,
                         int32_t* seq_len, int32_t* status, void* ws, size_t ws_bytes, void* stream) {
    g_err.clear();
    if (n < 0 || !y1 || !y1_off || !y2 || !y2_off || !opt || !seq1d || !seq1d_off || !len1 || !len2 || !identity ||
        !seq || !seq_off || !seq_len || !status || !ws) { g_err = "po_pair_decode_batch: null argument"; return PO_E_ARG; }
    int rc = po_launch_pair_decode(y1, y1_off, y2, y2_off, n, C, opt, seq1d, seq1d_off, len1, len2, identity,
                                   env_out, seq, seq_off, seq_len, status, ws, ws_bytes, (hipStream_t)stream);
    if (rc != PO_OK) { g_err = "po_pair_decode_batch: launch refused"; return rc; }
    HIPCHK(hipGetLastError());
    return PO_OK;
}

// -------------------------------------------------------------------------------- host-buffer forms
namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t b) { return hipMalloc(&p, b ? b : 16) == hipSuccess ? 0 : -1; }
};
#define UP(buf, src, bytes)                                                                      \
    do {                                                                                         \
        if ((buf).alloc(bytes)) { g_err = "hipMalloc failed"; return PO_E_HIP; }                 \
        if ((src) && (bytes)) HIPCHK(hipMemcpy((buf).p, (src), (bytes), hipMemcpyHostToDevice)); \
    } while (0)
#define DOWN(dst, buf, bytes)                                                                      \
    do {                                                                                           \
        if ((dst) && (bytes)) HIPCHK(hipMemcpy((dst), (buf).p, (bytes), hipMemcpyDeviceToHost));   \
    } while (0)
}  // namespace

int po_ingest_batch_h(const void* src_h, const int64_t* row_off_h, int n, int C, int mode, const int* perm_h, int reverse,
                      double* out_h) {
    g_err.clear();
    if (n <= 0) return PO_OK;
    const int64_t rows = row_off_h[n];
    const size_t esz = mode == PO_INGEST_LOGITS_F32 ? 4 : (mode == PO_INGEST_TRACE_U8 ? 1 : 8);
    DevBuf s, ro, o;
    UP(s, src_h, esz * (size_t)rows * C);
    UP(ro, row_off_h, sizeof(int64_t) * (n + 1));
    UP(o, nullptr, sizeof(double) * (size_t)rows * C);
    int rc = po_ingest_batch(s.p, (const int64_t*)ro.p, n, C, mode, perm_h, reverse, (double*)o.p, nullptr);
    if (rc != PO_OK) return rc;
    HIPCHK(hipDeviceSynchronize());
    DOWN(out_h, o, sizeof(double) * (size_t)rows * C);
    return PO_OK;
}

int po_viterbi_batch_h(const double* y_h, const int64_t* y_off_h, int n, int C, const char* alphabet, int kind,
                       int8_t* path_h,
                       char* seq_h, const int64_t* seq_off_h, int32_t* seq_len_h, int32_t* map_h,
                       int32_t* status_h) {
    g_err.clear();
    if (n <= 0) return PO_OK;
    const int64_t rows = y_off_h[n] - y_off_h[0];
    const int64_t seqb = seq_off_h[n];
    DevBuf y, yo, so, sq, sl, st, pt, mp, ws;
    UP(y, y_h + y_off_h[0] * C, sizeof(double) * rows * C);
    std::vector<int64_t> off(y_off_h, y_off_h + n + 1);
    for (auto& o : off) o -= y_off_h[0];
    UP(yo, off.data(), sizeof(int64_t) * (n + 1));
    UP(so, seq_off_h, sizeof(int64_t) * (n + 1));
    UP(sq, nullptr, (size_t)seqb);
    UP(sl, nullptr, sizeof(int32_t) * n);
    UP(st, nullptr, sizeof(int32_t) * n);
    UP(pt, nullptr, (size_t)rows);
    UP(mp, nullptr, sizeof(int32_t) * rows);
    const size_t wsb = po_viterbi_workspace_bytes(n, rows, C, kind);
    UP(ws, nullptr, wsb);
    int rc = po_viterbi_batch((const double*)y.p, (const int64_t*)yo.p, n, C, alphabet, kind, (int8_t*)pt.p, (char*)sq.p,
                              (const int64_t*)so.p, (int32_t*)sl.p, map_h ? (int32_t*)mp.p : nullptr,
                              (int32_t*)st.p, ws.p, wsb, nullptr);
    if (rc != PO_OK) return rc;
    HIPCHK(hipDeviceSynchronize());
    DOWN(seq_h, sq, (size_t)seqb);
    DOWN(seq_len_h, sl, sizeof(int32_t) * n);
    DOWN(status_h, st, sizeof(int32_t) * n);
    DOWN(path_h, pt, (size_t)rows);
    DOWN(map_h, mp, sizeof(int32_t) * rows);
    return PO_OK;
}

// decode driver in one call: raw basecaller output up, ingest on the device, Viterbi (beam_width <= 0) or 1-D beam
// search, strings down — `poreover decode` for a batch of files without a host-side log-softmax or a float64 upload
int po_decode_1d_batch_h(const void* src_h, const int64_t* row_off_h, int n, int C, int in_mode, const int* perm_h, int reverse,
                         const char* alphabet, int kind, int beam_width, int model, char* seq_h, const int64_t* seq_off_h,
                         int32_t* seq_len_h, int32_t* status_h) {
    g_err.clear();
    if (n <= 0) return PO_OK;
    if (!src_h || !row_off_h || !seq_h || !seq_off_h || !seq_len_h || !status_h) { g_err = "po_decode_1d_batch_h: null argument"; return PO_E_ARG; }
    if (in_mode < 0 || in_mode > 2 || row_off_h[0] != 0) { g_err = "po_decode_1d_batch_h: bad input mode / offsets"; return PO_E_ARG; }
    const int64_t rows = row_off_h[n];
    const size_t esz = in_mode == PO_INGEST_LOGITS_F32 ? 4 : (in_mode == PO_INGEST_TRACE_U8 ? 1 : 8);
    int64_t mx = 0;
    for (int i = 0; i < n; ++i) mx = std::max<int64_t>(mx, row_off_h[i + 1] - row_off_h[i]);
    DevBuf src, ro, y, so, sq, sl, st, ws;
    UP(src, src_h, esz * (size_t)rows * C);
    UP(ro, row_off_h, sizeof(int64_t) * (n + 1));
    UP(y, nullptr, sizeof(double) * (size_t)rows * C);
    UP(so, seq_off_h, sizeof(int64_t) * (n + 1));
    UP(sq, nullptr, (size_t)seq_off_h[n]);
    UP(sl, nullptr, sizeof(int32_t) * n);
    UP(st, nullptr, sizeof(int32_t) * n);
    int rc = po_launch_ingest(src.p, (const int64_t*)ro.p, n, C, in_mode, perm_h, reverse, rows, (double*)y.p, nullptr);
    if (rc != PO_OK) { g_err = "po_decode_1d_batch_h: bad C / mode / permutation"; return rc; }
    if (beam_width <= 0) {
        const size_t wsb = po_viterbi_workspace_bytes(n, rows, C, kind);
        UP(ws, nullptr, wsb);
        rc = po_viterbi_batch((const double*)y.p, (const int64_t*)ro.p, n, C, alphabet, kind, nullptr, (char*)sq.p,
                              (const int64_t*)so.p, (int32_t*)sl.p, nullptr, (int32_t*)st.p, ws.p, wsb, nullptr);
    } else {
        const size_t wsb = po_beam1d_workspace_bytes(n, rows, mx, C, beam_width, model);
        UP(ws, nullptr, wsb);
        rc = po_beam1d_batch((const double*)y.p, (const int64_t*)ro.p, n, C, alphabet, beam_width, model, (char*)sq.p,
                             (const int64_t*)so.p, (int32_t*)sl.p, (int32_t*)st.p, ws.p, wsb, nullptr);
    }
    if (rc != PO_OK) return rc;
    HIPCHK(hipDeviceSynchronize());
    DOWN(seq_h, sq, (size_t)seq_off_h[n]);
    DOWN(seq_len_h, sl, sizeof(int32_t) * n);
    DOWN(status_h, st, sizeof(int32_t) * n);
    return PO_OK;
}

int po_beam1d_batch_h(const double* y_h, const int64_t* y_off_h, int n, int C, const char* alphabet, int W,
                      int model, char* seq_h,
                      const int64_t* seq_off_h, int32_t* seq_len_h, int32_t* status_h) {
    g_err.clear();
    if (n <= 0) return PO_OK;
    const int64_t rows = y_off_h[n] - y_off_h[0];
    const int64_t seqb = seq_off_h[n];
    int64_t mx = 0;
    for (int i = 0; i < n; ++i) mx = std::max<int64_t>(mx, y_off_h[i + 1] - y_off_h[i]);
    DevBuf y, yo, so, sq, sl, st, ws;
    UP(y, y_h + y_off_h[0] * C, sizeof(double) * rows * C);
    std::vector<int64_t> off(y_off_h, y_off_h + n + 1);
    for (auto& o : off) o -= y_off_h[0];
    UP(yo, off.data(), sizeof(int64_t) * (n + 1));
    UP(so, seq_off_h, sizeof(int64_t) * (n + 1));
    UP(sq, nullptr, (size_t)seqb);
    UP(sl, nullptr, sizeof(int32_t) * n);
    UP(st, nullptr, sizeof(int32_t) * n);
    const size_t wsb = po_beam1d_workspace_bytes(n, rows, mx, C, W, model);
    UP(ws, nullptr, wsb);
    int rc = po_beam1d_batch((const double*)y.p, (const int64_t*)yo.p, n, C, alphabet, W, model, (char*)sq.p,
                             (const int64_t*)so.p, (int32_t*)sl.p, (int32_t*)st.p, ws.p, wsb, nullptr);
    if (rc != PO_OK) return rc;
    HIPCHK(hipDeviceSynchronize());
    DOWN(seq_h, sq, (size_t)seqb);
    DOWN(seq_len_h, sl, sizeof(int32_t) * n);
    DOWN(status_h, st, sizeof(int32_t) * n);
    return PO_OK;
}

int po_pair_gamma_batch_h(const double* y1_h, const int64_t* y1_off_h, const double* y2_h, const int64_t* y2_off_h,
                          const int32_t* env_h, const int64_t* env_off_h, int n, int C, int flavor, double* gamma00_h,
                          double* dense_out_h, const int64_t* dense_off_h, int32_t* status_h) {
    g_err.clear();
    if (n <= 0) return PO_OK;
    const int64_t r1 = y1_off_h[n] - y1_off_h[0], r2 = y2_off_h[n] - y2_off_h[0];
    int64_t m1 = 0, m2 = 0, mc = 0;
    for (int i = 0; i < n; ++i) {
        const int64_t U = y1_off_h[i + 1] - y1_off_h[i], V = y2_off_h[i + 1] - y2_off_h[i];
        m1 = std::max(m1, U); m2 = std::max(m2, V);
        int64_t cells = 0;
        if (env_h) for (int64_t u = 0; u <= U; ++u) { const int64_t w = env_h[2 * (env_off_h[i] + u) + 1] - env_h[2 * (env_off_h[i] + u)] + 1; cells += w > 0 ? w : 0; }
        else cells = (U + 1) * (V + 1);
        mc = std::max(mc, cells);
    }
    DevBuf a, ao, b, bo, ev, eo, g0, dn, dof, st, ws;
    UP(a, y1_h + y1_off_h[0] * C, sizeof(double) * r1 * C);
    UP(b, y2_h + y2_off_h[0] * C, sizeof(double) * r2 * C);
    std::vector<int64_t> o1(y1_off_h, y1_off_h + n + 1), o2(y2_off_h, y2_off_h + n + 1);
    for (auto& o : o1) o -= y1_off_h[0];
    for (auto& o : o2) o -= y2_off_h[0];
    UP(ao, o1.data(), sizeof(int64_t) * (n + 1));
    UP(bo, o2.data(), sizeof(int64_t) * (n + 1));
    if (env_h) { UP(ev, env_h, sizeof(int32_t) * 2 * (size_t)env_off_h[n]); UP(eo, env_off_h, sizeof(int64_t) * (n + 1)); }
    UP(g0, nullptr, sizeof(double) * n);
    if (dense_out_h) { UP(dn, nullptr, sizeof(double) * (size_t)dense_off_h[n]); UP(dof, dense_off_h, sizeof(int64_t) * (n + 1)); }
    UP(st, nullptr, sizeof(int32_t) * n);
    const size_t wsb = po_pair_gamma_workspace_bytes(n, mc, m1, m2);
    UP(ws, nullptr, wsb);
    int rc = po_pair_gamma_batch((const double*)a.p, (const int64_t*)ao.p, (const double*)b.p, (const int64_t*)bo.p,
                                 env_h ? (const int32_t*)ev.p : nullptr, env_h ? (const int64_t*)eo.p : nullptr, n, C, flavor,
                                 mc, (double*)g0.p, dense_out_h ? (double*)dn.p : nullptr,
                                 dense_out_h ? (const int64_t*)dof.p : nullptr, (int32_t*)st.p, ws.p, wsb, nullptr);
    if (rc != PO_OK) return rc;
    HIPCHK(hipDeviceSynchronize());
    DOWN(gamma00_h, g0, sizeof(double) * n);
    if (dense_out_h) DOWN(dense_out_h, dn, sizeof(double) * (size_t)dense_off_h[n]);
    DOWN(status_h, st, sizeof(int32_t) * n);
    return PO_OK;
}

int po_pair_prefix_search_env_batch_h(const double* y1_h, const int64_t* y1_off_h, const double* y2_h, const int64_t* y2_off_h,
                                      const int32_t* env_h, const int64_t* env_off_h, int n, int C, const char* alphabet,
                                      int flavor, char* seq_h, const int64_t* seq_off_h, int32_t* seq_len_h, double* logp_h,
                                      int32_t* status_h) {
    g_err.clear();
    if (n <= 0) return PO_OK;
    uint32_t ap = 0;
    const int A = pack_alphabet(alphabet, &ap);
    if (A < 0) { g_err = "po_pair_prefix_search_env_batch_h: alphabet must have 1..4 symbols"; return PO_E_ARG; }
    const int64_t r1 = y1_off_h[n] - y1_off_h[0], r2 = y2_off_h[n] - y2_off_h[0];
    int64_t m1 = 0, m2 = 0, mc = 0;
    std::vector<int64_t> dof((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) {
        const int64_t U = y1_off_h[i + 1] - y1_off_h[i], V = y2_off_h[i + 1] - y2_off_h[i];
        m1 = std::max(m1, U); m2 = std::max(m2, V);
        int64_t cells = (U + 1) * (V + 1);
        if (env_h) {   // stored cells of the envelope DP: sum over rows of end - start + 1
            cells = 0;
            for (int64_t u = 0; u <= U; ++u) { const int64_t w = (int64_t)env_h[2 * (env_off_h[i] + u) + 1] - env_h[2 * (env_off_h[i] + u)] + 1; cells += w > 0 ? w : 0; }
        }
        mc = std::max(mc, cells);
        dof[i + 1] = dof[i] + (U + 1) * (V + 1);
    }
    DevBuf a, ao, b, bo, g0, dn, dfo, st, st2, ws, so, sq, sl, lp, ws2, ev, eo;
    if (env_h) { UP(ev, env_h, sizeof(int32_t) * 2 * (size_t)env_off_h[n]); UP(eo, env_off_h, sizeof(int64_t) * (n + 1)); }
    UP(a, y1_h + y1_off_h[0] * C, sizeof(double) * r1 * C);
    UP(b, y2_h + y2_off_h[0] * C, sizeof(double) * r2 * C);
    std::vector<int64_t> o1(y1_off_h, y1_off_h + n + 1), o2(y2_off_h, y2_off_h + n + 1);
    for (auto& o : o1) o -= y1_off_h[0];
    for (auto& o : o2) o -= y2_off_h[0];
    UP(ao, o1.data(), sizeof(int64_t) * (n + 1));
    UP(bo, o2.data(), sizeof(int64_t) * (n + 1));
    UP(g0, nullptr, sizeof(double) * n);
    UP(dn, nullptr, sizeof(double) * (size_t)dof[n]);
    UP(dfo, dof.data(), sizeof(int64_t) * (n + 1));
    UP(st, nullptr, sizeof(int32_t) * n);
    const size_t wsb = po_pair_gamma_workspace_bytes(n, mc, m1, m2);
    UP(ws, nullptr, wsb);
    // gamma: dense (the Python paths' flavour), or the envelope DP of Gamma.h (its own arithmetic: logaddexp, -inf)
    // written out as a full matrix with -inf outside the stored ranges
    int rc = po_pair_gamma_batch((const double*)a.p, (const int64_t*)ao.p, (const double*)b.p, (const int64_t*)bo.p,
                                 env_h ? (const int32_t*)ev.p : nullptr, env_h ? (const int64_t*)eo.p : nullptr, n, C,
                                 env_h ? 0 : flavor, mc, (double*)g0.p, (double*)dn.p, (const int64_t*)dfo.p,
                                 (int32_t*)st.p, ws.p, wsb, nullptr);
    if (rc != PO_OK) return rc;
    const int64_t seqb = seq_off_h[n];
    UP(so, seq_off_h, sizeof(int64_t) * (n + 1));
    UP(sq, nullptr, (size_t)seqb);
    UP(sl, nullptr, sizeof(int32_t) * n);
    UP(lp, nullptr, sizeof(double) * n);
    UP(st2, nullptr, sizeof(int32_t) * n);
    const int64_t mr = std::max(m1, m2);
    const size_t wsb2 = po_pair_prefix_ws_bytes(n, mr);
    UP(ws2, nullptr, wsb2);
    rc = po_launch_pair_prefix_search((const double*)a.p, (const int64_t*)ao.p, (const double*)b.p, (const int64_t*)bo.p,
                                      (const double*)dn.p, (const int64_t*)dfo.p, n, C, A, ap, flavor, mr, (char*)sq.p,
                                      (const int64_t*)so.p, (int32_t*)sl.p, (double*)lp.p, (int32_t*)st2.p, ws2.p, wsb2, nullptr);
    if (rc != PO_OK) { g_err = "po_pair_prefix_search_env_batch_h: box too long for the LDS rows, or bad C"; return rc; }
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    DOWN(seq_h, sq, (size_t)seqb);
    DOWN(seq_len_h, sl, sizeof(int32_t) * n);
    DOWN(logp_h, lp, sizeof(double) * n);
    DOWN(status_h, st2, sizeof(int32_t) * n);
    return PO_OK;
}

int po_pair_prefix_search_batch_h(const double* y1_h, const int64_t* y1_off_h, const double* y2_h, const int64_t* y2_off_h,
                                  int n, int C, const char* alphabet, int flavor, char* seq_h, const int64_t* seq_off_h,
                                  int32_t* seq_len_h, double* logp_h, int32_t* status_h) {
    return po_pair_prefix_search_env_batch_h(y1_h, y1_off_h, y2_h, y2_off_h, nullptr, nullptr, n, C, alphabet, flavor, seq_h,
                                             seq_off_h, seq_len_h, logp_h, status_h);
}

int po_forward_vec_batch_h(const double* y_h, const int64_t* y_off_h, int n, int C, int s, int i, int flavor,
                           const double* previous_h, double* out_h) {
    g_err.clear();
    if (n <= 0) return PO_OK;
    if (!y_h || !y_off_h || !out_h) { g_err = "po_forward_vec_batch_h: null argument"; return PO_E_ARG; }
    const int64_t rows = y_off_h[n] - y_off_h[0];
    DevBuf y, yo, pv, out;
    UP(y, y_h + y_off_h[0] * C, sizeof(double) * rows * C);
    std::vector<int64_t> off(y_off_h, y_off_h + n + 1);
    for (auto& o : off) o -= y_off_h[0];
    UP(yo, off.data(), sizeof(int64_t) * (n + 1));
    if (previous_h) UP(pv, previous_h, sizeof(double) * rows);
    UP(out, nullptr, sizeof(double) * rows);
    int rc = po_launch_forward_vec((const double*)y.p, (const int64_t*)yo.p, n, C, s, i, flavor,
                                   previous_h ? (const double*)pv.p : nullptr, (double*)out.p, nullptr);
    if (rc != PO_OK) { g_err = "po_forward_vec_batch_h: bad symbol / label length, or previous row missing"; return rc; }
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    DOWN(out_h, out, sizeof(double) * rows);
    return PO_OK;
}

int po_align_scores_batch_h(const char* seqs_h, const int64_t* seq_off_h, int n, int band_width, int match, int mismatch,
                            int gap_cost, char* aln1_h, char* aln2_h, const int64_t* aln_off_h, int32_t* ncol_h,
                            int32_t* status_h) {
    g_err.clear();
    if (n <= 0) return PO_OK;
    int64_t m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) {
        m1 = std::max<int64_t>(m1, seq_off_h[2 * i + 1] - seq_off_h[2 * i]);
        m2 = std::max<int64_t>(m2, seq_off_h[2 * i + 2] - seq_off_h[2 * i + 1]);
    }
    DevBuf sq, so, a1, a2, ao, nc, st, ws;
    UP(sq, seqs_h, (size_t)seq_off_h[2 * n]);
    UP(so, seq_off_h, sizeof(int64_t) * (2 * n + 1));
    UP(a1, nullptr, (size_t)aln_off_h[n]);
    UP(a2, nullptr, (size_t)aln_off_h[n]);
    UP(ao, aln_off_h, sizeof(int64_t) * (n + 1));
    UP(nc, nullptr, sizeof(int32_t) * n);
    UP(st, nullptr, sizeof(int32_t) * n);
    const size_t wsb = po_align_workspace_bytes(n, m1, m2, band_width);
    UP(ws, nullptr, wsb);
    int rc = po_launch_align_scores((const char*)sq.p, (const int64_t*)so.p, n, band_width, match, mismatch, gap_cost, m1, m2,
                                    (char*)a1.p, (char*)a2.p, (const int64_t*)ao.p, (int32_t*)nc.p, (int32_t*)st.p, ws.p, wsb,
                                    nullptr);
    if (rc != PO_OK) { g_err = "po_align_scores_batch_h: workspace too small"; return rc; }
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    DOWN(aln1_h, a1, (size_t)aln_off_h[n]);
    DOWN(aln2_h, a2, (size_t)aln_off_h[n]);
    DOWN(ncol_h, nc, sizeof(int32_t) * n);
    DOWN(status_h, st, sizeof(int32_t) * n);
    return PO_OK;
}

int po_nw_matrix_batch(const char* seqs, const int64_t* seq_off, int n, int match, int mismatch, int gap_cost, int32_t* dp,
                       const int64_t* dp_off, int32_t* status, void* stream) {
    g_err.clear();
    if (n < 0 || !seqs || !seq_off || !dp || !dp_off) { g_err = "po_nw_matrix_batch: null argument"; return PO_E_ARG; }
    return po_launch_nw_matrix(seqs, seq_off, n, match, mismatch, gap_cost, dp, dp_off, status, (hipStream_t)stream);
}
int po_nw_matrix_batch_h(const char* seqs_h, const int64_t* seq_off_h, int n, int match, int mismatch, int gap_cost, int32_t* dp_h,
                         const int64_t* dp_off_h, int32_t* status_h) {
    g_err.clear();
    if (n <= 0) return PO_OK;
    DevBuf sq, so, dp, dpo, st;
    UP(sq, seqs_h, (size_t)seq_off_h[2 * n]);
    UP(so, seq_off_h, sizeof(int64_t) * (2 * n + 1));
    UP(dp, nullptr, sizeof(int32_t) * (size_t)dp_off_h[n]);
    UP(dpo, dp_off_h, sizeof(int64_t) * (n + 1));
    UP(st, nullptr, sizeof(int32_t) * n);
    const int rc = po_launch_nw_matrix((const char*)sq.p, (const int64_t*)so.p, n, match, mismatch, gap_cost, (int32_t*)dp.p,
                                       (const int64_t*)dpo.p, (int32_t*)st.p, nullptr);
    if (rc != PO_OK) return rc;
    HIPCHK(hipDeviceSynchronize());
    DOWN(dp_h, dp, sizeof(int32_t) * (size_t)dp_off_h[n]);
    DOWN(status_h, st, sizeof(int32_t) * n);
    return PO_OK;
}

int po_align_batch_h(const char* seqs_h, const int64_t* seq_off_h, int n, int band_width, char* aln1_h, char* aln2_h,
                     const int64_t* aln_off_h, int32_t* ncol_h, int32_t* status_h) {
    return po_align_scores_batch_h(seqs_h, seq_off_h, n, band_width, 2, -1, -1, aln1_h, aln2_h, aln_off_h, ncol_h, status_h);
}

int po_envelope_batch_h(const char* aln1_h, const char* aln2_h, const int64_t* aln_off_h, const int32_t* ncol_h, int n,
                        const int32_t* map1_h, const int64_t* map1_off_h, const int32_t* map2_h,
                        const int64_t* map2_off_h, const int32_t* U_h, const int32_t* V_h, int padding, int32_t* env_h,
                        const int64_t* env_off_h, int32_t* status_h) {
    g_err.clear();
    if (n <= 0) return PO_OK;
    int64_t mc = 0;
    for (int i = 0; i < n; ++i) mc = std::max<int64_t>(mc, ncol_h[i]);
    DevBuf a1, a2, ao, nc, m1, m1o, m2, m2o, u, v, ev, eo, st, ws;
    UP(a1, aln1_h, (size_t)aln_off_h[n]);
    UP(a2, aln2_h, (size_t)aln_off_h[n]);
    UP(ao, aln_off_h, sizeof(int64_t) * (n + 1));
    UP(nc, ncol_h, sizeof(int32_t) * n);
    UP(m1, map1_h, sizeof(int32_t) * (size_t)map1_off_h[n]);
    UP(m1o, map1_off_h, sizeof(int64_t) * (n + 1));
    UP(m2, map2_h, sizeof(int32_t) * (size_t)map2_off_h[n]);
    UP(m2o, map2_off_h, sizeof(int64_t) * (n + 1));
    UP(u, U_h, sizeof(int32_t) * n);
    UP(v, V_h, sizeof(int32_t) * n);
    UP(ev, nullptr, sizeof(int32_t) * 2 * (size_t)env_off_h[n]);
    UP(eo, env_off_h, sizeof(int64_t) * (n + 1));
    UP(st, nullptr, sizeof(int32_t) * n);
    const size_t wsb = po_envelope_workspace_bytes(n, mc);
    UP(ws, nullptr, wsb);
    int rc = po_envelope_batch((const char*)a1.p, (const char*)a2.p, (const int64_t*)ao.p, (const int32_t*)nc.p, n,
                               (const int32_t*)m1.p, (const int64_t*)m1o.p, (const int32_t*)m2.p, (const int64_t*)m2o.p,
                               (const int32_t*)u.p, (const int32_t*)v.p, padding, (int32_t*)ev.p, (const int64_t*)eo.p,
                               (int32_t*)st.p, ws.p, wsb, nullptr);
    if (rc != PO_OK) return rc;
    HIPCHK(hipDeviceSynchronize());
    DOWN(env_h, ev, sizeof(int32_t) * 2 * (size_t)env_off_h[n]);
    DOWN(status_h, st, sizeof(int32_t) * n);
    return PO_OK;
}

int po_prefix_search_batch_h(const double* y_h, const int64_t* y_off_h, int n, int C, const char* alphabet, char* seq_h,
                             const int64_t* seq_off_h, int32_t* seq_len_h, double* logp_h, int32_t* status_h) {
    g_err.clear();
    if (n <= 0) return PO_OK;
    const int64_t rows = y_off_h[n] - y_off_h[0];
    const int64_t seqb = seq_off_h[n];
    int64_t mx = 0;
    for (int i = 0; i < n; ++i) mx = std::max<int64_t>(mx, y_off_h[i + 1] - y_off_h[i]);
    DevBuf y, yo, so, sq, sl, lp, st, ws;
    UP(y, y_h + y_off_h[0] * C, sizeof(double) * rows * C);
    std::vector<int64_t> off(y_off_h, y_off_h + n + 1);
    for (auto& o : off) o -= y_off_h[0];
    UP(yo, off.data(), sizeof(int64_t) * (n + 1));
    UP(so, seq_off_h, sizeof(int64_t) * (n + 1));
    UP(sq, nullptr, (size_t)seqb);
    UP(sl, nullptr, sizeof(int32_t) * n);
    UP(lp, nullptr, sizeof(double) * n);
    UP(st, nullptr, sizeof(int32_t) * n);
    const size_t wsb = po_prefix_search_workspace_bytes(n, mx);
    UP(ws, nullptr, wsb);
    int rc = po_prefix_search_batch((const double*)y.p, (const int64_t*)yo.p, n, C, alphabet, (char*)sq.p,
                                    (const int64_t*)so.p, (int32_t*)sl.p, (double*)lp.p, (int32_t*)st.p, ws.p, wsb, nullptr);
    if (rc != PO_OK) return rc;
    HIPCHK(hipDeviceSynchronize());
    DOWN(seq_h, sq, (size_t)seqb);
    DOWN(seq_len_h, sl, sizeof(int32_t) * n);
    DOWN(logp_h, lp, sizeof(double) * n);
    DOWN(status_h, st, sizeof(int32_t) * n);
    return PO_OK;
}

int po_forward_batch_h(const double* y_h, const int64_t* y_off_h, int n, int C, const char* alphabet, int model,
                       const char* labels_h, const int64_t* label_off_h, double* logp_h, int32_t* status_h) {
    g_err.clear();
    if (n <= 0) return PO_OK;
    const int64_t rows = y_off_h[n] - y_off_h[0], nl = label_off_h[n] - label_off_h[0];
    int64_t mx = 0;
    for (int i = 0; i < n; ++i) mx = std::max<int64_t>(mx, y_off_h[i + 1] - y_off_h[i]);
    DevBuf y, yo, lb, lo, out, st, ws;
    UP(y, y_h + y_off_h[0] * C, sizeof(double) * rows * C);
    std::vector<int64_t> off(y_off_h, y_off_h + n + 1), lof(label_off_h, label_off_h + n + 1);
    for (auto& o : off) o -= y_off_h[0];
    for (auto& o : lof) o -= label_off_h[0];
    UP(yo, off.data(), sizeof(int64_t) * (n + 1));
    UP(lb, labels_h + label_off_h[0], (size_t)nl);
    UP(lo, lof.data(), sizeof(int64_t) * (n + 1));
    UP(out, nullptr, sizeof(double) * n);
    UP(st, nullptr, sizeof(int32_t) * n);
    const size_t wsb = po_forward_workspace_bytes(n, mx, model);
    UP(ws, nullptr, wsb);
    int rc = po_forward_batch((const double*)y.p, (const int64_t*)yo.p, n, C, alphabet, model, (const char*)lb.p,
                              (const int64_t*)lo.p, (double*)out.p, (int32_t*)st.p, ws.p, wsb, nullptr);
    if (rc != PO_OK) return rc;
    HIPCHK(hipDeviceSynchronize());
    DOWN(logp_h, out, sizeof(double) * n);
    DOWN(status_h, st, sizeof(int32_t) * n);
    return PO_OK;
}

int po_viterbi_acceptor_batch_h(const double* y_h, const int64_t* y_off_h, int n, int C, const char* alphabet,
                                int band_size, const char* labels_h, const int64_t* label_off_h, int32_t* path_h,
                                int32_t* status_h) {
    g_err.clear();
    if (n <= 0) return PO_OK;
    const int64_t rows = y_off_h[n] - y_off_h[0], nl = label_off_h[n] - label_off_h[0];
    int64_t mx = 0, ml = 0;
    for (int i = 0; i < n; ++i) {
        mx = std::max<int64_t>(mx, y_off_h[i + 1] - y_off_h[i]);
        ml = std::max<int64_t>(ml, label_off_h[i + 1] - label_off_h[i]);
    }
    DevBuf y, yo, lb, lo, pt, st, ws;
    UP(y, y_h + y_off_h[0] * C, sizeof(double) * rows * C);
    std::vector<int64_t> off(y_off_h, y_off_h + n + 1), lof(label_off_h, label_off_h + n + 1);
    for (auto& o : off) o -= y_off_h[0];
    for (auto& o : lof) o -= label_off_h[0];
    UP(yo, off.data(), sizeof(int64_t) * (n + 1));
    UP(lb, labels_h + label_off_h[0], (size_t)nl);
    UP(lo, lof.data(), sizeof(int64_t) * (n + 1));
    UP(pt, nullptr, sizeof(int32_t) * rows);
    UP(st, nullptr, sizeof(int32_t) * n);
    const size_t wsb = po_viterbi_acceptor_workspace_bytes(n, mx, ml);
    UP(ws, nullptr, wsb);
    int rc = po_viterbi_acceptor_batch((const double*)y.p, (const int64_t*)yo.p, n, C, alphabet, band_size,
                                       (const char*)lb.p, (const int64_t*)lo.p, (int32_t*)pt.p, (int32_t*)st.p, ws.p,
                                       wsb, nullptr);
    if (rc != PO_OK) return rc;
    HIPCHK(hipDeviceSynchronize());
    DOWN(path_h, pt, sizeof(int32_t) * rows);
    DOWN(status_h, st, sizeof(int32_t) * n);
    return PO_OK;
}

int po_viterbi_acceptor_cy_batch_h(const double* y_h, const int64_t* y_off_h, int n, int C, const char* alphabet,
                                   int band_size, const char* labels_h, const int64_t* label_off_h, int32_t* path_h,
                                   int32_t* status_h) {
    g_err.clear();
    if (band_size < 0) { g_err = "po_viterbi_acceptor_cy_batch_h: negative band"; return PO_E_ARG; }
    return po_viterbi_acceptor_batch_h(y_h, y_off_h, n, C, alphabet, -(band_size + 1), labels_h, label_off_h, path_h, status_h);
}

int po_beam2d_batch_h(const double* y1_h, const int64_t* y1_off_h, const double* y2_h, const int64_t* y2_off_h,
                      const int32_t* env_h, int n, int C, const char* alphabet, int W, int model, int method,
                      char* seq_h,
                      const int64_t* seq_off_h, int32_t* seq_len_h, int32_t* status_h) {
    g_err.clear();
    if (n <= 0) return PO_OK;
    const int64_t r1 = y1_off_h[n] - y1_off_h[0], r2 = y2_off_h[n] - y2_off_h[0];
    const int64_t seqb = seq_off_h[n];
    int64_t m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) {
        m1 = std::max<int64_t>(m1, y1_off_h[i + 1] - y1_off_h[i]);
        m2 = std::max<int64_t>(m2, y2_off_h[i + 1] - y2_off_h[i]);
    }
    DevBuf a, ao, b, bo, ev, so, sq, sl, st, ws;
    UP(a, y1_h + y1_off_h[0] * C, sizeof(double) * r1 * C);
    UP(b, y2_h + y2_off_h[0] * C, sizeof(double) * r2 * C);
    std::vector<int64_t> o1(y1_off_h, y1_off_h + n + 1), o2(y2_off_h, y2_off_h + n + 1);
    for (auto& o : o1) o -= y1_off_h[0];
    for (auto& o : o2) o -= y2_off_h[0];
    UP(ao, o1.data(), sizeof(int64_t) * (n + 1));
    UP(bo, o2.data(), sizeof(int64_t) * (n + 1));
    if (env_h) UP(ev, env_h + 2 * y1_off_h[0], sizeof(int32_t) * 2 * r1);
    UP(so, seq_off_h, sizeof(int64_t) * (n + 1));
    UP(sq, nullptr, (size_t)seqb);
    UP(sl, nullptr, sizeof(int32_t) * n);
    UP(st, nullptr, sizeof(int32_t) * n);
    // (without an envelope everything but "row" runs the grid method, as in the reference's dispatcher)
    const size_t wsb = po_beam2d_workspace_bytes(n, r1, r2, m1, m2, C, W, model,
                                                 (!env_h && method != PO_METHOD_ROW) ? PO_METHOD_GRID_NOENV : method);
    UP(ws, nullptr, wsb);
    int rc = po_beam2d_batch((const double*)a.p, (const int64_t*)ao.p, (const double*)b.p, (const int64_t*)bo.p,
                             env_h ? (const int32_t*)ev.p : nullptr, n, C, alphabet, W, model, method, (char*)sq.p,
                             (const int64_t*)so.p, (int32_t*)sl.p, (int32_t*)st.p, ws.p, wsb, nullptr);
    if (rc != PO_OK) return rc;
    HIPCHK(hipDeviceSynchronize());
    DOWN(seq_h, sq, (size_t)seqb);
    DOWN(seq_len_h, sl, sizeof(int32_t) * n);
    DOWN(status_h, st, sizeof(int32_t) * n);
    return PO_OK;
}

int po_pair_decode_from_1d_batch_h(const double* y1_h, const int64_t* y1_off_h, const double* y2_h, const int64_t* y2_off_h,
                                   int n, int C, const po_pair_options* opt, const char* seq1d_h,
                                   const int64_t* seq1d_off_h, const int32_t* len1_h, const int32_t* len2_h,
                                   const int32_t* map1_h, const int32_t* map2_h, double* identity_h, int32_t* env_out_h,
                                   char* seq_h, const int64_t* seq_off_h, int32_t* seq_len_h, int32_t* status_h) {
    g_err.clear();
    if (n <= 0) return PO_OK;
    const int64_t r1 = y1_off_h[n] - y1_off_h[0], r2 = y2_off_h[n] - y2_off_h[0];
    const int64_t seqb = seq_off_h[n], s1b = seq1d_off_h[2 * n];
    int64_t m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) {
        m1 = std::max<int64_t>(m1, y1_off_h[i + 1] - y1_off_h[i]);
        m2 = std::max<int64_t>(m2, y2_off_h[i + 1] - y2_off_h[i]);
    }
    DevBuf a, ao, b, bo, so, sq, sl, st, s1o, s1, l1, l2, idn, ev, mp1, mp2, ws;
    UP(a, y1_h + y1_off_h[0] * C, sizeof(double) * r1 * C);
    UP(b, y2_h + y2_off_h[0] * C, sizeof(double) * r2 * C);
    std::vector<int64_t> o1(y1_off_h, y1_off_h + n + 1), o2(y2_off_h, y2_off_h + n + 1);
    for (auto& o : o1) o -= y1_off_h[0];
    for (auto& o : o2) o -= y2_off_h[0];
    UP(ao, o1.data(), sizeof(int64_t) * (n + 1));
    UP(bo, o2.data(), sizeof(int64_t) * (n + 1));
    UP(so, seq_off_h, sizeof(int64_t) * (n + 1));
    UP(sq, nullptr, (size_t)seqb);
    UP(sl, nullptr, sizeof(int32_t) * n);
    UP(st, nullptr, sizeof(int32_t) * n);
    UP(s1o, seq1d_off_h, sizeof(int64_t) * (2 * n + 1));
    UP(s1, seq1d_h, (size_t)s1b);
    UP(l1, len1_h, sizeof(int32_t) * n);
    UP(l2, len2_h, sizeof(int32_t) * n);
    UP(mp1, map1_h + y1_off_h[0], sizeof(int32_t) * r1);
    UP(mp2, map2_h + y2_off_h[0], sizeof(int32_t) * r2);
    UP(idn, nullptr, sizeof(double) * n);
    UP(ev, nullptr, sizeof(int32_t) * 2 * r1);
    const size_t wsb = po_pair_decode_workspace_bytes(n, r1, r2, m1, m2, C, opt);
    UP(ws, nullptr, wsb);
    int rc = po_launch_pair_decode_from_1d((const double*)a.p, (const int64_t*)ao.p, (const double*)b.p, (const int64_t*)bo.p, n,
                                           C, opt, r1, r2, m1, m2, (const int32_t*)mp1.p, (const int32_t*)mp2.p, (char*)s1.p,
                                           (const int64_t*)s1o.p, (int32_t*)l1.p, (int32_t*)l2.p, (double*)idn.p,
                                           (int32_t*)ev.p, (char*)sq.p, (const int64_t*)so.p, (int32_t*)sl.p, (int32_t*)st.p,
                                           ws.p, wsb, nullptr);
    if (rc != PO_OK) { g_err = "po_pair_decode_from_1d_batch_h: launch refused"; return rc; }
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    DOWN(seq_h, sq, (size_t)seqb);
    DOWN(seq_len_h, sl, sizeof(int32_t) * n);
    DOWN(status_h, st, sizeof(int32_t) * n);
    DOWN(identity_h, idn, sizeof(double) * n);
    DOWN(env_out_h, ev, sizeof(int32_t) * 2 * r1);
    return PO_OK;
}

int po_pair_decode_batch_h(const double* y1_h, const int64_t* y1_off_h, const double* y2_h,
                           const int64_t* y2_off_h, int n, int C, const po_pair_options* opt, char* seq1d_h,
                           const int64_t* seq1d_off_h, int32_t* len1_h, int32_t* len2_h, double* identity_h,
                           int32_t* env_out_h, char* seq_h, const int64_t* seq_off_h, int32_t* seq_len_h,
                           int32_t* status_h) {
    g_err.clear();
    if (n <= 0) return PO_OK;
    const int64_t r1 = y1_off_h[n] - y1_off_h[0], r2 = y2_off_h[n] - y2_off_h[0];
    const int64_t seqb = seq_off_h[n], s1b = seq1d_off_h[2 * n];
    int64_t m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) {
        m1 = std::max<int64_t>(m1, y1_off_h[i + 1] - y1_off_h[i]);
        m2 = std::max<int64_t>(m2, y2_off_h[i + 1] - y2_off_h[i]);
    }
    DevBuf a, ao, b, bo, so, sq, sl, st, s1o, s1, l1, l2, idn, ev, ws;
    UP(a, y1_h + y1_off_h[0] * C, sizeof(double) * r1 * C);
    UP(b, y2_h + y2_off_h[0] * C, sizeof(double) * r2 * C);
    std::vector<int64_t> o1(y1_off_h, y1_off_h + n + 1), o2(y2_off_h, y2_off_h + n + 1);
    for (auto& o : o1) o -= y1_off_h[0];
    for (auto& o : o2) o -= y2_off_h[0];
    UP(ao, o1.data(), sizeof(int64_t) * (n + 1));
    UP(bo, o2.data(), sizeof(int64_t) * (n + 1));
    UP(so, seq_off_h, sizeof(int64_t) * (n + 1));
    UP(sq, nullptr, (size_t)seqb);
    UP(sl, nullptr, sizeof(int32_t) * n);
    UP(st, nullptr, sizeof(int32_t) * n);
    UP(s1o, seq1d_off_h, sizeof(int64_t) * (2 * n + 1));
    UP(s1, nullptr, (size_t)s1b);
    UP(l1, nullptr, sizeof(int32_t) * n);
    UP(l2, nullptr, sizeof(int32_t) * n);
    UP(idn, nullptr, sizeof(double) * n);
    UP(ev, nullptr, sizeof(int32_t) * 2 * r1);
    const size_t wsb = po_pair_decode_workspace_bytes(n, r1, r2, m1, m2, C, opt);
    UP(ws, nullptr, wsb);
    int rc = po_pair_decode_batch((const double*)a.p, (const int64_t*)ao.p, (const double*)b.p,
                                  (const int64_t*)bo.p, n, C, opt, (char*)s1.p, (const int64_t*)s1o.p,
                                  (int32_t*)l1.p, (int32_t*)l2.p, (double*)idn.p, (int32_t*)ev.p, (char*)sq.p,
                                  (const int64_t*)so.p, (int32_t*)sl.p, (int32_t*)st.p, ws.p, wsb, nullptr);
    if (rc != PO_OK) return rc;
    HIPCHK(hipDeviceSynchronize());
    DOWN(seq_h, sq, (size_t)seqb);
    DOWN(seq_len_h, sl, sizeof(int32_t) * n);
    DOWN(status_h, st, sizeof(int32_t) * n);
    DOWN(seq1d_h, s1, (size_t)s1b);
    DOWN(len1_h, l1, sizeof(int32_t) * n);
    DOWN(len2_h, l2, sizeof(int32_t) * n);
    DOWN(identity_h, idn, sizeof(double) * n);
    DOWN(env_out_h, ev, sizeof(int32_t) * 2 * r1);
    return PO_OK;
}

// -------------------------------------------------------------------------------- events / profile
void* po_event_create(void) {
    hipEvent_t e;
    return hipEventCreate(&e) == hipSuccess ? (void*)e : nullptr;
}
int po_event_record(void* ev, void* stream) {
    g_err.clear();
    HIPCHK(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return PO_OK;
}
int po_event_elapsed_ms(void* start, void* stop, float* ms) {
    g_err.clear();
    HIPCHK(hipEventSynchronize((hipEvent_t)stop));
    HIPCHK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return PO_OK;
}
void po_event_destroy(void* ev) { if (ev) (void)hipEventDestroy((hipEvent_t)ev); }

int po_profile_update_counter(uint64_t* device_counter) {
    g_err.clear();
    po_b2_set_update_counter((unsigned long long*)device_counter);
    return PO_OK;
}
int po_lae_peak(int iters, double* lae_per_s, void* stream) {
    g_err.clear();
    if (iters < 1 || !lae_per_s) { g_err = "po_lae_peak: bad argument"; return PO_E_ARG; }
    int rc = po_launch_lae_peak(iters, lae_per_s, (hipStream_t)stream);
    if (rc != PO_OK) g_err = "po_lae_peak: launch failed";
    return rc;
}
void po_profile_enable(int on) {
    g_prof_on = on != 0;
    po_b2_set_mark(g_prof_on ? b2_mark : nullptr);
}
void po_profile_reset(void) {
    prof_drain();
    std::lock_guard<std::mutex> lk(g_prof_mu);
    std::memset(g_prof_ms, 0, sizeof(g_prof_ms));
    std::memset(g_prof_n, 0, sizeof(g_prof_n));
}
int po_profile_get(int kernel, double* total_ms, int64_t* launches) {
    g_err.clear();
    if (kernel < 0 || kernel >= PO_K_COUNT) return PO_E_ARG;
    prof_drain();
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (total_ms) *total_ms = g_prof_ms[kernel];
    if (launches) *launches = g_prof_n[kernel];
    return PO_OK;
}
// internal: lets the pair pipeline time its own stages under their kernel ids
void po_prof_stage(int kernel, hipStream_t s, int begin, void** tok) {
    if (!g_prof_on) return;
    if (begin) {
        auto* r = new ProfRec{nullptr, nullptr, kernel};
        (void)hipEventCreate(&r->a);
        (void)hipEventCreate(&r->b);
        (void)hipEventRecord(r->a, s);
        *tok = r;
    } else if (*tok) {
        auto* r = (ProfRec*)*tok;
        (void)hipEventRecord(r->b, s);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back(*r);
        delete r;
        *tok = nullptr;
    }
}

}  // extern "C"
