// Host-to-strings pipeline of the pair decoder: the native replacement of the reference's per-pair worker
// processes (multiprocessing.Pool over pair_decode_helper, pair_decode.py:292-297) together with their trace loading
// (decode.load_logits / logit_to_log_likelihood, decode.py:34-51; transducer.reverse_complement, transducer.py:68-70;
// the Bonito column order, decode.py:79; the uint8 trace scaling, decode.py:89-93).
//
// The caller hands over n pairs as HOST arrays in the form the basecaller wrote them (float32 logits, uint8 traces
// or float64 log-probabilities) and gets the strings back.  The pairs are cut into WAVES; each wave goes through
// one of two SLOTS, a slot being {HIP stream, pinned staging buffers, device buffers, workspace}:
//
//     pack (host threads: item arrays -> pinned staging)  ->  H2D  ->  ingest kernels (log-softmax / trace scaling,
//     permutation, time reversal: 4 or 1 byte per value over PCIe instead of 8)  ->  the pair-decode launch chain
//     (po_launch_pair_decode_geom: Viterbi x2, alignment + envelope, pair beam search)  ->  D2H of the results
//
// all asynchronous on the slot's stream, so while the GPU decodes wave k the host packs and uploads wave k + 1 on the
// other stream, and the tail of one wave's persistent kernels overlaps the head of the next.  Buffers are allocated
// once and grow only; the working set is bounded by the wave size whatever n is (the drivers never hold more
// than two waves of device memory).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/poreover_hip.h"

extern "C" {
size_t po_pair_ws_bytes_impl(int, int64_t, int64_t, int64_t, int64_t, int, const po_pair_options*);
int po_launch_pair_decode_geom(const double*, const int64_t*, const double*, const int64_t*, int, int, const po_pair_options*,
                               int64_t, int64_t, int64_t, int64_t, const int32_t*, const int32_t*, char*, const int64_t*,
                               int32_t*, int32_t*, double*, int32_t*, char*, const int64_t*, int32_t*, int32_t*, void*, size_t,
                               hipStream_t);
int po_launch_ingest(const void*, const int64_t*, int, int, int, const int*, int, int64_t, double*, hipStream_t);
void po_set_error(const char* msg);
}

namespace {
inline size_t al256(size_t b) { return (b + 255) & ~size_t(255); }
inline double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct GrowBuf {   // grow-only buffer: device memory or pinned host memory
    void* p = nullptr;
    size_t cap = 0;
    bool host;
    explicit GrowBuf(bool host_) : host(host_) {}
    GrowBuf(const GrowBuf&) = delete;
    GrowBuf& operator=(const GrowBuf&) = delete;
    ~GrowBuf() { release(); }
    void release() {
        if (p) { if (host) (void)hipHostFree(p); else (void)hipFree(p); }
        p = nullptr; cap = 0;
    }
    bool ensure(size_t bytes) {
        if (bytes <= cap) return true;
        release();
        const size_t want = al256(bytes + bytes / 8);   // head-room: waves differ a little in size
        const hipError_t e = host ? hipHostMalloc(&p, want, hipHostMallocDefault) : hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return false; }
        cap = want;
        return true;
    }
};

struct Slot {
    hipStream_t st = nullptr;
    GrowBuf h_in{true}, h_off{true}, h_out{true};
    GrowBuf d_in{false}, d_y{false}, d_off{false}, d_out{false}, d_ws{false};
    // the wave in flight
    int first = 0, n = 0;
    bool busy = false;
    int64_t tr1 = 0, tr2 = 0;
    // layout of the wave's outputs inside d_out / h_out (same offsets on both sides)
    size_t o_seq1d = 0, o_seq = 0, o_l1 = 0, o_l2 = 0, o_len = 0, o_st = 0, o_id = 0, o_env = 0, out_bytes = 0;
    std::vector<int64_t> s1o, so;   // wave-local output offsets (also uploaded)
};
}  // namespace

struct po_pipeline {
    int device = 0, wave_pairs = 4096, threads = 8;
    int64_t wave_rows = (int64_t)64 << 20;
    Slot slot[2];
    double pack_ms = 0, wait_ms = 0, total_ms = 0;
    int waves = 0;
    std::string err;
};

namespace {
int fail(po_pipeline* p, int code, const std::string& msg) {
    p->err = msg;
    po_set_error(msg.c_str());
    return code;
}
#define PCHK(x)                                                                             \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) return fail(p, PO_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); \
    } while (0)

// copy the items [lo, hi) of a wave into the pinned staging buffer, split over host threads by bytes
void pack_items(const void* const* src, const int64_t* rows, int first, int n, size_t row_bytes, char* dst,
                const std::vector<int64_t>& off, int threads) {
    if (n <= 0) return;
    const int nt = std::max(1, std::min(threads, n));
    if (nt == 1) {
        for (int i = 0; i < n; ++i) std::memcpy(dst + (size_t)off[i] * row_bytes, src[first + i], (size_t)rows[first + i] * row_bytes);
        return;
    }
    const int64_t total = off[n];
    std::vector<std::thread> th;
    int lo = 0;
    for (int t = 0; t < nt; ++t) {
        const int64_t target = total * (t + 1) / nt;
        int hi = lo;
        while (hi < n && (off[hi + 1] <= target || t == nt - 1)) ++hi;
        if (t == nt - 1) hi = n;
        if (hi > lo)
            th.emplace_back([=, &off]() {
                for (int i = lo; i < hi; ++i)
                    std::memcpy(dst + (size_t)off[i] * row_bytes, src[first + i], (size_t)rows[first + i] * row_bytes);
            });
        lo = hi;
    }
    for (auto& x : th) x.join();
}
}  // namespace

extern "C" {

po_pipeline* po_pipeline_create(int device, int wave_pairs, int64_t wave_rows, int threads) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { po_set_error("po_pipeline_create: no such device"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { po_set_error("po_pipeline_create: hipSetDevice failed"); return nullptr; }
    po_pipeline* p = new po_pipeline();
    p->device = device;
    if (wave_pairs > 0) p->wave_pairs = wave_pairs;
    if (wave_rows > 0) p->wave_rows = wave_rows;
    const unsigned hc = std::thread::hardware_concurrency();
    p->threads = threads > 0 ? threads : (int)std::max(1u, std::min(16u, hc ? hc / 2 : 4u));
    for (auto& s : p->slot)
        if (hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking) != hipSuccess) {
            po_set_error("po_pipeline_create: hipStreamCreate failed");
            delete p;
            return nullptr;
        }
    return p;
}

void po_pipeline_destroy(po_pipeline* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (auto& s : p->slot) {
        if (s.st) { (void)hipStreamSynchronize(s.st); (void)hipStreamDestroy(s.st); }
    }
    delete p;
}

int po_pipeline_stats(po_pipeline* p, double* pack_ms, double* wait_ms, double* total_ms, int* waves) {
    if (!p) return PO_E_ARG;
    if (pack_ms) *pack_ms = p->pack_ms;
    if (wait_ms) *wait_ms = p->wait_ms;
    if (total_ms) *total_ms = p->total_ms;
    if (waves) *waves = p->waves;
    return PO_OK;
}

int po_pipeline_pair_decode(po_pipeline* p, const void* const* y1_h, const int64_t* rows1, const void* const* y2_h,
                            const int64_t* rows2, int n, int C, int in_mode, const int* perm1, const int* perm2,
                            int reverse2, const po_pair_options* opt, char* seq1d_h, const int64_t* seq1d_off_h,
                            int32_t* len1_h, int32_t* len2_h, double* identity_h, int32_t* env_out_h, char* seq_h,
                            const int64_t* seq_off_h, int32_t* seq_len_h, int32_t* status_h) {
    if (!p) return PO_E_ARG;
    if (n < 0 || !y1_h || !rows1 || !y2_h || !rows2 || !opt || !seq1d_h || !seq1d_off_h || !len1_h || !len2_h ||
        !identity_h || !seq_h || !seq_off_h || !seq_len_h || !status_h)
        return fail(p, PO_E_ARG, "po_pipeline_pair_decode: null argument");
    if (C < 1 || C > 8 || in_mode < 0 || in_mode > 2) return fail(p, PO_E_ARG, "po_pipeline_pair_decode: bad C / input mode");
    PCHK(hipSetDevice(p->device));
    const size_t esz = in_mode == PO_INGEST_LOGITS_F32 ? 4 : (in_mode == PO_INGEST_TRACE_U8 ? 1 : 8);
    const size_t row_in = esz * (size_t)C, row_y = sizeof(double) * (size_t)C;
    const double t_begin = now_ms();
    p->pack_ms = p->wait_ms = 0;
    p->waves = 0;
    std::vector<int64_t> env_row0;   // global row offset of every pair's envelope in env_out_h
    if (env_out_h) {
        env_row0.resize((size_t)n + 1, 0);
        for (int i = 0; i < n; ++i) env_row0[i + 1] = env_row0[i] + rows1[i];
    }

    // results of the wave a slot holds -> the caller's arrays (after the slot's stream has drained)
    auto drain = [&](Slot& s) -> int {
        if (!s.busy) return PO_OK;
        const double t0 = now_ms();
        PCHK(hipStreamSynchronize(s.st));
        p->wait_ms += now_ms() - t0;
        const char* ho = (const char*)s.h_out.p;
        const int32_t* l1 = (const int32_t*)(ho + s.o_l1);
        const int32_t* l2 = (const int32_t*)(ho + s.o_l2);
        const int32_t* ln = (const int32_t*)(ho + s.o_len);
        const int32_t* st = (const int32_t*)(ho + s.o_st);
        const double* idn = (const double*)(ho + s.o_id);
        const int64_t* o1 = (const int64_t*)s.h_off.p;
        for (int i = 0; i < s.n; ++i) {
            const int g = s.first + i;
            len1_h[g] = l1[i]; len2_h[g] = l2[i]; seq_len_h[g] = ln[i]; status_h[g] = st[i]; identity_h[g] = idn[i];
            const int64_t c1 = seq1d_off_h[2 * g + 1] - seq1d_off_h[2 * g], c2 = seq1d_off_h[2 * g + 2] - seq1d_off_h[2 * g + 1];
            const int64_t cc = seq_off_h[g + 1] - seq_off_h[g];
            if (l1[i] > c1 || l2[i] > c2 || ln[i] > cc) { status_h[g] = PO_E_CAP; seq_len_h[g] = 0; continue; }
            std::memcpy(seq1d_h + seq1d_off_h[2 * g], ho + s.o_seq1d + s.s1o[2 * i], (size_t)std::max(0, l1[i]));
            std::memcpy(seq1d_h + seq1d_off_h[2 * g + 1], ho + s.o_seq1d + s.s1o[2 * i + 1], (size_t)std::max(0, l2[i]));
            std::memcpy(seq_h + seq_off_h[g], ho + s.o_seq + s.so[i], (size_t)std::max(0, ln[i]));
            if (env_out_h)
                std::memcpy(env_out_h + 2 * env_row0[g], ho + s.o_env + sizeof(int32_t) * 2 * (size_t)o1[i],
                            sizeof(int32_t) * 2 * (size_t)rows1[g]);
        }
        s.busy = false;
        return PO_OK;
    };

    int first = 0, wave = 0;
    while (first < n) {
        // ---- plan the wave: pairs [first, first + wn)
        int wn = 0;
        int64_t r1 = 0, r2 = 0, m1 = 0, m2 = 0;
        while (first + wn < n && wn < p->wave_pairs) {
            const int64_t a = rows1[first + wn], b = rows2[first + wn];
            if (a < 0 || b < 0) return fail(p, PO_E_ARG, "po_pipeline_pair_decode: negative row count");
            if (wn > 0 && r1 + r2 + a + b > p->wave_rows) break;
            r1 += a; r2 += b; m1 = std::max(m1, a); m2 = std::max(m2, b);
            ++wn;
        }
        Slot& s = p->slot[wave & 1];
        int rc = drain(s);   // the wave this slot ran two waves ago
        if (rc != PO_OK) return rc;
        s.first = first; s.n = wn; s.tr1 = r1; s.tr2 = r2;

        // ---- offsets: [o1 (wn+1) | o2 (wn+1) | s1o (2wn+1) | so (wn+1)], pinned, uploaded as one block
        const size_t n_off = (size_t)(wn + 1) * 3 + (size_t)(2 * wn + 1);
        if (!s.h_off.ensure(sizeof(int64_t) * n_off) || !s.d_off.ensure(sizeof(int64_t) * n_off))
            return fail(p, PO_E_HIP, "po_pipeline_pair_decode: out of memory (offset tables)");
        int64_t* o1 = (int64_t*)s.h_off.p;
        int64_t* o2 = o1 + (wn + 1);
        int64_t* s1o = o2 + (wn + 1);
        int64_t* so = s1o + (2 * wn + 1);
        o1[0] = o2[0] = s1o[0] = so[0] = 0;
        for (int i = 0; i < wn; ++i) {
            const int64_t a = rows1[first + i], b = rows2[first + i];
            o1[i + 1] = o1[i] + a; o2[i + 1] = o2[i] + b;
            s1o[2 * i + 1] = s1o[2 * i] + a; s1o[2 * i + 2] = s1o[2 * i + 1] + b;   // a basecall has at most one base per frame
            so[i + 1] = so[i] + a + b;
        }
        s.s1o.assign(s1o, s1o + 2 * wn + 1);
        s.so.assign(so, so + wn + 1);
        std::vector<int64_t> off1(o1, o1 + wn + 1), off2(o2, o2 + wn + 1);

        // ---- buffers
        const size_t in1 = al256(row_in * (size_t)r1), in2 = al256(row_in * (size_t)r2);
        const size_t yb1 = al256(row_y * (size_t)r1), yb2 = al256(row_y * (size_t)r2);
        size_t o = 0;
        s.o_seq1d = o; o += al256((size_t)(r1 + r2) + 16);
        s.o_seq = o; o += al256((size_t)(r1 + r2) + 16);
        s.o_l1 = o; o += al256(sizeof(int32_t) * wn);
        s.o_l2 = o; o += al256(sizeof(int32_t) * wn);
        s.o_len = o; o += al256(sizeof(int32_t) * wn);
        s.o_st = o; o += al256(sizeof(int32_t) * wn);
        s.o_id = o; o += al256(sizeof(double) * wn);
        s.o_env = o; o += al256(sizeof(int32_t) * 2 * (size_t)r1);
        s.out_bytes = o;
        const size_t wsb = po_pair_ws_bytes_impl(wn, r1, r2, m1, m2, C, opt);
        const bool direct = (in_mode == PO_INGEST_F64 && !perm1 && !perm2 && !reverse2);   // log-probabilities as they are
        if (!s.h_in.ensure(in1 + in2) || !s.d_y.ensure(yb1 + yb2) || (!direct && !s.d_in.ensure(in1 + in2)) ||
            !s.d_out.ensure(s.out_bytes) || !s.h_out.ensure(s.out_bytes) || !s.d_ws.ensure(wsb))
            return fail(p, PO_E_HIP, "po_pipeline_pair_decode: out of memory (wave buffers; lower wave_pairs / wave_rows)");

        // ---- pack: item arrays -> pinned staging (the GPU is busy with the previous wave meanwhile)
        const double tp = now_ms();
        char* hin = (char*)s.h_in.p;
        pack_items(y1_h, rows1, first, wn, row_in, hin, off1, p->threads);
        pack_items(y2_h, rows2, first, wn, row_in, hin + in1, off2, p->threads);
        p->pack_ms += now_ms() - tp;

        // ---- upload + ingest + decode + download, all on the slot's stream
        double* dy1 = (double*)s.d_y.p;
        double* dy2 = (double*)((char*)s.d_y.p + yb1);
        PCHK(hipMemcpyAsync(s.d_off.p, s.h_off.p, sizeof(int64_t) * n_off, hipMemcpyHostToDevice, s.st));
        const int64_t* d_o1 = (const int64_t*)s.d_off.p;
        const int64_t* d_o2 = d_o1 + (wn + 1);
        const int64_t* d_s1o = d_o2 + (wn + 1);
        const int64_t* d_so = d_s1o + (2 * wn + 1);
        if (direct) {
            PCHK(hipMemcpyAsync(dy1, hin, row_in * (size_t)r1, hipMemcpyHostToDevice, s.st));
            PCHK(hipMemcpyAsync(dy2, hin + in1, row_in * (size_t)r2, hipMemcpyHostToDevice, s.st));
        } else {
            char* din = (char*)s.d_in.p;
            PCHK(hipMemcpyAsync(din, hin, row_in * (size_t)r1, hipMemcpyHostToDevice, s.st));
            PCHK(hipMemcpyAsync(din + in1, hin + in1, row_in * (size_t)r2, hipMemcpyHostToDevice, s.st));
            rc = po_launch_ingest(din, d_o1, wn, C, in_mode, perm1, 0, r1, dy1, s.st);
            if (rc == PO_OK) rc = po_launch_ingest(din + in1, d_o2, wn, C, in_mode, perm2, reverse2, r2, dy2, s.st);
            if (rc != PO_OK) return fail(p, rc, "po_pipeline_pair_decode: bad permutation / input mode");
        }
        char* dout = (char*)s.d_out.p;
        rc = po_launch_pair_decode_geom(dy1, d_o1, dy2, d_o2, wn, C, opt, r1, r2, m1, m2, nullptr, nullptr, dout + s.o_seq1d, d_s1o,
                                        (int32_t*)(dout + s.o_l1), (int32_t*)(dout + s.o_l2), (double*)(dout + s.o_id),
                                        (int32_t*)(dout + s.o_env), dout + s.o_seq, d_so, (int32_t*)(dout + s.o_len),
                                        (int32_t*)(dout + s.o_st), s.d_ws.p, s.d_ws.cap, s.st);
        if (rc != PO_OK) return fail(p, rc, "po_pipeline_pair_decode: launch refused (unsupported C / model / options)");
        PCHK(hipGetLastError());
        // results: everything but the envelope in one copy; the envelope only when asked for
        PCHK(hipMemcpyAsync(s.h_out.p, s.d_out.p, s.o_env, hipMemcpyDeviceToHost, s.st));
        if (env_out_h)
            PCHK(hipMemcpyAsync((char*)s.h_out.p + s.o_env, dout + s.o_env, sizeof(int32_t) * 2 * (size_t)r1, hipMemcpyDeviceToHost, s.st));
        s.busy = true;
        first += wn;
        ++wave;
    }
    p->waves = wave;
    int rc = drain(p->slot[wave & 1]);          // the older wave first, then the last one
    if (rc == PO_OK) rc = drain(p->slot[(wave + 1) & 1]);
    p->total_ms = now_ms() - t_begin;
    return rc;
}

}  // extern "C"
