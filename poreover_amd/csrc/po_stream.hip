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
// other stream, and the tail of one wave's persistent kernels overlaps the head of the next.  The working set is bounded
// by the wave size whatever n is.
//
// Round 5 — what the FIRST call of a process pays (the reference's CLI is one process per job, pair_decode.py:230-303).
// Rounds 3 - 4 staged a whole wave in pinned memory per slot: 1.3 GB of hipHostMalloc for the 10 000-pair job, ~ 0.14 ms
// per MB — 0.33 s of a 0.55 s first call, paid again whenever a wave outgrew its slot (profiles/r05_cold_start.txt).  Now:
//   * inputs go through a small fixed RING of pinned chunks (4 x 16 MB, allocated when the pipeline is created): host
//     threads pack chunk k + 1 while chunk k is on the wire, so a wave's upload also starts with its first chunk, not
//     after its last item has been packed;
//   * the pinned result buffer holds what is copied back (texts and lengths; the envelopes only when asked for);
//   * device buffers are sized ONCE per call, from the plan of its waves, before the first wave — nothing is freed and
//     re-allocated mid-job (hipFree waits for the device).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
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
        static const bool trace = getenv("PO_PIPE_TRACE") != nullptr;
        const double t0 = trace ? now_ms() : 0.0;
        release();
        const size_t want = al256(bytes + bytes / 8);   // head-room: waves differ a little in size
        const hipError_t e = host ? hipHostMalloc(&p, want, hipHostMallocDefault) : hipMalloc(&p, want);
        if (trace) fprintf(stderr, "[po_pipe] %s %.1f MB: %.2f ms\n", host ? "hipHostMalloc" : "hipMalloc", want / 1048576.0, now_ms() - t0);
        if (e != hipSuccess) { p = nullptr; return false; }
        cap = want;
        return true;
    }
};

// The pinned chunks every wave's input goes through (one ring per pipeline, allocated once).
struct StageRing {
    static constexpr int N = 4;
    size_t chunk = 0;
    char* buf[N] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[N] = {nullptr, nullptr, nullptr, nullptr};
    bool used[N] = {false, false, false, false};
    int next = 0;
    bool create(size_t chunk_bytes) {
        static const bool trace = getenv("PO_PIPE_TRACE") != nullptr;
        const double t0 = trace ? now_ms() : 0.0;
        chunk = chunk_bytes;
        for (int i = 0; i < N; ++i) {
            if (hipHostMalloc((void**)&buf[i], chunk, hipHostMallocDefault) != hipSuccess) { buf[i] = nullptr; return false; }
            if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) { ev[i] = nullptr; return false; }
        }
        if (trace) fprintf(stderr, "[po_pipe] staging ring %d x %.0f MB pinned: %.2f ms\n", N, chunk / 1048576.0, now_ms() - t0);
        return true;
    }
    void destroy() {
        for (int i = 0; i < N; ++i) {
            if (ev[i]) (void)hipEventDestroy(ev[i]);
            if (buf[i]) (void)hipHostFree(buf[i]);
            ev[i] = nullptr; buf[i] = nullptr; used[i] = false;
        }
    }
};

// Host threads that pack chunks: started once per pipeline, woken per chunk (a std::thread per chunk and worker would cost
// more than the copy of a small chunk).  run(f) calls f(0 .. n - 1), f(0) on the calling thread.
class WorkerPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    const std::function<void(int)>* job = nullptr;
    unsigned gen = 0;
    int pending = 0;
    bool stop = false;
    void loop(int i) {
        unsigned seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv_go.wait(lk, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            const std::function<void(int)>* f = job;
            lk.unlock();
            (*f)(i);
            lk.lock();
            if (--pending == 0) cv_done.notify_one();
        }
    }
  public:
    const int n;
    explicit WorkerPool(int n_) : n(std::max(1, n_)) {
        for (int i = 1; i < n; ++i) th.emplace_back([this, i] { loop(i); });
    }
    ~WorkerPool() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_go.notify_all();
        for (auto& t : th) t.join();
    }
    void run(const std::function<void(int)>& f) {
        if (n == 1) { f(0); return; }
        { std::lock_guard<std::mutex> lk(mu); job = &f; pending = n - 1; ++gen; }
        cv_go.notify_all();
        f(0);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
};

struct Slot {
    hipStream_t st = nullptr;
    GrowBuf h_off{true}, h_out{true};
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

// Hands out the waves of one call, in input order: pairs [first, first + wn) with their row totals and maxima.  One
// pipeline walks it alone; the pipelines of several devices share it (each takes the next wave when it has a free slot),
// which balances pairs of uneven length across the devices without a plan made in advance.
struct WavePlanner {
    const int64_t *rows1, *rows2;
    int n, wave_pairs;
    int64_t wave_rows;
    int next = 0;
    int handed = 0;      // waves handed out so far
    int ramp = 0;        // > 0: the first waves are short (ramp, 2 * ramp, ... up to wave_pairs): see PO_WAVE_RAMP
    int tail = 0;        // > 0: the LAST wave has at most this many pairs (what follows it is serial: its download, the
                         // copy into the caller's arrays, the caller's own work on its results): see PO_WAVE_TAIL
    bool bad = false;
    std::mutex mu;
    bool take(int* first, int* wn, int64_t* r1, int64_t* r2, int64_t* m1, int64_t* m2) {
        std::lock_guard<std::mutex> lk(mu);
        if (bad || next >= n) return false;
        int limit = wave_pairs;
        if (ramp > 0 && handed < 8) limit = std::min<long long>(wave_pairs, (long long)ramp << handed);
        ++handed;
        {
            const int remaining = n - next;
            if (tail > 0 && remaining > tail && remaining <= limit) limit = remaining - tail;
        }
        int k = 0;
        int64_t a1 = 0, a2 = 0, x1 = 0, x2 = 0;
        while (next + k < n && k < limit) {
            const int64_t a = rows1[next + k], b = rows2[next + k];
            if (a < 0 || b < 0) { bad = true; return false; }
            if (k > 0 && a1 + a2 + a + b > wave_rows) break;
            a1 += a; a2 += b; x1 = std::max(x1, a); x2 = std::max(x2, b);
            ++k;
        }
        *first = next; *wn = k; *r1 = a1; *r2 = a2; *m1 = x1; *m2 = x2;
        next += k;
        return true;
    }
};

constexpr int PO_MAX_SLOTS = 8;
// Wave size when the caller names none.  Measured on the 10 000-pair job (scripts/e2e_ab.sh, one MI355X, host float32 in ->
// strings out): three waves in flight and four even waves of 2 500 pairs 66.5k pairs/s; 2 x 4 096 + 1 808 on two slots 58.4k;
// 3 x 3 334 63.4k; 5 x 2 048 (the LDS-ring kernel's range) 50.8k; a short first wave (ramp) 54 - 59k.  The first wave is what
// the device waits for (pack + upload: 22 ms at 2 500 pairs, 35 ms at 4 096), even waves keep the last one from being a
// tail, and a third slot lets wave k + 2 upload while k decodes and k + 1 waits.  A job of at most 4 096 pairs stays one wave.
// Round 4 (beam2d_reg_kernel: 16 pairs per CU, a launch of 3 334 pairs decodes at ~ 110k pairs/s where 2 500 pairs reach
// ~ 85k), same job, same box: waves of 2 500 73.9k pairs/s end to end, 3 334 82.1k, 5 000 on two or three slots 75k, 2 000 x 5
// 69k, four slots 68k; and now a SHORT FIRST WAVE pays (1 250 pairs, then 2 500, then full waves: the device starts ~ 10 ms
// earlier and the short wave no longer decodes much slower per pair): 3 334 with that ramp 84.4k (2 500 with it 77.6k).
// ... and a SHORT LAST WAVE (round 4, after the pair beam got faster: 10 000 pairs resident 67 ms): what follows the last wave
// is serial — its download, the copy into the caller's arrays, the Python records of its pairs (~ 4.7 us each: 15 ms for
// 3 334 pairs).  The last wave is cut to 800 pairs (the rest of it goes before): 90.9k -> 98.4k pairs/s end to end (tail
// 500: 90.6k, 1 250: 75 - 84k; waves of 4 000 / 3 000 with the tail 87.8k / 86.3k).
constexpr int PO_WAVE_TARGET = 3334, PO_WAVE_MAX = 4096, PO_WAVE_RAMP = 1250, PO_WAVE_TAIL = 800;
static int auto_wave_pairs(int n) {
    if (n <= PO_WAVE_MAX) return PO_WAVE_MAX;
    const int waves = (n + PO_WAVE_TARGET - 1) / PO_WAVE_TARGET;
    return (n + waves - 1) / waves;
}
// a single pipeline's short first waves and short last wave (one place: po_pipeline_pair_decode runs this plan, po_wave_plan
// reports it; PO_WAVE_RAMP / PO_WAVE_TAIL override either, 0 = off)
static void ramp_and_tail(bool engine_plans, int n, int* ramp, int* tail) {
    *ramp = (engine_plans && n > PO_WAVE_MAX) ? PO_WAVE_RAMP : 0;
    *tail = (engine_plans && n > PO_WAVE_MAX) ? PO_WAVE_TAIL : 0;
    if (const char* e = getenv("PO_WAVE_RAMP")) *ramp = std::max(0, atoi(e));
    if (const char* e = getenv("PO_WAVE_TAIL")) *tail = std::max(0, atoi(e));
}
struct po_pipeline {
    int device = 0, wave_pairs = 0 /* 0: auto_wave_pairs(n) */, threads = 8;
    int64_t wave_rows = (int64_t)64 << 20;
    Slot slot[PO_MAX_SLOTS];
    StageRing ring;
    WorkerPool* pool = nullptr;
    int nslots = 3;   // waves in flight: one decoding, the next ones packed / uploading behind it (PO_PIPELINE_SLOTS)
    double pack_ms = 0, wait_ms = 0, total_ms = 0;
    int waves = 0, pairs = 0;
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

// Bytes [lo, hi) of a wave's input region (the items' rows back to back: item i at off[i] * row_bytes) -> dst[0, hi - lo),
// split over the pool's threads by bytes; items are cut where the range cuts them.
void pack_bytes(WorkerPool& pool, const void* const* src, int first, int n, size_t row_bytes, const std::vector<int64_t>& off,
                size_t lo, size_t hi, char* dst) {
    if (hi <= lo || n <= 0) return;
    const int nt = (hi - lo < ((size_t)1 << 20)) ? 1 : pool.n;
    const std::function<void(int)> work = [&](int t) {
        if (t >= nt) return;
        size_t a = lo + (hi - lo) * (size_t)t / (size_t)nt;
        const size_t b = lo + (hi - lo) * (size_t)(t + 1) / (size_t)nt;
        if (b <= a) return;
        // the item that holds byte a: the last one starting at or before it
        int i = (int)(std::upper_bound(off.begin(), off.begin() + n + 1, (int64_t)(a / row_bytes)) - off.begin()) - 1;
        if (i < 0) i = 0;
        while (a < b && i < n) {
            const size_t i0 = (size_t)off[i] * row_bytes, i1 = (size_t)off[i + 1] * row_bytes;
            if (a >= i1) { ++i; continue; }
            const size_t e = std::min(b, i1);
            std::memcpy(dst + (a - lo), (const char*)src[first + i] + (a - i0), e - a);
            a = e;
            if (a >= i1) ++i;
        }
    };
    if (nt == 1) work(0);
    else pool.run(work);
}
}  // namespace

namespace {
// binds the calling thread to `device` for a scope and gives it back the device it had
struct DeviceScope {
    int prev = -1;
    bool ok = true;
    explicit DeviceScope(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev == device) prev = -1;
        else ok = (hipSetDevice(device) == hipSuccess);
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};
}  // namespace

extern "C" {

void po_pipeline_destroy(po_pipeline* p);
po_pipeline* po_pipeline_create(int device, int wave_pairs, int64_t wave_rows, int threads) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { po_set_error("po_pipeline_create: no such device"); return nullptr; }
    // the caller's thread keeps the device it had: a pipeline's own work sets its device where it runs
    DeviceScope bind(device);
    if (!bind.ok) { po_set_error("po_pipeline_create: hipSetDevice failed"); return nullptr; }
    po_pipeline* p = new po_pipeline();
    p->device = device;
    if (wave_pairs > 0) p->wave_pairs = wave_pairs;
    else if (const char* e = getenv("PO_WAVE_PAIRS")) { const int v = atoi(e); if (v > 0) p->wave_pairs = v; }
    if (wave_rows > 0) p->wave_rows = wave_rows;
    if (const char* e = getenv("PO_PIPELINE_SLOTS")) { const int v = atoi(e); if (v >= 2 && v <= PO_MAX_SLOTS) p->nslots = v; }
    const unsigned hc = std::thread::hardware_concurrency();
    p->threads = threads > 0 ? threads : (int)std::max(1u, std::min(16u, hc ? hc / 2 : 4u));
    // (only the slots in use get a stream: HIP maps streams onto a handful of hardware queues — four by default — and two streams
    //  on one queue run one after the other; a process that has made other streams before, as bench.py's two-streams leg does,
    //  leaves fewer free ones)
    for (int k = 0; k < p->nslots; ++k)
        if (Slot& s = p->slot[k]; hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking) != hipSuccess) {
            po_set_error("po_pipeline_create: hipStreamCreate failed");
            po_pipeline_destroy(p);
            return nullptr;
        }
    size_t chunk_mb = 16;
    if (const char* e = getenv("PO_STAGE_MB")) { const int v = atoi(e); if (v >= 1 && v <= 1024) chunk_mb = (size_t)v; }
    if (!p->ring.create(chunk_mb << 20)) {
        po_set_error("po_pipeline_create: hipHostMalloc of the staging ring failed");
        po_pipeline_destroy(p);
        return nullptr;
    }
    p->pool = new WorkerPool(p->threads);
    return p;
}

void po_pipeline_destroy(po_pipeline* p) {
    if (!p) return;
    DeviceScope bind(p->device);
    for (auto& s : p->slot) {
        if (s.st) { (void)hipStreamSynchronize(s.st); (void)hipStreamDestroy(s.st); }
    }
    p->ring.destroy();
    delete p->pool;
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

}  // extern "C"

// the arguments of one call, as po_pipeline_pair_decode receives them
struct PairCall {
    const void* const* y1_h; const int64_t* rows1; const void* const* y2_h; const int64_t* rows2;
    int n, C, in_mode; const int* perm1; const int* perm2; int reverse2; const po_pair_options* opt;
    char* seq1d_h; const int64_t* seq1d_off_h; int32_t* len1_h; int32_t* len2_h; double* identity_h; int32_t* env_out_h;
    char* seq_h; const int64_t* seq_off_h; int32_t* seq_len_h; int32_t* status_h;
    const int64_t* env_row0;   // global row offset of every pair's envelope in env_out_h (NULL without env_out_h)
};

// One pipeline (one device, the calling thread) decodes the waves the planner hands it, two in flight.  Whatever way the
// call ends, no slot is left marked busy: a failed call's results are discarded, never drained into a later call's arrays.
// The pipelines of several devices start taking waves TOGETHER, after each has sized its buffers (the first call allocates: a
// pipeline that is ready early would otherwise have dealt itself most of a short job before the others are through).
struct StartGate {
    std::atomic<int> arrived{0};
    int n = 1;
};
static int pipeline_run(po_pipeline* p, WavePlanner& plan, const PairCall& c, StartGate* gate = nullptr) {
    struct Arrive {   // (every exit before the gate counts as arrived: nobody waits for a pipeline that has failed)
        StartGate* g;
        bool done = false;
        void now(bool wait) {
            if (!g || done) return;
            done = true;
            g->arrived.fetch_add(1);
            while (wait && g->arrived.load() < g->n) std::this_thread::yield();
        }
        ~Arrive() { now(false); }
    } arrive{gate};
    struct Quiesce {   // entry and every exit: both streams drained, both slots free
        po_pipeline* p;
        void run() {
            for (auto& s : p->slot) {
                if (s.busy) (void)hipStreamSynchronize(s.st);
                s.busy = false;
            }
            for (int k = 0; k < StageRing::N; ++k) {   // (a failed call may leave chunks on the wire: their streams are drained above)
                if (p->ring.used[k]) (void)hipEventSynchronize(p->ring.ev[k]);
                p->ring.used[k] = false;
            }
        }
        explicit Quiesce(po_pipeline* p_) : p(p_) { run(); }
        ~Quiesce() { run(); }
    } quiesce(p);
    const int C = c.C, in_mode = c.in_mode;
    const size_t esz = in_mode == PO_INGEST_LOGITS_F32 ? 4 : (in_mode == PO_INGEST_TRACE_U8 ? 1 : 8);
    const size_t row_in = esz * (size_t)C, row_y = sizeof(double) * (size_t)C;
    const double t_begin = now_ms();
    p->pack_ms = p->wait_ms = 0;
    p->waves = 0;
    p->pairs = 0;
    static const bool trace = getenv("PO_PIPE_TRACE") != nullptr;
    const bool direct = (in_mode == PO_INGEST_F64 && !c.perm1 && !c.perm2 && !c.reverse2);   // log-probabilities as they are

    // ---- every buffer at its largest, before the first wave: the plan of the call's waves is known (the planner is
    // deterministic), so nothing is freed and re-allocated while waves are in flight
    auto out_layout = [&](int wn, int64_t r1, int64_t r2, Slot* s) -> size_t {
        size_t o = 0, o_seq1d, o_seq, o_l1, o_l2, o_len, o_st, o_id, o_env;
        o_seq1d = o; o += al256((size_t)(r1 + r2) + 16);
        o_seq = o; o += al256((size_t)(r1 + r2) + 16);
        o_l1 = o; o += al256(sizeof(int32_t) * wn);
        o_l2 = o; o += al256(sizeof(int32_t) * wn);
        o_len = o; o += al256(sizeof(int32_t) * wn);
        o_st = o; o += al256(sizeof(int32_t) * wn);
        o_id = o; o += al256(sizeof(double) * wn);
        o_env = o; o += al256(sizeof(int32_t) * 2 * (size_t)r1);
        if (s) { s->o_seq1d = o_seq1d; s->o_seq = o_seq; s->o_l1 = o_l1; s->o_l2 = o_l2; s->o_len = o_len; s->o_st = o_st; s->o_id = o_id; s->o_env = o_env; s->out_bytes = o; }
        return o;
    };
    {
        WavePlanner sim;
        sim.rows1 = plan.rows1; sim.rows2 = plan.rows2; sim.n = plan.n; sim.wave_pairs = plan.wave_pairs; sim.wave_rows = plan.wave_rows;
        sim.ramp = plan.ramp; sim.tail = plan.tail;
        size_t mx_in = 0, mx_y = 0, mx_off = 0, mx_out = 0, mx_hout = 0, mx_ws = 0;
        int f = 0, wn = 0, nw = 0;
        int64_t r1 = 0, r2 = 0, m1 = 0, m2 = 0;
        while (sim.take(&f, &wn, &r1, &r2, &m1, &m2)) {
            ++nw;
            mx_in = std::max(mx_in, al256(row_in * (size_t)r1) + al256(row_in * (size_t)r2));
            mx_y = std::max(mx_y, al256(row_y * (size_t)r1) + al256(row_y * (size_t)r2));
            mx_off = std::max(mx_off, sizeof(int64_t) * ((size_t)(wn + 1) * 3 + (size_t)(2 * wn + 1)));
            const size_t ob = out_layout(wn, r1, r2, nullptr);
            mx_out = std::max(mx_out, ob);
            mx_hout = std::max(mx_hout, c.env_out_h ? ob : ob - al256(sizeof(int32_t) * 2 * (size_t)r1));
            mx_ws = std::max(mx_ws, po_pair_ws_bytes_impl(wn, r1, r2, m1, m2, C, c.opt));
        }
        const double t0 = now_ms();
        for (int k = 0; k < std::min(nw, p->nslots); ++k) {
            Slot& s = p->slot[k];
            if (!s.h_off.ensure(mx_off) || !s.d_off.ensure(mx_off) || !s.d_y.ensure(mx_y) || (!direct && !s.d_in.ensure(mx_in)) ||
                !s.d_out.ensure(mx_out) || !s.h_out.ensure(mx_hout) || !s.d_ws.ensure(mx_ws))
                return fail(p, PO_E_HIP, "po_pipeline_pair_decode: out of memory (wave buffers; lower wave_pairs / wave_rows)");
        }
        if (trace) fprintf(stderr, "[po_pipe] %d wave(s) planned, buffers sized in %.2f ms\n", nw, now_ms() - t0);
    }
    arrive.now(true);

    // results of the wave a slot holds -> the caller's arrays (after the slot's stream has drained)
    auto drain = [&](Slot& s) -> int {
        if (!s.busy) return PO_OK;
        const double t0 = now_ms();
        PCHK(hipStreamSynchronize(s.st));
        p->wait_ms += now_ms() - t0;
        if (trace) fprintf(stderr, "[po_pipe] wave of %d pairs (first %d) done at %.2f ms (waited %.2f ms)\n", s.n, s.first, now_ms() - t_begin, now_ms() - t0);
        const char* ho = (const char*)s.h_out.p;
        const int32_t* l1 = (const int32_t*)(ho + s.o_l1);
        const int32_t* l2 = (const int32_t*)(ho + s.o_l2);
        const int32_t* ln = (const int32_t*)(ho + s.o_len);
        const int32_t* st = (const int32_t*)(ho + s.o_st);
        const double* idn = (const double*)(ho + s.o_id);
        const int64_t* o1 = (const int64_t*)s.h_off.p;
        for (int i = 0; i < s.n; ++i) {
            const int g = s.first + i;
            c.len1_h[g] = l1[i]; c.len2_h[g] = l2[i]; c.seq_len_h[g] = ln[i]; c.identity_h[g] = idn[i];
            const int64_t c1 = c.seq1d_off_h[2 * g + 1] - c.seq1d_off_h[2 * g], c2 = c.seq1d_off_h[2 * g + 2] - c.seq1d_off_h[2 * g + 1];
            const int64_t cc = c.seq_off_h[g + 1] - c.seq_off_h[g];
            int code = st[i];
            if (l1[i] > c1 || l2[i] > c2 || ln[i] > cc) { code = PO_E_CAP; c.seq_len_h[g] = 0; }
            else {
                std::memcpy(c.seq1d_h + c.seq1d_off_h[2 * g], ho + s.o_seq1d + s.s1o[2 * i], (size_t)std::max(0, l1[i]));
                std::memcpy(c.seq1d_h + c.seq1d_off_h[2 * g + 1], ho + s.o_seq1d + s.s1o[2 * i + 1], (size_t)std::max(0, l2[i]));
                std::memcpy(c.seq_h + c.seq_off_h[g], ho + s.o_seq + s.so[i], (size_t)std::max(0, ln[i]));
                if (c.env_out_h)
                    std::memcpy(c.env_out_h + 2 * c.env_row0[g], ho + s.o_env + sizeof(int32_t) * 2 * (size_t)o1[i],
                                sizeof(int32_t) * 2 * (size_t)c.rows1[g]);
            }
            // the status is written LAST: a caller that watches status_h from another thread (it filled it with a value no
            // decode returns) may read a pair's outputs as soon as its status has changed
            std::atomic_thread_fence(std::memory_order_release);
            ((volatile int32_t*)c.status_h)[g] = code;
        }
        s.busy = false;
        return PO_OK;
    };

    int wave = 0;
    int first = 0, wn = 0;
    int64_t r1 = 0, r2 = 0, m1 = 0, m2 = 0;
    for (;;) {
        Slot& s = p->slot[wave % p->nslots];
        int rc = drain(s);   // the wave this slot ran two waves ago (before asking for another: a device takes work when it can start it)
        if (rc != PO_OK) return rc;
        if (!plan.take(&first, &wn, &r1, &r2, &m1, &m2)) break;
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
            const int64_t a = c.rows1[first + i], b = c.rows2[first + i];
            o1[i + 1] = o1[i] + a; o2[i + 1] = o2[i] + b;
            s1o[2 * i + 1] = s1o[2 * i] + a; s1o[2 * i + 2] = s1o[2 * i + 1] + b;   // a basecall has at most one base per frame
            so[i + 1] = so[i] + a + b;
        }
        s.s1o.assign(s1o, s1o + 2 * wn + 1);
        s.so.assign(so, so + wn + 1);
        std::vector<int64_t> off1(o1, o1 + wn + 1), off2(o2, o2 + wn + 1);

        // ---- buffers (sized before the first wave; ensure() only acts if a wave outgrows the plan)
        const size_t in1 = al256(row_in * (size_t)r1), in2 = al256(row_in * (size_t)r2);
        const size_t yb1 = al256(row_y * (size_t)r1), yb2 = al256(row_y * (size_t)r2);
        out_layout(wn, r1, r2, &s);
        const size_t wsb = po_pair_ws_bytes_impl(wn, r1, r2, m1, m2, C, c.opt);
        if (!s.d_y.ensure(yb1 + yb2) || (!direct && !s.d_in.ensure(in1 + in2)) || !s.d_out.ensure(s.out_bytes) ||
            !s.h_out.ensure(c.env_out_h ? s.out_bytes : s.o_env) || !s.d_ws.ensure(wsb))
            return fail(p, PO_E_HIP, "po_pipeline_pair_decode: out of memory (wave buffers; lower wave_pairs / wave_rows)");

        // ---- upload through the ring of pinned chunks: chunk k + 1 is packed (host threads) while chunk k is on the wire;
        // then ingest + decode + download, all on the slot's stream
        double* dy1 = (double*)s.d_y.p;
        double* dy2 = (double*)((char*)s.d_y.p + yb1);
        PCHK(hipMemcpyAsync(s.d_off.p, s.h_off.p, sizeof(int64_t) * n_off, hipMemcpyHostToDevice, s.st));
        const int64_t* d_o1 = (const int64_t*)s.d_off.p;
        const int64_t* d_o2 = d_o1 + (wn + 1);
        const int64_t* d_s1o = d_o2 + (wn + 1);
        const int64_t* d_so = d_s1o + (2 * wn + 1);
        char* din = direct ? nullptr : (char*)s.d_in.p;
        {
            StageRing& rg = p->ring;
            for (int region = 0; region < 2; ++region) {
                const size_t bytes = row_in * (size_t)(region ? r2 : r1);
                char* dst = direct ? (char*)(region ? dy2 : dy1) : din + (region ? in1 : 0);
                const void* const* src = region ? c.y2_h : c.y1_h;
                const std::vector<int64_t>& off = region ? off2 : off1;
                for (size_t lo = 0; lo < bytes; lo += rg.chunk) {
                    const size_t hi = std::min(bytes, lo + rg.chunk);
                    const int k = rg.next;
                    rg.next = (k + 1) % StageRing::N;
                    if (rg.used[k]) {   // (the copy that last used this chunk: long done unless the link is the bottleneck)
                        const double tw = now_ms();
                        PCHK(hipEventSynchronize(rg.ev[k]));
                        p->wait_ms += now_ms() - tw;
                    }
                    const double tp = now_ms();
                    pack_bytes(*p->pool, src, first, wn, row_in, off, lo, hi, rg.buf[k]);
                    p->pack_ms += now_ms() - tp;
                    PCHK(hipMemcpyAsync(dst + lo, rg.buf[k], hi - lo, hipMemcpyHostToDevice, s.st));
                    PCHK(hipEventRecord(rg.ev[k], s.st));
                    rg.used[k] = true;
                }
            }
        }
        if (trace) fprintf(stderr, "[po_pipe] wave %d: %d pairs, uploaded (enqueued) at %.2f ms\n", wave, wn, now_ms() - t_begin);
        if (!direct) {
            rc = po_launch_ingest(din, d_o1, wn, C, in_mode, c.perm1, 0, r1, dy1, s.st);
            if (rc == PO_OK) rc = po_launch_ingest(din + in1, d_o2, wn, C, in_mode, c.perm2, c.reverse2, r2, dy2, s.st);
            if (rc != PO_OK) return fail(p, rc, "po_pipeline_pair_decode: bad permutation / input mode");
        }
        char* dout = (char*)s.d_out.p;
        rc = po_launch_pair_decode_geom(dy1, d_o1, dy2, d_o2, wn, C, c.opt, r1, r2, m1, m2, nullptr, nullptr, dout + s.o_seq1d, d_s1o,
                                        (int32_t*)(dout + s.o_l1), (int32_t*)(dout + s.o_l2), (double*)(dout + s.o_id),
                                        (int32_t*)(dout + s.o_env), dout + s.o_seq, d_so, (int32_t*)(dout + s.o_len),
                                        (int32_t*)(dout + s.o_st), s.d_ws.p, s.d_ws.cap, s.st);
        if (rc != PO_OK) return fail(p, rc, "po_pipeline_pair_decode: launch refused (unsupported C / model / options)");
        PCHK(hipGetLastError());
        // results: everything but the envelope in one copy; the envelope only when asked for
        PCHK(hipMemcpyAsync(s.h_out.p, s.d_out.p, s.o_env, hipMemcpyDeviceToHost, s.st));
        if (c.env_out_h)
            PCHK(hipMemcpyAsync((char*)s.h_out.p + s.o_env, dout + s.o_env, sizeof(int32_t) * 2 * (size_t)r1, hipMemcpyDeviceToHost, s.st));
        s.busy = true;
        p->pairs += wn;
        ++wave;
    }
    if (plan.bad) return fail(p, PO_E_ARG, "po_pipeline_pair_decode: negative row count");
    p->waves = wave;
    int rc = PO_OK;
    for (int k = 0; k < p->nslots && rc == PO_OK; ++k) rc = drain(p->slot[(wave + k) % p->nslots]);   // oldest wave first
    p->total_ms = now_ms() - t_begin;
    return rc;
}

static int check_call(const PairCall& c, std::string* why) {
    if (c.n < 0 || !c.y1_h || !c.rows1 || !c.y2_h || !c.rows2 || !c.opt || !c.seq1d_h || !c.seq1d_off_h || !c.len1_h || !c.len2_h ||
        !c.identity_h || !c.seq_h || !c.seq_off_h || !c.seq_len_h || !c.status_h) { *why = "null argument"; return PO_E_ARG; }
    if (c.C < 1 || c.C > 8 || c.in_mode < 0 || c.in_mode > 2) { *why = "bad C / input mode"; return PO_E_ARG; }
    return PO_OK;
}

extern "C" {

int po_pipeline_pair_decode(po_pipeline* p, const void* const* y1_h, const int64_t* rows1, const void* const* y2_h,
                            const int64_t* rows2, int n, int C, int in_mode, const int* perm1, const int* perm2,
                            int reverse2, const po_pair_options* opt, char* seq1d_h, const int64_t* seq1d_off_h,
                            int32_t* len1_h, int32_t* len2_h, double* identity_h, int32_t* env_out_h, char* seq_h,
                            const int64_t* seq_off_h, int32_t* seq_len_h, int32_t* status_h) {
    if (!p) return PO_E_ARG;
    PairCall c{y1_h, rows1, y2_h, rows2, n, C, in_mode, perm1, perm2, reverse2, opt, seq1d_h, seq1d_off_h, len1_h, len2_h,
               identity_h, env_out_h, seq_h, seq_off_h, seq_len_h, status_h, nullptr};
    std::string why;
    if (check_call(c, &why) != PO_OK) return fail(p, PO_E_ARG, "po_pipeline_pair_decode: " + why);
    DeviceScope bind(p->device);   // (the caller's thread gets its own device back on every exit)
    if (!bind.ok) return fail(p, PO_E_HIP, "po_pipeline_pair_decode: hipSetDevice failed");
    std::vector<int64_t> env_row0;
    if (env_out_h) {
        env_row0.resize((size_t)n + 1, 0);
        for (int i = 0; i < n; ++i) env_row0[i + 1] = env_row0[i] + rows1[i];
        c.env_row0 = env_row0.data();
    }
    WavePlanner plan;
    plan.rows1 = rows1; plan.rows2 = rows2; plan.n = n; plan.wave_rows = p->wave_rows;
    plan.wave_pairs = p->wave_pairs > 0 ? p->wave_pairs : auto_wave_pairs(n);
    ramp_and_tail(p->wave_pairs <= 0, n, &plan.ramp, &plan.tail);   // a job of several waves starts with a short one, doubling, and ends with one
    // (Tried: a job within the latency-bound regime of the pair beam kernel — <= 2 048 pairs, one GPU's share of a multi-GPU
    //  job — cut into two waves whose kernels run side by side while the second uploads: 36.8 vs 36.9 ms for 1 250 pairs,
    //  three waves on the two slots 60 ms.  The upload is not what such a job waits for.  PO_PIPELINE_SPLIT=1 still does it.)
    {
        static const int split_env = [] { const char* e = getenv("PO_PIPELINE_SPLIT"); return e ? atoi(e) : 0; }();
        if (split_env > 0 && n >= 4) plan.wave_pairs = std::max(1, std::min(plan.wave_pairs, (n + 1) / 2));
    }
    return pipeline_run(p, plan, c);
}

// ---- several devices, one process -------------------------------------------------------------------------------
// The native replacement of the reference's fan-out over worker processes (pair_decode.py:292-297) on a multi-GPU node:
// one po_pipeline per device, each driven by its own host thread, all taking waves from one planner and writing their
// results straight into the caller's arrays at the pairs' own indices — input order, no gather, no inter-process copy.
struct po_multi {
    std::vector<po_pipeline*> pipes;
    std::vector<int> pairs_done;
    std::string err;
    int wave_pairs = 0;   // 0: auto_wave_pairs(n)
};

// The waves a call over `ndev` devices is cut into (no device is touched: what po_multi_pair_decode's planner hands out,
// in order): first[k], count[k] for k < returned number of waves (at most cap are written).
int po_wave_plan(const int64_t* rows1, const int64_t* rows2, int n, int wave_pairs, int64_t wave_rows, int ndev, int* first,
                 int* count, int cap) {
    if (!rows1 || !rows2 || n < 0 || ndev < 1) return PO_E_ARG;
    WavePlanner plan;
    plan.rows1 = rows1; plan.rows2 = rows2; plan.n = n;
    const int wp = wave_pairs > 0 ? wave_pairs : auto_wave_pairs(n);
    plan.wave_pairs = (ndev > 1) ? std::max(1, std::min(wp, (n + 2 * ndev - 1) / (2 * ndev))) : wp;
    plan.wave_rows = wave_rows > 0 ? wave_rows : ((int64_t)64 << 20);
    if (ndev == 1) ramp_and_tail(wave_pairs <= 0, n, &plan.ramp, &plan.tail);   // (po_pipeline_pair_decode's own plan)
    int k = 0, f = 0, c = 0;
    int64_t r1, r2, m1, m2;
    while (plan.take(&f, &c, &r1, &r2, &m1, &m2)) {
        if (k < cap) { if (first) first[k] = f; if (count) count[k] = c; }
        ++k;
    }
    return plan.bad ? PO_E_ARG : k;
}

po_multi* po_multi_create(const int* devices, int ndev, int wave_pairs, int64_t wave_rows, int threads) {
    if (!devices || ndev < 1) { po_set_error("po_multi_create: no devices"); return nullptr; }
    po_multi* m = new po_multi();
    if (wave_pairs > 0) m->wave_pairs = wave_pairs;
    for (int i = 0; i < ndev; ++i) {
        po_pipeline* p = po_pipeline_create(devices[i], wave_pairs, wave_rows, threads);
        if (!p) { for (auto* q : m->pipes) po_pipeline_destroy(q); delete m; return nullptr; }
        m->pipes.push_back(p);
    }
    m->pairs_done.assign((size_t)ndev, 0);
    if (wave_pairs <= 0) m->wave_pairs = m->pipes[0]->wave_pairs;   // (PO_WAVE_PAIRS, or 0 = auto)
    return m;
}

void po_multi_destroy(po_multi* m) {
    if (!m) return;
    for (auto* p : m->pipes) po_pipeline_destroy(p);
    delete m;
}

int po_multi_devices(po_multi* m) { return m ? (int)m->pipes.size() : 0; }

// pairs decoded by pipeline i in the last call, and its pack / wait / total milliseconds
int po_multi_stats(po_multi* m, int i, int* pairs, double* pack_ms, double* wait_ms, double* total_ms, int* waves) {
    if (!m || i < 0 || i >= (int)m->pipes.size()) return PO_E_ARG;
    if (pairs) *pairs = m->pipes[(size_t)i]->pairs;
    return po_pipeline_stats(m->pipes[(size_t)i], pack_ms, wait_ms, total_ms, waves);
}

int po_multi_pair_decode(po_multi* m, const void* const* y1_h, const int64_t* rows1, const void* const* y2_h,
                         const int64_t* rows2, int n, int C, int in_mode, const int* perm1, const int* perm2,
                         int reverse2, const po_pair_options* opt, char* seq1d_h, const int64_t* seq1d_off_h,
                         int32_t* len1_h, int32_t* len2_h, double* identity_h, int32_t* env_out_h, char* seq_h,
                         const int64_t* seq_off_h, int32_t* seq_len_h, int32_t* status_h) {
    if (!m || m->pipes.empty()) return PO_E_ARG;
    PairCall c{y1_h, rows1, y2_h, rows2, n, C, in_mode, perm1, perm2, reverse2, opt, seq1d_h, seq1d_off_h, len1_h, len2_h,
               identity_h, env_out_h, seq_h, seq_off_h, seq_len_h, status_h, nullptr};
    std::string why;
    if (check_call(c, &why) != PO_OK) { po_set_error(("po_multi_pair_decode: " + why).c_str()); return PO_E_ARG; }
    std::vector<int64_t> env_row0;
    if (env_out_h) {
        env_row0.resize((size_t)n + 1, 0);
        for (int i = 0; i < n; ++i) env_row0[i + 1] = env_row0[i] + rows1[i];
        c.env_row0 = env_row0.data();
    }
    const int nd = (int)m->pipes.size();
    // waves small enough that every device gets at least two (one decoding while the next uploads), never larger than
    // one pipeline's own wave size
    WavePlanner plan;
    plan.rows1 = rows1; plan.rows2 = rows2; plan.n = n;
    plan.wave_pairs = std::max(1, std::min(m->wave_pairs > 0 ? m->wave_pairs : auto_wave_pairs(n), (n + 2 * nd - 1) / (2 * nd)));
    plan.wave_rows = m->pipes[0]->wave_rows;
    std::vector<int> rcs((size_t)nd, PO_OK);
    std::vector<std::string> errs((size_t)nd);
    std::vector<std::thread> th;
    StartGate gate;
    gate.n = nd;
    for (int i = 0; i < nd; ++i)
        th.emplace_back([&, i]() {
            po_pipeline* p = m->pipes[(size_t)i];
            if (hipSetDevice(p->device) != hipSuccess) { rcs[(size_t)i] = PO_E_HIP; errs[(size_t)i] = "hipSetDevice failed"; gate.arrived.fetch_add(1); return; }
            rcs[(size_t)i] = pipeline_run(p, plan, c, &gate);
            if (rcs[(size_t)i] != PO_OK) {
                errs[(size_t)i] = p->err;
                std::lock_guard<std::mutex> lk(plan.mu);   // the other devices finish what they hold and stop taking waves
                plan.next = plan.n;
            }
        });
    for (auto& t : th) t.join();
    for (int i = 0; i < nd; ++i)
        if (rcs[(size_t)i] != PO_OK) {
            m->err = "device " + std::to_string(m->pipes[(size_t)i]->device) + ": " + errs[(size_t)i];
            po_set_error(m->err.c_str());
            return rcs[(size_t)i];
        }
    return PO_OK;
}

}  // extern "C"
