// Pair (2-D) CTC beam search, method "row_col" with an envelope, one-value model ("ctc"), W * (A + 1) <= 28:
// the element windows live in LDS rings, the HBM value store only takes what leaves them.
//
// Replaces (like beam2d_kernel, which stays the general form): decoding_cpp.cpp_beam_search_2d (decoding_cpp.pyx:107-139)
// -> beam_search_2d_by_row_col (BeamSearch.h:262-397) over PoreOverPrefixTree2D (PrefixTree.h:492-533) with
// Beam<..., node_greater_max_sym> (Beam.h:35-38,93-108).
//
// Why another kernel.  beam2d_kernel writes every update it executes as a 16-byte tagged entry into a ring store in
// HBM (8 GB for 4096 workgroups) because a pair's live window state did not fit the LDS at 16 workgroups per CU; its
// counters say 36 x the algorithmic bytes through the fabric and five of six vector instructions spent on
// bookkeeping (tags, 64-bit addresses, carried state through LDS tables, row-group marking) rather than on
// logaddexp.  Here one wave owns a pair with the whole per-element state in REGISTERS (the kernel runs 2 waves per
// SIMD, 256 VGPRs) and the window values of the <= 28 elements in an LDS ring of 32 times per (element, read):
//   * tier 1, LDS: ring[read][t & 31][row] — values of the times [hiw - 32, hiw) of the element that owns `row`;
//     lanes of one read touch one time and different rows: conflict-free.  Parents are read from the ring (the value a
//     parent computed in the previous iteration is simply there), so there is no exchange buffer.
//   * tier 2, HBM: the tagged ring store of beam2d_kernel, same addressing, same "absent reads as -inf" — but written
//     only when a value that can still be read leaves tier 1: EVICTION (a window wider than 32 times overwrites a slot
//     whose time is still inside the window) and SPILL (an element stops being an element: its parent left the beam).
//     Everything the reference's per-node maps would answer is in tier 1 or tier 2, so frozen parents, nodes that
//     re-enter the beam and windows hundreds of frames wide need no special cases — they are slower, not different.
//   * every element carries (lo, done, hiw): the times [lo, done) are computed in this incarnation, done is where
//     its window ended last; a main step computes [max(done, window start), window end) only (a recomputation of
//     anything else would rewrite the bits that are there: every input is unchanged), and the window maximum of the
//     rest is carried as (max, its time, last rise) exactly as in beam2d_kernel.
// The walk comes precomputed (beam2d_walk_kernel), the envelope checks, blank prefix sums and R from
// beam2d_prepass_kernel; pairs this kernel cannot hold (tier-2 row groups exhausted) go to beam2d_kernel through the
// same meta word the two-pairs-per-wave kernel uses.  Results are bit-identical to beam2d_kernel's: the same
// arithmetic in the same order within every chain.
#include <climits>

#include "po_beam2d_common.h"
#include "po_host.h"

namespace {

constexpr int RG_NRP = 26;    // LDS ring rows = elements that can be live at once
constexpr int RG_RL = 32;     // times per ring row
constexpr int RG_NY = 32;     // y rows per read resident in LDS
constexpr int RG_YC = 5;      // doubles per y row (A + 1 <= 5)
constexpr int RG_NGL = 96;   // tier-2 row groups tracked per pair
constexpr int RG_FRESH = INT_MIN / 2;

struct RingSmem {
    double ring[2][RG_RL][RG_NRP];
    double ybuf[2][RG_NY][RG_YC];
    int g_owner[RG_NGL], g_hi0[RG_NGL], g_hi1[RG_NGL];
    int ord[32];              // prune with exact score ties: candidate slots in node-id order (po_stl_prune)
    double csc[32];           // ... and their scores
    int sh[8];
    unsigned long long nupd, nupd_x;
    PoLaeTables lae;
};

__device__ __forceinline__ void rg_sync() { b2_sync_lds<64>(); }
__device__ __forceinline__ double rg_readlane_d(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}

}  // namespace

__global__ __launch_bounds__(64, 2) void beam2d_ring_kernel(X2Args a) {
    using Ent = Entry<1>;
    __shared__ RingSmem sm;
    const int lane = threadIdx.x, r = lane >> 5, s = lane & 31, hb = lane & 32;
    const int A = a.A, W = a.W, C = a.C;
    const int divA = (65536 + A - 1) / A;   // x / A == (x * divA) >> 16 for the slot numbers divided here
    Ent* const pool = (Ent*)(a.pool + (size_t)blockIdx.x * a.pool_bytes);
    const long long pool_entries = (long long)(a.pool_bytes / sizeof(Ent));
    int* const apl = a.arena + (size_t)blockIdx.x * 3 * a.arena_cap;
    int* const afc = apl + a.arena_cap;
    int* const acrow = afc + a.arena_cap;
    int* const g_hi = r ? sm.g_hi1 : sm.g_hi0;

    // ---- epoch tags across pairs and launches (as beam2d_kernel): no memset of the tier-2 store
    unsigned epoch = 0;
    auto clear_slice = [&]() {
        for (long long i = lane; i < pool_entries; i += 64) pool[i].tag = 0ull;
        __syncthreads();
    };
    {
        unsigned long long* stp = a.wgstate + 2 * (size_t)blockIdx.x;
        const unsigned long long w0 = stp[0], w1 = stp[1];
        const bool ok = (w0 == (a.magic ^ (unsigned long long)blockIdx.x));
        epoch = ok ? (unsigned)w1 : 0u;
        if (!ok) clear_slice();
    }
    po_lae_tables_load(&sm.lae, lane, 64);
    const PoLaeFast lae{&sm.lae};
    if (lane == 0) { sm.nupd = 0; sm.nupd_x = 0; }
    __syncthreads();
#ifdef PO_RING_TIMING
    // phase timers of workgroup 0 (wall_clock64: 100 MHz): see po_ring_launch for the names
    long long tk[32], tlast = wall_clock64();
    for (int i = 0; i < 32; ++i) tk[i] = 0;
    int tko = 0;
#define RT(i) do { const long long n_ = wall_clock64(); tk[tko + (i)] += n_ - tlast; tlast = n_; } while (0)
#define RTC(i, n) do { tk[(i)] += (n); } while (0)
#define RTX(i) do { const long long n_ = wall_clock64(); tk[(i)] += n_ - tlast; tlast = n_; } while (0)
#define RT_SET(o) do { tko = (o); } while (0)
#else
#define RT(i) do {} while (0)
#define RTC(i, n) do {} while (0)
#define RTX(i) do {} while (0)
#define RT_SET(o) do {} while (0)
#endif

    for (;;) {
        // ---------------------------------------------------------------- next pair from the queue
        int pi = 0;
        if (lane == 0) pi = atomicAdd(a.queue, 1);
        pi = __builtin_amdgcn_readfirstlane(pi);
        if (pi >= a.n) break;
        epoch++;
        if ((epoch & 0xffffu) == 0) { clear_slice(); epoch++; }
        const int2 m = a.meta[pi];
        if (m.y == X2_DEFERRED) continue;                 // beam2d_kernel decodes it after this kernel
        if (m.x != PO_OK || m.y < 0) {                    // refused by the pre-pass, or skipped upstream
            if (lane == 0) {
                a.seq_len[pi] = 0;
                if (m.y >= 0) a.status[pi] = m.x;
            }
            continue;
        }
        const int64_t o1 = a.y1_off[pi], o2 = a.y2_off[pi];
        const int U = (int)(a.y1_off[pi + 1] - o1), V = (int)(a.y2_off[pi + 1] - o2);
        const double* const yr = r ? a.y2 + o2 * C : a.y1 + o1 * C;      // this lane's read
        const int Tr = r ? V : U;
        const double* const cumr = r ? a.cum2 + (o2 - a.y2_off[0]) : a.cum1 + (o1 - a.y1_off[0]);
        const int4* const sched = a.sched + (o2 - a.y2_off[0]);
        const int nmain = a.nmain[pi];
        const int R2 = m.y, Rm2 = R2 - 1;
        const int NG = (int)min((long long)RG_NGL, pool_entries / ((long long)PO_A * 2 * R2));
        int st = PO_OK;

        // ---------------------------------------------------------------- tier-2 helpers (this lane's read)
        auto t2_read = [&](int row2, int node, int tq) -> double {
            double v = PO_NEG_INF;
            if (tq >= 0 && row2 >= 0) {
                const Ent e = pool[((size_t)row2 * 2 + r) * R2 + (tq & Rm2)];
                if (e.tag == make_tag(epoch, node, tq)) v = e.v[0];
            }
            return v;
        };
        auto t2_write = [&](int row2, int node, int tq, double v) {
            Ent e;
            e.tag = make_tag(epoch, node, tq);
            e.v[0] = v;
            pool[((size_t)row2 * 2 + r) * R2 + (tq & Rm2)] = e;
        };
        // ---------------------------------------------------------------- per-lane element state (slot s, read r)
        // table fields (the same in both halves of the wave)
        int e_id = 0, e_row2 = -1, e_lrow = 0, e_sym = 0, e_fc = -1, e_crow2 = -1, e_par = 0, e_gpar = -1, e_prow2 = -1,
            e_depth = 0, e_ps = PS_ROOT, e_alias = -1;
        bool live = false;
        // values of this read: [v_lo, v_done) computed in this incarnation, ring holds [v_hiw - 32, v_hiw)
        int v_lo = RG_FRESH, v_done = RG_FRESH, v_hiw = RG_FRESH, v_fresh = 0;   // v_fresh: 1 = seed from tier 2, 2 = brand new
        double v_self = PO_NEG_INF, v_mx = PO_NEG_INF;
        int v_mt = -1, v_td = 0;
        // a beam node whose parent is no element any more (FROZEN): the parent's last value and its time, taken when the
        // parent left — later times are absent (-inf), earlier ones (never asked for while window ends only grow) are in
        // tier 2.  fz_t = INT_MAX: nothing captured, always ask tier 2.
        double fz_val = PO_NEG_INF;
        int fz_t = INT_MAX;
        int nb = A, ne = A;
        unsigned lfree = (RG_NRP >= 32) ? 0xffffffffu : ((1u << RG_NRP) - 1u);
        int next_id = 1 + A;
        int gcur = 1;               // tier-2 group allocation cursor
        int yhi = 0;                // y rows [yhi - RG_NY, yhi) of this lane's read are in sm.ybuf
        int sel[6] = {0, 1, 2, 3, 4, 5};

        for (int q = lane; q < RG_NGL; q += 64) { sm.g_owner[q] = -1; sm.g_hi0[q] = 0; sm.g_hi1[q] = 0; }
        rg_sync();
        // root = node 0; its A children = nodes 1..A in row group 0 (BeamSearch.h:286-293), updated at t = 0 on both reads
        if (lane == 0) {
            apl[0] = po_pack_node(-1, A); afc[0] = 1; acrow[0] = 0;
            sm.g_owner[0] = 0; sm.g_hi0[0] = 1; sm.g_hi1[0] = 1;
            sm.sh[3] = 1;       // group allocation cursor
            sm.sh[4] = PO_OK;
        }
        if (s < A) {
            if (r == 0) { apl[1 + s] = po_pack_node(0, s); afc[1 + s] = -1; acrow[1 + s] = -1; }
            e_id = 1 + s; e_row2 = s; e_lrow = s; e_sym = sym_pack(s, A, true); e_fc = -1; e_crow2 = -1;
            e_par = 0; e_gpar = -1; e_prow2 = -1; e_depth = 1; e_ps = PS_ROOT; e_alias = -1;
            live = true;
            const double out = lae(0.0 + yr[s], PO_NEG_INF + yr[A]);   // update_prob(n, r, 0): parent = root at t = -1
            sm.ring[r][0][s] = out;
            v_lo = 0; v_done = 1; v_hiw = 1; v_fresh = 0; v_self = out;
            v_mx = out; v_mt = 0; v_td = 0;   // (the window maximum over [0, 1))
            lfree &= ~(1u << s);
        }
        lfree = (unsigned)__builtin_amdgcn_readfirstlane((int)(lfree & ~((1u << A) - 1u)));
        rg_sync();

        int mstep = 0, up = -1, vp = -1;
        // The walk's records, 64 at a time: lane l holds record 64 * batch + l of the current batch and of the next one
        // (requested a batch ahead: the load's latency never shows), the step's own record comes out with v_readlane.
        int4 rcur = sched[min(lane, max(nmain - 1, 0))], rnxt = sched[min(64 + lane, max(nmain - 1, 0))];
        auto rec_at = [&](int i) -> int4 {   // record of main step i (uniform i within the current batch)
            const int l = i & 63;
            return make_int4(__builtin_amdgcn_readlane(rcur.x, l), __builtin_amdgcn_readlane(rcur.y, l),
                             __builtin_amdgcn_readlane(rcur.z, l), __builtin_amdgcn_readlane(rcur.w, l));
        };
        int4 rec = rec_at(0);
        bool have_children = false;   // the table has its children slots (false only before the first expansion)
        bool tbl_fresh = true;        // the table has elements that have not computed yet (set by rebuild)
        bool tbl_uneven = false;      // a catch-up scan moved the beam nodes beyond their children
        unsigned long long cnt_ref = 0, cnt_x = 0;

        // ---------------------------------------------------------------- own ring / tier-2 value at time tq
        auto read_own = [&](int tq) -> double {
            if (tq < v_lo || tq >= v_hiw) return t2_read(e_row2, e_id, tq);     // an earlier incarnation's, or absent
            if (tq >= v_hiw - RG_RL) return sm.ring[r][tq & (RG_RL - 1)][e_lrow];
            return t2_read(e_row2, e_id, tq);                                   // evicted
        };

        // ---------------------------------------------------------------- y rows [t0, t0 + RG_NY) of this read -> LDS
        // (all of a lane's loads go out together — one memory round trip per reload, not one per element: a lone wave
        //  waits for every one of them, and a pair reloads ~470 times)
        auto y_reload = [&](int t0) {
            constexpr int PER = (RG_NY * RG_YC + 31) / 32;   // elements per lane (C <= RG_YC)
            double v[PER];
            int slot[PER];
            const int divC = (65536 + C - 1) / C;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int i = s + 32 * j;
                const int q = (i * divC) >> 16, c = i - q * C;
                const int t = t0 + q;
                const bool ok = i < RG_NY * C && t < Tr;
                slot[j] = ok ? (t & (RG_NY - 1)) * RG_YC + c : -1;
                v[j] = ok ? yr[(int64_t)t * C + c] : 0.0;
            }
            double* const yb = &sm.ybuf[r][0][0];
#pragma unroll
            for (int j = 0; j < PER; ++j)
                if (slot[j] >= 0) yb[slot[j]] = v[j];
        };

        // ---------------------------------------------------------------- one scan
        // Every participating lane computes [max(done, ws), we) of its read, all lanes of a read in lockstep on t (a
        // child at t reads its parent's t - 1, computed one iteration earlier or long ago: it is in the ring or in tier 2).
        // MAIN steps (is_main) track the window maximum; catch-up scans (BeamSearch.h:314-336) move the beam nodes only.
        double smx = PO_NEG_INF;   // out: max over this read's window (main steps)
        int c_plrow = 0;           // ring row of the parent element (e_ps >= 0), refreshed by rebuild
        // `uni`: the table is the previous main step's and every live lane's values end at the same time (no new
        // element, no catch-up since): nothing has to be asked of the parent's lane.
        auto scan = [&](bool is_main, int ws0, int we0, int ws1, int we1, int nlanes, bool uni) {
            const int ws = r ? ws1 : ws0, we = r ? we1 : we0;
            const bool part = live && s < nlanes && we > ws;
            // the window end moved back (an envelope with an occasional wide row): forget what lies beyond it, as the
            // reference's max does; recomputing it later rewrites the same bits
            if (part && v_fresh == 0 && v_done > we) {
                const double sv = read_own(we - 1);
                v_done = we; v_self = sv;
                if (v_done <= v_lo) { v_fresh = 2; }   // nothing left of this incarnation (cannot happen: lo <= ws < we)
            }
            int start = max(v_done, ws);
            double self = PO_NEG_INF;
            bool contin = false;   // continues where its values end (no new incarnation, no gap)
            if (part) {
                if (v_fresh != 0) {
                    start = ws;
                    self = (v_fresh == 1) ? t2_read(e_row2, e_id, start - 1) : PO_NEG_INF;
                    v_lo = start; v_done = start; v_hiw = start;
                } else if (start > v_done) {   // a gap (catch-ups went beyond the last window): the value at start - 1 was never computed
                    v_lo = start; v_done = start; v_hiw = start;
                } else {
                    self = v_self;
                    contin = true;
                }
            }
            const bool part2 = part && start < we;
            // ---- window maximum of the part [ws, start) that is not recomputed: carried from the previous step, unless its
            // time has left the window — then the stored values are looked at again (rr_need).  In the FAST loops the
            // ring reads for that are issued here and looked at after the iterations (the new values' maximum is
            // combined with the carried part's at the end: the new times are later, so `>=` keeps the latest maximum).
            double mx = PO_NEG_INF, cmx = PO_NEG_INF;
            int mt = -1, cmt = -1, td = ws, tr = INT_MIN;
            bool rr_need = false;
            const bool has_c = is_main && part && start > ws;
            if (has_c) {
                td = v_td;
#ifdef PO_ABL_NOREREAD   // timing ablation only (results are wrong)
                if (true) { cmx = v_mx; cmt = v_mt; }
#else
                if (v_mx == PO_NEG_INF || (v_mt >= ws && v_mt < start)) { cmx = v_mx; cmt = v_mt; }
#endif
                else rr_need = true;
            }
            auto rr_resolve = [&](const double* v4, bool have4) {   // the carried part's maximum from the stored values
                if (!rr_need) return;
                const bool inring = (ws >= v_lo && ws >= v_hiw - RG_RL);
                if (td <= ws) {   // non-increasing since before the window start (a node past its peak): its first value
                    cmx = (have4 && inring) ? v4[0] : read_own(ws);
                    cmt = ws;
                    return;
                }
                // otherwise only [ws, td] is looked at: from td on the values fall
                double pv = PO_NEG_INF;
                const int te = min(td + 1, start);
                td = ws;
                int tq0 = ws;
                if (have4 && inring) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int tq = ws + q;
                        if (tq < te) {
                            if (v4[q] >= cmx) { cmx = v4[q]; cmt = tq; }
                            if (tq > ws && v4[q] > pv) td = tq;
                            pv = v4[q];
                        }
                    }
                    tq0 = ws + 4;
                }
                for (int tq = tq0; tq < te; ++tq) {
                    const double v = read_own(tq);
                    if (v >= cmx) { cmx = v; cmt = tq; }
                    if (tq > ws && v > pv) td = tq;
                    pv = v;
                }
            };
            const int sym = sym_last(e_sym);
            if (is_main) RTX(uni ? 27 : 28);
            // ---- which loop.  FAST: every parent value an iteration needs is in the ring (or is the captured last value
            // of a frozen parent) and no slot that is overwritten can be read again — all but a few percent of the scans
            // (windows beyond 31 times, the root's children, elements restored from tier 2, catch-ups).
            bool fast = false;
            int p_lo = 0, p_done0 = 0, p_hiw0 = 0, p_start = 0, p_lrow = c_plrow;
            int tmin = 0, tph2 = 0;      // (half-uniform) first time computed by any lane / by the lanes that continue
            bool isP1 = false;           // starts before the continuing lanes do: a new element's full window
            if (uni) {
                const bool sl = part2 && (e_ps == PS_ROOT || (e_ps == PS_FROZEN && start - 1 < fz_t) || we - ws > RG_RL - 1);
                fast = is_main && (__ballot(sl) == 0ull);
                tmin = __builtin_amdgcn_readlane(start, 0);
                const int t1 = __builtin_amdgcn_readlane(start, 32);
                tmin = r ? t1 : tmin;   // (beam slot 0 is always live: every live lane of the half starts there)
                tph2 = tmin;
            }
            if (!fast) {
                // the parent's lane: what it holds and where it starts
                // (the values of [lo, hiw) exist: a window end that moved back lowers `done`, where the element resumes, not
                //  what the reference's maps hold)
                const int a_lo = (v_fresh != 0 && !part) ? INT_MAX : v_lo;
                const int a_done0 = (v_fresh != 0 && !part) ? INT_MAX : v_done;
                const int a_hiw = (v_fresh != 0 && !part) ? INT_MAX : v_hiw;
                const int a_start = part2 ? start : INT_MAX;
                const int plane = (e_ps >= 0) ? (hb | e_ps) : lane;
                p_lo = __shfl(a_lo, plane); p_done0 = __shfl(a_done0, plane); p_hiw0 = __shfl(a_hiw, plane);
                p_start = __shfl(a_start, plane); p_lrow = __shfl(e_lrow, plane);
                // the lanes that continue all start where the previous window ended
                const unsigned long long cb = __ballot(part2 && contin);
                const unsigned ch = r ? (unsigned)(cb >> 32) : (unsigned)cb;
                const int cl = (ch != 0u) ? (hb | __builtin_ctz(ch)) : lane;
                tph2 = (ch != 0u) ? __shfl(start, cl) : we;
                isP1 = part2 && !contin;
                const unsigned long long pb = __ballot(isP1);
                const unsigned ph = r ? (unsigned)(pb >> 32) : (unsigned)pb;
                tmin = (ph != 0u) ? min(ws, tph2) : tph2;
                bool sl = false;
                if (part2) {
                    const int tm0 = start - 1;
                    if (we - ws > RG_RL - 1) sl = true;
                    if (contin && start != tph2) sl = true;
                    if (e_ps >= 0) {
                        // [tm0, we - 2] must be there: old values (below p_done0) or computed in lockstep (from p_start on)
                        const bool plain0 = tm0 >= p_lo && tm0 >= p_hiw0 - RG_RL && (tm0 < p_hiw0 || tm0 >= p_start);
                        if (!plain0 || (p_start > p_hiw0 && p_hiw0 < we - 1)) sl = true;
                        if (isP1 && p_start < tph2) sl = true;   // (a new element under a parent that moves before the others do)
                    } else if (e_ps == PS_ROOT || isP1) sl = true;
                    else if (tm0 < fz_t) sl = true;
                }
                fast = is_main && (__ballot(sl) == 0ull);
#ifdef PO_RING_TIMING
                if (is_main && !fast) {
                    if (__ballot(part2 && we - ws > RG_RL - 1) != 0ull) RTC(uni ? 14 : 15, 1);
                    else if (__ballot(part2 && e_ps == PS_ROOT) != 0ull) RTC(23, 1);
                    else RTC(11, 1);
                }
#endif
                if (!fast) {   // the general loop walks every lane from the earliest start
                    int tm_ = part2 ? start : INT_MAX;
#pragma unroll
                    for (int off = 16; off >= 1; off >>= 1) tm_ = min(tm_, __shfl_xor(tm_, off));
                    tmin = tm_;
                }
            }
            double rr4[4] = {PO_NEG_INF, PO_NEG_INF, PO_NEG_INF, PO_NEG_INF};
            if (fast) {
                if (rr_need) {   // (window <= 31 times: these slots are not written in this scan)
                    const double* rp = &sm.ring[r][0][e_lrow];
#pragma unroll
                    for (int q = 0; q < 4; ++q) rr4[q] = rp[((ws + q) & (RG_RL - 1)) * RG_NRP];
                }
            } else {
                rr_resolve(rr4, false);
            }
            RT(1);
            if (fast) {
                // ---- y rows of [tmin, we) resident (the window has at most 31 times)
                const bool hwk = (tmin < we) && (__ballot(part2) != 0ull);
                if (hwk && !(tmin >= yhi - RG_NY && we <= yhi)) { y_reload(tmin); yhi = tmin + RG_NY; RTC(20, 1); }
                rg_sync();
                const double* const ringp = &sm.ring[r][0][p_lrow];
                double* const ringm = &sm.ring[r][0][e_lrow];
                const double* const yb_ = &sm.ybuf[r][0][0];
                // ---- phase 1: the new elements' windows up to where everybody else starts.  Their parents do not move
                // there: every operand of an iteration is known before the previous one ends, nothing is handed over.
                const int n1 = max(min(tph2, we) - tmin, 0);   // (half-uniform)
                const int n1max = max(__builtin_amdgcn_readlane(n1, 0), __builtin_amdgcn_readlane(n1, 32));
                RTC(16, 1); RTC(18, n1max);
                if (n1max > 0) {
                    int t = tmin;
                    double nya = 0.0, nyb = 0.0, npp = 0.0;
                    if (isP1) {
                        const double* yrow = yb_ + (t & (RG_NY - 1)) * RG_YC;
                        nya = yrow[sym]; nyb = yrow[A]; npp = ringp[((t - 1) & (RG_RL - 1)) * RG_NRP];
                    }
                    for (int k = 0; k < n1max; ++k) {
                        if (isP1 && k < n1) {
                            const double ya = nya, yb = nyb, pp = npp;
                            const int tn = t + 1;
                            const double* yrow = yb_ + (tn & (RG_NY - 1)) * RG_YC;
                            nya = yrow[sym]; nyb = yrow[A]; npp = ringp[(t & (RG_RL - 1)) * RG_NRP];   // (one past the end: read, never used)
                            const double out = lae(pp + ya, self + yb);
#ifdef PO_RING_TRACE_NODE
                            if (pi == 0 && e_id == PO_RING_TRACE_NODE) printf("V %d %d %d %.17g %.17g %.17g P1\n", e_id, r, t, out, pp, self);
#endif
                            ringm[(t & (RG_RL - 1)) * RG_NRP] = out;
                            if (out > self) tr = t;   // the last time a value rose
                            self = out;
                            mt = (out >= mx) ? t : mt;
                            mx = po_vmax(mx, out);
                            t = tn;
                        }
                    }
                    rg_sync();
                }
                RTX(24);
                // ---- phase 2: everybody, in lockstep (a child reads what its parent wrote one iteration earlier)
                const int t2 = min(tph2, we);
                const int n2 = we - t2;
                const int n2max = max(__builtin_amdgcn_readlane(n2, 0), __builtin_amdgcn_readlane(n2, 32));
                RTC(17, 1); RTC(19, n2max);
                for (int k = 0; k < n2max; ++k) {
                    const int t = t2 + k;
                    if (part2 && k < n2) {
                        const double* yrow = yb_ + (t & (RG_NY - 1)) * RG_YC;
                        const double ya = yrow[sym], yb = yrow[A];
                        const int tm = t - 1;
                        double pp = ringp[(tm & (RG_RL - 1)) * RG_NRP];
                        if (e_ps < 0) pp = (tm == fz_t) ? fz_val : PO_NEG_INF;
                        const double out = lae(pp + ya, self + yb);
#ifdef PO_RING_TRACE_NODE
                        if (pi == 0 && e_id == PO_RING_TRACE_NODE) printf("V %d %d %d %.17g %.17g %.17g P2 ps %d fzt %d\n", e_id, r, t, out, pp, self, e_ps, fz_t);
#endif
                        ringm[(t & (RG_RL - 1)) * RG_NRP] = out;
                        if (out > self) tr = t;
                        self = out;
                        mt = (out >= mx) ? t : mt;
                        mx = po_vmax(mx, out);
                    }
                    rg_sync();
                }
                if (part2) v_hiw = max(v_hiw, we);
                RTX(25);
            } else {
                const int span = (tmin == INT_MAX) ? 0 : we - tmin;   // (half-uniform)
                const int niter = max(__builtin_amdgcn_readlane(span, 0), __builtin_amdgcn_readlane(span, 32));
                RTC(21, 1); RTC(22, niter);
                int k = 0;
                while (k < niter) {
                    const int tcur = tmin + k;   // (garbage when this half has nothing to do: guarded by span)
                    const bool hw = k < span;    // this half still has times to compute
                    if (hw && !(tcur >= yhi - RG_NY && tcur < yhi)) { y_reload(tcur); yhi = tcur + RG_NY; RTC(20, 1); }
                    rg_sync();
                    const int cend = hw ? (min(we, yhi) - tmin) : niter;
                    const int kend = min(__builtin_amdgcn_readlane(cend, 0), __builtin_amdgcn_readlane(cend, 32));
                    for (; k < kend; ++k) {
                        const int t = tmin + k;
                        if (part2 && t >= start && t < we) {
                            const double ya = sm.ybuf[r][t & (RG_NY - 1)][sym], yb = sm.ybuf[r][t & (RG_NY - 1)][A];
                            const int tm = t - 1;
                            double pp;
                            if (e_ps >= 0) {
                                if (tm >= p_start || (tm >= p_lo && tm < p_hiw0 && tm >= p_hiw0 - RG_RL))
                                    pp = sm.ring[r][tm & (RG_RL - 1)][p_lrow];
                                else if (tm < p_lo || tm < p_hiw0) pp = t2_read(e_prow2, e_par, tm);    // an earlier incarnation's, or evicted
                                else pp = PO_NEG_INF;                                                    // never computed
                            } else if (e_ps == PS_ROOT) {
                                pp = (tm < 0) ? 0.0 : cumr[tm];
                            } else if (tm >= fz_t) {
                                pp = (tm == fz_t) ? fz_val : PO_NEG_INF;                                 // frozen parent: its last value, then nothing
                            } else {
                                pp = t2_read(e_prow2, e_par, tm);
                            }
                            const double out = lae(pp + ya, self + yb);
#ifdef PO_RING_TRACE_NODE
                            if (pi == 0 && e_id == PO_RING_TRACE_NODE) printf("V %d %d %d %.17g %.17g %.17g G ps %d fzt %d main %d\n", e_id, r, t, out, pp, self, e_ps, fz_t, (int)is_main);
#endif
                            double* slot = &sm.ring[r][t & (RG_RL - 1)][e_lrow];
                            if (t >= v_hiw) {
                                const int to = t - RG_RL;
                                if (to >= v_lo && to >= ws - 1) {   // the slot's old value can still be read: tier 2 takes it
                                    t2_write(e_row2, e_id, to, *slot);
                                    atomicMax(&g_hi[e_row2 >> 2], to + 1);
                                }
                                v_hiw = t + 1;
                            }
                            *slot = out;
                            if (out > self) tr = t;   // the last time a value rose
                            self = out;
                            mt = (out >= mx) ? t : mt;
                            mx = po_vmax(mx, out);
                        }
                        rg_sync();
                    }
                }
                RTX(26);
            }
            RT(2);
            if (fast) rr_resolve(rr4, true);
            if (has_c && !(mx >= cmx)) { mx = cmx; mt = cmt; }   // (new values, later in time, win ties)
            if (part2) { v_done = we; v_self = self; v_fresh = 0; }
            if (is_main) {
                if (part) { v_mx = mx; v_mt = mt; v_td = max(td, tr); }
                smx = part ? mx : PO_NEG_INF;
            }
            if (a.upd_count != nullptr) {
                const int lenx = part2 ? we - start : 0;
                int tot = lenx;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) tot += __shfl_xor(tot, off);
                cnt_x += (unsigned)tot;
            }
        };

        // ---------------------------------------------------------------- the next element table
        // Given the new beam (sel[0 .. nbn): slots of the present table, in rank order), builds the table of the next main
        // step: expansion of the beam nodes (BeamSearch.h:342-360: node ids in beam order), children slots, which old
        // element continues in which new slot (its ring row, carried maxima and times move with it), spill of the
        // elements that stop being elements, ring rows for the new ones.
        auto rebuild = [&](int nbn, int nu, int nv, int nce, int nre, int cu, int cv) {
            const int nbo = nb, neo = ne;
            const int nen = nbn * (A + 1);
            // ---- A. new beam lanes take their fields from the old slot sel[s]
            int mysel = sel[0];
#pragma unroll
            for (int i = 1; i < 6; ++i) mysel = (s == i) ? sel[i] : mysel;
            const bool rb = s < nbn;                       // this lane is a beam slot of the new table
            const bool rc = !rb && s < nen;                // ... a child slot
            const int j = rc ? (((s - nbn) * divA) >> 16) : 0, c = rc ? (s - nbn) - j * A : 0;
            int pj = sel[0];
#pragma unroll
            for (int i = 1; i < 6; ++i) pj = (j == i) ? sel[i] : pj;
            const int srcb = rb ? mysel : 0;
            int n_id = __shfl(e_id, hb | srcb), n_row2 = __shfl(e_row2, hb | srcb), n_sym = __shfl(e_sym, hb | srcb);
            int n_fc = __shfl(e_fc, hb | srcb), n_crow2 = __shfl(e_crow2, hb | srcb), n_par = __shfl(e_par, hb | srcb);
            int n_gpar = __shfl(e_gpar, hb | srcb), n_prow2 = __shfl(e_prow2, hb | srcb), n_depth = __shfl(e_depth, hb | srcb);
            // ---- every old element marks its tier-2 group with what it may still write there
            if (live && v_fresh == 0) atomicMax(&g_hi[e_row2 >> 2], v_hiw);
            // ---- B. expansion of the new beam nodes
            if (rb && n_fc == -2) { n_fc = afc[n_id]; n_crow2 = acrow[n_id]; }   // a node whose parent re-entered: the arena knows
            rg_sync();
            bool isnew = false, need_group = false;
            if (rb) {
                isnew = n_fc < 0;
                need_group = isnew || n_crow2 < 0 || n_crow2 >= NG || sm.g_owner[n_crow2] != n_id;   // (old rows recycled: all dead)
#ifdef PO_EMU_DEBUG
                if (r == 0) printf("NG nu %d slot %d id %d fc %d crow2 %d NG %d owner %d hi %d %d\n", nu, s, n_id, n_fc, n_crow2, NG, (n_crow2 >= 0 && n_crow2 < NG) ? sm.g_owner[n_crow2] : -9, (n_crow2 >= 0 && n_crow2 < NG) ? sm.g_hi0[n_crow2] : -9, (n_crow2 >= 0 && n_crow2 < NG) ? sm.g_hi1[n_crow2] : -9);
#endif
            }
            {
                const unsigned bn = (unsigned)__ballot(isnew && r == 0);
                if (isnew) {
                    n_fc = next_id + A * __popc(bn & ((1u << s) - 1u));
                    if (r == 0) afc[n_id] = n_fc;
                }
                next_id += A * __popc(bn);
                if (rb && !need_group) { atomicMax(&sm.g_hi0[n_crow2], nce); atomicMax(&sm.g_hi1[n_crow2], nre); }
                rg_sync();
                unsigned hg = (unsigned)__ballot(need_group && r == 0);
                while (hg != 0) {   // (uniform: every lane walks the group table, lane 0 writes)
                    const int jj = __builtin_ctz(hg);
                    hg &= hg - 1;
                    const int owner = __builtin_amdgcn_readlane(n_id, jj);
                    int gg = -1;
                    for (int tries = 0; tries < NG; ++tries) {
                        const int c = gcur;
                        gcur = (gcur + 1 == NG) ? 0 : gcur + 1;
                        if (sm.g_owner[c] < 0 || (sm.g_hi0[c] <= nu - 1 && sm.g_hi1[c] <= nv - 1)) { gg = c; break; }
                    }
                    if (gg < 0) { st = PO_E_NOMEM; gg = 0; }
                    rg_sync();   // (every lane has walked the table before lane 0 changes it)
                    if (lane == 0) { sm.g_owner[gg] = owner; sm.g_hi0[gg] = nce; sm.g_hi1[gg] = nre; acrow[owner] = gg; }
                    if (s == jj) n_crow2 = gg;
                    rg_sync();
                }
            }
            // ---- C. children slots take their parent's (new) fields
            const int p_id = __shfl(n_id, hb | j), p_fc = __shfl(n_fc, hb | j), p_crow2 = __shfl(n_crow2, hb | j);
            const int p_sym = __shfl(n_sym, hb | j), p_par = __shfl(n_par, hb | j), p_row2 = __shfl(n_row2, hb | j);
            const int p_depth = __shfl(n_depth, hb | j);
            const bool p_isnew = __shfl((int)isnew, hb | j) != 0;
#ifdef PO_EMU_DEBUG
            if (r == 0 && rb) printf("RB nu %d slot %d from %d id %d fc %d crow2 %d isnew %d needg %d\n", nu, s, srcb, n_id, n_fc, n_crow2, (int)isnew, (int)need_group);
#endif
            int n_alias = -1, n_ps = PS_FROZEN;
            if (rc) {
                n_id = p_fc + c; n_row2 = p_crow2 * PO_A + c; n_sym = sym_pack(c, sym_last(p_sym), false);
                n_par = p_id; n_gpar = p_par; n_prow2 = p_row2; n_depth = p_depth + 1; n_ps = j;
                n_fc = p_isnew ? -1 : -2; n_crow2 = p_isnew ? -1 : -2;
                if (p_isnew && r == 0) { apl[n_id] = po_pack_node(p_id, c); afc[n_id] = -1; acrow[n_id] = -1; }
            }
            // a child slot whose node is also a beam slot is the same node pushed twice (Beam::prune's std::unique)
            for (int i = 0; i < nbn; ++i) {
                const int bid = __builtin_amdgcn_readlane(n_id, i);
                if (rc && bid == n_id) n_alias = i;
            }
            // ---- D. which old slot continues here
            int src = -1;
            if (rb) src = mysel;
            else if (rc && n_alias < 0 && pj < nbo) {   // the parent was a beam node: its children were elements (or aliases of beam slots)
                if (have_children) src = nbo + A * pj + c;
            }
            {   // the parent enters the beam: a child of it was an element only as a beam node
                const bool look = !rb && rc && n_alias < 0 && pj >= nbo;
                for (int i = 0; i < nbo; ++i) {   // (wave-uniform loop: v_readlane)
                    const int oid = __builtin_amdgcn_readlane(e_id, i);
                    if (look && oid == n_id) src = i;
                }
            }
            {   // (an old child slot that was an alias hands over to the beam slot that held the node)
                const int oa = __shfl(e_alias, hb | max(src, 0));
                if (!rb && src >= nbo && oa >= 0) src = oa;
            }
            const bool nlive = (rb || (rc && n_alias < 0));
            // ---- E. old elements nobody continues: spill what a later step can still read, free the ring row
            if (lane == 0) { sm.sh[5] = 0; sm.sh[6] = 0; }
            rg_sync();
            if (nlive && src >= 0 && r == 0) atomicOr((unsigned*)&sm.sh[5], 1u << src);
            rg_sync();
            const unsigned claimed = (unsigned)sm.sh[5];
            const bool leaving = live && !((claimed >> s) & 1u);
            if (leaving) {
                if (v_fresh == 0) {
                    const int from = max(max(r ? cv : cu, v_lo), v_hiw - RG_RL);
                    const double* rp = &sm.ring[r][0][e_lrow];
                    for (int bt = from; bt < v_hiw; bt += 4) {   // (four ring reads in flight, then the stores)
                        double v4[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) v4[q] = rp[((bt + q) & (RG_RL - 1)) * RG_NRP];
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (bt + q < v_hiw) t2_write(e_row2, e_id, bt + q, v4[q]);
                    }
                }
                if (r == 0) atomicOr((unsigned*)&sm.sh[6], 1u << e_lrow);
            }
            rg_sync();
            lfree |= (unsigned)sm.sh[6];
            // ---- F. the lanes take their new identity
            const int gsrc = hb | max(src, 0);
            const int g_lrow = __shfl(e_lrow, gsrc), g_lo = __shfl(v_lo, gsrc), g_done = __shfl(v_done, gsrc);
            const int g_hiw = __shfl(v_hiw, gsrc), g_fresh = __shfl(v_fresh, gsrc), g_mt = __shfl(v_mt, gsrc), g_td = __shfl(v_td, gsrc);
            const double g_self = __shfl(v_self, gsrc), g_mx = __shfl(v_mx, gsrc);
            const int g_fc = __shfl(e_fc, gsrc), g_crow2 = __shfl(e_crow2, gsrc);
            // the last value of the node's parent as the old table knew it: of the parent's lane if it was an element,
            // else what was captured when it stopped being one
            double c_val;
            int c_t;
            const int op = __shfl(e_ps, gsrc);                     // the parent's slot in the old table (or ROOT / FROZEN)
            {
                const int opl = hb | max(op, 0);
                const int o_hiw = __shfl(v_hiw, opl), o_fresh = __shfl(v_fresh, opl), o_lrow = __shfl(e_lrow, opl);
                const double q_val = __shfl(fz_val, gsrc);
                const int q_t = __shfl(fz_t, gsrc);
                const double o_last = sm.ring[r][(o_hiw - 1) & (RG_RL - 1)][o_lrow];   // (the ring rows still hold the old table's values)
                c_val = (op >= 0) ? o_last : q_val;
                c_t = (op >= 0) ? ((o_fresh == 0) ? o_hiw - 1 : INT_MAX) : q_t;
            }
            const bool fresh = nlive && src < 0;
            e_id = n_id; e_row2 = n_row2; e_sym = n_sym; e_par = n_par; e_gpar = n_gpar; e_prow2 = n_prow2; e_depth = n_depth;
            e_alias = rc ? n_alias : -1;
            e_fc = n_fc; e_crow2 = n_crow2;
            if (rc && src >= 0) { e_fc = g_fc; e_crow2 = g_crow2; }   // a continuing child keeps what is known about its own children
            live = nlive;
            fz_val = c_val; fz_t = (nlive && src >= 0) ? c_t : INT_MAX;
            if (nlive && src >= 0) {
                e_lrow = g_lrow; v_lo = g_lo; v_done = g_done; v_hiw = g_hiw; v_fresh = g_fresh; v_mt = g_mt; v_td = g_td;
                v_self = g_self; v_mx = g_mx;
            } else {
                v_lo = RG_FRESH; v_done = RG_FRESH; v_hiw = RG_FRESH; v_fresh = (rc && p_isnew) ? 2 : 1;
                v_self = PO_NEG_INF; v_mx = PO_NEG_INF; v_mt = -1; v_td = 0; e_lrow = 0;
            }
            {   // ring rows for the new elements
                const unsigned fm = (unsigned)__ballot(fresh && r == 0);
                const int rank = __popc(fm & ((1u << s) - 1u));
                const int nf = __popc(fm);
                for (int q = 0; q < nf; ++q) {
                    const int rowq = (lfree != 0u) ? __builtin_ctz(lfree) : 0;
                    if (lfree == 0u) st = PO_E_NOMEM;   // (cannot happen: at most W * (A + 1) <= RG_NRP live elements)
                    lfree &= lfree - 1u;
                    if (fresh && rank == q) e_lrow = rowq;
                }
            }
            // ---- the parent slot of the beam nodes: a beam node, a child of a beam node, the root, or none (frozen)
            nb = nbn; ne = nen;
            {   // (wave-uniform loops: v_readlane)
                if (rb) n_ps = (e_par == 0) ? PS_ROOT : PS_FROZEN;
                for (int i = 0; i < nbn; ++i) {
                    const int bid = __builtin_amdgcn_readlane(e_id, i);
                    if (rb && e_par != 0 && bid == e_par) n_ps = i;
                }
                const bool fz = rb && n_ps == PS_FROZEN;
                for (int i = 0; i < nbn; ++i) {
                    const int bid = __builtin_amdgcn_readlane(e_id, i);
                    if (fz && bid == e_gpar) n_ps = nbn + A * i + sym_plast(e_sym);
                }
            }
            e_ps = n_ps;
            // ---- G. a frozen parent that is an element again.  A beam node whose parent had left the table computed its
            // newest values against "absent" (-inf beyond the parent's last time).  When the grandparent enters the beam the
            // parent comes back as one of its children, computes its whole window — times it never had — and the
            // reference's step, which recomputes every window in full, then gives the node (and everything below it)
            // different values over the part of the window it already had.  Those elements go back to the window start;
            // the part before it is what both sides keep.  (Everybody else's inputs are unchanged: skipping their old
            // times rewrites nothing.)
            {
                bool rew = rb && nlive && src >= 0 && op == PS_FROZEN && n_ps >= 0;
                if (__ballot(rew) != 0ull) {
                    for (int it = 0; it < W; ++it) {   // ... and their descendants in the table, generation by generation
                        const bool prew = __shfl((int)rew, hb | max(e_ps, 0)) != 0;
                        if (live && e_ps >= 0 && prew) rew = true;
                    }
                    const int wsn = r ? nv : nu;
                    if (rew && live && v_fresh == 0 && v_done > wsn) {
                        v_self = read_own(wsn - 1);
                        v_done = wsn;
                        v_lo = min(v_lo, wsn);   // (a window start that moved back: the times from it on are this incarnation's again)
                    }
                }
            }
            c_plrow = __shfl(e_lrow, hb | max(e_ps, 0));
            tbl_fresh = __ballot(live && v_fresh != 0) != 0ull;
            have_children = true;
            rg_sync();
        };

        // the first table: the root's children are the beam, their children come from the first expansion
        rebuild(A, rec.x, rec.y, rec.z, rec.w, 0, 0);
        RT_SET(0); RT(0);
        bool after_event = true;

        // ---------------------------------------------------------------- the diagonal walk (BeamSearch.h:300-393)
        while (st == PO_OK && mstep < nmain) {
            int u = rec.x, v = rec.y, ce = rec.z, re = rec.w;
            double sc = PO_NEG_INF;
            bool viol = false, run_viol = false;
#ifndef PO_RING_NO_RUN
            // ---- a RUN of main steps on the table as it stands.  After a step that kept the set of beam nodes, with every
            // live lane's values ending at the same time and nothing to catch up, the next step is: the new times of the two
            // windows (often none on a read: the envelope's window ends move a base at a time) in lockstep, the window maxima
            // from what is carried, the score, the one comparison per child.  The general step below does the same through
            // scan()'s prologue (which loop, which parent, which seed); here all of that is known.  The run ends at the first
            // step that is not of this kind (it is then done below) or that changes the beam (it is ranked below).
            if (!tbl_fresh && !tbl_uneven && nb == W && __ballot(live && e_ps == PS_ROOT) == 0ull) {
                const int sym = sym_last(e_sym);
                const double* const ringp = &sm.ring[r][0][c_plrow];
                double* const ringm = &sm.ring[r][0][e_lrow];
                const double* const yb_ = &sm.ybuf[r][0][0];
                for (;;) {
                    u = rec.x; v = rec.y; ce = rec.z; re = rec.w;
                    const int d0 = __builtin_amdgcn_readlane(v_done, 0), d1 = __builtin_amdgcn_readlane(v_done, 32);
                    if (!(u <= d0 && d0 <= ce && v <= d1 && d1 <= re) || ce - u > RG_RL - 1 || re - v > RG_RL - 1 || mstep + 1 >= nmain) break;
                    const int ws = r ? v : u, we = r ? re : ce, dr = r ? d1 : d0;
                    const bool part2 = live && dr < we;
                    if (__ballot(part2 && e_ps == PS_FROZEN && dr - 1 < fz_t) != 0ull) break;   // a frozen parent's older values: tier 2
                    // ---- the carried part [ws, dr) of the window: its maximum is what the previous step left, unless that
                    // time is now before the window start
                    const bool has_c = live && dr > ws;
                    double mx = PO_NEG_INF, cmx = PO_NEG_INF, self = v_self;
                    int mt = -1, cmt = -1, td = has_c ? v_td : ws, tr = INT_MIN;
                    bool rr_need = false;
                    if (has_c) {
                        if (v_mx == PO_NEG_INF || v_mt >= ws) { cmx = v_mx; cmt = v_mt; }
                        else rr_need = true;
                    }
                    double rr4[4] = {PO_NEG_INF, PO_NEG_INF, PO_NEG_INF, PO_NEG_INF};
                    if (rr_need) {   // (window <= 31 times: these slots are not written in this step)
#pragma unroll
                        for (int q = 0; q < 4; ++q) rr4[q] = ringm[((ws + q) & (RG_RL - 1)) * RG_NRP];
                    }
                    // ---- the new times [dr, we), everybody in lockstep
                    const int n2 = we - dr;   // (half-uniform, >= 0)
                    const int n2max = max(ce - d0, re - d1);
                    if (n2max > 0) {
                        if (n2 > 0 && !(dr >= yhi - RG_NY && we <= yhi)) { y_reload(dr); yhi = dr + RG_NY; }
                        rg_sync();
                        for (int k = 0; k < n2max; ++k) {
                            const int t = dr + k;
                            if (live && k < n2) {
                                const double* yrow = yb_ + (t & (RG_NY - 1)) * RG_YC;
                                const double ya = yrow[sym], yb = yrow[A];
                                const int tm = t - 1;
                                double pp = ringp[(tm & (RG_RL - 1)) * RG_NRP];
                                if (e_ps < 0) pp = (tm == fz_t) ? fz_val : PO_NEG_INF;
                                const double out = lae(pp + ya, self + yb);
#ifdef PO_RING_TRACE_NODE
                                if (pi == 0 && e_id == PO_RING_TRACE_NODE) printf("V %d %d %d %.17g %.17g %.17g RUN ps %d fzt %d\n", e_id, r, t, out, pp, self, e_ps, fz_t);
#endif
                                ringm[(t & (RG_RL - 1)) * RG_NRP] = out;
                                if (out > self) tr = t;
                                self = out;
                                mt = (out >= mx) ? t : mt;
                                mx = po_vmax(mx, out);
                            }
                            rg_sync();
                        }
                        if (part2) { v_hiw = max(v_hiw, we); v_done = we; v_self = self; }
                    }
                    if (rr_need) {   // the carried part's maximum from the stored values (as scan()'s rr_resolve)
                        const bool inring = (ws >= v_lo && ws >= v_hiw - RG_RL);
                        if (td <= ws) {
                            cmx = inring ? rr4[0] : read_own(ws);
                            cmt = ws;
                        } else {
                            double pv = PO_NEG_INF;
                            const int te = min(td + 1, dr);
                            td = ws;
                            int tq0 = ws;
                            if (inring) {
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const int tq = ws + q;
                                    if (tq < te) {
                                        if (rr4[q] >= cmx) { cmx = rr4[q]; cmt = tq; }
                                        if (tq > ws && rr4[q] > pv) td = tq;
                                        pv = rr4[q];
                                    }
                                }
                                tq0 = ws + 4;
                            }
                            for (int tq = tq0; tq < te; ++tq) {
                                const double vq = read_own(tq);
                                if (vq >= cmx) { cmx = vq; cmt = tq; }
                                if (tq > ws && vq > pv) td = tq;
                                pv = vq;
                            }
                        }
                    }
                    if (has_c && !(mx >= cmx)) { mx = cmx; mt = cmt; }   // (new values, later in time, win ties)
                    if (live) { v_mx = mx; v_mt = mt; v_td = max(td, tr); }
                    smx = live ? mx : PO_NEG_INF;
                    if (a.upd_count != nullptr) {
                        cnt_ref += (unsigned)(ne * ((ce - u) + (re - v)));
                        cnt_x += (unsigned)(__popcll(__ballot(live && r == 0)) * (ce - d0) + __popcll(__ballot(live && r == 1)) * (re - d1));
                    }
                    sc = smx + __shfl_xor(smx, 32);
#ifdef PO_RING_TRACE
                    if (pi == 0 && live && r == 0) printf("T %d %d %d %.17g\n", u, v, e_id, sc);
#endif
                    double scmin = rg_readlane_d(sc, 0);
                    for (int i = 1; i < nb; ++i) scmin = fmin(scmin, rg_readlane_d(sc, i));
                    viol = live && s >= nb && !(scmin > sc);
                    up = u; vp = v;
                    mstep++;
                    if ((mstep & 63) == 0) {
                        rcur = rnxt;
                        rnxt = sched[min(mstep + 64 + lane, max(nmain - 1, 0))];
                    }
                    rec = rec_at(min(mstep, nmain - 1));
                    RTC(12, 1);
                    if (__ballot(viol) != 0ull) { run_viol = true; break; }
                }
                RT(3);
            }
#endif
            if (!run_viol) {
            u = rec.x; v = rec.y; ce = rec.z; re = rec.w;
            // ---- catch-up steps between the previous main step and this one (:314-336): only the beam nodes, one time
            // at a time; a time the last main step's window covered is a no-op (the bits are there)
            {
                const int nbe = min(W, nb);
                const int d0 = __builtin_amdgcn_readlane(v_done, 0), d1 = __builtin_amdgcn_readlane(v_done, 32);
                if (a.upd_count != nullptr) cnt_ref += (unsigned)((max(u - up - 1, 0) + max(v - vp - 1, 0)) * nbe);
                if (u - 1 >= max(up + 1, d0)) { scan(false, up + 1, u, 0, 0, nbe, false); tbl_uneven = true; }
                if (v - 1 >= max(vp + 1, d1)) { scan(false, 0, 0, vp + 1, v, nbe, false); tbl_uneven = true; }
            }
            // ---- MAIN step at (u, v): windows [u, ce) x [v, re)  (:342-375)
            RT_SET(after_event ? 4 : 0); RTX(29);
            RTC(after_event ? 13 : 12, 1);
            {
                const int d0 = __builtin_amdgcn_readlane(v_done, 0), d1 = __builtin_amdgcn_readlane(v_done, 32);
                const bool uni = !tbl_fresh && !tbl_uneven && u <= d0 && d0 <= ce && v <= d1 && d1 <= re;
                scan(true, u, ce, v, re, 32, uni);
                tbl_fresh = false; tbl_uneven = false;
            }
            if (a.upd_count != nullptr) cnt_ref += (unsigned)(ne * ((ce - u) + (re - v)));
            // node_greater_max_sym: max over read 0's window + max over read 1's
            sc = smx + __shfl_xor(smx, 32);
#ifdef PO_RING_TRACE   // debugging builds only (scripts/trace_rowcol.py): every candidate's score before the prune
            if (pi == 0 && live && r == 0) printf("T %d %d %d %.17g\n", u, v, e_id, sc);
#endif
            // ---- prune (Beam.h:93-108).  Most steps keep the SET of beam nodes: iff every child is strictly below the smallest
            // beam score (a child AT it, ties included, goes the full way, as partial_sort decides them).  The order of the
            // beam nodes among themselves is not looked at: nothing is created while the set stays (every beam node has its
            // children), ties are decided on node ids, and the order matters only where nodes are created — the step in which
            // the set changes ranks everybody — and for the label: the last main step is always ranked.  (Round 3, late: the
            // first form also wanted the beam scores still in order, and rebuilt the table for every permutation of the same
            // five nodes.)
            viol = (nb != W) || (mstep + 1 == nmain);
            if (!viol) {   // (wave-uniform: nb == W and not the last step)
                double scmin = rg_readlane_d(sc, 0);
                for (int i = 1; i < nb; ++i) scmin = fmin(scmin, rg_readlane_d(sc, i));
                if (live && s >= nb) viol = !(scmin > sc);
            }
            up = u; vp = v;
            mstep++;
            if ((mstep & 63) == 0) {   // the next batch becomes the current one, the one after it is requested
                rcur = rnxt;
                rnxt = sched[min(mstep + 64 + lane, max(nmain - 1, 0))];
            }
            rec = rec_at(min(mstep, nmain - 1));
            RT(3);
#ifdef PO_RING_TIMING
            after_event = false;
#endif
            if (__ballot(viol) == 0ull) continue;
            }   // (!run_viol)
            const bool cand = live;
            RT_SET(8);
            // ---- full ranking among the distinct candidates
            const unsigned cm = (unsigned)__ballot(cand && r == 0);
            const int ncand = __popc(cm);
            // Only the beam nodes and the children that reach the smallest beam score can be among the W best (every
            // other child has W candidates above it), and nothing outside that set outranks a member of it: the ranks
            // are taken within it (a handful of candidates instead of W * (A + 1)).
            unsigned smask = cm;
            if (nb == W) {
                double thr = rg_readlane_d(sc, 0);
                for (int i = 1; i < nb; ++i) thr = fmin(thr, rg_readlane_d(sc, i));
                smask = (unsigned)__ballot(cand && r == 0 && (s < nb || sc >= thr));
            }
            int rank = 0, neq = 0;
            for (unsigned mm = smask; mm != 0u; mm &= mm - 1u) {
                const int o = __builtin_ctz(mm);
                const double so = rg_readlane_d(sc, o);
                const int io = __builtin_amdgcn_readlane(e_id, o);
                rank += ((so > sc) | (!(sc > so) & (io < e_id))) ? 1 : 0;
                neq += (so == sc) ? 1 : 0;
            }
            if (!((smask >> s) & 1u)) { rank = 64; neq = 0; }
            const int nbn = min(W, ncand);
#pragma unroll
            for (int jx = 0; jx < 6; ++jx) {
                const unsigned long long bj = __ballot(cand && r == 0 && rank == jx);
                sel[jx] = (bj != 0ull) ? (int)__builtin_ctzll(bj) : 0;
            }
            if (__ballot(cand && neq > 1 && rank < W) != 0ull) {
                // exact ties reaching into the beam: what libstdc++'s partial_sort / sort leave on the candidates in
                // creation order (po_device.h), replayed by one lane
                int pos = 0;   // (the replay runs over ALL candidates in creation order)
                for (int o = 0; o < ne; ++o) {
                    const int io = __builtin_amdgcn_readlane(e_id, o);
                    pos += (int)((cm >> o) & 1u) & ((io < e_id) ? 1 : 0);
                }
                if (cand && r == 0) { sm.ord[pos] = s; sm.csc[s] = sc; }
                rg_sync();
                if (lane == 0) {
                    const double* cp = sm.csc;
                    po_stl_prune<6>(sm.ord, ncand, W, [&](int slot) { return cp[slot]; });
                }
                rg_sync();
#pragma unroll
                for (int jx = 0; jx < 6; ++jx) sel[jx] = (jx < nbn) ? sm.ord[jx] : 0;
                rg_sync();
            }
            RT(0);
            rebuild(nbn, rec.x, rec.y, rec.z, rec.w, u, v);
            RT(1);
#ifdef PO_RING_TIMING
            after_event = true;
#endif
        }
        RT_SET(8);

        // ---------------------------------------------------------------- label of the top node
        if (st == PO_E_NOMEM && lane == 0) {   // out of tier-2 row groups: beam2d_kernel takes the pair
            a.meta[pi] = make_int2(PO_OK, X2_DEFERRED);
            a.queue[16] = 1;
        } else if (lane == 0) {
            int nout = 0;
            if (st == PO_OK) {
                int node = e_id;
                nout = e_depth;
                char* out = a.seq + a.seq_off[pi];
                const int cap = (int)(a.seq_off[pi + 1] - a.seq_off[pi]);
                if (nout > cap) { st = PO_E_CAP; nout = 0; }
                else
                    for (int i = nout - 1; i >= 0; --i) {
                        const int pk = apl[node];
                        out[i] = (char)((a.alphabet >> (8 * (po_node_last(pk) & 3))) & 0xffu);
                        node = po_node_parent(pk);
                    }
            }
            a.seq_len[pi] = nout;
            a.status[pi] = st;
        }
        if (a.upd_count != nullptr && lane == 0) { sm.nupd += cnt_ref; sm.nupd_x += cnt_x; }
        rg_sync();
        RT(2);
    }
#ifdef PO_RING_TIMING
    if (lane == 0 && a.dbg && blockIdx.x == 0)
        for (int i = 0; i < 32; ++i) a.dbg[i] = tk[i];
#endif
    if (lane == 0) {   // the next launch on this workspace continues from here
        unsigned long long* stp = a.wgstate + 2 * (size_t)blockIdx.x;
        stp[0] = a.magic ^ (unsigned long long)blockIdx.x;
        stp[1] = (unsigned long long)epoch;
        if (a.upd_count) { atomicAdd(a.upd_count, sm.nupd); atomicAdd(a.upd_count + 1, sm.nupd_x); }
    }
}

// resident workgroups per CU (LDS decides: 8)
extern "C" int po_ring_blocks_per_cu() {
#ifdef PO_EMU
    return 8;
#else
    static PoPerDeviceCache<1> per_cu;
    return per_cu.get(0, [] {
        int nblk = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)beam2d_ring_kernel, 64, 0) != hipSuccess || nblk <= 0) nblk = 8;
        if (const char* e = getenv("PO_RING_PER_CU")) { const int v = atoi(e); if (v > 0 && v < nblk) nblk = v; }
        if (getenv("PO_DEBUG_OCC")) fprintf(stderr, "[po] beam2d_ring_kernel: %d resident workgroups per CU, %zu B of LDS\n", nblk, sizeof(RingSmem));
        return nblk;
    });
#endif
}
extern "C" int po_ring_max_elements() { return RG_NRP; }
extern "C" int po_ring_ngl() { return RG_NGL; }
extern "C" void po_ring_launch(const void* x2args, int blocks, hipStream_t stream) {
    X2Args a = *(const X2Args*)x2args;
#ifdef PO_RING_TIMING
    static long long* dbg = nullptr;
    if (!dbg) { (void)hipMalloc((void**)&dbg, 32 * sizeof(long long)); }
    (void)hipMemsetAsync(dbg, 0, 32 * sizeof(long long), stream);
    a.dbg = dbg;
#endif
    hipLaunchKernelGGL(beam2d_ring_kernel, dim3(blocks), dim3(64), 0, stream, a);
#ifdef PO_RING_TIMING
    {
        long long h[32];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "[po_ring_timing] workgroup 0, 10 ns ticks.  steps after a steady prune | steps after a rebuild\n");
        const char* nm[4] = {"(pair setup / -)", "scan prologue", "scan loop", "catch-up test + score + prune test"};
        for (int i = 0; i < 4; ++i) fprintf(stderr, "   %-36s %12lld %12lld\n", nm[i], h[i], h[4 + i]);
        fprintf(stderr, "   ranking %lld, rebuild %lld, label walk + queue %lld\n", h[8], h[9], h[10]);
        fprintf(stderr, "   step top (record, catch-up tests) %lld; scan prologue up to the carried maximum: %lld (steady table) %lld (other)\n", h[29], h[27], h[28]);
        fprintf(stderr, "   loops: y + phase 1 %lld, phase 2 %lld, general %lld ticks\n", h[24], h[25], h[26]);
        fprintf(stderr, "   general main scans because of: a window beyond 31 times %lld (steady table) + %lld (new elements), the root's children %lld, other %lld\n", h[14], h[15], h[23], h[11]);
        fprintf(stderr, "   main steps: %lld after a steady prune, %lld after a rebuild; fast scans %lld: phase-1 iterations %lld, phase-2 iterations %lld; general scans %lld (iterations %lld); y reloads %lld\n",
                h[12], h[13], h[17], h[18], h[19], h[21], h[22], h[20]);
    }
#endif
}
