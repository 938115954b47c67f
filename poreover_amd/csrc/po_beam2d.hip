// Batched pair (2-D) CTC beam search: methods "row_col" and "row", with or without an envelope.
//
// Replaces decoding_cpp.cpp_beam_search_2d (decoding_cpp.pyx:107-139) -> beam_search(...)
// (BeamSearch.h:411-458) ->
//   method "row_col": beam_search_2d_by_row_col          (BeamSearch.h:262-397)   pair-decode CLI default
//   method "row":     beam_search_2d_by_row, envelope    (BeamSearch.h:110-172)   API default
//                     beam_search_2d_by_row, no envelope (BeamSearch.h:175-260)
// over the 2-D prefix trees of PrefixTree.h (:492-533 ctc, :578-633 flip-flop, :667-706 merge
// repeats) with Beam<..., node_greater_max_sym / node_greater_max> (Beam.h:20-38,93-108).
//
// What the reference does.  row_col walks the envelope along a diagonal (u, v): a MAIN step updates
// every element (the <= W beam nodes and their children) on read 0 over the look-ahead window
// [u, ce) and on read 1 over [v, re), then keeps the W elements with the largest
// max_t alpha0[t] + max_t alpha1[t]; a CATCH-UP step advances only u or v and updates only the beam
// nodes at that one time.  row visits every row u: beam nodes and children are updated on read 0 at
// time u only, every element is updated on read 1 over the whole row band [rs, re), and the score is
// alpha0[u] + max_t alpha1[t].  (Its `for b < beam_width` loop runs over a vector that GROWS while it
// is iterated, so while the beam is smaller than W, children pushed in the same row are expanded
// too — reproduced here, see "growing".)  Every update reads alpha of the node and of its parent at
// time t-1 from per-node std::unordered_map<int,double>s that are never erased; an absent entry
// reads as -inf.  Values written in one step are read again in later steps — by the same node
// (t = start-1), by beam nodes whose parent has left the beam (the parent's old values, "frozen"),
// and by children that become elements again when their parent re-enters the beam — so the maps
// ARE the algorithm's state and are reproduced exactly.
//
// Data layout on MI355X.
//   * VALUE STORE in HBM (L2-resident in practice): each node owns a ring row of R entries per read,
//     entry = {64-bit tag(epoch, node, t), K doubles}; a read hits iff the tag matches, so
//     "absent == -inf" needs no bookkeeping and rows are recycled without clearing.  R = pow2 >=
//     widest window + 2.  Rows are handed out per parent in groups of 4 (one per child symbol) and
//     recycled as soon as every time written into them lies below what any later step can read.
//   * The tree is an arena of packed (parent,last) words for the final label walk, plus first-child /
//     row-group words read only when a node (re-)enters the beam.
//   * Within a step the recurrence alpha[t] = lae(alpha_parent[t-1] + y[t][c], alpha[t-1] + y[t][blank])
//     is a wavefront over (depth, t): every element advances one t per iteration and takes its
//     parent's value of the previous iteration from an LDS exchange buffer (double-buffered, one
//     LDS hand-over per iteration).  Parents that do not move in the step (frozen, or the root's
//     blank prefix sums) are read straight from the store, one iteration ahead like the y rows.
//   * One workgroup per pair, thread = (read, element slot); the LDS layout is a compile-time struct
//     per beam-width class (W <= 6 / 12 / 25 -> 64 / 128 / 256 threads).  Workgroups are persistent
//     and pull pairs from an atomic queue.
#define PO_LAE_EARLY_TABLE 1   // (po_device.h; A/B in round 4: W = 10 -0.7 %, Bonito W = 5 -3 %, beam2d_kernel -1 %)
#define PO_LAE_BRANCHLESS 1    // (... the clamp instead of the small-argument test, and the two-instruction forms: Bonito W = 5
#define PO_LAE_TRIM 1          //  another -3 %, the others unchanged)
#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

#define PO_WANT_ZERO_KERNEL 1
#include "po_device.h"
#include "po_host.h"
#include "po_beam2d_common.h"

namespace {
constexpr int B2_YD = 256;   // doubles per read in the y window buffer (51 rows of 5, 32 rows of 8)
// The W <= 6 class (64 threads per pair) runs at 124 VGPRs, i.e. 4 waves per SIMD — if its LDS lets 16 workgroups
// share a CU.  The chains of dependent f64 operations in logaddexp leave a wave idle most of the time, so resident
// waves are what buys throughput here: 96 doubles of y rows per read (19 rows of 5) and 112 row groups bring the
// workgroup to 10 096 B (16 per CU) and the kernel from 65k to 85k pairs/s at 10 000 pairs.
//
// Waves per SIMD the W = 25 classes are compiled for (256-thread workgroups: a wave per SIMD each).  Unbounded the one-value kernel took
// 176 VGPRs, the three-value ones 204 - 206 — two workgroups per CU; at 168 (no VGPR spill / 12 - 13 spilled registers) three fit, LDS
// included: `row` / `row_col` W = 25, the Python API's defaults, 9.96k -> 13.76k / 11.28k -> 15.97k pairs/s; Bonito's and the flip-flop
// model's W = 25 launches + 26 .. 29 % (profiles/r06_ab_w25_occupancy.txt).  The W <= 12 class of the three-value models (128-thread
// workgroups, 195 - 198 VGPRs unbounded): flip-flop + 5 % at 168, merge repeats - 1 % — the first is bound, the second is not.
#ifndef PO_B2_W25_WAVES
#define PO_B2_W25_WAVES 3
#endif
#ifndef PO_B2_FF12_WAVES
#define PO_B2_FF12_WAVES 3
#endif
#ifndef B2_YD6
#define B2_YD6 96
#endif
#ifndef B2_NGL6
#define B2_NGL6 112
#endif
template <int MODEL, int WMAX>
struct B2Smem {
    static constexpr int K = (MODEL == PO_MODEL_CTC) ? 1 : 3;
    static constexpr int NCM = WMAX * (PO_A + 1);                      // element slots
    static constexpr int NCP = (NCM <= 32) ? 32 : (NCM <= 64 ? 64 : 128);  // threads per read
    int e[F_COUNT][NCM];    // element table: slots [0, nb) are the beam nodes in rank order
    int nx[F_COUNT][WMAX];  // next beam under construction
    int sel[WMAX];
    int cu_ps[WMAX];        // parent slots of the beam nodes in catch-up scans (parents among the beam nodes only)
    int stay[WMAX];         // beam slot was a beam slot in the previous main step (its children were elements then)
    // a beam node whose parent is no element any more: the parent's last value and its time, taken from the parent's
    // slot when it left (later times are absent, earlier ones are in the store).  fzt = INT_MAX: nothing captured.
    double fzv[2][WMAX][(MODEL == PO_MODEL_CTC) ? 1 : 3];
    int fzt[2][WMAX];
    int cnew[WMAX];         // its children got new rows in this step
    int badf[NCM];          // element may not skip redundant stores (see scan)
    int dup[NCM];
    // row groups tracked per pair: a step can open up to W new ones and a group lives for about a window
    static constexpr int NGL = (WMAX > 12) ? 768 : ((WMAX <= 6) ? B2_NGL6 : 256);
    int g_owner[NGL], g_hi0[NGL], g_hi1[NGL];
    int sh[20];
    double score[NCM];
    // (the window maxima of the scan just finished, mxs[r][s], live in xch[0][r][s][0]: the exchange buffer is idle
    //  between two scans, and xch[1] keeps the seeds of windows that continue)
    // incremental main steps: per (read, slot) the maximum of the previous main step's window, its time, and the time
    // from which the values of that window were non-increasing (ctd)
    double cmx[2][NCP];
    int cmt[2][NCP];
    int ctd[2][NCP];
    double xch[2][2][NCP][K];
    static constexpr int YD = (WMAX <= 6) ? B2_YD6 : B2_YD;
    double ybuf[2][YD];     // the y rows of the current step's windows, per read
    unsigned long long nupd;    // profiling: update_prob evaluations the reference's schedule makes for the current pair
    unsigned long long nupd_x;  // ... and those this kernel executed (incremental steps, no-op catch-ups)
    PoLaeTables lae;        // tables of the specialised logaddexp (po_device.h)
};

}  // namespace

// threads per workgroup for a beam-width class: 2 reads x (element slots padded to 32 / 64 / 128)
#define B2_THREADS(WM) ((WM) * (PO_A + 1) <= 32 ? 64 : ((WM) * (PO_A + 1) <= 64 ? 128 : 256))

// RC_ONLY: an instantiation without the `row` method's paths (is_row folds to false).  Same code, but the register
// allocator no longer has to cover both walks: 28 instead of 68 bytes of scratch per lane in the W <= 6 ctc kernel and
// +1.4 % on the headline configuration, so that one instantiation exists twice.
template <int MODEL, int WMAX, bool RC_ONLY = false>
// (4 waves per SIMD are what the W <= 6 class lives on — see B2_YD6.  It compiles to 124 / 128 VGPRs (ctc / merge
// repeats); an edit that costs four more registers costs a quarter of the throughput — check with
// -Rpass-analysis=kernel-resource-usage.  Forcing the bound here makes the allocator's choices 2 % worse today.)
// (The three-value models sit at 163 - 173 VGPRs in the W <= 6 class: 168 is the step between three waves per SIMD and
// two, so that bound is explicit too.)
__global__ __launch_bounds__(B2_THREADS(WMAX), ((MODEL == PO_MODEL_CTC && WMAX <= 12) ? 4 : (WMAX <= 6 ? 3 : ((MODEL == PO_MODEL_CTC || WMAX > 12) ? PO_B2_W25_WAVES : (MODEL == PO_MODEL_FLIPFLOP ? PO_B2_FF12_WAVES : 1))))) void beam2d_kernel(B2Args a) {
    using SM = B2Smem<MODEL, WMAX>;
    if (a.retry_flag != nullptr && *a.retry_flag == 0) return;   // retry pass and the first pass left nothing over: not one queue round trip
    constexpr int K = SM::K, NCM = SM::NCM, NCP = SM::NCP, nthr = 2 * NCP;
    using Ent = Entry<K>;
    __shared__ SM sm;
    const int tid = threadIdx.x;
    const int r = tid / NCP;           // read handled by this thread
    const int s = tid - r * NCP;       // element slot handled by this thread
    const int A = a.A, W = a.W, C = a.C;
    const int divA = (65536 + A - 1) / A;   // x / A == (x * divA) >> 16 for the slot numbers divided here (x < 1024, A <= 8)
    const bool is_row = RC_ONLY ? false : (a.method == PO_METHOD_ROW);

    // ---- per-workgroup workspace
    Ent* pool = (Ent*)(a.pool + (size_t)blockIdx.x * a.pool_bytes);
    const long long pool_entries = (long long)(a.pool_bytes / sizeof(Ent));
    int* apl = a.arena + (size_t)blockIdx.x * 5 * a.arena_cap;
    int* afc = apl + a.arena_cap;
    int* acrow = afc + a.arena_cap;
    double* amax = (double*)(acrow + a.arena_cap);  // row method: a node's max over its last non-empty row band
    double* cum0 = a.cum + (size_t)blockIdx.x * 2 * a.tcap;
    double* cum1 = cum0 + a.tcap;
    int* envt = a.envt + (size_t)blockIdx.x * 2 * a.vcap;
    // The store's tags carry a 16-bit epoch that is new for every pair a workgroup decodes, so entries of earlier
    // pairs never match and nothing has to be cleared between pairs.  The same holds BETWEEN LAUNCHES: the epoch
    // counter of a workgroup lives in the workspace next to a magic word naming the workspace geometry, a launch
    // continues where the previous one stopped, and the slice is cleared only when the word is not there (first
    // use of this memory) or when the 16 bits wrap around (every 65 535 pairs of this workgroup) — instead of a
    // memset of the whole store (8 GB for 4096 workgroups) before every launch.
    unsigned epoch = 0;
    auto clear_slice = [&]() {
        __syncthreads();
        for (long long i = tid; i < pool_entries; i += nthr) pool[i].tag = 0ull;
        __syncthreads();
    };
    {
        unsigned long long* stp = a.wgstate + 2 * (size_t)blockIdx.x;
        if (tid == 0) {
            const bool ok = (stp[0] == (a.magic ^ (unsigned long long)blockIdx.x));
            sm.sh[0] = ok ? 1 : 0;
            sm.sh[1] = ok ? (int)(unsigned)stp[1] : 0;
        }
        __syncthreads();
        const bool ok = sm.sh[0] != 0;
        epoch = (unsigned)sm.sh[1];
        __syncthreads();
        if (!ok) clear_slice();
    }
#ifdef PO_LAE_OCML
    const PoLaeOcml lae;
#else
    po_lae_tables_load(&sm.lae, tid, nthr);
    const PoLaeFast lae{&sm.lae};
#endif
#ifdef PO_B2_TIMING
    // three sets of counters: main steps that recompute their windows (0..11), incremental steps on the previous
    // step's table (12..23), incremental steps after a permutation of the beam (24..35); slot 11 of a set = its steps
    long long tk[37], tlast = 0;
    int tko = 0;
    for (int i = 0; i < 37; ++i) tk[i] = 0;
#define TK_START() do { tlast = wall_clock64(); } while (0)
#define TK(i) do { const long long n_ = wall_clock64(); tk[tko + (i)] += n_ - tlast; tlast = n_; } while (0)
#define TKC(i) do { tk[tko + (i)]++; } while (0)
#define TK_TYPE(o) do { tko = (o); } while (0)
#define TK_STEP() do { if (tko) tk[tko + 11]++; else tk[36]++; } while (0)
#else
#define TK_START() do {} while (0)
#define TK(i) do {} while (0)
#define TKC(i) do {} while (0)
#define TK_TYPE(o) do {} while (0)
#define TK_STEP() do {} while (0)
#endif

    for (;;) {
        // ---------------------------------------------------------------- next pair from the queue
        __syncthreads();
        if (tid == 0) {
            const int q = atomicAdd(a.queue, 1);
            sm.sh[0] = (a.order != nullptr && q < a.n) ? a.order[q] : q;   // longest pairs first (pair_order_kernel)
        }
        __syncthreads();
        const int pi = sm.sh[0];
        if (pi >= a.n) break;
        epoch++;
        if ((epoch & 0xffffu) == 0) { clear_slice(); epoch++; }   // (epoch 0 is what a cleared tag reads as)
        TK_START();
        if (a.only_meta && a.only_meta[pi].y != X2_DEFERRED) continue;  // done by beam2d_reg_kernel
        if (a.retry_nomem) {
            if (a.status[pi] != PO_E_NOMEM) continue;                  // decoded (or refused for good) by the first pass
        } else
        if (a.use_pre_status && a.status[pi] != PO_OK) {  // skipped upstream (pair_decode.py:372-375,395-398)
            if (tid == 0) a.seq_len[pi] = 0;
            continue;
        }
        const int64_t o1 = a.y1_off[pi], o2 = a.y2_off[pi];
        const int U = (int)(a.y1_off[pi + 1] - o1), V = (int)(a.y2_off[pi + 1] - o2);
        const double* yA = a.y1 + o1 * C;
        const double* yB = a.y2 + o2 * C;
        const double* yr_ = r ? yB : yA;
        const int32_t* env = a.env ? a.env + 2 * o1 : nullptr;
        int st = PO_OK;
        if (U < 1 || V < 1 || U > a.tcap || V > a.vcap || V > a.tcap || U >= (1 << 24) || V >= (1 << 24)) st = PO_E_ARG;
        if (!env && !is_row) st = PO_E_UNSUPPORTED;

        // ---------------------------------------------------------------- envelope pre-pass
        // bounds, widest row, and for row_col the transposed envelope (BeamSearch.h:270-284)
        int R = 32, NG = 0;
        if (st == PO_OK) {
            int bad = 0, wmax = env ? 0 : V;
            int unsup = 0, nonmono = 0;
            if (env)
                for (int u = tid; u < U; u += nthr) {
                    const int lo = env[2 * u], hi = env[2 * u + 1];
                    if (lo < hi && (lo < 0 || hi > V)) bad = 1;
                    if (lo < 0 || hi > V || (u > 0 && (lo < env[2 * u - 2] || hi < env[2 * u - 1]))) nonmono = 1;
                    // row method: a later row reads time rs-1, so row starts must not move backwards
                    // for the ring store to still hold it (every envelope the pipeline builds complies)
                    if (is_row && u > 0 && lo < env[2 * u - 2]) unsup = 1;
                    wmax = max(wmax, hi - lo);
                }
            const bool mono = !__syncthreads_or(nonmono);   // row bounds never move backwards (what build_envelope makes)
            if (!is_row)
                for (int x = tid; x < V; x += nthr) { envt[2 * x] = mono ? 0x7fffffff : -1; envt[2 * x + 1] = -1; }
            if (__syncthreads_or(bad)) st = PO_E_ENVELOPE;
            if (__syncthreads_or(unsup) && st == PO_OK) st = PO_E_UNSUPPORTED;
            if (st == PO_OK) {
                if (!is_row && mono) {
                    // the rows covering column x are then the contiguous range [a(x), b(x)]: row u is a(x) for the columns
                    // between the previous row's end and its own, and b(x) for those between its start and the next
                    // row's — every column is written once (as in beam2d_prepass_kernel)
                    for (int u = tid; u < U; u += nthr) {
                        const int lo = env[2 * u], hi = env[2 * u + 1];
                        const int hp = (u > 0) ? env[2 * u - 1] : 0, ln = (u + 1 < U) ? env[2 * u + 2] : V;
                        for (int x = hp; x < hi; ++x) envt[2 * x] = u;
                        for (int x = lo; x < ln; ++x) envt[2 * x + 1] = u;
                    }
                    __syncthreads();
                    for (int x = tid; x < V; x += nthr) {
                        const int a_ = envt[2 * x], b_ = envt[2 * x + 1];
                        if (a_ != 0x7fffffff && b_ >= a_) { envt[2 * x + 1] = b_ + 1; wmax = max(wmax, b_ + 1 - a_); }
                        else { envt[2 * x] = -1; envt[2 * x + 1] = -1; }
                    }
                } else if (!is_row) {
                    // each column x is always visited by thread x % nthr, rows in order: race-free
                    for (int u = 0; u < U; ++u) {
                        const int lo = env[2 * u], hi = env[2 * u + 1];
                        int x = lo + ((tid - lo) % nthr + nthr) % nthr;
                        for (; x < hi; x += nthr) {
                            if (envt[2 * x] < 0) { envt[2 * x] = u; envt[2 * x + 1] = u + 1; }
                            else envt[2 * x + 1]++;
                        }
                    }
                    __syncthreads();
                    for (int x = tid; x < V; x += nthr) wmax = max(wmax, envt[2 * x + 1] - envt[2 * x]);
                }
                if (tid == 0) sm.sh[1] = 0;
                __syncthreads();
                atomicMax(&sm.sh[1], wmax);
                __syncthreads();
                wmax = sm.sh[1];
                while (R < wmax + 2) R <<= 1;
                const long long ng = pool_entries / ((long long)PO_A * 2 * R);
                NG = (int)min((long long)SM::NGL, ng);
                if (NG < 2 * max(W, PO_A) + 4) st = PO_E_NOMEM;  // band too wide for the per-pair value store
            }
        }
        {   // node budget: every processed element creates at most A nodes per step
            const long long steps = is_row ? U : min(U, V);
            const long long need = 1 + A + (long long)A * max(W, A) * (steps + 1);
            if (st == PO_OK && (need > a.arena_cap || need >= (1 << 24))) st = PO_E_NOMEM;
        }
        if (st != PO_OK) {
            if (tid == 0) { a.status[pi] = st; a.seq_len[pi] = 0; if (st == PO_E_NOMEM) a.queue[8] = 1; }
            continue;
        }
        const int Rm = R - 1;

        // blank prefix sums of both reads = the CTC root's alpha (PrefixTree.h:509-515); serial in t so
        // the rounding is the reference's
        // (the first 32 / 64 threads of each read load that many frames at a time, coalesced, and add them in lane
        // order through v_readlane — one wave holds both reads when a read has 32 threads)
        {
            constexpr int SL = (NCP < 64) ? NCP : 64;
            const int Tn = r ? V : U, Tw = (NCP < 64) ? max(U, V) : Tn;   // (wave-uniform trip count)
            if (MODEL == PO_MODEL_CTC && s < SL) {
                double* cw = r ? cum1 : cum0;
                double acc = 0.0;
                for (int t0 = 0; t0 < Tw; t0 += SL) {
                    const int t = t0 + s;
                    const double x = (t < Tn) ? yr_[(int64_t)t * C + A] : 0.0;
                    double mine = 0.0;
#pragma unroll 4
                    for (int j = 0; j < SL; ++j) {   // (not unrolled further: the beam-width classes are register-tight)
                        double xj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), j),
                                                     __builtin_amdgcn_readlane(__double2loint(x), j));
                        if (NCP < 64) {   // lanes 32.. of the wave belong to read 1
                            const double xb = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), (j + 32) & 63),
                                                               __builtin_amdgcn_readlane(__double2loint(x), (j + 32) & 63));
                            xj = r ? xb : xj;
                        }
                        if (t0 + j < Tn) acc += xj;
                        if (s == j) mine = acc;
                    }
                    if (t < Tn) cw[t] = mine;
                }
            }
        }
        for (int g = tid; g < SM::NGL; g += nthr) { sm.g_owner[g] = -1; sm.g_hi0[g] = 0; sm.g_hi1[g] = 0; }
        __syncthreads();

        // ---------------------------------------------------------------- store access helpers
        auto st_read = [&](int row, int rr, int t, int node, double* out) {
            bool hit = false;
            if (t >= 0 && row >= 0) {
                const Ent e = pool[((size_t)row * 2 + rr) * R + (t & Rm)];
                hit = (e.tag == make_tag(epoch, node, t));
                if (hit) {
#pragma unroll
                    for (int k = 0; k < K; ++k) out[k] = e.v[k];
                }
            }
            if (!hit) {
#pragma unroll
                for (int k = 0; k < K; ++k) out[k] = PO_NEG_INF;
            }
        };
        auto st_write = [&](int row, int rr, int t, int node, const double* v) {
#ifdef PO_ABL_NOSTORE   // timing ablation only (results are wrong): keeps the value live, skips the store
            if (v[0] == 12345.678) pool[0].tag = 1;
            return;
#endif
            Ent e;
            e.tag = make_tag(epoch, node, t);
#pragma unroll
            for (int k = 0; k < K; ++k) e.v[k] = v[k];
            pool[((size_t)row * 2 + rr) * R + (t & Rm)] = e;
        };
        // values of the ROOT at time t (tree constructors, PrefixTree.h:499-516,585-598,674-688)
        auto root_at = [&](int rr, int t, double* out) {
            if (MODEL == PO_MODEL_CTC) {
                out[0] = (t < 0) ? 0.0 : (rr ? cum1 : cum0)[t];
            } else {
                double tmp[3];
                root_values<MODEL>(t, 0.0, tmp);
#pragma unroll
                for (int k = 0; k < K; ++k) out[k] = tmp[k];
            }
        };
        // a free row group: nothing a later step can read is stored in it.  lo0 / lo1 = smallest
        // time a later step can still read on read 0 / 1.  Without an envelope (row method) the read-1
        // band restarts at 0 in every row, so time never retires read-1 values; there a value can
        // only be read again as a frozen parent's, i.e. through a beam node's own row, its parent's
        // row or its children's group, and a group nobody in the beam points to is free.
        auto alloc_group = [&](int owner, int lo0, int lo1, int nbeam) -> int {
            int cur = sm.sh[3];
            int g = -1;
            for (int tries = 0; tries < NG; ++tries) {
                const int c = cur;
                cur = (cur + 1 == NG) ? 0 : cur + 1;
                bool free_ = sm.g_owner[c] < 0 || (sm.g_hi0[c] <= lo0 && sm.g_hi1[c] <= lo1);
                if (!free_ && !env && sm.g_hi0[c] <= lo0) {
                    bool ref = false;
                    for (int j = 0; j < nbeam; ++j)
                        ref |= (sm.e[F_ROW][j] / PO_A == c) || (sm.e[F_PROW][j] >= 0 && sm.e[F_PROW][j] / PO_A == c) ||
                               (sm.e[F_CROW][j] == c);
                    free_ = !ref;
                }
                if (free_) { g = c; break; }
            }
            sm.sh[3] = cur;
            if (g < 0) { sm.sh[4] = PO_E_NOMEM; g = 0; }
            sm.g_owner[g] = owner;
            sm.g_hi0[g] = 0; sm.g_hi1[g] = 0;
            return g;
        };

        // ---------------------------------------------------------------- tree + beam initialisation
        // root = node 0; its A children = nodes 1..A in row group 0 (BeamSearch.h:119-127,286-293)
        if (tid == 0) {
            apl[0] = po_pack_node(-1, A); afc[0] = 1; acrow[0] = 0;
            sm.g_owner[0] = 0; sm.g_hi0[0] = 1; sm.g_hi1[0] = 1;  // the root's children hold values at t = 0
            sm.sh[2] = 1 + A;  // next node id
            sm.sh[3] = 1;      // group allocation cursor
            sm.sh[4] = PO_OK;
            sm.sh[8] = INT_MIN; sm.sh[9] = INT_MIN;  // window ends of the previous row_col main step: none yet
            sm.sh[10] = 0; sm.sh[11] = 0; sm.sh[12] = 0; sm.sh[13] = 0; sm.sh[14] = 0; sm.sh[15] = 0;  // incremental steps: nothing to build on yet
            sm.sh[16] = INT_MIN; sm.sh[17] = INT_MIN;
            sm.sh[18] = INT_MIN; sm.sh[19] = INT_MIN;   // the latest time any main scan computed on read 0 / 1
        }
        if (tid < WMAX) { sm.stay[tid] = 0; sm.fzt[0][tid] = INT_MAX; sm.fzt[1][tid] = INT_MAX; }
        if (tid < A) {
            apl[1 + tid] = po_pack_node(0, tid); afc[1 + tid] = -1; acrow[1 + tid] = -1;
            sm.e[F_ID][tid] = 1 + tid; sm.e[F_ROW][tid] = tid; sm.e[F_PROW][tid] = -1; sm.e[F_PAR][tid] = 0;
            sm.e[F_GPAR][tid] = -1; sm.e[F_SYM][tid] = sym_pack(tid, A, true); sm.e[F_DEPTH][tid] = 1;
            sm.e[F_FC][tid] = -1; sm.e[F_CROW][tid] = -1;
        }
        if (s < A) {  // update_prob(n, r, 0) for both reads
            double sp[3] = {PO_NEG_INF, PO_NEG_INF, PO_NEG_INF}, pp[3], out[3];
            root_at(r, -1, pp);
            const double ya = yr_[s], yb = (MODEL == PO_MODEL_FLIPFLOP) ? yr_[s + A] : yr_[A];
            po_update<MODEL>(sp, pp, ya, yb, false, true, out, lae);
            st_write(s, r, 0, 1 + s, out);
            if (r == 1) amax[1 + s] = out[0];
        }
        int nb = A;  // beam size
        if (tid == 0) { sm.nupd = 0; sm.nupd_x = 0; }  // update_prob evaluations of this pair (profiling only; kept in LDS)
        __syncthreads();
        TK(0);  // pre-pass + init

        // ---------------------------------------------------------------- one scan
        // Every participating element advances over its window, parent values flowing through the
        // LDS exchange buffer.  t0x / lenx: window start / length on read x (len 0 = read untouched);
        // slots [skip_lo, skip_hi) do not move on read 0 (row method: beam nodes beyond the first W).
        auto scan = [&](bool is_main, bool reuse, int nelem, int skip_lo, int skip_hi, int t00, int len0_, int t01, int len1_) {
            // the window bounds come from (wave-uniform) vector loads: pin them to SGPRs so the iteration
            // loop is a scalar loop
            // INCREMENTAL main step (restricted to the simplest case): the previous prune left
            // the beam exactly as it was and nothing is created or re-allocated now, so every element sits in the slot
            // it had, and over the part of its windows the previous main step covered it would recompute the stored
            // bits.  Only the times beyond the previous window ends are computed (seeded from the store); the window
            // maximum of the rest is carried per (read, slot), or re-read from the store when its time has left the
            // window.  sh[12 + r]: cmx / cmt of read r were written for the current slot layout.
            const bool steady = is_main && reuse && sm.sh[10] != 0 && sm.sh[11] != 0;
            int l0 = len0_, l1 = len1_, s0 = t00, s1 = t01;
            if (steady && sm.sh[12] != 0) { s0 = min(max(sm.sh[8], t00), t00 + len0_); l0 = t00 + len0_ - s0; }
            if (steady && sm.sh[13] != 0) { s1 = min(max(sm.sh[9], t01), t01 + len1_); l1 = t01 + len1_ - s1; }
            const int len0 = __builtin_amdgcn_readfirstlane(l0), len1 = __builtin_amdgcn_readfirstlane(l1);
            const int t0 = r ? s1 : s0, len = r ? len1 : len0;
            const int t0f = r ? t01 : t00, lenf = r ? len1_ : len0_;   // the full window
            const bool partf = (s < nelem) && (lenf > 0) && !(r == 0 && s >= skip_lo && s < skip_hi);
            const bool part = partf && (len > 0);
            const int Lmax = max(len0, len1);
            // Everything this lane's slot holds, read unconditionally in one go (one LDS round trip instead of one per
            // block below; the slots beyond the table read harmless neighbours).
            const int sx = (s < SM::NCM) ? s : SM::NCM - 1;
            const double l_pm = sm.cmx[r][s];
            const int l_pt = sm.cmt[r][s], l_td = sm.ctd[r][s];
            const int l_id = sm.e[F_ID][sx], l_sy = sm.e[F_SYM][sx], l_row = sm.e[F_ROW][sx];
            const int l_ps = is_main ? sm.e[F_PSLOT][sx] : sm.cu_ps[(s < WMAX) ? s : WMAX - 1];
            const int l_prow = sm.e[F_PROW][sx], l_par = sm.e[F_PAR][sx];
            const int l_sdt = sm.sh[16 + r];
            double l_seed[K];
#pragma unroll
            for (int q = 0; q < K; ++q) l_seed[q] = sm.xch[1][r][s][q];
            int pslot = PS_ROOT, sym = 0;
            bool same = false, rootpar = false;
            double self[K], mx = PO_NEG_INF;
            int mt = -1;
            int td = t0f;   // the values of this window are non-increasing from time td on (maintained as they are computed)
            Ent wm_e;                    // a window maximum still on its way from the store (see below)
            unsigned long long wm_tag = 0;
            bool wm_pend = false;
            wm_e.tag = 0;
#pragma unroll
            for (int q = 0; q < K; ++q) wm_e.v[q] = PO_NEG_INF;
            if (partf && t0 > t0f) {   // the carried part [t0f, t0) of the window
                const double pm = l_pm;
                const int pt = l_pt;
                td = l_td;
#ifdef PO_ABL_NOREREAD   // timing ablation only (results are wrong)
                if (true) { mx = pm; mt = pt; }
#else
                if (pm == PO_NEG_INF || (pt >= t0f && pt < t0)) { mx = pm; mt = pt; }
#endif
                else {
                    // the carried maximum's time has left the window
                    const Ent* rp = pool + ((size_t)l_row * 2 + r) * R;
                    const unsigned long long tg = make_tag(epoch, l_id, 0);
                    if (td <= t0f) {
                        // ... whose carried part is non-increasing (a node past its peak decays frame by frame — the usual
                        // case): its maximum is its first value, one entry of the element's own ring row.  Requested
                        // here, looked at after the y rows have been waited for (one round trip instead of two).
                        wm_e = rp[t0f & Rm];
                        wm_tag = tg + (unsigned)t0f;
                        wm_pend = true;
                        mt = t0f;
                    } else {
                        // ... otherwise stored values are read back — only those of [t0f, td]: from td on the values fall,
                        // so x[td] is the maximum of the rest.  Eight loads in flight; no tag to compare (every value of
                        // the carried part is there: this element computed it in the steps since its window was last
                        // computed from its start, and the ring holds a whole window).
                        double pv = PO_NEG_INF;
                        const int te = min(td + 1, t0);
                        td = t0f;
#ifdef PO_RR4
                        for (int bt = t0f; bt < te; bt += 4) {
                            double v8[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) v8[q] = rp[(bt + q) & Rm].v[0];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
#else
                        for (int bt = t0f; bt < te; bt += 8) {
                            double v8[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) v8[q] = rp[(bt + q) & Rm].v[0];
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
#endif
                                const int tq = bt + q;
                                if (tq < te) {
                                    if (v8[q] >= mx) { mx = v8[q]; mt = tq; }
                                    if (tq > t0f && v8[q] > pv) td = tq;
                                    pv = v8[q];
                                }
                            }
                        }
                    }
                }
            }
            Ent* myrow = pool;          // this element's ring row on read r
            const Ent* prow = pool;     // a frozen parent's ring row
            unsigned long long tag0 = 0, ptag0 = 0;
            if (part) {
                const int node = l_id;
                pslot = l_ps;
                const int sy = l_sy;
                sym = sym_last(sy); same = (sym_plast(sy) == sym); rootpar = (sy >> 9) & 1;
                myrow = pool + ((size_t)l_row * 2 + r) * R;
                tag0 = make_tag(epoch, node, 0);
                // The value at t0 - 1.  A window that continues where this read's elements last computed (same slot
                // layout; sh[16 + r] = the time after the last value a main scan computed on read r) starts from the
                // value this lane computed last, which every main scan leaves in xch[1] (below) — no round trip to the
                // store.  (Window ends can move backwards — envelopes with a wide row now and then: the continuation
                // then starts before the last computed time, and the seed comes from the store.)  Catch-up scans only
                // touch the slots of the read they advance, and a read they advance restarts its windows from t0f.
                if (t0 > t0f && t0 == l_sdt) {
#pragma unroll
                    for (int k = 0; k < K; ++k) self[k] = l_seed[k];
                } else {
                    bool hit = false;
                    if (t0 >= 1) {
                        const Ent e = myrow[(t0 - 1) & Rm];
                        hit = (e.tag == tag0 + (unsigned)(t0 - 1));
#pragma unroll
                        for (int k = 0; k < K; ++k) self[k] = e.v[k];
                    }
                    if (!hit) {
#pragma unroll
                        for (int k = 0; k < K; ++k) self[k] = PO_NEG_INF;
                    }
#pragma unroll
                    for (int k = 0; k < K; ++k) sm.xch[1][r][s][k] = self[k];
                }
                if (pslot == PS_FROZEN) {
                    prow = pool + ((size_t)l_prow * 2 + r) * R;
                    ptag0 = make_tag(epoch, l_par, 0);
                }
            }
            // A frozen parent's values: nothing after its last one, and that one was captured when the parent left (see
            // prune_and_advance) — a beam node with a frozen parent (about three of the five, step after step) then never
            // touches the store in the iterations, and the loop below keeps no vector-memory load to wait for.
            bool fz_fast = false;
            int fz_tt = INT_MAX;
            double fz_v[K];
#pragma unroll
            for (int q = 0; q < K; ++q) fz_v[q] = PO_NEG_INF;
            if (is_main && !is_row && part && pslot == PS_FROZEN && s < WMAX) {
                fz_tt = sm.fzt[r][s];
#pragma unroll
                for (int q = 0; q < K; ++q) fz_v[q] = sm.fzv[r][s][q];
                fz_fast = (t0 - 1 >= fz_tt);
            }
            // Redundant stores (row_col main steps, `reuse`).  An element that was an element in the previous
            // main step too recomputes, over the part of its window that step already covered, exactly the
            // bits that are stored — if every ancestor of it inside the element set is in the same situation.
            // Beam nodes always come out of the previous step's element set; children were elements iff their
            // parent was a beam node then and kept its row group.  badf marks the others and is pushed down
            // the parent links; everyone else only stores from the previous window end on.
            bool bad = true;
            if (steady) bad = false;   // (every element was an element, in this slot, in the previous main step)
            else if (reuse) {
                bad = false;
                if (s < nelem && s >= nb) bad = !(sm.stay[pslot] && !sm.cnew[pslot]);
                if (r == 0 && s < nelem) sm.badf[s] = bad;
                b2_sync_lds<nthr>();
                for (;;) {
                    int changed = 0;
                    if (s < nelem && !bad && pslot >= 0 && sm.badf[pslot]) { bad = true; changed = 1; }
                    if (!b2_any<nthr>(changed)) break;
                    if (r == 0 && bad && s < nelem) sm.badf[s] = 1;
                    b2_sync_lds<nthr>();
                }
            }
            const int sfrom = bad ? INT_MIN : sm.sh[8 + r];  // first time whose value must be written
            const int ca = sym, cb = (MODEL == PO_MODEL_FLIPFLOP) ? sym + A : A;
            // The value at t-1 of a parent that does not move in this scan (frozen: from its ring row; root:
            // its closed form) is loaded one iteration ahead (rare).
            double pr_n[K];
            Ent pe_n;
            pe_n.tag = 0;
#pragma unroll
            for (int q = 0; q < K; ++q) { pr_n[q] = PO_NEG_INF; pe_n.v[q] = PO_NEG_INF; }
            auto fetch = [&](int k) {
                const int tp = t0 + k - 1;
                if (pslot == PS_ROOT) root_at(r, tp, pr_n);
                else if (tp >= 0) pe_n = prow[tp & Rm];
            };
            if (part && pslot < 0 && !fz_fast) fetch(0);
            // (wave-uniform: most scans have no such element, and then the iterations skip its handling with one
            //  scalar branch instead of two masked blocks)
            const bool any_static = __ballot(part && pslot < 0 && !fz_fast) != 0ull;
            // The y rows of both windows go through LDS, B2_YD doubles per read at a time (buffer row =
            // iteration index).  With no vector-memory LOAD left in the iteration loop the wave never
            // waits there for the acknowledgement of its value-store writes: vmcnt counts loads and stores
            // in order, so waiting for any load also waits for every store issued before it.
            const int yrows = SM::YD / C;
            for (int k0 = 0; k0 < Lmax; k0 += yrows) {
            {
                const int nrow = min(len, k0 + yrows) - k0;  // this read's rows in the chunk (<= 0: none)
                const double* src = yr_ + (int64_t)(t0 + k0) * C;
#ifdef PO_ABL_NOYLOAD     // timing ablation only (results are wrong)
                if (nrow * C > 1000000) sm.ybuf[r][s] = src[s];
#else
                for (int i = s; i < nrow * C; i += NCP) sm.ybuf[r][i] = src[i];
#endif
            }
            b2_sync_lds<nthr>();  // y rows (and, first time, the seeds in xch[1]) -> visible to the iterations
            if (wm_pend) { mx = (wm_e.tag == wm_tag) ? wm_e.v[0] : PO_NEG_INF; wm_pend = false; }
            TK(is_main ? 3 : 7);  // scan: self read + y rows
            const int kchunk = min(Lmax, k0 + yrows);
            const double* ypa = sm.ybuf[r] + ca;   // this lane's two y entries of the chunk's first row; one row on per iteration
            const double* ypb = sm.ybuf[r] + cb;
            const int xsl = (pslot >= 0) ? pslot : s;
            int tr = INT_MIN;
            for (int kv = k0; kv < kchunk; ++kv) {
                const int k = __builtin_amdgcn_readfirstlane(kv);  // keeps the loop counter and branch scalar
                if (part && k < len) {
                    const int t = t0 + k;
                    const double ya = *ypa, yb = *ypb;
                    ypa += C; ypb += C;
                    double pp[K], out[K];
#pragma unroll
                    for (int q = 0; q < K; ++q) pp[q] = sm.xch[(k + 1) & 1][r][xsl][q];
                    if (fz_fast) {
#pragma unroll
                        for (int q = 0; q < K; ++q) pp[q] = (t - 1 == fz_tt) ? fz_v[q] : PO_NEG_INF;
                    }
                    if (any_static) {
                        if (pslot < 0 && !fz_fast) {  // rare: the parent does not move in this scan
                            const bool hit = (t >= 1) && (pe_n.tag == ptag0 + (unsigned)(t - 1));
#pragma unroll
                            for (int q = 0; q < K; ++q) pp[q] = (pslot == PS_ROOT) ? pr_n[q] : (hit ? pe_n.v[q] : PO_NEG_INF);
                        }
                        if (pslot < 0 && !fz_fast && k + 1 < len) fetch(k + 1);
                    }
                    po_update<MODEL>(self, pp, ya, yb, same, rootpar && t == 0, out, lae);
                    // direct 16-byte store per lane.  (Tried: buffering 8 iterations in LDS and flushing
                    // row-contiguous 128-byte bursts to cut L2 requests — the flush's extra instructions
                    // and LDS cost more than the coalescing saved: 22k vs 29k pairs/s.  DESIGN.md §3.3.)
#ifdef PO_ABL_NOSTORE   // timing ablation only (results are wrong): keeps the value live, skips the store
                    if (out[0] == 12345.678) pool[0].tag = 1;
#else
                    if (t >= sfrom) {
                        Ent e;
                        // (t < 2^24 sits in the low word of the tag: a 32-bit add, no carry)
                        e.tag = (tag0 & 0xffffffff00000000ull) | (unsigned)((unsigned)tag0 + (unsigned)t);
#pragma unroll
                        for (int q = 0; q < K; ++q) e.v[q] = out[q];
                        myrow[(unsigned)(t & Rm)] = e;
                    }
#endif
                    if (out[0] > self[0]) tr = t;   // the last time a value rose (folded into td after the loop)
#pragma unroll
                    for (int q = 0; q < K; ++q) { self[q] = out[q]; sm.xch[k & 1][r][s][q] = out[q]; }
                    mt = (out[0] >= mx) ? t : mt;
                    asm("v_max_f64 %0, %1, %2" : "=v"(mx) : "v"(mx), "v"(out[0]));   // (no NaNs here; the builtin canonicalises both operands)
                }
                b2_sync_lds<nthr>();  // only xch crosses iterations; the stores stay in flight
            }
            td = max(td, tr);   // (times may be computed again after a window end moved back: td never decreases)
            TK(is_main ? 5 : 9);  // scan: iterations
            }
            if (wm_pend) mx = (wm_e.tag == wm_tag) ? wm_e.v[0] : PO_NEG_INF;   // (a scan without new times)
            if (s < nelem) sm.xch[0][r][s][0] = mx;   // "mxs"
            if (is_main && partf) { sm.cmx[r][s] = mx; sm.cmt[r][s] = mt; sm.ctd[r][s] = td; }
            if (is_main && part) {   // the last value of this window: the seed of a window that continues it
#pragma unroll
                for (int q = 0; q < K; ++q) sm.xch[1][r][s][q] = self[q];
            }
            if (is_main && tid == 0) {
                if (len0_ > 0) sm.sh[12] = 1;
                if (len1_ > 0) sm.sh[13] = 1;
                if (len0 > 0) { sm.sh[16] = s0 + len0; sm.sh[18] = max(sm.sh[18], s0 + len0); }   // xch[1] of read 0 / read 1 now holds the values at these times - 1
                if (len1 > 0) { sm.sh[17] = s1 + len1; sm.sh[19] = max(sm.sh[19], s1 + len1); }
            }
            if (a.upd_count != nullptr && tid == 0) {   // profiling (skipped slots counted too)
                sm.nupd += (unsigned)(nelem * (len0_ + len1_));
                sm.nupd_x += (unsigned)(nelem * (len0 + len1));
            }
        };

        // parent slot of beam slot j: the parent is an element if it is a beam node or a child of a
        // beam node (grand-parent in the beam); otherwise it does not move (PS_FROZEN) or is the root
        auto beam_parent = [&](int j, int nbm, bool with_children) -> int {
            const int par = sm.e[F_PAR][j];
            if (par == 0) return PS_ROOT;
            int ps = PS_FROZEN;
#pragma unroll 4
            for (int i = 0; i < nbm; ++i) if (sm.e[F_ID][i] == par) ps = i;
            if (ps < 0 && with_children) {
                const int gp = sm.e[F_GPAR][j];
#pragma unroll 4
                for (int i = 0; i < nbm; ++i) if (sm.e[F_ID][i] == gp) ps = nbm + A * i + sym_plast(sm.e[F_SYM][j]);
            }
            return ps;
        };

        // expansion of the processed elements + children slots.  Regular shape: every beam node is
        // processed, child c of beam node j sits at nb + A*j + c.  lo0/lo1: see alloc_group.
        bool steady_tbl = false;   // the last build_regular took the steady-table path (nothing allocated: it cannot fail)
        auto build_regular = [&](int lo0, int lo1, int hi0, int hi1) -> int {
            // STEADY TABLE.  After a prune that left the beam exactly as it was (same nodes, same slots: sh[14]), every
            // beam node was expanded in the previous main step and its children's row group was marked with that
            // step's window ends — later than anything a new allocation would have treated as free since — so nothing
            // is created, nothing re-allocated, and the element table of the previous step IS this step's, entry for
            // entry (catch-up scans keep their parent slots elsewhere).  Only the window ends written into the row
            // groups move on.
            steady_tbl = (sm.sh[14] != 0 && sm.sh[15] != 0);
            if (steady_tbl) {
                if (tid < nb) {
                    const int gc = sm.e[F_CROW][tid], go = sm.e[F_ROW][tid] / PO_A;
                    atomicMax(&sm.g_hi0[gc], hi0); atomicMax(&sm.g_hi1[gc], hi1);
                    atomicMax(&sm.g_hi0[go], hi0); atomicMax(&sm.g_hi1[go], hi1);
                }
                if (tid < nb) sm.cnew[tid] = 0;
                if (tid == 0) sm.sh[11] = 1;
                b2_sync_lds<nthr>();
                return nb * (A + 1);
            }
            // one lane per beam node (nb <= 25: the first wave).  New node ids are handed out in beam order — a prefix
            // count over the lanes that need them (ids break score ties); row groups, which only name storage, are
            // allocated one after the other by lane 0 and marked with this step's window ends at once.
            if (tid < 64) {
                const bool bl = tid < nb;
                bool isnew = false, need_group = false;
                int id = 0;
                if (bl) {
                    id = sm.e[F_ID][tid];
                    const int cr = sm.e[F_CROW][tid];
                    if (sm.e[F_FC][tid] < 0) { isnew = true; need_group = true; }
                    else if (cr < 0 || cr >= NG || sm.g_owner[cr] != id) need_group = true;  // old rows recycled: all dead
                }
                const unsigned long long bn = __ballot(isnew), bg = __ballot(need_group);
                const int base = sm.sh[2];
                if (isnew) {
                    // (lanes below this one that create nodes: mbcnt, not a 64-bit lane mask that lives in two registers)
                    const int fc = base + A * (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bn >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bn, 0u));
                    sm.e[F_FC][tid] = fc;
                    afc[id] = fc;
                    for (int c = 0; c < A; ++c) {
                        apl[fc + c] = po_pack_node(id, c); afc[fc + c] = -1; acrow[fc + c] = -1;
                        if (is_row) amax[fc + c] = PO_NEG_INF;
                    }
                }
                if (bl) sm.cnew[tid] = need_group ? 1 : 0;
                b2_sync_lds<64>();
                if (tid == 0) {
                    sm.sh[11] = (bg == 0) ? 1 : 0;   // nothing created, no row group re-allocated in this step
                    sm.sh[2] = base + A * __popcll(bn);
                    for (unsigned long long m = bg; m != 0; m &= m - 1) {
                        const int j = __builtin_ctzll(m);
                        const int idj = sm.e[F_ID][j];
                        const int g = alloc_group(idj, lo0, lo1, nb);
                        sm.e[F_CROW][j] = g;
                        acrow[idj] = g;
                        sm.g_hi0[g] = hi0; sm.g_hi1[g] = hi1;
                    }
                }
                b2_sync_lds<64>();
                if (bl) {   // this step writes up to (hi0, hi1) into the children's rows and the node's own
                    const int gc = sm.e[F_CROW][tid], go = sm.e[F_ROW][tid] / PO_A;
                    atomicMax(&sm.g_hi0[gc], hi0); atomicMax(&sm.g_hi1[gc], hi1);
                    atomicMax(&sm.g_hi0[go], hi0); atomicMax(&sm.g_hi1[go], hi1);
                }
            }
            b2_sync_lds<nthr>();
            const int ne = nb * (A + 1);
            if (tid == 0) sm.sh[15] = 1;   // a regular table now stands in e[]
            if (tid < ne) {
                if (tid < nb) {
                    sm.e[F_PSLOT][tid] = beam_parent(tid, nb, true);
                } else {
                    const int j = ((tid - nb) * divA) >> 16, c = (tid - nb) - j * A;
                    sm.e[F_ID][tid] = sm.e[F_FC][j] + c; sm.e[F_ROW][tid] = sm.e[F_CROW][j] * PO_A + c;
                    sm.e[F_SYM][tid] = sym_pack(c, sym_last(sm.e[F_SYM][j]), false);
                    sm.e[F_PSLOT][tid] = j; sm.e[F_FC][tid] = -2; sm.e[F_CROW][tid] = -2;
                }
            }
            return ne;
        };
        // growing shape (row method while the beam is smaller than W, BeamSearch.h:132-144): the loop
        // `for b < beam_width` runs over the vector it is appending to, so children pushed earlier
        // in the same row are processed (updated on read 0, expanded) as well.  Serial; first row only.
        auto build_growing = [&](int lo0, int lo1, int hi0, int hi1, int* nproc) -> int {
            if (tid == 0) {
                int next_id = sm.sh[2], ne = nb, np = 0;
                for (int b = 0; b < W && b < ne; ++b) {
                    np = b + 1;
                    const int id = sm.e[F_ID][b];
                    if (sm.e[F_FC][b] == -2) { sm.e[F_FC][b] = afc[id]; sm.e[F_CROW][b] = acrow[id]; }
                    bool need_group = false;
                    if (sm.e[F_FC][b] < 0) {
                        sm.e[F_FC][b] = next_id;
                        afc[id] = next_id;
                        for (int c = 0; c < A; ++c) {
                            apl[next_id + c] = po_pack_node(id, c); afc[next_id + c] = -1; acrow[next_id + c] = -1;
                            amax[next_id + c] = PO_NEG_INF;
                        }
                        next_id += A;
                        need_group = true;
                    } else if (sm.e[F_CROW][b] < 0 || sm.e[F_CROW][b] >= NG || sm.g_owner[sm.e[F_CROW][b]] != id) {
                        need_group = true;
                    }
                    if (need_group) {
                        const int g = alloc_group(id, lo0, lo1, ne);
                        sm.e[F_CROW][b] = g;
                        acrow[id] = g;
                    }
                    const int gc = sm.e[F_CROW][b], go = sm.e[F_ROW][b] / PO_A;
                    sm.g_hi0[gc] = max(sm.g_hi0[gc], hi0); sm.g_hi1[gc] = max(sm.g_hi1[gc], hi1);
                    sm.g_hi0[go] = max(sm.g_hi0[go], hi0); sm.g_hi1[go] = max(sm.g_hi1[go], hi1);
                    for (int c = 0; c < A && ne < NCM; ++c, ++ne) {
                        sm.e[F_ID][ne] = sm.e[F_FC][b] + c; sm.e[F_ROW][ne] = gc * PO_A + c;
                        sm.e[F_SYM][ne] = sym_pack(c, sym_last(sm.e[F_SYM][b]), false);
                        sm.e[F_PSLOT][ne] = b; sm.e[F_FC][ne] = -2; sm.e[F_CROW][ne] = -2;
                        sm.e[F_PAR][ne] = id; sm.e[F_GPAR][ne] = sm.e[F_PAR][b]; sm.e[F_PROW][ne] = sm.e[F_ROW][b];
                        sm.e[F_DEPTH][ne] = sm.e[F_DEPTH][b] + 1;
                    }
                }
                for (int j = 0; j < nb; ++j) {  // beam slots: parent among the elements, else staged
                    int ps = PS_FROZEN;
                    if (sm.e[F_PAR][j] == 0) ps = PS_ROOT;
                    else
                        for (int i = 0; i < ne; ++i) if (sm.e[F_ID][i] == sm.e[F_PAR][j]) ps = i;
                    sm.e[F_PSLOT][j] = ps;
                }
                sm.sh[2] = next_id; sm.sh[6] = ne; sm.sh[7] = np; sm.sh[11] = 0; sm.sh[15] = 0;
            }
            b2_sync_lds<nthr>();
            *nproc = sm.sh[7];
            return sm.sh[6];
        };

        // prune (Beam.h:93-108) + next beam table.  A slot whose node also sits in an earlier slot is
        // the same node pushed twice (std::unique).
        auto prune_and_advance = [&](int ne, bool regular) {
            // (the duplicate flags only depend on the node ids: unchanged while the table is — see "steady" in scan)
            const bool dup_valid = regular && sm.sh[14] != 0 && sm.sh[11] != 0;
            if (tid < ne && !dup_valid) {
                int d = 0;
                const int x = sm.e[F_ID][tid];
                if (regular) {
                    if (tid >= nb) {
#pragma unroll 4
                        for (int j = 0; j < nb; ++j) d |= (sm.e[F_ID][j] == x);
                    }
                } else {
                    for (int j = 0; j < tid; ++j) d |= (sm.e[F_ID][j] == x);
                }
                sm.dup[tid] = d;
            }
            if (tid == 0) sm.sh[5] = 0;
            b2_sync_lds<nthr>();
            // Most steps keep the beam as it is: that holds iff the beam nodes are still in order and the last of
            // them still beats every child — two comparisons per thread instead of a ranking against everybody.
            bool same_beam = false;
            if (regular && nb == W) {
                bool viol = false;
                {   // (every operand read unconditionally: one LDS round trip)
                    const int tx = (tid < SM::NCM) ? tid : SM::NCM - 1, tn = (tid + 1 < SM::NCM) ? tid + 1 : SM::NCM - 1;
                    const int dupf = sm.dup[tx];
                    const double sc = sm.score[tx], scn = sm.score[tn], scl = sm.score[nb - 1];
                    // (strictly: an exact tie is resolved by the full path below, as the reference's partial_sort does)
                    if (tid < ne && !dupf) {
                        if (tid >= nb) viol = !(scl > sc);
                        else if (tid + 1 < nb) viol = !(sc > scn);
                    }
                }
                same_beam = !b2_any<nthr>(viol);
#ifdef PO_ABL_ALWAYSSAME   // timing ablation only (results are wrong): every prune keeps the beam
                same_beam = true;
#endif
            }
            if (same_beam) {
                if (tid < nb) sm.stay[tid] = 1;
                if (tid == 0) { sm.sh[10] = 1; sm.sh[14] = 1; }
                b2_sync_lds<nthr>();
                return;
            }
            if (tid == 0) sm.sh[14] = 0;
            bool teq = false;
            if constexpr (nthr == 64) {
                // One wave per pair (the W <= 6 class).  Only the beam nodes and the children that reach the smallest
                // beam score can be among the W best (every other child has W candidates above it), and nothing outside
                // that set outranks a member of it: the ranks are taken within it — a handful of broadcast LDS reads
                // instead of W * (A + 1) candidates.  (Regular table with a full beam; otherwise everybody is in.)
                const bool cnd = tid < ne && !sm.dup[min(tid, SM::NCM - 1)];
                const int tx = min(tid, SM::NCM - 1);
                const double sc = sm.score[tx];
                const int id = sm.e[F_ID][tx];
                bool inS = cnd;
                if (regular && nb == W) {
                    double thr = sm.score[0];
                    for (int i = 1; i < nb; ++i) thr = fmin(thr, sm.score[i]);
                    inS = cnd && (tid < nb || sc >= thr);
                }
                const unsigned long long smk = __ballot(inS);
                int rank = 0, neq = 0;
                for (unsigned long long mm = smk; mm != 0ull; mm &= mm - 1ull) {   // (uniform)
                    const int o = __builtin_ctzll(mm);
                    const double so = sm.score[o];
                    const int io = sm.e[F_ID][o];
                    rank += ((so > sc) | (!(sc > so) & (io < id))) ? 1 : 0;
                    neq += (so == sc) ? 1 : 0;
                }
                if (inS && rank < W) sm.sel[rank] = tid;
                teq = inS && (neq > 1) && (rank < W);   // an exact score tie that reaches into the beam
                const int ncnd = __popcll(__ballot(cnd));
                if (tid == 0) sm.sh[5] = ncnd;
            } else
            if (tid < ne && !sm.dup[tid]) {
                const double sc = sm.score[tid];
                const int id = sm.e[F_ID][tid];
                int rank = 0, neq = 0;
                // (branch-free, every load unconditional: the LDS reads of eight candidates go out together)
#pragma unroll 8
                for (int o = 0; o < ne; ++o) {
                    const double so = sm.score[o];
                    const int io = sm.e[F_ID][o];
                    const int live = sm.dup[o] ? 0 : 1;
                    const int better = ((so > sc) | (!(sc > so) & (io < id))) ? 1 : 0;
                    rank += live & better;
                    neq += live & ((so == sc) ? 1 : 0);
                }
                if (rank < W) sm.sel[rank] = tid;
                teq = (neq > 1) && (rank < W);   // an exact score tie that reaches into the beam
                atomicAdd(&sm.sh[5], 1);
            }
            if (b2_any<nthr>(teq)) {
                // Exact ties (candidates that are all -inf, quantised inputs): the beam is what libstdc++'s partial_sort /
                // sort leave on the candidates in pointer = creation order (po_device.h).  Candidate slots in node-id
                // order (badf is free between scans), then one thread replays the algorithm.
                int* ord = sm.badf;
                if (tid < ne && !sm.dup[tid]) {
                    const int id = sm.e[F_ID][tid];
                    int pos = 0;
                    for (int o = 0; o < ne; ++o) pos += (!sm.dup[o] && sm.e[F_ID][o] < id) ? 1 : 0;
                    ord[pos] = tid;
                }
                b2_sync_lds<nthr>();
                if (tid == 0) {
                    const int m = sm.sh[5];
                    const double* scp = sm.score;
                    po_stl_prune<WMAX>(ord, m, W, [&](int slot) { return scp[slot]; });
                    for (int j = 0; j < min(W, m); ++j) sm.sel[j] = ord[j];
                }
            }
            b2_sync_lds<nthr>();
            const int nbn = min(W, sm.sh[5]);
            if (tid < nbn) sm.stay[tid] = (regular && sm.sel[tid] < nb) ? 1 : 0;
            // the last value of each new beam node's parent, as this step's table holds it (see B2Smem::fzv): of the
            // parent's slot if it is an element now, else what was captured earlier
            double cfv[2][K];
            int cft[2] = {INT_MAX, INT_MAX};
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int q = 0; q < K; ++q) cfv[rr][q] = PO_NEG_INF;
            if (regular && tid < nbn) {
                const int e = sm.sel[tid];
                const int op = sm.e[F_PSLOT][e];
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    if (op >= 0) {
                        // (after a window end moved back the store holds later values than the last one computed: no capture)
                        const int te = sm.sh[16 + rr];
                        cft[rr] = (te == INT_MIN || te != sm.sh[18 + rr]) ? INT_MAX : te - 1;
#pragma unroll
                        for (int q = 0; q < K; ++q) cfv[rr][q] = sm.xch[1][rr][op][q];
                    } else if (op == PS_FROZEN && e < nb) {
                        cft[rr] = sm.fzt[rr][e];
#pragma unroll
                        for (int q = 0; q < K; ++q) cfv[rr][q] = sm.fzv[rr][e][q];
                    }
                }
            }
            {
                // The same beam nodes in another order: every element of the next step was an element of this one, in
                // the slot the permutation says (beam slot i <- sel[i]; its children follow it), so the carried maxima
                // move along and the next step can still be incremental.  Anything else ends the run.
                const bool perm = regular && nbn == nb && !b2_any<nthr>(tid < nbn && sm.sel[tid] >= nb);
                double c0 = 0.0, c1 = 0.0, sf0[K], sf1[K];
                int t0_ = 0, t1_ = 0, d0_ = 0, d1_ = 0;
                if (perm && tid < ne) {
                    int src = tid;
                    if (tid < nb) src = sm.sel[tid];
                    else { const int j = ((tid - nb) * divA) >> 16; src = nb + A * sm.sel[j] + ((tid - nb) - j * A); }
                    c0 = sm.cmx[0][src]; c1 = sm.cmx[1][src]; t0_ = sm.cmt[0][src]; t1_ = sm.cmt[1][src];
                    d0_ = sm.ctd[0][src]; d1_ = sm.ctd[1][src];
#pragma unroll
                    for (int q = 0; q < K; ++q) { sf0[q] = sm.xch[1][0][src][q]; sf1[q] = sm.xch[1][1][src][q]; }
                }
                b2_sync_lds<nthr>();
                if (perm && tid < ne) {
                    sm.cmx[0][tid] = c0; sm.cmx[1][tid] = c1; sm.cmt[0][tid] = t0_; sm.cmt[1][tid] = t1_;
                    sm.ctd[0][tid] = d0_; sm.ctd[1][tid] = d1_;
#pragma unroll
                    for (int q = 0; q < K; ++q) { sm.xch[1][0][tid][q] = sf0[q]; sm.xch[1][1][tid][q] = sf1[q]; }
                }
                if (tid == 0) {
                    if (perm) sm.sh[10] = 1;
                    else { sm.sh[10] = 0; sm.sh[12] = 0; sm.sh[13] = 0; }
                }
            }
            if (tid < nbn) {
                const int e = sm.sel[tid];
                if (e < nb) {
#pragma unroll
                    for (int f = 0; f < F_COUNT; ++f) sm.nx[f][tid] = sm.e[f][e];
                } else {  // a child enters the beam; the arena knows whether it was ever expanded
                    const int p = sm.e[F_PSLOT][e];
                    const int id = sm.e[F_ID][e];
                    sm.nx[F_ID][tid] = id; sm.nx[F_ROW][tid] = sm.e[F_ROW][e]; sm.nx[F_PSLOT][tid] = PS_FROZEN;
                    sm.nx[F_SYM][tid] = sym_pack(sym_last(sm.e[F_SYM][e]), sym_last(sm.e[F_SYM][p]), false);
                    sm.nx[F_PAR][tid] = sm.e[F_ID][p]; sm.nx[F_GPAR][tid] = sm.e[F_PAR][p];
                    sm.nx[F_PROW][tid] = sm.e[F_ROW][p]; sm.nx[F_DEPTH][tid] = sm.e[F_DEPTH][p] + 1;
                    int fc = sm.e[F_FC][e], cr = sm.e[F_CROW][e];
                    if (fc == -2) { fc = afc[id]; cr = acrow[id]; }
                    sm.nx[F_FC][tid] = fc; sm.nx[F_CROW][tid] = cr;
                }
            }
            b2_sync_lds<nthr>();
            if (tid < nbn) {
#pragma unroll
                for (int f = 0; f < F_COUNT; ++f) sm.e[f][tid] = sm.nx[f][tid];
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    sm.fzt[rr][tid] = cft[rr];
#pragma unroll
                    for (int q = 0; q < K; ++q) sm.fzv[rr][tid][q] = cfv[rr][q];
                }
            }
            nb = nbn;
            b2_sync_lds<nthr>();
        };

        if (!is_row) {
            // ============================================================ row_col: the diagonal walk
            // u and v advance by at most one per round: the bounds of row u + 1 and column v + 1 are requested a round
            // ahead (a round otherwise starts with a dependent global round trip, and most catch-up rounds are no-ops)
            int u = 0, v = 0;
            const int2* env2 = (const int2*)env;
            const int2* envt2 = (const int2*)envt;
            b2_scalar_cache_inv();   // envt was written by this workgroup (vector stores, complete since the barrier after the pre-pass)
            int2 er_c = b2_sload2(env2), ec_c = b2_sload2(envt2), er_n = er_c, ec_n = ec_c;
            int we0 = INT_MIN, we1 = INT_MIN;   // = sh[8], sh[9] (window ends of the last main step), without the LDS round trip
            while (u <= U - 1 && v <= V - 1) {
                er_n = b2_sload2(env2 + min(u + 1, U - 1)); ec_n = b2_sload2(envt2 + min(v + 1, V - 1));
                const int ers = er_c.x, ere = er_c.y;
                const int ecs = ec_c.x, ece = ec_c.y;
                const bool row_ok = (v >= ers && v < ere);
                const bool col_ok = (u >= ecs && u < ece);
                const bool cu_v = (!row_ok && v < ers);                 // catch-up on read 1 (:314-322)
                const bool cu_u = !cu_v && (!col_ok && u < ecs);        // catch-up on read 0 (:328-336)
                if (cu_v || cu_u) {
                    const int nbe = min(W, nb);  // the reference indexes b < beam_width
                    // (node, t) of a beam node is already stored with these very bits when the last main step's
                    // window on this read covered t (see the store-skipping note in scan): nothing to do then
                    const bool need = cu_v ? (v >= we1) : (u >= we0);
                    if (need) {
                        if (tid < nbe) {
                            sm.cu_ps[tid] = beam_parent(tid, nbe, false);
                            if (cu_v) atomicMax(&sm.g_hi1[sm.e[F_ROW][tid] / PO_A], v + 1);
                            else atomicMax(&sm.g_hi0[sm.e[F_ROW][tid] / PO_A], u + 1);
                        }
                        b2_sync_mem<nthr>();  // store writes of earlier steps -> visible to this scan's reads
                        TK(6);
                        if (cu_v) scan(false, false, nbe, 0, 0, 0, 0, v, 1);
                        else scan(false, false, nbe, 0, 0, u, 1, 0, 0);
                        b2_sync_lds<nthr>();
                    }
                    TKC(11);
                    if (!need && a.upd_count != nullptr && tid == 0) sm.nupd += (unsigned)nbe;   // (a catch-up the reference computes and this kernel need not)
                    if (cu_v) { v++; ec_c = ec_n; } else { u++; er_c = er_n; }
                    continue;
                }
                if (!row_ok || !col_ok) { st = PO_E_ENVELOPE; break; }  // uninitialised bounds upstream (:309)
                // ---- MAIN step at (u, v): windows [u, ece) x [v, ere)  (:342-375)
                TK_TYPE((sm.sh[14] != 0 && sm.sh[15] != 0) ? 12 : ((sm.sh[10] != 0) ? 24 : 0));
                const int ne = build_regular(u - 1, v - 1, ece, ere);
                b2_sync_mem<nthr>();  // arena + store writes -> visible to the reads below
                if (!steady_tbl && sm.sh[4] != PO_OK) { st = sm.sh[4]; break; }
                TK(2);
                scan(true, true, ne, 0, 0, u, ece - u, v, ere - v);
                b2_sync_lds<nthr>();
                if (tid == 0) { sm.sh[8] = ece; sm.sh[9] = ere; }
                we0 = ece; we1 = ere;
                if (tid < ne) sm.score[tid] = sm.xch[0][0][tid][0] + sm.xch[0][1][tid][0];  // node_greater_max_sym (window maxima: see B2Smem)
                prune_and_advance(ne, true);
                TK(1);
                TK_STEP();
                TK_TYPE(0);
                u++;
                v++;
                er_c = er_n; ec_c = ec_n;
            }
        } else {
            // ============================================================ row: every row of read 0
            int2 rb_n = env ? ((const int2*)env)[0] : make_int2(0, V);   // (the next row's band is requested a row ahead)
            for (int u = env ? 0 : 1; u < U; ++u) {
                const int2 rb = rb_n;
                if (env) rb_n = ((const int2*)env)[min(u + 1, U - 1)];
                const int rs = rb.x, re = rb.y;
                const int wlen = max(0, re - rs);
                int ne, nproc;
                const bool regular = (nb == W);
                if (regular) { ne = build_regular(u - 1, rs - 1, u + 1, re); nproc = nb; }
                else ne = build_growing(u - 1, rs - 1, u + 1, re, &nproc);
                b2_sync_mem<nthr>();
                if (sm.sh[4] != PO_OK) { st = sm.sh[4]; break; }
                TK(2);
                // every element is updated on read 0 at time u except beam nodes beyond the first W
                // (possible only in the first row, when W < |alphabet|): `for b < beam_width` (:132)
                const int skip_lo = min(nproc, nb), skip_hi = nb;
                // the read-1 band of a row mostly repeats the previous row's: the same redundant-store rule as in
                // row_col applies (regular shape; in the growing first rows every element stores everything)
                scan(true, regular, ne, skip_lo, skip_hi, u, 1, rs, wlen);
                b2_sync_lds<nthr>();
                if (tid == 0) { sm.sh[8] = u + 1; sm.sh[9] = (wlen > 0) ? re : sm.sh[9]; }
                if (tid < ne) {  // node_greater_max: last_prob[0] + max_prob[1]
                    const int id = sm.e[F_ID][tid];
                    double m0 = sm.xch[0][0][tid][0], m1 = sm.xch[0][1][tid][0];   // window maxima of the scan (see B2Smem)
                    if (tid >= skip_lo && tid < skip_hi) {  // last_prob[0] is still the seed value at t = 0
                        double tmp[K];
                        st_read(sm.e[F_ROW][tid], 0, 0, id, tmp);
                        m0 = tmp[0];
                    }
                    if (wlen > 0) amax[id] = m1;  // reset_max happened at v == rs
                    else m1 = amax[id];           // empty band: max_prob[1] keeps its last value
                    sm.score[tid] = m0 + m1;
                }
                b2_sync_mem<nthr>();
                prune_and_advance(ne, regular);
                TK(1);
            }
        }

        // ---------------------------------------------------------------- label of the top node
        __syncthreads();
        if (tid == 0) {
            int nout = 0;
            if (st == PO_OK) {
                int node = sm.e[F_ID][0];
                nout = sm.e[F_DEPTH][0];
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
            if (st == PO_E_NOMEM) a.queue[8] = 1;   // (the retry pass looks at this word first)
        }
        if (a.upd_count && tid == 0) { atomicAdd(a.upd_count, sm.nupd); atomicAdd(a.upd_count + 1, sm.nupd_x); }
        TK(10);  // label walk
    }
    if (tid == 0) {   // the next launch on this workspace continues from here
        unsigned long long* stp = a.wgstate + 2 * (size_t)blockIdx.x;
        stp[0] = a.magic ^ (unsigned long long)blockIdx.x;
        stp[1] = (unsigned long long)epoch;
    }
#ifdef PO_B2_TIMING
    if (tid == 0 && a.dbg && blockIdx.x == 0)
        for (int i = 0; i < 37; ++i) a.dbg[i] = tk[i];
#endif
}

// (the pre-pass and the diagonal walk of the register-state kernel: po_beam2d_reg.hip; they are launched from this file)
#include "po_beam2d_pre.h"

// =================================================================================================
// method "grid" (beam_search_2d_grid, BeamSearch2.h:33-184; hidden upstream option): ONE BEAM PER CELL.
// Cell (u, v) of the band takes the beam of cell (u-1, v-1) — or the seed beam (the root's children) when
// that cell was never visited — updates every node of it and every child at (read 0, u) and (read 1, v),
// and keeps the W best by alpha0[u] + alpha1[v].  The cells are visited in row-major order and all beams
// share one tree whose per-node time maps are overwritten by later visits, so the order is part of the
// result: a pair is walked cell by cell by one workgroup, thread = (read, candidate slot); parallelism is
// the candidates of a cell and the pairs of the batch.  Within a cell every update reads times u-1 / v-1
// and writes u / v, so the candidates are independent; values live in the same tagged ring store as the
// other methods' (R >= widest row band + 2), a row group is recycled when nothing written to it can be
// read again (read 0: times < u-1; read 1: times < row start - 1, row starts must not move backwards),
// and the beams of the previous row are kept in HBM (6 ints per node).
namespace {
enum { G_ID, G_ROW, G_PAR, G_PROW, G_SYM, G_DEPTH, G_COUNT };
template <int MODEL, int WMAX>
struct GridSmem {
    static constexpr int K = (MODEL == PO_MODEL_CTC) ? 1 : 3;
    static constexpr int NCM = WMAX * (PO_A + 1);
    static constexpr int NCP = (NCM <= 32) ? 32 : 128;   // threads per read
    static constexpr int NGL = 3072;   // row groups tracked per pair (GRID_NGL on the host side)
    int e[G_COUNT][NCM];
    int fc[WMAX], crow[WMAX], isnew[WMAX];
    int dup[NCM];
    int ord[NCM];           // prune with exact score ties: candidate slots in node-id order (po_stl_prune)
    int sel[WMAX];
    int g_owner[NGL], g_hi0[NGL], g_hi1[NGL];
    int sh[8];
    double sc[2][NCP];
    unsigned long long nupd;
    PoLaeTables lae;
};
}  // namespace
#define GRID_THREADS(WM) ((WM) * (PO_A + 1) <= 32 ? 64 : 256)

template <int MODEL, int WMAX>
__global__ __launch_bounds__(GRID_THREADS(WMAX)) void beam2d_grid_kernel(B2Args a) {
    using SM = GridSmem<MODEL, WMAX>;
    static_assert(SM::NGL == 3072, "GRID_NGL");
    constexpr int K = SM::K, NCP = SM::NCP, nthr = 2 * NCP;
    using Ent = Entry<K>;
    __shared__ SM sm;
    const int tid = threadIdx.x;
    const int r = tid / NCP, s = tid - r * NCP;
    const int A = a.A, W = a.W, C = a.C;
    const int divA = (65536 + A - 1) / A;
    Ent* pool = (Ent*)(a.pool + (size_t)blockIdx.x * a.pool_bytes);
    const long long pool_entries = (long long)(a.pool_bytes / sizeof(Ent));
    int* apl = a.arena + (size_t)blockIdx.x * 3 * a.arena_cap;
    int* afc = apl + a.arena_cap;
    int* acrow = afc + a.arena_cap;
    double* cum0 = a.cum + (size_t)blockIdx.x * 2 * a.tcap;
    double* cum1 = cum0 + a.tcap;
    const int CI = 1 + W * G_COUNT;                                     // ints per stored cell: n, then the entries
    int* cellb = a.cellb + (size_t)blockIdx.x * 2 * a.vcap * CI;          // two rows of cells, indexed by column
    unsigned epoch = 0;
    po_lae_tables_load(&sm.lae, tid, nthr);
    const PoLaeFast lae{&sm.lae};

    for (;;) {
        __syncthreads();
        if (tid == 0) sm.sh[0] = atomicAdd(a.queue, 1);
        __syncthreads();
        const int pi = sm.sh[0];
        if (pi >= a.n) break;
        epoch++;
        if (a.use_pre_status && a.status[pi] != PO_OK) {
            if (tid == 0) a.seq_len[pi] = 0;
            continue;
        }
        const int64_t o1 = a.y1_off[pi], o2 = a.y2_off[pi];
        const int U = (int)(a.y1_off[pi + 1] - o1), V = (int)(a.y2_off[pi + 1] - o2);
        const double* yr_ = r ? a.y2 + o2 * C : a.y1 + o1 * C;
        const int32_t* env = a.env ? a.env + 2 * o1 : nullptr;
        int st = PO_OK;
        if (U < 1 || V < 1 || U > a.tcap || V > a.vcap || V > a.tcap || U >= (1 << 24) || V >= (1 << 24)) st = PO_E_ARG;
        int R = 32, NG = 0;
        if (st == PO_OK) {
            int bad = 0, unsup = 0, wmax = env ? 0 : V;
            if (env)
                for (int u = tid; u < U; u += nthr) {
                    const int lo = env[2 * u], hi = env[2 * u + 1];
                    if (lo < 0 || hi > V) bad = 1;                       // BeamSearch2.h reads y2 out of bounds there
                    if (u > 0 && lo < env[2 * u - 2]) unsup = 1;         // a later row would read times the store has dropped
                    wmax = max(wmax, hi - lo);
                }
            if (__syncthreads_or(bad)) st = PO_E_ENVELOPE;
            if (__syncthreads_or(unsup) && st == PO_OK) st = PO_E_UNSUPPORTED;
            if (st == PO_OK) {
                if (tid == 0) sm.sh[1] = 0;
                __syncthreads();
                atomicMax(&sm.sh[1], wmax);
                __syncthreads();
                wmax = sm.sh[1];
                while (R < wmax + 2) R <<= 1;
                NG = (int)min((long long)SM::NGL, pool_entries / ((long long)PO_A * 2 * R));
                if (NG < 2 * max(W, PO_A) + 4) st = PO_E_NOMEM;
            }
        }
        if (st != PO_OK) {
            if (tid == 0) { a.status[pi] = st; a.seq_len[pi] = 0; }
            continue;
        }
        const int Rm = R - 1;
        if (MODEL == PO_MODEL_CTC && s == 0) {   // blank prefix sums = the CTC root's alpha, serial in t
            double* cw = r ? cum1 : cum0;
            const int Tn = r ? V : U;
            double acc = 0.0;
            for (int t = 0; t < Tn; ++t) { acc += yr_[(int64_t)t * C + A]; cw[t] = acc; }
        }
        for (int g = tid; g < SM::NGL; g += nthr) { sm.g_owner[g] = -1; sm.g_hi0[g] = 0; sm.g_hi1[g] = 0; }
        __syncthreads();
        auto st_read = [&](int row, int rr, int t, int node, double* out) {
            bool hit = false;
            if (t >= 0 && row >= 0) {
                const Ent e = pool[((size_t)row * 2 + rr) * R + (t & Rm)];
                hit = (e.tag == make_tag(epoch, node, t));
                if (hit) {
#pragma unroll
                    for (int k = 0; k < K; ++k) out[k] = e.v[k];
                }
            }
            if (!hit) {
#pragma unroll
                for (int k = 0; k < K; ++k) out[k] = PO_NEG_INF;
            }
        };
        auto st_write = [&](int row, int rr, int t, int node, const double* v) {
            Ent e;
            e.tag = make_tag(epoch, node, t);
#pragma unroll
            for (int k = 0; k < K; ++k) e.v[k] = v[k];
            pool[((size_t)row * 2 + rr) * R + (t & Rm)] = e;
        };
        auto root_at = [&](int rr, int t, double* out) {
            if (MODEL == PO_MODEL_CTC) {
                out[0] = (t < 0) ? 0.0 : (rr ? cum1 : cum0)[t];
            } else {
                double tmp[3];
                root_values<MODEL>(t, 0.0, tmp);
#pragma unroll
                for (int k = 0; k < K; ++k) out[k] = tmp[k];
            }
        };
        // tree: root = node 0, its children = nodes 1..A in row group 0, which is never recycled (the seed
        // beam can come back at any cell); beam2d_seed (update at t = 0 on both reads)
        if (tid == 0) {
            apl[0] = po_pack_node(-1, A); afc[0] = 1; acrow[0] = 0;
            sm.g_owner[0] = 0; sm.g_hi0[0] = 1; sm.g_hi1[0] = 1;
            sm.sh[2] = 1 + A;   // next node id
            sm.sh[3] = 1;       // group allocation cursor
            sm.sh[4] = PO_OK;
            sm.nupd = 0;
        }
        if (tid < A) { apl[1 + tid] = po_pack_node(0, tid); afc[1 + tid] = -1; acrow[1 + tid] = -1; }
        if (s < A) {
            double sp[3] = {PO_NEG_INF, PO_NEG_INF, PO_NEG_INF}, pp[3], out[3];
            root_at(r, -1, pp);
            const double ya = yr_[s], yb = (MODEL == PO_MODEL_FLIPFLOP) ? yr_[s + A] : yr_[A];
            po_update<MODEL>(sp, pp, ya, yb, false, true, out, lae);
            st_write(s, r, 0, 1 + s, out);
        }
        __syncthreads();

        int prs = 0, pre = 0;   // the previous row's band
        for (int u = 0; u < U && st == PO_OK; ++u) {
            const int rs = env ? env[2 * u] : 0, re = env ? env[2 * u + 1] : V;
            int* cur = cellb + (size_t)(u & 1) * a.vcap * CI;
            const int* prv = cellb + (size_t)((u + 1) & 1) * a.vcap * CI;
            for (int v = rs; v < re; ++v) {
                // ---- 1. the predecessor's beam (BeamSearch2.h:137-146)
                const bool hp = u > 0 && v > 0 && (v - 1) >= prs && (v - 1) < pre;
                int nb = A;
                if (hp) {
                    const int* pc = prv + (size_t)(v - 1) * CI;
                    nb = pc[0];
                    if (tid < nb * G_COUNT) sm.e[tid % G_COUNT][tid / G_COUNT] = pc[1 + tid];
                } else if (tid < A) {
                    sm.e[G_ID][tid] = 1 + tid; sm.e[G_ROW][tid] = tid; sm.e[G_PAR][tid] = 0; sm.e[G_PROW][tid] = -1;
                    sm.e[G_SYM][tid] = sym_pack(tid, A, true); sm.e[G_DEPTH][tid] = 1;
                }
                __syncthreads();
                // ---- 2. expansion: children ids in beam order (ids break score ties), row groups
                if (tid < nb) {
                    const int id = sm.e[G_ID][tid];
                    sm.fc[tid] = afc[id]; sm.crow[tid] = acrow[id];
                }
                __syncthreads();
                if (tid == 0) {
                    int next = sm.sh[2];
                    for (int j = 0; j < nb; ++j) {
                        const int id = sm.e[G_ID][j];
                        int fcj = sm.fc[j], cr = sm.crow[j];
                        sm.isnew[j] = fcj < 0;
                        if (fcj < 0) {
                            if ((long long)next + A > a.arena_cap || next + A >= (1 << 24)) { sm.sh[4] = PO_E_NOMEM; break; }
                            fcj = next; next += A;
                            afc[id] = fcj;
                        }
                        if (cr < 0 || cr >= NG || sm.g_owner[cr] != id) {   // children never stored, or their rows recycled
                            int cursor = sm.sh[3], g = -1;
                            for (int tries = 0; tries < NG; ++tries) {
                                const int c = cursor;
                                cursor = (cursor + 1 == NG) ? 0 : cursor + 1;
                                if (c != 0 && (sm.g_owner[c] < 0 || (sm.g_hi0[c] <= u - 1 && sm.g_hi1[c] <= rs - 1))) { g = c; break; }
                            }
                            sm.sh[3] = cursor;
                            if (g < 0) { sm.sh[4] = PO_E_NOMEM; break; }
                            sm.g_owner[g] = id;
                            cr = g;
                            acrow[id] = g;
                        }
                        sm.g_hi0[cr] = u + 1; sm.g_hi1[cr] = v + 1;   // written below
                        sm.fc[j] = fcj; sm.crow[j] = cr;
                    }
                    sm.sh[2] = next;
                }
                __syncthreads();
                if (sm.sh[4] != PO_OK) { st = sm.sh[4]; break; }
                // ---- 3. candidate table: slots [0, nb) the beam, then the children
                const int ne = nb * (A + 1);
                if (r == 0 && s >= nb && s < ne) {
                    const int j = ((s - nb) * divA) >> 16, c = (s - nb) - j * A;
                    const int id = sm.fc[j] + c;
                    sm.e[G_ID][s] = id; sm.e[G_ROW][s] = sm.crow[j] * PO_A + c;
                    sm.e[G_PAR][s] = sm.e[G_ID][j]; sm.e[G_PROW][s] = sm.e[G_ROW][j];
                    sm.e[G_SYM][s] = sym_pack(c, sym_last(sm.e[G_SYM][j]), false);
                    sm.e[G_DEPTH][s] = sm.e[G_DEPTH][j] + 1;
                    if (sm.isnew[j]) { apl[id] = po_pack_node(sm.e[G_ID][j], c); afc[id] = -1; acrow[id] = -1; }
                    int d = 0;
                    for (int i = 0; i < nb; ++i) d |= (sm.e[G_ID][i] == id);   // Beam::prune dedupes by identity
                    sm.dup[s] = d;
                } else if (r == 0 && s < nb) {
                    sm.dup[s] = 0;
                    atomicMax(&sm.g_hi0[sm.e[G_ROW][s] / PO_A], u + 1);
                    atomicMax(&sm.g_hi1[sm.e[G_ROW][s] / PO_A], v + 1);
                }
                __syncthreads();
                // ---- 4. update_prob(node, 0, u) and (node, 1, v) of every candidate
                if (s < ne) {
                    const int id = sm.e[G_ID][s], row = sm.e[G_ROW][s], par = sm.e[G_PAR][s], sy = sm.e[G_SYM][s];
                    const int t = r ? v : u;
                    const int sym = sym_last(sy);
                    double self[K], pp[K], out[K];
                    st_read(row, r, t - 1, id, self);
                    if (par == 0) root_at(r, t - 1, pp);
                    else st_read(sm.e[G_PROW][s], r, t - 1, par, pp);
                    const double* yrow = yr_ + (int64_t)t * C;
                    const double ya = yrow[sym], yb = (MODEL == PO_MODEL_FLIPFLOP) ? yrow[sym + A] : yrow[A];
                    po_update<MODEL>(self, pp, ya, yb, sym_plast(sy) == sym, ((sy >> 9) & 1) && t == 0, out, lae);
                    st_write(row, r, t, id, out);
                    sm.sc[r][s] = out[0];
                }
                if (tid == 0) sm.nupd += 2u * (unsigned)ne;
                __syncthreads();
                // ---- 5. prune: the W best of the distinct candidates by alpha0[u] + alpha1[v]; an exact tie that reaches
                // into the beam (narrow bands: candidates that are all -inf) is resolved as libstdc++'s partial_sort / sort
                // leave the creation-ordered candidates (po_device.h), like every other prune
                const bool live = (r == 0) && s < ne && !sm.dup[s];
                bool teq = false;
                if (live) {
                    const double my = sm.sc[0][s] + sm.sc[1][s];
                    const int myid = sm.e[G_ID][s];
                    int rank = 0, neq = 0;
#pragma unroll 8
                    for (int o = 0; o < ne; ++o) {   // (branch-free: the candidates' LDS reads go out in batches)
                        const double so = sm.sc[0][o] + sm.sc[1][o];
                        const int io = sm.e[G_ID][o];
                        const int lv = sm.dup[o] ? 0 : 1;
                        rank += lv & (((so > my) | (!(my > so) & (io < myid))) ? 1 : 0);
                        neq += lv & ((so == my) ? 1 : 0);
                    }
                    if (rank < W) sm.sel[rank] = s;
                    teq = (neq > 1) && (rank < W);
                }
                const int ncand = __syncthreads_count(live);
                const int nn = min(W, ncand);
                if (__syncthreads_or(teq)) {
                    if (live) {
                        const int myid = sm.e[G_ID][s];
                        int pos = 0;
                        for (int o = 0; o < ne; ++o) pos += (!sm.dup[o] && sm.e[G_ID][o] < myid) ? 1 : 0;
                        sm.ord[pos] = s;
                    }
                    __syncthreads();
                    if (tid == 0) {
                        const double* s0 = sm.sc[0];
                        const double* s1 = sm.sc[1];
                        po_stl_prune<WMAX>(sm.ord, ncand, W, [&](int slot) { return s0[slot] + s1[slot]; });
                        for (int j = 0; j < nn; ++j) sm.sel[j] = sm.ord[j];
                    }
                    __syncthreads();
                }
                // ---- 6. the cell's beam goes to HBM for cell (u+1, v+1)
                int* cc = cur + (size_t)v * CI;
                if (tid < nn * G_COUNT) cc[1 + tid] = sm.e[tid % G_COUNT][sm.sel[tid / G_COUNT]];
                if (tid == 0) cc[0] = nn;
                __syncthreads();
            }
            prs = rs; pre = re;
        }
        // ---- the top of the last cell's beam, or of the seed beam if (U-1, V-1) was never visited (:176-183)
        __syncthreads();
        if (tid == 0) {
            int nout = 0;
            if (st == PO_OK) {
                int node = 1, depth = 1;
                const int rsl = env ? env[2 * (U - 1)] : 0, rel = env ? env[2 * (U - 1) + 1] : V;
                if (V - 1 >= rsl && V - 1 < rel) {
                    const int* lc = cellb + (size_t)((U - 1) & 1) * a.vcap * CI + (size_t)(V - 1) * CI;
                    node = lc[1 + G_ID]; depth = lc[1 + G_DEPTH];
                }
                nout = depth;
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
            if (a.upd_count) { atomicAdd(a.upd_count, sm.nupd); atomicAdd(a.upd_count + 1, sm.nupd); }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side: geometry, workspace layout, launch
namespace {
struct B2Geom {
    int threads, blocks, wclass;
    size_t pool_bytes, arena_cap, tcap, vcap;
    size_t off_queue, off_state, off_pool, off_arena, off_cum, off_envt, off_order, total;
    unsigned long long magic;
};
inline size_t al256(size_t b) { return (b + 255) & ~size_t(255); }

// device memory a pair-beam workspace may plan with (a fixed share of the board's memory: the size reported by
// po_*_workspace_bytes and the size a launch expects must agree whatever else is allocated)
size_t b2_mem_budget() { return po_dev_info().mem / 4; }

int b2_num_cus() { return po_dev_info().cus; }

// resident workgroups per CU of one kernel instance (registers and LDS decide), asked once from the runtime
template <int MODEL, int WMAX>
int b2_occ() {
    int nblk = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)beam2d_kernel<MODEL, WMAX>, B2_THREADS(WMAX), 0) !=
            hipSuccess || nblk <= 0)
        nblk = 12 * 64 / B2_THREADS(WMAX);
    return nblk;
}
int b2_blocks_per_cu(int model, int wclass) {
    static PoPerDeviceCache<9> cache;
    const int mi = model == PO_MODEL_CTC ? 0 : (model == PO_MODEL_MERGE ? 1 : 2), wi = wclass == 6 ? 0 : (wclass == 12 ? 1 : 2);
    return cache.get(mi * 3 + wi, [=] {
        if (mi == 0) return wi == 0 ? b2_occ<PO_MODEL_CTC, 6>() : (wi == 1 ? b2_occ<PO_MODEL_CTC, 12>() : b2_occ<PO_MODEL_CTC, 25>());
        if (mi == 1) return wi == 0 ? b2_occ<PO_MODEL_MERGE, 6>() : (wi == 1 ? b2_occ<PO_MODEL_MERGE, 12>() : b2_occ<PO_MODEL_MERGE, 25>());
        return wi == 0 ? b2_occ<PO_MODEL_FLIPFLOP, 6>() : (wi == 1 ? b2_occ<PO_MODEL_FLIPFLOP, 12>() : b2_occ<PO_MODEL_FLIPFLOP, 25>());
    });
}

B2Geom b2_geometry(int n, int64_t mr1, int64_t mr2, int W, int model, int method, int max_blocks = 0) {
    B2Geom g;
    const int K = (model == PO_MODEL_CTC) ? 1 : 3;
    g.wclass = W <= 6 ? 6 : (W <= 12 ? 12 : 25);
    // the small passes (pairs handed back by the x2 kernel, retries after PO_E_NOMEM) run the W <= 6 class on the
    // W <= 12 kernel (256 instead of 112 row groups) and the W <= 12 class on the W <= 25 kernel (768)
    if (max_blocks > 0 && g.wclass < 25) g.wclass = (g.wclass == 6) ? 12 : 25;
    g.threads = g.wclass == 6 ? 64 : (g.wclass == 12 ? 128 : 256);
    const int per_cu = b2_blocks_per_cu(model, g.wclass);
    g.blocks = b2_num_cus() * per_cu;
    if (max_blocks > 0 && g.blocks > max_blocks) g.blocks = max_blocks;
    if (g.blocks > n) g.blocks = n > 0 ? n : 1;
    // (tried: as many workgroups as make the rounds even — 3 334 for 10 000 pairs instead of 4 096 + a 44 % third round:
    //  95.5k vs 108k pairs/s kernel-only; resident waves are worth more than an even tail)
    // value store per workgroup: 4 / 8 MB (one / three values per entry); four times that for the widest beam class
    // and for the pass over the pairs the two-pairs-per-wave path handed back (max_blocks > 0: few workgroups, and
    // those pairs are the ones with windows hundreds of frames wide, whose live rows grow with the window squared)
    // (the W <= 6 class, with 16 workgroups per CU, gets half of that on its direct path: its 112 row groups fit
    // windows up to 126 frames in it, and a pair that runs out is decoded again by the retry pass below)
    g.pool_bytes = al256((K == 1 ? (size_t)4 : (size_t)8) << ((g.wclass == 25 || max_blocks > 0) ? 22 : (g.wclass == 6 ? 19 : 20)));
    const int64_t WM = W > PO_A ? W : PO_A;
    const int64_t steps = (method == PO_METHOD_ROW) ? mr1 : std::min(mr1, mr2);
    g.arena_cap = ((size_t)(1 + PO_A + (int64_t)PO_A * WM * (steps + 1)) + 1) & ~size_t(1);  // even: a double array follows
    g.tcap = (size_t)std::max(mr1, mr2);
    g.vcap = (size_t)mr2;
    {   // long reads: the per-workgroup arena grows with the read length — fewer resident workgroups rather than a
        // workspace beyond the board's memory (reads of 10^5 frames: ~40 MB each)
        const size_t per_block = g.pool_bytes + sizeof(int) * 5 * g.arena_cap + sizeof(double) * 2 * g.tcap + sizeof(int) * 2 * g.vcap;
        const size_t fit = std::max<size_t>(1, b2_mem_budget() / std::max<size_t>(per_block, 1));
        if ((size_t)g.blocks > fit) g.blocks = (int)fit;
    }
    size_t o = 0;
    g.off_queue = o; o += 256;
    g.off_state = o; o += al256(sizeof(unsigned long long) * 2 * (size_t)g.blocks);
    g.off_pool = o; o += g.pool_bytes * g.blocks;
    g.off_arena = o; o += al256(sizeof(int) * 5 * g.arena_cap * g.blocks);  // 3 int arrays + 1 double array
    // the word a workgroup leaves behind in its state slot: any change of the store's geometry changes it
    g.magic = 0x9e3779b97f4a7c15ull ^ ((unsigned long long)g.pool_bytes * 0x100000001b3ull) ^ ((unsigned long long)g.blocks << 40) ^
              ((unsigned long long)g.wclass << 32) ^ ((unsigned long long)K << 36);
    g.off_cum = o; o += al256(sizeof(double) * 2 * g.tcap * g.blocks);
    g.off_envt = o; o += al256(sizeof(int) * 2 * g.vcap * g.blocks);
    g.off_order = o; o += al256(sizeof(int) * (size_t)std::max(n, 1));   // the queue's order (pair_order_kernel)
    g.total = o + 256;
    return g;
}

// The order in which a launch's persistent workgroups take their pairs: longest first (frames of both reads, 2048
// classes, descending; within a class as the atomics fall).  The tail of a launch — the last pair of every workgroup,
// running on a half-empty device — is then made of the SHORTEST pairs instead of whatever came last in the input:
// worth a few percent on the bench's pairs (lengths within 10 %), most of the tail on real reads, whose lengths differ
// tenfold.  Results do not depend on it (every pair is decoded by one workgroup, alone).  One workgroup: a histogram,
// its prefix sums and a scatter, all in LDS.
constexpr int ORD_BINS = 2048;
__global__ __launch_bounds__(1024) void pair_order_kernel(const int64_t* y1_off, const int64_t* y2_off, int n, int* order) {
    __shared__ int hist[ORD_BINS];
    __shared__ int part[1024 / 64];
    __shared__ long long cmax_s;
    const int tid = threadIdx.x;
    for (int b = tid; b < ORD_BINS; b += 1024) hist[b] = 0;
    if (tid == 0) cmax_s = 1;
    __syncthreads();
    long long cm = 1;
    for (int i = tid; i < n; i += 1024) cm = max(cm, (long long)((y1_off[i + 1] - y1_off[i]) + (y2_off[i + 1] - y2_off[i])));
    atomicMax((unsigned long long*)&cmax_s, (unsigned long long)cm);
    __syncthreads();
    const long long cmax = cmax_s;
    auto bin_of = [&](int i) -> int {
        const long long c = (y1_off[i + 1] - y1_off[i]) + (y2_off[i + 1] - y2_off[i]);
        const long long b = (max(c, 0ll) * (ORD_BINS - 1)) / cmax;
        return ORD_BINS - 1 - (int)min(b, (long long)(ORD_BINS - 1));   // longest -> bin 0
    };
    for (int i = tid; i < n; i += 1024) atomicAdd(&hist[bin_of(i)], 1);
    __syncthreads();
    // exclusive prefix sums of the 2048 counts: two per thread, wave scan, wave totals through LDS
    const int c0 = hist[2 * tid], c1 = hist[2 * tid + 1];
    int v = c0 + c1;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o); if (lane >= o) v += t; }
    if (lane == 63) part[wave] = v;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += part[w];
    __syncthreads();
    const int ex = base + v - (c0 + c1);
    hist[2 * tid] = ex; hist[2 * tid + 1] = ex + c0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) order[atomicAdd(&hist[bin_of(i)], 1)] = i;
}

constexpr int X2_FB_BLOCKS = 64;   // workgroups of the beam2d_kernel pass over deferred pairs (16 / 32 MB of store each)
// Kernel routing is a process-wide setting of the library (po_set_pair_route), not something a launch looks up in the
// environment: PO_ROUTE_AUTO (the engine's choice: beam2d_reg_kernel wherever it applies — row_col with an envelope, W <= 12,
// every tree model — and beam2d_kernel elsewhere), PO_ROUTE_REG (the same, named), PO_ROUTE_LEGACY (always beam2d_kernel) —
// for the tests, which run the pair path on both kernels, and for A/B timing.  The environment variables PO_B2_LEGACY /
// PO_REG_NEVER / PO_X2_DEFER_ODD only give the INITIAL value, read once when the library is first used, so that a workspace
// size and the launch that follows always agree.  (Rounds 1 - 4 had two more kernels and routes — two pairs per wave, LDS
// rings — which the register-state kernel replaced at every size: DESIGN.md, appendix.)
struct B2Route { int route, defer_odd, debug_occ, no_order, reg_auto, chain_scan; };
B2Route& b2_route() {
    static B2Route r = [] {
        B2Route x;
        x.route = getenv("PO_B2_LEGACY") ? PO_ROUTE_LEGACY : (getenv("PO_REG_FORCE") ? PO_ROUTE_REG : PO_ROUTE_AUTO);
        x.reg_auto = getenv("PO_REG_NEVER") ? 0 : 1;      // A/B: beam2d_kernel for everything
        x.defer_odd = getenv("PO_X2_DEFER_ODD") ? 1 : 0;
        x.no_order = getenv("PO_B2_NO_ORDER") ? 1 : 0;   // A/B: pairs taken in input order
        x.debug_occ = getenv("PO_DEBUG_OCC") ? 1 : 0;
        x.chain_scan = getenv("PO_CHAIN_CLOSED") ? PO_CHAIN_CLOSED_FORM : PO_CHAIN_SERIAL;   // (po_set_chain_mode)
        return x;
    }();
    return r;
}
void (*g_b2_mark_fwd)(int begin, hipStream_t stream) = nullptr;   // set through po_b2_set_mark
// The register-state kernel (beam2d_reg_kernel, po_beam2d_reg.hip) with its batch-parallel pre-pass and walk kernels
// (po_beam2d_pre.h): the engine's choice (PO_ROUTE_AUTO) for row_col with an envelope at every batch size, every tree
// model, W <= 12.  PO_REG_NEVER / PO_ROUTE_LEGACY send everything to beam2d_kernel.
struct RegGeom {
    int blocks;        // pair waves (one-wave workgroups) of this launch
    size_t off_queue, off_meta, off_nmain, off_sched, off_envt, off_order, off_fb, fb_bytes, total;
};
extern "C" int po_reg_slots_per_cu(int model, int wide);
extern "C" int po_reg_max_elements(int wide);
extern "C" int po_reg_ngl(int wide);
extern "C" size_t po_reg_pool_bytes(int model, int wide);
extern "C" void po_reg_launch(const void* x2args, int slots, int model, int wide, hipStream_t stream);
// the 64-slot layout of the kernel (lane = element slot, the two reads one after the other): 7 <= W <= 12
inline int reg_wide(int W) {
    static const int force = getenv("PO_REG_FORCE_WIDE") ? 1 : 0;   // (experiments: the 64-slot layout for every width)
    return (W > 6 || force) ? 1 : 0;
}
struct RegPool;
RegPool* reg_pool(int model, int wide);
bool reg_eligible(int n, int W, int A, int model, int method) {
    const int rt = b2_route().route;
    // (every tree model since round 5: the kernel is templated on the model's values per entry)
    if (!(method == PO_METHOD_ROW_COL && W <= 12 && A >= 1 && W * (A + 1) <= po_reg_max_elements(reg_wide(W)))) return false;
    if (model != PO_MODEL_CTC && model != PO_MODEL_MERGE && model != PO_MODEL_FLIPFLOP) return false;
    if (!(rt == PO_ROUTE_REG || (rt == PO_ROUTE_AUTO && b2_route().reg_auto != 0))) return false;
    // The route needs the library's slice pool (below).  It is made HERE — by the workspace-size query that precedes every
    // launch, a host-side call — and not inside the enqueue-only launch (multi-GB hipMalloc calls, a memset and a
    // copy on the null stream do not belong in a stream capture).  A pool that cannot be allocated is remembered as such: the size
    // query then answers for beam2d_kernel and the launch takes that kernel — the two always agree.
    {   // (a host without a device can only PLAN — the CPU tests' size queries: answer for the engine's route)
        static const bool no_device = [] { int nd = 0; const bool none = hipGetDeviceCount(&nd) != hipSuccess || nd <= 0; (void)hipGetLastError(); return none; }();
        if (no_device) return true;
    }
    return reg_pool(model, reg_wide(W)) != nullptr;
}

// ---- The library's SLICE POOL of the register-state kernel (one per device, tree model and lane layout; made at the first
// launch that needs it, kept for the life of the process).  A slice = the value store of one pair wave (2 - 8 MB: 128 row
// groups at R = 128) + its tree arena.  Rounds 1 - 4 carved these out of every call's workspace: 12 GB for a 10 000-pair call,
// 28 GB per slot of a pipeline — and a hipMalloc of that size stalls for 0.5 - 2.5 s on this driver every now and then
// (scripts/micro/malloc_cost.hip: never below 4 GB, one in three at 16 GB), which is what a process's FIRST call paid.
// There are never more pair waves on the device than its register file and LDS admit, whatever the number of launches in
// flight: ONE pool of that many slices serves them all, in chunks of at most 3.5 GB; a wave claims a slice when it starts
// (po_beam2d_reg.hip).  The slices hold VALUES only (no tags, no epochs: presence is bookkeeping in the kernel), so nothing is ever
// memset or cleared.  po_reg_pool_prewarm makes a pool ahead of the first launch, po_reg_pool_release frees the current device's.
struct RegPool {
    int nslices = 0, spc_log2 = 0;
    size_t pool_bytes = 0, slice_bytes = 0;
    long long arena_cap = 0;
    char* chunk[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int* claim = nullptr;
    char* words = nullptr;                       // the allocation claim / defer_count / tickets live in
    unsigned long long* defer_count = nullptr;   // pairs handed to beam2d_kernel since the last reset (tests)
    unsigned* tickets = nullptr;                 // {take, give} tickets of the ring of free slices (claim)
};
std::mutex g_reg_pool_mu;
RegPool* g_reg_pools[PO_MAX_DEVICES][6] = {};
bool g_reg_pool_failed[PO_MAX_DEVICES][6] = {};   // (sticky until po_reg_pool_release: the size query and the launch must agree)
inline int reg_pool_key(int model, int wide) { return (model == PO_MODEL_CTC ? 0 : (model == PO_MODEL_MERGE ? 1 : 2)) * 2 + (wide ? 1 : 0); }
RegPool* reg_pool(int model, int wide) {
    const int dev = po_cur_device();
    const int key = reg_pool_key(model, wide);
    std::lock_guard<std::mutex> lk(g_reg_pool_mu);
    if (g_reg_pools[dev][key]) return g_reg_pools[dev][key];
    if (g_reg_pool_failed[dev][key]) return nullptr;
    RegPool* p = new RegPool();
    p->nslices = b2_num_cus() * po_reg_slots_per_cu(model, wide);
    // (PO_REG_POOL_SLICES: a smaller pool — several processes sharing one board, each with its own 9 GB otherwise; launches
    //  are cut to that many waves)
    if (const char* e = getenv("PO_REG_POOL_SLICES")) { const int v = atoi(e); if (v >= 64 && v < p->nslices) p->nslices = v & ~1; }
    p->pool_bytes = po_reg_pool_bytes(model, wide);
    // tree nodes a slice's arena holds: four per node that ever enters the beam.  1 024 beam entries per beam slot is ~ 4 x what
    // a T = 4000 pair makes; a pair that needs more is handed to beam2d_kernel (whose arenas are worst-case sized)
    const long long WM = wide ? 12 : 6;
    p->arena_cap = 1 + PO_A + (long long)PO_A * WM * 1024;
    // (+ the row headers: one int per store row and read)
    p->slice_bytes = al256(p->pool_bytes + sizeof(int) * 3 * (size_t)p->arena_cap + sizeof(int) * 2 * PO_A * (size_t)po_reg_ngl(wide));
    int spc = 1;
    while ((size_t)(2 * spc) * p->slice_bytes <= ((size_t)7 << 29) && 2 * spc <= p->nslices) spc *= 2;   // chunks of <= 3.5 GB
    while ((p->nslices + spc - 1) / spc > 8) spc *= 2;
    p->spc_log2 = 0;
    while ((1 << p->spc_log2) < spc) ++p->spc_log2;
    const int nchunks = (p->nslices + spc - 1) / spc;
    bool ok = true;
    for (int c = 0; c < nchunks && ok; ++c) {
        const int here = std::min(spc, p->nslices - c * spc);
        ok = hipMalloc((void**)&p->chunk[c], (size_t)here * p->slice_bytes) == hipSuccess;
    }
    // the ring of free slices (one int per slice), then — 256-byte aligned — the deferral counter and the two tickets
    const size_t ring = al256(sizeof(int) * (size_t)p->nslices), words = ring + 256;
    char* w = nullptr;
    ok = ok && hipMalloc((void**)&w, words) == hipSuccess && hipMemset(w, 0, words) == hipSuccess;
    if (ok) {   // every slice is free: ring word i holds slice i
        std::vector<int> ids((size_t)p->nslices);
        for (int i = 0; i < p->nslices; ++i) ids[(size_t)i] = i;
        ok = hipMemcpy(w, ids.data(), sizeof(int) * ids.size(), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) {
        (void)hipGetLastError();
        for (auto c : p->chunk) if (c) (void)hipFree(c);
        if (w) (void)hipFree(w);
        delete p;
        g_reg_pool_failed[dev][key] = true;
        if (b2_route().debug_occ) fprintf(stderr, "[po] register-state kernel pool (model %d, %s layout) could not be allocated: beam2d_kernel serves the route\n", model, wide ? "64-slot" : "32-slot");
        return nullptr;
    }
    p->words = w;
    p->claim = (int*)w;
    p->defer_count = (unsigned long long*)(w + ring);
    p->tickets = (unsigned*)(w + ring + 128);
    if (b2_route().debug_occ)
        fprintf(stderr, "[po] register-state kernel pool (model %d, %s layout): %d slices of %.2f MB in %d chunk(s)\n", model, wide ? "64-slot" : "32-slot",
                p->nslices, p->slice_bytes / 1048576.0, nchunks);
    g_reg_pools[dev][key] = p;
    return p;
}
// Per-call workspace of the route: the batch-sized arrays of the pre-pass and the walk, and the deferred-pairs pass.
RegGeom reg_geometry(int n, int64_t tr1, int64_t tr2, int64_t mr1, int64_t mr2, int W, int model) {
    (void)tr1;
    RegGeom g;
    g.blocks = b2_num_cus() * po_reg_slots_per_cu(model, reg_wide(W));
    if (g.blocks > n) g.blocks = n > 0 ? n : 1;
    const size_t np = (size_t)(n > 0 ? n : 1);
    size_t o = 0;
    g.off_queue = o; o += 256;
    g.off_meta = o; o += al256(sizeof(int2) * np);
    g.off_nmain = o; o += al256(sizeof(int) * np);
    g.off_sched = o; o += al256(sizeof(int4) * (size_t)(tr2 > 0 ? tr2 : 1));
    g.off_envt = o; o += al256(sizeof(int) * 2 * (size_t)tr2);
    g.off_order = o; o += al256(sizeof(int) * np);   // the queue's order (pair_order_kernel)
    g.off_fb = o;
    g.fb_bytes = b2_geometry(n, mr1, mr2, W, model, PO_METHOD_ROW_COL, X2_FB_BLOCKS).total;
    o += al256(g.fb_bytes);
    g.total = o + 256;
    return g;
}

// ---- grid method: one workgroup per pair in flight; per workgroup a value store, the tree arena, the blank
// prefix sums and two rows of cell beams
struct GridGeom {
    int threads, blocks, wclass;
    size_t pool_bytes, arena_cap, tcap, vcap, cell_ints;
    size_t off_queue, off_pool, off_arena, off_cum, off_cell, total;
};
constexpr int GRID_NGL = 3072;
GridGeom grid_geometry(int n, int64_t mr1, int64_t mr2, int W, int model, bool has_env) {
    GridGeom g;
    const int K = (model == PO_MODEL_CTC) ? 1 : 3;
    g.wclass = W <= 6 ? 6 : 25;
    g.threads = g.wclass == 6 ? 64 : 256;
    // value store: with an envelope, room for every row group at bands up to 62 wide (R = 64; wider bands get
    // fewer groups); without one a node keeps all V read-1 times (R >= V + 2), which only fits short reads
    int64_t Rh = 64;
    if (!has_env) while (Rh < mr2 + 2) Rh <<= 1;
    g.pool_bytes = al256(std::min<size_t>((size_t)GRID_NGL * PO_A * 2 * (size_t)Rh * (K == 1 ? 16 : 32), (size_t)1 << 30));
    // nodes: every beam node of every cell may be expanded; distinct ones per row are a few times W
    const int64_t WM = W > PO_A ? W : PO_A;
    g.arena_cap = (size_t)std::min<int64_t>((int64_t)1 << 24, 1 + PO_A + (int64_t)PO_A * WM * 8 * (mr1 + mr2 + 2));
    g.tcap = (size_t)std::max(mr1, mr2);
    g.vcap = (size_t)mr2;
    g.cell_ints = 2 * g.vcap * (size_t)(1 + W * G_COUNT);
    const size_t per_block = g.pool_bytes + sizeof(int) * 3 * g.arena_cap + sizeof(double) * 2 * g.tcap + sizeof(int) * g.cell_ints;
    g.blocks = b2_num_cus() * (g.wclass == 6 ? 4 : 1);
    g.blocks = (int)std::min<size_t>((size_t)g.blocks, std::max<size_t>(1, ((size_t)16 << 30) / per_block));  // <= 16 GB in all
    if (g.blocks > n) g.blocks = n > 0 ? n : 1;
    size_t o = 0;
    g.off_queue = o; o += 256;
    g.off_pool = o; o += g.pool_bytes * g.blocks;
    g.off_arena = o; o += al256(sizeof(int) * 3 * g.arena_cap * g.blocks);
    g.off_cum = o; o += al256(sizeof(double) * 2 * g.tcap * g.blocks);
    g.off_cell = o; o += al256(sizeof(int) * g.cell_ints * g.blocks);
    g.total = o + 256;
    return g;
}
template <int MODEL>
void grid_launch_w(const GridGeom& g, const B2Args& a, hipStream_t stream) {
    if (g.wclass == 6) hipLaunchKernelGGL((beam2d_grid_kernel<MODEL, 6>), dim3(g.blocks), dim3(g.threads), 0, stream, a);
    else hipLaunchKernelGGL((beam2d_grid_kernel<MODEL, 25>), dim3(g.blocks), dim3(g.threads), 0, stream, a);
}

template <int MODEL, int WMAX>
void b2_launch(const B2Geom& g, const B2Args& a, hipStream_t stream) {
    if constexpr (WMAX == 6) {
        if (a.method != PO_METHOD_ROW) {
            hipLaunchKernelGGL((beam2d_kernel<MODEL, WMAX, true>), dim3(g.blocks), dim3(g.threads), 0, stream, a);
            return;
        }
    }
    hipLaunchKernelGGL((beam2d_kernel<MODEL, WMAX>), dim3(g.blocks), dim3(g.threads), 0, stream, a);
}
template <int MODEL>
void b2_launch_w(const B2Geom& g, const B2Args& a, hipStream_t stream) {
    if (g.wclass == 6) b2_launch<MODEL, 6>(g, a, stream);
    else if (g.wclass == 12) b2_launch<MODEL, 12>(g, a, stream);
    else b2_launch<MODEL, 25>(g, a, stream);
}
}  // namespace

extern "C" size_t po_beam2d_ws_bytes_impl(int n, int64_t tr1, int64_t tr2, int64_t mr1, int64_t mr2, int C, int W,
                                          int model, int method) {
    (void)C;
    if (method == PO_METHOD_GRID) return grid_geometry(n, mr1, mr2, W, model, true).total;
    if (method == PO_METHOD_GRID_NOENV) return grid_geometry(n, mr1, mr2, W, model, false).total;
    if (reg_eligible(n, W, (model == PO_MODEL_FLIPFLOP) ? C / 2 : C - 1, model, method)) return reg_geometry(n, tr1, tr2, mr1, mr2, W, model).total;
    return b2_geometry(n, mr1, mr2, W, model, method).total + b2_geometry(n, mr1, mr2, W, model, method, X2_FB_BLOCKS).total;
}

namespace {
// The pair-beam kernels keep {magic, epoch} words per workgroup in the workspace instead of clearing their value
// store per launch.  That is only sound while the same layout is used on the same memory: a caller that reuses one
// buffer for waves of different geometry moves the sub-workspace, and state words of one layout can then survive
// inside the other's store while both keep writing tags into overlapping memory.  The host therefore remembers, per
// workspace base pointer, the layout of the last pair-beam launch there; a launch with another layout zeroes the
// (small) state region first, which makes every workgroup clear its slice.
struct B2Layout { size_t off_state, total; unsigned long long magic; };
std::mutex g_b2_layout_mu;
std::unordered_map<const void*, B2Layout> g_b2_layouts;
bool b2_ws_layout_changed(const void* ws, size_t off_state, size_t total, unsigned long long magic) {
    std::lock_guard<std::mutex> lk(g_b2_layout_mu);
    auto it = g_b2_layouts.find(ws);
    const bool same = it != g_b2_layouts.end() && it->second.off_state == off_state && it->second.total == total && it->second.magic == magic;
    // any other remembered buffer that overlaps this one is stale too (freed and reallocated memory)
    if (!same) {
        for (auto jt = g_b2_layouts.begin(); jt != g_b2_layouts.end();) {
            const char* b = (const char*)jt->first;
            if (jt->first != ws && b < (const char*)ws + total && (const char*)ws < b + jt->second.total) jt = g_b2_layouts.erase(jt);
            else ++jt;
        }
        g_b2_layouts[ws] = B2Layout{off_state, total, magic};
    }
    return !same;
}
unsigned long long* g_b2_upd_counter = nullptr;
void (*g_b2_mark)(int begin, hipStream_t stream) = nullptr;   // profiling: brackets the main pair beam kernel
}
extern "C" void po_b2_set_mark(void (*f)(int, hipStream_t)) { g_b2_mark = f; g_b2_mark_fwd = f; }
extern "C" int po_set_pair_route(int route, int defer_odd) {
    if (route != PO_ROUTE_AUTO && route != PO_ROUTE_LEGACY && route != PO_ROUTE_REG) return PO_E_ARG;   // (PO_ROUTE_X2 / PO_ROUTE_RING: kernels retired in round 5)
    b2_route().route = route;
    b2_route().defer_odd = defer_odd & 7;   // bit 0: odd pairs are handed on; bits 1, 2: starve the row groups / the arena
    return PO_OK;
}
// The slice pool ahead of time / given back (include/poreover_hip.h).
extern "C" int po_reg_pool_prewarm(int model, int beam_width) {
    if (model != PO_MODEL_CTC && model != PO_MODEL_MERGE && model != PO_MODEL_FLIPFLOP) return PO_E_ARG;
    if (beam_width < 1 || beam_width > 12) return PO_E_ARG;
    return reg_pool(model, reg_wide(beam_width)) ? PO_OK : PO_E_NOMEM;
}
extern "C" int po_reg_pool_release(void) {
    if (hipDeviceSynchronize() != hipSuccess) return PO_E_HIP;   // (no pair wave may still hold a slice)
    const int dev = po_cur_device();
    std::lock_guard<std::mutex> lk(g_reg_pool_mu);
    for (int k = 0; k < 6; ++k) {
        g_reg_pool_failed[dev][k] = false;
        RegPool* p = g_reg_pools[dev][k];
        if (!p) continue;
        for (auto c : p->chunk) if (c) (void)hipFree(c);
        if (p->words) (void)hipFree(p->words);
        delete p;
        g_reg_pools[dev][k] = nullptr;
    }
    return PO_OK;
}
// How beam2d_reg_kernel computes a NEW element's window (po_beam2d_reg.hip, "closed form"): see include/poreover_hip.h.
extern "C" int po_set_chain_mode(int mode) {
    if (mode != PO_CHAIN_SERIAL && mode != PO_CHAIN_CLOSED_FORM && mode != PO_CHAIN_CLOSED_GUARD3) return PO_E_ARG;
    b2_route().chain_scan = mode;
    return PO_OK;
}
extern "C" int po_get_chain_mode(void) { return b2_route().chain_scan; }
// profiling: a device counter that the pair beam kernels add their number of update_prob evaluations to
extern "C" void po_b2_set_update_counter(unsigned long long* dev_counter) { g_b2_upd_counter = dev_counter; }
// tests: pairs the register-state kernel (or its pre-pass) handed to beam2d_kernel on this device since the last reset, summed
// over the pools in use (synchronises the device)
extern "C" long long po_debug_deferred_pairs(int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    const int dev = po_cur_device();
    long long tot = 0;
    std::lock_guard<std::mutex> lk(g_reg_pool_mu);
    for (RegPool* p : g_reg_pools[dev]) {
        if (!p) continue;
        unsigned long long v = 0;
        if (hipMemcpy(&v, p->defer_count, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        tot += (long long)v;
        if (reset && hipMemset(p->defer_count, 0, sizeof(v)) != hipSuccess) return -1;
    }
    return tot;
}

// ---- logaddexp micro-benchmark: the peak rate of the specialised logaddexp on this device (4 independent
// chains per lane, tables in LDS, every lane busy) — the compute ceiling the pair kernels are priced against
__global__ __launch_bounds__(256) void lae_peak_kernel(int iters, double* sink) {
    __shared__ PoLaeTables tb;
    po_lae_tables_load(&tb, threadIdx.x, 256);
    __syncthreads();
    const PoLaeFast lae{&tb};
    const double seed = -1.0 - 1e-3 * (threadIdx.x + 1);
    double a0 = seed, a1 = seed - 0.5, a2 = seed - 1.5, a3 = seed - 2.5;
    const double y0 = -0.11, y1 = -2.3;
    for (int i = 0; i < iters; ++i) {
        a0 = lae(a0 + y1, a1 + y0); a1 = lae(a1 + y1, a2 + y0); a2 = lae(a2 + y1, a3 + y0); a3 = lae(a3 + y1, a0 + y0);
    }
    if (a0 + a1 + a2 + a3 == 12345.678) sink[0] = a0;
}
extern "C" int po_launch_lae_peak(int iters, double* lae_per_s, hipStream_t stream) {
    double* sink = nullptr;
    if (hipMalloc((void**)&sink, 8) != hipSuccess) return PO_E_HIP;
    const int blocks = b2_num_cus() * 8;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(lae_peak_kernel, dim3(blocks), dim3(256), 0, stream, 64, sink);  // warm-up
    (void)hipEventRecord(e0, stream);
    hipLaunchKernelGGL(lae_peak_kernel, dim3(blocks), dim3(256), 0, stream, iters, sink);
    (void)hipEventRecord(e1, stream);
    int rc = PO_OK;
    float ms = 0.f;
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f) rc = PO_E_HIP;
    else *lae_per_s = 4.0 * iters * 256.0 * blocks / (ms * 1e-3);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(sink);
    return rc;
}

namespace {
int b2_launch_legacy(const double* y1, const int64_t* y1_off, const double* y2, const int64_t* y2_off, const int32_t* env,
                     int n, int C, int A, uint32_t alphabet, int W, int model, int method, int64_t mr1, int64_t mr2,
                     char* seq, const int64_t* seq_off, int32_t* seq_len, int32_t* status, int use_pre_status, void* ws,
                     size_t ws_bytes, hipStream_t stream, int max_blocks, const int2* only_meta, int retry = 0,
                     const int* retry_flag = nullptr) {
    const B2Geom g = b2_geometry(n, mr1, mr2, W, model, method, max_blocks);
    if (ws_bytes < g.total) return PO_E_CAP;
    // direct path: a second, small pass (64 workgroups, four times the store) decodes the pairs whose live rows did
    // not fit the first one's store — windows hundreds of frames wide; an empty pass costs its 1 GB memset
    if (max_blocks == 0 && !retry) {
        const size_t rb = b2_geometry(n, mr1, mr2, W, model, method, X2_FB_BLOCKS).total;
        if (ws_bytes < g.total + rb) return PO_E_CAP;
    }
    char* w = (char*)ws;
    B2Args a;
    a.y1 = y1; a.y1_off = y1_off; a.y2 = y2; a.y2_off = y2_off; a.env = env;
    a.n = n; a.A = A; a.W = W; a.C = C; a.method = method; a.alphabet = alphabet;
    a.seq = seq; a.seq_off = seq_off; a.seq_len = seq_len; a.status = status;
    a.use_pre_status = use_pre_status;
    a.queue = (int*)(w + g.off_queue);
    a.pool = w + g.off_pool; a.pool_bytes = g.pool_bytes;
    a.arena = (int*)(w + g.off_arena); a.arena_cap = (long long)g.arena_cap;
    a.cum = (double*)(w + g.off_cum); a.tcap = (long long)g.tcap;
    a.envt = (int*)(w + g.off_envt); a.vcap = (long long)g.vcap;
    a.dbg = nullptr;
    a.only_meta = only_meta;
    a.cellb = nullptr;
    a.retry_nomem = retry;
    a.retry_flag = retry_flag;
    a.upd_count = g_b2_upd_counter;
    a.wgstate = (unsigned long long*)(w + g.off_state);
    a.magic = g.magic;
#ifdef PO_B2_TIMING
    static long long* dbg_buf = nullptr;
    if (!dbg_buf) { (void)hipMalloc((void**)&dbg_buf, 37 * sizeof(long long)); (void)hipMemset(dbg_buf, 0, 37 * sizeof(long long)); }
    a.dbg = dbg_buf;
#endif
    // the queue counter starts from zero on every launch; the value store is NOT cleared: its tags are told apart by
    // the epoch counters the workgroups keep in the workspace (see beam2d_kernel) — unless this memory was last used
    // with another layout (b2_ws_layout_changed)
    if (po_zero_async(w + g.off_queue, 256, stream) != hipSuccess) return PO_E_HIP;
    if (b2_ws_layout_changed(ws, g.off_state, g.total, g.magic) &&
        po_zero_async(w + g.off_state, sizeof(unsigned long long) * 2 * (size_t)g.blocks, stream) != hipSuccess)
        return PO_E_HIP;
    a.order = nullptr;
    if (n > g.blocks && !only_meta && !retry && !b2_route().no_order) {   // more pairs than resident workgroups: the order of the queue matters
        a.order = (int*)(w + g.off_order);
        hipLaunchKernelGGL(pair_order_kernel, dim3(1), dim3(1024), 0, stream, y1_off, y2_off, n, (int*)(w + g.off_order));
    }
    if (g_b2_mark && !only_meta && !retry) g_b2_mark(1, stream);
    if (model == PO_MODEL_CTC) b2_launch_w<PO_MODEL_CTC>(g, a, stream);
    else if (model == PO_MODEL_MERGE) b2_launch_w<PO_MODEL_MERGE>(g, a, stream);
    else if (model == PO_MODEL_FLIPFLOP) b2_launch_w<PO_MODEL_FLIPFLOP>(g, a, stream);
    else return PO_E_ARG;
    if (g_b2_mark && !only_meta && !retry) g_b2_mark(0, stream);
#ifdef PO_B2_TIMING
    {
        long long h[37];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h, a.dbg, sizeof(h), hipMemcpyDeviceToHost);
        const char* nm[12] = {"prepass+init", "main:prune+nextbeam", "main:expand+table", "main:scan selfread", "main:scan staging",
                              "main:scan iterations", "catchup:setup", "catchup:selfread", "catchup:staging", "catchup:iterations",
                              "label walk", "#steps (col 1: catch-ups)"};
        fprintf(stderr, "[po_b2_timing] block 0, wall_clock64 ticks (100 MHz => 10 ns each): full-window | steady table | after a permutation\n");
        for (int i = 0; i < 12; ++i) fprintf(stderr, "   %-26s %12lld %12lld %12lld\n", nm[i], h[i], h[12 + i], h[24 + i]);
        fprintf(stderr, "   %-26s %12lld\n", "#full-window main steps", h[36]);
    }
#endif
    if (max_blocks == 0 && !retry)
        return b2_launch_legacy(y1, y1_off, y2, y2_off, env, n, C, A, alphabet, W, model, method, mr1, mr2, seq, seq_off, seq_len,
                                status, use_pre_status, (char*)ws + g.total, ws_bytes - g.total, stream, X2_FB_BLOCKS, nullptr, 1,
                                a.queue + 8);
    return PO_OK;
}
}  // namespace

extern "C" int po_launch_beam2d_geom(const double* y1, const int64_t* y1_off, const double* y2, const int64_t* y2_off,
                                     const int32_t* env, int n, int C, int A, uint32_t alphabet, int W, int model,
                                     int method, int64_t tr1, int64_t tr2, int64_t mr1, int64_t mr2, char* seq,
                                     const int64_t* seq_off, int32_t* seq_len, int32_t* status, int use_pre_status,
                                     void* ws, size_t ws_bytes, hipStream_t stream) {
    if (n <= 0) return PO_OK;
    if (A < 1 || A > PO_A || W < 1 || W > 25) return PO_E_ARG;
    if (method != PO_METHOD_ROW_COL && method != PO_METHOD_ROW && method != PO_METHOD_GRID) return PO_E_ARG;
    if ((model == PO_MODEL_FLIPFLOP) ? (C != 2 * A) : (C != A + 1)) return PO_E_ARG;
    if (model != PO_MODEL_CTC && model != PO_MODEL_MERGE && model != PO_MODEL_FLIPFLOP) return PO_E_ARG;
    // without an envelope the reference's dispatcher knows "row" and sends everything else to grid (BeamSearch.h:441-458)
    if (method == PO_METHOD_GRID || (!env && method != PO_METHOD_ROW)) {
        const GridGeom g = grid_geometry(n, mr1, mr2, W, model, env != nullptr);
        if (ws_bytes < g.total) return PO_E_CAP;
        char* w = (char*)ws;
        B2Args a;
        a.y1 = y1; a.y1_off = y1_off; a.y2 = y2; a.y2_off = y2_off; a.env = env;
        a.n = n; a.A = A; a.W = W; a.C = C; a.method = PO_METHOD_GRID; a.alphabet = alphabet;
        a.seq = seq; a.seq_off = seq_off; a.seq_len = seq_len; a.status = status;
        a.use_pre_status = use_pre_status;
        a.queue = (int*)(w + g.off_queue);
        a.pool = w + g.off_pool; a.pool_bytes = g.pool_bytes;
        a.arena = (int*)(w + g.off_arena); a.arena_cap = (long long)g.arena_cap;
        a.cum = (double*)(w + g.off_cum); a.tcap = (long long)g.tcap;
        a.envt = nullptr; a.vcap = (long long)g.vcap;
        a.cellb = (int*)(w + g.off_cell);
        a.retry_nomem = 0;
        a.retry_flag = nullptr;
        a.order = nullptr;
        a.dbg = nullptr; a.only_meta = nullptr;
        a.upd_count = g_b2_upd_counter;
        a.wgstate = nullptr; a.magic = 0;   // (the grid kernel clears its store per launch)
        if (po_zero_async(w + g.off_queue, 256, stream) != hipSuccess) return PO_E_HIP;
        if (po_zero_async(w + g.off_pool, g.pool_bytes * g.blocks, stream) != hipSuccess) return PO_E_HIP;
        if (model == PO_MODEL_CTC) grid_launch_w<PO_MODEL_CTC>(g, a, stream);
        else if (model == PO_MODEL_MERGE) grid_launch_w<PO_MODEL_MERGE>(g, a, stream);
        else grid_launch_w<PO_MODEL_FLIPFLOP>(g, a, stream);
        return PO_OK;
    }
    if (reg_eligible(n, W, A, model, method)) {
        const RegGeom g = reg_geometry(n, tr1, tr2, mr1, mr2, W, model);
        if (ws_bytes < g.total) return PO_E_CAP;
        RegPool* const rp = reg_pool(model, reg_wide(W));   // (made by the size query before this launch: reg_eligible)
        if (!rp) return PO_E_NOMEM;                          // (cannot happen: eligibility said it exists)
        char* w = (char*)ws;
        X2Args a;
        a.y1 = y1; a.y1_off = y1_off; a.y2 = y2; a.y2_off = y2_off; a.env = env;
        a.n = n; a.A = A; a.W = W; a.C = C; a.alphabet = alphabet;
        a.seq = seq; a.seq_off = seq_off; a.seq_len = seq_len; a.status = status; a.use_pre_status = use_pre_status;
        a.queue = (int*)(w + g.off_queue);
        a.meta = (int2*)(w + g.off_meta);
        a.nmain = (int*)(w + g.off_nmain);
        a.sched = (int4*)(w + g.off_sched);
        a.envt = (int*)(w + g.off_envt);
        a.cum1 = nullptr; a.cum2 = nullptr;   // (the kernel adds the ctc root's alpha up as its scans pass the times)
        a.pool = nullptr; a.pool_bytes = rp->pool_bytes;
        a.arena = nullptr; a.arena_cap = rp->arena_cap;
        for (int c = 0; c < 8; ++c) a.slice_chunk[c] = rp->chunk[c];
        a.slice_spc_log2 = rp->spc_log2; a.nslices = rp->nslices; a.slice_bytes = rp->slice_bytes; a.slice_claim = rp->claim;
        a.slice_tickets = rp->tickets;
        a.defer_count = rp->defer_count;
        {   // PO_REG_PERSIST=0 / 1 pins the launch form (A/B)
            static const int env = [] { const char* e = getenv("PO_REG_PERSIST"); return e ? atoi(e) : -1; }();
            a.persist = env >= 0 ? (env != 0) : 1;
        }
        a.starve = (b2_route().defer_odd >> 1) & 3;
        a.wgstate = nullptr; a.magic = 0;   // (beam2d_kernel's epoch-tagged slices; this kernel's store is tag-free)
        a.dbg = nullptr;
        a.upd_count = g_b2_upd_counter;
        a.defer_odd = b2_route().defer_odd & 1;
        a.need_mono = 1;
        a.no_cum = 1;
        a.chain_scan = b2_route().chain_scan;
        a.order = nullptr;
        if (n > g.blocks && !b2_route().no_order) {   // more pairs than resident workgroups: longest first
            a.order = (int*)(w + g.off_order);
            hipLaunchKernelGGL(pair_order_kernel, dim3(1), dim3(1024), 0, stream, y1_off, y2_off, n, (int*)(w + g.off_order));
        }
        const int grid = a.persist ? std::min(g.blocks, rp->nslices) : n;
        a.pre_vcols = 0;
        a.ngl = po_reg_ngl(reg_wide(W));
        if (po_zero_async(w + g.off_queue, 256, stream) != hipSuccess) return PO_E_HIP;
        if (model == PO_MODEL_CTC) hipLaunchKernelGGL(beam2d_prepass_kernel<PO_MODEL_CTC>, dim3(n), dim3(64), 0, stream, a);
        else if (model == PO_MODEL_MERGE) hipLaunchKernelGGL(beam2d_prepass_kernel<PO_MODEL_MERGE>, dim3(n), dim3(64), 0, stream, a);
        else hipLaunchKernelGGL(beam2d_prepass_kernel<PO_MODEL_FLIPFLOP>, dim3(n), dim3(64), 0, stream, a);
        hipLaunchKernelGGL(beam2d_walk_kernel, dim3(n), dim3(64), 0, stream, a);
        if (g_b2_mark_fwd) g_b2_mark_fwd(1, stream);
        po_reg_launch(&a, grid, model, reg_wide(W), stream);
        if (g_b2_mark_fwd) g_b2_mark_fwd(0, stream);
        // pairs the pre-pass or the kernel deferred (tier-2 row groups exhausted, windows beyond the store's ring):
        // one small pass of beam2d_kernel, a no-op when there are none
        return b2_launch_legacy(y1, y1_off, y2, y2_off, env, n, C, A, alphabet, W, model, method, mr1, mr2, seq, seq_off,
                                seq_len, status, use_pre_status, w + g.off_fb, g.fb_bytes, stream, X2_FB_BLOCKS, a.meta, 0, a.queue + 16);
    }
    return b2_launch_legacy(y1, y1_off, y2, y2_off, env, n, C, A, alphabet, W, model, method, mr1, mr2, seq, seq_off, seq_len,
                            status, use_pre_status, ws, ws_bytes, stream, 0, nullptr);
}

// max rows are not part of the device-pointer ABI: read them back from the offset arrays
extern "C" int po_launch_beam2d(const double* y1, const int64_t* y1_off, const double* y2, const int64_t* y2_off,
                                const int32_t* env, int n, int C, int A, uint32_t alphabet, int W, int model,
                                int method, char* seq, const int64_t* seq_off, int32_t* seq_len, int32_t* status,
                                void* ws, size_t ws_bytes, hipStream_t stream) {
    if (n <= 0) return PO_OK;
    int64_t* h = (int64_t*)malloc(sizeof(int64_t) * 2 * (size_t)(n + 1));
    if (!h) return PO_E_NOMEM;
    int rc = PO_OK;
    if (hipMemcpyAsync(h, y1_off, sizeof(int64_t) * (n + 1), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipMemcpyAsync(h + n + 1, y2_off, sizeof(int64_t) * (n + 1), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
        rc = PO_E_HIP;
    int64_t m1 = 0, m2 = 0, tr1 = 0, tr2 = 0;
    if (rc == PO_OK) {
        for (int i = 0; i < n; ++i) {
            m1 = std::max<int64_t>(m1, h[i + 1] - h[i]);
            m2 = std::max<int64_t>(m2, h[n + 1 + i + 1] - h[n + 1 + i]);
        }
        tr1 = h[n] - h[0];
        tr2 = h[n + 1 + n] - h[n + 1];
    }
    free(h);
    if (rc != PO_OK) return rc;
    return po_launch_beam2d_geom(y1, y1_off, y2, y2_off, env, n, C, A, alphabet, W, model, method, tr1, tr2, m1, m2, seq,
                                 seq_off, seq_len, status, 0, ws, ws_bytes, stream);
}
