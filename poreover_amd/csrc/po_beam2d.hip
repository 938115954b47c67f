// Batched pair (2-D) CTC beam search restricted to an alignment envelope: method "row_col".
//
// Replaces decoding_cpp.cpp_beam_search_2d (decoding_cpp.pyx:107-139) -> beam_search(...,
// envelope_ranges, ..., method="row_col") (BeamSearch.h:411-427) -> beam_search_2d_by_row_col
// (BeamSearch.h:262-397) over the 2-D prefix trees of PrefixTree.h (:492-533 ctc, :578-633
// flip-flop, :667-706 merge-repeats), with Beam<..., node_greater_max_sym> (Beam.h:30-38,93-108).
//
// What the reference does.  It walks the envelope along a diagonal (u, v).  A MAIN step updates
// every element (the <= W beam nodes and their 4 children) on read 0 over the look-ahead window
// [u, ce) and on read 1 over [v, re), then keeps the W elements with the largest
// max_t alpha0[t] + max_t alpha1[t].  A CATCH-UP step advances only one of u, v and updates only
// the beam nodes at that single time.  Every update reads alpha of the node and of its parent at
// time t-1 from per-node std::unordered_map<int,double>s that are never erased; an absent
// entry reads as -inf.  Values written in one step are read again in later steps — by the same
// node (t = start-1), by beam nodes whose parent has left the beam (the parent's old look-ahead
// values, "frozen"), and by children that become elements again when their parent re-enters the
// beam — so the maps ARE the algorithm's state and have to be reproduced exactly.
//
// Data layout on MI355X.
//   * The maps become a VALUE STORE in HBM (L2-resident in practice): each node owns a ring
//     row of R entries per read, entry = {64-bit tag(epoch, node, t), K doubles}; a read hits iff
//     the tag matches, so absent == -inf falls out with no bookkeeping and rows can be recycled
//     without clearing.  R = pow2 >= widest window + 2 (every time a step can still read lies in
//     [u-1, u-1+R)).  Rows are handed out per parent in groups of 4 (one per child symbol) and
//     recycled as soon as every time written into them lies below u-1 / v-1 (nothing reads it).
//   * The tree shrinks to an arena of packed (parent,last) words for the final label walk, plus
//     first-child / row-group words that are read only when a node (re-)enters the beam.
//   * Within a step the recurrence alpha[t] = lae(alpha_parent[t-1] + y[t][c], alpha[t-1] +
//     y[t][blank]) is a wavefront over (depth, t): every element advances one t per iteration
//     and takes its parent's value of the previous iteration from an LDS exchange buffer
//     (double-buffered, one barrier per iteration).  Parents that are not elements ("frozen")
//     and the root are staged from the store / the blank prefix sums into LDS in chunks of 32
//     iterations, so the dependent chain never waits on HBM.
//   * One workgroup per pair, thread = (read, element slot); workgroups are persistent and pull
//     pairs from an atomic queue, so a launch fills the 256 CUs for any batch size and the
//     per-workgroup store (a few MB) is reused pair after pair.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "po_device.h"

namespace {

constexpr int B2_CH = 32;    // staging chunk: iterations per refill of the frozen-parent buffer
constexpr int B2_NGL = 256;  // row groups tracked per pair (LDS bookkeeping)

template <int K>
struct alignas(K == 1 ? 16 : 32) Entry {
    unsigned long long tag;
    double v[K];
};

__device__ __forceinline__ unsigned long long make_tag(unsigned epoch, int node, int t) {
    return ((unsigned long long)(epoch & 0xffffu) << 48) | ((unsigned long long)(node & 0xffffff) << 24) |
           (unsigned long long)(t & 0xffffff);
}

struct B2Args {
    const double* y1; const int64_t* y1_off;
    const double* y2; const int64_t* y2_off;
    const int32_t* env;
    int n, A, W, C;
    uint32_t alphabet;
    char* seq; const int64_t* seq_off; int32_t* seq_len; int32_t* status;
    int use_pre_status;       // status[] already holds skip / error codes for some pairs: leave those alone
    // workspace (per persistent workgroup unless noted)
    int* queue;               // one counter for the launch
    char* pool; size_t pool_bytes;
    int* arena; long long arena_cap;   // 3 arrays of arena_cap ints: packed(parent,last), first_child, row group
    double* cum; long long tcap;       // 2 arrays of tcap doubles: blank prefix sums of each read
    int* envt; long long vcap;         // 2 * vcap ints: transposed envelope
    long long* dbg;                    // optional phase cycle counters (PO_B2_TIMING builds)
};

__device__ __forceinline__ char* carve(char*& p, size_t bytes) {
    char* r = p;
    p += (bytes + 15) & ~size_t(15);
    return r;
}

// beam table: one entry per beam node, carried from step to step
struct BeamTab {
    int *id, *row, *prow, *par, *gpar, *plast, *last, *depth, *fc, *crow;
};
__device__ __forceinline__ void carve_beam(BeamTab& b, char*& p, int WM) {
    const size_t n = sizeof(int) * WM;  // spelled out: a loop over member pointers would go to scratch
    b.id = (int*)carve(p, n); b.row = (int*)carve(p, n); b.prow = (int*)carve(p, n); b.par = (int*)carve(p, n);
    b.gpar = (int*)carve(p, n); b.plast = (int*)carve(p, n); b.last = (int*)carve(p, n);
    b.depth = (int*)carve(p, n); b.fc = (int*)carve(p, n); b.crow = (int*)carve(p, n);
}

}  // namespace

template <int MODEL>
__global__ __launch_bounds__(256) void beam2d_rowcol_kernel(B2Args a) {
    constexpr int K = (MODEL == PO_MODEL_CTC) ? 1 : 3;
    using Ent = Entry<K>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int NCP = nthr >> 1;         // element slots per read (padded)
    const int r = tid / NCP;           // read handled by this thread
    const int s = tid - r * NCP;       // element slot handled by this thread
    const int A = a.A, W = a.W, C = a.C;
    const int WM = max(W, PO_A);
    const int NCmax = WM * (A + 1);

    // ---- LDS carve
    char* p = smem;
    BeamTab B, Bn;
    carve_beam(B, p, WM);
    carve_beam(Bn, p, WM);
    int* e_id = (int*)carve(p, sizeof(int) * NCmax);
    int* e_row = (int*)carve(p, sizeof(int) * NCmax);
    int* e_pslot = (int*)carve(p, sizeof(int) * NCmax);  // >=0: element slot of the parent; -1: staged
    int* e_sym = (int*)carve(p, sizeof(int) * NCmax);    // own symbol | same-as-parent << 8 | parent-is-root << 9
    int* sel = (int*)carve(p, sizeof(int) * WM);
    int* dup = (int*)carve(p, sizeof(int) * NCmax);
    int* b_stage = (int*)carve(p, sizeof(int) * WM);     // beam slot needs its parent staged (1) / root (2)
    int* g_owner = (int*)carve(p, sizeof(int) * B2_NGL);
    int* g_hi0 = (int*)carve(p, sizeof(int) * B2_NGL);    // one past the latest time written, read 0
    int* g_hi1 = (int*)carve(p, sizeof(int) * B2_NGL);    // ... read 1
    int* sh = (int*)carve(p, sizeof(int) * 16);
    double* score = (double*)carve(p, sizeof(double) * NCmax);
    double* mxs = (double*)carve(p, sizeof(double) * 2 * NCP);
    double* xch = (double*)carve(p, sizeof(double) * 2 * 2 * NCP * K);
    double* stg = (double*)carve(p, sizeof(double) * WM * 2 * B2_CH * K);

    // ---- per-workgroup workspace
    Ent* pool = (Ent*)(a.pool + (size_t)blockIdx.x * a.pool_bytes);
    const long long pool_entries = (long long)(a.pool_bytes / sizeof(Ent));
    int* apl = a.arena + (size_t)blockIdx.x * 3 * a.arena_cap;
    int* afc = apl + a.arena_cap;
    int* acrow = afc + a.arena_cap;
    double* cum0 = a.cum + (size_t)blockIdx.x * 2 * a.tcap;
    double* cum1 = cum0 + a.tcap;
    int* envt = a.envt + (size_t)blockIdx.x * 2 * a.vcap;
    unsigned epoch = 0;
#ifdef PO_B2_TIMING
    long long tk[12] = {0,0,0,0,0,0,0,0,0,0,0,0}, tlast = 0;
#define TK_START() do { tlast = wall_clock64(); } while (0)
#define TK(i) do { const long long n_ = wall_clock64(); tk[i] += n_ - tlast; tlast = n_; } while (0)
#define TKC(i) do { tk[i]++; } while (0)
#else
#define TK_START() do {} while (0)
#define TK(i) do {} while (0)
#define TKC(i) do {} while (0)
#endif

    for (;;) {
        // ---------------------------------------------------------------- next pair from the queue
        __syncthreads();
        if (tid == 0) sh[0] = atomicAdd(a.queue, 1);
        __syncthreads();
        const int pi = sh[0];
        if (pi >= a.n) break;
        epoch++;
        TK_START();
        if (a.use_pre_status && a.status[pi] != PO_OK) {  // skipped upstream (pair_decode.py:372-375,395-398)
            if (tid == 0) a.seq_len[pi] = 0;
            continue;
        }
        const int64_t o1 = a.y1_off[pi], o2 = a.y2_off[pi];
        const int U = (int)(a.y1_off[pi + 1] - o1), V = (int)(a.y2_off[pi + 1] - o2);
        const double* yA = a.y1 + o1 * C;
        const double* yB = a.y2 + o2 * C;
        const double* yr_ = r ? yB : yA;
        const int32_t* env = a.env + 2 * o1;
        const double* cum = r ? cum1 : cum0;
        int st = PO_OK;
        if (U < 1 || V < 1 || U > a.tcap || V > a.vcap || V > a.tcap || U >= (1 << 24) || V >= (1 << 24)) st = PO_E_ARG;

        // ---------------------------------------------------------------- envelope pre-pass
        // bounds, widest row, transposed envelope (BeamSearch.h:270-284), widest column
        int R = 32, NG = 0;
        if (st == PO_OK) {
            int bad = 0, wmax = 0;
            for (int u = tid; u < U; u += nthr) {
                const int lo = env[2 * u], hi = env[2 * u + 1];
                if (lo < hi && (lo < 0 || hi > V)) bad = 1;
                wmax = max(wmax, hi - lo);
            }
            for (int x = tid; x < V; x += nthr) { envt[2 * x] = -1; envt[2 * x + 1] = -1; }
            if (__syncthreads_or(bad)) st = PO_E_ENVELOPE;
            if (st == PO_OK) {
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
                // block max through LDS
                if (tid == 0) sh[1] = 0;
                __syncthreads();
                atomicMax(&sh[1], wmax);
                __syncthreads();
                wmax = sh[1];
                while (R < wmax + 2) R <<= 1;
                const long long ng = pool_entries / ((long long)PO_A * 2 * R);
                NG = (int)min((long long)B2_NGL, ng);
                if (NG < 2 * WM + 4) st = PO_E_NOMEM;  // envelope too wide for the per-pair store
            }
        }
        const long long arena_need = 1 + A + (long long)A * WM * (min(U, V) + 1);
        if (st == PO_OK && (arena_need > a.arena_cap || arena_need >= (1 << 24))) st = PO_E_NOMEM;
        if (st != PO_OK) {
            if (tid == 0) { a.status[pi] = st; a.seq_len[pi] = 0; }
            continue;
        }
        const int Rm = R - 1;
        // blank prefix sums of both reads = the CTC root's alpha (PrefixTree.h:509-515); serial
        // in t so the rounding is the reference's
        if (MODEL == PO_MODEL_CTC && s == 0) {
            double* cw = r ? cum1 : cum0;
            const int Tn = r ? V : U;
            double acc = 0.0;
            for (int t0 = 0; t0 < Tn; t0 += 8) {
                double b[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) b[q] = (t0 + q < Tn) ? yr_[(int64_t)(t0 + q) * C + A] : 0.0;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (t0 + q < Tn) { acc += b[q]; cw[t0 + q] = acc; }
            }
        }
        for (int g = tid; g < B2_NGL; g += nthr) { g_owner[g] = -1; g_hi0[g] = 0; g_hi1[g] = 0; }
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

        // ---------------------------------------------------------------- tree + beam initialisation
        // root = node 0; its A children = nodes 1..A in row group 0 (BeamSearch.h:286-293)
        if (tid == 0) {
            apl[0] = po_pack_node(-1, A); afc[0] = 1; acrow[0] = 0;
            g_owner[0] = 0; g_hi0[0] = 1; g_hi1[0] = 1;  // the root's children hold values at t = 0
            sh[2] = 1 + A;  // next node id
            sh[3] = 1;      // group allocation cursor
            sh[4] = PO_OK;
        }
        if (tid < A) {
            apl[1 + tid] = po_pack_node(0, tid); afc[1 + tid] = -1; acrow[1 + tid] = -1;
            B.id[tid] = 1 + tid; B.row[tid] = tid; B.prow[tid] = -1; B.par[tid] = 0; B.gpar[tid] = -1;
            B.plast[tid] = A; B.last[tid] = tid; B.depth[tid] = 1; B.fc[tid] = -1; B.crow[tid] = -1;
        }
        if (s < A) {  // update_prob(n, r, 0) for both reads
            double sp[3] = {PO_NEG_INF, PO_NEG_INF, PO_NEG_INF}, pp[3], out[3];
            root_at(r, -1, pp);
            const double ya = yr_[s], yb = (MODEL == PO_MODEL_FLIPFLOP) ? yr_[s + A] : yr_[A];
            po_update<MODEL>(sp, pp, ya, yb, false, true, out);
            st_write(s, r, 0, 1 + s, out);
        }
        int nb = A;  // beam size
        int u = 0, v = 0, step = 0;
        __syncthreads();
        TK(0);  // pre-pass + init

        // one scan = every participating element advances over its window, parent values flowing
        // through the LDS exchange buffer.  main: all elements of both reads; catch-up: beam
        // nodes of one read at one time (len == 1).
        // t0x / lenx: window start / length for read x (len 0 = read not touched).
        auto scan = [&](bool is_main, int nelem, int t00, int len0, int t01, int len1) {
            const int t0 = r ? t01 : t00, len = r ? len1 : len0;
            const bool part = (s < nelem) && (len > 0);
            const int Lmax = max(len0, len1);
            int node = 0, row = -1, pslot = -1, sym = 0;
            bool same = false, rootpar = false;
            double self[K], mx = PO_NEG_INF;
            if (part) {
                node = e_id[s]; row = e_row[s]; pslot = e_pslot[s];
                sym = e_sym[s] & 0xff; same = (e_sym[s] >> 8) & 1; rootpar = (e_sym[s] >> 9) & 1;
                st_read(row, r, t0 - 1, node, self);
#pragma unroll
                for (int k = 0; k < K; ++k) xch[((1 * 2 + r) * NCP + s) * K + k] = self[k];
            }
            const int64_t ycol_a = sym, ycol_b = (MODEL == PO_MODEL_FLIPFLOP) ? sym + A : A;
            TK(is_main ? 3 : 7);  // scan: self read
            for (int k0 = 0; k0 < Lmax; k0 += B2_CH) {
                // ---- stage the parents that are not moving in this scan (frozen / root): times
                //      t0-1+k0 .. for B2_CH iterations, both reads, every beam slot that needs it
                const int nbs = min(nelem, nb);
                for (int idx = tid; idx < nbs * 2 * B2_CH; idx += nthr) {
                    const int j = idx / (2 * B2_CH), rem = idx - j * 2 * B2_CH;
                    const int rr = rem / B2_CH, kk = rem - rr * B2_CH;
                    const int lr = rr ? len1 : len0, tr0 = rr ? t01 : t00;
                    const int mode = b_stage[j];
                    if (mode == 0) continue;  // parent moves in this scan: its values come through xch
                    if (k0 + kk >= lr) continue;
                    double out[K];
                    const int tt = tr0 - 1 + k0 + kk;
                    if (mode == 2) root_at(rr, tt, out);
                    else st_read(B.prow[j], rr, tt, B.par[j], out);
#pragma unroll
                    for (int k = 0; k < K; ++k) stg[((j * 2 + rr) * B2_CH + kk) * K + k] = out[k];
                }
                po_lds_barrier();
                TK(is_main ? 4 : 8);  // scan: staging
                const int kend = min(Lmax, k0 + B2_CH);
                for (int k = k0; k < kend; ++k) {
                    const bool act = part && (k < len);
                    double out[K];
                    if (act) {
                        const int t = t0 + k;
                        double pp[K];
                        if (pslot >= 0) {
#pragma unroll
                            for (int q = 0; q < K; ++q) pp[q] = xch[((((k + 1) & 1) * 2 + r) * NCP + pslot) * K + q];
                        } else {
#pragma unroll
                            for (int q = 0; q < K; ++q) pp[q] = stg[((s * 2 + r) * B2_CH + (k - k0)) * K + q];
                        }
                        // y[t][.] does not depend on the chain: the loads issue ahead and hit L1 (the
                        // windows of consecutive steps overlap); an LDS ring for y measured no faster
                        const double ya = yr_[(int64_t)t * C + ycol_a], yb = yr_[(int64_t)t * C + ycol_b];
                        po_update<MODEL>(self, pp, ya, yb, same, rootpar && t == 0, out);
                        st_write(row, r, t, node, out);
#pragma unroll
                        for (int q = 0; q < K; ++q) { self[q] = out[q]; xch[(((k & 1) * 2 + r) * NCP + s) * K + q] = out[q]; }
                        if (out[0] > mx) mx = out[0];
                    }
                    po_lds_barrier();  // only xch crosses iterations; the stores stay in flight
                }
                TK(is_main ? 5 : 9);  // scan: iterations
            }
            if (s < nelem) mxs[r * NCP + s] = mx;
        };

        // ================================================================ the diagonal walk
        while (u <= U - 1 && v <= V - 1) {
            const int ers = env[2 * u], ere = env[2 * u + 1];
            const int ecs = envt[2 * v], ece = envt[2 * v + 1];
            const bool row_ok = (v >= ers && v < ere);
            if (!row_ok && v < ers) {  // catch-up along read 1 (BeamSearch.h:314-322)
                const int nbe = min(W, nb);
                if (tid < nbe) {  // element table = beam nodes only
                    e_id[tid] = B.id[tid]; e_row[tid] = B.row[tid];
                    e_sym[tid] = B.last[tid] | ((B.plast[tid] == B.last[tid]) << 8) | ((B.par[tid] == 0) << 9);
                    int ps = -1, mode = 1;
                    if (B.par[tid] == 0) mode = 2;
                    else
                        for (int i = 0; i < nbe; ++i) if (B.id[i] == B.par[tid]) { ps = i; mode = 0; }
                    e_pslot[tid] = ps; b_stage[tid] = mode;
                    atomicMax(&g_hi1[B.row[tid] / PO_A], v + 1);
                }
                po_lds_barrier();
                __syncthreads();  // store writes of earlier steps -> visible to this scan's reads
                TK(6);  // catch-up: setup
                scan(false, nbe, 0, 0, v, 1);
                po_lds_barrier();
                TKC(11);
                v++;
                continue;
            }
            const bool col_ok = (u >= ecs && u < ece);
            if (!col_ok && u < ecs) {  // catch-up along read 0 (BeamSearch.h:328-336)
                const int nbe = min(W, nb);
                if (tid < nbe) {
                    e_id[tid] = B.id[tid]; e_row[tid] = B.row[tid];
                    e_sym[tid] = B.last[tid] | ((B.plast[tid] == B.last[tid]) << 8) | ((B.par[tid] == 0) << 9);
                    int ps = -1, mode = 1;
                    if (B.par[tid] == 0) mode = 2;
                    else
                        for (int i = 0; i < nbe; ++i) if (B.id[i] == B.par[tid]) { ps = i; mode = 0; }
                    e_pslot[tid] = ps; b_stage[tid] = mode;
                    atomicMax(&g_hi0[B.row[tid] / PO_A], u + 1);
                }
                po_lds_barrier();
                __syncthreads();
                TK(6);
                scan(false, nbe, u, 1, 0, 0);
                po_lds_barrier();
                TKC(11);
                u++;
                continue;
            }
            if (!row_ok || !col_ok) { st = PO_E_ENVELOPE; break; }  // uninitialised bounds upstream (:309)

            // ------------------------------------------------------------ MAIN step at (u, v)
            step++;
            // (1) expansion: fresh ids + a row group for beam nodes that never had children;
            //     nodes that re-entered keep their children (and the group, if it is still theirs)
            if (tid == 0) {
                int next_id = sh[2], cur = sh[3], err = PO_OK;
                for (int j = 0; j < nb; ++j) {
                    bool need_group = false;
                    if (B.fc[j] < 0) {
                        B.fc[j] = next_id;
                        afc[B.id[j]] = next_id;
                        for (int c = 0; c < A; ++c) { apl[next_id + c] = po_pack_node(B.id[j], c); afc[next_id + c] = -1; acrow[next_id + c] = -1; }
                        next_id += A;
                        need_group = true;
                    } else if (B.crow[j] < 0 || g_owner[B.crow[j]] != B.id[j]) {
                        need_group = true;  // its old rows were recycled: every value in them was dead
                    }
                    if (need_group) {
                        int g = -1;
                        for (int tries = 0; tries < NG; ++tries) {
                            const int c = cur;
                            cur = (cur + 1 == NG) ? 0 : cur + 1;
                            // every stored time is < hi; a step only reads times >= u-1 / v-1
                            if (g_owner[c] < 0 || (g_hi0[c] <= u - 1 && g_hi1[c] <= v - 1)) { g = c; break; }
                        }
                        if (g < 0) { err = PO_E_NOMEM; g = 0; }
                        g_owner[g] = B.id[j];
                        B.crow[j] = g;
                        acrow[B.id[j]] = g;
                    }
                    // this step writes [u, ece) x [v, ere) into the children's rows and the node's own
                    const int gc = B.crow[j], go = B.row[j] / PO_A;
                    g_hi0[gc] = max(g_hi0[gc], ece); g_hi1[gc] = max(g_hi1[gc], ere);
                    g_hi0[go] = max(g_hi0[go], ece); g_hi1[go] = max(g_hi1[go], ere);
                }
                sh[2] = next_id; sh[3] = cur;
                if (err != PO_OK) sh[4] = err;
            }
            po_lds_barrier();
            if (sh[4] != PO_OK) { st = sh[4]; break; }
            // (2) element table: beam slots [0, nb), child c of beam node j at nb + A*j + c
            const int NCc = nb * (A + 1);
            if (tid < NCc) {
                if (tid < nb) {
                    const int j = tid;
                    e_id[j] = B.id[j]; e_row[j] = B.row[j];
                    e_sym[j] = B.last[j] | ((B.plast[j] == B.last[j]) << 8) | ((B.par[j] == 0) << 9);
                    int ps = -1, mode = 1;
                    if (B.par[j] == 0) mode = 2;
                    else {
                        for (int i = 0; i < nb; ++i) if (B.id[i] == B.par[j]) { ps = i; mode = 0; }
                        if (ps < 0)
                            for (int i = 0; i < nb; ++i) if (B.id[i] == B.gpar[j]) { ps = nb + A * i + B.plast[j]; mode = 0; }
                    }
                    e_pslot[j] = ps; b_stage[j] = mode;
                } else {
                    const int j = (tid - nb) / A, c = (tid - nb) - j * A;
                    e_id[tid] = B.fc[j] + c; e_row[tid] = B.crow[j] * PO_A + c;
                    e_sym[tid] = c | ((B.last[j] == c) << 8);
                    e_pslot[tid] = j;
                }
            }
            po_lds_barrier();
            __syncthreads();  // arena + store writes -> visible to the reads below
            TK(2);  // main: expansion + element table
            // (3) the two look-ahead windows (BeamSearch.h:342-375)
            scan(true, NCc, u, ece - u, v, ere - v);
            po_lds_barrier();
            // (4) prune by max0 + max1 (node_greater_max_sym); a child that is also a beam node is
            //     the same node pushed twice (std::unique)
            if (tid < NCc) {
                score[tid] = mxs[tid] + mxs[NCP + tid];
                int d = 0;
                if (tid >= nb) {
                    const int x = e_id[tid];
                    for (int j = 0; j < nb; ++j) d |= (e_id[j] == x);
                }
                dup[tid] = d;
            }
            if (tid == 0) sh[5] = 0;
            po_lds_barrier();
            if (tid < NCc && !dup[tid]) {
                const double sc = score[tid];
                const int id = e_id[tid];
                int rank = 0;
                for (int o = 0; o < NCc; ++o)
                    if (!dup[o] && po_better(score[o], e_id[o], sc, id)) rank++;
                if (rank < W) sel[rank] = tid;
                atomicAdd(&sh[5], 1);
            }
            po_lds_barrier();
            const int nbn = min(W, sh[5]);
            // (5) next beam table; a promoted child learns from the arena whether it was ever expanded
            if (tid < nbn) {
                const int e = sel[tid];
                if (e < nb) {
                    Bn.id[tid] = B.id[e]; Bn.row[tid] = B.row[e]; Bn.prow[tid] = B.prow[e]; Bn.par[tid] = B.par[e];
                    Bn.gpar[tid] = B.gpar[e]; Bn.plast[tid] = B.plast[e]; Bn.last[tid] = B.last[e];
                    Bn.depth[tid] = B.depth[e]; Bn.fc[tid] = B.fc[e]; Bn.crow[tid] = B.crow[e];
                } else {
                    const int j = (e - nb) / A, c = (e - nb) - j * A;
                    const int id = B.fc[j] + c;
                    Bn.id[tid] = id; Bn.row[tid] = B.crow[j] * PO_A + c; Bn.prow[tid] = B.row[j]; Bn.par[tid] = B.id[j];
                    Bn.gpar[tid] = B.par[j]; Bn.plast[tid] = B.last[j]; Bn.last[tid] = c; Bn.depth[tid] = B.depth[j] + 1;
                    Bn.fc[tid] = afc[id]; Bn.crow[tid] = acrow[id];
                }
            }
            po_lds_barrier();
            if (tid < nbn) {
                B.id[tid] = Bn.id[tid]; B.row[tid] = Bn.row[tid]; B.prow[tid] = Bn.prow[tid]; B.par[tid] = Bn.par[tid];
                B.gpar[tid] = Bn.gpar[tid]; B.plast[tid] = Bn.plast[tid]; B.last[tid] = Bn.last[tid];
                B.depth[tid] = Bn.depth[tid]; B.fc[tid] = Bn.fc[tid]; B.crow[tid] = Bn.crow[tid];
            }
            nb = nbn;
            po_lds_barrier();
            TK(1);  // main: prune + next beam
            u++;
            v++;
        }

        // ---------------------------------------------------------------- label of the top node
        __syncthreads();
        if (tid == 0) {
            int nout = 0;
            if (st == PO_OK) {
                int node = B.id[0];
                nout = B.depth[0];
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
        TK(10);  // label walk
    }
#ifdef PO_B2_TIMING
    if (tid == 0 && a.dbg && blockIdx.x == 0)
        for (int i = 0; i < 12; ++i) a.dbg[i] = tk[i];
#endif
}

// ------------------------------------------------------------------------------------------------
// host side: geometry, workspace layout, launch
namespace {
struct B2Geom {
    int threads, blocks;
    size_t lds, pool_bytes, arena_cap, tcap, vcap;
    size_t off_queue, off_pool, off_arena, off_cum, off_envt, total;
};
inline size_t al256(size_t b) { return (b + 255) & ~size_t(255); }

int b2_num_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

B2Geom b2_geometry(int n, int64_t mr1, int64_t mr2, int W, int model) {
    B2Geom g;
    const int K = (model == PO_MODEL_CTC) ? 1 : 3;
    const int WM = W > PO_A ? W : PO_A, NC = WM * (PO_A + 1);
    int ncp = 32;
    while (ncp < NC) ncp <<= 1;
    g.threads = 2 * ncp;
    const int waves = g.threads / PO_WAVE;
    const int per_cu = waves <= 1 ? 12 : (waves == 2 ? 6 : 3);  // 3 waves per SIMD (<= 168 VGPRs)
    g.blocks = b2_num_cus() * per_cu;
    if (g.blocks > n) g.blocks = n > 0 ? n : 1;
    auto al = [](size_t b) { return (b + 15) & ~size_t(15); };
    g.lds = 20 * al(sizeof(int) * WM) + 4 * al(sizeof(int) * NC) + al(sizeof(int) * WM) + al(sizeof(int) * NC) +
            al(sizeof(int) * WM) + 3 * al(sizeof(int) * B2_NGL) + al(sizeof(int) * 16) + al(sizeof(double) * NC) +
            al(sizeof(double) * 2 * ncp) + al(sizeof(double) * 4 * ncp * K) + al(sizeof(double) * WM * 2 * B2_CH * K);
    g.pool_bytes = (K == 1 ? (size_t)4 : (size_t)8) << 20;  // value store per workgroup
    const int64_t mn = mr1 < mr2 ? mr1 : mr2;
    g.arena_cap = (size_t)(1 + PO_A + (int64_t)PO_A * WM * (mn + 1));
    g.tcap = (size_t)(mr1 > mr2 ? mr1 : mr2);
    g.vcap = (size_t)mr2;
    size_t o = 0;
    g.off_queue = o; o += 256;
    g.off_pool = o; o += al256(g.pool_bytes) * g.blocks;
    g.pool_bytes = al256(g.pool_bytes);
    g.off_arena = o; o += al256(sizeof(int) * 3 * g.arena_cap * g.blocks);
    g.off_cum = o; o += al256(sizeof(double) * 2 * g.tcap * g.blocks);
    g.off_envt = o; o += al256(sizeof(int) * 2 * g.vcap * g.blocks);
    g.total = o + 256;
    return g;
}
}  // namespace

extern "C" size_t po_beam2d_ws_bytes_impl(int n, int64_t tr1, int64_t tr2, int64_t mr1, int64_t mr2, int C, int W,
                                          int model, int method) {
    (void)tr1; (void)tr2; (void)C; (void)method;
    return b2_geometry(n, mr1, mr2, W, model).total;
}

extern "C" int po_launch_beam2d_geom(const double* y1, const int64_t* y1_off, const double* y2, const int64_t* y2_off,
                                     const int32_t* env, int n, int C, int A, uint32_t alphabet, int W, int model,
                                     int method, int64_t mr1, int64_t mr2, char* seq, const int64_t* seq_off,
                                     int32_t* seq_len, int32_t* status, int use_pre_status, void* ws,
                                     size_t ws_bytes, hipStream_t stream) {
    if (n <= 0) return PO_OK;
    if (A < 1 || A > PO_A || W < 1 || W > 25) return PO_E_ARG;
    if (method != PO_METHOD_ROW_COL) return PO_E_UNSUPPORTED;
    if (!env) return PO_E_UNSUPPORTED;
    if ((model == PO_MODEL_FLIPFLOP) ? (C != 2 * A) : (C != A + 1)) return PO_E_ARG;
    const B2Geom g = b2_geometry(n, mr1, mr2, W, model);
    if (ws_bytes < g.total) return PO_E_CAP;
    char* w = (char*)ws;
    B2Args a;
    a.y1 = y1; a.y1_off = y1_off; a.y2 = y2; a.y2_off = y2_off; a.env = env;
    a.n = n; a.A = A; a.W = W; a.C = C; a.alphabet = alphabet;
    a.seq = seq; a.seq_off = seq_off; a.seq_len = seq_len; a.status = status;
    a.use_pre_status = use_pre_status;
    a.queue = (int*)(w + g.off_queue);
    a.pool = w + g.off_pool; a.pool_bytes = g.pool_bytes;
    a.arena = (int*)(w + g.off_arena); a.arena_cap = (long long)g.arena_cap;
    a.cum = (double*)(w + g.off_cum); a.tcap = (long long)g.tcap;
    a.envt = (int*)(w + g.off_envt); a.vcap = (long long)g.vcap;
    a.dbg = nullptr;
#ifdef PO_B2_TIMING
    static long long* dbg_buf = nullptr;
    if (!dbg_buf) (void)hipMalloc((void**)&dbg_buf, 12 * sizeof(long long));
    a.dbg = dbg_buf;
#endif
    // queue counter and the store's tags start from zero on every launch
    if (hipMemsetAsync(w + g.off_queue, 0, 256, stream) != hipSuccess) return PO_E_HIP;
    if (hipMemsetAsync(w + g.off_pool, 0, g.pool_bytes * g.blocks, stream) != hipSuccess) return PO_E_HIP;
#define PO_LAUNCH_B2(M)                                                                                        \
    do {                                                                                                       \
        if (g.lds > 64 * 1024)                                                                                 \
            (void)hipFuncSetAttribute((const void*)beam2d_rowcol_kernel<M>,                                    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);                 \
        hipLaunchKernelGGL(beam2d_rowcol_kernel<M>, dim3(g.blocks), dim3(g.threads), g.lds, stream, a);        \
    } while (0)
    if (model == PO_MODEL_CTC) PO_LAUNCH_B2(PO_MODEL_CTC);
    else if (model == PO_MODEL_MERGE) PO_LAUNCH_B2(PO_MODEL_MERGE);
    else if (model == PO_MODEL_FLIPFLOP) PO_LAUNCH_B2(PO_MODEL_FLIPFLOP);
    else return PO_E_ARG;
#undef PO_LAUNCH_B2
#ifdef PO_B2_TIMING
    {
        long long h[12];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h, a.dbg, sizeof(h), hipMemcpyDeviceToHost);
        const char* nm[12] = {"prepass+init", "main:prune+nextbeam", "main:expand+table", "main:scan selfread", "main:scan staging",
                              "main:scan iterations", "catchup:setup", "catchup:selfread", "catchup:staging", "catchup:iterations",
                              "label walk", "#catchup steps"};
        fprintf(stderr, "[po_b2_timing] block 0, wall_clock64 ticks (100 MHz => 10 ns each):\n");
        for (int i = 0; i < 12; ++i) fprintf(stderr, "   %-24s %12lld\n", nm[i], h[i]);
    }
#endif
    return PO_OK;
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
    int64_t m1 = 0, m2 = 0;
    if (rc == PO_OK)
        for (int i = 0; i < n; ++i) {
            m1 = std::max<int64_t>(m1, h[i + 1] - h[i]);
            m2 = std::max<int64_t>(m2, h[n + 1 + i + 1] - h[n + 1 + i]);
        }
    free(h);
    if (rc != PO_OK) return rc;
    return po_launch_beam2d_geom(y1, y1_off, y2, y2_off, env, n, C, A, alphabet, W, model, method, m1, m2, seq, seq_off,
                                 seq_len, status, 0, ws, ws_bytes, stream);
}

