// Batched 1-D CTC beam search over a prefix tree.
//
// Replaces decoding_cpp.cpp_beam_search (decoding_cpp.pyx:88-103) -> beam_search
// (BeamSearch.h:400-408) -> beam_search_<Tree,Beam> (BeamSearch.h:18-58) with the three tree
// recurrences of PrefixTree.h: PoreOver 'ctc' (:478-488), Bonito 'ctc_merge_repeats' (:649-663),
// flip-flop 'ctc_flipflop' (:548-574).
//
// What the reference does per time step t: every beam node and every child of a beam node is
// updated from values at time t-1 ONLY (own alpha[t-1] and the parent's alpha[t-1]); a value
// that was never stored reads as -inf (PrefixTree.h:55-61).  alpha[t-1] of a node exists iff
// the node was in the candidate set (beam + children of beam nodes) of step t-1, or is the
// root.  So the exact live state is the previous step's candidate table (<= 5W nodes with
// their values), not the whole tree.  That table lives in LDS, double-buffered; the tree
// itself is reduced to an HBM arena of packed (parent, last) words — written once per node,
// read only for the final label walk — plus a first_child word per node that is read only
// when a node re-enters the beam after having dropped out of the candidate table.
//
// Mapping: ONE WAVE PER READ (the step-to-step dependency is serial; parallelism comes from
// the <= 5W candidates of a step and from thousands of reads in flight).  Lane s owns
// candidate slot s (slots [0,Wc) = beam nodes in rank order, slot Wc + 4*j + c = child c of
// beam node j), strided when 5W > 64.  Prune = dedupe by node id + rank by (score desc, id asc)
// + keep the top W (Beam.h:93-108; tie rule: see DESIGN.md).
// (po_device.h's logaddexp variants, A/B in round 4 on 1 000 reads: CTC W = 10 4.82 -> 4.68 ms, flip-flop W = 10 7.17 -> 7.10 ms,
//  W = 25 17.4 -> 17.3 / 23.1 -> 23.4 ms; same result bits)
#define PO_LAE_EARLY_TABLE 1
#define PO_LAE_BRANCHLESS 1
#define PO_LAE_TRIM 1
#include "po_device.h"

namespace {

template <int MODEL>
struct ModelTraits {
    static constexpr int K = (MODEL == PO_MODEL_CTC) ? 1 : 3;
    static constexpr int CMAX = (MODEL == PO_MODEL_FLIPFLOP) ? 2 * PO_A : PO_A + 1;
};

// a candidate as the ranking reads it (score, node id, duplicate flag): one ds_read_b128 instead of three reads
struct alignas(16) B1Cand { double sc; int id; int dup; };

// one candidate table (struct of arrays carved from dynamic LDS)
struct Table {
    int* id;      // [NC]  node id
    int* fc;      // [NC]  first child id; -1 = never expanded; -2 = unknown (ask the arena)
    int* depth;   // [NC]
    double* val;  // [K][NC] values at the table's time step (channel 0 = total)
    int* par;     // [WM]  beam slots only: parent id
    int* gpar;    // [WM]  grand-parent id (-1 if none)
    int* plast;   // [WM]  parent's last symbol (A for the root)
    int* last;    // [WM]  own last symbol
};

__device__ __forceinline__ char* carve(char*& p, size_t bytes) {
    char* r = p;
    p += (bytes + 15) & ~size_t(15);
    return r;
}

}  // namespace

#ifndef PO_B1_WAVES
#define PO_B1_WAVES 4   // waves per SIMD beam1d_wave_kernel is compiled for (5: 10 000 reads + 5 %, 1 000 reads - 1 .. 2.5 %: profiles/r06_ab_compiler_flags.txt)
#endif

// One row of y (N doubles at a wave-uniform address) through the SCALAR data cache: y is read-only input, the row
// of the next frame used to be requested a frame ahead by vector loads — and waiting for those (vmcnt counts in
// order) also waited for every tree-node store of the frame.  Scalar loads count on lgkmcnt.
template <int N>
__device__ __forceinline__ void b1_sload_row(const double* p, double* out) {
    const unsigned long long pv = (unsigned long long)p;
    const unsigned long long ps = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(pv >> 32)) << 32) |
                                  (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pv);
    unsigned long long v0, v1, v2, v3, v4, v5 = 0, v6 = 0, v7 = 0;
    if constexpr (N == 5) {
        asm volatile("s_load_dwordx2 %0, %5, 0x0\n\ts_load_dwordx2 %1, %5, 0x8\n\ts_load_dwordx2 %2, %5, 0x10\n\t"
                     "s_load_dwordx2 %3, %5, 0x18\n\ts_load_dwordx2 %4, %5, 0x20\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(v0), "=&s"(v1), "=&s"(v2), "=&s"(v3), "=&s"(v4) : "s"(ps) : "memory");
    } else {
        static_assert(N == 8, "row width");
        asm volatile("s_load_dwordx2 %0, %8, 0x0\n\ts_load_dwordx2 %1, %8, 0x8\n\ts_load_dwordx2 %2, %8, 0x10\n\t"
                     "s_load_dwordx2 %3, %8, 0x18\n\ts_load_dwordx2 %4, %8, 0x20\n\ts_load_dwordx2 %5, %8, 0x28\n\t"
                     "s_load_dwordx2 %6, %8, 0x30\n\ts_load_dwordx2 %7, %8, 0x38\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(v0), "=&s"(v1), "=&s"(v2), "=&s"(v3), "=&s"(v4), "=&s"(v5), "=&s"(v6), "=&s"(v7) : "s"(ps) : "memory");
    }
    const unsigned long long v[8] = {v0, v1, v2, v3, v4, v5, v6, v7};
#pragma unroll
    for (int c = 0; c < N; ++c) out[c] = __longlong_as_double((long long)v[c]);
}

// yr[i] for a per-lane index without a runtime-indexed private array (which would live in scratch memory): the row
// sits in scalar registers, the lane picks its entry with a compare-select chain
__device__ __forceinline__ double rg1_readlane_d(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
template <int N>
__device__ __forceinline__ double b1_pick(const double* yr, int i) {
    // (each entry passes through an empty asm first: a select between two LOADS of the row is otherwise folded into one
    // load from a selected ADDRESS, which pins the row in scratch memory and puts a scratch round trip — on gfx9 a drain of
    // the wave's store queue as well, vmcnt counts both — on every pick)
    double r = yr[0];
    asm("" : "+v"(r));
#pragma unroll
    for (int c = 1; c < N; ++c) {
        double v = yr[c];
        asm("" : "+v"(v));
        r = (i == c) ? v : r;
    }
    return r;
}

// minimum of x over the wave, as a wave-uniform value (row_shr steps, then the rows' results into lane 63)
__device__ __forceinline__ double b1_wave_min(double x) {
#define B1_STEPW(ctrl, rmask)                                                                                                  \
    x = fmin(x, __hiloint2double(__builtin_amdgcn_update_dpp(__double2hiint(x), __double2hiint(x), ctrl, rmask, 0xf, false),   \
                                 __builtin_amdgcn_update_dpp(__double2loint(x), __double2loint(x), ctrl, rmask, 0xf, false)))
    B1_STEPW(0x111, 0xf); B1_STEPW(0x112, 0xf); B1_STEPW(0x114, 0xf); B1_STEPW(0x118, 0xf);
    B1_STEPW(0x142, 0xa);   // row_bcast:15 -> rows 1, 3
    B1_STEPW(0x143, 0xc);   // row_bcast:31 -> rows 2, 3
#undef B1_STEPW
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
}

template <int MODEL>
__global__ __launch_bounds__(PO_WAVE) void beam1d_kernel(
    const double* __restrict__ y, const int64_t* __restrict__ y_off, int A, uint32_t alphabet, int W,
    int* __restrict__ arena_pl, int* __restrict__ arena_fc, char* __restrict__ seq,
    const int64_t* __restrict__ seq_off, int32_t* __restrict__ seq_len, int32_t* __restrict__ status) {
    constexpr int K = ModelTraits<MODEL>::K, CMAX = ModelTraits<MODEL>::CMAX;
    const int C = (MODEL == PO_MODEL_FLIPFLOP) ? 2 * A : A + 1;  // A = |alphabet| <= PO_A
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = po_lane();
    // specialised logaddexp (po_device.h): its tables sit at the front of the dynamic LDS block
    PoLaeTables* lae_t = (PoLaeTables*)smem;
    po_lae_tables_load(lae_t, lane, PO_WAVE);
    __syncthreads();
    const PoLaeFast lae{lae_t};
    const int r = blockIdx.x;
    const int64_t r0 = y_off[r];
    const int T = (int)(y_off[r + 1] - r0);
    const double* yr0 = y + r0 * C;
    const int WM = max(W, PO_A), NC = WM * (PO_A + 1);  // layout uses the maximum alphabet size
    // node arena of read r: closed-form offset, (1 + PO_A) + PO_A*WM*T entries (po_beam1d_arena_nodes)
    const int64_t aoff = (int64_t)r * (1 + PO_A) + (int64_t)PO_A * WM * (r0 - y_off[0]);
    const int64_t acap = (1 + PO_A) + (int64_t)PO_A * WM * T;
    int* apl = arena_pl + aoff;
    int* afc = arena_fc + aoff;

    // two candidate tables; selected by value each step (a runtime-indexed array of structs
    // would live in scratch memory)
    Table T0, T1;
    char* p = smem + ((sizeof(PoLaeTables) + 15) & ~size_t(15));
    for (int b = 0; b < 2; ++b) {
        Table& tb = b ? T1 : T0;
        tb.id = (int*)carve(p, sizeof(int) * NC);
        tb.fc = (int*)carve(p, sizeof(int) * NC);
        tb.depth = (int*)carve(p, sizeof(int) * NC);
        tb.val = (double*)carve(p, sizeof(double) * K * NC);
        tb.par = (int*)carve(p, sizeof(int) * WM);
        tb.gpar = (int*)carve(p, sizeof(int) * WM);
        tb.plast = (int*)carve(p, sizeof(int) * WM);
        tb.last = (int*)carve(p, sizeof(int) * WM);
    }
    int* sel = (int*)carve(p, sizeof(int) * WM);     // table-P slots forming the current beam
    int* isnew = (int*)carve(p, sizeof(int) * WM);   // beam node expanded in this very step
    int* dup = (int*)carve(p, sizeof(int) * NC);
    int* nsel = (int*)carve(p, sizeof(int) * WM);
    int* ord = (int*)carve(p, sizeof(int) * NC);     // prune with exact score ties: candidate slots in node-id order
    B1Cand* cand = (B1Cand*)carve(p, sizeof(B1Cand) * NC);   // the candidates packed for the ranking: one 16-byte read each
    int* stl_stk = (int*)carve(p, sizeof(int) * 48);         // the explicit stack of the exact-tie replay (po_stl_sort)
    int* kps = (int*)carve(p, sizeof(int) * NC);             // steady table: a slot's parent slot (-1 root, -2 none) ...
    int* kss = (int*)carve(p, sizeof(int) * NC);             // ... and its symbol | (same symbol as the parent) << 8

    if (T < 1) {
        if (lane == 0) { seq_len[r] = 0; status[r] = PO_E_ARG; }
        return;
    }
    if (acap < 1 + A) {
        if (lane == 0) { seq_len[r] = 0; status[r] = PO_E_NOMEM; }
        return;
    }

    int cur = 0;  // cur == 0: P (previous step) = T0, Q (being built) = T1; swapped each step
    int Pnb = 0;  // number of beam slots in P
    int next_id = 1 + A;
    int st = PO_OK;
    double blank_cum = 0.0;

    // ---- t = 0: the A children of the root (BeamSearch.h:25-30), no prune
    {
        const Table P = T0;
        double yr[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) yr[c] = (c < C) ? yr0[c] : 0.0;
        if (lane == 0) { apl[0] = po_pack_node(-1, A); afc[0] = 1; }
        if (lane < A) {
            double sp[3] = {PO_NEG_INF, PO_NEG_INF, PO_NEG_INF}, pp[3], out[3];
            root_values<MODEL>(-1, 0.0, pp);
            po_update<MODEL>(sp, pp, b1_pick<CMAX>(yr, lane), (MODEL == PO_MODEL_FLIPFLOP) ? b1_pick<CMAX>(yr, lane + A) : b1_pick<CMAX>(yr, A), false, true, out, lae);
            P.id[lane] = 1 + lane;
            P.fc[lane] = -1;
            P.depth[lane] = 1;
#pragma unroll
            for (int k = 0; k < K; ++k) P.val[k * NC + lane] = out[k];
            apl[1 + lane] = po_pack_node(0, lane);
            afc[1 + lane] = -1;
            sel[lane] = lane;
        }
        if (MODEL == PO_MODEL_CTC) blank_cum = b1_pick<CMAX>(yr, A);
    }
    int Wc = A;  // beam size entering step 1 (all root children; the first prune is at t = 1)
    __syncthreads();

    // STEADY TABLE (as in beam1d_wave_kernel): a frame that finds the beam exactly as the previous frame left it
    // rebuilds nothing — every slot updates its value in place (all reads, a fence, all
    // writes) from its own previous value and its parent's (kps / kss: left by the last frame that built a table), and the
    // prune is two comparisons per slot; anything but "strictly the same beam" runs the ranking on the table as it stands.
    bool stable = false;
    for (int t = 1; t < T; ++t) {
        const Table P = cur ? T1 : T0;
        const bool first = (t == 1);
        double yr[CMAX];
        auto load_y = [&](int tt) {
            if (C == CMAX) {   // (the standard alphabets: whole rows of CMAX doubles)
                b1_sload_row<CMAX>(yr0 + (int64_t)tt * C, yr);
            } else {
                // (other alphabets: per-lane loads, waited for HERE — a value still in flight at the join with the scalar-load
                // path would put an s_waitcnt vmcnt(0) on that path too, where it drains the frame's arena stores)
#pragma unroll
                for (int c = 0; c < CMAX; ++c) yr[c] = (c < C) ? yr0[(int64_t)tt * C + c] : 0.0;
#pragma unroll
                for (int c = 0; c < CMAX; ++c) asm volatile("" : "+v"(yr[c]));
            }
        };
        load_y(t);
        const int NCc0 = Wc * (A + 1);
        const bool ident_in = !first && Wc == Pnb && Wc <= PO_WAVE && (__ballot(lane < Wc && sel[min(lane, WM - 1)] != lane) == 0ull);
        const bool inplace = stable && ident_in && Wc == W && NCc0 <= 2 * PO_WAVE;
        if (inplace) {
            // A RUN of such frames (as in beam1d_wave_kernel): a lane's two slots keep their constants and their own values in
            // registers; a frame reads the parents' values from the table, updates, tests the beam on registers (the beam
            // slots are lanes 0 .. W-1 of the first half: next slot by wave_shl:1, last one by v_readlane) and only then
            // writes.  The first frame that does not keep the beam leaves the run with the table untouched and is redone by
            // the in-place frame below.
            int ps_[2], ks_[2];
            bool in_[2], dp_[2];
            double v_[2][3];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int sl = lane + h * PO_WAVE;
                in_[h] = sl < NCc0;
                const int sx = in_[h] ? sl : 0;
                ps_[h] = kps[sx]; ks_[h] = kss[sx]; dp_[h] = dup[sx] != 0;
#pragma unroll
                for (int k = 0; k < K; ++k) v_[h][k] = P.val[k * NC + sx];
            }
            for (;;) {
                if (t == T - 1) break;   // (the last frame is ranked: the label is the best node's)
                double o2[2][3];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    double pp[3];
#pragma unroll
                    for (int k = 0; k < K; ++k) pp[k] = P.val[k * NC + max(ps_[h], 0)];
                    if (ps_[h] == -1) root_values<MODEL>(t - 1, blank_cum, pp);
                    else if (ps_[h] < 0) {
#pragma unroll
                        for (int k = 0; k < K; ++k) pp[k] = PO_NEG_INF;
                    }
                    const int sym = ks_[h] & 0xff;
#pragma unroll
                    for (int k = 0; k < K; ++k) o2[h][k] = PO_NEG_INF;
                    if (in_[h])
                        po_update<MODEL>(v_[h], pp, b1_pick<CMAX>(yr, sym), (MODEL == PO_MODEL_FLIPFLOP) ? b1_pick<CMAX>(yr, sym + A) : b1_pick<CMAX>(yr, A),
                                         (ks_[h] >> 8) != 0, false, o2[h], lae);
                }
                const double sc0 = o2[0][0], sc1 = o2[1][0];
                // the same SET (see beam1d_wave_kernel): every child strictly below the smallest beam score
                const double scmin = b1_wave_min((lane < Wc) ? sc0 : HUGE_VAL);
                bool viol = false;
                if (lane >= Wc && in_[0] && !dp_[0]) viol = !(scmin > sc0);
                if (in_[1] && !dp_[1]) viol |= !(scmin > sc1);
                if (__ballot(viol) != 0ull) break;
                po_wave_sync();   // every read of the old values before the first write
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (in_[h]) {
#pragma unroll
                        for (int k = 0; k < K; ++k) { P.val[k * NC + lane + h * PO_WAVE] = o2[h][k]; v_[h][k] = o2[h][k]; }
                    }
                }
                po_wave_sync();
                if (MODEL == PO_MODEL_CTC) blank_cum += b1_pick<CMAX>(yr, A);
                ++t;
                if (t >= T) break;
                load_y(t);
            }
            if (t >= T) break;
            double o2[2][3];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int sl = lane + h * PO_WAVE;
                if (sl < NCc0) {
                    const int ps = kps[sl], ks = kss[sl];
                    const int sym = ks & 0xff;
                    double sp[3], pp[3];
#pragma unroll
                    for (int k = 0; k < K; ++k) sp[k] = P.val[k * NC + sl];
                    if (ps >= 0) { for (int k = 0; k < K; ++k) pp[k] = P.val[k * NC + ps]; }
                    else if (ps == -1) root_values<MODEL>(t - 1, blank_cum, pp);
                    else { for (int k = 0; k < K; ++k) pp[k] = PO_NEG_INF; }
                    po_update<MODEL>(sp, pp, b1_pick<CMAX>(yr, sym), (MODEL == PO_MODEL_FLIPFLOP) ? b1_pick<CMAX>(yr, sym + A) : b1_pick<CMAX>(yr, A),
                                     (ks >> 8) != 0, false, o2[h], lae);
                }
            }
            po_wave_sync();   // every read of the old values before the first write
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int sl = lane + h * PO_WAVE;
                if (sl < NCc0) {
#pragma unroll
                    for (int k = 0; k < K; ++k) P.val[k * NC + sl] = o2[h][k];
                }
            }
            po_wave_sync();
            bool viol = false;
            const double scl = P.val[Wc - 1];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int sl = lane + h * PO_WAVE;
                if (sl < NCc0) {
                    const double sc = o2[h][0];
                    if (sl < Wc) { if (sl + 1 < Wc) viol |= !(sc > P.val[sl + 1]); }
                    else if (!dup[sl]) viol |= !(scl > sc);
                }
            }
            if (t != T - 1 && __ballot(viol) == 0ull) {   // the same beam, strictly: nothing else moves
                if (MODEL == PO_MODEL_CTC) blank_cum += b1_pick<CMAX>(yr, A);
                continue;
            }
        }
        const Table Q = inplace ? P : (cur ? T0 : T1);
        const int NCc = NCc0;
        if (!inplace) {
        // ---- phase 1: beam slots
        bool need = false;
        for (int j = lane; j < Wc; j += PO_WAVE) {
            const int s = sel[j];
            int par, gpar, plast, last, pslot = -2;
            if (first) {
                par = 0; gpar = -1; plast = A; last = s; pslot = -1;
            } else if (s >= Pnb) {  // promoted from a child slot: its parent was beam slot b
                const int b = (s - Pnb) / A;
                par = P.id[b]; gpar = P.par[b]; plast = P.last[b]; last = (s - Pnb) % A; pslot = b;
            } else {
                par = P.par[s]; gpar = P.gpar[s]; plast = P.plast[s]; last = P.last[s];
                if (par == 0) pslot = -1;
                else {
#pragma unroll 4
                    for (int i = 0; i < Pnb; ++i) if (P.id[i] == par) pslot = i;
                    if (pslot < 0) {
#pragma unroll 4
                        for (int i = 0; i < Pnb; ++i) if (P.id[i] == gpar) pslot = Pnb + A * i + plast;
                    }
                }
            }
            double sp[3], pp[3], out[3];
#pragma unroll
            for (int k = 0; k < K; ++k) sp[k] = P.val[k * NC + s];
            if (pslot >= 0) { for (int k = 0; k < K; ++k) pp[k] = P.val[k * NC + pslot]; }
            else if (pslot == -1) root_values<MODEL>(t - 1, blank_cum, pp);
            else { for (int k = 0; k < K; ++k) pp[k] = PO_NEG_INF; }
            po_update<MODEL>(sp, pp, b1_pick<CMAX>(yr, last), (MODEL == PO_MODEL_FLIPFLOP) ? b1_pick<CMAX>(yr, last + A) : b1_pick<CMAX>(yr, A), plast == last, false, out, lae);
            int fc = P.fc[s];
            if (__ballot(fc == -2) != 0ull) {   // re-entered the beam: the arena remembers (rare; its load is waited for inside
                if (fc == -2) fc = afc[P.id[s]];   // the branch — at the join the wait would drain the frame's arena stores)
                po_settle(fc);
            }
            Q.id[j] = P.id[s]; Q.depth[j] = P.depth[s];
            Q.par[j] = par; Q.gpar[j] = gpar; Q.plast[j] = plast; Q.last[j] = last;
#pragma unroll
            for (int k = 0; k < K; ++k) Q.val[k * NC + j] = out[k];
            Q.fc[j] = fc;
            need = (fc == -1);
            kss[j] = last | ((plast == last) ? 0x100 : 0);
        }
        // ---- expansion: A fresh ids per beam node that has never had children
        //      (PrefixTree::expand, PrefixTree.h:439-446), in beam order
        {
            int base = next_id;
            for (int j0 = 0; j0 < Wc; j0 += PO_WAVE) {
                const int j = j0 + lane;
                const bool nd = (j < Wc) && need;  // `need` belongs to this lane's j in this chunk
                const unsigned long long m = __ballot(nd);
                if (nd) {
                    const int fc = base + A * __popcll(m & ((1ull << lane) - 1ull));
                    if ((int64_t)fc + A <= acap) {
                        Q.fc[j] = fc;
                        afc[Q.id[j]] = fc;
                        for (int c = 0; c < A; ++c) { apl[fc + c] = po_pack_node(Q.id[j], c); afc[fc + c] = -1; }
                    } else {
                        Q.fc[j] = 0;  // points at the root: harmless, the read is flagged below
                    }
                }
                if (j < Wc) isnew[j] = nd ? 1 : 0;
                base += A * __popcll(m);
            }
            if ((int64_t)base > acap) st = PO_E_NOMEM;
            next_id = base;
        }
        po_wave_sync();   // (one wave per read)
        if (st != PO_OK) break;

        // ---- phase 2: the A children of every beam node
        for (int s = Wc + lane; s < NCc; s += PO_WAVE) {
            const int j = (s - Wc) / A, c = (s - Wc) % A;
            const int x = Q.fc[j] + c, sj = sel[j];
            int slot = -2, fcx = isnew[j] ? -1 : -2;
            if (!isnew[j] && !first) {
#pragma unroll 4
                for (int i = 0; i < Pnb; ++i) if (P.id[i] == x) slot = i;
                if (slot < 0 && sj < Pnb) slot = Pnb + A * sj + c;
            }
            double sp[3], pp[3], out[3];
            if (slot >= 0) { fcx = P.fc[slot]; for (int k = 0; k < K; ++k) sp[k] = P.val[k * NC + slot]; }
            else { for (int k = 0; k < K; ++k) sp[k] = PO_NEG_INF; }
#pragma unroll
            for (int k = 0; k < K; ++k) pp[k] = P.val[k * NC + sj];
            po_update<MODEL>(sp, pp, b1_pick<CMAX>(yr, c), (MODEL == PO_MODEL_FLIPFLOP) ? b1_pick<CMAX>(yr, c + A) : b1_pick<CMAX>(yr, A), Q.last[j] == c, false, out, lae);
            Q.id[s] = x; Q.fc[s] = fcx; Q.depth[s] = Q.depth[j] + 1;
#pragma unroll
            for (int k = 0; k < K; ++k) Q.val[k * NC + s] = out[k];
            kps[s] = j; kss[s] = c | ((Q.last[j] == c) ? 0x100 : 0);   // (what the next frame needs if it finds this beam unchanged:
        }                                                              //  the parent's slot in THIS table)
        po_wave_sync();   // (one wave per read)
        for (int j = lane; j < Wc; j += PO_WAVE) {   // ... for a beam node: its parent among this table's beam slots, or a child of its grand-parent's
            const int par = Q.par[j], gpar = Q.gpar[j], plast = Q.plast[j];
            int ps = (par == 0) ? -1 : -2;
            if (par != 0) {
#pragma unroll 4
                for (int i = 0; i < Wc; ++i) if (Q.id[i] == par) ps = i;
                if (ps < 0) {
#pragma unroll 4
                    for (int i = 0; i < Wc; ++i) if (Q.id[i] == gpar) ps = Wc + A * i + plast;
                }
            }
            kps[j] = ps;
        }
        }   // !inplace

        // ---- phase 3: prune (Beam.h:93-108).  A child slot whose node is also a beam slot is the
        //      same node pushed twice: std::unique removes it.
        for (int s = lane; s < NCc; s += PO_WAVE) {
            int d = 0;
            if (s >= Wc) {
                const int x = Q.id[s];
#pragma unroll 4
                for (int j = 0; j < Wc; ++j) d |= (Q.id[j] == x);
            }
            dup[s] = d;
            B1Cand c;
            c.sc = Q.val[s]; c.id = Q.id[s]; c.dup = d;
            cand[s] = c;
        }
        po_wave_sync();   // (one wave per read)
        // the smallest FAMILY maximum (a beam node and its non-duplicate children: W distinct candidates reach it — see
        // beam1d_wave_kernel): the threshold of the two-stage selects below
        auto family_threshold = [&]() -> double {
            double fam = HUGE_VAL;
            if (lane < Wc) {
                fam = cand[lane].sc;
                for (int cc = 0; cc < A; ++cc) {
                    const int cs = Wc + A * lane + cc;
                    if (cs < NCc && !dup[cs]) fam = fmax(fam, cand[cs].sc);
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) fam = fmin(fam, __shfl_xor(fam, off));
            return fam;
        };
        int kept = 0;
        bool tie = false;
        // Two-stage select (tables of at most 64 candidates: W <= 12).  When the beam is full, its W own continuations
        // are W candidates at or above the smallest of their scores (thr): a child below thr cannot be among the W best,
        // and nothing below thr outranks anything at or above it — so the ranks are taken within {beam slots} +
        // {children >= thr}: about W + a few candidates instead of W * (A + 1), each one broadcast LDS read.  (Exact
        // ties reaching into the beam go the full way below.)
        if (NCc <= PO_WAVE) {
            const int s = lane;
            const bool valid = (s < NCc) && !dup[s];
            const B1Cand me = cand[min(s, NCc - 1)];
            double thr = PO_NEG_INF;
            if (Wc == W) thr = family_threshold();
            const bool inS = valid && me.sc >= thr;
            const unsigned long long sm_ = __ballot(inS);
            kept = __popcll(__ballot(valid));
            int rank = 0, neq = 0;
            for (unsigned long long mm = sm_; mm != 0ull; mm &= mm - 1ull) {   // (uniform)
                const B1Cand c = cand[__builtin_ctzll(mm)];
                rank += ((c.sc > me.sc) | (!(me.sc > c.sc) & (c.id < me.id))) ? 1 : 0;
                neq += (c.sc == me.sc) ? 1 : 0;
            }
            if (inS && rank < W) nsel[rank] = s;
            tie = __ballot(inS && (neq > 1) && (rank < W)) != 0ull;   // an exact tie that reaches into the beam
        } else if (NCc <= 2 * PO_WAVE && Wc == W && Wc <= PO_WAVE) {
            // the same two-stage select for tables of up to 128 candidates (W <= 25): a lane holds two slots, the set
            // {beam slots} + {children >= the smallest beam score} is two ballot masks, the ranks are taken within it
            const double thr = family_threshold();   // (Wc <= 64 here: one beam slot per lane)
            B1Cand me[2];
            bool val2[2], in2[2];
            unsigned long long smk[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int s = lane + h * PO_WAVE;
                val2[h] = (s < NCc) && !dup[min(s, NCc - 1)];
                me[h] = cand[min(s, NCc - 1)];
                in2[h] = val2[h] && me[h].sc >= thr;
                smk[h] = __ballot(in2[h]);
                kept += __popcll(__ballot(val2[h]));
            }
            int rank[2] = {0, 0}, neq[2] = {0, 0};
#pragma unroll
            for (int g = 0; g < 2; ++g)
                for (unsigned long long mm = smk[g]; mm != 0ull; mm &= mm - 1ull) {   // (uniform)
                    const B1Cand c = cand[g * PO_WAVE + __builtin_ctzll(mm)];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        rank[h] += ((c.sc > me[h].sc) | (!(me[h].sc > c.sc) & (c.id < me[h].id))) ? 1 : 0;
                        neq[h] += (c.sc == me[h].sc) ? 1 : 0;
                    }
                }
            bool teq = false;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (in2[h] && rank[h] < W) nsel[rank[h]] = lane + h * PO_WAVE;
                teq |= in2[h] && (neq[h] > 1) && (rank[h] < W);
            }
            tie = __ballot(teq) != 0ull;
        } else
        for (int s0 = 0; s0 < NCc; s0 += PO_WAVE) {
            const int s = s0 + lane;
            const bool valid = (s < NCc) && !dup[s];
            bool teq = false;
            if (valid) {
                const double sc = Q.val[s];
                const int id = Q.id[s];
                int rank = 0, neq = 0;
                // (branch-free, every load unconditional: the compiler batches the LDS reads of eight candidates
                //  instead of two dependent round trips per candidate)
#pragma unroll 8
                for (int o = 0; o < NCc; ++o) {
                    const B1Cand c = cand[o];
                    const int live = c.dup ? 0 : 1;
                    const int better = ((c.sc > sc) | (!(sc > c.sc) & (c.id < id))) ? 1 : 0;
                    rank += live & better;
                    neq += live & ((c.sc == sc) ? 1 : 0);
                }
                if (rank < W) nsel[rank] = s;
                teq = (neq > 1) && (rank < W);   // an exact tie that reaches into the beam
            }
            tie |= (__ballot(teq) != 0ull);
            kept += __popcll(__ballot(valid));
        }
        po_wave_sync();   // (one wave per read)
        if (tie) {
            // exact ties decide by what libstdc++'s partial_sort / sort leave (po_device.h): candidates in node-id
            // order (each live candidate's position = live candidates with a smaller id), then one lane replays it
            for (int s = lane; s < NCc; s += PO_WAVE) {
                if (!dup[s]) {
                    const int id = Q.id[s];
                    int pos = 0;
                    for (int o = 0; o < NCc; ++o) pos += ((dup[o] ? 0 : 1) & ((Q.id[o] < id) ? 1 : 0));
                    ord[pos] = s;
                }
            }
            po_wave_sync();   // (one wave per read)
            if (lane == 0) {
                const double* val0 = Q.val;
                po_stl_prune<64>(ord, kept, W, [&](int slot) { return val0[slot]; }, stl_stk);
                for (int j = 0; j < min(W, kept); ++j) nsel[j] = ord[j];
            }
            po_wave_sync();   // (one wave per read)
        }
        const int Wn = min(W, kept);
        for (int j = lane; j < Wn; j += PO_WAVE) sel[j] = nsel[j];
        if (MODEL == PO_MODEL_CTC) blank_cum += b1_pick<CMAX>(yr, A);
        if (!inplace) { cur ^= 1; stable = true; }   // (kps / kss describe the table just built; an in-place frame leaves everything as it is)
        Pnb = Wc;
        Wc = Wn;
        po_wave_sync();   // (one wave per read)
    }

    __syncthreads();
    // ---- label of the top node (PrefixTree::get_label, PrefixTree.h:449-457)
    if (lane == 0) {
        int n = 0;
        if (st == PO_OK) {
            const Table P = cur ? T1 : T0;
            int node = P.id[sel[0]];
            n = P.depth[sel[0]];
            char* out = seq + seq_off[r];
            const int cap = (int)(seq_off[r + 1] - seq_off[r]);
            if (n > cap) { st = PO_E_CAP; n = 0; }
            else
                for (int i = n - 1; i >= 0; --i) {
                    const int pk = apl[node];
                    out[i] = (char)((alphabet >> (8 * (po_node_last(pk) & 3))) & 0xffu);
                    node = po_node_parent(pk);
                }
        }
        seq_len[r] = n;
        status[r] = st;
    }
}

// ------------------------------------------------------------------------------------------------
// The same search with the candidate table in REGISTERS: W * (PO_A + 1) <= 64, i.e. W <= 12 (BASELINE configs 2 and 5:
// W = 10).  One wave per read, lane s = candidate slot s of the table; what a lane needs from another slot — the source
// slot of a beam node, its parent's and its own previous values, the beam's node ids — comes by v_readlane / ds_bpermute
// instead of LDS arrays and their search loops, and the two update passes of a frame (beam slots, then children: both
// read the PREVIOUS table only) are one pass with one logaddexp latency.  Semantics, arena layout and tie handling are
// beam1d_kernel's, statement for statement; only the exact-tie replay still goes through LDS.
// minimum of x over lanes 0 .. 15 (row 0 of the wave), as a wave-uniform value: four row_shr steps, lane 15 holds it
__device__ __forceinline__ double b1_row0_min(double x) {
#define B1_STEP(ctrl)                                                                                                     \
    x = fmin(x, __hiloint2double(__builtin_amdgcn_update_dpp(__double2hiint(x), __double2hiint(x), ctrl, 0xf, 0xf, false), \
                                 __builtin_amdgcn_update_dpp(__double2loint(x), __double2loint(x), ctrl, 0xf, 0xf, false)))
    B1_STEP(0x111); B1_STEP(0x112); B1_STEP(0x114); B1_STEP(0x118);
#undef B1_STEP
    return rg1_readlane_d(x, 15);
}

template <int MODEL>
__global__ __launch_bounds__(PO_WAVE, PO_B1_WAVES) void beam1d_wave_kernel(
    const double* __restrict__ y, const int64_t* __restrict__ y_off, int A, uint32_t alphabet, int W,
    int* __restrict__ arena_pl, int* __restrict__ arena_fc, char* __restrict__ seq,
    const int64_t* __restrict__ seq_off, int32_t* __restrict__ seq_len, int32_t* __restrict__ status) {
    constexpr int K = ModelTraits<MODEL>::K, CMAX = ModelTraits<MODEL>::CMAX;
    const int C = (MODEL == PO_MODEL_FLIPFLOP) ? 2 * A : A + 1;
    __shared__ PoLaeTables lae_tab;
    __shared__ int ord[64];
    __shared__ double tsc[64];
    __shared__ int stl_stk[48];
    // y rows, 32 frames at a time, double-buffered: the next block is requested a block ahead (four doubles per lane
    // held in registers for 32 frames) and written when the walk enters it, so no frame waits for memory — the row of a
    // frame is two broadcast-free LDS reads per lane, issued together with the table permutes
    __shared__ double yblk[2][32][CMAX];
    const int lane = po_lane();
    po_lae_tables_load(&lae_tab, lane, PO_WAVE);
    __syncthreads();
    const PoLaeFast lae{&lae_tab};
    const int r = blockIdx.x;
    const int64_t r0 = y_off[r];
    const int T = (int)(y_off[r + 1] - r0);
    const double* yr0 = y + r0 * C;
    const int WM = max(W, PO_A);
    const int64_t aoff = (int64_t)r * (1 + PO_A) + (int64_t)PO_A * WM * (r0 - y_off[0]);
    const int64_t acap = (1 + PO_A) + (int64_t)PO_A * WM * T;
    int* apl = arena_pl + aoff;
    int* afc = arena_fc + aoff;
    if (T < 1) { if (lane == 0) { seq_len[r] = 0; status[r] = PO_E_ARG; } return; }
    if (acap < 1 + A) { if (lane == 0) { seq_len[r] = 0; status[r] = PO_E_NOMEM; } return; }

    // ---- the previous step's table, one slot per lane
    int p_id = 0, p_fc = -1, p_depth = 0, p_par = 0, p_gpar = -1, p_plast = A, p_last = 0;
    double p_val[K];
#pragma unroll
    for (int k = 0; k < K; ++k) p_val[k] = PO_NEG_INF;
    int selv = lane;      // lane j < Wc: the previous table's slot of beam node j
    int Pnb = 0, Wc = A, next_id = 1 + A, st = PO_OK;
    double blank_cum = 0.0;
    {   // t = 0: the A children of the root (BeamSearch.h:25-30), no prune
        double yr[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) yr[c] = (c < C) ? yr0[c] : 0.0;
        if (lane == 0) { apl[0] = po_pack_node(-1, A); afc[0] = 1; }
        if (lane < A) {
            double sp[3] = {PO_NEG_INF, PO_NEG_INF, PO_NEG_INF}, pp[3], out[3];
            root_values<MODEL>(-1, 0.0, pp);
            po_update<MODEL>(sp, pp, b1_pick<CMAX>(yr, lane), (MODEL == PO_MODEL_FLIPFLOP) ? b1_pick<CMAX>(yr, lane + A) : b1_pick<CMAX>(yr, A), false, true, out, lae);
            p_id = 1 + lane; p_fc = -1; p_depth = 1;
#pragma unroll
            for (int k = 0; k < K; ++k) p_val[k] = out[k];
            apl[1 + lane] = po_pack_node(0, lane);
            afc[1 + lane] = -1;
        }
        if (MODEL == PO_MODEL_CTC) blank_cum = b1_pick<CMAX>(yr, A);
    }
    auto shf = [&](int v, int src) { return __shfl(v, src); };
    const int divA = (65536 + A - 1) / A;
    constexpr int YPL = (32 * CMAX + PO_WAVE - 1) / PO_WAVE;   // doubles of a block per lane
    const int divC = (65536 + C - 1) / C;
    double ynx[YPL];
    auto y_request = [&](int blk) {   // rows [32 blk, 32 blk + 32) -> registers
#pragma unroll
        for (int q = 0; q < YPL; ++q) {
            const int i = lane + q * PO_WAVE;
            const int64_t row = (int64_t)blk * 32 + ((i * divC) >> 16);
            ynx[q] = (i < 32 * C && row < T) ? yr0[(int64_t)blk * 32 * C + i] : 0.0;
        }
    };
    auto y_commit = [&](int blk) {    // registers -> LDS buffer of block blk
#pragma unroll
        for (int q = 0; q < YPL; ++q) {
            const int i = lane + q * PO_WAVE;
            if (i < 32 * C) { const int rw = (i * divC) >> 16; yblk[blk & 1][rw][i - rw * C] = ynx[q]; }
        }
    };
    y_request(0); y_commit(0);
    y_request(1);
    po_wave_sync();

    // STEADY TABLE.  Most frames keep the beam exactly as it was (the same nodes in the same order: nine frames in ten are
    // blank).  The table of such a frame is the previous one, slot for slot — every lane's node, parent slot, symbol and
    // duplicate flag are what they were — so nothing of it is rebuilt: a lane reads its own previous value and its
    // parent's, updates, and the prune is one comparison per child: strictly below the smallest beam score.  The ORDER of
    // the beam nodes among themselves is not looked at: nothing is created while the set stays (every beam node has its
    // children), ties are decided on node ids, so the order only matters when the set changes — and that frame ranks
    // everybody — and for the label, which is why the last frame is always ranked.  (A child AT the smallest beam score,
    // ties included, takes the full ranking below.)  `stable`: a table has been
    // built and the constants it left (k_*) describe its own slots.
    bool stable = false;
    int k_pslot = -2, k_sym = 0;
    bool k_samef = false, k_dup = false;
#ifdef PO_B1_COUNT
    int cnt_fast = 0, cnt_same = 0, cnt_gen = 0;
    long long tk_run = 0, tk_gen = 0, tk_last = wall_clock64(), tk_a = 0, tk_b = 0, tk_c = 0, tk_d = 0, tk_f = 0; int cnt_ff = 0, cnt_refetch = 0, cnt_tie = 0, cnt_S = 0, cnt_rankf = 0;
#define B1_KT(x) do { const long long n_ = wall_clock64(); (x) += n_ - tk_last; tk_last = n_; } while (0)
#else
#define B1_KT(x) do {} while (0)
#endif
    for (int t = 1; t < T; ++t) {
        const bool first = (t == 1);
        if ((t & 31) == 0) { y_commit(t >> 5); y_request((t >> 5) + 1); po_wave_sync(); }
        const double* yrow = &yblk[(t >> 5) & 1][t & 31][0];
        const int NCc = Wc * (A + 1);
        const bool rb = lane < Wc, rc = !rb && lane < NCc;
        const int j = rc ? (((lane - Wc) * divA) >> 16) : 0, c = rc ? (lane - Wc) - j * A : 0;   // (x / A for x < 64)
        const bool ident_in = !first && Wc == Pnb && (__ballot(rb && selv != lane) == 0ull);
        const bool fastf = stable && ident_in;
        double o_keep[3] = {PO_NEG_INF, PO_NEG_INF, PO_NEG_INF};
        bool have_o = false;   // (wave-uniform)
        if (fastf && Wc == W) {
            // A RUN of frames on the steady table.  A lone wave pays for every instruction of a frame, and the general frame
            // below spends ~300 on a steady one; here everything a lane needs is hoisted out of the run — a frame is the
            // parent's value (one permute), the update and two comparisons.  The first frame that does not keep the beam
            // strictly as it is leaves the run and goes through the regular path below, which takes the update the run computed (o_keep).
            const bool act = rb || rc;
            const int par_lane = rb ? max(k_pslot, 0) : j;
            const bool rootp = rb && k_pslot == -1, nop = rb && k_pslot < -1;
            const int iya = act ? k_sym : 0, iyb = (MODEL == PO_MODEL_FLIPFLOP) ? (act ? k_sym + A : 0) : A;
            const bool validc = rc && !k_dup;
            if constexpr (K == 1) {
            // (round 6; the one-value model — with three logaddexp per update the thrown-away frames cost more than the overlap
            //  gains: config 5 5.41 -> 5.78 ms, config 2 3.68 -> 3.59 ms) A lone wave's frame is latency: the update's ~ 40 dependent f64 operations and two table reads, THEN the
            // prune test's four dependent DPP steps, a lane read, a ballot and a branch.  The two do not depend on each other
            // once the frame's update is there — so the loop computes frame t + 1's update (as if frame t keeps the beam) in
            // the same basic block as frame t's prune test, without a branch between them (every lane computes; root / no-parent
            // values are selects), and the scheduler fills the update's latency slots with the test.  If frame t changes the
            // beam, frame t + 1's update is thrown away.  Same arithmetic, same results.
            auto upd_frame = [&](int tf, const double* pv, double bc, double* o_out, double& yb_out) {
                const double* yq = &yblk[(tf >> 5) & 1][tf & 31][0];
                const double ya = yq[iya], yb = yq[iyb];   // (requested before the permute: one LDS round trip for both)
                double pp[3], rv[3] = {PO_NEG_INF, PO_NEG_INF, PO_NEG_INF}, ov[3] = {PO_NEG_INF, PO_NEG_INF, PO_NEG_INF};
#pragma unroll
                for (int k = 0; k < K; ++k) pp[k] = __shfl(pv[k], par_lane);
                root_values<MODEL>(tf - 1, bc, rv);
#pragma unroll
                for (int k = 0; k < K; ++k) pp[k] = rootp ? rv[k] : (nop ? PO_NEG_INF : pp[k]);
                po_update<MODEL>(pv, pp, ya, yb, k_samef, false, ov, lae);
#pragma unroll
                for (int k = 0; k < K; ++k) o_out[k] = act ? ov[k] : PO_NEG_INF;
                yb_out = yb;
            };
            if (t < T - 1) {   // (the last frame is ranked: the label is the best node's)
                double o_cur[3] = {PO_NEG_INF, PO_NEG_INF, PO_NEG_INF}, yb_cur = 0.0;
                upd_frame(t, p_val, blank_cum, o_cur, yb_cur);
                for (;;) {
                    double o_nx[3] = {PO_NEG_INF, PO_NEG_INF, PO_NEG_INF}, yb_nx = 0.0;
                    const double bc_nx = blank_cum + yb_cur;   // (iyb == A for the ctc model: the blank column)
                    upd_frame(t + 1, o_cur, bc_nx, o_nx, yb_nx);   // (t + 1 <= T - 1; a frame that opens a y block reads stale rows: redone below)
                    const double sc = o_cur[0];
                    const double scmin = b1_row0_min(rb ? sc : HUGE_VAL);   // (W <= 12: the beam lanes sit in row 0)
                    if (__ballot(validc && !(scmin > sc)) != 0ull) {   // (the frame is ranked below: its update is this one)
#pragma unroll
                        for (int k = 0; k < K; ++k) o_keep[k] = o_cur[k];
                        have_o = true;
                        break;
                    }
#pragma unroll
                    for (int k = 0; k < K; ++k) p_val[k] = o_cur[k];
                    if (MODEL == PO_MODEL_CTC) blank_cum = bc_nx;
#ifdef PO_B1_COUNT
                    ++cnt_fast; ++cnt_same;
#endif
                    ++t;
                    // (the y block of frame t is committed BEFORE the run can be left: the last frame, ranked by the general path,
                    //  may be the one that opens a block — T - 1 a multiple of 32: the round's 1-D fuzz session, 61 of 51 806 reads)
                    const bool opens = (t & 31) == 0;
                    if (opens) { y_commit(t >> 5); y_request((t >> 5) + 1); po_wave_sync(); }
                    if (t == T - 1) break;
                    if (opens) {
                        upd_frame(t, p_val, blank_cum, o_cur, yb_cur);
                    } else {
#pragma unroll
                        for (int k = 0; k < K; ++k) o_cur[k] = o_nx[k];
                        yb_cur = yb_nx;
                    }
                }
            }
            } else {
            for (;;) {
                if (t == T - 1) break;   // (the last frame is ranked: the label is the best node's)
                const double* yq = &yblk[(t >> 5) & 1][t & 31][0];
                const double ya = yq[iya], yb = yq[iyb];   // (requested before the permute: one LDS round trip for both)
                double pp[3], o[3] = {PO_NEG_INF, PO_NEG_INF, PO_NEG_INF};
#pragma unroll
                for (int k = 0; k < K; ++k) pp[k] = __shfl(p_val[k], par_lane);
                if (rootp) root_values<MODEL>(t - 1, blank_cum, pp);
                else if (nop) {
#pragma unroll
                    for (int k = 0; k < K; ++k) pp[k] = PO_NEG_INF;
                }
                if (act) po_update<MODEL>(p_val, pp, ya, yb, k_samef, false, o, lae);
                const double sc = o[0];
                const double scmin = b1_row0_min(rb ? sc : HUGE_VAL);   // (W <= 12: the beam lanes sit in row 0)
                if (__ballot(validc && !(scmin > sc)) != 0ull) {   // (the frame is ranked below: its update is this one)
#pragma unroll
                    for (int k = 0; k < K; ++k) o_keep[k] = o[k];
                    have_o = true;
                    break;
                }
#pragma unroll
                for (int k = 0; k < K; ++k) p_val[k] = o[k];
                if (MODEL == PO_MODEL_CTC) blank_cum += yb;   // (iyb == A for this model: the blank column)
#ifdef PO_B1_COUNT
                ++cnt_fast; ++cnt_same;
#endif
                ++t;
                if (t >= T) break;
                if ((t & 31) == 0) { y_commit(t >> 5); y_request((t >> 5) + 1); po_wave_sync(); }
            }
            }
            B1_KT(tk_run);
            if (t >= T) break;
            yrow = &yblk[(t >> 5) & 1][t & 31][0];
        }
#ifdef PO_B1_COUNT   // debugging builds: how many frames of read 0 take which path
        if (fastf) ++cnt_fast;
        ++cnt_gen;
#endif
        int q_id, q_depth, q_fc, par, gpar, plast, last, pslot, slot = -2, s_self, s_parent, sym;
        bool samef, expanded = false;
        if (fastf) {
            q_id = p_id; q_depth = p_depth; q_fc = p_fc;
            par = p_par; gpar = p_gpar; plast = p_plast; last = p_last;
            pslot = k_pslot; slot = lane;
            s_self = lane; s_parent = rb ? max(k_pslot, 0) : j;
            sym = k_sym; samef = k_samef;
        } else {
        // ---- beam slots: fields from the previous table's slot selv
        const int src = rb ? selv : 0;
        q_id = shf(p_id, src); q_depth = shf(p_depth, src); q_fc = shf(p_fc, src);
        par = 0; gpar = -1; plast = A; last = 0; pslot = -2;
        int sl0 = -2;
        const int q_fc_pre = q_fc;
        const int xpre = shf(q_fc_pre, j) + c;   // a child's node id if its parent has children already
        {
            const int s_par = shf(p_par, src), s_gpar = shf(p_gpar, src), s_plast = shf(p_plast, src), s_last = shf(p_last, src);
            const int b = (src >= Pnb) ? (((src - Pnb) * divA) >> 16) : 0;
            const int b_id = shf(p_id, b), b_par = shf(p_par, b), b_last = shf(p_last, b);
            if (first) { par = 0; gpar = -1; plast = A; last = src; pslot = -1; }
            else if (src >= Pnb) { par = b_id; gpar = b_par; plast = b_last; last = (src - Pnb) - b * A; pslot = b; }
            else {
                par = s_par; gpar = s_gpar; plast = s_plast; last = s_last;
                if (par == 0) pslot = -1;
            }
            if (!first) {   // the parent's slot in the previous table: a beam slot, or a child of its grand-parent's
                // (the same pass over the previous beam's ids finds, for a child lane, its node's previous beam slot: its
                //  id is known before the expansion whenever its parent is not new)
                // (this kernel's beam has at most 12 slots: the passes over the beam's ids are unrolled over constant lanes — a
                //  lone wave paid ~ 60 cycles per trip of the v_readlane loops these were, and there are five of them per
                //  beam change: 40 % of the kernel at W = 10)
                int ps1 = -2, ps2 = -2;
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    const int bid = __builtin_amdgcn_readlane(p_id, i);
                    const bool in = i < Pnb;
                    if (in && bid == par) ps1 = i;
                    if (in && bid == gpar) ps2 = Pnb + A * i + plast;
                    if (in && bid == xpre) sl0 = i;
                }
                if (src < Pnb && par != 0) pslot = (ps1 >= 0) ? ps1 : ps2;
            }
        }
        const bool refetch = rb && q_fc == -2;
#ifdef PO_B1_COUNT
        if (__ballot(refetch) != 0ull) ++cnt_refetch;
#endif
        if (__ballot(refetch) != 0ull) {   // re-entered the beam: the arena remembers (waited for inside the rare branch: at the
            if (refetch) q_fc = afc[q_id];   // join the wait would drain the frame's arena stores, vmcnt counting both)
            po_settle(q_fc);
        }
        // ---- expansion: A fresh ids per beam node that has never had children, in beam order
        const bool need = rb && (q_fc == -1);
        {
            const unsigned long long m = __ballot(need);
            if (need) {
                const int fc = next_id + A * __popcll(m & ((1ull << lane) - 1ull));
                if ((int64_t)fc + A <= acap) {
                    q_fc = fc;
                    afc[q_id] = fc;
                    for (int cc = 0; cc < A; ++cc) { apl[fc + cc] = po_pack_node(q_id, cc); afc[fc + cc] = -1; }
                } else q_fc = 0;
            }
            next_id += A * __popcll(m);
            if ((int64_t)next_id > acap) st = PO_E_NOMEM;
            expanded = (m != 0ull) || (__ballot(refetch) != 0ull);
        }
        if (st != PO_OK) break;
        // ---- children: id, previous slot, fields of the parent beam node j
        const int pj_fc = shf(q_fc, j), pj_new = shf((int)need, j), pj_sel = shf(selv, j), pj_depth = shf(q_depth, j), pj_last = shf(last, j);
        int fcx = -2;
        if (rc) {
            q_id = pj_fc + c; q_depth = pj_depth + 1; fcx = pj_new ? -1 : -2;
        }
        if (!first) {
            int sl = sl0;
            if (__ballot(rb && q_fc_pre == -2) != 0ull) {   // (a parent's first child came from the arena: look again)
                sl = -2;
                for (int i = 0; i < Pnb; ++i) {
                    const int bid = __builtin_amdgcn_readlane(p_id, i);
                    if (bid == q_id) sl = i;
                }
            }
            if (rc && !pj_new) {
                slot = sl;
                if (slot < 0 && pj_sel < Pnb) slot = Pnb + A * pj_sel + c;
            }
        }
        {   // (every lane takes part in the permute: a lane that sits out cannot be read from)
            const int sfc = shf(p_fc, max(slot, 0));
            if (rc) { if (slot >= 0) fcx = sfc; q_fc = fcx; }
        }
        s_self = rb ? src : max(slot, 0); s_parent = rb ? max(pslot, 0) : pj_sel;
        sym = rb ? last : c;
        samef = rb ? (plast == last) : (pj_last == c);
        }
#ifdef PO_B1_COUNT
        { const long long n_ = wall_clock64(); if (fastf) { tk_f += n_ - tk_last; ++cnt_ff; } else tk_a += n_ - tk_last; tk_gen += n_ - tk_last; tk_last = n_; }
#endif
        // ---- one update for every slot of the new table
        double sp[3], pp[3], out[3] = {PO_NEG_INF, PO_NEG_INF, PO_NEG_INF};
#pragma unroll
        for (int k = 0; k < K; ++k) { sp[k] = fastf ? p_val[k] : __shfl(p_val[k], s_self); pp[k] = __shfl(p_val[k], s_parent); }
        if (rb) {
            if (pslot == -1) root_values<MODEL>(t - 1, blank_cum, pp);
            else if (pslot < 0) {
#pragma unroll
                for (int k = 0; k < K; ++k) pp[k] = PO_NEG_INF;
            }
        } else if (slot < 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) sp[k] = PO_NEG_INF;
        }
        const double ya = yrow[(rb || rc) ? sym : 0], yb = yrow[(MODEL == PO_MODEL_FLIPFLOP) ? ((rb || rc) ? sym + A : 0) : A];
        if (have_o) {
#pragma unroll
            for (int k = 0; k < K; ++k) out[k] = o_keep[k];
        } else if (rb || rc)
            po_update<MODEL>(sp, pp, ya, yb, samef, false, out, lae);
#ifdef PO_B1_COUNT
        { const long long n_ = wall_clock64(); tk_b += n_ - tk_last; tk_gen += n_ - tk_last; tk_last = n_; }
#endif
        // ---- prune (Beam.h:93-108): a child slot whose node is also a beam slot is the same node pushed twice
        bool dupf = fastf ? k_dup : false;
        const double sc = out[0];
        if (!fastf) {   // (the duplicate test: one pass over the beam lanes' ids)
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int bid = __builtin_amdgcn_readlane(q_id, i);
                if (i < Wc && rc && bid == q_id) dupf = true;
            }
        }
        const bool valid = (rb || rc) && !dupf;
        // the smallest beam score: the beam lanes sit in row 0 of the wave (W <= 12) — four DPP steps
        const double scmin = b1_row0_min(rb ? sc : HUGE_VAL);
        double thr = scmin;
        // the beam as it was?  (strictly: exact ties go through the ranking, as partial_sort decides them)
        bool same_beam = false;
        if (Wc == W && t != T - 1) same_beam = (__ballot(rc && valid && !(scmin > sc)) == 0ull);
        int Wn = Wc, nsel = lane;   // lane jx < Wn: the slot of the candidate of rank jx
#ifdef PO_B1_COUNT
        if (same_beam) ++cnt_same;
#endif
        if (!same_beam) {
        // Only candidates at or above a score that W DISTINCT candidates reach can be among the W best, and their ranks among
        // themselves are their ranks.  The W beam continuations give such a score (the smallest of them) — but in the frames
        // that change the beam most children are above the weakest beam node: 35 of 50 candidates went through the ranking
        // loop on average.  The best member of every FAMILY (a beam node and its A children) gives a higher one: W distinct
        // candidates again, and the loop sees the 11 - 14 that matter.
        if (Wc == W) {
            const double vs = valid ? sc : PO_NEG_INF;
            double fam = rb ? sc : PO_NEG_INF;
            for (int cc = 0; cc < A; ++cc) {   // (wave-uniform; every lane takes part in the permute)
                const int cl = min(Wc + A * min(lane, Wc - 1) + cc, PO_WAVE - 1);
                const double v = __shfl(vs, cl);
                if (rb) fam = fmax(fam, v);
            }
            thr = b1_row0_min(rb ? fam : HUGE_VAL);
        } else thr = PO_NEG_INF;
        const bool inS = valid && sc >= thr;
        const unsigned long long smk = __ballot(inS);
        const int kept = __popcll(__ballot(valid));
#ifdef PO_B1_COUNT
        cnt_S += __popcll(smk); ++cnt_rankf;
#endif
        int rank = 0, neq = 0;
        auto versus = [&](int o, bool on) {   // candidate o against this lane's
            const double so = rg1_readlane_d(sc, o);
            const int io = __builtin_amdgcn_readlane(q_id, o);
            rank += (on && ((so > sc) | (!(sc > so) & (io < q_id)))) ? 1 : 0;
            neq += (on && so == sc) ? 1 : 0;
        };
        for (unsigned long long mm = smk; mm != 0ull;) {   // (two candidates per trip: half the branches, the lane reads overlap)
            const int o1 = __builtin_ctzll(mm);
            mm &= mm - 1ull;
            const bool two = mm != 0ull;
            const int o2 = two ? __builtin_ctzll(mm) : o1;
            mm &= mm - 1ull;   // (0 stays 0)
            versus(o1, true);
            versus(o2, two);
        }
        Wn = min(W, kept);
        nsel = 0;
        if (__ballot(inS && neq > 1 && rank < W) != 0ull) {
#ifdef PO_B1_COUNT
            ++cnt_tie;
#endif
            // exact ties reaching into the beam: libstdc++'s partial_sort / sort on the candidates in node-id order
            int pos = 0;
            for (int o = 0; o < NCc; ++o) {
                const int io = __builtin_amdgcn_readlane(q_id, o);
                const int vo = (int)((__ballot(valid) >> o) & 1ull);
                pos += vo & ((io < q_id) ? 1 : 0);
            }
            if (valid) ord[pos] = lane;
            tsc[lane] = sc;
            po_wave_sync();
            if (lane == 0) {
                const double* tp = tsc;
                po_stl_prune<64>(ord, kept, W, [&](int slot_) { return tp[slot_]; }, stl_stk);
            }
            po_wave_sync();
            nsel = ord[min(lane, 63)];
            po_wave_sync();
        } else {   // (no tie reaches into the beam: the ranks below W are taken once each — rank r's slot goes to lane r through LDS)
            ord[lane] = 0;
            po_wave_sync();
            if (inS && rank < Wn) ord[rank] = lane;
            po_wave_sync();
            if (lane < Wn) nsel = ord[lane];
            po_wave_sync();
        }
        }
#ifdef PO_B1_COUNT
        { const long long n_ = wall_clock64(); tk_c += n_ - tk_last; tk_gen += n_ - tk_last; tk_last = n_; }
#endif
        // ---- the new table becomes the previous one
        p_id = q_id; p_fc = q_fc; p_depth = q_depth;
#pragma unroll
        for (int k = 0; k < K; ++k) p_val[k] = out[k];
        p_par = par; p_gpar = gpar; p_plast = plast; p_last = last;
        selv = nsel;
        if (!fastf) {
            // what the NEXT frame needs if it finds this frame's beam unchanged: every slot's parent slot in THIS table (the
            // path above knows it in the previous one) — the beam nodes' by one more pass over the beam's ids, a child's is
            // its beam node's lane.  Nothing will be expanded then: every beam node of this table has its children now.
            int ps1 = -2, ps2 = -2;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int bid = __builtin_amdgcn_readlane(q_id, i);
                const bool in = i < Wc;
                if (in && bid == par) ps1 = i;
                if (in && bid == gpar) ps2 = Wc + A * i + plast;
            }
            k_pslot = (par == 0) ? -1 : ((ps1 >= 0) ? ps1 : ps2);
            k_sym = sym; k_samef = samef; k_dup = dupf;
            stable = true;
        }
        if (MODEL == PO_MODEL_CTC) blank_cum += yrow[A];
        Pnb = Wc;
        Wc = Wn;
        B1_KT(tk_gen);
    }
#ifdef PO_B1_COUNT
    if (r == 0 && lane == 0) printf("[b1 count] read 0: %d frames, %d on the steady table, %d kept the beam; run loop %lld ticks (10 ns), general frames %lld ticks in %d frames (%d on a steady table: %lld ticks to the update; rebuild %lld, update %lld, prune + rank %lld, rest %lld; arena lookups in %d frames, tie replays in %d; %d ranking frames, %d candidates ranked in all)\n", T, cnt_fast, cnt_same, tk_run, tk_gen, cnt_gen, cnt_ff, tk_f, tk_a, tk_b, tk_c, tk_gen - tk_f - tk_a - tk_b - tk_c, cnt_refetch, cnt_tie, cnt_rankf, cnt_S);
#endif
    // ---- label of the top node (PrefixTree::get_label, PrefixTree.h:449-457)
    const int top = __builtin_amdgcn_readlane(selv, 0);
    const int node0 = __shfl(p_id, top), depth0 = __shfl(p_depth, top);
    if (lane == 0) {
        int n = 0;
        if (st == PO_OK) {
            int node = node0;
            n = depth0;
            char* out = seq + seq_off[r];
            const int cap = (int)(seq_off[r + 1] - seq_off[r]);
            if (n > cap) { st = PO_E_CAP; n = 0; }
            else
                for (int i = n - 1; i >= 0; --i) {
                    const int pk = apl[node];
                    out[i] = (char)((alphabet >> (8 * (po_node_last(pk) & 3))) & 0xffu);
                    node = po_node_parent(pk);
                }
        }
        seq_len[r] = n;
        status[r] = st;
    }
}

extern "C" size_t po_beam1d_lds_bytes(int W, int model) {
    const int K = (model == PO_MODEL_CTC) ? 1 : 3;
    const int WM = W > PO_A ? W : PO_A, NC = WM * (PO_A + 1);
    auto al = [](size_t b) { return (b + 15) & ~size_t(15); };
    size_t per = 3 * al(sizeof(int) * NC) + al(sizeof(double) * K * NC) + 4 * al(sizeof(int) * WM);
    return al(sizeof(PoLaeTables)) + 2 * per + 3 * al(sizeof(int) * WM) + 2 * al(sizeof(int) * NC) + al(16 * (size_t)NC) + al(sizeof(int) * 48) +
           2 * al(sizeof(int) * NC);
}

// node-arena entries for a batch: per read root + A children + A * max(W, A) new nodes per frame
extern "C" int64_t po_beam1d_arena_nodes(int n, int64_t total_rows, int W) {
    const int64_t WM = W > PO_A ? W : PO_A;
    return (int64_t)n * (1 + PO_A) + PO_A * WM * total_rows;
}

extern "C" int po_launch_beam1d(const double* y, const int64_t* y_off, int n, int C, int A, uint32_t alphabet,
                                int W, int model,
                                int* arena_pl, int* arena_fc, char* seq,
                                const int64_t* seq_off, int32_t* seq_len, int32_t* status,
                                hipStream_t stream) {
    if (n <= 0) return PO_OK;
    if (W < 1 || W > 64 || A < 1 || A > PO_A) return PO_E_ARG;
    const size_t lds = po_beam1d_lds_bytes(W, model);
    if (lds > 160 * 1024) return PO_E_ARG;
    const bool small = (W > PO_A ? W : PO_A) * (PO_A + 1) <= PO_WAVE && !getenv("PO_B1_TABLES");   // the table fits one wave's lanes
#define PO_LAUNCH_B1(M)                                                                                   \
    do {                                                                                                  \
        if (small) {                                                                                      \
            hipLaunchKernelGGL(beam1d_wave_kernel<M>, dim3(n), dim3(PO_WAVE), 0, stream, y, y_off, A, alphabet, W, arena_pl, \
                               arena_fc, seq, seq_off, seq_len, status);                                  \
            break;                                                                                        \
        }                                                                                                 \
        if (lds > 64 * 1024)                                                                              \
            (void)hipFuncSetAttribute((const void*)beam1d_kernel<M>,                                      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
        hipLaunchKernelGGL(beam1d_kernel<M>, dim3(n), dim3(PO_WAVE), lds, stream, y, y_off, A, alphabet, W, arena_pl, \
                           arena_fc, seq, seq_off, seq_len, status);                                      \
    } while (0)
    if (model == PO_MODEL_CTC) { if (C != A + 1) return PO_E_ARG; PO_LAUNCH_B1(PO_MODEL_CTC); }
    else if (model == PO_MODEL_MERGE) { if (C != A + 1) return PO_E_ARG; PO_LAUNCH_B1(PO_MODEL_MERGE); }
    else if (model == PO_MODEL_FLIPFLOP) { if (C != 2 * A) return PO_E_ARG; PO_LAUNCH_B1(PO_MODEL_FLIPFLOP); }
    else return PO_E_ARG;
#undef PO_LAUNCH_B1
    return PO_OK;
}
