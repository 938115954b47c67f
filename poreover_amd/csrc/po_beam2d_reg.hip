// Pair (2-D) CTC beam search, method "row_col" with a monotone envelope, one-value model ("ctc"), W * (A + 1) <= 32:
// the per-element state lives in REGISTERS (lane = (read, element slot), as beam2d_ring_kernel), every computed value
// goes once into the tagged value store in HBM (as beam2d_kernel), and the kernel keeps almost nothing in LDS — so that
// 16 pairs share a CU (4 waves per SIMD) instead of the ring kernel's 8.
//
// Replaces (like beam2d_kernel, which stays the general form): decoding_cpp.cpp_beam_search_2d (decoding_cpp.pyx:107-139)
// -> beam_search_2d_by_row_col (BeamSearch.h:262-397) over PoreOverPrefixTree2D (PrefixTree.h:492-533) with
// Beam<..., node_greater_max_sym> (Beam.h:35-38,93-108).
//
// Why a third kernel.  Round 3 measured the two ways of holding the window values: beam2d_kernel (HBM store, element
// tables in LDS, 4 waves per SIMD) and beam2d_ring_kernel (LDS rings + registers, 2 waves per SIMD).  The ring kernel's
// per-phase timers at full load (profiles/r04_ring_timing_*.txt) show a wave busy about half of the time with 2 waves
// per SIMD — the device is latency-bound there, and the rings (13 KB of a pair's 20 KB of LDS) are what keeps more
// pairs from being resident.  This kernel is the ring kernel's control structure without the rings:
//   * VALUE STORE (HBM, L2-resident in practice): the reference's per-node maps (PrefixTree.h:76-145), entry =
//     {tag(epoch, node, t), value}, ring rows of R entries, rows in groups of four per parent — beam2d_kernel's layout
//     and recycling rule.  Every read of a value "at rest" is a tagged lookup: absent reads as -inf, exactly
//     probability_at().  Written once per computed (node, read, time).
//   * REGISTERS: a lane's element (ids, rows, parent slot), where its values end (v_done), its last value (v_self),
//     the carried window maximum (value, time, last rise).  Within a scan a child takes its parent's previous value
//     from the parent's LANE (ds_bpermute) — no exchange buffer, no LDS ring, no fence per iteration.
//   * LDS (9.8 KB): 32 y rows per read, the staged windows of up to three parents for a step's new elements, the row
//     group table, the logaddexp tables.
//   * RUN loop: consecutive main steps that keep the set of beam nodes are one tight loop (the new times of the two
//     windows in lockstep, the carried maxima, the score, one comparison per child); the window maximum of a decaying
//     element needs one stored value per step, requested a step ahead.
//   * NEW ELEMENTS (a node entered the beam: its children compute their whole windows, ~ 12 dependent logaddexp
//     iterations on a few lanes): their parent's stored window is staged into LDS in one memory round trip, then the
//     chain runs on LDS and registers only; everybody else continues where it was (the stored bits of the part they
//     already have would be rewritten unchanged: every input is unchanged).
// The walk comes precomputed (beam2d_walk_kernel), envelope checks / blank prefix sums / R from beam2d_prepass_kernel;
// pairs this kernel cannot hold (row groups exhausted, non-monotone envelopes) go to beam2d_kernel through the meta
// word.  Results are bit-identical to the other kernels': the same arithmetic in the same order within every chain.
#include <climits>

#define PO_LAE_EARLY_TABLE 1   // (po_device.h: the exp table entry is requested before the polynomial — a lone wave's chain is latency)
#define PO_LAE_TRIM 1          // (... two instructions fewer: -|x1 - x2| through source modifiers, the exponent add in two)
#define PO_LAE_BRANCHLESS 1    // (... and exp's small-argument test is a select, not a branch: 10 000 pairs 68.3 -> 67.0 ms)
#include "po_beam2d_common.h"
#include "po_host.h"

namespace {

constexpr int RK_NY = 32;     // y rows per read resident in LDS
constexpr int RK_YC = 5;      // doubles per y row (A + 1 <= 5)
#ifndef PO_REG_NGL
#define PO_REG_NGL 96
#endif
#ifndef PO_REG_PS
#define PO_REG_PS 4
#endif
constexpr int RK_NGL = PO_REG_NGL;    // row groups tracked per pair
constexpr int RK_PS = PO_REG_PS;      // parents whose stored window one step can stage for its new elements (W <= 6)
constexpr int RK_PCAP = RK_PS * RK_NY;   // staged values per read: a block of RK_NY times of every parent
constexpr int RK_FRESH = INT_MIN / 2;

struct RegSmem {              // per pair wave
    double ybuf[2][RK_NY][RK_YC];
    double pst[2][RK_PCAP];
    int g_owner[RK_NGL], g_hi0[RK_NGL], g_hi1[RK_NGL];
    // the table fields only the table build (and the rare general scan) looks at, per element slot — the same in both halves of
    // the wave: in LDS they cost no register between two table builds (seven VGPRs of 128, in a kernel that spills)
    int f_fc[32], f_crow2[32], f_par[32], f_gpar[32], f_prow2[32], f_depth[32], f_alias[32];
    int ord[32];              // prune with exact score ties: candidate slots in node-id order (po_stl_prune)
    double csc[32];           // ... and their scores
    int sh[8];
    double rootcum[2];        // the root's alpha (blank prefix sum, PrefixTree.h:509-515) of each read at time rootT: added up as the
    int rootT[2];             // scans pass the times, while children of the root are in the table (the start of a pair)
    double pf0[2][8];         // a run's first step: the beam lanes' values at the window start, fetched with the staging of the
    int pf0_t[2][8];          // step before (their times; -1: none)
    unsigned long long nupd, nupd_x;
};

// ---- the job board (NPW > 1): NPW pair waves and one CHAIN wave per workgroup.  A step's new elements are a few lanes
// running ~ 20 dependent logaddexp iterations while the rest of their wave idles — 40 % of the kernel's vector
// instructions (profiles/r04_pmc_sq.json) at 8 - 16 busy lanes of 64.  Over that range every operand of a chain is at
// rest: the y rows and the parent's staged values sit in the poster's LDS, the seed is a number, the results go to the value
// store and to four numbers per chain.  So the pair wave POSTS its chains of a block (<= 32 times each) and sleeps; the
// chain wave runs the chains of ALL the workgroup's pair waves, one per lane, each lane at its own time — lanes pick up new
// chains whenever they are free — and reports back.  Same arithmetic in the same order within every chain: bit-identical.
constexpr int RK_JOBS = 48;   // chains one pair wave can post per block: 2 reads x W x A fresh children
struct RegJob { int code /* wave | read << 4 | staged parent << 5 | symbol << 8 */, t0, n, rowbase, node, pad_; double seed; };
struct RegRes { double self, mx; int mt, tr; };
union RegSlot { RegJob j; RegRes r; };
template <int NPW>
struct RegGroup {
    PoLaeTables lae;   // (first: at LDS address 0 the tables' offsets fit the immediate fields of ds_read2_b64 — one address per entry)
    RegSmem w[NPW];
    RegSlot slot[NPW > 1 ? NPW : 1][NPW > 1 ? RK_JOBS : 1];
    unsigned posted[8], taken[8], done[8];   // running totals per pair wave: chains posted / picked up / finished
    int pc_rm2[8];                           // per pair wave, for the pair it decodes: store ring mask ...
    unsigned pc_tagep[8];                    // ... and the epoch bits of its tags
    int exited;                              // pair waves that have left the kernel
};

__device__ __forceinline__ void rk_sync() { b2_sync_lds<64>(); }
// the smallest of x over lanes 0 .. n - 1 (n <= 16: the beam slots sit in row 0 of the wave), wave-uniform: four row_shr steps
// in the VALU and one pair of v_readlane instead of 2 n v_readlane and n - 1 minima
__device__ __forceinline__ double rk_row0_min(double x, int n, int lane) {
#ifdef PO_EMU
    double m_ = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 0), __builtin_amdgcn_readlane(__double2loint(x), 0));
    for (int i = 1; i < n; ++i)
        m_ = fmin(m_, __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), i), __builtin_amdgcn_readlane(__double2loint(x), i)));
    (void)lane;
    return m_;
#else
    x = (lane < n) ? x : __builtin_inf();
#define RK_STEP(ctrl)                                                                                                     \
    x = po_vmin(x, __hiloint2double(__builtin_amdgcn_update_dpp(__double2hiint(x), __double2hiint(x), ctrl, 0xf, 0xf, false), \
                                    __builtin_amdgcn_update_dpp(__double2loint(x), __double2loint(x), ctrl, 0xf, 0xf, false)))
    RK_STEP(0x111); RK_STEP(0x112); RK_STEP(0x114); RK_STEP(0x118);
#undef RK_STEP
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 15), __builtin_amdgcn_readlane(__double2loint(x), 15));
#endif
}
__device__ __forceinline__ double rk_readlane_d(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}

}  // namespace

#ifndef PO_REG_WAVES
#define PO_REG_WAVES 4
#endif
// COUNT: the instantiation po_profile_update_counter asks for (update_prob evaluations of the reference's schedule and executed
// ones, added up per step: ballots, a wave reduction per scan); the product path carries none of it.
template <int NPW, bool COUNT = false>
__global__ __launch_bounds__(NPW == 1 ? 64 : 64 * (NPW + 1), PO_REG_WAVES) void beam2d_reg_kernel(X2Args a) {
    using Ent = Entry<1>;
    __shared__ RegGroup<NPW> gsm;
    const int wave = (NPW == 1) ? 0 : (int)(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, r = lane >> 5, s = lane & 31, hb = lane & 32;
    RegSmem& sm = gsm.w[(wave < NPW) ? wave : 0];
    const int slotid = blockIdx.x * NPW + wave;      // this pair wave's slice of the workspace
    const bool use_board = (NPW > 1) && a.reg_board != 0;
    const int A = a.A, W = a.W, C = a.C;
    const int divA = (65536 + A - 1) / A;   // x / A == (x * divA) >> 16 for the slot numbers divided here
    Ent* const pool = (Ent*)(a.pool + (size_t)slotid * a.pool_bytes);
    const long long pool_entries = (long long)(a.pool_bytes / sizeof(Ent));
    int* const apl = a.arena + (size_t)slotid * 3 * a.arena_cap;
    int* const afc = apl + a.arena_cap;
    int* const acrow = afc + a.arena_cap;
    int* const g_hi = r ? sm.g_hi1 : sm.g_hi0;
    const bool is_pair_wave = wave < NPW && slotid < a.reg_slots;

    // ---- epoch tags across pairs and launches (as beam2d_kernel): no memset of the store
    unsigned epoch = 0;
    auto clear_slice = [&]() {
        for (long long i = lane; i < pool_entries; i += 64) pool[i].tag = 0ull;
        rk_sync();
    };
    if (is_pair_wave) {
        unsigned long long* stp = a.wgstate + 2 * (size_t)slotid;
        const unsigned long long w0 = stp[0], w1 = stp[1];
        const bool ok = (w0 == (a.magic ^ (unsigned long long)slotid));
        epoch = ok ? (unsigned)w1 : 0u;
        if (!ok) clear_slice();
    }
    po_lae_tables_load(&gsm.lae, (int)threadIdx.x, (int)blockDim.x);
    const PoLaeFast lae{&gsm.lae};
    if (lane == 0 && wave < NPW) { sm.nupd = 0; sm.nupd_x = 0; }
    if (threadIdx.x < 8) { gsm.posted[threadIdx.x] = 0u; gsm.taken[threadIdx.x] = 0u; gsm.done[threadIdx.x] = 0u; }
    if (threadIdx.x == 0) gsm.exited = 0;
    __syncthreads();
    unsigned posted_total = 0u;   // chains this pair wave has posted so far (the board keeps running totals)
    if constexpr (NPW > 1) {
        if (wave == NPW) {
            // ================================================================ the CHAIN wave
            // (seven pair waves wait for what this wave computes: it goes first whenever it can issue)
#ifndef PO_EMU
            __builtin_amdgcn_s_setprio(3);
#endif
            bool busy = false, fin = false;
            int jw = 0, jslot = 0, t = 0, k = 0, n = 0, rowbase = 0, node = 0, symo = 0, mt = -1, tr = INT_MIN, rm2 = 0;
            unsigned tagep = 0u;
            double self = PO_NEG_INF, mx = PO_NEG_INF;
            const double* yb = &gsm.w[0].ybuf[0][0][0];
            const double* ps = &gsm.w[0].pst[0][0];
            char* pl = a.pool;
            for (;;) {
#ifdef PO_EMU
                { static long itc2 = 0; static const bool dbg2_ = getenv("EMU_CHAIN_DEBUG") != nullptr; if (dbg2_ && lane == 0 && (++itc2 % 2000) == 0) { fprintf(stderr, "[chain loop %ld] exited %d", itc2, gsm.exited); for (int w2 = 0; w2 < NPW; ++w2) fprintf(stderr, "  w%d p%u t%u d%u", w2, gsm.posted[w2], gsm.taken[w2], gsm.done[w2]); fprintf(stderr, "\n"); } }
#endif
                // ---- free lanes pick up posted chains (totals: posted - taken chains of a pair wave are waiting)
                if (__ballot(!busy) != 0ull) {
                    // lane w looks at pair wave w's totals: one LDS round trip tells which pair waves have chains waiting
                    const int lw = lane & 7;
                    const unsigned po_l = ((volatile unsigned*)gsm.posted)[lw], tk_l = gsm.taken[lw];   // (taken: this wave's own)
                    unsigned wm = (unsigned)__ballot(lane < NPW && (int)(po_l - tk_l) > 0);
                    while (wm != 0u) {   // (wave-uniform)
                        const int w2 = __builtin_ctz(wm);
                        wm &= wm - 1u;
                        const unsigned po = (unsigned)__builtin_amdgcn_readlane((int)po_l, w2), tk = (unsigned)__builtin_amdgcn_readlane((int)tk_l, w2);
                        const int avail = (int)(po - tk);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        const unsigned long long fm = __ballot(!busy);
                        const int nf = __popcll(fm);
                        if (nf == 0) break;
                        const int give = min(avail, nf);
                        const int myrank = __popcll(fm & ((1ull << lane) - 1ull));
                        if (!busy && myrank < give) {
                            const int si = (int)((tk + (unsigned)myrank) % (unsigned)RK_JOBS);
                            const RegJob j = gsm.slot[w2][si].j;
                            busy = true; jw = w2; jslot = si;
                            const int rr = (j.code >> 4) & 1, myk = (j.code >> 5) & 7;
                            symo = (j.code >> 8) & 7;
                            t = j.t0; n = j.n; k = 0; rowbase = j.rowbase; node = j.node; self = j.seed;
                            mx = PO_NEG_INF; mt = -1; tr = INT_MIN;
                            yb = &gsm.w[w2].ybuf[rr][0][0]; ps = &gsm.w[w2].pst[rr][myk * RK_NY];
                            rm2 = gsm.pc_rm2[w2]; tagep = gsm.pc_tagep[w2];
                            pl = a.pool + (size_t)(blockIdx.x * NPW + w2) * a.pool_bytes;
                        }
                        rk_sync();
                        if (lane == 0) gsm.taken[w2] = tk + (unsigned)give;
                        rk_sync();
                    }
                }
                // ---- one iteration of every running chain (update_prob: PrefixTree.h:518-531)
                if (busy) {
                    const double* yrow = yb + (t & (RK_NY - 1)) * RK_YC;
                    const double ya = yrow[symo], ybl = yrow[A];
                    const double pp = ps[k];
                    const double out = lae(pp + ya, self + ybl);
                    Ent e;
                    e.tag = ((unsigned long long)(tagep | (((unsigned)node >> 8) & 0xffffu)) << 32) | (((unsigned)node << 24) | ((unsigned)t & 0xffffffu));
                    e.v[0] = out;
                    *(Ent*)(pl + (size_t)(unsigned)((rowbase + (t & rm2)) << 4)) = e;
                    if (out > self) tr = t;
                    self = out;
                    mt = (out >= mx) ? t : mt;
                    mx = po_vmax(mx, out);
                    ++t; ++k;
                    if (k == n) { busy = false; fin = true; }
                }
                // ---- finished chains: their poster reads the values back through the store, so the writes must have landed
                // before it is told (one wait for all the chains that finish in this iteration)
                if (__ballot(fin) != 0ull) {
                    if (fin) { RegRes rr_; rr_.self = self; rr_.mx = mx; rr_.mt = mt; rr_.tr = tr; gsm.slot[jw][jslot].r = rr_; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (the stores to the value store and the result, before the count)
                    if (fin) { atomicAdd(&gsm.done[jw], 1u); fin = false; }
                } else if (__ballot(busy) == 0ull) {
#ifdef PO_EMU
                    { static long itc = 0; static const bool dbg_ = getenv("EMU_CHAIN_DEBUG") != nullptr; if (dbg_ && lane == 0 && (++itc % 200000) == 0) { fprintf(stderr, "[chain idle] exited %d", gsm.exited); for (int w2 = 0; w2 < NPW; ++w2) fprintf(stderr, "  w%d p%u t%u d%u", w2, gsm.posted[w2], gsm.taken[w2], gsm.done[w2]); fprintf(stderr, "\n"); } }
#endif
                    bool idle = ((volatile int*)&gsm.exited)[0] >= NPW;
                    for (int w2 = 0; w2 < NPW; ++w2) idle = idle && (((volatile unsigned*)gsm.posted)[w2] == gsm.taken[w2]);
                    if (__builtin_amdgcn_readfirstlane((int)idle) != 0) break;
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            return;
        }
    }
    if (!is_pair_wave) {
        if (NPW > 1 && lane == 0) atomicAdd(&gsm.exited, 1);
        return;
    }
#ifdef PO_REG_TIMING
    // phase timers of workgroup 0 (wall_clock64: 100 MHz) and counts: see po_reg_launch for the names
    long long tk[40], tlast = wall_clock64();
    for (int i = 0; i < 40; ++i) tk[i] = 0;
#define KT(i) do { const long long n_ = wall_clock64(); tk[(i)] += n_ - tlast; tlast = n_; } while (0)
#define KC(i, n) do { tk[(i)] += (n); } while (0)
#ifdef PO_REG_TIMING2   // finer buckets inside the run loop and the table build (their time leaves buckets 0 and 8)
#define KT2(i) KT(i)
#else
#define KT2(i) do {} while (0)
#endif
#else
#define KT2(i) do {} while (0)
#define KT(i) do {} while (0)
#define KC(i, n) do {} while (0)
#endif

    for (;;) {
        // ---------------------------------------------------------------- next pair from the queue
        int pi = 0;
        if (lane == 0) {
            const int q = atomicAdd(a.queue, 1);
            pi = (a.order != nullptr && q < a.n) ? a.order[q] : q;   // longest pairs first (pair_order_kernel)
        }
        pi = __builtin_amdgcn_readfirstlane(pi);
        if (pi >= a.n) break;
        epoch++;
        if (__builtin_expect((epoch & 0xffffu) == 0, 0)) { clear_slice(); epoch++; }
        const int2 m = a.meta[pi];
        if (__builtin_expect(m.y == X2_DEFERRED, 0)) continue;                 // beam2d_kernel decodes it after this kernel
        if (__builtin_expect(m.x != PO_OK || m.y < 0, 0)) {                    // refused by the pre-pass, or skipped upstream
            if (lane == 0) {
                a.seq_len[pi] = 0;
                if (m.y >= 0) a.status[pi] = m.x;
            }
            continue;
        }
        const int64_t o1 = a.y1_off[pi], o2 = a.y2_off[pi];
        const int U = (int)(a.y1_off[pi + 1] - o1), V = (int)(a.y2_off[pi + 1] - o2);
        // (this lane's read: its rows, length and blank prefix sums are put together where they are used — a y reload every
        //  ~ 16 steps, the root's children at the start of a pair — rather than held in six registers across the walk)
        auto yr_ = [&]() -> const double* { return r ? a.y2 + o2 * C : a.y1 + o1 * C; };
        const int4* const sched = a.sched + (o2 - a.y2_off[0]);
        const int nmain = a.nmain[pi];
        const int R2 = m.y, Rm2 = R2 - 1;
        if (__builtin_expect(R2 > 256, 0)) {   // windows of 255 frames and more: the packed walk records below keep a window's length in 8 bits
            if (lane == 0) { a.meta[pi] = make_int2(PO_OK, X2_DEFERRED); a.queue[16] = 1; }
            continue;
        }
        const int NG = (int)min((long long)RK_NGL, pool_entries / ((long long)PO_A * 2 * R2));
        int st = PO_OK;

        // ---------------------------------------------------------------- the value store (this lane's read)
        // (entry index and byte offset stay within 32 bits: a workgroup's slice is a few MB — one v_lshl_add per access
        //  instead of 64-bit address arithmetic; the tag's words are put together from per-call constants the same way)
        const char* const poolb = (const char*)pool;
        const unsigned tag_ep = (epoch & 0xffffu) << 16;
        if (NPW > 1 && lane == 0) { gsm.pc_rm2[wave] = Rm2; gsm.pc_tagep[wave] = tag_ep; }
        auto t2_off = [&](int row2, int tq) -> unsigned { return (unsigned)(((row2 * 2 + r) * R2 + (tq & Rm2)) << 4); };
        auto t2_entry = [&](int row2, int tq) -> const Ent* { return (const Ent*)(poolb + (size_t)t2_off(row2, tq)); };
        auto tag_of = [&](int node, int tq) -> unsigned long long {   // == make_tag(epoch, node, tq) for 0 <= tq < 2^24
            const unsigned hi = tag_ep | (((unsigned)node >> 8) & 0xffffu), lo = ((unsigned)node << 24) | (unsigned)tq;   // (0 <= tq < 2^24: the pre-pass)
            return ((unsigned long long)hi << 32) | lo;
        };
        auto t2_read = [&](int row2, int node, int tq) -> double {
            double v = PO_NEG_INF;
            if (tq >= 0 && row2 >= 0) {
                const Ent e = *t2_entry(row2, tq);
                if (e.tag == tag_of(node, tq)) v = e.v[0];
            }
            return v;
        };
        auto t2_write = [&](int row2, int node, int tq, double v) {
            Ent e;
            e.tag = tag_of(node, tq);
            e.v[0] = v;
            *(Ent*)(const_cast<char*>(poolb) + (size_t)t2_off(row2, tq)) = e;
        };
        // ---------------------------------------------------------------- per-lane element state (slot s, read r)
        // table fields (the same in both halves of the wave)
        int e_id = 0, e_row2 = -1, e_sym = 0, e_ps = PS_ROOT;   // (first child, children's row group, parent, grandparent, the
                                                                  //  parent's row, depth, alias: sm.f_*)
        if (r == 0) {
            sm.f_fc[s] = -1; sm.f_crow2[s] = -1; sm.f_par[s] = 0; sm.f_gpar[s] = -1; sm.f_prow2[s] = -1;
            sm.f_depth[s] = (s < A) ? 1 : 0; sm.f_alias[s] = -1;
        }
        bool live = false;
        // values of this read: computed and stored up to v_done (exclusive); v_fresh: 1 = an element again, its last
        // value is in the store; 2 = a node that never computed
        int v_done = RK_FRESH, v_fresh = 0;
        double v_self = PO_NEG_INF, v_mx = PO_NEG_INF;
        int v_mt = -1, v_td = 0;
        // a beam node whose parent is no element any more (FROZEN): the parent's last value and its time, taken when the
        // parent left — later times are absent (-inf), earlier ones are in the store.  fz_t = INT_MAX: nothing captured.
        double fz_val = PO_NEG_INF;
        int fz_t = INT_MAX;
        // the stored value the next step's carried maximum may need (a decaying element: its value at the window start),
        // requested a step ahead: pf_t = its time (-1: none)
        // (pf_val / pf_t live in the run loop only: a long-lived entry in registers is what the allocator spills first, and a
        //  spilled prefetch is a wait at the point of issue)
        int nb = A, ne = A;
        int next_id = 1 + A;
        int gcur = 1;               // row group allocation cursor
        int yhi = 0;                // y rows [yhi - RK_NY, yhi) of this lane's read are in sm.ybuf
        int sel[6] = {0, 1, 2, 3, 4, 5};

        for (int q = lane; q < RK_NGL; q += 64) { sm.g_owner[q] = -1; sm.g_hi0[q] = 0; sm.g_hi1[q] = 0; }
        rk_sync();
        // root = node 0; its A children = nodes 1..A in row group 0 (BeamSearch.h:286-293), updated at t = 0 on both reads
        if (lane == 0) {
            apl[0] = po_pack_node(-1, A); afc[0] = 1; acrow[0] = 0;
            sm.g_owner[0] = 0; sm.g_hi0[0] = 1; sm.g_hi1[0] = 1;
        }
        if (s < A) {
            if (r == 0) { apl[1 + s] = po_pack_node(0, s); afc[1 + s] = -1; acrow[1 + s] = -1; }
            e_id = 1 + s; e_row2 = s; e_sym = sym_pack(s, A, true); e_ps = PS_ROOT;
            live = true;
            const double* const yr = yr_();
            const double out = lae(0.0 + yr[s], PO_NEG_INF + yr[A]);   // update_prob(n, r, 0): parent = root at t = -1
            t2_write(e_row2, e_id, 0, out);
            v_done = 1; v_fresh = 0; v_self = out;
            v_mx = out; v_mt = 0; v_td = 0;   // (the window maximum over [0, 1))
            if (s == 0) { sm.rootcum[r] = 0.0 + yr[A]; sm.rootT[r] = 0; }   // (serial in t from 0.0, as the reference adds)
        }
        rk_sync();

        int mstep = 0, up = -1, vp = -1;
        int pf0_step = -1;   // the main step sm.pf0 was filled for
        // The walk's records, 64 at a time: lane l holds record 64 * batch + l of the current batch and of the next one
        // (requested a batch ahead: the load's latency never shows), the step's own record comes out with v_readlane.
        // (kept PACKED, two words per record — time | window length << 24; times stay below 2^24 and a window below the
        //  store's ring length of <= 256 — : four registers for the two batches instead of eight)
        // (The walk INSIDE this kernel — the wave putting its own next 64 records together — was built and measured in
        //  round 4: 10 000 pairs 68.3 instead of 66.7 ms, a single pair 15.5 instead of 15.4: the walk is ~ 2 800 dependent
        //  rounds per pair, serial on this wave whether it runs here or in front; as a kernel of its own its waves fill the
        //  device 8 - 10 to a SIMD.  It stays a kernel, and got catch-up runs resolved in one round instead.)
        auto rec_load = [&](int i) -> int2 {
            const int4 q = sched[min(i, max(nmain - 1, 0))];
            return make_int2(q.x | ((q.z - q.x) << 24), q.y | ((q.w - q.y) << 24));
        };
        int2 rcur = rec_load(lane), rnxt = rec_load(64 + lane);
        auto rec_at = [&](int i) -> int4 {   // record of main step i (uniform i within the current batch)
            const int l = i & 63;
            const int px = __builtin_amdgcn_readlane(rcur.x, l), py = __builtin_amdgcn_readlane(rcur.y, l);
            const int uu = px & 0xffffff, vv = py & 0xffffff;
            return make_int4(uu, vv, uu + (int)((unsigned)px >> 24), vv + (int)((unsigned)py >> 24));
        };
        int4 rec = rec_at(0);
        bool have_children = false;   // the table has its children slots (false only before the first expansion)
        bool tbl_fresh = true;        // the table has elements that have not computed yet (set by rebuild)
        bool tbl_uneven = false;      // a catch-up scan moved the beam nodes beyond their children
        unsigned long long cnt_ref = 0, cnt_x = 0;

        auto read_own = [&](int tq) -> double { return t2_read(e_row2, e_id, tq); };

        // ---------------------------------------------------------------- y rows [t0, t0 + RK_NY) of this read -> LDS
        // (all of a lane's loads go out together: one memory round trip per reload)
        auto y_reload = [&](int t0) {
            constexpr int PER = (RK_NY * RK_YC + 31) / 32;   // elements per lane (C <= RK_YC)
            double v[PER];
            int slot[PER];
            const double* const yr = yr_();
            const int Tr = r ? V : U;
            const int divC = (65536 + C - 1) / C;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int i = s + 32 * j;
                const int q = (i * divC) >> 16, c = i - q * C;
                const int t = t0 + q;
                const bool ok = i < RK_NY * C && t < Tr;
                slot[j] = ok ? (t & (RK_NY - 1)) * RK_YC + c : -1;
                v[j] = ok ? yr[(int64_t)t * C + c] : 0.0;
            }
            double* const yb = &sm.ybuf[r][0][0];
#pragma unroll
            for (int j = 0; j < PER; ++j)
                if (slot[j] >= 0) yb[slot[j]] = v[j];
        };

        // ---------------------------------------------------------------- the carried part of a window's maximum
        // [ws, start) is not recomputed: its maximum is what the previous step left (cmx at cmt), unless that time has
        // left the window — then the stored values are looked at again.  A node past its peak decays frame by frame:
        // if the values were non-increasing since before the window start (td <= ws), the maximum is the first one.
        // pf: an entry requested a step ahead for exactly that time (pf_t), else the store is asked now.
        // Two parts.  carried_one (per lane): the single value of a decaying element; returns true when the lane needs the
        // rescan.  rescan_wave (the whole wave, uniform control flow): the lanes that need one are served in turn, each by
        // all 64 lanes — lane i asks for the stored value at ws + i of THAT lane's row (one memory round trip for the
        // range instead of one per time), the maximum comes from po_wave_max, its latest time and the last rise from
        // ballots.  (Done lane by lane and time by time, a late bump in an otherwise falling window was rescanned at
        // every step until the window start had passed it: ~ 10 dependent reads in each of 30 % of the run-loop steps.)
        auto carried_one = [&](int ws, double& cmx, int& cmt, int td, int pf_t = -1, double pf_val = 0.0) -> bool {
            if (td > ws) return true;
            cmx = (ws == pf_t) ? pf_val : read_own(ws);
            cmt = ws;
            return false;
        };
        auto rescan_wave = [&](bool need, int ws, int start, double& cmx, int& cmt, int& td) {
            unsigned long long m = __ballot(need);
            KC(21, __popcll(m)); KC(10, m != 0ull ? 1 : 0);
            // (two lanes per round, their loads issued together, measured in round 4: a single pair 14.57 -> 14.36 ms, but
            //  1 250 pairs 20.2 -> 20.5 and 4 096 pairs 30.6 -> 31.3 ms — twice the reduction code; one lane at a time stays)
            while (m != 0ull) {   // (wave-uniform)
                const int L = (int)__builtin_ctzll(m);
                m &= m - 1ull;
                const int wsL = __builtin_amdgcn_readlane(ws, L), teL = __builtin_amdgcn_readlane(min(td + 1, start), L);
                const int rowL = __builtin_amdgcn_readlane(e_row2, L), idL = __builtin_amdgcn_readlane(e_id, L);
                const int rowbase = (rowL * 2 + (L >> 5)) * R2;
                double bmx = PO_NEG_INF, pvc = PO_NEG_INF;
                int bmt = -1, btd = wsL;
                for (int base = wsL; base < teL; base += 64) {
                    const int tq = base + lane;
                    const bool valid = tq < teL;
                    double val = PO_NEG_INF;
                    if (valid) {
                        const Ent e = *(const Ent*)(poolb + (size_t)(unsigned)((rowbase + (tq & Rm2)) << 4));
                        if (e.tag == tag_of(idL, tq)) val = e.v[0];
                    }
                    const double mxv = po_wave_max(val);
                    const unsigned long long eq = __ballot(valid && val == mxv);   // (later times win ties: the highest lane)
                    if (mxv >= bmx && eq != 0ull) { bmx = mxv; bmt = base + 63 - (int)__builtin_clzll(eq); }
                    double prev = __shfl(val, (lane + 63) & 63);
                    if (lane == 0) prev = pvc;
                    const unsigned long long rs = __ballot(valid && tq > wsL && val > prev);
                    if (rs != 0ull) btd = base + 63 - (int)__builtin_clzll(rs);
                    pvc = rk_readlane_d(val, 63);
                }
                if (lane == L) { cmx = bmx; cmt = bmt; td = btd; }
            }
        };

        // ---------------------------------------------------------------- one scan (the general form)
        // Every participating lane computes [max(done, ws), we) of its read, all lanes of a read in lockstep on t: a child
        // at t takes its parent's t - 1 from the parent's lane when the parent computed it one iteration earlier (or holds
        // it as its last value), from the store otherwise.  MAIN steps (is_main) track the window maximum; catch-up scans
        // (BeamSearch.h:314-336) move the beam nodes only.
        double smx = PO_NEG_INF;   // out: max over this read's window (main steps)
        auto scan = [&](bool is_main, int ws0, int we0, int ws1, int we1, int nlanes) {
            const int ws = r ? ws1 : ws0, we = r ? we1 : we0;
            const bool part = live && s < nlanes && we > ws;
            // a window end that moves back cannot happen on a monotone envelope (the pre-pass hands the others to
            // beam2d_kernel); should it, the pair goes the same way
            // (st stays wave-uniform: the walk loop's condition reads it)
            if (is_main && __ballot(part && v_fresh == 0 && v_done > we) != 0ull) st = PO_E_NOMEM;
#ifdef PO_EMU_DEBUG
            if (is_main && part && v_fresh == 0 && v_done > we) printf("BACK lane %d id %d done %d ws %d we %d mstep %d nmain %d\n", lane, e_id, v_done, ws, we, mstep, nmain);
#endif
            int start = max(v_done, ws);
            double self = PO_NEG_INF;
            if (part) {
                if (v_fresh != 0) {
                    start = ws;
                    self = (v_fresh == 1) ? read_own(start - 1) : PO_NEG_INF;
                } else if (start > v_done) {
                    // a gap (catch-ups went beyond the last window): the value at start - 1 was never computed
                } else {
                    self = v_self;
                }
            }
            const bool part2 = part && start < we;
            double mx = PO_NEG_INF, cmx = PO_NEG_INF;
            int mt = -1, cmt = -1, td = ws, tr = INT_MIN;
            const bool has_c = is_main && part && start > ws;
            bool rsc = false;
            if (has_c) {
                td = v_td;
                if (v_mx == PO_NEG_INF || (v_mt >= ws && v_mt < start)) { cmx = v_mx; cmt = v_mt; }
                else rsc = carried_one(ws, cmx, cmt, td);
            }
            rescan_wave(rsc, ws, start, cmx, cmt, td);
            const int sym = sym_last(e_sym);
            const bool has_root = __ballot(live && e_ps == PS_ROOT) != 0ull;
            bool bad_root = false;
            // the parent's lane: where it starts and ends in this scan (its `self` is its value at p_start - 1 before the
            // first iteration, then at the time it computed last)
            const int plane = (e_ps >= 0) ? (hb | e_ps) : lane;
            const int p_start = __shfl(part2 ? start : INT_MAX, plane), p_we = __shfl(part2 ? we : INT_MIN, plane);
            int tm_ = part2 ? start : INT_MAX;
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) tm_ = min(tm_, __shfl_xor(tm_, off));
            const int tmin = tm_;                                   // (half-uniform)
            const int span = (tmin == INT_MAX) ? 0 : we - tmin;
            const int niter = max(__builtin_amdgcn_readlane(span, 0), __builtin_amdgcn_readlane(span, 32));
            int k = 0;
            while (k < niter) {
                const int tcur = tmin + k;   // (garbage when this half has nothing to do: guarded by span)
                const bool hw = k < span;    // this half still has times to compute
                rk_sync();                   // (every lane is done with the rows a reload overwrites)
                if (hw && !(tcur >= yhi - RK_NY && tcur < yhi)) { y_reload(tcur); yhi = tcur + RK_NY; }
                rk_sync();
                const int cend = hw ? (min(we, yhi) - tmin) : niter;
                const int kend = min(__builtin_amdgcn_readlane(cend, 0), __builtin_amdgcn_readlane(cend, 32));
                for (; k < kend; ++k) {
                    const int t = tmin + k;
                    const double ps_self = __shfl(self, plane);
                    if (part2 && t >= start && t < we) {
                        const double ya = sm.ybuf[r][t & (RK_NY - 1)][sym], yb = sm.ybuf[r][t & (RK_NY - 1)][A];
                        const int tm = t - 1;
                        double pp;
                        if (e_ps >= 0) {
                            if (tm >= p_start - 1 && tm < p_we && p_start != INT_MAX) pp = ps_self;
                            else pp = t2_read(sm.f_prow2[s], sm.f_par[s], tm);
                        } else if (e_ps == PS_ROOT) {
                            pp = 0.0;
                            if (tm >= 0) { pp = sm.rootcum[r]; bad_root = bad_root || (tm != sm.rootT[r]); }
                        } else if (tm >= fz_t) {
                            pp = (tm == fz_t) ? fz_val : PO_NEG_INF;                                 // frozen parent: its last value, then nothing
                        } else {
                            pp = t2_read(sm.f_prow2[s], sm.f_par[s], tm);
                        }
                        const double out = lae(pp + ya, self + yb);
#ifdef PO_RING_TRACE_NODE
                        if (pi == 0 && e_id == PO_RING_TRACE_NODE) printf("V %d %d %d %.17g %.17g %.17g G ps %d fzt %d main %d\n", e_id, r, t, out, pp, self, e_ps, fz_t, (int)is_main);
#endif
                        t2_write(e_row2, e_id, t, out);
                        if (out > self) tr = t;   // the last time a value rose
                        self = out;
                        mt = (out >= mx) ? t : mt;
                        mx = po_vmax(mx, out);
                    }
                    if (has_root) {   // the root's alpha moves on with the times this read's scans pass (every one of them, in order)
                        rk_sync();
                        if (s == 0 && k < span && t == sm.rootT[r] + 1) { sm.rootcum[r] += sm.ybuf[r][t & (RK_NY - 1)][A]; sm.rootT[r] = t; }
                        rk_sync();
                    }
                }
            }
            if (__ballot(bad_root) != 0ull) st = PO_E_NOMEM;   // (a time the sums have not reached: cannot happen — beam2d_kernel would take the pair)
            if (has_c && !(mx >= cmx)) { mx = cmx; mt = cmt; }   // (new values, later in time, win ties)
            if (part2) { v_done = we; v_self = self; v_fresh = 0; }
            if (is_main) {
                if (part) { v_mx = mx; v_mt = mt; v_td = max(td, tr); }
                smx = part ? mx : PO_NEG_INF;
            }
            KT(is_main ? 4 : 5); KC(is_main ? 16 : 17, 1); KC(18, niter);
            if constexpr (COUNT) {
                const int lenx = part2 ? we - start : 0;
                int tot = lenx;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) tot += __shfl_xor(tot, off);
                cnt_x += (unsigned)tot;
            }
        };

        // ---------------------------------------------------------------- a main step with NEW elements
        // After a rebuild: the continuing elements (all ending at the same time dr, as in a run) only need the new times
        // [dr, we); the fresh ones — children of a node that entered the beam — need their whole window [ws, we).  Over
        // [ws, dr) their parent does not move: its values are at rest in the store, so they are STAGED into LDS in one
        // memory round trip (with the fresh lanes' own seeds), and phase 1 runs the fresh lanes' chains on LDS and registers
        // only; phase 2 is the run loop's lockstep over the new times for everybody.  Returns false (nothing done) when
        // the step is not of this kind: scan() takes it.
        auto scan_new = [&](int u, int ce, int v, int re) -> bool {
            const int d0 = __builtin_amdgcn_readlane(v_done, 0), d1 = __builtin_amdgcn_readlane(v_done, 32);   // (beam slot 0 always continues)
#ifdef PO_EMU_DEBUG
            if (!(u <= d0 && d0 <= ce && v <= d1 && d1 <= re)) {
                if (lane == 0) printf("WHY0 u %d d0 %d ce %d v %d d1 %d re %d\n", u, d0, ce, v, d1, re);
                return false;
            }
#endif
            if (__builtin_expect(!(u <= d0 && d0 <= ce && v <= d1 && d1 <= re), 0)) return false;
            const int ws = r ? v : u, we = r ? re : ce, dr = r ? d1 : d0;
            const bool fresh = live && v_fresh != 0;
            const bool cont = live && v_fresh == 0;
            // everybody who continues ends at dr; fresh lanes hang under a continuing lane; no root, no older frozen values
            const bool bad = (cont && v_done != dr) || (live && e_ps == PS_ROOT) || (fresh && e_ps < 0);
            const bool pfresh = __shfl((int)fresh, hb | max(e_ps, 0)) != 0;
#ifdef PO_EMU_DEBUG
            if (__ballot(bad || (fresh && pfresh)) != 0ull) {
                const int w0 = (int)(__ballot(cont && v_done != dr) != 0), w1 = (int)(__ballot(live && e_ps == PS_ROOT) != 0), w2 = (int)(__ballot(fresh && e_ps < 0) != 0);
                const int w3 = 0, w4 = (int)(__ballot(fresh && pfresh) != 0);
                if (lane == 0) printf("WHY cont_done %d root %d freshfrozen %d frozenold %d pfresh %d\n", w0, w1, w2, w3, w4);
                return false;
            }
#else
            if (__builtin_expect(__ballot(bad || (fresh && pfresh)) != 0ull, 0)) return false;
#endif
            // ---- the parents to stage (beam slots with fresh children): at most RK_PS
            int pj[RK_PS], nps = 0;
#pragma unroll
            for (int k = 0; k < RK_PS; ++k) pj[k] = -1;
            bool many = false;
            for (int jj = 0; jj < nb; ++jj) {   // (wave-uniform)
                if (__ballot(fresh && e_ps == jj) == 0ull) continue;
                if (nps < RK_PS) {
#pragma unroll
                    for (int k = 0; k < RK_PS; ++k) if (k == nps) pj[k] = jj;
                    nps++;
                } else many = true;
            }
#ifdef PO_EMU_DEBUG
            if (many && lane == 0) printf("WHY many\n");
#endif
            const int n1 = dr - ws;   // (half-uniform, >= 0): times the fresh lanes compute before everybody else starts
            if (__builtin_expect(many, 0)) return false;
            int myk = 0;
#pragma unroll
            for (int k = 1; k < RK_PS; ++k) myk = (e_ps == pj[k] && pj[k] >= 0) ? k : myk;
            // the fresh lanes' own seeds (an element again: its last value is in the store)
            Ent se; se.tag = 0ull; se.v[0] = 0.0;
            const bool want_seed = fresh && v_fresh == 1 && ws - 1 >= 0;
            // (the same registers, other lanes: a continuing beam lane whose window maximum has left the window and whose
            //  values fall — the run that follows this step asks for its value at ws first thing)
            const bool want_pf = cont && s < nb && v_done > ws && v_mx != PO_NEG_INF && v_mt < ws && v_td <= ws;
            if (want_seed) se = *t2_entry(e_row2, ws - 1);
            else if (want_pf) se = *t2_entry(e_row2, ws);
            double mx = PO_NEG_INF, self = PO_NEG_INF;
            int mt = -1, tr = INT_MIN;
            const int sym = sym_last(e_sym);
            const double* const yb_ = &sm.ybuf[r][0][0];
            // ---- phase 1: the fresh lanes over [ws, dr) — every operand is at rest.  In blocks of RK_NY times: the parents'
            // stored values of the block are STAGED (lane i of a read asks for time ws - 1 + k0 + i of each parent: one memory
            // round trip for all of them, with the y rows of the block), then the chains run on LDS and registers only — a
            // load inside the chain loop would make the compiler wait for vmcnt(0) there, i.e. for every value-store write
            // of the iteration before.
            const int n1max = max(d0 - u, d1 - v);
            KT(1); KC(13, 1); KC(14, n1max);
            const double* const ps_ = &sm.pst[r][myk * RK_NY];
            for (int k0 = 0; k0 < n1max; k0 += RK_NY) {
                rk_sync();   // (every lane is done with the rows and staged values of the block before)
                // (the first two parents' entries are asked for BEFORE the y rows: one memory round trip for the rows and the
                //  staged values of the usual step — one or two nodes entered the beam — instead of one after the other)
                const int i = k0 + s, tq = ws - 1 + i;
                Ent e01[2];
                int pid01[2] = {0, 0};
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    e01[k].tag = 0ull; e01[k].v[0] = 0.0;
                    if (k < nps) {   // (wave-uniform)
                        const int jk = pj[k];
                        const int prow = __builtin_amdgcn_readlane(e_row2, jk);
                        pid01[k] = __builtin_amdgcn_readlane(e_id, jk);
                        if (i < n1 && tq >= 0) e01[k] = *t2_entry(prow, tq);
                    }
                }
                {
                    const int lo = ws + k0, hi = min(lo + RK_NY, dr);
                    if (hi > lo && !(lo >= yhi - RK_NY && hi <= yhi)) { y_reload(lo); yhi = lo + RK_NY; }
                }
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (k < nps && i < n1) sm.pst[r][k * RK_NY + s] = (tq >= 0 && e01[k].tag == tag_of(pid01[k], tq)) ? e01[k].v[0] : PO_NEG_INF;
                for (int k = 2; __builtin_expect(k < nps, 0); ++k) {   // (wave-uniform; three and more parents: rare)
                    int jk = pj[0];
#pragma unroll
                    for (int q = 1; q < RK_PS; ++q) jk = (k == q) ? pj[q] : jk;
                    const int prow = __builtin_amdgcn_readlane(e_row2, jk), pid = __builtin_amdgcn_readlane(e_id, jk);
                    if (i < n1) {
                        double val = PO_NEG_INF;
                        if (tq >= 0) {
                            const Ent e = *t2_entry(prow, tq);
                            if (e.tag == tag_of(pid, tq)) val = e.v[0];
                        }
                        sm.pst[r][k * RK_NY + s] = val;
                    }
                }
                if (k0 == 0 && want_seed && se.tag == tag_of(e_id, ws - 1)) self = se.v[0];
                rk_sync();
                const int k1 = min(n1max, k0 + RK_NY);
                if (use_board) {
                    // ---- the chains of this block go to the chain wave
                    const int nblk = fresh ? max(min(n1 - k0, RK_NY), 0) : 0;
                    const bool has = nblk > 0;
                    const unsigned long long hm = __ballot(has);
                    const int cnt = __popcll(hm);
                    if (cnt > 0) {
                        const int rank = __popcll(hm & ((1ull << lane) - 1ull));
                        if (has) {
                            RegJob j;
                            j.code = wave | (r << 4) | (myk << 5) | (sym << 8); j.t0 = ws + k0; j.n = nblk;
                            j.rowbase = (e_row2 * 2 + r) * R2; j.node = e_id; j.pad_ = 0; j.seed = self;
                            gsm.slot[wave][(posted_total + (unsigned)rank) % (unsigned)RK_JOBS].j = j;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        rk_sync();
                        if (lane == 0) ((volatile unsigned*)gsm.posted)[wave] = posted_total + (unsigned)cnt;
                        posted_total += (unsigned)cnt;
                        while ((int)(((volatile unsigned*)gsm.done)[wave] - posted_total) < 0) __builtin_amdgcn_s_sleep(1);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        if (has) {
                            const RegRes rr_ = gsm.slot[wave][(posted_total - (unsigned)cnt + (unsigned)rank) % (unsigned)RK_JOBS].r;
                            self = rr_.self;
                            mt = (rr_.mx >= mx) ? rr_.mt : mt;   // (later times win ties)
                            mx = po_vmax(mx, rr_.mx);
                            if (rr_.tr != INT_MIN) tr = rr_.tr;
                        }
                    }
                    continue;
                }
                // (the operands of an iteration are asked for one iteration ahead: a lone wave then waits for the LDS only
                //  inside logaddexp's own table lookups)
                double nya = 0.0, nyb = 0.0, npp = 0.0;
                if (fresh && k0 < n1) {
                    const double* yrow = yb_ + ((ws + k0) & (RK_NY - 1)) * RK_YC;
                    nya = yrow[sym]; nyb = yrow[A]; npp = ps_[0];
                }
#ifdef PO_REG_UNROLL2
#pragma unroll 2
#endif
                for (int k = k0; k < k1; ++k) {
                    if (fresh && k < n1) {
                        const int t = ws + k;
                        const double ya = nya, yb = nyb, pp = npp;
                        {   // (one past the end of the block: read, never used — the slots exist)
                            const double* yrow = yb_ + ((t + 1) & (RK_NY - 1)) * RK_YC;
                            nya = yrow[sym]; nyb = yrow[A]; npp = ps_[min(k - k0 + 1, RK_NY - 1)];
                        }
                        const double out = lae(pp + ya, self + yb);
#ifdef PO_RING_TRACE_NODE
                        if (pi == 0 && e_id == PO_RING_TRACE_NODE) printf("V %d %d %d %.17g %.17g %.17g P1\n", e_id, r, t, out, pp, self);
#endif
                        t2_write(e_row2, e_id, t, out);
                        if (out > self) tr = t;
                        self = out;
                        mt = (out >= mx) ? t : mt;
                        mx = po_vmax(mx, out);
                    }
                }
            }
            if (n1max == 0 && want_seed && se.tag == tag_of(e_id, ws - 1)) self = se.v[0];
            if (s < 8) {
                sm.pf0_t[r][s] = want_pf ? ws : -1;
                if (want_pf) sm.pf0[r][s] = (se.tag == tag_of(e_id, ws)) ? se.v[0] : PO_NEG_INF;
            }
            pf0_step = mstep;
            // the fresh lanes are ordinary continuing lanes now, ending at dr like everybody else: the run loop does the step
            if (fresh) {
                v_done = dr; v_self = self; v_fresh = 0;
                v_mx = mx; v_mt = mt; v_td = max(ws, tr);
            }
            KT(2);
            if constexpr (COUNT) {
                int tot = fresh ? n1 : 0;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) tot += __shfl_xor(tot, off);
                cnt_x += (unsigned)tot;
            }
            return true;
        };

        // ---------------------------------------------------------------- the next element table
        // Given the new beam (sel[0 .. nbn): slots of the present table, in rank order), builds the table of the next main
        // step: expansion of the beam nodes (BeamSearch.h:342-360: node ids in beam order), children slots, which old
        // element continues in which new slot (its carried maxima and times move with it).
        auto rebuild = [&](int nbn, int nu, int nv, int nce, int nre) {
            const int nbo = nb;
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
            int n_fc = sm.f_fc[srcb], n_crow2 = sm.f_crow2[srcb], n_par = sm.f_par[srcb];
            int n_gpar = sm.f_gpar[srcb], n_prow2 = sm.f_prow2[srcb], n_depth = sm.f_depth[srcb];
            KT2(32);
            // ---- every old element marks its row group with the times it has written there
            if (live && v_fresh == 0) atomicMax(&g_hi[e_row2 >> 2], v_done);
            // ---- B. expansion of the new beam nodes
            KC(20, __ballot(rb && n_fc == -2) != 0ull ? 1 : 0);
            if (__builtin_expect(rb && n_fc == -2, 0)) { n_fc = afc[n_id]; n_crow2 = acrow[n_id]; }   // a node whose parent re-entered: the arena knows
            rk_sync();
            bool isnew = false, need_group = false;
            if (rb) {
                isnew = n_fc < 0;
                need_group = isnew || n_crow2 < 0 || n_crow2 >= NG || sm.g_owner[n_crow2] != n_id;   // (old rows recycled: all dead)
            }
            {
                const unsigned bn = (unsigned)__ballot(isnew && r == 0);
                if (isnew) {
                    n_fc = next_id + A * __popc(bn & ((1u << s) - 1u));
                    if (r == 0) afc[n_id] = n_fc;
                }
                next_id += A * __popc(bn);
                if (rb && !need_group) { atomicMax(&sm.g_hi0[n_crow2], nce); atomicMax(&sm.g_hi1[n_crow2], nre); }
                rk_sync();
                unsigned hg = (unsigned)__ballot(need_group && r == 0);
                while (hg != 0) {   // (uniform: every lane walks the group table, lane 0 writes)
                    const int jj = __builtin_ctz(hg);
                    hg &= hg - 1;
                    const int owner = __builtin_amdgcn_readlane(n_id, jj);
                    int gg = -1;
                    for (int tries = 0; tries < NG; ++tries) {
                        const int cc = gcur;
                        gcur = (gcur + 1 == NG) ? 0 : gcur + 1;
                        if (sm.g_owner[cc] < 0 || (sm.g_hi0[cc] <= nu - 1 && sm.g_hi1[cc] <= nv - 1)) { gg = cc; break; }
                    }
                    if (gg < 0) { st = PO_E_NOMEM; gg = 0; }
                    rk_sync();   // (every lane has walked the table before lane 0 changes it)
                    if (lane == 0) { sm.g_owner[gg] = owner; sm.g_hi0[gg] = nce; sm.g_hi1[gg] = nre; acrow[owner] = gg; }
                    if (s == jj) n_crow2 = gg;
                    rk_sync();
                }
            }
            KT2(33);
            // ---- C. children slots take their parent's (new) fields
            const int p_id = __shfl(n_id, hb | j), p_fc = __shfl(n_fc, hb | j), p_crow2 = __shfl(n_crow2, hb | j);
            const int p_sym = __shfl(n_sym, hb | j), p_par = __shfl(n_par, hb | j), p_row2 = __shfl(n_row2, hb | j);
            const int p_depth = __shfl(n_depth, hb | j);
            const bool p_isnew = __shfl((int)isnew, hb | j) != 0;
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
                const int oa = sm.f_alias[max(src, 0)];
                if (!rb && src >= nbo && oa >= 0) src = oa;
            }
            const bool nlive = (rb || (rc && n_alias < 0));
            KT2(34);
            // ---- F. the lanes take their new identity
            const int gsrc = hb | max(src, 0);
            const int g_done = __shfl(v_done, gsrc), g_fresh = __shfl(v_fresh, gsrc), g_mt = __shfl(v_mt, gsrc), g_td = __shfl(v_td, gsrc);
            const double g_self = __shfl(v_self, gsrc), g_mx = __shfl(v_mx, gsrc);
            const int g_fc = sm.f_fc[max(src, 0)], g_crow2 = sm.f_crow2[max(src, 0)];
            // the last value of the node's parent as the old table knew it: of the parent's lane if it was an element,
            // else what was captured when it stopped being one
            double c_val;
            int c_t;
            const int op = __shfl(e_ps, gsrc);                     // the parent's slot in the old table (or ROOT / FROZEN)
            {
                const int opl = hb | max(op, 0);
                const int o_done = __shfl(v_done, opl), o_fresh = __shfl(v_fresh, opl);
                const double o_last = __shfl(v_self, opl);
                const double q_val = __shfl(fz_val, gsrc);
                const int q_t = __shfl(fz_t, gsrc);
                c_val = (op >= 0) ? o_last : q_val;
                c_t = (op >= 0) ? ((o_fresh == 0) ? o_done - 1 : INT_MAX) : q_t;
            }
            e_id = n_id; e_row2 = n_row2; e_sym = n_sym;
            if (rc && src >= 0) { n_fc = g_fc; n_crow2 = g_crow2; }   // a continuing child keeps what is known about its own children
            rk_sync();   // (every lane has read the old table's fields)
            if (r == 0) {
                sm.f_fc[s] = n_fc; sm.f_crow2[s] = n_crow2; sm.f_par[s] = n_par; sm.f_gpar[s] = n_gpar; sm.f_prow2[s] = n_prow2;
                sm.f_depth[s] = n_depth; sm.f_alias[s] = rc ? n_alias : -1;
            }
            live = nlive;
            fz_val = c_val; fz_t = (nlive && src >= 0) ? c_t : INT_MAX;
            if (nlive && src >= 0) {
                v_done = g_done; v_fresh = g_fresh; v_mt = g_mt; v_td = g_td; v_self = g_self; v_mx = g_mx;
            } else {
                v_done = RK_FRESH; v_fresh = (rc && p_isnew) ? 2 : 1;
                v_self = PO_NEG_INF; v_mx = PO_NEG_INF; v_mt = -1; v_td = 0;
            }
            KT2(35);
            // ---- the parent slot of the beam nodes: a beam node, a child of a beam node, the root, or none (frozen)
            nb = nbn; ne = nen;
            {   // (wave-uniform loops: v_readlane)
                if (rb) n_ps = (n_par == 0) ? PS_ROOT : PS_FROZEN;
                for (int i = 0; i < nbn; ++i) {
                    const int bid = __builtin_amdgcn_readlane(e_id, i);
                    if (rb && n_par != 0 && bid == n_par) n_ps = i;
                }
                const bool fz = rb && n_ps == PS_FROZEN;
                for (int i = 0; i < nbn; ++i) {
                    const int bid = __builtin_amdgcn_readlane(e_id, i);
                    if (fz && bid == n_gpar) n_ps = nbn + A * i + sym_plast(e_sym);
                }
            }
            e_ps = n_ps;
            KT2(36);
            // ---- G. a frozen parent that is an element again.  A beam node whose parent had left the table computed its
            // newest values against "absent" (-inf beyond the parent's last time).  When the grandparent enters the beam the
            // parent comes back as one of its children, computes its whole window — times it never had — and the
            // reference's step, which recomputes every window in full, then gives the node (and everything below it)
            // different values over the part of the window it already had.  Those elements go back to the window start;
            // the part before it is what both sides keep.  (Everybody else's inputs are unchanged: skipping their old
            // times rewrites nothing.)
            {
                bool rew = rb && nlive && src >= 0 && op == PS_FROZEN && n_ps >= 0;
                if (__builtin_expect(__ballot(rew) != 0ull, 0)) {
                    for (int it = 0; it < W; ++it) {   // ... and their descendants in the table, generation by generation
                        const bool prew = __shfl((int)rew, hb | max(e_ps, 0)) != 0;
                        if (live && e_ps >= 0 && prew) rew = true;
                    }
                    const int wsn = r ? nv : nu;
                    if (rew && live && v_fresh == 0 && v_done > wsn) {
                        v_self = read_own(wsn - 1);
                        v_done = wsn;
                    }
                }
            }
            tbl_fresh = __ballot(live && v_fresh != 0) != 0ull;
            have_children = true;
            rk_sync();
            KT(8);
        };

        // the first table: the root's children are the beam, their children come from the first expansion
        rebuild(A, rec.x, rec.y, rec.z, rec.w);

        // ---------------------------------------------------------------- the diagonal walk (BeamSearch.h:300-393)
        KT(9);
        while (st == PO_OK && mstep < nmain) {
            int u = rec.x, v = rec.y, ce = rec.z, re = rec.w;
            double sc = PO_NEG_INF;
            bool viol = false, run_viol = false;
            KT(6);
            // ---- a RUN of main steps on the table as it stands.  After a step that kept the set of beam nodes, with every
            // live lane's values ending at the same time and nothing to catch up, the next step is: the new times of the two
            // windows (often none on a read: the envelope's window ends move a base at a time) in lockstep, the window maxima
            // from what is carried, the score, the one comparison per child.  The run ends at the first step that is not of
            // this kind (it is then done below) or that changes the beam (it is ranked below).
            if (__builtin_expect(!tbl_fresh && !tbl_uneven && nb == W && __ballot(live && e_ps == PS_ROOT) == 0ull, 1)) {
                const int sym = sym_last(e_sym);
                const int plane = (e_ps >= 0) ? (hb | e_ps) : lane;
                const double* const yb_ = &sm.ybuf[r][0][0];
                Ent pf_e; pf_e.tag = 0ull; pf_e.v[0] = 0.0;   // the entry requested at the end of the previous step of this run
                int pf_t = -1;
                if (pf0_step == mstep && s < 8) {   // ... or with the staging of the new elements' step just before this run
                    pf_t = sm.pf0_t[r][s];
                    if (pf_t >= 0) { pf_e.tag = tag_of(e_id, pf_t); pf_e.v[0] = sm.pf0[r][s]; }
                }
                for (;;) {
                    u = rec.x; v = rec.y; ce = rec.z; re = rec.w;
                    const int d0 = __builtin_amdgcn_readlane(v_done, 0), d1 = __builtin_amdgcn_readlane(v_done, 32);
#ifdef PO_EMU_DEBUG
                    if ((!(u <= d0 && d0 <= ce && v <= d1 && d1 <= re)) && lane == 0) printf("RUNBRK u %d d0 %d ce %d v %d d1 %d re %d\n", u, d0, ce, v, d1, re);
#endif
                    if (!(u <= d0 && d0 <= ce && v <= d1 && d1 <= re) || mstep + 1 >= nmain) break;
                    const int ws = r ? v : u, we = r ? re : ce, dr = r ? d1 : d0;
                    const bool part2 = live && dr < we;
                    // (a frozen parent's older values would have to come from the store: only asked when there are new times)
                    if (__builtin_expect((ce > d0 || re > d1) && __ballot(part2 && e_ps == PS_FROZEN && dr - 1 < fz_t) != 0ull, 0)) break;
                    // ---- the carried part [ws, dr) of the window: its maximum is what the previous step left while that time
                    // is inside the window; else the stored values are looked at (the one most steps need — a decaying
                    // element's value at the window start — was requested a step ago: pf)
                    // A CHILD keeps its old maximum when that time has left the window: the maximum over an older, larger
                    // window is an upper bound, and a child only has to stay below the smallest beam score — its exact
                    // maximum is looked up below, if the bound does not settle that.  (40 of 50 lanes never ask the store.)
#ifdef PO_REG_TIMING2
                    KT2(22);
#ifndef PO_EMU
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (timing builds: what the drain of the wave's memory queue costs here)
#endif
                    KT2(23);
#endif
                    const bool has_c = live && dr > ws;
                    const bool child = s >= nb;
                    const bool bnd = has_c && child && !(v_mx == PO_NEG_INF || v_mt >= ws);
                    double mx = PO_NEG_INF, cmx = PO_NEG_INF, self = v_self;
                    int mt = -1, cmt = -1, td = has_c ? v_td : ws, tr = INT_MIN;
                    const double pf_val = (pf_t >= 0 && pf_e.tag == tag_of(e_id, pf_t)) ? pf_e.v[0] : PO_NEG_INF;
                    bool rsc = false;
                    if (has_c) {
                        if (v_mx == PO_NEG_INF || v_mt >= ws || child) { cmx = v_mx; cmt = v_mt; }
                        else rsc = carried_one(ws, cmx, cmt, td, pf_t, pf_val);
                    }
                    rescan_wave(rsc, ws, dr, cmx, cmt, td);
#ifdef PO_REG_TIMING2
                    {   // which way the carried maxima went (counts of steps; the longest rescan of the step)
                        const bool need = has_c && !child && !(v_mx == PO_NEG_INF || v_mt >= ws);
                        const bool one = need && v_td <= ws;
                        const int len = (need && !one) ? min(v_td + 1, dr) - ws : 0;
                        int lm = len;
                        for (int off = 32; off >= 1; off >>= 1) lm = max(lm, __shfl_xor(lm, off));
                        KC(37, __ballot(one && pf_t != ws) != 0ull ? 1 : 0);
                        KC(38, lm > 0 ? 1 : 0);
                        KC(39, lm);
                        KC(30, __ballot(one && pf_t == ws) != 0ull ? 1 : 0);
                    }
#endif
                    KT2(24);
                    KT2(25);
                    // ---- the new times [dr, we), everybody in lockstep: the parent's previous value comes from its lane
                    const int n2 = we - dr;   // (half-uniform, >= 0)
                    const int n2max = max(ce - d0, re - d1);
                    // (a frozen parent's captured value can only be asked for by the first new time — the test above — : one select
                    //  per iteration instead of two compares and two)
                    const bool fzl = e_ps < 0;
                    double pp_fz = (fzl && dr - 1 == fz_t) ? fz_val : PO_NEG_INF;
                    for (int k0 = 0; k0 < n2max; k0 += RK_NY) {   // (blocks of RK_NY times: the y rows are loaded between the loops)
                        {
                            const int lo = dr + k0, hi = min(lo + RK_NY, we);
                            rk_sync();
                            if (__builtin_expect(hi > lo && !(lo >= yhi - RK_NY && hi <= yhi), 0)) { y_reload(lo); yhi = lo + RK_NY; }
                            rk_sync();
                        }
                        KT2(26);
                        const int k1 = min(n2max, k0 + RK_NY);
                        for (int k = k0; k < k1; ++k) {
                            const int t = dr + k;
                            const double ps_self = __shfl(self, plane);
                            if (live && k < n2) {
                                const double* yrow = yb_ + (t & (RK_NY - 1)) * RK_YC;
                                const double ya = yrow[sym], yb = yrow[A];
                                const double pp = fzl ? pp_fz : ps_self;
                                pp_fz = PO_NEG_INF;
                                const double out = lae(pp + ya, self + yb);
#ifdef PO_RING_TRACE_NODE
                                if (pi == 0 && e_id == PO_RING_TRACE_NODE) printf("V %d %d %d %.17g %.17g %.17g RUN ps %d fzt %d\n", e_id, r, t, out, pp, self, e_ps, fz_t);
#endif
                                t2_write(e_row2, e_id, t, out);
                                if (out > self) tr = t;
                                self = out;
                                mt = (out >= mx) ? t : mt;
                                mx = po_vmax(mx, out);
                            }
                        }
                        KT2(27);
                    }
                    if (part2) { v_done = we; v_self = self; }
                    const double nmx = mx;   // the maximum over the new times alone
                    const int nmt = mt;
                    if (has_c && !(mx >= cmx)) { mx = cmx; mt = cmt; }   // (new values, later in time, win ties)
                    if (live) { v_mx = mx; v_mt = mt; v_td = max(td, tr); }
                    smx = live ? mx : PO_NEG_INF;
                    if constexpr (COUNT) {
                        cnt_ref += (unsigned)(ne * ((ce - u) + (re - v)));
                        cnt_x += (unsigned)(__popcll(__ballot(live && r == 0)) * (ce - d0) + __popcll(__ballot(live && r == 1)) * (re - d1));
                    }
                    sc = po_sum32(smx);
                    const double scmin = rk_row0_min(sc, nb, lane);
                    viol = live && child && !(scmin > sc);
                    // a child that reaches the smallest beam score on a bound: its exact maximum now (the stored values are
                    // looked at), then the score and the test again — the decision is the one exact maxima give
                    const bool hot = viol && bnd && !(nmx >= cmx);
                    if (__builtin_expect(__ballot(hot) != 0ull, 0)) {
                        double cx = PO_NEG_INF;
                        int ct = -1, td2 = td;
                        const bool rs2 = hot && carried_one(ws, cx, ct, td2, pf_t, pf_val);
                        rescan_wave(rs2, ws, dr, cx, ct, td2);
                        if (hot) {
                            const bool keep = (nmx >= cx);
                            v_mx = keep ? nmx : cx; v_mt = keep ? nmt : ct; v_td = max(td2, tr);
                            smx = v_mx;
                        }
                        sc = po_sum32(smx);
                        viol = live && child && !(scmin > sc);
                        KC(31, 1);
                    }
#ifdef PO_RING_TRACE
                    if (pi == 0 && live && r == 0) printf("T %d %d %d %.17g\n", u, v, e_id, sc);
#endif
                    KT2(28);
                    up = u; vp = v;
                    mstep++;
                    if (__builtin_expect((mstep & 63) == 0, 0)) {
                        rcur = rnxt;
                        rnxt = rec_load(mstep + 64 + lane);
                    }
                    rec = rec_at(min(mstep, nmain - 1));
                    {   // the stored value the next step's carried maximum will ask for, if any, requested now: the beam lanes,
                        // and a child whose bound has just had to be made exact (it will be again)
                        const int wsn = r ? rec.y : rec.x;
                        pf_t = -1;
                        if (live && (!child || hot) && v_done > wsn && v_mx != PO_NEG_INF && v_mt < wsn) { pf_t = wsn; pf_e = *t2_entry(e_row2, wsn); }
                    }
#ifdef PO_EMU_DEBUG
                    if (lane == 0) printf("STEP run\n");
#endif
                    KC(12, 1); KC(19, n2max);
                    KT2(29);
                    if (__ballot(viol) != 0ull) { run_viol = true; break; }
                }
                KT(0);
            }
            if (!run_viol) {
            u = rec.x; v = rec.y; ce = rec.z; re = rec.w;
            // ---- catch-up steps between the previous main step and this one (:314-336): only the beam nodes, one time
            // at a time; a time the last main step's window covered is a no-op (the bits are there)
            {
                const int nbe = min(W, nb);
                const int d0 = __builtin_amdgcn_readlane(v_done, 0), d1 = __builtin_amdgcn_readlane(v_done, 32);
                if constexpr (COUNT) cnt_ref += (unsigned)((max(u - up - 1, 0) + max(v - vp - 1, 0)) * nbe);
                if (__builtin_expect(u - 1 >= max(up + 1, d0), 0)) { scan(false, up + 1, u, 0, 0, nbe); tbl_uneven = true; }
                if (__builtin_expect(v - 1 >= max(vp + 1, d1), 0)) { scan(false, 0, 0, vp + 1, v, nbe); tbl_uneven = true; }
            }
            // ---- MAIN step at (u, v): windows [u, ce) x [v, re)  (:342-375)
            // (new elements first: their windows up to where everybody else stands — then the step is an ordinary one)
            if (__builtin_expect(tbl_fresh && !tbl_uneven && scan_new(u, ce, v, re), 1)) {
#ifdef PO_EMU_DEBUG
                if (lane == 0) printf("STEP new\n");
#endif
                tbl_fresh = false;
                continue;
            }
#ifdef PO_EMU_DEBUG
            if (lane == 0) printf("STEP general fresh %d uneven %d w %d %d\n", (int)tbl_fresh, (int)tbl_uneven, ce - u, re - v);
#endif
            scan(true, u, ce, v, re, 32);
            tbl_fresh = false; tbl_uneven = false;
            if constexpr (COUNT) cnt_ref += (unsigned)(ne * ((ce - u) + (re - v)));
            // node_greater_max_sym: max over read 0's window + max over read 1's
            sc = po_sum32(smx);
#ifdef PO_RING_TRACE   // debugging builds only: every candidate's score before the prune
            if (pi == 0 && live && r == 0) printf("T %d %d %d %.17g\n", u, v, e_id, sc);
#endif
            // ---- prune (Beam.h:93-108).  Most steps keep the SET of beam nodes: iff every child is strictly below the smallest
            // beam score (a child AT it, ties included, goes the full way, as partial_sort decides them).  The order of the
            // beam nodes among themselves is not looked at: nothing is created while the set stays (every beam node has its
            // children), ties are decided on node ids, and the order matters only where nodes are created — the step in which
            // the set changes ranks everybody — and for the label: the last main step is always ranked.
            viol = (nb != W) || (mstep + 1 == nmain);
            if (!viol) {   // (wave-uniform: nb == W and not the last step)
                const double scmin = rk_row0_min(sc, nb, lane);
                if (live && s >= nb) viol = !(scmin > sc);
            }
            up = u; vp = v;
            mstep++;
            if (__builtin_expect((mstep & 63) == 0, 0)) {   // the next batch becomes the current one, the one after it is requested
                rcur = rnxt;
                rnxt = rec_load(mstep + 64 + lane);
            }
            rec = rec_at(min(mstep, nmain - 1));
            KT(6);
            if (__ballot(viol) == 0ull) continue;
            }   // (!run_viol)
            const bool cand = live;
            // ---- full ranking among the distinct candidates
            const unsigned cm = (unsigned)__ballot(cand && r == 0);
            const int ncand = __popc(cm);
            // Only the beam nodes and the children that reach the smallest beam score can be among the W best (every
            // other child has W candidates above it), and nothing outside that set outranks a member of it: the ranks
            // are taken within it (a handful of candidates instead of W * (A + 1)).
            unsigned smask = cm;
            if (nb == W) {
                const double thr = rk_row0_min(sc, nb, lane);
                smask = (unsigned)__ballot(cand && r == 0 && (s < nb || sc >= thr));
            }
            int rank = 0, neq = 0;
            for (unsigned mm = smask; mm != 0u; mm &= mm - 1u) {
                const int o = __builtin_ctz(mm);
                const double so = rk_readlane_d(sc, o);
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
            if (__builtin_expect(__ballot(cand && neq > 1 && rank < W) != 0ull, 0)) {
                // exact ties reaching into the beam: what libstdc++'s partial_sort / sort leave on the candidates in
                // creation order (po_device.h), replayed by one lane
                int pos = 0;   // (the replay runs over ALL candidates in creation order)
                for (int o = 0; o < ne; ++o) {
                    const int io = __builtin_amdgcn_readlane(e_id, o);
                    pos += (int)((cm >> o) & 1u) & ((io < e_id) ? 1 : 0);
                }
                if (cand && r == 0) { sm.ord[pos] = s; sm.csc[s] = sc; }
                rk_sync();
                if (lane == 0) {
                    const double* cp = sm.csc;
                    po_stl_prune<6>(sm.ord, ncand, W, [&](int slot) { return cp[slot]; });
                }
                rk_sync();
#pragma unroll
                for (int jx = 0; jx < 6; ++jx) sel[jx] = (jx < nbn) ? sm.ord[jx] : 0;
                rk_sync();
            }
            KT(7);
            rebuild(nbn, rec.x, rec.y, rec.z, rec.w);
        }
        KT(6);
        // ---------------------------------------------------------------- label of the top node
        if (st == PO_E_NOMEM && lane == 0) {   // out of row groups (or a window end moved back): beam2d_kernel takes the pair
            a.meta[pi] = make_int2(PO_OK, X2_DEFERRED);
            a.queue[16] = 1;
        } else if (lane == 0) {
            int nout = 0;
            if (st == PO_OK) {
                int node = e_id;
                nout = sm.f_depth[0];
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
        if (COUNT && lane == 0) { sm.nupd += cnt_ref; sm.nupd_x += cnt_x; }
        rk_sync();
        KT(9);
    }
#ifdef PO_REG_TIMING
    if (lane == 0 && a.dbg && slotid == 0)
        for (int i = 0; i < 40; ++i) a.dbg[i] = tk[i];
#endif
    if (lane == 0) {   // the next launch on this workspace continues from here
        unsigned long long* stp = a.wgstate + 2 * (size_t)slotid;
        stp[0] = a.magic ^ (unsigned long long)slotid;
        stp[1] = (unsigned long long)epoch;
        if (COUNT && a.upd_count) { atomicAdd(a.upd_count, sm.nupd); atomicAdd(a.upd_count + 1, sm.nupd_x); }
        if (NPW > 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); atomicAdd(&gsm.exited, 1); }
    }
}

// The board variant: 7 pair waves + 1 chain wave per workgroup (512 threads; two workgroups per CU: 14 pair slots).
constexpr int RK_BOARD_NPW = 7;

// pair slots per CU (registers and LDS decide): 16 one-wave workgroups, or 2 x 7 pair waves with the job board
extern "C" int po_reg_slots_per_cu(int board) {
#ifdef PO_EMU
    return board ? 2 * RK_BOARD_NPW : 16;
#else
    static PoPerDeviceCache<2> per_cu;
    return per_cu.get(board ? 1 : 0, [board] {
        int nblk = 0;
        if (board) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)beam2d_reg_kernel<RK_BOARD_NPW, true>, 64 * (RK_BOARD_NPW + 1), 0) != hipSuccess || nblk <= 0) nblk = 2;
            if (getenv("PO_DEBUG_OCC")) fprintf(stderr, "[po] beam2d_reg_kernel<%d> (job board): %d resident workgroups per CU, %zu B of LDS\n", RK_BOARD_NPW, nblk, sizeof(RegGroup<RK_BOARD_NPW>));
            return nblk * RK_BOARD_NPW;
        }
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)beam2d_reg_kernel<1, false>, 64, 0) != hipSuccess || nblk <= 0) nblk = 16;
        if (const char* e = getenv("PO_REG_PER_CU")) { const int v = atoi(e); if (v > 0 && v < nblk) nblk = v; }
        if (getenv("PO_DEBUG_OCC")) fprintf(stderr, "[po] beam2d_reg_kernel: %d resident workgroups per CU, %zu B of LDS\n", nblk, sizeof(RegGroup<1>));
        return nblk;
    });
#endif
}
extern "C" int po_reg_blocks_per_cu() { return po_reg_slots_per_cu(0); }
extern "C" int po_reg_max_elements() { return 32; }
extern "C" int po_reg_ngl() { return RK_NGL; }
// `slots` pair slots (each with its own store slice and arena); board != 0: workgroups of RK_BOARD_NPW pair waves + a chain wave
extern "C" void po_reg_launch(const void* x2args, int slots, int board, hipStream_t stream) {
    X2Args a = *(const X2Args*)x2args;
    a.reg_slots = slots;
    a.reg_board = board ? 1 : 0;
#ifdef PO_REG_TIMING
    static long long* dbg = nullptr;
    if (!dbg) { (void)hipMalloc((void**)&dbg, 40 * sizeof(long long)); }
    (void)hipMemsetAsync(dbg, 0, 40 * sizeof(long long), stream);
    a.dbg = dbg;
#endif
    if (board) {
        const int wgs = (slots + RK_BOARD_NPW - 1) / RK_BOARD_NPW;
        hipLaunchKernelGGL((beam2d_reg_kernel<RK_BOARD_NPW, true>), dim3(wgs), dim3(64 * (RK_BOARD_NPW + 1)), 0, stream, a);
    } else {
        if (a.upd_count != nullptr) hipLaunchKernelGGL((beam2d_reg_kernel<1, true>), dim3(slots), dim3(64), 0, stream, a);
        else hipLaunchKernelGGL((beam2d_reg_kernel<1, false>), dim3(slots), dim3(64), 0, stream, a);
    }
#ifdef PO_REG_TIMING
    {
        long long h[40];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "[po_reg_timing] pair slot 0, 10 ns ticks\n");
        fprintf(stderr, "   run loop %lld (%lld steps, %lld new-time iterations)\n", h[0], h[12], h[19]);
        fprintf(stderr, "   steps with new elements: %lld steps; staging + carried maxima %lld, phase 1 %lld (%lld iterations), phase 2 + state %lld (%lld iterations)\n",
                h[13], h[1], h[2], h[14], h[3], h[15]);
        fprintf(stderr, "   general scans: main %lld ticks (%lld), catch-up %lld ticks (%lld); iterations %lld\n", h[4], h[16], h[5], h[17], h[18]);
        fprintf(stderr, "   step top + score + prune test %lld, ranking %lld, rebuild %lld, pair setup + label %lld\n", h[6], h[7], h[8], h[9]);
#ifdef PO_REG_TIMING2
        fprintf(stderr, "   inside the run loop: top + carried maxima %lld, prefetch %lld, y rows + syncs %lld, new-time iterations %lld, maxima + score + test %lld, record advance %lld (rest %lld)\n",
                h[24], h[25], h[26], h[27], h[28], h[29], h[0]);
        fprintf(stderr, "   run loop top before the carried maxima %lld, waiting for the memory queue to drain there %lld\n", h[22], h[23]);
        fprintf(stderr, "   table builds that ask the arena for a node's children: %lld; window rescans: %lld lanes in %lld calls\n", h[20], h[21], h[10]);
        fprintf(stderr, "   steps in which a child's bound had to be made exact: %lld\n", h[31]);
        fprintf(stderr, "   carried maxima in the run loop: steps with a prefetched entry used %lld, with a value asked for on the spot %lld, with a rescan %lld (longest rescans summed: %lld reads)\n", h[30], h[37], h[38], h[39]);
        fprintf(stderr, "   inside the table build: A fields %lld, B expansion + groups %lld, C + D children / sources %lld, F identity %lld, parent slots %lld (rest: G + end %lld)\n",
                h[32], h[33], h[34], h[35], h[36], h[8]);
#endif
    }
#endif
}
