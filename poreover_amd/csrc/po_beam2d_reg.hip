// Pair (2-D) beam search, method "row_col" with a monotone envelope — the REGISTER-STATE kernel: every tree model
// ("ctc", "ctc_merge_repeats", "ctc_flipflop") and every beam width up to 12, one wave per pair.
//
// Replaces: decoding_cpp.cpp_beam_search_2d (decoding_cpp.pyx:107-139) -> beam_search_2d_by_row_col (BeamSearch.h:262-397,
// dispatch :420-427) over PoreOverPrefixTree2D / BonitoPrefixTree2D / FlipFlopPrefixTree2D (PrefixTree.h:492-533, :665-706,
// :576-634) with Beam<..., node_greater_max_sym> (Beam.h:35-38,93-108).  beam2d_kernel (po_beam2d.hip) stays the general
// form (method row, no envelope, W > 12) and decodes the pairs this kernel hands on.
//
// The design (DESIGN.md §3.3 has the measurements and the history):
//   * VALUE STORE (HBM, L2-resident in practice): the reference's per-node maps (PrefixTree.h:76-145), entry = the node's K
//     values at one time (8 or 24 bytes, no tag), ring rows of R entries, rows in groups of four per parent.  Whether a
//     (node, time) is present — probability_at() answers -inf otherwise — follows from where the node's values end: an
//     element's lane knows, a row header remembers it for nodes that are no elements.  Written once per computed
//     (node, read, time).
//   * REGISTERS: a lane's element (ids, rows, parent slot), where its values end (v_done), its last values (v_self),
//     the carried window maximum (value, time, last rise).  Within a scan a child takes its parent's previous values
//     from the parent's LANE (ds_bpermute) — no exchange buffer, no LDS ring, no fence per iteration.
//   * LDS (10 - 13 KB): 32 y rows per read, the staged windows of the parents of a step's new elements, the row group
//     table, the element-table fields only the table build reads, the logaddexp tables.
//   * RUN loop: consecutive main steps that keep the set of beam nodes are one tight loop (the new times of the two
//     windows in lockstep, the carried maxima, the score, one comparison per child); the window maximum of a decaying
//     element needs one stored value per step, requested a step ahead.
//   * NEW ELEMENTS (a node entered the beam: its children compute their whole windows, ~ 20 dependent update_prob
//     iterations on a few lanes): their parent's stored window is staged into LDS in one memory round trip, then the
//     chain runs on LDS and registers only; everybody else continues where it was (the stored bits of the part they
//     already have would be rewritten unchanged: every input is unchanged).
//   * TWO LANE LAYOUTS, one source (template parameter NR = reads a lane serves):
//       NR = 1: lane = (read, element slot), 32 slots per read — W * (A + 1) <= 32, i.e. W <= 6: both reads' windows
//               advance side by side in the two halves of the wave;
//       NR = 2: lane = element slot, 64 slots — W <= 12: the same code runs read 0, then read 1, the per-read state
//               twice in registers; scores are sums within the lane.
// The walk comes precomputed (beam2d_walk_kernel), envelope checks and R from beam2d_prepass_kernel; pairs this kernel
// cannot hold (row groups exhausted, windows beyond 254 frames, non-monotone envelopes) go to beam2d_kernel through the
// meta word.  Results are bit-identical to beam2d_kernel's: the same arithmetic in the same order within every chain.
#include <climits>
#include <type_traits>

#define PO_LAE_EARLY_TABLE 1   // (po_device.h: the exp table entry is requested before the polynomial — a lone wave's chain is latency)
#define PO_LAE_TRIM 1          // (... two instructions fewer: -|x1 - x2| through source modifiers, the exponent add in two)
#define PO_LAE_BRANCHLESS 1    // (... and exp's small-argument test is a select, not a branch)
#include "po_beam2d_common.h"
#include "po_host.h"

namespace {

constexpr int RK_NY = 32;     // y rows per read resident in LDS
#ifndef PO_REG_NGL
#define PO_REG_NGL 96
#endif
#ifndef PO_REG_PS
#define PO_REG_PS 5   // (round 6, late: 4 -> 5 — at W = 5 every step's parents are staged; 7 % of the steps with new elements had five and
#endif                //  took the general scan, 19 us each at full load.  The 512 B of LDS come from `tie` sharing the staging buffer's bytes)
#ifndef PO_REG_PS_MR
#define PO_REG_PS_MR 3      // ... the merge-repeats model (three values, two of them staged: 1 KB of LDS per parent)
#endif
#ifndef PO_REG_PS_FF
#define PO_REG_PS_FF 3      // ... flip-flop
#endif
#ifndef PO_REG_STAGE_NB
#define PO_REG_STAGE_NB 2   // staged parents whose entries a 32-slot step asks for in ONE memory round trip (the others: one each)
#endif
constexpr int RK_NB = PO_REG_STAGE_NB;
constexpr int RK_FRESH = INT_MIN / 2;

// What the tree model and the lane layout decide (update_prob: PrefixTree.h:518-531 ctc, :690-704 merge repeats, :600-632
// flip-flop; the recurrences themselves are po_device.h::po_update, shared with every other beam kernel):
template <int MODEL, int NR>
struct RegCfg {
    static constexpr int K = (MODEL == PO_MODEL_CTC) ? 1 : 3;    // values per (node, read, time): alpha | alpha, gap, no_gap | alpha, flip, flop
    static constexpr int KP = (K == 1) ? 1 : 2;                   // how many of its PARENT's values an update reads ...
    static constexpr int PC0 = (MODEL == PO_MODEL_FLIPFLOP) ? 1 : 0;   // ... and which: {alpha} | {alpha, gap} | {flip, flop}
    static constexpr int YC = (MODEL == PO_MODEL_FLIPFLOP) ? 8 : 5;    // doubles per y row (A + 1 <= 5, or 2 A <= 8)
    static constexpr int EB = 8 * K;                              // bytes of a store entry: the values, nothing else (round 5)
    static constexpr int NS = (NR == 1) ? 32 : 64;                // element slots: W * (A + 1) <= NS
    static constexpr int WS = (NR == 1) ? 6 : 12;                 // widest beam
    static constexpr int NGL = (NR == 1) ? PO_REG_NGL : 2 * PO_REG_NGL;   // row groups tracked per pair
    // parents whose stored window one step can stage for its new elements (a step with more goes the general way)
    // (64 slots: ten — the reads go one after the other there, both use ONE staging buffer, and a block's 32 times leave half
    //  the read's 64 lanes free to ask for a second parent: at W = 10 a third of the steps with new elements have five to
    //  ten parents, and all of them went the general way)
    static constexpr int PS = (NR == 2) ? 10 : ((K == 1) ? PO_REG_PS : ((MODEL == PO_MODEL_FLIPFLOP) ? PO_REG_PS_FF : PO_REG_PS_MR));
    static constexpr int PF0N = (NR == 1) ? 8 : 16;               // beam slots whose window-start value a new-element step fetches ahead
    // waves per SIMD the register budget is set for (128 / 168 / 256 VGPRs)
#ifndef PO_REG_WAVES_K1W
#define PO_REG_WAVES_K1W 3   // <ctc, 64 slots>
#endif
#ifndef PO_REG_WAVES_K3N
#define PO_REG_WAVES_K3N 3   // <three values, 32 slots>
#endif
#ifndef PO_REG_WAVES_K1N
#define PO_REG_WAVES_K1N 4   // <ctc, 32 slots>: 128 VGPRs (5 = 96 VGPRs needs PO_REG_WPG_K1N=4 for the LDS: measured slower, below)
#endif
#ifndef PO_REG_WPG_K1N
#define PO_REG_WPG_K1N 1     // pair waves per workgroup (4: they share ONE copy of the logaddexp tables)
#endif
    // pair waves per workgroup.  The waves of a workgroup have nothing to do with each other but the logaddexp tables at the
    // front of the LDS block (2.5 KB).  Round 6 measured the kernel's time to be per-wave latency (12 / 14 / 16 waves per CU:
    // 73 / 59 / 52 ms per 10 000 pairs), so a fifth wave per SIMD was built: four waves on ONE copy of the tables are 7.9 KB each
    // (20 per CU fit) at 96 VGPRs — and the 96-register kernel is 23 % slower per wave (70 spilled registers against 33; at 16
    // waves per CU 63.8 ms, at 20 57.2 ms against 52.0 ms for 16 waves of 128 registers: profiles/r06_ab_occupancy.txt).  The
    // form stays selectable (-DPO_REG_WPG_K1N=4 -DPO_REG_WAVES_K1N=5); the default is one wave per workgroup.
    static constexpr int WPG = (K == 1 && NR == 1) ? PO_REG_WPG_K1N : 1;
    static constexpr int WAVES = (K == 1 && NR == 1) ? PO_REG_WAVES_K1N : (K == 1 ? PO_REG_WAVES_K1W : (NR == 1 ? PO_REG_WAVES_K3N : 2));
};
template <int K> struct RegVal { double v[K]; };

// closed-form chains (SCAN): what a chain's 8 lanes hand to the child's lane (and the child's seed the other way), per
// (half of the wave, symbol); nothing in the serial-chain kernel
struct RegCResEntry { double mx, last; int mt, tr; };
template <bool SCAN> struct RegCRes { RegCResEntry e[2][PO_A]; };
template <> struct RegCRes<false> { RegCResEntry e[1][1]; };
template <int MODEL, int NR, bool SCAN = false>
struct RegSmem {              // per pair wave
    using Cfg = RegCfg<MODEL, NR>;
    double ybuf[2][RK_NY][Cfg::YC];
    union {
        double pst[(NR == 1) ? 2 : 1][Cfg::PS * RK_NY][Cfg::KP];   // staged values (per read for NR = 1): a block of RK_NY times of every staged parent
        // ... and, in the same bytes (the staged values live inside one new-element step's scan; these between the scans and
        // the table build): the prune with exact score ties' candidate slots in node-id order (po_stl_prune) and their scores;
        // `ord` is also the table build's "this old slot continues" mark
        struct { int ord[Cfg::NS]; double csc[Cfg::NS]; } tie;
    };
    int g_owner[Cfg::NGL], g_hi0[Cfg::NGL], g_hi1[Cfg::NGL];
    // the table fields only the table build (and the rare general scan) looks at, per element slot — the same for both reads:
    // in LDS they cost no register between two table builds
    int f_fc[Cfg::NS], f_crow2[Cfg::NS], f_par[Cfg::NS], f_gpar[Cfg::NS], f_prow2[Cfg::NS], f_depth[Cfg::NS], f_alias[Cfg::NS];
    double rootcum[2];        // the ctc root's alpha (blank prefix sum, PrefixTree.h:509-515) of each read at time rootT: added up as the
    int rootT[2];             // scans pass the times, while children of the root are in the table (the start of a pair)
    double pf0[2][Cfg::PF0N]; // a run's first step: the beam lanes' values at the window start, fetched with the staging of the
    int pf0_t[2][Cfg::PF0N];  // step before (their times; -1: none)
    RegCRes<SCAN> cres;
    unsigned long long nupd, nupd_x;
};

template <int MODEL, int NR, bool SCAN = false>
struct RegGroup {
    PoLaeTables lae;   // (first: at LDS address 0 the tables' offsets fit the immediate fields of ds_read2_b64 — one address per entry)
    RegSmem<MODEL, NR, SCAN> w[RegCfg<MODEL, NR>::WPG];   // (one per pair wave)
};

__device__ __forceinline__ void rk_sync() { b2_sync_lds<64>(); }
// the smallest of x over lanes 0 .. n - 1 (n <= 16: the beam slots sit in row 0 of the wave), wave-uniform: four row_shr steps
// in the VALU and one pair of v_readlane instead of 2 n v_readlane and n - 1 minima
__device__ __forceinline__ double rk_row0_min(double x, int n, int lane) {
    x = (lane < n) ? x : __builtin_inf();
#define RK_STEP(ctrl)                                                                                                     \
    x = po_vmin(x, __hiloint2double(__builtin_amdgcn_update_dpp(__double2hiint(x), __double2hiint(x), ctrl, 0xf, 0xf, false), \
                                    __builtin_amdgcn_update_dpp(__double2loint(x), __double2loint(x), ctrl, 0xf, 0xf, false)))
    RK_STEP(0x111); RK_STEP(0x112); RK_STEP(0x114); RK_STEP(0x118);
#undef RK_STEP
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 15), __builtin_amdgcn_readlane(__double2loint(x), 15));
}
__device__ __forceinline__ double rk_readlane_d(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}

// ---- exchanges inside groups of 8 lanes, all in the VALU (DPP): the closed-form chains below give a chain 8 lanes
template <int CTRL>
__device__ __forceinline__ double rk_dpp_d(double x) {
    return __hiloint2double(__builtin_amdgcn_update_dpp(__double2hiint(x), __double2hiint(x), CTRL, 0xf, 0xf, false),
                            __builtin_amdgcn_update_dpp(__double2loint(x), __double2loint(x), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int rk_dpp_i(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, false); }
// lane ^ 1, lane ^ 2 (quad_perm) and 7 - lane (row_half_mirror: a lane of the OTHER quad — after the first two levels the four
// lanes of a quad hold the same value, so any of them does)
constexpr int RK_X1 = 0xB1, RK_X2 = 0x4E, RK_X4 = 0x141;
__device__ __forceinline__ double rk_g8_max(double x) {
    x = po_vmax(x, rk_dpp_d<RK_X1>(x));
    x = po_vmax(x, rk_dpp_d<RK_X2>(x));
    return po_vmax(x, rk_dpp_d<RK_X4>(x));
}
__device__ __forceinline__ int rk_g8_max_i(int x) {
    x = max(x, rk_dpp_i<RK_X1>(x));
    x = max(x, rk_dpp_i<RK_X2>(x));
    return max(x, rk_dpp_i<RK_X4>(x));
}
// exclusive prefix sum of `tot` over the 8 lanes of a group (and the group's total, the same bits in every lane: a + b == b + a):
// b0 / b1 / b2 = 1.0 where bit 0 / 1 / 2 of the lane number is set, else 0.0 — a multiply-add instead of two selects per level
// (the summands are finite)
__device__ __forceinline__ double rk_g8_excl(double tot, double b0, double b1, double b2, double& total) {
    const double t1 = rk_dpp_d<RK_X1>(tot);
    double e = b0 * t1;
    const double p2 = tot + t1;
    const double t2 = rk_dpp_d<RK_X2>(p2);
    e = __builtin_fma(b1, t2, e);
    const double p4 = p2 + t2;
    const double t4 = rk_dpp_d<RK_X4>(p4);
    e = __builtin_fma(b2, t4, e);
    total = p4 + t4;
    return e;
}

}  // namespace

// COUNT: the instantiation po_profile_update_counter asks for (update_prob evaluations of the reference's schedule and executed
// ones, added up per step: ballots, a wave reduction per scan); the product path carries none of it.
// (The job-board form of round 4 — seven pair waves posting their new elements' chains to a chain wave — was measured
//  slower at every batch size and is gone: DESIGN.md, appendix.)
// SCAN: the new elements' chains in closed form (po_set_chain_mode(PO_CHAIN_CLOSED_FORM); the one-value model) — its own instantiation:
// with both forms of the chain in one kernel the register allocation of EVERY phase suffered (the serial-only kernel ran at
// 69 instead of 51 ms per 10 000 pairs with the closed form merely compiled in).
template <int MODEL, int NR, bool COUNT = false, bool SCAN = false>
__global__
__launch_bounds__((64 * RegCfg<MODEL, NR>::WPG), (RegCfg<MODEL, NR>::WAVES))
void beam2d_reg_kernel(X2Args a) {
    static_assert(!SCAN || RegCfg<MODEL, NR>::K == 1, "closed-form chains: the one-value model");
    using Cfg = RegCfg<MODEL, NR>;
    constexpr int K = Cfg::K, KP = Cfg::KP, PC0 = Cfg::PC0, RK_YC = Cfg::YC, EB = Cfg::EB, RK_PS = Cfg::PS;
    constexpr int NS = Cfg::NS, WS = Cfg::WS, RK_NGL = Cfg::NGL, PF0N = Cfg::PF0N;
    using Val = RegVal<K>;        // a node's values at one time = a store entry
    using PVal = RegVal<KP>;      // ... the ones its children's updates read
    __shared__ RegGroup<MODEL, NR, SCAN> gsm;
    // lane = (read, slot) [NR = 1] or slot [NR = 2]; hb = first lane of this lane's read; lo_half = the lanes that do what is
    // done once per element slot (arena and table writes, candidate masks)
    const int lane = threadIdx.x & 63, s = (NR == 1) ? (lane & 31) : lane, hb = (NR == 1) ? (lane & 32) : 0;
    const bool lo_half = (NR == 1) ? (lane < 32) : true;
    auto RD = [&](int q) -> int { return (NR == 1) ? (lane >> 5) : q; };   // the read a lane's q-th state belongs to
    // one bit per element slot: 32 bits do for NR = 1 (scalar 32-bit operations; the 64-bit forms cost the W = 5 kernel 1.6 %)
    using SMask = typename std::conditional<NR == 1, unsigned, unsigned long long>::type;
    auto smask_of = [&](bool p) -> SMask { return (SMask)__ballot(p && lo_half); };
    auto sm_pop = [&](SMask m) -> int { return (NR == 1) ? __popc((unsigned)m) : __popcll((unsigned long long)m); };
    auto sm_ctz = [&](SMask m) -> int { return (NR == 1) ? __builtin_ctz((unsigned)m) : (int)__builtin_ctzll((unsigned long long)m); };
    RegSmem<MODEL, NR, SCAN>& sm = gsm.w[(Cfg::WPG == 1) ? 0 : (int)(threadIdx.x >> 6)];
    const int A = a.A, W = a.W, C = a.C;
    const int divA = (65536 + A - 1) / A;   // x / A == (x * divA) >> 16 for the slot numbers divided here
    // ---- this wave's SLICE of the library's pool: value store + tree arena.  The pool has one slice per pair wave the device
    // can hold (po_beam2d.hip::reg_pool), shared by every launch of this kernel on the device — the waves of a pipelined job on
    // their streams, the next call — so a workspace no longer carries megabytes per resident pair.  A wave CLAIMS a free slice
    // when it starts and gives it back when the queue is empty.  The hand-over between waves is an agent-scope release /
    // acquire pair (per-XCD L2s are not coherent, a CU's L1 is never refreshed by another CU's stores): once per wave's lifetime.
    // The free slices sit in a ring of nslices words (slice number, or -1: empty) with a take ticket and a give ticket: a wave
    // that starts takes the word its ticket names (and waits for it to be filled, should every slice be out: residency is what
    // the pool is sized for, so that cannot last), a wave that ends puts its slice into the word ITS ticket names.  Two atomics
    // per hand-over, no search — with a wave per pair (persist = 0) that is per pair.
    po_lae_tables_load(&gsm.lae, (int)threadIdx.x, (int)blockDim.x);
    const PoLaeFast lae{&gsm.lae};
    if (lane == 0) { sm.nupd = 0; sm.nupd_x = 0; }
    __syncthreads();   // (the only time the waves of a workgroup meet)
    const int gwave = (int)blockIdx.x * Cfg::WPG + (int)(threadIdx.x >> 6);   // this pair wave's number in the launch
    if (gwave >= a.reg_slots) return;
    int slotid = -1;
    {
        int v = -1;
        if (lane == 0) {
            const unsigned t = atomicAdd(&a.slice_tickets[0], 1u) % (unsigned)a.nslices;
            for (;;) {
                v = atomicExch(&a.slice_claim[t], -1);
                if (v >= 0) break;
                __builtin_amdgcn_s_sleep(8);
            }
        }
        slotid = __builtin_amdgcn_readfirstlane(v);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    char* const slice = a.slice_chunk[slotid >> a.slice_spc_log2] + (size_t)(slotid & ((1 << a.slice_spc_log2) - 1)) * a.slice_bytes;
    const long long pool_entries = (long long)(a.pool_bytes / EB);
    int* const apl = (int*)(slice + a.pool_bytes);
    int* const afc = apl + a.arena_cap;
    int* const acrow = afc + a.arena_cap;
    // row headers, one int per (store row, read): the time a node's stored values END at (exclusive), written when the node
    // stops being an element; RK_FRESH = nothing stored.  See "the value store" below.
    int* const rowhdr = acrow + a.arena_cap;
    // nodes a slice's arena holds: a pair that needs more goes to beam2d_kernel (starve: the tests' way to get there)
    const int arena_cap = (a.starve & 2) ? min((int)a.arena_cap, 1 + a.A + 24 * a.A) : (int)a.arena_cap;
    auto g_hi = [&](int r) -> int* { return r ? sm.g_hi1 : sm.g_hi0; };

#ifdef PO_REG_TIMING
    // phase timers of workgroup 0 (wall_clock64: 100 MHz) and counts: see po_reg_launch for the names
    long long tk[56], tlast = wall_clock64();
    for (int i = 0; i < 56; ++i) tk[i] = 0;
#define KT(i) do { const long long n_ = wall_clock64(); tk[(i)] += n_ - tlast; tlast = n_; } while (0)
#define KC(i, n) do { tk[(i)] += (n); } while (0)
#else
#define KT(i) do {} while (0)
#define KC(i, n) do {} while (0)
#endif

    for (int taken = 0;; ++taken) {
        // ---------------------------------------------------------------- next pair from the queue
        // (persist = 0: this wave's one pair is the one its number names; the launch has a wave per pair)
        if (!a.persist && taken > 0) break;
        int pi = 0;
        if (lane == 0) {
            const int q = a.persist ? atomicAdd(a.queue, 1) : gwave;
            pi = (a.order != nullptr && q < a.n) ? a.order[q] : q;   // longest pairs first (pair_order_kernel)
        }
        pi = __builtin_amdgcn_readfirstlane(pi);
        if (pi >= a.n) break;
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
        // (a read's rows, length and blank prefix sums are put together where they are used — a y reload every ~ 16 steps, the
        //  root's children at the start of a pair — rather than held in registers across the walk)
        auto yr_ = [&](int r) -> const double* { return r ? a.y2 + o2 * C : a.y1 + o1 * C; };
        const int4* const sched = a.sched + (o2 - a.y2_off[0]);
        const int nmain = a.nmain[pi];
        const int R2 = m.y, Rm2 = R2 - 1;
        if (__builtin_expect(R2 > 256, 0)) {   // windows of 255 frames and more: the packed walk records below keep a window's length in 8 bits
            if (lane == 0) { a.meta[pi] = make_int2(PO_OK, X2_DEFERRED); a.queue[16] = 1; if (a.defer_count) atomicAdd(a.defer_count, 1ull); }
            continue;
        }
        const int NG = (int)min((long long)((a.starve & 1) ? 12 : RK_NGL), pool_entries / ((long long)PO_A * 2 * R2));
        int st = PO_OK;

        // ---------------------------------------------------------------- the value store
        // The reference's per-node maps (PrefixTree.h:76-145): every node owns a ring row of R2 entries per read, an entry is
        // the node's K values at one time — 8 or 24 bytes, no tag (rounds 1 - 4 stored {tag(epoch, node, t), values}: twice the
        // bytes for the one-value model, and two thirds of the kernel's HBM-side traffic is these writes).  Whether (node, t) is
        // PRESENT — probability_at() answers -inf otherwise — is known without looking at the entry:
        //   * an ELEMENT's stored values are [.., v_done) of its lane (every read in this kernel asks for t >= window start - 1,
        //     and an element's first time is the window start of the step it was created in, or later);
        //   * a node that is NO element (a frozen parent's older values; the seed of a node that becomes an element again) has
        //     its end in the row's HEADER, written when it stopped being an element; a row group's headers are emptied when the
        //     group is handed to a parent, so nothing of an earlier owner — or of an earlier pair in this slice — is ever valid;
        //   * the ring cannot have wrapped over a time that is asked for: R2 >= widest window + 2.
        // (entry index and byte offset stay within 32 bits: a slice is a few MB)
        const char* const poolb = (const char*)slice;
        auto t2_off = [&](int r, int row2, int tq) -> unsigned { return (unsigned)(((row2 * 2 + r) * R2 + (tq & Rm2)) * EB); };
        auto val_neg = [&]() -> Val { Val x; for (int q = 0; q < K; ++q) x.v[q] = PO_NEG_INF; return x; };
        auto pval_neg = [&]() -> PVal { PVal x; for (int q = 0; q < KP; ++q) x.v[q] = PO_NEG_INF; return x; };
        auto pval_of = [&](const Val& x) -> PVal { PVal y; for (int q = 0; q < KP; ++q) y.v[q] = x.v[PC0 + q]; return y; };
        auto pval_shfl = [&](const PVal& x, int src) -> PVal { PVal y; for (int q = 0; q < KP; ++q) y.v[q] = __shfl(x.v[q], src); return y; };
        auto val_shfl = [&](const Val& x, int src) -> Val { Val y; for (int q = 0; q < K; ++q) y.v[q] = __shfl(x.v[q], src); return y; };
        // (value by value: an aggregate copy of the 24-byte entry keeps v_self and friends in scratch memory instead of registers)
        auto t2_load = [&](int r, int row2, int tq) -> Val {   // (all K values)
            const double* const p = (const double*)(poolb + (size_t)t2_off(r, row2, tq));
            Val v;
            for (int q = 0; q < K; ++q) v.v[q] = p[q];
            return v;
        };
        auto t2_load0 = [&](int r, int row2, int tq) -> double { return *(const double*)(poolb + (size_t)t2_off(r, row2, tq)); };   // (alpha alone)
        auto hdr_of = [&](int r, int row2) -> int* { return rowhdr + (row2 * 2 + r); };
        // values of (row, time) when `present` says they are there
        auto t2_read = [&](int r, int row2, int tq, bool present) -> Val {
            Val v = val_neg();
            if (present && tq >= 0 && row2 >= 0) {
                const double* const p = (const double*)(poolb + (size_t)t2_off(r, row2, tq));
                for (int q = 0; q < K; ++q) v.v[q] = p[q];
            }
            return v;
        };
        auto t2_write = [&](int r, int row2, int tq, const Val& v) {
            double* const p = (double*)(const_cast<char*>(poolb) + (size_t)t2_off(r, row2, tq));
            for (int q = 0; q < K; ++q) p[q] = v.v[q];
        };
        // update_prob of one element at one time: sp = its own values at t - 1, pk = its parent's (the ones read), ya / yb = the
        // two y entries of the row (own symbol; blank, or the symbol's flop column), same = parent->last == last
        // (first — a child of the root at t = 0 — happens at the start of a pair only: below)
        auto upd = [&](const Val& sp, const PVal& pk, double ya, double yb, bool same) -> Val {
            double pp[3] = {0.0, 0.0, 0.0};
            for (int q = 0; q < KP; ++q) pp[PC0 + q] = pk.v[q];
            Val o;
            po_update<MODEL>(sp.v, pp, ya, yb, same, false, o.v, lae);
            return o;
        };
        // ---------------------------------------------------------------- per-lane element state (slot s)
        // table fields (the same for both reads)
        int e_id = 0, e_row2 = -1, e_sym = 0, e_ps = PS_ROOT;   // (first child, children's row group, parent, grandparent, the
                                                                  //  parent's row, depth, alias: sm.f_*)
        if (lo_half) {
            sm.f_fc[s] = -1; sm.f_crow2[s] = -1; sm.f_par[s] = 0; sm.f_gpar[s] = -1; sm.f_prow2[s] = -1;
            sm.f_depth[s] = (s < A) ? 1 : 0; sm.f_alias[s] = -1;
        }
        // (emulator builds with -DPO_EMU_SHADOW: what rounds 1 - 4's TAGS would have answered, kept beside the store — every read's
        //  `present` is checked against it, and the first disagreement is reported)
#ifdef PO_EMU_SHADOW
        int sh_step = -1, sh_u = -1, sh_v = -1;
        auto sh_key = [&](int r, int row2, int tq) -> unsigned long long { return ((unsigned long long)(unsigned)slotid << 40) | (unsigned long long)t2_off(r, row2, tq); };
        auto sh_write = [&](int r, int row2, int tq, int node) { po_emu_shadow_put(sh_key(r, row2, tq), ((unsigned long long)(unsigned)pi << 48) | ((unsigned long long)(unsigned)node << 24) | (unsigned)tq); };
        auto sh_chk = [&](int r, int row2, int tq, bool present, int node, int site) {
            if (tq < 0 || row2 < 0) return;
            const bool hit = po_emu_shadow_get(sh_key(r, row2, tq)) == (((unsigned long long)(unsigned)pi << 48) | ((unsigned long long)(unsigned)node << 24) | (unsigned)tq);
            if (hit != present) printf("SHADOW site %d pair %d lane %d read %d row2 %d t %d node %d: present %d, a tag would %s (step %d, u %d v %d)\n", site, pi, (int)lane, r, row2, tq, node, (int)present, hit ? "HIT" : "MISS", sh_step, sh_u, sh_v);
        };
#define SH_WRITE(r, row2, tq, node) sh_write(r, row2, tq, node)
#define SH_CHK(r, row2, tq, present, node, site) sh_chk(r, row2, tq, present, node, site)
#else
#define SH_WRITE(r, row2, tq, node) do {} while (0)
#define SH_CHK(r, row2, tq, present, node, site) do {} while (0)
#endif
        bool live = false;
        // values of a read: computed and stored up to v_done (exclusive); v_fresh: 1 = an element again, its last
        // value is in the store; 2 = a node that never computed
        int v_done[NR], v_fresh[NR];
        Val v_self[NR];
        double v_mx[NR];
        int v_mt[NR], v_td[NR];
        // a beam node whose parent is no element any more (FROZEN): the parent's last value and its time, taken when the
        // parent left — later times are absent (-inf), earlier ones are in the store.  fz_t = INT_MAX: nothing captured.
        PVal fz_val[NR];
        int fz_t[NR];
        int yhi[NR];                // y rows [yhi - RK_NY, yhi) of the read are in sm.ybuf
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            v_done[q] = RK_FRESH; v_fresh[q] = 0; v_self[q] = val_neg(); v_mx[q] = PO_NEG_INF; v_mt[q] = -1; v_td[q] = 0;
            fz_val[q] = pval_neg(); fz_t[q] = INT_MAX; yhi[q] = 0;
        }
        int nb = A, ne = A;
        int next_id = 1 + A;
        int sel[WS];
#pragma unroll
        for (int i = 0; i < WS; ++i) sel[i] = i;

        for (int q = lane; q < RK_NGL; q += 64) { sm.g_owner[q] = -1; sm.g_hi0[q] = 0; sm.g_hi1[q] = 0; }
        rk_sync();
        // root = node 0; its A children = nodes 1..A in row group 0 (BeamSearch.h:286-293), updated at t = 0 on both reads
        if (lane == 0) {
            apl[0] = po_pack_node(-1, A); afc[0] = 1; acrow[0] = 0;
            sm.g_owner[0] = 0; sm.g_hi0[0] = 1; sm.g_hi1[0] = 1;
        }
        if (lane < 2 * PO_A) rowhdr[lane] = RK_FRESH;   // (row group 0: the root's children)
        if (s < A) {
            if (lo_half) { apl[1 + s] = po_pack_node(0, s); afc[1 + s] = -1; acrow[1 + s] = -1; }
            e_id = 1 + s; e_row2 = s; e_sym = sym_pack(s, A, true); e_ps = PS_ROOT;
            live = true;
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const int r = RD(q);
                const double* const yr = yr_(r);
                Val out;
                {   // update_prob(n, r, 0): parent = root at t = -1 (tree constructors: PrefixTree.h:467-476, :541-546, :641-647), first = true
                    double sp[3] = {PO_NEG_INF, PO_NEG_INF, PO_NEG_INF}, pp[3];
                    root_values<MODEL>(-1, 0.0, pp);
                    po_update<MODEL>(sp, pp, yr[s], (MODEL == PO_MODEL_FLIPFLOP) ? yr[s + A] : yr[A], false, true, out.v, lae);
                }
                t2_write(r, e_row2, 0, out); SH_WRITE(r, e_row2, 0, e_id);
                v_done[q] = 1; v_fresh[q] = 0; v_self[q] = out;
                v_mx[q] = out.v[0]; v_mt[q] = 0; v_td[q] = 0;   // (the window maximum over [0, 1))
                if (MODEL == PO_MODEL_CTC && s == 0) { sm.rootcum[r] = 0.0 + yr[A]; sm.rootT[r] = 0; }   // (serial in t from 0.0, as the reference adds)
            }
        }
        rk_sync();

        int mstep = 0, up = -1, vp = -1;
        int pf0_step = -1;   // the main step sm.pf0 was filled for
        // The walk's records, 64 at a time: lane l holds record 64 * batch + l of the current batch and of the next one
        // (requested a batch ahead: the load's latency never shows), the step's own record comes out with v_readlane.
        // (kept PACKED, two words per record — time | window length << 24; times stay below 2^24 and a window below the
        //  store's ring length of <= 256 — : four registers for the two batches instead of eight)
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
        // v_done of beam slot 0 (which always continues) on read 0 / read 1: where everybody stands
        auto done0 = [&]() -> int { return __builtin_amdgcn_readlane(v_done[0], 0); };
        auto done1 = [&]() -> int { return (NR == 1) ? __builtin_amdgcn_readlane(v_done[0], 32) : __builtin_amdgcn_readlane(v_done[NR - 1], 0); };

        // a lane's own stored values: the caller knows tq to be inside [.., v_done) (an element), or says how it knows
        auto read_own = [&](int r, int tq) -> double { return (tq >= 0) ? t2_load0(r, e_row2, tq) : PO_NEG_INF; };
        auto read_own_all = [&](int r, int tq, bool present) -> Val { return t2_read(r, e_row2, tq, present); };

        // ---------------------------------------------------------------- y rows [t0, t0 + RK_NY) of a read -> LDS
        // (all of a lane's loads go out together: one memory round trip per reload)
        auto y_reload = [&](int r, int t0) {
            constexpr int PER = (RK_NY * RK_YC + NS - 1) / NS;   // elements per lane (C <= RK_YC)
            double v[PER];
            int slot[PER];
            const double* const yr = yr_(r);
            const int Tr = r ? V : U;
            const int divC = (65536 + C - 1) / C;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int i = s + NS * j;
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
        // ballots.
        auto carried_one = [&](int r, int ws, double& cmx, int& cmt, int td, int pf_t = -1, double pf_val = 0.0) -> bool {
            if (td > ws) return true;
            if (ws != pf_t) SH_CHK(r, e_row2, ws, true, e_id, 1);
            cmx = (ws == pf_t) ? pf_val : read_own(r, ws);
            cmt = ws;
            return false;
        };
        auto rescan_wave = [&](int q, bool need, int ws, int start, double& cmx, int& cmt, int& td) {
            unsigned long long m = __ballot(need);
            KC(21, __popcll(m)); KC(10, m != 0ull ? 1 : 0);
            while (m != 0ull) {   // (wave-uniform)
                const int L = (int)__builtin_ctzll(m);
                m &= m - 1ull;
                const int wsL = __builtin_amdgcn_readlane(ws, L), teL = __builtin_amdgcn_readlane(min(td + 1, start), L);
                const int rowL = __builtin_amdgcn_readlane(e_row2, L);
                const int rowbase = (rowL * 2 + ((NR == 1) ? (L >> 5) : q)) * R2;
                double bmx = PO_NEG_INF, pvc = PO_NEG_INF;
                int bmt = -1, btd = wsL;
                for (int base = wsL; base < teL; base += 64) {
                    const int tq = base + lane;
                    const bool valid = tq < teL;
                    double val = PO_NEG_INF;
                    { const int idL_ = __builtin_amdgcn_readlane(e_id, L); (void)idL_; if (valid) SH_CHK((NR == 1) ? (L >> 5) : q, rowL, tq, true, idL_, 2); }
                    if (valid) val = *(const double*)(poolb + (size_t)(unsigned)((rowbase + (tq & Rm2)) * EB));   // (the lane computed every one of them)
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
        // Every participating lane computes [max(done, ws), we) of a read, all lanes of the read in lockstep on t: a child
        // at t takes its parent's t - 1 from the parent's lane when the parent computed it one iteration earlier (or holds
        // it as its last value), from the store otherwise.  MAIN steps (is_main) track the window maximum; catch-up scans
        // (BeamSearch.h:314-336) move the beam nodes only.
        double smx[NR];   // out: max over each read's window (main steps)
#pragma unroll
        for (int q = 0; q < NR; ++q) smx[q] = PO_NEG_INF;
        auto scan = [&](bool is_main, int ws0, int we0, int ws1, int we1, int nlanes) {
#pragma unroll
            for (int q = 0; q < NR; ++q) {
            const int r = RD(q);
            const int ws = r ? ws1 : ws0, we = r ? we1 : we0;
            const bool part = live && s < nlanes && we > ws;
            // a window end that moves back cannot happen on a monotone envelope (the pre-pass hands the others to
            // beam2d_kernel); should it, the pair goes the same way
            // (st stays wave-uniform: the walk loop's condition reads it)
            if (is_main && __ballot(part && v_fresh[q] == 0 && v_done[q] > we) != 0ull) st = PO_E_NOMEM;
            int start = max(v_done[q], ws);
            // (the lane's own state is read HERE, not inside the branches below: a branch that loads a double from the store next
            //  to one that loads it from v_self / fz_val has the two loads merged into one through a pointer — and the lane's
            //  state lives in scratch memory from then on, in every phase of the kernel; measured: Bonito pairs 6 % slower)
            const Val own = v_self[q];
            const PVal fzv = fz_val[q];
            Val self = val_neg();
            if (part) {
                if (v_fresh[q] != 0) {
                    start = ws;
                    if (v_fresh[q] == 1) SH_CHK(r, e_row2, start - 1, start - 1 < *hdr_of(r, e_row2), e_id, 3);
                    if (v_fresh[q] == 1) self = read_own_all(r, start - 1, start - 1 < *hdr_of(r, e_row2));   // (an element again: the header knows)
                } else if (start > v_done[q]) {
                    // a gap (catch-ups went beyond the last window): the value at start - 1 was never computed
                } else {
                    self = own;
                }
            }
            const bool part2 = part && start < we;
            double mx = PO_NEG_INF, cmx = PO_NEG_INF;
            int mt = -1, cmt = -1, td = ws, tr = INT_MIN;
            const bool has_c = is_main && part && start > ws;
            bool rsc = false;
            if (has_c) {
                td = v_td[q];
                if (v_mx[q] == PO_NEG_INF || (v_mt[q] >= ws && v_mt[q] < start)) { cmx = v_mx[q]; cmt = v_mt[q]; }
                else rsc = carried_one(r, ws, cmx, cmt, td);
            }
            rescan_wave(q, rsc, ws, start, cmx, cmt, td);
            const int sym = sym_last(e_sym), cb = (MODEL == PO_MODEL_FLIPFLOP) ? sym + A : A;
            const bool same = sym_plast(e_sym) == sym;
            const bool has_root = MODEL == PO_MODEL_CTC && __ballot(live && e_ps == PS_ROOT) != 0ull;
            bool bad_root = false;
            // the parent's lane: where it starts and ends in this scan (its `self` is its value at p_start - 1 before the
            // first iteration, then at the time it computed last)
            const int plane = (e_ps >= 0) ? (hb | e_ps) : lane;
            const int p_start = __shfl(part2 ? start : INT_MAX, plane), p_we = __shfl(part2 ? we : INT_MIN, plane);
            // where the parent's STORED values end: an element's v_done; an element AGAIN that has not computed since (a child slot
            // during catch-up steps, which only the beam nodes take part in) still has what it stored before it left — its row's
            // header says up to where (the round-5 fuzz: a beam node's catch-up read of such a parent was answered "absent")
            int own_end = RK_FRESH;
            if (live && v_fresh[q] == 0) own_end = v_done[q];
            else if (live && v_fresh[q] == 1 && e_row2 >= 0) own_end = *hdr_of(r, e_row2);
            const int p_done = __shfl(own_end, plane);
            int tm_ = part2 ? start : INT_MAX;
#pragma unroll
            for (int off = NS / 2; off >= 1; off >>= 1) tm_ = min(tm_, __shfl_xor(tm_, off));
            const int tmin = tm_;                                   // (uniform over the lanes of the read)
            const int span = (tmin == INT_MAX) ? 0 : we - tmin;
            const int niter = (NR == 1) ? max(__builtin_amdgcn_readlane(span, 0), __builtin_amdgcn_readlane(span, 32)) : __builtin_amdgcn_readlane(span, 0);
            int k = 0;
            while (k < niter) {
                const int tcur = tmin + k;   // (garbage when this read has nothing to do: guarded by span)
                const bool hw = k < span;    // this read still has times to compute
                rk_sync();                   // (every lane is done with the rows a reload overwrites)
                if (hw && !(tcur >= yhi[q] - RK_NY && tcur < yhi[q])) { y_reload(r, tcur); yhi[q] = tcur + RK_NY; }
                rk_sync();
                const int cend = hw ? (min(we, yhi[q]) - tmin) : niter;
                const int kend = (NR == 1) ? min(__builtin_amdgcn_readlane(cend, 0), __builtin_amdgcn_readlane(cend, 32)) : __builtin_amdgcn_readlane(cend, 0);
                for (; k < kend; ++k) {
                    const int t = tmin + k;
                    const PVal ps_self = pval_shfl(pval_of(self), plane);
                    if (part2 && t >= start && t < we) {
                        const double ya = sm.ybuf[r][t & (RK_NY - 1)][sym], yb = sm.ybuf[r][t & (RK_NY - 1)][cb];
                        const int tm = t - 1;
                        PVal pp;
                        if (e_ps >= 0) {
                            if (tm >= p_start - 1 && tm < p_we && p_start != INT_MAX) pp = ps_self;
                            else { SH_CHK(r, sm.f_prow2[s], tm, tm < p_done, sm.f_par[s], 4); pp = pval_of(t2_read(r, sm.f_prow2[s], tm, tm < p_done)); }
                        } else if (e_ps == PS_ROOT) {
                            // (t >= 1 here: the root's children got their t = 0 at the start of the pair; the other two models'
                            //  root holds nothing at times >= 0)
                            pp = pval_neg();
                            if (MODEL == PO_MODEL_CTC) {
                                pp.v[0] = 0.0;
                                if (tm >= 0) { pp.v[0] = sm.rootcum[r]; bad_root = bad_root || (tm != sm.rootT[r]); }
                            }
                        } else if (tm >= fz_t[q]) {
                            pp = (tm == fz_t[q]) ? fzv : pval_neg();                            // frozen parent: its last value, then nothing
                        } else {   // a frozen parent's older values: its row's header says where they end
                            const int prow = sm.f_prow2[s];
                            SH_CHK(r, prow, tm, prow >= 0 && tm < *hdr_of(r, max(prow, 0)), sm.f_par[s], 5);
                            pp = pval_of(t2_read(r, prow, tm, prow >= 0 && tm < *hdr_of(r, max(prow, 0))));
                        }
                        const Val out = upd(self, pp, ya, yb, same);
#ifdef PO_RING_TRACE_NODE
                        if (pi == 0 && e_id == PO_RING_TRACE_NODE) printf("V %d %d %d %.17g %.17g %.17g G ps %d fzt %d main %d\n", e_id, r, t, out.v[0], pp.v[0], self.v[0], e_ps, fz_t[q], (int)is_main);
#endif
                        t2_write(r, e_row2, t, out); SH_WRITE(r, e_row2, t, e_id);
                        if (out.v[0] > self.v[0]) tr = t;   // the last time a value rose
                        self = out;
                        mt = (out.v[0] >= mx) ? t : mt;
                        mx = po_vmax(mx, out.v[0]);
                    }
                    if (has_root) {   // the root's alpha moves on with the times this read's scans pass (every one of them, in order)
                        rk_sync();
                        if (s == 0 && k < span && t == sm.rootT[r] + 1) { sm.rootcum[r] += sm.ybuf[r][t & (RK_NY - 1)][A]; sm.rootT[r] = t; }
                        rk_sync();
                    }
                }
            }
            if (__ballot(bad_root) != 0ull) st = PO_E_NOMEM;   // (a time the sums have not reached: cannot happen — beam2d_kernel would take the pair)
            po_settle(cmx);   // (this scan is the rare path: what it loaded is final before it returns — po_settle)
            for (int k2 = 0; k2 < K; ++k2) po_settle(self.v[k2]);
            if (has_c && !(mx >= cmx)) { mx = cmx; mt = cmt; }   // (new values, later in time, win ties)
            if (part2) { v_done[q] = we; v_self[q] = self; v_fresh[q] = 0; }
            if (is_main) {
                if (part) { v_mx[q] = mx; v_mt[q] = mt; v_td[q] = max(td, tr); }
                smx[q] = part ? mx : PO_NEG_INF;
            }
            KT(is_main ? 4 : 5); KC(is_main ? 16 : 17, 1); KC(18, niter);
            if constexpr (COUNT) {
                const int lenx = part2 ? we - start : 0;
                int tot = lenx;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) tot += __shfl_xor(tot, off);
                cnt_x += (unsigned)tot;
            }
            }   // (q)
        };

        // ---------------------------------------------------------------- a main step with NEW elements
        // After a rebuild: the continuing elements (all ending at the same time dr, as in a run) only need the new times
        // [dr, we); the fresh ones — children of a node that entered the beam — need their whole window [ws, we).  Over
        // [ws, dr) their parent does not move: its values are at rest in the store, so they are STAGED into LDS in one
        // memory round trip (with the fresh lanes' own seeds), and phase 1 runs the fresh lanes' chains on LDS and registers
        // only; phase 2 is the run loop's lockstep over the new times for everybody.  Returns false (nothing done) when
        // the step is not of this kind: scan() takes it.
        auto scan_new = [&](int u, int ce, int v, int re) -> bool {
            const int d0 = done0(), d1 = done1();
            if (__builtin_expect(!(u <= d0 && d0 <= ce && v <= d1 && d1 <= re), 0)) { KC(22, 1); return false; }
            bool fresh[NR], fresh_any = false;
            {
                bool bad = false;
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    const int r = RD(q);
                    const int dr = r ? d1 : d0;
                    fresh[q] = live && v_fresh[q] != 0;
                    const bool cont = live && v_fresh[q] == 0;
                    // everybody who continues ends at dr; fresh lanes hang under a continuing lane; no root, no older frozen values
                    const bool pfresh = __shfl((int)fresh[q], hb | max(e_ps, 0)) != 0;
                    bad = bad || (cont && v_done[q] != dr) || (live && e_ps == PS_ROOT) || (fresh[q] && e_ps < 0) || (fresh[q] && pfresh);
#ifdef PO_REG_TIMING
                    KC(23, __ballot(cont && v_done[q] != dr) != 0ull); KC(24, __ballot(live && e_ps == PS_ROOT) != 0ull);
                    KC(25, __ballot(fresh[q] && e_ps < 0) != 0ull); KC(26, __ballot(fresh[q] && pfresh) != 0ull);
#endif
                    fresh_any = fresh_any || fresh[q];
                }
                if (__builtin_expect(__ballot(bad) != 0ull, 0)) return false;
            }
            // ---- the parents to stage (beam slots with fresh children): at most RK_PS
            // (their slots — below 16 — in four bits each of one wave-uniform word: an int array indexed by k ends up in scratch)
            unsigned long long pjb = 0ull;
            int nps = 0;
            auto pj = [&](int k) -> int { return (int)((pjb >> (4 * k)) & 15ull); };
            bool many = false;
            for (int jj = 0; jj < nb; ++jj) {   // (wave-uniform)
                if (__ballot(fresh_any && e_ps == jj) == 0ull) continue;
                if (nps < RK_PS) {
                    pjb |= (unsigned long long)jj << (4 * nps);
                    nps++;
                } else many = true;
            }
            if (__builtin_expect(many, 0)) { KC(27, 1); return false; }
            int myk = 0;
            for (int k = 1; k < nps; ++k) myk = (e_ps == pj(k)) ? k : myk;   // (wave-uniform loop)
            const int sym = sym_last(e_sym), cb = (MODEL == PO_MODEL_FLIPFLOP) ? sym + A : A;
            const bool same = sym_plast(e_sym) == sym;
            KT(1); KC(13, 1);
#pragma unroll
            for (int q = 0; q < NR; ++q) {
            const int r = RD(q);
            const int ws = r ? v : u, dr = r ? d1 : d0;
            const bool cont = live && v_fresh[q] == 0;
            const int n1 = dr - ws;   // (uniform over the read's lanes, >= 0): times the fresh lanes compute before everybody else starts
            // the fresh lanes' own seeds (an element again: its last value is in the store)
            Val se;
            for (int c = 0; c < K; ++c) se.v[c] = 0.0;
            int se_hdr = RK_FRESH;   // (a seed's row header: requested with the seed, looked at when both are there)
            const bool want_seed = fresh[q] && v_fresh[q] == 1 && ws - 1 >= 0;
            // (the same registers, other lanes: a continuing beam lane whose window maximum has left the window and whose
            //  values fall — the run that follows this step asks for its value at ws first thing)
            const bool want_pf = cont && s < nb && v_done[q] > ws && v_mx[q] != PO_NEG_INF && v_mt[q] < ws && v_td[q] <= ws;
            if (want_seed) { se = t2_load(r, e_row2, ws - 1); se_hdr = *hdr_of(r, e_row2); SH_CHK(r, e_row2, ws - 1, ws - 1 < se_hdr, e_id, 6); }
            else if (want_pf) { se = t2_load(r, e_row2, ws); SH_CHK(r, e_row2, ws, true, e_id, 7); }
            double mx = PO_NEG_INF;
            Val self = val_neg();
            int mt = -1, tr = INT_MIN;
            const double* const yb_ = &sm.ybuf[r][0][0];
            // ---- phase 1: the fresh lanes over [ws, dr) — every operand is at rest.  In blocks of RK_NY times: the parents'
            // stored values of the block are STAGED (lane i of a read asks for time ws - 1 + k0 + i of each parent: one memory
            // round trip for all of them, with the y rows of the block), then the chains run on LDS and registers only — a
            // load inside the chain loop would make the compiler wait for vmcnt(0) there, i.e. for every value-store write
            // of the iteration before.
            const int n1max = (NR == 1) ? max(d0 - u, d1 - v) : n1;
            KC(14, n1max);
            const double* const ps_ = &sm.pst[(NR == 1) ? r : 0][myk * RK_NY][0];
            for (int k0 = 0; k0 < n1max; k0 += RK_NY) {
                rk_sync();   // (every lane is done with the rows and staged values of the block before)
                // (the first two parents' entries are asked for BEFORE the y rows: one memory round trip for the rows and the
                //  staged values of the usual step — one or two nodes entered the beam — instead of one after the other)
                if constexpr (NR == 1) {
                const int i = k0 + s, tq = ws - 1 + i;
                const bool stg = i < n1;
                // (a staged parent is a continuing beam lane: its stored values end at its v_done on this read)
                Val e01[RK_NB];
                bool ok01[RK_NB];
#pragma unroll
                for (int k = 0; k < RK_NB; ++k) {
                    ok01[k] = false;
                    for (int c = 0; c < K; ++c) e01[k].v[c] = 0.0;
                    if (k < nps) {   // (wave-uniform)
                        const int jk = pj(k);
                        const int prow = __builtin_amdgcn_readlane(e_row2, jk);
                        const int pdone = __shfl(v_done[q], hb | jk);
                        ok01[k] = stg && tq >= 0 && tq < pdone;
                        { const int pid_ = __builtin_amdgcn_readlane(e_id, jk); (void)pid_; if (stg && tq >= 0) SH_CHK(r, prow, tq, tq < pdone, pid_, 8); }
                        if (ok01[k]) e01[k] = t2_load(r, prow, tq);
                    }
                }
                {
                    const int lo = ws + k0, hi = min(lo + RK_NY, dr);
                    if (hi > lo && !(lo >= yhi[q] - RK_NY && hi <= yhi[q])) { y_reload(r, lo); yhi[q] = lo + RK_NY; }
                }
#pragma unroll
                for (int k = 0; k < RK_NB; ++k)
                    if (k < nps && stg) {
                        for (int c = 0; c < KP; ++c) sm.pst[r][k * RK_NY + s][c] = ok01[k] ? e01[k].v[PC0 + c] : PO_NEG_INF;
                    }
                for (int k = RK_NB; __builtin_expect(k < nps, 0); ++k) {   // (wave-uniform; three and more parents: rare)
                    const int jk = pj(k);
                    const int prow = __builtin_amdgcn_readlane(e_row2, jk);
                    const int pdone = __shfl(v_done[q], hb | jk);
                    const int pid9_ = __builtin_amdgcn_readlane(e_id, jk); (void)pid9_;
                    if (stg) {
                        SH_CHK(r, prow, tq, tq < pdone, pid9_, 9);
                        const PVal val = pval_of(t2_read(r, prow, tq, tq < pdone));
                        for (int c = 0; c < KP; ++c) sm.pst[r][k * RK_NY + s][c] = val.v[c];
                    }
                }
                } else {
                // 64 lanes per read, 32 times per block: the lower half asks for parent k, the upper half for parent k + 1
                const int sl = s & (RK_NY - 1), half = s >> 5;
                const int i = k0 + sl, tq = ws - 1 + i;
                const bool stg = i < n1;
                auto parent_of = [&](int kk, int& prow, int& pdone) {   // (per lane: the two halves name different parents)
                    const int jk = pj(min(kk, RK_PS - 1));   // (kk >= nps: slot 0, never used)
                    prow = __shfl(e_row2, jk); pdone = __shfl(v_done[q], jk);
                };
                Val e0;
                for (int c = 0; c < K; ++c) e0.v[c] = 0.0;
                bool ok0 = false;
                {
                    int prow, pdone;
                    parent_of(half, prow, pdone);
                    ok0 = stg && half < nps && tq >= 0 && tq < pdone;
                    if (ok0) e0 = t2_load(r, prow, tq);
                }
                {
                    const int lo = ws + k0, hi = min(lo + RK_NY, dr);
                    if (hi > lo && !(lo >= yhi[q] - RK_NY && hi <= yhi[q])) { y_reload(r, lo); yhi[q] = lo + RK_NY; }
                }
                if (stg && half < nps)
                    for (int c = 0; c < KP; ++c) sm.pst[0][half * RK_NY + sl][c] = ok0 ? e0.v[PC0 + c] : PO_NEG_INF;
                for (int k = 2; __builtin_expect(k < nps, 0); k += 2) {   // (wave-uniform)
                    const int kk = k + half;
                    int prow, pdone;
                    parent_of(kk, prow, pdone);
                    if (stg && kk < nps) {
                        const PVal val = pval_of(t2_read(r, prow, tq, tq < pdone));
                        for (int c = 0; c < KP; ++c) sm.pst[0][kk * RK_NY + sl][c] = val.v[c];
                    }
                }
                }
                if (k0 == 0 && want_seed && ws - 1 < se_hdr) for (int c = 0; c < K; ++c) self.v[c] = se.v[c];
                rk_sync();
                const int k1 = min(n1max, k0 + RK_NY);
                // ---- the block's chains in CLOSED FORM (a.chain_scan; the one-value model).  A new element starts absent
                // (PrefixTree.h:518-531) and x_t = logaddexp(p_{t-1} + ya_t, x_{t-1} + yb_t) unrolls to
                //     x_t = B_t + log( exp(x_{ws-1}) + sum_{s <= t} exp(p_{s-1} + ya_s - B_s) ),   B_t = yb_ws + .. + yb_t :
                // ONE exp per (element, time), a prefix sum in the probability domain, ONE log — instead of ~ 20 dependent
                // logaddexp on 4 of a read's 32 lanes.  A chain gets 8 lanes (lane mI of the group holds the J <= 4 consecutive
                // times J mI .. J mI + J - 1 of the block), the four children of a staged parent the four groups of a read's half
                // of the wave [NR = 2: the halves take two parents], so a pass serves one parent on both reads.  The sums are
                // scaled by the chain's largest term m; a chain whose finite terms span more than 600 nats (exp(c - m) would
                // lose them) or whose blank column holds -inf sends the whole step to the general scan (this function returns
                // false: nothing of the lanes' state has been changed by then, and what the passes before it stored is
                // written again) — a serial chain in THIS kernel would cost every phase registers.  NOT the reference's
                // rounding: the values differ from the serial chain's by ~ 1e-12 (less than the serial chain differs from the
                // exact value: scripts/check_chain_scan.cpp), inside north_star's edit tolerance.  The
                // SCAN = false instantiation below is the reference's chain, operation for operation.
                if constexpr (SCAN) {
                {
                    const int half = lane >> 5, cg = (lane >> 3) & 3, mI = lane & 7;
                    const double guard = (a.chain_scan == 2) ? -3.0 : -600.0;   // (2: the tests' way to the hand-over below — most chains span 3 nats)
                    const int nvr = min(n1 - k0, RK_NY);          // times of this lane's read in the block (<= 0: none)
                    const int J = (k1 - k0 + 7) >> 3;               // items per lane, 1 .. 4 (wave-uniform)
                    const double b0 = (mI & 1) ? 1.0 : 0.0, b1 = (mI & 2) ? 1.0 : 0.0, b2 = (mI & 4) ? 1.0 : 0.0;
                    const int cbcol = (MODEL == PO_MODEL_FLIPFLOP) ? cg + A : A;
                    const int i0 = J * mI, tb = ws + k0 + i0;       // this lane's first item: index in the block, time
                    // B_t over the block (every group computes its own: the column is the group's for the flip-flop model)
                    double Bj[4];
                    bool badB;
                    {
                        double run = 0.0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            Bj[j] = 0.0;
                            if (j < J) {
                                const double yv = yb_[((tb + j) & (RK_NY - 1)) * RK_YC + cbcol];
                                run += (i0 + j < nvr) ? yv : 0.0;
                                Bj[j] = run;
                            }
                        }
                        double tot;
                        const double e = rk_g8_excl(run, b0, b1, b2, tot);
#pragma unroll
                        for (int j = 0; j < 4; ++j) Bj[j] += e;
                        badB = !(tot > -1e300);   // (-inf or NaN in the column: the closed form has no B)
                    }
                    const unsigned long long fmask = __ballot(fresh[q]);
                    for (int k = 0; k < nps; k += (NR == 1) ? 1 : 2) {   // (wave-uniform)
                        const int kk = (NR == 1) ? k : k + half;
                        const int jk = pj(min(kk, RK_PS - 1));                      // the parent's beam slot
                        const int clane = ((NR == 1) ? (half << 5) : 0) | ((nb + A * jk + cg) & (NS - 1));   // the lane of this group's child
                        const bool gv = kk < nps && cg < A && ((fmask >> clane) & 1ull) != 0ull;
                        // this lane as a CHILD of the pass: its seed (an element again, or the block before) goes to its chain's lanes
                        const bool mine = fresh[q] && ((NR == 1) ? (myk == k) : ((myk & ~1) == k));
                        auto& cmine = sm.cres.e[(NR == 1) ? half : (myk & 1)][sym & (PO_A - 1)];
                        auto& cgrp = sm.cres.e[half][cg];
                        if (mine) cmine.last = self.v[0];
                        rk_sync();
                        const double sd = gv ? cgrp.last : PO_NEG_INF;
                        const int crow = sm.f_crow2[jk] * PO_A + cg;
                        const double* const pk_ = &sm.pst[(NR == 1) ? r : 0][min(kk, RK_PS - 1) * RK_NY][0];
                        double xj[4];
                        double m = PO_NEG_INF;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            xj[j] = PO_NEG_INF;
                            if (j < J) {
                                const double pkv = pk_[(i0 + j) * KP], ya = yb_[((tb + j) & (RK_NY - 1)) * RK_YC + cg];
                                const double cj = (pkv + ya) - Bj[j];
                                xj[j] = (gv && i0 + j < nvr) ? cj : PO_NEG_INF;
                                m = po_vmax(m, xj[j]);
                            }
                        }
                        m = po_vmax(rk_g8_max(m), sd);
                        bool bad = (gv && badB) || (sd > PO_NEG_INF && sd - m < guard);
                        double run = 0.0;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (j < J) {
                                const double d = xj[j] - m;
                                bad = bad || (xj[j] > PO_NEG_INF && d < guard);
                                run += lae.ex(d);
                                xj[j] = run;
                            }
                        double tot;
                        double e = rk_g8_excl(run, b0, b1, b2, tot);
                        if (__ballot(sd > PO_NEG_INF) != 0ull) e += lae.ex(sd - m);   // (wave-uniform: most passes have no seed)
                        KC(34, 1);
                        if (__builtin_expect(__ballot(bad) != 0ull, 0)) { KC(35, 1); return false; }   // (wave-uniform; NR = 2: read 0 may be done — a consistent state)
                        {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (j < J) xj[j] = (gv && i0 + j < nvr) ? (Bj[j] + m) + lae.lg(xj[j] + e) : PO_NEG_INF;
                            double xl = xj[0];   // the lane's last item: what the next lane's first one is compared with
                            if (J > 1) xl = xj[1];
                            if (J > 2) xl = xj[2];
                            if (J > 3) xl = xj[3];
                            double prev = rk_dpp_d<0x111>(xl);   // (row_shr:1)
                            prev = (mI == 0) ? sd : prev;
                            double lm = PO_NEG_INF;
                            int lt = -1, ltr = INT_MIN;
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (j < J) {
                                    const int t = tb + j;
                                    if (gv && i0 + j < nvr) {
                                        Val o;
                                        o.v[0] = xj[j];
                                        t2_write(r, crow, t, o); SH_WRITE(r, crow, t, sm.f_fc[jk] + cg);
                                        if (xj[j] > prev) ltr = t;
                                        if (xj[j] >= lm) { lm = xj[j]; lt = t; }
                                        if (i0 + j == nvr - 1) cgrp.last = xj[j];
                                    }
                                    prev = xj[j];
                                }
                            const double gm = rk_g8_max(lm);
                            const int gt = rk_g8_max_i((lm == gm) ? lt : -1);   // (the latest time of the maximum, as the serial chain leaves it)
                            const int gtr = rk_g8_max_i(ltr);
                            if (gv && mI == 0) { cgrp.mx = gm; cgrp.mt = gt; cgrp.tr = gtr; }
                        }
                        rk_sync();
                        if (mine && nvr > 0) {
                            const double bmx = cmine.mx;
                            self.v[0] = cmine.last;
                            if (bmx >= mx) { mx = bmx; mt = cmine.mt; }
                            tr = max(tr, cmine.tr);
                        }
                        rk_sync();
                    }
                }
                } else {
                // (the operands of an iteration are asked for one iteration ahead: a lone wave then waits for the LDS only
                //  inside logaddexp's own table lookups)
                double nya = 0.0, nyb = 0.0;
                PVal npp;
                for (int c = 0; c < KP; ++c) npp.v[c] = 0.0;
                if (fresh[q] && k0 < n1) {
                    const double* yrow = yb_ + ((ws + k0) & (RK_NY - 1)) * RK_YC;
                    nya = yrow[sym]; nyb = yrow[cb];
                    for (int c = 0; c < KP; ++c) npp.v[c] = ps_[c];
                }
                for (int k = k0; k < k1; ++k) {
                    if (fresh[q] && k < n1) {
                        const int t = ws + k;
                        const double ya = nya, yb = nyb;
                        const PVal pp = npp;
                        {   // (one past the end of the block: read, never used — the slots exist)
                            const double* yrow = yb_ + ((t + 1) & (RK_NY - 1)) * RK_YC;
                            nya = yrow[sym]; nyb = yrow[cb];
                            const double* pn = ps_ + min(k - k0 + 1, RK_NY - 1) * KP;
                            for (int c = 0; c < KP; ++c) npp.v[c] = pn[c];
                        }
                        const Val out = upd(self, pp, ya, yb, same);
#ifdef PO_RING_TRACE_NODE
                        if (pi == 0 && e_id == PO_RING_TRACE_NODE) printf("V %d %d %d %.17g %.17g %.17g P1\n", e_id, r, t, out.v[0], pp.v[0], self.v[0]);
#endif
                        t2_write(r, e_row2, t, out); SH_WRITE(r, e_row2, t, e_id);
                        if (out.v[0] > self.v[0]) tr = t;
                        self = out;
                        mt = (out.v[0] >= mx) ? t : mt;
                        mx = po_vmax(mx, out.v[0]);
                    }
                }
                }   // (!SCAN)
            }
            if (n1max == 0 && want_seed && ws - 1 < se_hdr) for (int c = 0; c < K; ++c) self.v[c] = se.v[c];
            {   // (the slot's address from the lane number HERE: po_lane_here)
                const int ln = po_lane_here(), sh = (NR == 1) ? (ln & 31) : ln, rh = (NR == 1) ? (ln >> 5) : r;
                if (sh < PF0N) {
                    sm.pf0_t[rh][sh] = want_pf ? ws : -1;
                    if (want_pf) sm.pf0[rh][sh] = se.v[0];
                }
            }
            // the fresh lanes are ordinary continuing lanes now, ending at dr like everybody else: the run loop does the step
            if (fresh[q]) {
                v_done[q] = dr; v_self[q] = self; v_fresh[q] = 0;
                v_mx[q] = mx; v_mt[q] = mt; v_td[q] = max(ws, tr);
            }
            if constexpr (COUNT) {
                int tot = fresh[q] ? n1 : 0;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) tot += __shfl_xor(tot, off);
                cnt_x += (unsigned)tot;
            }
            }   // (q)
            pf0_step = mstep;
            KT(2);
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
            for (int i = 1; i < WS; ++i) mysel = (s == i) ? sel[i] : mysel;
            const bool rb = s < nbn;                       // this lane is a beam slot of the new table
            const bool rc = !rb && s < nen;                // ... a child slot
            const int j = rc ? (((s - nbn) * divA) >> 16) : 0, c = rc ? (s - nbn) - j * A : 0;
            int pj = sel[0];
#pragma unroll
            for (int i = 1; i < WS; ++i) pj = (j == i) ? sel[i] : pj;
            const int srcb = rb ? mysel : 0;
            int n_id = __shfl(e_id, hb | srcb), n_row2 = __shfl(e_row2, hb | srcb), n_sym = __shfl(e_sym, hb | srcb);
            int n_fc = sm.f_fc[srcb], n_crow2 = sm.f_crow2[srcb], n_par = sm.f_par[srcb];
            int n_gpar = sm.f_gpar[srcb], n_prow2 = sm.f_prow2[srcb], n_depth = sm.f_depth[srcb];
            KT(40);
            // ---- every old element marks its row group with the times it has written there
#pragma unroll
            for (int q = 0; q < NR; ++q)
                if (live && v_fresh[q] == 0) atomicMax(&g_hi(RD(q))[e_row2 >> 2], v_done[q]);
            // ---- B. expansion of the new beam nodes
            KC(20, __ballot(rb && n_fc == -2) != 0ull ? 1 : 0);
            if (__builtin_expect(__ballot(rb && n_fc == -2) != 0ull, 0)) {   // (wave-uniform branch: see po_settle)
                if (rb && n_fc == -2) { n_fc = afc[n_id]; n_crow2 = acrow[n_id]; }   // a node whose parent re-entered: the arena knows
                po_settle(n_fc, n_crow2);
            }
            rk_sync();
            KT(49);
            bool isnew = false, need_group = false;
            if (rb) {
                isnew = n_fc < 0;
                need_group = isnew || n_crow2 < 0 || n_crow2 >= NG || sm.g_owner[n_crow2] != n_id;   // (old rows recycled: all dead)
            }
            KC(36, sm_pop(smask_of(rb && mysel >= nbo))); KC(37, sm_pop(smask_of(rb && mysel >= nbo && !isnew))); KC(38, sm_pop(smask_of(rb && mysel >= nbo && !isnew && !need_group)));
            {
                const SMask bn = smask_of(isnew);
                if (isnew) {
                    n_fc = next_id + A * sm_pop(bn & (((SMask)1 << s) - (SMask)1));
                    if (lo_half) afc[n_id] = n_fc;
                }
                next_id += A * sm_pop(bn);
                if (__builtin_expect(next_id > arena_cap, 0)) st = PO_E_NOMEM;   // (the slice's arena is full: beam2d_kernel takes the pair)
                if (rb && !need_group) { atomicMax(&sm.g_hi0[n_crow2], nce); atomicMax(&sm.g_hi1[n_crow2], nre); }
                rk_sync();
                KT(50);
                const SMask hg = smask_of(need_group);
                if (hg != 0) {   // (wave-uniform)
                    // Round 6: every row group the step needs in ONE pass.  (Rounds 4 - 5 handed them out one after the other — walk
                    // the table from a cursor with three dependent LDS reads per entry, a fence, lane 0 writes the entry, a fence:
                    // 61 % of the table build's time, 12 % of the kernel's, profiles/r06_ab_row_groups.txt.)  Lane l looks at groups
                    // l, l + 64, ..: one LDS round trip for the whole table, a ballot per 64 groups; the k-th beam slot that needs a
                    // group takes the k-th free one (scalar bit work) and writes its entry itself.  WHICH free group a parent gets
                    // only names rows of the store: results do not depend on it.
                    constexpr int GPL = (RK_NGL + 63) / 64;
                    unsigned long long fm[GPL];
#pragma unroll
                    for (int g = 0; g < GPL; ++g) {
                        const int cc = lane + 64 * g;
                        bool fr = false;
                        if (cc < NG) {
                            const int ow = sm.g_owner[cc], h0 = sm.g_hi0[cc], h1 = sm.g_hi1[cc];   // (asked for together)
                            fr = ow < 0 || (h0 <= nu - 1 && h1 <= nv - 1);
                        }
                        fm[g] = __ballot(fr);
                    }
                    int mygg = 0;
                    for (SMask h2 = hg; h2 != 0; h2 &= h2 - (SMask)1) {   // (wave-uniform; one trip per group handed out)
                        const int jj = sm_ctz(h2);
                        int gg = -1;
#pragma unroll
                        for (int g = 0; g < GPL; ++g)
                            if (gg < 0 && fm[g] != 0ull) { gg = 64 * g + (int)__builtin_ctzll(fm[g]); fm[g] &= fm[g] - 1ull; }
                        KC(47, 1);
                        if (gg < 0) { st = PO_E_NOMEM; gg = 0; }   // (none free: beam2d_kernel takes the pair)
                        if (s == jj) mygg = gg;
                        if (lane < 2 * PO_A) rowhdr[gg * 2 * PO_A + lane] = RK_FRESH;   // (the group's rows hold nothing of their new owners yet)
                    }
                    rk_sync();   // (every lane has looked at the table before its entries change)
                    if (need_group) {
                        if (lo_half) { sm.g_owner[mygg] = n_id; sm.g_hi0[mygg] = nce; sm.g_hi1[mygg] = nre; acrow[n_id] = mygg; }
                        n_crow2 = mygg;
                    }
                    rk_sync();
                }
            }
            KT(41);
            // ---- C. children slots take their parent's (new) fields
            const int p_id = __shfl(n_id, hb | j), p_fc = __shfl(n_fc, hb | j), p_crow2 = __shfl(n_crow2, hb | j);
            const int p_sym = __shfl(n_sym, hb | j), p_par = __shfl(n_par, hb | j), p_row2 = __shfl(n_row2, hb | j);
            const int p_depth = __shfl(n_depth, hb | j);
            const bool p_isnew = __shfl((int)isnew, hb | j) != 0;
            // (children of a parent that got a NEW row group have nothing stored, whether the nodes are new or not)
            const bool p_newrows = __shfl((int)(isnew || need_group), hb | j) != 0;
            int n_alias = -1, n_ps = PS_FROZEN;
            if (rc) {
                n_id = p_fc + c; n_row2 = p_crow2 * PO_A + c; n_sym = sym_pack(c, sym_last(p_sym), false);
                n_par = p_id; n_gpar = p_par; n_prow2 = p_row2; n_depth = p_depth + 1; n_ps = j;
                n_fc = p_isnew ? -1 : -2; n_crow2 = p_isnew ? -1 : -2;
                if (p_isnew && lo_half && n_id < arena_cap) { apl[n_id] = po_pack_node(p_id, c); afc[n_id] = -1; acrow[n_id] = -1; }
            }
            // a child slot whose node is also a beam slot is the same node pushed twice (Beam::prune's std::unique)
            for (int i = 0; i < nbn; ++i) {
                const int bid = __builtin_amdgcn_readlane(n_id, i);
                if (rc && bid == n_id) n_alias = i;
            }
            KT(42);
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
            KT(43);
            // ---- E. an old element that continues in no slot of the new table LEAVES: where its stored values end goes into its
            // row's header — whoever asks for them while it is no element (a frozen parent's older values, its own seed should it
            // become an element again) finds the answer there.  (A handful of lanes per table build: writing every element's
            // header every time cost 12 GB of 4-byte stores per 10 000-pair launch.)
            {
                if (lo_half) sm.tie.ord[s] = 0;
                rk_sync();
                if (lo_half && nlive && src >= 0) sm.tie.ord[src] = 1;
                rk_sync();
                const bool leaving = live && sm.tie.ord[s] == 0;
#pragma unroll
                for (int q = 0; q < NR; ++q)
                    if (leaving && v_fresh[q] == 0) *hdr_of(RD(q), e_row2) = v_done[q];
                rk_sync();   // (ord is the tie replay's scratch as well)
            }
            KT(44);
            // ---- F. the lanes take their new identity
            const int gsrc = hb | max(src, 0);
            const int g_fc = sm.f_fc[max(src, 0)], g_crow2 = sm.f_crow2[max(src, 0)];
            const int op = __shfl(e_ps, gsrc);                     // the parent's slot in the old table (or ROOT / FROZEN)
            const int opl = hb | max(op, 0);
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const int g_done = __shfl(v_done[q], gsrc), g_fresh = __shfl(v_fresh[q], gsrc), g_mt = __shfl(v_mt[q], gsrc), g_td = __shfl(v_td[q], gsrc);
                const Val g_self = val_shfl(v_self[q], gsrc);
                const double g_mx = __shfl(v_mx[q], gsrc);
                // the last value of the node's parent as the old table knew it: of the parent's lane if it was an element,
                // else what was captured when it stopped being one
                const int o_done = __shfl(v_done[q], opl), o_fresh = __shfl(v_fresh[q], opl);
                const PVal o_last = pval_shfl(pval_of(v_self[q]), opl);
                const PVal q_val = pval_shfl(fz_val[q], gsrc);
                const int q_t = __shfl(fz_t[q], gsrc);
                PVal c_val;   // (value by value: a conditional between two aggregates becomes a select of scratch addresses)
                for (int c2 = 0; c2 < KP; ++c2) c_val.v[c2] = (op >= 0) ? o_last.v[c2] : q_val.v[c2];
                const int c_t = (op >= 0) ? ((o_fresh == 0) ? o_done - 1 : INT_MAX) : q_t;
                fz_val[q] = c_val; fz_t[q] = (nlive && src >= 0) ? c_t : INT_MAX;
                if (nlive && src >= 0) {
                    v_done[q] = g_done; v_fresh[q] = g_fresh; v_mt[q] = g_mt; v_td[q] = g_td; v_self[q] = g_self; v_mx[q] = g_mx;
                } else {
                    v_done[q] = RK_FRESH; v_fresh[q] = (rc && p_newrows) ? 2 : 1;
                    v_self[q] = val_neg(); v_mx[q] = PO_NEG_INF; v_mt[q] = -1; v_td[q] = 0;
                }
            }
            e_id = n_id; e_row2 = n_row2; e_sym = n_sym;
            if (rc && src >= 0) { n_fc = g_fc; n_crow2 = g_crow2; }   // a continuing child keeps what is known about its own children
            rk_sync();   // (every lane has read the old table's fields)
            if (lo_half) {
                sm.f_fc[s] = n_fc; sm.f_crow2[s] = n_crow2; sm.f_par[s] = n_par; sm.f_gpar[s] = n_gpar; sm.f_prow2[s] = n_prow2;
                sm.f_depth[s] = n_depth; sm.f_alias[s] = rc ? n_alias : -1;
            }
            live = nlive;
            KT(45);
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
            KT(46);
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
#pragma unroll
                    for (int q = 0; q < NR; ++q) {
                        const int r = RD(q);
                        const int wsn = r ? nv : nu;
                        if (rew && live && v_fresh[q] == 0 && v_done[q] > wsn) {
                            SH_CHK(r, e_row2, wsn - 1, true, e_id, 10);
                            v_self[q] = read_own_all(r, wsn - 1, true);   // (wsn - 1 < v_done: its own)
                            v_done[q] = wsn;
                        }
                        for (int k2 = 0; k2 < K; ++k2) po_settle(v_self[q].v[k2]);   // (the loads are waited for in THIS branch: po_settle)
                    }
                }
            }
            {
                bool fr = false;
#pragma unroll
                for (int q = 0; q < NR; ++q) fr = fr || v_fresh[q] != 0;
                tbl_fresh = __ballot(live && fr) != 0ull;
            }
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
#ifdef PO_EMU_SHADOW
            sh_step = mstep; sh_u = u; sh_v = v;
#endif
            double sc = PO_NEG_INF;
            bool viol = false, run_viol = false;
            KT(6);
            // ---- a RUN of main steps on the table as it stands.  After a step that kept the set of beam nodes, with every
            // live lane's values ending at the same time and nothing to catch up, the next step is: the new times of the two
            // windows (often none on a read: the envelope's window ends move a base at a time) in lockstep, the window maxima
            // from what is carried, the score, the one comparison per child.  The run ends at the first step that is not of
            // this kind (it is then done below) or that changes the beam (it is ranked below).
            if (__builtin_expect(!tbl_fresh && !tbl_uneven && nb == W && __ballot(live && e_ps == PS_ROOT) == 0ull, 1)) {
                const int sym = sym_last(e_sym), cb = (MODEL == PO_MODEL_FLIPFLOP) ? sym + A : A;
                const bool same = sym_plast(e_sym) == sym;
                const int plane = (e_ps >= 0) ? (hb | e_ps) : lane;
                const bool child = s >= nb;
                const bool fzl = e_ps < 0;
                double pf_e[NR];   // the value requested at the end of the previous step of this run
                int pf_t[NR];
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    pf_e[q] = 0.0; pf_t[q] = -1;
                    if (pf0_step == mstep) {   // ... or with the staging of the new elements' step just before this run  (wave-uniform)
                        const int ln = po_lane_here(), sh = (NR == 1) ? (ln & 31) : ln, rh = (NR == 1) ? (ln >> 5) : RD(q);
                        if (sh < PF0N) {
                            pf_t[q] = sm.pf0_t[rh][sh];
                            if (pf_t[q] >= 0) pf_e[q] = sm.pf0[rh][sh];
                        }
                    }
                }
                for (;;) {
                    u = rec.x; v = rec.y; ce = rec.z; re = rec.w;
                    const int d0 = done0(), d1 = done1();
                    if (!(u <= d0 && d0 <= ce && v <= d1 && d1 <= re) || mstep + 1 >= nmain) { KC(28, 1); break; }
                    // (a frozen parent's older values would have to come from the store: only asked when there are new times)
                    if (ce > d0 || re > d1) {
                        bool old = false;
#pragma unroll
                        for (int q = 0; q < NR; ++q) {
                            const int r = RD(q);
                            const int dr = r ? d1 : d0, we = r ? re : ce;
                            old = old || (live && dr < we && e_ps == PS_FROZEN && dr - 1 < fz_t[q]);
                        }
                        if (__builtin_expect(__ballot(old) != 0ull, 0)) { KC(29, 1); break; }
                    }
                    // ---- the carried part [ws, dr) of the window: its maximum is what the previous step left while that time
                    // is inside the window; else the stored values are looked at (the one most steps need — a decaying
                    // element's value at the window start — was requested a step ago: pf)
                    // A CHILD keeps its old maximum when that time has left the window: the maximum over an older, larger
                    // window is an upper bound, and a child only has to stay below the smallest beam score — its exact
                    // maximum is looked up below, if the bound does not settle that.  (40 of 50 lanes never ask the store.)
                    bool has_c[NR], bnd[NR];
                    double mx[NR], cmx[NR], pf_val[NR];
                    int mt[NR], cmt[NR], td[NR], tr[NR];
#pragma unroll
                    for (int q = 0; q < NR; ++q) {
                        const int r = RD(q);
                        const int ws = r ? v : u, dr = r ? d1 : d0;
                        has_c[q] = live && dr > ws;
                        bnd[q] = has_c[q] && child && !(v_mx[q] == PO_NEG_INF || v_mt[q] >= ws);
                        mx[q] = PO_NEG_INF; cmx[q] = PO_NEG_INF; mt[q] = -1; cmt[q] = -1; td[q] = has_c[q] ? v_td[q] : ws; tr[q] = INT_MIN;
                        // (Round 6: every value that comes from the store is waited for INSIDE a wave-uniform branch that is taken only
                        //  when some lane asked for one — po_settle.  Written as `pf_val = pf_t >= 0 ? pf_e : -inf` and a load in one arm of
                        //  a per-lane branch, the compiler's s_waitcnt vmcnt(0) sat on the common path behind the joins: one in front of the
                        //  carried maxima and TWO behind the new times' loop — where it waited for that loop's stores, a full round trip
                        //  per step.)
                        pf_val[q] = PO_NEG_INF;
                        if (__ballot(pf_t[q] >= 0) != 0ull) {   // (a value was requested a step ago)
                            double pe = pf_e[q];
                            po_settle(pe);
                            pf_val[q] = (pf_t[q] >= 0) ? pe : PO_NEG_INF;
                        }
                        const bool keep = has_c[q] && (v_mx[q] == PO_NEG_INF || v_mt[q] >= ws || child);
                        const bool gone = has_c[q] && !keep;             // the carried maximum has left the window ...
                        const bool one = gone && !(td[q] > ws);          // ... of a decaying element: the maximum is its value at ws
                        const bool rsc = gone && !one;
                        if (keep) { cmx[q] = v_mx[q]; cmt[q] = v_mt[q]; }
                        const bool ld = one && ws != pf_t[q];            // ... which was not the one requested
                        double lv = PO_NEG_INF;
                        if (__builtin_expect(__ballot(ld) != 0ull, 0)) {
                            if (ld) { SH_CHK(r, e_row2, ws, true, e_id, 1); lv = read_own(r, ws); }
                            po_settle(lv);
                        }
                        if (one) { cmx[q] = ld ? lv : pf_val[q]; cmt[q] = ws; }
                        rescan_wave(q, rsc, ws, dr, cmx[q], cmt[q], td[q]);
                    }
                    // ---- the new times [dr, we), everybody in lockstep: the parent's previous value comes from its lane
                    // (NR = 1: the two reads side by side in the halves of the wave; NR = 2: one after the other)
#pragma unroll
                    for (int q = 0; q < NR; ++q) {
                        const int r = RD(q);
                        const int we = r ? re : ce, dr = r ? d1 : d0;
                        const int n2 = we - dr;   // (uniform over the read's lanes, >= 0)
                        const int n2max = (NR == 1) ? max(ce - d0, re - d1) : n2;
                        const double* const yb_ = &sm.ybuf[r][0][0];
                        Val self = v_self[q];
                        // (a frozen parent's captured value can only be asked for by the first new time — the test above — : one select
                        //  per iteration instead of two compares and two)
                        PVal pp_fz = (fzl && dr - 1 == fz_t[q]) ? fz_val[q] : pval_neg();
                        for (int k0 = 0; k0 < n2max; k0 += RK_NY) {   // (blocks of RK_NY times: the y rows are loaded between the loops)
                            {
                                const int lo = dr + k0, hi = min(lo + RK_NY, we);
                                rk_sync();
                                if (__builtin_expect(hi > lo && !(lo >= yhi[q] - RK_NY && hi <= yhi[q]), 0)) { y_reload(r, lo); yhi[q] = lo + RK_NY; }
                                rk_sync();
                            }
                            const int k1 = min(n2max, k0 + RK_NY);
                            for (int k = k0; k < k1; ++k) {
                                const int t = dr + k;
                                const PVal ps_self = pval_shfl(pval_of(self), plane);
                                if (live && k < n2) {
                                    const double* yrow = yb_ + (t & (RK_NY - 1)) * RK_YC;
                                    const double ya = yrow[sym], yb = yrow[cb];
                                    const PVal pp = fzl ? pp_fz : ps_self;
                                    pp_fz = pval_neg();
                                    const Val out = upd(self, pp, ya, yb, same);
#ifdef PO_RING_TRACE_NODE
                                    if (pi == 0 && e_id == PO_RING_TRACE_NODE) printf("V %d %d %d %.17g %.17g %.17g RUN ps %d fzt %d\n", e_id, r, t, out.v[0], pp.v[0], self.v[0], e_ps, fz_t[q]);
#endif
                                    t2_write(r, e_row2, t, out); SH_WRITE(r, e_row2, t, e_id);
                                    if (out.v[0] > self.v[0]) tr[q] = t;
                                    self = out;
                                    mt[q] = (out.v[0] >= mx[q]) ? t : mt[q];
                                    mx[q] = po_vmax(mx[q], out.v[0]);
                                }
                            }
                        }
                        if (live && dr < we) { v_done[q] = we; v_self[q] = self; }
                    }
                    double nmx[NR];   // the maximum over the new times alone
                    int nmt[NR];
#pragma unroll
                    for (int q = 0; q < NR; ++q) {
                        nmx[q] = mx[q]; nmt[q] = mt[q];
                        if (has_c[q] && !(mx[q] >= cmx[q])) { mx[q] = cmx[q]; mt[q] = cmt[q]; }   // (new values, later in time, win ties)
                        if (live) { v_mx[q] = mx[q]; v_mt[q] = mt[q]; v_td[q] = max(td[q], tr[q]); }
                        smx[q] = live ? mx[q] : PO_NEG_INF;
                    }
                    if constexpr (COUNT) {
                        cnt_ref += (unsigned)(ne * ((ce - u) + (re - v)));
                        if (NR == 1) cnt_x += (unsigned)(__popcll(__ballot(live && lane < 32)) * (ce - d0) + __popcll(__ballot(live && lane >= 32)) * (re - d1));
                        else cnt_x += (unsigned)(__popcll(__ballot(live)) * ((ce - d0) + (re - d1)));
                    }
                    auto score = [&]() -> double { return (NR == 1) ? po_sum32(smx[0]) : smx[0] + smx[NR - 1]; };
                    sc = score();
                    const double scmin = rk_row0_min(sc, nb, lane);
                    viol = live && child && !(scmin > sc);
                    // a child that reaches the smallest beam score on a bound: its exact maximum now (the stored values are
                    // looked at), then the score and the test again — the decision is the one exact maxima give
                    bool hot[NR], hot_any = false;
#pragma unroll
                    for (int q = 0; q < NR; ++q) { hot[q] = viol && bnd[q] && !(nmx[q] >= cmx[q]); hot_any = hot_any || hot[q]; }
                    if (__builtin_expect(__ballot(hot_any) != 0ull, 0)) {
#pragma unroll
                        for (int q = 0; q < NR; ++q) {
                            const int r = RD(q);
                            const int ws = r ? v : u, dr = r ? d1 : d0;
                            double cx = PO_NEG_INF;
                            int ct = -1, td2 = td[q];
                            const bool rs2 = hot[q] && carried_one(r, ws, cx, ct, td2, pf_t[q], pf_val[q]);
                            po_settle(cx);   // (inside this branch: see the carried part above)
                            rescan_wave(q, rs2, ws, dr, cx, ct, td2);
                            if (hot[q]) {
                                const bool keep = (nmx[q] >= cx);
                                v_mx[q] = keep ? nmx[q] : cx; v_mt[q] = keep ? nmt[q] : ct; v_td[q] = max(td2, tr[q]);
                                smx[q] = v_mx[q];
                            }
                        }
                        sc = score();
                        viol = live && child && !(scmin > sc);
                        KC(31, 1);
                    }
#ifdef PO_RING_TRACE
                    if (pi == 0 && live && lo_half) printf("T %d %d %d %.17g\n", u, v, e_id, sc);
#endif
                    up = u; vp = v;
                    mstep++;
                    if (__builtin_expect((mstep & 63) == 0, 0)) {
                        rcur = rnxt;
                        rnxt = rec_load(mstep + 64 + lane);
                    }
                    rec = rec_at(min(mstep, nmain - 1));
                    // the stored value the next step's carried maximum will ask for, if any, requested now: the beam lanes,
                    // and a child whose bound has just had to be made exact (it will be again)
                    // (not on the step that leaves the run: the ranking and the table build would find a load in flight — po_settle)
                    const bool leave = __ballot(viol) != 0ull;
#pragma unroll
                    for (int q = 0; q < NR; ++q) {
                        const int r = RD(q);
                        const int wsn = r ? rec.y : rec.x;
                        pf_t[q] = -1;
                        if (!leave && live && (!child || hot[q]) && v_done[q] > wsn && v_mx[q] != PO_NEG_INF && v_mt[q] < wsn) { pf_t[q] = wsn; pf_e[q] = t2_load0(r, e_row2, wsn); SH_CHK(r, e_row2, wsn, true, e_id, 11); }
                    }
                    KC(12, 1);
                    if (leave) { run_viol = true; break; }
                }
                KT(0);
            }
            if (!run_viol) {
            u = rec.x; v = rec.y; ce = rec.z; re = rec.w;
            // ---- catch-up steps between the previous main step and this one (:314-336): only the beam nodes, one time
            // at a time; a time the last main step's window covered is a no-op (the bits are there)
            {
                const int nbe = min(W, nb);
                const int d0 = done0(), d1 = done1();
                if constexpr (COUNT) cnt_ref += (unsigned)((max(u - up - 1, 0) + max(v - vp - 1, 0)) * nbe);
                if (__builtin_expect(u - 1 >= max(up + 1, d0), 0)) { scan(false, up + 1, u, 0, 0, nbe); tbl_uneven = true; }
                if (__builtin_expect(v - 1 >= max(vp + 1, d1), 0)) { scan(false, 0, 0, vp + 1, v, nbe); tbl_uneven = true; }
            }
            // ---- MAIN step at (u, v): windows [u, ce) x [v, re)  (:342-375)
            // (new elements first: their windows up to where everybody else stands — then the step is an ordinary one)
            if (__builtin_expect(tbl_fresh && !tbl_uneven && scan_new(u, ce, v, re), 1)) {
                tbl_fresh = false;
                continue;
            }
            KC(30, tbl_fresh ? 0 : 1); KC(32, tbl_uneven ? 1 : 0); KC(33, nb != W ? 1 : 0);
            scan(true, u, ce, v, re, NS);
            tbl_fresh = false; tbl_uneven = false;
            if constexpr (COUNT) cnt_ref += (unsigned)(ne * ((ce - u) + (re - v)));
            // node_greater_max_sym: max over read 0's window + max over read 1's
            sc = (NR == 1) ? po_sum32(smx[0]) : smx[0] + smx[NR - 1];
#ifdef PO_RING_TRACE   // debugging builds only: every candidate's score before the prune
            if (pi == 0 && live && lo_half) printf("T %d %d %d %.17g\n", u, v, e_id, sc);
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
            const SMask cm = smask_of(cand);
            const int ncand = sm_pop(cm);
            // Only the beam nodes and the children that reach the smallest beam score can be among the W best (every
            // other child has W candidates above it), and nothing outside that set outranks a member of it: the ranks
            // are taken within it (a handful of candidates instead of W * (A + 1)).
            // (the 1-D kernels' higher threshold — the smallest FAMILY maximum, a beam node and its children — was tried here in
            //  round 5: a ranking sees 10.9 candidates at W = 5 and would see ~ 6, but the permutes that find the family maxima
            //  cost more than the trips they save: 51.9 against 51.1 ms)
            SMask smask = cm;
            if (nb == W) {
                const double thr = rk_row0_min(sc, nb, lane);
                smask = smask_of(cand && (s < nb || sc >= thr));
            }
            int rank = 0, neq = 0;
            for (SMask mm = smask; mm != 0; mm &= mm - (SMask)1) {
                const int o = sm_ctz(mm);
                const double so = rk_readlane_d(sc, o);
                const int io = __builtin_amdgcn_readlane(e_id, o);
                rank += ((so > sc) | (!(sc > so) & (io < e_id))) ? 1 : 0;
                neq += (so == sc) ? 1 : 0;
            }
            if (!((smask >> s) & (SMask)1)) { rank = 64; neq = 0; }
            const int nbn = min(W, ncand);
#pragma unroll
            for (int jx = 0; jx < WS; ++jx) {
                const SMask bj = smask_of(cand && rank == jx);
                sel[jx] = (bj != 0) ? sm_ctz(bj) : 0;
            }
            if (__builtin_expect(__ballot(cand && neq > 1 && rank < W) != 0ull, 0)) {
                // exact ties reaching into the beam: what libstdc++'s partial_sort / sort leave on the candidates in
                // creation order (po_device.h), replayed by one lane
                int pos = 0;   // (the replay runs over ALL candidates in creation order)
                for (int o = 0; o < ne; ++o) {
                    const int io = __builtin_amdgcn_readlane(e_id, o);
                    pos += (int)((cm >> o) & (SMask)1) & ((io < e_id) ? 1 : 0);
                }
                if (cand && lo_half) { sm.tie.ord[pos] = s; sm.tie.csc[s] = sc; }
                rk_sync();
                if (lane == 0) {
                    const double* cp = sm.tie.csc;
                    po_stl_prune<WS>(sm.tie.ord, ncand, W, [&](int slot) { return cp[slot]; });
                }
                rk_sync();
#pragma unroll
                for (int jx = 0; jx < WS; ++jx) sel[jx] = (jx < nbn) ? sm.tie.ord[jx] : 0;
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
            if (a.defer_count) atomicAdd(a.defer_count, 1ull);
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
        for (int i = 0; i < 56; ++i) a.dbg[i] = tk[i];
#endif
    if (COUNT && lane == 0 && a.upd_count) { atomicAdd(a.upd_count, sm.nupd); atomicAdd(a.upd_count + 1, sm.nupd_x); }
    // (everything this wave wrote into the slice leaves this XCD's L2 before another wave — any CU, any XCD — may claim it)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#ifndef PO_EMU
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    if (lane == 0) {
        const unsigned t = atomicAdd(&a.slice_tickets[1], 1u) % (unsigned)a.nslices;
        while (atomicCAS(&a.slice_claim[t], -1, slotid) != -1) __builtin_amdgcn_s_sleep(8);   // (its taker has a ticket: the word empties)
    }
}

// pair slots per CU (registers and LDS decide): 16 one-wave workgroups for the one-value model at W <= 6 (4 waves per SIMD),
// 12 for the three-value models and for 7 <= W <= 12 (3 waves per SIMD), 8 where both hold
namespace {
template <int MODEL, int NR>
int reg_occupancy() {
    int nblk = 0;
    constexpr int WPG = RegCfg<MODEL, NR>::WPG;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)beam2d_reg_kernel<MODEL, NR, false>, 64 * WPG, 0) != hipSuccess || nblk <= 0)
        nblk = 4 * RegCfg<MODEL, NR>::WAVES;
    else nblk *= WPG;
    if (const char* e = getenv("PO_REG_PER_CU")) { const int v = atoi(e); if (v > 0 && v < nblk) nblk = v; }
    if (getenv("PO_DEBUG_OCC")) fprintf(stderr, "[po] beam2d_reg_kernel<model %d, %d read(s) per lane>: %d resident pair waves per CU (workgroups of %d), %zu B of LDS per workgroup\n", MODEL, NR, nblk, WPG, sizeof(RegGroup<MODEL, NR>));
    return nblk;
}
template <int MODEL, int NR>
void reg_launch_model(const X2Args& a, int slots, hipStream_t stream) {
    constexpr int WPG = RegCfg<MODEL, NR>::WPG;
    const dim3 grid((slots + WPG - 1) / WPG), block(64 * WPG);   // (a.reg_slots = slots: the waves beyond it leave at once)
    if constexpr (RegCfg<MODEL, NR>::K == 1) {
        if (a.chain_scan && a.upd_count == nullptr) {   // (po_set_chain_mode(PO_CHAIN_CLOSED_FORM); the counting build is the serial chain's)
            hipLaunchKernelGGL((beam2d_reg_kernel<MODEL, NR, false, true>), grid, block, 0, stream, a);
            return;
        }
    }
    if (a.upd_count != nullptr) hipLaunchKernelGGL((beam2d_reg_kernel<MODEL, NR, true, false>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((beam2d_reg_kernel<MODEL, NR, false, false>), grid, block, 0, stream, a);
}
}  // namespace
// PO_REG_TU — poreover_amd/build.py compiles this file TWICE: 1 = the 32-slot kernels and the C entry points, 2 = the 64-slot kernels
// behind the two functions below (undefined: one translation unit with everything — the tools' and the emulator's build).  Two
// objects because they are compiled with different scheduler options (build.py: -amdgpu-use-amdgpu-trackers gains 1 % on the 32-slot
// one-value kernel and loses 1.5 % on the 64-slot one, profiles/r06_ab_compiler_flags.txt) and the option is per translation unit.
extern "C" __attribute__((visibility("hidden"))) int po_reg_wide_occupancy(int mi);
extern "C" __attribute__((visibility("hidden"))) void po_reg_wide_launch(const void* x2args, int slots, int model, hipStream_t stream);
#if !defined(PO_REG_TU) || PO_REG_TU == 2
extern "C" int po_reg_wide_occupancy(int mi) {
    return mi == 0 ? reg_occupancy<PO_MODEL_CTC, 2>() : (mi == 1 ? reg_occupancy<PO_MODEL_MERGE, 2>() : reg_occupancy<PO_MODEL_FLIPFLOP, 2>());
}
extern "C" void po_reg_wide_launch(const void* x2args, int slots, int model, hipStream_t stream) {
    const X2Args& a = *(const X2Args*)x2args;
    if (model == PO_MODEL_CTC) reg_launch_model<PO_MODEL_CTC, 2>(a, slots, stream);
    else if (model == PO_MODEL_MERGE) reg_launch_model<PO_MODEL_MERGE, 2>(a, slots, stream);
    else reg_launch_model<PO_MODEL_FLIPFLOP, 2>(a, slots, stream);
}
#endif
#if !defined(PO_REG_TU) || PO_REG_TU == 1
// wide != 0: the 64-slot layout (7 <= W <= 12)
extern "C" int po_reg_slots_per_cu(int model, int wide) {
#ifdef PO_EMU
    return (model == PO_MODEL_CTC ? 16 : 12) / (wide ? 2 : 1);
#else
    static PoPerDeviceCache<6> per_cu;
    const int mi = model == PO_MODEL_CTC ? 0 : (model == PO_MODEL_MERGE ? 1 : 2);
    return per_cu.get(mi * 2 + (wide ? 1 : 0), [mi, wide] {
        if (wide) return po_reg_wide_occupancy(mi);
        return mi == 0 ? reg_occupancy<PO_MODEL_CTC, 1>() : (mi == 1 ? reg_occupancy<PO_MODEL_MERGE, 1>() : reg_occupancy<PO_MODEL_FLIPFLOP, 1>());
    });
#endif
}
extern "C" int po_reg_max_elements(int wide) { return wide ? 64 : 32; }
extern "C" int po_reg_ngl(int wide) { return wide ? 2 * PO_REG_NGL : PO_REG_NGL; }
// bytes of value store per pair slot: 128 row groups at R = 128 of 8-byte (24-byte: three values) entries for W <= 6, twice that
// for the wide form
extern "C" size_t po_reg_pool_bytes(int model, int wide) { return (size_t)(model == PO_MODEL_CTC ? 1 : 3) << (wide ? 21 : 20); }
// `slots` pair slots (one-wave workgroups), each with its own store slice and arena
extern "C" void po_reg_launch(const void* x2args, int slots, int model, int wide, hipStream_t stream) {
    X2Args a = *(const X2Args*)x2args;
    a.reg_slots = slots;
#ifdef PO_REG_TIMING
    static long long* dbg = nullptr;
    if (!dbg) { (void)hipMalloc((void**)&dbg, 56 * sizeof(long long)); }
    (void)hipMemsetAsync(dbg, 0, 56 * sizeof(long long), stream);
    a.dbg = dbg;
#endif
    if (wide) {
        po_reg_wide_launch(&a, slots, model, stream);
    } else {
        if (model == PO_MODEL_CTC) reg_launch_model<PO_MODEL_CTC, 1>(a, slots, stream);
        else if (model == PO_MODEL_MERGE) reg_launch_model<PO_MODEL_MERGE, 1>(a, slots, stream);
        else reg_launch_model<PO_MODEL_FLIPFLOP, 1>(a, slots, stream);
    }
#ifdef PO_REG_TIMING
    {
        long long h[56];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "[po_reg_timing] pair slot 0, 10 ns ticks\n");
        fprintf(stderr, "   run loop %lld (%lld steps)\n", h[0], h[12]);
        fprintf(stderr, "   steps with new elements: %lld steps; staging + carried maxima %lld, phase 1 %lld (%lld iterations)\n", h[13], h[1], h[2], h[14]);
        fprintf(stderr, "   general scans: main %lld ticks (%lld), catch-up %lld ticks (%lld); iterations %lld\n", h[4], h[16], h[5], h[17], h[18]);
        fprintf(stderr, "   step top + score + prune test %lld, ranking %lld, rebuild %lld, pair setup + label %lld\n", h[6], h[7], h[8], h[9]);
        fprintf(stderr, "   new-element steps handed to the general scan: window order %lld, uneven ends %lld, root's children %lld, fresh without a parent lane %lld, fresh under fresh %lld, more than PS parents %lld\n", h[22], h[23], h[24], h[25], h[26], h[27]);
        fprintf(stderr, "   run loop left for the general scan: window order / last step %lld, a frozen parent's older values %lld; general main scans on a table that is not fresh %lld, uneven %lld, beam not full %lld\n", h[28], h[29], h[30], h[32], h[33]);
        fprintf(stderr, "   table build by part: A fields %lld, B marks + expansion + row groups %lld, C children %lld, D continuing slots %lld, E leaving headers %lld, F identity moves %lld, parent slots %lld, G rewind + rest %lld (the rest is in the total above)\n", h[40], h[41], h[42], h[43], h[44], h[45], h[46], h[8]);
        fprintf(stderr, "   row groups handed out: %lld; B by part: marks + arena look-up %lld, expansion %lld, row groups (in B above)\n", h[47], h[49], h[50]);
        fprintf(stderr, "   closed-form chains: %lld passes (a staged parent's children on both reads), %lld of them left to the serial chain\n", h[34], h[35]);
        fprintf(stderr, "   nodes entering the beam: %lld, of them expanded before %lld, with their children's row group still theirs %lld\n", h[36], h[37], h[38]);
        fprintf(stderr, "   table builds that ask the arena for a node's children: %lld; window rescans: %lld lanes in %lld calls; bounds made exact in %lld steps\n", h[20], h[21], h[10], h[31]);
    }
#endif
}
#endif   // PO_REG_TU: the C entry points
