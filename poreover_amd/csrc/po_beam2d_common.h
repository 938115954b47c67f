// Shared by the pair-beam translation units (po_beam2d.hip, po_beam2d_reg.hip): the tagged value store entry, kernel
// argument blocks, element-table field names and the wave-level hand-over helpers.  Everything sits in an anonymous
// namespace: each translation unit gets its own copy.
#pragma once
#include "po_device.h"

namespace {



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
    const int32_t* env;       // NULL: no envelope (method row only)
    int n, A, W, C, method;
    uint32_t alphabet;
    char* seq; const int64_t* seq_off; int32_t* seq_len; int32_t* status;
    int use_pre_status;       // status[] already holds skip / error codes for some pairs: leave those alone
    // workspace (per persistent workgroup unless noted)
    int* queue;               // one counter for the launch
    char* pool; size_t pool_bytes;
    int* arena; long long arena_cap;   // 3 int arrays (packed(parent,last), first_child, row group) + 1 double array
    double* cum; long long tcap;       // 2 arrays of tcap doubles: blank prefix sums of each read
    int* envt; long long vcap;         // 2 * vcap ints: transposed envelope
    long long* dbg;                    // optional phase cycle counters (PO_B2_TIMING builds)
    const int2* only_meta;             // non-NULL: decode only the pairs the two-pairs-per-wave path deferred (meta.y == -2)
    int* cellb;                        // grid method: two rows of per-cell beams per workgroup (2 * vcap * (1 + 6 W) ints)
    int retry_nomem;                   // second pass with a larger store: decode only the pairs the first one gave PO_E_NOMEM
    const int* order;                  // optional: the pair the q-th queue ticket stands for (longest first)
    const int* retry_flag;             // retry pass: the word the first pass sets when it hands a pair on (queue[8]); 0 -> return at once
    unsigned long long* upd_count;     // optional (po_profile_update_counter): update_prob evaluations {of the reference's schedule, executed}
    unsigned long long* wgstate;       // per workgroup {magic, epoch counter}: what its slice of the value store was last tagged with
    unsigned long long magic;          // names this workspace geometry: a slice whose state word differs is cleared before use
};

// F_PSLOT of an element whose parent does not move in the scan: a frozen parent (its values are read from
// its ring row in the store), or the root (closed form / blank prefix sums)
constexpr int PS_FROZEN = -1, PS_ROOT = -2;
// meta.y of a pair the two-pairs-per-wave kernel hands to beam2d_kernel (window too wide for its store
// geometry, or its row-group table ran out)
constexpr int X2_DEFERRED = -2;

// LDS hand-over between iterations.  One wave per workgroup: a wave's LDS operations execute in order,
// only the compiler needs fencing.  More waves: LDS-only barrier (outstanding stores are not waited for).
template <int NTHR>
__device__ __forceinline__ void b2_sync_lds() {
    if constexpr (NTHR == 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        po_lds_barrier();
    }
}

// Workgroup barrier that also makes earlier GLOBAL stores of the workgroup visible to its later loads.  With one
// wave per workgroup nothing has to be waited for: a wave's vector memory operations are performed in execution
// order and its CU's L1 is write-through, so a load issued after a store of the same wave sees it — __syncthreads()
// would drain every outstanding store (s_waitcnt vmcnt(0)) once per step for nothing.
template <int NTHR>
__device__ __forceinline__ void b2_sync_mem() {
    if constexpr (NTHR == 64) b2_sync_lds<64>();
    else __syncthreads();
}
// barrier + "does any thread of the workgroup say yes"
template <int NTHR>
__device__ __forceinline__ bool b2_any(bool p) {
    if constexpr (NTHR == 64) {
        b2_sync_lds<64>();
        return __ballot(p) != 0ull;
    } else {
        return __syncthreads_or(p) != 0;
    }
}

// One (uniform) int2 through the SCALAR data cache.  The walk of row_col reads two envelope entries per round; as vector
// loads they share the wave's vmcnt with its value-store writes, which complete in order — using an entry meant waiting
// for every store issued before the load was (a drain per round).  Scalar loads count on lgkmcnt.  The scalar cache is
// not coherent with vector stores: b2_scalar_cache_inv() before the first read of anything the kernel wrote itself.
__device__ __forceinline__ int2 b2_sload2(const int2* p) {
#ifdef PO_NO_SLOAD   // A/B switch
    return *p;
#else
    // (the "s" constraint does not move a pointer the compiler keeps in vector registers: readfirstlane does)
    const unsigned long long pv = (unsigned long long)p;
    const unsigned long long ps = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(pv >> 32)) << 32) |
                                  (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pv);
    unsigned long long v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ps) : "memory");
    return make_int2((int)(unsigned)v, (int)(unsigned)(v >> 32));
#endif
}
__device__ __forceinline__ void b2_scalar_cache_inv() {
#ifndef PO_NO_SLOAD
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}

// element-table field indices
enum { F_ID, F_ROW, F_PSLOT, F_SYM, F_FC, F_CROW, F_PAR, F_GPAR, F_PROW, F_DEPTH, F_COUNT };
// F_SYM packs: own symbol (bits 0-2) | parent's symbol (bits 4-6) | parent-is-root (bit 9)
__device__ __forceinline__ int sym_pack(int last, int plast, bool rootpar) { return last | (plast << 4) | (rootpar ? 512 : 0); }
__device__ __forceinline__ int sym_last(int s) { return s & 7; }
__device__ __forceinline__ int sym_plast(int s) { return (s >> 4) & 7; }


struct X2Args {
    const double* y1; const int64_t* y1_off;
    const double* y2; const int64_t* y2_off;
    const int32_t* env;
    int n, A, W, C;
    uint32_t alphabet;
    char* seq; const int64_t* seq_off; int32_t* seq_len; int32_t* status;
    int use_pre_status;
    int* queue;
    int2* meta;                    // per pair: {status, R}; R < 0: skipped upstream, leave status alone
    int4* sched;                   // the diagonal walk, one record per MAIN step: {u, v, column-window end, row-window
                                   // end}, at the pair's read-1 row offset (a pair has at most min(U, V) main steps)
    int* nmain;                    // per pair: number of main steps
    int* envt;                     // transposed envelope: 2 ints per read-1 row of the batch
    double* cum1; double* cum2;    // blank prefix sums at the batch row offsets (CTC root)
    char* pool; size_t pool_bytes; // value store per half-wave
    int* arena; long long arena_cap;  // per half-wave: 3 int arrays
    long long* dbg;
    int defer_odd;                    // test hook (PO_X2_DEFER_ODD): hand every odd pair to beam2d_kernel
    int need_mono;                    // the main kernel takes monotone envelopes only (beam2d_reg_kernel): others are deferred
    const int* order;                 // optional: the pair the q-th queue ticket stands for (longest first, pair_order_kernel)
    int pre_vcols;                    // pre-pass: columns its LDS table holds
    int ngl;                          // row groups the main kernel tracks per pair
    unsigned long long* upd_count;    // optional (po_profile_update_counter): update_prob evaluations {of the reference's schedule, executed}
    unsigned long long* wgstate;      // (unused since round 5: beam2d_reg_kernel's store carries no tags; kept for the argument block's layout)
    unsigned long long magic;
    int reg_slots;                    // beam2d_reg_kernel: pair waves of the launch
    // beam2d_reg_kernel: the library's pool of slices {value store of pool_bytes | tree arena: 3 x arena_cap ints}, one per
    // pair wave the device can hold; a wave claims one (slice_claim) when it starts.
    char* slice_chunk[8];             // slice i lives in chunk i >> slice_spc_log2 (chunks of <= 3.5 GB: see reg_pool)
    int slice_spc_log2, nslices;
    size_t slice_bytes;
    int* slice_claim;                 // the ring of free slices (slice number, -1 = empty) ...
    unsigned* slice_tickets;          // ... and its {take, give} tickets
    int persist;                      // 1: a wave takes pairs from the launch's queue until it is empty (as many waves as the device
                                      // holds); 0: one wave per pair, in queue order (the dispatcher interleaves concurrent launches)
    unsigned long long* defer_count;  // pairs handed to beam2d_kernel, counted for the tests (po_debug_deferred_pairs)
    int starve;                       // test hook (po_set_pair_route's defer_odd bits 1, 2): bit 0 = a dozen row groups only,
                                      // bit 1 = a tree arena of a few nodes only — every hand-over reason can be forced
    int no_cum;                       // pre-pass: leave the blank prefix sums out (beam2d_reg_kernel adds the root's alpha up as it goes)
    int chain_scan;                   // beam2d_reg_kernel: PO_CHAIN_CLOSED_FORM (1; 2 = its guard at 3 nats) = a new element's window in closed
                                      // form (one exp, a prefix sum, one log per time: not the reference's rounding, inside its tolerance),
                                      // 0 = the serial chain (po_set_chain_mode)
};

}  // namespace
