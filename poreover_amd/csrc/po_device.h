// Shared device helpers for the gfx950 decoding kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/poreover_hip.h"

#define PO_WAVE 64
#define PO_A 4  // alphabet "ACGT"

#define PO_NEG_INF (-__builtin_inf())

// Log.h:9-15 log_(): -inf for x <= 0 and for NaN
__device__ __forceinline__ double po_log_(double x) { return (x > 0) ? log(x) : PO_NEG_INF; }

// Log.h:17-23 logaddexp(): max + log_(1 + exp(min - max)); the (-inf, -inf) case yields -inf
// through log_(NaN), exactly as the reference does.
__device__ __forceinline__ double po_lae(double x1, double x2) {
    const bool ge = (x1 >= x2);
    const double hi = ge ? x1 : x2;
    const double d = ge ? (x2 - x1) : (x1 - x2);
    return hi + po_log_(1.0 + exp(d));
}

__device__ __forceinline__ int po_lane() { return threadIdx.x & (PO_WAVE - 1); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every
// outstanding global access (s_waitcnt vmcnt(0)); inside a dependent loop that turns each
// fire-and-forget global store into a full round trip.  Use this one when only LDS data is
// exchanged across the barrier and global visibility is established later by __syncthreads().
__device__ __forceinline__ void po_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Candidate ordering used by every prune: higher score first; exact ties by node creation
// order (ascending id).  The reference's tie order is heap-address order (Beam.h:96-107).
__device__ __forceinline__ bool po_better(double sa, int ia, double sb, int ib) {
    return (sa > sb) || (!(sb > sa) && ia < ib);
}

// node arena entry: parent id and last symbol packed as (parent << 3) | last  (last <= 4)
__device__ __forceinline__ int po_pack_node(int parent, int last) { return (parent << 3) | last; }
__device__ __forceinline__ int po_node_parent(int packed) { return packed >> 3; }
__device__ __forceinline__ int po_node_last(int packed) { return packed & 7; }

// ---- tree-model recurrences shared by the 1-D and pair kernels ------------------------------
// root values at time tm1 (tree constructors PrefixTree.h:467-476, :541-546, :641-647)
template <int MODEL>
__device__ __forceinline__ void root_values(int tm1, double blank_cum, double* out) {
    if (MODEL == PO_MODEL_CTC) {
        out[0] = (tm1 < 0) ? 0.0 : blank_cum;
    } else if (MODEL == PO_MODEL_MERGE) {
        out[0] = (tm1 < 0) ? 0.0 : PO_NEG_INF;
        out[1] = (tm1 < 0) ? 0.0 : PO_NEG_INF;
        out[2] = PO_NEG_INF;
    } else {
        const double h = log(0.5);
        out[0] = (tm1 < 0) ? 0.0 : PO_NEG_INF;
        out[1] = (tm1 < 0) ? h : PO_NEG_INF;
        out[2] = (tm1 < 0) ? h : PO_NEG_INF;
    }
}


// update_prob for one node at one time (PrefixTree.h:478-488,518-531 ctc; :649-663,690-704 merge
// repeats; :548-574,600-632 flip-flop).  sp = own values at t-1, pp = parent's values at t-1,
// ya = y[t][last]; yb = y[t][blank] (CTC models) or y[t][last + A] (flip-flop);
// same = (parent->last == last); first = (parent->depth == 0 && t == 0).
template <int MODEL>
__device__ __forceinline__ void po_update(const double* sp, const double* pp, double ya, double yb, bool same,
                                          bool first, double* out) {
    if (MODEL == PO_MODEL_CTC) {
        out[0] = po_lae(pp[0] + ya, sp[0] + yb);
    } else if (MODEL == PO_MODEL_MERGE) {
        const double gap = sp[0] + yb;
        double ng;
        if (first) ng = ya;
        else if (same) ng = po_lae(pp[1] + ya, sp[2] + ya);
        else ng = po_lae(pp[0] + ya, sp[2] + ya);
        out[0] = po_lae(gap, ng);
        out[1] = gap;
        out[2] = ng;
    } else {
        const double stay_flip = sp[1] + ya;
        const double stay_flop = sp[2] + yb;
        double ef, eo;
        if (first) {
            ef = ya;
            eo = yb;
        } else if (same) {
            ef = pp[2] + ya;
            eo = pp[1] + yb;
        } else {
            ef = po_lae(pp[1], pp[2]) + ya;
            eo = PO_NEG_INF;
        }
        const double flip = po_lae(ef, stay_flip);
        const double flop = po_lae(eo, stay_flop);
        out[0] = po_lae(flip, flop);
        out[1] = flip;
        out[2] = flop;
    }
}
