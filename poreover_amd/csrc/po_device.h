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

// Candidate ordering used by every prune: higher score first; exact ties by node creation
// order (ascending id).  The reference's tie order is heap-address order (Beam.h:96-107).
__device__ __forceinline__ bool po_better(double sa, int ia, double sb, int ib) {
    return (sa > sb) || (!(sb > sa) && ia < ib);
}

// node arena entry: parent id and last symbol packed as (parent << 3) | last  (last <= 4)
__device__ __forceinline__ int po_pack_node(int parent, int last) { return (parent << 3) | last; }
__device__ __forceinline__ int po_node_parent(int packed) { return packed >> 3; }
__device__ __forceinline__ int po_node_last(int packed) { return packed & 7; }
