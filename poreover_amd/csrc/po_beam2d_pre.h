// Per-pair pre-pass and diagonal walk of the row_col pair beam search for the register-state kernel (po_beam2d_reg.hip,
// launched through po_beam2d.hip's host code).  Kept in a
// header of their own so that tools/simt_emu can run them on the CPU next to the kernel under development.
#pragma once
#include "po_beam2d_common.h"

// ---- per-pair pre-pass: envelope bounds, transposed envelope, widest window -> R, node budget (BeamSearch.h:270-284)
// ONE wave per pair, no LDS, few registers: in the pipelined job this kernel of wave k + 1 runs next to a device full of
// beam2d_reg_kernel waves of wave k, and a workgroup only starts where those have left room — one wave slot of 128
// registers here and there, never four on one CU (round 5: the 256-thread form with a 35 KB column table waited for the
// pair beam kernel before it to drain, and the whole wave behind it).  The transposed envelope is written straight into its
// place in HBM: on a monotone envelope every column is written by exactly one row per bound.  (The ctc root's blank prefix
// sums were added up here through round 4; beam2d_reg_kernel adds them up as its scans pass the times.)
template <int MODEL>
__global__ __launch_bounds__(64) void beam2d_prepass_kernel(X2Args a) {
    constexpr int K = (MODEL == PO_MODEL_CTC) ? 1 : 3;
    constexpr int nthr = 64;
    const int pi = blockIdx.x, tid = threadIdx.x;
    if (a.use_pre_status && a.status[pi] != PO_OK) {
        if (tid == 0) { a.meta[pi] = make_int2(a.status[pi], -1); a.nmain[pi] = 0; }
        return;
    }
    const int64_t o1 = a.y1_off[pi], o2 = a.y2_off[pi];
    const int64_t b2 = a.y2_off[0];
    const int U = (int)(a.y1_off[pi + 1] - o1), V = (int)(a.y2_off[pi + 1] - o2);
    const int32_t* env = a.env + 2 * o1;
    int* envt = a.envt + 2 * (o2 - b2);
    const int A = a.A, W = a.W;
    int st = PO_OK, R = 32;
    if (U < 1 || V < 1 || U >= (1 << 24) || V >= (1 << 24)) st = PO_E_ARG;
    if (st == PO_OK) {
        // ONE pass over the rows checks the envelope and writes the transposed one as if it were monotone (what build_envelope
        // makes; should it turn out not to be, nothing reads what was written: the pair goes to beam2d_kernel).
        // Transposed envelope (BeamSearch.h:270-284): the first row that covers column x starts its range and every further
        // covering row extends it by one, i.e. [first row, first row + number of rows).  On a monotone envelope the rows covering
        // column x are the contiguous range [a(x), b(x)], a = the first row whose end lies beyond x, b = the last row that
        // starts at or before x: row u is a(x) for the columns between the previous row's end and its own, and b(x) for the
        // columns between its start and the next row's — every column has one writer per bound, no atomics, no initialising
        // pass (the columns nobody writes are known: a(x) for x at or beyond the last row's end, b(x) for x before the first
        // row's start).
        int bad = 0, wmax = 0, nonmono = 0;
        const int2* env2 = (const int2*)env;
        for (int u = tid; u < U; u += nthr) {
            const int2 e = env2[u];
            const int lo = e.x, hi = e.y;
            const int2 ep = (u > 0) ? env2[u - 1] : make_int2(0, 0);
            const int ln = (u + 1 < U) ? env2[u + 1].x : V;
            if (lo < hi && (lo < 0 || hi > V)) bad = 1;
            if (lo < 0 || hi > V || (u > 0 && (lo < ep.x || hi < ep.y))) nonmono = 1;
            wmax = max(wmax, hi - lo);
            const int hp = (u > 0) ? ep.y : 0;
            for (int x = max(hp, 0); x < min(hi, V); ++x) envt[2 * x] = u;
            for (int x = max(lo, 0); x < min(ln, V); ++x) envt[2 * x + 1] = u;
        }
        __threadfence_block();                             // (one wave: the stores above are in memory before the loads below go out)
        if (__syncthreads_or(bad)) st = PO_E_ENVELOPE;
        const bool mono = !__syncthreads_or(nonmono);     // row starts and ends never move backwards
        if (st == PO_OK && mono) {
            const int lo0 = env[0], hil = env[2 * (U - 1) + 1];
            for (int x = tid; x < V; x += nthr) {
                const int a_ = (x < hil) ? envt[2 * x] : 0x7fffffff, b_ = (x >= lo0) ? envt[2 * x + 1] : -1;
                const int c = (a_ != 0x7fffffff && b_ >= a_) ? b_ - a_ + 1 : 0;
                envt[2 * x] = c ? a_ : -1;
                envt[2 * x + 1] = c ? a_ + c : -1;
                wmax = max(wmax, c);
            }
        } else if (st == PO_OK && !a.need_mono) {   // (a caller's own envelope, and a launch that takes it: integer atomics, order-free)
            for (int x = tid; x < V; x += nthr) { envt[2 * x] = 0x7fffffff; envt[2 * x + 1] = 0; }
            __syncthreads();
            for (int u = tid; u < U; u += nthr) {
                const int lo = env[2 * u], hi = env[2 * u + 1];
                for (int x = lo; x < hi; ++x) { atomicMin(&envt[2 * x], u); atomicAdd(&envt[2 * x + 1], 1); }
            }
            __syncthreads();
            for (int x = tid; x < V; x += nthr) {
                const int c = envt[2 * x + 1], f = envt[2 * x];
                envt[2 * x] = c ? f : -1;
                envt[2 * x + 1] = c ? f + c : -1;
                wmax = max(wmax, c);
            }
        }
        if (st == PO_OK) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) wmax = max(wmax, __shfl_xor(wmax, off));
            while (R < wmax + 2) R <<= 1;
            const long long pool_entries = (long long)(a.pool_bytes / (8 * K));   // (beam2d_reg_kernel's entries: the values, no tag)
            const long long ng = pool_entries / ((long long)PO_A * 2 * R);
            // (node ids go into 24 tag bits; the slice's arena is smaller than this worst case and checked as nodes are made)
            const long long need = 1 + A + (long long)A * max(W, A) * ((long long)min(U, V) + 1);
            if (need >= (1 << 24)) st = PO_E_NOMEM;
            // too few row groups for this window width here, or (test hook) odd pairs: beam2d_kernel takes it
            // ... or an envelope whose row starts / ends move backwards, for a kernel that builds on windows that only
            // move forward (what build_envelope makes; anything else is a caller's own array)
            else if (min((long long)a.ngl, ng) < 8 * max(W, PO_A) || ((a.defer_odd & 1) && (pi & 1)) || (a.need_mono && !mono)) R = X2_DEFERRED;
        }
    }
    if (tid == 0) {
        a.meta[pi] = make_int2(st, R);
        a.nmain[pi] = 0;
        if (st == PO_OK && R == X2_DEFERRED) {   // the pass over the deferred pairs has something to do
            a.queue[16] = 1;
            if (a.defer_count) atomicAdd(a.defer_count, 1ull);
        }
    }
}

// ---- the diagonal walk itself (BeamSearch.h:300-341) depends on the envelope only: one wave per pair replays it
// and records the main steps, so the beam kernel neither reads the envelope nor loops over catch-up steps.
// Lane l caches envelope row ubase + l / column vbase + l; the walk reads them with v_readlane and refills a
// cache when it is left (one global round trip per <= 64 steps).  No LDS: many waves per CU hide its latency.
__global__ __launch_bounds__(64) void beam2d_walk_kernel(X2Args a) {
    const int pi = blockIdx.x, lane = threadIdx.x;
    const int2 mt = a.meta[pi];
    if (mt.x != PO_OK || mt.y < 0) return;   // refused, skipped upstream or deferred: no schedule needed
    const int64_t o1 = a.y1_off[pi], o2 = a.y2_off[pi], b2 = a.y2_off[0];
    const int U = (int)(a.y1_off[pi + 1] - o1), V = (int)(a.y2_off[pi + 1] - o2);
    const int2* env2 = (const int2*)(a.env + 2 * o1);
    const int2* envt2 = (const int2*)(a.envt + 2 * (o2 - b2));
    int4* sc = a.sched + (o2 - b2);
    // Main steps come in runs along a diagonal (u + l, v + l), l = 0, 1, ...: the wave tests 64 of them at once
    // (lane l looks at row u + l and column v + l, cached 64 at a time in registers and fetched with a lane
    // permute), records the run up to the first position that is not a main step, and resolves that one position
    // by the reference's rule (catch-up on read 1, catch-up on read 0, or uninitialised bounds).
    int u = 0, v = 0, m = 0, werr = 0, ubase = -1000, vbase = -1000;
    int2 erc = make_int2(0, 0), ecc = make_int2(0, 0);
    while (u <= U - 1 && v <= V - 1) {
        if (u < ubase || u >= ubase + 64) { ubase = u; erc = (ubase + lane < U) ? env2[ubase + lane] : make_int2(0, 0); }
        if (v < vbase || v >= vbase + 64) { vbase = v; ecc = (vbase + lane < V) ? envt2[vbase + lane] : make_int2(0, 0); }
        const int ul = u + lane, vl = v + lane;
        const int iu = ul - ubase, iv = vl - vbase;                    // cache slots of this lane's row / column
        const bool have = iu < 64 && iv < 64 && ul < U && vl < V;       // (lane 0 always has both)
        const int ers = __shfl(erc.x, iu & 63), ere = __shfl(erc.y, iu & 63);
        const int ecs = __shfl(ecc.x, iv & 63), ece = __shfl(ecc.y, iv & 63);
        const bool row_ok = (vl >= ers && vl < ere), col_ok = (ul >= ecs && ul < ece);
        const unsigned long long okb = __ballot(have && row_ok && col_ok);
        const unsigned long long haveb = __ballot(have);
        const int run = (~okb == 0ull) ? 64 : __builtin_ctzll(~okb);   // main steps from (u, v) on
        if (lane < run) sc[m + lane] = make_int4(ul, vl, ece, ere);
        m += run; u += run; v += run;
        if (run < 64 && ((haveb >> run) & 1ull)) {   // the position after the run is in range and is not a main step
            // Catch-ups come in runs too (a base of one read against a stretch of the other: 10 - 30 in a row, as many
            // positions as main steps on the bench's pairs), and a run of them is resolved at once:
            //  * read 1 (:314-322): row u starts beyond v — every (u, v') up to the row's start is the same case;
            //  * read 0 (:328-336): column v starts beyond u — (u', v) is the same case while u' stays before the column's
            //    start and row u' does not start beyond v (that would be a read-1 catch-up, which is tested first).
            const unsigned long long c1 = __ballot(!row_ok && vl < ers), c0 = __ballot(!col_ok && ul < ecs);
            if ((c1 >> run) & 1ull) v = __builtin_amdgcn_readlane(ers, run);
            else if ((c0 >> run) & 1ull) {
                const int cs = __builtin_amdgcn_readlane(ecs, run);
                const int rw = u + 1 + lane, ir = rw - ubase;            // lane l looks at row u + 1 + l
                const int rws = __shfl(erc.x, ir & 63);
                const unsigned long long same = __ballot(ir < 64 && rw < cs && rw < U && !(v < rws));
                u += 1 + ((~same == 0ull) ? 64 : __builtin_ctzll(~same));
            } else { werr = 1; break; }                // uninitialised bounds upstream (:309)
        }
        // (a run that ends where the caches or the reads end is simply continued by the next round)
    }
    if (lane == 0) {
        a.nmain[pi] = m;
        if (werr) a.meta[pi] = make_int2(PO_E_ENVELOPE, mt.y);
    }
}
