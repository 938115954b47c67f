// Pair pipeline between the two 1-D basecalls and the pair beam search, all on the device:
// alignment of the two basecalls, identity / length skips, and the alignment envelope.
//
// Replaces, for pair_decode.pair_decode_helper's default route (pair_decode.py:305-529):
//   align.global_pair_banded  (align/align.pyx:100-178)   band 500, match 2 / mismatch -1 / gap -1
//   align.global_pair         (align/align.pyx:29-98)     --alignment full
//   the identity and length skips (pair_decode.py:372-375, 391-398)
//   envelope.get_alignment_columns + add_block + build_envelope (decoding/envelope.py:5-87)
//   the --diagonal_envelope band (pair_decode.py:497-498)
//
// The banded aligner is NOT a textbook banded NW and is reproduced as written: its boundary
// initialisation is a no-op (the sparse matrix has no rows yet), rows 0..l1-1 are filled over
// [start, end) with `end` itself left at the default 0, reads outside a row's stored range give
// 0, seq[i-1] / seq[j-1] wrap to the last character at index 0, and the trace-back takes EVERY
// neighbour that equals the maximum in turn (no break), emitting up to three columns per pass.
//
// Mapping.  One workgroup (256 threads) per pair, persistent over an atomic queue.  A DP row is
// produced in parallel: c(j) = max(diag + score, up + gap) per cell, then the left-neighbour
// dependency cell(j) = max(c(j), cell(j-1) - 1) is an integer prefix-max of c(k) + k (exact,
// order-independent), done with a block scan.  The trace-back is inherently serial (one lane,
// three independent L2 loads per step).  The envelope is built with integer atomics on the
// output rows; its final fix-up pass carries one scalar (prev_end) and runs from LDS.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>

#define PO_WANT_ZERO_KERNEL 1
#include "po_device.h"
#include "po_host.h"

namespace {

constexpr int NW_MATCH = 2, NW_MISMATCH = -1, NW_GAP = -1, NW_BAND = 500;
// skewed-wavefront DP: cells per lane, columns per block (one wave), column blocks a basecall may span
constexpr int SK_PER = 8, SK_BW = SK_PER * 64, SK_MAXB = 256, SK_SEG = 1024;

struct PPArgs {
    const int64_t* y1_off; const int64_t* y2_off;  // U_i, V_i
    int n;
    const char* seq1d; const int64_t* seq1d_off;   // interleaved: read1 at [2i], read2 at [2i+1]
    const int32_t* len1; const int32_t* len2;
    const int32_t* map1; const int32_t* map2;       // frame of each base, at y*_off[i]
    const int32_t* st1; const int32_t* st2;         // status of the two 1-D decodes
    int padding, full_alignment, diagonal_envelope, diagonal_width;
    int band;                                       // banded alignment half-width (align.pyx:13: 500)
    int match, mismatch, gap;                       // Needleman-Wunsch scores (align.pyx:9-11: 2, -1, -1)
    long long seq_lds_cap;                          // bytes of dynamic LDS per basecall (0: read them from global memory)
    int mode;                                       // 0: align + skips + envelope; 1: align only; 2: envelope from a given alignment
    int retry_cap;                                  // second pass with the big DP slices: only the pairs the first one gave PO_E_CAP
    int maps_increasing;                            // map1 / map2 come from the engine's own Viterbi basecalls: strictly increasing frames
    int* cap_flag;                                  // set by the first pass when a pair gets PO_E_CAP: the second pass returns at once otherwise
    const int32_t* lenU; const int32_t* lenV;       // mode 2: U_i, V_i given explicitly (y*_off unused)
    const int64_t* map1_off; const int64_t* map2_off;  // mode 2: offsets of the frame maps
    char* aln_out1; char* aln_out2; const int64_t* aln_off; int32_t* ncol_out;  // mode 1 out / mode 2 in (forward order)
    const int64_t* env_off;                         // mode 2: row offsets of env
    int32_t* env; double* identity; int32_t* status;
    // workspace
    int* queue;
    int* dp; long long dp_cap;          // per workgroup: DP cells
    int* rowinfo; long long row_cap;    // per workgroup: start[row_cap], end[row_cap], off (as 2 ints) -> 4 ints/row
    char* aln; long long aln_cap;       // per workgroup: 2 * aln_cap chars
};

__device__ __forceinline__ int py_idx(int i, int len) { return i < 0 ? i + len : i; }  // python str index

}  // namespace

// SKEW (one-wave workgroups, banded alignment): the DP as a skewed wavefront — see the block comment at its code.
template <int NT, bool SKEW>
__global__ __launch_bounds__(NT) void pair_prep_kernel(PPArgs a) {
    static_assert(!SKEW || NT == PO_WAVE, "the skewed wavefront is a one-wave schedule");
    constexpr int PERMAX = (NT == 64) ? 16 : 8;  // consecutive DP cells per thread: a banded row (<= 1001 cells) fits one wave
    constexpr int NWAVES = NT / PO_WAVE;
    __shared__ int wsum[NWAVES];
    __shared__ int sh[8];
    __shared__ int lo_s[NT], hi_s[NT], pm[NT];
    // the DP row just filled, for the next row's reads: the fill never reads the table back from HBM (a
    // store -> barrier -> load round trip per row); the table is only written, for the trace-back
    __shared__ int rowbuf[2][SKEW ? 1 : PERMAX * NT + 4];
    // SKEW: per column block of SK_BW columns, the rows it is walked over and where its flag rows start
    __shared__ int blk_lo[SKEW ? SK_MAXB : 1], blk_hi[SKEW ? SK_MAXB : 1], blk_base[SKEW ? SK_MAXB : 1];
    __shared__ unsigned char stepbuf[SKEW ? SK_SEG : 4];   // SKEW: the trace-back's steps (three bits each) before they become columns
    extern __shared__ __attribute__((aligned(16))) char seq_lds[];  // 2 x a.seq_lds_cap bytes: the two basecalls
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* dp = a.dp + (size_t)blockIdx.x * a.dp_cap;
    int* r_start = a.rowinfo + (size_t)blockIdx.x * 4 * a.row_cap;
    int* r_end = r_start + a.row_cap;
    long long* r_off = (long long*)(r_end + a.row_cap);
    char* al1 = a.aln + (size_t)blockIdx.x * 2 * a.aln_cap;
    char* al2 = al1 + a.aln_cap;

    // LDS hand-over in the DP fill.  A one-wave workgroup's LDS and vector-memory operations are performed in order:
    // only the compiler needs a fence — __syncthreads() would also wait for the row's table stores to be acknowledged
    // (s_waitcnt vmcnt(0)), four times per row.
    auto psync = [&]() {
        if constexpr (NT == PO_WAVE) po_wave_sync();
        else __syncthreads();
    };
    // block-wide inclusive prefix max of one int per thread.  Within the wave: data-parallel-primitive moves instead of
    // six ds_bpermute round trips (this scan sits on the per-row dependency chain of the DP fill) — Hillis-Steele inside
    // each row of 16 lanes (row_shr 1, 2, 4, 8), then lane 15 of rows 0 / 2 into rows 1 / 3 (row_bcast:15) and lane 31
    // into rows 2 and 3 (row_bcast:31); a lane without a source keeps the identity.
    auto block_prefix_max = [&](int v) {
#ifdef PO_PP_SHFL_SCAN   // A/B switch: the permute form
#pragma unroll
        for (int o = 1; o < PO_WAVE; o <<= 1) {
            const int t = __shfl_up(v, o);
            if (lane >= o) v = max(v, t);
        }
#else
        v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x111, 0xf, 0xf, false));   // row_shr:1
        v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x112, 0xf, 0xf, false));   // row_shr:2
        v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x114, 0xf, 0xf, false));   // row_shr:4
        v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x118, 0xf, 0xf, false));   // row_shr:8
        v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
        v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
#endif
        if (lane == PO_WAVE - 1) wsum[wave] = v;
        psync();
        int carry = INT_MIN;
        for (int w = 0; w < wave; ++w) carry = max(carry, wsum[w]);
        psync();
        return max(v, carry);
    };
    // block-wide sum / exclusive prefix sum of one int per thread
    auto block_excl_sum = [&](int v, int* total) {
        int inc = v;
#pragma unroll
        for (int o = 1; o < PO_WAVE; o <<= 1) {
            const int t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == PO_WAVE - 1) wsum[wave] = inc;
        __syncthreads();
        int base = 0, tot = 0;
        for (int w = 0; w < NWAVES; ++w) { if (w < wave) base += wsum[w]; tot += wsum[w]; }
        __syncthreads();
        *total = tot;
        return base + inc - v;
    };

    if (a.retry_cap && *a.cap_flag == 0) return;   // nothing was left over (the usual case): not one queue round trip
    for (;;) {
        __syncthreads();
        if (tid == 0) sh[0] = atomicAdd(a.queue, 1);
        __syncthreads();
        const int pi = sh[0];
        if (pi >= a.n) break;
        if (a.retry_cap && a.status[pi] != PO_E_CAP) continue;   // decoded (or refused for good) by the first pass
#ifdef PO_PP_TIMING
        long long tk_[4] = {0, 0, 0, 0}, tl_ = wall_clock64();
#define PPTK(i) do { const long long n_ = wall_clock64(); tk_[i] += n_ - tl_; tl_ = n_; } while (0)
#else
#define PPTK(i) do {} while (0)
#endif
        const int64_t o1 = (a.mode == 2) ? a.map1_off[pi] : (a.mode == 1 ? 0 : a.y1_off[pi]);
        const int64_t o2 = (a.mode == 2) ? a.map2_off[pi] : (a.mode == 1 ? 0 : a.y2_off[pi]);
        const int U = (a.mode == 2) ? a.lenU[pi] : (a.mode == 1 ? 0 : (int)(a.y1_off[pi + 1] - o1));
        const int V = (a.mode == 2) ? a.lenV[pi] : (a.mode == 1 ? 0 : (int)(a.y2_off[pi + 1] - o2));
        int32_t* env = (a.mode == 1) ? nullptr : a.env + 2 * ((a.mode == 2) ? a.env_off[pi] : o1);

        if (a.mode == 0 && a.diagonal_envelope) {  // pair_decode.py:497-498, python float arithmetic u/U*V
            for (int u = tid; u < U; u += NT) {
                const int c = (int)((double)u / (double)U * (double)V);
                env[2 * u] = max(c - a.diagonal_width, 0);
                env[2 * u + 1] = min(c + a.diagonal_width, V);
            }
            if (tid == 0) { a.status[pi] = (U < 1 || V < 1) ? PO_E_ARG : PO_OK; a.identity[pi] = 0.0; }
            continue;
        }

        int l1, l2;
        const char *s1 = nullptr, *s2 = nullptr;
        if (a.mode == 2) {   // lengths = number of mapped bases; the strings themselves are not needed
            l1 = (int)(a.map1_off[pi + 1] - o1); l2 = (int)(a.map2_off[pi + 1] - o2);
        } else if (a.mode == 1) {
            l1 = (int)(a.seq1d_off[2 * pi + 1] - a.seq1d_off[2 * pi]);
            l2 = (int)(a.seq1d_off[2 * pi + 2] - a.seq1d_off[2 * pi + 1]);
        } else { l1 = a.len1[pi]; l2 = a.len2[pi]; }
        if (a.mode != 2) { s1 = a.seq1d + a.seq1d_off[2 * pi]; s2 = a.seq1d + a.seq1d_off[2 * pi + 1]; }
        // both basecalls go to LDS once: the DP reads one character per cell and the trace-back two per step;
        // from global memory each of those is a dependent round trip
        if (a.mode != 2 && a.seq_lds_cap > 0 && l1 >= 0 && l2 >= 0 && l1 <= a.seq_lds_cap && l2 <= a.seq_lds_cap) {
            char* ls1 = seq_lds;
            char* ls2 = seq_lds + a.seq_lds_cap;
            __syncthreads();
            for (int i = tid; i < l1; i += NT) ls1[i] = s1[i];
            for (int i = tid; i < l2; i += NT) ls2[i] = s2[i];
            __syncthreads();
            s1 = ls1; s2 = ls2;
        }
        const int32_t* m1 = (a.mode == 1) ? nullptr : a.map1 + o1;
        const int32_t* m2 = (a.mode == 1) ? nullptr : a.map2 + o2;
        int st = PO_OK;
        if (a.mode == 0 && a.st1[pi] != PO_OK) st = a.st1[pi];
        else if (a.mode == 0 && a.st2[pi] != PO_OK) st = a.st2[pi];
        else if (a.mode == 0 && abs(l1 - l2) > 1000) st = PO_SKIP_LENGTH;   // pair_decode.py:372-375
        // an empty basecall: global_pair_banded aligns it to gaps only, identity 0.0 -> the pair is skipped like any
        // other of low identity (pair_decode.py:395-398); two empty ones divide 0 by 0 upstream
        else if (a.mode == 0 && (l1 < 1) != (l2 < 1)) st = PO_SKIP_IDENTITY;
        else if (a.mode != 1 && (l1 < 1 || l2 < 1)) st = PO_E_ARG;
        const bool full = a.full_alignment != 0;
        const int nrows = full ? l1 + 1 : l1;
        if (st == PO_OK && a.mode != 2 && (nrows > a.row_cap || (long long)l1 + l2 + 8 > a.aln_cap)) st = PO_E_CAP;
        if (st != PO_OK) {
            if (tid == 0) { a.status[pi] = st; if (st == PO_E_CAP) *a.cap_flag = 1; if (a.identity) a.identity[pi] = 0.0; if (a.mode == 1) a.ncol_out[pi] = 0; }
            continue;
        }
        int ncol = 0;
        if (a.mode == 2) {  // the alignment is given (forward order): store it reversed like the trace-back does
            ncol = a.ncol_out[pi];
            if (ncol > a.aln_cap || ncol < 1) {
                if (tid == 0) a.status[pi] = (ncol < 1) ? PO_E_ARG : PO_E_CAP;
                continue;
            }
            const char* g1 = a.aln_out1 + a.aln_off[pi];
            const char* g2 = a.aln_out2 + a.aln_off[pi];
            for (int k = tid; k < ncol; k += NT) { al1[ncol - 1 - k] = g1[k]; al2[ncol - 1 - k] = g2[k]; }
            __syncthreads();
        } else {

        if constexpr (SKEW) {
        // ------------------------------------------------------------------ banded DP as a skewed wavefront
        // Lane L owns the SK_PER columns [cb + 8 L, cb + 8 L + 8) of a block of SK_BW = 512 columns and, at step tau,
        // fills row ilo + tau - L of them: the cell left of its strip was produced by lane L - 1 one step earlier (one
        // wave_shr:1 move), the diagonal one two steps earlier (the value it fetched in the previous step), the row
        // above is its own registers.  No scan, no LDS hand-over, no barrier: a row costs its 8 cells.  Cells outside
        // a row's [start, end) hold 0, which is what SparseMatrix::get answers for them.  A basecall wider than one
        // block is walked block by block (the rows whose band touches the block), the last column of a block handed
        // to the next one through a small array.
        //   The score table itself is never stored.  The trace-back (align.pyx:137-174) only asks, at a position
        // (i, j), which of  cell(i-1,j-1) + score,  cell(i-1,j) + gap,  cell(i,j-1) + gap  equal their maximum — the
        // operands of the fill at that position — so the fill stores those three bits per position (scored with the
        // DEFAULT match / mismatch as the reference's trace-back does), for every position of every (row, block) it
        // visits: one dword per lane and step, written coalesced in step order (slot = row - ilo + lane).  Visited are,
        // per block, the rows i with end(i) >= cb and start(i-1) <= cb + 511, plus the row below the last one: every
        // position with a computed neighbour.  Anywhere else the three neighbours read 0 and the walk decides from the
        // two characters alone.
        int nblocks = 0;
        int* const flags = dp;
        if (l1 >= 1 && l2 >= 1) {
            const double ratio = (double)l2 / (double)l1;
            auto row_s = [&](int i) -> int { const int c = (int)rint(ratio * (double)i); return max(c - a.band, 0); };
            auto row_e = [&](int i) -> int {
                const int c = (int)rint(ratio * (double)i);
                const int s_ = max(c - a.band, 0), e_ = min(c + a.band, l2 - 1);
                return e_ < s_ ? s_ : e_;
            };
            nblocks = row_e(l1 - 1) / SK_BW + 1;
            if (nblocks > SK_MAXB) {
                if (tid == 0) { a.status[pi] = PO_E_UNSUPPORTED; if (a.identity) a.identity[pi] = 0.0; if (a.mode == 1) a.ncol_out[pi] = 0; }
                continue;
            }
            psync();
            for (int b0 = 0; b0 < nblocks; b0 += PO_WAVE) {
                const int b = b0 + lane;
                if (b < nblocks) {
                    const int cb = b * SK_BW;
                    int lo = 0, hi = l1;        // first row in [0, l1] whose end reaches the block (row l1 counts as row l1 - 1)
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (row_e(min(mid, l1 - 1)) >= cb) hi = mid; else lo = mid + 1; }
                    int lo2 = 0, hi2 = l1 + 1;  // first row in [0, l1 + 1) whose predecessor starts beyond the block
                    while (lo2 < hi2) {
                        const int mid = (lo2 + hi2) >> 1;
                        if ((mid == 0 ? 0 : row_s(mid - 1)) > cb + SK_BW - 1) hi2 = mid; else lo2 = mid + 1;
                    }
                    blk_lo[b] = lo; blk_hi[b] = lo2 - 1;
                }
            }
            psync();
            if (lane == 0) {
                long long acc = 0;
                for (int b = 0; b < nblocks; ++b) { blk_base[b] = (int)acc; acc += blk_hi[b] - blk_lo[b] + 1 + (PO_WAVE - 1); }
                sh[1] = (acc * PO_WAVE > a.dp_cap) ? 1 : 0;
            }
            psync();
            if (sh[1]) {
                if (tid == 0) { a.status[pi] = PO_E_CAP; *a.cap_flag = 1; if (a.identity) a.identity[pi] = 0.0; if (a.mode == 1) a.ncol_out[pi] = 0; }
                continue;
            }
            PPTK(0);  // setup + block geometry
            const bool dflt = (a.match == NW_MATCH && a.mismatch == NW_MISMATCH);
            for (int b = 0; b < nblocks; ++b) {
                const int cb = b * SK_BW, ilo = blk_lo[b], ihi = blk_hi[b], nr = ihi - ilo + 1;
                int* const frow = flags + (size_t)blk_base[b] * PO_WAVE + lane;
                int* const bout = r_start + (size_t)(b & 1) * 2 * a.row_cap;          // last column of this block, by row
                const int* const bin = r_start + (size_t)((b & 1) ^ 1) * 2 * a.row_cap;  // ... of the previous block
                const int plo = b > 0 ? blk_lo[b - 1] : 0, phi = b > 0 ? blk_hi[b - 1] : -1;
                auto bin_val = [&](int i) -> int { return (i >= plo && i <= phi) ? bin[i] : 0; };
                const int j0 = cb + lane * SK_PER;
                int c2[SK_PER];
#pragma unroll
                for (int q = 0; q < SK_PER; ++q) c2[q] = s2[py_idx(min(j0 + q, l2) - 1, l2)];
                int prev[SK_PER];
#pragma unroll
                for (int q = 0; q < SK_PER; ++q) prev[q] = 0;
                int last = 0;                                        // this lane's newest right-most cell
                int lfp = (lane == 0 && b > 0) ? bin_val(ilo - 1) : 0;  // what it fetched from its left in the previous step
                int bch = 0, bnx = (b > 0) ? bin_val(ilo + lane) : 0;   // lane 0's left column, 64 rows per register
                const bool wr_b = (b + 1 < nblocks) && lane == PO_WAVE - 1;
                auto step = [&](auto DF, int tau, int i, int dg0, int lf) {
                    constexpr bool DFLT = decltype(DF)::value;
                    const bool real = i < l1;
                    const int c = (int)rint(ratio * (double)min(i, l1 - 1));
                    const int s_ = max(c - a.band, 0);
                    int e_ = min(c + a.band, l2 - 1);
                    if (e_ < s_) e_ = s_;
                    const unsigned w_ = real ? (unsigned)(e_ - s_) : 0u;
                    const int t0 = j0 - s_;
                    const int c1 = s1[py_idx(i - 1, l1)];
                    int diag = dg0, left = lf;
                    unsigned acc = 0;
#pragma unroll
                    for (int q = 0; q < SK_PER; ++q) {
                        const int up = prev[q];
                        const bool eq = (c1 == c2[q]);
                        const int d0 = diag + (eq ? a.match : a.mismatch);
                        const int d1 = up + a.gap, d2 = left + a.gap;
                        const int cm = max(d0, max(d1, d2));
                        int e0 = d0, mx = cm;
                        if constexpr (!DFLT) { e0 = diag + (eq ? NW_MATCH : NW_MISMATCH); mx = max(e0, max(d1, d2)); }
                        // three "is not the maximum" bits per position, first pushed = highest
                        acc = __builtin_amdgcn_alignbit(acc, (unsigned)(e0 - mx), 31);
                        acc = __builtin_amdgcn_alignbit(acc, (unsigned)(d1 - mx), 31);
                        acc = __builtin_amdgcn_alignbit(acc, (unsigned)(d2 - mx), 31);
                        const int cell = ((unsigned)(t0 + q) < w_) ? cm : 0;
                        diag = up; left = cell; prev[q] = cell;
                    }
                    last = left;
                    frow[(size_t)tau * PO_WAVE] = (int)acc;
                    if (wr_b) bout[i] = left;
                };
                for (int tau = 0; tau < nr + PO_WAVE - 1; ++tau) {
                    int bval = 0;
                    if (b > 0) {
                        if ((tau & (PO_WAVE - 1)) == 0) { bch = bnx; bnx = bin_val(ilo + tau + PO_WAVE + lane); }
                        bval = __builtin_amdgcn_readlane(bch, tau & (PO_WAVE - 1));
                    }
                    const int lf = __builtin_amdgcn_update_dpp(bval, last, 0x138, 0xf, 0xf, false);   // wave_shr:1; lane 0 keeps bval
                    const int dg0 = lfp;
                    lfp = lf;
                    const int i = ilo + tau - lane;
                    if (i >= ilo && i <= ihi) {
                        if (dflt) step(std::true_type{}, tau, i, dg0, lf);
                        else step(std::false_type{}, tau, i, dg0, lf);
                    }
                }
            }
        }
        PPTK(1);  // fill
        // ------------------------------------------------------------------ trace-back over the stored bits
        // The walk only follows the bits: a batch is 84 rows x 3 flag words around the diagonal through the current
        // position (four loads in flight, lane 3 k + p + 1 of load m holds row i - 21 m - k, words p = -1..1 around
        // column j - 21 m - k), read back with v_readlane on scalar indices; every step appends its three bits to an
        // LDS buffer and moves (i, j).  The alignment columns are written afterwards by all lanes at once: a step's
        // position and first column are prefix sums over the steps before it.
        {
            int i = __builtin_amdgcn_readfirstlane(l1), j = __builtin_amdgcn_readfirstlane(l2);
            const int cap = (int)a.aln_cap;
            int ci = i, cj = j, cn = 0;     // position / column count at the start of the buffered steps
            int nst = 0;
            auto put = [&](int nn, char x, char y) { if (nn < cap) { al1[nn] = x; al2[nn] = y; } };
            auto flush = [&]() {
                psync();
                for (int c0 = 0; c0 < nst; c0 += PO_WAVE) {
                    const int f = (c0 + lane < nst) ? (int)stepbuf[c0 + lane] : 0;
                    const int f0 = (f >> 2) & 1, f1 = (f >> 1) & 1, f2 = f & 1;
                    const int mine = (f0 + f1) | ((f0 + f2) << 10) | ((f0 + f1 + f2) << 20);
                    int v = mine;   // inclusive wave sum of the packed (rows, columns, alignment columns) moved
                    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
                    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
                    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
                    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
                    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
                    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
                    const int ex = v - mine;
                    int ii = ci - (ex & 1023), jj = cj - ((ex >> 10) & 1023), nn = cn + (ex >> 20);
                    if (f0) { ii--; jj--; put(nn++, s1[py_idx(ii, l1)], s2[py_idx(jj, l2)]); }
                    if (f1) { ii--; put(nn++, s1[py_idx(ii, l1)], '-'); }
                    if (f2) { jj--; put(nn++, '-', s2[py_idx(jj, l2)]); }
                    const int tot = __builtin_amdgcn_readlane(v, PO_WAVE - 1);
                    ci -= tot & 1023; cj -= (tot >> 10) & 1023; cn += tot >> 20;
                }
                nst = 0;
                psync();
            };
            while (i > 0 && j > 0) {
                const int I0 = i, J0 = j;
                int fw[4];
                unsigned long long vm[4];
                {
                    const int k = lane / 3, p = lane - 3 * k - 1;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const int kk = 21 * m + k;
                        const int Ik = I0 - kk, G = (max(J0 - kk, 0) >> 3) + p;
                        bool valid = lane < 63 && Ik >= 0 && G >= 0 && (G >> 6) < nblocks;
                        fw[m] = 0;
                        if (valid) {
                            const int b = G >> 6, lo = blk_lo[b];
                            valid = Ik >= lo && Ik <= blk_hi[b];
                            if (valid) { const int L = G & 63; fw[m] = flags[((size_t)blk_base[b] + (size_t)(Ik - lo + L)) * PO_WAVE + L]; }
                        }
                        vm[m] = __ballot(valid);
                    }
                }
                for (;;) {
                    const int k2 = I0 - i;
                    if (k2 >= 84) break;
                    const int pp = (j >> 3) - (max(J0 - k2, 0) >> 3);
                    if (pp < -1 || pp > 1) break;
                    const int m = k2 / 21, r = 3 * (k2 - 21 * m) + pp + 1;
                    const unsigned long long vmm = m == 0 ? vm[0] : (m == 1 ? vm[1] : (m == 2 ? vm[2] : vm[3]));
                    int t;   // f0 << 2 | f1 << 1 | f2
                    if ((vmm >> r) & 1ull) {
                        const int w = m == 0 ? __builtin_amdgcn_readlane(fw[0], r)
                                             : (m == 1 ? __builtin_amdgcn_readlane(fw[1], r)
                                                       : (m == 2 ? __builtin_amdgcn_readlane(fw[2], r) : __builtin_amdgcn_readlane(fw[3], r)));
                        t = ~((unsigned)w >> (21 - 3 * (j & 7))) & 7;
                    } else {   // no computed neighbour: 0 + score, 0 + gap, 0 + gap
                        const int sc = (s1[py_idx(i - 1, l1)] == s2[py_idx(j - 1, l2)]) ? NW_MATCH : NW_MISMATCH;
                        const int mx = max(sc, a.gap);
                        t = __builtin_amdgcn_readfirstlane((sc == mx ? 4 : 0) | (a.gap == mx ? 3 : 0));
                    }
                    if (lane == 0) stepbuf[nst] = (unsigned char)t;
                    nst++;
                    i -= ((t >> 2) & 1) + ((t >> 1) & 1);
                    j -= ((t >> 2) & 1) + (t & 1);
                    if (nst == SK_SEG) flush();
                    if (!(i > 0 && j > 0)) break;
                }
            }
            flush();
            // what is left of one sequence against gaps (align.pyx:168-174)
            int n = cn;
            if (i > 0) { for (int c = lane; c < i; c += PO_WAVE) put(n + c, s1[i - 1 - c], '-'); n += i; }
            else if (j > 0) { for (int c = lane; c < j; c += PO_WAVE) put(n + c, '-', s2[j - 1 - c]); n += j; }
            if (lane == 0) { sh[2] = n; sh[3] = (n > cap) ? 1 : 0; }
        }
        } else {
        // ------------------------------------------------------------------ DP row geometry
        // banded (align.pyx:119-125): center = int(np.round(l2 / l1 * i)); computed cells [start, end)
        // with end = min(center + band, l2 - 1); full: columns [0, l2] all computed
        for (int i = tid; i < nrows; i += NT) {
            int st_, en_;
            if (full) { st_ = 0; en_ = l2 + 1; }
            else {
                const int center = (int)rint((double)l2 / (double)l1 * (double)i);
                st_ = max(center - a.band, 0);
                en_ = min(center + a.band, l2 - 1);
                if (en_ < st_) en_ = st_;
            }
            r_start[i] = st_; r_end[i] = en_;
        }
        __syncthreads();
        if (tid == 0) {  // row offsets (serial prefix sum over <= a few thousand rows)
            long long acc = 0;
            for (int i = 0; i < nrows; ++i) { r_off[i] = acc; acc += r_end[i] - r_start[i]; }
            sh[1] = (acc > a.dp_cap) ? 1 : 0;
        }
        __syncthreads();
        if (sh[1]) {
            if (tid == 0) { a.status[pi] = PO_E_CAP; *a.cap_flag = 1; if (a.identity) a.identity[pi] = 0.0; if (a.mode == 1) a.ncol_out[pi] = 0; }
            continue;
        }
        // SparseMatrix<int>::get (SparseMatrix.h:51-57,108-115): default 0 outside the computed cells
        auto get = [&](int i, int j) -> int {
            if (i < 0 || i >= nrows) return 0;
            if (j < r_start[i] || j >= r_end[i]) return 0;
            return dp[r_off[i] + (j - r_start[i])];
        };

        PPTK(0);  // setup + row geometry
        // ------------------------------------------------------------------ DP fill, row by row
        int ps = 0, pe = 0;  // computed cells [ps, pe) of the previous row (none before row 0)
        for (int i = 0; i < nrows; ++i) {
            const int js = r_start[i], je = r_end[i], w = je - js;
            int* row = dp + r_off[i];
            int* cur = rowbuf[i & 1];
            const int* prv = rowbuf[(i & 1) ^ 1];
            if (full && i == 0) {  // dpMatrix[0, j] = gap * j (align.pyx:44-45)
                if (w > PERMAX * NT + 1) {
                    if (tid == 0) sh[1] = 1;
                    __syncthreads();
                    break;
                }
                for (int j = tid; j < w; j += NT) { row[j] = a.gap * j; cur[j] = a.gap * j; }
                __syncthreads();
                ps = js; pe = je;
                continue;
            }
            // cell (i-1, j) with SparseMatrix's default 0 outside the computed cells
            auto getp = [&](int j) -> int { return (j >= ps && j < pe) ? prv[j - ps] : 0; };
            // left boundary value and the character of seq1 this row scores against
            //   banded: cell(i, js-1) reads 0 (out of range); seq1[i-1] wraps at i == 0
            //   full:   cell(i, 0) = gap * i, filled cells start at j = 1; seq1[i-1]
            const int jfirst = full ? 1 : js;
            const int left0 = full ? a.gap * i : 0;
            const char c1 = s1[py_idx(i - 1, l1)];
            if (full && tid == 0) { row[0] = left0; cur[0] = left0; }
            const int cnt = je - jfirst;
            const int per = (cnt + NT - 1) / NT;  // consecutive columns per thread
            if (per > PERMAX) {  // a row wider than PERMAX * NT cells (only --alignment full on very long reads)
                if (tid == 0) sh[1] = 1;
                __syncthreads();
                break;
            }
            // local running max of c(k) + G k over this thread's columns (G = -gap: cell(j) = max(c(j), cell(j-1) + gap)
            // <=> cell(j) + G j = prefix max of c(k) + G k) ...
            const int j0 = jfirst + tid * per;
            int loc[PERMAX];
            int m = INT_MIN;
#pragma unroll
            for (int q = 0; q < PERMAX; ++q) {
                const int j = j0 + q;
                int val = INT_MIN;
                if (q < per && j < je) {
                    const char c2 = s2[py_idx(j - 1, l2)];
                    const int diag = getp(j - 1) + (c1 == c2 ? a.match : a.mismatch);
                    const int up = getp(j) + a.gap;
                    val = max(diag, up) - a.gap * j;
                }
                m = max(m, val);
                loc[q] = m;
            }
            // ... combined across threads (exclusive = inclusive of the previous thread), seeded with
            // the value left of the first computed cell
            const int incl = block_prefix_max(m);
            pm[tid] = incl;
            psync();
            const int excl = max(left0 - a.gap * (jfirst - 1), tid > 0 ? pm[tid - 1] : INT_MIN);
#pragma unroll
            for (int q = 0; q < PERMAX; ++q) {
                const int j = j0 + q;
                if (q < per && j < je) { const int cell = max(loc[q], excl) + a.gap * j; row[j - js] = cell; cur[j - js] = cell; }
            }
            psync();
            ps = js; pe = je;
        }
        if (sh[1]) {
            if (tid == 0) { a.status[pi] = PO_E_UNSUPPORTED; if (a.identity) a.identity[pi] = 0.0; if (a.mode == 1) a.ncol_out[pi] = 0; }
            continue;
        }

        PPTK(1);  // fill
        // ------------------------------------------------------------------ trace-back (align.pyx:56-95 == :137-174)
        // One wave walks the path.  Each step needs three table cells (a dependent global round trip), so lane q
        // loads the cells of position (i - q, j - q) speculatively: while the path moves diagonally (most steps
        // of two reads of the same molecule) the next step's cells are already in lane q + 1; any other move
        // ends the batch.  The decisions are taken in the reference's order, from the same values.
        if (tid < PO_WAVE) {
            int i = l1, j = l2, n = 0;
            const int cap = (int)a.aln_cap;
            bool ovf = false;
#define PP_EMIT(c1_, c2_) do { if (lane == 0) { if (n < cap) { al1[n] = (c1_); al2[n] = (c2_); } } if (n >= cap) ovf = true; n++; } while (0)
            while (i > 0 && j > 0) {
                const int ii = i - lane, jj = j - lane;
                int c0 = 0, c1v = 0, c2v = 0;
                if (ii > 0 && jj > 0) {
                    // (the reference's trace-back calls scoring_function without its match / mismatch arguments,
                    //  align.pyx:66,143: the default 2 / -1 whatever the fill used; gap_cost is the caller's)
                    const int sc = (s1[py_idx(ii - 1, l1)] == s2[py_idx(jj - 1, l2)]) ? NW_MATCH : NW_MISMATCH;
                    c0 = get(ii - 1, jj - 1) + sc; c1v = get(ii - 1, jj) + a.gap; c2v = get(ii, jj - 1) + a.gap;
                }
                for (int qv = 0; qv < PO_WAVE; ++qv) {
                    const int q = __builtin_amdgcn_readfirstlane(qv);
                    const int d0 = __builtin_amdgcn_readlane(c0, q), d1 = __builtin_amdgcn_readlane(c1v, q);
                    const int d2 = __builtin_amdgcn_readlane(c2v, q);
                    const int mx = max(d0, max(d1, d2));
                    const bool diag_only = (d0 == mx) && (d1 != mx) && (d2 != mx);
                    if (d0 == mx) { i--; j--; PP_EMIT(s1[py_idx(i, l1)], s2[py_idx(j, l2)]); }
                    if (d1 == mx) { i--; PP_EMIT(s1[py_idx(i, l1)], '-'); }
                    if (d2 == mx) { j--; PP_EMIT('-', s2[py_idx(j, l2)]); }
                    if (!diag_only || !(i > 0 && j > 0)) break;   // the speculation holds only along the diagonal
                }
            }
            while (i > 0 || j > 0) {
                if (i > 0) { i--; PP_EMIT(s1[py_idx(i, l1)], '-'); }
                else if (j > 0) { j--; PP_EMIT('-', s2[py_idx(j, l2)]); }
            }
#undef PP_EMIT
            if (lane == 0) { sh[2] = n; sh[3] = ovf ? 1 : 0; }
        }
        }  // !SKEW
        __syncthreads();
        ncol = sh[2];
        if (sh[3]) {
            if (tid == 0) { a.status[pi] = PO_E_CAP; *a.cap_flag = 1; if (a.identity) a.identity[pi] = 0.0; if (a.mode == 1) a.ncol_out[pi] = 0; }
            continue;
        }
        }  // mode != 2
        if (a.mode == 1) {  // alignment only: forward order out
            const int64_t ao = a.aln_off[pi];
            const int capo = (int)(a.aln_off[pi + 1] - ao);
            if (ncol > capo) { if (tid == 0) { a.status[pi] = PO_E_CAP; *a.cap_flag = 1; a.ncol_out[pi] = 0; } continue; }
            for (int k = tid; k < ncol; k += NT) { a.aln_out1[ao + k] = al1[ncol - 1 - k]; a.aln_out2[ao + k] = al2[ncol - 1 - k]; }
            if (tid == 0) { a.ncol_out[pi] = ncol; a.status[pi] = PO_OK; }
            continue;
        }
        // alignment is stored reversed: column k of the forward alignment is index ncol-1-k
        // identity = matches / columns (pair_decode.py:391-393)
        int matches = 0;
        for (int k = tid; k < ncol; k += NT) matches += (al1[k] == al2[k]);
        int tot_m;
        (void)block_excl_sum(matches, &tot_m);
        const double identity = (double)tot_m / (double)ncol;
        if (a.mode == 0 && identity < 0.5) {  // pair_decode.py:395-398
            if (tid == 0) { a.status[pi] = PO_SKIP_IDENTITY; a.identity[pi] = identity; }
            continue;
        }

        PPTK(2);  // trace-back + identity
        if constexpr (SKEW) if (a.maps_increasing) {
            // ------------------------------------------------------------------ envelope by base (envelope.py:46-87)
            // add_block paints the frames of ONE base of read 1 — [map1[b], map1[b+1]) — per alignment column, with the
            // frames of the column's base of read 2.  Along the alignment both base indices only grow and the frame maps
            // of the Viterbi basecalls grow strictly, so the minimum start over a base's columns is its FIRST column's
            // and the maximum end its LAST column's, and every frame of a base ends up with the same pair: two stores per
            // base instead of two atomics per (column, frame).  The fix-up's state machine (prev_end moves only where a
            // row starts beyond it) then needs one test per base — only a base's first frame can start beyond prev_end,
            // the others start at or before its own end — and the rows are written once, final.
            int* const blo = r_start;
            int* const bhi = r_start + a.row_cap + 8;
            int xbase = -1, ybase = -1;
            for (int k0 = 0; k0 < ncol; k0 += PO_WAVE) {
                const int k = k0 + lane;  // forward column index
                const bool in = k < ncol;
                const char ca = in ? al1[ncol - 1 - k] : '-', cb = in ? al2[ncol - 1 - k] : '-';
                const char cn = (k + 1 < ncol) ? al1[ncol - 2 - k] : '-';
                int tx, ty;
                const int ex = block_excl_sum(ca != '-' ? 1 : 0, &tx);
                const int ey = block_excl_sum(cb != '-' ? 1 : 0, &ty);
                if (in) {
                    const int xi = xbase + ex + (ca != '-' ? 1 : 0);
                    const int yi = ybase + ey + (cb != '-' ? 1 : 0);
                    const int i1 = min(max(xi, 0), l1 - 1), i2 = min(max(yi, 0), l2 - 1);
                    const int i1p = min(max(xi - (ca != '-' ? 1 : 0), 0), l1 - 1);
                    const int i1n = min(max(xi + (cn != '-' ? 1 : 0), 0), l1 - 1);
                    if (k == 0 || i1p != i1) blo[i1] = m2[i2];
                    if (k == ncol - 1 || i1n != i1) bhi[i1] = (i2 + 1 < l2) ? m2[i2 + 1] : V;
                }
                xbase += tx;
                ybase += ty;
            }
            psync();
            {   // frames before the first base: untouched rows (envelope.py:73-75 on the initial -1 / -1)
                const int f0 = min(m1[0], U);
                int lo = 0, hi = min(V, -1 + a.padding);
                if (lo > hi) lo = 0;
                for (int u = lane; u < f0; u += PO_WAVE) { env[2 * u] = lo; env[2 * u + 1] = hi; }
            }
            int pe_ = 0;
            for (int b0 = 0; b0 < l1; b0 += PO_WAVE) {
                const int b = b0 + lane;
                const bool in = b < l1;
                int lo = 0, hi = 0, r0 = 0, r1 = 0;
                if (in) {
                    lo = max(0, blo[b] - a.padding);
                    hi = min(V, bhi[b] + a.padding);
                    if (lo > hi) lo = 0;
                    r0 = m1[b];
                    r1 = min((b + 1 < l1) ? m1[b + 1] : U, U);
                }
                int lo_first = lo, pos = 0;
                for (;;) {
                    const unsigned long long mk = __ballot(in && r0 < r1 && lane >= pos && lo_first > pe_);
                    if (mk == 0ull) break;
                    const int qs = __builtin_ctzll(mk);
                    if (lane == qs) lo_first = pe_;
                    pe_ = __builtin_amdgcn_readlane(hi, qs);
                    pos = qs + 1;
                }
                for (int u = r0; u < r1; ++u) { env[2 * u] = (u == r0) ? lo_first : lo; env[2 * u + 1] = hi; }
            }
            if (tid == 0) { a.status[pi] = PO_OK; if (a.identity) a.identity[pi] = identity; }
            PPTK(3);  // envelope
#ifdef PO_PP_TIMING
            if (tid == 0 && pi == 0) printf("[pp timing] pair 0: setup %lld fill %lld traceback %lld envelope %lld ticks (10 ns)\n", tk_[0], tk_[1], tk_[2], tk_[3]);
#endif
            continue;
        }
        // ------------------------------------------------------------------ envelope (envelope.py:46-87)
        for (int u = tid; u < U; u += NT) { env[2 * u] = INT_MAX; env[2 * u + 1] = -1; }
        __syncthreads();
        // get_alignment_columns (:26-44): x_index / y_index = running count of non-gap characters - 1
        int xbase = -1, ybase = -1;
        for (int k0 = 0; k0 < ncol; k0 += NT) {
            const int k = k0 + tid;  // forward column index
            const bool in = k < ncol;
            const char ca = in ? al1[ncol - 1 - k] : '-', cb = in ? al2[ncol - 1 - k] : '-';
            int tx, ty;
            const int ex = block_excl_sum(ca != '-' ? 1 : 0, &tx);
            const int ey = block_excl_sum(cb != '-' ? 1 : 0, &ty);
            if (in) {
                const int xi = xbase + ex + (ca != '-' ? 1 : 0);
                const int yi = ybase + ey + (cb != '-' ? 1 : 0);
                const int i1 = min(max(xi, 0), l1 - 1), i2 = min(max(yi, 0), l2 - 1);
                const int sx = m1[i1], exr = (i1 + 1 < l1) ? m1[i1 + 1] : U;
                const int sy = m2[i2], eyr = (i2 + 1 < l2) ? m2[i2 + 1] : V;
                for (int i = sx; i < exr; ++i)  // add_block (:5-17)
                    if (i < U) { atomicMin(&env[2 * i], sy); atomicMax(&env[2 * i + 1], eyr); }
            }
            xbase += tx;
            ybase += ty;
        }
        __threadfence_block();
        __syncthreads();
        // padding (:73-75), then the sequential fix-ups (:78-85): prev_end only moves inside the 2nd if
        int prev_end = 0;
        for (int u0 = 0; u0 < U; u0 += NT) {
            const int u = u0 + tid;
            if (u < U) {
                int lo = env[2 * u], hi = env[2 * u + 1];
                if (lo == INT_MAX) lo = -1;
                lo = max(0, lo - a.padding);
                hi = min(V, hi + a.padding);
                lo_s[tid] = lo; hi_s[tid] = hi;
            }
            __syncthreads();
            // The fix-up is a state machine over the rows — prev_end moves only where a row starts beyond it — so the
            // first wave walks it from trigger to trigger: all 64 rows of a sub-chunk test `lo > prev_end` at once
            // (ballot), the first one that does takes prev_end as its start and hands its end on (readlane); a chunk
            // without triggers (the usual case) costs one ballot instead of 64 serial LDS round trips.
            if (tid < PO_WAVE) {
                const int cnt = min(NT, U - u0);
                int pe_ = prev_end;
                for (int b0 = 0; b0 < cnt; b0 += PO_WAVE) {
                    const int q = b0 + lane;
                    const bool in = q < cnt;
                    int lo = in ? lo_s[q] : 0;
                    const int hi = in ? hi_s[q] : 0;
                    if (lo > hi) lo = 0;
                    int pos = 0;   // rows below `pos` of this sub-chunk are settled
                    for (;;) {
                        const unsigned long long mk = __ballot(in && lane >= pos && lo > pe_);
                        if (mk == 0ull) break;
                        const int qs = __builtin_ctzll(mk);
                        if (lane == qs) lo = pe_;
                        pe_ = __builtin_amdgcn_readlane(hi, qs);
                        pos = qs + 1;
                    }
                    if (in) lo_s[q] = lo;
                }
                if (lane == 0) sh[4] = pe_;
            }
            __syncthreads();
            prev_end = sh[4];
            if (u < U) { env[2 * u] = lo_s[tid]; env[2 * u + 1] = hi_s[tid]; }
            __syncthreads();
        }
        if (tid == 0) { a.status[pi] = PO_OK; if (a.identity) a.identity[pi] = identity; }
        PPTK(3);  // envelope
#ifdef PO_PP_TIMING
        if (tid == 0 && pi == 0) printf("[pp timing] pair 0: setup %lld fill %lld traceback %lld envelope %lld ticks (10 ns)\n", tk_[0], tk_[1], tk_[2], tk_[3]);
#endif
    }
}

// ------------------------------------------------------------------------------------------------
extern "C" {
int po_launch_viterbi_strided(const double*, const int64_t*, int, int, int, uint32_t, int, int8_t*, char*,
                              const int64_t*, int, int, int32_t*, int32_t*, int32_t*, int8_t*, int8_t*, hipStream_t);
size_t po_beam2d_ws_bytes_impl(int, int64_t, int64_t, int64_t, int64_t, int, int, int, int);
int po_launch_beam2d_geom(const double*, const int64_t*, const double*, const int64_t*, const int32_t*, int, int, int,
                          uint32_t, int, int, int, int64_t, int64_t, int64_t, int64_t, char*, const int64_t*, int32_t*, int32_t*,
                          int, void*, size_t, hipStream_t);
void po_prof_stage(int kernel, hipStream_t s, int begin, void** tok);
}

// which kernel aligns with a band: process-wide like po_set_pair_route, initial value from PO_PP_LEGACY (read once)
static std::atomic<int> g_pp_legacy{-1};
extern "C" int po_set_align_route(int legacy) {
    g_pp_legacy.store(legacy ? 1 : 0);
    return PO_OK;
}
static bool pp_legacy() {
    int v = g_pp_legacy.load();
    if (v < 0) {
        const char* e = getenv("PO_PP_LEGACY");
        v = (e != nullptr && atoi(e) != 0) ? 1 : 0;
        g_pp_legacy.store(v);
    }
    return v != 0;
}

namespace {
void pp_launch(PPArgs a, int blocks, int one_wave, hipStream_t stream, long long max_len2 = 0) {
    // dynamic LDS for the two basecalls, when they fit (row_cap bounds their length)
    long long cap = (a.mode == 2) ? 0 : ((a.row_cap + 15) & ~15LL);
    if (cap > 24 * 1024) cap = 0;
    a.seq_lds_cap = cap;
    const size_t lds = (size_t)(2 * cap);
    // banded alignment on one-wave workgroups: the skewed wavefront (po_set_align_route(1) / PO_PP_LEGACY=1: the
    // row-at-a-time fill with the stored score table, kept for A/B runs and for basecalls beyond SK_MAXB column blocks)
    const bool legacy = pp_legacy();
    // (a basecall of read 2 beyond SK_MAXB column blocks — 131 072 bases — keeps the row-at-a-time kernel)
    const bool skew = one_wave && !a.full_alignment && a.mode != 2 && !legacy && max_len2 <= (long long)SK_MAXB * SK_BW;
    if (skew) hipLaunchKernelGGL((pair_prep_kernel<64, true>), dim3(blocks), dim3(64), lds, stream, a);
    else if (one_wave) hipLaunchKernelGGL((pair_prep_kernel<64, false>), dim3(blocks), dim3(64), lds, stream, a);
    else hipLaunchKernelGGL((pair_prep_kernel<256, false>), dim3(blocks), dim3(256), lds, stream, a);
}
}  // namespace

namespace {
inline size_t al256(size_t b) { return (b + 255) & ~size_t(255); }

// ints of DP slice the skewed-wavefront kernel needs for basecalls of at most l1 x l2 bases: one flag row of 64 dwords
// per (row, column block) pair it visits plus the 63 drain steps of every block.  A row's visited columns
// [start(i-1), end(i)] span at most 2 band + 1 cells plus the band's move from one row to the next (l2 in all).
inline size_t pp_skew_cells(int64_t l1, int64_t l2, int64_t band) {
    const int64_t nb = l2 / SK_BW + 1;
    const int64_t rows = 2 * (l1 + 1) + ((l1 + 1) * (2 * band + 1) + l2) / SK_BW + 1 + nb;
    return (size_t)(64 * (rows + 63 * nb));
}

struct PPGeom {
    int blocks, one_wave;
    size_t dp_cap, row_cap, aln_cap;
    // second pass for dense basecalls (more than rows / 4 bases: Bonito's stride, fast flip-flop models): a few
    // workgroups whose slices hold ANY basecall of these reads (a basecall has at most one base per frame)
    int big_blocks;
    size_t big_dp_cap, big_row_cap, big_aln_cap, off_big_dp, off_big_rows, off_big_aln;
    size_t off_queue, off_map1, off_map2, off_st1, off_st2, off_dp, off_rows, off_aln, off_ff, off_env, off_b2, total;
    size_t ff_bytes, b2_bytes;
};

// device memory the workspaces may plan with: a fixed share of the board's memory, so that the size a caller
// is told (po_*_workspace_bytes) and the size the launch expects agree whatever else is allocated
size_t pp_total_mem() { return po_dev_info().mem; }

int pp_num_cus() { return po_dev_info().cus; }

PPGeom pp_geometry(int n, int64_t tr1, int64_t tr2, int64_t mr1, int64_t mr2, int C, const po_pair_options* opt) {
    PPGeom g;
    // a banded DP row (<= 2 * 500 + 1 cells) fits one wave at 16 cells per lane: one-wave workgroups need no
    // block barriers and three times as many pairs are in flight; full alignment keeps 256 threads
    g.one_wave = opt->full_alignment ? 0 : 1;
    g.blocks = std::min(n > 0 ? n : 1, pp_num_cus() * (g.one_wave ? 16 : 4));
    // a basecall has at most one base per frame, so lengths are bounded by the row counts; the first pass
    // budgets for basecalls of at most max_rows / 4 bases (nanopore basecallers emit roughly one base per 8-10
    // frames); a pair with a longer one gets PO_E_CAP there and is aligned by the second pass, whose slices hold
    // a base per frame
    const int64_t lmax1 = std::max<int64_t>(64, mr1 / 4 + 8), lmax2 = std::max<int64_t>(64, mr2 / 4 + 8);
    g.row_cap = (size_t)(lmax1 + 1);
    const int64_t width = opt->full_alignment ? (lmax2 + 1) : std::min<int64_t>(lmax2 + 1, 2 * NW_BAND + 1);
    // (the skewed-wavefront kernel keeps trace-back flags, not the score table: 1.2 MB per workgroup at T = 4000 instead of the
    //  4.4 MB the row-at-a-time kernel needs — 13 GB less per 4 096 resident workgroups.  The table's size is reserved only
    //  when that kernel can run: po_set_align_route(1) / PO_PP_LEGACY, or a read 2 beyond SK_MAXB column blocks; a launch
    //  re-derives the geometry, so a route switched after the size query fails with PO_E_CAP instead of overrunning.)
    const bool skew_only = g.one_wave && !pp_legacy() && lmax2 <= (int64_t)SK_MAXB * SK_BW && mr2 + 8 <= (int64_t)SK_MAXB * SK_BW;
    g.dp_cap = skew_only ? pp_skew_cells(lmax1, lmax2, NW_BAND) : (size_t)((lmax1 + 1) * width);
    if (g.one_wave && !skew_only) g.dp_cap = std::max(g.dp_cap, pp_skew_cells(lmax1, lmax2, NW_BAND));
    g.aln_cap = (size_t)(lmax1 + lmax2 + 16);
    {   // the first pass's slices within 5 GB (T = 4000: workgroups of 1.2 MB — all sixteen per CU of a 10 000-pair call; a pipeline
        // slot's wave of 3 334 pairs: 4 GB — so that a wave's workspace stays a few GB: large allocations are what a process's first
        // call waits for) and within 1/16 of the board's memory (very long reads: fewer workgroups).  (2.5 GB — eight workgroups
        // per CU — until late in round 6: the kernel 2.38 instead of 1.96 ms per 10 000 pairs, 1 % of the step;
        // profiles/r06_ab_align_budget.txt.  The end-to-end figures and the first call of a process do not move.)
        static const size_t budget = [] { const char* e = getenv("PO_PP_BUDGET_MB"); return (size_t)(e ? atoi(e) : 5120) << 20; }();
        const size_t per_block = sizeof(int) * g.dp_cap + sizeof(int) * 4 * g.row_cap + 2 * g.aln_cap;
        // (the one-workgroup-per-CU floor applies against that budget only — a comfort figure; the board's 1/16 is a HARD
        //  cap: reads of 4e5 frames make a slice 105 MB, and a floor of 256 of them is 27 GB of workspace: ADVICE round 5)
        const size_t pb = std::max<size_t>(per_block, 1);
        const size_t fit_budget = std::max<size_t>(budget / pb, (size_t)pp_num_cus());
        const size_t fit_mem = std::max<size_t>(1, (pp_total_mem() / 16) / pb);
        g.blocks = (int)std::min<size_t>((size_t)g.blocks, std::max<size_t>(1, std::min(fit_budget, fit_mem)));
    }
    {
        const int64_t b1 = mr1 + 8, b2 = mr2 + 8;
        g.big_row_cap = (size_t)(b1 + 1);
        const int64_t bw = opt->full_alignment ? (b2 + 1) : std::min<int64_t>(b2 + 1, 2 * NW_BAND + 1);
        g.big_dp_cap = skew_only ? pp_skew_cells(b1, b2, NW_BAND) : (size_t)((b1 + 1) * bw);
        if (g.one_wave && !skew_only) g.big_dp_cap = std::max(g.big_dp_cap, pp_skew_cells(b1, b2, NW_BAND));
        g.big_aln_cap = (size_t)(b1 + b2 + 16);
        const size_t per_block = sizeof(int) * g.big_dp_cap + sizeof(int) * 4 * g.big_row_cap + 2 * g.big_aln_cap;
        const size_t fit = ((size_t)2 << 30) / std::max<size_t>(per_block, 1);     // 2 GB for the pass; at least one slice
        g.big_blocks = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(32, fit), (size_t)(n > 0 ? n : 1)));
        if (per_block > pp_total_mem() / 8) g.big_blocks = 0;   // (--alignment full on reads of 10^5 frames: no second pass)
    }
    const int kind = opt->model == PO_MODEL_FLIPFLOP ? PO_KIND_FLIPFLOP : PO_KIND_POREOVER;
    g.ff_bytes = (kind == PO_KIND_FLIPFLOP) ? al256((size_t)std::max(tr1, tr2) * 8) + al256((size_t)std::max(tr1, tr2)) : 0;
    g.b2_bytes = po_beam2d_ws_bytes_impl(n, tr1, tr2, mr1, mr2, C, opt->beam_width, opt->model, opt->method);
    size_t o = 0;
    g.off_queue = o; o += 256;
    g.off_map1 = o; o += al256(sizeof(int32_t) * (size_t)tr1);
    g.off_map2 = o; o += al256(sizeof(int32_t) * (size_t)tr2);
    g.off_st1 = o; o += al256(sizeof(int32_t) * (size_t)n);
    g.off_st2 = o; o += al256(sizeof(int32_t) * (size_t)n);
    g.off_dp = o; o += al256(sizeof(int) * g.dp_cap * g.blocks);
    g.off_rows = o; o += al256(sizeof(int) * 4 * g.row_cap * g.blocks);
    g.off_aln = o; o += al256(2 * g.aln_cap * g.blocks);
    g.off_big_dp = o; o += al256(sizeof(int) * g.big_dp_cap * g.big_blocks);
    g.off_big_rows = o; o += al256(sizeof(int) * 4 * g.big_row_cap * g.big_blocks);
    g.off_big_aln = o; o += al256(2 * g.big_aln_cap * g.big_blocks);
    g.off_ff = o; o += g.ff_bytes;
    g.off_env = o; o += al256(sizeof(int32_t) * 2 * (size_t)tr1);
    g.off_b2 = o; o += g.b2_bytes;
    g.total = o + 256;
    return g;
}
}  // namespace

extern "C" size_t po_pair_ws_bytes_impl(int n, int64_t tr1, int64_t tr2, int64_t mr1, int64_t mr2, int C,
                                        const po_pair_options* opt) {
    return pp_geometry(n, tr1, tr2, mr1, mr2, C, opt).total;
}

extern "C" int po_launch_pair_decode_geom(const double* y1, const int64_t* y1_off, const double* y2,
                                          const int64_t* y2_off, int n, int C, const po_pair_options* opt,
                                          int64_t tr1, int64_t tr2, int64_t mr1, int64_t mr2,
                                          const int32_t* ext_map1, const int32_t* ext_map2, char* seq1d,
                                          const int64_t* seq1d_off, int32_t* len1, int32_t* len2, double* identity,
                                          int32_t* env_out, char* seq, const int64_t* seq_off, int32_t* seq_len,
                                          int32_t* status, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (n <= 0) return PO_OK;
    const int A = PO_A;
    const uint32_t alphabet = 'A' | ('C' << 8) | ('G' << 16) | ((uint32_t)'T' << 24);
    if (opt->model < 0 || opt->model > 2) return PO_E_ARG;
    if ((opt->model == PO_MODEL_FLIPFLOP) ? (C != 2 * A) : (C != A + 1)) return PO_E_ARG;
    const int kind = opt->model == PO_MODEL_CTC ? PO_KIND_POREOVER
                                                 : (opt->model == PO_MODEL_MERGE ? PO_KIND_BONITO : PO_KIND_FLIPFLOP);
    const PPGeom g = pp_geometry(n, tr1, tr2, mr1, mr2, C, opt);
    if (ws_bytes < g.total) return PO_E_CAP;
    char* w = (char*)ws;
    int32_t* map1 = (int32_t*)(w + g.off_map1);
    int32_t* map2 = (int32_t*)(w + g.off_map2);
    int32_t* st1 = (int32_t*)(w + g.off_st1);
    int32_t* st2 = (int32_t*)(w + g.off_st2);
    int32_t* env = env_out ? env_out : (int32_t*)(w + g.off_env);
    int8_t* ffp = g.ff_bytes ? (int8_t*)(w + g.off_ff) : nullptr;
    int8_t* ffpath = ffp ? ffp + al256((size_t)std::max(tr1, tr2) * 8) : nullptr;
    void* tok = nullptr;
    int rc;
    // (1) the two 1-D basecalls with their frame maps (pair_decode.py:360-362, 377-382); with
    //     --single beam the caller supplies them (beam search + Viterbi acceptor, :363-370)
    if (ext_map1 && ext_map2) {
        map1 = const_cast<int32_t*>(ext_map1);
        map2 = const_cast<int32_t*>(ext_map2);
        if (po_zero_async(st1, sizeof(int32_t) * n, stream) != hipSuccess) return PO_E_HIP;
        if (po_zero_async(st2, sizeof(int32_t) * n, stream) != hipSuccess) return PO_E_HIP;
    } else if (!opt->diagonal_envelope) {
        po_prof_stage(PO_K_VITERBI, stream, 1, &tok);
        rc = po_launch_viterbi_strided(y1, y1_off, n, C, A, alphabet, kind, nullptr, seq1d, seq1d_off, 0, 2, len1, map1,
                                       st1, ffp, ffpath, stream);
        if (rc != PO_OK) return rc;
        rc = po_launch_viterbi_strided(y2, y2_off, n, C, A, alphabet, kind, nullptr, seq1d, seq1d_off, 1, 2, len2, map2,
                                       st2, ffp, ffpath, stream);
        if (rc != PO_OK) return rc;
        po_prof_stage(PO_K_VITERBI, stream, 0, &tok);
    } else {
        if (po_zero_async(len1, sizeof(int32_t) * n, stream) != hipSuccess) return PO_E_HIP;
        if (po_zero_async(len2, sizeof(int32_t) * n, stream) != hipSuccess) return PO_E_HIP;
    }
    // (2) alignment, skips, envelope
    PPArgs a;
    a.y1_off = y1_off; a.y2_off = y2_off; a.n = n;
    a.seq1d = seq1d; a.seq1d_off = seq1d_off; a.len1 = len1; a.len2 = len2;
    a.map1 = map1; a.map2 = map2; a.st1 = st1; a.st2 = st2;
    a.padding = opt->padding; a.full_alignment = opt->full_alignment;
    a.diagonal_envelope = opt->diagonal_envelope; a.diagonal_width = opt->diagonal_width;
    a.band = NW_BAND; a.mode = 0; a.retry_cap = 0;
    a.maps_increasing = (ext_map1 && ext_map2) ? 0 : 1;
    a.match = NW_MATCH; a.mismatch = NW_MISMATCH; a.gap = NW_GAP;
    a.lenU = a.lenV = nullptr; a.map1_off = a.map2_off = nullptr; a.aln_out1 = a.aln_out2 = nullptr;
    a.aln_off = nullptr; a.ncol_out = nullptr; a.env_off = nullptr;
    a.env = env; a.identity = identity; a.status = status;
    a.queue = (int*)(w + g.off_queue);
    a.cap_flag = a.queue + 32;   // (inside the 256 bytes cleared below)
    a.dp = (int*)(w + g.off_dp); a.dp_cap = (long long)g.dp_cap;
    a.rowinfo = (int*)(w + g.off_rows); a.row_cap = (long long)g.row_cap;
    a.aln = w + g.off_aln; a.aln_cap = (long long)g.aln_cap;
    if (po_zero_async(a.queue, 256, stream) != hipSuccess) return PO_E_HIP;
    po_prof_stage(PO_K_ALIGN, stream, 1, &tok);
    pp_launch(a, g.blocks, g.one_wave, stream, mr2 + 8);
    if (g.big_blocks > 0 && !opt->diagonal_envelope) {   // pairs whose basecalls did not fit the first pass's slices (none, usually)
        PPArgs b = a;
        b.retry_cap = 1;
        b.queue = a.queue + 16;
        b.dp = (int*)(w + g.off_big_dp); b.dp_cap = (long long)g.big_dp_cap;
        b.rowinfo = (int*)(w + g.off_big_rows); b.row_cap = (long long)g.big_row_cap;
        b.aln = w + g.off_big_aln; b.aln_cap = (long long)g.big_aln_cap;
        pp_launch(b, g.big_blocks, g.one_wave, stream, mr2 + 8);
    }
    po_prof_stage(PO_K_ALIGN, stream, 0, &tok);
    // (3) the pair beam search inside the envelope (pair_decode.py:166-173,511)
    po_prof_stage(PO_K_BEAM2D, stream, 1, &tok);
    rc = po_launch_beam2d_geom(y1, y1_off, y2, y2_off, env, n, C, A, alphabet, opt->beam_width, opt->model, opt->method,
                               tr1, tr2, mr1, mr2, seq, seq_off, seq_len, status, 1, w + g.off_b2, g.b2_bytes, stream);
    po_prof_stage(PO_K_BEAM2D, stream, 0, &tok);
    return rc;
}

extern "C" int po_launch_pair_decode(const double* y1, const int64_t* y1_off, const double* y2, const int64_t* y2_off,
                                     int n, int C, const po_pair_options* opt, char* seq1d, const int64_t* seq1d_off,
                                     int32_t* len1, int32_t* len2, double* identity, int32_t* env_out, char* seq,
                                     const int64_t* seq_off, int32_t* seq_len, int32_t* status, void* ws,
                                     size_t ws_bytes, hipStream_t stream) {
    if (n <= 0) return PO_OK;
    int64_t* h = (int64_t*)malloc(sizeof(int64_t) * 2 * (size_t)(n + 1));
    if (!h) return PO_E_NOMEM;
    int rc = PO_OK;
    if (hipMemcpyAsync(h, y1_off, sizeof(int64_t) * (n + 1), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipMemcpyAsync(h + n + 1, y2_off, sizeof(int64_t) * (n + 1), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
        rc = PO_E_HIP;
    int64_t m1 = 0, m2 = 0, t1 = 0, t2 = 0;
    if (rc == PO_OK) {
        for (int i = 0; i < n; ++i) {
            m1 = std::max<int64_t>(m1, h[i + 1] - h[i]);
            m2 = std::max<int64_t>(m2, h[n + 1 + i + 1] - h[n + 1 + i]);
        }
        t1 = h[n] - h[0];
        t2 = h[2 * n + 1] - h[n + 1];
    }
    free(h);
    if (rc != PO_OK) return rc;
    return po_launch_pair_decode_geom(y1, y1_off, y2, y2_off, n, C, opt, t1, t2, m1, m2, nullptr, nullptr, seq1d, seq1d_off, len1, len2,
                                      identity, env_out, seq, seq_off, seq_len, status, ws, ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------------
// standalone entry points of the two stages (align.global_pair / global_pair_banded; envelope.build_envelope)
extern "C" size_t po_align_ws_bytes(int n, int64_t max_len1, int64_t max_len2, int band) {
    const int blocks = std::min(n > 0 ? n : 1, pp_num_cus() * 4);
    const int64_t width = band > 0 ? std::min<int64_t>(max_len2 + 1, 2 * (int64_t)band + 1) : max_len2 + 1;
    size_t cells = (size_t)((max_len1 + 1) * width);
    if (band > 0) cells = std::max(cells, pp_skew_cells(max_len1, max_len2, band));
    return 256 + al256(sizeof(int) * cells * blocks) + al256(sizeof(int) * 4 * (size_t)(max_len1 + 2) * blocks) +
           al256(2 * (size_t)(max_len1 + max_len2 + 16) * blocks) + 256;
}

extern "C" int po_launch_align_scores(const char* seqs, const int64_t* seq_off, int n, int band, int match, int mismatch,
                                      int gap, int64_t max_len1, int64_t max_len2, char* aln1, char* aln2,
                                      const int64_t* aln_off, int32_t* ncol, int32_t* status, void* ws, size_t ws_bytes,
                                      hipStream_t stream);
extern "C" int po_launch_align(const char* seqs, const int64_t* seq_off, int n, int band, int64_t max_len1,
                               int64_t max_len2, char* aln1, char* aln2, const int64_t* aln_off, int32_t* ncol,
                               int32_t* status, void* ws, size_t ws_bytes, hipStream_t stream) {
    return po_launch_align_scores(seqs, seq_off, n, band, NW_MATCH, NW_MISMATCH, NW_GAP, max_len1, max_len2, aln1, aln2, aln_off,
                                  ncol, status, ws, ws_bytes, stream);
}
extern "C" int po_launch_align_scores(const char* seqs, const int64_t* seq_off, int n, int band, int match, int mismatch,
                                      int gap, int64_t max_len1, int64_t max_len2, char* aln1, char* aln2,
                                      const int64_t* aln_off, int32_t* ncol, int32_t* status, void* ws, size_t ws_bytes,
                                      hipStream_t stream) {
    if (n <= 0) return PO_OK;
    if (ws_bytes < po_align_ws_bytes(n, max_len1, max_len2, band)) return PO_E_CAP;
    const int blocks = std::min(n, pp_num_cus() * 4);
    const int64_t width = band > 0 ? std::min<int64_t>(max_len2 + 1, 2 * (int64_t)band + 1) : max_len2 + 1;
    char* w = (char*)ws;
    const int one_wave = band > 0 ? 1 : 0;
    PPArgs a = {};
    a.n = n; a.seq1d = seqs; a.seq1d_off = seq_off; a.mode = 1;
    a.full_alignment = band > 0 ? 0 : 1; a.band = band > 0 ? band : 0;
    a.match = match; a.mismatch = mismatch; a.gap = gap;
    a.aln_out1 = aln1; a.aln_out2 = aln2; a.aln_off = aln_off; a.ncol_out = ncol; a.status = status;
    size_t o = 0;
    a.queue = (int*)(w + o); o += 256;
    a.cap_flag = a.queue + 32;
    a.dp_cap = (long long)((max_len1 + 1) * width);
    if (band > 0) a.dp_cap = std::max<long long>(a.dp_cap, (long long)pp_skew_cells(max_len1, max_len2, band));
    a.dp = (int*)(w + o); o += al256(sizeof(int) * (size_t)a.dp_cap * blocks);
    a.row_cap = (long long)(max_len1 + 2); a.rowinfo = (int*)(w + o); o += al256(sizeof(int) * 4 * (size_t)a.row_cap * blocks);
    a.aln_cap = (long long)(max_len1 + max_len2 + 16); a.aln = w + o;
    if (po_zero_async(a.queue, 256, stream) != hipSuccess) return PO_E_HIP;
    pp_launch(a, blocks, one_wave, stream, max_len2);
    return PO_OK;
}

// ---- the dense DP matrix of align.global_pair (align.pyx:34-52), the third item the reference returns: (len1 + 1) x
// (len2 + 1) int32, row-major, boundary cells gap * i / gap * j, cell = max(diagonal + score, up + gap, left + gap).  An API
// completeness kernel, not a hot path: one wave per pair, a row in chunks of 64 columns — the left-neighbour dependency
// cell(j) = max(c(j), cell(j - 1) + gap) is a prefix maximum of c(k) - gap * k (integers: exact, order-free) —, the row
// above read back from the matrix itself (a wave's own stores, program order).
__global__ __launch_bounds__(64) void nw_matrix_kernel(const char* seqs, const int64_t* seq_off, int n, int match, int mismatch, int gap,
                                                       int32_t* dp, const int64_t* dp_off, int32_t* status) {
    const int p = (int)blockIdx.x, lane = (int)threadIdx.x;
    if (p >= n) return;
    const char* const s1 = seqs + seq_off[2 * p];
    const char* const s2 = seqs + seq_off[2 * p + 1];
    const int l1 = (int)(seq_off[2 * p + 1] - seq_off[2 * p]), l2 = (int)(seq_off[2 * p + 2] - seq_off[2 * p + 1]);
    int32_t* const M = dp + dp_off[p];
    const int64_t Wd = (int64_t)l2 + 1;
    for (int j = lane; j <= l2; j += 64) M[j] = gap * j;
    for (int i = 1; i <= l1; ++i) {
        po_wave_sync();
        const int32_t* const up_ = M + (int64_t)(i - 1) * Wd;
        int32_t* const row = M + (int64_t)i * Wd;
        const char ci = s1[i - 1];
        int carry = gap * i;   // cell(i, 0), as g(0) = cell - gap * 0
        if (lane == 0) row[0] = carry;
        for (int base = 1; base <= l2; base += 64) {
            const int j = base + lane;
            const bool ok = j <= l2;
            int v = INT_MIN / 2;
            if (ok) {
                const int sc = (ci == s2[j - 1]) ? match : mismatch;
                v = max(up_[j - 1] + sc, up_[j] + gap) - gap * j;
            }
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(v, d);
                if (lane >= d) v = max(v, o);
            }
            v = max(v, carry);
            if (ok) row[j] = v + gap * j;
            carry = __shfl(v, 63);
        }
    }
    if (lane == 0 && status) status[p] = PO_OK;
}
extern "C" int po_launch_nw_matrix(const char* seqs, const int64_t* seq_off, int n, int match, int mismatch, int gap, int32_t* dp,
                                   const int64_t* dp_off, int32_t* status, hipStream_t stream) {
    if (n <= 0) return PO_OK;
    hipLaunchKernelGGL(nw_matrix_kernel, dim3(n), dim3(64), 0, stream, seqs, seq_off, n, match, mismatch, gap, dp, dp_off, status);
    return hipGetLastError() == hipSuccess ? PO_OK : PO_E_HIP;
}

extern "C" size_t po_envelope_ws_bytes(int n, int64_t max_ncol) {
    const int blocks = std::min(n > 0 ? n : 1, pp_num_cus() * 4);
    return 256 + al256(2 * (size_t)(max_ncol + 16) * blocks) + 256;
}

extern "C" int po_launch_envelope(const char* aln1, const char* aln2, const int64_t* aln_off, const int32_t* ncol, int n,
                                  const int32_t* map1, const int64_t* map1_off, const int32_t* map2,
                                  const int64_t* map2_off, const int32_t* U, const int32_t* V, int padding,
                                  int64_t max_ncol, int32_t* env, const int64_t* env_off, int32_t* status, void* ws,
                                  size_t ws_bytes, hipStream_t stream) {
    if (n <= 0) return PO_OK;
    if (ws_bytes < po_envelope_ws_bytes(n, max_ncol)) return PO_E_CAP;
    const int blocks = std::min(n, pp_num_cus() * 4);
    char* w = (char*)ws;
    const int one_wave = 0;
    PPArgs a = {};
    a.n = n; a.mode = 2; a.padding = padding;
    a.match = NW_MATCH; a.mismatch = NW_MISMATCH; a.gap = NW_GAP;
    a.aln_out1 = const_cast<char*>(aln1); a.aln_out2 = const_cast<char*>(aln2); a.aln_off = aln_off;
    a.ncol_out = const_cast<int32_t*>(ncol);
    a.map1 = map1; a.map1_off = map1_off; a.map2 = map2; a.map2_off = map2_off; a.lenU = U; a.lenV = V;
    a.env = env; a.env_off = env_off; a.status = status;
    a.queue = (int*)w;
    a.cap_flag = a.queue + 32;
    a.aln_cap = (long long)(max_ncol + 16); a.aln = w + 256;
    a.dp = nullptr; a.dp_cap = 0; a.rowinfo = (int*)w; a.row_cap = 0;
    if (po_zero_async(a.queue, 256, stream) != hipSuccess) return PO_E_HIP;
    pp_launch(a, blocks, one_wave, stream);
    return PO_OK;
}

// pair decode with an externally supplied 1-D stage (pair_decode.py --single beam, :363-370): seq1d / len1 /
// len2 / map1 / map2 are INPUTS (maps: frame of every base, int32 at map + y*_off[i])
extern "C" int po_launch_pair_decode_from_1d(const double* y1, const int64_t* y1_off, const double* y2,
                                             const int64_t* y2_off, int n, int C, const po_pair_options* opt,
                                             int64_t tr1, int64_t tr2, int64_t mr1, int64_t mr2, const int32_t* map1,
                                             const int32_t* map2, char* seq1d, const int64_t* seq1d_off, int32_t* len1,
                                             int32_t* len2, double* identity, int32_t* env_out, char* seq,
                                             const int64_t* seq_off, int32_t* seq_len, int32_t* status, void* ws,
                                             size_t ws_bytes, hipStream_t stream) {
    if (!map1 || !map2) return PO_E_ARG;
    return po_launch_pair_decode_geom(y1, y1_off, y2, y2_off, n, C, opt, tr1, tr2, mr1, mr2, map1, map2, seq1d, seq1d_off,
                                      len1, len2, identity, env_out, seq, seq_off, seq_len, status, ws, ws_bytes, stream);
}
