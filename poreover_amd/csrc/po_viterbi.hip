// Batched argmax / Viterbi decode of (T, C) log-probability matrices.
//
// Replaces transducer.argmax_decode / poreover.viterbi_decode / bonito.viterbi_decode /
// transducer.viterbi_decode (reference transducer.py:27-59,72-73,83-89,94-103) and
// pair_decode.get_sequence_mapping (pair_decode.py:114-142).
//
// This is the one kernel on the path that is a pure HBM stream: 8*T*C bytes in, about
// T + 5*L bytes out per read (path, characters, frame map).  One workgroup per read; rows are
// copied HBM -> LDS with fully coalesced 8-byte-per-lane loads (a wave covers 512 contiguous
// bytes per instruction; all loads of a 512-frame tile are issued before the first use), the
// per-frame argmax is taken from LDS, and the emitted bases are compacted with a wave scan plus a
// 4-entry LDS scan across waves.
#include <cstdlib>

#include "po_device.h"

#define VT_THREADS 256
#define VT_WAVES (VT_THREADS / PO_WAVE)

// Exclusive prefix count of `flag` over the 256-thread block; returns this thread's offset
// and the block total through *total.  Uses LDS wsum[VT_WAVES].
__device__ __forceinline__ int block_exclusive_count(bool flag, int* wsum, int* total) {
    const unsigned long long mask = __ballot(flag);
    const int lane = po_lane(), wave = threadIdx.x >> 6;
    const int below = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(mask);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < VT_WAVES; ++w) {
        const int c = wsum[w];
        if (w < wave) base += c;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return base + below;
}

// kind: PO_KIND_POREOVER (blanks dropped, repeats kept) or PO_KIND_BONITO (groupby collapse).
//
// Streaming structure: a tile is VT_TILE = 512 frames (20 KB of log-probs at C = 5).  Each thread
// issues its VT_LD = 10 coalesced 8-byte loads of the tile back to back BEFORE the first use, so a
// workgroup keeps 20 KB in flight and a CU (6 resident workgroups) ~120 KB — enough to cover HBM
// latency.  Every thread then decodes VT_FPT = 2 consecutive frames from LDS.
// THREADS = 256 (four waves, the 512-frame tile above) or 64: ONE wave, a 128-frame tile and 5 KB of LDS — the form that finds
// room next to a device full of one-wave pair beam workgroups (the pipelined job: wave k + 1's Viterbi pass runs while wave k's
// pair beam kernel holds the device; a four-wave workgroup with 21 KB of LDS waits for a CU to drain).
#define VT_FPT 2
#define VT_CMAX (PO_A + 1)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void viterbi_ctc_kernel(
    const double* __restrict__ y, const int64_t* __restrict__ y_off, int C, uint32_t alphabet, int kind,
    int8_t* __restrict__ path, char* __restrict__ seq, const int64_t* __restrict__ seq_off, int so_base,
    int so_stride, int32_t* __restrict__ seq_len, int32_t* __restrict__ map, int32_t* __restrict__ status) {
    constexpr int VT_WAVES_ = THREADS / PO_WAVE, VT_TILE = THREADS * VT_FPT, VT_LD = (VT_TILE * VT_CMAX) / THREADS;
    __shared__ double tile[VT_TILE * VT_CMAX];
    __shared__ int wsum[VT_WAVES_];
    __shared__ int8_t pth[VT_TILE + 1];  // pth[0] = last state of the previous tile
    const int r = blockIdx.x, tid = threadIdx.x;
    const int64_t r0 = y_off[r];
    const int T = (int)(y_off[r + 1] - r0);
    const int blank = C - 1;
    const double* yr = y + r0 * C;
    const int64_t so = seq_off[so_base + (int64_t)r * so_stride];
    char* sq = seq + so;
    const int cap = (int)(seq_off[so_base + (int64_t)r * so_stride + 1] - so);
    int32_t* mp = map ? map + r0 : nullptr;
    int n_seq = 0, n_map = 0, st = PO_OK;

    // get_sequence_mapping('bonito') compares frame 0 with path[-1], i.e. the LAST frame
    int last_state = -1;
    if (kind == PO_KIND_BONITO && T > 0) {
        const double* row = yr + (int64_t)(T - 1) * C;
        int b = 0;
        double bv = row[0];
        for (int c = 1; c < C; ++c) {
            const double v = row[c];
            if (v > bv) { bv = v; b = c; }
        }
        last_state = b;
    }
    if (tid == 0) pth[0] = -1;
    __syncthreads();

    for (int t0 = 0; t0 < T; t0 += VT_TILE) {
        const int rows = min(VT_TILE, T - t0);
        const int nval = rows * C;
        const double* src = yr + (int64_t)t0 * C;
        // 16-byte loads (two doubles per lane and instruction), all issued before any use; a tile's byte offset in
        // the read is a multiple of 16, the read's own start only of 8 (dwordx4 needs 4-byte alignment)
        double2 reg[VT_LD / 2];
        const double2* src2 = (const double2*)src;
        const int nval2 = nval >> 1;
#pragma unroll
        for (int q = 0; q < VT_LD / 2; ++q) {
            const int i = tid + q * THREADS;
            reg[q] = (i < nval2) ? src2[i] : make_double2(0.0, 0.0);
        }
        const double tail = (nval & 1) ? src[nval - 1] : 0.0;   // odd number of values in a short last tile
#pragma unroll
        for (int q = 0; q < VT_LD / 2; ++q) {
            const int i = tid + q * THREADS;
            if (i < nval2) { tile[2 * i] = reg[q].x; tile[2 * i + 1] = reg[q].y; }
        }
        if ((nval & 1) && tid == 0) tile[nval - 1] = tail;
        __syncthreads();
        int p[VT_FPT];
#pragma unroll
        for (int f = 0; f < VT_FPT; ++f) {  // np.argmax: first maximum wins
            const int fr = tid * VT_FPT + f;
            p[f] = blank;
            if (fr < rows) {
                double bv = tile[fr * C];
                int b = 0;
                for (int c = 1; c < C; ++c) {
                    const double v = tile[fr * C + c];
                    if (v > bv) { bv = v; b = c; }
                }
                p[f] = b;
                pth[fr + 1] = (int8_t)b;
                if (path) path[r0 + t0 + fr] = (int8_t)b;
            }
        }
        __syncthreads();
        bool emit[VT_FPT], emit_map[VT_FPT];
        int ne = 0, nm = 0;
#pragma unroll
        for (int f = 0; f < VT_FPT; ++f) {
            const int fr = tid * VT_FPT + f, t = t0 + fr;
            const int prev = pth[fr];  // state of frame t-1 (-1 before frame 0)
            emit[f] = emit_map[f] = false;
            if (fr < rows && p[f] != blank) {
                if (kind == PO_KIND_POREOVER) {
                    emit[f] = emit_map[f] = true;
                } else {
                    emit[f] = (t == 0) || (p[f] != prev);
                    emit_map[f] = (p[f] != ((t == 0) ? last_state : prev));
                }
            }
            ne += emit[f];
            nm += emit_map[f];
        }
        // exclusive prefix of per-thread counts: wave scan + 4-entry cross-wave scan
        auto block_excl = [&](int v, int* total) {
            const int lane = po_lane(), wave = tid >> 6;
            int inc = v;
#pragma unroll
            for (int o = 1; o < PO_WAVE; o <<= 1) {
                const int t = __shfl_up(inc, o);
                if (lane >= o) inc += t;
            }
            if (lane == PO_WAVE - 1) wsum[wave] = inc;
            __syncthreads();
            int base = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < VT_WAVES_; ++w) {
                const int c = wsum[w];
                if (w < wave) base += c;
                tot += c;
            }
            __syncthreads();
            *total = tot;
            return base + inc - v;
        };
        // one scan for both compactions: bases in the low half, frame-map entries in the high half (<= 512 each)
        int tot_p;
        const int excl = block_excl(ne | (nm << 16), &tot_p);
        int pos_s = n_seq + (excl & 0xffff);
#pragma unroll
        for (int f = 0; f < VT_FPT; ++f)
            if (emit[f]) {
                if (pos_s < cap) sq[pos_s] = (char)((alphabet >> (8 * p[f])) & 0xffu);
                else st = PO_E_CAP;
                pos_s++;
            }
        n_seq += tot_p & 0xffff;
        if (mp) {
            int pos_m = n_map + (excl >> 16);
#pragma unroll
            for (int f = 0; f < VT_FPT; ++f)
                if (emit_map[f]) mp[pos_m++] = t0 + tid * VT_FPT + f;
            n_map += tot_p >> 16;
        }
        if (tid == 0) pth[0] = pth[rows];
        __syncthreads();
    }
    if (__syncthreads_or(st != PO_OK)) st = PO_E_CAP;
    // the reference asserts len(sequence_to_signal) == len(basecall) (pair_decode.py:379,382)
    if (st == PO_OK && mp && n_map != n_seq) st = PO_E_ARG;
    if (T < 1) st = PO_E_ARG;
    if (tid == 0) {
        seq_len[r] = n_seq;
        status[r] = st;
    }
}

// ---------------------------------------------------------------------------------------------
// Flip-flop Viterbi (transducer.py:35-59 with the 8x8 transition matrix :94-103).  The 0/1
// "transition" is ADDED to the log-probabilities, as in the reference (:44): illegal moves cost
// one nat less than legal ones rather than being forbidden.
// One 8-lane group per read (lane j = state j), 8 reads per wave.  ptr is int8[T][8] per read.
#define FF_S 8
__global__ __launch_bounds__(PO_WAVE) void flipflop_dp_kernel(const double* __restrict__ y,
                                                              const int64_t* __restrict__ y_off, int n, int A,
                                                              int8_t* __restrict__ ptr,
                                                              int8_t* __restrict__ path,
                                                              int32_t* __restrict__ status) {
    __shared__ __attribute__((aligned(8))) int8_t chunk[8][512 * FF_S];
    const int lane = po_lane(), g = lane >> 3, j = lane & 7;
    const int r = blockIdx.x * 8 + g;
    const bool live = r < n;
    const int64_t r0 = live ? y_off[r] : 0;
    const int T = live ? (int)(y_off[r + 1] - r0) : 0;
    int Tmax = T;
#pragma unroll
    for (int o = 8; o < PO_WAVE; o <<= 1) Tmax = max(Tmax, __shfl_xor(Tmax, o));
    const int S = 2 * A;  // live states; lanes j >= S of a group idle
    const double* yr = y + r0 * S;
    int8_t* pr = ptr + r0 * FF_S;
    double v = (T > 0 && j < S) ? yr[j] : PO_NEG_INF;
    // The frames are one dependent chain per read and a launch has little more than one wave per SIMD (8 reads per wave): a row
    // asked for in the frame that uses it is a memory round trip PER FRAME.  The rows of the next FF_PF frames are requested
    // while this block's frames are computed — and the block's back-pointers are written at the START of the next block, behind
    // those requests: gfx9 counts loads and stores on one counter, so the wait for the rows (s_waitcnt vmcnt(0)) also waits for
    // every store in flight; issued a block ahead of that wait, they have landed by then.
    constexpr int FF_PF = 16;
    auto row_at = [&](int t) -> double { return (t < T && j < S) ? yr[(int64_t)t * S + j] : 0.0; };
    double yq[FF_PF];
    int bq[FF_PF];      // the previous block's back-pointers (-1: none)
#pragma unroll
    for (int i = 0; i < FF_PF; ++i) { yq[i] = row_at(1 + i); bq[i] = -1; }
    for (int t0 = 1; t0 < Tmax + FF_PF; t0 += FF_PF) {   // (one trip more: the last block's stores)
        // (this block's rows — requested a block ago — are waited for HERE, before the next requests go out: left to their first
        //  use, the wait would sit behind those requests and, being vmcnt(0), wait for them too)
#pragma unroll
        for (int i = 0; i < FF_PF; ++i) po_settle(yq[i]);
        double yn[FF_PF];
#pragma unroll
        for (int i = 0; i < FF_PF; ++i) yn[i] = row_at(t0 + FF_PF + i);
#pragma unroll
        for (int i = 0; i < FF_PF; ++i)
            if (bq[i] >= 0) pr[(int64_t)(t0 - FF_PF + i) * FF_S + j] = (int8_t)bq[i];
#pragma unroll
        for (int f = 0; f < FF_PF; ++f) {
            const int t = t0 + f;
            bq[f] = -1;
            if (t < Tmax) {   // (wave-uniform)
                const bool on = t < T;
                const double yt = yq[f];
                double bv = 0;
                int bi = 0;
#pragma unroll
                for (int i = 0; i < FF_S; ++i) {
                    const double vi = __shfl(v, (g << 3) | i);
                    const double tr = (j < A) ? 1.0 : (((i % A) == (j - A)) ? 1.0 : 0.0);
                    const double cand = tr + vi;
                    if (i < S && (i == 0 || cand > bv)) { bv = cand; bi = i; }
                }
                if (on && j < S) {
                    bq[f] = bi;
                    v = yt + bv;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < FF_PF; ++i) yq[i] = yn[i];
    }
    // argmax of the final column (first maximum), held by every lane of the group
    double bv = __shfl(v, g << 3);
    int best = 0;
#pragma unroll
    for (int i = 1; i < FF_S; ++i) {
        const double vi = __shfl(v, (g << 3) | i);
        if (i < S && vi > bv) { bv = vi; best = i; }
    }
    __threadfence_block();
    // back-trace in 512-frame chunks staged through LDS (the chain of dependent reads then runs
    // at LDS latency instead of L2 latency)
    int cur = best;
    for (int lo = ((Tmax + 511) / 512 - 1) * 512; lo >= 0; lo -= 512) {  // uniform trip count
        const int hi = min(T, lo + 512), len = hi - lo;                  // len <= 0: nothing here
        {   // (a frame's eight back-pointers are eight consecutive bytes, 8-byte aligned: one load per frame, not one per byte)
            const unsigned long long* src = (const unsigned long long*)(pr + (int64_t)lo * FF_S);
            unsigned long long* dst = (unsigned long long*)&chunk[g][0];
            if (((unsigned long long)src & 7ull) == 0ull) {
                for (int f = j; f < len; f += 8) dst[f] = src[f];
            } else {   // (a workspace that is not 8-byte aligned: byte by byte)
                for (int i = j; i < len * FF_S; i += 8) chunk[g][i] = pr[(int64_t)lo * FF_S + i];
            }
        }
        __syncthreads();
        if (j == 0) {
            for (int t = hi - 1; t >= lo; --t) {
                path[r0 + t] = (int8_t)cur;
                if (t > 0) cur = chunk[g][(t - lo) * FF_S + cur];  // path[t-1] = ptr[t][path[t]]
            }
        }
        cur = __shfl(cur, g << 3);
        __syncthreads();
    }
    if (live && j == 0) status[r] = (T < 1) ? PO_E_ARG : PO_OK;
}

// remove_repeated(...).upper() (transducer.py:4-9,55) and get_sequence_mapping('flipflop')
// (pair_decode.py:124-132) from a stored flip-flop state path: a base is emitted at frame 0 and
// wherever the state changes.
__global__ __launch_bounds__(VT_THREADS) void flipflop_compact_kernel(
    const int8_t* __restrict__ path, const int64_t* __restrict__ y_off, int A, uint32_t alphabet,
    char* __restrict__ seq, const int64_t* __restrict__ seq_off, int so_base, int so_stride,
    int32_t* __restrict__ seq_len, int32_t* __restrict__ map, int32_t* __restrict__ status) {
    __shared__ int wsum[VT_WAVES];
    const int r = blockIdx.x, tid = threadIdx.x;
    const int64_t r0 = y_off[r];
    const int T = (int)(y_off[r + 1] - r0);
    const int64_t so = seq_off[so_base + (int64_t)r * so_stride];
    char* sq = seq + so;
    const int cap = (int)(seq_off[so_base + (int64_t)r * so_stride + 1] - so);
    int n_seq = 0, st = PO_OK;
    for (int t0 = 0; t0 < T; t0 += VT_THREADS) {
        const int t = t0 + tid;
        bool emit = false;
        int p = 0;
        if (t < T) {
            p = path[r0 + t];
            emit = (t == 0) || (p != path[r0 + t - 1]);
        }
        int tot;
        const int pos = n_seq + block_exclusive_count(emit, wsum, &tot);
        if (emit) {
            if (pos < cap) {
                sq[pos] = (char)((alphabet >> (8 * (p % A))) & 0xffu);
                if (map) map[r0 + pos] = t;
            } else st = PO_E_CAP;
        }
        n_seq += tot;
    }
    if (__syncthreads_or(st != PO_OK)) st = PO_E_CAP;
    if (tid == 0) {
        seq_len[r] = n_seq;
        if (status[r] == PO_OK) status[r] = st;
    }
}

extern "C" int po_launch_viterbi_strided(const double* y, const int64_t* y_off, int n, int C, int A,
                                         uint32_t alphabet, int kind, int8_t* path, char* seq,
                                         const int64_t* seq_off, int so_base, int so_stride, int32_t* seq_len,
                                         int32_t* map, int32_t* status, int8_t* ff_ptr, int8_t* ff_path,
                                         hipStream_t stream);

extern "C" int po_launch_viterbi(const double* y, const int64_t* y_off, int n, int C, int A, uint32_t alphabet,
                                 int kind, int8_t* path, char* seq, const int64_t* seq_off, int32_t* seq_len,
                                 int32_t* map, int32_t* status, int8_t* ff_ptr, int8_t* ff_path,
                                 hipStream_t stream) {
    return po_launch_viterbi_strided(y, y_off, n, C, A, alphabet, kind, path, seq, seq_off, 0, 1, seq_len, map,
                                     status, ff_ptr, ff_path, stream);
}

extern "C" int po_launch_viterbi_strided(const double* y, const int64_t* y_off, int n, int C, int A,
                                         uint32_t alphabet, int kind, int8_t* path, char* seq,
                                         const int64_t* seq_off, int so_base, int so_stride, int32_t* seq_len,
                                         int32_t* map, int32_t* status, int8_t* ff_ptr, int8_t* ff_path,
                                         hipStream_t stream) {
    if (n <= 0) return PO_OK;
    if (A < 1 || A > PO_A) return PO_E_ARG;
    if (kind == PO_KIND_FLIPFLOP) {
        if (C != 2 * A) return PO_E_ARG;
        int8_t* pth = path ? path : ff_path;
        hipLaunchKernelGGL(flipflop_dp_kernel, dim3((n + 7) / 8), dim3(PO_WAVE), 0, stream, y, y_off, n, A, ff_ptr,
                           pth, status);
        hipLaunchKernelGGL(flipflop_compact_kernel, dim3(n), dim3(VT_THREADS), 0, stream, pth, y_off, A, alphabet,
                           seq, seq_off, so_base, so_stride, seq_len, map, status);
        return PO_OK;
    }
    if (kind != PO_KIND_POREOVER && kind != PO_KIND_BONITO) return PO_E_ARG;
    if (C != A + 1) return PO_E_ARG;
    // (PO_VT_THREADS=256: the four-wave form, for A/B)
    static const int vt_threads = [] { const char* e = getenv("PO_VT_THREADS"); return (e && atoi(e) == 256) ? 256 : 64; }();
    if (vt_threads == 256)
        hipLaunchKernelGGL(viterbi_ctc_kernel<256>, dim3(n), dim3(256), 0, stream, y, y_off, C, alphabet, kind, path, seq,
                           seq_off, so_base, so_stride, seq_len, map, status);
    else
        hipLaunchKernelGGL(viterbi_ctc_kernel<64>, dim3(n), dim3(64), 0, stream, y, y_off, C, alphabet, kind, path, seq,
                           seq_off, so_base, so_stride, seq_len, map, status);
    return PO_OK;
}
