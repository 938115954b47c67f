// Batched 1-D prefix search (Graves) over windows of a (T, C) log-probability matrix.
//
// Replaces prefix_search.prefix_search_log_cy (reference prefix_search.py:176-238) with its helpers
// decoding_cy.forward_vec_log (decoding_cy.pyx:127-156: LOG_0 = -9999, log(exp(a) + exp(b))) and
// prefix_search.forward_vec_no_gap_log (prefix_search.py:67-79), as used by `decode --algorithm
// prefix` on consecutive windows of --window frames (decode.py:179-188).
//
// Per search level (one more symbol on the current prefix), for each symbol c:
//   alpha_ast[t] = (t == 0 ? (level == 1 ? 0 : -inf) : alpha_prev[t-1]) + y[t][c]
//   prefix_prob[c] = logsumexp_t alpha_ast[t]                       (scipy: max-shifted)
//   alpha_c[t]   = log(exp(y[t][blank] + alpha_c[t-1]) + exp(y[t][c] + alpha_prev[t-1])), alpha_c[0] special
//   label_prob[c] = alpha_c[T-1]
// then the best label / best prefix bookkeeping of the reference, in its order, and the stop test
// prefix_prob[best_prefix] < label_prob[top_label].
//
// Mapping: one workgroup (256 threads) per window; the alpha rows live in LDS.  The four alpha_c
// recurrences are serial in t and run on four lanes side by side; the reductions use all threads.
// The serial chain uses the reference's arithmetic step for step; the reductions sum in a different
// order than numpy (pairwise) does, so prefix_prob / the blank-only probability agree to rounding
// (tests compare with np.isclose, like the reference's own tests).
#include <algorithm>
#include <cstdlib>

#include "po_device.h"

namespace {
constexpr int PS_THREADS = 256;
constexpr int PS_WAVES = PS_THREADS / PO_WAVE;

struct PSArgs {
    const double* y; const int64_t* y_off; int n, C, A;
    uint32_t alphabet;
    char* seq; const int64_t* seq_off; int32_t* seq_len; double* logp; int32_t* status;
    char* curr; long long curr_cap;  // per workgroup: the current prefix
    int tcap;                        // LDS rows hold tcap doubles
};
}  // namespace

__global__ __launch_bounds__(PS_THREADS) void prefix_search_kernel(PSArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[PS_WAVES];
    __shared__ double shd[16];
    __shared__ int shi[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, pi = blockIdx.x;
    const int A = a.A, C = a.C, blank = a.A;
    const int64_t r0 = a.y_off[pi];
    const int T = (int)(a.y_off[pi + 1] - r0);
    const double* y = a.y + r0 * C;
    double* alpha_prev = (double*)smem;                 // [tcap]
    double* alpha = alpha_prev + a.tcap;                // [PO_A][tcap]
    char* curr = a.curr + (size_t)blockIdx.x * a.curr_cap;
    const double LOG0 = -9999.0;                        // decoding_cy.pyx:18
    if (T < 1 || T > a.tcap || T + 2 > a.curr_cap) {
        if (tid == 0) { a.seq_len[pi] = 0; a.logp[pi] = 0.0; a.status[pi] = (T < 1) ? PO_E_ARG : PO_E_CAP; }
        return;
    }
    auto block_max = [&](double v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
        if (lane == 0) red[wave] = v;
        __syncthreads();
        double r = red[0];
        for (int w = 1; w < PS_WAVES; ++w) r = fmax(r, red[w]);
        __syncthreads();
        return r;
    };
    auto block_sum = [&](double v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave] = v;
        __syncthreads();
        double r = 0;
        for (int w = 0; w < PS_WAVES; ++w) r += red[w];
        __syncthreads();
        return r;
    };
    // gap_prob = np.sum(y[:, -1]); alpha_prev = forward_vec_log(-1, 0, y) = running sum of blanks
    {
        double part = 0;
        for (int t = tid; t < T; t += PS_THREADS) part += y[(int64_t)t * C + blank];
        const double gp = block_sum(part);
        if (tid == 0) {
            shd[0] = gp;  // label_prob[top_label]
            double acc = 0;
            for (int t = 0; t < T; ++t) {
                acc = (t == 0) ? y[blank] : y[(int64_t)t * C + blank] + acc;
                alpha_prev[t] = acc;
            }
            shi[0] = 0;   // len(curr_label)
            shi[1] = 0;   // level at which top_label was set (0: the empty label)
            shi[2] = 0;   // its last symbol
            shi[3] = 0;   // stop flag
        }
        __syncthreads();
    }
    int st = PO_OK;
    for (int level = 1;; ++level) {
        if (level > T) { st = PO_E_DIVERGE; break; }   // prefix_forward[.., level-1] index error upstream
        const int curr_len = shi[0];
        // ---- prefix probabilities: logsumexp over t of alpha_ast, for each symbol
        for (int c = 0; c < A; ++c) {
            double m = PO_NEG_INF;
            for (int t = tid; t < T; t += PS_THREADS) {
                const double prev = (t == 0) ? (curr_len + 1 == 1 ? 0.0 : PO_NEG_INF) : alpha_prev[t - 1];
                m = fmax(m, prev + y[(int64_t)t * C + c]);
            }
            m = block_max(m);
            const double ms = (m > PO_NEG_INF && m < __builtin_inf()) ? m : 0.0;  // scipy: non-finite max -> 0
            double sacc = 0;
            for (int t = tid; t < T; t += PS_THREADS) {
                const double prev = (t == 0) ? (curr_len + 1 == 1 ? 0.0 : PO_NEG_INF) : alpha_prev[t - 1];
                sacc += exp(prev + y[(int64_t)t * C + c] - ms);
            }
            sacc = block_sum(sacc);
            if (tid == 0) shd[4 + c] = log(sacc) + ms;  // prefix_prob[curr + c]
        }
        // ---- label probabilities: the serial forward recurrences, one lane per symbol
        if (tid < A) {
            const int c = tid;
            double* al = alpha + (size_t)c * a.tcap;
            double fw = (level == 1) ? y[c] : LOG0;     // t == 0: i == 1 -> y[0, s], else stays LOG_0
            al[0] = fw;
            for (int t = 1; t < T; ++t) {
                fw = log(exp(y[(int64_t)t * C + blank] + fw) + exp(y[(int64_t)t * C + c] + alpha_prev[t - 1]));
                al[t] = fw;
            }
            shd[8 + c] = fw;  // label_prob[curr + c] = alpha[-1]
        }
        __syncthreads();
        // ---- bookkeeping in the reference's order (prefix_search.py:203-233)
        if (tid == 0) {
            int best = 0;
            double top = shd[0];
            for (int c = 0; c < A; ++c) {
                if (shd[8 + c] > top) { top = shd[8 + c]; shi[1] = level; shi[2] = c; }
                if (c > 0 && shd[4 + c] > shd[4 + best]) best = c;
            }
            shd[0] = top;
            if (shd[4 + best] < top) shi[3] = 1;
            else {
                curr[curr_len] = (char)((a.alphabet >> (8 * best)) & 0xffu);
                shi[0] = curr_len + 1;
                shi[4] = best;
            }
        }
        __syncthreads();
        if (shi[3]) break;
        const double* src = alpha + (size_t)shi[4] * a.tcap;
        for (int t = tid; t < T; t += PS_THREADS) alpha_prev[t] = src[t];
        __syncthreads();
    }
    __syncthreads();
    // top_label = curr[:top_level - 1] + top_symbol (curr only ever grows by appending)
    if (tid == 0) {
        int n = 0;
        if (st == PO_OK) {
            const int tl = shi[1];
            n = tl;  // 0 for the empty label
            char* out = a.seq + a.seq_off[pi];
            const int cap = (int)(a.seq_off[pi + 1] - a.seq_off[pi]);
            if (n > cap) { st = PO_E_CAP; n = 0; }
            else if (n > 0) {
                for (int i = 0; i < n - 1; ++i) out[i] = curr[i];
                out[n - 1] = (char)((a.alphabet >> (8 * shi[2])) & 0xffu);
            }
        }
        a.seq_len[pi] = n;
        a.logp[pi] = shd[0];
        a.status[pi] = st;
    }
}

namespace {
inline size_t al256(size_t b) { return (b + 255) & ~size_t(255); }
}

extern "C" size_t po_prefix_ws_bytes(int n, int64_t max_rows) { return al256((size_t)max_rows + 8) * (size_t)(n > 0 ? n : 1) + 256; }

extern "C" int po_launch_prefix_search(const double* y, const int64_t* y_off, int n, int C, int A, uint32_t alphabet,
                                       int64_t max_rows, char* seq, const int64_t* seq_off, int32_t* seq_len,
                                       double* logp, int32_t* status, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (n <= 0) return PO_OK;
    if (A < 1 || A > PO_A || C != A + 1) return PO_E_ARG;
    const size_t lds = sizeof(double) * (size_t)(PO_A + 1) * (size_t)max_rows;
    if (lds > 150 * 1024) return PO_E_UNSUPPORTED;  // windows longer than ~3800 frames do not fit the LDS rows
    const size_t per = al256((size_t)max_rows + 8);
    if (ws_bytes < per * n) return PO_E_CAP;
    PSArgs a;
    a.y = y; a.y_off = y_off; a.n = n; a.C = C; a.A = A; a.alphabet = alphabet;
    a.seq = seq; a.seq_off = seq_off; a.seq_len = seq_len; a.logp = logp; a.status = status;
    a.curr = (char*)ws; a.curr_cap = (long long)per; a.tcap = (int)max_rows;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)prefix_search_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(prefix_search_kernel, dim3(n), dim3(PS_THREADS), lds, stream, a);
    return PO_OK;
}

// ================================================================================================
// Batched PAIR prefix search on small boxes of two reads (dense gamma).
//
// Replaces prefix_search.pair_prefix_search_log / _cy (reference prefix_search.py:247-385) with
// pair_prefix_prob_log (:240-245): per search level and symbol c
//   alpha_ast_r[t] = (t == 0 ? (level == 1 ? 0 : -inf) : alpha_prev_r[t-1]) + y_r[t][c]        r = 1, 2
//   prefix_prob[c] = logsumexp_{u,v}(alpha_ast_1[u] + alpha_ast_2[v] + gamma[u+1][v+1]) - gamma[0][0]
//   alpha_c_r      = forward_vec_log(c, level, y_r, alpha_prev_r)                               (serial in t)
//   label_prob[c]  = alpha_c_1[U-1] + alpha_c_2[V-1] - gamma[0][0]
// and the reference's bookkeeping: label_prob is a dict over every label scored so far; while the best
// prefix probability is not below label_prob[top_label], top_label becomes the first maximum of the dict
// (insertion order) and the best prefix is extended.  gamma = the dense (U+1) x (V+1) matrix of
// pair_gamma_log (po_gamma.hip, either flavour).  flavor 0: prefix_search.py arithmetic (-inf,
// np.logaddexp); flavor 1: decoding_cy (LOG_0 = -9999, log(exp(a) + exp(b))).
//
// Mapping: one workgroup (256 threads) per box; alpha rows in LDS; the 2 x A forward recurrences run on
// 2A lanes side by side; the U x V log-sum-exp uses all threads (max-shifted like scipy; summation order
// differs from numpy's pairwise sum, so probabilities agree to rounding, as in the 1-D kernel).
namespace {
struct PPSArgs {
    const double* y1; const int64_t* y1_off; const double* y2; const int64_t* y2_off;
    const double* gm; const int64_t* gm_off;
    int n, C, A, flavor;
    uint32_t alphabet;
    char* seq; const int64_t* seq_off; int32_t* seq_len; double* logp; int32_t* status;
    char* curr; long long curr_cap;
    int tcap;   // LDS rows hold tcap doubles
};
__device__ __forceinline__ double pps_lae(double a, double b, int flavor) {
    if (flavor) return log(exp(a) + exp(b));
    // np.logaddexp: equal -> a + ln2; else max + log1p(exp(-|a-b|)); NaN only from NaN inputs
    if (a == b) return a + 0.6931471805599453;
    const double d = a - b;
    if (d > 0) return a + log1p(exp(-d));
    if (d <= 0) return b + log1p(exp(d));
    return d;  // NaN
}
}  // namespace

__global__ __launch_bounds__(PS_THREADS) void pair_prefix_search_kernel(PPSArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[PS_WAVES];
    __shared__ double shd[16];
    __shared__ int shi[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, pi = blockIdx.x;
    const int A = a.A, C = a.C, blank = a.A, tc = a.tcap;
    const int64_t r1 = a.y1_off[pi], r2 = a.y2_off[pi];
    const int U = (int)(a.y1_off[pi + 1] - r1), V = (int)(a.y2_off[pi + 1] - r2);
    const double* y1 = a.y1 + r1 * C;
    const double* y2 = a.y2 + r2 * C;
    const double* gm = a.gm + a.gm_off[pi];
    const int W1 = V + 1;
    // LDS rows: prev[2], ast[2], cand[2][PO_A]  (each tc doubles)
    double* prev1 = (double*)smem;
    double* prev2 = prev1 + tc;
    double* ast1 = prev2 + tc;
    double* ast2 = ast1 + tc;
    double* cand = ast2 + tc;  // [r][c][tc]
    char* curr = a.curr + (size_t)blockIdx.x * a.curr_cap;
    const double LOG0 = a.flavor ? -9999.0 : PO_NEG_INF;
    const int M = max(U, V);
    if (U < 1 || V < 1 || U > tc || V > tc || M + 4 > a.curr_cap) {
        if (tid == 0) { a.seq_len[pi] = 0; a.logp[pi] = 0.0; a.status[pi] = (U < 1 || V < 1) ? PO_E_ARG : PO_E_CAP; }
        return;
    }
    auto block_max = [&](double v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
        if (lane == 0) red[wave] = v;
        __syncthreads();
        double r = red[0];
        for (int w = 1; w < PS_WAVES; ++w) r = fmax(r, red[w]);
        __syncthreads();
        return r;
    };
    auto block_sum = [&](double v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave] = v;
        __syncthreads();
        double r = 0;
        for (int w = 0; w < PS_WAVES; ++w) r += red[w];
        __syncthreads();
        return r;
    };
    // label_prob[''] = sum of blanks of both reads; alpha_prev = forward_vec_log(-1, 0, y) (:262-268)
    if (tid < 2) {
        const double* y = tid ? y2 : y1;
        double* pv = tid ? prev2 : prev1;
        const int T = tid ? V : U;
        double g = 0.0, acc = 0.0;
        for (int t = 0; t < T; ++t) {
            const double b = y[(int64_t)t * C + blank];
            g += b;
            acc = (t == 0) ? b : b + acc;
            pv[t] = acc;
        }
        shd[12 + tid] = g;
    }
    __syncthreads();
    if (tid == 0) {
        shd[0] = shd[12] + shd[13];  // label_prob[top_label]
        shi[0] = 0;  // len(curr_label)
        shi[1] = 0;  // level at which top_label was scored (0: the empty label)
        shi[2] = 0;  // its last symbol
        shi[3] = 0;  // stop flag
    }
    __syncthreads();
    const double g00 = gm[0];
    int st = PO_OK;
    for (int level = 1;; ++level) {
        const int curr_len = shi[0];
        const bool depth_stop = curr_len > M;  // 'Max search depth exceeded' (:277-279): finish this level, then stop
        for (int c = 0; c < A; ++c) {
            for (int t = tid; t < U; t += PS_THREADS)
                ast1[t] = ((t == 0) ? (curr_len + 1 == 1 ? 0.0 : PO_NEG_INF) : prev1[t - 1]) + y1[(int64_t)t * C + c];
            for (int t = tid; t < V; t += PS_THREADS)
                ast2[t] = ((t == 0) ? (curr_len + 1 == 1 ? 0.0 : PO_NEG_INF) : prev2[t - 1]) + y2[(int64_t)t * C + c];
            __syncthreads();
            const int64_t N = (int64_t)U * V;
            double m = PO_NEG_INF;
            for (int64_t i = tid; i < N; i += PS_THREADS) {
                const int u = (int)(i / V), v = (int)(i - (int64_t)u * V);
                m = fmax(m, ast1[u] + ast2[v] + gm[(int64_t)(u + 1) * W1 + v + 1]);
            }
            m = block_max(m);
            const double ms = (m > PO_NEG_INF && m < __builtin_inf()) ? m : 0.0;  // scipy: non-finite max -> 0
            double sacc = 0;
            for (int64_t i = tid; i < N; i += PS_THREADS) {
                const int u = (int)(i / V), v = (int)(i - (int64_t)u * V);
                sacc += exp(ast1[u] + ast2[v] + gm[(int64_t)(u + 1) * W1 + v + 1] - ms);
            }
            sacc = block_sum(sacc);
            if (tid == 0) shd[4 + c] = log(sacc) + ms - g00;  // prefix_prob[curr + c]
            __syncthreads();
        }
        // forward rows of the A candidate labels on both reads: 2A serial recurrences side by side
        if (tid < 2 * A) {
            const int rr = tid / A, c = tid - rr * A;
            const double* y = rr ? y2 : y1;
            const double* pv = rr ? prev2 : prev1;
            const int T = rr ? V : U;
            double* al = cand + ((size_t)rr * PO_A + c) * tc;
            double fw = (level == 1) ? y[c] : LOG0;  // t == 0: i == 1 -> y[0, s], else stays LOG_0
            al[0] = fw;
            for (int t = 1; t < T; ++t) {
                fw = pps_lae(y[(int64_t)t * C + blank] + fw, y[(int64_t)t * C + c] + pv[t - 1], a.flavor);
                al[t] = fw;
            }
        }
        __syncthreads();
        if (tid == 0) {
            // label_prob of the A new labels; the dict's first maximum (insertion order) incl. them
            double top = shd[0];
            int tlev = shi[1], tsym = shi[2];
            for (int c = 0; c < A; ++c) {
                const double lp = cand[((size_t)0 * PO_A + c) * tc + U - 1] + cand[((size_t)1 * PO_A + c) * tc + V - 1] - g00;
                if (lp > top) { top = lp; tlev = level; tsym = c; }
            }
            int best = 0;
            for (int c = 1; c < A; ++c) if (shd[4 + c] > shd[4 + best]) best = c;
            if (shd[4 + best] < shd[0]) shi[3] = 1;  // compared with the top label BEFORE this level's labels
            else {
                shd[0] = top; shi[1] = tlev; shi[2] = tsym;
                curr[curr_len] = (char)((a.alphabet >> (8 * best)) & 0xffu);
                shi[0] = curr_len + 1;
                shi[4] = best;
                if (depth_stop) shi[3] = 1;
            }
        }
        __syncthreads();
        if (shi[3]) {
            // the rows are copied before the reference leaves the loop on depth_stop; nothing reads them after
            break;
        }
        const int best = shi[4];
        for (int t = tid; t < U; t += PS_THREADS) prev1[t] = cand[((size_t)0 * PO_A + best) * tc + t];
        for (int t = tid; t < V; t += PS_THREADS) prev2[t] = cand[((size_t)1 * PO_A + best) * tc + t];
        __syncthreads();
        if (level > M + 8) { st = PO_E_DIVERGE; break; }
    }
    __syncthreads();
    if (tid == 0) {
        int nout = 0;
        if (st == PO_OK) {
            nout = shi[1];  // 0 for the empty label; label = curr[:level-1] + symbol
            char* out = a.seq + a.seq_off[pi];
            const int cap = (int)(a.seq_off[pi + 1] - a.seq_off[pi]);
            if (nout > cap) { st = PO_E_CAP; nout = 0; }
            else if (nout > 0) {
                for (int i = 0; i < nout - 1; ++i) out[i] = curr[i];
                out[nout - 1] = (char)((a.alphabet >> (8 * shi[2])) & 0xffu);
            }
        }
        a.seq_len[pi] = nout;
        a.logp[pi] = shd[0];
        a.status[pi] = st;
    }
}

extern "C" size_t po_pair_prefix_ws_bytes(int n, int64_t max_rows) { return al256((size_t)max_rows + 16) * (size_t)(n > 0 ? n : 1) + 256; }

extern "C" int po_launch_pair_prefix_search(const double* y1, const int64_t* y1_off, const double* y2, const int64_t* y2_off,
                                            const double* gm, const int64_t* gm_off, int n, int C, int A, uint32_t alphabet,
                                            int flavor, int64_t max_rows, char* seq, const int64_t* seq_off,
                                            int32_t* seq_len, double* logp, int32_t* status, void* ws, size_t ws_bytes,
                                            hipStream_t stream) {
    if (n <= 0) return PO_OK;
    if (A < 1 || A > PO_A || C != A + 1) return PO_E_ARG;
    const size_t lds = sizeof(double) * (size_t)(4 + 2 * PO_A) * (size_t)max_rows;
    if (lds > 150 * 1024) return PO_E_UNSUPPORTED;  // boxes longer than ~1600 frames per read do not fit the LDS rows
    const size_t per = al256((size_t)max_rows + 16);
    if (ws_bytes < per * n) return PO_E_CAP;
    PPSArgs a;
    a.y1 = y1; a.y1_off = y1_off; a.y2 = y2; a.y2_off = y2_off; a.gm = gm; a.gm_off = gm_off;
    a.n = n; a.C = C; a.A = A; a.flavor = flavor; a.alphabet = alphabet;
    a.seq = seq; a.seq_off = seq_off; a.seq_len = seq_len; a.logp = logp; a.status = status;
    a.curr = (char*)ws; a.curr_cap = (long long)per; a.tcap = (int)max_rows;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)pair_prefix_search_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(pair_prefix_search_kernel, dim3(n), dim3(PS_THREADS), lds, stream, a);
    return PO_OK;
}

// ================================================================================================
// forward_vec_log (decoding_cy.pyx:127-156; prefix_search.py:81-96): one row of the CTC forward matrix —
// the log-probability, for every frame t, of having emitted a label whose last symbol is s by frame t, given
// the row `previous` of the label without that symbol.  i = length of the label (0: the empty label, whose row
// is the running sum of column s; 1: the row starts at y[0][s]).  A serial recurrence in t; one lane per item.
namespace {
struct FVArgs {
    const double* y; const int64_t* y_off; int n, C, s, i, flavor;
    const double* previous;   // same offsets as the output; NULL only for i == 0
    double* out;
};
}  // namespace

__global__ __launch_bounds__(64) void forward_vec_kernel(FVArgs a) {
    const int pi = blockIdx.x * 64 + threadIdx.x;
    if (pi >= a.n) return;
    const int64_t r0 = a.y_off[pi] - a.y_off[0];
    const int T = (int)(a.y_off[pi + 1] - a.y_off[pi]);
    const int C = a.C, sc = a.s < 0 ? C + a.s : a.s;   // s == -1 selects the blank column (python negative index)
    const double* y = a.y + a.y_off[pi] * C;
    const double* pv = a.previous ? a.previous + r0 : nullptr;
    double* fw = a.out + r0;
    const double LOG0 = a.flavor ? -9999.0 : PO_NEG_INF;
    double cur = LOG0;
    for (int t = 0; t < T; ++t) {
        const double* r = y + (int64_t)t * C;
        if (a.i == 0) cur = (t == 0) ? r[sc] : r[C - 1] + cur;
        else if (t == 0) cur = (a.i == 1) ? r[sc] : LOG0;
        else cur = pps_lae(r[C - 1] + cur, r[sc] + pv[t - 1], a.flavor);
        fw[t] = cur;
    }
}

extern "C" int po_launch_forward_vec(const double* y, const int64_t* y_off, int n, int C, int s, int i, int flavor,
                                     const double* previous, double* out, hipStream_t stream) {
    if (n <= 0) return PO_OK;
    if (C < 2 || s < -1 || s >= C || i < 0 || (i != 0 && !previous)) return PO_E_ARG;
    FVArgs a;
    a.y = y; a.y_off = y_off; a.n = n; a.C = C; a.s = s; a.i = i; a.flavor = flavor; a.previous = previous; a.out = out;
    hipLaunchKernelGGL(forward_vec_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, a);
    return PO_OK;
}
