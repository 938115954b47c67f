// Label lattices: log P(label | y) and the best alignment path of a known label.
//
// Replaces
//   decoding_cpp.cpp_forward          (decoding_cpp.pyx:49-65)  -> forward / forward_ (PrefixTree.h:710-759):
//       a chain of one prefix-tree node per label position, each updated for t = 0..T-1 with the
//       model's update_prob — i.e. the CTC forward lattice alpha[s][t] for the three tree models.
//   decoding_cpp.cpp_viterbi_acceptor (decoding_cpp.pyx:69-84)  -> viterbi_acceptor_poreover
//       (Forward.h:14-121): banded Viterbi alignment (emit vs stay, ties to emit) with traceback.
//       Reproduced as written: the value / pointer matrices are SparseMatrix rows with inclusive
//       ranges whose out-of-range writes are dropped and reads default (-inf / 0); rows 0 and 1
//       are pushed up front as [0, band], and the row pushed while processing label position l
//       lands at index l + 1 (so row l carries the band computed for l - 1).
//
// Mapping: one workgroup per item, lane = label position (chunks of 256 positions), all lanes step
// through time together; a lane takes its left neighbour's value of the previous frame from a
// double-buffered LDS array (one LDS-only barrier per frame).  The last lane of a chunk writes its
// row to HBM for the first lane of the next chunk.  Dependent chain of T frames: latency-bound.
#include <algorithm>
#include <cstdlib>

#include "po_device.h"

namespace {
constexpr int LT_THREADS = 256;

struct LTArgs {
    const double* y; const int64_t* y_off; int n, C, A;
    const char* labels; const int64_t* label_off;  // concatenated label characters
    uint32_t alphabet;
    int band;                 // acceptor only
    double* out;              // forward: log-probability per item
    int32_t* path;            // acceptor: int32[total_rows], at y_off[i]
    int32_t* status;
    double* rows; long long row_cap;   // per workgroup: 2 x row_cap x K doubles (chunk hand-over rows)
    int8_t* ptr; long long ptr_cap;    // acceptor, per workgroup: (L + 1) x T pointer bytes
};

__device__ __forceinline__ int sym_index(uint32_t alphabet, int A, char c) {
    for (int i = 0; i < A; ++i)
        if ((char)((alphabet >> (8 * i)) & 0xffu) == c) return i;
    return 0;  // std::unordered_map default-insert (PrefixTree.h:724, Forward.h:31): unknown -> 0
}
}  // namespace

template <int MODEL>
__global__ __launch_bounds__(LT_THREADS) void forward_kernel(LTArgs a) {
    constexpr int K = (MODEL == PO_MODEL_CTC) ? 1 : 3;
    __shared__ double xch[2][LT_THREADS][K];
    __shared__ PoLaeTables lae_t;
    const int tid = threadIdx.x, pi = blockIdx.x;
    po_lae_tables_load(&lae_t, tid, LT_THREADS);
    const PoLaeFast lae{&lae_t};
    __syncthreads();
    const int A = a.A, C = a.C;
    const int64_t r0 = a.y_off[pi];
    const int T = (int)(a.y_off[pi + 1] - r0);
    const double* y = a.y + r0 * C;
    const char* lab = a.labels + a.label_off[pi];
    const int L = (int)(a.label_off[pi + 1] - a.label_off[pi]);
    double* rowA = a.rows + (size_t)blockIdx.x * 2 * a.row_cap * K;
    double* rowB = rowA + a.row_cap * K;
    if (T < 1 || T > a.row_cap) {
        if (tid == 0) { a.out[pi] = __builtin_nan(""); a.status[pi] = PO_E_ARG; }
        return;
    }
    if (L == 0) {  // forward_ returns root->last_probability() == probability.at(0) (PrefixTree.h:749)
        if (tid == 0) {
            a.out[pi] = (MODEL == PO_MODEL_CTC) ? y[A] : __builtin_nan("");
            a.status[pi] = (MODEL == PO_MODEL_CTC) ? PO_OK : PO_E_ARG;
        }
        return;
    }
    double result = 0.0;
    double blank_cum = 0.0;  // CTC root: alpha[t] = sum of blanks up to t (lane 0 of chunk 0 tracks it)
    for (int c0 = 0; c0 < L; c0 += LT_THREADS) {
        const int nl = min(LT_THREADS, L - c0);
        const bool on = tid < nl;
        const int p = c0 + tid;
        const int sym = on ? sym_index(a.alphabet, A, lab[p]) : 0;
        const int psym = (on && p > 0) ? sym_index(a.alphabet, A, lab[p - 1]) : A;
        const bool same = (psym == sym), rootpar = (p == 0);
        const double* prow = ((c0 / LT_THREADS) & 1) ? rowA : rowB;   // written by the previous chunk
        double* nrow = ((c0 / LT_THREADS) & 1) ? rowB : rowA;
        double self[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { self[k] = PO_NEG_INF; xch[1][tid][k] = PO_NEG_INF; }
        blank_cum = 0.0;
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            double out[K];
            if (on) {
                double pp[K];
                if (tid > 0) {
#pragma unroll
                    for (int k = 0; k < K; ++k) pp[k] = xch[(t + 1) & 1][tid - 1][k];
                } else if (c0 == 0) {
                    double tmp[3];
                    root_values<MODEL>(t - 1, blank_cum, tmp);
#pragma unroll
                    for (int k = 0; k < K; ++k) pp[k] = tmp[k];
                } else {
#pragma unroll
                    for (int k = 0; k < K; ++k) pp[k] = (t > 0) ? prow[(size_t)(t - 1) * K + k] : PO_NEG_INF;
                }
                const double ya = y[(int64_t)t * C + sym];
                const double yb = (MODEL == PO_MODEL_FLIPFLOP) ? y[(int64_t)t * C + sym + A] : y[(int64_t)t * C + A];
                po_update<MODEL>(self, pp, ya, yb, same, rootpar && t == 0, out, lae);
#pragma unroll
                for (int k = 0; k < K; ++k) { self[k] = out[k]; xch[t & 1][tid][k] = out[k]; }
                if (tid == nl - 1) {
#pragma unroll
                    for (int k = 0; k < K; ++k) nrow[(size_t)t * K + k] = out[k];
                }
                if (MODEL == PO_MODEL_CTC && tid == 0 && c0 == 0) blank_cum += y[(int64_t)t * C + A];
            }
            po_lds_barrier();
        }
        if (on && tid == nl - 1) result = self[0];
        __syncthreads();  // the hand-over row is read by the next chunk
        if (c0 + LT_THREADS >= L && on && tid == nl - 1) { a.out[pi] = result; a.status[pi] = PO_OK; }
    }
}

__global__ __launch_bounds__(LT_THREADS) void acceptor_kernel(LTArgs a) {
    __shared__ double xch[2][LT_THREADS];
    __shared__ int sh[2];
    const int tid = threadIdx.x, pi = blockIdx.x;
    const int A = a.A, C = a.C, gap = a.A, band = a.band;
    const int64_t r0 = a.y_off[pi];
    const int T = (int)(a.y_off[pi + 1] - r0);
    const double* y = a.y + r0 * C;
    const char* lab = a.labels + a.label_off[pi];
    const int L = (int)(a.label_off[pi + 1] - a.label_off[pi]);
    int32_t* path = a.path + r0;
    double* rowA = a.rows + (size_t)blockIdx.x * 2 * a.row_cap;
    double* rowB = rowA + a.row_cap;
    int8_t* ptr = a.ptr + (size_t)blockIdx.x * a.ptr_cap;
    if (T < 1 || L < 1 || T > a.row_cap || (long long)(L + 1) * T > a.ptr_cap || band < 0) {
        if (tid == 0) a.status[pi] = (T < 1 || L < 1 || band < 0) ? PO_E_ARG : PO_E_CAP;
        return;
    }
    for (int64_t i = tid; i < (int64_t)(L + 1) * T; i += LT_THREADS) ptr[i] = 0;  // SparseMatrix<int> ptr(0)
    for (int t = tid; t < T; t += LT_THREADS) path[t] = gap;
    __syncthreads();
    // centre / band of label position l (Forward.h:59-60)
    auto rs_of = [&](int l) { return max(1, (int)((double)l * (double)T / (double)L) - band); };
    auto re_of = [&](int l) { return min(T, (int)((double)l * (double)T / (double)L) + band); };
    for (int c0 = 0; c0 < L; c0 += LT_THREADS) {
        const int nl = min(LT_THREADS, L - c0);
        const bool on = tid < nl;
        const int l = c0 + tid + 1;  // this lane's row (1-based label position)
        const int sym = on ? sym_index(a.alphabet, A, lab[l - 1]) : 0;
        const int rs = rs_of(l), re = re_of(l);
        // storage range of row index l (inclusive): row 1 = [0, band]; row l >= 2 = band of l - 1
        const int S = (l == 1) ? 0 : rs_of(l - 1), E = (l == 1) ? band : re_of(l - 1);
        const double* prow = ((c0 / LT_THREADS) & 1) ? rowA : rowB;
        double* nrow = ((c0 / LT_THREADS) & 1) ? rowB : rowA;
        double cur = PO_NEG_INF;  // stored v(l, t-1), -inf if that cell was never stored
        double cum = 0.0;         // row 0: running blank sum (lane of l == 1 only)
        xch[1][tid] = PO_NEG_INF;
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            double stored = PO_NEG_INF;  // v(l, t) as the matrix holds it after this frame
            if (on) {
                // v.get(l - 1, t - 1)
                double left;
                if (tid > 0) left = xch[(t + 1) & 1][tid - 1];
                else if (l == 1) left = (t - 1 >= 0 && t - 1 <= band) ? cum : PO_NEG_INF;  // row 0 = [0, band]
                else left = (t > 0) ? prow[t - 1] : PO_NEG_INF;
                if (l == 1 && t == 0 && 0 >= S && 0 <= E) {  // v.set(1, 0, y[0][label[0]]); ptr.set(1, 0, 1)
                    stored = y[sym];
                    ptr[(size_t)1 * T + 0] = 1;
                }
                if (t >= rs && t < re && t >= l - 1) {
                    const double emit = y[(int64_t)t * C + sym] + left;
                    const double stay = y[(int64_t)t * C + gap] + cur;
                    const bool take_emit = (emit >= stay);
                    if (t >= S && t <= E) {  // SparseRow::set drops out-of-range writes
                        stored = take_emit ? emit : stay;
                        ptr[(size_t)l * T + t] = take_emit ? 1 : 0;
                    }
                }
                cur = stored;
                xch[t & 1][tid] = stored;
                if (tid == nl - 1) nrow[t] = stored;
                if (l == 1) cum += y[(int64_t)t * C + gap];  // becomes v(0, t) for the next frame
            }
            po_lds_barrier();
        }
        __syncthreads();
    }
    __syncthreads();
    // traceback (Forward.h:100-115)
    if (tid == 0) {
        int l = L, t = T - 1, st = PO_OK;
        while (l > 0) {
            if (t < 0) { st = PO_E_DIVERGE; break; }  // the reference never terminates here
            if (ptr[(size_t)l * T + t] > 0) {
                path[t] = sym_index(a.alphabet, A, lab[l - 1]);
                l -= 1;
            }
            t -= 1;
        }
        a.status[pi] = st;
    }
}

// decoding_cy.viterbi_acceptor (decoding_cy.pyx:60-123), the Cython twin of the acceptor above, as it is written:
// dense matrices, row 0 = running blank sum, '>' (not '>=') between emit and stay, cells only for t >= l, pointer
// matrix initialised to the blank index (non-zero: a cell that was never computed reads as "emit" in the
// trace-back), trace-back with Python's negative-index wrap-around, and the band centre (l / L) * t that reuses
// the loop variable t of the previous loop (integer division: 0 for l < L, the last t of row L - 1 for l = L).
__global__ __launch_bounds__(LT_THREADS) void acceptor_cy_kernel(LTArgs a) {
    __shared__ double xch[2][LT_THREADS];
    const int tid = threadIdx.x, pi = blockIdx.x;
    const int A = a.A, C = a.C, gap = a.A;
    const int64_t r0 = a.y_off[pi];
    const int T = (int)(a.y_off[pi + 1] - r0);
    const double* y = a.y + r0 * C;
    const char* lab = a.labels + a.label_off[pi];
    const int L = (int)(a.label_off[pi + 1] - a.label_off[pi]);
    int32_t* path = a.path + r0;
    double* rowA = a.rows + (size_t)blockIdx.x * 2 * a.row_cap;
    double* rowB = rowA + a.row_cap;
    int8_t* ptr = a.ptr + (size_t)blockIdx.x * a.ptr_cap;
    if (T < 1 || L < 1 || T > a.row_cap || (long long)(L + 1) * T > a.ptr_cap) {
        if (tid == 0) a.status[pi] = (T < 1 || L < 1) ? PO_E_ARG : PO_E_CAP;
        return;
    }
    int badsym = 0;  // a label character outside the alphabet is a KeyError upstream
    for (int i = tid; i < L; i += LT_THREADS) {
        bool found = false;
        for (int k = 0; k < A; ++k) found |= ((char)((a.alphabet >> (8 * k)) & 0xffu) == lab[i]);
        badsym |= !found;
    }
    if (__syncthreads_or(badsym)) {
        if (tid == 0) a.status[pi] = PO_E_ARG;
        return;
    }
    for (int64_t i = tid; i < (int64_t)(L + 1) * T; i += LT_THREADS) ptr[i] = (int8_t)gap;
    for (int t = tid; t < T; t += LT_THREADS) path[t] = gap;
    __syncthreads();
    if (tid == 0) ptr[0] = 1;
    const int band_ = a.band > 0 ? a.band : T;
    const int hcommon = min(T, band_);                                   // rows l < L: [1, hcommon)
    const int tlast = (L > 1 && hcommon > 1) ? hcommon - 1 : T - 1;      // the loop variable when row L starts
    for (int c0 = 0; c0 < L; c0 += LT_THREADS) {
        const int nl = min(LT_THREADS, L - c0);
        const bool on = tid < nl;
        const int l = c0 + tid + 1;
        const int sym = on ? sym_index(a.alphabet, A, lab[l - 1]) : 0;
        const int lo = (l < L) ? 1 : max(1, tlast - band_);
        const int hi = (l < L) ? hcommon : min(T, tlast + band_);
        const double* prow = ((c0 / LT_THREADS) & 1) ? rowA : rowB;
        double* nrow = ((c0 / LT_THREADS) & 1) ? rowB : rowA;
        double cur = PO_NEG_INF;  // v(l, t-1)
        double cum = 0.0;         // row 0 (lane of l == 1): running blank sum
        xch[1][tid] = PO_NEG_INF;
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            double val = PO_NEG_INF;
            if (on) {
                double left;  // v(l-1, t-1)
                if (tid > 0) left = xch[(t + 1) & 1][tid - 1];
                else if (l == 1) left = (t >= 1) ? cum : PO_NEG_INF;
                else left = (t >= 1) ? prow[t - 1] : PO_NEG_INF;
                if (l == 1 && t == 0) val = y[sym];                      // v[1, 0] = y[0, label[0]]
                if (t >= lo && t < hi && t >= l) {
                    const double emit = y[(int64_t)t * C + sym] + left;
                    const double stay = y[(int64_t)t * C + gap] + cur;
                    const bool take_emit = (emit > stay);
                    val = take_emit ? emit : stay;
                    ptr[(size_t)l * T + t] = take_emit ? 1 : 0;
                }
                cur = val;
                xch[t & 1][tid] = val;
                if (tid == nl - 1) nrow[t] = val;
                if (l == 1) cum += y[(int64_t)t * C + gap];
            }
            po_lds_barrier();
        }
        __syncthreads();
    }
    __syncthreads();
    if (tid == 0) {
        int l = L, st = PO_OK;
        long long tt = T - 1;
        while (l > 0) {
            if (tt < -(long long)T) { st = PO_E_DIVERGE; break; }   // IndexError upstream
            const long long idx = tt < 0 ? tt + T : tt;              // wraparound
            if (ptr[(size_t)l * T + idx] != 0) {
                path[idx] = sym_index(a.alphabet, A, lab[l - 1]);
                l -= 1;
            }
            tt -= 1;
        }
        a.status[pi] = st;
    }
}

// ------------------------------------------------------------------------------------------------
namespace {
inline size_t al256(size_t b) { return (b + 255) & ~size_t(255); }
}

extern "C" size_t po_lattice_ws_bytes(int n, int64_t max_rows, int64_t max_label, int model, int acceptor) {
    const int K = (model == PO_MODEL_CTC) ? 1 : 3;
    size_t b = al256(sizeof(double) * 2 * (size_t)max_rows * K) * (size_t)(n > 0 ? n : 1);
    if (acceptor) b += al256((size_t)(max_label + 1) * (size_t)max_rows) * (size_t)(n > 0 ? n : 1);
    return b + 512;
}

extern "C" int po_launch_forward(const double* y, const int64_t* y_off, int n, int C, int A, uint32_t alphabet,
                                 int model, const char* labels, const int64_t* label_off, int64_t max_rows,
                                 double* out, int32_t* status, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (n <= 0) return PO_OK;
    if (A < 1 || A > PO_A) return PO_E_ARG;
    if ((model == PO_MODEL_FLIPFLOP) ? (C != 2 * A) : (C != A + 1)) return PO_E_ARG;
    const int K = (model == PO_MODEL_CTC) ? 1 : 3;
    const size_t per = al256(sizeof(double) * 2 * (size_t)max_rows * K);
    if (ws_bytes < per * n) return PO_E_CAP;
    LTArgs a = {};
    a.y = y; a.y_off = y_off; a.n = n; a.C = C; a.A = A; a.labels = labels; a.label_off = label_off;
    a.alphabet = alphabet; a.out = out; a.status = status;
    a.rows = (double*)ws; a.row_cap = (long long)(per / (sizeof(double) * 2 * K));
    if (model == PO_MODEL_CTC) hipLaunchKernelGGL(forward_kernel<PO_MODEL_CTC>, dim3(n), dim3(LT_THREADS), 0, stream, a);
    else if (model == PO_MODEL_MERGE) hipLaunchKernelGGL(forward_kernel<PO_MODEL_MERGE>, dim3(n), dim3(LT_THREADS), 0, stream, a);
    else if (model == PO_MODEL_FLIPFLOP) hipLaunchKernelGGL(forward_kernel<PO_MODEL_FLIPFLOP>, dim3(n), dim3(LT_THREADS), 0, stream, a);
    else return PO_E_ARG;
    return PO_OK;
}

extern "C" int po_launch_acceptor(const double* y, const int64_t* y_off, int n, int C, int A, uint32_t alphabet,
                                  int band, const char* labels, const int64_t* label_off, int64_t max_rows,
                                  int64_t max_label, int32_t* path, int32_t* status, void* ws, size_t ws_bytes,
                                  hipStream_t stream) {
    if (n <= 0) return PO_OK;
    if (A < 1 || A > PO_A || C != A + 1) return PO_E_ARG;
    const size_t per_rows = al256(sizeof(double) * 2 * (size_t)max_rows);
    const size_t per_ptr = al256((size_t)(max_label + 1) * (size_t)max_rows);
    if (ws_bytes < (per_rows + per_ptr) * n) return PO_E_CAP;
    LTArgs a = {};
    a.y = y; a.y_off = y_off; a.n = n; a.C = C; a.A = A; a.labels = labels; a.label_off = label_off;
    a.alphabet = alphabet; a.band = band; a.path = path; a.status = status;
    a.rows = (double*)ws; a.row_cap = (long long)(per_rows / (sizeof(double) * 2));
    a.ptr = (int8_t*)ws + per_rows * n; a.ptr_cap = (long long)per_ptr;
    if (band < 0) {   // negative band: the Cython twin, band = -band - 1 (0: the whole matrix)
        a.band = -band - 1;
        hipLaunchKernelGGL(acceptor_cy_kernel, dim3(n), dim3(LT_THREADS), 0, stream, a);
    } else {
        hipLaunchKernelGGL(acceptor_kernel, dim3(n), dim3(LT_THREADS), 0, stream, a);
    }
    return PO_OK;
}
