// Pair gamma DP: log-probability that two reads emit the same label, as a backward 2-D dynamic
// program over an alignment envelope, walked in anti-diagonals.
//
// Replaces
//   decoding_cpp.cpp_pair_gamma_log_envelope (decoding_cpp.pyx:168-188) -> pair_gamma_log_envelope (Gamma.h:15-98):
//       SparseMatrix rows 0..U with INCLUSIVE column ranges (SparseMatrix.h:35-57), default -inf, writes
//       outside a row dropped; cells [start, end-1] of rows u < U are computed (Gamma.h:61-64)
//   decoding_cy.pair_gamma_log (decoding_cy.pyx:177-220): the dense (U+1) x (V+1) matrix with LOG_0 = -9999
//       and log(exp(a) + exp(b)) instead of logaddexp   (flavor 1)
//
//   gamma(U,V) = gamma*(U,V) = 0; gamma(U,v) = sum_{v'>=v} y2[v'][blank]; gamma(u,V) = sum_{u'>=u} y1[u'][blank]
//   gamma*(u,v) = lae(gamma*(u,v+1) + y2[v][blank], gamma(u+1,v+1) + log sum_c exp(y1[u][c] + y2[v][c]))
//   gamma(u,v)  = lae(gamma(u+1,v) + y1[u][blank], gamma*(u,v))
//
// Cell (u,v) needs (u+1,v), (u,v+1) [diagonal d+1] and (u+1,v+1) [d+2], d = u + v: all cells of one
// anti-diagonal are independent.  One workgroup per pair walks d downwards; each thread tests its
// strided rows for membership (start_u <= d-u <= end_u - 1), so nothing is assumed about the shape
// of the envelope.  The two band matrices live in HBM (L2-resident: ~1.4 MB per T~4000 pair).
#include <algorithm>

#include "po_device.h"

namespace {
constexpr int GM_THREADS = 256;
struct GMArgs {
    const double* y1; const int64_t* y1_off; const double* y2; const int64_t* y2_off;
    const int32_t* env; const int64_t* env_off;   // (U_i + 1) inclusive rows per pair; NULL env_off: dense
    int n, C, flavor;
    double* out;            // gamma(0,0) per pair
    double* dense_out; const int64_t* dense_off;  // optional: full (U+1) x (V+1) gamma matrices
    int32_t* status;
    double* mat; long long mat_cap;      // per workgroup: 2 band matrices of mat_cap doubles
    long long* roff; long long row_cap;  // per workgroup: row offsets (U + 2)
    double* suf; long long suf_cap;      // per workgroup: suffix sums of blanks of both reads
};
}  // namespace

__global__ __launch_bounds__(GM_THREADS) void pair_gamma_kernel(GMArgs a) {
    __shared__ int sh[4];
    const int tid = threadIdx.x, pi = blockIdx.x;
    const int C = a.C, b = a.C - 1;
    const int64_t o1 = a.y1_off[pi], o2 = a.y2_off[pi];
    const int U = (int)(a.y1_off[pi + 1] - o1), V = (int)(a.y2_off[pi + 1] - o2);
    const double* y1 = a.y1 + o1 * C;
    const double* y2 = a.y2 + o2 * C;
    const bool dense = (a.env == nullptr);
    const int32_t* env = dense ? nullptr : a.env + 2 * a.env_off[pi];
    // flavor 0: Gamma.h; 1: decoding_cy.pair_gamma_log (dense, LOG_0 = -9999); 2: decoding_cy.pair_gamma_log_envelope
    // (decoding_cy.pyx:224-271: log(exp + exp), -inf defaults, every envelope cell with u < U and v < V is computed)
    const double LOG0 = (a.flavor == 1) ? -9999.0 : PO_NEG_INF;
    double* g = a.mat + (size_t)blockIdx.x * 2 * a.mat_cap;
    double* ga = g + a.mat_cap;
    long long* roff = a.roff + (size_t)blockIdx.x * a.row_cap;
    double* suf1 = a.suf + (size_t)blockIdx.x * 2 * a.suf_cap;
    double* suf2 = suf1 + a.suf_cap;
    auto rs = [&](int u) { return dense ? 0 : env[2 * u]; };
    auto re = [&](int u) { return dense ? V : env[2 * u + 1]; };   // inclusive end
    if (U < 1 || V < 1 || U + 2 > a.row_cap || max(U, V) + 1 > a.suf_cap) {
        if (tid == 0) { a.out[pi] = __builtin_nan(""); a.status[pi] = (U < 1 || V < 1) ? PO_E_ARG : PO_E_CAP; }
        return;
    }
    if (tid == 0) {
        long long acc = 0;
        for (int u = 0; u <= U; ++u) { roff[u] = acc; const int w = re(u) - rs(u) + 1; acc += (w > 0 ? w : 0); }
        roff[U + 1] = acc;
        sh[0] = (acc > a.mat_cap) ? 1 : 0;
        // suffix sums of the blank columns, in the reference's summation order (Gamma.h:39-55: from the cell forwards)
        // sum_{k>=v} computed forwards for each v would be O(T^2); the reference's order for entry v is
        // y[v] + y[v+1] + ... (left to right).  Reproduce exactly with a per-entry loop only for small T;
        // for large T use the backward recurrence (agrees to rounding).  Tests pin the small case exactly.
    }
    __syncthreads();
    if (sh[0]) {
        if (tid == 0) { a.out[pi] = __builtin_nan(""); a.status[pi] = PO_E_CAP; }
        return;
    }
    const long long ncell = roff[U + 1];
    for (long long i = tid; i < ncell; i += GM_THREADS) { g[i] = LOG0; ga[i] = LOG0; }
    // boundary sums: entry v = y[v] + y[v+1] + ... + y[T-1] summed left to right, as upstream
    const bool exact_sums = ((long long)U * U + (long long)V * V) <= (1 << 22);
    if (exact_sums) {
        for (int v = tid; v < V; v += GM_THREADS) { double s = 0.; for (int k = v; k < V; ++k) s += y2[(int64_t)k * C + b]; suf2[v] = s; }
        for (int u = tid; u < U; u += GM_THREADS) { double s = 0.; for (int k = u; k < U; ++k) s += y1[(int64_t)k * C + b]; suf1[u] = s; }
    } else if (tid < 2) {
        const double* yy = tid ? y2 : y1; double* sf = tid ? suf2 : suf1; const int Tn = tid ? V : U;
        double s = 0.;
        for (int k = Tn - 1; k >= 0; --k) { s += yy[(int64_t)k * C + b]; sf[k] = s; }
    }
    __syncthreads();
    auto in = [&](int u, int v) { return u >= 0 && u <= U && v >= rs(u) && v <= re(u); };
    auto G = [&](const double* m, int u, int v) { return in(u, v) ? m[roff[u] + (v - rs(u))] : (dense ? LOG0 : PO_NEG_INF); };
    // boundary cells (dropped when outside the stored ranges, SparseMatrix::set)
    for (int v = tid; v <= V; v += GM_THREADS)
        if (in(U, v)) g[roff[U] + (v - rs(U))] = (v == V) ? 0.0 : suf2[v];
    for (int u = tid; u < U; u += GM_THREADS)
        if (in(u, V)) g[roff[u] + (V - rs(u))] = suf1[u];
    if (tid == 0 && in(U, V)) ga[roff[U] + (V - rs(U))] = 0.0;
    __syncthreads();
    for (int d = U + V - 2; d >= 0; --d) {
        for (int u = min(U - 1, d) - tid; u >= 0 && u >= d - (V - 1); u -= GM_THREADS) {
            const int v = d - u;
            if (v < rs(u) || v > ((a.flavor == 2) ? min(re(u), V - 1) : re(u) - 1)) continue;   // cells [start, end-1] only (Gamma.h:61-64)
            const double* r1 = y1 + (int64_t)u * C;
            const double* r2 = y2 + (int64_t)v * C;
            const double gamma_eps = G(g, u + 1, v) + r1[b];
            const double gamma_ast_eps = G(ga, u, v + 1) + r2[b];
            double total2 = 0.;
            for (int t = 0; t < C - 1; ++t) total2 += exp(r1[t] + r2[t]);
            const double gamma_ast_ast = G(g, u + 1, v + 1) + log(total2);
            double x_ast, x;
            if (a.flavor) {
                x_ast = log(exp(gamma_ast_eps) + exp(gamma_ast_ast));
                x = log(exp(gamma_eps) + exp(x_ast));
            } else {
                x_ast = po_lae(gamma_ast_eps, gamma_ast_ast);
                x = po_lae(gamma_eps, x_ast);
            }
            ga[roff[u] + (v - rs(u))] = x_ast;
            g[roff[u] + (v - rs(u))] = x;
        }
        __syncthreads();
    }
    if (tid == 0) { a.out[pi] = G(g, 0, 0); a.status[pi] = PO_OK; }
    if (a.dense_out && dense) {
        double* dst = a.dense_out + a.dense_off[pi];
        for (long long i = tid; i < (long long)(U + 1) * (V + 1); i += GM_THREADS) dst[i] = g[i];
    } else if (a.dense_out) {
        // envelope: the (U+1) x (V+1) matrix as SparseMatrix::get shows it — -inf outside the stored ranges
        // (input of the pair prefix search with an envelope, PairPrefixSearch.cpp:79-229)
        double* dst = a.dense_out + a.dense_off[pi];
        for (long long i = tid; i < (long long)(U + 1) * (V + 1); i += GM_THREADS) {
            const int u = (int)(i / (V + 1)), v = (int)(i - (long long)u * (V + 1));
            dst[i] = G(g, u, v);
        }
    }
}

namespace {
inline size_t al256(size_t b) { return (b + 255) & ~size_t(255); }
}

extern "C" size_t po_gamma_ws_bytes(int n, int64_t max_cells, int64_t max_rows1, int64_t max_rows2) {
    const size_t nn = (size_t)(n > 0 ? n : 1);
    return al256(sizeof(double) * 2 * (size_t)max_cells) * nn + al256(sizeof(long long) * (size_t)(max_rows1 + 3)) * nn +
           al256(sizeof(double) * 2 * (size_t)(std::max(max_rows1, max_rows2) + 2)) * nn + 256;
}

extern "C" int po_launch_gamma(const double* y1, const int64_t* y1_off, const double* y2, const int64_t* y2_off,
                               const int32_t* env, const int64_t* env_off, int n, int C, int flavor, int64_t max_cells,
                               int64_t max_rows1, int64_t max_rows2, double* out, double* dense_out,
                               const int64_t* dense_off, int32_t* status, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (n <= 0) return PO_OK;
    if (C < 2 || C > 9) return PO_E_ARG;
    if (ws_bytes < po_gamma_ws_bytes(n, max_cells, max_rows1, max_rows2)) return PO_E_CAP;
    char* w = (char*)ws;
    GMArgs a;
    a.y1 = y1; a.y1_off = y1_off; a.y2 = y2; a.y2_off = y2_off; a.env = env; a.env_off = env_off;
    a.n = n; a.C = C; a.flavor = flavor; a.out = out; a.dense_out = dense_out; a.dense_off = dense_off; a.status = status;
    size_t o = 0;
    a.mat = (double*)(w + o); a.mat_cap = (long long)max_cells; o += al256(sizeof(double) * 2 * (size_t)max_cells) * n;
    a.roff = (long long*)(w + o); a.row_cap = (long long)(max_rows1 + 3); o += al256(sizeof(long long) * (size_t)(max_rows1 + 3)) * n;
    a.suf = (double*)(w + o); a.suf_cap = (long long)(std::max(max_rows1, max_rows2) + 2);
    // per-workgroup strides must match the carve above
    a.mat_cap = (long long)(al256(sizeof(double) * 2 * (size_t)max_cells) / (2 * sizeof(double)));
    a.row_cap = (long long)(al256(sizeof(long long) * (size_t)(max_rows1 + 3)) / sizeof(long long));
    a.suf_cap = (long long)(al256(sizeof(double) * 2 * (size_t)(std::max(max_rows1, max_rows2) + 2)) / (2 * sizeof(double)));
    hipLaunchKernelGGL(pair_gamma_kernel, dim3(n), dim3(GM_THREADS), 0, stream, a);
    return PO_OK;
}
