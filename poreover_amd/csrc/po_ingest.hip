// Trace ingest on the device: basecaller output -> (T, C) float64 log-probabilities.
//
// Replaces, from decoding/decode.py and decoding/transducer.py of the reference:
//   logit_to_log_likelihood (decode.py:34-39): x - logsumexp(x) per frame.  The reference does this in
//       the array's own precision (float32 logits stay float32, scipy.special.logsumexp) and only
//       then widens to float64 (transducer.py:16); so does this kernel, with the arithmetic of the scipy
//       version the reference runs on in this image (1.15.3; the test suite pins it on the reference's sample reads).
//   the uint8 flip-flop trace scaling (decode.py:89-93,99-103): log((x + 1e-7) / (255 + 1e-7)) in float64
//   the Bonito column permutation [1,2,3,4,0] (decode.py:79) and reverse_complement
//       (transducer.py:68-70,104-106): time reversal + column permutation — folded into the same pass.
//
// Pure streaming: 4 (or 1) bytes in, 8 bytes out per value; one thread per frame, a frame's C values
// are contiguous on both sides, so consecutive lanes touch consecutive 20/40-byte runs.  HBM-bound.
#include <algorithm>

#include "po_device.h"

namespace {
struct IGArgs {
    const void* src; const int64_t* row_off; int n, C, mode, reverse;
    int perm[8];
    double* out; int64_t total_rows;
};
}

// mode 0: float32 logits -> log-softmax; 1: uint8 trace -> log prob; 2: float64 values, copy
// (one-wave workgroups: they fit wherever a wave of the pair beam kernel of the wave before has left a slot)
__global__ __launch_bounds__(64) void ingest_kernel(IGArgs a) {
    const int C = a.C;
    for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < a.total_rows;
         row += (int64_t)gridDim.x * blockDim.x) {
        int64_t srow = row;
        if (a.reverse) {  // item of this row by binary search over the offsets, then mirror inside it
            int lo = 0, hi = a.n;
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.row_off[mid] <= row) lo = mid; else hi = mid; }
            srow = a.row_off[lo] + (a.row_off[lo + 1] - 1 - row);
        }
        double v[8];
        if (a.mode == 0) {
            const float* x = (const float*)a.src + srow * C;
            // scipy.special.logsumexp as shipped in this image (1.15.3, _logsumexp.py): the maximal elements leave
            // the sum (cnt of them), s = sum of exp(x - max) over the others in index order (numpy adds fewer than
            // 8 elements sequentially; the maximal ones contribute exp(-inf) = +0), lse = log1p(s / cnt) + log(cnt) + max
            float xv[8], m = x[0];
            for (int c = 0; c < C; ++c) { xv[c] = x[c]; m = fmaxf(m, xv[c]); }
            const float shift = isfinite(m) ? m : 0.f;
            float sum = 0.f, cnt = 0.f;
            for (int c = 0; c < C; ++c) {
                if (xv[c] == m) cnt += 1.f;
                else sum += expf(xv[c] - shift);
            }
            if (sum != 0.f) sum = sum / cnt;
            const float lse = (log1pf(sum) + logf(cnt)) + m;
            for (int c = 0; c < C; ++c) v[c] = (double)(xv[c] - lse);
        } else if (a.mode == 1) {
            const unsigned char* x = (const unsigned char*)a.src + srow * C;
            const double eps = 0.0000001;
            for (int c = 0; c < C; ++c) v[c] = log(((double)x[c] + eps) / (255 + eps));
        } else {
            const double* x = (const double*)a.src + srow * C;
            for (int c = 0; c < C; ++c) v[c] = x[c];
        }
        double* o = a.out + row * C;
        for (int c = 0; c < C; ++c) o[c] = v[a.perm[c]];
    }
}

extern "C" int po_launch_ingest(const void* src, const int64_t* row_off, int n, int C, int mode, const int* perm,
                                int reverse, int64_t total_rows, double* out, hipStream_t stream) {
    if (n <= 0 || total_rows <= 0) return PO_OK;
    if (C < 1 || C > 8 || mode < 0 || mode > 2) return PO_E_ARG;
    IGArgs a;
    a.src = src; a.row_off = row_off; a.n = n; a.C = C; a.mode = mode; a.reverse = reverse;
    for (int c = 0; c < 8; ++c) a.perm[c] = (perm && c < C) ? perm[c] : c;
    for (int c = 0; c < C; ++c) if (a.perm[c] < 0 || a.perm[c] >= C) return PO_E_ARG;
    a.out = out; a.total_rows = total_rows;
    const int64_t blocks = std::min<int64_t>((total_rows + 63) / 64, 256 * 32);
    hipLaunchKernelGGL(ingest_kernel, dim3((unsigned)blocks), dim3(64), 0, stream, a);
    return PO_OK;
}
