// TEST INFRASTRUCTURE ONLY — never linked, imported or executed by the product path.
//
// extern "C" shim around the reference's header-only C++ decoders, compiled by
// oracle/Makefile from the headers WHERE THEY LIE under /root/reference (nothing is
// copied into this repo). Output: oracle/_ref/libporef.so (git-ignored, travels to
// the GPU box with the gpurun snapshot).
//
// The reference reaches these functions through Cython (decoding_cpp.pyx:49-188),
// which only converts numpy arrays to double**/int** row-pointer tables; this shim
// does the same conversion from flat C arrays, so results are the reference's own.
//
//   ref_beam_search_1d      -> BeamSearch.h:400-408  (decoding_cpp.pyx:88-103)
//   ref_beam_search_2d      -> BeamSearch.h:411-458  (decoding_cpp.pyx:107-139)
//   ref_forward             -> PrefixTree.h:751-759  (decoding_cpp.pyx:49-65)
//   ref_viterbi_acceptor    -> Forward.h:14-121      (decoding_cpp.pyx:69-84)
//   ref_pair_gamma_envelope -> Gamma.h:15-98         (decoding_cpp.pyx:168-188)
#include <cstring>
#include <string>
#include <vector>

#include "BeamSearch.h"
#include "Forward.h"
#include "Gamma.h"

namespace {
std::vector<double*> rows_d(const double* y, int n, int c) {
    std::vector<double*> r(n > 0 ? n : 1);
    for (int i = 0; i < n; ++i) r[i] = const_cast<double*>(y) + (size_t)i * c;
    return r;
}
std::vector<int*> rows_i(const int* e, int n) {
    std::vector<int*> r(n > 0 ? n : 1);
    for (int i = 0; i < n; ++i) r[i] = const_cast<int*>(e) + (size_t)i * 2;
    return r;
}
// decoding_cpp.pyx:101,134 strips the root's '\0' with lstrip('\x00')
int emit(const std::string& s, char* out, int cap) {
    size_t b = 0;
    while (b < s.size() && s[b] == '\0') ++b;
    int n = (int)(s.size() - b);
    if (n + 1 > cap) return -1;
    std::memcpy(out, s.data() + b, n);
    out[n] = '\0';
    return n;
}
}  // namespace

extern "C" {

int ref_beam_search_1d(const double* y, int T, int C, const char* alphabet, int W,
                       const char* model, char* out, int cap) {
    auto r = rows_d(y, T, C);
    return emit(beam_search(r.data(), T, std::string(alphabet), W, std::string(model)), out, cap);
}

// env == NULL selects the no-envelope overload (BeamSearch.h:441)
int ref_beam_search_2d(const double* y1, int U, const double* y2, int V, int C,
                       const char* alphabet, const int* env, int W, const char* model,
                       const char* method, char* out, int cap) {
    auto r1 = rows_d(y1, U, C);
    auto r2 = rows_d(y2, V, C);
    std::string s;
    if (env) {
        auto re = rows_i(env, U);
        s = beam_search(r1.data(), r2.data(), U, V, std::string(alphabet), re.data(), W,
                        std::string(model), std::string(method));
    } else {
        s = beam_search(r1.data(), r2.data(), U, V, std::string(alphabet), W, std::string(model),
                        std::string(method));
    }
    return emit(s, out, cap);
}

double ref_forward(const double* y, int T, int C, const char* label, const char* alphabet,
                   const char* model) {
    auto r = rows_d(y, T, C);
    return forward(r.data(), T, std::string(label), std::string(alphabet), std::string(model));
}

// path_out[T]; the reference returns a string of digit characters (Forward.h:96-98)
int ref_viterbi_acceptor(const double* y, int T, int C, int band, const char* label,
                         const char* alphabet, int* path_out) {
    auto r = rows_d(y, T, C);
    std::string p = viterbi_acceptor_poreover(r.data(), T, band, std::string(label), std::string(alphabet));
    for (int t = 0; t < T && t < (int)p.size(); ++t) path_out[t] = p[t] - '0';
    return (int)p.size();
}

// env has U+1 rows, inclusive ends (Gamma.h:26-30)
double ref_pair_gamma_envelope(const double* y1, const double* y2, const int* env, int U, int V, int C) {
    auto r1 = rows_d(y1, U, C);
    auto r2 = rows_d(y2, V, C);
    auto re = rows_i(env, U + 1);
    return pair_gamma_log_envelope(r1.data(), r2.data(), re.data(), U, V, C);
}

}  // extern "C"
