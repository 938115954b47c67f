"""ctypes binding of oracle/_build/libpooracle.so (plain-C restatement) and, when present,
oracle/_ref/libporef.so (the reference's own C++ compiled through ref_shim.cpp).

TEST INFRASTRUCTURE ONLY — see po_oracle.h.  Function names mirror the reference's Python
API (decoding_cpp.pyx, transducer.py, align.pyx, envelope.py, pair_decode.py) so parity
tests read like the reference's own tests.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PO_ORACLE_SO selects another build of the same restatement (the sanitizer build: `make -C oracle asan`)
_ORACLE_SO = os.environ.get("PO_ORACLE_SO") or os.path.join(_HERE, "_build", "libpooracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libporef.so")

MODELS = {"ctc": 0, "ctc_merge_repeats": 1, "ctc_flipflop": 2}
METHODS = {"row": 0, "row_col": 1, "grid": 2}
KINDS = {"poreover": 0, "bonito": 1, "flipflop": 2}
E_CAP, E_ARG, E_ENVELOPE, E_NOMEM, E_DIVERGE = -1, -2, -3, -4, -5
SKIP_LENGTH, SKIP_IDENTITY = -10, -11

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force=False):
    """Compile the C restatement (and _ref when /root/reference is present)."""
    src = os.path.join(_HERE, "po_oracle.c")
    stale = (not os.path.exists(_ORACLE_SO)) or os.path.getmtime(_ORACLE_SO) < max(
        os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "po_oracle.h")))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    if os.path.isdir("/root/reference/poreover/decoding"):
        shim = os.path.join(_HERE, "ref_shim.cpp")
        if force or not os.path.exists(_REF_SO) or os.path.getmtime(_REF_SO) < os.path.getmtime(shim):
            subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    assert a.ndim == 2
    return a


def _env(e, U):
    if e is None:
        return None
    e = np.ascontiguousarray(e, dtype=np.intc)
    assert e.ndim == 2 and e.shape[1] == 2 and e.shape[0] >= U
    return e


class OracleError(RuntimeError):
    def __init__(self, code, what):
        super().__init__("%s failed with code %d" % (what, code))
        self.code = code


def _chk(rc, what):
    if rc < 0:
        raise OracleError(rc, what)
    return rc


class _Lib:
    def __init__(self, path):
        self.lib = C.CDLL(path)


_oracle = None
_ref = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        if not os.path.exists(_ORACLE_SO):
            build()
        L = C.CDLL(_ORACLE_SO)
        L.oracle_logaddexp.restype = C.c_double
        L.oracle_logaddexp.argtypes = [C.c_double, C.c_double]
        L.oracle_forward.restype = C.c_double
        L.oracle_pair_gamma_envelope.restype = C.c_double
        _oracle = L
    return _oracle


def have_ref():
    return os.path.exists(_REF_SO)


def ref_lib():
    global _ref
    if _ref is None:
        L = C.CDLL(_REF_SO)
        L.ref_forward.restype = C.c_double
        L.ref_pair_gamma_envelope.restype = C.c_double
        _ref = L
    return _ref


# ----------------------------------------------------------------------------- restatement
def logaddexp(a, b):
    return oracle_lib().oracle_logaddexp(a, b)


def viterbi_decode(y, kind="poreover", alphabet=None):
    y = _f64(y)
    T, Cc = y.shape
    path = np.zeros(T, dtype=np.intc)
    seq = C.create_string_buffer(T + 2)
    n = _chk(oracle_lib().oracle_viterbi_decode(
        y.ctypes.data_as(_dp), T, Cc, KINDS[kind], alphabet.encode() if alphabet else None,
        path.ctypes.data_as(_ip), seq, T + 2), "viterbi_decode")
    return seq.raw[:n].decode(), path.astype(np.int64)


def cpp_beam_search(y_, beam_width_=25, alphabet_="ACGT", model_="ctc"):
    y = _f64(y_)
    T, Cc = y.shape
    out = C.create_string_buffer(T + 2)
    n = _chk(oracle_lib().oracle_beam_search_1d(y.ctypes.data_as(_dp), T, Cc, alphabet_.encode(),
                                                int(beam_width_), MODELS[model_], out, T + 2),
             "beam_search_1d")
    return out.raw[:n].decode()


def cpp_beam_search_2d(y1_, y2_, envelope_ranges_=None, beam_width_=25, alphabet_="ACGT",
                       model_="ctc", method_="row"):
    y1, y2 = _f64(y1_), _f64(y2_)
    U, V = y1.shape[0], y2.shape[0]
    env = _env(envelope_ranges_, U)
    cap = U + V + 2
    out = C.create_string_buffer(cap)
    n = _chk(oracle_lib().oracle_beam_search_2d(
        y1.ctypes.data_as(_dp), U, y2.ctypes.data_as(_dp), V, y1.shape[1], alphabet_.encode(),
        env.ctypes.data_as(_ip) if env is not None else None, int(beam_width_), MODELS[model_],
        METHODS[method_], out, cap), "beam_search_2d")
    return out.raw[:n].decode()


def cpp_forward(y_, label_, alphabet_="ACGT", model_="ctc"):
    y = _f64(y_)
    return oracle_lib().oracle_forward(y.ctypes.data_as(_dp), y.shape[0], y.shape[1],
                                       label_.encode(), alphabet_.encode(), MODELS[model_])


def cpp_viterbi_acceptor(y_, label_, band_size=1000, alphabet_="ACGT"):
    y = _f64(y_)
    path = np.zeros(y.shape[0], dtype=np.intc)
    _chk(oracle_lib().oracle_viterbi_acceptor(y.ctypes.data_as(_dp), y.shape[0], y.shape[1],
                                              int(band_size), label_.encode(), alphabet_.encode(),
                                              path.ctypes.data_as(_ip)), "viterbi_acceptor")
    return path.astype(np.int64)


def viterbi_acceptor(y_, label_, alphabet="ACGT", band_size=0):
    y = _f64(y_)
    path = np.zeros(y.shape[0], dtype=np.intc)
    _chk(oracle_lib().oracle_viterbi_acceptor_cy(y.ctypes.data_as(_dp), y.shape[0], y.shape[1],
                                                 int(band_size), label_.encode(), alphabet.encode(),
                                                 path.ctypes.data_as(_ip)), "viterbi_acceptor_cy")
    return path.astype(np.int64)


def pair_gamma_log_envelope(y1_, y2_, envelope_inclusive):
    y1, y2 = _f64(y1_), _f64(y2_)
    U, V = y1.shape[0], y2.shape[0]
    env = _env(envelope_inclusive, U + 1)
    return oracle_lib().oracle_pair_gamma_envelope(y1.ctypes.data_as(_dp), y2.ctypes.data_as(_dp),
                                                   env.ctypes.data_as(_ip), U, V, y1.shape[1])


def pair_gamma_log(y1_, y2_, flavor="py"):
    y1, y2 = _f64(y1_), _f64(y2_)
    U, V = y1.shape[0], y2.shape[0]
    g = np.zeros((U + 1, V + 1))
    _chk(oracle_lib().oracle_pair_gamma_dense(y1.ctypes.data_as(_dp), U, y2.ctypes.data_as(_dp), V,
                                              y1.shape[1], 0 if flavor == "py" else 1,
                                              g.ctypes.data_as(_dp)), "pair_gamma_dense")
    return g


def forward_vec_log(s, i, y_, previous=None, flavor="py"):
    y = _f64(y_)
    fw = np.zeros(y.shape[0])
    prev = None if previous is None else np.ascontiguousarray(previous, dtype=np.float64)
    _chk(oracle_lib().oracle_forward_vec_log(int(s), int(i), y.ctypes.data_as(_dp), y.shape[0],
                                             y.shape[1], prev.ctypes.data_as(_dp) if prev is not None else None,
                                             0 if flavor == "py" else 1, fw.ctypes.data_as(_dp)),
         "forward_vec_log")
    return fw


def prefix_search_log(y_, flavor="py"):
    y = _f64(y_)
    out = C.create_string_buffer(y.shape[0] + 2)
    lp = C.c_double()
    n = _chk(oracle_lib().oracle_prefix_search_log(y.ctypes.data_as(_dp), y.shape[0], y.shape[1],
                                                   0 if flavor == "py" else 1, out, y.shape[0] + 2,
                                                   C.byref(lp)), "prefix_search_log")
    return out.raw[:n].decode(), lp.value


def pair_prefix_search_log(y1_, y2_, flavor="py", envelope_inclusive=None):
    """prefix_search.pair_prefix_search_log (flavor "py") / _cy ("cy"); with envelope_inclusive ((U + 1, 2) inclusive
    column ranges) gamma comes from the envelope DP of Gamma.h — the working form of PairPrefixSearch.cpp:79-229"""
    y1, y2 = _f64(y1_), _f64(y2_)
    cap = max(y1.shape[0], y2.shape[0]) + 8
    out = C.create_string_buffer(cap)
    lp = C.c_double()
    if envelope_inclusive is None:
        n = _chk(oracle_lib().oracle_pair_prefix_search_log(
            y1.ctypes.data_as(_dp), y1.shape[0], y2.ctypes.data_as(_dp), y2.shape[0], y1.shape[1],
            0 if flavor == "py" else 1, out, cap, C.byref(lp)), "pair_prefix_search_log")
    else:
        env = np.ascontiguousarray(envelope_inclusive, dtype=np.intc)
        assert env.shape == (y1.shape[0] + 1, 2)
        n = _chk(oracle_lib().oracle_pair_prefix_search_log_env(
            y1.ctypes.data_as(_dp), y1.shape[0], y2.ctypes.data_as(_dp), y2.shape[0], y1.shape[1],
            0 if flavor == "py" else 1, env.ctypes.data_as(_ip), out, cap, C.byref(lp)), "pair_prefix_search_log_env")
    return out.raw[:n].decode(), lp.value


def _align(fn, seq1, seq2, *extra):
    cap = 3 * (len(seq1) + len(seq2)) + 16
    a1, a2 = C.create_string_buffer(cap), C.create_string_buffer(cap)
    n = _chk(fn(seq1.encode(), len(seq1), seq2.encode(), len(seq2), *extra, a1, a2, cap), "align")
    return list(a1.raw[:n].decode()), list(a2.raw[:n].decode())


def _with_scores(fn, match, mismatch, gap_cost):
    L = oracle_lib()
    L.oracle_set_nw_scores(int(match), int(mismatch), int(gap_cost))
    try:
        return fn()
    finally:
        L.oracle_set_nw_scores(2, -1, -1)


def global_pair(seq1, seq2, match=2, mismatch=-1, gap_cost=-1):
    if (match, mismatch, gap_cost) != (2, -1, -1):
        return _with_scores(lambda: global_pair(seq1, seq2), match, mismatch, gap_cost)
    return _align(oracle_lib().oracle_global_pair, seq1, seq2)


def global_pair_banded(seq1, seq2, band_width=500, match=2, mismatch=-1, gap_cost=-1):
    if (match, mismatch, gap_cost) != (2, -1, -1):
        return _with_scores(lambda: global_pair_banded(seq1, seq2, band_width), match, mismatch, gap_cost)
    return _align(oracle_lib().oracle_global_pair_banded, seq1, seq2, int(band_width))


def get_sequence_mapping(path, kind):
    p = np.ascontiguousarray(path, dtype=np.intc)
    out = np.zeros(max(len(p), 1), dtype=np.intc)
    n = _chk(oracle_lib().oracle_sequence_mapping(p.ctypes.data_as(_ip), len(p), KINDS[kind],
                                                  out.ctypes.data_as(_ip)), "sequence_mapping")
    return out[:n].astype(np.int64)


def build_envelope(U, V, align1, align2, s2s1, s2s2, padding=150):
    a1, a2 = "".join(align1).encode(), "".join(align2).encode()
    assert len(a1) == len(a2)
    m1 = np.ascontiguousarray(s2s1, dtype=np.intc)
    m2 = np.ascontiguousarray(s2s2, dtype=np.intc)
    env = np.zeros((U, 2), dtype=np.intc)
    _chk(oracle_lib().oracle_build_envelope(U, V, a1, a2, len(a1), m1.ctypes.data_as(_ip), len(m1),
                                            m2.ctypes.data_as(_ip), len(m2), int(padding),
                                            env.ctypes.data_as(_ip)), "build_envelope")
    return env.astype(np.int64)


def diagonal_envelope(U, V, width):
    env = np.zeros((U, 2), dtype=np.intc)
    oracle_lib().oracle_diagonal_envelope(U, V, int(width), env.ctypes.data_as(_ip))
    return env.astype(np.int64)


class _Summary(C.Structure):
    _fields_ = [("len1", C.c_int), ("len2", C.c_int), ("ncol", C.c_int), ("identity", C.c_double),
                ("skipped", C.c_int)]


def pair_decode(y1_, y2_, kind="poreover", beam_width=5, method="row_col", padding=5,
                alignment="banded"):
    """pair_decode_helper stage chain with the CLI defaults; returns a dict."""
    y1, y2 = _f64(y1_), _f64(y2_)
    U, V = y1.shape[0], y2.shape[0]
    cap = U + V + 2
    s1, s2, cons = (C.create_string_buffer(cap) for _ in range(3))
    env = np.zeros((U, 2), dtype=np.intc)
    sm = _Summary()
    rc = oracle_lib().oracle_pair_decode(
        y1.ctypes.data_as(_dp), U, y2.ctypes.data_as(_dp), V, y1.shape[1], KINDS[kind],
        int(beam_width), METHODS[method], int(padding), 1 if alignment == "full" else 0, s1, s2, cons,
        cap, env.ctypes.data_as(_ip), C.byref(sm))
    if rc < 0 and rc not in (SKIP_LENGTH, SKIP_IDENTITY):
        raise OracleError(rc, "pair_decode")
    return {"seq1": s1.value.decode(), "seq2": s2.value.decode(),
            "consensus": cons.value.decode() if rc >= 0 else None, "envelope": env.astype(np.int64),
            "length1": sm.len1, "length2": sm.len2, "ncol": sm.ncol,
            "sequence_identity": sm.identity, "skipped": sm.skipped, "status": 0 if rc >= 0 else rc}


# ----------------------------------------------------------------------------- trace ingest (numpy)
def load_logits(arr, flatten=True):
    """decode.load_logits on an array instead of a file (decode.py:34-51): rows that already sum to 1 are
    probabilities -> log; otherwise (B, W, C) logits -> x - logsumexp(x, axis=2) in the array's own precision
    (float32 for `poreover call` output).  logsumexp is a third-party function the reference does not pin
    (requirements.txt names scipy without a version); the parity target is the reference run in this image, i.e.
    scipy 1.15.3, whose published algorithm (scipy/special/_logsumexp.py::_logsumexp) is restated here: the
    maximal elements are taken out of the sum (m = how many there are), s = sum(exp(x - max)) over the others,
    result = log1p(s / m) + log(m) + max."""
    a = np.asarray(arr)
    if np.isclose(np.sum(a[0]), 1):
        out = np.log(a)
    else:
        a_max = np.max(a, axis=2, keepdims=True)
        is_max = (a == a_max)
        m = np.sum(is_max.astype(a.dtype), axis=2, keepdims=True, dtype=a.dtype)
        shift = np.where(np.isfinite(a_max), a_max, np.asarray(0, dtype=a.dtype))
        with np.errstate(divide="ignore", invalid="ignore"):
            e = np.exp(np.where(is_max, -np.inf, a).astype(a.dtype) - shift)
            ssum = np.sum(e, axis=2, keepdims=True, dtype=a.dtype)
            ssum = np.where(ssum == 0, ssum, ssum / m)
            lse = (np.log1p(ssum) + np.log(m) + a_max).squeeze(axis=2)
        out = (a.T - lse.T).T
    if flatten and out.ndim > 2:
        out = np.concatenate(out)
    return out


def reverse_complement(log_prob, kind="poreover"):
    """transducer.reverse_complement (transducer.py:68-70 poreover / bonito, :104-106 flip-flop)"""
    perm = [3, 2, 1, 0, 7, 6, 5, 4] if kind == "flipflop" else [3, 2, 1, 0, 4]
    return np.ascontiguousarray(np.asarray(log_prob)[::-1, perm])


def trace_to_log_prob(trace_u8):
    """decode.py:89-93,99-103: uint8 flip-flop trace -> log((x + eps) / (255 + eps))"""
    eps = 0.0000001
    return np.log((np.asarray(trace_u8) + eps) / (255 + eps))


# ------------------------------------------------------------- the reference's own C++ (_ref)
def ref_beam_search(y_, beam_width_=25, alphabet_="ACGT", model_="ctc"):
    y = _f64(y_)
    out = C.create_string_buffer(y.shape[0] + 2)
    n = ref_lib().ref_beam_search_1d(y.ctypes.data_as(_dp), y.shape[0], y.shape[1], alphabet_.encode(),
                                     int(beam_width_), model_.encode(), out, y.shape[0] + 2)
    assert n >= 0
    return out.raw[:n].decode()


def ref_beam_search_2d(y1_, y2_, envelope_ranges_=None, beam_width_=25, alphabet_="ACGT", model_="ctc",
                       method_="row"):
    y1, y2 = _f64(y1_), _f64(y2_)
    U, V = y1.shape[0], y2.shape[0]
    env = _env(envelope_ranges_, U)
    cap = U + V + 2
    out = C.create_string_buffer(cap)
    n = ref_lib().ref_beam_search_2d(y1.ctypes.data_as(_dp), U, y2.ctypes.data_as(_dp), V, y1.shape[1],
                                     alphabet_.encode(),
                                     env.ctypes.data_as(_ip) if env is not None else None,
                                     int(beam_width_), model_.encode(), method_.encode(), out, cap)
    assert n >= 0
    return out.raw[:n].decode()


def ref_forward(y_, label_, alphabet_="ACGT", model_="ctc"):
    y = _f64(y_)
    return ref_lib().ref_forward(y.ctypes.data_as(_dp), y.shape[0], y.shape[1], label_.encode(),
                                 alphabet_.encode(), model_.encode())


def ref_viterbi_acceptor(y_, label_, band_size=1000, alphabet_="ACGT"):
    y = _f64(y_)
    path = np.zeros(y.shape[0], dtype=np.intc)
    ref_lib().ref_viterbi_acceptor(y.ctypes.data_as(_dp), y.shape[0], y.shape[1], int(band_size),
                                   label_.encode(), alphabet_.encode(), path.ctypes.data_as(_ip))
    return path.astype(np.int64)


def ref_pair_gamma_log_envelope(y1_, y2_, envelope_inclusive):
    y1, y2 = _f64(y1_), _f64(y2_)
    U, V = y1.shape[0], y2.shape[0]
    env = _env(envelope_inclusive, U + 1)
    return ref_lib().ref_pair_gamma_envelope(y1.ctypes.data_as(_dp), y2.ctypes.data_as(_dp),
                                             env.ctypes.data_as(_ip), U, V, y1.shape[1])
