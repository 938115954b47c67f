"""Drop-in for poreover.decoding.decoding_cpp (reference decoding_cpp.pyx:49-188): same names,
argument meaning and defaults; the work is done by the gfx950 engine (libporeover_hip.so)."""
import numpy as np

from .. import batch as _batch

DTYPE = np.float64


def _as2d(y_):
    y = np.asarray(y_, dtype=DTYPE, order="C")
    if y.ndim != 2:
        raise ValueError("expected a (T, C) matrix of log-probabilities")
    return y


def cpp_beam_search(y_, beam_width_=25, alphabet_="ACGT", model_="ctc"):
    """decoding_cpp.pyx:88-103"""
    return _batch.beam_search_batch([_as2d(y_)], beam_width_, alphabet_, model_)[0]


def cpp_beam_search_2d(y1_, y2_, envelope_ranges_=None, beam_width_=25, alphabet_="ACGT", model_="ctc",
                       method_="row"):
    """decoding_cpp.pyx:107-139; envelope_ranges_: (U, 2) half-open column range per row of y1_"""
    env = None if envelope_ranges_ is None else [np.asarray(envelope_ranges_, dtype=np.intc)]
    return _batch.beam_search_2d_batch([_as2d(y1_)], [_as2d(y2_)], env, beam_width_, alphabet_, model_,
                                       method_)[0]


def cpp_forward(y_, label_, alphabet_="ACGT", model_="ctc"):
    """decoding_cpp.pyx:49-65"""
    return float(_batch.forward_batch([_as2d(y_)], [label_], alphabet_, model_)[0])


def cpp_viterbi_acceptor(y_, label_, band_size=1000, alphabet_="ACGT"):
    """decoding_cpp.pyx:69-84 (the reference also prints "Mapping label" to stdout, Forward.h:20)"""
    return _batch.viterbi_acceptor_batch([_as2d(y_)], [label_], band_size, alphabet_)[0]


def cpp_pair_gamma_log_envelope(y1_, y2_, envelope_ranges_):
    """decoding_cpp.pyx:168-188: prints gamma(0,0) and returns None, as upstream.  envelope_ranges_: (U + 1, 2)
    with INCLUSIVE column ends (Gamma.h:26-30).  (batch.pair_gamma_batch returns the values.)"""
    print(float(_batch.pair_gamma_batch([_as2d(y1_)], [_as2d(y2_)], [np.asarray(envelope_ranges_, dtype=np.intc)])[0]))


def cpp_pair_prefix_search_log(y1_, y2_, envelope_ranges_, alphabet_="ACGT"):
    """decoding_cpp.pyx:143-164 -> pair_prefix_search_log (PairPrefixSearch.cpp:79-229): the pair prefix search with
    gamma restricted to an envelope ((U + 1, 2) inclusive column ranges); returns the label.  Upstream's version passes
    its gamma matrices by value (they stay empty) and reads past its forward rows; this is the working form: the
    search of the Python paths (prefix_search.py:247-314) on the envelope's gamma."""
    return _batch.pair_prefix_search_batch([_as2d(y1_)], [_as2d(y2_)], alphabet_, "py",
                                           [np.asarray(envelope_ranges_, dtype=np.intc)])[0][0]
