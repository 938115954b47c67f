"""Mirror of the parts of poreover.decoding.decoding_cy (reference decoding_cy.pyx) that have a GPU
implementation: the dense pair gamma DP, the forward row and the banded-envelope helper."""
import numpy as np

from .. import batch as _batch


def pair_gamma_log(y1, y2):
    """decoding_cy.pyx:177-220: the dense (U+1, V+1) gamma matrix (LOG_0 = -9999 arithmetic)"""
    return _batch.pair_gamma_batch([np.asarray(y1, dtype=np.float64)], [np.asarray(y2, dtype=np.float64)], None, "cy",
                                   return_matrix=True)[0]


def forward_vec_log(s, i, y, previous=None):
    """decoding_cy.pyx:127-156: forward row of a label of length i ending in symbol s (-1: blank), LOG_0 = -9999"""
    return _batch.forward_vec_batch([np.asarray(y, dtype=np.float64)], s, i,
                                    None if previous is None else [np.asarray(previous, dtype=np.float64)], "cy")[0]


def viterbi_acceptor(y, label_, alphabet="ACGT", band_size=0):
    """decoding_cy.pyx:60-123: best alignment path of a known label, the Cython twin of cpp_viterbi_acceptor
    (its own tie rule and band expression, reproduced as written); alphabet may be a dict like the reference's"""
    sym = "".join(alphabet.keys()) if hasattr(alphabet, "keys") else "".join(alphabet)
    return _batch.viterbi_acceptor_batch([np.asarray(y, dtype=np.float64)], [label_], band_size, sym, "cy")[0]


def diagonal_band_envelope(U, V, width):
    """decoding_cy.pyx:41-56: inclusive (start, end) per row around the main diagonal -> (U, 2) array"""
    out = []
    for u in range(U):
        center = int(np.round(V / U * u))
        out.append((max(center - width, 0), min(center + width, V - 1)))
    return np.array(out)
