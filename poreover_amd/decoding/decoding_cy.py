"""Mirror of poreover.decoding.decoding_cy (reference decoding_cy.pyx): the pair gamma DP (dense and inside an
envelope), the forward row, the Viterbi acceptor — on the GPU engine — and the module's small containers / scalar
helpers (PySparseMatrix, logsumexp, pair_prefix_prob_log*), which are host glue upstream too."""
import numpy as np

from .. import batch as _batch

LOG_0 = -9999.0
LOG_1 = 0.0


class PySparseMatrix:
    """decoding_cy.pyx:22-39 over SparseMatrix.h:61-117: rows with an inclusive [start, end] column range, default
    value -inf, writes outside a row (or past the last row) silently dropped."""

    def __init__(self):
        self.rows = []          # (start, end, values)
        self.default_value = -np.inf

    def push_row(self, start, end):
        self.rows.append((int(start), int(end), np.full(max(int(end) - int(start) + 1, 0), self.default_value)))
        return True

    def get(self, i, j):
        if 0 <= i < len(self.rows):
            s, e, v = self.rows[i]
            if s <= j <= e:
                return float(v[j - s])
        return self.default_value

    def set(self, i, j, value):
        if 0 <= i < len(self.rows):
            s, e, v = self.rows[i]
            if s <= j <= e:
                v[j - s] = value
        return True


def logsumexp(x):
    """decoding_cy.pyx:159-173: log(sum(exp(x))) without a shift, as written upstream"""
    x = np.asarray(x, dtype=np.float64)
    tot = 0.0
    for v in x:
        tot += np.exp(v)
    return float(np.log(tot)) if tot > 0 else -np.inf


def pair_prefix_prob_log_from_vec(alpha_ast1, alpha_ast2, gamma):
    """decoding_cy.pyx:326-335: log sum_{u,v} exp(a1[u] + a2[v] + gamma[u+1, v+1]) - gamma[0, 0], summed u-major"""
    a1, a2, g = (np.asarray(x, dtype=np.float64) for x in (alpha_ast1, alpha_ast2, gamma))
    tot = 0.0
    for u in range(len(a1)):
        for v in range(len(a2)):
            tot += np.exp(a1[u] + a2[v] + g[u + 1, v + 1])
    return float(np.log(tot) - g[0, 0])


def pair_prefix_prob_log(alpha_ast_ast, gamma):
    """decoding_cy.pyx:339-347: the same from the outer sum, summed v-major"""
    aa, g = np.asarray(alpha_ast_ast, dtype=np.float64), np.asarray(gamma, dtype=np.float64)
    tot = 0.0
    for v in range(aa.shape[1]):
        for u in range(aa.shape[0]):
            tot += np.exp(aa[u, v] + g[u + 1, v + 1])
    return float(np.log(tot) - g[0, 0])


def pair_gamma_log_envelope(y1, y2, envelope, envelope_indices, gamma_, gamma_ast):
    """decoding_cy.pyx:224-271: the gamma DP over the cells of `envelope` (a PySparseMatrix whose rows give the
    inclusive column range of every row u <= U), log(exp(a) + exp(b)) arithmetic with -inf outside; fills and returns
    gamma_ (a PySparseMatrix with the same rows).  envelope_indices is implied by the rows (upstream iterates over
    it in reverse row-major order, the order of the device's anti-diagonal wavefront); gamma_ast is scratch."""
    y1 = np.asarray(y1, dtype=np.float64)
    y2 = np.asarray(y2, dtype=np.float64)
    U, V = len(y1), len(y2)
    rows = [(gamma_.rows[u][0], gamma_.rows[u][1]) if u < len(gamma_.rows) else (0, -1) for u in range(U + 1)]
    mat = _batch.pair_gamma_batch([y1], [y2], [np.array(rows, dtype=np.int32)], "cy_env", return_matrix=True)[0]
    for u in range(min(U + 1, len(gamma_.rows))):
        s, e, v = gamma_.rows[u]
        for j in range(max(s, 0), min(e, V) + 1):
            v[j - s] = mat[u, j]
    return gamma_


def pair_gamma_log_envelope2(y1, y2, envelope):
    """decoding_cy.pyx:275-322 takes an instance of a Python sparse-envelope class that upstream no longer ships
    (its callers are gone); use pair_gamma_log_envelope."""
    raise NotImplementedError("pair_gamma_log_envelope2 needs upstream's removed Python envelope class; "
                              "pair_gamma_log_envelope computes the same values")


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


def diagonal_band_envelope(U, V, width, inside=1, outside=0):
    """decoding_cy.pyx:41-56: steps across the main diagonal -> (PySparseMatrix with `inside` in the band,
    (U, 2) inclusive ranges, (cells, 2) indices), as upstream returns them"""
    envelope = PySparseMatrix()
    ranges, indices = [], []
    for u in range(U):
        center = int(np.round(V / U * u))
        start, end = max(center - width, 0), min(center + width, V - 1)
        envelope.push_row(start, end)
        ranges.append((start, end))
        for v in range(start, end + 1):
            envelope.set(u, v, inside)
            indices.append((u, v))
    return envelope, np.array(ranges), np.array(indices)
