"""Mirror of poreover.decoding.transducer (reference transducer.py:11-106): containers for a (T, C)
table of log-probabilities with argmax / Viterbi decoding — decoding runs on the GPU engine."""
import numpy as np

from .. import batch as _batch


def remove_repeated(s):
    """transducer.py:4-9"""
    out = ''
    for i in range(len(s)):
        if (i == 0) or (s[i - 1] != s[i]):
            out += s[i]
    return out


class transducer:
    def __init__(self, log_prob, kind, alphabet):
        self.log_prob = np.asarray(log_prob).astype(np.float64)
        self.t_max = len(log_prob)
        self.alphabet = alphabet
        self.num_states = len(alphabet)
        self.kind = kind
        assert self.num_states == len(self.log_prob[0])

    def __getitem__(self, i):
        return self.log_prob.__getitem__(i)

    def _symbols(self):
        return "".join(a for a in self.alphabet if a != '' and a.isupper())

    def argmax_decode(self, return_path=False):
        """transducer.py:27-33 (blank -> '', repeats kept)"""
        seqs, paths = _batch.viterbi_batch([self.log_prob], "poreover", self._symbols(), return_path=True)
        return (seqs[0], paths[0]) if return_path else seqs[0]

    def viterbi_decode(self, return_path=False):
        """transducer.py:35-59"""
        seqs, paths = _batch.viterbi_batch([self.log_prob], self.kind, self._symbols(), return_path=True)
        return (seqs[0], paths[0]) if return_path else seqs[0]

    def __repr__(self):
        return 'transducer(kind=%s, alphabet=%s, t_max=%s)' % (self.kind, self.alphabet, self.t_max)


class poreover(transducer):
    def __init__(self, log_prob, alphabet="ACGT"):
        super().__init__(log_prob, 'poreover', np.array(list(alphabet) + ['']))

    def reverse_complement(self):
        """(A,C,G,T,-) -> (T,G,C,A,-), time reversed (transducer.py:68-70)"""
        self.log_prob = np.ascontiguousarray(self.log_prob[::-1, [3, 2, 1, 0, 4]])


class bonito(transducer):
    def __init__(self, log_prob, alphabet="ACGT"):
        super().__init__(log_prob, 'bonito', np.array(list(alphabet) + ['']))

    def reverse_complement(self):
        self.log_prob = np.ascontiguousarray(self.log_prob[::-1, [3, 2, 1, 0, 4]])


class flipflop(transducer):
    def __init__(self, log_prob):
        super().__init__(log_prob, 'flipflop', np.array(['A', 'C', 'G', 'T', 'a', 'c', 'g', 't']))

    def reverse_complement(self):
        """transducer.py:104-106"""
        self.log_prob = np.ascontiguousarray(self.log_prob[::-1, [3, 2, 1, 0, 7, 6, 5, 4]])
