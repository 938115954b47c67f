"""poreover.decoding.transducer (reference transducer.py:11-106) behind the same names: containers for a (T, C)
table of log-probabilities with argmax / Viterbi decoding — decoding runs on the GPU engine, and a container may hold
the basecaller's raw output (float32 logits, uint8 trace) for the device ingest instead of the float64 table.
remove_repeated is the reference's five-line helper restated as it is."""
import numpy as np

from .. import batch as _batch


def remove_repeated(s):
    """transducer.py:4-9"""
    out = ''
    for i in range(len(s)):
        if (i == 0) or (s[i - 1] != s[i]):
            out += s[i]
    return out


class DeferredTrace:
    """A basecaller output that has not been turned into log-probabilities yet: float32 logits (mode 0) or a uint8
    flip-flop trace (mode 1), as a C-contiguous (T, C) array, plus the host function that applies the reference's
    arithmetic to it (decode.logit_to_log_likelihood / the trace scaling of decode.model_from_trace).  The batched
    drivers hand `array` to the engine, whose ingest kernel does the same on the device (4 or 1 bytes per value over
    PCIe instead of 8); anything that reads .log_prob gets the host result."""

    def __init__(self, array, mode, host_fn):
        self.array = np.ascontiguousarray(array)
        assert self.array.ndim == 2
        self.mode = mode
        self.host_fn = host_fn


class transducer:
    def __init__(self, log_prob, kind, alphabet):
        self.alphabet = alphabet
        self.num_states = len(alphabet)
        self.kind = kind
        self._perm = list(range(self.num_states))   # pending column order / time reversal of a deferred trace
        self._rev = False
        if isinstance(log_prob, DeferredTrace):
            self._raw = log_prob
            self._log_prob = None
            self.t_max = len(log_prob.array)
            assert self.num_states == log_prob.array.shape[1]
        else:
            self._raw = None
            self._log_prob = np.asarray(log_prob).astype(np.float64)
            self.t_max = len(log_prob)
            assert self.num_states == len(self._log_prob[0])

    @property
    def log_prob(self):
        if self._log_prob is None:   # host ingest, the reference's own arithmetic
            lp = np.asarray(self._raw.host_fn(self._raw.array)).astype(np.float64)
            lp = lp[:, self._perm]
            self._log_prob = np.ascontiguousarray(lp[::-1] if self._rev else lp)
            self._raw = None
        return self._log_prob

    @log_prob.setter
    def log_prob(self, value):
        self._log_prob = value
        self._raw = None

    def engine_input(self):
        """(array, ingest mode, column permutation, time-reversed) for the engine's ingest stage: a deferred trace
        as it is, or the float64 log-probabilities (mode 2, identity)."""
        if self._raw is not None:
            return self._raw.array, self._raw.mode, list(self._perm), self._rev
        return np.ascontiguousarray(self._log_prob), 2, list(range(self.num_states)), False

    def _permute(self, perm, reverse=False):
        """log_prob[::-1 if reverse, perm] — applied now, or recorded for the device ingest"""
        if self._raw is not None:
            self._perm = [self._perm[c] for c in perm]
            self._rev = self._rev != bool(reverse)
        else:
            lp = self._log_prob[::-1] if reverse else self._log_prob
            self._log_prob = np.ascontiguousarray(lp[:, perm])

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
        self._permute([3, 2, 1, 0, 4], reverse=True)


class bonito(transducer):
    def __init__(self, log_prob, alphabet="ACGT"):
        super().__init__(log_prob, 'bonito', np.array(list(alphabet) + ['']))

    def reverse_complement(self):
        self._permute([3, 2, 1, 0, 4], reverse=True)


class flipflop(transducer):
    def __init__(self, log_prob):
        super().__init__(log_prob, 'flipflop', np.array(['A', 'C', 'G', 'T', 'a', 'c', 'g', 't']))

    def reverse_complement(self):
        """transducer.py:104-106"""
        self._permute([3, 2, 1, 0, 7, 6, 5, 4], reverse=True)
