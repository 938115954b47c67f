"""Synthetic basecaller outputs for tests and bench.py (SURVEY.md §8(d) recipe).

One reference sequence per pair, two independently mutated noisy reads of it, each rendered as
a (T, C) matrix of float32 logits with a +6 peak on the labelled column, then log-softmaxed in
float64 — the form the reference's decoders receive (decode.py:34-51).
"""
import numpy as np

__all__ = ["synth_pair", "synth_read", "synth_truth", "synth_pair_noise", "log_softmax"]


def log_softmax(logits):
    x = np.asarray(logits, dtype=np.float64)
    m = x.max(axis=-1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(axis=-1, keepdims=True))


def _mutate(rng, ref):
    out = []
    for b in ref:
        if rng.random() < 0.03:           # deletion
            continue
        if rng.random() < 0.03:           # substitution by a uniform other base
            b = (b + 1 + rng.integers(3)) % 4
        out.append(int(b))
        if rng.random() < 0.02:           # insertion after it
            out.append(int(rng.integers(4)))
    return np.asarray(out, dtype=np.int64)


def _render(rng, seq, T, flipflop, peak=6.0, sigma=1.0):
    C = 8 if flipflop else 5
    L = len(seq)
    if L > T:
        seq, L = seq[:T], T
    pos = np.sort(rng.choice(T, size=L, replace=False))
    if flipflop:
        # state persists until the next base; repeated bases alternate flip (c) / flop (c+4)
        lab = np.zeros(T, dtype=np.int64)
        state, prev_base, prev_state = int(seq[0]) if L else 0, -1, -1
        k = 0
        for t in range(T):
            if k < L and t >= pos[k]:
                b = int(seq[k])
                state = b + 4 if (b == prev_base and prev_state == b) else b
                prev_base, prev_state = b, state
                k += 1
            lab[t] = state
    else:
        lab = np.full(T, 4, dtype=np.int64)
        lab[pos] = seq
    logits = rng.normal(0, sigma, (T, C)).astype(np.float32)
    logits[np.arange(T), lab] += peak
    return log_softmax(logits)


def synth_pair(index, T=4000, base_seed=0, flipflop=False):
    """(y1, y2): float64 log-prob matrices of shapes (T, C) and (T2, C), T2 in [0.9T, 1.1T)."""
    rng = np.random.default_rng(base_seed + index)
    ref = rng.integers(4, size=max(1, int(T / 9.4)))
    T2 = int(T * rng.uniform(0.9, 1.1))
    y1 = _render(rng, _mutate(rng, ref), T, flipflop)
    y2 = _render(rng, _mutate(rng, ref), T2, flipflop)
    return y1, y2


def synth_read(index, T=4000, base_seed=0, flipflop=False):
    return synth_pair(index, T, base_seed, flipflop)[0]


def synth_truth(index, T=4000, base_seed=0):
    """The reference sequence both reads of synth_pair(index, T, base_seed) were mutated from (its first draw)."""
    rng = np.random.default_rng(base_seed + index)
    ref = rng.integers(4, size=max(1, int(T / 9.4)))
    return "".join("ACGT"[b] for b in ref)


def synth_pair_noise(index, T=4000, base_seed=0, peak=5.0, sigma=1.6):
    """(y1, y2, truth): the SAME sequence rendered twice with independent basecaller noise and no mutations — the
    setting the reference's pair decoding is for (README.md:5,12: two reads of one molecule).  The labelled column
    stands only `peak` above noise of width `sigma`, so a single read's Viterbi basecall has errors that the other
    read's evidence can correct; synth_pair's reads differ from their truth by real mutations, which no consensus
    of two can tell from signal."""
    rng = np.random.default_rng(base_seed + 7919 * 1000003 + index)
    ref = rng.integers(4, size=max(1, int(T / 9.4)))
    T2 = int(T * rng.uniform(0.9, 1.1))
    y1 = _render(rng, ref, T, False, peak, sigma)
    y2 = _render(rng, ref, T2, False, peak, sigma)
    return y1, y2, "".join("ACGT"[b] for b in ref)
