"""The banded aligner's skewed-wavefront kernel (pair_prep_kernel<64, true>: no score table, three trace-back bits per
position) against the oracle's restatement of align.global_pair_banded (align/align.pyx:100-178) and against the
row-at-a-time kernel it replaces (po_set_align_route(1)), over the geometries that exercise its parts: one and several
512-column blocks, bands narrower than the sequences (positions left of / right of / below the computed cells),
unequal lengths, non-default scores (the trace-back keeps the defaults), degenerate sequences."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from poreover_amd import _lib
    _lib.load()
    return _lib


def _mutated(rng, ref, p=0.08):
    out = []
    for b in ref:
        r = rng.random()
        if r < p / 3:
            continue
        if r < 2 * p / 3:
            b = "ACGT"[rng.integers(4)]
        out.append(b)
        if rng.random() < p / 3:
            out.append("ACGT"[rng.integers(4)])
    return "".join(out)


def _cases(seed):
    rng = np.random.default_rng(seed)
    rnd = lambda n: "".join("ACGT"[k] for k in rng.integers(4, size=n))
    cases = []
    for l in (1, 2, 3, 7, 8, 9, 63, 64, 65, 300, 425, 510, 511, 512, 513, 514, 700, 1023, 1024, 1025, 1300, 2100):
        ref = rnd(l)
        cases.append((_mutated(rng, ref) or "A", _mutated(rng, ref) or "C"))
    for l1, l2 in ((1, 40), (40, 1), (5, 600), (600, 5), (100, 900), (900, 100), (520, 1500), (1500, 520), (2, 1030)):
        cases.append((rnd(l1), rnd(l2)))
    cases.append(("A" * 300, "A" * 280))                 # every diagonal ties with every gap path somewhere
    cases.append(("AC" * 400, "CA" * 390))
    cases.append((rnd(700), rnd(700)))                   # unrelated: the path wanders to the band's edges
    return cases


def _run(pairs, band, scores, legacy):
    from poreover_amd import batch, _lib
    _lib.set_align_route(legacy)
    try:
        return batch.align_batch(pairs, band_width=band, match=scores[0], mismatch=scores[1], gap_cost=scores[2])
    finally:
        _lib.set_align_route(False)


@pytest.mark.parametrize("band", [500, 64, 17, 3, 1])
def test_skew_vs_oracle_and_legacy(eng, oracle, band):
    pairs = _cases(100 + band)
    new = _run(pairs, band, (2, -1, -1), legacy=False)
    old = _run(pairs, band, (2, -1, -1), legacy=True)
    for k, (p, n, o) in enumerate(zip(pairs, new, old)):
        assert n == o, ("legacy", band, k, len(p[0]), len(p[1]))
        if len(p[0]) * min(len(p[1]), 2 * band + 1) <= 700 * 1001:     # the oracle is a per-cell loop
            w1, w2 = oracle.global_pair_banded(p[0], p[1], band)
            assert (n[0], n[1]) == ("".join(w1), "".join(w2)), ("oracle", band, k, len(p[0]), len(p[1]))


@pytest.mark.parametrize("scores", [(3, -2, -2), (1, -3, -1), (2, -1, -3), (5, 0, -1), (2, 2, -1)])
def test_skew_scores(eng, oracle, scores):
    pairs = _cases(7)[:14] + _cases(8)[22:]
    for band in (500, 20):
        new = _run(pairs, band, scores, legacy=False)
        old = _run(pairs, band, scores, legacy=True)
        for k, (p, n, o) in enumerate(zip(pairs, new, old)):
            assert n == o, ("legacy", scores, band, k)
            if len(p[0]) <= 520 and len(p[1]) <= 520:
                w1, w2 = oracle.global_pair_banded(p[0], p[1], band, *scores)
                assert (n[0], n[1]) == ("".join(w1), "".join(w2)), ("oracle", scores, band, k)


def test_skew_long_reads(eng):
    """Basecalls of several thousand bases: 10 - 16 column blocks, the band (500) sliding across them."""
    rng = np.random.default_rng(5)
    pairs = []
    for l in (5000, 8200):
        ref = "".join("ACGT"[k] for k in rng.integers(4, size=l))
        pairs.append((_mutated(rng, ref, 0.12), _mutated(rng, ref, 0.12)))
    pairs.append((pairs[0][0][:4000], pairs[0][1]))      # the shorter read ends early: the path leaves the band
    assert _run(pairs, 500, (2, -1, -1), legacy=False) == _run(pairs, 500, (2, -1, -1), legacy=True)
    assert _run(pairs, 40, (2, -1, -1), legacy=False) == _run(pairs, 40, (2, -1, -1), legacy=True)


def test_skew_many_pairs_one_launch(eng):
    """More pairs than resident workgroups: slices, block tables and boundary arrays are reused pair after pair."""
    rng = np.random.default_rng(11)
    pairs = []
    for k in range(6000):
        l = int(rng.integers(1, 140)) if k % 3 else int(rng.integers(500, 560))
        ref = "".join("ACGT"[c] for c in rng.integers(4, size=l))
        pairs.append((_mutated(rng, ref) or "G", _mutated(rng, ref) or "T"))
    assert _run(pairs, 500, (2, -1, -1), legacy=False) == _run(pairs, 500, (2, -1, -1), legacy=True)
