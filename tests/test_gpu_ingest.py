"""GPU: device ingest (logits -> log-likelihood, uint8 trace scaling, column permutation, reverse
complement) vs the reference's own outputs (golden) and numpy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from poreover_amd import _lib, batch
    _lib.load()
    return batch


def test_logits_to_log_likelihood_golden(eng, golden_inputs):
    lg = golden_inputs["ingest_logits"]                       # (3, 40, 5) float32, as `poreover call` writes
    want = golden_inputs["ingest_logits_out"]                 # the reference's load_logits(flatten=True), float32
    got = eng.ingest_batch([np.concatenate(lg)])[0]
    assert got.dtype == np.float64 and got.shape == want.shape
    # float32 arithmetic on both sides (device expf/logf vs scipy): agreement to 2 float32 ulp
    assert np.allclose(got, want.astype(np.float64), rtol=0, atol=5e-7)
    assert np.allclose(np.exp(got).sum(axis=1), 1.0, atol=1e-6)
    bon = eng.ingest_batch([np.concatenate(lg)], perm=[1, 2, 3, 4, 0])[0]      # decode.py:79
    assert np.array_equal(bon, got[:, [1, 2, 3, 4, 0]])


def test_trace_scaling_and_revcomp(eng, golden_inputs):
    rng = np.random.default_rng(3)
    tr = [rng.integers(0, 256, size=(n, 8), dtype=np.uint8) for n in (17, 300)]
    got = eng.ingest_batch(tr)
    for g, t in zip(got, tr):
        want = np.log((t + 0.0000001) / (255 + 0.0000001))     # decode.py:91-92
        assert np.allclose(g, want, rtol=1e-15, atol=0)
    y = golden_inputs["revcomp_poreover_in"]
    rc = eng.ingest_batch([y, y[:7]], perm=[3, 2, 1, 0, 4], reverse=True)       # transducer.py:68-70
    assert np.array_equal(rc[0], golden_inputs["revcomp_poreover_out"])
    assert np.array_equal(rc[1], y[:7][::-1][:, [3, 2, 1, 0, 4]])
    ff = np.arange(48, dtype=np.float64).reshape(6, 8)
    assert np.array_equal(eng.ingest_batch([ff], perm=[3, 2, 1, 0, 7, 6, 5, 4], reverse=True)[0],
                          ff[::-1][:, [3, 2, 1, 0, 7, 6, 5, 4]])                  # transducer.py:104-106
    with pytest.raises(ValueError):
        eng.ingest_batch([np.zeros((3, 5), dtype=np.int32)])
