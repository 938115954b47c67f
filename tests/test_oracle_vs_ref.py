"""Oracle restatement vs the reference's own C++ (oracle/_ref/libporef.so, built from
/root/reference by oracle/Makefile).  Skipped where _ref has not been built."""
import numpy as np
import pytest

from oracle import po_oracle as O
from poreover_amd.synth import synth_pair

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref/libporef.so not built")


@pytest.mark.parametrize("idx", range(4))
@pytest.mark.parametrize("ff", [False, True])
def test_random_sweep(oracle, idx, ff):
    y1, y2 = synth_pair(100 + idx, T=360, flipflop=ff)
    U, V = len(y1), len(y2)
    env = O.diagonal_envelope(U, V, 15)
    for m in (["ctc_flipflop"] if ff else ["ctc", "ctc_merge_repeats"]):
        for W in (3, 5, 10):
            assert O.cpp_beam_search(y1, W, model_=m) == O.ref_beam_search(y1, W, model_=m)
            for meth in ("row", "row_col"):
                assert (O.cpp_beam_search_2d(y1, y2, env, W, model_=m, method_=meth)
                        == O.ref_beam_search_2d(y1, y2, env, W, model_=m, method_=meth)), (m, W, meth)
        b1, b2 = y1[:60], y2[:50]
        for meth in ("row", "grid"):
            assert (O.cpp_beam_search_2d(b1, b2, None, 4, model_=m, method_=meth)
                    == O.ref_beam_search_2d(b1, b2, None, 4, model_=m, method_=meth))
        lab = O.cpp_beam_search(y1[:100], 5, model_=m)
        assert O.cpp_forward(y1[:100], lab, model_=m) == O.ref_forward(y1[:100], lab, model_=m)
    if not ff:
        # grid + envelope: CTC only (the other models collapse to all -inf scores in a narrow band,
        # where the reference's order is heap-address order — DESIGN.md "tie rule")
        n = 80
        e2 = O.diagonal_envelope(n, n * V // U, 4)
        a1, a2 = y1[:n], y2[:n * V // U]
        assert (O.cpp_beam_search_2d(a1, a2, e2, 4, method_="grid")
                == O.ref_beam_search_2d(a1, a2, e2, 4, method_="grid"))
        seq, _ = O.viterbi_decode(y1)
        for band in (1000, 30):
            assert O.cpp_viterbi_acceptor(y1, seq, band).tolist() == O.ref_viterbi_acceptor(y1, seq, band).tolist()
        eg = np.array([(max(0, int(u / U * V) - 10), min(V, int(u / U * V) + 10)) for u in range(U + 1)])
        eg[U] = (max(0, V - 10), V)
        assert O.pair_gamma_log_envelope(y1, y2, eg) == O.ref_pair_gamma_log_envelope(y1, y2, eg)


def test_pipeline_envelope_t4000(oracle):
    """One full-size pair: oracle pipeline envelope + both pair methods vs the reference C++."""
    y1, y2 = synth_pair(0, T=4000)
    out = O.pair_decode(y1, y2, "poreover", 5, "row_col")
    assert out["consensus"] == O.ref_beam_search_2d(y1, y2, out["envelope"], 5, method_="row_col")
    assert O.cpp_beam_search(y1, 10) == O.ref_beam_search(y1, 10)
