"""GPU: the multi-GPU decode path (BASELINE config 4: pairs sharded across the GPUs of a node, host gather).
One spawned worker process per device (dist.run_sharded), results in input order.  With one GPU in the box both
workers share device 0 — same code path, same processes, same gather."""
import argparse
import os

import numpy as np
import pytest

from poreover_amd.synth import synth_pair

pytestmark = pytest.mark.gpu


def _devices():
    from poreover_amd import _lib
    n = _lib.load().po_device_count()
    return [0, 1] if n >= 2 else [0, 0]      # world_size = min(2, device_count) devices, two workers either way


def test_pair_decode_batch_sharded_vs_oracle(oracle):
    from poreover_amd import batch
    y1s, y2s = [], []
    for i in range(11):
        a, b = synth_pair(8100 + i, T=300 + 170 * (i % 5))
        y1s.append(a); y2s.append(b)
    got = batch.pair_decode_batch_sharded(y1s, y2s, devices=_devices(), kind="poreover", beam_width=5, method="row_col")
    one = batch.pair_decode_batch(y1s, y2s, "poreover", 5, "row_col")
    assert len(got) == 11
    for i in range(11):
        want = oracle.pair_decode(y1s[i], y2s[i], "poreover", 5, "row_col")
        assert got[i]["status"] == want["status"] == one[i]["status"], i
        assert (got[i]["seq1"], got[i]["seq2"]) == (want["seq1"], want["seq2"]), i
        assert got[i]["consensus"] == want["consensus"] == one[i]["consensus"], i
        if want["status"] == 0:
            assert np.array_equal(got[i]["envelope"], want["envelope"]), i


def test_driver_sharded_files(tmp_path, golden, golden_inputs):
    """pair-decode driver over a pairs file, two worker processes: records equal the reference's, in input order"""
    from poreover_amd.decoding import pair_decode
    recs = [r for r in golden["pairs"] if r["kind"] == "poreover"][:8]
    lines = []
    for r in recs:
        for k in ("y1", "y2"):
            np.save(tmp_path / ("p%d_%s.npy" % (r["index"], k)), np.exp(golden_inputs["pair%d_%s" % (r["index"], k)]))
        lines.append("p%d_y1.npy\tp%d_y2.npy" % (r["index"], r["index"]))
    a = argparse.Namespace(dir=str(tmp_path), basecaller="poreover", reverse_complement=False, out=str(tmp_path / "o"),
                           threads=1, method="envelope", single="viterbi", logging="info", debug=False, algorithm="beam",
                           alignment="banded", beam_width=5, debug_envelope=False, diagonal_envelope=False,
                           diagonal_width=50, padding=5, skip_matches=False, skip_threshold=10,
                           beam_search_method="row_col", window=200)
    pairs = [l.split() for l in lines]
    got = pair_decode.decode_pairs(pairs, a, devices=_devices())
    for r, x in zip(recs, got):
        run = r["runs"]["row_col_w5_banded"]
        assert len(x) == run["n_out"]
        if len(x) == 3:
            assert "".join(x[1].split("\n")[1:]) == "".join(run["fasta_2d"].split("\n")[1:])
            assert x[2]["length1"] == run["summary"]["length1"]
