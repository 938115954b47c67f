"""GPU: the multi-GPU decode path (BASELINE config 4: pairs sharded across the GPUs of a node).  ONE process drives
every device (po_multi_pair_decode: a pipeline and a host thread per device, waves dealt as devices become free,
results written in input order); the file drivers' `split` / `--skip_matches` routes keep one spawned worker per
device (dist.run_sharded).  With one GPU in the box both pipelines share device 0 — same code path, same threads."""
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


def test_multi_device_stream_256_pairs_vs_oracle(oracle):
    """the in-process multi-device pipeline on devices [0, 0] (or [0, 1]): 256 pairs of uneven length, every consensus
    equal to the oracle's and to the single-device pipeline's, input order kept, every pair decoded exactly once, both
    pipelines used"""
    from poreover_amd import batch
    from poreover_amd.synth import synth_pair
    rng = np.random.default_rng(77)
    y1s, y2s = [], []
    for i in range(256):
        a, b = synth_pair(20000 + i, T=int(rng.integers(150, 700)))
        y1s.append(a); y2s.append(b)
    st = {}
    got = batch.pair_decode_stream(y1s, y2s, "poreover", 5, "row_col", devices=_devices(), wave_pairs=40, stats=st)
    one = batch.pair_decode_stream(y1s, y2s, "poreover", 5, "row_col")
    assert len(got) == 256 and sum(d["pairs"] for d in st["per_device"]) == 256
    assert all(d["pairs"] > 0 for d in st["per_device"]), st
    bad = 0
    for i in range(256):
        want = oracle.pair_decode(y1s[i], y2s[i], "poreover", 5, "row_col")
        assert got[i]["status"] == one[i]["status"] == want["status"], i
        assert (got[i]["seq1"], got[i]["seq2"]) == (one[i]["seq1"], one[i]["seq2"]), i
        assert got[i]["consensus"] == one[i]["consensus"], i
        bad += got[i]["consensus"] != want["consensus"]
    assert bad == 0, "%d of 256 pairs differ from the oracle" % bad


def _digest(seq1, seq2, cons):
    import hashlib
    return hashlib.md5(("%s|%s|%s" % (seq1, seq2, cons if cons is not None else "")).encode()).hexdigest()[:10]


def _gen4000(seed):
    return synth_pair(seed, T=4000)


def test_shard_of_the_strong_scaling_job():
    """1250 pairs of T = 4000 — rank 1's share of the 10 000-pair job on eight GPUs, i.e. bench seeds 1250 .. 2499 —
    through the pipelined host layer on the engine's route and on beam2d_kernel: every record equals the committed
    digest of the CPU oracle's result for that seed (tests/golden/batch_digest.json, all 10 000 bench pairs)."""
    import json
    from multiprocessing import get_context
    from poreover_amd import batch, _lib
    lo, n = 1250, 1250
    with get_context("spawn").Pool(min(8, os.cpu_count() or 1)) as pool:   # (this process holds a HIP context)
        pairs = pool.map(_gen4000, range(lo, lo + n), chunksize=16)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "batch_digest.json")) as f:
        recs = json.load(f)["records"][lo:lo + n]
    assert len(recs) == n
    y1s = [p[0].astype(np.float64) for p in pairs]; y2s = [p[1].astype(np.float64) for p in pairs]
    for route in ("auto", "legacy"):
        _lib.set_pair_route(route)
        try:
            got = batch.pair_decode_stream(y1s, y2s, "poreover", 5, "row_col", devices=_devices())
        finally:
            _lib.set_pair_route("auto")
        bad = [i for i, (g, r) in enumerate(zip(got, recs))
               if [g["status"], g["length1"], g["length2"], len(g["consensus"] or ""), _digest(g["seq1"], g["seq2"], g["consensus"])] != r]
        assert bad == [], (route, bad[:5])   # (bit-identical by construction: every other gate of the suite is exact too)


def test_eight_pipelines_512_uneven_pairs_vs_oracle(oracle):
    """po_multi_pair_decode with EIGHT pipelines — the shape of the 8-GPU node: range(n) when the box has that many
    devices, else the same device eight times (same code path: eight host threads, eight sets of slots, one planner) — on
    520 pairs of very uneven length: every pair decoded once, in input order, equal to the oracle's consensus."""
    from poreover_amd import batch, _lib
    nd = _lib.load().po_device_count()
    devs = list(range(8)) if nd >= 8 else ([i % nd for i in range(8)] if nd >= 2 else [0] * 8)
    rng = np.random.default_rng(91)
    base = []
    for i in range(26):
        a, b = synth_pair(52000 + i, T=int(np.exp(rng.uniform(np.log(120), np.log(2400)))))
        base.append((a, b, oracle.pair_decode(a, b, "poreover", 5, "row_col")))
    idx = rng.integers(len(base), size=520)
    st = {}
    got = batch.pair_decode_stream([base[i][0] for i in idx], [base[i][1] for i in idx], "poreover", 5, "row_col",
                                   devices=devs, wave_pairs=24, stats=st)
    assert len(got) == 520 and sum(d["pairs"] for d in st["per_device"]) == 520 and len(st["per_device"]) == 8
    assert sum(1 for d in st["per_device"] if d["pairs"] > 0) >= 4, st     # (the planner deals waves to whoever is free)
    for k, i in enumerate(idx):
        w = base[i][2]
        assert got[k]["status"] == w["status"] and (got[k]["consensus"] or "") == (w["consensus"] or "") and got[k]["seq1"] == w["seq1"], (k, int(i))


def test_bench_inprocess_devices_leg():
    """bench.py --inprocess_devices (the strong-scaling job driven by ONE process) runs as a child process on a small
    job, so that the driver's multi-GPU bench is never the first execution of that leg"""
    import json
    import subprocess
    import sys
    from poreover_amd import _lib
    nd = _lib.load().po_device_count()
    devs = ",".join(str(i % max(nd, 1)) for i in range(2))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--pairs", "256", "--steps", "1", "--warmup", "1", "--cpu_sample", "0",
                          "--no_secondary", "--inprocess_devices", devs], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["value"] > 0 and "strong_scaling" in rec and rec["scaling"] == "strong" and rec["n_gpus"] == 1 and "weak_scaling" not in rec
    assert "inprocess" in rec["strong_scaling"] and rec["strong_scaling"]["inprocess"]["pairs_per_s"] > 0, rec["strong_scaling"].keys()


def test_bench_two_ranks_one_job():
    """bench.py at N = 2 (round 6: the headline is ONE job split over the ranks, `"scaling": "strong"`, the weak figure beside
    it), both ranks on device 0 (--share_device: this box has one GPU; the figures mean nothing, the code path is the
    driver's 2 / 4 / 8-GPU run)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share_device", "--pairs", "192", "--steps", "1",
                          "--warmup", "1", "--cpu_sample", "0", "--no_secondary"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["value"] > 0
    assert rec["config"]["pairs_per_job"] == 192 and rec["config"]["pairs_per_gpu"] == 96
    assert rec["weak_scaling"]["scaling"] == "weak" and rec["weak_scaling"]["pairs_per_gpu"] == 192 and rec["weak_scaling"]["value"] > 0
    assert rec["strong_scaling"]["n_gpus"] == 2 and rec["strong_scaling"]["pairs"] == 192
    assert rec["parity_check"]["digest_mismatches"] == 0 and rec["parity_check"]["pairs_checked"] == 96
    assert "roofline" in rec and rec["roofline"]["avg_launch_ms"] > 0


def test_failed_call_leaves_the_pipeline_usable():
    """a call that fails inside the wave loop (here: a column permutation the ingest kernel refuses) must not leave a
    slot marked busy: the next call on the same cached pipeline decodes correctly"""
    from poreover_amd import batch, _lib
    from poreover_amd.synth import synth_pair
    y1s, y2s = [], []
    for i in range(9):
        a, b = synth_pair(40000 + i, T=260)
        y1s.append(a.astype(np.float32)); y2s.append(b.astype(np.float32))
    good = batch.pair_decode_stream(y1s, y2s, "poreover", 5, "row_col", wave_pairs=3)
    with pytest.raises(_lib.EngineError):
        batch.pair_decode_stream(y1s, y2s, "poreover", 5, "row_col", wave_pairs=3, perm1=[0, 1, 2, 3, 9])
    again = batch.pair_decode_stream(y1s[:4], y2s[:4], "poreover", 5, "row_col", wave_pairs=3)
    assert [r["consensus"] for r in again] == [r["consensus"] for r in good[:4]]
