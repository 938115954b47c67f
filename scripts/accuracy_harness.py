"""Accuracy / parity-at-scale report (north-star edit budget <= 0.1 %).

For N synthetic pairs (T ~ 4000, W = 5, row_col, padding 5):
  * the GPU pipeline's 1-D basecalls and consensus,
  * the same chain on the CPU through the oracle (test infrastructure: the C restatement pinned against the
    reference, oracle/po_oracle.c) in worker processes — run BEFORE the GPU is initialised,
  * edit distance GPU vs CPU (the parity figure), and of read 1 / read 2 / consensus vs the synthetic truth
    (what the pair decode buys).
Writes one JSON object to stdout (and to --out)."""
import argparse
import json
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poreover_amd.synth import synth_pair, synth_truth  # noqa: E402


def edit_distance(a, b):
    """Levenshtein distance, one numpy row at a time (the left dependency is a running minimum)."""
    if not a or not b:
        return max(len(a), len(b))
    x = np.frombuffer(a.encode(), dtype=np.uint8)
    y = np.frombuffer(b.encode(), dtype=np.uint8)
    idx = np.arange(len(y) + 1)
    prev = idx.copy()
    for i in range(1, len(x) + 1):
        t = np.empty(len(y) + 1, dtype=np.int64)
        t[0] = i
        t[1:] = np.minimum(prev[1:] + 1, prev[:-1] + (y != x[i - 1]))
        prev = np.minimum.accumulate(t - idx) + idx
    return int(prev[-1])


def _cpu_one(job):
    from oracle import po_oracle as O
    i, T = job
    y1, y2 = synth_pair(i, T=T)
    r = O.pair_decode(y1, y2, "poreover", 5, "row_col")
    return r["seq1"], r["seq2"], r.get("consensus") if r["status"] == 0 else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=256)
    ap.add_argument("--T", type=int, default=4000)
    ap.add_argument("--procs", type=int, default=0)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    jobs = [(100000 + i, args.T) for i in range(args.pairs)]
    procs = args.procs or min(32, os.cpu_count() or 1)
    t0 = time.time()
    with Pool(procs) as pool:      # CPU first: no fork after the GPU is up
        cpu = pool.map(_cpu_one, jobs, chunksize=4)
    t_cpu = time.time() - t0
    from poreover_amd import batch
    pairs = [synth_pair(i, T=args.T) for i, _ in jobs]
    t0 = time.time()
    gpu = batch.pair_decode_batch([p[0] for p in pairs], [p[1] for p in pairs], "poreover", 5, "row_col")
    t_gpu = time.time() - t0
    tot = {"gpu_vs_cpu_edits": 0, "cpu_consensus_bases": 0, "identical_consensus": 0, "identical_1d": 0, "decoded": 0,
           "read1_edits": 0, "read2_edits": 0, "consensus_edits": 0, "truth_bases": 0}
    for (i, _), c, g in zip(jobs, cpu, gpu):
        tot["identical_1d"] += int((g["seq1"], g["seq2"]) == (c[0], c[1]))
        if c[2] is None or g["consensus"] is None:
            assert (c[2] is None) == (g["consensus"] is None), "skip decisions differ"
            continue
        tot["decoded"] += 1
        tot["identical_consensus"] += int(g["consensus"] == c[2])
        tot["gpu_vs_cpu_edits"] += 0 if g["consensus"] == c[2] else edit_distance(g["consensus"], c[2])
        tot["cpu_consensus_bases"] += len(c[2])
        truth = synth_truth(i, args.T)
        tot["truth_bases"] += len(truth)
        tot["read1_edits"] += edit_distance(g["seq1"], truth)
        tot["read2_edits"] += edit_distance(g["seq2"], truth)
        tot["consensus_edits"] += edit_distance(g["consensus"], truth)
    rep = {"pairs": args.pairs, "T": args.T, "beam_width": 5, "method": "row_col", **tot,
           "gpu_vs_cpu_edit_fraction": tot["gpu_vs_cpu_edits"] / max(tot["cpu_consensus_bases"], 1),
           "error_rate_read1": tot["read1_edits"] / max(tot["truth_bases"], 1),
           "error_rate_read2": tot["read2_edits"] / max(tot["truth_bases"], 1),
           "error_rate_consensus": tot["consensus_edits"] / max(tot["truth_bases"], 1),
           "cpu_seconds_wall": round(t_cpu, 2), "cpu_procs": procs, "gpu_seconds_wall_incl_h2d": round(t_gpu, 2),
           "note": "CPU side = oracle (C restatement of the reference, pinned against it by tests/); budget: "
                   "gpu_vs_cpu_edit_fraction <= 0.001"}
    txt = json.dumps(rep, indent=1)
    print(txt)
    if args.out:
        with open(args.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
