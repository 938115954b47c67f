#!/usr/bin/env python3
"""Generate tests/golden/batch_digest.json: per-pair digests of what the CPU oracle (oracle/po_oracle.c, pinned to
the reference by tests/test_oracle_golden.py, test_oracle_vs_ref.py and test_oracle_real.py) returns for the first
N pairs of the bench workload — synth_pair(seed, T=4000) for seed = 0 .. N-1, CLI defaults (W = 5, row_col, banded
alignment, padding 5).  bench.py's rank 0 decodes exactly these pairs first, and tests/test_gpu_batch_scale.py runs
them through po_pair_decode_batch: both compare with this file, so parity at batch scale needs no CPU decode on the
GPU box.  Each record: status, lengths of the two 1-D basecalls and of the consensus, and the first 10 hex digits of
the md5 of "seq1|seq2|consensus".

    python3 tests/golden/make_batch_digest.py [--pairs 1024]
"""
import argparse
import hashlib
import json
import os
import sys
from multiprocessing import Pool

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)


def digest(seq1, seq2, cons):
    return hashlib.md5(("%s|%s|%s" % (seq1, seq2, cons if cons is not None else "")).encode()).hexdigest()[:10]


def one(seed):
    from oracle import po_oracle as O
    from poreover_amd.synth import synth_pair
    y1, y2 = synth_pair(seed, T=4000)
    r = O.pair_decode(y1, y2, "poreover", 5, "row_col")
    return [int(r["status"]), len(r["seq1"]), len(r["seq2"]), len(r["consensus"] or ""),
            digest(r["seq1"], r["seq2"], r["consensus"])]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1024)
    args = ap.parse_args()
    from oracle import po_oracle as O
    O.build()
    with Pool(min(8, os.cpu_count() or 1)) as pool:
        recs = pool.map(one, range(args.pairs), chunksize=8)
    out = {"T": 4000, "beam_width": 5, "method": "row_col", "kind": "poreover", "seed0": 0, "records": recs}
    with open(os.path.join(HERE, "batch_digest.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote %d records; decoded %d, bases %d" % (len(recs), sum(r[0] == 0 for r in recs), sum(r[3] for r in recs)))


if __name__ == "__main__":
    main()
