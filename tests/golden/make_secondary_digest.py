#!/usr/bin/env python3
"""Generate tests/golden/secondary_digest.json: digests of what the CPU oracle (oracle/po_oracle.c, pinned to the reference
by tests/test_oracle_*.py) returns for the SECONDARY legs of bench.py — the configurations next to the headline that the
bench used to check by status only:

    config2            1 000 reads (read 1 of bench seeds 0 .. 999), 1-D beam search W = 10, model ctc
    config5            1 000 flip-flop reads (synth_read(500000 + i, flipflop)), 1-D beam search W = 10, model ctc_flipflop
    pair_bonito_W5     the first N pairs of the bench workload, whole pair-decode chain, kind bonito (ctc_merge_repeats), row_col W = 5
    pair_row_W5        ... kind poreover, method row, W = 5
    pair_row_col_W10   ... kind poreover, method row_col, W = 10
    pair_flipflop_W5   flip-flop pairs synth_pair(700000 + i, flipflop), kind flipflop, row_col W = 5
    pair_row_W25       the bench pairs, kind poreover, method row, W = 25 (the defaults of cpp_beam_search_2d, decoding_cpp.pyx:107)
    pair_row_col_W25   ... method row_col, W = 25

Records: 1-D legs [status, length, md5(seq)[:10]]; pair legs [status, len1, len2, len consensus, md5("seq1|seq2|consensus")[:10]]
(the format of batch_digest.json).  bench.py and tests/test_gpu_batch_scale.py compare with this file, so these legs need no
CPU decode on the GPU box.

    python3 tests/golden/make_secondary_digest.py [--pairs 1024] [--reads 1000] [--only LEG]
"""
import argparse
import hashlib
import json
import os
import sys
import time
from multiprocessing import Pool

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
T = 4000


def digest(*parts):
    return hashlib.md5("|".join(p if p is not None else "" for p in parts).encode()).hexdigest()[:10]


def read_ctc(i):
    from oracle import po_oracle as O
    from poreover_amd.synth import synth_pair
    y1, _ = synth_pair(i, T=T)
    try:
        s = O.cpp_beam_search(y1, 10, model_="ctc")
        return [0, len(s), digest(s)]
    except O.OracleError as e:
        return [e.code, 0, digest("")]


def read_ff(i):
    from oracle import po_oracle as O
    from poreover_amd.synth import synth_read
    y = synth_read(500000 + i, T, 0, True)
    try:
        s = O.cpp_beam_search(y, 10, model_="ctc_flipflop")
        return [0, len(s), digest(s)]
    except O.OracleError as e:
        return [e.code, 0, digest("")]


def pair(job):
    from oracle import po_oracle as O
    from poreover_amd.synth import synth_pair
    seed, kind, W, method, ff = job
    y1, y2 = synth_pair(seed, T=T, flipflop=ff)
    try:
        r = O.pair_decode(y1, y2, kind, W, method)
        return [int(r["status"]), len(r["seq1"] or ""), len(r["seq2"] or ""), len(r["consensus"] or ""), digest(r["seq1"], r["seq2"], r["consensus"])]
    except O.OracleError as e:   # (the reference's own assertions, e.g. the frame map of pair_decode.py:379)
        return [int(e.code), 0, 0, 0, digest("", "", "")]


LEGS = {
    "config2": lambda n, m: (read_ctc, range(m)),
    "config5": lambda n, m: (read_ff, range(m)),
    "pair_bonito_W5": lambda n, m: (pair, [(i, "bonito", 5, "row_col", False) for i in range(n)]),
    "pair_row_W5": lambda n, m: (pair, [(i, "poreover", 5, "row", False) for i in range(n)]),
    "pair_row_col_W10": lambda n, m: (pair, [(i, "poreover", 10, "row_col", False) for i in range(n)]),
    "pair_flipflop_W5": lambda n, m: (pair, [(700000 + i, "flipflop", 5, "row_col", True) for i in range(n)]),
    # the literal defaults of the Python API (decoding_cpp.pyx:107: beam_width_ = 25, method_ = "row"), and row_col at that width
    "pair_row_W25": lambda n, m: (pair, [(i, "poreover", 25, "row", False) for i in range(n)]),
    "pair_row_col_W25": lambda n, m: (pair, [(i, "poreover", 25, "row_col", False) for i in range(n)]),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1024)
    ap.add_argument("--reads", type=int, default=1000)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    from oracle import po_oracle as O
    O.build()
    path = os.path.join(HERE, "secondary_digest.json")
    out = json.load(open(path)) if os.path.exists(path) else {"T": T, "legs": {}}
    with Pool(min(8, os.cpu_count() or 1)) as pool:
        for name, mk in LEGS.items():
            if args.only and name != args.only:
                continue
            fn, jobs = mk(args.pairs, args.reads)
            t0 = time.time()
            recs = pool.map(fn, jobs, chunksize=4)
            out["legs"][name] = recs
            print("%s: %d records, %d with status 0, %.0f s" % (name, len(recs), sum(r[0] == 0 for r in recs), time.time() - t0), flush=True)
            with open(path, "w") as f:
                json.dump(out, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
