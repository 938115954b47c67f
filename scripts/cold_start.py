#!/usr/bin/env python3
"""What the FIRST full-size call of a process costs (the reference's CLI is one process per job, pair_decode.py:230-303):
python scripts/cold_start.py [--pairs 10000] [--T 4000] [--reps 3].  Host float32 logits in -> strings out through
batch.pair_decode_stream, no warm-up call; prints library load, first call, later calls.  PO_PIPE_TRACE=1 lists every
buffer allocation of the pipeline with its duration."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _gen(job):
    from poreover_amd.synth import synth_pair
    lo, hi, T = job
    out = []
    for i in range(lo, hi):
        y1, y2 = synth_pair(i, T=T)
        out.append((y1.astype(np.float32), y2.astype(np.float32)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=10000)
    ap.add_argument("--T", type=int, default=4000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--procs", type=int, default=min(32, os.cpu_count() or 1))
    args = ap.parse_args()
    import multiprocessing as mp
    step = (args.pairs + args.procs - 1) // args.procs
    with mp.get_context("fork").Pool(args.procs) as pool:
        parts = pool.map(_gen, [(lo, min(lo + step, args.pairs), args.T) for lo in range(0, args.pairs, step)])
    pairs = [p for part in parts for p in part]
    l1 = [p[0] for p in pairs]; l2 = [p[1] for p in pairs]
    t0 = time.perf_counter()
    from poreover_amd import batch, _lib
    _lib.load()
    t_load = time.perf_counter() - t0
    times = []
    for rep in range(args.reps):
        st = {}
        t0 = time.perf_counter()
        res = batch.pair_decode_stream(l1, l2, "poreover", 5, "row_col", stats=st)
        times.append(time.perf_counter() - t0)
        print("call %d: %.4f s (%.0f pairs/s)  pipeline %s" % (rep, times[-1], len(l1) / times[-1], {k: round(v, 1) if isinstance(v, float) else v for k, v in st.items()}), flush=True)
    ok = sum(1 for r in res if r["status"] == 0)
    print({"pairs": len(l1), "decoded": ok, "lib_load_s": round(t_load, 3), "first_call_s": round(times[0], 4),
           "later_calls_s": [round(t, 4) for t in times[1:]]})


if __name__ == "__main__":
    main()
