#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Replays the rounds scripts/fuzz_parity.py draws in pair mode (same seed, same --focus) on
the CPU emulation of the register-state kernel and compares with the oracle — for a round that faults on the GPU:
python tools/simt_emu/replay_fuzz.py --seed 411 --focus --rounds 8 [--lib tools/simt_emu/_build_asan/libemu_pair_beam.so]
(with the ASan build: LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0)"""
import argparse
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
sys.path.insert(0, HERE)


def oracle_one(job):
    from oracle import po_oracle as O
    y1, y2, env, W = job
    try:
        return O.cpp_beam_search_2d(y1, y2, env, W, model_="ctc", method_="row_col"), 0
    except O.OracleError as e:
        return "", e.code


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--focus", action="store_true")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--first", type=int, default=0, help="skip the rounds before this one (they are drawn, not run)")
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--kernel", default="reg")
    ap.add_argument("--group", type=int, default=8, help="pairs per emulated launch")
    ap.add_argument("--dump", default="", help="replay the round saved by fuzz_parity.py --dump instead of drawing rounds")
    ap.add_argument("--lib", default=os.path.join(HERE, "_build", "libemu_pair_beam.so"))
    args = ap.parse_args()
    from fuzz_parity import draw_round
    from check_emu import run_emu_batch
    kid = 1
    rng = np.random.default_rng(args.seed)
    bad = 0
    with ProcessPoolExecutor(args.procs) as pool:
        for rnd in range(args.rounds):
            if args.dump:
                d = np.load(args.dump)
                n = sum(1 for k in d.files if k.startswith("y1_"))
                model, method, W, style = str(d["model"]), str(d["method"]), int(d["W"]), str(d["style"])
                y1s = [d["y1_%d" % i] for i in range(n)]; y2s = [d["y2_%d" % i] for i in range(n)]; envs = [d["env_%d" % i] for i in range(n)]
            else:
                model, method, W, style, y1s, y2s, envs = draw_round(rng, args.focus)
            n = len(y1s)
            print("round", rnd, dict(model=model, method=method, W=W, style=style, n=n,
                                     widest=max(int((e[:, 1] - e[:, 0]).max()) for e in envs)), flush=True)
            if rnd < args.first or model != "ctc" or method != "row_col" or W > 6:
                continue
            want = list(pool.map(oracle_one, [(a, b, e, W) for a, b, e in zip(y1s, y2s, envs)]))
            cases = [(a, b, np.asarray(e, dtype=np.int32), w[0], w[1]) for a, b, e, w in zip(y1s, y2s, envs, want)]
            jobs = [(cases[i:i + args.group], W, args.lib, kid, len(cases[i:i + args.group])) for i in range(0, n, args.group)]
            res = [r for out in pool.map(run_emu_batch, jobs) for r in out]
            for i, ((got, st, deferred, dt, upd), c) in enumerate(zip(res, cases)):
                if deferred:
                    continue
                if not ((st == c[4]) and (c[4] != 0 or got == c[3])):
                    bad += 1
                    print("MISMATCH round", rnd, "index", i, "status", st, c[4], "U V", len(c[0]), len(c[1]), flush=True)
    print("replay: %d mismatches" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
