#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Runs the register-state pair beam kernel (beam2d_reg_kernel, every model and lane layout) on the CPU (SIMT emulator, `make -C tools/simt_emu`)
and compares its strings with the oracle's: python tools/simt_emu/check_emu.py [--n 16] [--T 300] [--seed 1] [--procs 8]
Envelopes: the pipeline's own (Viterbi + banded alignment) and the fuzz script's stairs / wobble / bursts."""
import argparse
import ctypes as C
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


MODELS = {"ctc": (0, "poreover", "ctc"), "merge": (1, "bonito", "ctc_merge_repeats"), "flipflop": (2, "flipflop", "ctc_flipflop")}
MODEL = [os.environ.get("EMU_MODEL", "ctc")]


def make_case(job):
    seed, T, W, style = job
    from oracle import po_oracle as O
    from poreover_amd.synth import synth_pair
    from fuzz_parity import jagged
    _, kind, model_ = MODELS[MODEL[0]]
    rng = np.random.default_rng(seed)
    y1, y2 = synth_pair(int(rng.integers(1 << 30)), T=T, flipflop=(kind == "flipflop"))
    if style == "pipeline":
        try:
            r = O.pair_decode(y1, y2, kind, W, "row_col")
        except O.OracleError:   # (the reference's own assertions refuse some pairs: e.g. Bonito's frame map, pair_decode.py:379)
            return None
        if r["status"] != 0 or r.get("envelope") is None:
            return None
        env = np.asarray(r["envelope"], dtype=np.int32)
    else:
        env = jagged(rng, len(y1), len(y2), style, int(rng.integers(2, 12))).astype(np.int32)
    try:
        want = O.cpp_beam_search_2d(y1, y2, env, W, model_=model_, method_="row_col")
        code = 0
    except O.OracleError as e:
        want, code = "", e.code
    return y1, y2, env, want, code


def run_emu(job):
    case, W, lib_path = job[:3]
    kernel = job[3] if len(job) > 3 else 0
    y1, y2, env, want, code = case
    lib = C.CDLL(lib_path)
    n = 1
    y1 = np.ascontiguousarray(y1, dtype=np.float64); y2 = np.ascontiguousarray(y2, dtype=np.float64)
    o1 = np.array([0, len(y1)], dtype=np.int64); o2 = np.array([0, len(y2)], dtype=np.int64)
    cap = len(y1) + len(y2) + 8
    seq = np.zeros(cap, dtype=np.uint8); so = np.array([0, cap], dtype=np.int64)
    sl = np.zeros(1, dtype=np.int32); st = np.zeros(1, dtype=np.int32)
    upd = np.zeros(2, dtype=np.uint64)
    env = np.ascontiguousarray(env, dtype=np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    alphabet = int.from_bytes(b"ACGT", "little")
    t0 = time.time()
    nblocks = int(os.environ.get("EMU_SLOTS", "1"))
    deferred = lib.emu_ring_pair_beam(p(y1), p(o1), p(y2), p(o2), p(env), n, y1.shape[1], 4, C.c_uint32(alphabet), W, p(seq), p(so), p(sl), p(st), nblocks, p(upd), kernel, MODELS[MODEL[0]][0])
    got = bytes(seq[: sl[0]]).decode()
    return got, int(st[0]), deferred, time.time() - t0, int(upd[1])


def run_emu_batch(job):
    """several pairs in ONE emulated launch (pair waves of one workgroup working side by side)"""
    cases, W, lib_path, kernel, slots = job
    lib = C.CDLL(lib_path)
    n = len(cases)
    y1 = np.ascontiguousarray(np.concatenate([c[0] for c in cases]), dtype=np.float64)
    y2 = np.ascontiguousarray(np.concatenate([c[1] for c in cases]), dtype=np.float64)
    env = np.ascontiguousarray(np.concatenate([c[2] for c in cases]), dtype=np.int32)
    o1 = np.zeros(n + 1, dtype=np.int64); o2 = np.zeros(n + 1, dtype=np.int64); so = np.zeros(n + 1, dtype=np.int64)
    for i, c in enumerate(cases):
        o1[i + 1] = o1[i] + len(c[0]); o2[i + 1] = o2[i] + len(c[1]); so[i + 1] = so[i] + len(c[0]) + len(c[1]) + 8
    seq = np.zeros(int(so[-1]), dtype=np.uint8)
    sl = np.zeros(n, dtype=np.int32); st = np.zeros(n, dtype=np.int32); upd = np.zeros(2, dtype=np.uint64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    alphabet = int.from_bytes(b"ACGT", "little")
    t0 = time.time()
    deferred = lib.emu_ring_pair_beam(p(y1), p(o1), p(y2), p(o2), p(env), n, y1.shape[1], 4, C.c_uint32(alphabet), W, p(seq), p(so), p(sl), p(st), slots, p(upd), kernel, MODELS[MODEL[0]][0])
    dt = time.time() - t0
    return [(bytes(seq[so[i]: so[i] + sl[i]]).decode(), int(st[i]), deferred if st[i] == -100 else 0, dt / n, 0) for i in range(n)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--T", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--W", type=int, default=5)
    ap.add_argument("--lib", default=os.path.join(HERE, "_build", "libemu_pair_beam.so"))
    ap.add_argument("--styles", default="pipeline,stairs,wobble,bursts")
    ap.add_argument("--batch", type=int, default=1, help="pairs per emulated launch (with as many pair slots)")
    ap.add_argument("--slots", type=int, default=0, help="with --batch: pair slots of the emulated launch (0 = one per pair; 1 = every pair of a "
                    "batch goes through ONE slice of the value store, one after the other)")
    ap.add_argument("--kernel", default="reg", help="reg (kept for old command lines)")
    ap.add_argument("--model", default=MODEL[0], help="ctc | merge | flipflop (the register-state kernel serves all three)")
    args = ap.parse_args()
    MODEL[0] = args.model
    os.environ["EMU_MODEL"] = args.model   # (the pool's workers read it)
    styles = args.styles.split(",")
    rng = np.random.default_rng(args.seed)
    jobs = []
    for i in range(args.n):
        T = int(rng.integers(max(20, args.T // 3), args.T + 1))
        W = args.W if args.W > 0 else int([1, 2, 3, 4, 5, 5, 5, 6][rng.integers(8)])
        jobs.append((int(rng.integers(1 << 30)), T, W, styles[i % len(styles)]))
    from concurrent.futures import ProcessPoolExecutor   # (a worker the emulator aborts raises BrokenProcessPool; mp.Pool would hang)
    with ProcessPoolExecutor(args.procs) as pool:
        cases = list(pool.map(make_case, jobs))
        keep = [(c, j) for c, j in zip(cases, jobs) if c is not None]
        kid = 1
        if args.batch <= 1:
            res = list(pool.map(run_emu, [(c, j[2], args.lib, kid) for c, j in keep]))
        else:   # batches of equal beam width
            res = [None] * len(keep)
            groups = {}
            for i, (c, j) in enumerate(keep):
                groups.setdefault(j[2], []).append(i)
            jobs2, where = [], []
            for W_, idx in groups.items():
                for b in range(0, len(idx), args.batch):
                    part = idx[b:b + args.batch]
                    jobs2.append(([keep[i][0] for i in part], W_, args.lib, kid, args.slots if args.slots > 0 else len(part))); where.append(part)
            for part, out in zip(where, pool.map(run_emu_batch, jobs2)):
                for i, o in zip(part, out):
                    res[i] = o
    bad = 0
    tot_t = 0.0
    for (c, j), (got, st, deferred, dt, upd) in zip(keep, res):
        tot_t += dt
        y1, y2, env, want, code = c
        if deferred:
            print("deferred", j); continue
        ok = (st == code) and (code != 0 or got == want)
        if not ok:
            bad += 1
            print("MISMATCH", j, "status", st, code, "len", len(got), len(want))
    print("emu check: %d cases, %d mismatches, %.1f s of emulation" % (len(keep), bad, tot_t))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
