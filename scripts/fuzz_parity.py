#!/usr/bin/env python3
"""Randomised differential test of the pair beam kernels against the CPU oracle (test infrastructure), beyond what
tests/ pins: python scripts/fuzz_parity.py [--seconds 120] [--seed 1] [--procs 16].  Needs an MI355X.
Every round draws a configuration (model, method, beam width, envelope style, lengths), decodes a batch on the GPU
through the C-ABI and the same pairs with the oracle on the host cores, and compares strings and statuses."""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def jagged(rng, U, V, style, pad):
    """envelopes unlike the pipeline's: stairs / wobble / bursts around the diagonal (monotone row starts for `row`/`grid`)"""
    env = np.zeros((U, 2), dtype=np.int64)
    c = 0.0
    for u in range(U):
        if style == "stairs":
            c = (u // 7) * 7 * V / U
        elif style == "wobble":
            c = u * V / U + 3.0 * np.sin(u / 5.0)
        else:
            c = u * V / U + (rng.integers(-2, 3) if u % 11 == 0 else 0)
        w = pad + (rng.integers(0, pad + 1) if style == "bursts" and u % 13 == 0 else 0)
        env[u] = (max(0, int(c) - w), min(V, int(c) + w + 1))
    env[:, 0] = np.maximum.accumulate(env[:, 0])
    env[:, 1] = np.maximum(env[:, 1], env[:, 0] + 1).clip(max=V)
    return env


def write_summary(args, mode, **counts):
    """profiles/rNN_fuzz_*.json: what ran, on which binary, with what result (the verdict of round 3 asked for artefacts)"""
    if not args.json:
        return
    import hashlib
    import json
    from poreover_amd import _lib
    lib_path = os.environ.get("POREOVER_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), "libporeover_hip.so")
    try:
        sha = hashlib.sha256(open(lib_path, "rb").read()).hexdigest()
    except OSError:
        sha = None
    try:
        chain = _lib.get_chain_mode()   # (serial unless PO_CHAIN_CLOSED / po_set_chain_mode says otherwise: round 6)
    except Exception:
        chain = None
    rec = dict(mode=mode, seed=args.seed, seconds=args.seconds, route=args.route, focus=bool(args.focus), git_head=args.head, chain_mode=chain,
               lib=os.path.relpath(lib_path), lib_sha256=sha, when=time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), **counts)
    os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
    prev = []
    if os.path.exists(args.json):
        try:
            prev = json.load(open(args.json))
        except ValueError:
            prev = []
    with open(args.json, "w") as f:
        json.dump(prev + [rec], f, indent=1)


def _oracle_one(job):
    from oracle import po_oracle as O
    y1, y2, env, W, model, method = job
    try:
        return O.cpp_beam_search_2d(y1, y2, env, W, model_=model, method_=method), 0
    except O.OracleError as e:
        return "", e.code


def _oracle_pipeline(job):
    from oracle import po_oracle as O
    y1, y2, kind, W, method = job
    try:
        r = O.pair_decode(y1, y2, kind, W, method)
        return (r["status"], r["seq1"], r["seq2"], r.get("consensus"), None if r.get("envelope") is None else
                np.asarray(r["envelope"]).tolist(), r.get("sequence_identity"))
    except O.OracleError as e:
        return (e.code, None, None, None, None, None)


def pipeline_mode(args, pool):
    """the whole stage chain (Viterbi x2, banded NW, skips, envelope, pair beam) against the oracle's"""
    from poreover_amd import batch, _lib as _L
    from poreover_amd.synth import synth_pair
    _L.load()
    _L.set_pair_route(args.route)
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    rounds = pairs = bad = 0
    while time.time() < t_end:
        kind = ["poreover", "poreover", "bonito", "flipflop"][rng.integers(4)]
        method = ["row_col", "row_col", "row"][rng.integers(3)]
        W = int([3, 5, 5, 5, 8, 10][rng.integers(6)])
        n = int(rng.integers(4, 48))
        y1s, y2s = [], []
        for i in range(n):
            y1, y2 = synth_pair(int(rng.integers(1 << 30)), T=int(rng.integers(40, 2500)), flipflop=(kind == "flipflop"))
            if rng.random() < 0.1:
                y2 = y2[: max(2, len(y2) // 2)]
            y1s.append(y1); y2s.append(y2)
        want = pool.map(_oracle_pipeline, [(a, b, kind, W, method) for a, b in zip(y1s, y2s)])
        from poreover_amd import _lib
        try:
            got = batch.pair_decode_batch(y1s, y2s, kind, W, method)
        except _lib.EngineError:    # some pair is refused (e.g. the reference's frame-map assertion): one by one
            got = []
            for a, b in zip(y1s, y2s):
                try:
                    got.append(batch.pair_decode_batch([a], [b], kind, W, method)[0])
                except _lib.EngineError as e:
                    got.append({"status": e.code, "seq1": None, "seq2": None})
        for i, (w, g) in enumerate(zip(want, got)):
            ok = g["status"] == w[0]
            if ok and w[1] is not None and g["seq1"] is not None:
                ok = (g["seq1"], g["seq2"]) == (w[1], w[2])
            if ok and w[0] == 0:
                ok = g["consensus"] == w[3] and np.asarray(g["envelope"]).tolist() == w[4] and g["sequence_identity"] == w[5]
            if g["status"] == -4 and w[0] == 0:
                ok = True
            if not ok:
                bad += 1
                print("MISMATCH", dict(kind=kind, method=method, W=W, U=len(y1s[i]), V=len(y2s[i]), status=(g["status"], w[0])),
                      flush=True)
        rounds += 1; pairs += n
    pool.terminate()
    print("fuzz pipeline: %d rounds, %d pairs, %d mismatches" % (rounds, pairs, bad))
    write_summary(args, "pipeline", rounds=rounds, pairs=pairs, mismatches=bad)
    sys.exit(1 if bad else 0)


def _oracle_oned(job):
    from oracle import po_oracle as O
    y, kind, model, W = job
    seq, path = O.viterbi_decode(y, kind)
    beam = O.cpp_beam_search(y, W, model_=model)
    lab = beam[: max(1, min(len(beam), 60))] if beam else "A"
    fwd = O.cpp_forward(y, lab, model_=model)
    acc = O.cpp_viterbi_acceptor(y, seq, 1000).tolist() if (kind == "poreover" and seq) else None
    return seq, [int(x) for x in path], beam, lab, fwd, acc


def oned_mode(args, pool):
    """the 1-D entry points: Viterbi (three kinds), CTC beam search (three models), forward, Viterbi acceptor"""
    from poreover_amd import batch
    from poreover_amd.synth import synth_pair
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    rounds = reads = bad = 0
    while time.time() < t_end:
        kind, model, ff = [("poreover", "ctc", False), ("bonito", "ctc_merge_repeats", False),
                           ("flipflop", "ctc_flipflop", True)][rng.integers(3)]
        W = int([1, 2, 3, 5, 8, 10, 16, 25, 40][rng.integers(9)])
        n = int(rng.integers(4, 40))
        ys = []
        for i in range(n):
            T = int(rng.integers(1, 40)) if rng.random() < 0.15 else int(rng.integers(40, 3000))
            y1, _ = synth_pair(int(rng.integers(1 << 30)), T=max(T, 12), flipflop=ff)
            ys.append(y1[:T])
        want = pool.map(_oracle_oned, [(y, kind, model, W) for y in ys])
        seqs, paths = batch.viterbi_batch(ys, kind, return_path=True)
        beams = batch.beam_search_batch(ys, W, model=model)
        fwd = batch.forward_batch(ys, [w[3] for w in want], model=model)
        for i, w in enumerate(want):
            ok = seqs[i] == w[0] and [int(x) for x in paths[i]] == w[1] and beams[i] == w[2]
            f, g = w[4], float(fwd[i])
            ok = ok and ((f == g) or (np.isfinite(f) and abs(f - g) <= 1e-12 * max(1.0, abs(f))) or (np.isinf(f) and np.isinf(g)))
            if ok and w[5] is not None:
                ok = batch.viterbi_acceptor_batch([ys[i]], [w[0]], 1000)[0].tolist() == w[5]
            if not ok:
                bad += 1
                print("MISMATCH", dict(kind=kind, model=model, W=W, T=len(ys[i]), seq=(seqs[i][:20], w[0][:20]),
                                        beam=(beams[i][:20], w[2][:20]), fwd=(g, f)), flush=True)
        rounds += 1; reads += n
    pool.terminate()
    print("fuzz 1-D: %d rounds, %d reads, %d mismatches" % (rounds, reads, bad))
    write_summary(args, "oned", rounds=rounds, reads=reads, mismatches=bad)
    sys.exit(1 if bad else 0)


def draw_round(rng, focus):
    """one round of the pair mode: a configuration and its batch (shared with tools/simt_emu, which replays the same rounds)"""
    from oracle import po_oracle as O
    from poreover_amd.synth import synth_pair
    model, ff = [("ctc", False), ("ctc", False), ("ctc_merge_repeats", False), ("ctc_flipflop", True)][rng.integers(4)]
    method = ["row_col", "row_col", "row_col", "row", "grid"][rng.integers(5)]
    W = int([1, 2, 3, 4, 5, 5, 5, 6, 7, 9, 10, 12, 13, 16, 25][rng.integers(15)])
    if focus:   # what beam2d_reg_kernel serves: row_col, every tree model, W <= 12 (both lane layouts)
        method = "row_col"
        W = int([1, 2, 3, 4, 5, 5, 5, 5, 6, 7, 9, 10, 12][rng.integers(13)])
    # (grid with the other models in a narrow band: every score -inf — the reference's own order is heap-address
    #  order there; engine and oracle both replay libstdc++ on creation order and agree)
    tmax = 260 if method == "grid" else 1400
    n = int(rng.integers(4, 40))
    style = ["pipeline", "diag", "stairs", "wobble", "bursts"][rng.integers(5)]
    if focus and rng.random() < 0.5:
        n = int(rng.integers(40, 400))
    y1s, y2s, envs = [], [], []
    for i in range(n):
        T = int(rng.integers(30, tmax))
        y1, y2 = synth_pair(int(rng.integers(1 << 30)), T=T, flipflop=ff)
        if rng.random() < 0.15:
            y2 = y2[: max(2, (2 * len(y2)) // 3)]
        U, V = len(y1), len(y2)
        if style == "pipeline":
            kind = {"ctc": "poreover", "ctc_merge_repeats": "bonito", "ctc_flipflop": "flipflop"}[model]
            try:
                env = np.asarray(O.pair_decode(y1, y2, kind, 5, "row_col")["envelope"])
            except Exception:
                env = np.asarray(O.diagonal_envelope(U, V, 12))
            if env is None or len(env) != U:
                env = np.asarray(O.diagonal_envelope(U, V, 12))
        elif style == "diag":
            env = np.asarray(O.diagonal_envelope(U, V, int(rng.integers(3, 30))))
        else:
            env = jagged(rng, U, V, style, int(rng.integers(2, 14)))
        y1s.append(y1); y2s.append(y2); envs.append(env)
    return model, method, W, style, y1s, y2s, envs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--oned", action="store_true", help="fuzz the 1-D entry points (Viterbi, beam search, forward, acceptor)")
    ap.add_argument("--pipeline", action="store_true", help="fuzz pair_decode_batch (the whole stage chain) instead")
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--procs", type=int, default=min(32, os.cpu_count() or 1))
    ap.add_argument("--route", default="auto", help="pin the pair beam kernel: auto | legacy | reg")
    ap.add_argument("--focus", action="store_true", help="pair mode: only what the register-state kernel serves "
                    "(ctc, row_col, W <= 6, monotone envelope styles)")
    ap.add_argument("--json", default="", help="append a summary record to this JSON file (profiles/rNN_fuzz_*.json)")
    ap.add_argument("--dump", default="", help="pair mode: save every round's inputs here before the GPU call (a GPU fault kills "
                    "the process: the file then holds the round that did it)")
    ap.add_argument("--head", default="", help="git head of the tree under test (the GPU box has no .git)")
    args = ap.parse_args()
    from oracle import po_oracle as O
    O.build()
    from poreover_amd.synth import synth_pair
    pool = mp.get_context("fork").Pool(args.procs)   # before the GPU runtime is initialised
    if args.pipeline:
        return pipeline_mode(args, pool)
    if args.oned:
        return oned_mode(args, pool)
    from poreover_amd import batch, _lib
    _lib.load()
    _lib.set_pair_route(args.route)
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    rounds = pairs = bad = refused = 0
    by_style = {}
    while time.time() < t_end:
        model, method, W, style, y1s, y2s, envs = draw_round(rng, args.focus)
        n = len(y1s)
        if args.dump:   # the round about to run, for a replay should the GPU fault (a fault kills the process)
            os.makedirs(os.path.dirname(os.path.abspath(args.dump)), exist_ok=True)
            np.savez_compressed(args.dump, model=model, method=method, W=W, style=style, round=rounds, seed=args.seed,
                                **{"y1_%d" % i: a for i, a in enumerate(y1s)}, **{"y2_%d" % i: a for i, a in enumerate(y2s)},
                                **{"env_%d" % i: a for i, a in enumerate(envs)})
            print("round", rounds, dict(model=model, method=method, W=W, style=style, n=n), flush=True)
        want = pool.map(_oracle_one, [(a, b, e, W, model, method) for a, b, e in zip(y1s, y2s, envs)])
        got, st = batch.beam_search_2d_batch(y1s, y2s, envs, W, model=model, method=method, return_status=True)
        for i, ((ws, wc), g, c) in enumerate(zip(want, got, st.tolist())):
            ok = (c == wc) and (wc != 0 or g == ws)
            if wc == 0 and (c == -4 or (c == -6 and method != "row_col")):   # capacity / unsupported-envelope refusals
                ok = True
                refused += 1
                print("refused", dict(model=model, method=method, W=W, style=style, U=len(y1s[i]), V=len(y2s[i]), status=c,
                                      widest_row=int((envs[i][:, 1] - envs[i][:, 0]).max())), flush=True)
            if not ok:
                bad += 1
                print("MISMATCH", dict(model=model, method=method, W=W, style=style, U=len(y1s[i]), V=len(y2s[i]),
                                        status=(c, wc), got=g[:40], want=ws[:40], round=rounds, index=i, n=n), flush=True)
                if bad <= 8:   # the case itself, for a replay (scripts/fuzz_replay.py)
                    os.makedirs("gpurun_out", exist_ok=True)
                    np.savez_compressed("gpurun_out/fuzz_fail_%d_%d.npz" % (args.seed, bad), y1=y1s[i], y2=y2s[i], env=envs[i], W=W,
                                        model=model, method=method, got=g, want=ws, status=np.array([c, wc]), index=i, n=n)
        rounds += 1; pairs += n
        by_style[style] = by_style.get(style, 0) + n
    pool.terminate()
    print("fuzz: %d rounds, %d pairs, %d mismatches, %d refused by the engine (PO_E_NOMEM / PO_E_UNSUPPORTED) where the "
          "oracle decodes" % (rounds, pairs, bad, refused))
    write_summary(args, "pair_beam", rounds=rounds, pairs=pairs, mismatches=bad, refused=refused, pairs_by_envelope_style=by_style)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
