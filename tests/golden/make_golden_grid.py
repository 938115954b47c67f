#!/usr/bin/env python3
"""Golden vectors for the hidden `grid` pair method (BeamSearch2.h:33-184), produced by RUNNING the reference
(same scratch build and import as make_golden.py; build container only) -> tests/golden/golden_grid.json.
Inputs are the committed CSV fixture (inputs.npz), the toy matrices of golden.json and seeded synthetic
pairs (poreover_amd.synth), so only the outputs are stored.

    python3 tests/golden/make_golden_grid.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402


def main():
    MG.build_reference()
    decoding, _ = MG.import_reference()
    from poreover_amd.synth import synth_pair
    G = {"script": "tests/golden/make_golden_grid.py"}
    inputs = np.load(os.path.join(HERE, "inputs.npz"))
    gold = json.load(open(os.path.join(HERE, "golden.json")))
    # the reference's own (commented-out) grid tests: the CSV fixture against itself in a +-10 band
    y = np.log(inputs["poreover_csv_prob"])
    T = len(y)
    env10 = np.array([(max(0, i - 10), min(i + 10, T)) for i in range(T)])
    G["csv_self_grid"] = {"W%d" % W: decoding.cpp_beam_search_2d(y, y, env10.tolist(), beam_width_=W, method_="grid")
                          for W in (5, 10)}
    # toy pairs of tests/test_prefix.py, no envelope
    pm = gold["prefix_prob"]
    toy = {}
    with np.errstate(divide="ignore"):
        for a, b in (("p1", "p2"), ("p1", "p3"), ("p2", "p3")):
            ya, yb = np.log(np.array(pm[a])), np.log(np.array(pm[b]))
            toy[a + "_" + b] = decoding.cpp_beam_search_2d(ya, yb, alphabet_="AB", beam_width_=5, method_="grid")
    G["toy_noenv_grid"] = toy
    # seeded synthetic pairs: (seed, T, flipflop, model, W, band or None)
    cases = []
    for seed, T_, ff, model, W, band in ((9001, 140, False, "ctc", 4, 6), (9002, 200, False, "ctc", 5, 9),
                                          (9003, 160, False, "ctc", 10, 7), (9004, 60, False, "ctc", 4, None),
                                          (9005, 50, False, "ctc_merge_repeats", 4, None),
                                          (9006, 50, True, "ctc_flipflop", 4, None), (9007, 300, False, "ctc", 25, 8)):
        y1, y2 = synth_pair(seed, T=T_, flipflop=ff)
        U, V = len(y1), len(y2)
        env = None
        if band is not None:
            env = np.array([(max(0, int(u * V / U) - band), min(V, int(u * V / U) + band)) for u in range(U)])
        out = decoding.cpp_beam_search_2d(y1, y2, None if env is None else env.tolist(), beam_width_=W, model_=model,
                                          method_="grid")
        cases.append({"seed": seed, "T": T_, "flipflop": ff, "model": model, "W": W, "band": band, "out": out})
    G["synthetic"] = cases
    json.dump(G, open(os.path.join(HERE, "golden_grid.json"), "w"), indent=1)
    print("wrote golden_grid.json:", {k: (len(v) if hasattr(v, "__len__") else v) for k, v in G.items()})


if __name__ == "__main__":
    main()
