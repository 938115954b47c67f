#!/usr/bin/env python3
"""Generate tests/golden/extra_golden.json by RUNNING the reference: the API corners added in round 2 —
align.global_pair / global_pair_banded with non-default match / mismatch / gap_cost (align.pyx:29,100),
prefix_search_log_cy(return_forward=True) (prefix_search.py:176-238), decoding_cy.pair_prefix_prob_log(_from_vec)
and logsumexp on small inputs.

    python3 tests/golden/make_golden_extra.py
"""
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

sys.path.insert(0, MG.REPO)


def main():
    MG.build_reference()
    decoding, align = MG.import_reference()
    from poreover.decoding import prefix_search as ref_ps
    from poreover_amd.synth import synth_pair
    G = {}
    rng = np.random.default_rng(99)
    cases = []
    for n, scores in ((40, (1, -1, -2)), (90, (3, -2, -1)), (150, (2, -3, -3)), (300, (5, -4, -2)), (64, (1, 0, -1))):
        a = "".join("ACGT"[i] for i in rng.integers(4, size=n))
        b = "".join(ch for ch in a if rng.random() > 0.06)
        b = "".join(("ACGT"[rng.integers(4)] if rng.random() < 0.05 else ch) for ch in b)
        f1, f2, _ = align.global_pair(a, b, *scores)
        b1, b2 = align.global_pair_banded(a, b, 25, *scores)
        cases.append({"s1": a, "s2": b, "scores": list(scores), "full": ["".join(f1), "".join(f2)],
                      "banded25": ["".join(b1), "".join(b2)]})
    G["align_scores"] = cases
    y1, _ = synth_pair(40, T=400)
    fw = {}
    for lo, hi in ((0, 100), (100, 250)):
        lab, mat = ref_ps.prefix_search_log_cy(y1[lo:hi], return_forward=True)
        fw["%d_%d" % (lo, hi)] = {"label": lab, "shape": list(mat.shape), "matrix": [[MG.jf(x) for x in row] for row in mat]}
    G["prefix_return_forward"] = fw
    g = np.asarray(decoding.decoding_cy.pair_gamma_log(y1[:12], y1[20:30]))
    a1 = rng.normal(-3, 1, 12); a2 = rng.normal(-3, 1, 10)
    G["pair_prefix_prob"] = {"alpha1": [MG.jf(x) for x in a1], "alpha2": [MG.jf(x) for x in a2],
                             "from_vec": MG.jf(decoding.decoding_cy.pair_prefix_prob_log_from_vec(a1, a2, g)),
                             "outer": MG.jf(decoding.decoding_cy.pair_prefix_prob_log(np.add.outer(a1, a2), g))}
    # decoding_cy.pair_gamma_log_envelope (decoding_cy.pyx:224-271) on small banded pairs: the reference's own Cython,
    # its PySparseMatrix rows pushed as the tests of this repo push them, the cells in row-major order
    cyg = []
    for seed, U_, V_, half in ((7300, 14, 11, 3), (7301, 20, 23, 4), (7302, 9, 9, 2)):
        ya, yb = synth_pair(seed, T=40)
        ya, yb = np.ascontiguousarray(ya[:U_]), np.ascontiguousarray(yb[:V_])
        rows = [(max(0, int(u * V_ / U_) - half), min(V_, int(u * V_ / U_) + half)) for u in range(U_ + 1)]

        def fresh():
            m = decoding.decoding_cy.PySparseMatrix()
            for s_, e_ in rows:
                m.push_row(s_, e_)
            return m
        cells = np.array([(u, v) for u in range(U_ + 1) for v in range(rows[u][0], rows[u][1] + 1)], dtype=np.intp)
        gm = decoding.decoding_cy.pair_gamma_log_envelope(ya, yb, fresh(), cells, fresh(), fresh())
        cyg.append({"seed": seed, "U": U_, "V": V_, "half": half, "rows": [list(r) for r in rows],
                    "cells": cells.tolist(), "gamma": [MG.jf(gm.get(int(u), int(v))) for u, v in cells]})
    G["pair_gamma_cy_envelope"] = cyg
    with open(os.path.join(HERE, "extra_golden.json"), "w") as f:
        json.dump(G, f, indent=0, sort_keys=True)
    print("wrote extra_golden.json", os.path.getsize(os.path.join(HERE, "extra_golden.json")))
    shutil.rmtree(MG.SCRATCH, ignore_errors=True)


if __name__ == "__main__":
    main()
