#!/usr/bin/env python3
"""Generate tests/golden/nw_matrix_golden.json by RUNNING the reference: the dense DP matrix align.global_pair returns as its
third item (align.pyx:34-52,98) for the 24 alignment cases of golden.json ("nw") and the 5 score cases of extra_golden.json
("align_scores"), as {shape, sha256 of the int32 row-major bytes, the four corner cells, the last row} — plus two small
matrices in full.

    python3 tests/golden/make_golden_nw_matrix.py
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402


def rec(mat):
    m = np.ascontiguousarray(np.asarray(mat), dtype=np.int32)
    return {"shape": list(m.shape), "sha256": hashlib.sha256(m.tobytes()).hexdigest(),
            "corners": [int(m[0, 0]), int(m[0, -1]), int(m[-1, 0]), int(m[-1, -1])], "last_row": m[-1].tolist()}


def main():
    MG.build_reference()
    _, align = MG.import_reference()
    G = {"nw": [], "align_scores": [], "small": []}
    with open(os.path.join(HERE, "golden.json")) as f:
        for c in json.load(f)["nw"]:
            G["nw"].append(rec(align.global_pair(c["s1"], c["s2"])[2]))
    with open(os.path.join(HERE, "extra_golden.json")) as f:
        for c in json.load(f)["align_scores"]:
            G["align_scores"].append(rec(align.global_pair(c["s1"], c["s2"], *c["scores"])[2]))
    for a, b, sc in (("ACGTACGTTT", "ACGTCGTTTA", (2, -1, -1)), ("GATTACA", "GCATGCT", (1, -1, -2)), ("A", "", (2, -1, -1)), ("", "", (2, -1, -1))):
        m = np.asarray(align.global_pair(a, b, *sc)[2], dtype=np.int32)
        G["small"].append({"s1": a, "s2": b, "scores": list(sc), "matrix": m.tolist()})
    with open(os.path.join(HERE, "nw_matrix_golden.json"), "w") as f:
        json.dump(G, f, separators=(",", ":"), sort_keys=True)
    print("wrote nw_matrix_golden.json", os.path.getsize(os.path.join(HERE, "nw_matrix_golden.json")))
    shutil.rmtree(MG.SCRATCH, ignore_errors=True)


if __name__ == "__main__":
    main()
