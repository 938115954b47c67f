"""Replay of the cases scripts/fuzz_parity.py saved (gpurun_out/fuzz_fail_*.npz) on every route of the pair beam search:
   python scripts/fuzz_replay.py gpurun_out/fuzz_fail_21_1.npz ..."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import po_oracle as O   # the checker of this ad-hoc script
from poreover_amd import _lib, batch

O.build()
for f in sys.argv[1:]:
    d = np.load(f, allow_pickle=True)
    y1, y2, env = d["y1"], d["y2"], d["env"]
    W, model, method = int(d["W"]), str(d["model"]), str(d["method"])
    kind = {"ctc": "poreover", "ctc_merge_repeats": "bonito", "ctc_flipflop": "flipflop"}[model]
    want = O.cpp_beam_search_2d(y1, y2, env, W, model_=model, method_=method)
    print(f, dict(W=W, model=model, method=method, U=len(y1), V=len(y2)), "saved want == oracle now:", str(d["want"]) == want)
    for route in ("auto", "reg", "legacy"):
        try:
            _lib.set_pair_route(route)
            got, st = batch.beam_search_2d_batch([y1], [y2], [env], W, model=model, method=method, return_status=True)
            same = got[0] == want
            first = next((k for k, (a, b) in enumerate(zip(got[0], want)) if a != b), min(len(got[0]), len(want)))
            print("  route %-6s status %d  %s  len %d / %d  first difference at %d" % (route, int(st[0]), "OK" if same else "MISMATCH", len(got[0]), len(want), first))
        except Exception as e:
            print("  route %-6s refused: %s" % (route, e))
    _lib.set_pair_route("auto")
