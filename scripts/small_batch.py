"""Pair-beam kernel time at small and large batch sizes on chosen routes (the per-GPU shard of the strong-scaling job):
python scripts/small_batch.py [sizes...]   (PO_ROUTES=legacy,reg selects the routes, PO_MODEL=ctc | ctc_merge_repeats the tree model,
PO_W the beam width)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np   # noqa: E402
from poreover_amd import batch, _lib   # noqa: E402
from poreover_amd.synth import synth_pair   # noqa: E402
from oracle import po_oracle as O   # noqa: E402  (envelopes for this ad-hoc script)

sizes = [int(x) for x in sys.argv[1:]] or [1, 64, 256, 625, 1250, 2500]
routes = os.environ.get("PO_ROUTES", "legacy,reg").split(",")
model = os.environ.get("PO_MODEL", "ctc")
W = int(os.environ.get("PO_W", "5"))
lib = _lib.load()
nb = 16
base = []
for i in range(nb):
    y1, y2 = synth_pair(i, T=4000)
    base.append((y1, y2, O.pair_decode(y1, y2, "poreover", W, "row_col")["envelope"]))
want = [O.cpp_beam_search_2d(b[0], b[1], b[2], W, model_=model, method_="row_col") for b in base]
for n in sizes:
    y1s = [base[i % nb][0] for i in range(n)]; y2s = [base[i % nb][1] for i in range(n)]; envs = [base[i % nb][2] for i in range(n)]
    line = "n=%5d:" % n
    for route in routes:
        _lib.set_pair_route(route)
        best = None
        bad = 0
        for rep in range(3):
            lib.po_profile_enable(1); lib.po_profile_reset()
            got = batch.beam_search_2d_batch(y1s, y2s, envs, W, model=model, method="row_col")
            ms = C.c_double(); cnt = C.c_int64(); lib.po_profile_get(_lib.K_BEAM2D, C.byref(ms), C.byref(cnt))
            best = ms.value if best is None else min(best, ms.value)
            bad += sum(1 for i in range(n) if got[i] != want[i % nb])
        line += "  %s %.2f ms (%.0f pairs/s)%s" % (route, best, n / best * 1e3, "" if not bad else " MISMATCHES %d" % bad)
    _lib.set_pair_route("auto")
    print(line, flush=True)
