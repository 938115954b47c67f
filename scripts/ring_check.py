"""Development check of beam2d_ring_kernel: route "ring" vs route "legacy" vs the oracle on synthetic pairs.
Usage: python scripts/ring_check.py [npairs] [T] [W] [time_n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np   # noqa: E402
from poreover_amd import batch, _lib   # noqa: E402
from poreover_amd.synth import synth_pair   # noqa: E402
from oracle import po_oracle as O   # noqa: E402  (the checker)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
T = int(sys.argv[2]) if len(sys.argv) > 2 else 600
W = int(sys.argv[3]) if len(sys.argv) > 3 else 5
tn = int(sys.argv[4]) if len(sys.argv) > 4 else 0
_lib.load()
y1s, y2s, envs, want = [], [], [], []
for i in range(n):
    Ti = T if i % 3 else max(60, T // 3)
    y1, y2 = synth_pair(1000 + i, T=Ti)
    r = O.pair_decode(y1, y2, "poreover", W, "row_col")
    y1s.append(y1); y2s.append(y2); envs.append(np.asarray(r["envelope"])); want.append(r["consensus"])
res = {}
for route in ("legacy", "ring"):
    _lib.set_pair_route(route)
    try:
        res[route] = batch.beam_search_2d_batch(y1s, y2s, envs, W, model="ctc", method="row_col")
    except Exception as e:   # noqa: BLE001
        print(route, "FAILED:", e)
        res[route] = [None] * n
_lib.set_pair_route("auto")
bad = 0
for i in range(n):
    for route in ("legacy", "ring"):
        g = res[route][i]
        if g != want[i]:
            bad += 1
            k = 0
            if g is not None:
                while k < min(len(g), len(want[i])) and g[k] == want[i][k]:
                    k += 1
            print("MISMATCH pair %d route %s: len %s vs %d, first difference at base %d (U=%d V=%d)" % (
                i, route, None if g is None else len(g), len(want[i]), k, len(y1s[i]), len(y2s[i])))
print("ring_check: %d pairs, T=%d, W=%d: %d mismatches" % (n, T, W, bad))
if tn:
    import ctypes as C
    lib = _lib.load()
    Y1 = [y1s[i % n] for i in range(tn)]; Y2 = [y2s[i % n] for i in range(tn)]; E = [envs[i % n] for i in range(tn)]
    for route in ("legacy", "ring", "legacy", "ring"):
        _lib.set_pair_route(route)
        lib.po_profile_enable(1); lib.po_profile_reset()
        t0 = time.time(); out = batch.beam_search_2d_batch(Y1, Y2, E, W, model="ctc", method="row_col"); dt = time.time() - t0
        ms = C.c_double(); cnt = C.c_int64(); lib.po_profile_get(_lib.K_BEAM2D, C.byref(ms), C.byref(cnt))
        ok = sum(1 for i in range(tn) if out[i] == want[i % n])
        print("%s: n=%d kernel %.1f ms -> %.0f pairs/s (%d/%d identical to the oracle)" % (route, tn, ms.value, tn / (ms.value / 1e3), ok, tn))
    _lib.set_pair_route("auto")
