"""Quick timing of the pair beam kernel through the host-buffer entry point (H2D included)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poreover_amd import batch, _lib
from poreover_amd.synth import synth_pair
from oracle import po_oracle as O   # only to build envelopes for this ad-hoc script
_lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
kind = sys.argv[3] if len(sys.argv) > 3 else "poreover"      # poreover | bonito | flipflop
model = {"poreover": "ctc", "bonito": "ctc_merge_repeats", "flipflop": "ctc_flipflop"}[kind]
method = sys.argv[4] if len(sys.argv) > 4 else "row_col"    # row_col | row
nb = 16
base = []
for i in range(nb):
    y1, y2 = synth_pair(i, T=4000, flipflop=(kind == "flipflop"))
    base.append((y1, y2, O.pair_decode(y1, y2, kind, 5, "row_col")["envelope"]))
y1s = [base[i % nb][0] for i in range(n)]; y2s = [base[i % nb][1] for i in range(n)]; envs = [base[i % nb][2] for i in range(n)]
lib = _lib.load()
if os.environ.get("PO_ROUTE"): _lib.set_pair_route(os.environ["PO_ROUTE"])   # auto | x2 | legacy
import ctypes as C
for rep in range(2):
    lib.po_profile_enable(1); lib.po_profile_reset()
    t0 = time.time(); out = batch.beam_search_2d_batch(y1s, y2s, envs, W, model=model, method=method); dt = time.time() - t0
    ms = C.c_double(); cnt = C.c_int64(); lib.po_profile_get(_lib.K_BEAM2D, C.byref(ms), C.byref(cnt))
    bases = sum(len(s) for s in out)
    print("beam2d " + method + " " + model + " W=%d n=%d: call %.3f s; kernel %.1f ms (%d launch) -> %.0f pairs/s kernel-only, %.3f Mbases/s" % (
        W, n, dt, ms.value, cnt.value, n / (ms.value / 1e3), bases / (ms.value / 1e3) / 1e6))
