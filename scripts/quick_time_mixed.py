"""Pair beam kernel on pairs of very different lengths (T from 400 to 8000, log-uniform, input order random): what the
longest-first queue order (pair_order_kernel; PO_B2_NO_ORDER=1 switches it off) is for.  python scripts/quick_time_mixed.py [n]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import po_oracle as O   # only to build envelopes for this ad-hoc script
from poreover_amd import _lib, batch
from poreover_amd.synth import synth_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(3)
nb = 48
base = []
for i in range(nb):
    T = int(np.exp(rng.uniform(np.log(400), np.log(8000))))
    y1, y2 = synth_pair(i, T=T)
    base.append((y1, y2, O.pair_decode(y1, y2, "poreover", 5, "row_col")["envelope"]))
idx = rng.integers(nb, size=n)
y1s = [base[i][0] for i in idx]; y2s = [base[i][1] for i in idx]; envs = [base[i][2] for i in idx]
lib = _lib.load()
for rep in range(3):
    lib.po_profile_enable(1); lib.po_profile_reset()
    t0 = time.time(); out = batch.beam_search_2d_batch(y1s, y2s, envs, W, model="ctc", method="row_col"); dt = time.time() - t0
    ms = C.c_double(); cnt = C.c_int64(); lib.po_profile_get(_lib.K_BEAM2D, C.byref(ms), C.byref(cnt))
    print("mixed lengths W=%d n=%d: kernel %.1f ms -> %.0f pairs/s, %.2f Mframes/s" % (W, n, ms.value, n / (ms.value / 1e3), sum(len(a) + len(b) for a, b in zip(y1s, y2s)) / ms.value / 1e3))
# the whole stage chain on the same pairs (Viterbi x2, banded alignment + envelope, pair beam): stage milliseconds
for rep in range(2):
    lib.po_profile_enable(1); lib.po_profile_reset()
    out = batch.pair_decode_batch(y1s, y2s, "poreover", W, "row_col")
    st = {}
    for name, k in (("viterbi", _lib.K_VITERBI), ("align+envelope", _lib.K_ALIGN), ("pair beam", _lib.K_BEAM2D)):
        ms = C.c_double(); cnt = C.c_int64(); lib.po_profile_get(k, C.byref(ms), C.byref(cnt)); st[name] = round(ms.value, 2)
    print("chain:", st)
