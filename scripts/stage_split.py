"""Pair beam stage of n pairs split into its main kernel and everything around it (pre-pass, walk, fallback pass, launch
gaps):  python scripts/stage_split.py 1250   (ad-hoc; the oracle only builds the envelopes)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poreover_amd import batch, _lib
from poreover_amd.synth import synth_pair
from oracle import po_oracle as O
lib = _lib.load()
n = int(sys.argv[1])
base = []
for i in range(8):
    y1, y2 = synth_pair(i, T=4000)
    base.append((y1, y2, O.pair_decode(y1, y2, "poreover", 5, "row_col")["envelope"]))
y1s = [base[i % 8][0] for i in range(n)]; y2s = [base[i % 8][1] for i in range(n)]; envs = [base[i % 8][2] for i in range(n)]
for rep in range(3):
    lib.po_profile_enable(1); lib.po_profile_reset()
    batch.beam_search_2d_batch(y1s, y2s, envs, 5, model="ctc", method="row_col")
    out = {}
    for name, k in (("stage", _lib.K_BEAM2D), ("main kernel", _lib.K_BEAM2D_MAIN)):
        ms = C.c_double(); cnt = C.c_int64(); lib.po_profile_get(k, C.byref(ms), C.byref(cnt)); out[name] = round(ms.value, 3)
    print(n, out)
