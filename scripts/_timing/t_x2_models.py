import sys, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo") else ".")
import numpy as np
from poreover_amd import batch, _lib
from poreover_amd.synth import synth_pair
from oracle import po_oracle as O
_lib.load()
for kind, model, ff in (("bonito", "ctc_merge_repeats", False), ("flipflop", "ctc_flipflop", True)):
    y1s, y2s, envs, want = [], [], [], []
    for i in range(8):
        y1, y2 = synth_pair(7000 + i, T=300 + 50 * i, flipflop=ff)
        env = O.pair_decode(y1, y2, kind, 5, "row_col")["envelope"]
        y1s.append(y1); y2s.append(y2); envs.append(env)
        want.append(O.cpp_beam_search_2d(y1, y2, env, 5, model_=model, method_="row_col"))
    got = batch.beam_search_2d_batch(y1s, y2s, envs, 5, model=model, method="row_col")
    print(kind, "parity:", got == want)
