"""debug: path statistics of beam2d_kernel's carried window maxima (needs the -DPO_B2_PATHSTAT build)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from poreover_amd import batch, _lib
from poreover_amd.synth import synth_pair
from oracle import po_oracle as O
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
base = []
for i in range(16):
    y1, y2 = synth_pair(i, T=4000)
    base.append((y1, y2, O.pair_decode(y1, y2, "poreover", 5, "row_col")["envelope"]))
y1s = [base[i % 16][0] for i in range(n)]; y2s = [base[i % 16][1] for i in range(n)]; envs = [base[i % 16][2] for i in range(n)]
d = torch.zeros(8, dtype=torch.int64, device="cuda")
lib.po_profile_update_counter(d.data_ptr())
batch.beam_search_2d_batch(y1s, y2s, envs, 5, model="ctc", method="row_col")
torch.cuda.synchronize()
v = d.cpu().tolist()
print("ref-schedule updates %d executed %d | lane exits: monotone %d full %d | main scans %d steady %d with-exit %d with-full-reread %d" % tuple(v))
