"""Workload statistics of the row_col walk on the bench's synthetic pairs (design input for the pair-beam kernels).

Builds a scratch copy of the oracle with -DPO_ORACLE_STATS (counters inside beam2d_row_col) and decodes a few pairs.
Usage: python scripts/rowcol_stats.py [npairs] [T] [flipflop]
"""
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
so = "/tmp/st/libpooracle_stats.so"
os.makedirs("/tmp/st", exist_ok=True)
subprocess.check_call(["gcc", "-O2", "-std=c99", "-ffp-contract=off", "-fPIC", "-shared", "-DPO_ORACLE_STATS",
                       os.path.join(os.path.dirname(__file__), "..", "oracle", "po_oracle.c"), "-o", so, "-lm"])
os.environ["PO_ORACLE_SO"] = so
from oracle import po_oracle as O   # noqa: E402
from poreover_amd.synth import synth_pair   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
ff = len(sys.argv) > 3
lib = O.oracle_lib()
lib.oracle_rowcol_stats.restype = C.POINTER(C.c_longlong)
for i in range(n):
    y1, y2 = synth_pair(i, T=T, flipflop=ff)
    O.pair_decode(y1, y2, kind="flipflop" if ff else "poreover")
s = lib.oracle_rowcol_stats()
names = {0: "main steps", 1: "catch-ups", 2: "catch-ups beyond the last window", 3: "sum len0", 4: "sum len1",
         5: "window end moved back", 6: "new times read0", 7: "new times read1", 8: "prune: same beam",
         9: "prune: permutation", 10: "prune: set changed", 11: "entering nodes", 12: "re-entries (has children)",
         13: "re-entries with live stale values", 14: "frozen-parent beam nodes (node-steps)",
         15: "frozen claim violated", 16: "frozen value at done-1 exists", 17: "exact ties in beam", 18: "pairs",
         19: "max window", 20: "steps with >1 entering", 21: "pairs maxw<=14", 22: "pairs maxw<=30",
         23: "pairs maxw<=62", 24: "pairs maxw>62", 25: "nodes created"}
for k in sorted(names):
    print("%-45s %12d   per pair %.1f" % (names[k], s[k], s[k] / max(1, s[18])))
