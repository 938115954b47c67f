#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  First step at which the emulated ring kernel's candidate scores differ from the oracle's,
for one check_emu.py case:  python tools/simt_emu/trace_case.py SEED T W STYLE   (builds the -DPO_RING_TRACE emulation)"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
job = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
if len(sys.argv) > 5 and sys.argv[5] == "emu":
    import check_emu as cr
    c = cr.make_case(job)
    r = cr.run_emu((c, job[2], os.path.join(HERE, "_build_trace", "libemu_pair_beam.so"), 1 if os.environ.get("EMU_KERNEL") == "reg" else 0))
    sys.stdout.flush()
    print("\nRESULT", r[0])
    sys.exit(0)
if len(sys.argv) > 5 and sys.argv[5] == "oracle":
    import check_emu as cr
    c = cr.make_case(job)
    sys.stdout.flush()
    print("\nRESULT", c[3])
    sys.exit(0)
subprocess.check_call(["make", "-s", "-C", HERE, "OUT=" + os.path.join(HERE, "_build_trace"), "EMU_DEFS=-DPO_RING_TRACE"])
so = "/tmp/st/libpooracle_trace.so"
os.makedirs("/tmp/st", exist_ok=True)
subprocess.check_call(["gcc", "-O2", "-std=c99", "-ffp-contract=off", "-fPIC", "-shared", "-DPO_ORACLE_TRACE",
                       os.path.join(ROOT, "oracle", "po_oracle.c"), "-o", so, "-lm"])
emu = subprocess.run([sys.executable, __file__] + sys.argv[1:5] + ["emu"], capture_output=True, text=True, timeout=600)
env = dict(os.environ, PO_ORACLE_SO=so)
ora = subprocess.run([sys.executable, __file__] + sys.argv[1:5] + ["oracle"], capture_output=True, text=True, env=env, timeout=600)
if emu.returncode != 0:
    print("emulation failed:", emu.stderr[-2000:])


def parse(text):
    steps, order, res = {}, [], None
    for ln in text.splitlines():
        if ln.startswith("RESULT"):
            res = ln[7:]
        if not ln.startswith("T "):
            continue
        f = ln.split()
        if len(f) != 5:
            continue
        k = (int(f[1]), int(f[2]))
        if k not in steps:
            steps[k] = {}; order.append(k)
        steps[k][int(f[3])] = float(f[4])
    return steps, order, res


# (the oracle prints the trace of every beam search it runs: the pipeline's pair decode first, the explicit one last)
want, order, wres = parse(ora.stdout)
got, gorder, gres = parse(emu.stdout)
print("oracle steps", len(order), "emu steps", len(gorder), "strings equal:", wres == gres)
for k in order:
    g, w = got.get(k), want[k]
    if g is None:
        print("step", k, "missing in the emulation"); break
    if set(g) != set(w) or any(abs(g[i] - w[i]) > 1e-9 * max(1.0, abs(w[i])) and not (g[i] == w[i]) for i in w):
        print("first difference at step (u, v) =", k, "index", order.index(k))
        for nid in sorted(set(w) | set(g)):
            a, b = w.get(nid), g.get(nid)
            print("   node %6d  oracle %-24s emu %-24s %s" % (nid, repr(a), repr(b), "" if a == b else "<--"))
        break
else:
    print("all steps agree")
