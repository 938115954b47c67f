"""Step-by-step comparison of beam2d_reg_kernel with the oracle on ONE saved case (tests/golden/fuzz_cases/*.npz):
every candidate's score before every prune.
   on the GPU box:  POREOVER_HIP_LIB=scripts/variants/libporeover_hip_ringtrace.so python scripts/trace_rowcol.py gpu CASE > gpurun_out/trace_gpu.txt
   here (CPU):      python scripts/trace_rowcol.py cpu CASE gpurun_out/trace_gpu.txt
(the library: scripts/build_file_variant.sh po_beam2d_reg ringtrace -DPO_RING_TRACE)"""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode, case = sys.argv[1], sys.argv[2]
d = np.load(case, allow_pickle=True)
y1, y2, env = d["y1"], d["y2"], d["env"]
W, model, method = int(d["W"]), str(d["model"]), str(d["method"])
if mode == "gpu":
    from poreover_amd import _lib, batch
    _lib.set_pair_route("reg")
    got = batch.beam_search_2d_batch([y1], [y2], [env], W, model=model, method=method)
    sys.stdout.flush()
    print("RESULT", got[0])
else:
    so = "/tmp/st/libpooracle_trace.so"
    os.makedirs("/tmp/st", exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-ffp-contract=off", "-fPIC", "-shared", "-DPO_ORACLE_TRACE",
                           os.path.join(os.path.dirname(__file__), "..", "oracle", "po_oracle.c"), "-o", so, "-lm"])
    code = ("import os,sys,numpy as np; sys.path.insert(0,%r); os.environ['PO_ORACLE_SO']=%r; from oracle import po_oracle as O; "
            "d=np.load(%r,allow_pickle=True); print('RESULT', O.cpp_beam_search_2d(d['y1'],d['y2'],d['env'],int(d['W']),model_=str(d['model']),method_=str(d['method'])))"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), so, case))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout

    def parse(text):
        steps = {}
        order = []
        for ln in text.splitlines():
            if not ln.startswith("T "):
                continue
            f = ln.split()
            if len(f) != 5:
                print("malformed line:", ln[:80])
                continue
            _, u, v, nid, sc = f
            k = (int(u), int(v))
            if k not in steps:
                steps[k] = {}
                order.append(k)
            steps[k][int(nid)] = float(sc)
        return steps, order
    want, order = parse(out)
    got, _ = parse(open(sys.argv[3]).read())
    print("oracle steps", len(order), "gpu steps", len(got))
    for k in order:
        g = got.get(k)
        w = want[k]
        if g is None:
            print("step", k, "missing on the gpu"); break
        if g != w:
            print("first difference at step (u, v) =", k, "index", order.index(k))
            for nid in sorted(set(w) | set(g)):
                a, b = w.get(nid), g.get(nid)
                print("   node %6d  oracle %-24s gpu %-24s %s" % (nid, repr(a), repr(b), "" if a == b else "<--"))
            break
    else:
        print("all steps agree")
