"""Build libporeover_hip.so in-tree with hipcc for gfx950 (MI355X).  No JIT cache, no torch
extension machinery: the .so sits next to the sources so it travels with the repo snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libporeover_hip.so")
SOURCES = ["po_capi.hip", "po_viterbi.hip", "po_beam1d.hip", "po_beam2d.hip", "po_beam2d_reg.hip", "po_pair.hip", "po_lattice.hip", "po_prefix.hip", "po_ingest.hip", "po_gamma.hip", "po_stream.hip"]
HEADERS = ["po_device.h", os.path.join("..", "..", "include", "poreover_hip.h")]


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest(deps):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    os.makedirs(os.path.join(CSRC, "_obj"), exist_ok=True)
    for s in srcs:
        o = os.path.join(CSRC, "_obj", os.path.basename(s) + ".o")
        if force or not os.path.exists(o) or os.path.getmtime(o) < _newest([s] + deps[len(srcs):]):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                   "-Wno-unused-value", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
        objs.append(o)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
