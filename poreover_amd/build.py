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
# Per-object compiler options (measured, round 6: profiles/r06_ab_compiler_flags.txt).  -amdgpu-use-amdgpu-trackers (the AMDGPU register-pressure
# trackers in the machine scheduler) is worth 1 % on the 32-slot pair kernel, costs 1.5 % on the 64-slot pair kernel and 5 % on beam2d_kernel's
# W = 25 class; in po_beam1d.hip it gives beam1d_wave_kernel 1.5 % and takes 4.5 % from beam1d_kernel (W = 25), so that file goes without.
# po_beam2d_reg.hip is compiled twice (PO_REG_TU: the 32-slot kernels + the C entry points | the 64-slot kernels), each object with what it
# runs best with.  (object name, source, extra options)
TRACKERS = ["-mllvm", "-amdgpu-use-amdgpu-trackers"]
OBJECTS = [(s, s, []) for s in SOURCES if s != "po_beam2d_reg.hip"] + [
    ("po_beam2d_reg.hip", "po_beam2d_reg.hip", ["-DPO_REG_TU=1"] + TRACKERS),
    ("po_beam2d_reg_wide.hip", "po_beam2d_reg.hip", ["-DPO_REG_TU=2"]),
]


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest(deps + [os.path.abspath(__file__)]):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    os.makedirs(os.path.join(CSRC, "_obj"), exist_ok=True)
    me = os.path.abspath(__file__)   # (the options live here)
    for oname, sname, extra in OBJECTS:
        s = os.path.join(CSRC, sname)
        o = os.path.join(CSRC, "_obj", oname + ".o")
        if force or not os.path.exists(o) or os.path.getmtime(o) < _newest([s, me] + deps[len(srcs):]):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                   "-Wno-unused-value", "-Wno-unused-function", *extra, "-c", s, "-o", o]
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
