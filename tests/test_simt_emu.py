"""CPU: the pair-beam kernels' LOGIC without a GPU.  tools/simt_emu compiles the product's own kernel sources
(po_beam2d_pre.h, po_beam2d_ring.hip, po_beam2d_reg.hip) with g++ against a lane-by-lane emulation of the HIP execution
model (one fibre per lane, cross-lane operations as rendezvous points) and runs them on small pairs; the strings must be the
oracle's.  Both lane schedules (ascending / descending) must agree: a result that changes with the schedule means a kernel
relies on lockstep execution across an LDS hand-over without a fence.  This is test infrastructure around the SAME sources
the GPU library is built from — the `-m gpu` suite stays the parity gate on real hardware."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(REPO, "tools", "simt_emu")


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "-C", EMU])
    return os.path.join(EMU, "_build", "libemu_pair_beam.so")


@pytest.mark.parametrize("kernel", ["reg", "ring"])
@pytest.mark.parametrize("sched", ["0", "1"])
def test_emulated_kernel_matches_oracle(emu_lib, oracle, kernel, sched):
    env = dict(os.environ, EMU_SCHED=sched)
    out = subprocess.run([sys.executable, os.path.join(EMU, "check_ring.py"), "--n", "18", "--T", "320", "--W", "0", "--seed", "31",
                          "--procs", "4", "--kernel", kernel, "--styles", "pipeline,stairs,wobble", "--lib", emu_lib],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert "0 mismatches" in out.stdout
