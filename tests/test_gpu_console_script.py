"""GPU: the installable form (setup.cfg / pyproject.toml, round 6).  The reference installs ONE console script, `poreover`
(/root/reference setup.py:26 -> poreover.__main__:main); this package installs `poreover-amd` and the drop-in alias `poreover`.
`pip install` into a scratch directory (no network, no build isolation), then the installed script decodes the reference's
CSV table (tests/poreover.csv upstream, a golden fixture here) and the FASTA file equals what the reference wrote."""
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _install(target):
    out = subprocess.run([sys.executable, "-m", "pip", "install", "--no-build-isolation", "--no-deps", "--no-index", "--target", str(target), REPO],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    subprocess.run(["rm", "-rf", os.path.join(REPO, "build"), os.path.join(REPO, "poreover_amd.egg-info")])
    return out


def test_console_scripts_install_and_show_the_reference_cli(tmp_path):
    """CPU part (also runs under -m gpu): both scripts exist and print the decode / pair-decode sub-commands"""
    _install(tmp_path / "site")
    env = dict(os.environ, PYTHONPATH=str(tmp_path / "site"))
    for name in ("poreover-amd", "poreover"):
        script = tmp_path / "site" / "bin" / name
        assert script.exists(), os.listdir(tmp_path / "site")
        out = subprocess.run([sys.executable, str(script), "pair-decode", "--help"], capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=120)
        assert out.returncode == 0, out.stderr[-1000:]
        for flag in ("--basecaller", "--reverse_complement", "--beam_width", "--padding", "--skip_matches", "--alignment"):
            assert flag in out.stdout
    assert (tmp_path / "site" / "poreover_amd" / "libporeover_hip.so").exists()      # the HIP library travels as package data


@pytest.mark.gpu
def test_installed_script_decodes(tmp_path, golden, golden_inputs):
    from poreover_amd.decoding import decode
    _install(tmp_path / "site")
    prob = golden_inputs["poreover_csv_prob"]
    csv = tmp_path / "poreover.csv"
    with open(csv, "w") as f:
        f.write("A,C,G,T,\n")
        np.savetxt(f, prob, delimiter=",", fmt="%.18e")
    env = dict(os.environ, PYTHONPATH=str(tmp_path / "site"))
    for algo, want in (("viterbi", golden["csv"]["viterbi"]), ("beam", golden["csv"]["beam_w25"])):
        out = subprocess.run([sys.executable, str(tmp_path / "site" / "bin" / "poreover"), "decode", str(csv), "--out", str(tmp_path / algo),
                              "--algorithm", algo], capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        assert open(str(tmp_path / algo) + ".fasta").read() == decode.fasta_format("poreover", want) + "\n"
