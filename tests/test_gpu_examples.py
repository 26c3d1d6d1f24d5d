"""The reference's example programs re-hosted on the device backend (examples/*.py), run as programs on its own data."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from gtsam_petercdev_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run(script, *args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script), *args], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    return out.stdout


def number(text, label):
    return float(re.search(re.escape(label) + r"\s*([-+0-9.eE]+)", text).group(1))


def test_Pose3SLAMExample_g2o(tmp_path):
    """examples/Pose3SLAMExample_g2o.cpp on pose3example.txt: the errors the reference's binary prints (SURVEY §6.2:
    64 941.32 -> 19 130.66), and a g2o file of the result that reads back."""
    outf = tmp_path / "out3.g2o"
    text = run("Pose3SLAMExample_g2o.py", os.path.join(ROOT, "tests", "golden", "pose3example.txt"), str(outf))
    assert abs(number(text, "initial error=") - 64941.3) < 0.1 and abs(number(text, "final error=") - 19130.7) < 0.1
    back = _lib.read_g2o(str(outf), is3D=True)
    assert back.n_vars == 5 and back.n_factors == 6 + 1


def test_Pose2SLAMExample_g2o(tmp_path, golden_dir):
    """examples/Pose2SLAMExample_g2o.cpp: default file, an output file, an iteration cap and the robust kernels."""
    text = run("Pose2SLAMExample_g2o.py")
    assert "Adding prior on pose 0" in text and "Value 0: (gtsam::Pose2)" in text
    e0, e1 = number(text, "initial error="), number(text, "final error=")
    assert e1 < 0.5 * e0                                    # (noisyToyGraph: the residual floor is its noise)
    outf = tmp_path / "out2.g2o"
    f = os.path.join(golden_dir, "pose2example.txt")
    plain = run("Pose2SLAMExample_g2o.py", f, str(outf), "20")
    assert "User required to perform maximum  20 iterations" in plain and "done!" in plain
    assert number(plain, "final error=") < number(plain, "initial error=")
    back = _lib.load2d(str(outf), noise_format=0)
    assert back.n_vars == _lib.load2d(f, noise_format=0).n_vars
    for kernel in ("huber", "tukey"):
        robust = run("Pose2SLAMExample_g2o.py", f, str(tmp_path / f"out_{kernel}.g2o"), "20", kernel)
        assert f"Using robust kernel: {kernel}" in robust
        assert number(robust, "final error=") <= number(plain, "final error=") + 1e-9   # a robust loss never exceeds 1/2 r^2


def test_SFMExample_bal(tmp_path):
    """examples/SFMExample_bal.cpp on dubrovnik-3-7-pre: 7 tracks on 3 cameras, LM to the noise floor; the written BAL file
    holds the optimum (re-running on it ends at the same floor)."""
    outf = tmp_path / "dub_opt.txt"
    text = run("SFMExample_bal.py", os.path.join(ROOT, "tests", "golden", "dubrovnik-3-7-pre.txt"), str(outf))
    assert "read 7 tracks on 3 cameras" in text
    e = number(text, "final error:")
    assert 0.0 < e < 0.05        # (0.0199833 without the two priors: tests/testGeneralSFMFactorB.cpp:44-63)
    again = run("SFMExample_bal.py", str(outf))
    # (the two priors of the re-run sit at the optimum itself, so its floor is a little lower; the reader keeps floats)
    assert 0.5 * e < number(again, "final error:") <= e + 1e-6
