"""Race hygiene: kernels must give the same bits when ANOTHER process's waves share the CUs.  A kernel with a hand-off that only
holds while the waves of a block run in step passes every single-process test and fails a few times per thousand launches next to a
neighbour (profiles/r05_race_halo_prologue.md: the halo-form K loop's first iteration, found this way).  The test starts a second
copy of scripts/probes/conv_hold.py on the same GPU and runs the hold loop itself: one layer per kernel family (halo form 3x3,
per-tap 1x1, small-grid, strided, concat), forward + statistics, backward-data and backward-weight on fixed operands; every launch's
checksum must equal the first launch's."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAYERS = ["dec4.c2", "dec3.c1", "l1.conv2", "l1.conv3", "l3.conv2", "l2.0.conv2", "dec2.c1"]


def test_convolution_kernels_are_bit_stable_next_to_a_second_process():
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "probes", "conv_hold.py"), "800"] + LAYERS
    procs = [subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for _ in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out += "\nTIMEOUT"
        outs.append(out)
    for p, out in zip(procs, outs):
        lines = [ln for ln in out.splitlines() if "launches that differ" in ln]
        assert p.returncode == 0 and len(lines) == len(LAYERS), out[-2000:]
        assert not any("DIFFERS" in ln for ln in lines), "\n".join(lines)
