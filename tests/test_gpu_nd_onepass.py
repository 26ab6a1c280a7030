"""GPU: the opt-in one-pass form of the N-d reduction tile (``PTHIP_ND_ONEPASS=1``: the kernel's last workgroup per output
tile folds the splits through self-validating pairs and per-tile tickets, codegen_tile.tile_reduce_source ``finish``).
Not the default (profiles/r8_nd_onepass.txt: no faster than the second launch it replaces) but kept correct: the golden
reduction cases run under it in a subprocess — the switch is read at import — against the reference C linker's outputs,
twice per executable (the pairs a launch consumed must be clean for the next one)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_golden_reductions_under_the_one_pass_form():
    env = {**os.environ, "PTHIP_ND_ONEPASS": "1"}
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "test_golden_case and (careduce or layout_fuzz_f64 or elemwise_axis_reduce or var_std or c2_)"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=1200)
    tail = (p.stdout or "")[-1500:]
    assert p.returncode == 0, tail + (p.stderr or "")[-1500:]
    assert " passed" in tail and "failed" not in tail, tail
