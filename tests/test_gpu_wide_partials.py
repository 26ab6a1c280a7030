"""GPU: the tail kernel's wave-level fold of deferred reductions with MORE than 64 partials per reduction (two to four
values per lane, added in index order, then the butterfly; codegen.TAIL_WAVE_Q).  The default many-term launch hands over
at most 64 partials per term (dispatch/wide.TERM_CAP), so the wider fold is exercised here with the cap raised — in a
subprocess, because the caps are read when the modules are imported."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle"))
import np_graph
from pytensor_amd import configs, ffi
from pytensor_amd.executor import HipExecutable
from pytensor_amd.ir import Graph
ffi.init(0)
d = json.load(open(os.path.join({root!r}, "tests", "golden", "wide_200.json")))
g, names = Graph.from_dict(d), d["input_names"]
vals = configs.wide200_inputs(N=200_000, K=16, G=8)
ins = [vals[n] for n in names]
params = set(configs.wide200_params())
exe = HipExecutable(g, resident=[k for k, n in enumerate(names) if n not in params])
got = exe(*ins)
plan = exe.freeze(*ins)
rep = plan(*ins)
same = all(np.array_equal(a, b) for a, b in zip(got, rep))
want = np_graph.run_graph(g, ins)
worst = 0.0
for a, b in zip(got, want):
    b = np.asarray(b)
    tol = 1e-12 * np.abs(b) + 8 * 2.0**-52 * 200_000 * 64.0  # sums of N terms in another order (tests/test_gpu_fullsize._wide_tolerance)
    worst = max(worst, float(np.max(np.abs(np.asarray(a) - b) / tol)))
exe.profile_nodes(ins, reps=1)
tails = [k for k in exe.last_kernel_times if k.startswith("tail_")]
multi = [n for n in exe.graph.nodes if n.op == "MultiElemwise"]
print(json.dumps({{"worst": worst, "replay_identical": same, "tails": len(tails), "multi": len(multi)}}))
"""


@pytest.mark.parametrize("cap,groups", [(128, 6144), (256, 12288)])
def test_many_term_graph_with_more_than_64_partials_per_term(cap, groups):
    env = {**os.environ, "PTHIP_WIDE_CAP": str(cap), "PTHIP_WIDE_GROUPS": str(groups)}
    p = subprocess.run([sys.executable, "-c", WORKER.format(root=ROOT)], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["multi"] == 1 and r["tails"] >= 1, r
    assert r["replay_identical"], "a replayed plan must reproduce the eager bits"
    assert r["worst"] <= 1.0, f"|hip - oracle| / tol = {r['worst']}"


def test_many_term_graph_finishing_its_own_reductions():
    """PTHIP_WIDE_FINISH=1: each term's last workgroup folds the term's pairs (one-pass form, per-term tickets)"""
    env = {**os.environ, "PTHIP_WIDE_FINISH": "1"}
    p = subprocess.run([sys.executable, "-c", WORKER.format(root=ROOT)], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["multi"] == 1 and r["tails"] >= 1, r
    assert r["replay_identical"], "a replayed plan must reproduce the eager bits"
    assert r["worst"] <= 1.0, f"|hip - oracle| / tol = {r['worst']}"
