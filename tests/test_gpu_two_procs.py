"""GPU: two processes on ONE device through the persistent / cooperative linear-algebra kernels (PyMC's default launch
of several chains).  Each process evaluates Cholesky(2048), a vector SolveTriangular(4096) and Det(1024) in a loop for a
few seconds while the other does the same; every result of every call must be correct and no call may raise — the
task-graph kernels claim their tasks through tickets (no co-residency assumption, csrc/linalg.hip), and an LU panel
that cannot get its workgroups resident makes the executor switch to the launch-per-step forms and evaluate again
(executor.DeviceWaitExpired -> pthip_set_safe_mode).  Reference semantics: LAPACK on the host, which shares nothing."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_processes_share_one_device():
    env = {**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_two_proc_worker.py"), str(seed), "6"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for seed in (11, 12)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, f"worker failed:\n{se[-3000:]}"
    res = [json.loads(so.strip().splitlines()[-1]) for so, _ in outs]
    assert all(r["calls"] >= 3 for r in res), res
    print(res)


def test_safe_mode_forms_are_correct():
    """the launch-per-step forms the fallback lands on, forced: same graph, same checks, one process"""
    env = {**os.environ, "PTHIP_TEST_SAFE_MODE": "1"}
    p = subprocess.run([sys.executable, "-c",
                        "import sys,os;sys.path.insert(0,%r);from pytensor_amd import ffi;ffi.init(0);ffi.lib().pthip_set_safe_mode(1);"
                        "f=%r;sys.argv=[f,'21','0.5'];exec(compile(open(f).read(),f,'exec'),{'__file__':f,'__name__':'__main__'})"
                        % (os.path.dirname(HERE), os.path.join(HERE, "_two_proc_worker.py"))],
                       capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["calls"] >= 3
