"""CPU: host-side logic of the executor that needs no device (the device calls are stubbed)."""
import pytest

from pytensor_amd import executor, ffi
from util import load_case


class _FakeLib:
    def __init__(self, already):
        self.already, self.calls = already, 0

    def pthip_set_safe_mode(self, on):
        self.calls += 1
        was, self.already = self.already, 1
        return was


class _FakePlan:
    closed = False

    def close(self):
        self.closed = True


def _exe(monkeypatch, already, stale_plan):
    g, *_ = load_case("c1_gauss")
    exe = executor.HipExecutable(g, auto_freeze=True)
    lib = _FakeLib(already)
    monkeypatch.setattr(ffi, "lib", lambda: lib)
    state = {"n": 0}

    def boom(*a):
        state["n"] += 1
        raise executor.DeviceWaitExpired("expired")

    monkeypatch.setattr(exe, "_call", boom)
    monkeypatch.setattr(exe, "_call_eager", lambda *a: "eager-result")
    plan = _FakePlan() if stale_plan else None
    exe._auto_plan = plan
    return exe, lib, plan


def test_first_expired_wait_switches_to_safe_mode_and_retries(monkeypatch):
    exe, lib, plan = _exe(monkeypatch, already=0, stale_plan=True)
    with pytest.warns(RuntimeWarning, match="launch-per-step"):
        assert exe(1.0) == "eager-result"
    assert plan.closed and exe._auto_plan is None and lib.calls == 1


def test_stale_plan_of_another_executable_is_dropped_not_fatal(monkeypatch):
    """safe mode is process-wide; executable B's plan was captured with the cooperative kernels BEFORE
    executable A switched the process over.  B's expired wait must drop the plan and evaluate eagerly, not raise."""
    exe, lib, plan = _exe(monkeypatch, already=1, stale_plan=True)
    assert exe(1.0) == "eager-result"
    assert plan.closed and exe._auto_plan is None and exe.stats["safe_mode_retries"] == 1


def test_expired_wait_under_safe_mode_without_a_stale_plan_is_an_error(monkeypatch):
    exe, lib, plan = _exe(monkeypatch, already=1, stale_plan=False)
    with pytest.raises(executor.DeviceWaitExpired):
        exe(1.0)
