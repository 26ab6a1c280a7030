"""Helpers for the end-to-end GPU tests: the real drop-in (``pytensor.function(mode="hip")``)
next to the reference's own C linker (``mode="CVM"``) in the same process.

TEST INFRASTRUCTURE.  The importable reference copy (``oracle/_ref``, built by
``oracle/make_ref.py`` from ``/root/reference``) travels to the GPU box as a built artefact;
nothing here reads ``/root/reference`` at run time.  Shaped after the reference's own helper
for a JIT linker, ``tests/link/pytorch/test_basic.py:41-87`` (``compare_pytorch_and_py``).
"""
import os

import numpy as np
import pytest

import make_ref

F64_RTOL = 1e-12  # north_star: fp64 within 1e-12 rtol
F32_RTOL = 1e-5  # north_star: fp32 within 1e-5


def activate():
    """Import the reference copy + register ``mode="hip"``; skip (loudly) when it is absent."""
    if not make_ref.importable():
        pytest.skip("oracle/_ref (importable reference copy) is not present on this box: "
                    "run `python oracle/make_ref.py` where /root/reference exists")
    make_ref.activate()
    import pytensor

    import pytensor_amd

    pytensor_amd.register()
    return pytensor


def have_gpu():
    from pytensor_amd import ffi

    return ffi.device_count() > 0


def reference_mode():
    """The oracle's runtime: the C linker under the CVM; ``Mode("py")`` with the
    same rewrites only if the box has no g++ (said in the test id through `reference_mode_name`)."""
    import pytensor
    from pytensor.compile.mode import Mode

    if pytensor.config.cxx:
        return Mode(linker="cvm", optimizer="fast_run")
    return Mode(linker="py", optimizer="fast_run")


def reference_mode_name():
    import pytensor

    return "CVM" if pytensor.config.cxx else "py(no g++)"


def assert_close(got, want, what="", rtol=None, atol=0.0):
    """Element-wise: bit-exact for bool/int, ``|got-want| <= atol + rtol*|want|`` for floats
    (``atol`` only where the caller states why — a cancelling sum)."""
    assert isinstance(got, np.ndarray), f"{what}: the hip linker must return host ndarrays, got {type(got)}"
    want = np.asarray(want)
    assert got.dtype == want.dtype, f"{what}: dtype {got.dtype} != {want.dtype}"
    assert got.shape == want.shape, f"{what}: shape {got.shape} != {want.shape}"
    if want.dtype.kind in "biu":
        np.testing.assert_array_equal(got, want, err_msg=what)
        return
    if rtol is None:
        rtol = F64_RTOL if want.dtype == np.float64 else F32_RTOL
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol, equal_nan=True, err_msg=what)


def compare_hip_and_cvm(graph_inputs, graph_outputs, test_inputs, *, calls=3, rtol=None, atol=0.0, hip_mode="hip",
                        must_freeze=None, **fn_kwargs):
    """Compile with ``mode="hip"`` and with the reference C linker, call both on the same values.

    The hip function is called ``calls`` times (eager → hipGraph capture → replay) and every
    call must agree with the reference.  Returns ``(hip_fn, hip_results_of_last_call)``.
    """
    import pytensor

    single = not isinstance(graph_outputs, (list, tuple))
    outs = [graph_outputs] if single else list(graph_outputs)
    f_hip = pytensor.function(list(graph_inputs), outs, mode=hip_mode, **fn_kwargs)
    f_ref = pytensor.function(list(graph_inputs), outs, mode=reference_mode(), **fn_kwargs)
    want = f_ref(*test_inputs)
    got = None
    for c in range(calls):
        got = f_hip(*test_inputs)
        assert len(got) == len(want)
        for k, (a, b) in enumerate(zip(got, want)):
            assert_close(a, b, f"output {k}, call {c} (hip vs {reference_mode_name()})", rtol=rtol, atol=atol)
    if must_freeze is not None:
        exe = hip_executable(f_hip)
        assert (exe._auto_plan is not None) == must_freeze, f"frozen plan present: {exe._auto_plan is not None}"
    return f_hip, (got[0] if single else got)


def hip_executable(fn):
    """The ``HipExecutable`` behind a compiled ``Function`` (for white-box assertions)."""
    from pytensor_amd.executor import HipExecutable

    exe = fn.vm.jit_fn  # link/basic.py:716 — `fn.jit_fn = jit_fn`
    assert isinstance(exe, HipExecutable), type(exe)
    return exe
