"""GPU: the reference's own Scan test classes and ``TestGemm``'s value cases under the hip linker.

``tests/scan/test_basic.py`` of the reference (``oracle/_ref/tests``, a built artefact — the test
code is the reference's, never committed): ``TestScan`` (:249), ``TestGradUntil`` (:2458),
``TestExamples`` (:2712) and the module-level Scan tests compile with the DEFAULT mode, so the
module is imported — and every test run — with ``config.mode`` set to the registered ``hip`` mode
(``config.mode`` accepts a ``Mode`` instance, configdefaults.py:34-63).  What such a test then
checks is what it always checked: values against NumPy loops, gradients against finite
differences (``utt.verify_grad``), shapes, error messages — now produced by ``HipLinker``.
Tests that never reach the default mode (they build their own ``Mode(linker="py"|"cvm")``) or
that assert properties of the C VM are listed at the bottom of each class with the reason.

``tests/tensor/test_blas.py::TestGemm`` (:82) hard-codes ``gemm_inplace`` and the linker strings
``"c|py"``/``"py"``/``"c"`` in ``cmp``; ``TestGemmHip`` keeps the class's cases (``test_basic_*``,
``test_shape_0``, ``test_transposes``) and swaps in ``gemm_no_inplace`` under
``Mode(optimizer=None, linker="hip")`` — the destructive variant cannot exist for a linker that
lists ``inplace`` in ``incompatible_rewrites``.
"""
import numpy as np
import pytest

import e2e_util as E
import make_ref

pytestmark = pytest.mark.gpu

if not make_ref.importable():
    pytest.skip("oracle/_ref (importable reference copy incl. its tests/) is not present", allow_module_level=True)

E.activate()

import pytensor  # noqa: E402
from pytensor import config  # noqa: E402
from pytensor.compile.mode import Mode, get_mode  # noqa: E402

HIP = get_mode("hip")

_saved_mode = config.mode
config.mode = HIP  # module-level `mode_with_opt = get_default_mode()` of the reference's test module
try:
    from tests.scan import test_basic as ref_scan  # noqa: E402
finally:
    config.mode = _saved_mode
from tests import unittest_tools as utt  # noqa: E402
from tests.tensor import test_blas as ref_blas  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    import os

    if not E.have_gpu() and not os.environ.get("PTHIP_LOWER_ONLY"):  # (builder's check without a GPU: how far does each test compile)
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")


@pytest.fixture(autouse=True)
def _hip_is_the_default_mode():
    with config.change_flags(mode=HIP):
        yield


def test_the_default_mode_is_hip_in_this_module():
    from pytensor.compile.mode import get_default_mode

    from pytensor_amd.linker import HipLinker

    assert isinstance(get_default_mode().linker, HipLinker)
    assert isinstance(ref_scan.mode_with_opt.linker, HipLinker)


class TestScanHip(ref_scan.TestScan):
    # -- never reach the default mode: they parametrise over Mode(linker="py"/"cvm") themselves
    test_no_step = None
    test_no_steps_sit_sot = None
    test_no_steps_nit_sot = None
    test_sequence_is_scan = None
    test_inner_graph_cloning = None  # Mode(optimizer=None) with the default (C) linker
    # -- assert properties of the C VM / the C cache
    test_monitor_mode = None  # MonitorMode wraps the VM's per-node callback (link/vm.py)
    test_inner_storage_leak = None  # counts storage cells of the inner VM function
    # -- compare draws with NumPy's PCG64 stream value by value: the device samplers are
    #    counter-based (Philox); a PCG64 generator is re-keyed, parity is distributional.
    #    Random draws INSIDE a Scan are lowered and exercised by test_grad_multiple_outs_some_truncate,
    #    test_grad_multiple_outs_some_uncomputable, test_pushforward_2 (which only need the draws to be
    #    the same in the functions they compare).
    test_simple_shared_random = None


class TestGradUntilHip(ref_scan.TestGradUntil):
    pass


class TestExamplesHip(ref_scan.TestExamples):
    # -- build Mode(linker="py") explicitly
    test_eliminate_seqs = None
    test_eliminate_nonseqs = None
    # -- compares binomial draws with NumPy's PCG64 stream value by value (see TestScanHip)
    test_gibbs_chain = None


# module-level Scan tests that use the default mode
def test_ref_mintap_onestep():
    ref_scan.test_mintap_onestep()


def test_ref_constant_folding_n_steps():
    ref_scan.test_constant_folding_n_steps()


@pytest.mark.parametrize("single_step", (True, False))
def test_ref_scan_mapped_and_non_traced_output_ordering(single_step):
    ref_scan.test_scan_mapped_and_non_traced_output_ordering(single_step)


def test_ref_single_step_untraced_sit_sot():
    ref_scan.test_single_step_untraced_sit_sot()


# ---------------------------------------------------------------------------------------------
# tests/tensor/test_blas.py::TestGemm — value cases through gemm_no_inplace under the hip linker
# ---------------------------------------------------------------------------------------------
class TestGemmHip(ref_blas.TestGemm):
    def cmp(self, z_, a_, x_, y_, b_):
        from pytensor.tensor import as_tensor_variable
        from pytensor.tensor.blas import gemm_no_inplace

        for dtype in ["float32", "float64"]:  # (complex: compile-time NotImplementedError)
            z, a, x, y, b = (np.asarray(p, dtype=dtype) for p in (z_, a_, x_, y_, b_))
            z_orig = z.copy()
            tz, ta, tx, ty, tb = (as_tensor_variable(p).type() for p in (z, a, x, y, b))
            f = pytensor.function([tz, ta, tx, ty, tb], gemm_no_inplace(tz, ta, tx, ty, tb),
                                  mode=Mode(optimizer=None, linker="hip"))
            for _ in range(3):  # eager, capture, replay
                got = f(z, a, x, y, b)
                utt.assert_allclose(self._gemm(z_orig, a, x, y, b), got)
                np.testing.assert_array_equal(z, z_orig)  # functional: the input is untouched

    def test_transposes(self):
        from pytensor import shared
        from pytensor.tensor.blas import gemm_no_inplace

        rng = np.random.default_rng(seed=utt.fetch_seed())
        A, B, C = (rng.random((4, 5))[:, :4] for _ in range(3))

        def t(z, x, y, a=1.0, b=0.0, dt="float64"):
            z, a, x, y, b = (np.asarray(p, dtype=dt) for p in (z, a, x, y, b))
            z_after = self._gemm(z, a, x, y, b)
            tz, ta, tx, ty, tb = (shared(p) for p in (z, a, x, y, b))
            f = pytensor.function([], gemm_no_inplace(tz, ta, tx, ty, tb), mode=Mode(optimizer=None, linker="hip"))
            for _ in range(3):
                utt.assert_allclose(z_after, f())
            y_T = ty.get_value(borrow=True).T
            ty.set_value(tx.get_value(borrow=True).T, borrow=True)
            tx.set_value(y_T, borrow=True)
            # the transposed product: (x y)^T = y^T x^T
            utt.assert_allclose(self._gemm(z, a, y.T, x.T, b), f())

        t(C, A, B)
        t(C.T, A, B)
        t(C, A.T, B, dt="float32")
        t(C, A, B.T)
        t(C.T, A.T, B)
        t(C, A.T, B.T, dt="float32")
        t(C, A[:, :2], B[:2, :])
        t(C.T, A[:, :2], B[:2, :], dt="float32")
        t(C, A[:2, :].T, B[:2, :])
        t(C.T, A[:2, :].T, B[:2, :], dt="float32")
        t(C, A[:2, :].T, B[:, :2].T)
        t(C.T, A[:2, :].T, B[:, :2].T, dt="float32")

    # -- assert the graph the C backend's InplaceBlasOpt builds (Gemm{inplace}, mode="CVM")
    test_factorised_scalar = None
    # -- destroy_map protocol of the in-place Op (graph construction only; no linker involved)
    test_destroy_map0 = None
    test_destroy_map1 = None
    test_destroy_map2 = None
    test_destroy_map3 = None
    test_destroy_map4 = None  # inplace_func + gemm_inplace
    test_non_contiguous = None  # inplace_func + gemm_inplace over strided shared values: covered by TestBlasStridesHip
