"""GPU: the reference's own indexing / shape / Blockwise test modules under the hip linker.

``tests/tensor/test_subtensor.py`` (the bit-exact indexing tier of SURVEY §8a: ``Subtensor``,
``IncSubtensor``, ``AdvancedSubtensor``, ``AdvancedIncSubtensor``, boolean masks, ``take``),
``tests/tensor/test_shape.py`` (``Shape_i``, ``Reshape``, ``SpecifyShape``) and
``tests/tensor/test_blockwise.py`` of the reference (``oracle/_ref/tests``, a built artefact) compile
with the DEFAULT mode: each module is imported — and every test run — with ``config.mode`` set to the
registered ``hip`` mode (same mechanism as ``tests/test_gpu_refsuite_linalg.py``).  What is not run
is listed in ``NOT_RUN_IDS`` with the reason.  First run of these modules found: a boolean mask next to
a slice renumbered the slice's operands away; ``inc_subtensor(..., ignore_duplicates=True)``;
the runtime-broadcast ``ValueError`` of a vector-index update; ``Blockwise`` of an inlined
``OpFromGraph`` and of an op whose lowering looks at its node.
"""
import importlib

import pytest

import e2e_util as E
import make_ref

pytestmark = pytest.mark.gpu

if not make_ref.importable():
    pytest.skip("oracle/_ref (importable reference copy incl. its tests/) is not present", allow_module_level=True)

E.activate()

from pytensor import config  # noqa: E402
from pytensor.compile.mode import get_mode  # noqa: E402

HIP = get_mode("hip")

MODULES = {
    "subtensor": "tests.tensor.test_subtensor",
    "shape": "tests.tensor.test_shape",
    "blockwise": "tests.tensor.test_blockwise",
    "basic": "tests.tensor.test_basic",
}

# test name (as exported) -> reason it is not run under the hip linker
NOT_RUN = {}
# substrings of a test id -> reason (applied at collection through tests/conftest.py)
NOT_RUN_IDS = {
    "Test_blockwise__Inplace": "asserts destroy_map of the rewritten graph (`inplace` rewrites are incompatible with this linker)",
    "Subtensor::test_grad_list": "asserts `node.op.inplace` in the rewritten graph",
    "test_blockwise__perform_method_per_node": "a test-only Op (NodeDependentPerformOp) with no device lowering",
    "test_blockwise__blockwise_infer_core_shape": "a test-only Op (TestOpWithInferShape) with no device lowering",
    "test_blockwise__blockwise_shape": "a test-only Op (MyTestOp) with no device lowering",
    "test_blockwise__eig_blockwise": "Eig returns complex eigenvalues (DESIGN §7)",
    "complex": "complex operands (DESIGN §7)",
    "Triangle::test_tri": "loops over complex dtypes inside the test (DESIGN §7)",
    "test_basic__identity": "loops over complex dtypes inside the test (DESIGN §7)",
    "Choose::test_numpy_compare_tuple": "Choose over a typed list (DESIGN §7)",
}


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    import os

    if not E.have_gpu() and not os.environ.get("PTHIP_LOWER_ONLY"):
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")


@pytest.fixture(autouse=True)
def _hip_is_the_default_mode():
    with config.change_flags(mode=HIP):
        yield


def _export():
    saved = config.mode
    config.mode = HIP  # module-level `mode = get_default_mode()` constants of the reference's modules
    try:
        mods = {k: importlib.import_module(m) for k, m in MODULES.items()}
    finally:
        config.mode = saved
    g = globals()
    for key, mod in mods.items():
        for name, obj in vars(mod).items():
            if getattr(obj, "__module__", None) != mod.__name__:
                continue  # (helpers imported from elsewhere)
            if name.startswith("test_") and callable(obj):
                new = f"test_{key}__{name[5:]}"
            elif name.startswith("Test") and isinstance(obj, type):
                new = f"Test_{key}__{name[4:]}"
            else:
                continue
            if new in NOT_RUN:
                continue
            g[new] = obj


_export()


def pytest_collection_modifyitems_for_this_module(items):
    """called from tests/conftest.py: mark the listed parametrisations as skipped, reason attached"""
    for item in items:
        if not item.nodeid.startswith("tests/test_gpu_refsuite_index.py"):
            continue
        for sub, why in NOT_RUN_IDS.items():
            if sub in item.nodeid:
                item.add_marker(pytest.mark.skip(reason=f"not run under the hip linker: {why}"))
                break
