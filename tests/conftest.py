import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: marker used by the reference's own test classes (tests/test_gpu_refsuite.py)")


def pytest_collection_modifyitems(config, items):
    """Parametrisations of the reference's own test modules that are not run under the hip linker
    (complex operands, pivoted QR): skipped with the reason, listed in the module itself."""
    for name in ("test_gpu_refsuite_linalg", "test_gpu_refsuite_index", "test_gpu_refsuite_math"):
        mod = sys.modules.get(name) or sys.modules.get("tests." + name)
        if mod is not None:
            mod.pytest_collection_modifyitems_for_this_module(items)
