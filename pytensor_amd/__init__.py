"""pytensor_amd — an MI355X-native execution backend (``mode="hip"``) for PyTensor.

Layout (only what the hot path needs):

* ``csrc/``      hand-written HIP kernels for gfx950 + the C-ABI (``libpthip.so``)
* ``ffi.py``     ctypes binding of ``include/pthip.h`` (fails loudly if the library is missing)
* ``ir.py``      portable lowered-graph IR
* ``lower.py``   PyTensor ``FunctionGraph`` → IR (needs PyTensor)
* ``linker.py``  ``HipLinker(JITLinker)`` + mode/linker registration (needs PyTensor)
* ``codegen.py`` ``Composite`` scalar graph → fused HIP kernel source
* ``executor.py``/``dispatch/``  runs the IR on the device
"""

__version__ = "0.1.0"


def register():
    """Register ``mode="hip"`` with an importable PyTensor (idempotent)."""
    from pytensor_amd import linker  # noqa: F401

    return linker.HipLinker
