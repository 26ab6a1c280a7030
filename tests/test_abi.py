"""CPU: the C-ABI library loads and exports every symbol include/pthip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "pthip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pthip_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge

    ge.build()
    lib = ctypes.CDLL(os.path.join(ROOT, "pytensor_amd", "libpthip.so"))
    names = _declared()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_ffi_signatures_cover_the_header():
    from pytensor_amd import ffi

    assert sorted(ffi.SIGNATURES) == _declared()


def test_jit_compiles_without_gpu():
    from pytensor_amd import ffi

    code = ffi.jit_compile('extern "C" __global__ void k(double* x){ x[threadIdx.x] = exp(x[threadIdx.x]); }', "k.hip")
    assert code[:4] == b"\x7fELF"


def test_product_path_fails_loudly_without_gpu():
    from pytensor_amd import ffi
    from pytensor_amd.executor import HipExecutable
    from util import load_case

    if ffi.device_count() > 0:
        pytest.skip("a GPU is visible")
    g, ins, *_ = load_case("c1_gauss")
    exe = HipExecutable(g)
    with pytest.raises(ffi.HipError):
        exe(*ins)


def test_optional_special_function_helpers_compile_without_gpu():
    """the long scalar helpers (incomplete gamma / beta and their inverses, polygamma) are emitted only
    into kernels that use them: each combination must pass hiprtc on its own"""
    from pytensor_amd import codegen, ffi

    for op, nin in (("NdtriExp", 1), ("PolyGamma", 2), ("GammaIncInv", 2), ("GammaIncCInv", 2), ("BetaIncInv", 3), ("GammaInc", 2), ("BetaInc", 3)):
        body = {"in_dtypes": ["float64"] * nin, "out_dtypes": ["float64"],
                "body": [{"op": op, "in": [["i", k] for k in range(nin)], "dtype": "float64"}], "outs": [["t", 0]]}
        names = [f"v{k}" for k in range(nin)]
        src = (codegen.prelude_for(body) + '\nextern "C" __global__ void probe(const double* a, double* o) {\n'
               + "".join(f"  const double {nm} = a[{k}];\n" for k, nm in enumerate(names)) + "  double r;\n"
               + codegen.emit_body(body, names, ["r"], indent="  ") + "\n  o[0] = r;\n}\n")
        assert len(ffi.jit_compile(src, f"probe_{op}.hip")) > 1000, op


def test_generated_tail_kernel_with_the_device_join_compiles_without_gpu():
    """both forms of the generated tail kernel carry the plan's device-side join in their prologue (wait for the other
    stream's signal word, put it back; a wait given up is reported through the done word as 2) and pass hiprtc"""
    from pytensor_amd import codegen, ffi

    spec = {"ext": [{"kind": "P", "dtype": "float64"}], "slots": [{"dtype": "float64", "scalar": True}],
            "steps": [{"op": "rsum", "src": ("e", 0), "red": "Add", "acc_dtype": "float64", "dtype": "float64", "out": 0}], "outs": [0]}
    plain = codegen.tail_chain_source("tail_probe", spec)
    preload = codegen.tail_chain_source("tail_probe_p", spec, codegen.tail_preload_sizes(spec, [64], [64]))
    for name, src in (("tail_probe", plain), ("tail_probe_p", preload)):
        assert "int* join_src" in src and "__hip_atomic_store(done_dst, 2," in src and "if (join_fail_) return;" in src
        # the wait sits in front of every operand request
        assert src.index("join_src != nullptr") < src.index("e0[")
        assert len(ffi.jit_compile(src, name + ".hip")) > 1000


def test_bench_traffic_falls_back_when_the_profiler_pass_fails(monkeypatch):
    """bench.live_pmc_traffic: without a device the rocprofv3 child fails — a reason comes back, never an
    exception, so the bench line is still printed (with the committed summary as ``traffic``); and no
    counter passes are started from inside a profiled run."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    from pytensor_amd import ffi

    if ffi.device_count() <= 0:
        traffic, why = bench.live_pmc_traffic("gchain_")
        assert traffic is None and isinstance(why, str) and why
    monkeypatch.setenv("ROCPROF_OUTPUT_PATH", "/tmp/x")
    traffic, why = bench.live_pmc_traffic("gchain_")
    assert traffic is None and "profiled" in why


def test_device_status_word_maps_to_the_reference_exceptions():
    """Kernels cannot raise: they set bits of a device word, the executor raises at its next
    synchronisation — IndexError (take / inc_subtensor out of range), LinAlgError (np.linalg.inv of a
    singular matrix, non-converged SVD / Eigh), RuntimeError for an expired bounded wait of a persistent
    linear-algebra kernel (never a hang)."""
    import numpy as np
    import pytest

    from pytensor_amd.executor import raise_device_status

    raise_device_status(0)
    with pytest.raises(IndexError):
        raise_device_status(1)
    for bit, msg in ((2, "Singular matrix"), (4, "SVD did not converge"), (8, "Eigenvalues did not converge")):
        with pytest.raises(np.linalg.LinAlgError, match=msg):
            raise_device_status(bit)
    with pytest.raises(RuntimeError, match="persistent linear-algebra kernel"):
        raise_device_status(16)
    with pytest.raises(RuntimeError):  # the wait bit wins over an index error raised by the aborted launch's neighbours
        raise_device_status(16 | 1)
