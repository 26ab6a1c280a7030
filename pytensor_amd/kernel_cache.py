"""JIT cache for generated kernels: source → gfx950 code object → loaded function.

Analogue of the reference's C-module cache (pytensor/link/c/cmodule.py:612
``ModuleCache``): keyed by a hash of the generated source; code objects persist
in-tree under ``pytensor_amd/_kcache/`` (git-ignored ``*.hsaco``), so kernels
compiled by ``build()`` without a GPU are reused on the GPU box.
"""

from __future__ import annotations

import ctypes as C
import os

from pytensor_amd import ffi
from pytensor_amd.codegen import source_key

_DEFAULT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_kcache")
CACHE_DIR = os.environ.get("PTHIP_KCACHE") or _DEFAULT_DIR  # (tests patch this; the flag below wins when set)


def cache_dir() -> str:
    """``config.hip__cache_dir`` when PyTensor is loaded and the flag is set, else ``CACHE_DIR``."""
    import sys

    pt = sys.modules.get("pytensor")
    d = getattr(getattr(pt, "config", None), "hip__cache_dir", "") if pt is not None else ""
    return d or CACHE_DIR

_code = {}  # key -> bytes
_funcs = {}  # (key, name) -> function handle
_modules = {}


def compile_source(src: str, name: str) -> str:
    """Compile (or fetch) ``src``; returns the cache key.  Needs no GPU."""
    if "_Float16" in src:
        # float16 is rounded after every scalar op (NumPy semantics, what the reference's Elemwise
        # `perform` does): HIP's default -ffp-contract=fast would turn `half(a*b) - c` into one
        # v_fma_f16 and skip the intermediate rounding
        src = "#pragma clang fp contract(off)\n" + src
    key = source_key(src)
    if key in _code:
        return key
    dump = os.environ.get("PTHIP_KERNEL_SRC_DIR")  # diagnostic: keep every generated source
    if dump:
        try:
            os.makedirs(dump, exist_ok=True)
            with open(os.path.join(dump, f"{name}.hip"), "w") as fh:
                fh.write(src)
        except OSError:
            pass
    cdir = cache_dir()
    path = os.path.join(cdir, f"{key}.hsaco")
    if os.path.exists(path):
        with open(path, "rb") as fh:
            _code[key] = fh.read()
        return key
    code = ffi.jit_compile(src, name + ".hip")
    _code[key] = code
    try:
        os.makedirs(cdir, exist_ok=True)
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as fh:
            fh.write(code)
        os.replace(tmp, path)
    except OSError:
        pass
    return key


def get_function(src: str, name: str) -> int:
    key = compile_source(src, name)
    f = _funcs.get((key, name))
    if f is not None:
        return f
    lib = ffi.lib()
    mod = _modules.get(key)
    if mod is None:
        m = C.c_void_p()
        code = _code[key]
        buf = C.create_string_buffer(code, len(code))
        ffi.check(lib.pthip_module_load(buf, len(code), C.byref(m)))
        _modules[key] = (m, buf)
        mod = _modules[key]
    fn = C.c_void_p()
    ffi.check(lib.pthip_module_get_function(mod[0], name.encode(), C.byref(fn)))
    _funcs[(key, name)] = fn.value
    return fn.value
