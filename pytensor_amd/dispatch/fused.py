"""Handlers of the IR-level fusions that have no single reference Op
(``pytensor_amd/fusion.py``): ``GemvChain`` / ``GemvFinish``.

``GemvChain`` reads the matrix once for the forward ``A@x`` and the backward ``A.T@w``
(reference: two ``Gemv.perform`` calls, pytensor/tensor/blas/gemv.py:64-108, with an
``Elemwise.perform`` between them).  When the operands do not meet the fast kernel's
layout requirements the handler falls back to the three unfused launches — same
results, two passes over the matrix.
"""

from __future__ import annotations

import os
import struct

import numpy as np

from pytensor_amd import codegen, ffi, kernel_cache
from pytensor_amd.device import DeviceArray
from pytensor_amd.dispatch import handler
from pytensor_amd.dispatch.blas import _scalar, gemv_device
from pytensor_amd.dispatch.elemwise import BLOCK, MAX_GRID, _body_key, alloc_partials, finish_partials, launch_elemwise
from pytensor_amd.executor import HostValue

CHAIN_RG = int(os.environ.get("PTHIP_CHAIN_RG", 0))  # 0 = auto


def _fast_ok(A, x1, y1, e_ins, N, K):
    if str(A.dtype) != "float64" or A.ndim != 2 or A.strides[1] != 1 or A.strides[0] % 2 or A.ptr % 16:
        return False
    if K % 2 or K > 1024 or N == 0:
        return False
    if not x1.is_contiguous() or x1.ptr % 16 or str(x1.dtype) != "float64":
        return False
    if y1 is not None and (not y1.is_contiguous() or y1.shape != (N,)):
        return False
    for a in e_ins:
        if a.size == 1:
            continue
        if a.shape != (N,) or not a.is_contiguous():
            return False
    return True


@handler("GemvChain")
def gemv_chain(node, inputs, env):
    p = node.params
    body, spec, r_pos, w_out, store_r = p["scalar"], p["reduce"], p["r_pos"], p["w_out"], p["store_r"]
    y1, alpha1, A, x1, beta1, *e_rest = inputs
    alpha1, beta1 = _scalar(env, alpha1), _scalar(env, beta1)
    A, x1 = env.to_device(A), env.to_device(x1)
    y1d = None if beta1 == 0.0 else env.to_device(y1)
    e_ins = [env.to_device(v) for v in e_rest]
    N, K = A.shape
    nout = len(body["out_dtypes"])
    if x1.shape != (K,):
        raise ValueError(f"Shape mismatch: A.shape[1] != x.shape[0] ({A.shape}, {x1.shape})")
    if not _fast_ok(A, x1, y1d, e_ins, N, K):
        return _fallback(node, env, alpha1, A, x1, beta1, y1d, e_ins)
    # runtime broadcast rule of Elemwise (elemwise.py:825-840) for the vector inputs
    g = env.graph
    for vid, a in zip(node.inputs[5:], e_ins):
        if a.size == 1 and a.ndim == 1 and N != 1 and g.vars[vid].shape[0] != 1:
            raise ValueError("Runtime broadcasting not allowed (GemvChain elementwise input)")
    C = (K + 127) // 128
    RG = CHAIN_RG or max(4, 32 // C)
    while RG * C > 32 and RG > 4:
        RG //= 2
    e_modes = []
    it = iter(e_ins)
    for pos in range(len(body["in_dtypes"])):
        if pos == r_pos:
            e_modes.append("R")
        else:
            a = next(it)
            e_modes.append("S" if a.size == 1 else "V")
    rs = [None if r is None else (r["op"], r["acc_dtype"]) for r in spec]
    rkey = "".join("-" if r is None else r["op"][0] + r["acc_dtype"][0] + r["acc_dtype"][-1] for r in spec)
    name = f"gchain_{_body_key(body)}_{''.join(e_modes)}_{rkey}_w{w_out}_c{C}_g{RG}_{int(store_r)}{int(y1d is not None)}".replace("-", "x")
    src = codegen.gemv_chain_source(name, body, e_modes, rs, w_out, C, RG, store_r, y1d is not None)
    fn = kernel_cache.get_function(src, name)
    ngroups = (N + RG - 1) // RG
    grid = max(1, min((ngroups + 3) // 4, MAX_GRID))
    args = [("q", N), ("q", K), ("q", A.ptr), ("q", A.strides[0]), ("q", x1.ptr), ("q", y1d.ptr if y1d is not None else 0), ("d", alpha1), ("d", beta1)]
    args += [("q", a.ptr) for a in e_ins]
    outs = []
    r_out = None
    if store_r:
        r_out = DeviceArray.empty((N,), "float64")
        args.append(("q", r_out.ptr))
    parts = alloc_partials(spec, grid)
    stored = [None] * nout
    for k in range(nout):
        if spec[k] is None:
            stored[k] = DeviceArray.empty((N,), body["out_dtypes"][k])
            args.append(("q", stored[k].ptr))
        else:
            args.append(("q", parts[k].ptr))
    partT = DeviceArray.empty((grid, K), "float64")
    args.append(("q", partT.ptr))
    buf = struct.pack("<" + "".join(a[0] for a in args), *[a[1] for a in args])
    kt = env.kernel_timer
    tok = kt.begin() if kt is not None else None
    ffi.check(env.lib.pthip_launch(fn, grid, 1, 1, BLOCK, 1, 1, 0, buf, len(buf)))
    if kt is not None:
        kt.end(name, tok)
    finals = finish_partials(env, spec, parts, grid)
    res = [r_out] if store_r else []
    for k in range(nout):
        res.append(stored[k] if spec[k] is None else finals[k])
    res.append(partT)
    return res


def _fallback(node, env, alpha1, A, x1, beta1, y1d, e_ins):
    """Three launches, two passes over A (exactly the unfused graph)."""
    p = node.params
    body, spec, r_pos, w_out, store_r = p["scalar"], p["reduce"], p["r_pos"], p["w_out"], p["store_r"]
    N, K = A.shape
    r = gemv_device(env, alpha1, A, x1, beta1, y1d)
    ins = list(e_ins)
    ins.insert(r_pos, r)
    shape = (N,)
    outs, parts, grid = launch_elemwise(body, ins, shape, body["out_dtypes"], spec, env)
    finals = finish_partials(env, spec, parts, grid) if grid else [None] * len(spec)
    res = [r] if store_r else []
    for k, sp in enumerate(spec):
        if sp is None:
            res.append(outs[k])
        elif grid == 0:
            res.append(env.to_device(HostValue(np.asarray({"Add": 0, "Mul": 1}[sp["op"]], dtype=sp["dtype"]))))
        else:
            res.append(finals[k])
    w = outs[w_out]
    At = A.view((K, N), (A.strides[1], A.strides[0]))
    t = gemv_device(env, 1.0, At, w, 0.0, None)
    res.append(t.view((1, K), (K, 1)))
    return res


@handler("GemvFinish")
def gemv_finish(node, inputs, env):
    part, y2, alpha2, beta2 = inputs
    alpha2, beta2 = _scalar(env, alpha2), _scalar(env, beta2)
    nparts, M = part.shape
    out = DeviceArray.empty((M,), part.dtype)
    y2d = None if beta2 == 0.0 else env.to_device(y2)
    if y2d is not None and y2d.shape != (M,):
        raise ValueError(f"Shape mismatch: y.shape[0] != A.shape[0] ({y2d.shape}, {M})")
    ffi.check(
        env.lib.pthip_gemv_finish(
            ffi.np_dtype_code(part.dtype), M, nparts, part.ptr, alpha2, beta2,
            y2d.ptr if y2d is not None else None, (y2d.strides[0] if M > 1 else 1) if y2d is not None else 0, out.ptr,
        )
    )
    return [out]
