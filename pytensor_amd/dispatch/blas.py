"""BLAS-family ops: ``Gemv``, ``Ger``, ``Dot22``, ``Dot22Scalar``, ``Gemm``, ``BatchedDot``, ``Dot``.

Reference: pytensor/tensor/blas/ (gemv.py:16, ger.py:8, gemm.py:76,248,298,
batched.py:18) and ``Dot`` (pytensor/tensor/math.py:3041).  The stride→layout
decisions made by the reference's C glue (c_code/codegen.py:159-250: pick N/T flags
from strides, copy when no unit stride 111-157) are made here on the host and passed
to the kernels as element strides.
"""

from __future__ import annotations

import os

import numpy as np

from pytensor_amd import ffi
from pytensor_amd.device import DeviceArray
from pytensor_amd.dispatch import handler
from pytensor_amd.executor import HostValue


def _scalar(env, v) -> float:
    return float(np.asarray(env.to_host(v)).reshape(-1)[0])


def _host_scalar(v):
    """The value of alpha/beta when the host knows it (a constant or shape arithmetic); ``None``
    when it is *computed on the device* (e.g. ``-lr`` of an SGD update, ``lr`` a function input):
    reading it back would be a host synchronisation per call and cannot be captured in a hipGraph,
    so those cases run the product with alpha=1, beta=0 and a fused ``beta*y + alpha*t`` epilogue
    kernel that takes the scalars as device operands (`_axpby`)."""
    if isinstance(v, HostValue):
        return float(np.asarray(v.a).reshape(-1)[0])
    if isinstance(v, DeviceArray):
        return None
    return float(np.asarray(v).reshape(-1)[0])


def _axpby(env, alpha, t, beta, y):
    """``beta*y + alpha*t`` (``y`` broadcast to ``t``; ``beta == 0`` never reads ``y``, as
    gemv.py:79-86 / gemm.py:183-216 do) with alpha / beta as host floats or device scalars."""
    from pytensor_amd.dispatch.elemwise import launch_elemwise

    dt = str(t.dtype)

    def operand(v):
        h = _host_scalar(v) if not isinstance(v, float) else v
        if h is not None:
            return HostValue(np.asarray(h, dtype=dt))
        v = env.to_device(v)
        return v.view((1,) * t.ndim, (0,) * t.ndim) if t.ndim else v

    a_op, b_op = operand(alpha), operand(beta)
    bh = _host_scalar(beta) if not isinstance(beta, float) else beta
    if y is None or bh == 0.0:
        body = {"in_dtypes": [dt, dt], "out_dtypes": [dt],
                "body": [{"op": "Mul", "in": [["i", 1], ["i", 0]], "dtype": dt}], "outs": [["t", 0]]}
        outs, _, _ = launch_elemwise(body, [t, a_op], t.shape, [dt], None, env)
        return outs[0]
    yb = y
    if y.shape != t.shape:  # Gemm: z may be broadcast along a static length-1 axis (gemm.py:194-198)
        ysh = (1,) * (t.ndim - y.ndim) + y.shape
        yst = (0,) * (t.ndim - y.ndim) + y.strides
        yb = y.view(ysh, tuple(0 if s == 1 and d != 1 else st for s, d, st in zip(ysh, t.shape, yst)))
    zero = ["c", "0x0.0p+0", dt]
    body = {
        "in_dtypes": [dt, dt, dt, dt], "out_dtypes": [dt],
        "body": [
            {"op": "Mul", "in": [["i", 1], ["i", 0]], "dtype": dt},  # alpha * t
            {"op": "Mul", "in": [["i", 2], ["i", 3]], "dtype": dt},  # beta * y
            {"op": "EQ", "in": [["i", 2], zero], "dtype": "bool"},
            {"op": "Switch", "in": [["t", 2], zero, ["t", 1]], "dtype": dt},  # beta == 0: y is not used
            {"op": "Add", "in": [["t", 3], ["t", 0]], "dtype": dt},
        ],
        "outs": [["t", 4]],
    }
    outs, _, _ = launch_elemwise(body, [t, a_op, b_op, yb], t.shape, [dt], None, env)
    return outs[0]


def _dt(x) -> int:
    return ffi.np_dtype_code(x.dtype)


def _unit_stride_2d(x: DeviceArray) -> DeviceArray:
    """Like the reference's C glue (codegen.py:111-157): copy operands that have no
    unit stride (or negative strides) into a fresh row-major buffer."""
    ok = all(s >= 0 for s in x.strides) and (1 in x.strides or x.size <= 1 or min(x.shape) == 1)
    return x if ok else x.contiguous()


def gemv_device(env, alpha, A, x, beta, y):
    lib = env.lib
    M, N = A.shape
    if x.shape[0] != N:
        raise ValueError(f"Shape mismatch: A.shape[1] != x.shape[0] ({A.shape}, {x.shape})")
    A = _unit_stride_2d(A)
    out = DeviceArray.empty((M,), A.dtype)
    if y is not None and y.shape[0] != M:
        raise ValueError(f"Shape mismatch: y.shape[0] != A.shape[0] ({y.shape}, {A.shape})")
    sA0, sA1 = A.strides
    if N == 1:
        sA1 = 1
    if M == 1 and sA1 != 1:
        sA0 = 1
    ws_bytes = lib.pthip_gemv_workspace(_dt(A), M, N, sA0, sA1)
    ws = DeviceArray.empty((ws_bytes,), "uint8") if ws_bytes else None
    # (row layout: one kernel; column layout: the streaming kernel plus its fixed-order finish)
    env.timed(
        f"gemv_{A.dtype}_{M}x{N}_{'row' if sA1 == 1 else 'col'}",
        lambda: ffi.check(
            lib.pthip_gemv(
                _dt(A), M, N, float(alpha), A.ptr, sA0, sA1, x.ptr, x.strides[0] if x.shape[0] > 1 else 1,
                float(beta), y.ptr if y is not None else None, (y.strides[0] if y.shape[0] > 1 else 1) if y is not None else 0,
                out.ptr, ws.ptr if ws is not None else None, ws_bytes,
            )
        ),
    )
    return out


@handler("Gemv")
def gemv(node, inputs, env):
    y, alpha, A, x, beta = inputs
    ah, bh = _host_scalar(alpha), _host_scalar(beta)
    A, x = env.to_device(A), env.to_device(x)
    y = env.to_device(y)
    if ah is None or bh is None:  # alpha/beta computed on the device
        if y.shape[0] != A.shape[0]:
            raise ValueError(f"Shape mismatch: y.shape[0] != A.shape[0] ({y.shape}, {A.shape})")
        return [_axpby(env, alpha, gemv_device(env, 1.0, A, x, 0.0, None), beta, y)]
    return [gemv_device(env, ah, A, x, bh, None if bh == 0.0 else y)]


@handler("Ger")
def ger(node, inputs, env):
    A, alpha, x, y = inputs
    ah = _host_scalar(alpha)
    A, x, y = env.to_device(A), env.to_device(x), env.to_device(y)
    if ah is None:  # device alpha: A + (alpha*x) y^T
        x, alpha = _axpby(env, alpha, x, 0.0, None), 1.0
    else:
        alpha = ah
    M, N = A.shape
    if x.shape[0] != M or y.shape[0] != N:
        raise ValueError("Ger: shape mismatch")
    out = DeviceArray.empty((M, N), A.dtype)
    ffi.check(
        env.lib.pthip_ger(_dt(A), M, N, alpha, A.ptr, A.strides[0], A.strides[1], x.ptr, x.strides[0], y.ptr, y.strides[0], out.ptr)
    )
    return [out]


def gemm_device(env, alpha, A, B, beta=0.0, Cm=None, batch=None):
    """out = beta*C + alpha*A@B on MFMA tiles; A (.., M, K), B (.., K, N)."""
    lib = env.lib
    if batch is None:
        M, K = A.shape
        K2, N = B.shape
        nb = 1
        sAb = sBb = 0
        sA, sB = A.strides, B.strides
    else:
        nb, M, K = A.shape
        nb2, K2, N = B.shape
        if nb != nb2:
            raise TypeError(f"Inputs [{A.shape}, {B.shape}] must have the same size in axis 0")
        sAb, sBb = A.strides[0], B.strides[0]
        sA, sB = A.strides[1:], B.strides[1:]
    if K != K2:
        raise ValueError(f"Shape mismatch: x has {K} cols but y has {K2} rows")
    shape = (M, N) if batch is None else (nb, M, N)
    out = DeviceArray.empty(shape, A.dtype)
    if out.size == 0:
        return out
    if Cm is not None and beta != 0.0:
        cs = Cm.strides
        cshape = Cm.shape
        sC0 = 0 if cshape[-2] == 1 and M != 1 else cs[-2]
        sC1 = 0 if cshape[-1] == 1 and N != 1 else cs[-1]
        sCb = cs[0] if batch is not None else 0
        cptr = Cm.ptr
    else:
        sC0 = sC1 = sCb = 0
        cptr = None
        beta = 0.0
    env.timed(
        f"gemm_{A.dtype}_b{nb}_{M}x{N}x{K}",
        lambda: ffi.check(
            lib.pthip_gemm(
                _dt(A), nb, M, N, K, float(alpha), A.ptr, sAb, sA[0], sA[1], B.ptr, sBb, sB[0], sB[1],
                float(beta), cptr, sCb, sC0, sC1, out.ptr,
            )
        ),
    )
    return out


def _prep2d(x):
    return _unit_stride_2d(x)


@handler("Dot22")
def dot22(node, inputs, env):
    x, y = (env.to_device(i) for i in inputs)
    return [gemm_device(env, 1.0, _prep2d(x), _prep2d(y))]


@handler("Dot22Scalar")
def dot22scalar(node, inputs, env):
    x, y, a = inputs
    x, y = env.to_device(x), env.to_device(y)
    ah = _host_scalar(a)
    if ah is None:
        return [_axpby(env, a, gemm_device(env, 1.0, _prep2d(x), _prep2d(y)), 0.0, None)]
    return [gemm_device(env, ah, _prep2d(x), _prep2d(y))]


@handler("Gemm")
def gemm(node, inputs, env):
    z, a, x, y, b = inputs
    z, x, y = env.to_device(z), env.to_device(x), env.to_device(y)
    ah, bh = _host_scalar(a), _host_scalar(b)
    if ah is None or bh is None:
        return [_axpby(env, a, gemm_device(env, 1.0, _prep2d(x), _prep2d(y)), b, z)]
    return [gemm_device(env, ah, _prep2d(x), _prep2d(y), bh, z)]


def _prep3d(x):
    ok = all(s >= 0 for s in x.strides) and (1 in x.strides[1:] or x.size <= 1)
    return x if ok else x.contiguous()


@handler("BatchedDot")
def batched_dot(node, inputs, env):
    x, y = (env.to_device(i) for i in inputs)
    return [gemm_device(env, 1.0, _prep3d(x), _prep3d(y), batch=True)]


def _int_dot(env, x, y, out_dtype):
    """Integer ``Dot`` (no BLAS in the reference either: ``np.dot`` on integer arrays): operands
    cast to the result dtype, wrap-around arithmetic, bit-exact."""
    from pytensor_amd.dispatch.elemwise import _cast

    if x.ndim > 2 or y.ndim > 2:
        raise NotImplementedError("Dot with ndim > 2")
    K = x.shape[-1]
    if K != y.shape[0]:
        raise ValueError(f"shapes {x.shape} and {y.shape} not aligned: {K} (dim {x.ndim - 1}) != {y.shape[0]} (dim 0)")
    x = x if str(x.dtype) == out_dtype else _cast(env, x, out_dtype)
    y = y if str(y.dtype) == out_dtype else _cast(env, y, out_dtype)
    xm = x if x.ndim == 2 else x.view((1, K), (0, x.strides[0]))
    ym = y if y.ndim == 2 else y.view((K, 1), (y.strides[0], 0))
    M, N = xm.shape[0], ym.shape[1]
    out_shape = (() if x.ndim == 1 else (M,)) + (() if y.ndim == 1 else (N,))
    out = DeviceArray.empty(out_shape, out_dtype)
    if K == 0:
        ffi.check(env.lib.pthip_memset(out.ptr, 0, out.nbytes))
    elif out.size:
        ffi.check(
            env.lib.pthip_imatmul(ffi.np_dtype_code(out_dtype), M, N, K, xm.ptr, xm.strides[0], xm.strides[1],
                                  ym.ptr, ym.strides[0], ym.strides[1], out.ptr)
        )
    return out


@handler("Dot")
def dot(node, inputs, env):
    x, y = (env.to_device(i) for i in inputs)
    out_dtype = str(env.graph.vars[node.outputs[0]].dtype)
    if np.dtype(out_dtype).kind in "iu":
        return [_int_dot(env, x, y, out_dtype)]
    # Dot.make_node accepts mixed operand dtypes and upcasts (tensor/math.py Dot; np.dot does the
    # same): f32 @ f64, int64 @ f64 reach this handler unchanged — cast to the result dtype first
    from pytensor_amd.dispatch.elemwise import _cast

    x = x if str(x.dtype) == out_dtype else _cast(env, x, out_dtype)
    y = y if str(y.dtype) == out_dtype else _cast(env, y, out_dtype)
    if x.ndim == 2 and y.ndim == 2:
        return [gemm_device(env, 1.0, _prep2d(x), _prep2d(y))]
    if x.ndim == 2 and y.ndim == 1:
        return [gemv_device(env, 1.0, x, y, 0.0, None)]
    if x.ndim == 1 and y.ndim == 2:
        yt = y.view((y.shape[1], y.shape[0]), (y.strides[1], y.strides[0]))
        return [gemv_device(env, 1.0, yt, x, 0.0, None)]
    if x.ndim == 1 and y.ndim == 1:
        if x.shape != y.shape:
            raise ValueError(f"shapes {x.shape} and {y.shape} not aligned")
        A = x.view((1, x.shape[0]), (0, x.strides[0]))
        r = gemv_device(env, 1.0, A if x.strides[0] == 1 or x.shape[0] <= 1 else A.contiguous(), y, 0.0, None)
        return [r.view((), ())]
    raise NotImplementedError("Dot with ndim > 2")


class LazySeq(DeviceArray):
    """Result buffer of a hoisted sequence product whose rows are produced on demand:
    ``produce(t0, t1)`` enqueues ``out[t0:t1] = seq[t0:t1] @ W`` on the CURRENT stream.  Only the
    ``Scan`` step driver ever sees one (``fusion.hoist_scan_seq_dots`` marks the node ``lazy`` when
    the Scan is its sole consumer): it computes the first chunk before the loop and every further
    chunk on a second stream one chunk ahead of the steps that read it — the steps are
    latency-bound (PMC: 52-66 % of the wave cycles waiting on memory, the matrix cores 25-33 % busy),
    the big product is MFMA-bound, and the two share the CUs."""

    __slots__ = ("producer", "steps", "chunk", "_keep")


@handler("SeqDot22")
def seq_dot22(node, inputs, env):
    """``out[t] = seq[t] @ W`` for every step of a Scan at once (fusion.hoist_scan_seq_dots):
    one (T·B × K)@(K × N) MFMA GEMM instead of T skinny ones."""
    seq, W = (env.to_device(i) for i in inputs)
    T, B, K = seq.shape
    if W.shape[0] != K:
        raise ValueError(f"Shape mismatch: x has {K} cols but y has {W.shape[0]} rows")
    # (T, B, K) -> (T*B, K) needs the first two dims to be mergeable
    if seq.strides[0] != B * seq.strides[1] or seq.strides[2] != 1:
        seq = seq.contiguous()
    flat = seq.view((T * B, K), (seq.strides[1], seq.strides[2]))
    N = W.shape[1]
    W2 = _prep2d(W)
    # chunks of >= 4096 rows: enough 128x128 tiles that the GEMM neither splits K (which would
    # allocate a partial-slab workspace on the side stream) nor leaves CUs without a tile
    chunk = max(1, -(-int(os.environ.get("PTHIP_SCAN_CHUNK_ROWS", "4096")) // max(B, 1)))
    lazy_ok = (node.params.get("lazy") and os.environ.get("PTHIP_SCAN_OVERLAP", "0") == "1" and env.scheduler is None
               and T >= 3 * chunk and B * N > 0 and K > 0 and (N + 127) // 128 * ((chunk * B + 127) // 128) >= 128)
    if not lazy_ok:
        out = gemm_device(env, 1.0, flat, W2)
        return [out.view((T, B, N), (B * N, N, 1))]
    buf = DeviceArray.empty((T * B, N), seq.dtype)
    out = LazySeq(buf.buf, buf.offset, (T, B, N), (B * N, N, 1), buf.dtype)
    sA, sB = flat.strides, W2.strides

    def produce(t0, t1):
        rows = (t1 - t0) * B
        if rows <= 0:
            return
        a_ptr = flat.ptr + t0 * B * sA[0] * flat.itemsize
        o_ptr = buf.ptr + t0 * B * N * buf.itemsize
        env.timed(
            f"gemm_{flat.dtype}_b1_{rows}x{N}x{K}",
            lambda: ffi.check(env.lib.pthip_gemm(_dt(flat), 1, rows, N, K, 1.0, a_ptr, 0, sA[0], sA[1], W2.ptr, 0, sB[0], sB[1],
                                                 0.0, None, 0, 0, 0, o_ptr)),
        )

    out.producer, out.steps, out.chunk = produce, T, chunk
    out._keep = (flat, W2, buf)  # operands of launches that are still to come
    return [out]


@handler("GemmPartials")
def gemm_partials(node, inputs, env):
    """``A @ B`` left as split-K slabs ``(S, M, N)`` for the Elemwise node that consumes it
    (fusion.defer_gemm_finish): that kernel adds the slabs in order and applies the Gemm's
    alpha/beta, so the separate finish launch of ``pthip_gemm`` disappears."""
    A, B = (_prep2d(env.to_device(i)) for i in inputs)
    M, K = A.shape
    K2, N = B.shape
    if K != K2:
        raise ValueError(f"Shape mismatch: x has {K} cols but y has {K2} rows")
    if str(A.dtype) != str(B.dtype):
        raise TypeError("GemmPartials: dtype mismatch")
    S = int(env.lib.pthip_gemm_nslabs(1, M, N, K)) if M and N else 1
    part = DeviceArray.empty((S, M, N), A.dtype)
    if part.size:
        ffi.check(
            env.lib.pthip_gemm_partials(_dt(A), 1, M, N, K, A.ptr, 0, A.strides[0], A.strides[1], B.ptr, 0, B.strides[0], B.strides[1], part.ptr, S)
        )
    return [part]
