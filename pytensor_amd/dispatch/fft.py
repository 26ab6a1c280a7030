"""``RFFTOp`` / ``IRFFTOp`` (pytensor/tensor/fft.py:11, 76): real FFTs over the trailing axes of a
batch, real and imaginary parts stacked on a last axis of length 2.

Reference: perform 39-48 (``np.fft.rfftn(a, s)``), 109-117 (``np.fft.irfftn(inp, s) * prod(s)``).
Correct-first tier: every transformed axis is a dense DFT on the MFMA GEMM path — the (exactly
reduced) cosine / sine tables are generated on the device, a real axis costs two products, a complex
axis four.  O(n^2) per axis instead of O(n log n): right for the sizes of spectral features inside a
model graph (n up to a few thousand), refused above ``MAX_N``.
"""

from __future__ import annotations

import numpy as np

from pytensor_amd import ffi
from pytensor_amd.device import DeviceArray, contiguous_strides, copy_into
from pytensor_amd.dispatch import handler
from pytensor_amd.dispatch.linalg import _require_float

MAX_N = 16384


def _cs(shape):
    return contiguous_strides(tuple(shape))


def _ew(env, ops, ins, in_dtypes, out_dtype, shape):
    from pytensor_amd.dispatch.elemwise import launch_elemwise

    body = {"in_dtypes": list(in_dtypes), "out_dtypes": [out_dtype], "body": ops, "outs": [["t", len(ops) - 1]]}
    (out,), _, _ = launch_elemwise(body, ins, tuple(shape), [out_dtype], None, env)
    return out


def _iota(env, n):
    out = DeviceArray.empty((n,), "int64")
    ffi.check(env.lib.pthip_arange(ffi.np_dtype_code(np.dtype("int64")), n, 0.0, 1.0, 0, 1, out.ptr))
    return out


def _tables(env, n, rows, cols, dt, weights=None):
    """(cos, sin)(2 pi ((t f) mod n) / n) as (rows x cols) matrices of dtype ``dt``; the angle is
    reduced exactly in integers, sines of multiples of pi are exact zeros.  ``weights`` (host vector
    over the row index) scales both (the 1, 2, ..., 2, 1 of a complex-to-real transform)."""
    if n > MAX_N:
        raise NotImplementedError(f"hip linker: FFT axis of length {n} (dense-DFT tier, up to {MAX_N})")
    t = _iota(env, rows).view((rows, cols), (1, 0))
    f = _iota(env, cols).view((rows, cols), (0, 1))
    two_pi_n = float(2.0 * np.pi / n).hex()
    base = [{"op": "Mul", "in": [["i", 0], ["i", 1]], "dtype": "int64"},
            {"op": "Mod", "in": [["t", 0], ["c", n, "int64"]], "dtype": "int64"},
            {"op": "Cast", "in": [["t", 1]], "dtype": "float64"},
            {"op": "Mul", "in": [["t", 2], ["c", two_pi_n, "float64"]], "dtype": "float64"}]
    cosm = _ew(env, base + [{"op": "Cos", "in": [["t", 3]], "dtype": "float64"}], [t, f], ["int64", "int64"], "float64", (rows, cols))
    # sin: exactly 0 where 2 m is a multiple of n
    sin_ops = base + [{"op": "Sin", "in": [["t", 3]], "dtype": "float64"},
                      {"op": "Mul", "in": [["t", 1], ["c", 2, "int64"]], "dtype": "int64"},
                      {"op": "Mod", "in": [["t", 5], ["c", n, "int64"]], "dtype": "int64"},
                      {"op": "EQ", "in": [["t", 6], ["c", 0, "int64"]], "dtype": "bool"},
                      {"op": "Switch", "in": [["t", 7], ["c", (0.0).hex(), "float64"], ["t", 4]], "dtype": "float64"}]
    sinm = _ew(env, sin_ops, [t, f], ["int64", "int64"], "float64", (rows, cols))
    if weights is not None:
        from pytensor_amd.executor import HostValue

        w = env.to_device(HostValue(np.asarray(weights, dtype="float64"))).view((rows, cols), (1, 0))
        mul = [{"op": "Mul", "in": [["i", 0], ["i", 1]], "dtype": "float64"}]
        cosm = _ew(env, mul, [cosm, w], ["float64"] * 2, "float64", (rows, cols))
        sinm = _ew(env, mul, [sinm, w], ["float64"] * 2, "float64", (rows, cols))
    if np.dtype(dt) != np.dtype("float64"):
        from pytensor_amd.dispatch.elemwise import _cast

        cosm, sinm = _cast(env, cosm, dt), _cast(env, sinm, dt)
    return cosm, sinm


def _fit_axis(env, x: DeviceArray, axis: int, n: int) -> DeviceArray:
    """``x`` truncated or zero-padded to length ``n`` along ``axis`` (what ``s`` does to the input)"""
    cur = x.shape[axis]
    if cur == n:
        return x
    if cur > n:
        shape = list(x.shape)
        shape[axis] = n
        return x.view(shape, x.strides)
    shape = list(x.shape)
    shape[axis] = n
    out = DeviceArray.empty(shape, x.dtype)
    if out.size:
        ffi.check(env.lib.pthip_memset(out.ptr, 0, out.nbytes))
    if x.size:
        copy_into(out.view(x.shape, out.strides), x)
    return out


def _to_last(x: DeviceArray, axis: int) -> DeviceArray:
    order = [d for d in range(x.ndim) if d != axis] + [axis]
    return x.view([x.shape[d] for d in order], [x.strides[d] for d in order])


def _from_last(x: DeviceArray, axis: int) -> DeviceArray:
    nd = x.ndim
    order = list(range(axis)) + [nd - 1] + list(range(axis, nd - 1))
    return x.view([x.shape[d] for d in order], [x.strides[d] for d in order])


def _rows(x: DeviceArray):
    """contiguous (rows, n) matrix of an array whose transformed axis is last"""
    xc = x.contiguous()
    n = xc.shape[-1]
    r = xc.size // n if n else 0
    return xc.view((r, n), (n, 1)), xc.shape


def _complex_axis(env, re, im, axis, n, sign):
    """DFT of length ``n`` along ``axis`` of re + i im with kernel cos(a) + i sign sin(a): forward
    sign = -1 (re' = re C + im S, im' = im C - re S), inverse sign = +1"""
    from pytensor_amd.dispatch.blas import gemm_device

    re, im = _fit_axis(env, re, axis, n), _fit_axis(env, im, axis, n)
    C, S = _tables(env, n, n, n, re.dtype)
    rm, shp = _rows(_to_last(re, axis))
    imm, _ = _rows(_to_last(im, axis))
    t1 = gemm_device(env, 1.0, rm, C)
    new_re = gemm_device(env, -float(sign), imm, S, 1.0, t1)
    t2 = gemm_device(env, 1.0, imm, C)
    new_im = gemm_device(env, float(sign), rm, S, 1.0, t2)
    back = lambda m: _from_last(m.view(shp, _cs(shp)), axis)
    return back(new_re), back(new_im)


def _s_tuple(env, s, k):
    vals = [int(v) for v in np.asarray(env.to_host(s)).ravel()]
    if len(vals) > k:
        raise ValueError("Shape and axes have different lengths.")
    if any(v < 1 for v in vals):
        raise ValueError(f"Invalid number of FFT data points ({vals}) specified.")
    return vals


@handler("RFFTOp")
def rfft(node, inputs, env):
    from pytensor_amd.dispatch.blas import gemm_device

    a = env.to_device(inputs[0])
    _require_float(a, "RFFTOp")
    s = _s_tuple(env, inputs[1], a.ndim)
    axes = list(range(a.ndim - len(s), a.ndim))
    n = s[-1]
    h = n // 2 + 1
    x = _fit_axis(env, a, axes[-1], n)
    for ax, m in zip(axes[:-1], s[:-1]):
        x = _fit_axis(env, x, ax, m)
    xm, shp = _rows(x)
    C, S = _tables(env, n, n, h, a.dtype)
    oshape = (*shp[:-1], h)
    re = gemm_device(env, 1.0, xm, C).view(oshape, _cs(oshape))
    im = gemm_device(env, -1.0, xm, S).view(oshape, _cs(oshape))
    for ax, m in zip(axes[:-1], s[:-1]):
        re, im = _complex_axis(env, re, im, ax, m, -1)
    out = DeviceArray.empty((*re.shape, 2), a.dtype)
    if out.size:
        st = out.strides[:-1]
        copy_into(out.view(re.shape, st), re)
        copy_into(out.view(re.shape, st, 1), im)
    return [out]


@handler("IRFFTOp")
def irfft(node, inputs, env):
    from pytensor_amd.dispatch.blas import gemm_device

    a = env.to_device(inputs[0])
    _require_float(a, "IRFFTOp")
    if a.shape[-1] != 2:
        raise ValueError("IRFFTOp: the last axis holds the real and imaginary parts")
    core = a.ndim - 1
    s = _s_tuple(env, inputs[1], core)
    axes = list(range(core - len(s), core))
    part = lambda k: a.view(a.shape[:-1], a.strides[:-1], k * a.strides[-1])
    re, im = part(0), part(1)
    for ax, m in zip(axes[:-1], s[:-1]):
        re, im = _complex_axis(env, re, im, ax, m, +1)
    n = s[-1]
    h = n // 2 + 1
    re, im = _fit_axis(env, re, axes[-1], h), _fit_axis(env, im, axes[-1], h)
    w = np.full(h, 2.0)
    w[0] = 1.0
    if n % 2 == 0:
        w[-1] = 1.0
    C, S = _tables(env, n, h, n, a.dtype, weights=w)
    rm, shp = _rows(re)
    imm, _ = _rows(im)
    t1 = gemm_device(env, 1.0, rm, C)
    out = gemm_device(env, -1.0, imm, S, 1.0, t1)
    oshape = (*shp[:-1], n)
    return [out.view(oshape, _cs(oshape))]
