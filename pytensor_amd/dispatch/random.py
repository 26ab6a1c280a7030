"""``RandomVariable`` nodes: draws from the device Philox stream (csrc/random.hip).

Reference: RandomVariable.perform (pytensor/tensor/random/op.py: ``rng_fn(rng, *params, size)``,
outputs ``(advanced rng, draws)``), size / parameter broadcasting as ``_infer_shape`` does it
(op.py), the distribution classes of random/basic.py.  The streams are this backend's own (see
``pytensor_amd/rng.py`` and SURVEY §8f row 4): what is reproduced is the distribution, the
output shape / dtype, and the threading of the generator through the graph.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from pytensor_amd import ffi
from pytensor_amd.device import DeviceArray, contiguous_strides, copy_into
from pytensor_amd.dispatch import handler
from pytensor_amd.executor import HostValue
from pytensor_amd.rng import RngState

# distribution name (RandomVariable.name) -> (code of csrc/random.hip, number of parameters)
DISTRIBUTIONS = {
    "uniform": (0, 2), "normal": (1, 2), "halfnormal": (2, 2), "lognormal": (3, 2), "exponential": (4, 1),
    "laplace": (5, 2), "logistic": (6, 2), "cauchy": (7, 2), "halfcauchy": (8, 2), "gumbel": (9, 2),
    "weibull": (10, 1), "pareto": (11, 2), "triangular": (12, 3), "gamma": (13, 2), "beta": (14, 2),
    "invgamma": (15, 2), "t": (16, 3), "bernoulli": (17, 1), "geometric": (18, 1), "poisson": (19, 1),
    "integers": (20, 2), "binomial": (21, 2), "negative_binomial": (22, 2), "wald": (23, 2), "truncexpon": (24, 3),
    "gengamma": (25, 3), "beta_binomial": (26, 3), "vonmises": (27, 2), "hypergeometric": (28, 3),
}

# vector-valued RandomVariables with a handler of their own below
STRUCTURED = ("categorical", "dirichlet", "multivariate_normal", "multinomial", "permutation", "choice_without_replacement")


def _size_tuple(env, size, size_is_none):
    if size_is_none:
        return None
    return tuple(int(s) for s in np.asarray(env.to_host(size)).ravel())


def _param_operand(x: DeviceArray, shape):
    """(array kept alive, pointer, dtype code, stride): one element broadcast (stride 0) or a
    contiguous array of the output shape (stride 1)"""
    if x.size == 1:
        return x, x.ptr, ffi.np_dtype_code(x.dtype), 0
    if tuple(x.shape) == tuple(shape) and x.is_contiguous():
        return x, x.ptr, ffi.np_dtype_code(x.dtype), 1
    full = DeviceArray.empty(shape, x.dtype)
    copy_into(full, x)
    return full, full.ptr, ffi.np_dtype_code(full.dtype), 1


def _words(a):
    arr = np.ascontiguousarray(a, dtype=np.uint64)
    return arr, arr.ctypes.data_as(C.c_void_p)


@handler("RandomVariable")
def random_variable(node, inputs, env):
    p = node.params
    rng, size, *params = inputs
    if not isinstance(rng, RngState):
        rng = RngState.from_generator(rng.a.item() if isinstance(rng, HostValue) else rng)
    name = p["name"]
    out_dtype = np.dtype(p["dtype"])
    size = _size_tuple(env, size, p["size_is_none"])
    devs = [env.to_device(x) for x in params]
    key_arr, key_ptr = _words(rng.key_words())
    ctr_arr, ctr_ptr = _words(rng.counter_words())

    if name == "categorical":
        (pr,) = devs
        k = pr.shape[-1]
        batch = tuple(pr.shape[:-1])
        shape = batch if size is None else size
        rows = int(np.prod(shape)) if shape else 1
        out = DeviceArray.empty(shape, "int64")
        if rows:
            if pr.ndim == 1 or int(np.prod(batch)) == 1:
                prc, stride = pr.contiguous(), 0
            else:
                if tuple(batch) != tuple(shape):
                    full = DeviceArray.empty((*shape, k), pr.dtype)
                    copy_into(full, pr)
                    pr = full
                prc, stride = pr.contiguous(), k
            ffi.check(env.lib.pthip_random_categorical(ffi.np_dtype_code(prc.dtype), rows, k, key_ptr, ctr_ptr, prc.ptr, stride, out.ptr))
        out = out if out_dtype == np.dtype("int64") else _cast(env, out, out_dtype)
        return [rng.advanced(rows), out]

    if name == "permutation":
        # PermutationRV.rng_fn (random/basic.py:2143: rng.permutation(x), x an int = arange(x)): a
        # uniformly random order = the argsort of n iid uniform keys (Generator.random's numbers)
        (x,) = devs
        core = int(p.get("ndims_params", [1])[0])
        if x.ndim != core:
            raise NotImplementedError("hip linker: permutation with batch dimensions")
        n = int(np.asarray(env.to_host(params[0]))) if core == 0 else x.shape[0]
        order, _ = _random_order(env, n, None, key_ptr, ctr_ptr)
        out = order if core == 0 else _take_rows(env, x, order)
        return [rng.advanced((n + 3) // 4), out if np.dtype(out.dtype) == out_dtype else _cast(env, out.contiguous(), out_dtype)]

    if name == "choice_without_replacement":
        # ChoiceWithoutReplacement.rng_fn (random/basic.py:2029: rng.choice(a, p=p, size=core_shape,
        # replace=False)).  Without p: the head of a random permutation.  With p: the smallest
        # exponential keys E_i / p_i (Efraimidis & Spirakis 2006) — successive sampling without
        # replacement proportional to p, the distribution NumPy's renormalise-and-redraw loop has.
        a, *rest = devs
        pr = rest[0] if len(rest) == 2 else None
        core = int(p.get("ndims_params", [a.ndim])[0])
        if a.ndim != core or (pr is not None and pr.ndim != 1):
            raise NotImplementedError("hip linker: choice without replacement with batch dimensions")
        core_shape = tuple(int(v) for v in np.asarray(env.to_host(inputs[-1])).ravel())
        take = int(np.prod(core_shape)) if core_shape else 1
        n = int(np.asarray(env.to_host(params[0]))) if core == 0 else a.shape[0]
        if take > n:
            raise ValueError("Cannot take a larger sample than population when replace is False")
        if pr is not None and pr.shape[0] != n:
            raise ValueError("a and p must have same size")
        order, keys = _random_order(env, n, pr, key_ptr, ctr_ptr)
        if pr is not None and take and not np.isfinite(np.asarray(env.to_host(keys.view((1,), (1,), take - 1)))).all():
            raise ValueError("Fewer non-zero entries in p than size")
        head = order.view((take,), (1,))
        out = head if core == 0 else _take_rows(env, a, head)
        shape = (*core_shape, *tuple(a.shape[1:]))
        out = out.contiguous().view(shape, contiguous_strides(shape))
        blocks = n if pr is not None else (n + 3) // 4
        return [rng.advanced(blocks), out if np.dtype(out.dtype) == out_dtype else _cast(env, out, out_dtype)]

    if name == "multinomial":
        # MultinomialRV.rng_fn (random/basic.py:1798-1812): n broadcast against p's batch dimensions
        nn, pr = devs
        k = pr.shape[-1]
        batch = np.broadcast_shapes(tuple(nn.shape), tuple(pr.shape[:-1]))
        shape = tuple(batch) if size is None else size
        rows = int(np.prod(shape)) if shape else 1
        out = DeviceArray.empty((*shape, k), "int64")
        if rows and k:
            nk, nptr, ndt, nst = _param_operand(nn, shape)
            if pr.ndim == 1 or int(np.prod(pr.shape[:-1])) == 1:
                prc, stride = pr.contiguous(), 0
            else:
                if tuple(pr.shape[:-1]) != tuple(shape):
                    full = DeviceArray.empty((*shape, k), pr.dtype)
                    copy_into(full, pr)
                    pr = full
                prc, stride = pr.contiguous(), k
            ffi.check(env.lib.pthip_random_multinomial(ffi.np_dtype_code(prc.dtype), rows, k, key_ptr, ctr_ptr, nptr, ndt, nst,
                                                       prc.ptr, stride, out.ptr))
            env.keepalive.extend((nk, prc))
        out = out if out_dtype == np.dtype("int64") else _cast(env, out, out_dtype)
        return [rng.advanced(rows), out]

    if name == "dirichlet":
        # DirichletRV (random/basic.py:942): unit-scale gammas normalised along the last axis
        (al,) = devs
        k = al.shape[-1]
        shape = (*(tuple(al.shape[:-1]) if size is None else size), k)
        n = int(np.prod(shape))
        g = _draw(env, "gamma", [al, env.to_device(HostValue(np.asarray(1.0)))], shape, np.dtype("float64"), key_ptr, ctr_ptr)
        return [rng.advanced(n), _normalise_rows(env, g, out_dtype)]

    if name == "multivariate_normal":
        # MvNormalRV.rng_fn (random/basic.py:914-936): mean + A z, z ~ N(0, I), A A^T = cov with A the
        # Cholesky factor, U sqrt(s) (svd) or V sqrt(w) (eigh)
        mean, cov = devs
        method = p.get("method", "cholesky")
        if cov.ndim != 2 or mean.ndim < 1:
            raise NotImplementedError("hip linker: multivariate_normal with a batched covariance")
        from pytensor_amd.dispatch.blas import gemm_device
        from pytensor_amd.dispatch.linalg import cholesky_device

        k = mean.shape[-1]
        lead = tuple(mean.shape[:-1]) if size is None else size  # rng_fn 915-916: batch shape of the parameters
        if np.broadcast_shapes(tuple(mean.shape[:-1]), lead) != tuple(lead):
            raise ValueError(f"multivariate_normal: size {lead} does not match the batch shape of mean {tuple(mean.shape[:-1])}")
        rows = int(np.prod(lead)) if lead else 1
        fdt = np.dtype(cov.dtype)
        zero, one = (env.to_device(HostValue(np.asarray(v, dtype=fdt))) for v in (0.0, 1.0))
        z = _draw(env, "normal", [zero, one], (rows, k), fdt, key_ptr, ctr_ptr)
        if method == "cholesky":
            L = cholesky_device(env, cov, True)
        else:
            from pytensor_amd.dispatch.elemwise import launch_elemwise

            if method == "svd":
                from pytensor_amd.dispatch.decomp import svd_device

                M, d, _ = svd_device(env, cov, True, True)
            else:
                from pytensor_amd.dispatch.lu import eigh as eigh_handler

                d, M = eigh_handler(type("_N", (), {"params": {"lower": True}}), [cov], env)
            dt = str(fdt)
            body = {"in_dtypes": [dt, dt], "out_dtypes": [dt],
                    "body": [{"op": "Sqrt", "in": [["i", 1]], "dtype": dt}, {"op": "Mul", "in": [["i", 0], ["t", 0]], "dtype": dt}], "outs": [["t", 1]]}
            (L,), _, _ = launch_elemwise(body, [M, d.view((k, k), (0, d.strides[0]))], (k, k), [dt], None, env)
        m2 = mean if str(mean.dtype) == str(fdt) else _cast(env, mean, fdt)
        if m2.ndim == 1:
            mrows = m2.view((1, k), (0, m2.strides[0]))
        else:  # a mean per draw: broadcast to the batch shape, one row each
            full = DeviceArray.empty((*lead, k), fdt)
            lead_pad = len(lead) + 1 - m2.ndim
            copy_into(full, m2.view((*lead, k), (0,) * lead_pad + tuple(0 if (sdim == 1 and t != 1) else st for sdim, st, t in zip(m2.shape, m2.strides, (*lead, k)[lead_pad:]))))
            mrows = full.view((rows, k), (k, 1))
        out = gemm_device(env, 1.0, z, L.view((k, k), (L.strides[1], L.strides[0])), 1.0, mrows)
        out = out.view((*lead, k), contiguous_strides((*lead, k)))
        return [rng.advanced(rows * k), out if out_dtype == fdt else _cast(env, out, out_dtype)]

    if name not in DISTRIBUTIONS:
        raise NotImplementedError(f"hip linker: no device sampler for the {name!r} RandomVariable")
    code, nparams = DISTRIBUTIONS[name]
    if len(devs) != nparams:
        raise TypeError(f"{name}: expected {nparams} parameters, got {len(devs)}")
    bshape = np.broadcast_shapes(*[tuple(d.shape) for d in devs]) if devs else ()
    if size is None:
        shape = tuple(bshape)
    else:
        shape = size
        if np.broadcast_shapes(bshape, shape) != tuple(shape):
            raise ValueError(f"{name}: size {shape} does not match the parameter batch shape {tuple(bshape)}")
    n = int(np.prod(shape)) if shape else 1
    kernel_dtype = out_dtype if out_dtype.name in ("float64", "float32", "int64") else np.dtype("int64" if out_dtype.kind in "iub" else "float64")
    out = DeviceArray.empty(shape, kernel_dtype)
    if n:
        ops = [_param_operand(d, shape) for d in devs]
        ptrs = (C.c_void_p * 3)(*[o[1] for o in ops], *([None] * (3 - len(ops))))
        dts = (C.c_int * 3)(*[o[2] for o in ops], *([0] * (3 - len(ops))))
        sts = (C.c_int64 * 3)(*[o[3] for o in ops], *([0] * (3 - len(ops))))
        ffi.check(env.lib.pthip_random(code, ffi.np_dtype_code(kernel_dtype), n, key_ptr, ctr_ptr, len(ops),
                                       C.cast(ptrs, C.c_void_p), C.cast(dts, C.c_void_p), C.cast(sts, C.c_void_p), out.ptr))
        env.keepalive.extend(o[0] for o in ops)
    if kernel_dtype != out_dtype:
        out = _cast(env, out, out_dtype)
    blocks = (n + 3) // 4 if name == "uniform" else n
    return [rng.advanced(blocks), out]


def _draw(env, name, devs, shape, kernel_dtype, key_ptr, ctr_ptr) -> DeviceArray:
    """one pthip_random launch: per-element draws of ``name`` into a fresh contiguous array"""
    code, _ = DISTRIBUTIONS[name]
    n = int(np.prod(shape)) if shape else 1
    out = DeviceArray.empty(shape, kernel_dtype)
    if n:
        ops = [_param_operand(d, shape) for d in devs]
        ptrs = (C.c_void_p * 3)(*[o[1] for o in ops], *([None] * (3 - len(ops))))
        dts = (C.c_int * 3)(*[o[2] for o in ops], *([0] * (3 - len(ops))))
        sts = (C.c_int64 * 3)(*[o[3] for o in ops], *([0] * (3 - len(ops))))
        ffi.check(env.lib.pthip_random(code, ffi.np_dtype_code(kernel_dtype), n, key_ptr, ctr_ptr, len(ops),
                                       C.cast(ptrs, C.c_void_p), C.cast(dts, C.c_void_p), C.cast(sts, C.c_void_p), out.ptr))
        env.keepalive.extend(o[0] for o in ops)
    return out


def _random_order(env, n, weights, key_ptr, ctr_ptr):
    """(argsort of n random keys, the sorted keys): uniform keys, or Exp(1) / weight keys"""
    from pytensor_amd.dispatch.elemwise import launch_elemwise

    one = env.to_device(HostValue(np.asarray(1.0)))
    if weights is None:
        keys = _draw(env, "uniform", [env.to_device(HostValue(np.asarray(0.0))), one], (n,), np.dtype("float64"), key_ptr, ctr_ptr)
    else:
        e = _draw(env, "exponential", [one], (n,), np.dtype("float64"), key_ptr, ctr_ptr)
        wdt = str(weights.dtype)
        body = {"in_dtypes": ["float64", wdt], "out_dtypes": ["float64"],
                "body": [{"op": "Cast", "in": [["i", 1]], "dtype": "float64"}, {"op": "TrueDiv", "in": [["i", 0], ["t", 0]], "dtype": "float64"}],
                "outs": [["t", 1]]}
        (keys,), _, _ = launch_elemwise(body, [e, weights], (n,), ["float64"], None, env)
    order = DeviceArray.empty((n,), "int64")
    vals = DeviceArray.empty((n,), "float64")
    if n:
        ffi.check(env.lib.pthip_sort(ffi.np_dtype_code(np.dtype("float64")), 1, n, keys.contiguous().ptr, vals.ptr, order.ptr))
    env.keepalive.append(keys)
    return order, vals


def _take_rows(env, x: DeviceArray, order: DeviceArray) -> DeviceArray:
    """x[order] along axis 0 (pthip_take_rows)"""
    x = x.contiguous()
    rest = tuple(x.shape[1:])
    inner = int(np.prod(rest)) if rest else 1
    out = DeviceArray.empty((order.size, *rest), x.dtype)
    if out.size:
        ffi.check(env.lib.pthip_take_rows(x.itemsize, order.size, inner, x.ptr, x.shape[0], inner, order.ptr, out.ptr))
    return out


def _normalise_rows(env, g: DeviceArray, out_dtype) -> DeviceArray:
    """g / g.sum(-1, keepdims=True) of a contiguous array"""
    from pytensor_amd.dispatch.elemwise import device_reduce, launch_elemwise

    k = g.shape[-1]
    rows = g.size // k if k else 0
    dt = str(g.dtype)
    if not g.size:
        return DeviceArray.empty(g.shape, out_dtype)
    sums = device_reduce(env, "Add", g, rows, k, 1, k, 1, 0, dt, dt, (rows,))
    body = {"in_dtypes": [dt, dt], "out_dtypes": [str(out_dtype)],
            "body": [{"op": "TrueDiv", "in": [["i", 0], ["i", 1]], "dtype": dt}, {"op": "Cast", "in": [["t", 0]], "dtype": str(out_dtype)}],
            "outs": [["t", 1]]}
    g2 = g.view((rows, k), (k, 1))
    outs, _, _ = launch_elemwise(body, [g2, sums.view((rows, 1), (1, 0))], (rows, k), [str(out_dtype)], None, env)
    return outs[0].view(g.shape, contiguous_strides(g.shape))


def _cast(env, x: DeviceArray, dtype) -> DeviceArray:
    from pytensor_amd.dispatch.elemwise import launch_elemwise

    body = {"in_dtypes": [str(x.dtype)], "out_dtypes": [str(dtype)],
            "body": [{"op": "Cast", "in": [["i", 0]], "dtype": str(dtype)}], "outs": [["t", 0]]}
    outs, _, _ = launch_elemwise(body, [x], x.shape, [str(dtype)], None, env)
    return outs[0].view(x.shape, contiguous_strides(x.shape))
