"""Shape / alloc / copy / assert ops (bit-exact tier).

Reference: pytensor/tensor/basic.py (Alloc 1545, AllocEmpty 4197, MakeVector 1900,
Join 2405, ExtractDiag 3636, ScalarFromTensor 684, TensorFromScalar 627),
pytensor/tensor/shape.py (Shape 71, Shape_i 201, SpecifyShape 384, Reshape 613),
pytensor/raise_op.py:26 (CheckAndRaise), pytensor/compile/ops.py (ViewOp 87,
DeepCopyOp 121).
"""

from __future__ import annotations

import numpy as np

from pytensor_amd.device import DeviceArray, contiguous_strides, copy_into
from pytensor_amd.dispatch import handler
from pytensor_amd.executor import HOST_MAX, HostValue


def _ints(env, vals):
    return [int(env.to_host(v)) for v in vals]


@handler("Shape_i")
def shape_i(node, inputs, env):
    return [HostValue(np.asarray(inputs[0].shape[node.params["i"]], dtype="int64"))]


@handler("Shape")
def shape(node, inputs, env):
    return [HostValue(np.asarray(inputs[0].shape, dtype="int64"))]


@handler("MakeVector")
def make_vector(node, inputs, env):
    dt = node.params["dtype"]
    if all(isinstance(i, HostValue) for i in inputs):
        return [HostValue(np.asarray([i.a for i in inputs], dtype=dt).reshape(len(inputs)))]
    out = DeviceArray.empty((len(inputs),), dt)
    for k, v in enumerate(inputs):
        v = env.to_device(v)
        if str(v.dtype) != dt:
            from pytensor_amd.dispatch.elemwise import _cast

            v = _cast(env, v, dt)
        copy_into(out.view((1,), (1,), k), v.view((1,), (0,)))
    return [out]


@handler("ScalarFromTensor")
def scalar_from_tensor(node, inputs, env):
    (x,) = inputs
    if isinstance(x, HostValue):
        return [HostValue(np.asarray(x.a).reshape(()))]
    return [x.view((), ())]


@handler("TensorFromScalar")
def tensor_from_scalar(node, inputs, env):
    (x,) = inputs
    if isinstance(x, HostValue):
        return [HostValue(np.asarray(x.a).reshape(()))]
    return [x.view((), ())]


@handler("ViewOp", "SpecifyShape")
def view_op(node, inputs, env):
    x = inputs[0]
    if node.op == "SpecifyShape":
        for d, s in enumerate(inputs[1:]):
            if isinstance(s, HostValue) and s.a.dtype == object:
                continue
            sv = env.to_host(s)
            if sv is not None and x.shape[d] != int(sv):
                raise AssertionError(f"SpecifyShape: dim {d} of input has shape {x.shape[d]}, expected {int(sv)}.")
    return [x]


@handler("DeepCopyOp")
def deep_copy(node, inputs, env):
    (x,) = inputs
    if isinstance(x, HostValue):
        return [HostValue(x.a.copy())]
    out = DeviceArray.empty(x.shape, x.dtype)
    copy_into(out, x)
    return [out]


@handler("Alloc")
def alloc(node, inputs, env):
    v, *shape = inputs
    shape = tuple(_ints(env, shape))
    v = env.to_device(v)
    out = DeviceArray.empty(shape, v.dtype)
    copy_into(out, v)
    return [out]


@handler("AllocEmpty")
def alloc_empty(node, inputs, env):
    # contents unspecified in the reference (basic.py:4197+); pooled memory is reused as is
    return [DeviceArray.empty(tuple(_ints(env, inputs)), node.params["dtype"])]


@handler("Reshape")
def reshape(node, inputs, env):
    x, shp = inputs
    new = [int(s) for s in np.asarray(env.to_host(shp)).ravel()]
    if isinstance(x, HostValue):
        return [HostValue(x.a.reshape(new))]
    n = x.size
    if new.count(-1) > 1:
        raise ValueError("can only specify one unknown dimension")
    if -1 in new:
        known = int(np.prod([s for s in new if s != -1])) if len(new) > 1 else 1
        if known == 0 or n % known:
            raise ValueError(f"cannot reshape array of size {n} into shape {tuple(new)}")
        new[new.index(-1)] = n // known
    if int(np.prod(new)) != n:
        raise ValueError(f"cannot reshape array of size {n} into shape {tuple(new)}")
    xc = x.contiguous()  # NumPy copies too when the layout does not allow a view
    return [xc.view(new, contiguous_strides(new))]


@handler("ExtractDiag")
def extract_diag(node, inputs, env):
    (x,) = inputs
    x = env.to_device(x)
    off, a1, a2 = node.params["offset"], node.params["axis1"], node.params["axis2"]
    n1, n2 = x.shape[a1], x.shape[a2]
    if off >= 0:
        ln = max(0, min(n1, n2 - off))
        start = off * x.strides[a2]
    else:
        ln = max(0, min(n1 + off, n2))
        start = -off * x.strides[a1]
    keep = [d for d in range(x.ndim) if d not in (a1, a2)]
    shape = [x.shape[d] for d in keep] + [ln]
    strides = [x.strides[d] for d in keep] + [x.strides[a1] + x.strides[a2]]
    return [x.view(shape, strides, start if ln else 0)]


@handler("Join")
def join(node, inputs, env):
    axis = node.params["axis"]
    tensors = [env.to_device(t) for t in inputs]
    nd = tensors[0].ndim
    for t in tensors:
        if t.ndim != nd:
            raise TypeError("Only tensors with the same number of dimensions can be joined")
        for d in range(nd):
            if d != axis and t.shape[d] != tensors[0].shape[d]:
                raise ValueError(
                    f"all the input array dimensions except for the concatenation axis must match exactly, "
                    f"but along dimension {d}, got {tensors[0].shape[d]} and {t.shape[d]}"
                )
    total = sum(t.shape[axis] for t in tensors)
    shape = list(tensors[0].shape)
    shape[axis] = total
    # Join.make_node upcasts its operands to their common dtype (basic.py: `as_tensor_variable_args` ->
    # `ps.upcast`); the output variable carries it
    odt = np.dtype(env.graph.vars[node.outputs[0]].dtype) if node.outputs[0] in env.graph.vars else np.result_type(*[t.dtype for t in tensors])
    out = DeviceArray.empty(shape, odt)
    off = 0
    for t in tensors:
        if t.size:
            if np.dtype(t.dtype) != odt:
                from pytensor_amd.dispatch.elemwise import _cast

                t = _cast(env, t, odt)
            sub = out.view(t.shape, out.strides, off * out.strides[axis])
            copy_into(sub, t)
        off += t.shape[axis]
    return [out]


@handler("Split")
def split(node, inputs, env):
    # Split.perform (pytensor/tensor/basic.py:2268-2283): np.split at the running sums of
    # `splits` — views of the input (view_map 2237), no data moves
    x, splits = inputs
    axis = node.params["axis"]
    sizes = [int(v) for v in np.asarray(env.to_host(splits)).ravel()]
    if len(sizes) != node.params["len_splits"]:
        raise ValueError("Length of splits is not equal to n_splits")
    x = env.to_device(x)
    if sum(sizes) != x.shape[axis]:
        raise ValueError(f"Split sizes sum to {sum(sizes)}; expected {x.shape[axis]}")
    if any(v < 0 for v in sizes):
        raise ValueError("Split sizes cannot be negative")
    outs, off = [], 0
    for v in sizes:
        shape = list(x.shape)
        shape[axis] = v
        outs.append(x.view(shape, x.strides, off * x.strides[axis] if v else 0))
        off += v
    return outs


@handler("CheckAndRaise")
def check_and_raise(node, inputs, env):
    x, *conds = inputs
    for c in conds:
        cv = env.to_host(c)
        if not np.all(cv):
            raise _exception_class(node.params["exc_type"])(node.params["msg"])
    return [x]


def _exception_class(name: str):
    """The exception class a ``CheckAndRaise`` was built with (raise_op.py:26), by name: the IR is
    serialisable, so the class travels as a string — builtins and the NumPy ``LinAlgError`` that
    ``cholesky(on_error="raise")`` / ``solve`` checks use (linalg/decomposition/cholesky.py:194)."""
    import builtins

    if name == "LinAlgError":
        return np.linalg.LinAlgError
    cls = getattr(builtins, name, None)
    return cls if isinstance(cls, type) and issubclass(cls, BaseException) else RuntimeError


@handler("HostPerform")
def host_perform(node, inputs, env):
    """D2H → ``Op.perform`` → H2D for ops without a device kernel (live PyTensor only),
    cf. numba's object-mode fallback (link/numba/dispatch/basic.py:228-263)."""
    op, pnode = node.params["op"], node.params["node"]
    host_in = []
    for v, var in zip(inputs, pnode.inputs):
        a = env.to_host(v)
        host_in.append(a if getattr(var.type, "ndim", None) is not None else a[()])
    storage = [[None] for _ in pnode.outputs]
    op.perform(pnode, host_in, storage)
    outs = []
    for cell in storage:
        a = np.asarray(cell[0])
        outs.append(HostValue(a) if a.dtype.kind in "iub" and a.size <= HOST_MAX and a.ndim <= 1 else env.to_device(HostValue(a)))
    return outs


@handler("IfElse")
def ifelse(node, inputs, env):
    """``IfElse`` (pytensor/ifelse.py:42): the executor has read the condition and run only the
    branch taken (executor.py ``branch_guards``); what is left is handing that branch's values on."""
    n_out = len(node.outputs)
    b = env.branch
    return list(inputs[1 + b * n_out : 1 + (b + 1) * n_out])
