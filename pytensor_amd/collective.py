"""``all_reduce`` as a PyTensor ``Op`` (needs PyTensor): the explicit collective of the ``hip`` mode.

Not a reference Op — the reference has no distributed layer; the shape follows its ``Op``
protocol (pytensor/graph/op.py: ``make_node`` / ``perform`` / ``pullback``-style gradient /
``infer_shape``) so that the node lives in ordinary graphs, is rewritten around like any other
``Apply`` and is lowered by ``HipLinker`` to the IR node ``AllReduce`` (dispatch/extra.py →
``comm.all_reduce_device``: RCCL over xGMI on the device buffer).  Under every other linker
``perform`` runs the same reduction on host arrays (gloo or RCCL through a staging tensor).
"""

from __future__ import annotations

from pytensor.graph.basic import Apply
from pytensor.graph.op import Op
from pytensor.tensor.basic import as_tensor_variable

from pytensor_amd import comm


class AllReduce(Op):
    """``out = reduce_over_ranks(x)``, same shape and dtype on every rank."""

    __props__ = ("op",)

    def __init__(self, op: str = "sum"):
        if op not in comm.OPS:
            raise ValueError(f"AllReduce: unknown reduction {op!r} (one of {comm.OPS})")
        self.op = op

    def make_node(self, x):
        x = as_tensor_variable(x)
        if x.type.dtype not in comm.DTYPES:
            raise TypeError(f"all_reduce: dtype {x.type.dtype} has no collective on every transport (supported: {', '.join(comm.DTYPES)})")
        return Apply(self, [x], [x.type()])

    def perform(self, node, inputs, output_storage):
        output_storage[0][0] = comm.all_reduce_host(inputs[0], self.op)

    def infer_shape(self, node, input_shapes):
        return [input_shapes[0]]

    def L_op(self, inputs, outputs, output_grads):
        # y = sum_r x_r on every rank; each rank holds its own cost L_r(y): dL/dx_r = sum_r' dL_r'/dy
        if self.op != "sum":
            raise NotImplementedError(f"gradient of all_reduce({self.op!r})")
        return [AllReduce("sum")(output_grads[0])]

    pullback = L_op


def all_reduce(x, op: str = "sum"):
    """Sum / product / max / min of ``x`` over all ranks of the job (identity on one rank)."""
    return AllReduce(op)(x)
