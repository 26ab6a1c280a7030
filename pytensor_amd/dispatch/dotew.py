"""``DotEpilogue`` / ``PackB16`` — a skinny product and the ``Composite`` that consumes it, one launch.

Produced by ``gemmfuse.fuse_dot_epilogue`` inside ``Scan`` steps: ``out = body(.., A_d @ W_d, ..)``
with ``W_d`` loop constants.  Reference ops restated: ``Dot22``/``Gemm`` (pytensor/tensor/blas/
gemm.py:76,248 — the alpha/beta epilogue already lives in ``body``) followed by ``Elemwise``
(pytensor/tensor/elemwise.py:755).  The generated kernel (codegen.dot_epilogue_source) covers
``K % 16 == 0``, rows of A 16-byte aligned, M up to ``MAX_ROWS``; anything else runs the plain
MFMA GEMM followed by the ordinary elementwise kernel — same values, two launches per product.
"""

from __future__ import annotations

import os
import struct

from pytensor_amd import codegen, ffi, kernel_cache
from pytensor_amd.device import DeviceArray
from pytensor_amd.dispatch import handler
from pytensor_amd.dispatch.blas import _prep2d, gemm_device
from pytensor_amd.dispatch.elemwise import _body_key, _placed, _scalar_bits, _scalar_or_device, launch_elemwise
from pytensor_amd.executor import HostValue

MAX_ROWS = 1024  # above this the 16x16-tile/full-K scheme re-reads too much; GEMM + Elemwise


def _ceil16(n: int) -> int:
    return (n + 15) // 16 * 16


@handler("PackB16")
def pack_b16(node, inputs, env):
    B = env.to_device(inputs[0])
    if B.ndim != 2 or B.itemsize not in (4, 8):
        raise TypeError(f"PackB16: expected a float32/float64 matrix, got {B.dtype} with {B.ndim} dims")
    K, N = B.shape
    out = DeviceArray.empty((_ceil16(K) * _ceil16(N),), B.dtype)
    if out.size:
        ffi.check(env.lib.pthip_pack_b16(B.itemsize, K, N, B.ptr, B.strides[0], B.strides[1], out.ptr))
    return [out]


@handler("DotEpilogue")
def dot_epilogue(node, inputs, env):
    body = node.params["scalar"]
    dpos = list(node.params["dot_inputs"])
    nb = len(body["in_dtypes"])
    plain = node.params.get("plain")
    if plain is not None:
        # a product moved here because it shares this node's left operand (gemmfuse.
        # _hoist_shared_left_operand): only when its shape equals the node's own can it ride in the
        # same tiles; otherwise it is a plain GEMM and the node runs in its original form
        q = plain["moved"][0]
        Am, Bm = env.to_device(inputs[q]), env.to_device(inputs[nb + 2 * (len(dpos) - 1)])
        q0 = plain["dot_inputs"][0]
        A0, B0 = env.to_device(inputs[q0]), env.to_device(inputs[nb])
        if (Am.shape[0], Bm.shape[1]) != (A0.shape[0], B0.shape[1]) or Am.shape[1] != A0.shape[1]:
            nbp = len(plain["scalar"]["in_dtypes"])
            sub = type("_Plain", (), {"params": {"scalar": plain["scalar"], "dot_inputs": plain["dot_inputs"]},
                                      "inputs": list(node.inputs[:nbp]) + list(node.inputs[nb : nb + 2 * len(plain["dot_inputs"])]),
                                      "outputs": list(node.outputs[: plain["moved"][1]])})
            outs = dot_epilogue(sub, list(inputs[:nbp]) + list(inputs[nb : nb + 2 * len(plain["dot_inputs"])]), env)
            return list(outs) + [gemm_device(env, 1.0, _prep2d(Am), _prep2d(Bm))]
    ins = [_scalar_or_device(env, i) for i in inputs[:nb]]
    extra = inputs[nb:]
    dots = {}
    shape = None
    for j, q in enumerate(dpos):
        A, B, Bp = (env.to_device(v) for v in (ins[q], extra[2 * j], extra[2 * j + 1]))
        if A.ndim != 2 or B.ndim != 2:
            raise TypeError("DotEpilogue: operands of a product must be matrices")
        if A.shape[1] != B.shape[0]:
            raise ValueError(f"Shape mismatch: x has {A.shape[1]} cols but y has {B.shape[0]} rows")
        if shape is not None and shape != (A.shape[0], B.shape[1]):
            raise ValueError(f"Incompatible Elemwise input shapes {[shape, (A.shape[0], B.shape[1])]}")
        shape = (A.shape[0], B.shape[1])
        dots[q] = (A, B, Bp)
    M, N = shape
    for k, a in enumerate(ins):
        if k in dots or (isinstance(a, HostValue) and a.a.size == 1):
            continue
        if a.ndim != 2 or any(a.shape[d] not in (1, shape[d]) for d in range(2)):
            raise ValueError(f"Incompatible Elemwise input shapes {[shape] + [tuple(i.shape) for i in ins if not isinstance(i, HostValue)]}")
        static = env.graph.vars[node.inputs[k]].shape
        for d in range(2):
            if a.shape[d] == 1 and shape[d] != 1 and static[d] != 1:
                raise ValueError(
                    f"Runtime broadcasting not allowed. One input had a distinct dimension length of 1 along axis {d}, "
                    "but the static type does not mark it as broadcastable"
                )
    out_dtypes = body["out_dtypes"]
    placed = _placed(env, node, shape, out_dtypes)
    T = body["in_dtypes"][dpos[0]]
    K = dots[dpos[0]][0].shape[1]
    fast = (
        M > 0 and N > 0 and 0 < K <= codegen.DOTEW_MAX_K and K % 16 == 0 and M <= MAX_ROWS
        and T in ("float32", "float64") and all(body["in_dtypes"][q] == T for q in dpos)
    )
    for q in dpos:
        A, B, Bp = dots[q]
        fast = fast and A.shape[1] == K and str(A.dtype) == T and str(B.dtype) == T
        fast = fast and A.strides[1] == 1 and (A.strides[0] * A.itemsize) % 16 == 0 and A.ptr % 16 == 0
        fast = fast and Bp.size == _ceil16(K) * _ceil16(N) and Bp.is_contiguous() and Bp.ptr % 16 == 0
    if not fast:
        # shapes/layouts the generated kernel does not cover: MFMA GEMM, then the Elemwise kernel
        ins2 = [gemm_device(env, 1.0, _prep2d(dots[k][0]), _prep2d(dots[k][1])) if k in dots else a for k, a in enumerate(ins)]
        outs, _, _ = launch_elemwise(body, ins2, shape, out_dtypes, None, env, (), placed)
        return outs
    outs = [
        placed[k] if placed and placed[k] is not None else DeviceArray.empty(shape, out_dtypes[k])
        for k in range(len(out_dtypes))
    ]
    byvalue = {k for k, a in enumerate(ins) if isinstance(a, HostValue)}
    # register buffers of 8 k-groups (two in flight = every load of a K=1024 product): 16.0 us per
    # GRU step; 16 (both products of the update gate in flight, ~300 VGPRs) measured 17.4
    # (profiles/r2f_c5_chunk.txt)
    chunk = int(os.environ.get("PTHIP_DOTEW_CHUNK", 0)) or codegen.DOTEW_CHUNK
    name = f"dotew_{_body_key(body)}_k{K}_d{'_'.join(map(str, dpos))}_u{chunk}" + ("_c" + "_".join(map(str, sorted(byvalue))) if byvalue else "")
    # products with the very same left operand stream it once
    share = {}
    for t2, q2 in enumerate(dpos):
        for q1 in dpos[:t2]:
            a1, a2 = dots[q1][0], dots[q2][0]
            if q1 not in share and a1.ptr == a2.ptr and a1.shape == a2.shape and a1.strides == a2.strides and q1 not in share.values():
                share[q2] = q1
                break
    if share:
        name += "_h" + "_".join(f"{a}.{b}" for a, b in sorted(share.items())).replace(".", "x")
    # Left operands that an earlier step kernel ALSO stored in the MFMA operand order (Scan steps:
    # `r*h` of this step, `h` of the previous one — dispatch/scan.py hands the context over) are read
    # from that image; outputs a later step kernel multiplies from the left are stored in it too.
    ctx = getattr(env, "scan_ctx", None)
    use_pack = ctx is not None and os.environ.get("PTHIP_DOTEW_PACKA", "1") != "0"
    packed_a = {}
    if use_pack:
        for q in dpos:
            src = ctx["tap_src"].get(node.inputs[q], (node.inputs[q], 0))
            ent = ctx["packed"].get((src[0], ctx["t"] - src[1]))
            if ent is not None and ent.size == _ceil16(M) * K and str(ent.dtype) == T:
                packed_a[q] = ent
    pack_outs = []
    if use_pack and N % 16 == 0:
        pack_outs = [k for k, o in enumerate(node.outputs[: len(out_dtypes)]) if o in ctx["pack_vars"] and out_dtypes[k] == T]
    if packed_a:
        name += "_pa" + "_".join(str(q) for q in sorted(packed_a))
    if pack_outs:
        name += "_po" + "_".join(map(str, pack_outs))
    lds_a = os.environ.get("PTHIP_DOTEW_LDSA", "0") == "1" and T == "float32" and K % 128 == 0 and chunk % 2 == 0 and not packed_a
    if lds_a:
        name += "_la"
    var = os.environ.get("PTHIP_DOTEW_VAR", "acc4")  # accumulator chains; other values: tools/dotew_variants.py timing-only decompositions
    if var:
        name += "_" + var.replace(",", "_")
    src = codegen.dot_epilogue_source(name, body, dpos, K, byvalue, chunk, share, lds_a, var, set(packed_a), pack_outs)
    fn = kernel_cache.get_function(src, name)
    args = [M, N]
    for k, a in enumerate(ins):
        if k in dots:
            A, _, Bp = dots[k]
            if k in packed_a:
                args += [packed_a[k].ptr, 0, Bp.ptr]
            else:
                args += [A.ptr, A.strides[0], Bp.ptr]
        elif k in byvalue:
            args.append(_scalar_bits(a, body["in_dtypes"][k]))
        else:
            args += [a.ptr, 0 if a.shape[0] == 1 and M != 1 else a.strides[0], 0 if a.shape[1] == 1 and N != 1 else a.strides[1]]
    for o in outs:
        args += [o.ptr, o.strides[0]]
    for k in pack_outs:
        pk = DeviceArray.empty((_ceil16(M) * N,), out_dtypes[k])
        args.append(pk.ptr)
        ctx["packed"][(node.outputs[k], ctx["t"])] = pk
    buf = struct.pack(f"<{len(args)}q", *args)
    env.timed(name, lambda: ffi.check(env.lib.pthip_launch(fn, (N + 15) // 16, (M + 15) // 16, 1, codegen.BLOCK, 1, 1, 0, buf, len(buf))))
    return outs
