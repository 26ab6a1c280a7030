"""Handlers of the IR-level fusions that have no single reference Op
(``pytensor_amd/fusion.py``): ``GemvChain`` / ``GemvFinish``.

``GemvChain`` reads the matrix once for the forward ``A@x`` and the backward ``A.T@w``
(reference: two ``Gemv.perform`` calls, pytensor/tensor/blas/gemv.py:64-108, with an
``Elemwise.perform`` between them), optionally gathering ``table[idx]`` inputs
(``AdvancedSubtensor``, subtensor.py:1932) and scatter-adding one output
(``AdvancedIncSubtensor``, subtensor.py:2275) in the same pass.  When the operands do
not meet the fast kernel's layout requirements the handler falls back to the unfused
launches — same results, two passes over the matrix.
"""

from __future__ import annotations

import os
import struct

import numpy as np

from pytensor_amd import codegen, ffi, kernel_cache
from pytensor_amd.device import DeviceArray, copy_into
from pytensor_amd.dispatch import handler
from pytensor_amd.dispatch.blas import _scalar, gemv_device
from pytensor_amd.dispatch.elemwise import BLOCK, MAX_GRID, _body_key, alloc_partials, finish_partials, launch_elemwise
from pytensor_amd.executor import HostValue

CHAIN_RG = int(os.environ.get("PTHIP_CHAIN_RG", 0))  # 0 = auto
# Workgroups of the one-pass kernel (230 VGPRs -> 2 workgroups per CU resident).  Measured
# on C4 (evals/s): 448: 3252, 480: 3220, 512: 3209, 1024: 3274, 2048: 3331, 4096: 3282 —
# a persistent grid that leaves CUs free for the overlapped latency chain does not pay.
CHAIN_GRID = int(os.environ.get("PTHIP_CHAIN_GRID", 2048))
MAX_SCATTER_BINS = 256  # 64 bins per accumulator register of a lane, up to four (codegen.gemv_chain_source)
MAX_CHAIN_K = 4096  # columns the one-pass kernel holds in registers (32 chunks of 128: one row per group)


def _unpack(node, inputs, env):
    """Split the flat input list of a GemvChain node into its parts."""
    p = node.params
    body = p["scalar"]
    gather = set(p.get("gather") or [])
    it = iter(inputs[5:])
    e_vals = []  # per elementwise input position != r_pos: DeviceArray | (table, idx)
    for pos in range(len(body["in_dtypes"])):
        if pos == p["r_pos"]:
            e_vals.append(None)
        elif pos in gather:
            e_vals.append((env.to_device(next(it)), env.to_device(next(it))))
        else:
            e_vals.append(env.to_device(next(it)))
    sidx = base = None
    if p.get("scatter_out") is not None:
        sidx = env.to_device(next(it))
        base = next(it)
        if p.get("scatter_len_input"):
            base = _Bins(int(np.asarray(env.to_host(base)).reshape(-1)[0]))  # zeros(n): only n exists
        else:
            base = env.to_device(base)
    return e_vals, sidx, base


class _Bins:
    """Stand-in for a never-materialised ``zeros(n)`` scatter base: just its shape."""

    ndim = 1

    def __init__(self, n):
        self.shape = (n,)


def _fast_ok(A, x1, y1, e_vals, sidx, base, N, K):
    if os.environ.get("PTHIP_GCHAIN_FAST", "1") == "0":  # (measurements: the unfused launches beside the one-pass kernel)
        return False
    if str(A.dtype) not in ("float64", "float32") or A.ndim != 2 or A.strides[1] != 1 or A.ptr % A.itemsize:
        return False
    if K > MAX_CHAIN_K or K == 0 or N == 0:
        return False
    if not x1.is_contiguous() or x1.dtype != A.dtype or (y1 is not None and y1.dtype != A.dtype):
        return False
    if str(A.dtype) == "float32" and (K % 4 or A.strides[0] % 4 or A.ptr % 16):
        # float32 rows that cannot be read in 16-byte packs: 4- and 8-byte loads per lane make the one pass slower than
        # the two unfused ones (K = 127: 1.04 ms against 0.68, profiles/r6b_gchain_f32_sweep.txt)
        return False
    if y1 is not None and (not y1.is_contiguous() or y1.shape != (N,)):
        return False
    for v in e_vals:
        if v is None:
            continue
        if isinstance(v, tuple):
            table, idx = v
            if not table.is_contiguous() or table.ndim != 1 or idx.shape != (N,) or not idx.is_contiguous():
                return False
            if table.shape[0] == 0:
                return False
        elif v.size != 1 and (v.shape != (N,) or not v.is_contiguous()):
            return False
    if sidx is not None:
        # (more than MAX_SCATTER_BINS bins: still ONE pass over the matrix — the kernel stores the scattered output and
        #  the deterministic scatter kernel of csrc/index.hip adds it up afterwards, gemv_chain below)
        if sidx.shape != (N,) or not sidx.is_contiguous() or base.ndim != 1 or base.shape[0] <= 0:
            return False
    return True


@handler("GemvChain")
def gemv_chain(node, inputs, env):
    p = node.params
    body, spec, r_pos, w_out, store_r = p["scalar"], p["reduce"], p["r_pos"], p["w_out"], p["store_r"]
    nout = len(body["out_dtypes"])
    out_store = p.get("out_store") or [s is None for s in spec]
    scatter_out = p.get("scatter_out")
    y1, alpha1, A, x1, beta1 = inputs[:5]
    alpha1, beta1 = _scalar(env, alpha1), _scalar(env, beta1)
    A, x1 = env.to_device(A), env.to_device(x1)
    y1d = None if beta1 == 0.0 else env.to_device(y1)
    e_vals, sidx, base = _unpack(node, inputs, env)
    N, K = A.shape
    if x1.shape != (K,):
        raise ValueError(f"Shape mismatch: A.shape[1] != x.shape[0] ({A.shape}, {x1.shape})")
    if not _fast_ok(A, x1, y1d, e_vals, sidx, base, N, K):
        return _fallback(node, env, alpha1, A, x1, beta1, y1d, e_vals, sidx, base)
    # the scatter-add rides in the kernel's registers up to MAX_SCATTER_BINS bins; beyond that the kernel stores the
    # scattered output (N x 8 bytes next to the N x K x 8 of the matrix) and a second, small launch bins it
    late_scatter = scatter_out is not None and base.shape[0] > MAX_SCATTER_BINS
    if late_scatter:
        out_store = list(out_store)
        scatter_pos, scatter_out = scatter_out, None
        late_forced = not out_store[scatter_pos] and spec[scatter_pos] is None
        out_store[scatter_pos] = True
        if spec[scatter_pos] is not None:
            return _fallback(node, env, alpha1, A, x1, beta1, y1d, e_vals, sidx, base)  # (a reduced AND scattered output: not a shape the passes produce)
    C = (K + 127) // 128
    f32x4 = (str(A.dtype) == "float32" and K % 4 == 0 and A.strides[0] % 4 == 0 and A.ptr % 16 == 0 and os.environ.get("PTHIP_GCHAIN_F32X4", "1") != "0")
    if f32x4:
        C = 2 * ((K + 255) // 256)  # 16-byte loads of four float columns: chunk PAIRS of 256 columns
    RG = CHAIN_RG or 32
    while RG * C > 32 and RG > 1:  # rows per group: the row registers hold RG*C <= 32 packs (K <= 1024: >= 4 rows; 2048: 2; 4096: 1)
        RG //= 2
    # 16-byte packs need every row to start on a 16-byte boundary; otherwise two 8-byte loads per chunk (odd K, odd lda)
    # (K <= 64 leaves half the lanes of the one chunk without a pack; a column per lane instead measured slower:
    #  K = 64: 238 us with packs, 262 us with 8-byte loads on all lanes — profiles/r5_gchain_sweep.txt)
    pk_bytes = 2 * A.itemsize  # (a pack = two elements of the matrix: 16 bytes of fp64, 8 of fp32)
    pack = 2 if (K % 2 == 0 and A.strides[0] % 2 == 0 and A.ptr % pk_bytes == 0 and x1.ptr % pk_bytes == 0) else 1
    atype = str(A.dtype)
    if f32x4:
        pack = 4
    e_modes = []
    for pos, v in enumerate(e_vals):
        if v is None:
            e_modes.append("R")
        elif isinstance(v, tuple):
            e_modes.append("G")
        else:
            e_modes.append("S" if v.size == 1 else "V")
    rs = [None if r is None else (r["op"], r["acc_dtype"]) for r in spec]
    rkey = "".join("-" if r is None else r["op"][0] + r["acc_dtype"][0] + r["acc_dtype"][-1] for r in spec)
    skey = "".join("1" if s else "0" for s in out_store)
    name = (
        f"gchain_{_body_key(body)}_{''.join(e_modes)}_{rkey}_w{w_out}_c{C}_g{RG}_{int(store_r)}{int(y1d is not None)}"
        f"_s{skey}_{'n' if scatter_out is None else scatter_out}" + ("" if pack == 2 else f"_p{pack}") + ("" if atype == "float64" else "_f32")
    ).replace("-", "x")
    sgroups = 2
    if scatter_out is not None:
        sgroups = max(1, (base.shape[0] + 63) // 64)
        name += f"_b{sgroups}"
    src = codegen.gemv_chain_source(name, body, e_modes, rs, w_out, C, RG, store_r, y1d is not None, out_store, scatter_out, sgroups, pack, atype)
    fn = kernel_cache.get_function(src, name)
    ngroups = (N + RG - 1) // RG
    grid = max(1, min((ngroups + 3) // 4, CHAIN_GRID))
    args = [("q", N), ("q", K), ("q", A.ptr), ("q", A.strides[0]), ("q", x1.ptr), ("q", y1d.ptr if y1d is not None else 0), ("d", alpha1), ("d", beta1)]
    for v in e_vals:
        if v is None:
            continue
        if isinstance(v, tuple):
            args += [("q", v[0].ptr), ("q", v[1].ptr), ("q", v[0].shape[0])]
        else:
            args.append(("q", v.ptr))
    r_out = None
    if store_r:
        r_out = DeviceArray.empty((N,), atype)
        args.append(("q", r_out.ptr))
    parts = alloc_partials(spec, grid)
    stored = [None] * nout
    for k in range(nout):
        if spec[k] is not None:
            args.append(("q", parts[k].ptr))
        elif out_store[k]:
            stored[k] = DeviceArray.empty((N,), body["out_dtypes"][k])
            args.append(("q", stored[k].ptr))
    partT = DeviceArray.empty((grid, K), "float64")
    args.append(("q", partT.ptr))
    partS = None
    if scatter_out is not None:
        bins = base.shape[0]
        partS = DeviceArray.empty((grid, bins), "float64")
        args += [("q", sidx.ptr), ("q", bins), ("q", partS.ptr)]
    args.append(("q", _status_ptr(env)))
    buf = struct.pack("<" + "".join(a[0] for a in args), *[a[1] for a in args])
    kt = env.kernel_timer
    tok = kt.begin() if kt is not None else None
    ffi.check(env.lib.pthip_launch(fn, grid, 1, 1, BLOCK, 1, 1, 0, buf, len(buf)))
    if kt is not None:
        kt.end(name, tok)
    finals = finish_partials(env, spec, parts, grid, p.get("defer_reduce") or ())
    res = [r_out] if store_r else []
    for k in range(nout):
        res.append(finals[k] if spec[k] is not None else stored[k])
    res.append(partT)
    if scatter_out is not None:
        res.append(partS)
    elif late_scatter:
        res.append(_scatter_rows(env, stored[scatter_pos], sidx, base.shape[0]))
        if late_forced:
            res[(1 if store_r else 0) + scatter_pos] = None  # (nothing outside the node reads it: stored only for the scatter)
    return res


_STATUS = [None]


def _status_ptr(env) -> int:
    """device address of the runtime's error flag (read + cleared by pthip_check_status)"""
    if _STATUS[0] is None:
        _STATUS[0] = env.lib.pthip_status_ptr()
    return _STATUS[0]


def _fallback(node, env, alpha1, A, x1, beta1, y1d, e_vals, sidx, base):
    """Unfused launches, two passes over A (exactly the original graph)."""
    p = node.params
    body, spec, r_pos, w_out, store_r = p["scalar"], p["reduce"], p["r_pos"], p["w_out"], p["store_r"]
    scatter_out = p.get("scatter_out")
    N, K = A.shape
    r = gemv_device(env, alpha1, A, x1, beta1, y1d)
    ins = []
    for v in e_vals:
        if v is None:
            ins.append(r)
        elif isinstance(v, tuple):
            table, idx = v
            g = DeviceArray.empty((idx.shape[0],), table.dtype)
            if g.size:
                tc = table.contiguous()
                ffi.check(env.lib.pthip_take_rows(table.itemsize, idx.shape[0], 1, tc.ptr, tc.shape[0], 1, idx.contiguous().ptr, g.ptr))
            ins.append(g)
        else:
            ins.append(v)
    outs, parts, grid = launch_elemwise(body, ins, (N,), body["out_dtypes"], spec, env)
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
    if scatter_out is not None:
        res.append(_scatter_rows(env, outs[scatter_out], sidx, base.shape[0]))
    return res


def _scatter_rows(env, values, sidx, bins):
    """``zeros(bins)[sidx] += values`` (AdvancedIncSubtensor, subtensor.py:2275) as a (1, bins) partial slab: the
    deterministic scatter kernel of csrc/index.hip."""
    acc = DeviceArray.empty((bins,), values.dtype)
    ffi.check(env.lib.pthip_memset(acc.ptr, 0, acc.nbytes))
    n_idx = sidx.shape[0]
    if n_idx and bins:
        lib = env.lib
        ws_bytes = lib.pthip_scatter_rows_workspace(n_idx, bins, 1)
        ws = DeviceArray.empty((ws_bytes,), "uint8") if ws_bytes else None
        sv = values.contiguous()
        ffi.check(
            lib.pthip_scatter_rows(ffi.np_dtype_code(values.dtype), 1, n_idx, 1, acc.ptr, bins, sidx.contiguous().ptr, sv.ptr, 1,
                                   ws.ptr if ws is not None else None, ws_bytes)
        )
    return acc.view((1, bins), (bins, 1))


@handler("GemvFinish")
def gemv_finish(node, inputs, env):
    part, y2, alpha2, beta2 = inputs
    alpha2, beta2 = _scalar(env, alpha2), _scalar(env, beta2)
    nparts, M = part.shape
    out = DeviceArray.empty((M,), part.dtype)
    odt = str(env.graph.vars[node.outputs[0]].dtype)
    y2d = None if beta2 == 0.0 else env.to_device(y2)
    if y2d is not None and y2d.shape != (M,):
        raise ValueError(f"Shape mismatch: y.shape[0] != A.shape[0] ({y2d.shape}, {M})")
    if y2d is not None and y2d.dtype != part.dtype:
        from pytensor_amd.dispatch.elemwise import _cast

        y2d = _cast(env, y2d.contiguous(), str(part.dtype))  # (float32 graph, float64 slabs: the epilogue runs in the slabs' precision)
    ffi.check(
        env.lib.pthip_gemv_finish(
            ffi.np_dtype_code(part.dtype), M, nparts, part.ptr, alpha2, beta2,
            y2d.ptr if y2d is not None else None, (y2d.strides[0] if M > 1 else 1) if y2d is not None else 0, out.ptr,
        )
    )
    if odt != str(part.dtype) and odt in ("float32", "float64"):
        from pytensor_amd.dispatch.elemwise import _cast

        out = _cast(env, out, odt)  # float64 partial slabs of a float32 graph: rounded once, at the end
    return [out]
