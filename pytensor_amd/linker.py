"""``HipLinker`` — the drop-in boundary: ``pytensor.function(..., mode="hip")``.

Mirrors the reference's JIT linkers (pytensor/link/basic.py:582-745 ``JITLinker``;
closest sibling pytensor/link/pytorch/linker.py:5-104).  Importing this module
requires PyTensor to be importable; everything below the boundary
(``pytensor_amd.executor``, the C-ABI) does not.

What the linker does, per the reference contract:

* ``fgraph_convert``  lowers the rewritten ``FunctionGraph`` to the portable IR
  (``pytensor_amd.lower.lower_fgraph``) and wraps it in a ``HipExecutable``;
* ``create_thunk_inputs`` hands over the storage cells of *all* fgraph inputs
  (explicit + shared), exactly like pytorch/linker.py:97-104;
* ``jit_compile`` returns the callable whose positional arguments are the host
  ``ndarray``s and whose result is a tuple of host ``ndarray``s
  (link/basic.py:670-684 consumes it).

Registration follows pytensor/compile/mode.py:59-63 (``register_linker``) and
mode.py:624-631 (``register_mode``); the inner-graph rewrites are registered
per pytensor/compile/rewriting.py:128-168 and
pytensor/scan/rewriting/inner_graph.py:28-91 (functional variants, as the JAX /
PyTorch linkers use).
"""

from __future__ import annotations

from pytensor.compile.mode import Mode, register_linker, register_mode, predefined_linkers, predefined_modes
from pytensor.graph.rewriting.db import RewriteDatabaseQuery
from pytensor.link.basic import JITLinker


# The trace-reducing Scan rewrite (scan/rewriting/trace.py:888 `scan_reduce_trace_prealloc`, the C backend's variant: a state
# nobody reads back keeps taps + 1 rows instead of n_steps + taps).  Rounds 2-5 excluded it (as the Numba linker does) without
# taking the JIT backends' `no_prealloc` variant in its place, so NO trace was ever shortened: a 1000-step recurrence read
# at [-1] kept 1002 rows (reference tests/scan/rewriting/test_trace.py).  The no-prealloc form (taps rows) lets a step's
# output land on the row of its oldest tap — the step kernels here read and write rows in separate launches, and three
# reference tests fail with it — so it is the prealloc form that is on.  PTHIP_SCAN_SAVE_MEM=0: as before.
_SCAN_SAVE_MEM = __import__("os").environ.get("PTHIP_SCAN_SAVE_MEM", "1") != "0"


class HipLinker(JITLinker):
    """A `Linker` that runs a ``FunctionGraph`` on MI355X through hand-written HIP kernels."""

    required_rewrites = ("minimum_compile",)
    # Appendix C of SURVEY.md: keep ``fusion`` and ``BlasOpt`` (they produce the
    # Elemwise{Composite} / Gemm / Gemv / Dot22 nodes we have kernels for);
    # drop C-only and in-place rewrites (kernels are functional).
    incompatible_rewrites = (
        "cxx_only",
        "inplace",
        *(() if _SCAN_SAVE_MEM else ("scan_reduce_trace_prealloc",)),
        # Softmax / LogSoftmax stay whole and run as one kernel (csrc/softmax.hip), like the
        # JAX / PyTorch / MLX linkers that dispatch their own softmax (rewriting/ofg.py:46-62).
        # XLogY, XLog1PY, LogSumExp, LogAddExp are still inlined at specialize (ofg.py:16-17).
        "inline_symbolic_for_fusion",
    )

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.last_ir = None  # the lowered IR of the most recent fgraph (for export)

    def fgraph_convert(self, fgraph, input_storage=None, storage_map=None, **kwargs):
        from pytensor_amd.lower import lower_fgraph

        # No silent CPU fallback: an Op without a device lowering is an error unless the user
        # opts in to the D2H -> Op.perform -> H2D detour (config.hip__allow_host_perform).
        from pytensor.configdefaults import config

        graph = lower_fgraph(fgraph, allow_host_fallback=bool(config.hip__allow_host_perform))
        self.last_ir = graph
        # shared variables = data: uploaded once and kept in HBM (re-uploaded when the storage cell
        # holds a different array or the array's content fingerprint changed: executor._refresh_resident)
        from pytensor.compile.sharedvalue import SharedVariable

        self._resident = [k for k, v in enumerate(fgraph.inputs) if isinstance(v, SharedVariable)]
        # `updates=`: output i is stored into the storage cell of input j after every call
        # (compile/executor.py:712-716); the executor keeps that value in HBM (fg.py:188)
        self._update_map = dict(getattr(fgraph, "update_mapping", None) or {})
        return graph

    def jit_compile(self, graph):
        from pytensor.configdefaults import config

        from pytensor_amd.executor import HipExecutable

        # repeated calls with one input signature replay a captured hipGraph (the CVM analogue)
        return HipExecutable(graph, resident=getattr(self, "_resident", ()), auto_freeze=bool(config.hip__auto_freeze),
                             update_map=getattr(self, "_update_map", None),
                             device=None if config.hip__device < 0 else int(config.hip__device))

    def create_thunk_inputs(self, storage_map):
        # cf. pytensor/link/pytorch/linker.py:97-104: every fgraph input,
        # shared variables included, is passed on each call; the executor keeps
        # device-resident copies of arrays it has seen (identity-keyed cache).
        return [storage_map[n] for n in self.fgraph.inputs]


def _add_config_flags():
    """``hip__*`` flags, declared the way every backend flag of the reference is
    (pytensor/configdefaults.py:183 ``config.add``; sources: ``PYTENSOR_FLAGS``, ``~/.pytensorrc``,
    ``pytensor.config.change_flags``).  The ``PTHIP_*`` environment variables of earlier rounds
    remain as the defaults, so existing set-ups keep working."""
    import os

    from pytensor.configdefaults import config
    from pytensor.configparser import BoolParam, EnumStr, IntParam, StrParam

    if hasattr(config, "hip__resident"):
        return
    env = os.environ.get
    modes = ["guard", "strict", "sampled", "trust"]
    default = env("PTHIP_RESIDENT", "guard")
    config.add(
        "hip__resident",
        "How the hip linker keeps the HBM copy of a shared variable coherent with its host array "
        "(which the reference reads on every call): 'guard' write-protects the array's pages and "
        "re-uploads after the first store (sound for every CPU store, free when nothing changes; arrays up to 64 KiB, "
        "read-only arrays and memory pinned / registered with the HIP runtime are hashed instead; while an array is "
        "guarded, a kernel-side write INTO it — file.readinto, socket.recv_into, np.fromfile into a view, an MPI "
        "receive — fails with EFAULT where the reference would accept it: use set_value, or 'strict'); 'strict' hashes the "
        "whole array on every call; 'sampled' checks 256 elements (may miss sparse in-place edits); "
        "'trust' checks identity only.",
        EnumStr(default, [m for m in modes if m != default], mutable=True),
        in_c_key=False,
    )
    config.add(
        "hip__device",
        "HIP device ordinal the process binds to (-1: device 0, or the rank's device under pytensor_amd.replicas).",
        IntParam(int(env("PTHIP_DEVICE", "-1")), mutable=True),
        in_c_key=False,
    )
    config.add(
        "hip__cache_dir",
        "Directory of the generated-kernel cache (gfx950 code objects keyed by source hash; the analogue of compiledir). "
        "Empty: pytensor_amd/_kcache next to the package.",
        StrParam(env("PTHIP_KCACHE", ""), mutable=True),
        in_c_key=False,
    )
    config.add(
        "hip__auto_freeze",
        "Capture the launch sequence of a repeated call signature into a hipGraph plan and replay it (the CVM analogue).",
        BoolParam(env("PTHIP_AUTO_FREEZE", "1") != "0", mutable=True),
        in_c_key=False,
    )
    config.add(
        "hip__allow_host_perform",
        "Run Ops without a device lowering through Op.perform on the host (D2H, perform, H2D) instead of raising at compile time.",
        BoolParam(env("PTHIP_ALLOW_HOST_PERFORM") == "1", mutable=True),
        in_c_key=False,
    )


def _register():
    _add_config_flags()
    from pytensor.compile.rewriting import rewrite_ofg_inner_graph, _ofg_inner_optimizer
    from pytensor.scan.rewriting.inner_graph import rewrite_scan_inner_graph

    if "hip" not in predefined_linkers:
        register_linker("hip", HipLinker())
    if "HIP" not in predefined_modes:
        register_mode(
            "HIP",
            Mode(HipLinker(), RewriteDatabaseQuery(include=["fast_run"])),
        )

    if HipLinker not in rewrite_ofg_inner_graph.registry:

        @rewrite_ofg_inner_graph.register(HipLinker)
        def _hip_rewrite_ofg_inner_graph(linker, op, node, inner, *, mode):
            _ofg_inner_optimizer(mode, op).rewrite(inner)

    if HipLinker not in rewrite_scan_inner_graph.registry:
        from pytensor.scan.rewriting.inner_graph import scan_inner_optimizer

        @rewrite_scan_inner_graph.register(HipLinker)
        def _hip_rewrite_scan_inner_graph(linker, op, node, inner, *, mode):
            scan_inner_optimizer(op, mode).rewrite(inner)


_register()
