"""``HipExecutable`` — runs a lowered graph (``pytensor_amd.ir.Graph``) on the device.

This is the callable ``HipLinker.jit_compile`` returns: positional host
``ndarray``s in (all ``fgraph.inputs``, shared variables included), a tuple of
freshly-owned host ``ndarray``s out — the contract of the thunk built in
pytensor/link/basic.py:670-684 (cf. pytorch/linker.py:81-88).

Execution model (MI355X-first, no tracing compiler):

* every node maps to zero or more launches on ONE in-order HIP stream through the
  C-ABI (``include/pthip.h``); views launch nothing;
* shape arithmetic (``Shape_i``, ``MakeVector``, integer ``Elemwise`` on shapes,
  ``ScalarFromTensor`` of those) stays on the host, as control plane — tensor
  data never does: there is NO CPU fallback for data ops, the module raises if
  ``libpthip.so`` or the GPU is missing;
* inputs marked resident (shared variables) are uploaded once and cached by array
  identity;
* the only host↔device crossings are H2D of non-resident inputs at entry and D2H
  of outputs at exit, followed by the single stream synchronisation of the call;
* ``freeze()`` captures the whole launch sequence of one call signature into a
  hipGraph (arena-backed buffers, pinned staging) so that subsequent calls cost
  one native call instead of one Python dispatch per node — the analogue of the
  reference's CVM (pytensor/link/c/c_code/lazylinker_c.c:749).
"""

from __future__ import annotations

import ctypes as C
import os
import weakref

import numpy as np

from pytensor_amd import coherence, ffi
from pytensor_amd.device import DeviceArray
from pytensor_amd.ir import Graph

HOST_MAX = 64  # host-resident values are tiny integer/bool arrays (shape math)


class HostValue:
    """A small host-resident value (shape arithmetic / scalar indices)."""

    __slots__ = ("a", "dev")

    def __init__(self, a):
        self.a = np.asarray(a)
        self.dev = None  # cached device copy (constants are uploaded once)

    @property
    def shape(self):
        return self.a.shape

    @property
    def dtype(self):
        return self.a.dtype

    @property
    def ndim(self):
        return self.a.ndim

    @property
    def size(self):
        return self.a.size

    def __repr__(self):
        return f"HostValue({self.a!r})"


# ---------------------------------------------------------------------------------------
# Resident (shared-variable) inputs: coherence between the host array and its HBM copy
# ---------------------------------------------------------------------------------------
# The reference backends read a shared variable's storage cell on every call; what keeps the
# resident copy honest (write-protected pages by default, content hashes for small arrays) is in
# pytensor_amd/coherence.py.  `HipExecutable.invalidate_resident()` forces a re-upload.


class ResidentEntry:
    """One shared-variable input kept in HBM: a *stable* device buffer (frozen plans capture its
    address; a new host value of the same shape is copied into it in place), the host array it
    mirrors (strong reference: keeps `id` unique, and the memory mapped while its pages are
    write-protected) and the coherence token watching that array (coherence.watch)."""

    __slots__ = ("key", "dev", "host", "fp")

    def __init__(self, key, dev, host, fp):
        self.key, self.dev, self.host, self.fp = key, dev, host, fp

    def rewatch(self, key, host, a):
        coherence.release(self.fp)
        self.key, self.host, self.fp = key, host, coherence.watch(a)

    def __del__(self):
        coherence.release(self.fp)


def _resident_key(value, a):
    return (id(value), a.ctypes.data, a.shape, a.strides, a.dtype.str)


class DeferredReduce:
    """A 0-d value whose second-stage reduction has not run yet: the per-workgroup partials of a
    fused Elemwise+reduce kernel (``parts``: ``[grid]`` in the accumulator dtype).  Produced only
    for outputs whose every consumer is a ``Tail`` node (tailfuse.py), which folds the sum into
    its one kernel; anything else that touches the value forces it (``Env.to_device``)."""

    __slots__ = ("parts", "grid", "spec", "_forced")
    ndim, shape, size = 0, (), 1

    def __init__(self, parts, grid, spec):
        self.parts, self.grid, self.spec, self._forced = parts, int(grid), spec, None

    @property
    def dtype(self):
        return np.dtype(self.spec["dtype"])

    def force(self, env) -> DeviceArray:
        if self._forced is None:
            from pytensor_amd.dispatch.elemwise import device_reduce

            r = self.spec
            self._forced = device_reduce(env, r["op"], self.parts, 1, self.grid, 1, 0, 1, 0, r["acc_dtype"], r["dtype"], ())
        return self._forced


class KernelTimer:
    """HIP-event brackets around individual generated-kernel launches (profiling only)."""

    # what a two-event bracket of ONE launch reports beyond the launch itself (event packets, the
    # command processor's gaps either side): measured once per process with a 256-byte fill kernel as
    # (mean bracket of one launch) - (per-launch time of the same launches back to back between one
    # pair of events), and subtracted from every bracket.  ~3-4 us on MI355X; without it the bracketed
    # time of a 22 us kernel exceeded the graph replay that contains it.
    overhead_ms = None

    @classmethod
    def calibrate(cls, reps=40):
        if cls.overhead_ms is not None:
            return cls.overhead_ms
        lib = ffi.lib()
        buf = C.c_void_p()
        ffi.check(lib.pthip_alloc(256, C.byref(buf)))
        e0, e1 = C.c_void_p(), C.c_void_p()
        ffi.check(lib.pthip_event_create(C.byref(e0)))
        ffi.check(lib.pthip_event_create(C.byref(e1)))
        ms = C.c_float()
        try:
            for _ in range(5):
                ffi.check(lib.pthip_memset(buf, 0, 256))
            ffi.check(lib.pthip_synchronize())
            br = []
            for _ in range(reps):
                ffi.check(lib.pthip_event_record(e0))
                ffi.check(lib.pthip_memset(buf, 0, 256))
                ffi.check(lib.pthip_event_record(e1))
                ffi.check(lib.pthip_event_synchronize(e1))
                ffi.check(lib.pthip_event_elapsed_ms(e0, e1, C.byref(ms)))
                br.append(ms.value)
            ffi.check(lib.pthip_event_record(e0))
            for _ in range(reps):
                ffi.check(lib.pthip_memset(buf, 0, 256))
            ffi.check(lib.pthip_event_record(e1))
            ffi.check(lib.pthip_event_synchronize(e1))
            ffi.check(lib.pthip_event_elapsed_ms(e0, e1, C.byref(ms)))
            br.sort()
            cls.overhead_ms = max(0.0, br[len(br) // 2] - ms.value / reps)
        finally:
            lib.pthip_event_destroy(e0)
            lib.pthip_event_destroy(e1)
            lib.pthip_free(buf)
        return cls.overhead_ms

    def __init__(self):
        self.lib = ffi.lib()
        self.pending = []
        self.totals = {}
        self.overhead = self.calibrate()

    def begin(self):
        e0, e1 = C.c_void_p(), C.c_void_p()
        ffi.check(self.lib.pthip_event_create(C.byref(e0)))
        ffi.check(self.lib.pthip_event_create(C.byref(e1)))
        ffi.check(self.lib.pthip_event_record(e0))
        return e0, e1

    def end(self, name, tok):
        ffi.check(self.lib.pthip_event_record(tok[1]))
        self.pending.append((name, tok))

    def resolve(self):
        for name, (e0, e1) in self.pending:
            ms = C.c_float()
            ffi.check(self.lib.pthip_event_synchronize(e1))
            ffi.check(self.lib.pthip_event_elapsed_ms(e0, e1, C.byref(ms)))
            t = self.totals.setdefault(name, [0.0, 0])
            t[0] += max(ms.value - self.overhead, 0.0)
            t[1] += 1
            self.lib.pthip_event_destroy(e0)
            self.lib.pthip_event_destroy(e1)
        self.pending = []

    def mean_ms(self):
        return {k: v[0] / v[1] for k, v in self.totals.items()}


def _has_op(graph, name: str) -> bool:
    """``name`` anywhere in the graph, inner graphs of Scan / Tail nodes included."""
    for n in graph.nodes:
        if n.op == name:
            return True
        inner = n.params.get("inner") if isinstance(n.params, dict) else None
        if inner is not None and _has_op(inner, name):
            return True
        if n.op == "Tail" and any(m.op == name for m in n.params.get("nodes", ())):
            return True
    return False


def branch_guards(g):
    """Lazy ``IfElse`` (pytensor/ifelse.py:42; the VM evaluates the condition, then only the branch
    taken): ``guard[k] = (IfElse node index, branch)`` for every node whose results reach the graph
    outputs only through that branch's inputs — the innermost one when conditionals nest — and the
    node lists per (IfElse, branch) in execution order."""
    nodes = g.nodes
    guard = [None] * len(nodes)
    members = {}
    if not any(n.op == "IfElse" for n in nodes):
        return guard, members
    uses = {}
    for k, n in enumerate(nodes):
        for pos, i in enumerate(n.inputs):
            uses.setdefault(i, []).append((k, pos))
    outs = set(g.outputs)
    for k, n in enumerate(nodes):
        if n.op != "IfElse":
            continue
        n_out = len(n.outputs)
        for b in (0, 1):
            lo, hi = 1 + b * n_out, 1 + (b + 1) * n_out
            inside = set()
            for j in range(k - 1, -1, -1):
                m = nodes[j]
                if any(o in outs for o in m.outputs):
                    continue
                us = [u for o in m.outputs for u in uses.get(o, ())]
                if us and all((uk == k and lo <= up < hi) or uk in inside for uk, up in us):
                    inside.add(j)
            mine = [j for j in sorted(inside) if guard[j] is None]
            for j in mine:
                guard[j] = (k, b)
            members[(k, b)] = mine
    return guard, members


class DeviceWaitExpired(RuntimeError):
    """status bit 4: a bounded dependency wait of a cooperative linear-algebra kernel expired"""


def raise_device_status(word: int):
    """The device error word (kernels cannot raise): bit 0 = an index was out of range
    (IndexError, like the reference's take/inc_subtensor), bit 1 = np.linalg.inv met an exactly
    singular matrix (LinAlgError), bit 2 / bit 3 = the Jacobi SVD / eigensolver did not converge
    (LinAlgError, the messages of np.linalg.svd / eigh), bit 4 = a bounded dependency wait of a
    persistent linear-algebra kernel expired (RuntimeError)."""
    if word & 2:
        raise np.linalg.LinAlgError("Singular matrix")
    if word & 4:
        raise np.linalg.LinAlgError("SVD did not converge")
    if word & 8:
        raise np.linalg.LinAlgError("Eigenvalues did not converge")
    if word & 16:
        raise DeviceWaitExpired(
            "hip linker: a persistent linear-algebra kernel (Cholesky / triangular solve / LU panel) gave up "
            "waiting for another workgroup (not all of its workgroups were resident — another kernel holding "
            "compute units on a second stream?); PTHIP_CHOL=steps / PTHIP_TRSM=generic select the launch-per-step forms"
        )
    if word & 1:
        raise IndexError("index out of bounds (device-side check)")


class Env:
    """Per-call state handed to the node handlers."""

    def __init__(self, exe: "HipExecutable"):
        self.exe = exe
        self.lib = ffi.lib()
        self.keepalive = []  # host arrays whose async H2D may still be in flight
        self.graph = exe.graph  # graph whose node is currently running (inner graphs swap it)
        self.node_events = None  # set by HipExecutable.profile_nodes
        self.kernel_timer = None  # KernelTimer: brackets individual generated-kernel launches
        self.scheduler = None  # StreamScheduler of a multi-stream frozen plan
        self.branch = 0  # branch taken by the IfElse node that is running
        self.donated = frozenset()  # input positions of the running node that may be overwritten
        self.node_key = None  # (executable id, node index) of the running node
        # var id -> DeviceArray the producing kernel should write into directly (a Scan's trace
        # slot): handlers that support it skip their own allocation, the caller skips the copy
        self.placement = {}
        # frozen plans: (device status word, pinned destination) — the Tail kernel, being the last
        # launch, carries the error word to the host along with its results
        self.tail_status = None
        self.tail_status_done = False
        self.tail_done_word = False
        self.tail_launch_mark = -1
        # frozen multi-stream plans: nodes of the latency-chain segment that read these graph inputs from the
        # pinned staging block itself instead of its device copy (plan.py _A_DIRECT)
        self.direct_nodes = None
        self.direct_inputs = None
        # inside a Scan step: which step-kernel outputs to keep in the MFMA operand order as well, and
        # the packed images of recent steps (dispatch/scan.py::_pack_plan, dispatch/dotew.py)
        self.scan_ctx = None

    def timed(self, name, launch):
        """Run ``launch()`` (ONE kernel launch) between two HIP events when a KernelTimer is
        attached (HipExecutable.profile_nodes); plain call otherwise."""
        kt = self.kernel_timer
        if kt is None:
            return launch()
        tok = kt.begin()
        r = launch()
        kt.end(name, tok)
        return r

    def to_device(self, v) -> DeviceArray:
        if isinstance(v, DeviceArray):
            return v
        if isinstance(v, DeferredReduce):
            return v.force(self)
        if isinstance(v, HostValue):
            if v.dev is None:
                a = v.a if v.a.flags.c_contiguous else np.ascontiguousarray(v.a)
                self.keepalive.append(a)
                v.dev = DeviceArray.from_host(a)
            return v.dev
        raise TypeError(f"cannot move {type(v)} to the device")

    def to_host(self, v) -> np.ndarray:
        """Value of a (small) variable on the host — a synchronisation point when
        the value lives on the device."""
        if isinstance(v, HostValue):
            return v.a
        if isinstance(v, DeferredReduce):
            v = v.force(self)
        if isinstance(v, DeviceArray):
            if self.exe._capturing:
                raise ffi.HipError("data-dependent host read inside a frozen (hipGraph) plan")
            return v.to_host(sync=True)
        return np.asarray(v)


class HipExecutable:
    """``auto_freeze``: after one eager call, a second call with the same input signature
    (shapes/dtypes, resident array identities, integer scalar values) captures the launch
    sequence into a :class:`~pytensor_amd.plan.FrozenPlan`; later calls with that signature
    replay it (one native call).  A different signature runs eagerly and re-arms the capture;
    graphs that cannot be frozen (data-dependent host reads) stay eager.  This is what
    ``pytensor.function(..., mode="hip")`` gets (``HipLinker.jit_compile``); direct users of
    the class opt in."""

    def __init__(self, graph: Graph, resident=(), device: int | None = None, fuse=True, auto_freeze=False,
                 update_map=None, tail=True):
        from pytensor_amd import dispatch  # registers handlers
        from pytensor_amd.passes import run_pipeline

        self.auto_freeze = bool(auto_freeze) and os.environ.get("PTHIP_AUTO_FREEZE", "1") != "0"
        self._auto_plan = None
        self._auto_sig = None  # signature seen on the previous eager call
        self._auto_failed = False
        self.source_graph = graph
        # per-node segment ids for multi-stream plans (fusion.segment_graph), or None
        self.graph, self.segments = run_pipeline(graph, fuse, tail=tail)
        self.resident = set(resident)
        # a captured launch sequence would replay the same Philox counters: graphs that draw
        # random numbers run eagerly (sampling graphs are not the logp+grad hot path)
        self.has_rng = any(v.kind == "rng" for v in self.graph.vars.values())
        # an all-reduce drains the stream and runs on RCCL's: not capturable either
        self.has_collective = _has_op(self.graph, "AllReduce")
        if self.has_rng or self.has_collective:
            self.auto_freeze = False
        self._handlers = dispatch.HANDLERS
        self._resident_cache = {}  # input position -> ResidentEntry
        self._res_gen = C.c_uint64(0)  # bumped whenever a resident is re-uploaded, re-watched or invalidated (csrc/fastplan.c)
        # update feedback (compile/executor.py:712-716 stores output i into the storage cell of
        # input j after every call): {output index: resident input position}.  The new value is
        # copied device-to-device into the resident buffer at the end of the call, so the next
        # call neither uploads it nor leaves the captured plan (an SGD-style `updates=` loop).
        self.update_map = {int(o): int(i) for o, i in (update_map or {}).items() if int(i) in self.resident}
        self.stats = {"resident_uploads": 0, "eager_calls": 0, "captures": 0, "replays": 0, "capture_failures": 0}
        self._warm = False
        self._const_cache = {}
        self._device = device
        self._capturing = False
        self._plans = {}
        # lazy IfElse: nodes that only feed one branch run when (and if) that branch is taken
        self._guard, self._branch_nodes = branch_guards(self.graph)
        if self._branch_nodes:
            self.auto_freeze = False  # the condition is read on the host: nothing to capture
        self._scalar_outs = [k for k, vid in enumerate(self.graph.outputs) if self.graph.vars[vid].kind == "scalar"]
        self._last_use = self._compute_last_use()
        self._donations = self._compute_donations()
        # fail loudly and early if the library / device is unusable
        ffi.lib()

    # ------------------------------------------------------------------
    def _compute_last_use(self):
        last = {}
        guard = self._guard

        def when(k):  # a guarded node runs when its (outermost) IfElse does
            while guard[k] is not None:
                k = guard[k][0]
            return k

        for k, n in enumerate(self.graph.nodes):
            for i in n.inputs:
                last[i] = max(last.get(i, -1), when(k))
        keep = set(self.graph.outputs)
        free_after = [[] for _ in self.graph.nodes]
        for vid, k in last.items():
            if vid not in keep and self.graph.vars[vid].const is None:
                free_after[k].append(vid)
        return free_after

    # ops whose outputs are freshly allocated, exclusively owned buffers (never views)
    _FRESH_OPS = frozenset(
        ["Alloc", "AllocEmpty", "Elemwise", "ElemwiseReduce", "GemvChain", "AdvancedSubtensor", "Gemv", "Gemm", "Dot22",
         "Dot22Scalar", "BatchedDot", "Ger", "Join", "DeepCopyOp", "IncSubtensor", "AdvancedIncSubtensor",
         "Cholesky", "SolveTriangular", "CholeskySolve", "Blockwise", "GemvFinish", "SeqDot22", "CholeskyTrsv"]
    )

    def _compute_donations(self):
        """Per node: input positions whose buffer the node may overwrite in place — the
        value is produced by a fresh-buffer op, this node is its only consumer and it is not
        a graph output.  The functional replacement for the reference's ``inplace`` rewrites
        (``destroy_map``), decided on the lowered graph instead of by graph rewriting."""
        g = self.graph
        producer, consumers = {}, {}
        for k, n in enumerate(g.nodes):
            for o in n.outputs:
                producer[o] = n.op
            for i in n.inputs:
                consumers[i] = consumers.get(i, 0) + 1
        outs = set(g.outputs)
        res = []
        for n in g.nodes:
            d = set()
            for pos, v in enumerate(n.inputs):
                if producer.get(v) in self._FRESH_OPS and consumers.get(v) == 1 and v not in outs:
                    d.add(pos)
            res.append(d)
        return res

    def _ensure_device(self):
        if ffi.device_count() <= 0:
            raise ffi.HipError("no HIP device visible: the hip linker has no CPU fallback")
        ffi.init(self._device if self._device is not None else 0)

    def _const(self, vid, env):
        v = self.graph.vars[vid]
        if vid in self._const_cache:
            return self._const_cache[vid]
        if v.const is None and v.kind == "tensor":  # (a NoneConst argument is a constant whose value IS None)
            raise KeyError(f"hip linker: variable {vid} ({v.name}) has no value yet and is not a constant")
        a = np.asarray(v.const)
        if v.kind != "tensor" or a.size <= HOST_MAX:
            # small constants (alpha/beta of Gemv, fill values, shapes) stay on the host and
            # are uploaded lazily, once, when a kernel needs them as an operand
            val = HostValue(a)
        else:
            val = DeviceArray.from_host(a)
            env.keepalive.append(a)
        self._const_cache[vid] = val
        return val

    def _input(self, pos, vid, value, env):
        var = self.graph.vars[vid]
        if isinstance(value, (DeviceArray, HostValue)):
            return value
        if var.kind == "rng":  # numpy Generator (or its state dict) -> Philox key + counter
            from pytensor_amd.rng import RngState

            return RngState.from_generator(value)
        if var.kind != "tensor":
            return HostValue(np.asarray(value))
        a = np.asarray(value)
        if str(a.dtype) != var.dtype:
            raise TypeError(f"input {pos} ({var.name}): expected dtype {var.dtype}, got {a.dtype}")
        if a.ndim != var.ndim:
            raise TypeError(f"input {pos} ({var.name}): expected {var.ndim} dimensions, got {a.ndim}")
        for d, s in enumerate(var.shape):
            if s is not None and a.shape[d] != s:
                raise TypeError(f"input {pos} ({var.name}): static shape {var.shape} violated by {a.shape}")
        if pos in self.resident:
            return self._refresh_resident(pos, value, a, env)
        if a.dtype.kind in "iub" and a.ndim == 0:
            return HostValue(a)
        dev = DeviceArray.from_host(a)
        env.keepalive.append(a)
        return dev

    def _refresh_resident(self, pos, value, a=None, env=None) -> DeviceArray:
        """Device copy of resident input ``pos`` for the host value in the storage cell: reused
        when the cell still holds the array that was uploaded (same object, same memory, same
        content fingerprint); otherwise uploaded again — *into the same device buffer* when
        shape and dtype are unchanged, so captured plans stay valid across ``set_value``."""
        ent = self._resident_cache.get(pos)
        if ent is not None and value is ent.host and ent.key is not None:
            # the very object that was uploaded (an ndarray cannot move its memory): only the
            # content can have changed
            if coherence.clean(ent.fp, value):
                return ent.dev
            a, key = value, ent.key
        else:
            if a is None:
                a = np.asarray(value)
            key = _resident_key(value, a)
            if ent is not None and ent.key == key and coherence.clean(ent.fp, a):
                return ent.dev
        self.stats["resident_uploads"] += 1
        self._res_gen.value += 1  # native FastPlans of this executable re-validate through the Python path (csrc/fastplan.c)
        # Watch FIRST, copy second: a store from another thread between the two is then either in the copy (it came
        # before the protection / the hash) or marks the value dirty (it faulted, or the next hash differs) — with the
        # copy first it could land after the snapshot and before the watch and never be seen.  pthip_h2d reads
        # write-protected pages through its pinned bounce buffers (runtime.hip), the protection stays.
        if ent is not None and ent.dev.shape == a.shape and ent.dev.dtype == a.dtype:
            ent.rewatch(key, value, a)
            if a.size:
                c = a if a.flags.c_contiguous else np.ascontiguousarray(a)
                ffi.check(ffi.lib().pthip_h2d(ent.dev.ptr, c.ctypes.data, c.nbytes))
                if env is not None:
                    env.keepalive.append(c)
            return ent.dev
        if ent is not None:
            coherence.release(ent.fp)
            ent.fp = None
        fp = coherence.watch(a)
        dev = DeviceArray.from_host(a)
        if env is not None:
            env.keepalive.append(a)
        self._resident_cache[pos] = ResidentEntry(key, dev, value, fp)
        return dev

    def invalidate_resident(self, pos=None):
        """Forget what is known about the host side of resident inputs (all, or one position):
        the next call uploads them again (needed only in the opt-in `sampled` / `trust` modes)."""
        self._res_gen.value += 1
        for p, ent in self._resident_cache.items():
            if pos is None or p == pos:
                ent.key = None

    # -- update feedback -------------------------------------------------------------------
    def _feed_updates_device(self, outs, dry=False):
        """Enqueue ``resident[pos] <- outs[o]`` (device to device) for every update pair — behind,
        in stream order, every read of the old values: the graph's own kernels AND the copies that
        carry the outputs to the host (an output may BE the old resident: ``insert_deepcopy``,
        compile/aliasing.py:165-260, does not copy an output that aliases an updated input).
        Two phases, because ``Function`` stores all updates after the call, i.e. simultaneously
        (compile/executor.py:712-716): a source that lives in ANY resident buffer written here
        (``{w_prev: w, w: f(w)}``, a swap ``{x: y, y: x}``, a shifted view of its own buffer) is
        first copied aside; only then is any resident written.  Returns the (pos, o) pairs fed.
        ``dry``: stage the copies but write nothing — the sizing pass of a frozen plan runs for real and
        must not advance the shared state, yet has to make the same allocations as the capture."""
        from pytensor_amd.device import copy_into

        todo = []
        for o, pos in self.update_map.items():
            ent = self._resident_cache.get(pos)
            src = outs[o]
            if ent is None or not isinstance(src, DeviceArray) or src.shape != ent.dev.shape or src.dtype != ent.dev.dtype:
                continue  # (shape changed: the next call uploads the new host value)
            todo.append((pos, o, ent, src))
        written = {id(ent.dev.buf) for _, _, ent, _ in todo}
        staged = []
        for pos, o, ent, src in todo:
            same = src.buf is ent.dev.buf and src.offset == ent.dev.offset and src.strides == ent.dev.strides
            if not same and id(src.buf) in written:
                src = src.contiguous_copy()
            staged.append((pos, o, ent, src, same))
        for pos, o, ent, src, same in staged:
            if not same and not dry:
                copy_into(ent.dev, src)
        return [(pos, o) for pos, o, *_ in staged]

    def _feed_updates_host(self, fed, host):
        """The arrays just returned are what ``Function`` installs in the storage cells: record
        them as the host mirrors of the (already updated) resident buffers."""
        for pos, o in fed:
            ent = self._resident_cache[pos]
            h = host[o]
            coherence.release(ent.fp)
            ent.key, ent.host, ent.fp = _resident_key(h, h), h, coherence.watch_update_fed(h)

    # ------------------------------------------------------------------
    def run_device(self, inputs, env=None):
        """Run the graph; returns the list of output values (DeviceArray/HostValue)."""
        g = self.graph
        if len(inputs) != len(g.inputs):
            raise TypeError(f"expected {len(g.inputs)} inputs, got {len(inputs)}")
        env = env or Env(self)
        outer_graph, env.graph = env.graph, g
        try:
            return self._run_nodes(g, inputs, env), env
        finally:
            env.graph = outer_graph

    def _run_nodes(self, g, inputs, env):
        vals = {}
        for pos, (vid, value) in enumerate(zip(g.inputs, inputs)):
            vals[vid] = self._input(pos, vid, value, env)
        handlers = self._handlers
        evs = env.node_events if env.exe is self else None  # inner graphs are not itemised
        sched = env.scheduler if env.exe is self else None  # multi-stream plans (plan.py)
        guard = self._guard
        open_branches, taken = set(), {}  # lazy IfElse: (node, branch) pairs whose nodes may run
        todo = list(range(len(g.nodes)))[::-1]  # a stack: a taken branch pushes its nodes in front
        while todo:
            k = todo.pop()
            node = g.nodes[k]
            if guard[k] is not None and guard[k] not in open_branches:
                continue
            if node.op == "IfElse" and k not in taken:
                # ifelse.py:300-345 (the lazy thunk): the condition first, then only the taken branch
                cond = vals.get(node.inputs[0])
                if cond is None:
                    cond = self._const(node.inputs[0], env)
                b = 0 if np.asarray(env.to_host(cond)).item() != 0 else 1
                taken[k] = b
                open_branches.add((k, b))
                todo.append(k)
                todo.extend(reversed(self._branch_nodes.get((k, b), ())))
                continue
            ins = []
            lazy = None
            if node.op == "IfElse":
                # the inputs of the branch NOT taken were never computed (lazy, ifelse.py:300-345)
                # and are not constants: the handler gets a placeholder for them
                n_out = len(node.outputs)
                b = taken[k]
                lazy = set(range(1 + (1 - b) * n_out, 1 + (2 - b) * n_out))
            direct = env.direct_inputs if (env.direct_nodes is not None and env.exe is self and k in env.direct_nodes) else None
            for pos, i in enumerate(node.inputs):
                v = vals.get(i)
                if direct is not None and i in direct:
                    v = direct[i]
                if v is None:
                    v = None if (lazy is not None and pos in lazy and g.vars[i].const is None and g.vars[i].kind == "tensor") else self._const(i, env)
                ins.append(v)
            h = handlers.get(node.op)
            if h is None:
                raise NotImplementedError(f"hip linker: no device handler for {node.op}")
            if node.op == "IfElse":
                env.branch = taken[k]
            try:
                if evs is not None:
                    ffi.check(env.lib.pthip_event_record(evs[2 * k]))
                if sched is not None:
                    sched.before_node(k, node)
                env.donated = self._donations[k]
                env.node_key = (id(self), k)
                outs = h(node, ins, env)
                if evs is not None:
                    ffi.check(env.lib.pthip_event_record(evs[2 * k + 1]))
            except Exception as e:
                # the analogue of raise_with_op (pytensor/link/utils.py:271): say which Apply failed
                desc = ", ".join(
                    f"{type(v).__name__}{getattr(v, 'shape', '')}:{getattr(v, 'dtype', '')}" for v in ins
                )
                msg = f"\nhip linker: while running node {k} {node.op}({desc})"
                if e.args and isinstance(e.args[0], str):
                    e.args = (e.args[0] + msg, *e.args[1:])
                raise
            for o, val in zip(node.outputs, outs):
                vals[o] = val
            for dead in self._last_use[k]:
                vals.pop(dead, None)
        outs = []
        for o in g.outputs:
            v = vals.get(o)
            if v is None:
                v = self._const(o, env)
            outs.append(v)
        return outs

    def _signature(self, inputs):
        sig = []
        for pos, (vid, v) in enumerate(zip(self.graph.inputs, inputs)):
            a = v if isinstance(v, np.ndarray) else np.asarray(v)
            if pos in self.resident and self.graph.vars[vid].kind == "tensor":
                # content changes are absorbed by an in-place re-upload; only the geometry matters
                sig.append(("r", a.shape, a.dtype.str))
                continue
            if self.graph.vars[vid].kind != "tensor" or (a.dtype.kind in "iub" and a.ndim == 0):
                sig.append(("b", a.dtype.str, a.tolist()))  # baked into a plan
            else:
                sig.append((a.shape, a.dtype.str))
        return tuple(sig)

    def __call__(self, *inputs):
        try:
            return self._call(*inputs)
        except DeviceWaitExpired:
            # A cooperative kernel (the LU panel exchanges a column per step among its workgroups) did not get all of its
            # workgroups onto the device within its spin limit — another process's kernels held the compute units
            # (several PyMC chains on one GPU).  Not an error of the graph: evaluate it again with the launch-per-step
            # forms, which wait for nobody, and stay on them (include/pthip.h pthip_set_safe_mode).
            # Safe mode is PROCESS-wide but plans are per executable: after executable A has switched the process over,
            # executable B (or a Scan's inner executable) still holds a plan captured WITH the cooperative kernels and
            # can hit the expired wait on its next replay.  So: always drop this executable's stale plan and evaluate
            # once eagerly under safe mode; only a failure of THAT evaluation is an error.
            lib = ffi.lib()
            already = bool(lib.pthip_set_safe_mode(1))
            had_stale_plan = self._auto_plan is not None
            if already and not had_stale_plan:
                raise  # the launch-per-step forms themselves expired: something else is wrong
            if not already:
                import warnings

                warnings.warn("hip linker: a cooperative linear-algebra kernel gave up waiting for its own workgroups (the device is shared "
                              "with other work); evaluating again with the launch-per-step forms and staying on them for this process",
                              RuntimeWarning, stacklevel=2)
            if had_stale_plan:
                self._auto_plan.close()  # (captured with the cooperative kernels)
                self._auto_plan = None
            self.stats["safe_mode_retries"] = self.stats.get("safe_mode_retries", 0) + 1
            return self._call_eager(*inputs)

    def _call(self, *inputs):
        if self.auto_freeze:
            if self._auto_plan is not None:
                # the plan checks the signature of what it is given itself (before launching
                # anything): no second pass over the inputs on the replay path
                from pytensor_amd.plan import SignatureChanged

                try:
                    res = self._auto_plan(*inputs)
                    self.stats["replays"] += 1
                    return res
                except SignatureChanged:
                    self._auto_plan.close()  # the signature moved on: capture again later
                    self._auto_plan = None
                sig = self._signature(inputs)
            else:
                sig = self._signature(inputs)
            if self._auto_plan is None and sig == self._auto_sig and not self._auto_failed:
                plan = None
                try:
                    plan = self.freeze(*inputs, multi_stream="auto")
                except Exception as e:  # noqa: BLE001 — *any* capture-time failure means "not freezable":
                    # data-dependent host reads, an allocation that missed the arena, a HIP call that
                    # stream capture does not support, a handler's NotImplementedError under capture.
                    # Eager execution is unaffected; explicit freeze() calls still raise.
                    self._auto_failed = True
                    self.stats["capture_failures"] += 1
                    if "data-dependent host read" not in str(e) and os.environ.get("PTHIP_WARN_CAPTURE", "1") != "0":
                        import warnings

                        warnings.warn(f"hip linker: hipGraph capture failed, staying on the eager path: {type(e).__name__}: {e}",
                                      RuntimeWarning, stacklevel=2)
                if plan is not None:
                    self._auto_plan, self._auto_plan_sig = plan, sig
                    self.stats["captures"] += 1
                    self.stats["replays"] += 1
                    return plan(*inputs)
            self._auto_sig = sig
        return self._call_eager(*inputs)

    def _call_eager(self, *inputs):
        self._ensure_device()
        self.stats["eager_calls"] += 1
        outs, env = self.run_device(inputs)
        self._warm = True
        lib = ffi.lib()
        host = []
        for o, vid in zip(outs, self.graph.outputs):
            var = self.graph.vars[vid]
            if var.kind == "rng":  # the advanced generator, as a numpy Generator(Philox) again
                host.append(o.to_generator())
            elif isinstance(o, HostValue):
                host.append(np.array(o.a, dtype=var.dtype if var.kind == "tensor" else o.a.dtype, copy=True))
            else:
                host.append(o.to_host(sync=False))
        ffi.check(lib.pthip_synchronize())
        st = C.c_int(0)
        ffi.check(lib.pthip_check_status(C.byref(st)))
        raise_device_status(st.value)  # (a call that raises commits no update: Function does not either)
        env.keepalive.clear()
        if self.update_map:
            # only now: the outputs (old resident values among them) are on the host, the call succeeded
            fed = self._feed_updates_device(outs)
            if fed:
                self._feed_updates_host(fed, host)
        for k in self._scalar_outs:  # a ScalarType output is a NumPy scalar, not a 0-d array (scalar/basic.py ScalarType.filter)
            host[k] = host[k][()]
        if not self.graph.outputs:
            return None  # link/basic.py:690-699: a function without outputs must return None
        return tuple(host)

    # ------------------------------------------------------------------
    def profile_nodes(self, inputs, reps=10):
        """Device time of every node (HIP events on the context stream around each
        handler, eager mode, averaged over ``reps`` runs): ``[(k, op, ms), ...]``."""
        self._ensure_device()
        lib = ffi.lib()
        n = len(self.graph.nodes)
        evs = []
        for _ in range(2 * n):
            e = C.c_void_p()
            ffi.check(lib.pthip_event_create(C.byref(e)))
            evs.append(e)
        tot = [0.0] * n
        self.last_kernel_times = {}
        kt = KernelTimer()
        try:
            self(*inputs)  # warm: kernels compiled, residents uploaded
            for _ in range(reps):
                env = Env(self)
                env.node_events = evs
                env.kernel_timer = kt
                self.run_device(inputs, env)
                ffi.check(lib.pthip_synchronize())
                for k in range(n):
                    ms = C.c_float()
                    ffi.check(lib.pthip_event_elapsed_ms(evs[2 * k], evs[2 * k + 1], C.byref(ms)))
                    tot[k] += ms.value
                kt.resolve()
            self.last_kernel_times = kt.mean_ms()
        finally:
            for e in evs:
                lib.pthip_event_destroy(e)
        return [(k, self.graph.nodes[k].op, tot[k] / reps) for k in range(n)]

    # ------------------------------------------------------------------
    def freeze(self, *inputs, fetch_outputs=True, multi_stream=True):
        """Capture the launch sequence for this input signature into a hipGraph and
        return a :class:`FrozenPlan` (see ``pytensor_amd/plan.py``).

        ``multi_stream="auto"`` (what the linker path uses): when the graph has a latency
        chain to overlap, capture both the one-graph and the three-graph/two-stream form, time
        a few replays of each and keep the faster — the two extra ``hipGraphLaunch`` calls cost
        ≈20 µs of host time, more than the overlap saves on small problems (measured,
        tools/bench_small.py: N=3000: 142 vs 162 µs per call; N=1e6: 300 vs 249 µs)."""
        from pytensor_amd.plan import FrozenPlan

        if self.has_rng:
            raise NotImplementedError("hip linker: a graph that draws random numbers cannot be frozen "
                                      "(a replay would repeat the captured Philox counters)")
        if self.has_collective:
            raise NotImplementedError("hip linker: a graph with an all-reduce cannot be frozen "
                                      "(the collective runs outside the captured stream)")
        self._ensure_device()
        if not self._warm:
            # never run before: constants are uploaded lazily and would otherwise be allocated
            # inside the plan's arena: one plain device pass first (no update feedback)
            _, env = self.run_device(inputs)
            ffi.check(ffi.lib().pthip_synchronize())
            env.keepalive.clear()
            self._warm = True
        if multi_stream != "auto":
            return FrozenPlan(self, inputs, fetch_outputs=fetch_outputs, multi_stream=bool(multi_stream))
        single = FrozenPlan(self, inputs, fetch_outputs=fetch_outputs, multi_stream=False)
        if self.segments is None:
            return single
        multi = FrozenPlan(self, inputs, fetch_outputs=fetch_outputs, multi_stream=True)
        if not multi.segmented:
            multi.close()
            return single
        import time

        def wall(plan, reps=12):
            for _ in range(3):
                plan._replay(True)
            t0 = time.perf_counter()
            for _ in range(reps):
                plan._replay(True)
            return time.perf_counter() - t0

        # the contest replays the plans for real: with update feedback that would advance the
        # shared state, so it is snapshotted and put back
        saved = [(pos, self._resident_cache[pos].dev, self._resident_cache[pos].dev.contiguous_copy())
                 for pos in set(self.update_map.values()) if pos in self._resident_cache]
        t_single, t_multi = wall(single), wall(multi)
        from pytensor_amd.device import copy_into

        for _pos, dev, snap in saved:
            copy_into(dev, snap)
        keep, drop = (single, multi) if t_single <= t_multi else (multi, single)
        drop.close()
        return keep
