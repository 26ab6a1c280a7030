"""``FrozenPlan`` — one ``Function`` call as one (or three) hipGraph launches.

The reference's fastest runtime is the CVM (pytensor/link/c/c_code/lazylinker_c.c:749
``CLazyLinker_call``): one native call per ``Function.__call__`` that walks
pre-resolved node tables instead of dispatching from Python.  The MI355X-native
analogue is a captured hipGraph: for a fixed input signature (shapes/dtypes, resident
arrays) the launch sequence, every intermediate buffer and both PCIe staging copies are
frozen once and replayed.

How: (1) a warm-up run inside a private pool *arena* discovers every allocation;
(2) the arena is rewound and the same sequence is run again under stream capture —
allocations are now served from the arena's free lists (no ``hipMalloc`` while
capturing); all non-resident inputs travel in ONE pinned staging block (one H2D), all
outputs are gathered by one ``pthip_pack`` launch into one block (one D2H);
(3) replays copy the small parameter arrays into the staging block, launch, synchronise
once and copy the outputs out.

**Segmented plans.**  ``fusion.segment_graph`` splits the nodes into A (the single-CU,
latency-bound Cholesky/solve chain), B (the HBM-streaming work, independent of A) and
C (what combines them).  ROCm 7.2 executes the branches of one captured graph one after
the other (measured: profiles/r1f_*), so A, B and C are captured as three graphs and
replayed as  ``H2D → [A on stream 1 ‖ B on stream 0] → C on stream 0``: the ≈0.25 ms
latency chain hides under the streaming kernels.  Segmented plans run with the arena in
no-reuse mode (stream order no longer serialises all users of a block).

Data-dependent host reads (a ``ScalarFromTensor`` of a computed value used as a
shape) cannot be frozen; ``freeze`` raises and the caller keeps the eager path.
"""

from __future__ import annotations

import ctypes as C
import os
import weakref

import numpy as np

from pytensor_amd import coherence, ffi
from pytensor_amd.device import Buffer, DeviceArray, contiguous_strides
from pytensor_amd.executor import Env, HostValue

_ALIGN = 256


def _round(n):
    return (n + _ALIGN - 1) // _ALIGN * _ALIGN


class _PinnedBlock:
    """One pinned host allocation carved into aligned NumPy views."""

    def __init__(self, specs):
        self.offsets = []
        off = 0
        for shape, dtype in specs:
            self.offsets.append(off)
            n = int(np.prod(shape)) if len(shape) else 1
            off += _round(max(n * np.dtype(dtype).itemsize, 1))
        self.nbytes = max(off, _ALIGN)
        p = C.c_void_p()
        ffi.check(ffi.lib().pthip_host_alloc(self.nbytes, C.byref(p)))
        self.ptr = p.value
        coherence.register_pinned(self.ptr, self.nbytes)  # (these pages are never write-protected)
        raw = (C.c_char * self.nbytes).from_address(self.ptr)
        self.views = []
        for (shape, dtype), o in zip(specs, self.offsets):
            n = int(np.prod(shape)) if len(shape) else 1
            self.views.append(np.frombuffer(raw, dtype=dtype, count=n, offset=o).reshape(shape))

    def free(self):
        if self.ptr:
            self.views = []
            coherence.unregister_pinned(self.ptr)
            ffi.lib().pthip_host_free(self.ptr)
            self.ptr = 0


class _ResultRing:
    """Pinned result blocks that are handed to the caller as the output arrays themselves.

    A replay's D2H lands in a block of this ring and the views on it are returned without a copy
    (an 800 kB gradient costs ~40 us to copy — more than the evaluation).  The block comes back
    when the last array on it is collected; while every block is still held by the caller, calls
    fall back to the plan's fixed block and copy out of it (fresh arrays either way,
    link/basic.py:670-684)."""

    def __init__(self, specs, limit):
        self.specs, self.limit = specs, limit
        self.free, self.made, self.closed = [], 0, False
        self._layout = None  # (dtype, count, shape) per result, offsets from the first block

    def take(self):
        if self.free:
            return self.free.pop()
        if self.made < self.limit:
            try:
                blk = _PinnedBlock(self.specs)
            except ffi.HipError:  # no more pinnable memory: this ring stays at its current size
                self.limit = self.made
                return None
            self.made += 1
            return blk
        return None

    def hand_out(self, blk):
        """fresh views of ``blk`` whose collection returns it to the ring"""
        if self._layout is None:
            self._layout = [(np.dtype(dtype), int(np.prod(shape)) if len(shape) else 1, tuple(shape), o)
                            for (shape, dtype), o in zip(self.specs[:-1], blk.offsets)]
            self._ctype = C.c_char * blk.nbytes
        raw = self._ctype.from_address(blk.ptr)
        weakref.finalize(raw, self._give_back, blk).atexit = False
        fb = np.frombuffer
        return [fb(raw, dtype=dt, count=n, offset=o).reshape(shape) for dt, n, shape, o in self._layout]

    def _give_back(self, blk):
        if self.closed:
            blk.free()
        else:
            self.free.append(blk)

    def close(self):
        self.closed = True
        while self.free:
            self.free.pop().free()


class _Src:
    """A raw device range handed to the pack kernel (the error word)."""

    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes


class _PinnedAsBuffer:
    """The pinned (device-visible) output block seen as a buffer, so that a ``DeviceArray`` can
    name a slot of it: the Tail kernel stores its results there directly (no pack launch)."""

    __slots__ = ("ptr", "nbytes")

    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes


class _SegmentSwitch:
    """Scheduler hook: tells the plan when the executor crosses a segment boundary (A → B → C) —
    during the warm-up pass to size the segments, during the build pass to close the running
    capture / recording and open the next one."""

    def __init__(self, plan, seg):
        self.plan = plan
        self.seg = seg
        self.capturing = False
        self.sizing = False

    def before_node(self, k, node):
        if k > 0 and self.seg[k] != self.seg[k - 1]:
            if self.capturing:
                self.plan._segment_boundary(self.seg[k - 1], self.seg[k])
                self.plan._end_segment()
                self.plan._between_segments(self.seg[k - 1], self.seg[k])
                self.plan._begin_segment()
            elif self.sizing:
                self.plan._note_boundary()


# results up to this size are written by the pack kernel directly into pinned host memory
_ZEROCOPY_MAX = int(os.environ.get("PTHIP_ZEROCOPY_MAX", 1 << 16))

# larger results: this many pinned blocks per plan are handed out as the result arrays (0: always copy)
_OUT_RING = int(os.environ.get("PTHIP_OUT_RING", 4))

# a segment of at most this many launches is replayed as a recorded launch list (direct launches:
# no hipGraph launch floor, no graph-to-graph boundary); longer ones as captured hipGraphs
_LIST_MAX = int(os.environ.get("PTHIP_LIST_MAX", 8))

# completion by polling the word the plan's last kernel (the Tail) stores behind its results, instead of waiting
# for the stream's completion signal (tools/ubench/call_lat.hip: ~5 us per call)
_POLL = os.environ.get("PTHIP_PLAN_POLL", "1") != "0"

# the latency-chain segment reads its staged parameters straight from the pinned staging block when every
# consumer there reads them once (a right-hand side): it no longer depends on the parameter upload, and no
# event sits between that upload and the streaming segment
_A_DIRECT = os.environ.get("PTHIP_PLAN_A_DIRECT", "1") != "0"
# the two streams of a segmented plan meet on the device: segment A ends with a one-thread launch that stores a
# signal word (pthip_join_signal), the Tail node that opens segment C waits for it inside its own launches — no event
# between the streams (5.6 us between `gchain` and the next launch with one, profiles/r4z_c4_timeline.md)
_DEV_JOIN = os.environ.get("PTHIP_PLAN_DEVICE_JOIN", "1") != "0"
# ... which leans on the caches having been invalidated when the waiting kernel started (csrc/tail_device.h): checked
# once per process on the device itself (pthip_join_probe: in-place results that change every round, readers without a
# fence); a device / driver / partition mode where that does not hold keeps the event.  PTHIP_JOIN_PROBE=0 skips it.
_JOIN_PROBED = [None]


def _device_join_ok(lib) -> bool:
    if _JOIN_PROBED[0] is None:
        if os.environ.get("PTHIP_JOIN_PROBE", "1") == "0":
            _JOIN_PROBED[0] = True
        else:
            bad = C.c_int(-1)
            ffi.check(lib.pthip_join_probe(64, C.byref(bad)))
            _JOIN_PROBED[0] = bad.value == 0
            if bad.value:
                import warnings

                warnings.warn(f"pytensor_amd: the device-side join probe failed on this device ({bad.value}); "
                              "segmented plans keep an event between their streams", RuntimeWarning, stacklevel=2)
    return _JOIN_PROBED[0]
_A_DIRECT_OPS = ("CholeskyTrsv", "SolveTriangular", "CholeskySolve")


class _ReplayDesc(C.Structure):
    """``pthip_replay_desc`` (include/pthip.h)"""

    _fields_ = [(n, C.c_void_p) for n in ("ga", "la", "gb", "lb", "gc", "lc", "dev_in", "host_in")] + [
        ("in_bytes", C.c_size_t), ("dev_out", C.c_void_p), ("out_bytes", C.c_size_t), ("flags", C.c_int)]


try:  # the native call path of a plan (host code only; built by __graft_entry__.build next to libpthip.so)
    from pytensor_amd import _fastplan as _FASTPLAN
except ImportError:  # pragma: no cover (not built: the Python path below serves every plan)
    _FASTPLAN = None

_CAPTURE_ACTIVE = [None]  # the plan currently inside warm-up/capture, if any
_DEFERRED = []  # plans whose release was requested during somebody else's capture


def _drain_deferred():
    while _DEFERRED and _CAPTURE_ACTIVE[0] is None:
        _DEFERRED.pop().close()


class SignatureChanged(TypeError, ValueError):
    """A replay was asked for inputs of another signature (shape / dtype of a staged or resident
    input, value of a baked scalar) than the plan was captured for.  Raised before anything is
    launched; the auto-freeze path of ``HipExecutable`` answers it with an eager call."""


class FrozenPlan:
    def __init__(self, exe, inputs, fetch_outputs=True, multi_stream=True):
        self._closed = False
        _drain_deferred()
        self.exe = exe
        self.fetch_outputs = fetch_outputs  # False: device-side timing only (no pack/D2H node)
        self.lib = ffi.lib()
        g = exe.graph
        if len(inputs) != len(g.inputs):
            raise TypeError(f"expected {len(g.inputs)} inputs, got {len(inputs)}")
        self._arena = C.c_void_p()
        self._graphs = []  # segments in order: ("graph", hipGraphExec handle) | ("list", LaunchList handle)
        self._seg_sizes = []  # launches per segment, measured in the warm-up pass
        self._mark = 0
        # update feedback writes the resident buffers: a recording pass executes what it records,
        # so such plans are captured (capture does not execute)
        self._use_lists = _LIST_MAX > 0 and not exe.update_map
        self._staged = []  # input positions travelling through the staging block
        self._baked = {}  # position -> host value baked into the plan (int scalars)
        self._resident_devs = {}  # position -> DeviceArray (kept alive: its address is in the graphs)
        self._fed = []  # (resident position, output index) pairs written back inside the graph
        self._sig = []
        specs = []
        for pos, (vid, value) in enumerate(zip(g.inputs, inputs)):
            var = g.vars[vid]
            a = np.asarray(value)
            self._sig.append((a.shape, str(a.dtype)))
            if pos in exe.resident and var.kind == "tensor":
                # the plan captures the address of the executable's stable resident buffer; new
                # host values of the same geometry are copied into it (executor._refresh_resident)
                self._resident_devs[pos] = exe._refresh_resident(pos, value, a)
            elif var.kind != "tensor" or (a.dtype.kind in "iub" and a.ndim == 0):
                self._baked[pos] = a.copy()
            else:
                self._staged.append(pos)
                specs.append((a.shape, a.dtype))
        self._in_block = _PinnedBlock(specs)
        self._in_view = {pos: k for k, pos in enumerate(self._staged)}
        self._dev_in = Buffer(self._in_block.nbytes) if self._staged else None  # outside the arena
        seg = exe.segments if multi_stream else None
        self.segmented = seg is not None and len(set(seg)) == 3 and seg[0] == 0
        self._switch = _SegmentSwitch(self, seg) if self.segmented else None
        # segment-A nodes that may read staged parameters from the pinned block itself (see _A_DIRECT)
        self._a_direct_nodes = None
        # (every node reading the pinned block, no upload at all: measured 0.1952-0.1974 ms per evaluation of config #4
        #  against 0.1936-0.1944 with the 2 KB upload, profiles/r4_c4_all_direct_ab.txt: the streaming kernel's
        #  workgroups each fetch their parameters over the host link)
        if self.segmented and _A_DIRECT and self._staged:
            staged_vids = {g.inputs[pos] for pos in self._staged}
            users = [k for k, n in enumerate(g.nodes) if seg[k] == 0 and any(i in staged_vids for i in n.inputs)]
            if all(g.nodes[k].op in _A_DIRECT_OPS and g.nodes[k].inputs[0] not in staged_vids for k in users):
                self._a_direct_nodes = frozenset(users)
        # device-side join: segment C opens with the Tail node (its launches take the signal word)
        self._join_word = None
        self._join_used = False
        if self.segmented and _DEV_JOIN and g.nodes[seg.index(2)].op == "Tail" and _device_join_ok(self.lib):
            slot = C.c_void_p()
            ffi.check(self.lib.pthip_ticket_slot(C.byref(slot)))
            self._join_word = slot.value
        self._async_pending = False
        self._fast = None  # csrc/fastplan.c: the replay path as one native call, built after the first Python-path call
        self._fast_off = os.environ.get("PTHIP_FASTPLAN", "1") == "0"
        self._fast_misses = 0
        self._poll = False  # set by the capture pass when the Tail kernel is the plan's last launch
        self._desc = None
        self._out_block = None
        self._out_meta = None
        self._dev_out = None  # packed results on the device, copied out by the replay call itself
        self._ring = None
        self._keep = []
        _CAPTURE_ACTIVE[0] = self
        try:
            self._build(inputs)
        except Exception:
            _CAPTURE_ACTIVE[0] = None
            self.close()
            raise
        finally:
            _CAPTURE_ACTIVE[0] = None

    # ------------------------------------------------------------------
    def _note_boundary(self):
        n = int(self.lib.pthip_launch_count())
        self._seg_sizes.append(n - self._mark)
        self._mark = n

    def _segment_boundary(self, prev, nxt):
        """inside the capture / recording of segment ``prev``, about to close it"""
        if prev == 0 and self._join_word:
            ffi.check(self.lib.pthip_join_signal(self._join_word))  # segment A's last launch

    def _between_segments(self, prev, nxt):
        """no capture / recording is running"""
        if nxt == 2 and self._join_word:
            # a recorded segment C runs now: its join must find the word set whether or not A's signal ran (a captured
            # hipGraph does not); _build puts it back to 0
            self._set_join_word(1)

    def _set_join_word(self, value):
        self._join_host = np.array([value], dtype=np.int32)
        ffi.check(self.lib.pthip_h2d(self._join_word, self._join_host.ctypes.data, 4))
        ffi.check(self.lib.pthip_synchronize())

    def _segment_is_list(self, k):
        return self._use_lists and k < len(self._seg_sizes) and 0 < self._seg_sizes[k] <= _LIST_MAX

    def _begin_segment(self):
        self._cur_is_list = self._segment_is_list(len(self._graphs))
        ffi.check(self.lib.pthip_record_begin() if self._cur_is_list else self.lib.pthip_capture_begin())
        self.exe._capturing = True
        if self._switch is not None:
            self._switch.capturing = True

    def _end_segment(self):
        self.exe._capturing = False
        if self._switch is not None:
            self._switch.capturing = False
        h = C.c_void_p()
        if self._cur_is_list:
            n = C.c_int64(0)
            if self.lib.pthip_record_end(C.byref(h), C.byref(n)) != 0:
                self._list_failed = True  # (no exception: a traceback would keep arena blocks alive)
                h = C.c_void_p()
            self._graphs.append(("list", h))
        else:
            ffi.check(self.lib.pthip_capture_end(C.byref(h)))
            self._graphs.append(("graph", h))

    def _abort_segment(self):
        """close a running capture / recording after a failure, releasing what it produced"""
        self.exe._capturing = False
        h = C.c_void_p()
        if getattr(self, "_cur_is_list", False):
            self.lib.pthip_record_end(C.byref(h), None)
            if h:
                self.lib.pthip_list_destroy(h)
        else:
            self.lib.pthip_capture_end(C.byref(h))
            if h:
                self.lib.pthip_graph_destroy(h)

    def _upload_params(self):
        if self._dev_in is not None:
            ffi.check(self.lib.pthip_h2d(self._dev_in.ptr, self._in_block.ptr, self._in_block.nbytes))

    def _run_once(self, inputs, capture):
        exe, lib = self.exe, self.lib
        env = Env(exe)
        blk = self._in_block
        # the parameter H2D is a plain async copy issued before the graphs (both A and B
        # read the parameters; it is not part of any captured segment)
        if not capture:
            self._upload_params()
        dev_inputs = []
        for pos, value in enumerate(inputs):
            k = self._in_view.get(pos)
            if k is not None:
                v = blk.views[k]
                dev_inputs.append(DeviceArray(self._dev_in, blk.offsets[k], v.shape, contiguous_strides(v.shape), v.dtype))
            elif pos in self._baked:
                dev_inputs.append(HostValue(self._baked[pos]))
            elif pos in self._resident_devs:
                dev_inputs.append(self._resident_devs[pos])
            else:  # non-tensor shared input (cannot happen for freezable graphs: rng graphs stay eager)
                dev_inputs.append(value)
        if self._a_direct_nodes is not None:
            pin_in = _PinnedAsBuffer(blk.ptr, blk.nbytes)
            env.direct_nodes = self._a_direct_nodes
            env.direct_inputs = {exe.graph.inputs[pos]: DeviceArray(pin_in, blk.offsets[k], blk.views[k].shape, contiguous_strides(blk.views[k].shape), blk.views[k].dtype)
                                 for pos, k in self._in_view.items()}
        placed = {}
        if capture and self.fetch_outputs and self._out_block is not None and self._out_block.nbytes <= _ZEROCOPY_MAX:
            # results of the Tail node (the last launch) go straight into the pinned block
            ob = self._out_block
            g = exe.graph
            tail_outs = {o for n in g.nodes if n.op == "Tail" for o in n.outputs}
            pin = _PinnedAsBuffer(ob.ptr, ob.nbytes)
            k = 0
            for vid, meta in zip(g.outputs, self._out_meta):
                if meta is not None:
                    continue
                v = ob.views[k]
                if vid in tail_outs and g.outputs.count(vid) == 1:
                    placed[vid] = DeviceArray(pin, ob.offsets[k], v.shape, contiguous_strides(v.shape), v.dtype)
                k += 1
            if placed:
                env.placement.update(placed)
                env.tail_status = (lib.pthip_status_ptr(), ob.ptr + ob.offsets[-1], (ob.ptr + ob.offsets[-1] + 4) if _POLL else 0)
                if self._join_word:
                    env.tail_join = self._join_word
        if capture:
            env.scheduler = self._switch
            self._begin_segment()
        elif self._switch is not None:
            env.scheduler = self._switch
            self._switch.sizing = True
        if not capture:
            self._seg_sizes = []
            self._mark = int(lib.pthip_launch_count())
        ok = False
        try:
            outs, env = exe.run_device(dev_inputs, env)
            if self._out_block is None:
                specs = []
                self._out_meta = []
                for o in outs:
                    if isinstance(o, HostValue):
                        self._out_meta.append(np.array(o.a, copy=True))
                    else:
                        self._out_meta.append(None)
                        specs.append((o.shape, o.dtype))
                # last slot: the device error word (index out of range / singular inverse) rides
                # along with the results — a replay costs no extra sync to learn about it
                # (second word: the completion flag the Tail kernel sets behind its results, polled by the replay)
                specs.append(((2,), np.dtype("int32")))
                self._out_block = _PinnedBlock(specs)
                self._out_block.views[-1][:] = 0
                self._out_specs = specs
            ob = self._out_block
            dev_outs = [o.contiguous() for o in outs if not isinstance(o, HostValue)]
            if dev_outs and self.fetch_outputs:
                dev_outs = dev_outs + [_Src(lib.pthip_status_ptr(), 4)]
            if dev_outs and self.fetch_outputs and ob.nbytes <= _ZEROCOPY_MAX:
                # small results: the pack kernel stores straight into the pinned (device-visible,
                # coherent) host block — no copy node after it; what the Tail kernel already wrote
                # there (results, error word) is skipped
                todo = [(o, off) for o, off in zip(dev_outs, ob.offsets)
                        if not (isinstance(getattr(o, "buf", None), _PinnedAsBuffer) or (isinstance(o, _Src) and env.tail_status_done))]
                for c0 in range(0, len(todo), 16):
                    chunk = todo[c0 : c0 + 16]
                    n = len(chunk)
                    srcs = (C.c_void_p * n)(*[o.ptr for o, _ in chunk])
                    nb = (C.c_int64 * n)(*[o.nbytes for o, _ in chunk])
                    offs = (C.c_int64 * n)(*[off for _, off in chunk])
                    ffi.check(lib.pthip_pack(n, srcs, nb, offs, ob.ptr))
            elif dev_outs and self.fetch_outputs:
                dev_out = Buffer(ob.nbytes)
                # gather every output into one block: chunks of <= 16 buffers per launch
                for c0 in range(0, len(dev_outs), 16):
                    chunk = dev_outs[c0 : c0 + 16]
                    n = len(chunk)
                    srcs = (C.c_void_p * n)(*[o.ptr for o in chunk])
                    nb = (C.c_int64 * n)(*[o.nbytes for o in chunk])
                    offs = (C.c_int64 * n)(*ob.offsets[c0 : c0 + n])
                    ffi.check(lib.pthip_pack(n, srcs, nb, offs, dev_out.ptr))
                # the D2H is not part of the plan: the replay call issues it into the block of that call
                if capture:
                    self._keep.append(dev_out)
                    self._dev_out = dev_out
                    if self._ring is None and _OUT_RING > 0:
                        self._ring = _ResultRing(self._out_specs, _OUT_RING)
            if capture:
                self._join_used = bool(getattr(env, "tail_join_used", False))
                # poll mode needs the Tail kernel (which stores the completion word) to be the LAST launch
                self._poll = bool(_POLL and self.fetch_outputs and getattr(env, "tail_done_word", False) and self._dev_out is None
                                  and getattr(env, "tail_launch_mark", -1) == int(lib.pthip_launch_count()) and not exe.update_map)
                self._keep += [dev_outs, env.keepalive]
                if exe.update_map:
                    # update feedback inside the graph: resident[pos] <- outs[o] after every read
                    # of the old value (the warm-up pass must NOT do this: it runs for real)
                    self._fed = exe._feed_updates_device(outs)
            elif exe.update_map:
                exe._feed_updates_device(outs, dry=True)  # (same staging allocations as the capture, no writes)
            ok = True
        finally:
            if self._switch is not None:
                self._switch.sizing = False
            if capture:
                if ok:
                    self._end_segment()
                else:  # abort the capture / recording cleanly
                    self._abort_segment()
            elif ok:
                self._note_boundary()  # the last (or only) segment, output packing included
        return outs

    def _build(self, inputs):
        lib = self.lib
        for pos, k in self._in_view.items():
            np.copyto(self._in_block.views[k], np.asarray(inputs[pos]))
        # (1) warm-up inside the arena (single stream; discovers every allocation)
        ffi.check(lib.pthip_arena_begin(C.byref(self._arena)))
        if self.segmented:
            ffi.check(lib.pthip_arena_set_no_reuse(self._arena, 1))
        try:
            outs = self._run_once(inputs, capture=False)
            del outs
            ffi.check(lib.pthip_synchronize())
        finally:
            ffi.check(lib.pthip_arena_end())
        # (2) rewind + capture / record (one segment, or three)
        for attempt in (0, 1):
            self._list_failed = False
            ffi.check(lib.pthip_arena_begin(C.byref(self._arena)))
            try:
                outs = self._run_once(inputs, capture=True)
                del outs
            finally:
                ffi.check(lib.pthip_arena_end())
            if not self._list_failed:
                break
            # a sequence a launch list cannot repeat (an upload inside it): hipGraphs instead
            ffi.check(lib.pthip_synchronize())
            self._release_segments()
            self._keep.clear()  # arrays of the discarded pass hold arena blocks: rewind needs them gone
            self._fed = []
            self._use_lists = False
        if self.segmented and len(self._graphs) != 3:
            raise ffi.HipError(f"segmented plan captured {len(self._graphs)} graphs instead of 3")
        if self._join_word:
            self._set_join_word(0)  # (whatever the build passes left there)
        if self._dev_out is not None:
            # the result copy is issued per call (not captured): take the runtime's one-time set-up of
            # that copy path (8 ms on the second large D2H of a process, measured) here
            for _ in range(2):
                ffi.check(lib.pthip_d2h(self._out_block.ptr, self._dev_out.ptr, self._out_block.nbytes))
                ffi.check(lib.pthip_synchronize())
            if self._ring is not None:
                first = self._ring.take()  # the first call's block (pinning memory is slow)
                if first is not None:
                    self._ring.free.append(first)

    # ------------------------------------------------------------------
    def _make_desc(self):
        if self.segmented:
            sa, sb, sc = self._graphs
        else:
            sa, sb, sc = None, self._graphs[0], None
        g = lambda seg: seg[1] if (seg is not None and seg[0] == "graph") else None
        l = lambda seg: seg[1] if (seg is not None and seg[0] == "list") else None
        v = lambda h: h.value if isinstance(h, C.c_void_p) else h
        nb = self._in_block.nbytes if self._dev_in is not None else 0
        do = self._dev_out
        d = _ReplayDesc(v(g(sa)), v(l(sa)), v(g(sb)), v(l(sb)), v(g(sc)), v(l(sc)), self._dev_in.ptr if nb else None,
                        self._in_block.ptr if nb else None, nb, do.ptr if do is not None else None,
                        self._out_block.nbytes if (do is not None and self._out_block is not None) else 0,
                        (1 if (self._a_direct_nodes is not None and self.segmented) else 0)
                        | (2 if (self._join_used and self.segmented and self._poll) else 0))
        self._desc = d
        self._desc_ref = C.byref(d)
        ob = self._out_block
        self._done_ptr = (ob.ptr + ob.offsets[-1] + 4) if (ob is not None and self._poll) else None
        self._sync_mode = 2 if self._poll else 1

    def _replay(self, sync, out_block=None):
        if self._desc is None:
            self._make_desc()
        ob = (out_block or self._out_block) if self._dev_out is not None else None
        mode = self._sync_mode if sync else 0
        if mode == 2 and self._async_pending:
            mode, self._async_pending = 1, False
        rc = self.lib.pthip_plan_replay4(self._desc_ref, ob.ptr if ob is not None else None, self._done_ptr, mode)
        if rc:
            ffi.check(rc)
        if mode:
            self._async_pending = False  # this replay waited for the stream: nothing of an earlier launch_async is in flight

    # ------------------------------------------------------------------
    def _build_fast(self):
        """The replay path as one native call (csrc/fastplan.c) when the plan is of the simple, common kind:
        results written straight into the pinned block by the plan's kernels, no update feedback, no baked
        scalars, every resident watched by a write-protection slot (or not at all).  ``None`` otherwise."""
        if _FASTPLAN is None or self._fast_off or not self.fetch_outputs or self._dev_out is not None or self._fed or self._baked:
            return None
        if self._out_block is None or self._ring is not None or self._out_meta is None:
            return None
        if self._desc is None:
            self._make_desc()
        exe, ob, lib = self.exe, self._out_block, self.lib
        res = []
        for pos in self._resident_devs:
            ent = exe._resident_cache.get(pos)
            if ent is None or ent.key is None or ent.host is None:
                return None
            tok = ent.fp
            if tok is None:
                slot = -1
            elif isinstance(tok, coherence._Guard) and tok.slot >= 0:
                slot = tok.slot
            else:
                self._fast_off = True  # (do not ask again on every call)
                return None  # a content hash is checked by the Python path (overlapped with the replay)
            res.append((pos, ent.host, slot))
        staged = [(pos, self._in_block.ptr + self._in_block.offsets[k], self._in_block.views[k].dtype, tuple(self._in_block.views[k].shape))
                  for pos, k in self._in_view.items()]
        outs, k = [], 0
        scalars = set(exe._scalar_outs)
        for q, meta in enumerate(self._out_meta):
            if meta is not None:
                outs.append((None, np.array(meta, order="C", copy=True), meta.dtype, tuple(meta.shape), int(q in scalars)))  # (ascontiguousarray would make a 0-d value 1-d)
            else:
                shape, dtype = self._out_specs[k]
                outs.append((ob.ptr + ob.offsets[k], None, np.dtype(dtype), tuple(shape), int(q in scalars)))
                k += 1
        addr = lambda f: C.cast(f, C.c_void_p).value
        try:
            return _FASTPLAN.FastPlan(len(self._sig), staged, res, outs, addr(lib.pthip_plan_replay4), addr(lib.pthip_guard_clean), C.addressof(self._desc),
                                      self._done_ptr or 0, ob.ptr + ob.offsets[-1], self._sync_mode, C.addressof(exe._res_gen), exe._res_gen.value)
        except Exception:  # noqa: BLE001 (an exotic dtype or rank: the Python path serves the plan)
            self._fast_off = True
            return None

    def launch_async(self):
        """Enqueue one replay (parameters already in the staging block); no host sync."""
        self._async_pending = True  # its completion word may still arrive: the next waiting call uses the stream
        self._replay(False)

    def __call__(self, *inputs):
        lib = self.lib
        fast = self._fast
        if fast is not None and not self._async_pending:
            r = fast(inputs)
            if r.__class__ is tuple:
                return r if r else None
            if r is None:
                # not the captured situation (signature, a dirty resident, a non-contiguous argument): the full
                # checks below decide what happens; the native path is rebuilt after a successful call
                self._fast = None
                self._fast_misses += 1
                if self._fast_misses > 8:
                    self._fast_off = True
            elif r < 0:
                ffi.check(-r)
            else:
                from pytensor_amd.executor import raise_device_status

                st = C.c_int(0)
                ffi.check(lib.pthip_check_status(C.byref(st)))  # clears the device word
                self._out_block.views[-1][0] = 0
                raise_device_status(int(r))
        if len(inputs) != len(self._sig):
            raise TypeError(f"expected {len(self._sig)} inputs, got {len(inputs)}")
        views = self._in_block.views
        for pos, k in self._in_view.items():
            v = views[k]
            a = inputs[pos]
            if getattr(a, "shape", None) != v.shape or getattr(a, "dtype", None) != v.dtype:
                a = np.asarray(a)
                if a.shape != v.shape or a.dtype != v.dtype:
                    raise SignatureChanged(f"frozen plan: input {pos} changed signature {self._sig[pos]} -> {(a.shape, str(a.dtype))}")
            v[...] = a
        exe = self.exe
        late = []  # residents still held by the object that was uploaded, watched by a content hash: check deferred
        for pos, dev in self._resident_devs.items():
            v = inputs[pos]
            ent = exe._resident_cache[pos]
            if v is ent.host and ent.key is not None:
                tok = ent.fp
                if tok is None:
                    continue
                if tok.late:
                    late.append((pos, ent, v))
                elif not tok.clean(v):
                    # write-protected pages saw a store (coherence._Guard): upload before launching
                    ent.key = None
                    exe._refresh_resident(pos, v)
            elif exe._refresh_resident(pos, v) is not dev:
                raise SignatureChanged(f"frozen plan: resident input {pos} changed shape or dtype; re-freeze")
        for pos, b in self._baked.items():
            if not np.array_equal(np.asarray(inputs[pos]), b):
                raise SignatureChanged(f"frozen plan: scalar input {pos} is baked into the plan and changed")
        ring = self._ring
        mine = ring.take() if ring is not None else None  # the pinned block this call's results land in
        ob = mine or self._out_block
        if not late:
            self._replay(True, mine)  # one native call: H2D, graphs, D2H, stream synchronisation
        else:
            # launch first, hash the resident host arrays while the GPU works (a borrowed shared
            # value edited in place, pytensor_amd/coherence.py): the check costs the call nothing
            self._replay(False, mine)
            dirty = [pos for pos, ent, v in late if not ent.fp.clean(v)]
            if dirty:
                # the replay in flight read stale data: discard it, upload what changed (and put
                # back every update-fed resident, which that replay advanced), run again
                ffi.check(lib.pthip_synchronize())
                for pos in set(dirty) | {p for p, _ in self._fed}:
                    exe._resident_cache[pos].key = None
                    exe._refresh_resident(pos, inputs[pos])
                self._replay(False, mine)
            ffi.check(lib.pthip_synchronize())
        if self.fetch_outputs and ob is not None and ob.views[-1][0]:
            from pytensor_amd.executor import raise_device_status

            word = int(ob.views[-1][0])
            st = C.c_int(0)
            ffi.check(lib.pthip_check_status(C.byref(st)))  # clears the device word
            ob.views[-1][0] = 0
            if mine is not None:
                ring.free.append(mine)
            # a call that raises commits no update (compile/executor.py:712-716 is not reached), but the
            # replay has already advanced the update-fed residents: the next call uploads the storage
            # cells' (old) values again
            for p, _ in self._fed:
                self.exe._resident_cache[p].key = None
            raise_device_status(word)
        fresh = ring.hand_out(mine) if mine is not None else None
        res = []
        k = 0
        for meta in self._out_meta:
            if meta is not None:
                res.append(meta.copy())
            else:
                res.append(fresh[k] if fresh is not None else ob.views[k].copy())
                k += 1
        if self._fed:
            self.exe._feed_updates_host(self._fed, res)
        for k in self.exe._scalar_outs:
            res[k] = res[k][()]
        if not res and not self._out_meta:
            return None  # a function without outputs returns None (link/basic.py:690-699)
        if self._fast is None and not self._fast_off:
            self._fast = self._build_fast()
            if self._fast is None and not late:
                self._fast_off = True  # not a plan of the simple kind: do not ask again
        return tuple(res)

    def close(self):
        """Release the graphs, the arena and the staging blocks.  Idempotent.  When called
        (e.g. by the garbage collector through ``__del__``) while ANOTHER plan is being
        captured, the release is deferred: a stream synchronisation would invalidate the
        running capture."""
        if self._closed:
            return
        if _CAPTURE_ACTIVE[0] and _CAPTURE_ACTIVE[0] is not self:
            _DEFERRED.append(self)  # keeps the object alive until it is safe to free
            return
        self._closed = True
        self._fast = None
        self._fast_off = True
        lib = self.lib
        try:
            lib.pthip_synchronize()
            self._release_segments()
            self._keep.clear()
            if self._arena:
                lib.pthip_arena_destroy(self._arena)
                self._arena = C.c_void_p()
            self._dev_in = None
            self._dev_out = None
            if self._ring is not None:
                self._ring.close()
            for blk in (self._in_block, self._out_block):
                if blk is not None:
                    blk.free()
        except Exception:
            pass

    def _release_segments(self):
        for kind, h in self._graphs:
            if h:
                (self.lib.pthip_graph_destroy if kind == "graph" else self.lib.pthip_list_destroy)(h)
        self._graphs = []

    def __del__(self):
        self.close()
