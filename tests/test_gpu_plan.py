"""GPU: frozen hipGraph plans (single- and two-stream) reproduce the eager results bit for bit."""
import numpy as np
import pytest

from util import golden_cases, load_case

pytestmark = pytest.mark.gpu

# graphs that draw random numbers run eagerly: a replay would repeat the captured Philox counters
FREEZABLE = [c for c in golden_cases() if not c.startswith("random_")]


@pytest.fixture(scope="module")
def hip():
    from pytensor_amd import ffi

    if ffi.device_count() <= 0:
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    ffi.init(0)
    return ffi


@pytest.mark.parametrize("name", FREEZABLE)
@pytest.mark.parametrize("multi", [False, True])
def test_frozen_plan_equals_eager(hip, name, multi):
    from pytensor_amd.executor import HipExecutable

    g, ins, cvm, py, meta = load_case(name)
    exe = HipExecutable(g)
    want = exe(*ins)
    try:
        plan = exe.freeze(*ins, multi_stream=multi)
    except hip.HipError as e:
        if "data-dependent host read" in str(e):
            pytest.skip("graph reads device data on the host: not freezable")
        raise
    for rep in range(3):
        got = plan(*ins)
        for k, (a, b) in enumerate(zip(got, want)):
            np.testing.assert_array_equal(a, b, err_msg=f"{name} out{k} replay {rep}")
    plan.close()


def test_random_graphs_refuse_to_freeze(hip):
    from pytensor_amd.executor import HipExecutable

    g, ins, cvm, py, meta = load_case("random_uniform_philox")
    exe = HipExecutable(g, auto_freeze=True)
    assert exe.has_rng and not exe.auto_freeze
    with pytest.raises(NotImplementedError):
        exe.freeze(*ins)


def test_plan_new_parameters_same_signature(hip):
    """replays pick up new parameter values (the staging block is rewritten per call)"""
    import np_graph
    from pytensor_amd import configs
    from pytensor_amd.executor import HipExecutable
    from util import assert_parity

    g, ins, cvm, py, meta = load_case("c4_hier_small")
    names = meta["input_names"]
    resident = [k for k, n in enumerate(names) if n in configs.C4_DATA]
    exe = HipExecutable(g, resident=resident)
    exe(*ins)
    plan = exe.freeze(*ins)
    for chain in range(3):
        v = configs.c4_inputs(N=257, K=16, G=8, chain=chain)
        cur = [ins[k] if k in resident else np.asarray(v[n]) for k, n in enumerate(names)]
        got = plan(*cur)
        ref = np_graph.run_graph(g, cur)
        for k, (a, b) in enumerate(zip(got, ref)):
            assert_parity(a, b, 1e-12, f"chain {chain} out{k}")
    plan.close()


def test_bench_multi_rank_flow_on_one_gpu(hip, tmp_path):
    """The N>1 launch contract (torch.distributed.run, barrier, max-over-ranks) end to end,
    with two ranks sharing GPU 0 over gloo (RCCL refuses two ranks on one device).  The bench is
    started WITHOUT a launcher: ``python bench.py --gpus 2`` starts its own two ranks under
    ``torch.distributed.run`` (``replicas.ensure_world``), so both entry shapes are exercised."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PTHIP_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--rows", "20000"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"] is None


def test_coexists_with_torch_hip_runtime(hip):
    """bench.py's multi-rank path imports torch (which bundles its own libamdhip64):
    our library must keep working next to it."""
    import torch

    from pytensor_amd.executor import HipExecutable

    g, ins, cvm, py, meta = load_case("c1_gauss")
    a = HipExecutable(g)(*ins)
    if torch.cuda.is_available():
        t = torch.ones(1024, device="cuda").sum().item()
        assert t == 1024.0
    b = HipExecutable(g)(*ins)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


def test_auto_freeze_is_what_the_linker_path_gets(hip):
    """``HipLinker.jit_compile`` builds ``HipExecutable(..., auto_freeze=True)``: call 1 eager,
    call 2 captures, later calls replay; a new signature falls back to eager and re-arms; a
    graph that reads device data on the host stays eager."""
    from pytensor_amd.executor import HipExecutable

    g, ins, cvm, py, meta = load_case("c4_hier_small")
    names = [g.vars[v].name for v in g.inputs]
    resident = [k for k, n in enumerate(names) if n in ("y", "X", "gidx", "Sigma")]
    exe = HipExecutable(g, resident=resident, auto_freeze=True)
    first = exe(*ins)
    assert exe._auto_plan is None
    second = exe(*ins)
    assert exe._auto_plan is not None
    third = exe(*ins)
    for a, b, c, ref in zip(first, second, third, cvm):
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(a, c)
        np.testing.assert_allclose(a, ref, rtol=1e-12)  # north_star, element-wise
    # new parameter values, same signature: still the plan, new results
    ins2 = [a if k in resident else (a * 1.01 if a.dtype.kind == "f" else a) for k, a in enumerate(ins)]
    plan = exe._auto_plan
    got = exe(*ins2)
    assert exe._auto_plan is plan
    want = HipExecutable(g, resident=resident)(*ins2)
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a, b)
    # a resident array replaced by a new object of the same geometry (shared.set_value): copied
    # into the SAME device buffer, so the captured plan stays valid (round 2; it used to drop
    # back to eager and capture again)
    ups = exe.stats["resident_uploads"]
    ins3 = list(ins)
    ins3[resident[0]] = ins[resident[0]].copy()
    got3 = exe(*ins3)
    assert exe._auto_plan is plan and exe.stats["resident_uploads"] == ups + 1
    for a, b in zip(got3, first):
        np.testing.assert_array_equal(a, b)
    # new CONTENT (in-place edit of the borrowed array): seen, re-uploaded, still the plan
    ins3[resident[0]][...] = ins3[resident[0]] * 2.0
    got4 = exe(*ins3)
    assert exe._auto_plan is plan and exe.stats["resident_uploads"] == ups + 2
    want4 = HipExecutable(g, resident=resident)(*ins3)
    for a, b in zip(got4, want4):
        np.testing.assert_array_equal(a, b)
    assert not np.array_equal(got4[0], got3[0])
    # a new SHAPE is a new signature: eager, then captured again
    n_old = ins[resident[0]].shape[0]
    ins5 = [a[: n_old - 7] if (k in resident and a.shape[:1] == (n_old,)) else a for k, a in enumerate(ins)]
    exe(*ins5)
    assert exe._auto_plan is None
    exe(*ins5)
    assert exe._auto_plan is not None and exe._auto_plan is not plan

    # not freezable: stays eager, still correct
    g, ins, cvm, py, meta = load_case("softmax_shapes")
    exe = HipExecutable(g, auto_freeze=True)
    for _ in range(3):
        out = exe(*ins)
    assert exe._auto_plan is None and exe._auto_failed
    from util import assert_parity

    for k, (a, ref) in enumerate(zip(out, cvm)):
        assert_parity(a, ref, None, f"softmax_shapes out{k}", case="softmax_shapes", k=k, py=py[k])


def test_auto_multi_stream_picks_a_plan_and_matches_eager(hip):
    """freeze(multi_stream="auto") times the one- and the two-stream capture and keeps one;
    either way the replay is bit-identical to the eager run."""
    from pytensor_amd.executor import HipExecutable

    for name in ("c4_hier_small", "c1_gauss"):
        g, ins, cvm, py, meta = load_case(name)
        exe = HipExecutable(g)
        want = exe(*ins)
        plan = exe.freeze(*ins, multi_stream="auto")
        for a, b in zip(plan(*ins), want):
            np.testing.assert_array_equal(a, b)
        if exe.segments is None:
            assert not plan.segmented
        plan.close()


def test_arena_destroy_adopts_blocks_that_are_still_referenced(hip):
    """A block handed out inside a plan's arena may outlive the plan (arrays in frames that a
    stored exception traceback keeps alive).  Destroying the arena must not release it: its owner
    would free a dangling pointer later, and the pool would hand a live block out twice."""
    import ctypes as C

    from pytensor_amd.device import Buffer

    lib = hip.lib()
    arena = C.c_void_p()
    hip.check(lib.pthip_arena_begin(C.byref(arena)))
    held = Buffer(4096)  # allocated inside the arena, outlives it
    hip.check(lib.pthip_arena_end())
    hip.check(lib.pthip_arena_destroy(arena))
    others = [Buffer(4096) for _ in range(64)]
    assert held.ptr not in {b.ptr for b in others}
    del held  # the late free returns a block the pool knows about
    more = [Buffer(4096) for _ in range(64)]
    ptrs = [b.ptr for b in others + more]
    assert len(set(ptrs)) == len(ptrs)


def test_small_segments_replay_as_launch_lists(hip, monkeypatch):
    """Plan segments of a few launches are recorded launch lists (direct launches at replay), long
    ones captured hipGraphs; both forms give the eager path's bits."""
    import pytensor_amd.plan as plan_mod
    from pytensor_amd.executor import HipExecutable

    g, ins, cvm, py, meta = load_case("c1_gauss")
    exe = HipExecutable(g)
    want = exe(*ins)
    p = exe.freeze(*ins)
    assert [k for k, _ in p._graphs] == ["list"] and p._seg_sizes[0] <= plan_mod._LIST_MAX
    for _ in range(3):
        for a, b in zip(p(*ins), want):
            np.testing.assert_array_equal(a, b)
    p.close()
    # the same graph captured as a hipGraph
    monkeypatch.setattr(plan_mod, "_LIST_MAX", 0)
    q = exe.freeze(*ins)
    assert [k for k, _ in q._graphs] == ["graph"]
    for a, b in zip(q(*ins), want):
        np.testing.assert_array_equal(a, b)
    q.close()
    monkeypatch.undo()
    # a Scan unrolled into thousands of launches stays a hipGraph
    g, ins, cvm, py, meta = load_case("c5_gru")
    exe = HipExecutable(g)
    want = exe(*ins)
    p = exe.freeze(*ins)
    assert "graph" in [k for k, _ in p._graphs]
    for a, b in zip(p(*ins), want):
        np.testing.assert_array_equal(a, b)
    p.close()
    # the segmented two-stream plan of config #4: three lists
    g, ins, cvm, py, meta = load_case("c4_hier")
    names = meta["input_names"]
    exe = HipExecutable(g, resident=[k for k, n in enumerate(names) if n in ("y", "X", "gidx", "Sigma")])
    want = exe(*ins)
    p = exe.freeze(*ins)
    assert p.segmented and [k for k, _ in p._graphs] == ["list", "list", "list"]
    for a, b in zip(p(*ins), want):
        np.testing.assert_array_equal(a, b)
    p.close()


def test_segmented_plan_joins_its_streams_on_the_device(hip, monkeypatch):
    """The two streams of config #4's plan meet through a signal word the Tail node's launches wait for
    (``pthip_join_signal`` at the end of segment A, descriptor flag bit 1), not through an event: the same bits
    as the eager path over many replays with changing parameters, in all three segment forms (launch lists,
    hipGraphs, mixed), and the event form stays available (PTHIP_PLAN_DEVICE_JOIN=0)."""
    import pytensor_amd.plan as plan_mod
    from pytensor_amd.executor import HipExecutable

    g, ins, cvm, py, meta = load_case("c4_hier")
    names = meta["input_names"]
    res = [k for k, n in enumerate(names) if n in ("y", "X", "gidx", "Sigma")]
    exe = HipExecutable(g, resident=res)
    rng = np.random.default_rng(5)

    def perturbed(j):
        out = list(ins)
        for k, a in enumerate(ins):
            if k not in res and isinstance(a, np.ndarray) and a.dtype.kind == "f":
                out[k] = a + 1e-3 * j * rng.standard_normal(a.shape)
        return out

    for list_max, join in ((None, True), (0, True), (2, True), (None, False)):  # lists; hipGraphs; A a hipGraph, B and C lists; events
        if list_max is not None:
            monkeypatch.setattr(plan_mod, "_LIST_MAX", list_max)
        monkeypatch.setattr(plan_mod, "_DEV_JOIN", join)
        p = exe.freeze(*ins)
        assert p.segmented
        assert p._join_used == join and bool(p._join_word) == join
        for j in range(6):
            args = perturbed(j)
            want = exe(*args)
            for a, b in zip(p(*args), want):
                np.testing.assert_array_equal(a, b)
        assert bool(p._desc.flags & 2) == join
        p.close()
        monkeypatch.undo()


def test_device_join_probe_and_many_replays(hip, monkeypatch):
    """What the device-side join assumes (readers that waited for the word find the other stream's in-place results
    with plain loads, csrc/tail_device.h) is probed on the device once per process (``pthip_join_probe``; a failing
    probe keeps the event between the streams) — and held against the event form over many replays whose segment-A
    results change on every call."""
    import ctypes as C

    import pytensor_amd.plan as plan_mod
    from pytensor_amd import ffi
    from pytensor_amd.executor import HipExecutable

    bad = C.c_int(-1)
    ffi.check(ffi.lib().pthip_join_probe(256, C.byref(bad)))
    assert bad.value == 0, f"device-side join probe: {bad.value}"
    assert plan_mod._device_join_ok(ffi.lib())
    # a failing probe switches the join off for new plans
    g, ins, cvm, py, meta = load_case("c4_hier")
    names = meta["input_names"]
    res = [k for k, n in enumerate(names) if n in ("y", "X", "gidx", "Sigma")]
    exe = HipExecutable(g, resident=res)
    monkeypatch.setattr(plan_mod, "_JOIN_PROBED", [False])
    p_ev = exe.freeze(*ins)
    assert p_ev.segmented and not p_ev._join_word
    monkeypatch.undo()
    p_dev = exe.freeze(*ins)
    assert p_dev._join_word
    rng = np.random.default_rng(11)
    for j in range(300):
        args = list(ins)
        for k, a in enumerate(ins):
            if k not in res and isinstance(a, np.ndarray) and a.dtype.kind == "f":
                args[k] = a + 1e-2 * rng.standard_normal(a.shape)
        for a, b in zip(p_dev(*args), p_ev(*args)):
            np.testing.assert_array_equal(a, b)
    p_dev.close()
    p_ev.close()


def test_device_join_that_gives_up_is_retried_by_the_host(hip, monkeypatch):
    """When the latency-chain segment is still running 1 ms after the tail kernel started waiting (here: two 4096^3
    fp64 products recorded in front of the signal; in the field: a profiler that serialises kernels), the kernel
    stores 2 into the done word and leaves; ``pthip_plan_replay4`` waits for the chain's stream and runs the closing
    segment again.  After three such calls the descriptor goes back to the event.  Results: the eager path's bits."""
    import ctypes as C

    import pytensor_amd.plan as plan_mod
    from pytensor_amd import ffi
    from pytensor_amd.device import DeviceArray
    from pytensor_amd.executor import HipExecutable

    lib = ffi.lib()
    n = 4096
    A = DeviceArray.from_host(np.full((n, n), 1e-3))
    out = DeviceArray.empty((n, n), "float64")
    orig = plan_mod.FrozenPlan._segment_boundary

    def slow_chain(self, prev, nxt):
        if prev == 0 and self._join_word:
            for _ in range(2):
                ffi.check(lib.pthip_gemm(ffi.np_dtype_code("float64"), 1, n, n, n, 1.0, A.ptr, 0, n, 1, A.ptr, 0, n, 1, 0.0, None, 0, 0, 0, out.ptr))
        orig(self, prev, nxt)

    monkeypatch.setattr(plan_mod.FrozenPlan, "_segment_boundary", slow_chain)
    g, ins, cvm, py, meta = load_case("c4_hier")
    names = meta["input_names"]
    exe = HipExecutable(g, resident=[k for k, nm in enumerate(names) if nm in ("y", "X", "gidx", "Sigma")])
    want = exe(*ins)
    p = exe.freeze(*ins)
    assert p.segmented and p._join_used
    for _ in range(6):  # three retried calls, then the event
        for a, b in zip(p(*ins), want):
            np.testing.assert_array_equal(a, b)
    assert p._desc.flags & 2
    p.close()


def test_large_results_are_handed_out_without_a_copy_and_stay_valid(hip):
    """Results above the zero-copy pack limit land in a pinned block of the call's own (plan.py
    ``_ResultRing``) and are returned as views on it: every call's arrays are distinct and keep
    their values while later calls run (link/basic.py:670-684: thunks return fresh arrays), also
    when the caller holds more results than the ring has blocks, and the blocks come back."""
    import gc

    from pytensor_amd.executor import HipExecutable
    from pytensor_amd.ir import Graph

    n = 50_000
    g = Graph(name="big_out")
    x = g.new_var("float64", (None,), name="x")
    s = g.new_var("float64", (None,), name="s")
    y, z = g.new_var("float64", (None,)), g.new_var("float64", (None,))
    body = lambda op: {"in_dtypes": ["float64", "float64"], "out_dtypes": ["float64"],
                       "body": [{"op": op, "in": [["i", 0], ["i", 1]], "dtype": "float64"}], "outs": [["t", 0]]}
    g.add_node("Elemwise", {"scalar": body("Mul")}, [x, s], [y])
    g.add_node("Elemwise", {"scalar": body("Add")}, [x, s], [z])
    g.inputs, g.outputs = [x, s], [y, z]
    xv = np.random.default_rng(0).normal(size=n)
    exe = HipExecutable(g, resident=[0])
    full = lambda v: np.full(n, float(v))
    exe(xv, full(1))
    plan = exe.freeze(xv, full(1))
    assert plan._ring is not None
    held = [plan(xv, full(k)) for k in range(7)]  # more than the ring holds
    assert plan._ring.made == plan._ring.limit and not plan._ring.free
    for k, (a, b) in enumerate(held):
        assert a.flags.writeable and np.array_equal(a, xv * k) and np.array_equal(b, xv + k)
    ptrs = {a.ctypes.data for a, _ in held}
    assert len(ptrs) == 7
    del held, a, b
    gc.collect()
    assert len(plan._ring.free) == plan._ring.limit  # every block came back
    first = plan(xv, full(2))
    again = plan(xv, full(3))
    assert np.array_equal(first[0], xv * 2.0) and np.array_equal(again[0], xv * 3.0)
    keep = first[1]
    del first, again
    plan.close()  # the array still held keeps its block; it is released with the array
    assert np.array_equal(keep, xv + 2.0)
    del keep
    gc.collect()


@pytest.mark.parametrize("mode", ["trust", "guard"])
def test_native_path_sees_invalidate_resident_and_foreign_uploads(hip, mode):
    """ADVICE r4 (csrc/fastplan.c): the native call path of a plan vouches for a resident only while the executable's
    resident generation is the one it was built at.  ``invalidate_resident()`` — the documented way to force a
    re-upload in mode ``trust``, where no guard slot exists — and an upload into the shared device buffer by ANOTHER
    plan of the same executable must both send the next call through the Python path, which uploads the new value."""
    import np_graph
    from pytensor_amd import coherence, configs
    from pytensor_amd.executor import HipExecutable
    from util import assert_parity

    coherence.set_mode(mode)
    try:
        g, ins, cvm, py, meta = load_case("c4_hier")
        names = meta["input_names"]
        resident = [k for k, n in enumerate(names) if n in configs.C4_DATA]
        # every resident above 64 KiB: in mode "guard" smaller ones are watched by a content hash, which the Python
        # path checks — such a plan never gets the native path
        v = configs.c4_inputs(N=9001, K=96, G=8)
        ins = [np.array(v[n]) for n in names]  # private, writable copies
        exe = HipExecutable(g, resident=resident)
        exe(*ins)
        plan = exe.freeze(*ins)
        for _ in range(3):
            first = plan(*ins)
        assert plan._fast is not None, "the native path was not built: this test would not exercise it"
        ypos = names.index("y")
        if mode == "trust":
            ins[ypos][...] = ins[ypos] * 0.5 + 1.0  # in place: nothing watches the array in this mode
            exe.invalidate_resident()
        else:
            ins[ypos][...] = ins[ypos] * 0.5 + 1.0  # the write-protected pages fault: the guard slot turns dirty
        got = plan(*ins)
        ref = np_graph.run_graph(g, ins)
        for k, (a, b) in enumerate(zip(got, ref)):
            assert_parity(a, b, 1e-12, f"after the in-place change ({mode}) out{k}")
        assert not np.array_equal(got[0], first[0])
        # a second plan of the same executable uploads a NEW array object into the shared device buffer ...
        other = [a for a in ins]
        other[ypos] = ins[ypos] + 2.0
        plan2 = exe.freeze(*other)
        plan2(*other)
        # ... the first plan, called again with ITS array (unchanged since its last call), must notice
        for _ in range(2):
            got = plan(*ins)
        for k, (a, b) in enumerate(zip(got, ref)):
            assert_parity(a, b, 1e-12, f"after another plan's upload ({mode}) out{k}")
        plan.close()
        plan2.close()
    finally:
        coherence.set_mode(None)
