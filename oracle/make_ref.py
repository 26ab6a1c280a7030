"""Build ``oracle/_ref/`` — a Python-3.10-importable copy of the reference.

TEST INFRASTRUCTURE.  The reference (``/root/reference``, requires-python >=3.12,
pyproject.toml:13) does not import under this image's Python 3.10.12.  This
recipe copies ``pytensor/`` and the reference's own ``tests/`` into ``oracle/_ref/``
(git-ignored: no reference source enters the history; NOT gpurun-ignored since round 2, so the
built copy travels to the GPU box like any other built artefact and the drop-in can be
exercised end to end there — ``tests/test_gpu_e2e.py``, ``tests/test_gpu_refsuite.py``, the
``cpu_baseline`` leg of ``bench.py``) and applies the three mechanical shims from SURVEY.md
§8c / Appendix A:

1. PEP 695/646 syntax in 11 files (``type X = ...``, ``def f[T](``, ``*tuple[...]``);
2. hashable ``slice`` in ``MetaType``'s generated ``__hash__`` (graph/utils.py:217-219);
3. ``BaseException.add_note`` polyfill (link/utils.py:322, compile/executor.py:564).

Nothing here is product code; no reference source enters the git history.

Usage:  python oracle/make_ref.py   (idempotent; prints the PYTHONPATH to use)
"""

from __future__ import annotations

import ast
import os
import re
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("PYTENSOR_REFERENCE", "/root/reference")
DST = os.path.join(HERE, "_ref")
COMPILEDIR = os.environ.get("PTHIP_REF_COMPILEDIR", "/tmp/pthip_ref_compiledir")

PEP695_FILES = [
    "gradient.py",
    "graph/traversal.py",
    "graph/basic.py",
    "graph/utils.py",
    "graph/rewriting/unify.py",
    "scalar/basic.py",
    "link/numba/dispatch/linalg/solvers/lu_solve.py",
    "tensor/reshape.py",
    "tensor/subtensor.py",
    "tensor/random/variable.py",
    "xtensor/random/variable.py",
]

HASH_OLD = """                def __hash__(self):
                    return hash((type(self), tuple(getattr(self, a) for a in props)))
"""
HASH_NEW = """                def __hash__(self):
                    def _h(v):
                        if isinstance(v, slice):
                            return ("__slice__", _h(v.start), _h(v.stop), _h(v.step))
                        if isinstance(v, (tuple, list)):
                            return tuple(_h(x) for x in v)
                        return v

                    return hash((type(self), tuple(_h(getattr(self, a)) for a in props)))
"""


def build(force: bool = False) -> str:
    pkg = os.path.join(DST, "pytensor")
    stamp = os.path.join(DST, ".built")
    if os.path.exists(stamp) and not force:
        return DST
    if not os.path.isdir(os.path.join(REF, "pytensor")):
        raise FileNotFoundError(f"reference not found at {REF}")
    if os.path.exists(DST):
        shutil.rmtree(DST)
    os.makedirs(DST)
    shutil.copytree(os.path.join(REF, "pytensor"), pkg)
    os.system(f"chmod -R u+w {DST}")
    for f in PEP695_FILES:
        p = os.path.join(pkg, f)
        s = open(p).read()
        s = re.sub(r"^type (\w+) = ", r"\1 = ", s, flags=re.M)
        s = re.sub(r"^(\s*def \w+)\[[^\]]*\]\(", r"\1(", s, flags=re.M)
        s = s.replace('tuple[Optional["Op"], *tuple["Variable", ...]]', "tuple")
        open(p, "w").write(s)
    p = os.path.join(pkg, "graph/utils.py")
    s = open(p).read()
    assert HASH_OLD in s, "reference changed: MetaType.__hash__ shim does not apply"
    open(p, "w").write(s.replace(HASH_OLD, HASH_NEW))
    for f, old in [("link/utils.py", "exc_value.add_note("), ("compile/executor.py", "e.add_note(")]:
        p = os.path.join(pkg, f)
        s = open(p).read()
        assert old in s
        var = old.split(".")[0]
        open(p, "w").write(s.replace(old, f'getattr({var}, "add_note", lambda *_a: None)('))
    # the reference's own test-suite (subclassed under mode="hip" by tests/test_gpu_refsuite.py)
    tdst = os.path.join(DST, "tests")
    shutil.copytree(os.path.join(REF, "tests"), tdst)
    os.system(f"chmod -R u+w {tdst}")
    for root, _, files in os.walk(tdst):
        for f in files:
            if f.endswith(".py"):
                p = os.path.join(root, f)
                s = open(p).read()
                s2 = re.sub(r"^type (\w+) = ", r"\1 = ", s, flags=re.M)
                s2 = re.sub(r"^(\s*def \w+)\[[^\]]*\]\(", r"\1(", s2, flags=re.M)
                if s2 != s:
                    open(p, "w").write(s2)
    bad = []
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                try:
                    ast.parse(open(os.path.join(root, f)).read())
                except SyntaxError as e:  # pragma: no cover
                    bad.append((os.path.join(root, f), e))
    if bad:
        raise RuntimeError(f"shim incomplete: {bad}")
    open(stamp, "w").write("ok\n")
    return DST


def activate() -> None:
    """Make the shimmed reference importable in this process (C linker, cvm)."""
    dst = build()
    if dst not in sys.path:
        sys.path.insert(0, dst)
    # The reference C linker's compile cache lives OUTSIDE the repo tree: its hundreds of per-Op
    # thunk modules are neither product nor portable (the directory name carries the kernel
    # release and the key the host's -march flags, so a cache built here misses on the GPU box
    # anyway), and in-tree they crowd the product's one library out of the driver's record of
    # loaded native code.
    os.environ.setdefault("PYTENSOR_FLAGS", f"base_compiledir={COMPILEDIR},linker=cvm")


def available() -> bool:
    """The pristine reference is present (build container): fixtures can be regenerated."""
    return os.path.isdir(os.path.join(REF, "pytensor"))


def importable() -> bool:
    """A built copy exists (build container, or the GPU box where ``oracle/_ref`` travelled)."""
    return os.path.exists(os.path.join(DST, ".built")) or available()


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
