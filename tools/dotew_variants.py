"""Timing decomposition of the Scan step kernels (config #5's generated ``dotew_*``): the GRU scan at
T steps for each PTHIP_DOTEW_VAR value, one child process per variant (the variant is baked into
the generated kernel's name and source).  Values other than "" / "acc2" give WRONG results — they
exist to price the parts of the kernel (operand streaming, MFMA chain, launch + epilogue floor).

usage: python tools/dotew_variants.py [T] [variant ...]      (on the MI355X box)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, json
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tools")
import bench_configs as bc
from pytensor_amd import configs, ffi
ffi.init(0)
T = {T}
v = configs.c5_inputs(T=T, B=64, H=1024)
td, tw = bc.run_case("c5_gru", v, 5, check=False, kernel_reps=1)
k = {{n: round(ms * 1e3, 2) for n, ms in sorted(bc.KERNELS["c5_gru"].items(), key=lambda t: -t[1])[:4]}}
print("RESULT " + json.dumps({{"us_per_step": td / T * 1e3, "kernels_us_eager": k}}))
"""


def main():
    args = sys.argv[1:]
    T = int(args[0]) if args and args[0].isdigit() else 256
    variants = [a for a in args if not a.isdigit()] or ["none", "acc2", "nomfma", "noload", "noload,nomfma", "apacked", "acc2,apacked"]
    for var in variants:
        env = dict(os.environ, PTHIP_DOTEW_VAR="" if var == "none" else var)
        r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, T=T)], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        print(f"{var:16s}", line[0][7:] if line else ("FAILED: " + r.stderr[-400:]), flush=True)


if __name__ == "__main__":
    main()
