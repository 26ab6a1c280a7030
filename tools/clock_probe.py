"""Sample the shader clock (rocm-smi) while a kernel loop runs — is a GEMM's distance from the
datasheet peak the kernel or the clock?   usage: python tools/clock_probe.py [f64|f32]"""
import ctypes as C
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytensor_amd import ffi  # noqa: E402
from pytensor_amd.device import DeviceArray  # noqa: E402


def main(kind="f64"):
    ffi.init(0)
    lib = ffi.lib()
    dtype = "float64" if kind == "f64" else "float32"
    n = 4096
    A, B, out = (DeviceArray.empty((n, n), dtype) for _ in range(3))
    for x in (A, B):
        ffi.check(lib.pthip_memset(x.ptr, 0, x.nbytes))
    dt = ffi.np_dtype_code(dtype)
    stop = []
    samples = []

    def poll():
        while not stop:
            try:
                o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
                samples.append([l.strip() for l in o.splitlines() if "sclk" in l or "Power" in l or "fclk" in l][:3])
            except Exception as e:  # noqa: BLE001
                samples.append([repr(e)])
            time.sleep(0.3)

    th = threading.Thread(target=poll)
    th.start()
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 4.0:
        for _ in range(50):
            ffi.check(lib.pthip_gemm(dt, 1, n, n, n, 1.0, A.ptr, 0, n, 1, B.ptr, 0, n, 1, 0.0, None, 0, 0, 0, out.ptr))
        ffi.check(lib.pthip_synchronize())
        reps += 50
    el = time.perf_counter() - t0
    stop.append(1)
    th.join()
    print("TFLOP/s sustained:", round(2.0 * n**3 * reps / el / 1e12, 1))
    for s in samples[:: max(1, len(samples) // 6)]:
        print(s)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "f64")
