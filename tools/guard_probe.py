"""GPU-box probe for csrc/guard.hip: (1) is an H2D copy FROM write-protected pageable pages fine
(the runtime may pin them)?  (2) can pinned (hipHostMalloc) pages be write-protected on the CPU
side while the device keeps writing into them (update-fed shared values land in pinned result
blocks)?  Each probe runs in a child process: a failure must not take the caller down."""
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P1 = r"""
import sys, ctypes as C, numpy as np, mmap
sys.path.insert(0, %r)
from pytensor_amd import ffi, coherence
from pytensor_amd.device import DeviceArray
ffi.init(0)
lib = ffi.lib()
hip = C.CDLL("libamdhip64.so")
n = 1 << 22
a = np.random.default_rng(0).normal(size=n)
d = DeviceArray.empty((n,), a.dtype)
tok = coherence.watch(a)
assert tok.clean(a)
# raw hipMemcpy from the protected pages (bypasses pthip_h2d's own opening of the slot)
rc = hip.hipMemcpy(C.c_void_p(d.ptr), C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), 1)
print("hipMemcpy H2D from PROT_READ pages rc =", rc, "still clean:", tok.clean(a))
back = d.to_host(sync=True)
print("round trip equal:", bool(np.array_equal(back, a)))
rc = hip.hipMemcpyAsync(C.c_void_p(d.ptr), C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), 1, C.c_void_p(lib.pthip_stream()))
lib.pthip_synchronize()
print("hipMemcpyAsync rc =", rc, "still clean:", tok.clean(a))
a[12345] = 1.0
print("after a store: clean =", tok.clean(a))
"""
P2 = r"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, %r)
from pytensor_amd import ffi, coherence
from pytensor_amd.device import DeviceArray
ffi.init(0)
lib = ffi.lib()
nb = 8 << 20
hp = C.c_void_p()
ffi.check(lib.pthip_host_alloc(nb, C.byref(hp)))
h = np.frombuffer((C.c_char * nb).from_address(hp.value), dtype=np.float64)
h[:] = 1.0
src = DeviceArray.from_host(np.full(nb // 8, 3.0))
tok = coherence.watch(h)
print("pinned block watched with", type(tok).__name__, "clean:", tok.clean(h))
ffi.check(lib.pthip_d2h(hp.value, src.ptr, nb)); lib.pthip_synchronize()
print("device wrote into the protected pinned block: h[5] =", h[5], "clean (a device write is not a CPU store):", tok.clean(h))
h[777] = 9.0
print("CPU store into it: clean =", tok.clean(h), "value", h[777])
tok.release()
ffi.check(lib.pthip_d2h(hp.value, src.ptr, nb)); lib.pthip_synchronize()
print("after release the block still works:", h[777])
"""
for name, src in (("guard_on_pinned", P2), ("h2d_from_protected", P1)):
    try:
        r = subprocess.run([sys.executable, "-u", "-c", src % ROOT], capture_output=True, text=True, timeout=40)
        print(f"== {name}: rc={r.returncode}\n{r.stdout}{r.stderr[-600:]}", flush=True)
    except subprocess.TimeoutExpired as e:
        print(f"== {name}: TIMED OUT after 40 s\n{(e.stdout or b'').decode(errors='replace') if isinstance(e.stdout, bytes) else e.stdout}", flush=True)
