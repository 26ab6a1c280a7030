import sys, os, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import e2e_util as E
pytensor = E.activate()
import pytensor.tensor as ptt
rng = np.random.default_rng(11)
Xv, yv = rng.normal(size=(4096, 64)), rng.normal(size=4096)
X, y = pytensor.shared(Xv, name="X"), pytensor.shared(yv, name="y")
w = pytensor.shared(rng.normal(size=64) * 0.1, name="w")
lr = ptt.dscalar("lr")
loss = ((ptt.dot(X, w) - y) ** 2).mean()
g = pytensor.grad(loss, w)
f = pytensor.function([lr], loss, updates={w: w - lr * g}, mode="hip")
exe = E.hip_executable(f)
print("update_map", exe.update_map, "resident", exe.resident)
print(f(0.05))
ins = [c.storage[0] for c in f.input_storage]
print([type(i) for i in ins])
try:
    p = exe.freeze(*[np.asarray(0.05), *[c.storage[0] for c in f.input_storage[1:]]], multi_stream="auto")
    print("freeze ok", p)
except Exception:
    traceback.print_exc()
for _ in range(3):
    print(f(0.05), exe.stats, exe._auto_failed)
