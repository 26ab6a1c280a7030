import sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'oracle')
import e2e_util
pytensor = e2e_util.activate()
import pytensor.tensor as pt
x = pt.dvector("x"); y = pt.dvector("y")
f = pytensor.function([x, y], pt.pow(x, y), mode="hip")
xs, ys = np.meshgrid(np.arange(-20, 21, dtype=np.float64), np.arange(0, 16, dtype=np.float64))
xs, ys = xs.ravel(), ys.ravel()
got = f(xs, ys)
want = np.array([float(int(a) ** int(b)) for a, b in zip(xs, ys)])
bad = got != want
print("float64 pow, integer-valued operands: inexact", int(bad.sum()), "of", bad.size)
for a, b, g, w in list(zip(xs[bad], ys[bad], got[bad], want[bad]))[:12]:
    print("  ", a, b, repr(g), w)
xi = pt.lvector("xi"); yi = pt.lvector("yi")
fi = pytensor.function([xi, yi], pt.pow(xi, yi), mode="hip")
gi = fi(xs.astype(np.int64), ys.astype(np.int64))
wi = np.array([int(a) ** int(b) for a, b in zip(xs, ys)], dtype=object)
ok = np.array([int(g) == int(w) for g, w in zip(gi, wi) if abs(int(w)) < 2**53])
print("int64 pow: wrong", int((~ok).sum()), "of", ok.size)
xs2 = np.random.default_rng(0).uniform(0.1, 10, 200000); ys2 = np.random.default_rng(1).uniform(-5, 5, 200000)
g2 = f(xs2, ys2); w2 = np.power(xs2.astype(np.longdouble), ys2.astype(np.longdouble))
ulp = np.abs(np.nextafter(w2.astype(np.float64), np.inf) - w2.astype(np.float64))
print("random pow: max ulp err", float((np.abs(g2.astype(np.longdouble) - w2).astype(np.float64) / ulp).max()))
print("pow(x,1)==x:", bool(np.all(f(xs2, np.ones_like(xs2)) == xs2)), " pow(x,2)==x*x:", bool(np.all(f(xs2, 2*np.ones_like(xs2)) == xs2*xs2)), " pow(x,0.5)==sqrt:", bool(np.all(f(xs2, 0.5*np.ones_like(xs2)) == np.sqrt(xs2))))
k = np.arange(-60, 61, dtype=np.float64)
for nm, fn, arg, want in [("log2(2^k)", pt.log2, 2.0 ** k, k), ("log10(10^k)", pt.log10, 10.0 ** np.arange(0, 23), np.arange(0, 23, dtype=np.float64)),
                          ("exp2(k)", pt.exp2, k, 2.0 ** k), ("sqrt(n^2)", pt.sqrt, np.arange(0, 2000, dtype=np.float64) ** 2, np.arange(0, 2000, dtype=np.float64)),
                          ("log(1),exp(0),log1p(0),expm1(0)", lambda v: pt.stack([pt.log(v[0:1] + 1), pt.exp(v[0:1]), pt.log1p(v[0:1]), pt.expm1(v[0:1])]).ravel(), np.zeros(1), np.array([0.0, 1.0, 0.0, 0.0]))]:
    g = pytensor.function([x], fn(x), mode="hip")(arg)
    print(nm, "inexact:", int((g != want).sum()), "of", want.size, [(float(a), float(b)) for a, b in zip(g[g != want][:3], want[g != want][:3])])
f32 = pytensor.function([pt.fvector("a"), pt.fvector("b")], pt.pow(pt.fvector("a"), pt.fvector("b")), mode="hip", on_unused_input="ignore") if False else None
a32 = pt.fvector("a32"); b32 = pt.fvector("b32")
g32 = pytensor.function([a32, b32], pt.pow(a32, b32), mode="hip")(xs.astype(np.float32), ys.astype(np.float32))
w32 = np.array([float(int(a) ** int(b)) for a, b in zip(xs, ys)]).astype(np.float32)
print("float32 pow integer-valued: inexact", int((g32 != w32).sum()), "of", w32.size)
