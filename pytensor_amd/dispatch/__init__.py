"""Node handlers: IR op name → ``handler(node, inputs, env) -> outputs``.

Mirrors the per-backend ``dispatch`` packages of the reference
(pytensor/link/pytorch/dispatch/, pytensor/link/jax/dispatch/), one module per Op
family.  Handlers receive/return ``DeviceArray`` (HBM) or ``HostValue``
(shape arithmetic) values.
"""

HANDLERS = {}


def handler(*names):
    def deco(f):
        for n in names:
            HANDLERS[n] = f
        return f

    return deco


from pytensor_amd.dispatch import basic, blas, decomp, dotew, elemwise, extra, fft, fused, linalg, lu, misc, random, scan, subtensor, tail, wide  # noqa: E402,F401
