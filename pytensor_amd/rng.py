"""RNG values at the boundary of the hip linker.

The reference passes ``numpy.random.Generator`` objects through the graph
(``RandomGeneratorType``, pytensor/tensor/random/type.py); each ``RandomVariable`` node returns
the advanced generator as its first output (random/op.py).  The device samplers
(csrc/random.hip) are counter-based, so inside the executor an RNG value is a :class:`RngState`
(Philox key + 256-bit counter); at the boundary it is a real ``Generator(Philox)``:

* a ``Generator`` over ``numpy.random.Philox`` with an empty buffer is taken over as it is — its
  first uniform draws on the device are the numbers ``Generator.random`` returns on the host;
* any other bit generator (the default PCG64, ...) is re-keyed: its state is hashed with
  ``SeedSequence`` into a Philox key, counter 0 (the JAX linker does the analogous copy,
  link/jax/dispatch/random.py:58-80).  The streams differ from the reference's either way, so
  the draws are distributionally — not numerically — the reference's.
"""

from __future__ import annotations

import numpy as np

_MASK64 = (1 << 64) - 1


class RngState:
    __slots__ = ("key", "counter")

    def __init__(self, key, counter: int):
        self.key = (int(key[0]) & _MASK64, int(key[1]) & _MASK64)
        self.counter = int(counter) & ((1 << 256) - 1)

    def advanced(self, blocks: int) -> "RngState":
        return RngState(self.key, self.counter + int(blocks))

    def key_words(self):
        return np.array(self.key, dtype=np.uint64)

    def counter_words(self):
        return np.array([(self.counter >> (64 * j)) & _MASK64 for j in range(4)], dtype=np.uint64)

    # ---- boundary ----
    @classmethod
    def from_generator(cls, gen) -> "RngState":
        if isinstance(gen, RngState):
            return gen
        if isinstance(gen, dict):  # a state dict is a valid RandomGeneratorType value too
            st = gen
        else:
            st = gen.bit_generator.state
        if st["bit_generator"] == "Philox":
            ctr = 0
            for j, w in enumerate(np.asarray(st["state"]["counter"], dtype=np.uint64)):
                ctr |= int(w) << (64 * j)
            # a partly consumed 4-word buffer is dropped: every node starts on a block boundary
            return cls([int(w) for w in np.asarray(st["state"]["key"], dtype=np.uint64)], ctr)
        words = []

        def flatten(o):
            if isinstance(o, dict):
                for k in sorted(o):
                    flatten(o[k])
            elif isinstance(o, np.ndarray):
                words.extend(int(x) for x in o.ravel())
            elif isinstance(o, (int, np.integer)):
                words.append(int(o))
            elif isinstance(o, str):
                words.extend(o.encode())

        flatten(st)
        ent = []
        for wv in words:  # SeedSequence takes non-negative ints of any size
            ent.append(wv if wv >= 0 else -wv)
        key = np.random.SeedSequence(ent).generate_state(2, np.uint64)
        return cls([int(key[0]), int(key[1])], 0)

    def to_generator(self):
        bg = np.random.Philox(key=self.key_words(), counter=self.counter_words())
        return np.random.Generator(bg)

    def __repr__(self):
        return f"RngState(key={self.key}, counter={self.counter})"
