"""Deterministic synthetic tensors (there are no checkpoints or datasets offline).

Values come from numpy's Philox bit generator (a counter-based generator whose raw stream numpy guarantees
stable) through an explicit Box-Muller transform, keyed by (seed, tensor name) — so the same weights and inputs
are regenerated bit-for-bit on the GPU box, in the fixture generator that imports the reference, and in the
CPU tests, independent of torch's RNG implementation and of generation order.
"""
import zlib

import numpy as np
import torch


def randn(name: str, shape, seed: int = 0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    key = (int(seed) << 32) ^ zlib.crc32(name.encode())
    bg = np.random.Philox(key=key)
    m = (n + 1) // 2
    raw = bg.random_raw(2 * m).astype(np.uint64)
    u = ((raw >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)   # (0,1), 53 bits
    r = np.sqrt(-2.0 * np.log(u[:m]))
    th = 2.0 * np.pi * u[m:]
    z = np.concatenate([r * np.cos(th), r * np.sin(th)])[:n].astype(np.float32)
    return torch.from_numpy(z.reshape(tuple(shape)))
