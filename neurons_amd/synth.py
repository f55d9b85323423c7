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


def gpu_random_state_dict(schema, seed: int, device) -> dict:
    """Seeded weights generated on the GPU with torch's generator (fast for the 1.3-2.5 G-parameter full-size networks;
    NOT bit-reproducible across torch versions, so never used for committed fixtures).  Same scaling rules as
    ``unet3d.random_state_dict``: Linear/conv ~ N(0, 1/fan_in), norm gains ~ 1 + 0.1 N, biases ~ 0.02 N (norm biases 0.1 N)."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, shape in schema.items():
        z = torch.randn(tuple(shape), generator=g, device=device, dtype=torch.float32)
        if name.endswith(".bias"):
            is_norm = ".norm" in name or "norms." in name or "conv_norm_out" in name or "ff_norm" in name
            z = (0.1 if is_norm else 0.02) * z
        elif len(shape) == 1:
            z = 1.0 + 0.1 * z
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            z = z / (fan_in ** 0.5)
        sd[name] = z
    return sd
