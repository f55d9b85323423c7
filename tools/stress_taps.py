"""Layer-by-layer localisation at FULL width: the engine's activation taps (nr_net_set_debug) against the fp32 oracle's, one U-Net
evaluation at the headline shape (CFG batch 2, 16 frames, 32x32 latent), on un-stressed and on stress-L1 weights (neurons_amd.synth.stress_state_dict).
Prints rel-L2 per tap (block outputs in execution order) so that an op that loses precision in the heavy-tailed regime stands out as a jump.
Usage (GPU box): python tools/stress_taps.py [none L1 ...]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neurons_amd import _lib, NativeUNet3D  # noqa: E402
from neurons_amd.synth import STRESS_LEVELS, gpu_random_state_dict, stress_state_dict  # noqa: E402
from neurons_amd.unet3d import UNet3DConfig, state_dict_schema  # noqa: E402
from oracle import animatediff_oracle as O  # noqa: E402


def rel(a, b):
    return ((a.float() - b.float()).pow(2).mean().sqrt() / (b.float().pow(2).mean().sqrt() + 1e-20)).item()


def main():
    dev = torch.device("cuda", 0)
    cfg = UNet3DConfig()
    oc = O.OracleConfig.from_native(cfg)
    g = torch.Generator(device=dev).manual_seed(0)
    sample = torch.randn(2, 4, 16, 32, 32, generator=g, device=dev)
    ctx = torch.randn(2, 77, cfg.cross_attention_dim, generator=g, device=dev)
    lib = _lib.load()
    table = {}
    for level in (sys.argv[1:] or ["none", "L1"]):
        sd = gpu_random_state_dict(state_dict_schema(cfg, _lib.NR_KIND_UNET3D), 1, dev)
        if level != "none":
            stress_state_dict(sd, 7, **STRESS_LEVELS[level])
        net = NativeUNet3D(cfg).to(dev)
        net.load_state_dict({k: v.cpu() for k, v in sd.items()})
        _lib.check(lib.nr_net_set_debug(net._handle(), 1))
        got_eps = net(sample, 481, encoder_hidden_states=ctx).sample
        taps = {}
        with torch.no_grad():
            want = O.unet3d_forward(sd, oc, sample, 481, ctx, taps=taps)
        n = lib.nr_net_num_taps(net._h)
        rows = []
        for i in range(n):
            name = lib.nr_net_tap_name(net._h, i).decode()
            ref = taps[name]
            b, c, f, h, w = ref.shape
            buf = np.empty(b * f * h * w * c, dtype=np.float32)
            r_, c_ = C.c_int32(), C.c_int32()
            _lib.check(lib.nr_net_read_tap(net._h, i, buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(r_), C.byref(c_)))
            got = torch.from_numpy(buf).reshape(b, f, h, w, c).permute(0, 4, 1, 2, 3).to(dev)
            amax = ref.abs().max().item() / (ref.pow(2).mean().sqrt().item() + 1e-20)
            rows.append((name, rel(got, ref), amax))
            del got
        rows.append(("eps (network output)", rel(got_eps, want), want.abs().max().item() / want.pow(2).mean().sqrt().item()))
        table[level] = rows
        del net, sd, taps
        torch.cuda.empty_cache()
    levels = list(table)
    print(f"{'tap (block output, execution order)':58s} " + " ".join(f"{lv + ' rel-L2':>14s} {'max/rms':>8s}" for lv in levels))
    for i in range(len(table[levels[0]])):
        print(f"{table[levels[0]][i][0]:58s} " + " ".join(f"{table[lv][i][1]:14.3e} {table[lv][i][2]:8.1f}" for lv in levels))
    for lv in levels:
        r = [x[1] for x in table[lv]]
        jumps = sorted(((r[i] / max(r[i - 1], 1e-9), table[lv][i][0]) for i in range(1, len(r))), reverse=True)[:3]
        print(f"[{lv}] worst tap {max(r):.3e}; largest tap-to-tap growth: " + "; ".join(f"x{j:.2f} at {nm}" for j, nm in jumps))


if __name__ == "__main__":
    main()
