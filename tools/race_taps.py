"""Localise the run-to-run nondeterminism of the overlapped step: activation taps of both networks after repeated
forward_with_controlnet calls on identical inputs; prints the first taps that differ from the first run."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from neurons_amd import _lib, NativeSparseCtrl, NativeUNet3D  # noqa: E402
from neurons_amd.unet3d import random_state_dict  # noqa: E402
from tiny_configs import tiny_ctrl_config, tiny_unet_config  # noqa: E402

lib = _lib.load()


def taps(net):
    out = {}
    n = lib.nr_net_num_taps(net._h)
    for i in range(n):
        name = lib.nr_net_tap_name(net._h, i).decode()
        buf = np.empty(1 << 22, dtype=np.float32)
        rows, cc = C.c_int32(), C.c_int32()
        _lib.check(lib.nr_net_read_tap(net._h, i, buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(rows), C.byref(cc)))
        out[f"{i:03d}:{name}"] = buf[:rows.value * cc.value].copy()
    return out


g = np.load(os.path.join(ROOT, "tests", "golden", "tiny_networks.npz"))
ucfg, ccfg = tiny_unet_config(), tiny_ctrl_config()
unet = NativeUNet3D(ucfg).to("cuda")
unet.load_state_dict(random_state_dict(ucfg, _lib.NR_KIND_UNET3D, seed=11))
ctrl = NativeSparseCtrl(ccfg).to("cuda")
ctrl.load_state_dict(random_state_dict(ccfg, _lib.NR_KIND_SPARSECTRL, seed=12))
_lib.check(lib.nr_net_set_debug(unet._handle(), 1))
_lib.check(lib.nr_net_set_debug(ctrl._handle(), 1))
sample, ctx = torch.from_numpy(g["sample"]).cuda(), torch.from_numpy(g["ctx"]).cuda()
cond, mask = torch.from_numpy(g["cond"]).cuda(), torch.from_numpy(g["mask"]).cuda()
ref = None
for r in range(12):
    out = unet.forward_with_controlnet(ctrl, sample, int(g["t"]), ctx, cond, mask, 1.0).sample
    torch.cuda.synchronize()
    tu, tc = taps(unet), taps(ctrl)
    cur = {**{"U:" + k: v for k, v in tu.items()}, **{"C:" + k: v for k, v in tc.items()}, "out": out.cpu().numpy().ravel()}
    if ref is None:
        ref = cur
        print("taps:", len(tu), "+", len(tc))
        continue
    bad = [k for k in ref if not np.array_equal(ref[k], cur[k])]
    def mag(k):
        d = np.abs(ref[k] - cur[k])
        return f"{k.split(':', 1)[-1] if k != 'out' else k}[{(d > 0).mean() * 100:.1f}% max {d.max():.2e}]"
    bu = [k for k in bad if k.startswith("U:")]
    bc = [k for k in bad if k.startswith("C:")]
    print(f"run {r}: U {len(bu)} C {len(bc)} | U first: {[mag(k) for k in bu[:4]]} | C first: {[mag(k) for k in bc[:4]]}", flush=True)
    for k in (bu[:1] + bc[:1]):
        idx = np.nonzero(ref[k] != cur[k])[0]
        Cc = 64 if ref[k].size % 64 == 0 else 1
        print("   ", k, "n =", idx.size, "elements (row, ch, ref, got):", [(int(i // Cc), int(i % Cc), float(ref[k][i]), float(cur[k][i])) for i in idx[:12]])
