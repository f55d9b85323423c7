"""Time the native first-stage decoder on the two reference workloads (16-frame 256^2 clip, one 768^2 keyframe), dump
per-launch timings, and time the fp32 PyTorch restatement (oracle, on the same GPU) beside it for context.
Usage (GPU box): python tools/vae_profile.py gpurun_out/vae_ops_clip.csv gpurun_out/vae_ops_key.csv"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import gpu_random_state_dict  # noqa: E402
from neurons_amd.vae import NativeVAEDecoder, VAEDecoderConfig, vae_decoder_state_dict_schema  # noqa: E402
from oracle import vae_oracle as V  # noqa: E402

dev = torch.device("cuda", 0)
cfg = VAEDecoderConfig()
sd = gpu_random_state_dict(vae_decoder_state_dict_schema(cfg), 3, dev)
dec = NativeVAEDecoder(cfg).to(dev)
dec.load_state_dict({k: v.cpu() for k, v in sd.items()})


def timed(fn, n=5):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, z, path in (("clip 16 x 32x32 -> 256^2", torch.randn(16, 4, 32, 32, device=dev), sys.argv[1]),
                      ("keyframe 1 x 96x96 -> 768^2", torch.randn(1, 4, 96, 96, device=dev), sys.argv[2])):
    ms = timed(lambda: dec.decode(z, z_scale=1 / 0.18215, unit_range=True))
    os.environ["NR_PROFILE_CSV"] = path
    for _ in range(2):
        p = dec.profile_last()
    tot = sum(v["ms"] for v in p.values())
    fl = sum(v["flops"] for v in p.values())
    print(f"{name}: native {ms:.2f} ms/decode (sum of launches {tot:.2f} ms, {fl / 1e12:.2f} TFLOP -> {fl / tot / 1e9:.0f} TF/s)",
          {k: round(v["ms"], 3) for k, v in p.items()}, f"arena {dec.workspace_bytes() / 2**20:.0f} MiB")
    with torch.no_grad():
        if z.shape[0] == 16:      # the reference decodes frame by frame (pipeline_animation.py:249-250)
            ref = timed(lambda: [V.decode(sd, z[i:i + 1] / 0.18215, 4, 2) for i in range(16)], n=2)
        else:
            ref = timed(lambda: V.decode(sd, z / 0.18215, 4, 2), n=2)
    print(f"{name}: fp32 PyTorch restatement on the same GPU {ref:.1f} ms -> native is {ref / ms:.1f}x")

# encoder: 17 images of 256^2 (16 blurry frames + control image, scripts/neuroclips_video.py:263-283)
from neurons_amd.vae import NativeVAEEncoder, vae_encoder_state_dict_schema  # noqa: E402
esd = gpu_random_state_dict(vae_encoder_state_dict_schema(cfg), 4, dev)
enc = NativeVAEEncoder(cfg).to(dev)
enc.load_state_dict({k: v.cpu() for k, v in esd.items()})
x = torch.rand(16, 3, 256, 256, device=dev)
ms = timed(lambda: enc.moments(x, 2.0, -1.0))
if len(sys.argv) > 3:
    os.environ["NR_PROFILE_CSV"] = sys.argv[3]
for _ in range(2):
    p = enc.profile_last()
tot = sum(v["ms"] for v in p.values())
fl = sum(v["flops"] for v in p.values())
print(f"encode 16 x 256^2: native {ms:.2f} ms (sum of launches {tot:.2f} ms, {fl / 1e12:.2f} TFLOP -> {fl / tot / 1e9:.0f} TF/s)",
      {k: round(v["ms"], 3) for k, v in p.items()})
with torch.no_grad():
    ref = timed(lambda: V.encode_moments(esd, 2 * x - 1, 4, 2), n=2)
print(f"encode 16 x 256^2: fp32 PyTorch restatement on the same GPU {ref:.1f} ms -> native is {ref / ms:.1f}x")
