import sys, time, torch
sys.path.insert(0, '/root/repo')
from neurons_amd.clip import CLIPTextConfig, NativeCLIPTextModel, clip_state_dict_schema
from oracle import clip_oracle as CO
cfg = CLIPTextConfig(); dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(1)
sd = {k: (torch.randn(s, generator=g, device=dev) * (0.02 if len(s) == 1 else s[-1] ** -0.5)) for k, s in clip_state_dict_schema(cfg).items()}
for k in sd:
    if 'norm' in k and k.endswith('weight'): sd[k] = sd[k] + 1
enc = NativeCLIPTextModel(cfg).to(dev); enc.load_state_dict({k: v.cpu() for k, v in sd.items()})
ids = torch.randint(0, 49408, (2, 77), generator=g, device=dev)
def timed(fn, n=20):
    fn(); fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
a = timed(lambda: enc(ids))
with torch.no_grad():
    b = timed(lambda: CO.clip_text_forward(sd, ids, 12, 12))
p = enc.profile_last()
print(f"CLIP text encode (2 x 77 tokens): native {a:.3f} ms, fp32 PyTorch restatement {b:.3f} ms", {k: round(v['ms'], 3) for k, v in p.items()})
