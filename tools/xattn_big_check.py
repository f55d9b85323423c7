import sys, torch
sys.path.insert(0, '/root/repo')
from neurons_amd import ops
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(0)
C = 320
for (nimg, hw, ipc) in [(64, 1024, 16), (256, 1024, 16), (1024, 1024, 16), (128, 1024, 2)]:
    nctx = (nimg + ipc - 1) // ipc
    t = torch.randn(nimg * hw, C, generator=g, device=dev).to(torch.bfloat16)
    kv = torch.randn(nctx * 77, 2 * C, generator=g, device=dev).to(torch.bfloat16)
    wq, wo = (torch.randn(C, C, generator=g, device=dev) * C ** -0.5 for _ in range(2))
    o = ops.xattn_fused(t, nimg, hw, ipc, torch.ones(C, device=dev), torch.zeros(C, device=dev), wq, wo, torch.zeros(C, device=dev), kv, 77)
    torch.cuda.synchronize()
    print(nimg, hw, ipc, bool(torch.isfinite(o.float()).all()), flush=True)
