"""TEST INFRASTRUCTURE (oracle-type caller restatement; never imported by neurons_amd/ or by bench.py's timed region).
Caller with the call sequence of the reference's ``utils.unclip_recon`` (utils.py:302-350), written against the
ATTRIBUTES of a ``diffusion_engine`` object only — exactly what the reference function touches, in its order:

    diffusion_engine.ema_scope()                                         (:307)
    diffusion_engine.sampler.discretization(diffusion_engine.sampler.num_steps)   (:324)
    denoiser(x, sigma, c) -> diffusion_engine.denoiser(diffusion_engine.model, x, sigma, c)   (:337-338, a closure)
    diffusion_engine.sampler(denoiser, noised_z, cond=c, uc=uc)          (:340)
    diffusion_engine.decode_first_stage(samples_z)                       (:343)

The reference draws four random tensors inside (z :308, uc tokens :318, noise :323, offset :328-330), partly on the CPU and partly on
``device``; they cannot be replayed from a seed on another device, so this caller takes them as ``draws`` (the fixture
tests/golden/unclip_tiny.npz stores the ones the reference's own run made).  Used with an oracle-backed engine on the CPU
(test_sgm_oracle_golden.py) and with ``neurons_amd.sgm.NativeDiffusionEngine`` on the GPU (test_sgm_gpu.py)."""
import torch


def call_like_unclip_recon(x, diffusion_engine, vector_suffix, draws, num_samples=1, offset_noise_level=0.04, device="cpu"):
    assert x.ndim == 3
    if x.shape[0] == 1:
        x = x[[0]]
    with torch.no_grad(), diffusion_engine.ema_scope():
        z = draws["z"].to(device)
        tokens = x
        c = {"crossattn": tokens.repeat(num_samples, 1, 1).to(z.device), "vector": vector_suffix.repeat(num_samples, 1).to(z.device)}
        tokens = draws["uc_tokens"].to(x.device)
        uc = {"crossattn": tokens.repeat(num_samples, 1, 1).to(z.device), "vector": vector_suffix.repeat(num_samples, 1).to(z.device)}
        for k in c:
            c[k], uc[k] = map(lambda y: y[k][:num_samples].to(device), (c, uc))
        noise = draws["noise"].to(z.device)
        sigmas = diffusion_engine.sampler.discretization(diffusion_engine.sampler.num_steps)
        sigma = sigmas[0].to(z.device)
        if offset_noise_level > 0.0:
            off = draws["offset"].to(z.device)
            noise = noise + offset_noise_level * off.reshape(-1, *([1] * (z.ndim - 1)))
        noised_z = z + noise * sigma.reshape(*([1] * z.ndim))
        noised_z = noised_z / torch.sqrt(1.0 + sigmas[0] ** 2.0).to(z.device)

        def denoiser(x, sigma, c):
            return diffusion_engine.denoiser(diffusion_engine.model, x, sigma, c)

        samples_z = diffusion_engine.sampler(denoiser, noised_z, cond=c, uc=uc)
        samples_x = diffusion_engine.decode_first_stage(samples_z)
        return torch.clamp((samples_x * .8 + .2), min=0.0, max=1.0)
