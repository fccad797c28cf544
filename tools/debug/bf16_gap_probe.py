"""Developer probe: engine bf16 error vs the bf16-storage oracle gap, per tensor, for the AutoencoderKL / discriminator cases."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
import gpu_util as G
from param_gen import gen_param, normal, eeg_windows
from oracle import aekl as A, losses as Ls, quant as Q
from eegldm.models import AutoencoderKL, PatchDiscriminator

CASES = {"c32_32_64": [32, 32, 64], "c2_2_4": [2, 2, 4], "c4_16_32": [4, 16, 32], "c4_4_16": [4, 4, 16], "c8_8_16": [8, 8, 16]}
for name, nc in CASES.items():
    cfg = dict(num_channels=nc, latent_channels=1, in_channels=1, out_channels=1, num_res_blocks=2, norm_num_groups=1)
    B, L = 2, 256
    shapes = A.aekl_param_shapes(cfg)
    sd0 = {k: torch.from_numpy(gen_param(11, k, s)) for k, s in shapes.items()}
    x0 = torch.from_numpy(eeg_windows(B, seed=5, length=L, pad=8))
    eps = torch.from_numpy(normal((B, 1, L // 4), seed=6)); dy = None
    def run(emul):
        global dy
        sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
        x = x0.clone().requires_grad_(True)
        with Q.bf16_storage(emul):
            recon, mu, sg = A.forward(sd, cfg, x, eps)
            if dy is None:
                dy = torch.from_numpy(normal(tuple(recon.shape), seed=7))
            ((recon * dy).sum() + 0.3 * Ls.kl_loss(mu, sg)).backward()
        return recon.detach(), x.grad, {k: v.grad for k, v in sd.items()}
    r32, dx32, g32 = run(False); rq, dxq, gq = run(True)
    net = AutoencoderKL(spatial_dims=1, attention_levels=[False] * 3, dtype="bfloat16", **cfg); net.load_state_dict(sd0)
    r, mu, sg = net(x0, eps=eps); net.zero_grad(); dx = net.backward(dy, kl_weight=0.3, need_dx=True)
    gap = G.grads_rel_errors(gq, g32, 3e-2); err = G.grads_rel_errors(net.grad_dict(), g32, 3e-2)
    print(f"{name}: recon {G.rel_l2(r, r32):.2e} (gap {G.rel_l2(rq, r32):.2e}) dx {G.rel_l2(dx, dx32):.2e} (gap {G.rel_l2(dxq, dx32):.2e})")
    for k in sorted(err, key=lambda k: -err[k] / max(gap[k], 2 ** -7))[:6]:
        print(f"    {k:45s} n={g32[k].numel():6d} engine {err[k]:.2e} gap {gap[k]:.2e} ratio {err[k]/max(gap[k],1e-9):.1f}")
