"""Developer probe: engine-bf16 error vs bf16-storage-oracle gap for the AutoencoderKL, RMS over several input seeds (is a large
single-draw ratio on a 4-element tensor luck or a systematic precision loss?)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
import gpu_util as G
from param_gen import gen_param, normal, eeg_windows
from oracle import aekl as A, losses as Ls, quant as Q
from eegldm.models import AutoencoderKL

for name, nc in {"c4_4_16": [4, 4, 16], "c4_16_32": [4, 16, 32], "c2_2_4": [2, 2, 4]}.items():
    cfg = dict(num_channels=nc, latent_channels=1, in_channels=1, out_channels=1, num_res_blocks=2, norm_num_groups=1)
    B, L = 2, 256
    shapes = A.aekl_param_shapes(cfg)
    sd0 = {k: torch.from_numpy(gen_param(11, k, s)) for k, s in shapes.items()}
    net = AutoencoderKL(spatial_dims=1, attention_levels=[False] * 3, dtype="bfloat16", **cfg); net.load_state_dict(sd0)
    E, Gp = {k: 0.0 for k in shapes}, {k: 0.0 for k in shapes}
    NS = 8
    for s in range(NS):
        x0 = torch.from_numpy(eeg_windows(B, seed=100 + s, length=L, pad=8))
        eps = torch.from_numpy(normal((B, 1, L // 4), seed=200 + s)); dy = torch.from_numpy(normal((B, 1, L), seed=300 + s))
        def run(emul):
            sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
            with Q.bf16_storage(emul):
                recon, mu, sg = A.forward(sd, cfg, x0, eps)
                ((recon * dy).sum() + 0.3 * Ls.kl_loss(mu, sg)).backward()
            return {k: v.grad for k, v in sd.items()}
        g32, gq = run(False), run(True)
        net(x0, eps=eps); net.zero_grad(); net.backward(dy, kl_weight=0.3)
        gap = G.grads_rel_errors(gq, g32, 3e-2); err = G.grads_rel_errors(net.grad_dict(), g32, 3e-2)
        for k in shapes:
            E[k] += err[k] ** 2 / NS; Gp[k] += gap[k] ** 2 / NS
    print(name)
    for k in sorted(shapes, key=lambda k: -(E[k] / max(Gp[k], 1e-12)))[:8]:
        print(f"    {k:45s} rms engine {E[k] ** 0.5:.2e} rms gap {Gp[k] ** 0.5:.2e} ratio {(E[k] / max(Gp[k], 1e-18)) ** 0.5:.2f}")
