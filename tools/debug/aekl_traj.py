"""Developer tool: the AEKL [32,32,64] + PatchDiscriminator GAN trajectory on the GPU engines with the SAME initial weights, windows and
posterior noise as a CPU-oracle run (oracle.steps.aekl_train_step, B = 8, lr 1e-3 / 5e-4, reference loss weights): losses every 5 steps."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from param_gen import gen_param, eeg_windows, normal
import eegldm
from eegldm.models import AutoencoderKL, PatchDiscriminator
from eegldm.training import Adam, aekl_train_step
init = sys.argv[1] if len(sys.argv) > 1 else "gen_param"
for dtype in ("float32", "bfloat16"):
    ae = AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=[32, 32, 64], latent_channels=1, num_res_blocks=2,
                       norm_num_groups=1, attention_levels=[False, False, False], dtype=dtype, device=0)
    disc = PatchDiscriminator(spatial_dims=1, num_layers_d=3, num_channels=64, in_channels=1, out_channels=1, kernel_size=3,
                              norm="BATCH", bias=False, padding=1, dtype=dtype, device=0)
    if init == "gen_param":
        ae.load_state_dict({k: torch.from_numpy(gen_param(42, k, tuple(v.shape))) for k, v in ae.state_dict().items()})
        dsd = disc.state_dict()
        disc.load_state_dict({k: (torch.from_numpy(gen_param(43, k, tuple(v.shape))) if v.dtype.is_floating_point and "running" not in k and "num_batches" not in k else v) for k, v in dsd.items()})
    og, od = Adam(ae, lr=1e-3), Adam(disc, lr=5e-4)
    xs = torch.from_numpy(eeg_windows(64, seed=777)).cuda()
    lo = torch.zeros(6, device="cuda")
    for i in range(1, 41):
        s = ((i - 1) * 8) % 64
        ew = torch.from_numpy(normal((8, 1, 768), seed=100 + i)).cuda()
        ae.zero_grad(); disc.zero_grad()
        aekl_train_step(ae, disc, xs[s:s + 8], ew, 0.01, 1e-9, 1e4, True, losses_out=lo)
        og.step(); od.step()
        if i % 5 == 0 or i == 1:
            v = [round(float(t), 5) for t in lo.cpu()]
            print(dtype, init, i, dict(zip(["recons", "spectral", "kl", "gen", "d_fake", "d_real"], v)), flush=True)
    del ae, disc
