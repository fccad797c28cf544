"""Developer tool: are the fused train steps bit-reproducible?  Runs the LDM step (config_ldm UNet) and the AEKL/GAN step ([2,2,4] autoencoder +
PatchDiscriminator + spectral loss) twice from identical state and inputs and reports which outputs differ.
   python tools/debug/det_check_steps.py [float32|bfloat16] [B_ldm] [B_aekl]         (EEGLDM_DETERMINISTIC=1 must give zeros everywhere)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import eegldm
from eegldm.models import UNetModel, AutoencoderKL, PatchDiscriminator
from eegldm.schedulers import DDPMScheduler
from eegldm.training import Adam, aekl_train_step, ldm_train_step, randn
dtype = sys.argv[1] if len(sys.argv) > 1 else "float32"
BL = int(sys.argv[2]) if len(sys.argv) > 2 else 64
BA = int(sys.argv[3]) if len(sys.argv) > 3 else 64
ctx = eegldm.default_context(0)


def differing(a, b, entries):
    return [k for k, (o, n, _s) in entries.items() if not torch.equal(a[o:o + n], b[o:o + n])]


# ---- LDM step
L = 768
net = UNetModel(image_size=L, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4], channel_mult=[1, 2, 4], resblock_updown=True, dtype=dtype)
g = torch.Generator().manual_seed(0); sd = net.state_dict()
sd0 = {k: (torch.randn(v.shape, generator=g) * 0.02 if float(v.abs().sum()) == 0 else v.cpu().clone()) for k, v in sd.items()}
sched = DDPMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195)
lat = torch.randn(BL, 1, L, generator=g).cuda(); nz = torch.randn(BL, 1, L, generator=g).cuda(); t = torch.randint(0, 1000, (BL,), generator=g).cuda()


def ldm_run(steps=3):
    net.load_state_dict(sd0); opt = Adam(net, lr=1e-4); losses = []
    for _ in range(steps):
        net.zero_grad(); losses.append(float(ldm_train_step(net, sched, lat, nz, t))); opt.step()
    torch.cuda.synchronize()
    return losses, net.flat_grad.clone(), net.flat.clone()


l1, g1, p1 = ldm_run(); l2, g2, p2 = ldm_run()
dg = differing(g1, g2, net.entries)
print(f"LDM step {dtype} B={BL}: losses equal {l1 == l2}; gradients differing in {len(dg)} of {len(net.entries)} parameters {dg[:8]}; parameters after 3 steps equal {bool(torch.equal(p1, p2))}", flush=True)

# ---- AEKL / GAN step
LA = 3072
ae = AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=[2, 2, 4], latent_channels=1, num_res_blocks=2, norm_num_groups=1,
                   attention_levels=[False, False, False], dtype=dtype, device=0)
disc = PatchDiscriminator(spatial_dims=1, num_layers_d=3, num_channels=64, in_channels=1, out_channels=1, kernel_size=3, norm="BATCH", bias=False, padding=1, dtype=dtype, device=0)
a0, d0 = {k: v.cpu().clone() for k, v in ae.state_dict().items()}, {k: v.cpu().clone() for k, v in disc.state_dict().items()}
x = torch.randn(BA, 1, LA, generator=g).cuda(); lo = torch.zeros(6, device="cuda")


def aekl_run(steps=4):
    ae.load_state_dict(a0); disc.load_state_dict(d0); og, od = Adam(ae, lr=5e-3), Adam(disc, lr=5e-4); hist = []
    for i in range(steps):
        eps = randn(ctx, (BA, 1, LA // 4), seed=5, offset=i * BA * LA)
        ae.zero_grad(); disc.zero_grad()
        aekl_train_step(ae, disc, x, eps, 0.01, 1e-9, 1e4, True, losses_out=lo)
        hist.append(lo.cpu().clone()); og.step(); od.step()
    torch.cuda.synchronize()
    return hist, ae.flat_grad.clone(), disc.flat_grad.clone(), ae.flat.clone(), disc.flat.clone()


h1, ga1, gd1, pa1, pd1 = aekl_run(); h2, ga2, gd2, pa2, pd2 = aekl_run()
names = ["l1", "spectral", "kl", "gen", "disc_fake", "disc_real"]
ld = [(i, [names[j] for j in range(6) if h1[i][j] != h2[i][j]]) for i in range(len(h1)) if not torch.equal(h1[i], h2[i])]
print(f"AEKL/GAN step {dtype} B={BA}: loss values differing (step, which) {ld[:4]}; last-step gradients differing: autoencoder {differing(ga1, ga2, ae.entries)[:10]} "
      f"discriminator {differing(gd1, gd2, disc.entries)[:10]}; parameters after 4 steps equal: autoencoder {bool(torch.equal(pa1, pa2))} discriminator {bool(torch.equal(pd1, pd2))}", flush=True)
