"""Round-6 developer tool: A/B of the AEKL [2,2,4] + PatchDiscriminator GAN step (BASELINE configs[1], B = 256, L = 3072, bf16) between
environment switches of the library, alternating inside ONE process (eegldm_debug_reload_env) so that box-to-box spread cancels.

    python tools/r06/aekl_ab.py "EEGLDM_DISC_NO_FUSED_TAIL=1" ["OTHER=1,MORE=2" ...]      # first column is always the default build
"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib
from eegldm.models import AutoencoderKL, PatchDiscriminator
from eegldm.training import Adam, aekl_train_step, randn
B, L, dtype = int(os.environ.get("AB_B", "256")), 3072, os.environ.get("AB_DTYPE", "bfloat16")
ctx = eegldm.default_context(0)
ae = AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=[2, 2, 4], latent_channels=1, num_res_blocks=2,
                   norm_num_groups=1, attention_levels=[False, False, False], dtype=dtype, device=0)
disc = PatchDiscriminator(spatial_dims=1, num_layers_d=3, num_channels=64, in_channels=1, out_channels=1, kernel_size=3,
                          norm="BATCH", bias=False, padding=1, dtype=dtype, device=0)
og, od = Adam(ae, lr=5e-3), Adam(disc, lr=5e-4)
x = torch.randn(B, 1, L, device="cuda")
lo = torch.zeros(6, device="cuda")
def step(i):
    eps = randn(ctx, (B, 1, L // 4), seed=5, offset=i * B * L)
    ae.zero_grad(); disc.zero_grad()
    aekl_train_step(ae, disc, x, eps, 0.01, 1e-9, 1e4, True, losses_out=lo)
    og.step(); od.step()
def setenv(spec):
    kv = dict(s.split("=", 1) for s in spec.split(",") if s)
    for k, v in kv.items(): os.environ[k] = v
    lib.eegldm_debug_reload_env()
    return kv
def clearenv(kv):
    for k in kv: os.environ.pop(k, None)
    lib.eegldm_debug_reload_env()
variants = [""] + sys.argv[1:]
res = {v: [] for v in variants}
it = 0
for rep in range(4):
    for v in variants:
        kv = setenv(v)
        for i in range(3): step(it); it += 1
        torch.cuda.synchronize(); t0 = time.time()
        n = 20
        for i in range(n): step(it); it += 1
        torch.cuda.synchronize(); res[v].append((time.time() - t0) / n * 1e3)
        clearenv(kv)
for v in variants:
    r = sorted(res[v])
    print(f"{v or 'default':50s} ms/step min {r[0]:.3f} med {r[len(r)//2]:.3f}  all {[round(t, 3) for t in res[v]]}")
print("losses", [round(float(t), 4) for t in lo.cpu()])
