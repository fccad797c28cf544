"""Developer tool: DDIM-50 of two windows under different kernel-path switches (and the fp32 engine), pairwise rel-L2 of the final latents --
how much of the few-row chain's difference from the general kernels is bf16 rounding order amplified by 50 chained forwards."""
import os, subprocess, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SCRIPT = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
import eegldm
from eegldm.models import UNetModel, AutoencoderKL
from eegldm.training import randn
from eegldm.sampling import ddim_sample, make_sampling_scheduler
dt = sys.argv[2]
ctx = eegldm.default_context(0)
torch.manual_seed(0)      # the module's default init draws from torch's global generator: same weights in every process
u = UNetModel(image_size=768, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4], channel_mult=[1, 2, 4],
              resblock_updown=True, dtype=dt)
g = torch.Generator().manual_seed(42); sd = u.state_dict()
u.load_state_dict({k: (torch.randn(v.shape, generator=g) * 0.02 if float(v.abs().sum()) == 0.0 else v) for k, v in sd.items()})
ae = AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=[32, 32, 64], latent_channels=1, num_res_blocks=2, norm_num_groups=1,
                   attention_levels=[False] * 3, dtype=dt)
steps = int(sys.argv[3])
x, z = ddim_sample(u, ae, make_sampling_scheduler(steps), randn(ctx, (2, 1, 768), seed=4242))
np.savez(sys.argv[1], z=z.float().cpu().numpy())
''' % ROOT
OFF = {"EEGLDM_NO_CONV_SKINNY": "1", "EEGLDM_NO_EVAL_GN_FUSE": "1", "EEGLDM_SAMPLE_NO_EMB_TABLE": "1", "EEGLDM_GN_NO_FEW_SLAB_NARROW": "1", "EEGLDM_SAMPLE_OWN_STREAM": "1"}
VAR = {"fp32": ({}, "float32"), "general": (OFF, "bfloat16"), "skinny_only": ({"EEGLDM_NO_EVAL_GN_FUSE": "1"}, "bfloat16"), "few_row": ({}, "bfloat16"),
       "general_narrow_gn": (dict(OFF, EEGLDM_GN_NO_FEW_SLAB_NARROW="0") if False else {k: v for k, v in OFF.items() if k != "EEGLDM_GN_NO_FEW_SLAB_NARROW"}, "bfloat16")}
for steps in (5, 50):
    z = {}
    for name, (env_extra, dt) in VAR.items():
        out = f"/tmp/ddim_ab_{name}.npz"
        env = dict(os.environ); env.update(env_extra)
        r = subprocess.run([sys.executable, "-c", SCRIPT, out, dt, str(steps)], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        z[name] = np.load(out)["z"]
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    print(f"--- {steps} DDIM steps: rel-L2 of the final latents")
    for a in z:
        print(f"{a:18s}", " ".join(f"{rel(z[a], z[b]):.3e}" for b in z))
