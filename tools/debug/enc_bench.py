"""Developer tool: the frozen stage-1 encode of the LDM step alone (AutoencoderKL [32,32,64], B = 256, L = 3072, bf16), fused vs layer by layer."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm.models import AutoencoderKL
from eegldm.training import randn
ctx = eegldm.default_context(0)
ae = AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=[32, 32, 64], latent_channels=1, num_res_blocks=2,
                   norm_num_groups=1, attention_levels=[False] * 3, dtype="bfloat16")
B, L = 256, 3072
x = torch.randn(B, 1, L, device="cuda"); eps = randn(ctx, (B, 1, L // 4), seed=3)
for mode in ("fused", "layers", "fused"):
    if mode == "layers": os.environ["EEGLDM_AEKL_NO_FUSED_ENC"] = "1"
    else: os.environ.pop("EEGLDM_AEKL_NO_FUSED_ENC", None)
    for _ in range(3): z = ae.encode_stage_2_inputs(x, eps=eps)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20): z = ae.encode_stage_2_inputs(x, eps=eps)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 20
    print(f"frozen encode B={B} L={L} [{mode}]: {dt*1e6:.0f} us  z std {float(z.std()):.5f}")
