import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm.models import UNetModel
net = UNetModel(image_size=64, in_channels=1, out_channels=1, model_channels=32, num_res_blocks=1, attention_resolutions=[2], channel_mult=[1, 2], resblock_updown=True, dtype="bfloat16")
for B, L in [(1, 64), (3, 72), (2, 66), (2, 4), (0, 64), (2, 63)]:
    try:
        y = net(torch.randn(B, 1, L), timesteps=torch.randint(0, 1000, (B,)))
        print(B, L, "ok", tuple(y.shape), bool(torch.isfinite(y).all()))
    except Exception as e:
        print(B, L, "ERR", type(e).__name__, str(e)[:150])
try:
    net(torch.randn(2, 1, 64), timesteps=torch.tensor([5]))
except Exception as e:
    print("t mismatch ERR", type(e).__name__, str(e)[:150])
try:
    net(torch.randn(2, 2, 64), timesteps=torch.tensor([5, 6]))
except Exception as e:
    print("chan mismatch ERR", type(e).__name__, str(e)[:150])
