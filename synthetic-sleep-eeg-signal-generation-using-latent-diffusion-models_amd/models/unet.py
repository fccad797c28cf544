"""`UNetModel` -- host-side mirror of the reference denoiser's interface
(/root/reference/src/models/unet.py:330-563: same constructor kwargs, same
`forward(x, timesteps=...)`, same 278-key `state_dict()`), executing on the
hand-written HIP kernels of libeegldm.so through the C ABI.

torch is used for device memory only.  Two ways to train: the fused native step
(eegldm.training.ldm_train_step: add_noise + forward + MSE + backward in one call, the
fast path), or -- once `parameters()` has been handed to a torch optimizer -- the
reference's own loop body: `model(x=..., timesteps=...)` then returns a tensor with a
grad_fn whose backward is the hand-written native backward (eegldm.autograd).
"""
import ctypes as C
import math
from collections import OrderedDict

import torch

from .._lib import lib, check, ptr, default_context, UNetCfg, F32, BF16, F16
from ..autograd import FlatModule, _UNetFn

_DT = {"float32": F32, "fp32": F32, torch.float32: F32, "bfloat16": BF16, "bf16": BF16, torch.bfloat16: BF16, "float16": F16, "fp16": F16, "half": F16, torch.float16: F16, F32: F32, BF16: BF16, F16: F16}
_ZERO_INIT_SUFFIX = ("out_layers.3.weight", "out_layers.3.bias", "proj_out.weight", "proj_out.bias", "out.2.weight", "out.2.bias")


class UNetModel(FlatModule):
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, num_classes=None, num_heads=1,
                 num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 n_embed=None, dtype="float32", device=0, ctx=None):
        if not 0.0 <= float(dropout) < 1.0:
            raise ValueError("dropout must be in [0, 1)")
        self.dropout = float(dropout)
        if num_classes is not None or n_embed is not None:
            raise NotImplementedError("class-conditional / codebook heads are not used by the reference configs")
        num_heads, num_head_channels, num_heads_upsample = int(num_heads), int(num_head_channels), int(num_heads_upsample)
        if num_heads < 1 or num_head_channels == 0:
            raise ValueError("num_heads >= 1 and num_head_channels = -1 or > 0 (unet.py:146-153)")
        self.conv_resample, self.num_heads, self.num_head_channels = bool(conv_resample), num_heads, num_head_channels
        self.num_heads_upsample = num_heads if num_heads_upsample == -1 else num_heads_upsample      # unet.py:354-355
        self.use_scale_shift_norm, self.resblock_updown = bool(use_scale_shift_norm), bool(resblock_updown)
        self.image_size, self.in_channels, self.model_channels, self.out_channels = image_size, in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions, self.channel_mult = num_res_blocks, list(attention_resolutions), list(channel_mult)
        self.dtype = _DT[dtype]
        self.ctx = ctx or default_context(device if isinstance(device, int) else torch.device(device).index or 0)
        self.device = torch.device("cuda", self.ctx.device)
        cfg = UNetCfg()
        cfg.in_channels, cfg.out_channels, cfg.model_channels, cfg.num_res_blocks = in_channels, out_channels, model_channels, num_res_blocks
        cfg.n_mult = len(self.channel_mult)
        for i, m in enumerate(self.channel_mult):
            cfg.channel_mult[i] = int(m)
        cfg.n_attn = len(self.attention_resolutions)
        for i, a in enumerate(self.attention_resolutions):
            cfg.attention_resolutions[i] = int(a)
        cfg.num_heads, cfg.dtype = self.num_heads, self.dtype
        cfg.num_head_channels = self.num_head_channels if self.num_head_channels > 0 else 0
        cfg.num_heads_upsample = self.num_heads_upsample
        cfg.use_scale_shift_norm = int(self.use_scale_shift_norm)
        cfg.resample_layers = 0 if self.resblock_updown else 1                          # Downsample / Upsample layers (unet.py:462-470,493-498)
        cfg.resample_pool_only = 0 if self.conv_resample else 1
        h = C.c_void_p()
        check(lib.eegldm_unet_create(self.ctx.h, C.byref(cfg), C.byref(h)))
        self.h = h
        if self.dropout > 0.0:      # nn.Dropout(p) of every ResBlock (unet.py:289): training-mode forwards only, masks from the device Philox stream
            self.set_dropout_seed(0x0D50)
        self.n_flat = int(lib.eegldm_unet_num_params(self.h))
        self.entries = OrderedDict()
        name = C.create_string_buffer(256)
        off, numel, ndim, shape = C.c_long(), C.c_long(), C.c_int(), (C.c_int * 3)()
        for i in range(lib.eegldm_unet_num_entries(self.h)):
            check(lib.eegldm_unet_entry(self.h, i, name, 256, C.byref(off), C.byref(numel), C.byref(ndim), shape))
            self.entries[name.value.decode()] = (off.value, numel.value, tuple(shape[k] for k in range(ndim.value)))
        self.flat = torch.zeros(self.n_flat, device=self.device, dtype=torch.float32)
        self.flat_grad = torch.zeros(self.n_flat, device=self.device, dtype=torch.float32)
        check(lib.eegldm_unet_bind(self.h, ptr(self.flat), ptr(self.flat_grad)))
        self.training = True
        self.reset_parameters()

    def set_dropout_seed(self, seed):
        """(Re)start the dropout mask stream: the same seed gives the same masks for the same sequence of training forwards
        (torch's counterpart is `torch.manual_seed` ahead of the loop)."""
        check(lib.eegldm_unet_set_dropout(self.h, self.dropout, int(seed)))

    # ------------------------------------------------------------------ parameters
    def reset_parameters(self, generator=None):
        """torch.nn default init (kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in)) for weights and biases),
        GroupNorm (1, 0), and the reference's zero_module layers (unet.py:39-45)."""
        sd = OrderedDict()
        for k, (_o, _n, shape) in self.entries.items():
            if len(shape) == 1 and (".in_layers.0." in k or ".out_layers.0." in k or ".norm." in k or k.startswith("out.0.")):
                sd[k] = torch.ones(shape) if k.endswith("weight") else torch.zeros(shape)
            elif k.endswith(_ZERO_INIT_SUFFIX):
                sd[k] = torch.zeros(shape)
            else:
                wshape = shape if len(shape) > 1 else self.entries[k[:-4] + "weight"][2]
                bound = 1.0 / math.sqrt(int(torch.tensor(wshape[1:]).prod()))
                sd[k] = (torch.rand(shape, generator=generator) * 2 - 1) * bound
        self.load_state_dict(sd)

    def state_dict(self):
        out = OrderedDict()
        for k, (o, n, shape) in self.entries.items():
            t = self.flat[o:o + n]
            if len(shape) == 3:       # packed [K][Cout][Cin] -> reference (Cout, Cin, K)
                t = t.reshape(shape[2], shape[0], shape[1]).permute(1, 2, 0)
            out[k] = t.reshape(shape).clone()
        return out

    def grad_dict(self):
        out = OrderedDict()
        for k, (o, n, shape) in self.entries.items():
            t = self.flat_grad[o:o + n]
            if len(shape) == 3:
                t = t.reshape(shape[2], shape[0], shape[1]).permute(1, 2, 0)
            out[k] = t.reshape(shape).clone()
        return out

    def load_state_dict(self, sd, strict=True):
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}   # DataParallel prefix (compute_mmds.py:475-478)
        missing = [k for k in self.entries if k not in sd]
        extra = [k for k in sd if k not in self.entries]
        if strict and (missing or extra):
            raise KeyError(f"state_dict mismatch: missing {missing[:4]}..., unexpected {extra[:4]}...")
        for k, (o, n, shape) in self.entries.items():
            if k not in sd:
                continue
            v = torch.as_tensor(sd[k]).detach().to(torch.float32)
            if tuple(v.shape) != tuple(shape):
                raise ValueError(f"{k}: shape {tuple(v.shape)} != {tuple(shape)}")
            if len(shape) == 3:
                v = v.permute(2, 0, 1)
            self.flat[o:o + n].copy_(v.reshape(-1).to(self.device))
        self.sync_weights()

    def sync_weights(self):
        check(lib.eegldm_unet_sync_weights(self.h))
        self._mark_synced()

    def zero_grad(self, set_to_none=True):
        self.flat_grad.zero_()

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def to(self, *a, **k):
        return self

    # ------------------------------------------------------------------ compute
    def forward(self, x, timesteps=None, context=None, y=None, **kwargs):
        assert timesteps is not None, "need to implement no-timestep usage"      # unet.py:520-521 (same assertion text)
        x = x.to(self.device, torch.float32).contiguous()
        t = timesteps.to(self.device, torch.int64).contiguous()
        if x.dim() != 3:
            raise ValueError(f"UNetModel expects x of shape (B, {self.in_channels}, L), got {tuple(x.shape)}")
        B, Cc, L = x.shape
        if Cc != self.in_channels:
            raise ValueError(f"UNetModel was built with in_channels={self.in_channels}, got x with {Cc} channels")
        if tuple(t.shape) != (B,):
            raise ValueError(f"timesteps must have shape ({B},) to match the batch, got {tuple(t.shape)}")
        if B == 0:
            return torch.empty(B, self.out_channels, L, device=self.device, dtype=torch.float32)      # empty batch in, empty batch out (as the reference modules do)
        levels = len(self.channel_mult)
        if L % (1 << (levels - 1)) != 0:
            raise ValueError(f"L={L} must be divisible by {1 << (levels - 1)} (one halving per resolution level; the reference's "
                             "length-mismatch crop at unet.py:544-551 is not implemented)")
        vec = 4 if self.dtype == F32 else 8              # contiguous-dimension granularity of the MFMA operand loads
        for lvl in range(levels):
            if (1 << lvl) in self.attention_resolutions and (L >> lvl) % vec != 0:
                raise ValueError(f"attention at downsample rate {1 << lvl} sees T={L >> lvl} positions; this engine needs T to be a "
                                 f"multiple of {vec} ({'fp32' if vec == 4 else 'bf16'}): use L divisible by {vec << lvl}")
        self._sync_if_stale()                            # a torch optimizer updated the flat parameter in place: refresh the compute copies
        if self.training and self._wants_graph(x):
            return _UNetFn.apply(self, x, t, self._flat_param())
        return self._forward_native(x, t)

    def _forward_native(self, x, t):
        B, _c, L = x.shape
        out = torch.empty(B, self.out_channels, L, device=self.device, dtype=torch.float32)
        check(lib.eegldm_unet_forward(self.h, ptr(x), ptr(t), ptr(out), B, L, 1 if self.training else 0))
        self._bump_tape()                                # the executor's single tape now belongs to THIS call (eegldm.autograd)
        return out

    __call__ = forward

    def backward(self, dy, need_dx=False):
        dy = dy.to(self.device, torch.float32).contiguous()
        dx = torch.empty(dy.shape[0], self.in_channels, dy.shape[2], device=self.device) if need_dx else None
        check(lib.eegldm_unet_backward(self.h, ptr(dy), ptr(dx)))
        self._bump_tape()                                # consumed
        return dx

    def __del__(self):
        try:
            lib.eegldm_unet_destroy(self.h)
        except Exception:
            pass
