"""`AutoencoderKL` and `PatchDiscriminator` -- host-side mirrors of the monai-generative classes the
reference instantiates (/root/reference/src/train_autoencoderkl.py:129-137, config/config_aekl_eeg.yaml:19-40):
same constructor kwargs, methods (`forward`, `encode`, `sampling`, `decode`, `reconstruct`,
`encode_stage_2_inputs`, `decode_stage_2_outputs`) and state_dict keys, executing on libeegldm.so."""
import ctypes as C
import math

import torch

from .._lib import lib, check, ptr, default_context, AeklCfg, DiscCfg
from ._flat import DT, read_entries, unpack, pack_into
from ..autograd import FlatModule, _AeklFn, _DiscFn


class _Flat(FlatModule):
    def _init_flat(self, n_params):
        self.flat = torch.zeros(n_params, device=self.device, dtype=torch.float32)
        self.flat_grad = torch.zeros(n_params, device=self.device, dtype=torch.float32)
        self.training = True

    def state_dict(self):
        return unpack(self.flat, self.entries)

    def grad_dict(self):
        return unpack(self.flat_grad, self.entries)

    def zero_grad(self, set_to_none=True):
        self.flat_grad.zero_()

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def to(self, *a, **k):
        return self

    def _default_init(self, generator=None, conv_std=None):
        sd = {}
        for k, (_o, _n, shape) in self.entries.items():
            if len(shape) == 1 and ("norm" in k or ".adn.N." in k or k.split(".")[-2].isdigit() and "blocks" in k):
                sd[k] = torch.ones(shape) if k.endswith("weight") else torch.zeros(shape)
                if conv_std and k.endswith("weight"):
                    sd[k] = 1.0 + conv_std * torch.randn(shape, generator=generator)
            elif conv_std and len(shape) == 3:
                sd[k] = conv_std * torch.randn(shape, generator=generator)
            else:
                wshape = shape if len(shape) > 1 else self.entries[k[:-4] + "weight"][2]
                bound = 1.0 / math.sqrt(int(torch.tensor(wshape[1:]).prod()))
                sd[k] = (torch.rand(shape, generator=generator) * 2 - 1) * bound
        return sd


class AutoencoderKL(_Flat):
    def __init__(self, spatial_dims=1, in_channels=1, out_channels=1, num_res_blocks=2, num_channels=(32, 64, 64, 64),
                 attention_levels=None, latent_channels=3, norm_num_groups=32, norm_eps=1e-6,
                 with_encoder_nonlocal_attn=False, with_decoder_nonlocal_attn=False, dtype="float32", device=0, ctx=None, **_ignored):
        if spatial_dims != 1:
            raise NotImplementedError("the reference is 1-D (config_aekl_eeg.yaml:21)")
        num_channels = list(num_channels)
        if attention_levels is not None and any(attention_levels) or with_encoder_nonlocal_attn or with_decoder_nonlocal_attn:
            raise NotImplementedError("every reference config disables attention in the autoencoder (config_aekl_eeg.yaml:26-28)")
        if isinstance(num_res_blocks, (list, tuple)):
            if len(set(num_res_blocks)) != 1:
                raise NotImplementedError("per-level num_res_blocks")
            num_res_blocks = num_res_blocks[0]
        if abs(norm_eps - 1e-6) > 1e-12:
            raise NotImplementedError("norm_eps is fixed to the MONAI default 1e-6")
        self.in_channels, self.out_channels, self.num_channels = in_channels, out_channels, num_channels
        self.latent_channels, self.num_res_blocks, self.norm_num_groups = latent_channels, num_res_blocks, norm_num_groups
        self.dtype = DT[dtype]
        self.ctx = ctx or default_context(device if isinstance(device, int) else torch.device(device).index or 0)
        self.device = torch.device("cuda", self.ctx.device)
        cfg = AeklCfg()
        cfg.in_channels, cfg.out_channels, cfg.n_levels = in_channels, out_channels, len(num_channels)
        for i, c in enumerate(num_channels):
            cfg.num_channels[i] = int(c)
        cfg.latent_channels, cfg.num_res_blocks, cfg.norm_num_groups, cfg.dtype = latent_channels, num_res_blocks, norm_num_groups, self.dtype
        h = C.c_void_p()
        check(lib.eegldm_aekl_create(self.ctx.h, C.byref(cfg), C.byref(h)))
        self.h = h
        self.entries = read_entries(self.h, lib.eegldm_aekl_num_entries, lib.eegldm_aekl_entry)
        self._init_flat(int(lib.eegldm_aekl_num_params(self.h)))
        check(lib.eegldm_aekl_bind(self.h, ptr(self.flat), ptr(self.flat_grad)))
        self.down = 2 ** (len(num_channels) - 1)
        self.load_state_dict(self._default_init())

    def load_state_dict(self, sd, strict=True):
        pack_into(self.flat, self.entries, sd, strict)
        self.sync_weights()

    def sync_weights(self):
        check(lib.eegldm_aekl_sync_weights(self.h))
        self._mark_synced()

    def _x(self, x):
        return x.to(self.device, torch.float32).contiguous()

    def _win(self, x, channels, latent=False):
        """Validated (B, channels, L) input on the device; L must survive the strided convolutions (a multiple of `down`)."""
        x = self._x(x)
        if x.dim() != 3 or x.shape[1] != channels:
            raise ValueError(f"AutoencoderKL expects a (B, {channels}, L) tensor here, got {tuple(x.shape)}")
        if not latent and x.shape[2] % self.down != 0:
            raise ValueError(f"L={x.shape[2]} must be a multiple of {self.down} (one stride-2 downsample per level but the last; "
                             "the loader pads 3000-sample windows to 3072, dataset.py:12-19)")
        return x

    def encode(self, x):
        x = self._win(x, self.in_channels); B, _c, L = x.shape
        mu = torch.empty(B, self.latent_channels, L // self.down, device=self.device); sg = torch.empty_like(mu)
        if B == 0:
            return mu, sg
        check(lib.eegldm_aekl_encode(self.h, ptr(x), None, None, ptr(mu), ptr(sg), B, L))
        self._bump_tape()
        return mu, sg

    def sampling(self, z_mu, z_sigma, eps=None):
        eps = torch.randn_like(z_sigma) if eps is None else eps.to(z_sigma.device)
        return z_mu + eps * z_sigma        # elementwise glue on caller tensors; the fused path is encode_stage_2_inputs

    def encode_stage_2_inputs(self, x, eps=None, scale_factor=None):
        """z = mu + eps * sigma (Stage1Wrapper, training.py:15-26), optionally times scale_factor (training.py:426)."""
        x = self._win(x, self.in_channels); B, _c, L = x.shape
        z = torch.empty(B, self.latent_channels, L // self.down, device=self.device)
        if B == 0:
            return z
        if eps is None:
            eps = torch.randn(z.shape, device=self.device)
        eps = self._x(eps)
        check(lib.eegldm_aekl_encode(self.h, ptr(x), ptr(eps), ptr(z), None, None, B, L))
        self._bump_tape()
        if scale_factor is not None and float(scale_factor) != 1.0:
            check(lib.eegldm_axpy(self.ctx.h, ptr(z), ptr(z), float(scale_factor) - 1.0, z.numel()))
        return z

    def decode(self, z):
        z = self._win(z, self.latent_channels, latent=True); B, _c, Ll = z.shape
        out = torch.empty(B, self.out_channels, Ll * self.down, device=self.device)
        if B == 0:
            return out
        check(lib.eegldm_aekl_decode(self.h, ptr(z), ptr(out), B, Ll))
        self._bump_tape()
        return out

    decode_stage_2_outputs = decode

    def reconstruct(self, x):
        mu, _ = self.encode(x)
        return self.decode(mu)

    def forward(self, x, eps=None, kl_out=None):
        """(reconstruction, z_mu, z_sigma) as the reference's call at train_autoencoderkl.py:204.  With a torch optimizer on `parameters()`
        the three outputs carry a grad_fn (eegldm.autograd): the KL written with tensor ops on z_mu / z_sigma (train_autoencoderkl.py:210-211)
        back-propagates through the native backward."""
        x = self._win(x, self.in_channels); B, _c, L = x.shape
        if B == 0:
            e = torch.empty(0, self.latent_channels, L // self.down, device=self.device)
            return torch.empty(0, self.out_channels, L, device=self.device), e, e.clone()
        if eps is None:
            eps = torch.randn(B, self.latent_channels, L // self.down, device=self.device)
        eps = self._x(eps)
        self._sync_if_stale()
        if kl_out is None and self.training and self._wants_graph(x):
            return _AeklFn.apply(self, x, eps, self._flat_param())
        return self._forward_native(x, eps, kl_out)

    def _forward_native(self, x, eps, kl_out=None):
        B, _c, L = x.shape
        recon = torch.empty(B, self.out_channels, L, device=self.device)
        mu = torch.empty(B, self.latent_channels, L // self.down, device=self.device); sg = torch.empty_like(mu)
        check(lib.eegldm_aekl_forward(self.h, ptr(x), ptr(eps), ptr(recon), ptr(mu), ptr(sg), ptr(kl_out), B, L))
        self._bump_tape()
        return recon, mu, sg

    __call__ = forward

    def backward(self, d_recon, kl_weight=0.0, need_dx=False):
        d = self._x(d_recon)
        dx = torch.empty(d.shape[0], self.in_channels, d.shape[2], device=self.device) if need_dx else None
        check(lib.eegldm_aekl_backward(self.h, ptr(d), float(kl_weight), ptr(dx)))
        self._bump_tape()
        return dx

    def __del__(self):
        try:
            lib.eegldm_aekl_destroy(self.h)
        except Exception:
            pass


class PatchDiscriminator(_Flat):
    def __init__(self, spatial_dims=1, num_channels=64, in_channels=1, out_channels=1, num_layers_d=3, kernel_size=4,
                 activation=None, norm="BATCH", bias=False, padding=1, dropout=0.0, last_conv_kernel_size=None,
                 dtype="float32", device=0, ctx=None):
        if spatial_dims != 1 or str(norm).upper() != "BATCH" or dropout not in (0, 0.0) or activation is not None:
            raise NotImplementedError("reference config: spatial_dims=1, norm='BATCH', LeakyReLU(0.2), no dropout (config_aekl_eeg.yaml:30-40)")
        if last_conv_kernel_size not in (None, kernel_size):
            raise NotImplementedError("last_conv_kernel_size != kernel_size")
        self.in_channels, self.out_channels, self.num_layers_d = in_channels, out_channels, num_layers_d
        self.kernel_size, self.padding = kernel_size, padding
        self.dtype = DT[dtype]
        self.ctx = ctx or default_context(device if isinstance(device, int) else torch.device(device).index or 0)
        self.device = torch.device("cuda", self.ctx.device)
        cfg = DiscCfg()
        cfg.in_channels, cfg.out_channels, cfg.num_channels, cfg.num_layers_d = in_channels, out_channels, num_channels, num_layers_d
        cfg.kernel_size, cfg.padding, cfg.bias, cfg.dtype = kernel_size, padding, int(bool(bias)), self.dtype
        h = C.c_void_p()
        check(lib.eegldm_disc_create(self.ctx.h, C.byref(cfg), C.byref(h)))
        self.h = h
        self.entries = read_entries(self.h, lib.eegldm_disc_num_entries, lib.eegldm_disc_entry)
        self.buf_entries = read_entries(self.h, lib.eegldm_disc_num_buffer_entries, lib.eegldm_disc_buffer_entry)
        self._init_flat(int(lib.eegldm_disc_num_params(self.h)))
        self.buffers = torch.zeros(int(lib.eegldm_disc_num_buffers(self.h)), device=self.device)
        for k, (o, n, _s) in self.buf_entries.items():
            if k.endswith("running_var"):
                self.buffers[o:o + n] = 1.0
        check(lib.eegldm_disc_bind(self.h, ptr(self.flat), ptr(self.flat_grad), ptr(self.buffers)))
        # MONAI initialise_weights: conv N(0, 0.02), BatchNorm weight N(1, 0.02), bias 0
        sd = self._default_init(conv_std=0.02)
        self.load_state_dict(sd, strict=False)

    def state_dict(self):
        sd = unpack(self.flat, self.entries)
        for k, (o, n, shape) in self.buf_entries.items():
            v = self.buffers[o:o + n].clone()
            sd[k] = v.reshape(shape) if shape else v.reshape(()).round().to(torch.int64)
        return sd

    def load_state_dict(self, sd, strict=True):
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        pack_into(self.flat, self.entries, sd, strict, extra_ok=tuple(self.buf_entries))
        for k, (o, n, _s) in self.buf_entries.items():
            if k in sd:
                self.buffers[o:o + n].copy_(torch.as_tensor(sd[k]).to(torch.float32).reshape(-1).to(self.device))
        self.sync_weights()

    def sync_weights(self):
        check(lib.eegldm_disc_sync_weights(self.h))
        self._mark_synced()

    def forward(self, x, features=True):
        """The reference's return value: the list of per-block feature maps -- initial conv + LeakyReLU, one per conv + BatchNorm +
        LeakyReLU layer, then the final conv's logits (the entry the trainer uses, train_autoencoderkl.py:213).  The earlier maps are
        copied out of the native forward's tape as fp32 (B, C, L) tensors; `features=False` skips those copies and leaves None in
        their places (the train step itself is one native call and never comes through here)."""
        x = x.to(self.device, torch.float32).contiguous(); B, _c, L = x.shape
        self._sync_if_stale()
        if self._wants_graph(x):            # the logits carry a grad_fn (the feature maps below are copies, as before)
            logits = _DiscFn.apply(self, x, self._flat_param())
        else:
            logits = self._forward_native(x, 1 if self.training else 0)
        feats = [None] * (self.num_layers_d + 1)
        if features:
            Cc, Lf = C.c_int(), C.c_int()
            for i in range(self.num_layers_d + 1):
                check(lib.eegldm_disc_feature(self.h, i, None, C.byref(Cc), C.byref(Lf)))
                f = torch.empty(B, Cc.value, Lf.value, device=self.device)
                check(lib.eegldm_disc_feature(self.h, i, ptr(f), None, None))
                feats[i] = f
        return feats + [logits]

    __call__ = forward

    def _forward_native(self, x, training):
        B, _c, L = x.shape
        Lo = L
        for _ in range(self.num_layers_d):
            Lo = (Lo + 2 * self.padding - self.kernel_size) // 2 + 1          # initial + (num_layers_d - 1) stride-2 convs; the rest keep L
        logits = torch.empty(B, self.out_channels, Lo, device=self.device)
        check(lib.eegldm_disc_forward(self.h, ptr(x), ptr(logits), B, L, int(training)))
        self._bump_tape()
        return logits

    def backward(self, dlogits, need_dx=False, param_grads=True, in_shape=None):
        d = dlogits.to(self.device, torch.float32).contiguous()
        dx = torch.empty(in_shape, device=self.device) if need_dx else None
        check(lib.eegldm_disc_backward(self.h, ptr(d), ptr(dx), 1 if param_grads else 0))
        self._bump_tape()
        return dx

    def __del__(self):
        try:
            lib.eegldm_disc_destroy(self.h)
        except Exception:
            pass
