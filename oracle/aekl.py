"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of MONAI-Generative `AutoencoderKL` and `PatchDiscriminator`
as the reference instantiates them (config/config_aekl_eeg.yaml:19-40;
/root/reference/src/train_autoencoderkl.py:129-137).

PARITY: `monai-generative` is an un-pinned dependency (/root/reference/requirements.txt:12), its source is not under
/root/reference, and the reference holds no tests or golden vectors for it, so WHAT MONAI computes is restated from the
published monai-generative 0.2.x algorithm (unpinned at that library boundary: that AutoencoderKL / PatchDiscriminator have
this structure, their state-dict key names, norm_num_groups semantics).  The restatement ITSELF is pinned against the
reference's own code (tests/golden/make_golden_r2.py -> aekl_twin_32_32_64.npz, disc_twin_k3.npz; tests/test_oracle_golden.py):
  * the local AutoencoderKL (/root/reference/src/models/ae_kl.py:123-291) with its mid-attention blocks removed is exactly this
    structure (conv, ResBlocks, right-pad stride-2 Downsample, nearest x2 + conv Upsample, final norm + conv, 1x1 heads,
    clamp(-30, 20), sigma = exp(log_var / 2), post_quant_conv) at [32,32,64] with GroupNorm(32): forward, input gradient and
    all 126 parameter gradients of recon.dy + 0.3 KL agree to 1e-4;
  * the local PatchGAN Discriminator (/root/reference/src/models/discriminator.py:15-84) with kernel-3 convs in place of its
    hard-coded kernel 4 is the configured PatchDiscriminator: logits, gradients, BatchNorm running statistics agree.
The GroupNorm group count (32 there, norm_num_groups = 1 in the configs) is a parameter of these functions.
"""
import torch
import torch.nn.functional as F

from .quant import q as sq, qw as wq

NORM_EPS = 1e-6


def _conv(sd, p, x, stride=1, padding=1):
    return sq(F.conv1d(x, wq(sd[p + ".conv.weight"]), sd.get(p + ".conv.bias"), stride=stride, padding=padding))


def _gn(sd, p, x, groups):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps=NORM_EPS)


def _resblock(sd, p, x, groups):
    h = sq(F.silu(_gn(sd, p + ".norm1", x, groups)))
    h = _conv(sd, p + ".conv1", h)
    h = sq(F.silu(_gn(sd, p + ".norm2", h, groups)))
    h = _conv(sd, p + ".conv2", h)
    if (p + ".nin_shortcut.conv.weight") in sd:
        x = _conv(sd, p + ".nin_shortcut", x, padding=0)
    return sq(x + h)


def aekl_plan(num_channels, num_res_blocks=2):
    """Block lists for encoder / decoder: (kind, cin, cout)."""
    nrb = [num_res_blocks] * len(num_channels) if isinstance(num_res_blocks, int) else list(num_res_blocks)
    enc = [("conv3", None, num_channels[0])]
    oc = num_channels[0]
    for i, c in enumerate(num_channels):
        ic, oc = oc, c
        for _ in range(nrb[i]):
            enc.append(("res", ic, oc)); ic = oc
        if i != len(num_channels) - 1:
            enc.append(("down", ic, ic))
    enc += [("gn", oc, oc), ("conv3", oc, None)]
    rev = list(reversed(num_channels)); rnrb = list(reversed(nrb))
    dec = [("conv3", None, rev[0])]
    oc = rev[0]
    for i, c in enumerate(rev):
        ic, oc = oc, c
        for _ in range(rnrb[i]):
            dec.append(("res", ic, oc)); ic = oc
        if i != len(rev) - 1:
            dec.append(("up", ic, ic))
    dec += [("gn", oc, oc), ("conv3", oc, None)]
    return enc, dec


def _run(sd, prefix, plan, x, groups):
    for i, (kind, _ci, _co) in enumerate(plan):
        p = f"{prefix}.blocks.{i}"
        if kind == "conv3":
            x = _conv(sd, p, x)
        elif kind == "res":
            x = _resblock(sd, p, x, groups)
        elif kind == "down":              # pad right only, then stride-2 conv, padding 0
            x = _conv(sd, p + ".conv", F.pad(x, (0, 1)), stride=2, padding=0)
        elif kind == "up":                # nearest x2 then conv3
            x = _conv(sd, p + ".conv", F.interpolate(x, scale_factor=2.0, mode="nearest"))
        elif kind == "gn":                # NO nonlinearity after the final norm
            x = sq(_gn(sd, p, x, groups))
    return x


def encode(sd, cfg, x):
    enc, _ = aekl_plan(cfg["num_channels"], cfg.get("num_res_blocks", 2))
    h = _run(sd, "encoder", enc, x, cfg.get("norm_num_groups", 1))
    z_mu = _conv(sd, "quant_conv_mu", h, padding=0)
    z_log_var = torch.clamp(_conv(sd, "quant_conv_log_sigma", h, padding=0), -30.0, 20.0)
    return z_mu, torch.exp(z_log_var / 2)


def decode(sd, cfg, z):
    _, dec = aekl_plan(cfg["num_channels"], cfg.get("num_res_blocks", 2))
    z = _conv(sd, "post_quant_conv", z, padding=0)
    return _run(sd, "decoder", dec, z, cfg.get("norm_num_groups", 1))


def forward(sd, cfg, x, eps):
    """returns (reconstruction, z_mu, z_sigma); eps replaces randn_like for parity."""
    z_mu, z_sigma = encode(sd, cfg, x)
    z = z_mu + eps * z_sigma
    return decode(sd, cfg, z), z_mu, z_sigma


def aekl_param_shapes(cfg):
    nc, lat = cfg["num_channels"], cfg["latent_channels"]
    cin, cout = cfg.get("in_channels", 1), cfg.get("out_channels", 1)
    enc, dec = aekl_plan(nc, cfg.get("num_res_blocks", 2))
    s = {}

    def conv(p, ci, co, k):
        s[p + ".conv.weight"] = (co, ci, k); s[p + ".conv.bias"] = (co,)

    def add(prefix, plan, first_in, last_out):
        for i, (kind, ci, co) in enumerate(plan):
            p = f"{prefix}.blocks.{i}"
            if kind == "conv3":
                conv(p, first_in if ci is None else ci, last_out if co is None else co, 3)
            elif kind == "res":
                s[p + ".norm1.weight"] = (ci,); s[p + ".norm1.bias"] = (ci,)
                conv(p + ".conv1", ci, co, 3)
                s[p + ".norm2.weight"] = (co,); s[p + ".norm2.bias"] = (co,)
                conv(p + ".conv2", co, co, 3)
                if ci != co:
                    conv(p + ".nin_shortcut", ci, co, 1)
            elif kind in ("down", "up"):
                conv(p + ".conv", ci, co, 3)
            elif kind == "gn":
                s[p + ".weight"] = (ci,); s[p + ".bias"] = (ci,)

    add("encoder", enc, cin, lat)
    add("decoder", dec, lat, cout)
    conv("quant_conv_mu", lat, lat, 1)
    conv("quant_conv_log_sigma", lat, lat, 1)
    conv("post_quant_conv", lat, lat, 1)
    return s


# ---------------------------------------------------------------- discriminator
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def disc_param_shapes(cfg):
    nch, nl, k = cfg["num_channels"], cfg["num_layers_d"], cfg["kernel_size"]
    s = {"initial_conv.conv.weight": (nch, cfg["in_channels"], k), "initial_conv.conv.bias": (nch,)}
    ic, oc = nch, nch * 2
    for l_ in range(nl):
        s[f"{l_}.conv.weight"] = (oc, ic, k)
        if cfg.get("bias", False):
            s[f"{l_}.conv.bias"] = (oc,)
        s[f"{l_}.adn.N.weight"] = (oc,); s[f"{l_}.adn.N.bias"] = (oc,)
        s[f"{l_}.adn.N.running_mean"] = (oc,); s[f"{l_}.adn.N.running_var"] = (oc,)
        s[f"{l_}.adn.N.num_batches_tracked"] = ()
        ic, oc = oc, oc * 2
    s["final_conv.conv.weight"] = (cfg["out_channels"], ic, k); s["final_conv.conv.bias"] = (cfg["out_channels"],)
    return s


def disc_forward(sd, cfg, x, training=True, running=None):
    """Returns the list of feature maps (caller uses [-1],
    /root/reference/src/train_autoencoderkl.py:213).  `running`: optional dict
    receiving updated running stats (train mode: biased var normalises,
    unbiased var feeds running_var, momentum 0.1)."""
    pad, nl = cfg.get("padding", 1), cfg["num_layers_d"]
    outs = []
    h = F.conv1d(sq(x), wq(sd["initial_conv.conv.weight"]), sd["initial_conv.conv.bias"], stride=2, padding=pad)
    h = sq(F.leaky_relu(h, 0.2))
    outs.append(h)
    for l_ in range(nl):
        stride = 1 if l_ == nl - 1 else 2
        h = sq(F.conv1d(h, wq(sd[f"{l_}.conv.weight"]), sd.get(f"{l_}.conv.bias"), stride=stride, padding=pad))
        rm = sd[f"{l_}.adn.N.running_mean"].clone(); rv = sd[f"{l_}.adn.N.running_var"].clone()
        h = F.batch_norm(h, rm, rv, sd[f"{l_}.adn.N.weight"], sd[f"{l_}.adn.N.bias"], training=training,
                         momentum=BN_MOMENTUM, eps=BN_EPS)
        if running is not None:
            running[f"{l_}.adn.N.running_mean"] = rm; running[f"{l_}.adn.N.running_var"] = rv
        h = sq(F.leaky_relu(h, 0.2))
        outs.append(h)
    k = sd["final_conv.conv.weight"].shape[-1]
    h = F.conv1d(h, wq(sd["final_conv.conv.weight"]), sd["final_conv.conv.bias"], stride=1, padding=(k - 1) // 2)
    outs.append(h)
    return outs
