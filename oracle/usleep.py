"""ORACLE (test infrastructure) -- CPU restatement of the U-Sleep feature extractor the reference uses for its FID
(/root/reference/src/models/usleep.py:20-287, called from /root/reference/src/compute_fid.py:357-386) in torch functional ops.

Structure (usleep.py:165-252): channels c_0 = in_chans, c_{i+1} = int(f_i * sqrt(complexity_factor)) with f_0 = n_time_filters,
f_{i+1} = int(f_i * sqrt(2)); `depth` encoder blocks [conv k 'same' -> ELU -> BatchNorm1d; the block output is the skip; zero-pad
1 + 1 when the length is odd; MaxPool(2)] (:20-51), bottom conv k -> ELU -> BatchNorm (:193-200), `depth` decoder blocks
[nearest x2 -> conv k=2 'same' (PyTorch pads an even kernel on the RIGHT) -> ELU -> BatchNorm -> crop both to the common length ->
cat([x, skip]) -> conv k -> ELU -> BatchNorm] (:54-98), classifier conv1 -> tanh -> AvgPool1d(input_size) -> conv1 -> ELU -> conv1
(:218-247).  forward returns (y_pred, decoder output, bottom) (:249-287).

Pinned: tests/golden/usleep_d12.npz (make_golden_r3.py imports the reference class with param_gen weights; the trained weights
`/project/params.pt` of compute_fid.py:367 are not in the tree), eval-mode and train-mode BatchNorm."""
import math

import torch
import torch.nn.functional as F

BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def usleep_channels(in_chans=2, depth=12, n_time_filters=5, complexity_factor=1.67):
    ch, f = [in_chans], n_time_filters
    for _ in range(depth + 1):
        ch.append(int(f * math.sqrt(complexity_factor)))       # usleep.py:168-171 (np.sqrt on python floats: same values)
        f = int(f * math.sqrt(2))
    return ch


def usleep_kernel_size(sfreq=100, time_conv_size_s=9 / 128):
    k = int(round(time_conv_size_s * sfreq))                   # np.round(...).astype(int): 7 at 100 Hz, 9 at 128 Hz
    if k % 2 == 0:
        raise ValueError("time_conv_size must be an odd number (usleep.py:157-163)")
    return k


def usleep_param_shapes(in_chans=2, sfreq=100, depth=12, n_time_filters=5, complexity_factor=1.67, n_classes=5, time_conv_size_s=9 / 128,
                        with_skip_connection=True):
    """state_dict keys and shapes in the reference's order (usleep.py: encoder.{i}.block_prepool.{0,2}, bottom.{0,2},
    decoder.{i}.block_preskip.{1,3} / block_postskip.{0,2}, clf.{0,3,5})."""
    ch = usleep_channels(in_chans, depth, n_time_filters, complexity_factor)
    k = usleep_kernel_size(sfreq, time_conv_size_s)
    s = {}

    def conv(p, co, ci, kk):
        s[p + ".weight"] = (co, ci, kk); s[p + ".bias"] = (co,)

    def bn(p, c):
        s[p + ".weight"] = (c,); s[p + ".bias"] = (c,); s[p + ".running_mean"] = (c,); s[p + ".running_var"] = (c,); s[p + ".num_batches_tracked"] = ()

    for i in range(depth):
        conv(f"encoder.{i}.block_prepool.0", ch[i + 1], ch[i], k); bn(f"encoder.{i}.block_prepool.2", ch[i + 1])
    conv("bottom.0", ch[-1], ch[-2], k); bn("bottom.2", ch[-1])
    rc = ch[::-1]
    for i in range(depth):
        conv(f"decoder.{i}.block_preskip.1", rc[i + 1], rc[i], 2); bn(f"decoder.{i}.block_preskip.3", rc[i + 1])
        conv(f"decoder.{i}.block_postskip.0", rc[i + 1], (2 if with_skip_connection else 1) * rc[i + 1], k); bn(f"decoder.{i}.block_postskip.2", rc[i + 1])
    conv("clf.0", ch[1], ch[1], 1); conv("clf.3", n_classes, ch[1], 1); conv("clf.5", n_classes, n_classes, 1)
    return s


def _same(x, w, b):
    """nn.Conv1d(padding='same'): total padding k - 1, the odd remainder goes to the RIGHT (torch's _conv_forward for 'same')."""
    k = w.shape[-1]
    left = (k - 1) // 2
    return F.conv1d(F.pad(x, (left, k - 1 - left)), w, b)


def _bn(x, sd, p, training, running):
    rm, rv = sd[p + ".running_mean"].clone(), sd[p + ".running_var"].clone()
    y = F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], training=training, momentum=BN_MOMENTUM, eps=BN_EPS)
    if running is not None and training:
        running[p + ".running_mean"] = rm; running[p + ".running_var"] = rv
    return y


def usleep_forward(sd, x, depth=12, input_size=3000, training=False, running=None, with_skip_connection=True):
    """x (B, in_chans, T) -> (y_pred, decoder output (B, c_1, T), bottom (B, c_{depth+1}, T / 2^depth rounded up)).
    training=True = BatchNorm on batch statistics: what compute_fid.py literally runs (it never calls model.eval())."""
    res = []
    for i in range(depth):
        p = f"encoder.{i}.block_prepool"
        x = _bn(F.elu(_same(x, sd[p + ".0.weight"], sd[p + ".0.bias"])), sd, p + ".2", training, running)
        res.append(x)
        if x.shape[-1] % 2:
            x = F.pad(x, (1, 1))                                 # ConstantPad1d(1, 0): BOTH sides, zeros take part in the max
        x = F.max_pool1d(x, 2, 2)
    k = sd["bottom.0.weight"].shape[-1]
    x = _bn(F.elu(F.conv1d(x, sd["bottom.0.weight"], sd["bottom.0.bias"], padding=(k - 1) // 2)), sd, "bottom.2", training, running)
    bottom = x
    for i, r in enumerate(res[::-1]):
        p = f"decoder.{i}"
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        x = _bn(F.elu(_same(x, sd[p + ".block_preskip.1.weight"], sd[p + ".block_preskip.1.bias"])), sd, p + ".block_preskip.3", training, running)
        if with_skip_connection:
            n = min(x.shape[-1], r.shape[-1])
            x = torch.cat([x[..., :n], r[..., :n]], 1)
        x = _bn(F.elu(_same(x, sd[p + ".block_postskip.0.weight"], sd[p + ".block_postskip.0.bias"])), sd, p + ".block_postskip.2", training, running)
    y = torch.tanh(F.conv1d(x, sd["clf.0.weight"], sd["clf.0.bias"]))
    y = F.avg_pool1d(y, input_size)
    y = F.conv1d(F.elu(F.conv1d(y, sd["clf.3.weight"], sd["clf.3.bias"])), sd["clf.5.weight"], sd["clf.5.bias"])
    if y.shape[-1] == 1:
        y = y[:, :, 0]
    return y, x, bottom


def fid_features(sd, windows, depth=12, training=False):
    """compute_fid.py:373-384: crop the loader's 36-sample pads when present, duplicate the single EEG channel into the 2-channel
    input, take the bottleneck activation and squeeze its length-1 time axis -> (B, c_{depth+1}).  (The script unpacks two values
    from a forward that returns three, `predict, outputs = model(...)`: it cannot run as written; `outputs.squeeze(-1)` only makes
    sense for the (B, C, 1) bottleneck, so that is the feature.)"""
    if windows.shape[-1] == 3072:
        windows = windows[:, :, 36:-36]
    _y, _x, bottom = usleep_forward(sd, torch.cat([windows, windows], 1), depth=depth, input_size=windows.shape[-1], training=training)
    return bottom.squeeze(-1)
