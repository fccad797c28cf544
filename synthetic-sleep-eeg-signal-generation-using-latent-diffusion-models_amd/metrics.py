"""Quality metrics of reconstructed / generated windows on the device, with the interfaces the reference's
evaluation code uses:

* `MultiScaleSSIMMetric(spatial_dims=1, data_range=1.0, kernel_size=7)(y_pred, y) -> (B, 1)`
  -- /root/reference/src/compute_mmds.py:297-408 (the reference's local 1-D adaptation of MONAI's metric), call site :487-503.
* `compute_psd(windows, sfreq=100, fmax=18) -> (psds (B, n_freqs), freqs)` and `mean_psd_db`
  -- mne `Epochs.compute_psd(fmax=18)` + average + 10 log10 as used at /root/reference/src/sample_trials.py:172-181
  (multitaper, DPSS half-bandwidth 4, low-bias tapers, normalization "length").  The DPSS tapers (a few KB, once per window
  length) come from scipy on the host; everything per window runs in libeegldm (eegldm_psd_multitaper).
* `band_powers`: integrates a PSD over the classical sleep-EEG bands.
* `USleep(...)`, `fid_features(model, windows)`, `FIDMetric()(y_pred, y)`, `FeatureMoments`
  -- the FID of /root/reference/src/compute_fid.py:341-419: the U-Sleep feature extractor (/root/reference/src/models/usleep.py:101-287,
  same constructor kwargs, state_dict keys and forward return value) runs in libeegldm (csrc/usleep.hip), the features' mean and
  covariance are accumulated on the device in fp64 (eegldm_feature_moments), and the 302 x 302 trace-of-square-root is host linear
  algebra in fp64 -- where monai-generative's FIDMetric (scipy.linalg.sqrtm) does it too.
"""
import ctypes as C
import functools

import numpy as np
import torch

from ._lib import lib, check, ptr, default_context, USleepCfg

BANDS = {"delta": (0.5, 4.0), "theta": (4.0, 8.0), "alpha": (8.0, 13.0), "sigma": (11.0, 16.0), "beta": (13.0, 18.0)}


def gaussian_1d(kernel_size, sigma):
    """compute_mmds.py:185-195: exp(-(d/sigma)^2/2), d = -(k-1)/2 ... (k-1)/2, normalised to sum 1."""
    dist = np.arange((1 - kernel_size) / 2, (1 + kernel_size) / 2, 1.0, dtype=np.float32)
    g = np.exp(-np.power(dist / np.float32(sigma), 2) / 2).astype(np.float32)
    return g / g.sum()


class MultiScaleSSIMMetric:
    def __init__(self, spatial_dims=1, data_range=1.0, kernel_type="gaussian", kernel_size=11, kernel_sigma=1.5, k1=0.01, k2=0.03,
                 weights=(0.0448, 0.2856, 0.3001, 0.2363, 0.1333), reduction="mean", device=0, ctx=None):
        if spatial_dims != 1:
            raise NotImplementedError("the reference evaluates 1-D windows (compute_mmds.py:487)")
        ks = int(kernel_size[0] if isinstance(kernel_size, (tuple, list)) else kernel_size)
        sg = float(kernel_sigma[0] if isinstance(kernel_sigma, (tuple, list)) else kernel_sigma)
        if str(kernel_type).lower() == "gaussian":
            self.kernel = gaussian_1d(ks, sg)
        elif str(kernel_type).lower() == "uniform":
            self.kernel = np.full(ks, 1.0 / ks, np.float32)
        else:
            raise ValueError(kernel_type)
        self.data_range, self.k1, self.k2 = float(data_range), float(k1), float(k2)
        self.weights = np.asarray(weights, np.float32)
        self.ctx = ctx or default_context(device)
        self.device = torch.device("cuda", self.ctx.device)

    def __call__(self, y_pred, y):
        if y_pred.shape != y.shape:
            raise ValueError(f"y_pred and y should have same shapes, got {y_pred.shape} and {y.shape}.")
        if y_pred.dim() != 3:
            raise ValueError(f"y_pred should have 3 dimensions (batch, channel, length) when using 1 spatial dimension, got {y_pred.dim()}.")
        a = y_pred.to(self.device, torch.float32).contiguous(); b = y.to(self.device, torch.float32).contiguous()
        B, Cc, L = a.shape
        ks, ns = len(self.kernel), len(self.weights)
        div = max(1, ns - 1) ** 2
        if L // div <= ks - 1:
            raise ValueError(f"For a given number of `weights` parameters {ns} and kernel size {ks}, the image height must be larger than "
                             f"{(ks - 1) * div}.")
        out = torch.empty(B, 1, device=self.device)
        if B:
            check(lib.eegldm_ms_ssim_1d(self.ctx.h, ptr(a), ptr(b), ptr(out), B, Cc, L, (C.c_float * ks)(*self.kernel.tolist()), ks,
                                        (C.c_float * ns)(*self.weights.tolist()), ns, self.data_range, self.k1, self.k2))
        return out


@functools.lru_cache(maxsize=8)
def dpss_tapers(n_times, half_nbw=4.0, low_bias=True):
    """Tapers and weights as mne's multitaper path builds them: scipy dpss(N, NW, Kmax = int(2 NW), sym=False, norm=2), keep the tapers
    whose concentration exceeds 0.9 (low_bias), weights = sqrt(concentration)."""
    from scipy.signal.windows import dpss
    tapers, ratios = dpss(n_times, half_nbw, int(2 * half_nbw), sym=False, norm=2, return_ratios=True)
    keep = ratios > 0.9 if low_bias else np.ones_like(ratios, bool)
    if not keep.any():
        keep = np.zeros_like(ratios, bool); keep[0] = True
    return np.ascontiguousarray(tapers[keep], np.float32), np.sqrt(ratios[keep]).astype(np.float32)


def compute_psd(windows, sfreq=100.0, fmin=0.0, fmax=18.0, bandwidth=None, low_bias=True, device=0, ctx=None):
    """windows (B, 1, L) or (B, L) -> (psds (B, n_freqs) device tensor [V^2/Hz], freqs numpy)."""
    ctx = ctx or default_context(device)
    dev = torch.device("cuda", ctx.device)
    x = windows.to(dev, torch.float32)
    if x.dim() == 3:
        if x.shape[1] != 1:
            raise ValueError("single-channel windows expected")
        x = x[:, 0]
    x = x.contiguous()
    B, L = x.shape
    half_nbw = 4.0 if bandwidth is None else float(bandwidth) * L / (2.0 * sfreq)
    tapers, w = dpss_tapers(L, half_nbw, low_bias)
    freqs = np.fft.rfftfreq(L, 1.0 / sfreq)
    n_bins = int(np.searchsorted(freqs, fmax, side="right"))
    k0 = int(np.searchsorted(freqs, fmin, side="left"))
    td = torch.from_numpy(tapers).to(dev)
    psd = torch.empty(B, n_bins, device=dev)
    if B:
        check(lib.eegldm_psd_multitaper(ctx.h, ptr(x), ptr(td), (C.c_float * len(w))(*w.tolist()), len(w), float(sfreq), n_bins, ptr(psd), B, L))
    return psd[:, k0:], freqs[k0:n_bins]


def mean_psd_db(psds):
    """sample_trials.py:176-181: average the epochs' spectra, then 10 log10."""
    return 10.0 * torch.log10(psds.mean(dim=0))


def band_powers(psds, freqs, bands=None):
    """Trapezoidal integral of each window's PSD over the bands -> {band: (B,) tensor}."""
    out = {}
    f = torch.as_tensor(freqs, device=psds.device, dtype=psds.dtype)
    for name, (lo, hi) in (bands or BANDS).items():
        m = (f >= lo) & (f <= hi)
        out[name] = torch.trapezoid(psds[:, m], f[m], dim=1) if int(m.sum()) > 1 else torch.zeros(psds.shape[0], device=psds.device)
    return out


# ---------------------------------------------------------------------------------------------------- FID on U-Sleep features
class USleep:
    """Host-side mirror of /root/reference/src/models/usleep.py::USleep (constructor kwargs, `forward(x) -> (y_pred, x, bottom)`,
    state_dict keys), forward only, executing on libeegldm.  `train()` / `eval()` select BatchNorm on batch or running statistics; a
    freshly constructed module is in TRAIN mode like any nn.Module -- /root/reference/src/compute_fid.py:357-386 never calls
    `.eval()`, so the reference's FID features are computed with batch statistics; `fid_features` below defaults to eval mode (the
    intended use of a trained stager) and takes `batch_stats=True` to reproduce the script literally."""

    def __init__(self, in_chans=2, sfreq=128, depth=12, n_time_filters=5, complexity_factor=1.67, with_skip_connection=True, n_classes=5,
                 input_size_s=30, time_conv_size_s=9 / 128, ensure_odd_conv_size=False, apply_softmax=False, device=0, ctx=None):
        k = int(np.round(time_conv_size_s * sfreq))
        if k % 2 == 0:
            if ensure_odd_conv_size:
                k += 1
            else:
                raise ValueError("time_conv_size must be an odd number to accomodate the upsampling step in the decoder blocks.")   # usleep.py:160-163
        self.in_chans, self.depth, self.n_classes, self.apply_softmax = in_chans, depth, n_classes, apply_softmax
        self.input_size = int(np.ceil(input_size_s * sfreq))
        self.ctx = ctx or default_context(device if isinstance(device, int) else torch.device(device).index or 0)
        self.device = torch.device("cuda", self.ctx.device)
        cfg = USleepCfg(in_chans, depth, n_time_filters, n_classes, k, self.input_size, 1 if with_skip_connection else 0, float(complexity_factor))
        h = C.c_void_p()
        check(lib.eegldm_usleep_create(self.ctx.h, C.byref(cfg), C.byref(h)))
        self.h = h
        self.channels = [int(lib.eegldm_usleep_channel(h, i)) for i in range(depth + 2)]
        self.entries = {}                      # key -> (kind, offset, numel, shape), reference state_dict order
        name = C.create_string_buffer(256)
        kind, off, numel, ndim, shape = C.c_int(), C.c_long(), C.c_long(), C.c_int(), (C.c_int * 3)()
        for i in range(int(lib.eegldm_usleep_num_entries(h))):
            check(lib.eegldm_usleep_entry(h, i, name, 256, C.byref(kind), C.byref(off), C.byref(numel), C.byref(ndim), shape))
            self.entries[name.value.decode()] = (kind.value, off.value, numel.value, tuple(shape[j] for j in range(ndim.value)))
        self.flat = torch.zeros(int(lib.eegldm_usleep_num_params(h)), device=self.device)
        self.buffers = torch.zeros(int(lib.eegldm_usleep_num_buffers(h)), device=self.device)
        check(lib.eegldm_usleep_bind(h, ptr(self.flat), ptr(self.buffers)))
        self.training = True
        self.load_state_dict(self._default_init())

    def _default_init(self, generator=None):
        sd = {}
        for k, (_kind, _o, _n, shape) in self.entries.items():
            leaf = k.split(".")[-1]
            if leaf == "running_var" or (leaf == "weight" and len(shape) == 1):
                sd[k] = torch.ones(shape)
            elif len(shape) == 3:          # nn.Conv1d default: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
                sd[k] = (torch.rand(shape, generator=generator) * 2 - 1) / float(np.sqrt(shape[1] * shape[2]))
            elif leaf == "bias" and k[:-4] + "weight" in self.entries and len(self.entries[k[:-4] + "weight"][3]) == 3:
                w = self.entries[k[:-4] + "weight"][3]
                sd[k] = (torch.rand(shape, generator=generator) * 2 - 1) / float(np.sqrt(w[1] * w[2]))
            else:
                sd[k] = torch.zeros(shape, dtype=torch.int64 if leaf == "num_batches_tracked" else torch.float32)
        return sd

    def _buf(self, kind):
        return self.flat if kind == 0 else self.buffers

    def state_dict(self):
        out = {}
        for k, (kind, o, n, shape) in self.entries.items():
            v = self._buf(kind)[o:o + n].clone()
            out[k] = v.reshape(shape) if shape else v.reshape(()).round().to(torch.int64)
        return out

    def load_state_dict(self, sd, strict=True):
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        missing = [k for k in self.entries if k not in sd]; extra = [k for k in sd if k not in self.entries]
        if strict and (missing or extra):
            raise KeyError(f"state_dict mismatch: missing {missing[:4]}, unexpected {extra[:4]}")
        for k, (kind, o, n, shape) in self.entries.items():
            if k not in sd:
                continue
            v = torch.as_tensor(sd[k]).detach().to(torch.float32)
            if tuple(v.shape) != tuple(shape):
                raise ValueError(f"{k}: shape {tuple(v.shape)} != {tuple(shape)}")
            self._buf(kind)[o:o + n].copy_(v.reshape(-1).to(self.device))

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def to(self, *a, **k):
        return self

    def forward(self, x, features_only=False):
        """(B, C, T) or (B, S, C, T) -> (y_pred, decoder output, bottleneck) as usleep.py:249-287; `features_only` skips the decoder and
        returns (None, None, bottleneck)."""
        x = x.to(self.device, torch.float32)
        if x.dim() == 4:                              # (B, S, C, T) -> (B, C, S * T)
            x = x.permute(0, 2, 1, 3).flatten(start_dim=2)
        if x.dim() != 3 or x.shape[1] != self.in_chans:
            raise ValueError(f"USleep expects (B, {self.in_chans}, T) or (B, S, {self.in_chans}, T), got {tuple(x.shape)}")
        x = x.contiguous(); B, _c, T = x.shape
        Lb = T
        for _ in range(self.depth):
            Lb = (Lb + (2 if Lb % 2 else 0)) // 2
        bottom = torch.empty(B, self.channels[-1], Lb, device=self.device)
        if B == 0:
            return (None, None, bottom) if features_only else (torch.empty(0, self.n_classes, device=self.device), torch.empty(0, self.channels[1], T, device=self.device), bottom)
        if features_only:
            check(lib.eegldm_usleep_forward(self.h, ptr(x), None, None, ptr(bottom), B, T, 1 if self.training else 0))
            return None, None, bottom
        if T < self.input_size:
            raise ValueError(f"T={T} is shorter than input_size={self.input_size} (the classifier's AvgPool1d window)")
        S = T // self.input_size
        y = torch.empty(B, self.n_classes, S, device=self.device); dec = torch.empty(B, self.channels[1], T, device=self.device)
        check(lib.eegldm_usleep_forward(self.h, ptr(x), ptr(y), ptr(dec), ptr(bottom), B, T, 1 if self.training else 0))
        if self.apply_softmax:
            y = torch.softmax(y, dim=1)
        if S == 1:
            y = y[:, :, 0]
        return y, dec, bottom

    __call__ = forward

    def __del__(self):
        try:
            lib.eegldm_usleep_destroy(self.h)
        except Exception:
            pass


def fid_features(model, windows, batch_stats=False):
    """compute_fid.py:373-384: windows (B, 1, 3072) from the loader (the 36-sample pads are cropped) or (B, 1, 3000) from the sampler's
    sample_{i}.npy files -> duplicate the EEG channel into U-Sleep's two inputs -> bottleneck activation with its length-1 time axis
    squeezed: (B, c_{depth+1}) = (B, 302) for the reference configuration."""
    w = windows.to(model.device, torch.float32)
    if w.dim() != 3 or w.shape[1] != 1:
        raise ValueError(f"single-channel windows (B, 1, T) expected, got {tuple(w.shape)}")
    if w.shape[-1] == 3072:
        w = w[:, :, 36:-36]
    was = model.training
    model.train(bool(batch_stats))
    try:
        _y, _d, bottom = model.forward(torch.cat([w, w], 1), features_only=True)
    finally:
        model.train(was)
    return bottom.squeeze(-1)


class FeatureMoments:
    """Streaming mean / unbiased covariance of feature batches, accumulated on the device in fp64 (eegldm_feature_moments) -- the
    reference concatenates every batch's features in host memory first (compute_fid.py:371-388)."""

    def __init__(self, dim, device=0, ctx=None):
        self.ctx = ctx or default_context(device)
        self.dim, self.n = int(dim), 0
        dev = torch.device("cuda", self.ctx.device)
        self.sum = torch.zeros(self.dim, dtype=torch.float64, device=dev); self.outer = torch.zeros(self.dim, self.dim, dtype=torch.float64, device=dev)

    def update(self, feats):
        f = feats.to(self.sum.device, torch.float32).contiguous()
        if f.dim() != 2 or f.shape[1] != self.dim:
            raise ValueError("Inputs should have (number images, number of features) shape.")
        if f.shape[0]:
            check(lib.eegldm_feature_moments(self.ctx.h, ptr(f), f.shape[0], self.dim, ptr(self.sum), ptr(self.outer)))
            self.n += int(f.shape[0])
        return self

    def finalize(self):
        """(mean (D,), covariance (D, D)) as float64 numpy arrays."""
        if self.n < 2:
            raise ValueError("at least two feature vectors are needed for a covariance")
        s = self.sum.cpu().numpy(); o = self.outer.cpu().numpy()
        mu = s / self.n
        return mu, (o - self.n * np.outer(mu, mu)) / (self.n - 1)


def frechet_distance(mu_x, sigma_x, mu_y, sigma_y):
    """|mu_x - mu_y|^2 + tr(S_x) + tr(S_y) - 2 tr((S_x S_y)^(1/2)).  The trace of the square root is the sum of the square roots of the
    eigenvalues of S_x S_y, which are those of the SYMMETRIC positive semi-definite matrix S_x^(1/2) S_y S_x^(1/2): two `eigh` calls in
    fp64 instead of a general (complex, ill-conditioned for rank-deficient covariances: 64 synthetic windows against 302 features in
    compute_fid.py:405) matrix square root."""
    mu_x, mu_y = np.asarray(mu_x, np.float64), np.asarray(mu_y, np.float64)
    sx, sy = np.asarray(sigma_x, np.float64), np.asarray(sigma_y, np.float64)
    w, v = np.linalg.eigh((sx + sx.T) / 2)
    root = (v * np.sqrt(np.clip(w, 0.0, None))) @ v.T
    m = root @ ((sy + sy.T) / 2) @ root
    ev = np.linalg.eigvalsh((m + m.T) / 2)
    d = mu_x - mu_y
    return float(d @ d + np.trace(sx) + np.trace(sy) - 2.0 * np.sqrt(np.clip(ev, 0.0, None)).sum())


class FIDMetric:
    """monai-generative's `FIDMetric()(y_pred, y)` as compute_fid.py:412-414 calls it: (N, D) feature tensors -> scalar tensor."""

    def __init__(self, device=0, ctx=None):
        self.ctx = ctx or default_context(device)

    def __call__(self, y_pred, y):
        if y_pred.dim() > 2 or y.dim() > 2:
            raise ValueError("Inputs should have (number images, number of features) shape.")
        a = FeatureMoments(y_pred.shape[1], ctx=self.ctx).update(y_pred).finalize()
        b = FeatureMoments(y.shape[1], ctx=self.ctx).update(y).finalize()
        return torch.tensor(frechet_distance(a[0], a[1], b[0], b[1]), dtype=torch.float64)
