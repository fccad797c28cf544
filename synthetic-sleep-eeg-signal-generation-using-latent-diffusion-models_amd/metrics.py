"""Quality metrics of reconstructed / generated windows on the device, with the interfaces the reference's
evaluation code uses:

* `MultiScaleSSIMMetric(spatial_dims=1, data_range=1.0, kernel_size=7)(y_pred, y) -> (B, 1)`
  -- /root/reference/src/compute_mmds.py:297-408 (the reference's local 1-D adaptation of MONAI's metric), call site :487-503.
* `compute_psd(windows, sfreq=100, fmax=18) -> (psds (B, n_freqs), freqs)` and `mean_psd_db`
  -- mne `Epochs.compute_psd(fmax=18)` + average + 10 log10 as used at /root/reference/src/sample_trials.py:172-181
  (multitaper, DPSS half-bandwidth 4, low-bias tapers, normalization "length").  The DPSS tapers (a few KB, once per window
  length) come from scipy on the host; everything per window runs in libeegldm (eegldm_psd_multitaper).
* `band_powers`: integrates a PSD over the classical sleep-EEG bands.
"""
import ctypes as C
import functools

import numpy as np
import torch

from ._lib import lib, check, ptr, default_context

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
