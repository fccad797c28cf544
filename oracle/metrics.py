"""ORACLE (test infrastructure only -- never imported by the product path).

* ms_ssim_1d: line-by-line torch restatement of the reference's LOCAL 1-D MS-SSIM
  (/root/reference/src/compute_mmds.py: _gaussian_kernel :172-211 with the 2-D product commented out at :197-200,
  compute_ssim_and_cs :214-276, MultiScaleSSIMMetric._compute_metric :336-402).  The source is in the reference tree, so this
  restatement is checked against it by reading; it cannot be imported here (the module's top-level imports need monai /
  generative, which are absent), hence no golden vectors: PARITY UNPINNED beyond the closed forms in the tests
  (identical inputs -> 1; SSIM of a constant offset; symmetric in its arguments).
* psd_multitaper: numpy restatement of mne.time_frequency.psd_array_multitaper as called through Epochs.compute_psd(fmax=18)
  (/root/reference/src/sample_trials.py:172-181); mne is absent: PARITY UNPINNED (closed form: Parseval / a pure sinusoid's peak).
"""
import numpy as np
import torch
import torch.nn.functional as F


def gaussian_kernel_1d(kernel_size, sigma):
    dist = torch.arange(start=(1 - kernel_size) / 2, end=(1 + kernel_size) / 2, step=1)
    gauss = torch.exp(-torch.pow(dist / sigma, 2) / 2)
    return (gauss / gauss.sum()).unsqueeze(dim=0)           # (1, k)


def ssim_and_cs(y_pred, y, data_range=1.0, kernel_size=11, kernel_sigma=1.5, k1=0.01, k2=0.03):
    C = y_pred.size(1)
    kernel = gaussian_kernel_1d(kernel_size, kernel_sigma).expand(C, 1, kernel_size).to(y_pred)
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    mu_x = F.conv1d(y_pred, kernel, groups=C); mu_y = F.conv1d(y, kernel, groups=C)
    mu_xx = F.conv1d(y_pred * y_pred, kernel, groups=C); mu_yy = F.conv1d(y * y, kernel, groups=C); mu_xy = F.conv1d(y_pred * y, kernel, groups=C)
    sigma_x, sigma_y, sigma_xy = mu_xx - mu_x * mu_x, mu_yy - mu_y * mu_y, mu_xy - mu_x * mu_y
    cs = (2 * sigma_xy + c2) / (sigma_x + sigma_y + c2)
    ssim = ((2 * mu_x * mu_y + c1) / (mu_x ** 2 + mu_y ** 2 + c1)) * cs
    return ssim, cs


def ms_ssim_1d(y_pred, y, data_range=1.0, kernel_size=11, kernel_sigma=1.5, k1=0.01, k2=0.03,
               weights=(0.0448, 0.2856, 0.3001, 0.2363, 0.1333)):
    w = torch.tensor(weights, dtype=torch.float)
    y_pred, y = y_pred.float(), y.float()
    ms = []
    for _ in range(len(w)):
        ssim, cs = ssim_and_cs(y_pred, y, data_range, kernel_size, kernel_sigma, k1, k2)
        ms.append(torch.relu(cs.view(cs.shape[0], -1).mean(1)))
        y_pred = F.avg_pool1d(y_pred, kernel_size=2); y = F.avg_pool1d(y, kernel_size=2)
    ms[-1] = torch.relu(ssim.view(ssim.shape[0], -1).mean(1))
    ms = torch.stack(ms)
    val = torch.prod(ms ** w.view(-1, 1), dim=0)
    return val.view(val.shape[0], -1).mean(1, keepdim=True)


def psd_multitaper(x, sfreq=100.0, fmax=18.0, half_nbw=4.0):
    """x (B, L) float -> (psd (B, n_freqs), freqs), float64 arithmetic."""
    from scipy.signal.windows import dpss
    x = np.asarray(x, np.float64)
    L = x.shape[-1]
    tapers, ratios = dpss(L, half_nbw, int(2 * half_nbw), sym=False, norm=2, return_ratios=True)
    keep = ratios > 0.9
    tapers, wts = tapers[keep], np.sqrt(ratios[keep])
    x = x - x.mean(-1, keepdims=True)
    x_mt = np.fft.rfft(x[:, None, :] * tapers[None], axis=-1)            # (B, K, F)
    x_mt[..., 0] /= np.sqrt(2.0)
    if L % 2 == 0:
        x_mt[..., -1] /= np.sqrt(2.0)
    psd = (np.abs(wts[None, :, None] * x_mt) ** 2).sum(1) * 2.0 / (wts ** 2).sum()
    psd /= sfreq
    freqs = np.fft.rfftfreq(L, 1.0 / sfreq)
    m = freqs <= fmax
    return psd[:, m], freqs[m]
