"""-m gpu: device quality metrics against the oracle restatements (oracle/metrics.py) and closed forms."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from param_gen import eeg_windows, normal  # noqa: E402


@pytest.mark.parametrize("ks,L,C", [(7, 3000, 1), (11, 3000, 1), (16, 3072, 1), (7, 375, 2), (3, 48, 1)])
def test_ms_ssim_1d_matches_oracle(ks, L, C):
    """compute_mmds.py:487-503: MultiScaleSSIMMetric(spatial_dims=1, data_range=1.0, kernel_size=7) on cropped windows."""
    import gpu_util as G
    from eegldm.metrics import MultiScaleSSIMMetric
    from oracle.metrics import ms_ssim_1d
    B = 5
    pad = 36 if L == 3072 else 0
    x = torch.from_numpy(eeg_windows(B * C, seed=1, length=L + 2 * (36 - pad), pad=36))[:, :, 36 - pad:36 - pad + L].reshape(B, C, L).contiguous()
    y = (x + 0.03 * torch.from_numpy(normal((B, C, L), seed=2))).clamp(0, 1)
    y[0] = x[0]                                     # identical pair -> exactly 1
    y[1] = 1.0 - x[1]                               # anti-correlated: negative cs clamps to 0 (torch.relu at compute_mmds.py:386)
    m = MultiScaleSSIMMetric(spatial_dims=1, data_range=1.0, kernel_size=ks)
    got = m(x, y)
    want = ms_ssim_1d(x, y, kernel_size=ks)
    assert got.shape == (B, 1)
    G.assert_close(got, want, rtol=2e-4, atol=2e-5, name="ms-ssim")
    assert abs(float(got[0]) - 1.0) < 1e-5 and float(got[1]) == 0.0
    G.assert_close(m(y, x), got, rtol=1e-5, atol=1e-6, name="symmetry")
    with pytest.raises(ValueError, match="must be larger than"):
        m(x[:, :, :ks * 16 - 16], y[:, :, :ks * 16 - 16])


def test_psd_multitaper_matches_oracle_and_closed_forms():
    """sample_trials.py:172-181: compute_psd(fmax=18) -> average -> 10 log10, on (B,1,3000) windows at 100 Hz."""
    import gpu_util as G
    from eegldm.metrics import band_powers, compute_psd, mean_psd_db
    from oracle.metrics import psd_multitaper
    B, L = 6, 3000
    x = torch.from_numpy(eeg_windows(B, seed=3, length=3072))[:, :, 36:-36].contiguous() * 1e-4
    t = np.arange(L) / 100.0
    x[0, 0] = torch.from_numpy((3e-5 * np.sin(2 * np.pi * 10.0 * t)).astype(np.float32))       # pure 10 Hz alpha
    psd, freqs = compute_psd(x, sfreq=100.0, fmax=18.0)
    want, wf = psd_multitaper(x[:, 0].numpy(), 100.0, 18.0)
    assert psd.shape == want.shape == (B, 541) and np.allclose(freqs, wf)
    assert G.rel_l2(psd, want) < 2e-4, G.rel_l2(psd, want)
    G.assert_close(psd, want, rtol=2e-3, atol=1e-4 * float(want.max()), name="psd")
    # closed forms: the sinusoid's spectrum peaks at 10 Hz and its alpha-band power is A^2/2 (the tapers spread it over +-4/30 Hz)
    assert abs(freqs[int(psd[0].argmax())] - 10.0) < 0.15
    bp = band_powers(psd, freqs)
    assert abs(float(bp["alpha"][0]) - 0.5 * 3e-5 ** 2) < 0.03 * 0.5 * 3e-5 ** 2
    assert float(bp["delta"][0]) < 1e-3 * float(bp["alpha"][0])
    db = mean_psd_db(psd)
    assert db.shape == (541,) and torch.isfinite(db[1:]).all()
    G.assert_close(db[1:], 10 * np.log10(want.mean(0))[1:], rtol=0, atol=2e-2, name="mean dB")
