"""ORACLE (test infrastructure) -- the Frechet distance the reference computes with monai-generative's FIDMetric
(/root/reference/src/compute_fid.py:412-414: `FIDMetric()(samples_features, synthetic_features)`).

monai-generative is an un-pinned dependency (requirements.txt:12) whose source is absent: this restates its published algorithm
(generative/metrics/fid.py, 0.2.x) -- "parity unpinned", checked by closed forms in tests/test_oracle_usleep_fid.py:
    y, y_pred -> float64;  mu = mean over the batch;  sigma = cov(rowvar=False), unbiased (N - 1);
    covmean = scipy.linalg.sqrtm(sigma_pred @ sigma);  when not finite: add eps = 1e-6 on both diagonals and retry;
    a complex result keeps its real part (imaginary diagonal must be ~0);
    FID = |mu_pred - mu|^2 + tr(sigma_pred) + tr(sigma) - 2 tr(covmean)."""
import numpy as np
import scipy.linalg
import torch


def _cov(x):
    x = x.t()
    x = x - x.mean(dim=1, keepdim=True)
    return x.matmul(x.t()) / (x.shape[1] - 1)


def _sqrtm(m):
    r = scipy.linalg.sqrtm(m.detach().cpu().numpy().astype(np.float64), disp=False)
    r = r[0] if isinstance(r, tuple) else r
    return torch.from_numpy(np.asarray(r))


def frechet_distance(mu_x, sigma_x, mu_y, sigma_y, epsilon=1e-6):
    diff = mu_x - mu_y
    covmean = _sqrtm(sigma_x.mm(sigma_y))
    if not torch.isfinite(covmean).all():
        off = torch.eye(sigma_x.shape[0], dtype=sigma_x.dtype) * epsilon
        covmean = _sqrtm((sigma_x + off).mm(sigma_y + off))
    if torch.is_complex(covmean):
        if not torch.allclose(torch.diagonal(covmean).imag, torch.tensor(0, dtype=torch.double), atol=1e-3):
            raise ValueError(f"Imaginary component {torch.max(torch.abs(covmean.imag))} too high.")
        covmean = covmean.real
    return diff.dot(diff) + torch.trace(sigma_x) + torch.trace(sigma_y) - 2 * torch.trace(covmean)


def fid(y_pred, y):
    y_pred, y = y_pred.double(), y.double()
    if y.ndimension() > 2:
        raise ValueError("Inputs should have (number images, number of features) shape.")
    return frechet_distance(y_pred.mean(0), _cov(y_pred), y.mean(0), _cov(y))
