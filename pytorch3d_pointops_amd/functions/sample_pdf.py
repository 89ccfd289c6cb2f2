"""sample_pdf -- the API of the reference's functions/sample_pdf.py (:14-66 `sample_pdf`, :69-148
`sample_pdf_python`) on the HIP inverse-CDF kernel (csrc/sample_pdf.hip).  Like the reference, not
re-exported from functions/__init__.
"""
import torch

from .. import _C


def _quantiles(batch_shape, n_samples: int, det: bool, device) -> torch.Tensor:
    """(..., n_samples) fp32 quantiles in [0, 1]: evenly spaced (det) or torch.rand -- the reference draws
    them the same way (:53-57), so a seeded generator yields the same samples."""
    shape = tuple(batch_shape) + (n_samples,)
    if not det:
        return torch.rand(shape, dtype=torch.float32, device=device)
    return torch.linspace(0.0, 1.0, n_samples, dtype=torch.float32, device=device).expand(shape).contiguous()


def sample_pdf(
    bins: torch.Tensor,
    weights: torch.Tensor,
    n_samples: int,
    det: bool = False,
    eps: float = 1e-5,
) -> torch.Tensor:
    """Draw `n_samples` values per row from the piecewise-constant densities with bin edges `bins`
    (..., n_bins + 1) and non-negative bin masses `weights` (..., n_bins); `eps` is added to every mass
    (empty bins).  Returns (..., n_samples).  Not differentiable (raises NotImplementedError when a gradient
    is requested, as the reference does)."""
    if torch.is_grad_enabled() and (bins.requires_grad or weights.requires_grad):
        raise NotImplementedError("sample_pdf differentiability.")
    if weights.min() <= -eps:
        raise ValueError("Negative weights provided.")
    n_bins = weights.shape[-1]
    if bins.shape[-1] != n_bins + 1 or bins.shape[:-1] != weights.shape[:-1]:
        raise ValueError("Inconsistent shapes of bins and weights: " + f"{bins.shape}{weights.shape}")
    out = _quantiles(bins.shape[:-1], n_samples, det, bins.device)
    # the kernel turns the quantiles into samples in place
    _C.sample_pdf(bins.reshape(-1, n_bins + 1), weights.reshape(-1, n_bins), out.view(-1, n_samples), eps)
    return out


def sample_pdf_python(
    bins: torch.Tensor,
    weights: torch.Tensor,
    N_samples: int,
    det: bool = False,
    eps: float = 1e-5,
) -> torch.Tensor:
    """The same sampling in plain torch ops (any device): normalised CDF, `searchsorted`, linear
    interpolation inside the hit bin.  Mirrors the reference's fallback (:69-148) in behaviour."""
    mass = weights + eps
    if mass.min() <= 0:
        raise ValueError("Negative weights provided.")
    cdf = torch.cumsum(mass / mass.sum(-1, keepdim=True), dim=-1)
    cdf = torch.nn.functional.pad(cdf, (1, 0))  # leading 0: one CDF value per bin edge
    u = _quantiles(cdf.shape[:-1], N_samples, det, cdf.device).to(cdf.dtype)
    upper = torch.searchsorted(cdf, u, right=True).clamp(max=cdf.shape[-1] - 1)  # edge above the quantile
    lower = (upper - 1).clamp(min=0)
    c_lo, c_hi = cdf.gather(-1, lower), cdf.gather(-1, upper)
    b_lo, b_hi = bins.gather(-1, lower), bins.gather(-1, upper)
    width = c_hi - c_lo
    width = torch.where(width < eps, torch.ones_like(width), width)
    return b_lo + (u - c_lo) / width * (b_hi - b_lo)
