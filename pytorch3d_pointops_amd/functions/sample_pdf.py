"""sample_pdf -- same API as the reference's functions/sample_pdf.py (:14-66 sample_pdf,
:69-148 sample_pdf_python).  Like the reference, not re-exported from functions/__init__."""
import torch

from .. import _C


def sample_pdf(
    bins: torch.Tensor,
    weights: torch.Tensor,
    n_samples: int,
    det: bool = False,
    eps: float = 1e-5,
) -> torch.Tensor:
    """Draw `n_samples` per row from the piecewise-constant PDFs given by `bins` (..., n_bins+1)
    and non-negative `weights` (..., n_bins); same contract as the reference (functions/sample_pdf.py:14-66).
    `det=True` uses uniformly spaced quantiles, otherwise torch.rand (same RNG consumption)."""
    if torch.is_grad_enabled() and (bins.requires_grad or weights.requires_grad):
        raise NotImplementedError("sample_pdf differentiability.")
    if weights.min() <= -eps:
        raise ValueError("Negative weights provided.")
    batch_shape = bins.shape[:-1]
    n_bins = weights.shape[-1]
    if n_bins + 1 != bins.shape[-1] or weights.shape[:-1] != batch_shape:
        shapes = f"{bins.shape}{weights.shape}"
        raise ValueError("Inconsistent shapes of bins and weights: " + shapes)
    output_shape = batch_shape + (n_samples,)

    if det:
        u = torch.linspace(0.0, 1.0, n_samples, device=bins.device, dtype=torch.float32)
        output = u.expand(output_shape).contiguous()
    else:
        output = torch.rand(output_shape, dtype=torch.float32, device=bins.device)

    _C.sample_pdf(bins.reshape(-1, n_bins + 1), weights.reshape(-1, n_bins), output.reshape(-1, n_samples), eps)
    return output


def sample_pdf_python(
    bins: torch.Tensor,
    weights: torch.Tensor,
    N_samples: int,
    det: bool = False,
    eps: float = 1e-5,
) -> torch.Tensor:
    """Pure-torch variant (searchsorted on the normalised CDF); any device.
    reference: functions/sample_pdf.py:69-148."""
    weights = weights + eps
    if weights.min() <= 0:
        raise ValueError("Negative weights provided.")
    pdf = weights / weights.sum(dim=-1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if det:
        u = torch.linspace(0.0, 1.0, N_samples, device=cdf.device, dtype=cdf.dtype)
        u = u.expand(list(cdf.shape[:-1]) + [N_samples]).contiguous()
    else:
        u = torch.rand(list(cdf.shape[:-1]) + [N_samples], device=cdf.device, dtype=cdf.dtype)
    inds = torch.searchsorted(cdf, u, right=True)
    below = (inds - 1).clamp(0)
    above = inds.clamp(max=cdf.shape[-1] - 1)
    inds_g = torch.stack([below, above], -1).view(*below.shape[:-1], below.shape[-1] * 2)
    cdf_g = torch.gather(cdf, -1, inds_g).view(*below.shape, 2)
    bins_g = torch.gather(bins, -1, inds_g).view(*below.shape, 2)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom = torch.where(denom < eps, torch.ones_like(denom), denom)
    t = (u - cdf_g[..., 0]) / denom
    return bins_g[..., 0] + t * (bins_g[..., 1] - bins_g[..., 0])
