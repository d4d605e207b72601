"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): fp64 restatement of the optimizer step the reference's loop
performs -- unscale + clip_grad_norm_ (utils/utils.py:343-364) followed by apex FusedAdam(adam_w_mode=True)
(utils/optim_factory.py:185-186; same update as torch.optim.AdamW).  Pinned in tests against torch.optim.AdamW /
torch.nn.utils.clip_grad_norm_ themselves (the reference's apex is not installed here)."""
import numpy as np


def clip_coef(grads, max_norm, inv_scale=1.0):
    """-> (norm, factor to apply to the raw gradients): torch.nn.utils.clip_grad_norm_ semantics."""
    norm = float(np.sqrt(sum(float((np.asarray(g, np.float64) ** 2).sum()) for g in grads))) * inv_scale
    coef = inv_scale
    if max_norm is not None and max_norm > 0:
        coef *= min(1.0, max_norm / (norm + 1e-6))
    return norm, coef


def adam_step(p, g, m, v, step, lr, beta1, beta2, eps, wd, adam_w_mode=True, bias_correction=True):
    """One update of one tensor, all fp64; returns (p, m, v)."""
    p, g, m, v = (np.asarray(x, np.float64) for x in (p, g, m, v))
    if not adam_w_mode:
        g = g + wd * p
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step if bias_correction else 1.0
    bc2 = 1 - beta2 ** step if bias_correction else 1.0
    u = (m / bc1) / (np.sqrt(v / bc2) + eps)
    if adam_w_mode:
        u = u + wd * p
    return p - lr * u, m, v
