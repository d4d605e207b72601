"""Task heads of VlmoModule (models/vlmo/heads.py:86-138) with identical parameter names.

The two vocabulary heads -- the MLM decoder tied to the word embedding (768 -> 30 522) and the MIM head
(768 -> 8 192 visual tokens) -- have a fused loss path (``loss_and_pred``): a HIP GEMM whose epilogue keeps per-row
running (max, sum exp, arg-max, label logit) per 64-column chunk, so the [rows, vocabulary] logits never reach HBM, and
a backward that recomputes (softmax - onehot) tile-wise into the bf16 operand of the input- and weight-gradient GEMMs
(VLMO_EPI_CE / VLMO_EPI_CE_BWD, csrc/gemm.hip).  ``forward`` still returns the logits (reference contract:
`mlm_logits` / `mim_logits` in the output dict); ``config.train.fused_ce`` selects the fused path in the objectives.
The small heads (ITC projection + normalise, 2-way ITM, pooler) are a few MFLOP and stay torch ops."""
import torch
import torch.nn as nn

from . import hip


class _PaddedShadows:
    """bf16 copies of a vocabulary head's weight padded to a multiple of 64 rows (W [Vp, d], W^T [d, Vp]) and its fp32
    bias padded with -1e30 (so padded columns vanish from the soft-max), refreshed when a version counter changes."""

    def __init__(self):
        self.key, self.val = None, None

    def get(self, weight, bias):
        key = (weight._version, weight.data_ptr(), None if bias is None else (bias._version, bias.data_ptr()))
        if self.key != key:
            V, d = weight.shape
            Vp = (V + 63) // 64 * 64
            w = torch.zeros((Vp, d), dtype=torch.bfloat16, device=weight.device)
            w[:V] = weight.detach()
            wt = w.t().contiguous()
            b = torch.full((Vp,), -1e30, dtype=torch.float32, device=weight.device)
            b[:V] = bias.detach() if bias is not None else 0.0
            self.key, self.val = key, (w, wt, b, Vp)
        return self.val


class LinearCrossEntropyFn(torch.autograd.Function):
    """mean cross-entropy of ``x @ W^T + b`` against ``labels`` (rows with ``ignore_index`` excluded) and the
    arg-max prediction per row, without materialising the logits (heads.py:86-112 + objectives.py:57-68,571-582)."""

    @staticmethod
    def forward(ctx, x, weight, bias, labels, ignore_index, shadows):
        n, d = x.shape
        V = weight.shape[0]
        w, wt, b, Vp = shadows.get(weight, bias)
        xb = x.detach().to(torch.bfloat16).contiguous()
        lab = labels.to(torch.int32).contiguous()
        nch = Vp // 64
        dev = x.device
        part = torch.empty((n, nch, 4), dtype=torch.float32, device=dev)
        hip.gemm_nt(hip.EPI_CE, xb, w, n, Vp, d, part, bias=b, row_index=lab, ldo=nch)
        lse = torch.empty(n, dtype=torch.float32, device=dev)
        rows = torch.empty(n, dtype=torch.float32, device=dev)
        pred = torch.empty(n, dtype=torch.int32, device=dev)
        hip.ce_reduce(part, nch, lab, ignore_index, lse, rows, pred, n)
        valid = labels != ignore_index
        nvalid = valid.sum().clamp(min=1).to(torch.float32)
        ctx.save_for_backward(xb, lab, lse, valid, nvalid, w, wt, b)
        ctx.dims = (n, d, V, Vp, bias is not None)
        ctx.in_dtypes = (x.dtype, weight.dtype, None if bias is None else bias.dtype)
        ctx.mark_non_differentiable(pred)
        return rows.sum() / nvalid, pred

    @staticmethod
    def backward(ctx, dloss, _dpred):
        xb, lab, lse, valid, nvalid, w, wt, b = ctx.saved_tensors
        n, d, V, Vp, has_bias = ctx.dims
        dev = xb.device
        rs = (valid.to(torch.float32) * (dloss.to(torch.float32) / nvalid)).contiguous()
        dlog = torch.empty((n, Vp), dtype=torch.bfloat16, device=dev)
        hip.gemm_nt(hip.EPI_CE_BWD, xb, w, n, Vp, d, dlog, bias=b, resid=lse, row_scale=rs, row_index=lab)
        dx = torch.empty((n, d), dtype=torch.float32, device=dev)
        hip.gemm_nt(hip.EPI_F32, dlog, wt, n, d, Vp, dx)
        dw = torch.zeros((Vp, d), dtype=torch.float32, device=dev)
        hip.gemm_tn(dlog, xb, dw, n, Vp, d)
        db = None
        if has_bias:
            db = torch.zeros(Vp, dtype=torch.float32, device=dev)
            hip.colsum(dlog, db, n, Vp)
            db = db[:V]
        # autograd wants every gradient in its input's dtype: under torch.autocast the activations that reach this head may
        # be half precision (multimodal.py:276-279 runs the module inside autocast)
        xd, wd, bd = ctx.in_dtypes
        return dx.to(xd), dw[:V].to(wd), (db.to(bd) if db is not None else None), None, None, None


class BertPredictionHeadTransform(nn.Module):
    """transformers BertPredictionHeadTransform: dense -> GELU -> LayerNorm(eps 1e-12)."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)

    def forward(self, hidden_states):
        return self.LayerNorm(nn.functional.gelu(self.dense(hidden_states)))


class MLMHead(nn.Module):
    """heads.py:86-101 (decoder weight tied to the word embedding, vlmo_module.py:53-55)."""

    def __init__(self, config, weight=None):
        super().__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.bias = nn.parameter.Parameter(torch.zeros(config.vocab_size))
        if weight is not None:
            self.decoder.weight = weight

    def forward(self, x):
        return self.decoder(self.transform(x)) + self.bias

    def loss_and_pred(self, x, labels, ignore_index=-100):
        """fused decoder + cross-entropy: (mean loss over rows whose label is not ignore_index, arg-max [n])."""
        if not hasattr(self, '_ce_shadows'):
            object.__setattr__(self, '_ce_shadows', _PaddedShadows())
        return LinearCrossEntropyFn.apply(self.transform(x), self.decoder.weight, self.bias, labels, ignore_index,
                                          self._ce_shadows)


class MIMHead(nn.Module):
    """heads.py:104-112."""

    def __init__(self, hidden_size, vocab_size):
        super().__init__()
        self.fc = nn.Linear(hidden_size, vocab_size)

    def forward(self, x):
        return self.fc(x)

    def loss_and_pred(self, x, labels, ignore_index=-100):
        if not hasattr(self, '_ce_shadows'):
            object.__setattr__(self, '_ce_shadows', _PaddedShadows())
        return LinearCrossEntropyFn.apply(x, self.fc.weight, self.fc.bias, labels, ignore_index, self._ce_shadows)


class ITCHead(nn.Module):
    """heads.py:115-127."""

    def __init__(self, hidden_size, out_size):
        super().__init__()
        self.dense = nn.ModuleDict({'v': nn.Linear(hidden_size, out_size), 'l': nn.Linear(hidden_size, out_size)})

    def forward(self, hidden_states, route=None):
        return nn.functional.normalize(self.dense[route](hidden_states), dim=-1)


class ITMHead(nn.Module):
    """heads.py:130-138."""

    def __init__(self, hidden_size):
        super().__init__()
        self.fc = nn.Linear(hidden_size, 2)

    def forward(self, x):
        return self.fc(x)
