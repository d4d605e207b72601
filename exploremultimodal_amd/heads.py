"""Task heads of VlmoModule (models/vlmo/heads.py:86-138) with identical parameter names.

Small GEMMs on a few thousand gathered rows; they run as stock torch ops on the device
(bf16 autocast is the caller's choice).  SURVEY.md section 8f ranks fusing them next."""
import torch
import torch.nn as nn


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


class MIMHead(nn.Module):
    """heads.py:104-112."""

    def __init__(self, hidden_size, vocab_size):
        super().__init__()
        self.fc = nn.Linear(hidden_size, vocab_size)

    def forward(self, x):
        return self.fc(x)


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
