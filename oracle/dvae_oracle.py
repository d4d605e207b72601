"""Plain-PyTorch fp32 restatement of the dall_e dVAE encoder (TEST INFRASTRUCTURE).

Follows dall_e/encoder.py:13-133 (EncoderBlock :13-46, Encoder :49-133),
dall_e/utils.py:11-48 (Conv2d: same padding (kw-1)//2, params ``w``/``b``) and
models/modeling_discrete_vae.py:246-248 (get_codebook_indices = argmax dim 1).
State-dict keys are the reference's (``blocks.group_1.block_1.res_path.conv_1.w``).
"""
import torch
import torch.nn.functional as F


def _conv(sd, p, x):
    w = sd[p + '.w']
    return F.conv2d(x, w, sd[p + '.b'], padding=(w.shape[-1] - 1) // 2)


def encoder_block(sd, p, x, post_gain):
    """encoder.py:24-46: id_path(x) + post_gain * res_path(x)."""
    idp = _conv(sd, p + '.id_path', x) if (p + '.id_path.w') in sd else x
    r = x
    for i in (1, 2, 3, 4):
        r = _conv(sd, p + f'.res_path.conv_{i}', F.relu(r))
    return idp + post_gain * r


def encoder(sd, x, group_count=4, n_blk_per_group=2):
    """encoder.py:123-133 (+ the ValueError checks)."""
    if x.dim() != 4:
        raise ValueError(f'input shape {x.shape} is not 4d')
    if x.shape[1] != sd['blocks.input.w'].shape[1]:
        raise ValueError('input channel mismatch')
    if x.dtype != torch.float32:
        raise ValueError('input must have dtype torch.float32')
    post_gain = 1.0 / (group_count * n_blk_per_group) ** 2
    x = _conv(sd, 'blocks.input', x)
    for g in range(1, group_count + 1):
        for b in range(1, n_blk_per_group + 1):
            x = encoder_block(sd, f'blocks.group_{g}.block_{b}', x, post_gain)
        if g < group_count:
            x = F.max_pool2d(x, 2)
    return _conv(sd, 'blocks.output.conv', F.relu(x))


def get_codebook_indices(sd, images, **kw):
    return torch.argmax(encoder(sd, images, **kw), dim=1)
