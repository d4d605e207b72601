"""Per-kernel parity of the C-ABI HIP kernels (through exploremultimodal_amd.hip)
against plain PyTorch fp32 math of the same op on the same bf16-rounded inputs.
Tolerances (stated per test): bf16 outputs 2^-8 relative (one rounding) plus the
fp32-accumulation noise; fp32 outputs 1e-3 relative to the row scale."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from exploremultimodal_amd import hip  # noqa: E402

DEV = 'cuda'


def _rand(*shape, scale=1.0, seed=0, dtype=torch.bfloat16):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dtype)


def _close(got, ref, rtol, atol, what=''):
    got, ref = got.float(), ref.float()
    err = (got - ref).abs()
    lim = atol + rtol * ref.abs()
    bad = err > lim
    if bad.any():
        idx = bad.nonzero()[0].tolist()
        raise AssertionError(
            f'{what}: {int(bad.sum())}/{bad.numel()} mismatches, max err {err.max().item():.4g} '
            f'first at {idx}: got {got[tuple(idx)].item():.6g} ref {ref[tuple(idx)].item():.6g}')


@pytest.mark.parametrize('tile', [0, 3, 4])
@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (300, 256, 128), (1000, 384, 768), (77, 128, 192)])
def test_gemm_nt_exact_integers(tile, M, N, K):
    """Asymmetric small-integer operands: products and fp32 sums are exact, so any
    fragment-layout or swizzle error shows up as a wrong integer."""
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    B = torch.randint(-2, 3, (N, K), generator=g).float()
    B[:, 0] += torch.arange(N).float() % 5     # break symmetries
    ref = A @ B.t()
    Ad, Bd = A.to(DEV).bfloat16(), B.to(DEV).bfloat16()
    out = torch.full((M, N), float('nan'), device=DEV)
    hip.gemm_nt(hip.EPI_F32, Ad, Bd, M, N, K, out, tile=tile)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), ref), (out.cpu() - ref).abs().max()


@pytest.mark.parametrize('tile', [0, 3, 4])
def test_gemm_nt_bias_bf16_and_padding_rows(tile):
    M, N, K = 333, 256, 256
    A, B = _rand(M, K, seed=1), _rand(N, K, scale=0.1, seed=2)
    bias = _rand(N, seed=3, dtype=torch.float32)
    out = torch.full((M + 5, N), 7.0, device=DEV, dtype=torch.bfloat16)
    hip.gemm_nt(hip.EPI_BIAS, A, B, M, N, K, out, bias=bias, tile=tile)
    ref = A.float() @ B.float().t() + bias
    _close(out[:M], ref, 1 / 128, 1e-2, 'bias epilogue')
    assert (out[M:] == 7.0).all(), 'rows beyond M were written'
    out2 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    hip.gemm_nt(hip.EPI_BIAS, A, B, M, N, K, out2, bias=bias, relu=True, tile=tile)
    _close(out2, ref.relu(), 1 / 128, 1e-2, 'bias+relu epilogue')


@pytest.mark.parametrize('tile', [4, 8, 106, 110, 309, 310, 311, 312, 313, 315, 317, 319, 320])
@pytest.mark.parametrize('M,N,K', [(1000, 384, 768), (4099, 2304, 768), (77, 128, 192)])
def test_gemm_nt_tile4_small_integers(tile, M, N, K):
    """The 256x128x32 and 192x256x64 tiles and the 16x16x32-MFMA tiles 309..320 ((16 * (tile - 300)) x 256, odd heights = wave groups of different size; 106..110 = 32 * (tile - 100) rows: a new fragment
    layout, a new accumulator -> LDS map, staging with a ragged last instruction at 224 / 288 rows) exist for a few
    epilogues only (bf16 output): integer operands small enough
    that every output is an integer below 256 in magnitude, i.e. exact in bf16 -- a fragment-layout or swizzle error of
    this tile shape shows up as a wrong integer, ragged edges in M and N included."""
    g = torch.Generator().manual_seed(M + N)
    A = (torch.rand(M, K, generator=g) < 0.05).float() * torch.randint(-2, 3, (M, K), generator=g).float()
    B = (torch.rand(N, K, generator=g) < 0.05).float() * torch.randint(-2, 3, (N, K), generator=g).float()
    B[:, 0] = (torch.arange(N) % 3).float()
    A[:, 0] = (torch.arange(M) % 2).float() + 1
    bias = (torch.arange(N) % 7 - 3).float()
    ref = A @ B.t() + bias
    assert ref.abs().max() < 256
    out = torch.full((M, N), float('nan'), device=DEV, dtype=torch.bfloat16)
    hip.gemm_nt(hip.EPI_BIAS, A.to(DEV).bfloat16(), B.to(DEV).bfloat16(), M, N, K, out, bias=bias.to(DEV), tile=tile)
    assert torch.equal(out.float().cpu(), ref), (out.float().cpu() - ref).abs().max()


def test_gemm_nt_f16_dtype():
    M, N, K = 200, 128, 128
    A, B = _rand(M, K, seed=1, dtype=torch.float16), _rand(N, K, scale=0.1, seed=2, dtype=torch.float16)
    out = torch.empty(M, N, device=DEV, dtype=torch.float16)
    hip.gemm_nt(hip.EPI_BIAS, A, B, M, N, K, out)
    _close(out, A.float() @ B.float().t(), 1 / 512, 1e-2, 'f16 gemm')


@pytest.mark.parametrize('tile', [-1, 313, 317, 319, 110])
def test_gemm_nt_gelu_resid_dgelu_epilogues(tile):
    M, N, K = 500, 384, 128
    A, B = _rand(M, K, seed=4), _rand(N, K, scale=0.15, seed=5)
    bias = _rand(N, seed=6, dtype=torch.float32) * 0.1
    acc = A.float() @ B.float().t() + bias
    # fc1: u and gelu(u)
    u = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    hh = torch.empty_like(u)
    hip.gemm_nt(hip.EPI_BIAS_GELU, A, B, M, N, K, u, out2=hh, bias=bias, tile=tile)
    _close(u, acc, 1 / 128, 1e-2, 'gelu.u')
    _close(hh, F.gelu(acc), 1 / 128, 1e-2, 'gelu.h')
    # residual: x + gamma * (acc + bias) * row_scale
    resid = _rand(M, N, seed=7, dtype=torch.float32)
    gamma = _rand(N, seed=8, dtype=torch.float32)
    rs = (torch.rand(M, device=DEV) > 0.3).float() / 0.7
    xo = torch.empty(M, N, device=DEV)
    zd = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    hip.gemm_nt(hip.EPI_RESID, A, B, M, N, K, xo, out2=zd, bias=bias, gamma=gamma, resid=resid, row_scale=rs, tile=tile)
    _close(zd, acc, 1 / 128, 1e-2, 'resid.zd')
    _close(xo, resid + gamma * acc * rs[:, None], 1e-3, 1e-3, 'resid.out')
    # dgelu: acc * gelu'(aux)
    aux = _rand(M, N, seed=9)
    dg = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    hip.gemm_nt(hip.EPI_DGELU, A, B, M, N, K, dg, aux=aux, tile=tile)
    a = aux.float().requires_grad_(True)
    F.gelu(a).backward(torch.ones_like(a))
    _close(dg, (A.float() @ B.float().t()) * a.grad, 1 / 128, 1e-2, 'dgelu')


@pytest.mark.parametrize('tile', [-1, 0, 317])
def test_gemm_nt_saved_gelu_derivative(tile):
    """VlmoEpilogue.relu bit 2: the fc1 epilogue writes d h / d u = GELU'(u) * dropout mask / (1 - p) in place of the
    pre-activation, the GELU-derivative epilogue multiplies by it -- the pair gives the same du as the pre-activation path
    (bf16 rounding of the saved factor instead of the saved u), with the SAME dropout mask as the forward's h."""
    M, N, K = 500, 384, 128
    A, B = _rand(M, K, seed=4), _rand(N, K, scale=0.15, seed=5)
    bias = _rand(N, seed=6, dtype=torch.float32) * 0.1
    acc = A.float() @ B.float().t() + bias
    drop = hip.drop_params(0.1, True)
    for dp in ((0, 1.0), drop):
        g = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        hh = torch.empty_like(g)
        hip.gemm_nt(hip.EPI_BIAS_GELU, A, B, M, N, K, g, out2=hh, bias=bias, relu=4, drop=dp, seed=11, tile=tile)
        a = acc.clone().requires_grad_(True)
        F.gelu(a).backward(torch.ones_like(a))
        keep = (hh != 0) | (g != 0)                 # the mask the forward applied: a dropped element is zero in BOTH outputs
        scale = dp[1]
        _close(hh, F.gelu(acc) * keep * scale, 1 / 128, 1e-2, 'h')
        _close(g, a.grad * keep * scale, 1 / 128, 1e-2, 'saved derivative')
        if dp[0]:
            assert 0.05 < 1 - keep.float().mean().item() < 0.15
        # backward: dh . W2^T-like product times the saved factor
        A2, B2 = _rand(M, K, seed=14), _rand(N, K, scale=0.15, seed=15)
        du = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        hip.gemm_nt(hip.EPI_DGELU, A2, B2, M, N, K, du, aux=g, relu=4, tile=tile)
        _close(du, (A2.float() @ B2.float().t()) * g.float(), 1 / 128, 1e-2, 'du')


@pytest.mark.parametrize('M', [500, 1000, 77])
def test_gemm_nt_dgelu_column_partials(M):
    """VlmoEpilogue.colpart: [ceil(M/16), N] fp32, EVERY row written by exactly one epilogue pass (sums of the pass's rows
    in the row of its first 16-row block, zeros in the others), so that the fc1 bias gradient is the plain fold of the
    rows whatever tile height ran: all tiles, ragged M."""
    N, K = 384, 128
    A, B = _rand(M, K, seed=4), _rand(N, K, scale=0.15, seed=5)
    aux = _rand(M, N, seed=9)
    a = aux.float().requires_grad_(True)
    F.gelu(a).backward(torch.ones_like(a))
    want = (A.float() @ B.float().t()) * a.grad
    nblk = (M + 15) // 16
    ref16 = torch.stack([want[16 * b:16 * b + 16].sum(0) for b in range(nblk)])
    for tile in (0, 3, 8, 106, 309, 310, 311, 312, 313, 315, 316, 317, 319, 320):
        cp = torch.full((nblk + 1, N), float('nan'), device=DEV)
        dg2 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        hip.gemm_nt(hip.EPI_DGELU, A, B, M, N, K, dg2, aux=aux, colpart=cp, tile=tile)
        _close(dg2, want, 1 / 128, 1e-2, 'dgelu tile')
        assert torch.isnan(cp[nblk]).all(), f'tile {tile}: partial rows beyond ceil(M/16) were written'
        assert not torch.isnan(cp[:nblk]).any(), f'tile {tile}: a partial row was left unwritten'
        _close(cp[:nblk].sum(0), want.sum(0), 2e-3, 2e-3 * want.sum(0).abs().max().item(), f'fold of the partials, tile {tile}')
        # a row is either all zeros or the sum of one or two consecutive 16-row blocks starting at its own
        for b in range(nblk):
            row = cp[b]
            if (row == 0).all():
                continue
            one = ref16[b]
            two = ref16[b] + (ref16[b + 1] if b + 1 < nblk else 0)
            tol = 2e-3 * max(one.abs().max().item(), two.abs().max().item()) + 1e-6
            assert (row - one).abs().max().item() <= tol or (row - two).abs().max().item() <= tol, (tile, b)


def test_dropout_epilogue_consistency_with_backward():
    """EPI_RESID dropout mask == vlmo_resid_bwd mask; drop rate ~ p; survivors scaled 1/(1-p)."""
    M, N, K = 512, 256, 64
    A, B = _rand(M, K, seed=1), _rand(N, K, scale=0.3, seed=2)
    resid = torch.zeros(M, N, device=DEV)
    drop = hip.drop_params(0.1, True)
    xo, zd = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    hip.gemm_nt(hip.EPI_RESID, A, B, M, N, K, xo, out2=zd, resid=resid, drop=drop, seed=1234)
    acc = A.float() @ B.float().t()
    kept = zd.float() != 0
    rate = 1 - kept.float().mean().item()
    assert abs(rate - 0.1) < 0.01, rate
    _close(zd[kept], (acc / 0.9)[kept], 1 / 100, 1e-2, 'dropout survivors')
    dx = torch.ones(M, N, device=DEV)
    dz = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    hip.resid_bwd(dx, None, None, None, dz, None, None, M, N, drop=drop, seed=1234)
    assert torch.equal(dz.float() != 0, kept), 'forward and backward dropout masks differ'
    zd2 = torch.empty_like(zd)
    hip.gemm_nt(hip.EPI_RESID, A, B, M, N, K, xo, out2=zd2, resid=resid, drop=drop, seed=99)
    assert not torch.equal(zd2.float() != 0, kept), 'seed does not change the mask'


@pytest.mark.parametrize('M,N1,N2', [(64, 128, 128), (1000, 256, 384), (333, 128, 776), (4096, 768, 768)])
def test_gemm_tn_wgrad(M, N1, N2):
    A, B = _rand(M, N1, seed=1), _rand(M, N2, seed=2)
    C = torch.zeros(N1, N2, device=DEV)
    hip.gemm_tn(A, B, C, M, N1, N2)
    ref = A.float().t() @ B.float()
    _close(C, ref, 2e-3, 2e-3 * math.sqrt(M), 'wgrad')
    hip.gemm_tn(A, B, C, M, N1, N2, alpha=0.5, splits=3)
    _close(C, 1.5 * ref, 2e-3, 3e-3 * math.sqrt(M), 'wgrad accumulate')
    hip.gemm_tn(A, B, C, M, N1, N2, alpha=-1.5, splits=3, slab=False)      # atomic path
    _close(C, 0 * ref, 2e-3, 6e-3 * math.sqrt(M), 'wgrad atomics')


@pytest.mark.parametrize('force', [1001, 2001, 2003])
def test_gemm_tn_tile_variants_exact(force):
    """both tile shapes of the wgrad kernel (128x128 and 256x256 sub-image tiles), exact integer data,
    ragged M / N1 / N2 (splits >= 1000 is the test hook that forces a tile shape)."""
    M, N1, N2 = 700, 384, 520
    g = torch.Generator().manual_seed(force)
    A = torch.randint(-3, 4, (M, N1), generator=g).float()
    B = torch.randint(-2, 3, (M, N2), generator=g).float()
    A[:, 0] += torch.arange(M).float() % 3
    C = torch.zeros(N1, N2, device=DEV)
    hip.gemm_tn(A.to(DEV).bfloat16(), B.to(DEV).bfloat16(), C, M, N1, N2, splits=force)
    assert torch.equal(C.cpu(), A.t() @ B)
    hip.gemm_tn(A.to(DEV).bfloat16(), B.to(DEV).bfloat16(), C, M, N1, N2, splits=force, slab=False)
    assert torch.equal(C.cpu(), 2 * (A.t() @ B))


def test_gemm_tn_exact_integers():
    M, N1, N2 = 200, 128, 256
    g = torch.Generator().manual_seed(5)
    A = torch.randint(-3, 4, (M, N1), generator=g).float()
    B = torch.randint(-2, 3, (M, N2), generator=g).float()
    A[0] += torch.arange(N1).float() % 3
    C = torch.zeros(N1, N2, device=DEV)
    hip.gemm_tn(A.to(DEV).bfloat16(), B.to(DEV).bfloat16(), C, M, N1, N2, splits=1)
    assert torch.equal(C.cpu(), A.t() @ B)


@pytest.mark.parametrize('M,d', [(37, 128), (1000, 768), (260, 1024), (64, 256)])
def test_layernorm_fwd_bwd(M, d):
    x = _rand(M, d, seed=1, dtype=torch.float32) * 2 + 0.5
    w = _rand(d, seed=2, dtype=torch.float32) * 0.1 + 1
    b = _rand(d, seed=3, dtype=torch.float32) * 0.1
    y = torch.empty(M, d, device=DEV, dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    hip.ln_fwd(x, w, b, y, mean, rstd, None, M, d, 1e-12)
    xr = x.clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (d,), wr, br, 1e-12)
    _close(y, ref, 1 / 128, 1e-2, 'ln fwd')
    _close(mean, x.mean(1), 1e-5, 1e-5, 'ln mean')
    # fp32 output through a row map (final norm scatter)
    perm = torch.randperm(M, device=DEV).int()
    y32 = torch.empty(M, d, device=DEV)
    hip.ln_fwd(x, w, b, y32, mean, rstd, perm, M, d, 1e-12)
    _close(y32[perm.long()], ref, 1e-4, 1e-4, 'ln fwd f32 rowmap')
    # backward
    dy = _rand(M, d, seed=4)
    dres = _rand(M, d, seed=5, dtype=torch.float32)
    ref.backward(dy.float())
    dx = torch.empty(M, d, device=DEV)
    dw, db = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    hip.ln_bwd(dy, None, x, w, mean, rstd, dres, dx, dw, db, M, d)
    _close(dx, xr.grad + dres, 1e-3, 1e-3, 'ln dx')
    _close(dw, wr.grad, 1e-3, 1e-3 * math.sqrt(M), 'ln dw')
    _close(db, br.grad, 1e-3, 1e-3 * math.sqrt(M), 'ln db')
    # fp32 dy through the row map, no residual
    dy32 = torch.empty(M, d, device=DEV)
    dy32[perm.long()] = dy.float()
    dx2 = torch.empty(M, d, device=DEV)
    hip.ln_bwd(dy32, perm, x, w, mean, rstd, None, dx2, None, None, M, d)
    _close(dx2, xr.grad, 1e-3, 1e-3, 'ln dx rowmap')


def _attn_case(B, lenA, lenB, heads, seed=0, mask_frac=0.3):
    """packed rows: all A segments, then all B segments."""
    d = heads * 64
    M = B * (lenA + lenB)
    qkv = _rand(M, 3 * d, seed=seed, scale=1.0)
    seg = torch.tensor([[b * lenA, lenA, B * lenA + b * lenB, lenB] for b in range(B)], dtype=torch.int32)
    keymask = torch.ones(M, dtype=torch.int32)
    g = torch.Generator().manual_seed(seed + 1)
    for b in range(B):
        if b % 2 == 1 and lenA > 4:
            p0 = int(torch.randint(2, lenA, (1,), generator=g))
            keymask[b * lenA + p0:(b + 1) * lenA] = 0
    return qkv, seg.to(DEV), keymask.to(DEV), M, d


def _hash32(x):
    """host replica of csrc/common.h hash32 on uint64 arrays holding 32-bit values"""
    m = np.uint64(0xFFFFFFFF)
    x = x ^ (x >> np.uint64(16))
    x = (x * np.uint64(0x7feb352d)) & m
    x = x ^ (x >> np.uint64(15))
    x = (x * np.uint64(0x846ca68b)) & m
    return x ^ (x >> np.uint64(16))


def _attn_keep_mask(seed, nseq, heads, N, thresh):
    """[nseq, heads, N(q), N(key)] bool: the keep mask of the attention kernels' counter-based dropout
    (csrc/attention.hip att_key / att_mix): top 16 bits of mix((q * 512 + key) * G + key(seed, bh)) >= thresh."""
    m = np.uint64(0xFFFFFFFF)
    bh = (np.arange(nseq, dtype=np.uint64)[:, None] * np.uint64(heads) + np.arange(heads, dtype=np.uint64)[None, :])
    lo, hi = np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32)
    akey = (_hash32((lo ^ ((bh * np.uint64(0x9E3779B9)) & m)) & m) + hi) & m
    q = np.arange(N, dtype=np.uint64)[None, None, :, None]
    key = np.arange(N, dtype=np.uint64)[None, None, None, :]
    x = ((((q * np.uint64(512) + key) & m) * np.uint64(0x9E3779B1)) + akey[:, :, None, None]) & m
    x = x ^ (x >> np.uint64(15))
    x = (x * np.uint64(0x7feb352d)) & m
    return torch.from_numpy((x >> np.uint64(16)) >= np.uint64(thresh))


def _attn_ref(qkv, seg, keymask, heads, d, dctx=None, keep=None, inv_keep=1.0):
    """fp32 torch reference per sequence (vlmo.py:79-95); returns ctx (+ dqkv).  keep: optional dropout keep
    mask [nseq, heads, N, N] applied to the probabilities (vlmo.py:93)."""
    q = qkv.float().clone().requires_grad_(dctx is not None)
    ctx = torch.zeros(q.shape[0], d, device=q.device)
    outs = []
    for si, s in enumerate(seg.tolist()):
        rows = torch.cat([torch.arange(s[0], s[0] + s[1]), torch.arange(s[2], s[2] + s[3])]).to(q.device)
        x = q[rows]
        N = x.shape[0]
        qq, kk, vv = [x[:, i * d:(i + 1) * d].reshape(N, heads, 64).transpose(0, 1) for i in range(3)]
        att = (qq @ kk.transpose(-2, -1)) * 64 ** -0.5
        att = att.masked_fill(~keymask[rows].bool()[None, None, :], float('-inf')).softmax(-1)
        if keep is not None:
            att = att * keep[si, :, :N, :N].to(att.device) * inv_keep
        o = (att @ vv).transpose(0, 1).reshape(N, d)
        outs.append((rows, o))
    ctx = torch.zeros(q.shape[0], d, device=q.device)
    for rows, o in outs:
        ctx = ctx.index_add(0, rows, o)
    if dctx is None:
        return ctx.detach()
    ctx.backward(dctx.float())
    return ctx.detach(), q.grad


@pytest.mark.parametrize('B,lenA,lenB,heads', [(3, 16, 17, 2), (2, 64, 0, 2), (2, 0, 197, 1),
                                               (2, 64, 197, 2), (1, 40, 197, 3), (2, 24, 50, 4), (3, 16, 0, 2), (2, 0, 256, 1),
                                               (2, 60, 197, 1), (1, 64, 224, 2), (2, 70, 200, 1)])
def test_attention_fwd_bwd(B, lenA, lenB, heads):
    qkv, seg, keymask, M, d = _attn_case(B, lenA, lenB, heads, seed=B + lenA)
    N = lenA + lenB
    ctx = torch.zeros(M, d, device=DEV, dtype=torch.bfloat16)
    lse = torch.zeros(B, heads, ((N + 31) // 32) * 32, device=DEV)
    hip.attn_fwd(qkv, seg, B, keymask, ctx, lse.view(B * heads, -1), heads, d, N, 64 ** -0.5)
    dctx = _rand(M, d, seed=77)
    ref_ctx, ref_dqkv = _attn_ref(qkv, seg, keymask, heads, d, dctx)
    _close(ctx, ref_ctx, 1 / 64, 1e-2, 'attn ctx')
    dqkv = torch.zeros(M, 3 * d, device=DEV, dtype=torch.bfloat16)
    qv = torch.full((B, 2 * d), float('nan'), device=DEV)
    hip.attn_bwd(qkv, ctx, dctx, lse.view(B * heads, -1), seg, B, keymask, dqkv, heads, d, N, 64 ** -0.5, qv_colsum=qv)
    scale = ref_dqkv.abs().max().item()
    _close(dqkv, ref_dqkv, 1 / 32, 2e-2 * scale, 'attn dqkv')
    # per-sequence column sums of dq | dv (what the q_bias / v_bias gradient is folded from)
    for si, sg in enumerate(seg.tolist()):
        rows = torch.cat([torch.arange(sg[0], sg[0] + sg[1]), torch.arange(sg[2], sg[2] + sg[3])]).to(DEV)
        want = torch.cat([ref_dqkv[rows, :d].sum(0), ref_dqkv[rows, 2 * d:].sum(0)])
        _close(qv[si], want, 1 / 32, 2e-2 * scale * max(1.0, len(rows) ** 0.5), 'attn qv column sums')


def test_attention_mixed_lengths_in_one_launch():
    """Below the fusion layer the image and the text sequences of a batch go into ONE launch (engine.Plan.seg_sep:
    longest first): the workgroups are sized for the longest sequence and a short one uses the first waves only."""
    B, T, P, heads = 3, 24, 197, 2
    d = heads * 64
    M = B * (T + P)
    qkv = _rand(M, 3 * d, seed=5, scale=1.0)
    seg = torch.tensor([[B * T + b * P, P, 0, 0] for b in range(B)] + [[b * T, T, 0, 0] for b in range(B)], dtype=torch.int32).to(DEV)
    keymask = torch.ones(M, dtype=torch.int32)
    keymask[T + 9:2 * T] = 0
    keymask = keymask.to(DEV)
    for drop, seed in ((None, 0), (hip.drop_params(0.1, True), 0x5EED5EED1)):
        ctx = torch.zeros(M, d, device=DEV, dtype=torch.bfloat16)
        lse = torch.zeros(2 * B * heads, 224, device=DEV)
        kw = dict(drop=drop, seed=seed) if drop else {}
        hip.attn_fwd(qkv, seg, 2 * B, keymask, ctx, lse, heads, d, P, 64 ** -0.5, **kw)
        keep = _attn_keep_mask(seed, 2 * B, heads, P, drop[0]) if drop else None
        dctx = _rand(M, d, seed=79)
        ref_ctx, ref_dqkv = _attn_ref(qkv, seg, keymask, heads, d, dctx, keep=keep, inv_keep=drop[1] if drop else 1.0)
        _close(ctx, ref_ctx, 1 / 64, 1e-2, 'attn ctx (mixed)')
        dqkv = torch.zeros(M, 3 * d, device=DEV, dtype=torch.bfloat16)
        hip.attn_bwd(qkv, ctx, dctx, lse, seg, 2 * B, keymask, dqkv, heads, d, P, 64 ** -0.5, **kw)
        _close(dqkv, ref_dqkv, 1 / 32, 2e-2 * ref_dqkv.abs().max().item(), 'attn dqkv (mixed)')


def test_attention_backward_nine_tile_sequences_in_a_mixed_launch():
    """One launch with sequences on both sides of the 256-token boundary between the single-pass and the two-phase
    backward (257 .. 288 tokens take the latter; with VLMO_ATTN_BWD_SPLIT=1 two launches: the tile pairs that touch the
    ninth tile in the two-phase kernel, the rest in the single-pass kernel, which adds the first launch's partial rows),
    a packed two-range sequence, key masks that cover a whole fringe, dropout: dq / dk / dv of every row and the
    per-sequence column sums against the fp32 reference with the replicated mask."""
    heads = 2
    d = heads * 64
    lens = [270, 200, 257, 64, 288, 256]
    starts = np.cumsum([0] + lens[:-1]).tolist()
    M = sum(lens)
    qkv = _rand(M, 3 * d, seed=11, scale=1.0)
    # the 270-token sequence is packed from two row ranges like a fused text + image sequence
    seg = [[s, n, 0, 0] for s, n in zip(starts, lens)]
    seg[0] = [starts[0], 70, starts[0] + 70, 200]
    seg = torch.tensor(seg, dtype=torch.int32).to(DEV)
    keymask = torch.ones(M, dtype=torch.int32)
    keymask[starts[0] + 40:starts[0] + 70] = 0
    keymask[starts[2] + 250:starts[2] + 257] = 0          # the whole fringe of the 257-token sequence and a few more
    keymask = keymask.to(DEV)
    nseq, N = len(lens), max(lens)
    for drop, seed in ((None, 0), (hip.drop_params(0.1, True), 0xABCDEF0123)):
        kw = dict(drop=drop, seed=seed) if drop else {}
        ctx = torch.zeros(M, d, device=DEV, dtype=torch.bfloat16)
        lse = torch.zeros(nseq * heads, 288, device=DEV)
        hip.attn_fwd(qkv, seg, nseq, keymask, ctx, lse, heads, d, N, 64 ** -0.5, **kw)
        keep = _attn_keep_mask(seed, nseq, heads, N, drop[0]) if drop else None
        dctx = _rand(M, d, seed=80)
        ref_ctx, ref_dqkv = _attn_ref(qkv, seg, keymask, heads, d, dctx, keep=keep, inv_keep=drop[1] if drop else 1.0)
        _close(ctx, ref_ctx, 1 / 64, 1e-2, 'attn ctx (split launch)')
        dqkv = torch.full((M, 3 * d), float('nan'), device=DEV, dtype=torch.bfloat16)
        qv = torch.full((nseq, 2 * d), float('nan'), device=DEV)
        hip.attn_bwd(qkv, ctx, dctx, lse, seg, nseq, keymask, dqkv, heads, d, N, 64 ** -0.5, qv_colsum=qv, **kw)
        scale = ref_dqkv.abs().max().item()
        _close(dqkv, ref_dqkv, 1 / 32, 2e-2 * scale, 'attn dqkv (split launch)')
        for si, sg in enumerate(seg.tolist()):
            rows = torch.cat([torch.arange(sg[0], sg[0] + sg[1]), torch.arange(sg[2], sg[2] + sg[3])]).to(DEV)
            want = torch.cat([ref_dqkv[rows, :d].sum(0), ref_dqkv[rows, 2 * d:].sum(0)])
            _close(qv[si], want, 1 / 32, 2e-2 * scale * len(rows) ** 0.5, f'attn qv column sums (split launch, sequence {si})')


@pytest.mark.parametrize('B,lenA,lenB,heads', [(2, 64, 197, 2), (3, 16, 17, 1)])
def test_attention_dropout_forward_and_backward_use_the_same_mask(B, lenA, lenB, heads):
    """attention dropout (vlmo.py:93): the mask is regenerated in the backward kernels from (seed, sequence, head,
    query, key); a host replica of the hash gives the fp32 reference the SAME mask, so ctx and dq/dk/dv are checked
    element-wise -- forward, dQ phase and dK/dV phase must all agree on it."""
    qkv, seg, keymask, M, d = _attn_case(B, lenA, lenB, heads, seed=17 + B)
    N = lenA + lenB
    drop = hip.drop_params(0.1, True)
    seed = 0x1234567ABCDEF01
    ctx = torch.zeros(M, d, device=DEV, dtype=torch.bfloat16)
    lse = torch.zeros(B * heads, ((N + 31) // 32) * 32, device=DEV)
    hip.attn_fwd(qkv, seg, B, keymask, ctx, lse, heads, d, N, 64 ** -0.5, drop=drop, seed=seed)
    keep = _attn_keep_mask(seed, B, heads, N, drop[0])
    assert abs(keep.float().mean().item() - 0.9) < 0.01
    dctx = _rand(M, d, seed=78)
    ref_ctx, ref_dqkv = _attn_ref(qkv, seg, keymask, heads, d, dctx, keep=keep, inv_keep=drop[1])
    _close(ctx, ref_ctx, 1 / 64, 1e-2, 'attn ctx (dropout)')
    dqkv = torch.zeros(M, 3 * d, device=DEV, dtype=torch.bfloat16)
    hip.attn_bwd(qkv, ctx, dctx, lse, seg, B, keymask, dqkv, heads, d, N, 64 ** -0.5, drop=drop, seed=seed)
    scale = ref_dqkv.abs().max().item()
    _close(dqkv, ref_dqkv, 1 / 32, 2e-2 * scale, 'attn dqkv (dropout)')


def test_attention_dropout_statistics():
    B, lenA, lenB, heads = 2, 64, 197, 2
    qkv, seg, keymask, M, d = _attn_case(B, lenA, lenB, heads, seed=3)
    qkv[:, :2 * d] = 0          # uniform attention: ctx = mean of (kept v)/(1-p)
    ctx0 = torch.empty(M, d, device=DEV, dtype=torch.bfloat16)
    ctx1 = torch.empty_like(ctx0)
    lse = torch.zeros(B * heads, 288, device=DEV)
    hip.attn_fwd(qkv, seg, B, None, ctx0, lse, heads, d, 261, 0.125)
    hip.attn_fwd(qkv, seg, B, None, ctx1, lse, heads, d, 261, 0.125, drop=hip.drop_params(0.1, True), seed=5)
    # E[ctx1] = ctx0 ; var small: compare batch means
    assert abs(ctx1.float().mean().item() - ctx0.float().mean().item()) < 5e-3
    assert (ctx1.float() - ctx0.float()).abs().mean().item() > 1e-3   # dropout did something


def test_colsum_cast_patchify():
    M, N = 1000, 768
    x = _rand(M, N, seed=1)
    out = torch.zeros(N, device=DEV)
    hip.colsum(x, out, M, N)
    _close(out, x.float().sum(0), 1e-3, 1e-2, 'colsum')
    w = _rand(300, 200, seed=2, dtype=torch.float32)
    a = torch.empty(300, 200, device=DEV, dtype=torch.bfloat16)
    at = torch.empty(200, 300, device=DEV, dtype=torch.bfloat16)
    hip.cast_weight(w, a, at)
    assert torch.equal(a, w.bfloat16()) and torch.equal(at, w.bfloat16().t().contiguous())
    img = _rand(2, 3, 64, 64, seed=3, dtype=torch.float32)
    p = torch.empty(2 * 16, 3 * 256, device=DEV, dtype=torch.bfloat16)
    hip.patchify(img, p, 16)
    ref = F.unfold(img, 16, stride=16).transpose(1, 2).reshape(32, 768)
    assert torch.equal(p, ref.bfloat16())


def test_embed_img_fwd_bwd():
    B, npatch, d = 3, 16, 128
    proj = _rand(B * npatch, d, seed=1)
    cls_tok, mask_tok = [_rand(d, seed=s, dtype=torch.float32) for s in (2, 3)]
    pos = _rand(npatch + 1, d, seed=4, dtype=torch.float32)
    type_row = _rand(d, seed=5, dtype=torch.float32)
    masked = (torch.rand(B, npatch) < 0.4).to(torch.uint8).to(DEV)
    x = torch.empty(B, npatch + 1, d, device=DEV)
    hip.embed_img_finish(proj, cls_tok, mask_tok, pos, type_row, masked, x, B, npatch, d)
    pr = proj.float().view(B, npatch, d).clone().requires_grad_(True)
    leaves = [t.clone().requires_grad_(True) for t in (cls_tok, mask_tok, pos, type_row)]
    w = masked.float().unsqueeze(-1)
    xr = pr * (1 - w) + leaves[1] * w
    xr = torch.cat([leaves[0].expand(B, 1, d), xr], 1) + leaves[2] + leaves[3]
    _close(x, xr, 1e-6, 1e-6, 'embed_img fwd')
    dx = _rand(B, npatch + 1, d, seed=6, dtype=torch.float32)
    xr.backward(dx)
    dproj = torch.empty(B * npatch, d, device=DEV, dtype=torch.bfloat16)
    dcls, dmask, dtype_row = [torch.zeros(d, device=DEV) for _ in range(3)]
    dpos = torch.zeros(npatch + 1, d, device=DEV)
    hip.embed_img_bwd(dx, masked, dproj, dcls, dmask, dpos, dtype_row, B, npatch, d)
    _close(dproj.view(B, npatch, d), pr.grad, 1 / 128, 1e-3, 'dproj')
    for got, leaf, name in ((dcls, leaves[0], 'dcls'), (dmask, leaves[1], 'dmask'),
                            (dpos, leaves[2], 'dpos'), (dtype_row, leaves[3], 'dtype')):
        _close(got, leaf.grad, 1e-4, 1e-4, name)


def test_embed_txt_fwd_bwd():
    B, T, d, V = 3, 16, 128, 50
    ids = torch.randint(0, V, (B, T)).to(DEV)
    ids[0, 5:] = 0
    tabs = [_rand(V, d, seed=1, dtype=torch.float32), _rand(T, d, seed=2, dtype=torch.float32),
            _rand(d, seed=3, dtype=torch.float32), _rand(d, seed=4, dtype=torch.float32) * 0.1 + 1,
            _rand(d, seed=5, dtype=torch.float32), _rand(d, seed=6, dtype=torch.float32)]
    word, pos, bt0, lw, lb, t0 = tabs
    x = torch.empty(B * T, d, device=DEV)
    xhat, rstd = torch.empty(B * T, d, device=DEV), torch.empty(B * T, device=DEV)
    hip.embed_txt_fwd(ids, word, pos, bt0, lw, lb, t0, x, xhat, rstd, B, T, d, 1e-12)
    L = [t.clone().requires_grad_(True) for t in tabs]
    e = F.embedding(ids, L[0], padding_idx=0) + L[2] + L[1][:T]
    xr = F.layer_norm(e, (d,), L[3], L[4], 1e-12) + L[5]
    _close(x.view(B, T, d), xr, 1e-4, 1e-4, 'embed_txt fwd')
    dx = _rand(B * T, d, seed=7, dtype=torch.float32)
    xr.backward(dx.view(B, T, d))
    outs = [torch.zeros_like(t) for t in tabs]
    hip.embed_txt_bwd(dx, ids, xhat, rstd, lw, outs[0], outs[1], outs[2], outs[3], outs[4], outs[5], B, T, d)
    for got, leaf, name in zip(outs, L, ('dword', 'dpos', 'dbtype0', 'dlnw', 'dlnb', 'dtype0')):
        _close(got, leaf.grad, 1e-3, 1e-3, name)


@pytest.mark.parametrize('tile', [0, 3, 4, 8, 313, 317])
@pytest.mark.parametrize('epi', ['bias_gelu', 'resid'])
def test_gemm_nt_grouped_equals_separate_launches(tile, epi):
    """vlmo_gemm_nt_grouped (the per-modality expert FFNs in one launch) is bit-identical to one launch per
    group, including the dropout streams (per-group seed, group-local element index); ragged group sizes."""
    g = torch.Generator().manual_seed(7)
    Ms, N, K = [300, 1000, 129], 384, 128
    As = [torch.randn(m, K, generator=g).to(DEV).bfloat16() for m in Ms]
    Bs = [(torch.randn(N, K, generator=g) * 0.1).to(DEV).bfloat16() for _ in Ms]
    biases = [torch.randn(N, generator=g).to(DEV) for _ in Ms]
    drop = hip.drop_params(0.1, True)
    if epi == 'bias_gelu':
        code, odt = hip.EPI_BIAS_GELU, torch.bfloat16
        mk = lambda i, m: dict(bias=biases[i], out2=torch.empty(m, N, device=DEV, dtype=torch.bfloat16), seed=100 + i)
        common = dict(drop=drop)
    else:
        code, odt = hip.EPI_RESID, torch.float32
        gamma = torch.rand(N, generator=g).to(DEV)
        resids = [torch.randn(m, N, generator=g).to(DEV) for m in Ms]
        mk = lambda i, m: dict(bias=biases[i], resid=resids[i], out2=torch.empty(m, N, device=DEV, dtype=torch.bfloat16),
                               seed=200 + i)
        common = dict(drop=drop, gamma=gamma)
    sep, sep_kw = [], []
    for i, m in enumerate(Ms):
        out = torch.empty(m, N, device=DEV, dtype=odt)
        kw = mk(i, m)
        hip.gemm_nt(code, As[i], Bs[i], m, N, K, out, tile=tile, **common, **kw)
        sep.append(out)
        sep_kw.append(kw)
    outs = [torch.full((m, N), float('nan'), device=DEV, dtype=odt) for m in Ms]
    grp_kw = [mk(i, m) for i, m in enumerate(Ms)]
    hip.gemm_nt_grouped(code, As, Bs, Ms, N, K, outs, per_group=grp_kw, tile=tile, **common)
    for i in range(len(Ms)):
        assert torch.equal(outs[i], sep[i]), i
        assert torch.equal(grp_kw[i]['out2'], sep_kw[i]['out2']), i


@pytest.mark.parametrize('drop', [0.0, 0.1])
def test_ln_resid_bwd_equals_ln_bwd_then_resid_bwd(drop):
    """vlmo_ln_resid_bwd = vlmo_ln_bwd followed by vlmo_resid_bwd on the dx it wrote: dx and dz bit-identical
    (same arithmetic, same dropout stream), the four column sums equal up to summation order."""
    g = torch.Generator().manual_seed(11)
    M, d = 1000, 768
    dy = torch.randn(M, d, generator=g).to(DEV).bfloat16()
    x = torch.randn(M, d, generator=g).to(DEV)
    w = (1 + 0.1 * torch.randn(d, generator=g)).to(DEV)
    mean = x.mean(1)
    rstd = (x.var(1, unbiased=False) + 1e-12).rsqrt()
    dres = torch.randn(M, d, generator=g).to(DEV)
    zd = torch.randn(M, d, generator=g).to(DEV).bfloat16()
    gamma = torch.rand(d, generator=g).to(DEV)
    scale = (torch.rand(7, generator=g) + 0.5).to(DEV)
    ridx = torch.randint(0, 7, (M,), generator=g).to(torch.int32).to(DEV)
    dp = hip.drop_params(drop, True)
    z = lambda *s: torch.zeros(*s, device=DEV)
    dx0, dw0, db0, dz0, dg0, dpb0 = torch.empty(M, d, device=DEV), z(d), z(d), torch.empty(M, d, device=DEV, dtype=torch.bfloat16), z(d), z(d)
    hip.ln_bwd(dy, None, x, w, mean, rstd, dres, dx0, dw0, db0, M, d)
    hip.resid_bwd(dx0, zd, gamma, scale, dz0, dg0, dpb0, M, d, drop=dp, seed=5, row_index=ridx)
    dx1, dw1, db1, dz1, dg1, dpb1 = torch.empty(M, d, device=DEV), z(d), z(d), torch.empty(M, d, device=DEV, dtype=torch.bfloat16), z(d), z(d)
    hip.ln_resid_bwd(dy, x, w, mean, rstd, dres, dx1, dw1, db1, zd, gamma, scale, ridx, dz1, dg1, dpb1, M, d, drop=dp, seed=5)
    assert torch.equal(dx0, dx1) and torch.equal(dz0, dz1)
    for a, b_ in ((dw0, dw1), (db0, db1), (dg0, dg1), (dpb0, dpb1)):
        torch.testing.assert_close(a, b_, rtol=1e-4, atol=1e-3)


def test_gemm_tn_multi_exact_integers():
    """vlmo_gemm_tn_multi: several weight-gradient problems of different shapes in ONE launch, exact small-integer
    data.  Covers the no-split path (>= 192 tiles: in-place store / accumulate), the split path (few tiles: fp32
    atomics, incl. the memset of a non-accumulating output), edge tiles (N not a multiple of 256), a token count that
    is not a multiple of 64, and leading dimensions larger than the used columns."""
    g = torch.Generator().manual_seed(11)

    def prob(M, N1, N2, lda_extra=0, ldb_extra=0, acc=True):
        A = torch.randint(-2, 3, (M, N1 + lda_extra), generator=g).float()
        B = torch.randint(-2, 3, (M, N2 + ldb_extra), generator=g).float()
        B[:, 0] += torch.arange(M).float() % 3
        C0 = torch.randint(-4, 5, (N1, N2), generator=g).float()
        ref = A[:, :N1].t() @ B[:, :N2] + (C0 if acc else 0)
        return (A.to(DEV).bfloat16(), B.to(DEV).bfloat16(), C0.to(DEV).clone(), M, N1, N2, acc), ref

    # few tiles -> token dimension split, atomics
    small = [prob(700, 256, 512), prob(333, 96, 264, acc=False), prob(1000, 512, 256, lda_extra=64, ldb_extra=8)]
    hip.gemm_tn_multi([p for p, _ in small])
    torch.cuda.synchronize()
    for (p, ref) in small:
        assert torch.equal(p[2].cpu(), ref), (p[2].cpu() - ref).abs().max()
    # >= 192 tiles together -> every tile owned by one workgroup, in-place accumulate / store
    big = [prob(300, 768, 3072), prob(200, 3072, 768, acc=False), prob(130, 768, 768), prob(257, 2304, 768),
           prob(64, 1000, 264, acc=False)]
    hip.gemm_tn_multi([p for p, _ in big])
    torch.cuda.synchronize()
    for (p, ref) in big:
        assert torch.equal(p[2].cpu(), ref), (p[2].cpu() - ref).abs().max()


def test_gemm_tn_multi_more_than_16_problems():
    g = torch.Generator().manual_seed(5)
    probs, refs = [], []
    for i in range(19):
        M, N1, N2 = 64 + 8 * i, 256, 264
        A = torch.randint(-2, 3, (M, N1), generator=g).float()
        B = torch.randint(-2, 3, (M, N2), generator=g).float()
        probs.append((A.to(DEV).bfloat16(), B.to(DEV).bfloat16(), torch.zeros(N1, N2, device=DEV), M, N1, N2, True))
        refs.append(A.t() @ B)
    hip.gemm_tn_multi(probs)
    torch.cuda.synchronize()
    for p, r in zip(probs, refs):
        assert torch.equal(p[2].cpu(), r)


def test_colwork_multi():
    """vlmo_colwork_multi: partial-row folds into up to four outputs and bf16 column sums, batched in one launch."""
    g = torch.Generator().manual_seed(3)
    d = 192
    ws = torch.randn(300, 4 * d, generator=g).to(DEV)
    o = [torch.randn(d, generator=g).to(DEV) for _ in range(4)]
    o0 = [t.clone() for t in o]
    ws2 = torch.randn(40, 2 * d, generator=g).to(DEV)
    p, q = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    x = (torch.randn(5000, 520, generator=g)).to(DEV).bfloat16()
    cs = torch.zeros(512, device=DEV)
    x2 = (torch.randn(100, 64, generator=g)).to(DEV).bfloat16()
    cs2 = torch.ones(64, device=DEV)
    hip.colwork_multi([(0, ws, 300, 4 * d, o, d), (1, x, 5000, 512, [cs], 512), (0, ws2, 40, 2 * d, [p, None], d),
                       (1, x2, 100, 64, [cs2], 64)])
    torch.cuda.synchronize()
    for k in range(4):
        _close(o[k], o0[k] + ws[:, k * d:(k + 1) * d].sum(0), 1e-5, 1e-4, f'fold out{k}')
    _close(p, ws2[:, :d].sum(0), 1e-5, 1e-4, 'fold with a NULL output')
    assert torch.equal(q, torch.zeros_like(q))
    _close(cs, x[:, :512].float().sum(0), 1e-5, 1e-3, 'colsum')
    _close(cs2, 1 + x2.float().sum(0), 1e-5, 1e-3, 'colsum small')


def test_gemm_tn_multi_more_tiles_than_one_placement_table():
    """20 problems x 64 tiles = 1 280 workgroups: more than one launch's placement table (1 024) holds, so the call
    is cut into several launches; mixed reduction lengths exercise the cost-balanced XCD placement."""
    g = torch.Generator().manual_seed(8)
    probs, refs = [], []
    for i in range(20):
        M = 64 * (1 + i % 3) + (5 if i % 4 == 0 else 0)
        A = torch.randint(-2, 3, (M, 2048), generator=g).float()
        B = torch.randint(-2, 3, (M, 2048), generator=g).float()
        probs.append((A.to(DEV).bfloat16(), B.to(DEV).bfloat16(), torch.zeros(2048, 2048, device=DEV), M, 2048, 2048, i % 2 == 0))
        refs.append(A.t() @ B)
    hip.gemm_tn_multi(probs)
    torch.cuda.synchronize()
    for p, r in zip(probs, refs):
        assert torch.equal(p[2].cpu(), r)


def test_cast_weight_multi_matches_single():
    """vlmo_cast_weight_multi: every job's W and W^T copies equal the one-weight kernel's (ragged shapes, > 72 jobs)."""
    g = torch.Generator().manual_seed(11)
    shapes = [(768, 768), (2304, 768), (3072, 768), (768, 3072), (100, 36), (33, 65), (8, 4)] * 12      # 84 jobs
    jobs, want = [], []
    for r, c in shapes:
        src = torch.randn(r, c, generator=g).to(DEV)
        w, wt = torch.empty(r, c, dtype=torch.bfloat16, device=DEV), torch.empty(c, r, dtype=torch.bfloat16, device=DEV)
        jobs.append((src, w, wt))
        want.append((src.bfloat16(), src.t().contiguous().bfloat16()))
    jobs[3] = (jobs[3][0], jobs[3][1], None)          # no transposed copy for this one
    hip.cast_weight_multi(jobs)
    for q, ((_, w, wt), (rw, rwt)) in enumerate(zip(jobs, want)):
        assert torch.equal(w, rw), q
        if wt is not None:
            assert torch.equal(wt, rwt), q
