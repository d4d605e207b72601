"""dVAE encoder on the HIP engine (fp16, as the reference runs it on GPU) against
(a) plain PyTorch fp32 convolution of the same fp16-rounded operands per kernel and
(b) the reference's own logits / codebook ids (tests/golden/dvae_*.npz).
Tolerances: conv outputs 2^-10 relative (one fp16 rounding) + accumulation noise;
end-to-end logits 3e-2 abs (8 blocks of fp16 activations, logit scale ~1); ids may
differ from the reference only where the reference's own top-2 logit gap is below
that noise."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda'

from exploremultimodal_amd import hip  # noqa: E402


def _nhwc(x):      # [B,C,H,W] -> [B*H*W, C]
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


@pytest.mark.parametrize('B,H,W,Cin,Cout,kw', [(2, 12, 12, 64, 64, 3), (1, 16, 20, 128, 256, 3),
                                               (2, 7, 9, 64, 128, 1), (3, 14, 14, 256, 64, 3),
                                               (11, 64, 60, 64, 256, 3),       # 165 tiles of 256 x 256: the ping-pong kernel
                                               (2, 9, 11, 64, 32, 3), (1, 10, 12, 64, 192, 3),     # ragged channel tiles of the shared-dx kernels
                                               (1, 5, 7, 128, 36, 3)])         # 36 channels: not a multiple of 8 -> generic kernel
def test_conv2d_nhwc_vs_torch(B, H, W, Cin, Cout, kw):
    g = torch.Generator().manual_seed(Cin + Cout + kw)
    x = torch.randn(B, Cin, H, W, generator=g).half().to(DEV)
    w = (torch.randn(Cout, Cin, kw, kw, generator=g) / (Cin * kw * kw) ** 0.5).half().to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    ref = F.conv2d(x.float(), w.float(), b, padding=(kw - 1) // 2)
    wq = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous()
    out = torch.empty(B * H * W, Cout, device=DEV, dtype=torch.float16)
    hip.conv2d_nhwc(hip.EPI_BIAS, _nhwc(x), B, H, W, Cin, kw, wq, Cout, out, bias=b)
    got = out.view(B, H, W, Cout).permute(0, 3, 1, 2).float()
    assert (got - ref).abs().max().item() <= 4e-3 + 2e-3 * ref.abs().max().item()
    out2 = torch.empty_like(out)
    hip.conv2d_nhwc(hip.EPI_BIAS, _nhwc(x), B, H, W, Cin, kw, wq, Cout, out2, bias=b, relu=True)
    assert (out2.view(B, H, W, Cout).permute(0, 3, 1, 2).float() - ref.relu()).abs().max().item() <= 4e-3 + 2e-3 * ref.abs().max().item()


def test_conv2d_exact_integers_borders():
    """integer data: exact sums; catches any wrong tap/offset/zero-padding at the image border."""
    B, H, W, Cin, Cout = 2, 5, 6, 64, 128
    g = torch.Generator().manual_seed(3)
    x = torch.randint(-2, 3, (B, Cin, H, W), generator=g).float()
    w = torch.randint(-1, 2, (Cout, Cin, 3, 3), generator=g).float()
    ref = F.conv2d(x, w, None, padding=1)
    out = torch.empty(B * H * W, Cout, device=DEV)
    hip.conv2d_nhwc(hip.EPI_F32, _nhwc(x.half().to(DEV)), B, H, W, Cin, 3,
                    w.permute(0, 2, 3, 1).reshape(Cout, -1).half().contiguous().to(DEV), Cout, out)
    assert torch.equal(out.view(B, H, W, Cout).permute(0, 3, 1, 2).cpu(), ref)


def test_dual_epilogue_maxpool_im2col_argmax():
    M, N, K = 300, 256, 64
    g = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=g).half().to(DEV)
    Wt = (torch.randn(N, K, generator=g) / 8).half().to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    resid = torch.randn(M, N, generator=g).half().to(DEV)
    raw, rel = torch.empty(M, N, device=DEV, dtype=torch.float16), torch.empty(M, N, device=DEV, dtype=torch.float16)
    hip.gemm_nt(hip.EPI_DUAL, A, Wt, M, N, K, raw, out2=rel, bias=bias, resid=resid, beta=1 / 64)
    ref = resid.float() + (A.float() @ Wt.float().t() + bias) / 64
    assert (raw.float() - ref).abs().max().item() <= 4e-3
    assert (rel.float() - ref.relu()).abs().max().item() <= 4e-3
    # maxpool
    x = torch.randn(2, 64, 8, 6, generator=g).half().to(DEV)
    rp, lp = torch.empty(2 * 4 * 3, 64, device=DEV, dtype=torch.float16), torch.empty(2 * 4 * 3, 64, device=DEV, dtype=torch.float16)
    hip.maxpool2_nhwc(_nhwc(x), rp, lp, 2, 8, 6, 64)
    pr = F.max_pool2d(x.float(), 2)
    assert torch.equal(rp.view(2, 4, 3, 64).permute(0, 3, 1, 2).float(), pr)
    assert torch.equal(lp.view(2, 4, 3, 64).permute(0, 3, 1, 2).float(), pr.relu())
    # im2col of the 7x7 stem
    img = torch.rand(2, 3, 10, 12, generator=g).to(DEV)
    cols = torch.empty(2 * 10 * 12, 192, device=DEV, dtype=torch.float16)
    hip.dvae_im2col(img, cols, 7, 192)
    refc = F.unfold(img, 7, padding=3).transpose(1, 2).reshape(-1, 147)
    assert torch.equal(cols[:, :147], refc.half()) and (cols[:, 147:] == 0).all()
    # fused arg-max epilogue
    Nv = 1000
    Wv = torch.randn(Nv, K, generator=g).half().to(DEV)
    bv = torch.randn(Nv, generator=g).to(DEV)
    part = torch.empty(M, (Nv + 63) // 64, 2, device=DEV)
    hip.gemm_nt(hip.EPI_ARGMAX, A, Wv, M, Nv, K, part, bias=bv, ldo=(Nv + 63) // 64)
    ids = torch.empty(M, dtype=torch.int64, device=DEV)
    hip.argmax_reduce(part, (Nv + 63) // 64, ids, M)
    logits = A.float() @ Wv.float().t() + bv
    assert (ids == logits.argmax(1)).float().mean().item() > 0.99
    picked = logits.gather(1, ids[:, None])[:, 0]
    assert (logits.max(1).values - picked).max().item() < 1e-3


@pytest.mark.parametrize('B,C,H,W,Kpad', [(2, 3, 112, 112, 192), (1, 3, 16, 224, 192), (3, 3, 8, 9, 192),
                                           (1, 1, 8, 40, 64), (2, 5, 8, 8, 256)])
def test_im2col_row_form_equals_unfold(B, C, H, W, Kpad):
    """The stem's patch matrix (dall_e/encoder.py:75, utils.py:14 padding (kw-1)//2) from the LDS-staged row kernel:
    every column of every pixel equal to F.unfold, padding columns zero; widths that are not multiples of the pixels
    per pass, one and five input channels."""
    g = torch.Generator().manual_seed(7)
    img = (torch.rand(B, C, H, W, generator=g) - 0.3).to(DEV)
    cols = torch.full((B * H * W, Kpad), float('nan'), device=DEV, dtype=torch.float16)
    hip.dvae_im2col(img, cols, 7, Kpad)
    refc = F.unfold(img, 7, padding=3).transpose(1, 2).reshape(-1, C * 49)
    assert torch.equal(cols[:, :C * 49], refc.half())
    assert (cols[:, C * 49:] == 0).all()


def _encoder(**kw):
    from exploremultimodal_amd.dvae import Encoder
    enc = Encoder(**kw)
    r = enc.load_state_dict(synth.synth_dvae_state_dict(0, **kw), strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    return enc.to(DEV)


def test_encoder_small_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, 'dvae_small.npz'))
    enc = _encoder(n_hid=256, vocab_size=1024)
    gen = torch.Generator().manual_seed(99)
    x = (0.8 * torch.rand(2, 3, 32, 32, generator=gen) + 0.1).to(DEV)
    logits = enc(x)
    ref = torch.from_numpy(g['logits'])
    err = (logits.cpu() - ref).abs()
    print('dvae_small logits max err', err.max().item(), 'mean', err.mean().item(), 'scale', ref.abs().mean().item())
    assert err.max().item() <= 3e-2
    ids = enc.codebook_indices(x).cpu()
    assert ids.shape == (2, 4, 4) and ids.dtype == torch.int64
    diff = ids.numpy() != g['ids']
    assert (g['top2_gap'][diff] < 3e-2).all(), 'ids differ where the reference has a clear winner'
    assert torch.equal(ids, logits.argmax(1).cpu()) or diff.mean() < 0.2


def test_encoder_full_ids_vs_reference(golden_dir):
    """BASELINE shape: 112x112 input -> 14x14 ids, 8192-way."""
    g = np.load(os.path.join(golden_dir, 'dvae_full_b2.npz'))
    enc = _encoder()
    gen = torch.Generator().manual_seed(99)
    x = (0.8 * torch.rand(2, 3, 112, 112, generator=gen) + 0.1).to(DEV)
    ids = enc.codebook_indices(x).cpu().numpy()
    assert ids.shape == (2, 14, 14)
    diff = ids != g['ids']
    gaps = g['top2_gap']
    print('dvae_full: id agreement', 1 - diff.mean(), 'max gap among mismatches', gaps[diff].max() if diff.any() else 0,
          'median top-2 gap', np.median(gaps))
    # gap histogram of the reference's top-2 logits: all positions vs the positions where the ids differ (SURVEY 8c:
    # >= 99 % exact, every mismatch a near-tie below the fp16 activation resolution)
    edges = [0, 1e-3, 3e-3, 1e-2, 3e-2, 1e-1, 1e9]
    print('top-2 gap histogram (edges', edges[:-1], '): all', np.histogram(gaps, edges)[0].tolist(),
          'mismatched', np.histogram(gaps[diff], edges)[0].tolist())
    assert 1 - diff.mean() >= 0.99
    assert (gaps[diff] < 3e-2).all()
    lm = enc(x).amax(1).cpu().numpy()
    assert np.abs(lm - g['logits_max']).max() <= 3e-2


def test_output_convolution_row_split_equals_one_launch():
    """Encoder._row_parts: at 12 images of 112 x 112 the output convolution has 10 x 32 = 320 tiles of 256 x 256 = one
    full dispatch round + a quarter; the rows of the partial round go out as 128 x 128 tiles (a second launch).  Logits
    (fp32 epilogue) and ids (fused arg-max epilogue) against the one-launch form: the same products in another
    summation order."""
    from exploremultimodal_amd import dvae
    enc = _encoder()
    assert enc._row_parts(12 * 196, 8192) == [(0, 2048, 3), (2048, 2352, 0)]
    assert enc._row_parts(2 * 196, 8192) == [(0, 392, -1)] and enc._row_parts(64 * 196, 8192)[1] == (12288, 12544, 0)
    gen = torch.Generator().manual_seed(5)
    x = (0.8 * torch.rand(12, 3, 112, 112, generator=gen) + 0.1).to(DEV)
    old = dvae.OUTPUT_ROW_SPLIT
    try:
        dvae.OUTPUT_ROW_SPLIT = True
        lo_s, ids_s = enc(x).float(), enc.codebook_indices(x)
        dvae.OUTPUT_ROW_SPLIT = False
        assert enc._row_parts(12 * 196, 8192) == [(0, 2352, -1)]
        lo_1, ids_1 = enc(x).float(), enc.codebook_indices(x)
    finally:
        dvae.OUTPUT_ROW_SPLIT = old
    assert (lo_s - lo_1).abs().max().item() <= 1e-4
    assert torch.equal(lo_s[:10], lo_1[:10])        # images 0..9 = matrix rows below 1 960: the same tiles in both forms
    assert (ids_s == ids_1).float().mean().item() >= 0.999
    assert torch.equal(ids_s, lo_s.argmax(1)) or (ids_s == lo_s.argmax(1)).float().mean().item() >= 0.999


def test_encoder_input_checks():
    from exploremultimodal_amd.dvae import Encoder, create_d_vae
    enc = Encoder(n_hid=256, vocab_size=512).to(DEV)
    with pytest.raises(ValueError):
        enc(torch.zeros(3, 32, 32, device=DEV))
    with pytest.raises(ValueError):
        enc(torch.zeros(1, 4, 32, 32, device=DEV))
    with pytest.raises(ValueError):
        enc(torch.zeros(1, 3, 32, 32, device=DEV, dtype=torch.float64))
    with pytest.raises(ValueError):
        Encoder(n_hid=32)
    with pytest.raises(NotImplementedError):
        create_d_vae(None, 'customized', 112, 'cpu')


@pytest.mark.parametrize('M,N,k1,k2,dt,scale', [(300, 256, 128, 256, torch.float16, 1 / 64),      # EncoderBlock tail, small tiles
                                                (1000, 512, 1024, 1024, torch.float16, 1.0),     # [x | x] . [w_hi | w_lo], 256x256 ping-pong
                                                (257, 768, 64, 1536, torch.float16, 0.5)])
def test_gemm_nt_two_segments(M, N, k1, k2, dt, scale):
    """vlmo_gemm_nt_2src: out = seg_scale * A W1^T + A2 W2^T + bias with W = [W1 | W2] one matrix; both sources with
    their own leading dimension (A is a column slice of a wider buffer)."""
    g = torch.Generator().manual_seed(M + N)
    wide = torch.randn(M, k1 + 64, generator=g).to(dt).to(DEV)
    A = wide[:, :k1]                      # lda = k1 + 64
    A2 = torch.randn(M, k2, generator=g).to(dt).to(DEV)
    W = (torch.randn(N, k1 + k2, generator=g) / (k1 + k2) ** 0.5).to(dt).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    ref = scale * (A.float() @ W[:, :k1].float().t()) + A2.float() @ W[:, k1:].float().t() + b
    out = torch.empty(M, N, device=DEV, dtype=dt)
    hip.gemm_nt(hip.EPI_BIAS, A, W, M, N, k1 + k2, out, bias=b, A2=A2, k1=k1, seg_scale=scale)
    tol = 2 ** -8 if dt == torch.float16 else 2 ** -6
    assert (out.float() - ref).abs().max().item() <= tol * (1 + ref.abs().max().item())
    # same source twice (the output convolution's hi / lo weight halves)
    if k1 == k2:
        ref2 = A2.float() @ (W[:, :k1].float() + W[:, k1:].float()).t() + b
        hip.gemm_nt(hip.EPI_BIAS, A2, W, M, N, 2 * k1, out, bias=b, A2=A2, k1=k1)
        assert (out.float() - ref2).abs().max().item() <= tol * (1 + ref2.abs().max().item())
    with pytest.raises(RuntimeError, match='k1'):
        hip.gemm_nt(hip.EPI_BIAS, A, W, M, N, k1 + k2, out, bias=b, A2=A2, k1=k1 + 8)
    with pytest.raises(RuntimeError, match='f16'):
        hip.gemm_nt(hip.EPI_BIAS, A.bfloat16(), W.bfloat16(), M, N, k1 + k2, out.bfloat16(), bias=b, A2=A2.bfloat16(), k1=k1)
