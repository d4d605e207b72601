"""Host logic of the engine that needs no GPU: the flat gradient-bucket layout that engine._fill_grads carves for
vlmo_stack_bwd must be the layout engine.shared_layout / expert_layout advertise to the ZeRO-2 optimizer
(zero.ZeroAdam re-homes the parameters by these offsets), including the hole the zero k-bias leaves."""
import types

import torch

from exploremultimodal_amd import engine


def test_bucket_layouts_match_the_carved_gradients():
    d, hid, nexp = 8, 32, 2
    mk = lambda *shape: torch.nn.Parameter(torch.zeros(*shape))
    shared = (mk(d), mk(d), mk(d), mk(d), mk(3 * d, d), mk(d), mk(d), mk(d, d), mk(d), mk(d), mk(d))
    experts = [(mk(hid, d), mk(hid), mk(d, hid), mk(d)) for _ in range(nexp)]
    shared_n = 6 * d + 3 * d * d + d * d + d + 3 * d
    exp_n = 2 * hid * d + hid + d
    flats = [torch.arange(shared_n, dtype=torch.float32)] + \
            [torch.arange(exp_n, dtype=torch.float32) + 1000 * (e + 1) for e in range(nexp)]

    class Desc:                 # stands in for the ctypes descriptor: accepts attribute and indexed writes
        def __init__(self):
            for n in ('dw1', 'db1', 'dw2', 'db2'):
                object.__setattr__(self, n, [0, 0])
    grads = engine._fill_grads(Desc(), flats, d, hid, nexp)
    assert len(grads) == 11 + 4 * nexp
    for (p, off), g in zip(sorted(engine.shared_layout(shared, d), key=lambda t: [id(x) for x in shared].index(id(t[0]))),
                           grads[:11]):
        assert g.shape == p.shape
        assert g.reshape(-1)[0].item() == float(off), (p.shape, off)          # flat holds arange: first element = offset
        assert g.numel() == p.numel()
    for e in range(nexp):
        for (p, off), g in zip(engine.expert_layout(experts[e], d, hid), grads[11 + 4 * e: 15 + 4 * e]):
            assert g.shape == p.shape and g.reshape(-1)[0].item() == 1000.0 * (e + 1) + off
    # q_bias and v_bias sit at the two ends of the 3d-wide qkv-bias slot: the d elements between them have no parameter
    lay = dict((id(p), off) for p, off in engine.shared_layout(shared, d))
    assert lay[id(shared[6])] - lay[id(shared[5])] == 2 * d
    covered = sum(p.numel() for p in shared)
    assert shared_n - covered == d


def test_wgrad_batch_by_tile_count(monkeypatch):
    """engine.wgrad_batch_for: default two blocks per launch; VLMO_WGRAD_BATCH=0 = the count whose tiles fill whole rounds."""
    from exploremultimodal_amd import engine
    assert engine.wgrad_batch_for(768, 3072) == engine.WGRAD_BATCH == 2
    monkeypatch.setattr(engine, 'WGRAD_BATCH', 0)
    assert engine.wgrad_batch_for(768, 3072) == 2        # 108 tiles per block: 216 = one round of 256 CUs
    assert engine.wgrad_batch_for(1024, 4096) == 4       # 192 per block: 768 = exactly three rounds


def test_weight_gradient_stream_rule(monkeypatch):
    """engine._use_side_stream: forced by VLMO_OVERLAP_WGRAD (engine.OVERLAP_WGRAD True / False), else the side stream
    under a gradient reducer and for passes of fewer than ONE_STREAM_ROWS rows, the caller's stream otherwise."""
    sink = object()
    monkeypatch.setattr(engine, 'OVERLAP_WGRAD', None)
    assert engine._use_side_stream(None, 64 * 261) is False          # VLMo-Base at 64 pairs: one stream
    assert engine._use_side_stream(None, 32 * 261) is True           # VLMo-Large at 32 pairs: 8 352 rows
    assert engine._use_side_stream(None, engine.ONE_STREAM_ROWS) is False
    assert engine._use_side_stream(sink, 64 * 261) is True           # a reducer waits on per-block grad_ready events
    monkeypatch.setattr(engine, 'OVERLAP_WGRAD', False)
    assert engine._use_side_stream(sink, 100) is False
    monkeypatch.setattr(engine, 'OVERLAP_WGRAD', True)
    assert engine._use_side_stream(None, 1 << 20) is True


def test_dvae_output_convolution_row_parts(monkeypatch):
    """dvae.Encoder._row_parts: the rows of a last dispatch round of 256 x 256 tiles that is at most a quarter full go
    out as a second launch of 128 x 128 tiles; everything else stays one launch with the library's own tile choice."""
    from exploremultimodal_amd import dvae
    rp = dvae.Encoder._row_parts
    monkeypatch.setattr(dvae, 'OUTPUT_ROW_SPLIT', True)
    assert rp(64 * 196, 8192) == [(0, 12288, 3), (12288, 12544, 0)]          # 1 568 tiles = 6 rounds + 32
    assert rp(32 * 196, 8192) == [(0, 6144, 3), (6144, 6272, 0)]
    assert rp(2 * 196, 8192) == [(0, 392, -1)]                               # less than one round
    assert rp(8 * 256, 8192) == [(0, 2048, -1)]                              # exactly one round
    assert rp(8 * 256 + 3 * 256, 8192) == [(0, 2816, -1)]                    # last round 96 of 256 tiles: kept
    for M, N in ((12544, 8192), (6272, 8192), (5000, 2048), (70000, 512)):
        parts = rp(M, N)
        assert parts[0][0] == 0 and parts[-1][1] == M
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        if len(parts) == 2:
            r = parts[0][1]
            assert r % 256 == 0 and (r // 256) * -(-N // 256) % 256 == 0     # the big launch fills whole rounds
    monkeypatch.setattr(dvae, 'OUTPUT_ROW_SPLIT', False)
    assert rp(64 * 196, 8192) == [(0, 12544, -1)]
