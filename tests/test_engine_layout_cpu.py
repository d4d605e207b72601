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
