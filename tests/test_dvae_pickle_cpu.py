"""dvae.load_model against the wire format of the published dVAE weights: a pickled ``dall_e.encoder.Encoder`` module
(dall_e/__init__.py:12-21).  tests/golden/dvae_encoder_pickle.pkl is the REFERENCE's Encoder saved by torch.save
(oracle/gen_golden.py `dvae_encoder_pickle`; parameter storages emptied, so it is the object graph: class paths,
attribute names and values, module tree).  No GPU needed: everything up to the engine's weight shadows is host code."""
import os

import torch

from exploremultimodal_amd import dvae, synth

PKL = os.path.join(os.path.dirname(__file__), 'golden', 'dvae_encoder_pickle.pkl')
KW = dict(n_hid=64, vocab_size=512, n_blk_per_group=2)


def _filled(enc, sd):
    with torch.no_grad():
        for name, p in enc.named_parameters():
            p.set_(sd[name].clone())
    return enc


def test_reference_pickle_loads_onto_the_mirror_classes():
    enc = dvae.load_model(PKL, device='cpu')
    assert type(enc) is dvae.Encoder
    assert 'dall_e' not in str(type(enc).__module__)
    import sys
    assert 'dall_e.encoder' not in sys.modules, 'load_model leaked its stand-in modules'
    ours = dvae.Encoder(**KW)
    # same module tree and parameter names as a mirror built from scratch (= the state-dict keys of the reference)
    assert [n for n, _ in enc.named_modules()] == [n for n, _ in ours.named_modules()]
    assert [n for n, _ in enc.named_parameters()] == [n for n, _ in ours.named_parameters()]
    for (n, a), (_, b) in zip(enc.named_modules(), ours.named_modules()):
        assert type(a) is type(b), n
    # every attribute the engine reads (dvae.Encoder._features / forward) is present with the reference's value
    for k in ('group_count', 'n_hid', 'n_blk_per_group', 'input_channels', 'vocab_size'):
        assert getattr(enc, k) == getattr(ours, k), k
    for (n, a), (_, b) in zip(enc.named_modules(), ours.named_modules()):
        if isinstance(a, dvae.Conv2d):
            assert (a.n_in, a.n_out, a.kw, a.use_float16) == (b.n_in, b.n_out, b.kw, b.use_float16), n
        if isinstance(a, dvae.EncoderBlock):
            assert (a.n_in, a.n_out, a.n_hid) == (b.n_in, b.n_out, b.n_hid) and a.post_gain == b.post_gain, n
            assert isinstance(a.id_path, dvae.Conv2d) == isinstance(b.id_path, dvae.Conv2d), n
    assert enc.blocks.output.conv.use_float16 is False          # the fp32 layer that decides the arg-max


def test_loaded_pickle_builds_the_same_weight_shadows():
    sd = synth.synth_dvae_state_dict(0, **KW)
    enc = _filled(dvae.load_model(PKL, device='cpu'), sd)
    ours = dvae.Encoder(**KW)
    ours.load_state_dict(sd, strict=True)
    for (n, a), (_, b) in zip(enc.named_modules(), ours.named_modules()):
        if not isinstance(a, dvae.Conv2d):
            continue
        wa, ba = a.shadow_split() if not a.use_float16 else a.shadow()
        wb, bb = b.shadow_split() if not b.use_float16 else b.shadow()
        assert torch.equal(wa, wb) and torch.equal(ba, bb), n


def test_dalle_vae_wrapper_reads_encoder_pkl(tmp_path):
    """models/modeling_discrete_vae.py:233-236: Dalle_VAE.load_model(model_dir) -> <dir>/encoder.pkl."""
    import shutil
    shutil.copy(PKL, tmp_path / 'encoder.pkl')
    vae = dvae.create_d_vae(str(tmp_path), 'dall-e', image_size=112, device='cpu')
    assert type(vae.encoder) is dvae.Encoder and vae.encoder.vocab_size == 512
