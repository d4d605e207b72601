"""C-ABI gradient exchange (include/vlmo_hip.h vlmo_comm_*, vlmo_grad_pack / unpack) on one rank: RCCL communicator
bootstrap, the three collectives as identities at world size 1, and the pack / unpack passes against torch."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_pack_unpack_match_torch():
    from exploremultimodal_amd import hip
    g = torch.Generator(device='cuda').manual_seed(5)
    for n in (8, 2048, 1_000_003 - 3, 7_340_032):        # multiples of 8 and a ragged tail below
        for tail in (0, 5):
            x = torch.randn(n + tail, device='cuda', generator=g) * 3
            out = torch.empty(n + tail, dtype=torch.bfloat16, device='cuda')
            hip.grad_pack(x, out, 0.125)
            assert torch.equal(out, (x * 0.125).to(torch.bfloat16))      # 0.125: exact scaling, one rounding
            back = torch.empty(n + tail, device='cuda')
            hip.grad_unpack(out, back)
            assert torch.equal(back, out.float())
    # a scale that is not a power of two: product rounded once, as torch.mul(out=bf16) does
    x = torch.randn(4096, device='cuda', generator=g)
    out = torch.empty(4096, dtype=torch.bfloat16, device='cuda')
    hip.grad_pack(x, out, 1.0 / 3)
    want = torch.empty_like(out)
    torch.mul(x, 1.0 / 3, out=want)
    assert torch.equal(out, want)
    with pytest.raises(RuntimeError, match='alignment'):
        hip.grad_pack(x[1:], out[1:], 1.0)


def test_communicator_single_rank_collectives():
    from exploremultimodal_amd import hip
    assert hip.comm_available()
    uid = hip.comm_unique_id()
    assert len(uid) == hip.COMM_ID_BYTES
    comm = hip.comm_init(uid, 0, 1)
    try:
        side = torch.cuda.Stream()
        for dt in (torch.float32, torch.bfloat16):
            x = torch.randn(100_000, device='cuda').to(dt)
            ref = x.clone()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                hip.comm_all_reduce(comm, x)
                shard = torch.empty_like(x)
                hip.comm_reduce_scatter(comm, shard, x)
                full = torch.zeros_like(x)
                hip.comm_all_gather(comm, full, shard)
            side.synchronize()
            assert torch.equal(x, ref) and torch.equal(shard, ref) and torch.equal(full, ref)
        with pytest.raises(RuntimeError, match='communicator'):
            hip.comm_all_reduce(0, x)
    finally:
        torch.cuda.synchronize()
        hip.comm_destroy(comm)
    with pytest.raises(RuntimeError, match='rank'):
        hip.comm_init(uid, 2, 1)
