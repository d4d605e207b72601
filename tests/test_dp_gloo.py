"""world_size-2 CPU (gloo) tests of the gradient reducer: averaged gradients equal the
mean of the per-rank gradients, parameters that a pass does not touch keep grad=None
(find_unused_parameters semantics of multimodal.py:82-89), reduce-scatter mode returns
this rank's slice of the averaged flat bucket."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.embed = nn.Linear(8, 16)
        self.blocks = nn.ModuleList([nn.ModuleDict({'a': nn.Linear(16, 16), 'unused': nn.Linear(16, 16)})
                                     for _ in range(3)])
        self.norm = nn.LayerNorm(16)

    def forward(self, x):
        x = self.embed(x)
        for b in self.blocks:
            x = x + torch.tanh(b['a'](x))
        return self.norm(x)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from exploremultimodal_amd.dp import GradReducer
        torch.manual_seed(100 + rank)            # different init per rank: sync_params must fix it
        model = Tiny()
        red = GradReducer(model, reduce_scatter=(mode == 'rs'))
        ref = Tiny()
        ref.load_state_dict(model.state_dict())
        w0 = [torch.zeros_like(p) for p in model.parameters()]
        for w, p in zip(w0, model.parameters()):
            w.copy_(p.detach())
            dist.broadcast(w, src=0)
        assert all(torch.equal(w, p.detach()) for w, p in zip(w0, model.parameters())), 'params not synced'
        for step in range(2):
            g = torch.Generator().manual_seed(7 + rank + 10 * step)
            x = torch.randn(5, 8, generator=g)
            for m in (model, ref):
                for p in m.parameters():
                    p.grad = None
            loss = model(x).square().mean()
            red.prepare(loss)
            loss.backward()
            red.finish()
            ref(x).square().mean().backward()
            for (n, p), pr in zip(model.named_parameters(), ref.parameters()):
                if 'unused' in n:
                    assert p.grad is None and pr.grad is None, n
                    continue
                want = pr.grad.clone()
                dist.all_reduce(want)
                want /= world
                if mode == 'rs':
                    continue
                assert torch.allclose(p.grad, want, atol=1e-6), (n, (p.grad - want).abs().max())
            if mode == 'rs':
                for b in red.buckets:
                    full = torch.zeros(b.padded)
                    for p, off, u in zip(b.params, b.offsets, b.used):
                        if u:
                            pr = dict(ref.named_parameters())[
                                [n for n, q_ in model.named_parameters() if q_ is p][0]]
                            full[off:off + p.numel()] = pr.grad.reshape(-1)
                    dist.all_reduce(full)
                    full /= world
                    n = b.padded // world
                    assert torch.allclose(b.shard, full[rank * n:(rank + 1) * n], atol=1e-6), b.name
        q.put((rank, 'ok'))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _sink_worker(rank, world, port, q):
    """The engine-sink protocol (expect / acquire / release / finish) that BlockFn drives, on CPU tensors:
    two 'passes' accumulate into one bucket before it is reduced; a second step re-zeroes it."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from exploremultimodal_amd.dp import GradReducer
        model = Tiny()
        red = GradReducer(model, engine_sink=False)
        group = tuple(model.blocks[0]['a'].parameters())
        other = tuple(model.blocks[1]['a'].parameters())
        n = sum(p.numel() for p in group)
        for step in range(2):
            red.expect(group)
            red.expect(group)          # the block is used by two passes of this step
            red.expect(other)
            red._armed = True
            for k in range(2):
                flat = red.acquire(group, n, torch.device('cpu'))
                flat += (rank + 1) * (k + 1) * (step + 1)
                red.release(group)
            f2 = red.acquire(other, n, torch.device('cpu'))
            f2 += 10.0 * (rank + 1)
            red.release(other)
            red.finish()
            want = sum((r + 1) * 3 * (step + 1) for r in range(world)) / world
            got = red.sinks[red._key(group)].flat[:n]
            assert torch.allclose(got, torch.full_like(got, want)), (step, got[:3], want)
            got2 = red.sinks[red._key(other)].flat[:n]
            assert torch.allclose(got2, torch.full_like(got2, 10.0 * sum(r + 1 for r in range(world)) / world))
        q.put((rank, 'ok'))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


class _FakeEngineFn(torch.autograd.Function):
    """Stand-in for engine.StackFn on CPU: registers its sink groups in forward, accumulates `scale` into the sink
    bucket in backward (what vlmo_stack_bwd does with the weight gradients)."""

    @staticmethod
    def forward(ctx, x, red, groups, scale):
        ctx.sink, ctx.sink_groups, ctx.scale = red, [groups], scale
        for g in groups:
            red.expect(g)
        return x * 1.0

    @staticmethod
    def backward(ctx, dy):
        for g in ctx.sink_groups[0]:
            n = sum(p.numel() for p in g)
            flat = ctx.sink.acquire(g, n, torch.device('cpu'))
            flat += ctx.scale
            for p in g:
                if p.grad is None:
                    p.grad = flat[:p.numel()].view_as(p)   # placeholder view, as the engine installs
        ctx.sink.release_all(ctx.sink_groups[0])
        return dy, None, None, None


def _sink_edge_worker(rank, world, port, q):
    """(a) a pass whose output is dropped from the loss (train/pretrain/multimodal.py:281-284 drops non-finite task
    losses): prepare(loss) must count only the engine nodes that will run, else the bucket keeps local gradients;
    (b) gradient accumulation (update_grad=False micro-steps): finish(accumulate=True) keeps the averaged bucket and
    the next backward adds to it."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from exploremultimodal_amd.dp import GradReducer
        model = Tiny()
        red = GradReducer(model, engine_sink=False)
        group = tuple(model.blocks[0]['a'].parameters())
        n = sum(p.numel() for p in group)
        mean = lambda f: sum(f(r) for r in range(world)) / world
        for step in range(2):
            x = torch.ones(3, requires_grad=True)
            y1 = _FakeEngineFn.apply(x, red, [group], float(rank + 1))
            y2 = _FakeEngineFn.apply(x, red, [group], 100.0 * (rank + 1))      # this pass is dropped from the loss
            loss = y1.sum()
            red.prepare(loss)
            loss.backward()
            red.finish()
            got = red.sinks[red._key(group)].flat[:n]
            want = mean(lambda r: r + 1.0)
            assert torch.allclose(got, torch.full_like(got, want)), ('dropped pass', step, got[:3], want)
            del y2
        # accumulation: two micro-steps without an optimizer step, then the update step
        tot = 0.0
        for micro, acc in enumerate([True, True, False]):
            x = torch.ones(3, requires_grad=True)
            y = _FakeEngineFn.apply(x, red, [group], float((rank + 1) * (micro + 1)))
            loss = y.sum()
            red.prepare(loss)
            loss.backward()
            red.finish(accumulate=acc)
            tot += mean(lambda r: (r + 1.0) * (micro + 1))
            got = red.sinks[red._key(group)].flat[:n]
            assert torch.allclose(got, torch.full_like(got, tot)), ('accumulate', micro, got[:3], tot)
        # after the update step the bucket starts from zero again
        x = torch.ones(3, requires_grad=True)
        loss = _FakeEngineFn.apply(x, red, [group], 1.0).sum()
        red.prepare(loss)
        loss.backward()
        red.finish()
        got = red.sinks[red._key(group)].flat[:n]
        assert torch.allclose(got, torch.ones_like(got)), got[:3]
        q.put((rank, 'ok'))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _rs_accumulate_worker(rank, world, port, q):
    """Gradient partition (reduce_scatter=True) + gradient accumulation with fp32 communication (what gloo and
    `--comm-dtype fp32` use): the sink's flat buffer is the rank-LOCAL accumulator there, so packing must not scale it in
    place -- after every micro-step the shard is this rank's slice of mean_r(sum of the micro-steps so far)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from exploremultimodal_amd.dp import GradReducer
        model = Tiny()
        red = GradReducer(model, reduce_scatter=True, engine_sink=False)
        group = tuple(model.blocks[0]['a'].parameters())
        n = sum(p.numel() for p in group)
        mean = lambda f: sum(f(r) for r in range(world)) / world
        for window in range(2):
            tot = 0.0
            for micro, acc in enumerate([True, True, False]):
                x = torch.ones(3, requires_grad=True)
                y = _FakeEngineFn.apply(x, red, [group], float((rank + 1) * (micro + 1) * (window + 1)))
                loss = y.sum()
                red.prepare(loss)
                loss.backward()
                red.finish(accumulate=acc)
                tot += mean(lambda r: (r + 1.0) * (micro + 1) * (window + 1))
                sb = red.sinks[red._key(group)]
                assert sb.has_grad
                lo = rank * (sb.padded // world)
                valid = max(0, min(n - lo, sb.shard.numel()))
                assert torch.allclose(sb.shard[:valid], torch.full((valid,), tot)), (window, micro, sb.shard[:3], tot)
            for p in group:
                p.grad = None
        q.put((rank, 'ok'))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_reduce_scatter_with_accumulation_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rs_accumulate_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == 'ok', f'rank {rank}: {msg}'


def test_engine_sink_dropped_pass_and_accumulation_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sink_edge_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == 'ok', f'rank {rank}: {msg}'


def test_engine_sink_protocol_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sink_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == 'ok', f'rank {rank}: {msg}'


@pytest.mark.parametrize('mode', ['allreduce', 'rs', 'rs_ag'])
def test_grad_reducer_world2_gloo(mode, monkeypatch):
    """rs_ag: VLMO_DP_COLLECTIVE=rs_ag replaces every all-reduce by reduce-scatter + all-gather on the bucket in
    place (the A/B switch for the first multi-GPU run): same averaged gradients."""
    if mode == 'rs_ag':
        monkeypatch.setenv('VLMO_DP_COLLECTIVE', 'rs_ag')
        mode = 'allreduce'
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == 'ok', f'rank {rank}: {msg}'


def _gather_worker(rank, world, port, q):
    """GatherLayer (objectives.py:392-426) + the global-negatives ITC arithmetic of objectives.py:99-108: the mean over
    ranks of the per-rank losses and every rank's feature gradient equal the single-process computation on the
    concatenated batch."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import torch.nn.functional as F
        from exploremultimodal_amd.objectives import GatherLayer
        bs, dim, temp = 4, 6, 3.0
        g = torch.Generator().manual_seed(0)
        I = F.normalize(torch.randn(world * bs, dim, generator=g), dim=1)
        T = F.normalize(torch.randn(world * bs, dim, generator=g), dim=1)
        i_feat = I[rank * bs:(rank + 1) * bs].clone().requires_grad_(True)
        t_feat = T[rank * bs:(rank + 1) * bs].clone().requires_grad_(True)
        i_all = torch.roll(GatherLayer.apply(i_feat, None, rank), -bs * rank, 0)
        t_all = torch.roll(GatherLayer.apply(t_feat, None, rank), -bs * rank, 0)
        tgt = torch.arange(bs)
        loss = (F.cross_entropy(i_feat @ t_all.t() * temp, tgt) + F.cross_entropy(t_feat @ i_all.t() * temp, tgt)) / 2
        loss.backward()
        # single-process reference: the DDP-averaged objective is the mean of the per-rank losses
        Ir, Tr = I.clone().requires_grad_(True), T.clone().requires_grad_(True)
        tg = torch.arange(world * bs)
        ref = (F.cross_entropy(Ir @ Tr.t() * temp, tg) + F.cross_entropy(Tr @ Ir.t() * temp, tg)) / 2
        ref.backward()
        tot = loss.detach().clone()
        dist.all_reduce(tot)
        assert torch.allclose(tot / world, ref.detach(), atol=1e-6), (tot / world, ref)
        # gradient of the MEAN loss w.r.t. this rank's features = this rank's autograd gradient / world
        assert torch.allclose(i_feat.grad / world, Ir.grad[rank * bs:(rank + 1) * bs], atol=1e-6)
        assert torch.allclose(t_feat.grad / world, Tr.grad[rank * bs:(rank + 1) * bs], atol=1e-6)
        q.put((rank, 'ok'))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_gather_layer_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == 'ok', f'rank {rank}: {msg}'


def _zero_worker(rank, world, port, q):
    """ZeRO-2 (conf/ds_stage/l2.yaml): reduce-scattered gradients -> sharded AdamW on this rank's slice ->
    all-gather of the updated parameters must equal the REPLICATED step (every rank running torch.optim.AdamW on the
    all-reduced gradients) parameter for parameter, with global-norm clipping, name-group lr / weight decay, an
    engine-style sink bucket (with the hole the qkv bias leaves) beside the hook buckets.  On CPU the two device
    kernels of ZeroAdam are replaced by their torch restatement (test infrastructure; the product path has none)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from exploremultimodal_amd.dp import GradReducer
        from exploremultimodal_amd.zero import ZeroAdam

        class CpuZero(ZeroAdam):
            def _views(self, tab):
                import ctypes
                out = []
                for (pp, gp, mp_, vp, n, gi, _p) in tab['ent']:
                    mk = lambda a: torch.frombuffer((ctypes.c_float * n).from_address(a), dtype=torch.float32)
                    out.append((mk(pp), mk(gp), mk(mp_), mk(vp), gi))
                return out

            def _local_sqnorm(self, tab):
                return sum((g.double() ** 2).sum() for _, g, _, _, _ in self._views(tab)).float()

            def _apply(self, tab, a, ctl):
                if ctl is not None and float(ctl[2]) != 0:
                    return
                gs = float(ctl[1]) if ctl is not None else 1.0
                for p, g, m, v, gi in self._views(tab):
                    grp = self.param_groups[gi]
                    g = g * gs
                    m.mul_(a.beta1).add_(g, alpha=1 - a.beta1)
                    v.mul_(a.beta2).addcmul_(g, g, value=1 - a.beta2)
                    u = (m * a.inv_bc1) / ((v * a.inv_bc2).sqrt() + a.eps) + grp['weight_decay'] * p
                    p.sub_(grp['lr'] * u)

        torch.manual_seed(5)
        model, ref = Tiny(), Tiny()
        ref.load_state_dict(model.state_dict())
        red = GradReducer(model, reduce_scatter=True, engine_sink=False)
        # block 1's linear goes through the engine-sink protocol with a hole in its flat layout
        sink_lin = model.blocks[1]['a']
        group = (sink_lin.weight, sink_lin.bias)
        layout = [(sink_lin.weight, 0), (sink_lin.bias, sink_lin.weight.numel() + 7)]
        sink_n = sink_lin.weight.numel() + 7 + sink_lin.bias.numel()
        red._sink_params.update(id(p) for p in group)

        def groups_of(m):
            dec = [p for n, p in m.named_parameters() if p.dim() > 1]
            nodec = [p for n, p in m.named_parameters() if p.dim() <= 1]
            return [{'params': dec, 'lr': 3e-2, 'weight_decay': 0.1}, {'params': nodec, 'lr': 1e-2, 'weight_decay': 0.0}]
        opt = CpuZero(red, groups_of(model), betas=(0.9, 0.98), eps=1e-5)    # well-conditioned m / (sqrt(v) + eps)
        ropt = torch.optim.AdamW(groups_of(ref), betas=(0.9, 0.98), eps=1e-5)
        for step in range(3):
            g = torch.Generator().manual_seed(11 + rank + 10 * step)
            x = torch.randn(6, 8, generator=g)
            for m in (model, ref):
                for p in m.parameters():
                    p.grad = None
            loss = model(x).square().mean()
            red.prepare(loss)
            loss.backward()
            # hand block 1's gradients to the sink bucket as the engine would
            flat = red.acquire(group, sink_n, torch.device('cpu'), layout=layout)
            for p, off in layout:
                flat[off:off + p.numel()] += p.grad.reshape(-1)
                p.grad = None
            red.release_all([group])
            red.finish()
            opt.step(clip_grad=0.5)
            ref(x).square().mean().backward()
            for p in ref.parameters():
                if p.grad is not None:
                    dist.all_reduce(p.grad)
                    p.grad /= world
            torch.nn.utils.clip_grad_norm_([p for p in ref.parameters() if p.grad is not None], 0.5)
            ropt.step()
            for (n, p), pr in zip(model.named_parameters(), ref.parameters()):
                if 'unused' in n:
                    continue
                assert torch.allclose(p.detach(), pr.detach(), rtol=1e-4, atol=5e-5), (step, n, (p - pr).abs().max())  # updates are lr-sized (3e-2): 5e-5 is fp32 summation-order noise through m / sqrt(v)
        # the moments exist only for this rank's slices
        total = sum(pt.padded for pt in opt.parts)
        assert sum(pt.m.numel() for pt in opt.parts) * world == total

        # A step in which NOTHING feeds the sink bucket (every pass that uses block 1 dropped from the loss): its
        # persistent shard still holds the previous step's gradient and must not be applied again -- torch.optim skips
        # grad=None parameters (no update, no weight decay, no moment decay), and so must the sharded step.
        def one_step(step, feed_sink, mdl, rd, op, rmdl, rop):
            g = torch.Generator().manual_seed(11 + rank + 10 * step)
            x = torch.randn(6, 8, generator=g)
            for m in (mdl, rmdl):
                for p in m.parameters():
                    p.grad = None
            lin = mdl.blocks[1]['a']
            grp = (lin.weight, lin.bias)
            lay = [(lin.weight, 0), (lin.bias, lin.weight.numel() + 7)]
            loss = mdl(x).square().mean()
            rd.prepare(loss)
            loss.backward()
            if feed_sink:
                flat = rd.acquire(grp, sink_n, torch.device('cpu'), layout=lay)
                for p, off in lay:
                    flat[off:off + p.numel()] += p.grad.reshape(-1)
            for p in grp:
                p.grad = None
            if feed_sink:
                rd.release_all([grp])
            rd.finish()
            op.step(clip_grad=0.5)
            rmdl(x).square().mean().backward()
            if not feed_sink:
                for p in rmdl.blocks[1]['a'].parameters():
                    p.grad = None
            for p in rmdl.parameters():
                if p.grad is not None:
                    dist.all_reduce(p.grad)
                    p.grad /= world
            torch.nn.utils.clip_grad_norm_([p for p in rmdl.parameters() if p.grad is not None], 0.5)
            rop.step()
            for (n, p), pr in zip(mdl.named_parameters(), rmdl.parameters()):
                if 'unused' not in n:
                    assert torch.allclose(p.detach(), pr.detach(), rtol=1e-4, atol=5e-5), (step, n, (p - pr).abs().max())

        before = [p.detach().clone() for p in group]
        one_step(3, False, model, red, opt, ref, ropt)
        assert all(torch.equal(b_, p.detach()) for b_, p in zip(before, group)), 'a bucket without a gradient was stepped'
        one_step(4, True, model, red, opt, ref, ropt)

        # resume in the natural order: build, load_state_dict (before any backward has created the gradient buckets), train
        sd_model = {k: v.clone() for k, v in model.state_dict().items()}
        sd_opt = opt.state_dict()
        assert all('key' in st for st in sd_opt['parts'])
        red.close()
        model2 = Tiny()
        model2.load_state_dict(sd_model)
        red2 = GradReducer(model2, reduce_scatter=True, engine_sink=False)
        lin2 = model2.blocks[1]['a']
        red2._sink_params.update(id(p) for p in (lin2.weight, lin2.bias))
        opt2 = CpuZero(red2, groups_of(model2), betas=(0.9, 0.98), eps=1e-5)
        opt2.load_state_dict(sd_opt)
        assert opt2.parts is None and opt2.step_count == opt.step_count
        one_step(5, True, model2, red2, opt2, ref, ropt)       # `ref` / `ropt` simply keep going
        bad = dict(sd_opt)
        bad['parts'] = list(reversed(sd_opt['parts']))
        opt3 = CpuZero(red2, groups_of(model2), betas=(0.9, 0.98), eps=1e-5)
        opt3.parts = opt2.parts
        try:
            opt3.load_state_dict(bad)
            raise AssertionError('a permuted ZeroAdam state was accepted')
        except ValueError:
            pass
        q.put((rank, 'ok'))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_zero2_sharded_step_equals_replicated_step_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_zero_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == 'ok', f'rank {rank}: {msg}'


def _tune_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    os.environ.pop('VLMO_DP_COLLECTIVE', None)
    os.environ.pop('VLMO_DP_AUTOTUNE', None)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from exploremultimodal_amd.dp import GradReducer
        torch.manual_seed(5)
        model = Tiny()
        red = GradReducer(model)
        assert red.tuned is not None and set(red.tuned['candidates_ms']) == {'torch/all_reduce', 'torch/rs_ag'}, red.tuned
        assert red.tuned['chosen'] in red.tuned['candidates_ms']
        # every rank made the same choice (the per-candidate MAX over the ranks decides)
        got = [None] * world
        dist.all_gather_object(got, (red.tuned['chosen'], red.rs_ag, red.tuned['candidates_ms']))
        assert all(g_ == got[0] for g_ in got), got
        d = red.describe()
        assert d['ranks_in_communicator'] == world and d['collective'] in ('all_reduce', 'rs_ag') and d['bytes_per_step'] > 0
        assert d['autotune']['chosen'] == red.tuned['chosen']
        # ... and the exchange still averages
        g = torch.Generator().manual_seed(3 + rank)
        x = torch.randn(5, 8, generator=g)
        loss = model(x).square().mean()
        red.prepare(loss)
        loss.backward()
        red.finish()
        ref = Tiny()
        ref.load_state_dict(model.state_dict())
        ref(x).square().mean().backward()
        for (n, p), pr in zip(model.named_parameters(), ref.parameters()):
            if 'unused' in n:
                continue
            want = pr.grad.clone()
            dist.all_reduce(want)
            want /= world
            assert torch.allclose(p.grad, want, rtol=1e-5, atol=1e-6), n
        # an explicit choice is kept: no tuning of the collective form
        os.environ['VLMO_DP_COLLECTIVE'] = 'rs_ag'
        red2 = GradReducer(Tiny())
        assert red2.rs_ag and list(red2.tuned['candidates_ms']) == ['torch/rs_ag']
        q.put((rank, 'ok'))
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc() + str(e)))
    finally:
        dist.destroy_process_group()


def test_exchange_autotune_world2_gloo():
    """With more than one rank the reducer times all-reduce against reduce-scatter + all-gather on a bucket-sized buffer at
    start-up (the choice must not need the builder on a multi-GPU box), all ranks agree on the
    choice, GradReducer.describe() reports what a scaling run needs, and an explicit VLMO_DP_COLLECTIVE is kept."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tune_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == 'ok', f'rank {rank}: {msg}'
