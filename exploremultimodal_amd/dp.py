"""Data-parallel gradient averaging over RCCL (xGMI), overlapped with backward.

Replaces the reference's DistributedDataParallel wrap
(train/pretrain/multimodal.py:82-89, find_unused_parameters=True) and is the
reduce half of its DeepSpeed ZeRO path (conf/ds_stage/l2.yaml).  One process
per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm ("gloo" works for
the CPU tests).

Design (MI355X-first):
  * parameters are grouped into flat buckets that follow the backward order:
    one bucket per transformer Block (24-33 MB in bf16 for VLMo-Base) plus one for
    embeddings / final norm / heads;
  * the set of parameters that will receive a gradient in THIS pass is read off
    the autograd graph in prepare(loss) (static expert routing => identical on
    every rank), which is what DDP's find_unused_parameters does;
  * when the last expected gradient of a bucket has been accumulated
    (post-accumulate-grad hook) the bucket is packed into a flat comm buffer
    (bf16 by default: half the bytes per link) and its all-reduce is issued on a
    side stream while the backward of the next block runs;
  * finish() waits for the side stream, unpacks (1/world scaling fused into the
    unpack) and leaves p.grad as views of the flat fp32 bucket.
`reduce_scatter=True` keeps only this rank's 1/W slice of every bucket reduced
(ZeRO-2 gradient partition, ds_stage/l2.yaml) and exposes it as
`bucket.shard`; the full gradient is then NOT written back.

Transformer blocks of the HIP engine skip the pack step altogether: while a reducer
is attached (engine.GRAD_SINK), BlockFn.backward accumulates its weight gradients
straight into a persistent flat fp32 buffer owned by the reducer ("sink bucket"),
p.grad are views of it, and the whole buffer is cast to bf16 and all-reduced as soon
as the last backward call that contributes to it in this step has been enqueued.
Only parameters outside the blocks (embeddings, final norm, heads) use the hook path.
"""
import re

import torch
import torch.distributed as dist


class _Bucket:
    def __init__(self, name, params, device, comm_dtype, world):
        self.name, self.params = name, params
        self.offsets, n = [], 0
        for p in params:
            self.offsets.append(n)
            n += p.numel()
        self.numel = n
        self.padded = ((n + world * 8 - 1) // (world * 8)) * (world * 8)
        self.flat = torch.zeros(self.padded, dtype=torch.float32, device=device)
        self.comm = self.flat if comm_dtype == torch.float32 else torch.zeros(
            self.padded, dtype=comm_dtype, device=device)
        self.shard = None
        self.expected = 0
        self.pending = 0
        self.work = None
        self.launched = False
        self.used = []
        self.has_grad = False   # reduced in the current accumulation window (see GradReducer.prepare)
        self.had = []


class _Arena:
    """Contiguous storage for the sink buckets of ONE transformer block (its shared parameters and up to three
    experts): buckets that become ready together are reduced as one collective over their common range."""

    def __init__(self, capacity, device, comm_dtype):
        self.flat = torch.zeros(capacity, dtype=torch.float32, device=device)
        self.comm = self.flat if comm_dtype == torch.float32 else torch.empty(capacity, dtype=comm_dtype, device=device)
        self.used = 0


class _SinkBucket:
    """Flat fp32 gradient storage of one engine block call signature (see module docstring)."""

    def __init__(self, numel, device, comm_dtype, world, arena=None):
        self.numel = numel
        self.padded = ((numel + world * 8 - 1) // (world * 8)) * (world * 8)
        self.arena, self.offset = None, 0
        if arena is not None and arena.used + self.padded <= arena.flat.numel():
            self.arena, self.offset = arena, arena.used
            arena.used += self.padded
            self.flat = arena.flat[self.offset:self.offset + self.padded]
            self.comm = self.flat if comm_dtype == torch.float32 else arena.comm[self.offset:self.offset + self.padded]
        else:
            self.flat = torch.zeros(self.padded, dtype=torch.float32, device=device)
            self.comm = self.flat if comm_dtype == torch.float32 else torch.empty(
                self.padded, dtype=comm_dtype, device=device)
        self.expected = 0       # backward calls still to come in this step
        self.fresh = True       # zero before the first accumulation of the step
        self.reduced = False    # a collective has been issued for this step's contents
        self.work = None
        self.shard = None
        self.has_grad = False   # a collective has been issued for it in the current accumulation window


class _Enqueued:
    """Work handle of a collective issued through the C-ABI communicator: it is ordered by the stream it was enqueued
    on, there is nothing to wait for on the host."""
    def wait(self):
        return True


class GradReducer:
    def __init__(self, module, process_group=None, comm_dtype=None, reduce_scatter=False,
                 broadcast_params=True, engine_sink=True, comm=None):
        """comm: 'torch' (collectives through torch.distributed, backend nccl = RCCL) or 'native' (the library's own RCCL
        communicator, include/vlmo_hip.h vlmo_comm_*: the collectives are plain enqueues on the communication stream);
        default from VLMO_DP_COMM, else 'torch'.
        comm_dtype: torch.bfloat16 (half the bytes per link, one pack and one unpack pass over every bucket), torch.float32
        (the buckets are exchanged in place), or None = decided at start-up: timed with the other candidates when there is
        more than one rank (autotune), fp32 at world size 1 (nothing crosses a link, the two passes would be pure cost)."""
        import os
        self._auto_dtype = comm_dtype is None
        if comm_dtype is None:
            comm_dtype = torch.bfloat16
        self.rs_ag = os.environ.get('VLMO_DP_COLLECTIVE', 'all_reduce') == 'rs_ag'
        self.comm_mode = comm or os.environ.get('VLMO_DP_COMM', 'torch')
        if self.comm_mode not in ('torch', 'native'):
            raise ValueError(f"comm must be 'torch' or 'native', got {self.comm_mode!r}")
        self.native = None
        self.pg = process_group if process_group is not None else dist.group.WORLD
        self.world = dist.get_world_size(self.pg)
        self.rank = dist.get_rank(self.pg)
        self.reduce_scatter = reduce_scatter
        params = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        if not params:
            raise ValueError('no trainable parameters')
        self.device = params[0][1].device
        self.on_gpu = self.device.type == 'cuda'
        if not self.on_gpu:
            comm_dtype = torch.float32          # gloo has no bf16 sum on every build
        self.comm_dtype = comm_dtype
        groups = {}
        for n, p in params:
            m = re.search(r'blocks\.(\d+)\.', n)
            key = f'block{int(m.group(1)):03d}' if m else 'rest'
            groups.setdefault(key, []).append(p)
        # backward order: last block first, embeddings/rest last
        order = sorted((k for k in groups if k != 'rest'), reverse=True) + (['rest'] if 'rest' in groups else [])
        self.buckets = [_Bucket(k, groups[k], self.device, comm_dtype, self.world) for k in order]
        self._of = {}
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._of[p] = (b, i)
                p.register_post_accumulate_grad_hook(self._hook)
        self.comm_stream = None
        if self.on_gpu:
            # a stream that really runs beside the caller's stream and the weight-gradient stream (engine.pick_stream)
            from . import engine
            dev = self.device
            with torch.cuda.device(dev):
                busy = [torch.cuda.current_stream(dev), engine._side_stream(dev)]
                make = lambda: torch.cuda.Stream(device=dev)
                self.comm_stream = engine.pick_stream(dev, make, busy) if engine.PROBE_STREAMS else make()
        if self.comm_mode == 'native':
            if not self.on_gpu:
                raise RuntimeError("comm='native' is the RCCL communicator of the HIP library: GPU parameters only")
            from . import hip
            # bootstrap over the existing process group: rank 0 draws the id, everybody joins
            uid = [hip.comm_unique_id() if self.rank == 0 else None]
            src = dist.get_global_rank(self.pg, 0) if self.pg is not dist.group.WORLD else 0
            dist.broadcast_object_list(uid, src=src, group=self.pg)
            with torch.cuda.device(self.device):
                self.native = hip.comm_init(uid[0], self.rank, self.world)
        self._armed = False
        self.sinks = {}
        self.arenas = {}
        self._sink_params = set()
        if engine_sink and self.on_gpu:
            from . import engine
            engine.GRAD_SINK = self
        if broadcast_params and self.world > 1:
            self.sync_params(module)
        # Which form of the exchange is faster depends on the node (xGMI is point to point: a ring all-reduce is per-link
        # bound, reduce-scatter + all-gather use all links; SURVEY.md 5.8 estimates 4.6 vs 0.66 ms for VLMo-Base) and nobody
        # could measure it on the one-GPU development boxes: with more than one rank, time the candidates on a buffer of a
        # real bucket's size at start-up and keep the fastest.  Explicit choices (VLMO_DP_COLLECTIVE, comm=...) are kept.
        self.tuned = None
        if self.world > 1 and os.environ.get('VLMO_DP_AUTOTUNE', '1') != '0':
            # the library's own communicator joins the candidates only on request (VLMO_DP_TRY_NATIVE=1): it has run at
            # world size 1 only, and a second RCCL communicator whose creation went wrong on one rank would hang the job
            # at start-up -- not a risk to take by default for ~0.3 ms per step
            self.autotune(try_native=(comm is None and 'VLMO_DP_COMM' not in os.environ and self.on_gpu
                                      and os.environ.get('VLMO_DP_TRY_NATIVE', '0') == '1'),
                          try_collective='VLMO_DP_COLLECTIVE' not in os.environ,
                          try_dtype=self._auto_dtype and self.on_gpu)
        elif self._auto_dtype and self.world == 1:
            self._set_comm_dtype(torch.float32)

    # ------------------------------------------------------------ start-up choice of the exchange
    def _set_comm_dtype(self, dtype):
        """Switch the exchange dtype before any sink bucket exists (the hook buckets' wire buffers are re-made)."""
        if dtype == self.comm_dtype:
            return
        assert not self.sinks and not self.arenas, 'the exchange dtype is fixed once gradients have been accumulated'
        self.comm_dtype = dtype
        for b in self.buckets:
            b.comm = b.flat if dtype == torch.float32 else torch.zeros(b.padded, dtype=dtype, device=self.device)

    def _time_exchange(self, buf, reps, wire=None):
        """seconds per in-place sum of `buf` over the ranks with the current settings (max over `reps` after 2 warm-ups is
        not needed: the median of the timed repetitions; every rank times its own, the caller takes the max over ranks)."""
        import time
        ts = []
        for i in range(reps + 2):
            if self.on_gpu:
                with torch.cuda.stream(self.comm_stream):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    if wire is not None:            # fp32 bucket through a narrower wire buffer: pack -> sum -> unpack
                        self._pack(buf, wire)
                        self._c_all_reduce(wire).wait()
                        self._unpack_into(wire, buf)
                    else:
                        buf.mul_(1.0 / self.world)
                        self._c_all_reduce(buf).wait()
                    b.record()
                b.synchronize()
                t = a.elapsed_time(b) * 1e-3
            else:
                dist.barrier(group=self.pg)
                t0 = time.perf_counter()
                self._c_all_reduce(buf if wire is None else wire).wait()
                t = time.perf_counter() - t0
            if i >= 2:
                ts.append(t)
        ts.sort()
        return ts[len(ts) // 2]

    def autotune(self, try_native=False, try_collective=True, try_dtype=False, numel=None, reps=5):
        """Pick (communicator, collective form) by timing them on a buffer the size of the largest gradient bucket.  Every
        rank measures, the per-candidate MAX over the ranks decides, so all ranks make the same choice.  Returns the
        table of candidates -> seconds and stores it (with the choice) in self.tuned."""
        if self.world < 2:
            return None
        n = numel or max(b.padded for b in self.buckets)
        n = ((n + self.world * 8 - 1) // (self.world * 8)) * (self.world * 8)
        buf = torch.zeros(n, dtype=torch.float32 if self.on_gpu else self.comm_dtype, device=self.device)
        dtypes = [torch.bfloat16, torch.float32] if try_dtype else [self.comm_dtype]
        wires = {d: (None if d == buf.dtype else torch.zeros(n, dtype=d, device=self.device)) for d in dtypes}
        native0 = self.native
        made_native = None
        if try_native and self.native is None:
            # the library's own communicator joins the candidates; a failure to create it on ANY rank drops it everywhere
            ok = torch.ones(1, dtype=torch.int32, device=self.device)
            try:
                from . import hip
                uid = [hip.comm_unique_id() if self.rank == 0 else None]
                src = dist.get_global_rank(self.pg, 0) if self.pg is not dist.group.WORLD else 0
                dist.broadcast_object_list(uid, src=src, group=self.pg)
                with torch.cuda.device(self.device):
                    made_native = hip.comm_init(uid[0], self.rank, self.world)
            except Exception:       # noqa: BLE001 -- any failure means "not a candidate"
                ok.zero_()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.pg)
            if int(ok.item()) == 0 and made_native is not None:
                from . import hip
                hip.comm_destroy(made_native)
                made_native = None
        comms = [('torch', None)] if native0 is None else [('native', native0)]
        if made_native is not None:
            comms.append(('native', made_native))
        forms = ['all_reduce', 'rs_ag'] if (try_collective and n % self.world == 0) else ['rs_ag' if self.rs_ag else 'all_reduce']
        cands, times = [], []
        for cname, handle in comms:
            for form in forms:
                for d in dtypes:
                    self.native, self.rs_ag = handle, form == 'rs_ag'
                    cands.append((cname, form, handle, d))
                    times.append(self._time_exchange(buf, reps, wires[d]))
        tt = torch.tensor(times, dtype=torch.float64, device=self.device if self.on_gpu else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=self.pg)
        best = int(torch.argmin(tt).item())
        cname, form, handle, dsel = cands[best]
        self.native, self.rs_ag, self.comm_mode = handle, form == 'rs_ag', cname
        self._set_comm_dtype(dsel)
        if made_native is not None and handle is not made_native:
            from . import hip
            if self.on_gpu:
                torch.cuda.synchronize(self.device)
            hip.comm_destroy(made_native)
        name = lambda c, f, d: f'{c}/{f}' + (('/' + str(d).replace('torch.', '')) if len(dtypes) > 1 else '')
        self.tuned = {'elements': n, 'chosen': name(cname, form, dsel),
                      'candidates_ms': {name(c, f, d): round(float(t) * 1e3, 4) for (c, f, _, d), t in zip(cands, tt.tolist())}}
        if self.rank == 0:
            print(f'[GradReducer] exchange of a {n / 1e6:.1f} M-element gradient bucket over {self.world} ranks: '
                  f'{self.tuned["candidates_ms"]} ms -> {self.tuned["chosen"]}', flush=True)
        return self.tuned

    # optional timing of the exchange: event pairs on the communication stream around pack -> collective -> unpack
    timing = False

    def _t0(self):
        if self.timing and self.on_gpu:
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream(self.device))
            return e
        return None

    def _t1(self, e0):
        if e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(torch.cuda.current_stream(self.device))
            self.__dict__.setdefault('_tev', []).append((e0, e1))

    def comm_ms(self):
        """milliseconds the communication stream spent in pack / collective / unpack since timing was switched on
        (synchronises); resets the record."""
        if self.on_gpu:
            torch.cuda.synchronize(self.device)
        tot = sum(a.elapsed_time(b) for a, b in self.__dict__.get('_tev', []))
        self.__dict__['_tev'] = []
        return tot

    def agree_finite(self, *losses):
        """True when every given loss is finite on EVERY rank.  The reference's loop drops a non-finite task loss per rank
        (train/pretrain/multimodal.py:281-284), which DDP's fixed bucket order tolerates; this reducer issues a bucket's
        collective as soon as its last expected contribution has been enqueued, so ranks that run different sets of
        passes would issue different collective sequences.  Decide the drop together: ``if not red.agree_finite(l): skip``."""
        ok = torch.stack([torch.isfinite(l.detach()).all() for l in losses]).all().to(torch.float32).reshape(1)
        if self.world > 1:
            ok = ok.to(self.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.pg)
        return bool(ok.item())

    def describe(self):
        """What a scaling run needs to check this reducer: the size of the communicator as the COMMUNICATOR reports it, the
        form of the exchange, bytes on the wire per step and rank (before the ring / tree factor)."""
        ranks = self.world
        if self.native is not None:
            from . import hip
            ranks = hip.comm_count(self.native)
        return {'ranks_in_communicator': ranks, 'communicator': self.comm_mode, 'backend': dist.get_backend(self.pg),
                'collective': ('reduce_scatter' if self.reduce_scatter else ('rs_ag' if self.rs_ag else 'all_reduce')),
                'comm_dtype': str(self.comm_dtype).replace('torch.', ''), 'bytes_per_step': self.bytes_per_step(),
                'autotune': self.tuned}

    def close(self):
        """Detach from the engine (block gradients go back through autograd) and drop the buckets."""
        from . import engine
        if engine.GRAD_SINK is self:
            engine.GRAD_SINK = None
        self.sinks.clear()
        self.arenas.clear()
        self._sink_params.clear()
        self._armed = False
        if self.native is not None:
            from . import hip
            torch.cuda.synchronize(self.device)
            hip.comm_destroy(self.native)
            self.native = None

    # ------------------------------------------------------------ collectives
    # All of them are issued inside `with torch.cuda.stream(self.comm_stream)` on the GPU; the returned handle's
    # wait() orders that stream after the collective (torch) or is a no-op (native: same stream already).
    def _c_all_reduce(self, t):
        """Sum of a flat bucket over the ranks, in place.  rs_ag (VLMO_DP_COLLECTIVE=rs_ag): the same result as a
        reduce-scatter + all-gather pair on the bucket in place (what a ring all-reduce is made of, issued as the two
        collectives RCCL schedules over all xGMI links) -- an A/B switch for the first multi-GPU run."""
        split = self.rs_ag and self.world > 1 and t.numel() % self.world == 0
        n = t.numel() // self.world
        mine = t[self.rank * n:(self.rank + 1) * n] if split else None
        if self.native is not None:
            from . import hip
            if split:
                hip.comm_reduce_scatter(self.native, mine, t)
                hip.comm_all_gather(self.native, t, mine)
            else:
                hip.comm_all_reduce(self.native, t)
            return _Enqueued()
        if split:
            dist.reduce_scatter_tensor(mine, t, group=self.pg)
            return dist.all_gather_into_tensor(t, mine, group=self.pg, async_op=True)
        return dist.all_reduce(t, group=self.pg, async_op=True)

    def _c_reduce_scatter(self, out, src):
        if self.native is not None:
            from . import hip
            hip.comm_reduce_scatter(self.native, out, src)
            return _Enqueued()
        return dist.reduce_scatter_tensor(out, src, group=self.pg, async_op=True)

    def all_gather(self, out, src):
        """out [world * n] <- every rank's src [n] (src may be out's own slice), on the CURRENT stream."""
        if self.native is not None:
            from . import hip
            hip.comm_all_gather(self.native, out, src)
        else:
            dist.all_gather_into_tensor(out, src, group=self.pg)

    def all_reduce_now(self, t):
        """in-place sum on the CURRENT stream (small tensors: the squared gradient norm of the sharded step)."""
        if self.native is not None:
            from . import hip
            hip.comm_all_reduce(self.native, t)
        else:
            dist.all_reduce(t, group=self.pg)

    def _pack(self, flat, comm):
        """comm (bf16) = flat (fp32) / world in ONE pass on the current stream."""
        if self.on_gpu and comm.dtype == torch.bfloat16 and flat.data_ptr() % 16 == 0 and comm.data_ptr() % 16 == 0:
            from . import hip
            hip.grad_pack(flat, comm, 1.0 / self.world)
        else:
            torch.mul(flat, 1.0 / self.world, out=comm)

    def _unpack_into(self, comm, flat):
        if self.on_gpu and comm.dtype == torch.bfloat16 and flat.data_ptr() % 16 == 0 and comm.data_ptr() % 16 == 0:
            from . import hip
            hip.grad_unpack(comm, flat)
        else:
            flat.copy_(comm)

    # ------------------------------------------------------------------ setup
    def sync_params(self, module):
        """rank 0's parameters and buffers everywhere (what DDP does at wrap time)."""
        with torch.no_grad():
            ts = list(module.parameters()) + list(module.buffers())
            for t in ts:
                dist.broadcast(t.data, src=dist.get_global_rank(self.pg, 0) if self.pg is not dist.group.WORLD else 0,
                               group=self.pg)
            # the broadcast wrote through .data: bump the version counters so that bf16 weight shadows cached by a
            # forward that ran before the reducer was built are re-cast (engine.ShadowCache keys on _version)
            torch.autograd.graph.increment_version(ts)

    def bytes_per_step(self):
        el = 2 if self.comm_dtype in (torch.bfloat16, torch.float16) else 4
        return sum(b.padded for b in self.buckets) * el

    # ------------------------------------------------------------ per-step API
    def prepare(self, loss):
        """Find the parameters this backward will touch (autograd graph walk) and arm the hooks."""
        used = set()
        seen, stack = set(), [loss.grad_fn] if loss.grad_fn is not None else []
        while stack:
            fn = stack.pop()
            if fn is None or fn in seen:
                continue
            seen.add(fn)
            v = getattr(fn, 'variable', None)
            if v is not None and v in self._of:
                used.add(v)
            for nxt, _ in fn.next_functions:
                stack.append(nxt)
        if getattr(self, '_window_done', True):
            # first backward of a new accumulation window (the previous finish() was an update step): nothing has been
            # reduced yet.  zero.ZeroAdam reads these flags: a bucket that receives no gradient in this window (every pass
            # feeding it dropped, modality absent) must not be stepped with the PREVIOUS window's shard.
            for b in self.buckets:
                b.has_grad, b.had = False, [False] * len(b.params)
            for sb in self.sinks.values():
                sb.has_grad = False
            self._window_done = False
        for b in self.buckets:
            # parameters fed by an engine sink bucket never fire the accumulate hook
            b.used = [(p in used) and (id(p) not in self._sink_params) for p in b.params]
            b.expected = b.pending = sum(b.used)
            b.work, b.launched = None, False
        # Engine sinks: forward() counted one expected backward call per grad-enabled block call (expect()), but a
        # pass whose output does not reach `loss` (a non-finite task loss dropped from the sum as in
        # train/pretrain/multimodal.py:281-284, an objective that returned python 0.) never runs its backward.
        # The graph says which engine nodes WILL run: count those instead, so that a bucket is reduced after its
        # last real contribution and never left holding local, un-averaged gradients.
        counts = {}
        for fn in seen:
            groups = getattr(fn, 'sink_groups', None)
            if groups is None or getattr(fn, 'sink', None) is not self:
                continue
            flat = groups if (groups and isinstance(groups[0], tuple)) else [g_ for blk in groups for g_ in blk]
            for g_ in flat:
                k = self._key(g_)
                counts[k] = counts.get(k, 0) + 1
        for k, sb in self.sinks.items():
            sb.expected = counts.get(k, 0)
        self._pending_expect = {k: c for k, c in counts.items() if k not in self.sinks}
        self._armed = True

    # ---- engine sink protocol (called from engine.BlockFn) -------------------------------------------
    @staticmethod
    def _key(group):
        return tuple(id(p) for p in group)

    def expect(self, group):
        key = self._key(group)
        self._sink_params.update(key)      # known before prepare(loss): these never take the hook path
        sb = self.sinks.get(key)
        if sb is not None:
            sb.expected += 1
        else:
            self._pending_expect = getattr(self, '_pending_expect', {})
            self._pending_expect[key] = self._pending_expect.get(key, 0) + 1

    def acquire(self, group, numel, device, arena_key=None, arena_numel=0, layout=None, lazy_zero=False):
        """Flat fp32 storage for one parameter group of a block call.  ``arena_key`` (any hashable naming the block)
        places the groups of one block next to each other (``arena_numel`` = room for all of them) so that
        release_all() can reduce them in one collective.  ``lazy_zero``: return ``(flat, fresh)`` and leave a fresh
        bucket (first contribution of the accumulation window) UN-zeroed: the caller either overwrites it (the engine's
        weight-gradient launch in store mode) or zeroes it itself."""
        key = self._key(group)
        sb = self.sinks.get(key)
        if sb is None:
            arena = None
            if arena_key is not None and not self.reduce_scatter:
                arena = self.arenas.get(arena_key)
                if arena is None:
                    pad = self.world * 8
                    arena = self.arenas[arena_key] = _Arena(arena_numel + 4 * pad, device, self.comm_dtype)
            sb = _SinkBucket(numel, device, self.comm_dtype, self.world, arena)
            sb.params = tuple(group)
            sb.layout = layout          # [(parameter, offset in the flat bucket)]: what zero.ZeroAdam shards
            sb.expected = getattr(self, '_pending_expect', {}).pop(key, 1)
            self.sinks[key] = sb
            self._sink_params.update(key)
        if sb.work is not None:         # previous step's reduction was never finish()ed: retire it first
            sb.work.wait()
            if self.on_gpu:
                torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
            sb.work, sb.fresh, sb.unpacked, sb.reduced = None, True, False, False
        fresh = sb.fresh
        if fresh:
            if not lazy_zero:
                sb.flat.zero_()
            sb.fresh = False
        return (sb.flat[:numel], fresh) if lazy_zero else sb.flat[:numel]

    def release(self, group):
        self.release_all([group])

    def release_all(self, groups, ready_event=None):
        """The backward call that fed these groups has been enqueued.  Buckets whose last contribution of the step
        this was are reduced now; neighbours in one block arena go out as ONE collective.  ready_event: native
        event (hip.event_create) recorded when these gradients are complete (engine.StackFn: the whole stack's
        backward is one call, so "everything enqueued so far on the current stream" would be the END of it)."""
        self._ready_event = ready_event
        ready = []
        for g in groups:
            sb = self.sinks[self._key(g)]
            sb.expected -= 1
            if sb.expected <= 0:
                ready.append(sb)
        ready.sort(key=lambda b: (id(b.arena), b.offset))
        i = 0
        while i < len(ready):
            j = i
            if ready[i].arena is not None:
                while (j + 1 < len(ready) and ready[j + 1].arena is ready[i].arena
                       and ready[j + 1].offset == ready[j].offset + ready[j].padded):
                    j += 1
            if j == i:
                self._launch_sink(ready[i])
            else:
                self._launch_range(ready[i:j + 1])
            i = j + 1

    def _comm_wait(self):
        """Order the communication stream after the producers of the gradients about to be reduced."""
        ev = getattr(self, '_ready_event', None)
        if ev is not None:
            from . import hip
            hip.stream_wait_event(self.comm_stream, ev)
        else:
            self.comm_stream.wait_stream(torch.cuda.current_stream(self.device))

    def _launch_range(self, sbs):
        """One pack / collective / unpack over the contiguous arena range of several ready buckets."""
        a = sbs[0].arena
        lo, hi = sbs[0].offset, sbs[-1].offset + sbs[-1].padded
        flat, comm = a.flat[lo:hi], a.comm[lo:hi]
        self._comm_wait()
        with torch.cuda.stream(self.comm_stream):
            t0 = self._t0()
            if comm.data_ptr() != flat.data_ptr():
                self._pack(flat, comm)
            elif self.world > 1:
                flat.mul_(1.0 / self.world)
            work = self._c_all_reduce(comm)
            work.wait()
            if comm.data_ptr() != flat.data_ptr():
                self._unpack_into(comm, flat)
            self._t1(t0)
        for sb in sbs:
            sb.work, sb.unpacked, sb.reduced, sb.has_grad = work, True, True, True

    def _launch_sink(self, sb):
        sb.reduced = sb.has_grad = True
        # everything on the communication stream (ordered after the block backward that just finished on the
        # current stream): the pack would otherwise sit in the dgrad chain's critical path 30 times per step
        if self.on_gpu:
            self._comm_wait()
            ctxm = torch.cuda.stream(self.comm_stream)
        else:
            from contextlib import nullcontext
            ctxm = nullcontext()
        with ctxm:
            t0 = self._t0()
            src = sb.comm
            if sb.comm is not sb.flat:
                self._pack(sb.flat, sb.comm)        # ONE pass: 1/world scaling + fp32 -> bf16 pack
            elif self.reduce_scatter:
                # fp32 communication + gradient partition: the reduce-scatter writes only the shard, so `flat` stays the
                # LOCAL accumulator that the next micro-step of a gradient-accumulation window adds to -- it must never
                # be scaled in place (it would carry g1 / W into the next pack: (g1 / W + g2) / W)
                if getattr(sb, 'rs_pack', None) is None:
                    sb.rs_pack = torch.empty_like(sb.flat)
                src = torch.mul(sb.flat, 1.0 / self.world, out=sb.rs_pack)
            elif self.world > 1:
                sb.flat.mul_(1.0 / self.world)
            if self.reduce_scatter:
                # persistent buffers: the ZeRO-2 optimizer (zero.ZeroAdam) keeps device tables of their addresses
                n = sb.padded // self.world
                if getattr(sb, 'shard_comm', None) is None:
                    sb.shard_comm = torch.empty(n, dtype=sb.comm.dtype, device=self.device)
                    sb.shard32 = sb.shard_comm if sb.comm.dtype == torch.float32 else torch.empty(
                        n, dtype=torch.float32, device=self.device)
                sb.work = self._c_reduce_scatter(sb.shard_comm, src)
            else:
                sb.work = self._c_all_reduce(sb.comm)
            if self.on_gpu:
                # the unpack follows its collective on the communication stream (work.wait() orders this stream
                # after the collective, it does not block the host): it overlaps the rest of the backward pass
                # instead of running 30 times on the main stream at the end of the step
                sb.work.wait()
                self._unpack_sink(sb)
                self._t1(t0)

    def _unpack_sink(self, sb):
        if self.reduce_scatter:
            if sb.shard32 is not sb.shard_comm:
                self._unpack_into(sb.shard_comm, sb.shard32)
            sb.shard = sb.shard32
        elif sb.comm is not sb.flat:
            self._unpack_into(sb.comm, sb.flat)  # ONE pass: bf16 -> fp32 unpack (already averaged)
        sb.unpacked = True

    def _hook(self, p):
        if not self._armed:
            return
        b, _ = self._of[p]
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b):
        """Pack (1/world folded in), reduce and unpack one hook bucket, all on the communication stream."""
        inv = 1.0 / self.world
        if self.on_gpu:
            self.comm_stream.wait_stream(torch.cuda.current_stream(self.device))
            ctxm = torch.cuda.stream(self.comm_stream)
        else:
            from contextlib import nullcontext
            ctxm = nullcontext()
        with ctxm:
            t0 = self._t0()
            # pack: grads of this pass, zeros for parameters this pass did not touch
            b.had = []
            for p, off, u in zip(b.params, b.offsets, b.used):
                v = b.comm[off:off + p.numel()]
                # gradient partition: p.grad stays the rank-local sum over the micro-steps of an accumulation window, so a
                # parameter an EARLIER micro-step touched is packed again even when this one did not use it
                has = (u or self.reduce_scatter) and p.grad is not None
                if has:
                    torch.mul(p.grad.reshape(-1), inv, out=v)
                    if self.on_gpu:
                        p.grad.record_stream(self.comm_stream)
                else:
                    v.zero_()       # e.g. img_mask_token in a pass without masked patches: backward gives None
                b.had.append(has)
            if self.reduce_scatter:
                n = b.padded // self.world
                if getattr(b, 'shard_comm', None) is None:
                    b.shard_comm = torch.empty(n, dtype=b.comm.dtype, device=self.device)
                    b.shard32 = b.shard_comm if b.comm.dtype == torch.float32 else torch.empty(
                        n, dtype=torch.float32, device=self.device)
                b.work = self._c_reduce_scatter(b.shard_comm, b.comm)
            else:
                b.work = self._c_all_reduce(b.comm)
            if self.on_gpu:
                b.work.wait()           # orders the communication stream after the collective (no host block)
                self._unpack(b)
                self._t1(t0)
        b.launched = b.has_grad = True

    def _unpack(self, b):
        if self.reduce_scatter:
            if b.shard32 is not b.shard_comm:
                self._unpack_into(b.shard_comm, b.shard32)
            b.shard = b.shard32
        elif b.comm is not b.flat:
            self._unpack_into(b.comm, b.flat)        # bf16 -> fp32 unpack (already averaged)
        b.unpacked = True

    def finish(self, accumulate=False):
        """Wait for every bucket and hand the averaged gradients back.  ``accumulate=True`` (a micro-step of gradient
        accumulation: ``update_grad=False`` in the reference's loop, train/pretrain/multimodal.py:260,316-323) keeps
        the buckets' contents, so the next backward ADDS to the averaged gradients: sum_r (g_prev + g_r) / W =
        g_prev + mean_r g_r because g_prev is identical on every rank.  Otherwise the sink buckets are re-zeroed
        before the next step's first contribution."""
        if not self._armed:
            raise RuntimeError('GradReducer.finish() without prepare()')
        self._armed = False
        self._window_done = not accumulate
        inv = 1.0 / self.world
        for b in self.buckets:
            if b.expected == 0:
                continue
            if not b.launched:        # a hook did not fire (grad was None): flush what we have
                self._launch(b)
            b.work.wait()
        self._ready_event = None
        for sb in self.sinks.values():
            # acquired in this step but never released to zero (a backward call that was expected did not run):
            # reduce what it holds now -- every rank walks its sinks in creation order, so ranks that agree on which
            # passes ran issue the same collectives
            if sb.work is None and not sb.fresh and not getattr(sb, 'reduced', False):
                self._launch_sink(sb)
        for sb in self.sinks.values():
            if sb.work is not None:
                sb.work.wait()
        if self.on_gpu:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        for sb in self.sinks.values():
            if sb.work is not None:
                if not getattr(sb, 'unpacked', False):
                    self._unpack_sink(sb)
                sb.work, sb.unpacked = None, False
            sb.fresh, sb.expected, sb.reduced = (not accumulate), 0, False
        for b in self.buckets:
            if b.expected == 0:
                continue
            if not getattr(b, 'unpacked', False):
                self._unpack(b)
            b.unpacked = False
            if self.reduce_scatter:
                continue
            for p, off, has in zip(b.params, b.offsets, b.had):
                if has:
                    p.grad = b.flat[off:off + p.numel()].view_as(p)
