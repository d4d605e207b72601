"""ZeRO-2 style training step: gradient partition + sharded AdamW + parameter all-gather.

Replaces the reference's DeepSpeed stage-2 path (conf/ds_stage/l2.yaml:1-6 -- ``stage: 2, reduce_scatter: true,
overlap_comm: true`` -- wired in train/pretrain/multimodal.py:61-79) on RCCL:

  * ``GradReducer(reduce_scatter=True)`` (dp.py) leaves every rank with its 1/W slice of each averaged flat
    gradient bucket (one bucket per transformer block group, one for embeddings / norms / heads);
  * ``ZeroAdam`` keeps the Adam moments ONLY for that slice, updates the matching slice of the fp32 master
    parameters with the same HIP multi-tensor kernel as ``optim.FusedAdam`` (the slice is cut into one table entry
    per parameter it overlaps, so the name-based lr / weight-decay groups of utils/optim_factory.py:22-90 keep
    working), then all-gathers the updated parameters.

To make slice updates and the all-gather operate on contiguous memory the parameters of a bucket are re-homed into
one flat fp32 buffer laid out exactly like the bucket's gradients (``p.data`` becomes a view; modules, state dicts
and the engine see the same tensors as before).  Optimizer state per rank: 8 bytes x parameters / W.

The global gradient norm for clipping is the all-reduced sum of the per-rank slice norms, so every rank applies the
same factor as ``torch.nn.utils.clip_grad_norm_`` would on the full averaged gradient.
"""
import torch
import torch.distributed as dist

from . import hip
from .optim import CHUNK


class _Part:
    """One gradient bucket as the optimizer sees it: flat fp32 parameters, this rank's slice, its moments."""

    def __init__(self, bucket, entries, world, rank, device):
        self.bucket = bucket
        self.padded = bucket.padded
        self.n = bucket.padded // world
        self.lo, self.hi = rank * self.n, (rank + 1) * self.n
        self.pflat = torch.zeros(bucket.padded, dtype=torch.float32, device=device)
        self.entries = entries              # [(param, offset, lr-group index)]
        with torch.no_grad():
            for p, off, _ in entries:
                self.pflat[off:off + p.numel()].copy_(p.detach().reshape(-1))
                p.data = self.pflat[off:off + p.numel()].view_as(p)
        self.m = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.v = torch.zeros(self.n, dtype=torch.float32, device=device)


class ZeroAdam:
    """``opt = ZeroAdam(reducer, param_groups, betas=..., eps=..., adam_w_mode=True)``;
    ``norm = opt.step(clip_grad=5.0)`` after ``reducer.finish()``.  ``param_groups`` as produced by
    ``optim.get_parameter_groups`` (dicts with 'params', 'lr', 'weight_decay').  The partition is built at the
    first step (the engine's gradient buckets exist once a backward has run)."""

    def __init__(self, reducer, param_groups, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 adam_w_mode=True, bias_correction=True):
        if not reducer.reduce_scatter:
            raise ValueError('ZeroAdam needs GradReducer(reduce_scatter=True)')
        self.reducer = reducer
        self.param_groups = []
        for g in param_groups:
            g = dict(g)
            g.setdefault('lr', lr)
            g.setdefault('weight_decay', weight_decay)
            g['params'] = list(g['params'])
            self.param_groups.append(g)
        self.betas, self.eps = tuple(betas), eps
        self.adam_w_mode, self.bias_correction = adam_w_mode, bias_correction
        self.step_count = 0
        self.parts = None
        self._tab = None
        self.last_ctl = None
        self._pending_state = None
        self._pstep = {}            # id(parameter) -> optimizer steps taken (torch.optim.AdamW keeps one per parameter)

    # ------------------------------------------------------------------------------------------ partition
    def _group_of(self):
        m = {}
        for gi, g in enumerate(self.param_groups):
            for p in g['params']:
                m[id(p)] = gi
        return m

    def _build(self):
        red = self.reducer
        gof = self._group_of()
        dev = red.device
        parts, owned = [], set()
        for sb in red.sinks.values():
            if sb.layout is None:
                raise RuntimeError('engine sink bucket without a parameter layout')
            ents = [(p, off, gof[id(p)]) for p, off in sb.layout if p.requires_grad and id(p) in gof]
            owned.update(id(p) for p, _, _ in ents)
            parts.append(_Part(sb, ents, red.world, red.rank, dev))
        for b in red.buckets:
            ents = [(p, off, gof[id(p)]) for p, off in zip(b.params, b.offsets)
                    if id(p) not in owned and id(p) not in red._sink_params and id(p) in gof]
            if ents:
                parts.append(_Part(b, ents, red.world, red.rank, dev))
        torch.autograd.graph.increment_version([p for pt in parts for p, _, _ in pt.entries])
        self.parts = parts
        self._tab = None
        if self._pending_state is not None:
            self._apply_state(self._pending_state)
            self._pending_state = None

    @staticmethod
    def _part_key(pt):
        """Identity of a partition entry across processes: the bucket's padded size and its parameters' (offset, numel)."""
        return [int(pt.padded)] + [[int(off), int(p.numel())] for p, off, _ in pt.entries]

    def _apply_state(self, sd):
        saved = sd['parts']
        if len(saved) != len(self.parts):
            raise ValueError(f'ZeroAdam state has {len(saved)} partition entries, this run built {len(self.parts)}')
        for i, (pt, st) in enumerate(zip(self.parts, saved)):
            if 'key' in st and st['key'] != self._part_key(pt):
                raise ValueError(f'ZeroAdam state entry {i} does not describe the same bucket (saved {st["key"][:3]}..., '
                                 f'built {self._part_key(pt)[:3]}...): same model, objectives and world size are required')
            if st['m'].numel() != pt.m.numel():
                raise ValueError(f'ZeroAdam state entry {i}: slice of {st["m"].numel()} elements, expected {pt.m.numel()}')
            pt.m.copy_(st['m'])
            pt.v.copy_(st['v'])
            for (p, _, _), n_ in zip(pt.entries, st.get('steps', [int(sd['step'])] * len(pt.entries))):
                self._pstep[id(p)] = int(n_)

    def _entries(self):
        """(p_ptr, g_ptr, m_ptr, v_ptr, numel, group, param) per (bucket slice n parameter) overlap that received a
        gradient in the current accumulation window."""
        ent = []
        for pt in self.parts:
            shard = getattr(pt.bucket, 'shard32', None)
            # `shard32` is a persistent buffer: it still holds the PREVIOUS window's gradient when nothing fed the bucket in
            # this one (every pass that uses it dropped from the loss, modality absent).  torch.optim / FusedAdam skip
            # `grad is None`; so does this: only buckets reduced in the current window (GradReducer.prepare / _launch*)
            if shard is None or not getattr(pt.bucket, 'has_grad', False):
                continue
            # parameters of a hook bucket that received no gradient in this step (an objective skipped; unused heads)
            # are left alone, as torch.optim / FusedAdam skip `grad is None` (no moment decay, no weight decay)
            had = getattr(pt.bucket, 'had', None)
            skip = set()
            if had is not None and hasattr(pt.bucket, 'offsets'):
                skip = {id(p) for p, h in zip(pt.bucket.params, had) if not h}
            for p, off, gi in pt.entries:
                if id(p) in skip:
                    continue
                lo, hi = max(off, pt.lo), min(off + p.numel(), pt.hi)
                if lo >= hi:
                    continue
                ent.append((pt.pflat.data_ptr() + 4 * lo, shard.data_ptr() + 4 * (lo - pt.lo),
                            pt.m.data_ptr() + 4 * (lo - pt.lo), pt.v.data_ptr() + 4 * (lo - pt.lo), hi - lo, gi, p))
        return ent

    def _table(self, ent):
        """Device tables of the multi-tensor kernels for the entries `ent` (cached by their addresses)."""
        dev = self.reducer.device
        sig = tuple(e[:5] for e in ent)
        if self._tab is None:
            self._tab = {}
        tab = self._tab.get(sig)
        if tab is not None:
            return tab
        if len(self._tab) > 8:          # the sets of parameters stepping together change rarely
            self._tab.clear()
        nt = len(ent)
        chunk_tensor, chunk_start = [], []
        for t, e in enumerate(ent):
            for o in range(0, e[4], CHUNK):
                chunk_tensor.append(t)
                chunk_start.append(o)
        nc = len(chunk_tensor)
        host_i = torch.empty(5 * nt + nc, dtype=torch.int64)
        for t, e in enumerate(ent):
            host_i[t], host_i[nt + t], host_i[2 * nt + t], host_i[3 * nt + t], host_i[4 * nt + t] = e[:5]
        if nc:
            host_i[5 * nt:] = torch.tensor(chunk_start, dtype=torch.int64)
        tab = {'sig': sig, 'ent': ent, 'nt': nt}
        tab['dev_i'] = host_i.to(dev)
        tab['dev_f'] = torch.empty(max(2 * nt, 1), dtype=torch.float32, device=dev)
        tab['dev_c'] = torch.tensor(chunk_tensor, dtype=torch.int32).to(dev)
        tab['partial'] = torch.empty(max(nc, 1), dtype=torch.float32, device=dev)
        tab['ctl'] = torch.zeros(4, dtype=torch.float32, device=dev)
        if dev.type == 'cuda':
            tl = hip.TensorList()
            base = tab['dev_i'].data_ptr()
            tl.p, tl.g, tl.m, tl.v = base, base + 8 * nt, base + 16 * nt, base + 24 * nt
            tl.numel, tl.chunk_start = base + 32 * nt, base + 40 * nt
            tl.lr, tl.wd = tab['dev_f'].data_ptr(), tab['dev_f'].data_ptr() + 4 * nt
            tl.chunk_tensor = tab['dev_c'].data_ptr()
            tl.n_chunks, tl.chunk = nc, CHUNK
            tab['tl'] = tl
        self._tab[sig] = tab
        return tab

    # ------------------------------------------------------------------------------------------ kernels
    def _local_sqnorm(self, tab):
        """sum of squares of this rank's gradient slices -> 0-dim device tensor."""
        hip.mt_grad_norm(tab['tl'], 1.0, 0.0, tab['partial'], tab['ctl'])
        return tab['ctl'][0] * tab['ctl'][0]

    def _apply(self, tab, args, ctl):
        hip.mt_adam(tab['tl'], args, ctl)

    # ------------------------------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, clip_grad=None):
        red = self.reducer
        if self.parts is None:
            self._build()
        elif any(sb not in [pt.bucket for pt in self.parts] for sb in red.sinks.values()):
            raise RuntimeError('new gradient buckets appeared after the partition was built: run one backward of '
                               'every objective before the first ZeroAdam.step()')
        ent = self._entries()
        tab = self._table(ent)          # every stepping entry: the gradient norm covers all of them
        self.step_count += 1
        ctl = None
        if clip_grad is not None:
            sq = self._local_sqnorm(tab).reshape(1).clone()
            if red.world > 1:
                red.all_reduce_now(sq)
            norm = sq.sqrt()
            bad = ~torch.isfinite(norm)
            coef = torch.clamp(clip_grad / (norm + 1e-6), max=1.0) if clip_grad > 0 else torch.ones_like(norm)
            ctl = tab['ctl']
            ctl[0:1] = norm
            ctl[1:2] = torch.where(bad, torch.zeros_like(coef), coef)
            ctl[2:3] = bad.to(torch.float32)
        # per-parameter step counters (torch.optim.AdamW semantics, as optim.FusedAdam): a parameter that had no gradient
        # in some step falls behind; the bias corrections are per launch, so entries are bucketed by their counter --
        # one launch in the usual case
        by_step = {}
        for e in ent:
            by_step.setdefault(self._pstep.get(id(e[6]), 0), []).append(e)
        b1, b2 = self.betas
        for step0, es in sorted(by_step.items()):
            tb = tab if len(by_step) == 1 else self._table(es)
            n_ = tb['nt']
            if n_:
                lrwd = torch.tensor([self.param_groups[e[5]]['lr'] for e in es] +
                                    [self.param_groups[e[5]]['weight_decay'] for e in es], dtype=torch.float32)
                tb['dev_f'][:2 * n_].copy_(lrwd, non_blocking=True)
            a = hip.AdamArgs()
            a.beta1, a.beta2, a.eps = b1, b2, self.eps
            if self.bias_correction:
                a.inv_bc1, a.inv_bc2 = 1.0 / (1.0 - b1 ** (step0 + 1)), 1.0 / (1.0 - b2 ** (step0 + 1))
            else:
                a.inv_bc1 = a.inv_bc2 = 1.0
            a.adam_w_mode = 1 if self.adam_w_mode else 0
            self._apply(tb, a, ctl)
        for pid in {id(e[6]) for e in ent}:
            self._pstep[pid] = self._pstep.get(pid, 0) + 1
        # every rank's updated slice -> every rank's full flat parameters (in place: the slice is the rank's
        # own chunk of the output buffer)
        if red.world > 1:
            # EVERY partitioned bucket is gathered, stepped in this window or not: whether a bucket received a gradient is
            # rank-local knowledge (bucket.has_grad), and a collective issued by some ranks only hangs the job; gathering
            # an unchanged slice is a no-op in value.  (Ranks must still agree on which PASSES run -- see
            # GradReducer.agree_finite -- because the reducer issues its collectives as the buckets become ready.)
            for pt in self.parts:
                if getattr(pt.bucket, 'shard32', None) is None:
                    continue
                red.all_gather(pt.pflat, pt.pflat[pt.lo:pt.hi])
        # the parameters changed behind autograd's back: invalidate the engine's bf16 weight shadows
        torch.autograd.graph.increment_version([p for pt in self.parts for p, _, _ in pt.entries])
        self.last_ctl = ctl
        return ctl[0] if ctl is not None else None

    def zero_grad(self, set_to_none=True):
        for g in self.param_groups:
            for p in g['params']:
                p.grad = None

    # rank-local state (like DeepSpeed's zero_pp_rank_* files): moments of this rank's slices + the step counter
    def state_dict(self):
        return {'step': self.step_count,
                'parts': [{'key': self._part_key(pt), 'm': pt.m.clone(), 'v': pt.v.clone(),
                           'steps': [self._pstep.get(id(p), 0) for p, _, _ in pt.entries]} for pt in (self.parts or [])]}

    def load_state_dict(self, sd):
        """Natural resume order works: build, load_state_dict, train.  The partition only exists once a backward has run
        (the engine's gradient buckets are created by it), so before that the state is kept and applied, validated
        against each bucket's identity, when the first step() builds the partition."""
        self.step_count = int(sd['step'])
        if not sd['parts']:
            return
        if self.parts is None:
            self._pending_state = sd
        else:
            self._apply_state(sd)
