#!/usr/bin/env python
"""Headline benchmark: image-text pairs/s, forward+backward, VLMo-Base, bf16, on
1/2/4/8 MI355X (BASELINE.json configs[1] / configs[2]).

A "step" = one VLMO.forward_features(img, txt, masks) in VL mode (training-mode
semantics: dropout / attention dropout / drop-path 0.1 as conf/model/vlmo_base.yaml:23-25)
+ backward from a scalar loss, on a synthetic batch already resident in HBM.
For N > 1 every rank runs the same per-GPU batch (weak scaling) and gradients are
averaged over RCCL (exploremultimodal_amd/dp.py).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W        (starts its own N ranks when no launcher did)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (see README / DESIGN.md section "Measurement").
"""
import argparse
import json
import os
import sys
import time
from functools import partial

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')   # see exploremultimodal_amd/__init__.py

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def spawn_ranks(n, cmd, port=None, env=None, timeout=None):
    """`python bench.py --gpus N` from a plain shell (no launcher, WORLD_SIZE unset): start one child process per rank
    with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (what utils/utils.py:298-334
    `init_distributed_mode` reads from the launcher's environment), relay rank 0's stdout (the ONE JSON line) and
    return the first non-zero exit code (the other ranks are then terminated).  The parent never touches the GPU:
    this runs before torch is imported, and a process that has initialised HIP is never re-exec'ed."""
    import socket
    import subprocess
    if port is None:
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
    procs = []
    for r in range(n):
        e = dict(os.environ if env is None else env)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                 MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # rank 0 owns stdout; the other ranks print nothing there by contract, anything they do print goes to stderr
        procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else sys.stderr))
    rc, t_end = 0, (time.time() + timeout) if timeout else None
    out0 = b''
    try:
        import threading
        buf = []
        rd = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
        rd.start()
        alive = set(range(n))
        while alive:
            for r in sorted(alive):
                c = procs[r].poll()
                if c is not None:
                    alive.discard(r)
                    if c != 0 and rc == 0:
                        rc = c
            if rc != 0 or (t_end and time.time() > t_end):
                if rc == 0:
                    rc = 124
                break
            time.sleep(0.05)
        for r in alive:             # a rank failed (or the time limit passed): end exactly the children started here
            procs[r].terminate()
        for r in alive:
            try:
                procs[r].wait(10)
            except subprocess.TimeoutExpired:
                procs[r].kill()
        rd.join(10)
        out0 = buf[0] if buf else b''
    finally:
        for p_ in procs:
            if p_.poll() is None:
                p_.kill()
    if rc == 0:
        sys.stdout.write(out0.decode())
        sys.stdout.flush()
    else:
        sys.stderr.write(out0.decode())
    return rc


def _wanted_gpus(argv):
    for i, a in enumerate(argv):
        if a == '--gpus' and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith('--gpus='):
            return int(a.split('=', 1)[1])
    return 1


if __name__ == '__main__' and 'WORLD_SIZE' not in os.environ and _wanted_gpus(sys.argv[1:]) > 1:
    sys.exit(spawn_ranks(_wanted_gpus(sys.argv[1:]), [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))

import torch

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0


def fwd_flops_per_pair(d, L, F, T, P=197, patch_k=768):
    """matmul flops only, 2 per MAC."""
    blk = lambda n: 24 * n * d * d + 4 * n * n * d
    return 2 * (P - 1) * patch_k * d + F * (blk(P) + blk(T)) + (L - F) * blk(P + T)


def build_model(preset, device, drop=0.1):
    from exploremultimodal_amd.vlmo import VLMO, LayerNorm
    from exploremultimodal_amd import synth
    mc = synth.make_config(preset).model
    torch.manual_seed(0)
    m = VLMO(img_size=mc.img_size, patch_size=mc.patch_size, in_chans=mc.in_chans, num_classes=0,
             embed_dim=mc.embed_dim, depth=mc.depth, num_heads=mc.num_heads, mlp_ratio=mc.mlp_ratio,
             qkv_bias=True, drop_rate=drop, attn_drop_rate=drop, drop_path_rate=drop,
             norm_layer=partial(LayerNorm, eps=1e-12), init_values=mc.init_values, vocab_size=mc.vocab_size,
             max_text_len=mc.max_text_len, fusion_layer=mc.fusion_layer)
    for b in m.blocks[:mc.fusion_layer]:       # VlmoModule._freeze_params, vlmo_module.py:165-167
        del b.mlp['vl']
    return m.to(device), mc


def cpu_baseline(preset, B=8, reps=5):
    """The CPU oracle (plain PyTorch fp32 restatement of the reference) timed on the host cores:
    VL forward+backward at batch B (median of `reps` after one warm-up) -> pairs/s, plus BASELINE.json configs[0]
    (single VL forward, batch 2).  Reported baseline, not a target."""
    from oracle import synth, vlmo_oracle
    # the GPU box exposes many more logical CPUs than this job's share: use the affinity mask, capped at
    # the documented 16-core share of a 1-GPU box (oversubscribing the host stalls for minutes)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    mc = synth.make_config(preset).model
    sd = synth.synth_backbone_state_dict(mc, 0)
    for t in sd.values():
        t.requires_grad_(True)

    def run(Bx, backward, n):
        batch = synth.synth_batch(mc, Bx, seed=1234, mim=False)
        im = torch.ones(Bx, synth.num_img_tokens(mc), dtype=torch.int64)
        ts = []
        for r in range(n + 1):
            t0 = time.perf_counter()
            with torch.set_grad_enabled(backward):
                x, _ = vlmo_oracle.forward_features(sd, mc, img=batch['image'], txt=batch['text_ids'], img_attn_masks=im,
                                                    txt_attn_masks=batch['text_mask'])
                if backward:
                    x.square().mean().backward()
            ts.append(time.perf_counter() - t0)
            for t in sd.values():
                t.grad = None
        ts = sorted(ts[1:])
        return ts[len(ts) // 2]

    med = run(B, True, reps)
    fwd2 = run(2, False, reps)
    return {'value': round(B / med, 3), 'unit': 'pairs/s', 'cores': cores, 'kind': 'port',
            'sample': f'oracle VLMo-{preset} VL fwd+bwd fp32, batch {B}, median of {reps} after 1 warm-up '
                      f'({med:.2f} s per step); configs[0] single VL forward at batch 2: {fwd2:.3f} s '
                      f'({2 / fwd2:.2f} pairs/s forward only)',
            'config0_forward_b2_s': round(fwd2, 4)}


def _latest_profile(suffix):
    """Newest profiles/rNN*<suffix> by round number (the PMC passes of this same command are committed per round)."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]*' + suffix)):
        m = re.match(r'r(\d\d)(.*)', os.path.basename(f))
        key = (int(m.group(1)), m.group(2))
        if best is None or key > best[0]:
            best = (key, f)
    return best[1] if best else None


def main():
    # stdout carries exactly ONE JSON line: libraries that print to file descriptor 1 while they initialise
    # (RCCL's version banner at communicator creation) are sent to stderr until the result is ready
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=64, help='per-GPU batch (pairs)')
    ap.add_argument('--preset', default='base')
    ap.add_argument('--tile', type=int, default=None)
    ap.add_argument('--comm-dtype', choices=['auto', 'bf16', 'fp32'], default='auto',
                    help='gradient exchange dtype: bf16 packs / unpacks every bucket, fp32 exchanges it in place, auto = timed at start-up')
    ap.add_argument('--merge-passes', action='store_true',
                    help='full objective only: batch the backbone passes by mode (V / L / VL): 3 passes instead of 7')
    ap.add_argument('--optimizer', action='store_true',
                    help='also run the fused AdamW step with global-norm clip 5.0 (conf/train/pretrain_mum.yaml:28-36,54,75-80) '
                         'inside the timed step; the headline metric is forward+backward only, so this is off by default')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--logits', action='store_true', help='full objective: materialise mlm/mim logits as the reference does')
    ap.add_argument('--no-dropout', action='store_true')
    ap.add_argument('--force-reducer', action='store_true', help='run the RCCL gradient reducer even at world size 1')
    ap.add_argument('--objective', default='vl', choices=['vl', 'full'],
                    help="vl = one VL forward_features + backward (headline metric); full = VlmoModule.forward with "
                         "[mlm, mim, itc, itm] incl. the in-loop dVAE tokenizer (BASELINE.json configs[4])")
    ap.add_argument('--rehearse-gloo', action='store_true',
                    help='multi-rank REHEARSAL on fewer GPUs than ranks: gloo collectives, ranks share the visible GPUs '
                         '(exercises the world>1 code paths; the numbers mean nothing)')
    ap.add_argument('--zero2', action='store_true', help='reduce-scatter gradients (ZeRO-2 style partition, ds_stage/l2.yaml)')
    args = ap.parse_args()

    t_start = time.perf_counter()
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus != world:       # a launcher set WORLD_SIZE: the environment is authoritative (utils/utils.py:298-334)
        if rank == 0:
            print(f'[bench] --gpus {args.gpus} but WORLD_SIZE={world}: running {world} ranks', file=sys.stderr)
        args.gpus = world
    if args.rehearse_gloo:
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1 or args.force_reducer:
        import torch.distributed as dist
        if 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1')
        if args.rehearse_gloo:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)

    from exploremultimodal_amd import engine, hip
    from exploremultimodal_amd import synth
    hip.lib()
    drop = 0.0 if args.no_dropout else 0.1
    if args.objective == 'full':
        from exploremultimodal_amd.build import build_model as build_module
        cfg = synth.make_config(args.preset, loss_names=['mlm', 'mim', 'itc', 'itm'], drop_rate=drop,
                                attn_drop_rate=drop, drop_path_rate=drop)
        cfg.train.merge_passes = args.merge_passes
        cfg.train.fused_ce = not args.logits       # vocabulary heads: fused HIP cross-entropy, no logits in HBM
        torch.manual_seed(0)
        model, mc = build_module(cfg).to(dev), cfg.model
    else:
        model, mc = build_model(args.preset, dev, drop=drop)
    model.train()
    if args.tile is not None:
        engine.DEFAULT_TILE = args.tile
    reducer = None
    if dist is not None:
        from exploremultimodal_amd.dp import GradReducer
        reducer = GradReducer(model, dist.group.WORLD, reduce_scatter=args.zero2,
                              comm_dtype=(torch.float32 if (args.comm_dtype == 'fp32' or args.rehearse_gloo) else
                                          (torch.bfloat16 if args.comm_dtype == 'bf16' else None)))

    B = args.batch
    batch = synth.synth_batch(mc, B, seed=1234 + rank, mim=args.objective == 'full')
    if args.objective == 'full':        # what the input hand-off (prefetch.DataLoaderX) attaches to every batch on the host
        from exploremultimodal_amd.objectives import attach_row_indices
        attach_row_indices(batch)
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    P = synth.num_img_tokens(mc)
    img = batch['image'].to(dev)
    ids, tmask = batch['text_ids'].to(dev), batch['text_mask'].to(dev)
    imask = torch.ones(B, P, dtype=torch.int64, device=dev)
    R = torch.randn(B, mc.max_text_len + P, mc.embed_dim, device=dev) / (B * 1000.0)

    opt = None
    if args.optimizer:
        from types import SimpleNamespace as NS
        from exploremultimodal_amd import optim
        if not hasattr(model, 'config'):      # bare VLMO backbone: give the factory the attributes it reads by name
            model.config = NS(model=mc)
        ocfg = NS(opt=NS(name='fusedadamw', eps=1e-8, betas=[0.9, 0.98], momentum=0.9), weight_decay=0.01,
                  base_lr=2e-4, lr_mult_head=1, lr_mult_fusion=1)
        if args.zero2 and reducer is not None:
            # ZeRO-2 (conf/ds_stage/l2.yaml): sharded AdamW over the reduce-scattered gradient slices + all-gather
            from exploremultimodal_amd.zero import ZeroAdam
            skip = model.no_weight_decay() if hasattr(model, 'no_weight_decay') else {}
            opt = ZeroAdam(reducer, optim.get_parameter_groups(model, base_lr=ocfg.base_lr, lr_mult_head=1, lr_mult_fusion=1,
                                                                weight_decay=ocfg.weight_decay, skip_list=skip),
                           betas=(0.9, 0.98), eps=1e-8)
        else:
            opt = optim.create_optimizer(ocfg, model)

    def step():
        for p in model.parameters():
            p.grad = None
        if args.objective == 'full':
            ret = model(dict(dbatch))
            loss = sum(v for k, v in ret.items() if 'task_loss' in k)
        else:
            x, _ = model.forward_features(img=img, txt=ids, img_attn_masks=imask, txt_attn_masks=tmask)
            loss = (x * R).sum()
        if reducer is not None:
            reducer.prepare(loss)
        loss.backward()
        if reducer is not None:
            reducer.finish()
        if opt is not None:
            opt.step(clip_grad=5.0)
        return loss

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print(f'[bench +{time.perf_counter() - t_start:.1f}s] {msg}', file=sys.stderr, flush=True)

    log('model and batch ready')
    for i in range(args.warmup):
        step()
        if i == 0:
            torch.cuda.synchronize()
            log('first step done')
    barrier()
    log('warm-up done')
    hip.profile_start()                   # HIP-event pairs around every GEMM launch of the timed region (in-library)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    t_issue = time.perf_counter() - t0    # host time to ENQUEUE the steps (launch-bound if close to dt)
    barrier()
    dt = time.perf_counter() - t0
    prof = hip.profile_stop()
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    assert torch.isfinite(loss).item(), 'non-finite loss'
    # host cost of enqueueing one step, measured with an EMPTY device queue (in the timed loop above the host runs
    # ahead until the launch queue is full and then waits for the GPU: t_issue there is back-pressure, not host work)
    hs = []
    for _ in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        hs.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    host_ms = sorted(hs)[1] * 1e3
    comm_ms_per_step = None
    if reducer is not None:
        # three more steps (all ranks: the exchange is collective) with event pairs on the communication stream
        reducer.timing = True
        for _ in range(3):
            step()
        comm_ms_per_step = round(reducer.comm_ms() / 3, 4)
        reducer.timing = False
    if os.environ.get('VLMO_SYNC_PROBE'):
        # measurement aid: the host <-> device synchronisation points of one step, innermost frame inside this repository
        import collections
        import traceback
        import warnings
        sites, root = collections.Counter(), os.path.dirname(os.path.abspath(__file__))

        def _show(message, category, filename, lineno, file=None, line=None):
            if 'synchronizing' in str(message) and 'prototype feature' not in str(message):
                where = [f'{os.path.relpath(fr.filename, root)}:{fr.lineno} {fr.line}' for fr in traceback.extract_stack()[:-1]
                         if fr.filename.startswith(root)]
                sites[where[-1] if where else f'{filename}:{lineno}'] += 1
        old_show, warnings.showwarning = warnings.showwarning, _show
        warnings.simplefilter('always')
        torch.cuda.set_sync_debug_mode('warn')
        step()
        torch.cuda.set_sync_debug_mode('default')
        warnings.showwarning = old_show
        torch.cuda.synchronize()
        for k, v in sites.most_common():
            log(f'sync probe: {v:3d} x {k}')
        log(f'sync probe: {sum(sites.values())} synchronisation(s) in one step')
    log(f'timed region done: {dt / args.steps * 1e3:.2f} ms/step (host enqueue with an empty queue {host_ms:.2f} ms/step; '
        f'{t_issue / args.steps * 1e3:.2f} ms/step while the queue is full)')
    if rank == 0:
        ms = dt / args.steps * 1e3
        pairs = B * world * args.steps / dt
        fl = 3 * fwd_flops_per_pair(mc.embed_dim, mc.depth, mc.fusion_layer, mc.max_text_len, P,
                                    mc.in_chans * mc.patch_size ** 2)
        if args.objective == 'full':   # 4 VL + 2 V + 1 L backbone passes, 3 of the VL-sized ones on 2B... per pair:
            d_, L_, F_, T_ = mc.embed_dim, mc.depth, mc.fusion_layer, mc.max_text_len
            blk = lambda n: 24 * n * d_ * d_ + 4 * n * n * d_
            v_only = 2 * (P - 1) * mc.in_chans * mc.patch_size ** 2 * d_ + L_ * blk(P)
            l_only = L_ * blk(T_)
            fl = 3 * (4 * fl // 3 + 2 * v_only + l_only) + int(52.119e9)      # + dVAE forward per image
        # per-symbol totals from the HIP-event pairs recorded in the timed region (hip.profile_start / _stop)
        per = {sym: [sec, flops, n] for sym, (sec, flops, n) in prof.items()}
        import re
        tf = _latest_profile('traffic.json')
        try:
            pmc = json.load(open(tf)) if tf else None
        except (OSError, ValueError):
            pmc = None

        def roof_of(item):
            sym, (tsec, flops, n) = item
            ach = flops / tsec / 1e12
            r = {'bound': 'mfma', 'kernel': sym, 'achieved': round(ach, 1), 'peak': PEAK_BF16_TFLOPS,
                 'unit': 'TFLOP/s', 'frac': round(ach / PEAK_BF16_TFLOPS, 4), 'traffic': None,
                 'launches_per_step': n // args.steps, 'avg_launch_us': round(tsec / n * 1e6, 2),
                 'flops_per_launch': round(flops / n), 's_per_step': round(tsec / args.steps, 6)}
            # HBM bytes per launch: from the committed PMC passes of this same command (another run: flagged as such)
            try:
                epi = {'bias': 0, 'bias_gelu': 1, 'resid': 2, 'dgelu': 3, 'f32': 4, 'dual': 5, 'argmax': 6}
                m16 = re.match(r'gemm_nt16_kernel<(\w+),Hx256>', sym)
                if m16:
                    # the 16x16x32 kernels of one epilogue run at several tile heights (one symbol per height in the
                    # profile): launch-weighted mean of their bytes per launch
                    pat16 = f'gemm_nt16_kernelIDF16bLi\\d+ELi{epi[m16.group(1)]}ELi\\dE'
                    hits = [v for k, v in pmc['kernels'].items() if re.search(pat16, k)]
                    tot = sum(v['launches'] for v in hits)
                    r['traffic'] = round(sum((v['fetch_bytes_per_launch'] + v['write_bytes_per_launch']) * v['launches'] for v in hits) / tot)
                    r['traffic_measured_in_run'] = False
                    r['traffic_source'] = os.path.relpath(tf, ROOT)
                    return r
                m = re.match(r'(gemm_[a-z_]+_kernel)<(?:(\w+),)?(\d+)x(\d+)>', sym)
                if 'multi' in m.group(1):
                    pat = m.group(1)
                else:
                    pat = f'{m.group(1)}IDF16bLi{m.group(3)}ELi{m.group(4)}ELi\\dELi\\dE'
                    if m.group(2):
                        pat += f'Li{epi[m.group(2)]}E'
                for k, v in pmc['kernels'].items():
                    if re.search(pat, k):
                        r['traffic'] = round(v['fetch_bytes_per_launch'] + v['write_bytes_per_launch'])
                        r['traffic_measured_in_run'] = False
                        r['traffic_source'] = os.path.relpath(tf, ROOT)
                        break
            except (KeyError, ValueError, AttributeError, TypeError):
                pass
            return r

        # `roofline` = the kernel symbol with the largest ELAPSED share of the step (event time on its launch stream; in
        # the overlapped step this includes time a launch waits for compute units the other stream holds, so it is the
        # lower of the two rates one could quote); `roofline_by_flops` = the symbol that carries the most flops
        roof = roof_of(max(per.items(), key=lambda kv: kv[1][0])) if per else None
        roof_fl = roof_of(max(per.items(), key=lambda kv: kv[1][1])) if per else None
        out = {
            'metric': 'image-text pairs/sec fwd+bwd, VLMo-Base, 1/2/4/8 MI355X; % bf16 MFMA roofline',
            'value': round(pairs, 2), 'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': (f'VLMo-{args.preset} VL forward_features fwd+bwd' if args.objective == 'vl' else
                                    f'VLMo-{args.preset} VlmoModule.forward [mlm,mim,itc,itm] + in-loop dVAE, fwd+bwd'
                                    + (', ZeRO-2 grad partition' if args.zero2 else '')
                                    + (', passes merged by mode' if args.merge_passes else '')) +
                                   f', per-GPU batch {B} synthetic 224x224 image + {mc.max_text_len}-token text '
                                   f'pairs, dropout/drop-path {0.0 if args.no_dropout else 0.1}'
                                   + (', + fused AdamW step with clip 5.0' if args.optimizer else ''),
                       'global_batch': B * world, 'seq_len': mc.max_text_len + P,
                       'parallelism': f'dp{world}'},
            'step_tflops': round(fl * B * world * args.steps / dt / 1e12, 1),
            'step_mfma_frac': round(fl * B * args.steps / dt / 1e12 / PEAK_BF16_TFLOPS, 4),
            'flops_per_pair': fl,
            'roofline': roof,
            'roofline_by_flops': roof_fl,
            'kernels': {k: {'s_per_step': round(v[0] / args.steps, 6), 'tflops': round(v[1] / v[0] / 1e12, 1)}
                        for k, v in per.items()},
        }
        # north_star: "rocprof HBM GB/s and MFMA-busy counters reported against chip peak": whole-step figures of the
        # committed PMC passes of this same command (profiles/rNN_traffic.json, tools/profile_round.sh +
        # tools/summarize_pmc.py).  They were NOT measured by this run: they sit in their own block with the profile's
        # step time, and are dropped when that differs from this run's by more than 10 % (another build or workload).
        if pmc is not None and 'step' in pmc and args.objective == 'vl' and args.preset == 'base' and B == 64:
            pms = pmc.get('ms_per_step_unprofiled')
            if pms and abs(pms - ms) / ms <= 0.10:
                out['from_profile'] = {'measured_in_run': False, 'source': os.path.relpath(tf, ROOT),
                                       'profile_ms_per_step': pms, 'hbm_gbs': pmc['step'].get('hbm_gbs'),
                                       'hbm_frac': round(pmc['step'].get('hbm_gbs', 0) / PEAK_HBM_GBS, 4),
                                       'mfma_busy': pmc['step'].get('mfma_busy')}
        out['host_enqueue_ms'] = round(host_ms, 3)
        # which stream the weight gradients of the VL pass ran on (engine._use_side_stream: forced by VLMO_OVERLAP_WGRAD,
        # else the side stream under a reducer or for passes of fewer than engine.ONE_STREAM_ROWS rows)
        out['wgrad_stream'] = 'side' if engine._use_side_stream(engine.GRAD_SINK, B * (mc.max_text_len + P)) else 'main'
        if reducer is not None:
            # what a scaling run needs to check the exchange: communicator size as the communicator reports it, the
            # form chosen at start-up (GradReducer.autotune), bytes per step, time of the communication stream per step
            out['comm'] = dict(reducer.describe(), comm_stream_ms_per_step=comm_ms_per_step)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.preset)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
    if reducer is not None:
        reducer.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
