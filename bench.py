#!/usr/bin/env python3
"""Headline benchmark: synthetic-image TRAINING throughput of the ViT encoder hot path.

  python bench.py --gpus N --steps K --warmup W         (any N: with no WORLD_SIZE in the environment and N > 1 the
                                                         script starts N ranks of itself, vit_amd/launch.py)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...      (runs as that rank)

One step = forward + label-smoothing CE + backward (+ bucketed RCCL gradient all-reduce when
N > 1, overlapped with backward) + AdamW, on one per-GPU batch of synthetic images
(train.py:530-554 of the reference).  Default workload: ViT-B/16-224 LAI + disentangled
(BASELINE.json's metric config: 24 layers x 1024 x 4096 x 16 heads, 197 tokens, per-GPU batch 128,
bf16 compute / fp32 master weights, LAI stage w = 20).  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     : the dominant kernel (bf16 MFMA GEMM): algorithmic FLOPs per launch / HIP-event
                 time per launch, summed over every GEMM launch of every 5th timed step (--timer-every:
                 events around all ~600 launches of every step cost 2.4 % of the step), vs 2.5 PF.  On the
                 FAVOR+ workloads: the FAVOR+ kernels, algorithmic bytes vs 8 TB/s (bound "hbm").
                 traffic = PMC bytes per launch from profiles/ (null when that file was measured on
                 other kernel sources).  In the step the weight-gradient GEMMs run BESIDE the others on a
                 second stream, so those in-step launch times include shared CUs (the schedule, not the
                 kernel); roofline.isolated is the same kernel on the step's own shapes and epilogues,
                 each timed ALONE (8 launches, HIP events) after the timed region: the kernel's quality.
  timed_kernels: the same line for every timed launch group (GEMM layouts, attention, FAVOR+).
  cpu_baseline : the CPU oracle (oracle/ref_cpu.py, the reference's algorithm in plain fp32
                 PyTorch) timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0         # HBM3E (same guide)

WORKLOADS = {
    # name: (attention variant, preset, img, patch, classes, default per-GPU batch)
    'dislai_b16': ('disentangled_lai_attention', 'B', 224, 16, 1000, 128),
    'dis_b16': ('disentangled_attention', 'B', 224, 16, 1000, 128),
    'lai_s16': ('transformer_timm_lai', 'S', 224, 16, 1000, 128),
    'dislai_l16': ('disentangled_lai_attention', 'L', 224, 16, 1000, 64),
    'regular_cifar': ('transformer', 'cifar', 32, 4, 10, 128),
    'performer_s16_384': ('performer_timm', 'S', 384, 16, 1000, 64),      # config 5, FAVOR+ m = 64
    'disperf_s16': ('disentangled_performer', 'S', 224, 16, 1000, 128),   # SURVEY 8(f)-1, m = 16 (train.py:75)
    # config 5's other leg: the same timm ViT-S/16-384 with its softmax attention left in place (577 tokens, no mask)
    'softmax_s16_384': ('transformer_timm_lai', 'S', 384, 16, 1000, 64),
}
MASK_EPOCHS = [10, 20, 30, 40]
STAGE_EPOCH = {8: 0, 12: 10, 18: 20, 20: 30, -1: 40}


def train_flops_per_image(variant, p, N, P, C, B, w):
    """Algorithmic FLOPs of one training step per image = 3 x forward; flash
    recompute and LN / GELU / softmax are not counted."""
    D, F, L, H = p['dim'], p['mlp'], p['depth'], p['heads']
    fwd = 2.0 * (N - 1) * 3 * P * P * D + 2.0 * D * C
    wbar = N if w < 0 else (N * (2 * w + 1) - w * (w + 1)) / N
    lin = 8.0 * N * D * D + 4.0 * N * D * F
    if 'performer' in variant:
        m = 16 if variant == 'disentangled_performer' else 64
        attn = 8.0 * N * D * m + 2.0 * N * m * H          # FAVOR+ with m random features
        table = 4.0 * N * D * D / B if variant == 'disentangled_performer' else 0.0    # pos_query / pos_key projections
    elif variant.startswith('disentangled'):
        attn = 8.0 * N * wbar * D
        table = 4.0 * (2 * N - 1) * D * D / B
    else:
        attn = 4.0 * N * wbar * D
        table = 0.0
    fwd += L * (lin + attn + table)
    return 3.0 * fwd


def launch_plan(gpus, env):
    """'spawn': start `gpus` ranks of this script (vit_amd/launch.py); 'run': this process is the job (N = 1) or one
    rank of it.  Decided from the environment alone, before anything touches the GPU."""
    have = all(k in env for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'))
    if not have:
        return 'spawn' if gpus > 1 else 'run'
    if int(env['WORLD_SIZE']) != gpus:
        raise SystemExit(f'--gpus {gpus} but WORLD_SIZE={env["WORLD_SIZE"]}: pass the same N to the launcher and to --gpus')
    return 'run'


def isolated_gemms(M, D, F, dev, reps=8):
    """The step's GEMM shapes with the step's epilogues, each timed ALONE on an idle GPU (2 warm-up + `reps` launches
    between one pair of HIP events on the launch stream).  Kernel quality, free of the step's schedule: in the step the
    weight-gradient GEMMs share the CUs with the forward / dX ones on purpose.  M = rows of the token matrix."""
    from vit_amd import ops
    bf = torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(7)

    def rnd(*shape, dtype=bf):
        return (torch.randn(*shape, device=dev, generator=g, dtype=torch.float32) * 0.05).to(dtype)

    xD, xF, x3 = rnd(M, D), rnd(M, F), rnd(M, 3 * D)
    w3, wDD, wFD, wDF, wq = rnd(3 * D, D), rnd(D, D), rnd(F, D), rnd(D, F), rnd(D, 3 * D)
    b3, bD, bF = rnd(3 * D, dtype=torch.float32), rnd(D, dtype=torch.float32), rnd(F, dtype=torch.float32)
    oD, oF, o3, pre = torch.empty_like(xD), torch.empty_like(xF), torch.empty_like(x3), torch.empty_like(xF)
    gDD, gFD, gDF, g3D = (torch.zeros(s, device=dev, dtype=torch.float32) for s in ((D, D), (F, D), (D, F), (3 * D, D)))
    A = ops
    cases = [   # (name, kernel group, flops, launch)
        ('qkv fwd (bias)', 'gemm_NT_bf16', 2.0 * M * 3 * D * D, lambda: A.gemm(xD, w3, bias=b3, out=o3)),
        ('proj fwd (bias + residual)', 'gemm_NT_bf16', 2.0 * M * D * D, lambda: A.gemm(xD, wDD, bias=bD, residual=xD, out=oD)),
        ('fc1 fwd (bias + GELU + pre-activation copy)', 'gemm_NT_bf16', 2.0 * M * F * D,
         lambda: A.gemm(xD, wFD, bias=bF, act=A.ACT_GELU, aux_out=pre, out=oF)),
        ('fc2 fwd (bias + residual)', 'gemm_NT_bf16', 2.0 * M * D * F, lambda: A.gemm(xF, wDF, bias=bD, residual=xD, out=oD)),
        ('fc2 dX (dGELU)', 'gemm_NT_bf16', 2.0 * M * F * D, lambda: A.gemm(xD, wFD, act=A.ACT_DGELU, aux_in=pre, out=oF)),
        ('fc1 dX (residual)', 'gemm_NT_bf16', 2.0 * M * D * F, lambda: A.gemm(xF, wDF, residual=xD, out=oD)),
        ('proj dX', 'gemm_NT_bf16', 2.0 * M * D * D, lambda: A.gemm(xD, wDD, out=oD)),
        ('qkv dX (residual)', 'gemm_NT_bf16', 2.0 * M * D * 3 * D, lambda: A.gemm(x3, wq, residual=xD, out=oD)),
        ('fc2 dW', 'gemm_TN_bf16', 2.0 * M * D * F, lambda: A.gemm(xD, xF, trans_a=True, trans_b=False, out=gDF, accumulate=True)),
        ('fc1 dW', 'gemm_TN_bf16', 2.0 * M * D * F, lambda: A.gemm(xF, xD, trans_a=True, trans_b=False, out=gFD, accumulate=True)),
        ('proj dW', 'gemm_TN_bf16', 2.0 * M * D * D, lambda: A.gemm(xD, xD, trans_a=True, trans_b=False, out=gDD, accumulate=True)),
        ('qkv dW', 'gemm_TN_bf16', 2.0 * M * D * 3 * D, lambda: A.gemm(x3, xD, trans_a=True, trans_b=False, out=g3D, accumulate=True)),
    ]
    rows, agg = [], {}
    for name, group, flops, fn in cases:
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        sec = e0.elapsed_time(e1) * 1e-3 / reps
        rows.append({'gemm': name, 'kernel': group, 'us': 1e6 * sec, 'tflops': flops / sec / 1e12})
        a = agg.setdefault(group, [0.0, 0.0])
        a[0] += flops
        a[1] += sec
    return rows, {k: v[0] / v[1] / 1e12 for k, v in agg.items()}


def cpu_baseline(args, variant, preset, img, patch, classes, w):
    """Oracle (port of the reference algorithm) on host cores, bounded sample."""
    from oracle import ref_cpu as R
    from vit_amd.train import PRESETS
    p = PRESETS[preset]
    host_cpus = os.cpu_count() or 1
    cores = min(16, host_cpus)      # the GPU box gives one job 16 CPUs (OMP_NUM_THREADS there); os.cpu_count() is the whole machine
    torch.set_num_threads(cores)
    me = MASK_EPOCHS if 'lai' in variant else None
    depth = min(p['depth'], args.cpu_layers)
    torch.manual_seed(0)
    if variant == 'transformer_timm_lai':
        net = R.RefVisionTransformerLAI(img_size=img, patch_size=patch, num_classes=classes, embed_dim=p['dim'],
                                        depth=depth, num_heads=p['heads'], mlp_ratio=p['mlp'] / p['dim'],
                                        mask_epochs=me)
    elif variant == 'performer_timm':
        # train.py:147-165: a timm ViT whose attention modules are swapped for PerformerAttention (its forward takes no epoch)
        class NoEpoch(torch.nn.Module):
            def __init__(self, mod):
                super().__init__()
                self.mod = mod

            def forward(self, x, epoch=None):
                return self.mod(x)
        net = R.RefVisionTransformerLAI(img_size=img, patch_size=patch, num_classes=classes, embed_dim=p['dim'],
                                        depth=depth, num_heads=p['heads'], mlp_ratio=p['mlp'] / p['dim'], mask_epochs=None)
        for blk in net.blocks:
            blk.attn = NoEpoch(R.RefPerformerAttention(p['dim'], num_heads=p['heads'], qkv_bias=True, n_orf=64))
    elif variant == 'disentangled_performer':
        tr = R.RefDisentangledPerformerTransformer(d_model=p['dim'], nhead=p['heads'], num_encoder_layers=depth,
                                                   dim_feedforward=p['mlp'], normalize_before=True, norf=16)
        net = R.RefVisionTransformer(img_size=img, patch_size=patch, num_classes=classes, transformer=tr)
    elif variant == 'transformer':
        tr = R.RefTransformer(d_model=p['dim'], dropout=0.0, nhead=p['heads'], dim_feedforward=p['mlp'],
                              num_encoder_layers=depth, normalize_before=True)
        net = R.RefVisionTransformer(img_size=img, patch_size=patch, num_classes=classes, transformer=tr)
    else:
        tr = R.RefDisentangledTransformer(d_model=p['dim'], dropout=0.0, nhead=p['heads'], dim_feedforward=p['mlp'],
                                          num_encoder_layers=depth, normalize_before=True, pos_att_type='c2p|p2c',
                                          max_position_embeddings=p['dim'], relative_attention=True,
                                          lai='lai' in variant, mask_epochs=me)
        net = R.RefVisionTransformer(img_size=img, patch_size=patch, num_classes=classes, transformer=tr)
    opt = torch.optim.AdamW([q for q in net.parameters()], lr=1e-4)
    Bc = args.cpu_batch
    x = torch.randn(Bc, 3, img, img)
    y = torch.randint(0, classes, (Bc,))
    epoch = STAGE_EPOCH[w] if me is not None else (0 if variant == 'performer_timm' else None)   # the LAI shell's forward wants one
    R.train_step(net, opt, x, y, epoch)           # warm-up
    t0 = time.perf_counter()
    n = 0
    while n < args.cpu_steps:
        R.train_step(net, opt, x, y, epoch)
        n += 1
    dt = (time.perf_counter() - t0) / n
    # the sample runs `depth` of the model's layers; scale the per-image time to the full depth
    # with the measured per-layer cost (layers are identical; embed/head are < 1 % of the step)
    scale = p['depth'] / depth
    return {'value': Bc / (dt * scale), 'unit': 'images/sec', 'cores': cores, 'host_cpus': host_cpus, 'kind': 'port',
            'sample': f'oracle/ref_cpu.py fp32, batch {Bc}, {depth} of {p["depth"]} layers timed over {n} step(s) '
                      f'(fwd+bwd+AdamW), per-image time scaled x{scale:g} to full depth'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='dislai_b16', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=0, help='per-GPU batch (0 = workload default)')
    ap.add_argument('--band', type=int, default=20, help='LAI half width stage: 8, 12, 18, 20 or -1 (unmasked)')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--bucket-mb', type=int, default=128)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=8)
    ap.add_argument('--cpu-steps', type=int, default=3)
    ap.add_argument('--cpu-layers', type=int, default=24)
    ap.add_argument('--no-kernel-timer', action='store_true')
    ap.add_argument('--timer-every', type=int, default=5,
                    help='HIP events around the kernels on every K-th step of the timed region (1 = every step)')
    ap.add_argument('--graph', action='store_true',
                    help='replay the step as one captured HIP graph, gradient exchange included (vit_amd.graph; the per-GEMM timer is off)')
    ap.add_argument('--no-max-overlap', action='store_true',
                    help='the library\'s own K splits (256 / tiles: 16 / 4 / 4) in the proj / fc1 / fc2 weight-gradient GEMMs and the '
                         'bias-gradient column sums on the main stream, instead of the default throughput schedule (two K splits = '
                         'VITK_DW_SPLITK=16:4,64:2,36:4, the dW GEMMs holding 64 / 128 CUs BESIDE the main stream, column sums on the side '
                         'stream: +3-4 %% images/s on the headline).  With this flag the in-step launch times describe each kernel '
                         'alone again (round 2\'s default); roofline.isolated measures that directly either way')
    ap.add_argument('--dropout', type=float, default=0.0,
                    help='dropout probability at every nn.Dropout site of the model (reference default: 0.1, train.py:79)')
    ap.add_argument('--comm-dtype', default='f32', choices=['f32', 'bf16'],
                    help='what the gradient buckets travel as (N > 1): f32 = the reference\'s DDP, bf16 = half the bytes')
    ap.add_argument('--reserve-cus', type=int, default=-1,
                    help='CUs the persistent GEMM launches leave free for RCCL\'s kernels (-1: 0 at N = 1, 16 at N > 1)')
    ap.add_argument('--no-isolated', action='store_true', help='skip the isolated per-shape GEMM pass after the timed region')
    ap.add_argument('--force-reducer', action='store_true',
                    help='diagnostic: run the gradient reducer (flat slabs, hooks, all-reduce) even at world size 1')
    args = ap.parse_args()

    # One process per GPU.  Started plain with --gpus N > 1: start N ranks of this very command line as CHILD processes
    # (before any GPU call; nothing is exec'ed), rank 0 prints the JSON line on our stdout.
    if launch_plan(args.gpus, os.environ) == 'spawn':
        from vit_amd.launch import spawn_ranks
        raise SystemExit(spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__), *sys.argv[1:]]))

    import torch.distributed as dist
    from vit_amd import ops
    from vit_amd.train import PRESETS, build_model, train_step, takes_epoch

    from vit_amd import functional as VF
    max_overlap = not args.no_max_overlap
    if not max_overlap:
        VF._DW_SPLITK = os.environ.get('VITK_DW_SPLITK', '')
        VF.SIDE_COLSUM = os.environ.get('VITK_SIDE_COLSUM', '0') != '0'
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    reserve = args.reserve_cus if args.reserve_cus >= 0 else (16 if world > 1 else 0)
    ops.set_reserved_cus(reserve)
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f'rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible): '
                         f'--gpus {args.gpus} needs {args.gpus} devices on this node (one process per GPU over RCCL)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1 or args.force_reducer:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        from vit_amd.ddp import init_process_group
        init_process_group('nccl', device=dev)      # RCCL's stream in the high-priority class: a hardware queue of its own

    variant, preset, img, patch, classes, dbatch = WORKLOADS[args.workload]
    if args.workload == 'softmax_s16_384':
        args.band = -1
    B = args.batch or dbatch
    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    p = PRESETS[preset]
    N = (img // patch) ** 2 + 1

    torch.manual_seed(0)       # identical init on every rank (rank 0 is broadcast anyway)
    net = build_model(variant, preset, img_size=img, patch_size=patch, num_classes=classes,
                      mask_epochs=MASK_EPOCHS if 'lai' in variant else None, dtype=dtype, device=dev, dropout=args.dropout)
    if args.dropout > 0:
        from vit_amd import dropout as VD
        VD.manual_seed(4321 + rank)
        net.train()
    from vit_amd.optim import AdamW
    use_graph = args.graph          # with a reducer the RCCL exchange is captured too (vit_amd.graph)
    opt = AdamW(net.parameters(), lr=1e-4, capturable=use_graph)
    reducer = None
    if world > 1 or args.force_reducer:
        from vit_amd.ddp import GradReducer
        reducer = GradReducer(net, bucket_mb=args.bucket_mb, always_reduce=args.force_reducer,
                              comm_dtype=torch.bfloat16 if args.comm_dtype == 'bf16' else torch.float32)
    epoch = STAGE_EPOCH[args.band] if takes_epoch(net) else None
    w = args.band if ('lai' in variant and epoch is not None) else -1

    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn(B, 3, img, img, device=dev, generator=g)
    y = torch.randint(0, classes, (B,), device=dev, generator=g)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if use_graph:
        from vit_amd.graph import GraphedTrainStep
        graphed = GraphedTrainStep(net, opt, x, y, epoch=epoch, warmup=max(2, args.warmup), reducer=reducer)
        step_fn = lambda: graphed.step(x, y)      # noqa: E731
        args.no_kernel_timer = True
    else:
        step_fn = lambda: train_step(net, opt, x, y, epoch, reducer=reducer)      # noqa: E731
    for _ in range(args.warmup):
        loss = step_fn()
    # The per-kernel HIP events are recorded on every `--timer-every`-th step of the timed region only: an event pair
    # around each of a step's ~600 launches keeps consecutive kernels from overlapping head to tail and costs 2.4 % of
    # the headline step (same box: 1885 img/s with events on every step, 1919 on every 4th, 1927 on every 10th, 1932
    # with none; the measured roofline fraction is the same at every rate: 0.3655-0.3662).
    timer = None if args.no_kernel_timer else []
    every = max(1, args.timer_every)
    timed_steps = 0
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        on = timer is not None and i % every == 0
        ops.KERNEL_TIMER = timer if on else None
        timed_steps += int(on)
        loss = step_fn()
    barrier()
    elapsed = time.perf_counter() - t0
    ops.KERNEL_TIMER = None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    loss_val = float(loss)

    ms_per_step = 1e3 * elapsed / args.steps
    value = world * B * args.steps / elapsed
    fl_img = train_flops_per_image(variant, p, N, patch, classes, B, w)
    model_tflops_gpu = fl_img * B * args.steps / elapsed / 1e12

    roofline = None
    kernels = None
    if timer:
        span_elapsed = elapsed * timed_steps / args.steps      # wall time of the steps that carried events (to first order)
        agg = {}
        spans = []                 # (start, end) of every GEMM launch on the first event's clock, for the union below
        base = timer[0][2]
        for name, work, e0, e1, unit in timer:
            a = agg.setdefault(name, [0.0, 0.0, 0, unit])
            a[0] += work
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
            if unit == 'flop':
                t0 = base.elapsed_time(e0) * 1e-3
                spans.append((t0, t0 + e0.elapsed_time(e1) * 1e-3))
        peak = PEAK_BF16_TFLOPS if dtype == torch.bfloat16 else PEAK_F32_TFLOPS

        def line(name):
            work, sec, cnt, unit = agg[name]
            if unit == 'flop':
                return {'bound': 'mfma', 'kernel': name, 'achieved': work / sec / 1e12, 'peak': peak, 'unit': 'TFLOP/s',
                        'frac': work / sec / 1e12 / peak, 'launches': cnt, 'avg_launch_us': 1e6 * sec / cnt,
                        'avg_gflop_per_launch': work / cnt / 1e9, 'share_of_step_time': sec / span_elapsed}
            return {'bound': 'hbm', 'kernel': name, 'achieved': work / sec / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                    'frac': work / sec / 1e9 / PEAK_HBM_GBS, 'launches': cnt, 'avg_launch_us': 1e6 * sec / cnt,
                    'avg_mb_per_launch': work / cnt / 1e6, 'share_of_step_time': sec / span_elapsed}

        gemms = [k for k in agg if agg[k][3] == 'flop']
        favor = [k for k in agg if k.startswith('favor')]
        # the roofline object: the GEMM that takes most of the step -- except on the FAVOR+ workloads, whose own kernels
        # (byte-bound streaming over q, k, v) are the path under test: there the slower of favor_fwd / favor_bwd
        dom = max(favor, key=lambda k: agg[k][1]) if favor else max(gemms, key=lambda k: agg[k][1])
        roofline = line(dom)
        # HBM-side bytes per launch of that kernel: measured with rocprofv3 --pmc in runs of their own (FETCH_SIZE and
        # WRITE_SIZE cannot share a pass, and PMC runs cannot be nested in this one); tools/collect_pmc.sh over this same
        # command writes the summary, which is used only while it was measured on the kernel sources of this tree.
        headline = args.workload == 'dislai_b16' and args.band == 20 and B == dbatch
        tname = 'pmc_gemm_traffic.json' if headline else f'pmc_traffic_{args.workload}.json'
        tpath = os.path.join(ROOT, 'profiles', tname)
        traffic, note = None, f'no profiles/{tname}'
        if os.path.exists(tpath) and (headline or (args.band == 20 and B == dbatch)):
            from vit_amd.srchash import kernel_source_hash
            tj = json.load(open(tpath))
            pmc_names = {'favor_fwd': ['favor_kmax_kernel', 'favor_kv_kernel', 'favor_out_kernel'],
                         'favor_bwd': ['favor_bwd_q_kernel', 'favor_bwd_kv_kernel']}.get(dom, [dom])
            if tj.get('kernel_source_sha256') != kernel_source_hash():
                note = (f'profiles/{tname} was measured on other kernel sources (git {tj.get("git_head", "?")[:10]}): '
                        'stale, re-run tools/collect_pmc.sh')
            elif all(k in tj.get('kernels', {}) for k in pmc_names):
                traffic = sum(tj['kernels'][k]['hbm_bytes_per_launch'] for k in pmc_names)
                note = f'profiles/{tname}, rocprofv3 --pmc passes at git {tj.get("git_head", "?")[:10]} (+ working tree), same kernel sources'
            else:
                note = f'profiles/{tname} has no entry for {pmc_names}'
        roofline['timer'] = f'HIP events on {timed_steps} of the {args.steps} timed steps (every {every}th), on the launch streams'
        roofline['traffic'] = traffic
        roofline['traffic_unit'] = 'bytes/launch (PMC)'
        roofline['traffic_source'] = note
        if gemms:
            # the weight-gradient GEMMs run on a second stream and overlap the others (functional._dw): summed per-launch
            # durations would count the shared time twice, so the aggregate is taken over the UNION of the launches' spans
            spans.sort()
            busy, cur0, cur1 = 0.0, None, None
            for t0, t1 in spans:
                if cur1 is None or t0 > cur1:
                    if cur1 is not None:
                        busy += cur1 - cur0
                    cur0, cur1 = t0, t1
                else:
                    cur1 = max(cur1, t1)
            if cur1 is not None:
                busy += cur1 - cur0
            roofline['all_gemm_tflops'] = sum(agg[k][0] for k in gemms) / busy / 1e12
            roofline['all_gemm_share_of_step_time'] = busy / span_elapsed
            roofline['overlap_note'] = ('dW GEMMs (gemm_TN) run on a second stream beside the dX GEMMs: per-launch durations include '
                                        'the time they share CUs; all_gemm_* are over the union of the launch spans')
        kernels = [line(k) for k in sorted(agg, key=lambda k: -agg[k][1])]
        if max_overlap and gemms:
            roofline['schedule_note'] = ('default schedule: the dW GEMMs hold 32 / 128 CUs beside the main stream, so in-step launch '
                                         'times of the NT GEMMs include shared CUs; roofline.isolated = the same kernel alone')

    # Kernel quality apart from the schedule: every GEMM shape of one encoder layer, with its epilogue, timed alone.
    if roofline is not None and rank == 0 and dtype == torch.bfloat16 and not args.no_isolated and roofline['bound'] == 'mfma':
        rows_tok = B * N
        if variant in ('disentangled_attention', 'disentangled_lai_attention'):
            rows_tok += VF._tail_rows(rows_tok, 2 * N - 1)      # the relative-table rows ride along (functional.DisLayerFn)
        torch.cuda.synchronize()
        per_shape, by_kernel = isolated_gemms(rows_tok, p['dim'], p['mlp'], dev)
        iso = by_kernel.get(roofline['kernel'])
        roofline['isolated'] = {
            'what': f'each GEMM of one encoder layer ([{rows_tok}, {p["dim"]}] tokens, MLP {p["mlp"]}) with its fused epilogue, alone on the '
                    'GPU: 2 warm-up + 8 launches between one pair of HIP events, after the timed region',
            'kernel': roofline['kernel'], 'achieved': iso, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
            'frac': None if iso is None else iso / PEAK_BF16_TFLOPS,
            'by_kernel_tflops': by_kernel, 'per_shape': per_shape}

    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline and variant != 'performer':      # model.Performer has no oracle shell
            cpu = cpu_baseline(args, variant, preset, img, patch, classes, w)
        out = {
            'metric': 'train images/sec (node) ViT-B/16-224 LAI+Dis' if args.workload == 'dislai_b16'
                      else f'train images/sec (node) {args.workload}',
            'value': value, 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'{args.workload}: {variant} preset {preset} ({p["depth"]}L x {p["dim"]} x '
                                   f'{p["mlp"]} x {p["heads"]}h), {img}x{img}/{patch}, {N} tokens, LAI w={w}, '
                                   f'dropout {args.dropout:g}, AdamW lr 1e-4, label smoothing 0.1',
                       'per_gpu_batch': B, 'global_batch': B * world, 'parallelism': f'dp{world}',
                       'hip_graph': bool(use_graph), 'max_overlap': bool(max_overlap), 'reserved_cus': reserve,
                       'grad_bucket_mb': args.bucket_mb if reducer is not None else None,
                       'grad_comm_dtype': args.comm_dtype if reducer is not None else None},
            'train_gflop_per_image': fl_img / 1e9, 'model_tflops_per_gpu': model_tflops_gpu,
            'mfma_roofline_frac_per_gpu': model_tflops_gpu / (PEAK_BF16_TFLOPS if dtype == torch.bfloat16
                                                                else PEAK_F32_TFLOPS),
            'loss': loss_val, 'roofline': roofline, 'cpu_baseline': cpu, 'timed_kernels': kernels,
        }
        print(json.dumps(out), flush=True)
    if world > 1 or args.force_reducer:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
