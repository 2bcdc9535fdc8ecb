"""Where do the torch copy / cast kernels of a training step come from?  Runs a few steps of a bench workload under
torch.profiler (shapes + Python stacks) and prints the aten::copy_ / aten::to / aten::add call sites by device time.
usage (GPU box): python3 tools/find_copies.py [--workload dislai_b16] [--batch 32]"""
import argparse
import os
import sys
from collections import defaultdict

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                        # noqa: E402
from vit_amd.optim import AdamW                                     # noqa: E402
from vit_amd.train import build_model, takes_epoch, train_step     # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--workload', default='dislai_b16')
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--band', type=int, default=20)
args = ap.parse_args()
variant, preset, img, patch, classes, _ = bench.WORKLOADS[args.workload]
dev = torch.device('cuda', 0)
torch.manual_seed(0)
net = build_model(variant, preset, img_size=img, patch_size=patch, num_classes=classes,
                  mask_epochs=bench.MASK_EPOCHS if 'lai' in variant else None, dtype=torch.bfloat16, device=dev)
opt = AdamW(net.parameters(), lr=1e-4)
epoch = bench.STAGE_EPOCH[args.band] if takes_epoch(net) else None
x = torch.randn(args.batch, 3, img, img, device=dev)
y = torch.randint(0, classes, (args.batch,), device=dev)
for _ in range(3):
    train_step(net, opt, x, y, epoch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(2):
        train_step(net, opt, x, y, epoch)
    torch.cuda.synchronize()
agg = defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.name not in ('aten::copy_', 'aten::add', 'aten::add_', 'aten::cat', 'aten::fill_', 'aten::zero_'):
        continue
    dt = getattr(ev, 'device_time_total', 0) or getattr(ev, 'cuda_time_total', 0)
    site = next((s for s in ev.stack if '/vit_amd/' in s or 'bench.py' in s), ev.stack[0] if ev.stack else '?')
    key = (ev.name, site.strip()[-90:], str(ev.input_shapes)[:70])
    agg[key][0] += 1
    agg[key][1] += dt
for (name, site, shapes), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f'{t / 2e3:8.2f} ms/step {n / 2:6.1f}/step  {name:12s} {shapes:70s} {site}')
