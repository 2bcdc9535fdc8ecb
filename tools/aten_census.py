"""Which torch (aten) ops launch kernels inside one training step, how often, and from which line of
vit_amd/ -- the list the "un-fused leftovers" are swept from.  usage: python tools/aten_census.py [workload]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from vit_amd.train import build_model, train_step, takes_epoch
from vit_amd.optim import AdamW

wl = sys.argv[1] if len(sys.argv) > 1 else 'dislai_b16'
variant, preset, img, patch, classes, B = bench.WORKLOADS[wl]
dev = torch.device('cuda', 0)
torch.manual_seed(0)
net = build_model(variant, preset, img_size=img, patch_size=patch, num_classes=classes,
                  mask_epochs=bench.MASK_EPOCHS if 'lai' in variant else None, dtype=torch.bfloat16, device=dev)
opt = AdamW(net.parameters(), lr=1e-4)
reducer = None
if '--reducer' in sys.argv:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
    os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
    dist.init_process_group('nccl', device_id=dev)
    from vit_amd.ddp import GradReducer
    reducer = GradReducer(net)
epoch = bench.STAGE_EPOCH[20] if takes_epoch(net) else None
x = torch.randn(B, 3, img, img, device=dev)
y = torch.randint(0, classes, (B,), device=dev)
for _ in range(3):
    train_step(net, opt, x, y, epoch, reducer=reducer)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    train_step(net, opt, x, y, epoch, reducer=reducer)
    torch.cuda.synchronize()
cnt = collections.Counter()
dur = collections.Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith('aten::'):
        continue
    kt = sum(k.duration for k in ev.kernels) if ev.kernels else 0
    if not ev.kernels:
        continue
    where = '?'
    for fr in ev.stack or []:
        if 'vit_amd/' in fr or 'bench.py' in fr:
            where = fr.split('vit_amd/')[-1] if 'vit_amd/' in fr else fr
            break
    cnt[(ev.name, where)] += 1
    dur[(ev.name, where)] += kt
tot = 0
for k, n in sorted(cnt.items(), key=lambda kv: -dur[kv[0]]):
    print(f'{n:5d} x {k[0]:28s} {dur[k]:9.0f} us  {k[1]}')
    tot += dur[k]
print(f'total {sum(cnt.values())} launches, {tot / 1e3:.2f} ms of kernels per step')
