"""Control for tests/test_gpu_ddp_world2.py: no vit_amd code at all.  Two ranks on cuda:0, an async gloo all_reduce of a
CUDA tensor between torch matmuls, 40 steps.
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 \
        tools/gloo_one_gpu_control.py [elements]
On this image (torch 2.10.0+rocm7.0, one MI355X shared by both ranks) it hangs in work.wait() on both ranks: 5 runs of 5
(four before step 5, one after step 30) -- the same hang as the reducer rehearsals, so that hang is gloo's CUDA staging
with two processes on one device, not the reducer.  faulthandler ends a hung run after 40 s."""
import os, sys, faulthandler
import torch, torch.distributed as dist
rank = int(os.environ['RANK'])
faulthandler.dump_traceback_later(40, exit=True, file=sys.stderr)
torch.cuda.set_device(0)
dist.init_process_group('gloo')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
a = torch.randn(2048, 2048, device='cuda')
flat = torch.zeros(n, device='cuda')
for i in range(40):
    b = a @ a
    flat.add_(b.view(-1)[:n] * 1e-6)
    h = dist.all_reduce(flat, async_op=True)
    c = b @ a
    h.wait()
    flat.mul_(0.5)
    if i % 5 == 4:
        print(f'[control rank {rank}] step {i} queued', file=sys.stderr, flush=True)
torch.cuda.synchronize()
print(f'[control rank {rank}] done {float(flat[0])}', file=sys.stderr, flush=True)
dist.barrier()
dist.destroy_process_group()
