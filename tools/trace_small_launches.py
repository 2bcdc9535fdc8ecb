"""Which Python lines launch the small torch kernels (fills, copies, adds) of a training step?
    python tools/trace_small_launches.py [--workload dislai_b16] [--band 20]
torch.profiler with stacks over ONE step after warm-up; prints, per aten op that launches a kernel, the innermost
vit_amd frame and the call count.  Diagnostic only (not part of the product path)."""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench as B  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='dislai_b16')
    ap.add_argument('--band', type=int, default=20)
    args = ap.parse_args()
    from vit_amd.train import build_model, train_step, takes_epoch
    from vit_amd.optim import AdamW
    variant, preset, img, patch, classes, batch = B.WORKLOADS[args.workload]
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    net = build_model(variant, preset, img_size=img, patch_size=patch, num_classes=classes,
                      mask_epochs=B.MASK_EPOCHS if 'lai' in variant else None, dtype=torch.bfloat16, device=dev)
    opt = AdamW(net.parameters(), lr=1e-4)
    epoch = B.STAGE_EPOCH[args.band] if takes_epoch(net) else None
    x = torch.randn(batch, 3, img, img, device=dev)
    y = torch.randint(0, classes, (batch,), device=dev)
    for _ in range(3):
        train_step(net, opt, x, y, epoch)
    torch.cuda.synchronize()
    # torch.profiler's with_stack comes back empty here: count at the Python level instead
    import traceback
    count = collections.Counter()

    def site():
        for fr in reversed(traceback.extract_stack()[:-2]):
            if 'vit_amd' in fr.filename or fr.filename.endswith('bench.py'):
                return f'{os.path.relpath(fr.filename)}:{fr.lineno} {fr.line}'
        return '?'

    def wrap(owner, name, is_method):
        orig = getattr(owner, name)

        def f(*a, **k):
            r = orig(*a, **k)
            t = a[0] if is_method else r
            if isinstance(t, torch.Tensor) and t.is_cuda:
                count[(name, site())] += 1
            return r
        setattr(owner, name, f)
        return orig

    saved = [(o, n, wrap(o, n, m)) for o, n, m in [
        (torch, 'zeros', False), (torch, 'zeros_like', False), (torch, 'cat', False), (torch, 'full', False),
        (torch.Tensor, 'zero_', True), (torch.Tensor, 'fill_', True), (torch.Tensor, 'copy_', True),
        (torch.Tensor, 'contiguous', True), (torch.Tensor, 'to', True), (torch.Tensor, 'add_', True),
        (torch.Tensor, 'clone', True), (torch.Tensor, 'float', True), (torch.Tensor, 'bfloat16', True)]]
    train_step(net, opt, x, y, epoch)
    torch.cuda.synchronize()
    for o, n, orig in saved:
        setattr(o, n, orig)
    for (name, where), n in sorted(count.items(), key=lambda kv: -kv[1])[:60]:
        print(f'{n:5d}  {name:12s} {where}')


if __name__ == '__main__':
    main()
