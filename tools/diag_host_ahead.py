"""Is the Python thread ahead of the GPU through a headline step?  Host time stamps and HIP events at the same points
(after forward + loss, after backward, after the optimizer step); no synchronisation inside the measured steps.
usage: python tools/diag_host_ahead.py [--steps 6]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd.optim import AdamW  # noqa: E402
from vit_amd.train import build_model, label_smoothing_ce  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=6)
    a = ap.parse_args()
    me = [10, 20, 30, 40]
    net = build_model('disentangled_lai_attention', 'B', mask_epochs=me, dtype=torch.bfloat16)
    opt = AdamW(net.parameters(), lr=1e-4)
    x = torch.randn(128, 3, 224, 224, device='cuda')
    y = torch.randint(0, 1000, (128,), device='cuda')
    epoch = 30
    rec = []
    for step in range(a.steps + 3):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        t = [time.perf_counter()]
        ev[0].record()
        loss = label_smoothing_ce(net(x, epoch), y, 0.1)
        t.append(time.perf_counter()); ev[1].record()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        t.append(time.perf_counter()); ev[2].record()
        opt.step()
        t.append(time.perf_counter()); ev[3].record()
        rec.append((t, ev))
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    print('step   host: fwd  bwd  opt (ms)    gpu: fwd  bwd  opt (ms)    host start -> next host start')
    for i, (t, ev) in enumerate(rec):
        h = [1e3 * (t[k + 1] - t[k]) for k in range(3)]
        g = [ev[k].elapsed_time(ev[k + 1]) for k in range(3)]
        nxt = 1e3 * ((rec[i + 1][0][0] if i + 1 < len(rec) else t_end) - t[0])
        print(f'{i:3d}   {h[0]:7.2f} {h[1]:7.2f} {h[2]:7.2f}      {g[0]:7.2f} {g[1]:7.2f} {g[2]:7.2f}      {nxt:7.2f}')


if __name__ == '__main__':
    main()
