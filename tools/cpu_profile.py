"""Host-side cost of one training step (cProfile): batch 8 keeps the GPU far ahead, so wall time = CPU time."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd.train import build_model, train_step
from vit_amd.optim import AdamW

wl = sys.argv[1] if len(sys.argv) > 1 else 'disentangled_lai_attention'
preset = sys.argv[2] if len(sys.argv) > 2 else 'B'
net = build_model(wl, preset, mask_epochs=[10, 20, 30, 40] if 'lai' in wl else None)
opt = AdamW(net.parameters(), lr=1e-4)
img = torch.randn(8, 3, 224, 224, device='cuda')
tgt = torch.randint(0, 1000, (8,), device='cuda')
ep = 30 if 'lai' in wl else None
for _ in range(3):
    train_step(net, opt, img, tgt, ep)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    train_step(net, opt, img, tgt, ep)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f'host time per step {(t1 - t0) / 5 * 1e3:.2f} ms (until the last launch returned), '
      f'{(time.perf_counter() - t0) / 5 * 1e3:.2f} ms with the GPU drained')
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    train_step(net, opt, img, tgt, ep)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
