"""Time the attention kernels alone at the headline shape (per-GPU batch 128, 197 tokens, 16 heads x 64,
disentangled bias, band w): forward, backward, backward with the fused bias-gradient sums.
usage: python tools/bench_attn.py [--batch 128] [--w 20] [--plain] [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('VITK_LIB'):          # a variant build of the library (tools/lai_phases.sh)
    from vit_amd import _lib
    _lib.LIB_PATH = os.environ['VITK_LIB']
from vit_amd import ops  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--tokens', type=int, default=197)
    ap.add_argument('--heads', type=int, default=16)
    ap.add_argument('--dh', type=int, default=64)
    ap.add_argument('--w', type=int, default=20)
    ap.add_argument('--plain', action='store_true')
    ap.add_argument('--iters', type=int, default=20)
    a = ap.parse_args()
    B, N, H, dh = a.batch, a.tokens, a.heads, a.dh
    D = H * dh
    dev = 'cuda'
    qkv = (torch.randn(B * N, 3 * D, device=dev) * 0.5).bfloat16()
    q, k, v = (qkv[:, i * D:(i + 1) * D] for i in range(3))
    pk = pq = dpk = dpq = None
    if not a.plain:
        tab = (torch.randn(2 * N - 1, 2 * D, device=dev) * 0.5).bfloat16()
        pq, pk = tab[:, :D], tab[:, D:]
        dtab = torch.zeros(2 * N - 1, 2 * D, device=dev)
        dpq, dpk = dtab[:, :D], dtab[:, D:]
    scale = (dh * (1 if a.plain else 3)) ** -0.5
    ldp = 2 * D
    out, lse = ops.attn_fwd(q, k, v, B, H, N, dh, 3 * D, pk, pq, ldp, scale, a.w)
    dout = torch.randn_like(out)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = (dqkv[:, i * D:(i + 1) * D] for i in range(3))
    from vit_amd._lib import lib
    ws = torch.empty(lib().vitk_attn_bwd_workspace_bytes(ops._dt(q), B, H, N, dh), dtype=torch.uint8, device=dev)
    dbias = torch.zeros(3 * D, device=dev)

    def bwd(db):
        ops.attn_bwd(q, k, v, B, H, N, dh, 3 * D, pk, pq, ldp, scale, a.w, out, dout, lse, dq, dk, dv, 3 * D,
                     dpk, dpq, workspace=ws, ld_dp=ldp, dbias=db)

    t_f = timed(lambda: ops.attn_fwd(q, k, v, B, H, N, dh, 3 * D, pk, pq, ldp, scale, a.w), a.iters)
    t_b = timed(lambda: bwd(None), a.iters)
    t_bb = timed(lambda: bwd(dbias), a.iters)
    t_cs = timed(lambda: ops.colsum(dqkv, out=dbias), a.iters)
    print(f'B {B} N {N} H {H} dh {dh} w {a.w} bias {not a.plain}: fwd {t_f:.1f} us  bwd {t_b:.1f} us  '
          f'bwd+dbias {t_bb:.1f} us  (separate colsum {t_cs:.1f} us)')


if __name__ == '__main__':
    main()
