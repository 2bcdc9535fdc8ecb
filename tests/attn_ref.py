"""Closed-form attention reference used by the GPU kernel tests (plain torch, any device/dtype)."""
import torch


def attention_ref(q, k, v, pk, pq, scale, w):
    """q,k,v [B,N,H,dh]; pk,pq [2N-1,H,dh] or None; returns out [B,N,H,dh], lse [B,H,N], probs."""
    B, N, H, dh = q.shape
    s = torch.einsum('bihd,bjhd->bhij', q, k)
    idx = torch.arange(N, device=q.device)
    delta = idx[:, None] - idx[None, :] + (N - 1)
    if pk is not None:
        c2p = torch.einsum('bihd,rhd->bhir', q, pk)            # [B,H,N,R]
        s = s + torch.gather(c2p, -1, delta.expand(B, H, N, N))
        p2c = torch.einsum('bjhd,rhd->bhjr', k, pq)            # [B,H,N(j),R]
        s = s + torch.gather(p2c, -1, delta.t().expand(B, H, N, N)).transpose(-1, -2)
    s = s * scale
    if w is not None and w >= 0:
        s = s.masked_fill((idx[:, None] - idx[None, :]).abs() > w, float('-inf'))
    lse = torch.logsumexp(s, -1)
    p = torch.softmax(s, -1)
    out = torch.einsum('bhij,bjhd->bihd', p, v)
    return out, lse, p
