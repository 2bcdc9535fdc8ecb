"""Deterministic inputs / weights shared by ``tools/gen_goldens.py`` (which feeds them to the
reference) and the tests (which feed them to the oracle and to the HIP path).

Only seeds and shapes live here; the expected outputs live in ``tests/golden/*.npz``.
Every tensor is drawn from its own CPU ``torch.Generator`` seeded by (seed, crc32(name)), so
values do not depend on parameter registration order.
"""
import zlib

import torch


def _gen(seed, name):
    g = torch.Generator(device='cpu')
    g.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 63))
    return g


def randn(seed, name, *shape, scale=1.0):
    return torch.randn(*shape, generator=_gen(seed, name), dtype=torch.float32) * scale


def fill_state_dict(sd, seed):
    """Return a new state_dict (same keys / shapes, float32) with deterministic values:
    matrices ~ N(0, 1/fan_in), LayerNorm-like ``weight`` vectors ~ 1 + 0.1 N(0,1),
    other vectors ~ 0.1 N(0,1), embedding tables / tokens ~ N(0,1) * 0.5."""
    out = {}
    for k, v in sd.items():
        shape = tuple(v.shape)
        if v.dim() >= 2 and ('rel_embeddings' in k or 'cls_token' in k or 'pos_embed' in k):
            t = randn(seed, k, *shape, scale=0.5)
        elif v.dim() >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = randn(seed, k, *shape, scale=fan_in ** -0.5)
        elif k.endswith('weight') and ('norm' in k.lower()):
            t = 1.0 + randn(seed, k, *shape, scale=0.1)
        else:
            t = randn(seed, k, *shape, scale=0.1)
        out[k] = t
    return out


def sub(t, step=1, max_elems=6144):
    """Strided subsample used to keep large goldens small: flatten, then take every k-th
    element with k the smallest odd stride >= step that leaves <= max_elems values."""
    flat = t.detach().reshape(-1)
    k = max(step, -(-flat.numel() // max_elems))
    if k > 1 and k % 2 == 0:
        k += 1
    return flat[::k].contiguous()


LAI_MASK_EPOCHS = [10, 20, 30, 40]
# one epoch per stage: w=8, 12, 18, 20, unmasked
LAI_STAGE_EPOCHS = [0, 10, 20, 30, 40]

# (name, B, N, D, H, subsample step)
ATTN_SHAPES = [
    ('s', 2, 65, 128, 4, 1),      # dh 32
    ('m', 1, 197, 768, 12, 7),    # dh 64
    ('l', 1, 197, 320, 4, 5),     # dh 80
]
