"""Dropout streams for the fused kernels (the reference's nn.Dropout sites, train.py:79 ``--dropout 0.1``).

A dropout SITE (one nn.Dropout call of the reference) owns one stream (p, seed, offset): element i of the site's tensor is
kept iff hash(seed, offset, i) >= p * 2^32, evaluated inside the kernels (csrc/common.h: drop_keep) -- in the forward
and again in the backward, so no mask is ever stored.  ``seed`` is process-wide (``manual_seed``; it follows
``torch.initial_seed()`` until set), ``offset`` counts the sites drawn so far: every forward of every site gets a fresh
stream, and a run is reproducible from its seed.

The reference draws its masks from torch's Philox generator; a fused kernel cannot reproduce those bits, so parity at
p > 0 is defined with THIS mask handed to the oracle (ops.dropout_mask, tests/test_gpu_dropout.py).

Captured HIP graphs bake (seed, offset) into the kernel arguments: a replay would repeat its masks, so
vit_amd.graph.GraphedTrainStep refuses models whose dropout is active.
"""
import torch

_state = {'seed': None, 'offset': 0, 'log': None}


def manual_seed(seed):
    """Seed of every dropout stream drawn from now on; resets the site counter."""
    _state['seed'] = int(seed) & 0xFFFFFFFFFFFFFFFF
    _state['offset'] = 0


def get_state():
    return _state['seed'], _state['offset']


def next_stream(p):
    """(p, seed, offset) for one dropout site; (0, 0, 0) when p == 0 (no stream is consumed)."""
    p = float(p)
    if p <= 0.0:
        return (0.0, 0, 0)
    if not p < 1.0:
        raise ValueError('dropout probability must be in [0, 1)')
    if _state['seed'] is None:
        _state['seed'] = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
    off = _state['offset']
    _state['offset'] = off + 1
    s = (p, _state['seed'], off)
    if _state['log'] is not None:
        _state['log'].append(s)
    return s


class record:
    """Context manager: the streams drawn inside, in order (tests hand their masks to the oracle)."""

    def __enter__(self):
        self.prev = _state['log']
        _state['log'] = self.streams = []
        return self.streams

    def __exit__(self, *exc):
        _state['log'] = self.prev
        return False


def active(module):
    """True when ``module`` is in training mode and any nn.Dropout / attention dropout below it has p > 0."""
    if not module.training:
        return False
    for m in module.modules():
        if isinstance(m, torch.nn.Dropout) and m.p > 0:
            return True
        if isinstance(m, torch.nn.MultiheadAttention) and m.dropout > 0:
            return True
    return False
