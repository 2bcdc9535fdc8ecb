"""Checkpoints in the reference's format, plus the resume path it lacks.

Reference: train.py:581-593 writes ``{'model', 'optimizer', 'scheduler', 'epoch', 'args'}`` with
``save_on_master`` (:619-620 -- which, despite its name, saves on EVERY rank) to ``checkpoint.pth`` each epoch and to
``checkpoint_<name>_<epoch:04>.pth`` every ``save_freq`` epochs; :609-615 writes the final model without the
scheduler.  Nothing in the reference ever loads one.

Here: the same dictionary (``vit_amd.optim.AdamW.state_dict()`` has torch.optim.AdamW's layout, so either optimizer
loads the other's file), written by rank 0 only, through a temporary file so that a killed job never leaves a
truncated checkpoint; and ``load_checkpoint`` to resume.  Weights are copied in place, which bumps their version
counters: the bf16 copies the GEMMs read (functional.shadow) are rebuilt at their next use.
"""
import argparse
import os

import torch
import torch.distributed as dist


def is_main_process():
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def save_on_master(obj, path):
    """torch.save(obj, path) on rank 0 only (atomically: write, then rename); other ranks return at once."""
    if not is_main_process():
        return
    path = os.fspath(path)
    tmp = path + '.tmp'
    torch.save(obj, tmp)
    os.replace(tmp, path)


def checkpoint_dict(model, optimizer, scheduler, epoch, args):
    d = {'model': model.state_dict(), 'optimizer': optimizer.state_dict(), 'epoch': epoch, 'args': args}
    if scheduler is not None:
        d['scheduler'] = scheduler.state_dict()
    return d


def save_checkpoint(path, model, optimizer, scheduler, epoch, args=None):
    """One checkpoint of train.py:586-593 (``scheduler=None`` gives the final-model form of :610-615)."""
    save_on_master(checkpoint_dict(model, optimizer, scheduler, epoch, args), path)


def load_checkpoint(path, model, optimizer=None, scheduler=None, map_location='cpu'):
    """Restore what save_checkpoint wrote (or the reference's own files); returns (epoch, args).
    Loaded with ``weights_only=True``: only tensors, containers, numbers and an argparse.Namespace are accepted."""
    with torch.serialization.safe_globals([argparse.Namespace]):
        ck = torch.load(os.fspath(path), map_location=map_location, weights_only=True)
    for key in ('model', 'epoch'):
        if key not in ck:
            raise KeyError(f'{path}: not a checkpoint of this format (no {key!r})')
    sd = ck['model']
    if sd and all(k.startswith('module.') for k in sd):
        # the reference saves model.state_dict() of the DDP- / DataParallel-wrapped model in its multi-GPU modes
        # (train.py:588 after :635 / :646): every key then carries the wrapper's 'module.' prefix
        sd = {k[len('module.'):]: v for k, v in sd.items()}
    model.load_state_dict(sd)
    if optimizer is not None:
        if 'optimizer' not in ck:
            raise KeyError(f'{path}: no optimizer state')
        optimizer.load_state_dict(ck['optimizer'])
    if scheduler is not None:
        if 'scheduler' not in ck:
            raise KeyError(f'{path}: no scheduler state (a final-model file, train.py:610-615?)')
        scheduler.load_state_dict(ck['scheduler'])
    return ck['epoch'], ck.get('args')
