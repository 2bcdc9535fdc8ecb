"""The train step as ONE HIP graph: for the launch-bound configurations.

A step of the reference loop (train.py:530-554) is ~1100 kernel launches for the 24-layer model and
~600 for the 12-layer ones; where the kernels are short (S-preset at small batch, the CIFAR plumbing
config) the Python thread cannot keep ahead of the GPU and the step takes 1.3-3.5x its kernel time.
Every launch of this package goes to torch's current stream through the C-ABI and nothing in a step
synchronises or reads device memory on the host, so the whole step -- forward, loss, backward, AdamW
with its weight copies -- captures into one graph; the only values that change between replays
(AdamW's bias corrections and learning rate) live in device memory (vitk_adamw_dyn).

Data parallel: with ``reducer`` (vit_amd.ddp.GradReducer, built by the eager warm-up steps) the gradient exchange is
part of the graph -- the hooks' bucket all-reduces are RCCL launches on RCCL's stream, forked from and joined to the
capturing stream by events, all of which a stream capture records; the .grad views into the flat slabs, the AdamW
launch tables over them and the bucket layout are fixed after the first step, so every replay runs the same exchange.
Every rank must capture and replay in step (a collective inside a graph is still a collective).
"""
import torch

from .optim import AdamW
from .train import train_step


class GraphedTrainStep:
    """``step(img, target) -> loss`` replaying a captured train_step(net, optimizer, ...).

    optimizer must be vit_amd.optim.AdamW(capturable=True).  ``epoch`` (the LAI stage) is fixed per capture:
    build another GraphedTrainStep when the stage changes (four times in a training run).
    """

    def __init__(self, net, optimizer, img, target, epoch=None, smoothing=0.1, warmup=3, reducer=None):
        if not isinstance(optimizer, AdamW) or not optimizer.capturable:
            raise ValueError('GraphedTrainStep needs vit_amd.optim.AdamW(capturable=True)')
        from . import dropout as VD
        if VD.active(net):
            raise ValueError('GraphedTrainStep: dropout is active (p > 0 in training mode); a captured graph bakes the '
                             'dropout streams into its kernel arguments and every replay would repeat the same masks')
        self.net, self.optimizer, self.epoch, self.smoothing = net, optimizer, epoch, smoothing
        self.reducer = reducer
        if reducer is not None and warmup < 2:
            raise ValueError('GraphedTrainStep: the reducer builds its buckets on the first eager step; warmup >= 2')
        self.img = img.clone()
        self.target = target.clone()
        # warm-up on a side stream (weight copies, optimizer state, kernel attributes all get created eagerly)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                optimizer.update_dyn()
                train_step(net, optimizer, self.img, self.target, epoch, smoothing, reducer=reducer)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        optimizer.update_dyn()
        torch.cuda.synchronize()
        with torch.cuda.graph(self.graph):
            self.loss = train_step(net, optimizer, self.img, self.target, epoch, smoothing, reducer=reducer)
        # the capture itself executed nothing: that step's dyn values are consumed by the first replay
        self._pending = True

    def step(self, img, target):
        self.img.copy_(img, non_blocking=True)
        self.target.copy_(target, non_blocking=True)
        if self._pending:
            self._pending = False
        else:
            self.optimizer.update_dyn()
        self.graph.replay()
        return self.loss
