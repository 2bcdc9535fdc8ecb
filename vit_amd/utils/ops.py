"""Host-side helpers mirroring the reference's ``utils/ops.py`` for the hot-path subset:
build_relative_position (:61-70), AverageMeter (:82), accuracy (:106), the distributed helpers
(:124-169).  XSoftmax (:14, never called) and the log-bucket positions (:72, unreachable with
position_buckets = -1) are dead code upstream and are not restated."""
import os

import torch
import torch.distributed as dist


def build_relative_position(query_size, key_size, bucket_size=-1, max_position=-1):
    """int64 [1, query_size, key_size] with entry (i, j) = i - j.  The kernels regenerate this
    from indices; the tensor exists for API parity and tests."""
    if bucket_size > 0 and max_position > 0:
        raise NotImplementedError('log-bucketed positions are unreachable in the reference (position_buckets=-1)')
    q = torch.arange(query_size, dtype=torch.long)
    k = torch.arange(key_size, dtype=torch.long)
    return (q[:, None] - k[None, :]).unsqueeze(0)


class AverageMeter(object):
    """Computes and stores the average and current value"""

    def __init__(self, name, fmt=':f'):
        self.name, self.fmt = name, fmt
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count

    def __str__(self):
        return ('{name} {val' + self.fmt + '} ({avg' + self.fmt + '})').format(**self.__dict__)


def accuracy(output, target, topk=(1,)):
    """Top-k accuracies in percent, computed on the device without a sync."""
    with torch.no_grad():
        maxk = max(topk)
        _, pred = output.topk(maxk, 1, True, True)
        correct = pred.t().eq(target.view(1, -1))
        return [correct[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / target.size(0)) for k in topk]


def init_distributed_mode(args):
    """One process per GPU, env:// rendezvous; 'nccl' is RCCL on ROCm (gloo for CPU rehearsals)."""
    args.num_gpu = 1
    args.local_rank = int(os.environ.get('LOCAL_RANK', getattr(args, 'local_rank', 0)))
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        args.device = 'cuda:%d' % args.local_rank
        torch.cuda.set_device(args.local_rank)
    else:
        args.device = 'cpu'
    if not dist.is_initialized():
        from ..ddp import init_process_group      # RCCL's stream in the high-priority class (a hardware queue of its own)
        init_process_group('nccl' if use_gpu else 'gloo', init_method='env://')
    args.world_size = dist.get_world_size()
    args.rank = dist.get_rank()


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0
