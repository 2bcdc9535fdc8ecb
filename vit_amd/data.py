"""Device-side input normalisation: the reference's loaders end in
``ToTensor()`` + ``Normalize(mean, std)`` on CPU workers (train.py:340-341 CIFAR-10, :362-363 / :385
ImageNet) and ship fp32 tensors.  Here a loader may stop before those two transforms and ship the raw uint8
pixels (4x fewer bytes over PCIe, no per-pixel float work on the host): the patch-embedding gather applies
x / 255 and (x - mean) / std on the fly (``vitk_patchify_u8``), bit-identical in fp32 to torchvision's arithmetic.
Augmentations (flip, ColorJitter, RandomResizedCrop: train.py:359-361) stay where they are, before this step.
"""
CIFAR10_NORM = ((0.4914, 0.4822, 0.4465), (0.2023, 0.1994, 0.2010))       # train.py:341
IMAGENET_NORM = ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))            # train.py:363


def set_input_norm(net, mean, std, layout='nchw'):
    """Let ``net`` (model.VisionTransformer / VisionTransformerLAI) take uint8 images: [B,C,H,W] ('nchw', e.g.
    torchvision's PILToTensor) or [B,H,W,C] ('nhwc', decoder order).  Float images are still taken as already
    normalised, as the reference's loaders deliver them."""
    if layout not in ('nchw', 'nhwc'):
        raise ValueError("layout must be 'nchw' or 'nhwc'")
    mean, std = tuple(float(v) for v in mean), tuple(float(v) for v in std)
    if len(mean) != len(std) or not 1 <= len(mean) <= 4 or any(v == 0 for v in std):
        raise ValueError('mean / std: one non-degenerate pair per channel, at most 4 channels')
    net.input_norm = (mean, std, layout)
    return net


# ------------------------------------------------------------------------------------------------------------------
# Sharded loading and batched augmentation (train.py:356-400).  The reference builds, per rank, an ImageFolder with
# RandomHorizontalFlip + ColorJitter(0.4, 0.4, 0.4) + RandomResizedCrop(size, scale=(0.7, 1)) on CPU workers and a
# torch.utils.data.DistributedSampler.  Here the workers only decode (uint8 out); the three augmentations run as a few
# batched tensor ops on the GPU (one launch sequence per batch, not per image), then the uint8 patchify takes over.
# torchvision is not part of this image: the formulas below are torchvision's documented ones (functional_tensor:
# brightness = blend with zero, contrast = blend with the mean of the grey image, saturation = blend with the grey
# image, grey = 0.299 R + 0.587 G + 0.114 B; RandomResizedCrop.get_params: area uniform in scale, aspect log-uniform
# in [3/4, 4/3], ten tries then a centre crop), pinned by closed-form tests -- "parity unpinned" against torchvision's
# random streams, which are a property of its generator use, not of the transform.
import math

import torch


def shard_loader(dataset, batch_size, rank=None, world=None, shuffle=True, seed=0, num_workers=0, drop_last=False,
                 prefetch_factor=None):
    """(loader, sampler) for one rank: torch's DistributedSampler (train.py:367, 388: padded to a multiple of the
    world size, rank r takes indices r, r + world, ...) when world > 1, RandomSampler / SequentialSampler otherwise
    (train.py:370, 391); pinned host buffers as in the reference.  Call ``sampler.set_epoch(e)`` once per epoch when
    it has one (the reference never does, so its ranks see the same order every epoch; seed is its default 0)."""
    import torch.distributed as dist
    from torch.utils import data as tud
    if world is None:
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    if world > 1:
        sampler = tud.DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=shuffle, seed=seed, drop_last=drop_last)
    elif shuffle:
        sampler = tud.RandomSampler(dataset, generator=torch.Generator().manual_seed(seed))
    else:
        sampler = tud.SequentialSampler(dataset)
    kw = dict(prefetch_factor=prefetch_factor) if (num_workers > 0 and prefetch_factor) else {}
    loader = tud.DataLoader(dataset, batch_size=batch_size, sampler=sampler, num_workers=num_workers, pin_memory=True,
                            drop_last=drop_last, **kw)
    return loader, sampler


def _grey(x):
    return (0.299 * x[:, 0:1] + 0.587 * x[:, 1:2] + 0.114 * x[:, 2:3])


def _blend(a, b, f):
    return (f * a + (1.0 - f) * b).clamp_(0.0, 255.0)


def crop_params(n, height, width, scale=(0.7, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), generator=None):
    """RandomResizedCrop.get_params for n images at once (CPU tensors): top, left, h, w, int64 [n] each."""
    area = height * width
    top = torch.zeros(n, dtype=torch.int64)
    left = torch.zeros(n, dtype=torch.int64)
    hh = torch.zeros(n, dtype=torch.int64)
    ww = torch.zeros(n, dtype=torch.int64)
    done = torch.zeros(n, dtype=torch.bool)
    for _ in range(10):
        target = area * torch.empty(n).uniform_(scale[0], scale[1], generator=generator)
        logr = torch.empty(n).uniform_(math.log(ratio[0]), math.log(ratio[1]), generator=generator)
        ar = torch.exp(logr)
        w = torch.round(torch.sqrt(target * ar)).long()
        h = torch.round(torch.sqrt(target / ar)).long()
        ok = (w > 0) & (w <= width) & (h > 0) & (h <= height) & ~done
        u = torch.rand(n, generator=generator)
        v = torch.rand(n, generator=generator)
        t = (u * (height - h + 1).clamp_min(1)).long()
        l_ = (v * (width - w + 1).clamp_min(1)).long()
        top, left = torch.where(ok, t, top), torch.where(ok, l_, left)
        hh, ww = torch.where(ok, h, hh), torch.where(ok, w, ww)
        done |= ok
    # fallback: central crop at the nearest legal aspect (whole image when the aspect is already legal)
    in_ratio = width / height
    if in_ratio < ratio[0]:
        fw, fh = width, int(round(width / ratio[0]))
    elif in_ratio > ratio[1]:
        fh, fw = height, int(round(height * ratio[1]))
    else:
        fw, fh = width, height
    top = torch.where(done, top, torch.full_like(top, (height - fh) // 2))
    left = torch.where(done, left, torch.full_like(left, (width - fw) // 2))
    hh = torch.where(done, hh, torch.full_like(hh, fh))
    ww = torch.where(done, ww, torch.full_like(ww, fw))
    return top, left, hh, ww


def augment_batch(img_u8, size, generator=None, flip_p=0.5, jitter=(0.4, 0.4, 0.4), scale=(0.7, 1.0)):
    """The training transform of train.py:359-361 on a whole uint8 batch [B, 3, H, W] on its device (no per-image
    launches): horizontal flip, ColorJitter(brightness, contrast, saturation) in a random order per BATCH
    (torchvision permutes per image; the distribution of each factor is the same), RandomResizedCrop(size, scale)
    with bilinear resampling.  Returns uint8 [B, 3, size, size], ready for the uint8 patch-embedding path.
    ``generator``: a CPU torch.Generator (the parameters are a few numbers per image, drawn on the host)."""
    if img_u8.dtype != torch.uint8 or img_u8.dim() != 4 or img_u8.shape[1] != 3:
        raise ValueError('augment_batch: uint8 [B, 3, H, W] images')
    B, _, H, W = img_u8.shape
    dev = img_u8.device
    x = img_u8.float()
    flip = torch.rand(B, generator=generator) < flip_p
    x = torch.where(flip.to(dev)[:, None, None, None], x.flip(-1), x)
    fac = [torch.empty(B).uniform_(max(0.0, 1.0 - j), 1.0 + j, generator=generator).to(dev)[:, None, None, None] for j in jitter]
    for op in torch.randperm(3, generator=generator).tolist():
        if jitter[op] == 0:
            continue
        if op == 0:
            x = _blend(x, torch.zeros_like(x), fac[0])
        elif op == 1:
            # torchvision rounds nothing here for float images; for uint8 it blends in float and truncates at the end
            x = _blend(x, _grey(x).mean(dim=(1, 2, 3), keepdim=True), fac[1])
        else:
            x = _blend(x, _grey(x), fac[2])
    top, left, hh, ww = crop_params(B, H, W, scale=scale, generator=generator)
    # one grid_sample for all crops: output pixel centres mapped into each image's crop box (align_corners=False)
    ys = (torch.arange(size, device=dev, dtype=torch.float32) + 0.5) / size
    top, left, hh, ww = (t.to(dev).float() for t in (top, left, hh, ww))
    gy = ((top[:, None] + ys[None, :] * hh[:, None]) / H) * 2.0 - 1.0          # [B, size]
    gx = ((left[:, None] + ys[None, :] * ww[:, None]) / W) * 2.0 - 1.0
    grid = torch.stack([gx[:, None, :].expand(B, size, size), gy[:, :, None].expand(B, size, size)], dim=-1)
    out = torch.nn.functional.grid_sample(x, grid, mode='bilinear', padding_mode='border', align_corners=False)
    return out.round_().clamp_(0.0, 255.0).to(torch.uint8)
