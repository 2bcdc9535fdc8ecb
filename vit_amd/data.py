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
