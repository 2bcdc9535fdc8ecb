"""Host logic of vit_amd/data.py (input normalisation config).  CPU."""
import pytest
import torch

from vit_amd import data


def test_set_input_norm_validates_and_stores():
    net = torch.nn.Identity()
    assert data.set_input_norm(net, *data.IMAGENET_NORM) is net
    assert net.input_norm == (data.IMAGENET_NORM[0], data.IMAGENET_NORM[1], 'nchw')
    data.set_input_norm(net, [0.1307], [0.3081], layout='nhwc')            # MNIST, train.py:291-292
    assert net.input_norm == ((0.1307,), (0.3081,), 'nhwc')
    for bad in (dict(mean=(0.5,), std=(0.5, 0.5)), dict(mean=(0.5,) * 5, std=(0.5,) * 5), dict(mean=(0.5,), std=(0.0,)),
                dict(mean=(0.5,), std=(0.5,), layout='hwc')):
        with pytest.raises(ValueError):
            data.set_input_norm(net, **bad)


def test_reference_constants():
    assert data.CIFAR10_NORM == ((0.4914, 0.4822, 0.4465), (0.2023, 0.1994, 0.2010))    # train.py:341
    assert data.IMAGENET_NORM == ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))         # train.py:363


def test_shard_loader_partitions_like_distributed_sampler():
    ds = torch.utils.data.TensorDataset(torch.arange(103))
    seen = []
    for r in range(4):
        loader, sampler = data.shard_loader(ds, batch_size=8, rank=r, world=4, shuffle=True, seed=3)
        assert isinstance(sampler, torch.utils.data.DistributedSampler)
        sampler.set_epoch(1)
        idx = torch.cat([b[0] for b in loader])
        ref = list(torch.utils.data.DistributedSampler(ds, num_replicas=4, rank=r, shuffle=True, seed=3))   # epoch 0
        assert len(idx) == 26 and len(ref) == 26               # 103 padded to 104
        seen.append(idx)
    allidx = torch.cat(seen)
    assert set(allidx.tolist()) == set(range(103))             # every sample on some rank, one repeated as padding
    loader, sampler = data.shard_loader(ds, batch_size=8, world=1, rank=0, shuffle=False)
    assert isinstance(sampler, torch.utils.data.SequentialSampler)
    assert torch.equal(torch.cat([b[0] for b in loader]), torch.arange(103))


def test_augment_batch_closed_forms():
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (6, 3, 40, 56), dtype=torch.uint8, generator=g)
    # no jitter, no flip, full-area crop at the image's own size and aspect: identity
    same = data.augment_batch(img[:, :, :, :40], 40, generator=g, flip_p=0.0, jitter=(0, 0, 0), scale=(1.0, 1.0))
    t, l_, h, w = data.crop_params(1000, 40, 40, scale=(1.0, 1.0), generator=g)
    full = (h == 40) & (w == 40)
    assert full.float().mean() > 0.3                                       # aspect draws near 1 give the whole image ...
    assert torch.all((h * w).float() <= 40 * 40 * 1.0 + 1)                 # ... and no crop exceeds the image
    assert same.shape == (6, 3, 40, 40) and same.dtype == torch.uint8
    # always flip, nothing else, whole image: exact mirror (centre crop fallback is the whole square image)
    sq = img[:, :, :, :40].contiguous()
    params_full = lambda *a, **k: (torch.zeros(6, dtype=torch.int64), torch.zeros(6, dtype=torch.int64),
                                   torch.full((6,), 40), torch.full((6,), 40))
    keep = data.crop_params
    data.crop_params = params_full
    try:
        assert torch.equal(data.augment_batch(sq, 40, generator=g, flip_p=1.0, jitter=(0, 0, 0)), sq.flip(-1))
        assert torch.equal(data.augment_batch(sq, 40, generator=g, flip_p=0.0, jitter=(0, 0, 0)), sq)
        # brightness alone: x * f, clamped -- recover f from a mid-grey image
        grey = torch.full((6, 3, 40, 40), 100, dtype=torch.uint8)
        out = data.augment_batch(grey, 40, generator=g, flip_p=0.0, jitter=(0.4, 0, 0)).float()
        f = out[:, 0, 0, 0] / 100.0
        assert torch.all((f >= 0.59) & (f <= 1.41)) and f.std() > 0.05
        assert torch.all(out == out[:, :1, :1, :1])                        # uniform in, uniform out
        # saturation / contrast leave a grey image grey (its grey version / mean is itself)
        assert torch.equal(data.augment_batch(grey, 40, generator=g, flip_p=0.0, jitter=(0, 0.4, 0.4)), grey)
        # a 2x downscale of the whole image = 2x2 box average under bilinear sampling at pixel centres
        half = data.augment_batch(sq, 20, generator=g, flip_p=0.0, jitter=(0, 0, 0)).float()
        box = sq.float().view(6, 3, 20, 2, 20, 2).mean(dim=(3, 5))
        assert (half - box).abs().max() <= 0.5 + 1e-3
    finally:
        data.crop_params = keep
    # area law of the sampled crops
    t, l_, h, w = data.crop_params(4000, 224, 224, generator=g)
    frac = (h * w).float() / (224 * 224)
    # large areas at a non-square aspect do not fit and are redrawn: the mean sits below the 0.85 of the uniform draw
    assert 0.69 < frac.min() and frac.max() < 1.02 and 0.77 < frac.mean().item() < 0.85
    assert torch.all(t + h <= 224) and torch.all(l_ + w <= 224) and torch.all(t >= 0) and torch.all(l_ >= 0)
    r = w.float() / h.float()
    assert r.min() > 0.73 and r.max() < 1.36
