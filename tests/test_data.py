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
