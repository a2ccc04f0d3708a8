"""Drop-in for the reference's `src.backbones.ResNet34.Model` ("DeTone" 4-point regressor): a
torchvision-layout resnet34 with a 2-channel conv1 and an 8-way fc (src/backbones/ResNet34.py:6-50),
executed by the gfx950 kernels.  State-dict keys `resnet34.{conv1,bn1,layer1..4,fc}.*` as upstream.
"""
import os

import torch
import torch.nn as nn

from .. import net


class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))


class _ResNet34(nn.Module):
    def __init__(self, in_ch, num_out):
        super().__init__()
        self.conv1 = nn.Conv2d(in_ch, 64, 7, 2, 3, bias=False)            # ResNet34.py:17
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for i, (n, c) in enumerate(zip((3, 4, 6, 3), (64, 128, 256, 512))):
            blocks = []
            for j in range(n):
                blocks.append(_BasicBlock(cin, c, 2 if (j == 0 and i > 0) else 1))
                cin = c
            setattr(self, 'layer%d' % (i + 1), nn.Sequential(*blocks))
        self.fc = nn.Linear(512, num_out, bias=True)                       # ResNet34.py:19


class Model(nn.Module):

    def __init__(self, **kwargs):
        super().__init__()
        self.patch_keys = kwargs['PATCH_KEYS']
        self.target_keys = kwargs['TARGET_KEYS']
        pre = kwargs.get('PRETRAINED_RESNET')
        if pre is True:
            raise RuntimeError("PRETRAINED_RESNET=True downloads the torchvision ImageNet checkpoint upstream (ResNet34.py:15); "
                               "no network here - pass the path of resnet34-333f7ec4.pth as PRETRAINED_RESNET instead")
        self.resnet34 = _ResNet34(2, 8)
        self.variant = str.lower(kwargs['VARIANT']) if 'VARIANT' in kwargs else 'oneline'
        assert 'oneline' in self.variant or 'doubleline' in self.variant, 'Only OneLine or DoubleLine variant is supported'
        # optional extra kwarg (ignored upstream): conv arithmetic 'f32' (default: fp32 accuracy, kernels.PRECISION), 'f32-mfma' (fp32-input MFMA only) or 'bf16' (bf16 MFMA operands,
        # fp32 accumulate/storage - BASELINE.json configs[3])
        self.precision = kwargs.get('PRECISION', os.environ.get('BIHOME_PRECISION', 'f32'))
        net.to_kernel_layout_(self)
        if isinstance(pre, str) and pre:           # torchvision resnet34 ImageNet state dict; conv1 / fc stay fresh (ResNet34.py:15-19)
            from ..weights import load_imagenet_resnet34
            load_imagenet_resnet34(self, pre)
            net.to_kernel_layout_(self)
        self._runner = None
        net.install_counter_hooks(self)

    def _build(self):
        r = self.resnet34
        prog = net.Program()
        s = prog.conv(0, r.conv1, in_nchw=True)
        s = prog.maxpool(prog.bn(s, r.bn1, relu=True))
        for name in ('layer1', 'layer2', 'layer3', 'layer4'):
            for blk in getattr(r, name):
                s = prog.basic_block(s, blk)
        s = prog.conv(prog.gap(s), r.fc)
        return net.Runner(self, prog, trainable=True, precision=self.precision)

    def single_forward(self, x, groups=1):
        if self._runner is None:
            net.to_kernel_layout_(self)
            self._runner = self._build()
        return self._runner(x.contiguous(), groups).reshape(-1, 4, 2)       # ResNet34.py:28

    def forward(self, data):
        e1, e2 = self.patch_keys
        p1, p2 = data[e1], data[e2]
        x12 = torch.cat([p1, p2], dim=1)
        if self.variant == 'doubleline':
            B = p1.shape[0]
            out = self.single_forward(torch.cat([x12, torch.cat([p2, p1], dim=1)], dim=0), groups=2)
            data[self.target_keys[0]] = out[:B]
            data[self.target_keys[1]] = out[B:]
        else:
            data[self.target_keys[0]] = self.single_forward(x12)
        return data

    def predict_homography(self, data):
        return self.forward(data)

    def state_dict(self, *args, **kwargs):
        net.flush_counters(self)
        return super().state_dict(*args, **kwargs)
