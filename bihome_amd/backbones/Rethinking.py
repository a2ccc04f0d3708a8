"""Drop-in for the reference's `src.backbones.Rethinking.Model` ("Zeng" perspective-field network,
RESNET_BLOCK='ResNet34'), executed by hand-written gfx950 kernels.

Same plugin contract: `Model(**cfg['MODEL']['BACKBONE'])`, `forward(data) -> data`
adds `data[TARGET_KEYS[0]]` (and `[1]` for VARIANT='DoubleLine'), `predict_homography = forward`, and the
same state-dict keys (`layerN.k.upper_branch.j.weight` ...), so upstream checkpoints load.  The
nn.Conv2d / nn.BatchNorm2d leaves below are parameter containers only; the arithmetic is
bihome_amd.net.run_forward / run_backward.  Reference: src/backbones/Rethinking.py:13-156, 284-316 and
src/backbones/utils.py:60-131.
"""
import os

import torch
import torch.nn as nn

from .. import net


class _Branches(nn.Module):
    """A residual unit holding `upper_branch` (and optionally `lower_branch`) Sequentials."""

    def __init__(self, upper, lower=None):
        super().__init__()
        self.upper_branch = nn.Sequential(*upper)
        if lower is not None:
            self.lower_branch = nn.Sequential(*lower)


def _conv(cin, cout, k, s, p, bias=False):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=s, padding=p, bias=bias)


def resnet34_unit(cin, cout, stride=1):
    # utils.py:85-131 (ConvBlock when the shape changes, IdentityBlock otherwise)
    upper = [_conv(cin, cout, 3, stride, 1), nn.BatchNorm2d(cout), nn.ReLU(),
             _conv(cout, cout, 3, 1, 1), nn.BatchNorm2d(cout)]
    lower = [_conv(cin, cout, 1, stride, 0), nn.BatchNorm2d(cout)] if cin != cout else None
    return _Branches(upper, lower)


def deconv_unit(c):
    # utils.py:60-82 (ResNet50DeconvBlock, used by both variants)
    upper = [nn.ConvTranspose2d(c, c, kernel_size=2, stride=2), _conv(c, c, 3, 1, 1), nn.BatchNorm2d(c), nn.ReLU(),
             _conv(c, c // 2, 1, 1, 0), nn.BatchNorm2d(c // 2)]
    lower = [nn.ConvTranspose2d(c, c // 2, kernel_size=2, stride=2, bias=False), nn.BatchNorm2d(c // 2)]
    return _Branches(upper, lower)


class Model(nn.Module):

    def __init__(self, **kwargs):
        super().__init__()
        self.image_size = kwargs.get('IMAGE_SIZE')
        self.patch_keys = kwargs['PATCH_KEYS']
        self.target_keys = kwargs['TARGET_KEYS']
        self.resnet_block = kwargs['RESNET_BLOCK']
        self.variant = str.lower(kwargs['VARIANT']) if 'VARIANT' in kwargs else 'oneline'
        assert 'oneline' in self.variant or 'doubleline' in self.variant, 'Only OneLine or DoubleLine variant is supported'
        if self.resnet_block != 'ResNet34':
            raise NotImplementedError("only RESNET_BLOCK='ResNet34' is built (every shipped config uses it)")
        pre = kwargs.get('PRETRAINED_RESNET')
        if pre is True:
            raise RuntimeError("PRETRAINED_RESNET=True downloads resnet34-333f7ec4.pth upstream (Rethinking.py:178-183); no "
                               "network here - pass the path of that file as PRETRAINED_RESNET instead")
        S, U, D = nn.Sequential, resnet34_unit, deconv_unit
        # optional extra kwarg (upstream hard-wires 2 = two grayscale patches, Rethinking.py:31): channels per patch;
        # 3 gives the 6-channel stem of BASELINE.json configs[4] (256x256 RGB pairs)
        self.patch_channels = int(kwargs.get('PATCH_CHANNELS', 1))
        self.layer1 = S(_conv(2 * self.patch_channels, 64, 7, 2, 3), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(3, 2, 1))
        self.layer2 = S(U(64, 64), U(64, 64), U(64, 64))
        self.layer3 = S(U(64, 128, 2), U(128, 128), U(128, 128), U(128, 128))
        self.layer4 = S(U(128, 256, 2), *[U(256, 256) for _ in range(5)], D(256))
        self.layer5 = S(U(128, 128), U(128, 128), U(128, 128), D(128))
        self.layer6 = S(U(64, 64), U(64, 64), D(64))
        self.layer7 = S(U(32, 32), D(32))
        self.layer8 = S(nn.Conv2d(16, 128, 1), nn.BatchNorm2d(128), nn.ReLU(), nn.Conv2d(128, 2, 1))
        # optional extra kwarg (ignored upstream): conv arithmetic 'f32' (default: fp32 accuracy, kernels.PRECISION), 'f32-mfma' (fp32-input MFMA only) or 'bf16' (bf16 MFMA operands,
        # fp32 accumulate/storage - BASELINE.json configs[3])
        self.precision = kwargs.get('PRECISION', os.environ.get('BIHOME_PRECISION', 'f32'))
        net.to_kernel_layout_(self)
        self._runner = None
        net.install_counter_hooks(self)
        self.fuse_tail = os.environ.get("BIHOME_FUSE_TAIL", "1") != "0"
        if isinstance(pre, str) and pre:           # path of the torchvision resnet34 ImageNet state dict (Rethinking.py:158-282)
            from ..weights import load_imagenet_resnet34
            load_imagenet_resnet34(self, pre)
            net.to_kernel_layout_(self)

    # ---- program ---------------------------------------------------------------------------------
    def _build(self):
        prog = net.Program()
        s = 0
        first = True
        for i in range(1, 9):
            layer = getattr(self, 'layer%d' % i)
            if first:
                # network input arrives NCHW (two stacked grayscale planes): the first conv reads it as is
                mods = list(layer)
                s = prog.conv(s, mods[0], in_nchw=True)
                s = prog.sequential(s, mods[1:])
                first = False
            elif i == 8:
                mods = list(layer)
                if self.fuse_tail:
                    # conv1x1 + BN + ReLU + conv1x1 in one pass over the 16-channel input (csrc/tail.hip)
                    s = prog.tail(s, mods[0], mods[1], mods[3])
                else:
                    s = prog.sequential(s, mods[:-1])
                    s = prog.conv(s, mods[-1], out_nchw=True)   # perspective field leaves as NCHW
            else:
                s = prog.sequential(s, layer)
        return net.Runner(self, prog, trainable=True, precision=self.precision)

    def _forward(self, x, groups=1):
        if self._runner is None:
            net.to_kernel_layout_(self)
            self._runner = self._build()
        return self._runner(x.contiguous(), groups)

    def forward(self, data):
        e1, e2 = self.patch_keys
        p1, p2 = data[e1], data[e2]
        x12 = torch.cat([p1, p2], dim=1)
        if self.variant == 'doubleline':
            # both directions in ONE pass, BatchNorm statistics kept per direction (groups=2)
            B = p1.shape[0]
            out = self._forward(torch.cat([x12, torch.cat([p2, p1], dim=1)], dim=0), groups=2)
            data[self.target_keys[0]] = out[:B]
            data[self.target_keys[1]] = out[B:]
            data['_bh_pf_stacked'] = out
        else:
            data[self.target_keys[0]] = self._forward(x12, groups=1)
        return data

    def predict_homography(self, data):
        return self.forward(data)

    def state_dict(self, *args, **kwargs):
        net.flush_counters(self)
        return super().state_dict(*args, **kwargs)
