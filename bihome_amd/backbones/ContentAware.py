"""Drop-in for the reference's `src.backbones.ContentAware.Model` (Zhang et al. "Content-Aware Unsupervised Deep Homography
Estimation" baseline, src/backbones/ContentAware.py:84-192): a mask predictor and a 1 -> 4 -> 8 -> 1-channel feature extractor applied
to each patch, G = mask * features, and a torchvision-layout resnet34 (2-channel conv1, 8-way fc) on cat(G1, G2) - executed by the
gfx950 kernels.  State-dict keys `mask_predictor.layerN.*`, `feature_extractor.layerN.*`, `resnet34.*` as upstream.

Scope: every shipped zhang-* config sets FIX_MASK: True - the mask is all ones and the mask predictor never runs
(its parameters exist for checkpoint compatibility); FIX_MASK: False raises NotImplementedError (it needs a Sigmoid layer, the
per-sample max normalisation and the warp adjoint w.r.t. the image).  The feature extractor's last BatchNorm has ONE channel
(csrc/bn1.hip); its tiny-channel convolutions run on the generic implicit-GEMM kernels."""
import os

import torch
import torch.nn as nn

from .. import net
from .ResNet34 import _ResNet34


def _cbr(cin, cout, act=nn.ReLU):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1, bias=False), nn.BatchNorm2d(cout), act())


class MaskPredictor(nn.Module):
    """ContentAware.py:6-52 (parameter container; with fix_mask the forward is ones_like)."""

    def __init__(self, fix_mask=False, normalization_strength=-1):
        super().__init__()
        self.fix_mask, self.normalization_strength = fix_mask, normalization_strength
        self.layer1, self.layer2, self.layer3, self.layer4 = _cbr(1, 4), _cbr(4, 8), _cbr(8, 16), _cbr(16, 32)
        self.layer5 = _cbr(32, 1, nn.Sigmoid)

    def forward(self, x):
        if not self.fix_mask:
            raise NotImplementedError("bihome_amd ContentAware: FIX_MASK False (a trained mask predictor) is not built - every shipped "
                                      "zhang-* config fixes the mask to ones (DESIGN.md 7)")
        return torch.ones_like(x)                                             # :38-39


class FeatureExtractor(nn.Module):
    """ContentAware.py:55-81: three Conv3x3 + BatchNorm + ReLU layers, 1 -> 4 -> 8 -> 1 channels, full resolution.  Called on the two
    patches (backbone) and on the two warped patches (TripletHead.py:59,68) - each call with its own BatchNorm batch statistics:
    `groups` stacks calls along the batch axis and keeps their statistics apart."""

    def __init__(self, precision="f32"):
        super().__init__()
        self.layer1, self.layer2, self.layer3 = _cbr(1, 4), _cbr(4, 8), _cbr(8, 1)
        self.precision = precision
        self._runner = None

    def _build(self):
        prog = net.Program()
        s = prog.bn(prog.conv(0, self.layer1[0], in_nchw=True), self.layer1[1], relu=True)
        s = prog.bn(prog.conv(s, self.layer2[0]), self.layer2[1], relu=True)
        # (one output channel: the NCHW tensor [N,1,h,w] is also its NHWC form)
        s = prog.bn(prog.conv(s, self.layer3[0], out_nchw=True), self.layer3[1], relu=True)
        return net.Runner(self, prog, trainable=True, precision=self.precision)

    def forward(self, x, groups=1):
        if self._runner is None:
            net.to_kernel_layout_(self)
            self._runner = self._build()
        return self._runner(x.contiguous(), groups).reshape(x.shape)

    def retrieve_weights(self):                                               # :76-80
        return {name: p.data for name, p in self.named_parameters()}


class Model(nn.Module):

    def __init__(self, **kwargs):
        super().__init__()
        self.patch_keys = kwargs['PATCH_KEYS']
        self.mask_keys = kwargs['MASK_KEYS']
        self.feature_keys = kwargs['FEATURE_KEYS']
        self.target_keys = kwargs['TARGET_KEYS']
        pre = kwargs.get('PRETRAINED_RESNET')
        if pre is True:
            raise RuntimeError("PRETRAINED_RESNET=True downloads the torchvision ImageNet checkpoint upstream (ContentAware.py:106); "
                               "no network here - pass the path of resnet34-333f7ec4.pth as PRETRAINED_RESNET instead")
        self.precision = kwargs.get('PRECISION', os.environ.get('BIHOME_PRECISION', 'f32'))
        strength = kwargs['MASK_NORMALIZATION_STRENGTH'] if 'MASK_NORMALIZATION_STRENGTH' in kwargs else -1
        self.mask_predictor = MaskPredictor(fix_mask=kwargs['FIX_MASK'], normalization_strength=strength)       # :93-94
        self.feature_extractor = FeatureExtractor(self.precision)
        self.variant = str.lower(kwargs['VARIANT'])
        assert 'oneline' in self.variant or 'doubleline' in self.variant, 'Only OneLine or DoubleLine variant is supported'
        # Upstream (ContentAware.py:101-114) calls init() BEFORE it builds the pretrained resnet, so on the pretrained path the replaced
        # 2-channel conv1 and the fc keep torch's default initialisation (kaiming_uniform, a = sqrt(5)) and only the mask predictor / feature
        # extractor get kaiming_normal; without a checkpoint init() runs over everything.
        if isinstance(pre, str) and pre:
            self.init()
            self.resnet34 = _ResNet34(2, 8)                                    # :106-110 (default init: conv1 / fc as upstream)
        else:
            self.resnet34 = _ResNet34(2, 8)
            self.init()                                                        # :113-114 (kaiming / ones / zeros)
        net.to_kernel_layout_(self)
        if isinstance(pre, str) and pre:
            from ..weights import load_imagenet_resnet34
            load_imagenet_resnet34(self, pre)
            net.to_kernel_layout_(self)
        self._runner = None
        net.install_counter_hooks(self)

    def init(self):                                                            # :116-122
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _build(self):
        r = self.resnet34
        prog = net.Program()
        s = prog.conv(0, r.conv1, in_nchw=True)
        s = prog.maxpool(prog.bn(s, r.bn1, relu=True))
        for name in ('layer1', 'layer2', 'layer3', 'layer4'):
            for blk in getattr(r, name):
                s = prog.basic_block(s, blk)
        s = prog.conv(prog.gap(s), r.fc)
        return net.Runner(r, prog, trainable=True, precision=self.precision)

    def _resnet(self, g, groups):
        if self._runner is None:
            net.to_kernel_layout_(self)
            self._runner = self._build()
        return self._runner(g.contiguous(), groups).reshape(-1, 4, 2)

    def _features(self, p1, p2):
        """m1, f1, m2, f2, g1, g2 of ContentAware.py:126-135; the two extractor calls are stacked (statistics per call)."""
        B = p1.shape[0]
        f = self.feature_extractor(torch.cat([p1, p2], 0), groups=2)
        f1, f2 = f[:B], f[B:]
        m1, m2 = self.mask_predictor(p1), self.mask_predictor(p2)
        return m1, f1, m2, f2, f1, f2                                          # (FIX_MASK: g = 1 * f)

    def forward(self, data):                                                   # :146-173
        e1, e2 = self.patch_keys
        m1k, m2k = self.mask_keys
        f1k, f2k = self.feature_keys
        p1, p2 = data[e1], data[e2]
        B = p1.shape[0]
        data[m1k], data[f1k], data[m2k], data[f2k], g1, g2 = self._features(p1, p2)
        g12 = torch.cat([g1, g2], 1)
        if self.variant == 'doubleline':
            # main pass and auxiliary pass (g2, g1) in ONE resnet pass, BatchNorm statistics per pass (groups = 2)
            o = self._resnet(torch.cat([g12, torch.cat([g2, g1], 1)], 0), groups=2)
            data[self.target_keys[0]], data[self.target_keys[1]] = o[:B], o[B:]
        else:
            data[self.target_keys[0]] = self._resnet(g12, groups=1)
        return data

    def predict_homography(self, data):                                        # :175-187
        e1, e2 = self.patch_keys
        data[self.mask_keys[0]], _, data[self.mask_keys[1]], _, g1, g2 = self._features(data[e1], data[e2])
        data[self.target_keys[0]] = self._resnet(torch.cat([g1, g2], 1), groups=1)
        return data

    def retrieve_weights(self):                                                # :189-192
        return {name: p.data for name, p in self.resnet34.named_parameters()}

    def state_dict(self, *args, **kwargs):
        net.flush_counters(self)
        return super().state_dict(*args, **kwargs)
