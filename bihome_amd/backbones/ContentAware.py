"""Drop-in for the reference's `src.backbones.ContentAware.Model` (Zhang et al. "Content-Aware Unsupervised Deep Homography
Estimation" baseline, src/backbones/ContentAware.py:84-192): a mask predictor and a 1 -> 4 -> 8 -> 1-channel feature extractor applied
to each patch, G = mask * features, and a torchvision-layout resnet34 (2-channel conv1, 8-way fc) on cat(G1, G2) - executed by the
gfx950 kernels.  State-dict keys `mask_predictor.layerN.*`, `feature_extractor.layerN.*`, `resnet34.*` as upstream.

Every shipped zhang-* config sets FIX_MASK: True - the mask is all ones and the mask predictor never runs.  FIX_MASK: False (round 4):
the predictor's five Conv3x3 + BatchNorm layers run on the conv executor (statistics per call), csrc/mask.hip applies the Sigmoid, the
per-sample max normalisation (MASK_NORMALIZATION_STRENGTH > 0) and G = mask * features in one launch, and its adjoint takes the
mask gradients the TripletHead sends back (direct and through the warp, csrc/warp.hip bh_warp_bwd_img_f).  The last BatchNorm of the
predictor and of the extractor has ONE channel (csrc/bn1.hip); the tiny-channel convolutions run on the generic implicit-GEMM kernels."""
import os

import torch
import torch.nn as nn

from .. import kernels as K
from .. import net
from .ResNet34 import _ResNet34


def _cbr(cin, cout, act=nn.ReLU):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1, bias=False), nn.BatchNorm2d(cout), act())


@K.scoped_function
class _MaskGate(torch.autograd.Function):
    """(y, f) -> (m, g): m = normalise(sigmoid(y)), g = m * f (ContentAware.py:24-26,28-35,128-134) - bh_mask_fwd / bh_mask_bwd.  f None:
    the mask alone.  The gradient of m comes from the head (TripletHead: the masks weight the loss), the gradient of g from the resnet."""

    @staticmethod
    def forward(ctx, y, f, strength):
        y = y.contiguous()
        f = f.contiguous() if f is not None else None
        m, g, smax, imax = K.mask_fwd(y, f, strength)
        ctx.saved = (y, f.detach() if f is not None else None, m, smax, imax)
        ctx.strength, ctx.want_gf = strength, f is not None and f.requires_grad
        ctx.set_materialize_grads(False)
        return (m, g) if f is not None else m

    @staticmethod
    def backward(ctx, g_m, g_g=None):
        y, f, m, smax, imax = ctx.saved
        ctx.saved = None
        g_m = g_m.contiguous() if g_m is not None else None
        g_g = g_g.contiguous() if g_g is not None else None
        g_y, g_f = K.mask_bwd(y, f, m, smax, imax, g_m, g_g, ctx.strength, want_gf=ctx.want_gf)
        return g_y, g_f, None


class MaskPredictor(nn.Module):
    """ContentAware.py:6-52.  fix_mask: ones_like (the parameters exist for checkpoint compatibility).  Trained: five Conv3x3 + BatchNorm
    layers (ReLU after the first four) on the conv executor up to the last BatchNorm, then _MaskGate.  `groups` stacks calls along the
    batch axis and keeps their BatchNorm statistics apart (the backbone calls the predictor once per patch, :127,:133)."""

    def __init__(self, fix_mask=False, normalization_strength=-1, precision="f32"):
        super().__init__()
        self.fix_mask, self.normalization_strength = fix_mask, normalization_strength
        self.layer1, self.layer2, self.layer3, self.layer4 = _cbr(1, 4), _cbr(4, 8), _cbr(8, 16), _cbr(16, 32)
        self.layer5 = _cbr(32, 1, nn.Sigmoid)
        self.precision = precision
        self._runner = None

    def _build(self):
        prog = net.Program()
        s = prog.bn(prog.conv(0, self.layer1[0], in_nchw=True), self.layer1[1], relu=True)
        for layer in (self.layer2, self.layer3, self.layer4):
            s = prog.bn(prog.conv(s, layer[0]), layer[1], relu=True)
        s = prog.bn(prog.conv(s, self.layer5[0], out_nchw=True), self.layer5[1], relu=False)     # (the Sigmoid is _MaskGate's)
        return net.Runner(self, prog, trainable=True, precision=self.precision)

    def pre_sigmoid(self, x, groups=1):
        if self._runner is None:
            net.to_kernel_layout_(self)
            self._runner = self._build()
        return self._runner(x.contiguous(), groups).reshape(x.shape)

    def forward(self, x):
        if self.fix_mask:
            return torch.ones_like(x)                                         # :38-39
        return _MaskGate.apply(self.pre_sigmoid(x), None, float(self.normalization_strength))    # :41-50


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
        self.mask_predictor = MaskPredictor(fix_mask=kwargs['FIX_MASK'], normalization_strength=strength, precision=self.precision)   # :93-94
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
        x = torch.cat([p1, p2], 0)
        f = self.feature_extractor(x, groups=2)
        f1, f2 = f[:B], f[B:]
        if self.mask_predictor.fix_mask:
            m1, m2 = self.mask_predictor(p1), self.mask_predictor(p2)
            return m1, f1, m2, f2, f1, f2                                      # (FIX_MASK: g = 1 * f)
        y = self.mask_predictor.pre_sigmoid(x, groups=2)                       # :127,:133 - one call per patch: statistics per call
        m, g = _MaskGate.apply(y, f, float(self.mask_predictor.normalization_strength))
        return m[:B], f1, m[B:], f2, g[:B], g[B:]

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
