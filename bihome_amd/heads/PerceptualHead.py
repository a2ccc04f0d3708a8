"""Drop-in for the reference's `src.heads.PerceptualHead.Model` (the biHomE head) on gfx950 kernels.

Contract: `Model(backbone, **cfg['MODEL']['HEAD'])`; `forward(data) -> (loss, delta_gt,
delta_hat[B,4,2])`; `predict_homography(data) -> (delta_hat[B,4,2], None)`; state-dict keys
`auxiliary_resnet.resnet.{conv1,bn1,layer1.*}` and `backbone.*` as upstream.  Only the branch the shipped
biHomE configs select is built (TRIPLET_LOSS 'double-line', TRIPLET_DISTANCE 'l1', TRIPLET_AGGREGATION
'channel-agnostic', str TRIPLET_MARGIN, SAMPLING_STRATEGY 'downsample-mask', AUXILIARY_RESNET 'resnet34'
layer 1); anything else raises.  Reference: src/heads/PerceptualHead.py:15-767, src/heads/ransac_utils.py:26-161,
src/data/utils.py:7-59.

Data flow of one training forward (both directions stacked along the batch axis, "2B"):
    pf[2B,2,h,w] --bh_dlt_fwd--> delta_hat[2B,4,2] --bh_h4pt_fwd--> H[2B,9]
    patches[2B,1,h,w], H --bh_warp_fwd--> warped[2B,1,h,w] + pooled coverage[2B,h/4,w/4]
    extractor(patches) (no grad), extractor(warped) (dgrad only)  -> NHWC features [2B,h/4,w/4,64]
    bh_triplet_l1_fwd + bh_bihome_loss_fwd -> loss
and the adjoint chain back to pf in `_BiHomELoss.backward` / `_DltFunction.backward`.
"""
import os

import torch
import torch.nn as nn

from .. import kernels as K
from .. import net


# -----------------------------------------------------------------------------------------------
# frozen ResNet-34 stem + layer1 (parameter container with torchvision's names)
# -----------------------------------------------------------------------------------------------
class _BasicBlock(nn.Module):
    def __init__(self, cin, cout=None, stride=1):
        super().__init__()
        cout = cin if cout is None else cout
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))


class _ResNetStem(nn.Module):
    """conv1, bn1, layer1 .. layer<output_layer> of a torchvision resnet34 (the later layers are Identity upstream and own
    no parameters: PerceptualHead.py:26-33)."""

    def __init__(self, output_layer=1):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for i, (n, c) in enumerate(zip((3, 4, 6, 3), (64, 128, 256, 512))):
            if i + 1 > output_layer:
                break
            blocks = []
            for j in range(n):
                blocks.append(_BasicBlock(cin, c, 2 if (j == 0 and i > 0) else 1))
                cin = c
            setattr(self, 'layer%d' % (i + 1), nn.Sequential(*blocks))
        self.out_channels = cin


class AuxiliaryResnet(nn.Module):
    """PerceptualHead.py:15-76 for AUXILIARY_RESNET='resnet34', AUXILIARY_RESNET_OUTPUT_LAYER 1..4 (:24-33,:62-67).
    Output is NHWC [N, h/s, w/s, C] with (s, C) = (4, 64), (8, 128), (16, 256), (32, 512) - the layout the triplet
    kernel reads."""
    keep_warped = False        # True: the loss nodes also write the warped patches (head.last["warped"]) when the stem makes them itself

    def __init__(self, **kwargs):
        super().__init__()
        self.output_layer = int(kwargs.get('AUXILIARY_RESNET_OUTPUT_LAYER', 1))
        if kwargs.get('AUXILIARY_RESNET', 'resnet34') != 'resnet34' or not 1 <= self.output_layer <= 4:
            raise NotImplementedError("only AUXILIARY_RESNET='resnet34' with OUTPUT_LAYER 1..4 is built")
        if kwargs.get('WITH_PROJECTION_HEAD') is not None:
            raise NotImplementedError("WITH_PROJECTION_HEAD is not used by any shipped config")
        self.resnet = _ResNetStem(self.output_layer)
        self.stride = 4 << (self.output_layer - 1)      # feature-map stride = the mask downsample factor (:450)
        self.freeze = kwargs.get('AUXILIARY_RESNET_FREEZE', True)
        if not self.freeze:
            raise NotImplementedError("AUXILIARY_RESNET_FREEZE=False is not used by any shipped config")
        for p in self.resnet.parameters():          # PerceptualHead.py:36-39
            p.requires_grad = False
        self.precision = kwargs.get('PRECISION', os.environ.get('BIHOME_PRECISION', 'f32'))
        net.to_kernel_layout_(self)
        self._runners = {}
        self._fold = {}             # one BatchNorm-folding cache for every program over these modules
        net.install_counter_hooks(self)

    @staticmethod
    def _gray_weight(w):
        # x.repeat(1,3,1,1) then a 3-channel conv (PerceptualHead.py:52-55) == 1-channel conv with summed weights
        return w.sum(dim=1, keepdim=True)

    def _runner(self, in_ch):
        if in_ch not in self._runners:
            r = self.resnet
            prog = net.Program()
            # a 1-channel NCHW image is already NHWC; RGB input is read as NCHW by the first conv
            s = prog.conv(0, r.conv1, in_nchw=(in_ch != 1), weight_fn=self._gray_weight if in_ch == 1 else None)
            s = prog.bn(s, r.bn1, relu=True)
            s = prog.maxpool(s)
            for li in range(1, self.output_layer + 1):
                for blk in getattr(r, 'layer%d' % li):
                    s = prog.basic_block(s, blk)
            self._runners[in_ch] = net.Runner(self, prog, trainable=False, precision=self.precision, fold_cache=self._fold)
        return self._runners[in_ch]

    def forward(self, x, groups=1):
        """x [N,1|3,h,w] NCHW -> NHWC features; BatchNorms follow self.training (batch statistics in
        training although the weights are frozen - SURVEY.md 7), one statistic set per group."""
        net.to_kernel_layout_(self)
        N, C, h, w = x.shape
        x = x.contiguous()
        if C == 1:
            x = x.view(N, h, w, 1)
        return self._runner(C)(x, groups)

    def state_dict(self, *args, **kwargs):
        net.flush_counters(self)
        return super().state_dict(*args, **kwargs)


# -----------------------------------------------------------------------------------------------
# autograd nodes
# -----------------------------------------------------------------------------------------------
@K.scoped_function
class _DltFunction(torch.autograd.Function):
    """pf[N,2,h,w], choice[N,n*P] -> delta_hat[N,n,4,2], Hdlt[N,n,3,3]  (ransac_utils.py:47-74 +
    PerceptualHead.py:125-146,175-178)."""

    @staticmethod
    def forward(ctx, pf, choice, n, P):
        pf = pf.contiguous()
        Hd, dh, eig = K.dlt_fwd(pf, choice, n, P)
        ctx.save_for_backward(pf, choice, eig)
        ctx.n, ctx.P = n, P
        ctx.set_materialize_grads(False)            # an unused output (Hdlt when nothing scores the hypotheses) stays None
        N = pf.shape[0]
        return dh.view(N, n, 4, 2), Hd.view(N, n, 3, 3)

    @staticmethod
    def backward(ctx, g_dh, g_H):
        pf, choice, eig = ctx.saved_tensors
        N = pf.shape[0]
        g = (g_dh.contiguous().view(-1, 4, 2) if g_dh is not None
             else torch.zeros(N * ctx.n, 4, 2, dtype=torch.float32, device=pf.device))
        gH = g_H.reshape(-1, 9).to(torch.float64).contiguous() if g_H is not None else None   # from the hypothesis scores
        return K.dlt_bwd(pf, choice, eig, g, ctx.n, ctx.P, g_H=gH), None, None, None


@K.scoped_function
class _DsacScores(torch.autograd.Function):
    """DSACSoftmax.__score_hypotheses (ransac_utils.py:76-128): scores[N,n] = softmax(-sum_points |H.coord - map|_1), with
    its adjoint w.r.t. the perspective field (every point) and the hypotheses' homographies."""

    @staticmethod
    def forward(ctx, pf, Hd):
        pf = pf.contiguous()
        N, n = Hd.shape[0], Hd.shape[1]
        Hflat = Hd.reshape(N * n, 9).contiguous()
        scores, _ = K.dsac_scores_fwd(pf, Hflat, n)
        ctx.save_for_backward(pf, Hflat, scores)
        ctx.n = n
        return scores

    @staticmethod
    def backward(ctx, g_scores):
        pf, Hflat, scores = ctx.saved_tensors
        g_pf, g_Hd = K.dsac_scores_bwd(pf, Hflat, scores, g_scores.contiguous(), ctx.n)
        return g_pf, g_Hd.to(torch.float32).view(-1, ctx.n, 3, 3)


@K.scoped_function
class _ScaleSamples(torch.autograd.Function):
    """y[b] = x[b // rep] * s[b] (multihead_resnet_loss' score weighting, PerceptualHead.py:276-280)."""

    @staticmethod
    def forward(ctx, x, s, rep):
        x, s = x.contiguous(), s.contiguous()
        ctx.save_for_backward(x, s)
        ctx.rep = rep
        return K.scale_samples_fwd(x, s, rep)

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        want_gx = ctx.needs_input_grad[0]
        if want_gx and ctx.rep != 1:
            raise NotImplementedError("gradient w.r.t. a repeated feature map is not needed by any branch")
        g_x, g_s = K.scale_samples_bwd(g.contiguous(), x, s, ctx.rep, want_gx)
        return g_x, g_s, None


def _extractor_dgrad_into_warp(aux, featw, wl, gfeatw, src, H64, gcov, pool, gH):
    """The adjoint of  H -> extractor(warp(src, H))  given the feature gradient: extractor dgrad (one NetFunction node), then the warp's
    adjoint into gH (+=).  Round 6: where the extractor's stem kernel applies (one-channel patches, default arithmetic) the two are ONE
    launch - the Runner's `input_sink` tells the stem's dgrad what its gradient image is for (kernels.conv_dgrad warp_sink), the image is
    never written and warp_bwd4_kernel never runs; every other case takes the two calls as before."""
    runner = aux._runner(src.shape[1])
    sink = dict(src=src, H64=H64, g_cov=gcov, pool=pool, gH=gH, done=False) if (src.shape[1] == 1 and src.is_contiguous()) else None
    runner.input_sink = sink
    try:
        (gwarp,) = torch.autograd.grad(featw, wl, gfeatw, allow_unused=True)
    finally:
        runner.input_sink = None
    if sink is None or not sink["done"]:
        K.warp_bwd(src, H64, gwarp.contiguous(), gcov, pool, gH=gH)


def _extractor_of_warp(aux, src, H64, pool, groups, want_cov=True):
    """warped, cov, wl, featw = the homography warp of src (PerceptualHead.py:371,392 `_warp` -> src/data/utils.py:54-59), the pooled
    coverage of the warped all-ones mask (:380-382,447-459), and the extractor's features of the warped image as a differentiable function
    of it (:377,398).  Round 6: where the extractor's stem kernel applies (one-channel patches, default arithmetic, pool 4) the warp is made
    INSIDE the stem's forward launch (the Runner's `input_source` -> kernels.conv_fwd warp_src -> bh_stem7_fwd_warp): warp_fwd4_kernel
    never runs, `cov` (and `warped`, when somebody wants to look at it) are written on the way, bitwise what the separate launch writes."""
    B, C, h, w = src.shape
    if C != 1 or not src.is_contiguous():
        warped, cov = K.warp_fwd(src, H64, pool, want_cov=want_cov)
        with torch.enable_grad():
            wl = warped.detach().requires_grad_(True)
            featw = aux(wl, groups=groups)
        return warped, cov, wl, featw
    warped = torch.empty_like(src)
    cov = torch.empty(B, h // pool, w // pool, dtype=torch.float32, device=src.device) if want_cov else None
    runner = aux._runner(1)
    # the warped image itself has no reader on this path (the frozen extractor's stem has no weight gradient; the stem's dgrad does not
    # need its input): the fused kernel writes it only on request (aux.keep_warped, for inspection) - `warped` is then None
    want_image = bool(getattr(aux, "keep_warped", False)) or runner.flat is not None
    source = dict(src=src, H64=H64, pool=pool, cov=cov, want_image=want_image, done=False, filled=False)
    runner.input_source = source
    try:
        with torch.enable_grad():
            wl = warped.detach().requires_grad_(True)
            featw = aux(wl, groups=groups)
    finally:
        runner.input_source = None
    return (warped if (want_image or not source["done"]) else None), cov, wl, featw


@K.scoped_function
class _BiHomELoss(torch.autograd.Function):
    """triplet_resnet_loss, double-line branch (PerceptualHead.py:320-714) for stacked directions.
    patches[2B,1,h,w] = cat(patch_1, patch_2); delta[2B,4,2] = cat(delta_hat_12, delta_hat_21)."""

    @staticmethod
    def forward(ctx, delta, patches, head):
        B2, _, h, w = patches.shape
        B = B2 // 2
        aux = head.auxiliary_resnet
        delta = delta.contiguous()
        ready = getattr(head, "_feat_ready", None)
        if ready is not None:                                   # prefetched on the side stream under the backbone
            feat = ready[0]
            torch.cuda.current_stream().wait_event(ready[1])
        else:
            with torch.no_grad():
                feat = aux(patches, groups=2)                   # :358,:367  (patch_1 then patch_2 statistics)
        H64, H32 = K.h4pt_fwd(delta, h)                          # _warp -> four_point_to_homography :237-243
        pool = aux.stride                                        # :450 downsample_factor = mask size // feature size
        warped, cov, wl, featw = _extractor_of_warp(aux, patches, H64, pool, groups=2)      # :371-382,:392-401,:447-459
        f1, f2, f1w, f2w = feat[:B], feat[B:], featw.detach()[:B], featw.detach()[B:]
        m1w, m2w = cov[:B], cov[B:]
        M1, M2, numden = K.triplet_l1_fwd(f1, f2, f1w, f2w, m1w, m2w)      # :559-561,:609-653
        loss4 = K.bihome_loss_fwd(numden, H64[:B], H64[B:], head.triplet_mu)   # :656-665
        ctx.head, ctx.B, ctx.pool = head, B, pool
        ctx.saved = (delta, patches, H64, feat, featw, wl, cov, M1, M2, numden)
        head.last = {"loss4": loss4, "H_4pt": H32, "warped": warped, "coverage": cov, "numden": numden,
                     "f1": f1, "f2": f2, "f1w": f1w}
        return loss4[0]

    @staticmethod
    def backward(ctx, g_loss):
        delta, patches, H64, feat, featw, wl, cov, M1, M2, numden = ctx.saved
        ctx.saved = None
        B, head = ctx.B, ctx.head
        h = patches.shape[-1]
        g = g_loss.reshape(1).to(torch.float32).contiguous()
        fw = featw.detach()
        gfeatw, gcov, gH = K.bihome_loss_bwd(g, feat[:B], feat[B:], fw[:B], fw[B:], cov[:B], cov[B:], None, None, M1, M2, numden,
                                             H64[:B], H64[B:], head.triplet_mu, joined=True)      # (both directions in one tensor each)
        _extractor_dgrad_into_warp(head.auxiliary_resnet, featw, wl, gfeatw, patches, H64, gcov, ctx.pool, gH)
        gdelta = K.h4pt_bwd(delta, H64, gH, h)
        return gdelta, None, None


@K.scoped_function
class _IHomELoss(torch.autograd.Function):
    """triplet_resnet_loss, one-line branch (iHomE; PerceptualHead.py:320-538): only patch_1 is warped, hinge with a
    numeric margin.  patches[2B,1,h,w] = cat(patch_1, patch_2); delta[B,4,2] = delta_hat_12."""

    @staticmethod
    def forward(ctx, delta, patches, head, scores=None, n=1):
        """delta [B*n,4,2] (n hypotheses per sample, :352-361), scores [B*n] or None (:505-511)."""
        B2, _, h, w = patches.shape
        B = B2 // 2
        aux = head.auxiliary_resnet
        delta = delta.contiguous()
        ready = getattr(head, "_feat_ready", None)
        if ready is not None:
            feat = ready[0]
            torch.cuda.current_stream().wait_event(ready[1])
        else:
            with torch.no_grad():
                # (upstream extracts the features of the n-fold repeated patches: identical batch statistics, so one copy)
                feat = aux(patches, groups=2)                   # :358,:367 (patch_1, then patch_2)
        H64, H32 = K.h4pt_fwd(delta, h)                          # :371 -> four_point_to_homography
        pool = aux.stride
        p1 = patches[:B].contiguous() if n == 1 else patches[:B].repeat_interleave(n, 0)   # :352 one copy per hypothesis
        warped, cov, wl, featw = _extractor_of_warp(aux, p1, H64, pool, groups=1)           # :371-382 + downsample (:447-451)
        sw = scores.contiguous() if scores is not None else None
        loss, T, numden, per = K.oneline_loss_fwd(feat[:B], feat[B:], featw.detach(), cov, head.triplet_margin, rep=n,
                                                  sample_w=sw)    # :474-533
        ctx.head, ctx.pool, ctx.n = head, pool, n
        ctx.saved = (delta, p1, H64, feat[B:], featw, wl, cov, T, numden, sw, per)
        head.last = {"loss4": loss, "H_4pt": H32, "warped": warped, "coverage": cov, "f1": feat[:B], "f2": feat[B:],
                     "f1w": featw.detach()}
        return loss[0]

    @staticmethod
    def backward(ctx, g_loss):
        delta, p1, H64, f2, featw, wl, cov, T, numden, sw, per = ctx.saved
        ctx.saved = None
        h = p1.shape[-1]
        g = g_loss.reshape(1).to(torch.float32).contiguous()
        gfw, gcov = K.oneline_loss_bwd(g, f2, featw.detach(), cov, T, numden, rep=ctx.n, sample_w=sw)
        gH = torch.zeros_like(H64)
        _extractor_dgrad_into_warp(ctx.head.auxiliary_resnet, featw, wl, gfw, p1, H64, gcov, ctx.pool, gH)
        gdelta = K.h4pt_bwd(delta, H64, gH, h)
        g_scores = per * g if sw is not None else None           # d loss / d score_b = loss_b
        return gdelta, None, None, g_scores, None


@K.scoped_function
class _WarpFeatures(torch.autograd.Function):
    """multihead_resnet_loss (PerceptualHead.py:245-315): features of the warped patch_1 as a differentiable function
    of delta_hat.  p1[B,1,h,w], delta[B,4,2] -> NHWC features [B,h/s,w/s,C]."""

    @staticmethod
    def forward(ctx, delta, p1, head):
        aux = head.auxiliary_resnet
        delta = delta.contiguous()
        h = p1.shape[-1]
        H64, H32 = K.h4pt_fwd(delta, h)                          # _warp :237-243
        warped, _, wl, featw = _extractor_of_warp(aux, p1, H64, aux.stride, groups=1, want_cov=False)      # :272-273
        ctx.saved = (delta, p1, H64, featw, wl)
        ctx.pool, ctx.aux = aux.stride, aux
        head.last = {"H_4pt": H32, "warped": warped}
        return featw.detach()

    @staticmethod
    def backward(ctx, g):
        delta, p1, H64, featw, wl = ctx.saved
        ctx.saved = None
        gH = torch.zeros_like(H64)
        _extractor_dgrad_into_warp(ctx.aux, featw, wl, g.contiguous(), p1, H64, None, ctx.pool, gH)
        return K.h4pt_bwd(delta, H64, gH, p1.shape[-1]), None, None


def _tb_scalars(data, groups):
    """The head's TensorBoard side channel (PerceptualHead.py:286-298,678-697): only when the driver injected
    data['summary_writer'] (log steps, train.py:312-314).  Each value is a device sync, as upstream."""
    sw, step = data['summary_writer'], data['summary_writer_step']
    for tag, key, val in groups:
        sw.add_scalars(tag, {key: float(val)}, step)


# -----------------------------------------------------------------------------------------------
@K.scoped_module
class Model(nn.Module):

    def __init__(self, backbone, **kwargs):
        super().__init__()
        self.backbone = backbone
        self.patch_size = kwargs['PATCH_SIZE']
        self.patch_keys = kwargs['PATCH_KEYS']
        self.delta_hat_keys = kwargs['DELTA_HAT_KEYS']
        if len(self.delta_hat_keys):
            self.hypothesis_no = 1
        else:
            self.pf_keys = kwargs['PF_KEYS']
            self.hypothesis_no = kwargs['RANSAC_HYPOTHESIS_NO']
            self.point_per_hypothesis = kwargs['POINTS_PER_HYPOTHESIS']
            if kwargs.get('SCORING_METHOD', 'repr_error') != 'repr_error':
                raise NotImplementedError("only SCORING_METHOD='repr_error' is built")
        self.triplet_version = kwargs['TRIPLET_LOSS']
        self.multihead = self.triplet_version == ''           # PerceptualHead.py:108,:230-235 -> multihead_resnet_loss
        common = ('dual' not in self.triplet_version and kwargs.get('TRIPLET_DISTANCE') == 'l1'
                  and not len(kwargs.get('MASK_KEYS', [])) and 'upsample' not in str(kwargs.get('SAMPLING_STRATEGY', ''))
                  and not kwargs.get('MASK_CRD', False))
        self.one_line = 'one-line' in self.triplet_version
        if self.multihead:                 # features out, a torch loss is applied by the driver (train.py:318-322)
            ok = True
        elif self.one_line:                # iHomE: PerceptualHead.py:465-538, hinge with a numeric margin
            ok = common and isinstance(kwargs.get('TRIPLET_MARGIN'), (int, float))
        else:                              # biHomE: PerceptualHead.py:540-665
            ok = (common and 'double-line' in self.triplet_version and isinstance(kwargs.get('TRIPLET_MARGIN'), str)
                  and kwargs.get('TRIPLET_AGGREGATION') == 'channel-agnostic')
        if not ok:
            raise NotImplementedError("built: biHomE (double-line / l1 / channel-agnostic / str margin), iHomE (one-line / l1 / "
                                      "numeric margin) and the multihead feature loss (TRIPLET_LOSS ''), no MASK_KEYS / MASK_CRD, downsample-mask - see SURVEY.md 2 for what is out of "
                                      "scope")
        self.triplet_mu = kwargs.get('TRIPLET_MU', 0.0)
        self.triplet_margin = kwargs.get('TRIPLET_MARGIN')
        self.auxiliary_resnet = AuxiliaryResnet(**kwargs)
        self.last = {}
        # The features of the two unwarped patches depend on the batch only, not on the backbone: their extractor pass is
        # enqueued on the side HIP stream when the backbone's forward starts and runs under the backbone's kernels; the
        # loss picks the result up behind an event.  (The reference runs the same extractor calls, in the same order,
        # after the backbone: PerceptualHead.py:358,367.)
        self._prefetched = None
        self.prefetch_features = os.environ.get("BIHOME_OVERLAP", "1") != "0"      # default on, see net.Runner
        if isinstance(backbone, nn.Module):
            backbone.register_forward_pre_hook(self._prefetch_hook)

    def _stack_patches(self, data):
        e1, e2 = self.patch_keys
        p1, p2 = data[e1], data[e2]
        B = p1.shape[0]
        return torch.cat([p1.reshape(B, -1, self.patch_size, self.patch_size),
                          p2.reshape(B, -1, self.patch_size, self.patch_size)], 0)

    def _prefetch_hook(self, module, args):
        self._prefetched = None
        if not (self.prefetch_features and self.training and torch.is_grad_enabled() and args and isinstance(args[0], dict)):
            return
        data = args[0]
        if self.patch_keys[0] not in data or not data[self.patch_keys[0]].is_cuda:
            return
        patches = self._stack_patches(data)
        main = torch.cuda.current_stream()
        side = net.side_stream(patches.device)
        side.wait_stream(main)
        with torch.cuda.stream(side), torch.no_grad():
            feat = self.auxiliary_resnet(patches, groups=2)     # :358,:367 (patch_1 then patch_2 statistics)
            ev = torch.cuda.Event()
            ev.record(side)
        patches.record_stream(side)
        feat.record_stream(main)
        self._prefetched = (data[self.patch_keys[0]], data[self.patch_keys[1]], patches, feat, ev)

    # ---- DSAC -----------------------------------------------------------------------------------------
    @staticmethod
    def sample_choice(n_points, count, device):
        """ransac_utils.py:54-56: torch.multinomial with weights arange(N) (p(i) ~ i; a reference quirk kept)."""
        w = torch.arange(0, n_points, dtype=torch.float32, device=device)
        return torch.multinomial(w, count, replacement=True)

    def _choices(self, data, key, B, N, device):
        if key in data:                                         # test hook: indices supplied (bit-exact pin)
            c = data[key].to(device=device, dtype=torch.int64).reshape(B, -1).contiguous()
            # the DLT kernels gather pf[id] and scatter-add into g_pf[id]: caller-supplied indices must lie in [0, N).
            # (Host check with a device sync: this hook is the test / replay path; indices drawn by `sample_choice` are in
            # range by construction.  A device-side assert would abort the process on ROCm instead of raising.)
            capturing = torch.cuda.is_current_stream_capturing()     # (no host sync inside a HIP-graph capture)
            if c.numel() and not capturing and not bool(((c >= 0) & (c < N)).all()):
                raise ValueError("bihome_amd: data['%s'] holds indices outside [0, %d)" % (key, N))
            return c
        n, P = self.hypothesis_no, self.point_per_hypothesis
        return self.sample_choice(N, B * P * n, device).reshape(B, -1)

    def _stacked_pf(self, data):
        pf12, pf21 = data[self.pf_keys[0]], data[self.pf_keys[1]]
        st = data.get('_bh_pf_stacked')
        B = pf12.shape[0]
        if (st is not None and st.shape[0] == 2 * B and pf12.data_ptr() == st.data_ptr()
                and pf21.data_ptr() == st[B:].data_ptr()):
            return st                                           # produced stacked by this build's backbone
        return torch.cat([pf12, pf21], 0)

    def forward(self, data):
        e1, e2 = self.patch_keys
        p1, p2 = data[e1], data[e2]
        if not p1.is_cuda:
            raise RuntimeError("bihome_amd heads run on the MI355X only; no CPU fallback (use oracle/ for CPU checks)")
        B = p1.shape[0]
        if self.multihead:
            return self._forward_multihead(data, p1, p2, B)
        if self.one_line:
            return self._forward_one_line(data, p1, p2, B)
        if not len(self.delta_hat_keys):
            pf = self._stacked_pf(data)
            N = pf.shape[-1] * pf.shape[-2]
            c12 = self._choices(data, 'choice_12', B, N, pf.device)
            c21 = self._choices(data, 'choice_21', B, N, pf.device)
            dh, Hd = _DltFunction.apply(pf, torch.cat([c12, c21], 0), self.hypothesis_no, self.point_per_hypothesis)
            if self.hypothesis_no != 1:
                # (upstream's double-line branch cannot run with n > 1 either: `eye` is built for B samples, h1 h2 for B*n,
                #  PerceptualHead.py:660-662; the one-line and multihead branches do take several hypotheses - built below)
                raise NotImplementedError("double-line training with RANSAC_HYPOTHESIS_NO > 1 is not defined upstream")
            delta = dh.reshape(2 * B, 4, 2)
            self.last_dlt = Hd
        else:
            delta = torch.cat([data[self.delta_hat_keys[0]].reshape(B, 4, 2), data[self.delta_hat_keys[1]].reshape(B, 4, 2)], 0)
        pre, self._prefetched = self._prefetched, None
        if pre is not None and pre[0] is p1 and pre[1] is p2:
            patches, self._feat_ready = pre[2], (pre[3], pre[4])
        else:
            patches, self._feat_ready = self._stack_patches(data), None
        loss = _BiHomELoss.apply(delta, patches, self)
        self._feat_ready = None
        if 'summary_writer' in data:                            # PerceptualHead.py:678-697 (syncs; log steps only)
            f1, f2, f1w = self.last["f1"], self.last["f2"], self.last["f1w"]
            nd = self.last["numden"]                            # [B, (num1, den1, num2, den2)]
            eye = torch.eye(3, device=p1.device, dtype=torch.float32)
            _tb_scalars(data, [('feature_space', 'patch_1_f', f1.mean()), ('feature_space', 'patch_2_f', f2.mean()),
                               ('feature_space', 'patch_1_f_prime', f1w.mean()),
                               ('loss_comp', 'l1', (f2 - f1w).abs().mean()), ('loss_comp', 'l3', (f2 - f1).abs().mean()),
                               ('h', 'h1', ((self.last["H_4pt"][:B] - eye) ** 2).sum()),
                               ('loss_den', 'l1_den', nd[:, 1].min()), ('loss_den', 'l2_den', nd[:, 3].min())])
        delta_gt = data['delta'] if 'delta' in data else None
        return loss, delta_gt, delta[:B]

    def _delta_12(self, data, B):
        """(delta_hat_12 [B*n,4,2], scores [B*n] | None): n hypotheses per sample from the DLT on pf_hat_12 with their
        softmax(-reprojection error) scores (:154-178, ransac_utils.py:147-161), or the backbone's own delta_hat (:211-214).
        A single hypothesis has score exactly 1 (and no gradient through it): scores = None."""
        if not len(self.delta_hat_keys):
            pf = data[self.pf_keys[0]].contiguous()
            N = pf.shape[-1] * pf.shape[-2]
            n = self.hypothesis_no
            c12 = self._choices(data, 'choice_12', B, N, pf.device)
            dh, Hd = _DltFunction.apply(pf, c12, n, self.point_per_hypothesis)
            self.last_dlt = Hd
            scores = _DsacScores.apply(pf, Hd).reshape(B * n) if n > 1 else None
            return dh.reshape(B * n, 4, 2), scores
        return data[self.delta_hat_keys[0]].reshape(B, 4, 2), None

    def _expected_delta(self, delta, scores, B):
        """:309-312,:708-710: the returned delta_hat is the score-weighted mean over the hypotheses."""
        if scores is None:
            return delta
        n = self.hypothesis_no
        return (delta.reshape(B, n, 4, 2) * scores.reshape(B, n, 1, 1)).sum(1)

    def _forward_multihead(self, data, p1, p2, B):
        """multihead_resnet_loss (PerceptualHead.py:245-315): returns (features of patch_2, features of the warped patch_1,
        delta_gt, delta_hat) - ground truth first, the driver applies its torch loss (train.py:318-322).  The feature
        tensors are handed over in the reference's NCHW shape (views of the kernels' NHWC maps)."""
        delta, scores = self._delta_12(data, B)
        P = self.patch_size
        n = 1 if scores is None else self.hypothesis_no
        aux = self.auxiliary_resnet
        f2 = aux(p2.reshape(B, -1, P, P), groups=1).detach()      # :269 (frozen weights: no gradient path)
        p1r = p1.reshape(B, -1, P, P).contiguous()
        f1w = _WarpFeatures.apply(delta, p1r if n == 1 else p1r.repeat_interleave(n, 0), self)     # :265,:272-273
        if scores is not None:                                  # :276-280 both feature maps times the hypothesis score
            f1w = _ScaleSamples.apply(f1w, scores, 1)
            f2 = _ScaleSamples.apply(f2, scores, n)
        if 'summary_writer' in data:                            # :286-298
            eye = torch.eye(3, device=p1.device, dtype=torch.float32)
            _tb_scalars(data, [('feature_space', 'patch_2_f', f2.mean()), ('feature_space', 'patch_1_f_prime', f1w.mean()),
                               ('loss_comp', 'l1', (f2 - f1w).abs().mean()),
                               ('h', 'h1', ((self.last["H_4pt"] - eye) ** 2).sum())])
        delta_gt = data['delta'] if 'delta' in data else None
        return f2.permute(0, 3, 1, 2), f1w.permute(0, 3, 1, 2), delta_gt, self._expected_delta(delta, scores, B)

    def _forward_one_line(self, data, p1, p2, B):
        """One direction only (PerceptualHead.py:154-176,222-223): delta_hat_12 from the DLT on pf_hat_12 (or given)."""
        delta, scores = self._delta_12(data, B)
        pre, self._prefetched = self._prefetched, None
        if pre is not None and pre[0] is p1 and pre[1] is p2:
            patches, self._feat_ready = pre[2], (pre[3], pre[4])
        else:
            patches, self._feat_ready = self._stack_patches(data), None
        loss = _IHomELoss.apply(delta, patches, self, scores, 1 if scores is None else self.hypothesis_no)
        self._feat_ready = None
        if 'summary_writer' in data:                            # :678-692
            f1, f2, f1w = self.last["f1"], self.last["f2"], self.last["f1w"]
            if scores is not None:                              # (upstream logs the n-fold repeated maps)
                f1, f2 = (t.repeat_interleave(self.hypothesis_no, 0) for t in (f1, f2))
            eye = torch.eye(3, device=p1.device, dtype=torch.float32)
            _tb_scalars(data, [('feature_space', 'patch_1_f', f1.mean()), ('feature_space', 'patch_2_f', f2.mean()),
                               ('feature_space', 'patch_1_f_prime', f1w.mean()),
                               ('loss_comp', 'l1', (f2 - f1w).abs().mean()), ('loss_comp', 'l3', (f2 - f1).abs().mean()),
                               ('h', 'h1', ((self.last["H_4pt"] - eye) ** 2).sum())])
        delta_gt = data['delta'] if 'delta' in data else None
        return loss, delta_gt, self._expected_delta(delta, scores, B)

    def predict_homography(self, data):
        if len(self.delta_hat_keys):
            return data[self.delta_hat_keys[0]], None
        pf = data[self.pf_keys[0]].contiguous()
        B, _, h, w = pf.shape
        n, P = self.hypothesis_no, self.point_per_hypothesis
        choice = self._choices(data, 'choice', B, h * w, pf.device)
        Hd, dh, _ = K.dlt_fwd(pf, choice, n, P)
        if n == 1:
            return dh.view(B, 4, 2), None
        err, best = K.dsac_score(pf, Hd.view(-1, 9), n)         # argmax softmax(-err) == argmin err (:755-757)
        self.last.update(best=best, repr_error=err)
        return dh.view(B, n, 4, 2)[torch.arange(B, device=pf.device), best], None

    def state_dict(self, *args, **kwargs):
        net.flush_counters(self)
        return super().state_dict(*args, **kwargs)
