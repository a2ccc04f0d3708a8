"""Drop-in for the reference's `src.heads.TripletHead.Model` (the triplet loss of Zhang et al., src/heads/TripletHead.py:9-212) on top
of the ContentAware backbone: warp both patches and both (all-ones) masks with the regressed 4-point offsets, run the backbone's
trainable feature extractor on the warped patches, and reduce

    ln1 = sum_b sum_p m1' m2 h(|f1' - f2| - |f1 - f2| + margin) / max(sum_p m1' m2, 1),   ln2 mirrored,   ln3 = sum ||H1 H2 - I||_F^2
    loss = ln1 + ln2 + mu ln3      (OneLine: ln1)

with the HIP kernels: bh_h4pt_fwd/bwd, bh_warp_fwd/bwd (pool = 1: the warped ones-mask IS the pooled coverage), the conv stack executor
for the extractor, bh_zhang_triplet_fwd/bwd and bh_bihome_loss_fwd.  Trained masks (FIX_MASK False, round 4): each mask rides through the
warp as a second channel next to its patch, the triplet adjoint also returns the gradients of the unwarped masks, and the warp's
adjoint w.r.t. the image (bh_warp_bwd_img_f) takes the warped masks' gradients back to the mask predictor.

Reference quirk kept (results must equal the reference's): with a NUMERIC margin and 'channel-agnostic' aggregation - the shipped
zhang-orig config - `torch.max(sum_c l1 - sum_c l3 + margin, zeros_like(l1))` (:104-105,:141-142) broadcasts [B,h,w] against [B,1,h,w]
to [B,B,h,w], and the sums that follow count every sample B times: ln1 and ln2 carry a factor B.  Reproduced as that factor."""
import torch
import torch.nn as nn

from .. import kernels as K
from .. import net


@K.scoped_function
class _ZhangTripletLoss(torch.autograd.Function):
    """delta[2B | B,4,2] = cat(delta_hat_12, delta_hat_21) (one line: delta_hat_12), patches[2B,1,h,w] = cat(patch_1, patch_2),
    feat[2B,1,h,w] = the backbone's features of the unwarped patches (they carry gradients: the extractor is trainable),
    masks[2B,1,h,w] = cat(mask_1, mask_2) of a trained mask predictor, or None (FIX_MASK: all ones)."""

    @staticmethod
    def forward(ctx, delta, patches, feat, masks, head):
        B2, _, h, w = patches.shape
        B = B2 // 2
        double = head.variant == 'doubleline'
        fe = head.backbone.feature_extractor
        delta = delta.contiguous()
        H64, H32 = K.h4pt_fwd(delta, h)                                       # _warp :30-35 -> four_point_to_homography
        src = patches if double else patches[:B].contiguous()
        m1 = m2 = None
        if masks is None:
            warped, cov = K.warp_fwd(src, H64, 1)                             # :58,:60 (patch and ones-mask), :67,:69
        else:
            md = masks.detach()
            m1, m2 = (md[:B].reshape(B, h, w) if double else None), md[B:].reshape(B, h, w)
            src = torch.stack([src[:, 0], (md if double else md[:B])[:, 0]], 1).contiguous()      # [nB, 2, h, w]: patch, mask
            both, _ = K.warp_fwd(src, H64, 1, want_cov=False)                 # :58,:60,:67,:69 with the predicted masks
            warped, cov = both[:, 0:1].contiguous(), both[:, 1].contiguous()
        with torch.enable_grad():
            wl = warped.detach().requires_grad_(True)
            featw = fe(wl, groups=2 if double else 1)                         # :59,:68: one call per warped patch
        fd = feat.detach()
        f1, f2 = fd[:B], fd[B:]
        fw = featw.detach()
        f1w, f2w = fw[:B], (fw[B:] if double else None)
        m1w, m2w = cov[:B], (cov[B:] if double else None)
        hinge = not isinstance(head.triplet_margin, str)
        margin = float(head.triplet_margin) if hinge else 0.0
        T1, T2, numden = K.zhang_triplet_fwd(f1, f2, f1w, f2w, m1w, m2w, margin, hinge, m1=m1, m2=m2)
        eye = torch.eye(3, dtype=torch.float64, device=delta.device).reshape(1, 9).expand(B, 9).contiguous()
        loss4 = K.bihome_loss_fwd(numden, H64[:B], H64[B:] if double else eye, head.mu if double else 0.0)     # {loss, ln1, ln2, ln3}
        # the reference's broadcast (module docstring): a factor B on ln1 / ln2 for numeric margin + channel-agnostic
        rep = float(B) if (hinge and head.triplet_channel_aggregation == 'channel-agnostic') else 1.0
        ln1, ln2, ln3 = loss4[1] * rep, loss4[2] * rep, loss4[3]
        loss = ln1 + ln2 + head.mu * ln3 if double else ln1
        ctx.head, ctx.B, ctx.double, ctx.hinge, ctx.rep = head, B, double, hinge, rep
        ctx.saved = (delta, src, H64, fd, featw, wl, cov, T1, T2, numden, m1, m2)
        ctx.trained_masks = masks is not None
        head.last = {"ln1": ln1.detach(), "ln2": ln2.detach(), "ln3": ln3.detach(), "H_4pt": H32, "warped": warped, "coverage": cov,
                     "f1": f1, "f2": f2, "f1w": f1w, "f2w": f2w}
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        delta, src, H64, fd, featw, wl, cov, T1, T2, numden, m1, m2 = ctx.saved
        ctx.saved = None
        B, head, double = ctx.B, ctx.head, ctx.double
        h = src.shape[-1]
        g = (g_loss.reshape(1).to(torch.float32) * ctx.rep).contiguous()
        fw = featw.detach()
        grads = K.zhang_triplet_bwd(g, fd[:B], fd[B:], fw[:B], fw[B:] if double else None, cov[:B], cov[B:] if double else None, T1, T2,
                                    numden, ctx.hinge, m1=m1, m2=m2, mask_grads=ctx.trained_masks)
        g_f1, g_f2, g_f1w, g_f2w, g_m1w, g_m2w = grads[:6]
        gfeatw = torch.cat([g_f1w, g_f2w], 0) if double else g_f1w
        (gwarp,) = torch.autograd.grad(featw, wl, gfeatw)                     # extractor backward: weight gradients + d / d warped patch
        gcov = torch.cat([g_m1w, g_m2w], 0) if double else g_m1w
        gH = torch.zeros_like(H64)
        if double:
            # ln3 = sum ||H1 H2 - I||^2 (:150-152): 3 x 3 products on B samples (host plumbing, float64)
            H1, H2 = H64[:B].view(B, 3, 3), H64[B:].view(B, 3, 3)
            R = 2.0 * head.mu * g_loss.double() * (torch.matmul(H1, H2) - torch.eye(3, dtype=torch.float64, device=H64.device))
            gH[:B] = torch.matmul(R, H2.transpose(1, 2)).reshape(B, 9)
            gH[B:] = torch.matmul(H1.transpose(1, 2), R).reshape(B, 9)
        g_masks = None
        if ctx.trained_masks:
            # the warped mask is an image like the warped patch: its gradient w.r.t. H through the two-channel adjoint, its gradient w.r.t.
            # the mask itself through the transposed gather; the unwarped masks get theirs straight from the loss
            K.warp_bwd(src, H64, torch.stack([gwarp[:, 0], gcov], 1).contiguous(), None, 1, gH=gH)
            g_src = K.warp_bwd_img(H64, gcov.unsqueeze(1).contiguous())
            g_m1, g_m2 = grads[6], grads[7]
            g_masks = torch.cat([g_src[:B] + g_m1.unsqueeze(1), (g_src[B:] if double else 0) + g_m2.unsqueeze(1)], 0)
        else:
            K.warp_bwd(src, H64, gwarp.contiguous(), gcov.contiguous(), 1, gH=gH)
        gdelta = K.h4pt_bwd(delta, H64, gH, h)
        return gdelta, None, torch.cat([g_f1, g_f2], 0), g_masks, None


@K.scoped_module
class Model(nn.Module):

    def __init__(self, backbone, **kwargs):
        super().__init__()
        self.backbone = backbone
        self.patch_keys = kwargs['PATCH_KEYS']
        self.mask_keys = kwargs['MASK_KEYS']
        self.feature_keys = kwargs['FEATURE_KEYS']
        self.target_keys = kwargs['TARGET_KEYS']
        self.ld = kwargs['LD']
        self.mu = kwargs['MU']
        assert self.ld == 2, 'Only ld==2 is supported at the moment'
        self.variant = str.lower(kwargs['VARIANT'])
        assert self.variant == 'oneline' or self.variant == 'doubleline', 'Supported variants: OneLine or DoubleLine'
        self.triplet_margin = kwargs['TRIPLET_MARGIN']
        self.triplet_channel_aggregation = kwargs['TRIPLET_AGGREGATION']
        assert self.triplet_channel_aggregation in ('channel-aware', 'channel-agnostic'), 'Do not know this aggregation technique'
        self.last = {}

    def forward(self, data):                                                   # :37-199
        e1, e2 = self.patch_keys
        f1k, f2k = self.feature_keys
        o1 = self.target_keys[0]
        p1, p2 = data[e1], data[e2]
        feat = torch.cat([data[f1k], data[f2k]], 0)
        patches = torch.cat([p1, p2], 0)
        if self.variant == 'doubleline':
            delta = torch.cat([data[o1], data[self.target_keys[1]]], 0)
        else:
            delta = data[o1]
        masks = None
        if not getattr(self.backbone.mask_predictor, 'fix_mask', False):       # :46,:51 (trained masks; FIX_MASK: the all-ones mask is implicit)
            masks = torch.cat([data[self.mask_keys[0]], data[self.mask_keys[1]]], 0)
        loss = _ZhangTripletLoss.apply(delta, patches, feat, masks, self)
        if 'summary_writer' in data:                                           # :158-186 (same tags / keys)
            sw, step, L = data['summary_writer'], data['summary_writer_step'], self.last
            f1, f2, f1w = L["f1"], L["f2"], L["f1w"]
            sw.add_scalars('feature_space', {'patch_2_f': f2.mean().item()}, step)
            sw.add_scalars('feature_space', {'patch_1_f_prime': f1w.mean().item()}, step)
            sw.add_scalars('feature_space', {'patch_1_f': f1.mean().item()}, step)
            if self.variant == 'doubleline':
                sw.add_scalars('feature_space', {'patch_2_f_prime': L["f2w"].mean().item()}, step)
            sw.add_scalars('loss_comp', {'l1': (f2 - f1w).abs().mean().item()}, step)
            sw.add_scalars('loss_comp', {'l3': (f1 - f2).abs().mean().item()}, step)
            B = f1.shape[0]
            eye = torch.eye(3, dtype=L["H_4pt"].dtype, device=f1.device).unsqueeze(0)
            sw.add_scalars('h', {'h1': ((L["H_4pt"][:B] - eye) ** 2).sum().item()}, step)
            if self.variant == 'doubleline':
                sw.add_scalars('loss_comp', {'l2': (f1 - L["f2w"]).abs().mean().item()}, step)
                sw.add_scalars('loss_comp', {'ln1': L["ln1"].item()}, step)
                sw.add_scalars('loss_comp', {'ln2': L["ln2"].item()}, step)
                sw.add_scalars('loss_comp', {'ln3': self.mu * L["ln3"].item()}, step)
                sw.add_scalars('h', {'h2': ((L["H_4pt"][B:] - eye) ** 2).sum().item()}, step)
        delta_gt = data['delta'] if 'delta' in data else None
        delta_hat = data[o1] if o1 in data else None
        return loss, delta_gt, delta_hat

    def predict_homography(self, data):                                        # :201-212
        delta_hat = data[self.target_keys[0]]
        _, H32 = K.h4pt_fwd(delta_hat.contiguous(), data[self.patch_keys[0]].shape[-1])
        return delta_hat, H32

    def state_dict(self, *args, **kwargs):
        net.flush_counters(self)
        return super().state_dict(*args, **kwargs)
