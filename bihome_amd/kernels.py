"""Thin tensor-level wrappers over the C ABI (include/bihome.h): argument checking, raw device
pointers and the current HIP stream.  No arithmetic happens here."""
import ctypes
import os

import torch

from ._lib import (AMAX_FLOATS, BN_DETERMINISTIC, F_DETERMINISTIC, GEOMETRY_FIELDS, ROUTE_DETERMINISTIC, ROUTE_WX3_PC, ROUTE_WX3_SHARED, BhBnAdj, BhBnIn, BhBnReduce, BhConvDesc,
                   BhPack3x3Job, check, lib)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


# Deterministic calls (include/bihome.h "Deterministic calls"): the C library has NO mode - every launch carries its own bit.  What is
# kept here is host-side convenience: a default for newly built models (BIHOME_DETERMINISTIC=1 / set_deterministic) and a scope that the
# models' forward / backward passes open around their launches with THEIR mode (net.Runner.det, the heads' .det), so two models of one
# process can run in different modes.
_DET_DEFAULT = os.environ.get("BIHOME_DETERMINISTIC", "0") == "1"
_DET_SCOPE = []


def set_deterministic(on=True):
    """The mode of calls made OUTSIDE a model's scope and the default of models built from now on: every cross-workgroup sum
    order-independent - bit-identical training steps from run to run, HIP-graph replays identical to eager steps.  Returns the
    previous default.  A model keeps the mode it was built with (net.Runner.det, head.det)."""
    global _DET_DEFAULT
    prev, _DET_DEFAULT = _DET_DEFAULT, bool(on)
    return prev


def deterministic():
    """The mode of the calls being made right now (innermost det_scope, else the default)."""
    return _DET_SCOPE[-1] if _DET_SCOPE else _DET_DEFAULT


class det_scope:
    """with det_scope(on): ... - the wrappers below set the per-call bit of every launch inside from `on`."""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        _DET_SCOPE.append(self.on)
        return self

    def __exit__(self, *exc):
        _DET_SCOPE.pop()


def scoped_function(cls):
    """Class decorator for torch.autograd.Function subclasses that launch kernels: forward records the mode in force (ctx.det),
    backward re-opens it - a backward pass runs wherever autograd calls it, long after the model's own scope was closed."""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *a, **k):
        ctx.det = deterministic()
        return fwd(ctx, *a, **k)

    def backward(ctx, *g):
        with det_scope(ctx.det):
            return bwd(ctx, *g)
    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
    return cls


def scoped_module(cls):
    """Class decorator for the head modules: the mode is fixed when the module is built (self.det) and forward /
    predict_homography open a scope with it."""
    init = cls.__init__

    def __init__(self, *a, **k):
        init(self, *a, **k)
        if not hasattr(self, "det"):
            self.det = deterministic()
    cls.__init__ = __init__
    for name in ("forward", "predict_homography"):
        fn = cls.__dict__.get(name)
        if fn is None:
            continue

        def wrap(fn):
            def method(self, *a, **k):
                with det_scope(self.det):
                    return fn(self, *a, **k)
            method.__name__, method.__doc__ = fn.__name__, fn.__doc__
            return method
        setattr(cls, name, wrap(fn))
    return cls


def _mark(d):
    _route_det(d)
    dp = getattr(d, "bh_packed", None)
    if dp is not None:
        _route_det(dp)
    return d


def _fdet():
    return F_DETERMINISTIC if deterministic() else 0


def _route_det(d):
    """The descriptor with its BH_ROUTE_DETERMINISTIC bit set from the current scope (in place: descriptors are per-call host objects)."""
    d.route = (d.route | ROUTE_DETERMINISTIC) if deterministic() else (d.route & ~ROUTE_DETERMINISTIC)
    return d


_raw_stream, _cur_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice


def _stream():
    # (the raw handle of torch's current stream: two C calls - torch.cuda.current_stream() builds a Stream object through four Python
    #  frames, ~140 times per training step)
    return ctypes.c_void_p(_raw_stream(_cur_device()))


# ------------------------------------------------------------------------------------------------
# optional per-launch timing (bench.py's roofline leg): HIP events on the launch stream around each call
# ------------------------------------------------------------------------------------------------
TIMING_DETAIL = False   # True: key by launch geometry as well (per-layer tables)
TIMING = None       # None, or dict name -> {"events": [(start, end)], "flops": float, "bytes": float, "n": int}


class _Timed:
    __slots__ = ("name", "flops", "bytes", "ev")

    def __init__(self, name, flops=0.0, nbytes=0.0):
        self.name, self.flops, self.bytes = name, flops, nbytes

    def __enter__(self):
        if TIMING is not None:
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.ev[0].record()
        return self

    def __exit__(self, *exc):
        if TIMING is not None:
            self.ev[1].record()
            r = TIMING.setdefault(self.name, {"events": [], "flops": 0.0, "bytes": 0.0, "n": 0})
            r["events"].append(self.ev)
            r["flops"] += self.flops
            r["bytes"] += self.bytes
            r["n"] += 1


_VARIANT_CACHE = {}


def conv_variant(d, which, accumulate=False, bn_groups=0):
    """Kernel symbol(s) a launch described by `d` runs, as the LIBRARY reports it (bh_conv_variant executes the real
    dispatch code with the launches replaced by a name record): 'fwd' / 'dgrad' / 'wgrad'.  Used for the roofline
    attribution in bench.py and the per-launch timing tables; several launches are joined by '+'."""
    key = (tuple(getattr(d, f) for f in GEOMETRY_FIELDS), which, bool(accumulate), int(bn_groups))
    v = _VARIANT_CACHE.get(key)
    if v is None:
        buf = ctypes.create_string_buffer(256)
        if d.precision == 4 and not d.a_bound:
            # (routing does not depend on the magnitude records, but a precision-4 launch without one is refused: describe it with a dummy)
            d = _with_layout(d, d.w_layout)
            d.a_bound = d.b_bound = 256
        check(lib.bh_conv_variant(ctypes.byref(d), {"fwd": 0, "dgrad": 1, "wgrad": 2, "wgrad_det": 3, "dgrad_bnr_z": 4, "dgrad_bnr_y": 5, "dgrad_colsum": 6}[which], int(bool(accumulate)), int(bn_groups),
                                  buf, 256), "bh_conv_variant")
        v = _VARIANT_CACHE[key] = buf.value.decode()
    return v


def _conv_variant(d, which, accumulate=False, bn_groups=0):
    if TIMING is None:
        return ""
    v = conv_variant(d, which, accumulate, bn_groups)
    if TIMING_DETAIL:
        v += " %s N%d %dx%d C%d->%d k%d s%d%s" % (which.split("_")[0], d.N, d.Hi, d.Wi, d.Ci, d.Co, d.kh, d.stride, " T" if d.transposed else "")
    return v


def _bni_name(v):
    """The library's name of a plain launch with the BatchNorm-on-load template argument (second to last) switched on."""
    import re
    if v.startswith("conv3x3_pc_kernel<"):                 # conv3x3_pc_kernel<DGRAD,BNI,EM>
        return re.sub(r"<(false|true),false,", r"<\1,true,", v, count=1)
    # conv3x3_halo_kernel<..,BNI,NP,MAP4> / wgrad_x3_kernel<CB,BNI,NP>
    return re.sub(r",false,(\d)((?:,(?:false|true))*)>", r",true,\1\2>", v, count=1)


def conv_flops(d):
    """Algorithmic flops of one conv launch: 2 * MACs (same count for fwd, dgrad and wgrad)."""
    if d.transposed:
        return 2.0 * d.N * d.Hi * d.Wi * d.Ci * d.Co * d.kh * d.kw
    return 2.0 * d.N * d.Ho * d.Wo * d.Co * d.Ci * d.kh * d.kw


def _chk(t, dtype=torch.float32):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError("bihome_amd kernels need device tensors (got %s); there is no CPU path" % t.device)
    if t.dtype != dtype:
        raise TypeError("expected %s, got %s" % (dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")


# ------------------------------------------------------------------------------------------------
# geometry
# ------------------------------------------------------------------------------------------------
def h4pt_fwd(delta, size):
    """delta [B,4,2] f32 -> (H64 [B,9] f64, H32 [B,3,3] f32)."""
    _chk(delta)
    B = delta.shape[0]
    H64 = torch.empty(B, 9, dtype=torch.float64, device=delta.device)
    H32 = torch.empty(B, 3, 3, dtype=torch.float32, device=delta.device)
    check(lib.bh_h4pt_fwd(_p(delta), B, float(size), float(size), _p(H64), _p(H32), _stream()), "bh_h4pt_fwd")
    return H64, H32


def h4pt_bwd(delta, H64, gH, size):
    _chk(delta); _chk(H64, torch.float64); _chk(gH, torch.float64)
    B = delta.shape[0]
    gd = torch.empty(B, 4, 2, dtype=torch.float32, device=delta.device)
    check(lib.bh_h4pt_bwd(_p(delta), _p(H64), _p(gH), B, float(size), float(size), _p(gd), _stream()), "bh_h4pt_bwd")
    return gd


def dlt_fwd(pf, choice, n, P):
    """pf [B,2,h,w] f32, choice [B,n*P] i64 -> Hdlt [B*n,3,3], delta_hat [B*n,4,2], eig [B*n,96] f64."""
    _chk(pf); _chk(choice, torch.int64)
    B, _, h, w = pf.shape
    assert choice.shape == (B, n * P), (choice.shape, B, n, P)
    Hd = torch.empty(B * n, 3, 3, dtype=torch.float32, device=pf.device)
    dh = torch.empty(B * n, 4, 2, dtype=torch.float32, device=pf.device)
    eig = torch.empty(B * n, 96, dtype=torch.float64, device=pf.device)
    check(lib.bh_dlt_fwd(_p(pf), _p(choice), B, n, P, h, w, _p(Hd), _p(dh), _p(eig), _stream()), "bh_dlt_fwd")
    return Hd, dh, eig


def dlt_bwd(pf, choice, eig, g_delta, n, P, g_H=None):
    """g_H [B*n,9] float64 (optional): gradient reaching the homography itself (hypothesis scoring)."""
    _chk(pf); _chk(choice, torch.int64); _chk(eig, torch.float64); _chk(g_delta); _chk(g_H, torch.float64)
    B, _, h, w = pf.shape
    g_pf = torch.zeros_like(pf)
    check(lib.bh_dlt_bwd_f(_p(pf), _p(choice), _p(eig), _p(g_delta), _p(g_H), B, n, P, h, w, _p(g_pf), _fdet(), _stream()), "bh_dlt_bwd")
    return g_pf


def dsac_scores_fwd(pf, Hd, n):
    """scores[B,n] = softmax(-reprojection error) (ransac_utils.py:76-128); returns (scores, err)."""
    _chk(pf); _chk(Hd)
    B, _, h, w = pf.shape
    err = torch.empty(B, n, dtype=torch.float32, device=pf.device)
    scores = torch.empty_like(err)
    check(lib.bh_dsac_score(_p(pf), _p(Hd), B, n, h, w, _p(err), None, _stream()), "bh_dsac_score")
    check(lib.bh_dsac_scores_fwd(_p(err), B, n, _p(scores), _stream()), "bh_dsac_scores_fwd")
    return scores, err


def dsac_scores_bwd(pf, Hd, scores, g_scores, n):
    """-> (g_pf [B,2,h,w], g_Hd [B*n,9] float64)."""
    _chk(pf); _chk(Hd); _chk(scores); _chk(g_scores)
    B, _, h, w = pf.shape
    g_err = torch.empty_like(scores)
    g_Hd = torch.empty(B * n, 9, dtype=torch.float64, device=pf.device)
    g_pf = torch.zeros_like(pf)
    check(lib.bh_dsac_scores_bwd_f(_p(pf), _p(Hd), _p(scores), _p(g_scores), B, n, h, w, _p(g_err), _p(g_Hd), _p(g_pf), _fdet(), _stream()),
          "bh_dsac_scores_bwd")
    return g_pf, g_Hd


def scale_samples_fwd(x, s, rep):
    """y[b] = x[b // rep] * s[b]: x [Bx, ...] (one row per sample), s [Bx * rep] -> y [Bx * rep, ...]."""
    _chk(x); _chk(s)
    Bn = s.numel()
    L = x.numel() // x.shape[0]
    y = torch.empty((Bn,) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
    check(lib.bh_scale_samples_fwd(_p(x), _p(s), Bn, L, rep, _p(y), _stream()), "bh_scale_samples_fwd")
    return y


def scale_samples_bwd(g_y, x, s, rep, want_gx):
    _chk(g_y); _chk(x); _chk(s)
    Bn = s.numel()
    L = x.numel() // x.shape[0]
    g_x = torch.empty_like(g_y) if want_gx else None
    g_s = torch.empty(Bn, dtype=torch.float32, device=x.device)
    check(lib.bh_scale_samples_bwd_f(_p(g_y), _p(x), _p(s), Bn, L, rep, _p(g_x), _p(g_s), _fdet(), _stream()), "bh_scale_samples_bwd")
    return g_x, g_s


def dsac_score(pf, Hd, n):
    _chk(pf); _chk(Hd)
    B, _, h, w = pf.shape
    err = torch.empty(B, n, dtype=torch.float32, device=pf.device)
    best = torch.empty(B, dtype=torch.int64, device=pf.device)
    check(lib.bh_dsac_score(_p(pf), _p(Hd), B, n, h, w, _p(err), _p(best), _stream()), "bh_dsac_score")
    return err, best


# ------------------------------------------------------------------------------------------------
# warp
# ------------------------------------------------------------------------------------------------
def warp_fwd(img, H64, pool=4, want_cov=True):
    """img [B,C,h,w], H64 [B,9] -> (out [B,C,h,w], cov [B,h/pool,w/pool] or None)."""
    _chk(img); _chk(H64, torch.float64)
    B, C, h, w = img.shape
    out = torch.empty_like(img)
    cov = torch.empty(B, h // pool, w // pool, dtype=torch.float32, device=img.device) if want_cov else None
    with _Timed("warp_fwd_kernel", 0.0, 4.0 * (img.numel() + out.numel() + (cov.numel() if want_cov else 0))):     # 8 B/px/channel
        check(lib.bh_warp_fwd_f(_p(img), _p(H64), B, C, h, w, pool, _p(out), _p(cov), _fdet(), _stream()), "bh_warp_fwd")
    return out, cov


def mask_coverage_fwd(H64, h, w, pool=4):
    _chk(H64, torch.float64)
    B = H64.shape[0]
    cov = torch.empty(B, h // pool, w // pool, dtype=torch.float32, device=H64.device)
    check(lib.bh_warp_fwd_f(None, _p(H64), B, 1, h, w, pool, None, _p(cov), _fdet(), _stream()), "bh_warp_fwd(cov)")
    return cov


def warp_bwd(img, H64, g_out, g_cov, pool=4, gH=None):
    _chk(img); _chk(H64, torch.float64); _chk(g_out); _chk(g_cov)
    B, C, h, w = img.shape
    if gH is None:
        gH = torch.zeros(B, 9, dtype=torch.float64, device=img.device)
    with _Timed("warp_bwd_kernel", 0.0, 4.0 * (img.numel() + g_out.numel() + (g_cov.numel() if g_cov is not None else 0))):
        check(lib.bh_warp_bwd_f(_p(img), _p(H64), _p(g_out), _p(g_cov), B, C, h, w, pool, _p(gH), _fdet(), _stream()), "bh_warp_bwd")
    return gH


def warp_bwd_img(H64, g_out):
    """Adjoint of warp_fwd w.r.t. the image (the trained masks of the Zhang baseline, TripletHead.py:60,69): g_out[B,C,h,w] -> g_img."""
    _chk(H64, torch.float64); _chk(g_out)
    B, C, h, w = g_out.shape
    g_img = torch.empty_like(g_out)
    flags = _fdet()
    n = lib.bh_warp_bwd_img_scratch_doubles(B, C, h, w, flags)
    scratch = torch.empty(n, dtype=torch.float64, device=g_out.device) if n else None
    check(lib.bh_warp_bwd_img_f(_p(H64), _p(g_out), B, C, h, w, _p(g_img), _p(scratch), flags, _stream()), "bh_warp_bwd_img_f")
    return g_img


# ------------------------------------------------------------------------------------------------
# trained content masks (Zhang baseline, FIX_MASK False)
# ------------------------------------------------------------------------------------------------
def mask_fwd(y, f, strength):
    """y[N,1,h,w] = the mask predictor's last BatchNorm output, f = the features (None: mask only) -> (m, g | None, smax[N], imax[N])
    (ContentAware.py:24-26,28-35,128-134: Sigmoid, per-sample max normalisation when strength > 0, G = m * f)."""
    _chk(y); _chk(f)
    N, P = y.shape[0], y.numel() // y.shape[0]
    m = torch.empty_like(y)
    g = torch.empty_like(y) if f is not None else None
    smax = torch.empty(N, dtype=torch.float32, device=y.device)
    imax = torch.empty(N, dtype=torch.int32, device=y.device)
    check(lib.bh_mask_fwd(_p(y), _p(f), N, P, float(strength), _p(m), _p(g), _p(smax), _p(imax), _stream()), "bh_mask_fwd")
    return m, g, smax, imax


def mask_bwd(y, f, m, smax, imax, g_m, g_g, strength, want_gf=True):
    """-> (g_y, g_f | None); g_m / g_g None = no gradient from that consumer."""
    for t in (y, f, m, g_m, g_g):
        _chk(t)
    N, P = y.shape[0], y.numel() // y.shape[0]
    g_y = torch.empty_like(y)
    g_f = torch.empty_like(y) if (want_gf and g_g is not None) else None
    check(lib.bh_mask_bwd(_p(y), _p(f), _p(m), _p(smax), _p(imax), _p(g_m), _p(g_g), N, P, float(strength), _p(g_y), _p(g_f), _stream()),
          "bh_mask_bwd")
    return g_y, g_f


# ------------------------------------------------------------------------------------------------
# triplet
# ------------------------------------------------------------------------------------------------
def triplet_l1_fwd(f1, f2, f1w, f2w, m1w, m2w, m1=None, m2=None):
    """features NHWC [B,hf,wf,C]; masks [B,hf,wf]."""
    for t in (f1, f2, f1w, f2w, m1w, m2w, m1, m2):
        _chk(t)
    B, hf, wf, C = f1.shape
    M1 = torch.empty(B, hf, wf, dtype=torch.float32, device=f1.device)
    M2 = torch.empty_like(M1)
    numden = torch.empty(B, 4, dtype=torch.float64, device=f1.device)
    with _Timed("triplet_fwd_kernel", 0.0, 4.0 * (4 * f1.numel() + 4 * M1.numel())):      # 4 feature maps in, masks in, M1 / M2 out
        check(lib.bh_triplet_l1_fwd_f(_p(f1), _p(f2), _p(f1w), _p(f2w), _p(m1w), _p(m2w), _p(m1), _p(m2), B, hf * wf, C,
                                      _p(M1), _p(M2), _p(numden), _fdet(), _stream()), "bh_triplet_l1_fwd")
    return M1, M2, numden


def oneline_loss_fwd(f1, f2, f1w, m1w, margin, m2=None, rep=1, sample_w=None):
    """iHomE one-line hinge loss (PerceptualHead.py:465-538): returns (loss[1], T[B,hf,wf], numden[B,2], per_sample[B]).
    rep > 1: B = samples * rep hypotheses; f1 / f2 / m2 hold one entry per sample; sample_w[B] = DSAC scores."""
    for t in (f1, f2, f1w, m1w, m2, sample_w):
        _chk(t)
    B, hf, wf, C = f1w.shape
    T = torch.empty(B, hf, wf, dtype=torch.float32, device=f1.device)
    numden = torch.empty(B, 2, dtype=torch.float64, device=f1.device)
    per = torch.empty(B, dtype=torch.float32, device=f1.device)
    loss = torch.empty(1, dtype=torch.float32, device=f1.device)
    check(lib.bh_oneline_loss_fwd_f(_p(f1), _p(f2), _p(f1w), _p(m1w), _p(m2), B, hf * wf, C, float(margin), rep, _p(sample_w), _p(T),
                                    _p(numden), _p(per), _p(loss), _fdet(), _stream()), "bh_oneline_loss_fwd")
    return loss, T, numden, per


def oneline_loss_bwd(g_loss, f2, f1w, m1w, T, numden, m2=None, rep=1, sample_w=None):
    _chk(g_loss); _chk(sample_w)
    B, hf, wf, C = f1w.shape
    g_f1w = torch.empty_like(f1w)
    g_m1w = torch.empty(B, hf, wf, dtype=torch.float32, device=f1w.device)
    check(lib.bh_oneline_loss_bwd(_p(g_loss), _p(f2), _p(f1w), _p(m1w), _p(m2), _p(T), _p(numden), B, hf * wf, C, rep, _p(sample_w),
                                  _p(g_f1w), _p(g_m1w), _stream()), "bh_oneline_loss_bwd")
    return g_f1w, g_m1w


def zhang_triplet_fwd(f1, f2, f1w, f2w, m1w, m2w, margin, hinge, m1=None, m2=None):
    """Zhang content-aware triplet loss on one-channel full-resolution features [B,1,h,w] / masks [B,h,w] (TripletHead.py:75-152):
    returns (T1, T2 | None, numden[B,4]).  f2w / m2w None: one line."""
    for t in (f1, f2, f1w, f2w, m1w, m2w, m1, m2):
        _chk(t)
    B, hw = f1.shape[0], f1.numel() // f1.shape[0]
    T1 = torch.empty(B, hw, dtype=torch.float32, device=f1.device)
    T2 = torch.empty_like(T1) if f2w is not None else None
    numden = torch.empty(B, 4, dtype=torch.float64, device=f1.device)
    check(lib.bh_zhang_triplet_fwd(_p(f1), _p(f2), _p(f1w), _p(f2w), _p(m1w), _p(m2w), _p(m1), _p(m2), B, hw, float(margin), int(bool(hinge)),
                                   _p(T1), _p(T2), _p(numden), _stream()), "bh_zhang_triplet_fwd")
    return T1, T2, numden


def zhang_triplet_bwd(g_loss, f1, f2, f1w, f2w, m1w, m2w, T1, T2, numden, hinge, m1=None, m2=None, mask_grads=False):
    """-> (g_f1, g_f2, g_f1w, g_f2w | None, g_m1w, g_m2w | None), shaped like their tensors; mask_grads (trained masks): two more,
    (g_m1, g_m2) - the gradients of the unwarped masks."""
    _chk(g_loss)
    B, hw = f1.shape[0], f1.numel() // f1.shape[0]
    g_f1, g_f2, g_f1w = torch.empty_like(f1), torch.empty_like(f2), torch.empty_like(f1w)
    g_m1w = torch.empty_like(m1w)
    g_f2w = torch.empty_like(f2w) if f2w is not None else None
    g_m2w = torch.empty_like(m2w) if f2w is not None else None
    g_m1 = torch.empty_like(m1w) if mask_grads else None
    g_m2 = torch.empty_like(m1w) if mask_grads else None
    check(lib.bh_zhang_triplet_bwd_m(_p(g_loss), _p(f1), _p(f2), _p(f1w), _p(f2w), _p(m1w), _p(m2w), _p(m1), _p(m2), _p(T1), _p(T2), _p(numden),
                                     B, hw, int(bool(hinge)), _p(g_f1), _p(g_f2), _p(g_f1w), _p(g_f2w), _p(g_m1w), _p(g_m2w), _p(g_m1), _p(g_m2),
                                     _stream()), "bh_zhang_triplet_bwd_m")
    if mask_grads:
        return g_f1, g_f2, g_f1w, g_f2w, g_m1w, g_m2w, g_m1, g_m2
    return g_f1, g_f2, g_f1w, g_f2w, g_m1w, g_m2w


def bihome_loss_fwd(numden, H1, H2, mu):
    _chk(numden, torch.float64); _chk(H1, torch.float64); _chk(H2, torch.float64)
    loss4 = torch.empty(4, dtype=torch.float32, device=numden.device)
    check(lib.bh_bihome_loss_fwd(_p(numden), _p(H1), _p(H2), numden.shape[0], float(mu), _p(loss4), _stream()),
          "bh_bihome_loss_fwd")
    return loss4


def bihome_loss_bwd(g_loss, f1, f2, f1w, f2w, m1w, m2w, m1, m2, M1, M2, numden, H1, H2, mu, joined=False):
    """joined: the two directions' outputs are the halves of ONE [2B, ...] tensor each and (g_fw, g_mw, gH) are returned whole (the
    caller's extractor / warp adjoints take both directions in one call: no torch.cat of 33 MB)."""
    _chk(g_loss)
    B, hf, wf, C = f1.shape
    g_fw = torch.empty((2 * B,) + tuple(f1w.shape[1:]), dtype=torch.float32, device=f1.device)
    g_mw = torch.empty((2 * B,) + tuple(m1w.shape[1:]), dtype=torch.float32, device=f1.device)
    gH = torch.empty(2 * B, 9, dtype=torch.float64, device=f1.device)
    g_f1w, g_f2w, g_m1w, g_m2w, gH1, gH2 = g_fw[:B], g_fw[B:], g_mw[:B], g_mw[B:], gH[:B], gH[B:]
    with _Timed("triplet_bwd_kernel", 0.0, 4.0 * (6 * f1.numel() + 6 * M1.numel())):      # 4 feature maps in, 2 gradients out
        check(lib.bh_bihome_loss_bwd(_p(g_loss), _p(f1), _p(f2), _p(f1w), _p(f2w), _p(m1w), _p(m2w), _p(m1), _p(m2), _p(M1),
                                     _p(M2), _p(numden), _p(H1), _p(H2), B, hf * wf, C, float(mu), _p(g_f1w), _p(g_f2w),
                                     _p(g_m1w), _p(g_m2w), _p(gH1), _p(gH2), _stream()), "bh_bihome_loss_bwd")
    if joined:
        return g_fw, g_mw, gH
    return g_f1w, g_f2w, g_m1w, g_m2w, gH1, gH2


# ------------------------------------------------------------------------------------------------
# conv stacks
# ------------------------------------------------------------------------------------------------
# conv arithmetic (bh_conv_desc.precision).  'f32' (default) asks for fp32 ACCURACY and lets the library pick the evaluation.
# Round 4: the packed 3x3 kernels (forward, dgrad, weight gradient: 80 % of the step's flops) evaluate every product from TWO FP16
# pieces per operand of the operand times a power-of-two scale per tensor - 22 significand bits and a sign, three partial products
# on v_mfma_f32_32x32x16_f16 with fp32 accumulate, results rescaled exactly (precision 4, "f16x2": ~2^-22 per product; against
# float64 the kernels measure at or BELOW the error of the fp32-input MFMA kernels on every shape tested, tests/test_f16x2_gpu.py,
# because sixteen products enter one fp32 accumulate rounding instead of two).  Everything else runs v_mfma_f32_32x32x2_f32.
# 'f32x3' (precision 2, the default of rounds 2-3): the exact cut into three bf16 pieces, six partial products per product.
# 'f32-mfma' forces the fp32-input MFMA everywhere (precision 0); 'bf16' rounds the operands to bf16 (precision 1).
# 'f32x2' (precision 3): two bf16 pieces per operand, both rounded to nearest, three products - ~4e-6 per product (13x the fp32
# rounding, 500x below 'bf16'); a separately reported reduced-precision arithmetic.
PRECISION = {"f32": 4, "fp32": 4, "f16x2": 4, "f32x3": 2, "f32-mfma": 0, "bf16": 1, "f32x2": 3}
SPLIT_PIECES = {2: 3, 3: 2, 4: 2}   # precision -> pieces per operand of the split-operand kernels
SPLIT_LAYOUT = {2: 2, 3: 3, 4: 4}   # precision -> bh_conv_desc.w_layout of the packed split weights
F16X2 = 4


def amax_record(device):
    """A zeroed magnitude record (include/bihome.h BH_AMAX_FLOATS); net.run_forward / run_backward cut theirs from one arena."""
    return torch.zeros(AMAX_FLOATS, dtype=torch.float32, device=device)


def absmax(x, rec=None):
    """Magnitude record of a tensor by a streaming pass (the fallback where no producer kernel left one)."""
    _chk(x)
    if rec is None:
        rec = amax_record(x.device)
    with _Timed("absmax_kernel", 0.0, 4.0 * x.numel()):
        check(lib.bh_absmax(_p(x), x.numel(), _p(rec), _stream()), "bh_absmax")
    return rec


def amax_of(x, make=True):
    """The magnitude record attached to a tensor (or BnOnLoad) by its producer; make: measure it when there is none."""
    rec = x.amax if isinstance(x, BnOnLoad) else getattr(x, "_bh_amax", None)
    if rec is None and make:
        if isinstance(x, BnOnLoad):
            raise RuntimeError("BatchNorm-on-load operand without a magnitude record (bn_fwd_coeffs(..., amax=...))")
        rec = x._bh_amax = absmax(x)
    return rec


def packed_layout(precision):
    """bh_conv_desc.w_layout of WeightPacker copies for this precision (1: fp32 fragments, 2 / 3: three / two bf16 pieces)."""
    return SPLIT_LAYOUT.get(int(precision), 1)
_ENV_ROUTE = int(os.environ.get("BIHOME_ROUTE", "0"))     # benchmarks: OR these BH_ROUTE_* bits into every conv descriptor


def conv_desc(N, Hi, Wi, Ci, Co, k, stride, pad, transposed=False, in_nchw=False, out_nchw=False, precision=0, route=0):
    d = BhConvDesc()
    d.precision = int(precision)
    d.route = int(route) | _ENV_ROUTE | (ROUTE_DETERMINISTIC if deterministic() else 0)
    d.N, d.Hi, d.Wi, d.Ci, d.Co = N, Hi, Wi, Ci, Co
    d.kh = d.kw = k
    d.stride, d.pad = stride, pad
    d.transposed, d.in_nchw, d.out_nchw = int(transposed), int(in_nchw), int(out_nchw)
    if transposed:
        d.Ho, d.Wo = Hi * stride, Wi * stride
    else:
        d.Ho, d.Wo = (Hi + 2 * pad - k) // stride + 1, (Wi + 2 * pad - k) // stride + 1
    return d


def conv_out_shape(d):
    return (d.N, d.Co, d.Ho, d.Wo) if d.out_nchw else (d.N, d.Ho, d.Wo, d.Co)


def _with_layout(d, w_layout):
    d2 = BhConvDesc()
    ctypes.memmove(ctypes.byref(d2), ctypes.byref(d), ctypes.sizeof(BhConvDesc))
    d2.w_layout = w_layout
    return d2


# the halo-tiled 3x3 kernels: one workgroup per tile position (csrc/conv3x3.hip) / persistent producer-consumer workgroups (csrc/conv3x3_pc.hip)
_C3_KERNELS = ("conv3x3_halo_kernel", "conv3x3_pc_kernel")


def packs_3x3(d):
    """True when the halo-tiled 3x3 kernel takes both the forward and the dgrad of this conv (then the fragment-ordered
    weight copies of bh_conv3x3_pack can be used for it)."""
    # (asked with the layout the packed copies would have: the 4 x 4 map form of the halo kernel exists for split operands only)
    q = _with_layout(d, packed_layout(d.precision)) if (d.w_layout == 0 and int(d.precision) in SPLIT_PIECES) else d
    try:
        return conv_variant(q, "fwd").startswith(_C3_KERNELS) and conv_variant(q, "dgrad").startswith(_C3_KERNELS)
    except Exception:
        return False


class WeightPacker:
    """Fragment-ordered copies (forward and dgrad operand order) of the 3x3 conv weights of one module tree, refreshed by
    ONE bh_conv3x3_pack launch whenever a parameter version changed (every optimizer step in training, once for frozen
    weights).  Buffers and the device job table are allocated once (addresses stay fixed: HIP-graph safe)."""
    _serials = 0

    def __init__(self, split=False, f16=False):
        # split: False / 0 = fp32 fragments (w_layout 1); True / 3 = three exact bf16 pieces (w_layout 2, precision 2);
        # 2 = two rounded bf16 pieces (w_layout 3, precision 3 "f32x2"); 2 with f16: two fp16 pieces of w 2^k (w_layout 4, precision 4)
        self.pieces = 0 if not split else (2 if (split == 2 and split is not True) else 3)
        self.split = self.pieces > 0
        self.f16 = bool(f16)
        if self.f16 and self.pieces != 2:
            raise ValueError("fp16 pieces: two per weight")
        self.layout = 4 if self.f16 else {0: 1, 3: 2, 2: 3}[self.pieces]
        self.job_split = 3 if self.f16 else {0: 0, 3: 1, 2: 2}[self.pieces]
        self.entries = {}          # id(weight) -> (weight, pf, pd)
        WeightPacker._serials += 1
        self.serial = WeightPacker._serials       # names this packer in cache keys (net.run_forward's plan cache)
        self.table = None
        self.versions = None
        self.dirty = False         # a training forward ran since the last pack: the optimizer has (probably) moved the weights

    def get(self, weight, need_dgrad=True):
        e = self.entries.get(id(weight))
        if e is None:
            Co, Ci = weight.shape[0], weight.shape[1]
            n = weight.numel() * self.pieces // 2 if self.split else weight.numel()
            if self.f16:
                n += 16                       # the layer's sixteen partial maxima of |w| (bh_conv3x3_pack_f16)
            pf = torch.empty(n, dtype=torch.float32, device=weight.device)
            pd = torch.empty(n, dtype=torch.float32, device=weight.device) if need_dgrad else None
            e = self.entries[id(weight)] = (weight, pf, pd)
            self.table = None
        return e[1], e[2]

    def invalidate(self):
        """The weights changed behind the version counters (graph replay, broadcast into .data): repack at the next refresh."""
        self.versions = None
        self.dirty = True

    def refresh(self, training=False):
        """Repack when a parameter changed.  Version counters catch load_state_dict / copy_ / the default optimizers, but
        torch's FUSED optimizers update parameters without bumping them - so trainable weights are repacked on every
        training forward, and on the first inference forward after one."""
        if not self.entries:
            return
        vers = tuple((w._version, w.data_ptr()) for w, _, _ in self.entries.values())
        trainable = any(w.requires_grad for w, _, _ in self.entries.values())
        force = trainable and (training or self.dirty)
        self.dirty = bool(training and trainable)
        if self.table is not None and vers == self.versions and not force:
            return
        if self.table is None or any(p != q[1] for p, q in zip(self._ptrs, vers)):
            jobs = (BhPack3x3Job * len(self.entries))()
            for j, (w, pf, pd) in zip(jobs, self.entries.values()):
                if not w.permute(0, 2, 3, 1).is_contiguous():
                    raise RuntimeError("conv weight is not in kernel (channels_last) layout")
                j.w, j.pf, j.pd = w.data_ptr(), pf.data_ptr(), (pd.data_ptr() if pd is not None else None)
                j.Co, j.Ci, j.split = w.shape[0], w.shape[1], self.job_split
            raw = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8)
            dev = next(iter(self.entries.values()))[0].device
            self.table = raw.to(dev)
            self._ptrs = [v[1] for v in vers]
        if self.f16:
            check(lib.bh_conv3x3_pack_f16(_p(self.table), len(self.entries), _stream()), "bh_conv3x3_pack_f16")
        else:
            check(lib.bh_conv3x3_pack(_p(self.table), len(self.entries), _stream()), "bh_conv3x3_pack")
        self.versions = vers


def packer_for_precision(precision):
    """A WeightPacker whose copies are what the packed kernels of this precision read."""
    p = int(precision)
    return WeightPacker(split=SPLIT_PIECES.get(p, 0), f16=p == F16X2)


class BnOnLoad:
    """A training-mode BatchNorm(+ReLU) whose APPLY rides in its consumer: `z` is the BatchNorm's input, `table` the
    [groups][C] x (scale, shift) coefficients made by bn_fwd_coeffs.  The f32x3 3x3 forward and weight-gradient kernels take
    it as their input operand and transform it while staging (zero padding stays zero): the BatchNorm's output is never stored."""
    __slots__ = ("z", "table", "groups", "relu", "amax")

    def __init__(self, z, table, groups, relu, amax=None):
        self.z, self.table, self.groups, self.relu = z, table, groups, relu
        self.amax = amax            # magnitude record of the (never stored) BatchNorm output: precision 4 consumers

    @property
    def shape(self):
        return self.z.shape

    def struct(self):
        b = BhBnIn()
        b.table, b.groups, b.relu = self.table.data_ptr(), int(self.groups), int(bool(self.relu))
        return b


def bn_fwd_coeffs(stats, gamma, beta, rmean, rvar, groups, rows, C, eps, momentum, amax=None):
    """(scale, shift) table of a training-mode BatchNorm from its forward sums (+ the running-statistics update).
    amax: zeroed magnitude record - receives the a-priori bound of the BatchNorm's output."""
    table = torch.empty((groups, C, 2), dtype=torch.float32, device=stats.device)
    check(lib.bh_bn_fwd_coeffs_amax(_p(stats), _p(gamma), _p(beta), _p(rmean), _p(rvar), groups, rows, C, float(eps), float(momentum),
                                    _p(table), _p(amax), _stream()), "bh_bn_fwd_coeffs")
    return table


def conv_fwd(x, w, bias, d, bn_sums=None, groups=1, res=None, relu=False, wpacked=None, amax=None, warp_src=None):
    """wpacked: the forward buffer of WeightPacker for this conv (then `w` is only used for the byte count).
    bn_sums: zeroed float64 sums buffer - the conv also accumulates the batch statistics of its output for the
    BatchNorm that follows (bn_fwd(..., stats=bn_sums, stats_ready=True)).  res / relu: inference epilogue
    y = act(conv + bias + res) (BatchNorm folded into w, bias by the caller).
    warp_src (round 6; the extractor's one-plane stem only): dict(src, H64, pool, cov[, want_image]) - x is an UNFILLED buffer for the
    homography warp of `src`; where the fused kernel applies, the stem makes the warped pixels itself (bh_stem7_fwd_warp: warp_src["cov"]
    and - unless want_image is False - x are written on the way; warp_src["done"] is set), else bh_warp_fwd fills them here and the conv
    runs as always."""
    _mark(d)
    if warp_src is not None:
        ws_ = warp_src
        _chk(ws_["src"]); _chk(ws_["H64"], torch.float64); _chk(ws_["cov"]); _chk(x); _chk(w); _chk(bias); _chk(bn_sums, torch.float64)
        B_, h_, w_ = ws_["src"].shape[0], ws_["src"].shape[-2], ws_["src"].shape[-1]
        if (res is None and not relu and wpacked is None and amax is None and d.Ci == 1 and d.precision == F16X2 and ws_["pool"] == 4
                and not d.transposed and d.kh == 7 and d.kw == 7 and d.stride == 2 and d.pad == 3 and d.Co == 64 and not d.out_nchw
                and d.Ho % 8 == 0 and d.Wo % 8 == 0 and d.Ho * 2 == d.Hi and d.Wo * 2 == d.Wi and d.N * (d.Ho // 8) * (d.Wo // 8) >= 256
                and w.is_contiguous() and os.environ.get("BIHOME_WARP_IN_STEM_FWD", "1") != "0"):
            y = torch.empty(conv_out_shape(d), dtype=torch.float32, device=x.device)
            with _Timed("stem7_fwd_f16_kernel<1,true>", conv_flops(d), 4.0 * (2 * x.numel() + y.numel() + w.numel())):
                # (want_image False: nobody reads the warped image - the kernel keeps it to itself, x stays unfilled)
                rc = lib.bh_stem7_fwd_warp(_p(ws_["src"]), _p(ws_["H64"]), 4, _p(w), _p(bias), _p(y), ctypes.byref(d),
                                           _p(x) if ws_.get("want_image", True) else None, _p(ws_["cov"]), _p(bn_sums), groups, _stream())
            if rc != -2:                                 # (BH_E_UNSUPPORTED: the geometry is not the fp16-piece stem's - the two calls)
                check(rc, "bh_stem7_fwd_warp")
                ws_["done"] = ws_["filled"] = True
                return y
        with _Timed("warp_fwd_kernel", 0.0, 4.0 * (2 * x.numel() + (ws_["cov"].numel() if ws_["cov"] is not None else 0))):
            check(lib.bh_warp_fwd_f(_p(ws_["src"]), _p(ws_["H64"]), B_, 1, h_, w_, int(ws_["pool"]), _p(x), _p(ws_["cov"]), _fdet(), _stream()),
                  "bh_warp_fwd")
        ws_["filled"] = True
    if isinstance(x, BnOnLoad):
        # the BatchNorm in front of this conv is applied on load (packed f32x3 forward only)
        bol, x = x, x.z
        _chk(x); _chk(bias); _chk(bn_sums, torch.float64)
        if d.kh == 1 and wpacked is None and res is None and not relu:
            # round 4: a 1x1 conv behind BatchNorm + ReLU (decoder units) - the generic kernel transforms its A operand while staging
            _chk(w)
            y = torch.empty(conv_out_shape(d), dtype=torch.float32, device=x.device)
            bs = bol.struct()
            with _Timed((conv_variant(d, "fwd") + " bnin" + (" fwd N%d %dx%d C%d->%d k1" % (d.N, d.Hi, d.Wi, d.Ci, d.Co) if TIMING_DETAIL else "")) if TIMING is not None else "",
                        conv_flops(d), 4.0 * (x.numel() + y.numel() + w.numel())):
                check(lib.bh_conv_fwd_bnin(_p(x), _p(w), _p(bias), _p(y), ctypes.byref(d), _p(bn_sums), groups, ctypes.byref(bs), _stream()),
                      "bh_conv_fwd_bnin(1x1)")
            return y
        if wpacked is None or res is not None or relu:
            raise RuntimeError("BatchNorm-on-load needs the packed f32x3 3x3 forward")
        y = torch.empty(conv_out_shape(d), dtype=torch.float32, device=x.device)
        dp = getattr(d, "bh_packed", None) or _with_layout(d, packed_layout(d.precision))
        _route_det(dp)
        if dp.precision == F16X2:
            dp.a_bound = amax_of(bol).data_ptr()
        bs = bol.struct()
        # (the library's name of the plain launch with the last template argument - BatchNorm-on-load - switched on)
        with _Timed(_bni_name(_conv_variant(dp, "fwd", bn_groups=groups if bn_sums is not None else 0)) if TIMING is not None else "",
                    conv_flops(d), 4.0 * (x.numel() + y.numel() + w.numel())):
            check(lib.bh_conv_fwd_bnin(_p(x), _p(wpacked), _p(bias), _p(y), ctypes.byref(dp), _p(bn_sums), groups, ctypes.byref(bs),
                                       _stream()), "bh_conv_fwd_bnin")
        return y
    _chk(x); _chk(w); _chk(bias); _chk(res)
    y = torch.empty(conv_out_shape(d), dtype=torch.float32, device=x.device)
    if wpacked is not None:
        d, w = _route_det(getattr(d, "bh_packed", None) or _with_layout(d, packed_layout(d.precision))), wpacked
        if d.precision == F16X2:
            d.a_bound = amax_of(x).data_ptr()
    with _Timed(_conv_variant(d, "fwd", bn_groups=groups if bn_sums is not None else 0), conv_flops(d),
                4.0 * (x.numel() + y.numel() + w.numel())):
        if res is not None or relu:
            check(lib.bh_conv_fwd_act(_p(x), _p(w), _p(bias), _p(res), _p(y), ctypes.byref(d), int(bool(relu)), _stream()),
                  "bh_conv_fwd_act")
        elif amax is not None and bn_sums is None and wpacked is None:
            # (amax: zeroed magnitude record - the generic kernel measures max |y| in its epilogue; precision-4 consumers of y)
            check(lib.bh_conv_fwd_amax(_p(x), _p(w), _p(bias), _p(y), ctypes.byref(d), _p(amax), _stream()), "bh_conv_fwd_amax")
            y._bh_amax = amax
        elif bn_sums is None:
            check(lib.bh_conv_fwd(_p(x), _p(w), _p(bias), _p(y), ctypes.byref(d), _stream()), "bh_conv_fwd")
        else:
            _chk(bn_sums, torch.float64)
            check(lib.bh_conv_fwd_bnstats(_p(x), _p(w), _p(bias), _p(y), ctypes.byref(d), _p(bn_sums), groups, _stream()),
                  "bh_conv_fwd_bnstats")
    return y


_STEM_WT = {}          # (id(param), device, Ci) -> [version, table]: one table per stem parameter, NEVER freed or replaced - a
                       # captured HIP graph holds its address; a changed weight is written into the same storage
_STEM_WT_DIRTY = set()


def invalidate_stem_tables():
    """The weights may have changed behind the version counters (graph replay, broadcast): rebuild every table IN PLACE at its next use."""
    _STEM_WT_DIRTY.update(_STEM_WT.keys())


def _stem_dgrad_two_step(gy, w, d, wkey=None):
    """dgrad of the 1- or 3-input-channel 7x7/2 stem as (1x1 MFMA GEMM gy x w^T -> per-source-pixel tap table) + col2im.
    w is the kernel-layout weight [Co][7][7][Ci]; the result is NCHW [N,Ci,Hi,Wi] (== NHWC for Ci = 1).
    wkey: the SOURCE parameter and its version, (param, param._version) - `w` itself may be a per-forward temporary (the
    extractor's channel-summed stem weight) whose address and version repeat although the parameter changed; without a
    key the transposed table is rebuilt on every call (64 x 52 floats)."""
    cols = 49 * d.Ci
    ld = (cols + 3) // 4 * 4                                                          # 52 / 148
    import weakref
    key = None if wkey is None else (id(wkey[0]), str(w.device), d.Ci)
    ent = _STEM_WT.get(key) if key is not None else None
    if ent is not None and (ent[2]() is not wkey[0] or ent[1].shape != (ld, 1, 1, d.Co)):
        ent = None                                                                    # (the id of a dead parameter was reused)
    if ent is None:
        wt = torch.zeros(ld, 1, 1, d.Co, dtype=torch.float32, device=w.device)        # [tap*Ci + c, padded][Co]
        wt[:cols, 0, 0, :] = w.reshape(d.Co, cols).t()
        if key is not None:
            _STEM_WT[key] = [wkey[1], wt, weakref.ref(wkey[0])]
            _STEM_WT_DIRTY.discard(key)
    else:
        wt = ent[1]
        if ent[0] != wkey[1] or key in _STEM_WT_DIRTY:
            wt[:cols, 0, 0, :] = w.reshape(d.Co, cols).t()                            # same storage: graph-safe
            ent[0] = wkey[1]
            _STEM_WT_DIRTY.discard(key)
    d1 = conv_desc(d.N, d.Ho, d.Wo, d.Co, ld, 1, 1, 0, precision=d.precision)
    tm = conv_fwd(gy, wt, None, d1)                                                   # [N,Ho,Wo,ld]
    shape = (d.N, d.Hi, d.Wi, 1) if d.Ci == 1 else (d.N, d.Ci, d.Hi, d.Wi)
    gx = torch.empty(shape, dtype=torch.float32, device=gy.device)
    with _Timed("col2im_c1_kernel", 0.0, 4.0 * (tm.numel() + gx.numel())):
        check(lib.bh_col2im_c1(_p(tm), _p(gx), ctypes.byref(d), ld, _stream()), "bh_col2im_c1")
    return gx


def dgrad_bn_reduce_ok(d):
    """True when conv_dgrad(..., bn_reduce=...) is available for this conv (the halo-tiled 3x3 kernel takes its dgrad)."""
    return conv_variant(d, "dgrad").startswith(_C3_KERNELS)


def bias_grad_from_sums(sums, gbias, groups, C):
    """gbias[C] += column sums accumulated by conv_dgrad(..., colsum=sums)."""
    _chk(sums, torch.float64); _chk(gbias)
    check(lib.bh_bias_grad_from_sums(_p(sums), _p(gbias), groups, C, _stream()), "bh_bias_grad_from_sums")


def conv_dgrad(gy, w, d, out=None, bn_reduce=None, wkey=None, wpacked=None, colsum=None, warp_sink=None):
    """warp_sink (round 6; the extractor stem's dgrad on a one-channel image only): dict(src, H64, g_cov, pool, gH) - the image is the
    homography warp of `src`; where the fused kernel applies, the warp's adjoint is accumulated into gH, warp_sink["done"] is set and
    None is returned instead of the gradient image.
    colsum (only when dgrad_bn_reduce_ok(d), no `out`, no bn_reduce): zeroed bn_stats_buffer(1, Ci) - the per-channel sums
    of the gradient written are accumulated in the epilogue (bias gradient of the producer of this conv's input).
    wpacked: the dgrad buffer of WeightPacker for this conv.
    wkey: (param, param._version) of the parameter `w` was derived from (cache key of derived weight tables).
    bn_reduce (only when dgrad_bn_reduce_ok(d)): dict(z, y, stats, gamma, beta, eps, relu, sums, groups) of the
    BatchNorm whose output gradient this call completes - its backward sums are accumulated into `sums` (zeroed
    bn_stats_buffer) in the conv epilogue; pass them to bn_bwd(..., sums_ready=sums)."""
    _chk(gy); _chk(w)
    _mark(d)
    if wpacked is not None:
        d, w = _route_det(getattr(d, "bh_packed", None) or _with_layout(d, packed_layout(d.precision))), wpacked
        if d.precision == F16X2:
            d.a_bound = amax_of(gy).data_ptr()
    if colsum is not None:
        assert out is None and bn_reduce is None
        _chk(colsum, torch.float64)
        out = torch.empty((d.N, d.Hi, d.Wi, d.Ci), dtype=torch.float32, device=gy.device)
        with _Timed(_conv_variant(d, "dgrad_colsum"), conv_flops(d), 4.0 * (gy.numel() + out.numel() + w.numel())):
            check(lib.bh_conv_dgrad_colsum(_p(gy), _p(w), _p(out), ctypes.byref(d), _p(colsum), _stream()), "bh_conv_dgrad_colsum")
        return out
    if bn_reduce is not None:
        acc = out is not None
        if out is None:
            out = torch.empty((d.N, d.Hi, d.Wi, d.Ci), dtype=torch.float32, device=gy.device)
        b = bn_reduce
        _chk(b["z"]); _chk(b["y"]); _chk(b["stats"], torch.float64); _chk(b["sums"], torch.float64)
        # (amax_d, round 6: a zeroed magnitude record that receives max |mask(d)| of the gradient written - conv_wgrad_bnadj's scale bound)
        _chk(b.get("amax_d"))
        st = BhBnReduce(_p(b["z"]), _p(b["y"]), _p(b["stats"]), _p(b["gamma"]), _p(b["beta"]), float(b["eps"]), int(bool(b["relu"])),
                        _p(b.get("amax_d")))
        with _Timed(_conv_variant(d, "dgrad_bnr_y" if b["y"] is not None else "dgrad_bnr_z", acc, int(b["groups"])), conv_flops(d),
                    4.0 * (gy.numel() + out.numel() * ((3 if acc else 2) + (1 if b["y"] is not None else 0)) + w.numel())):
            check(lib.bh_conv_dgrad_bnreduce(_p(gy), _p(w), _p(out), ctypes.byref(d), int(acc), ctypes.byref(st), _p(b["sums"]),
                                             int(b["groups"]), _stream()), "bh_conv_dgrad_bnreduce")
        if acc:
            out._bh_amax = None              # (a record left by an earlier producer no longer bounds the sum: as add_ does)
        return out
    if (out is None and not d.transposed and d.kh == 7 and d.stride == 2 and not d.out_nchw
            and (d.Ci == 1 or (d.Ci == 3 and d.in_nchw)) and d.Co % 4 == 0 and d.N * d.Ho * d.Wo >= 4096):
        if (d.Ci == 1 and d.Co == 64 and d.pad == 3 and d.Ho % 8 == 0 and d.Wo % 8 == 0 and d.Ho * 2 == d.Hi and d.Wo * 2 == d.Wi
                and w.is_contiguous() and os.environ.get("BIHOME_STEM_DGRAD_FUSED", "1") != "0"):
            # one kernel (round 4): a workgroup owns a 16 x 16 image tile - window GEMM into a tap table in LDS + gather; no tap table in HBM
            if warp_sink is not None and not deterministic() and os.environ.get("BIHOME_WARP_IN_STEM_DGRAD", "1") != "0":
                # round 6: the image is the homography warp of warp_sink["src"] and its gradient has one consumer, the warp's adjoint: the
                # kernel applies it to every pixel it has just summed (bh_stem7_dgrad_c1_warp) - no gradient image, no warp_bwd launch
                ws_ = warp_sink
                _chk(ws_["src"]); _chk(ws_["H64"], torch.float64); _chk(ws_["gH"], torch.float64); _chk(ws_["g_cov"])
                # (the name keeps the plain kernel's: rocprofv3 prints the template argument, bh_conv_variant-style)
                with _Timed("stem7_dgrad_c1_kernel<true>", conv_flops(d), 4.0 * (gy.numel() + ws_["src"].numel())):
                    check(lib.bh_stem7_dgrad_c1_warp(_p(gy), _p(w), None, ctypes.byref(d), _p(ws_["src"]), _p(ws_["H64"]), _p(ws_["g_cov"]),
                                                     int(ws_["pool"]), _p(ws_["gH"]), _stream()), "bh_stem7_dgrad_c1_warp")
                ws_["done"] = True
                return None
            gx = torch.empty((d.N, d.Hi, d.Wi, 1), dtype=torch.float32, device=gy.device)
            with _Timed("stem7_dgrad_c1_kernel", conv_flops(d), 4.0 * (gy.numel() + gx.numel())):
                check(lib.bh_stem7_dgrad_c1(_p(gy), _p(w), _p(gx), ctypes.byref(d), _stream()), "bh_stem7_dgrad_c1")
            return gx
        return _stem_dgrad_two_step(gy, w, d, wkey)
    acc = out is not None
    if (not d.transposed and d.stride == 2 and not d.in_nchw and not d.out_nchw and d.Co % 4 == 0 and d.Ci % 4 == 0
            and d.Hi % 2 == 0 and d.Wi % 2 == 0 and d.Ho * 2 == d.Hi and d.Wo * 2 == d.Wi
            and ((d.kh == 3 and d.kw == 3 and d.pad == 1) or (d.kh == 1 and d.kw == 1 and d.pad == 0))):
        # stride-2 dgrad by output parity class (csrc/conv_gemm.hip bh_conv_dgrad_s2)
        if out is None:
            out = torch.empty((d.N, d.Hi, d.Wi, d.Ci), dtype=torch.float32, device=gy.device)
        wpack = torch.empty(w.numel(), dtype=torch.float32, device=gy.device)
        nlaunch = 2 if d.kh == 3 else (2 + (0 if acc else 1))         # (weight pack + ONE launch for the parity classes: round 6)
        with _Timed("conv_dgrad_s2(%d kernels)" % nlaunch + (" N%d %dx%d C%d->%d k%d" % (d.N, d.Hi, d.Wi, d.Ci, d.Co, d.kh) if TIMING_DETAIL else ""),
                    conv_flops(d) / 4.0 * (1.0 if d.kh == 1 else 1.0), 4.0 * (gy.numel() + out.numel() * (2 if acc else 1) + 2 * w.numel())):
            check(lib.bh_conv_dgrad_s2(_p(gy), _p(w), _p(out), ctypes.byref(d), int(acc), _p(wpack), _stream()), "bh_conv_dgrad_s2")
        if acc:
            out._bh_amax = None
        return out
    if out is None:
        shape = (d.N, d.Ci, d.Hi, d.Wi) if d.in_nchw else (d.N, d.Hi, d.Wi, d.Ci)
        out = torch.empty(shape, dtype=torch.float32, device=gy.device)
    with _Timed(_conv_variant(d, "dgrad", acc), conv_flops(d), 4.0 * (gy.numel() + out.numel() * (2 if acc else 1) + w.numel())):
        check(lib.bh_conv_dgrad(_p(gy), _p(w), _p(out), ctypes.byref(d), int(acc), _stream()), "bh_conv_dgrad")
    if acc:
        out._bh_amax = None
    return out


def wgrad_det_bytes(d):
    """Workspace bytes of the deterministic weight-gradient form for this conv (0: not available for the shape)."""
    _route_det(d)
    if d.precision == 4 and not d.a_bound:
        # (a precision-4 launch is described WITH magnitude records - conv_wgrad attaches them -, and the fp16-piece forms size their
        #  workspace differently from what a description without them falls back to: ask with placeholders, as conv_variant does)
        d = _with_layout(d, d.w_layout)
        d.a_bound = d.b_bound = 256
    return int(lib.bh_conv_wgrad_det_bytes(ctypes.byref(d)))


_STEM_WGRAD_WS = {}      # (device, bytes, stream) -> the stem weight gradient's partial-sum workspace: launches of ONE stream serialise on it;
                         # another stream (a second model's side stream) gets its own.  Allocated by a step's first eager run - GraphedStep
                         # warms up eagerly before it captures, so the buffer is never born inside a graph's private pool


def conv_wgrad_bnadj(x, dout, gw, d, ws, bna, amax_d):
    """Round 6 (bh_conv_wgrad_bnadj): the weight gradient of a 3x3 conv whose output feeds a training-mode BatchNorm, from the gradient
    `dout` of that BatchNorm's OUTPUT - the adjoint is applied while the kernel stages its operand.  x: the conv's input (tensor or
    BnOnLoad); bna: dict(z, y, stats, sums, gamma, beta, eps, relu, groups) as conv_dgrad's bn_reduce plus the backward sums it filled;
    amax_d: the magnitude record conv_dgrad(..., bn_reduce=dict(amax_d=...)) left.  Returns False where the kernel does not apply
    (nothing was launched: run bn_bwd and conv_wgrad as before)."""
    _route_det(d)
    if d.precision != F16X2 or not getattr(d, "bh_wx3", True) or ws is None or deterministic():
        return False
    bol = x if isinstance(x, BnOnLoad) else None
    xt = x.z if bol is not None else x
    _chk(xt); _chk(dout); _chk(gw); _chk(ws); _chk(amax_d); _chk(bna["z"]); _chk(bna["y"]); _chk(bna["stats"], torch.float64); _chk(bna["sums"], torch.float64)
    d.a_bound, d.b_bound = amax_of(x).data_ptr(), amax_d.data_ptr()
    st = BhBnAdj(_p(bna["z"]), _p(bna["y"]), _p(bna["stats"]), _p(bna["sums"]), _p(bna["gamma"]), _p(bna["beta"]), float(bna["eps"]),
                 int(bool(bna["relu"])), int(bna["groups"]))
    bs = bol.struct() if bol is not None else None
    name = ""
    if TIMING is not None:
        name = conv_variant(d, "wgrad_det").replace(",false,false>", ",false,false,true>").replace(",true,false>", ",true,false,true>") + (" bnin" if bol is not None else "")
    with _Timed(name, conv_flops(d), 4.0 * (xt.numel() + dout.numel() * (3 if bna["y"] is not None else 2) + gw.numel())):
        rc = lib.bh_conv_wgrad_bnadj(_p(xt), _p(dout), _p(gw), ctypes.byref(d), _p(ws), ws.numel() * 4, ctypes.byref(bs) if bs is not None else None,
                                     ctypes.byref(st), _stream())
    if rc == -2:
        return False
    check(rc, "bh_conv_wgrad_bnadj")
    return True


def conv_wgrad_batch(items, ws):
    """Round 6 (bh_conv_wgrad_batch): the fp16-piece weight gradients of 2 .. 4 layers of one geometry in one launch.  items: [(x, gy, gw, d)],
    x a tensor or BnOnLoad (all of one kind); every d of the same geometry and route.  Returns False where the kernel form does not apply
    (nothing launched: call conv_wgrad per layer)."""
    n = len(items)
    d0 = items[0][3]
    if not (1 <= n <= 4) or d0.precision != F16X2 or ws is None or deterministic():
        return False
    xs, bols = [], []
    for x, gy, gw, d in items:
        _route_det(d)
        bol = x if isinstance(x, BnOnLoad) else None
        xt = x.z if bol is not None else x
        _chk(xt); _chk(gy); _chk(gw)
        d.a_bound, d.b_bound = amax_of(x).data_ptr(), amax_of(gy).data_ptr()
        xs.append(xt); bols.append(bol)
    if any((b is None) != (bols[0] is None) for b in bols):
        return False
    _chk(ws)
    PA = ctypes.c_void_p * n
    xa, ga, wa = PA(*[t.data_ptr() for t in xs]), PA(*[it[1].data_ptr() for it in items]), PA(*[it[2].data_ptr() for it in items])
    da = (ctypes.POINTER(BhConvDesc) * n)(*[ctypes.pointer(it[3]) for it in items])
    structs = [b.struct() for b in bols] if bols[0] is not None else None
    ba = (ctypes.POINTER(BhBnIn) * n)(*[ctypes.pointer(b) for b in structs]) if structs is not None else None
    name = ""
    if TIMING is not None:
        name = (conv_variant(d0, "wgrad_det") + (" bnin" if structs is not None else "")) + " x%d layers" % n
    nb = sum(4.0 * (t.numel() + it[1].numel() + it[2].numel()) for t, it in zip(xs, items))
    with _Timed(name, n * conv_flops(d0), nb):
        rc = lib.bh_conv_wgrad_batch(n, xa, ga, wa, da, _p(ws), ws.numel() * 4, ba, _stream())
    if rc == -2:
        return False
    check(rc, "bh_conv_wgrad_batch")
    return True


def conv_wgrad(x, gy, gw, gbias, d, det_ws=None):
    """gw += x^T gy (split-K MFMA kernel, fp32 atomics); gbias += column sums of gy (separate launch, own timing entry so
    that the wgrad entry is the kernel rocprofv3 lists under the same name).
    det_ws: float32 workspace of >= wgrad_det_bytes(d) bytes - the split-K partial tiles are stored there and added in a
    fixed order by a second launch (bitwise repeatable, no atomics); ignored where the shape has no deterministic form."""
    _route_det(d)
    if d.precision == F16X2 and getattr(d, "bh_wx3", True):
        # both records -> the fp16-piece kernel; (a description without them runs the exact three-piece form)
        d.a_bound, d.b_bound = amax_of(x).data_ptr(), amax_of(gy).data_ptr()
    if isinstance(x, BnOnLoad):
        bol, x = x, x.z
        _chk(x); _chk(gy); _chk(gw); _chk(gbias)
        if d.kh == 1:
            # round 4: 1x1 conv behind BatchNorm + ReLU - the small-channel kernel transforms x while staging (no workspace outside
            # deterministic calls)
            bs = bol.struct()
            ws_ = det_ws if deterministic() else None
            with _Timed((conv_variant(d, "wgrad") + " bnin") if TIMING is not None else "", conv_flops(d), 4.0 * (x.numel() + gy.numel() + gw.numel())):
                check(lib.bh_conv_wgrad_bnin(_p(x), _p(gy), _p(gw), _p(gbias), ctypes.byref(d), _p(ws_), ws_.numel() * 4 if ws_ is not None else 0,
                                             ctypes.byref(bs), _stream()), "bh_conv_wgrad_bnin(1x1)")
            return
        need = wgrad_det_bytes(d)
        if det_ws is None or not (0 < need <= det_ws.numel() * 4):
            raise RuntimeError("BatchNorm-on-load needs the f32x3 weight gradient and its workspace")
        bs = bol.struct()
        with _Timed((_bni_name(conv_variant(d, "wgrad_det")) if TIMING is not None else ""), conv_flops(d),
                    4.0 * (x.numel() + gy.numel() + gw.numel())):
            check(lib.bh_conv_wgrad_bnin(_p(x), _p(gy), _p(gw), _p(gbias), ctypes.byref(d), _p(det_ws), det_ws.numel() * 4,
                                         ctypes.byref(bs), _stream()), "bh_conv_wgrad_bnin")
        return
    _chk(x); _chk(gy); _chk(gw); _chk(gbias)
    if (d.kh == 7 and d.stride == 2 and d.Ci == 2 and d.in_nchw and gbias is None and gw.is_contiguous()
            and os.environ.get("BIHOME_STEM_WGRAD_FUSED", "1") != "0"):
        # the backbone's stem (round 4): dedicated kernel + ordered reduction through a private workspace (no atomics: every mode)
        need = lib.bh_stem7_wgrad_ws_bytes(ctypes.byref(d))
        if need:
            if torch.cuda.is_current_stream_capturing():
                # under a graph capture the workspace is born in THAT graph's private pool: it must not outlive the capture in a module-level
                # cache (a second graph capturing on the same stream handle would be handed a buffer the first graph's pool owns - round-5
                # ADVICE); stream order inside the capture keeps a freed block safe for later allocations of the same stream
                ws = torch.empty(need // 4, dtype=torch.float32, device=x.device)
            else:
                key = (str(x.device), int(need), int(torch.cuda.current_stream(x.device).cuda_stream))
                ws = _STEM_WGRAD_WS.get(key)
                if ws is None:
                    if len(_STEM_WGRAD_WS) >= 8:           # (streams come and go: the cache stays bounded)
                        _STEM_WGRAD_WS.clear()
                    ws = _STEM_WGRAD_WS[key] = torch.empty(need // 4, dtype=torch.float32, device=x.device)
            with _Timed("stem7_wgrad_kernel<2>+stem7_wgrad_reduce_kernel<2>", conv_flops(d), 4.0 * (x.numel() + gy.numel() + gw.numel())):
                check(lib.bh_stem7_wgrad(_p(x), _p(gy), _p(gw), ctypes.byref(d), _p(ws), need, _stream()), "bh_stem7_wgrad")
            return
    det = deterministic()
    if det and det_ws is None:
        raise RuntimeError("deterministic mode: conv_wgrad needs a workspace (det_ws) - bh_conv_wgrad has only the atomic form")
    if det_ws is not None:
        need = wgrad_det_bytes(d)
        if det and not (0 < need <= det_ws.numel() * 4):
            raise RuntimeError("deterministic mode: workspace of %d bytes, %d needed" % (det_ws.numel() * 4, need))
        if 0 < need <= det_ws.numel() * 4:
            with _Timed((conv_variant(d, "wgrad_det") if TIMING is not None else ""), conv_flops(d), 4.0 * (x.numel() + gy.numel() + gw.numel())):
                # (deterministic mode: the bias gradient rides in the same call - its limb entries are the tail of the workspace)
                check(lib.bh_conv_wgrad_det(_p(x), _p(gy), _p(gw), _p(gbias) if det else None, ctypes.byref(d), _p(det_ws),
                                            det_ws.numel() * 4, _stream()), "bh_conv_wgrad_det")
            if gbias is not None and not det:
                check(lib.bh_conv_bias_grad(_p(gy), _p(gbias), ctypes.byref(d), _stream()), "bh_conv_bias_grad")
            return
    with _Timed((conv_variant(d, "wgrad") if TIMING is not None else "") + (" N%d %dx%d C%d->%d k%d s%d%s" % (d.N, d.Hi, d.Wi, d.Ci, d.Co, d.kh, d.stride, " T" if d.transposed else "") if TIMING_DETAIL else ""), conv_flops(d), 4.0 * (x.numel() + gy.numel() + gw.numel())):
        check(lib.bh_conv_wgrad(_p(x), _p(gy), _p(gw), None, ctypes.byref(d), _stream()), "bh_conv_wgrad")
    if gbias is not None:
        with _Timed("bias_grad(colsum)", 0.0, 4.0 * gy.numel()):
            check(lib.bh_conv_bias_grad(_p(gy), _p(gbias), ctypes.byref(d), _stream()), "bh_conv_bias_grad")


BN_SUM_STRIDE = 16                                     # doubles between entries (one 128-byte line each; csrc/common.h)
BN_SUM_SLOTS = 1


def bn_stats_doubles(groups, C):
    return BN_SUM_SLOTS * groups * C * 2 * BN_SUM_STRIDE    # == lib.bh_bn_stats_doubles(groups, C)


def bn_stats_buffer(groups, C, device):
    """Zeroed [BN_SUM_SLOTS,groups,C,2,BN_SUM_STRIDE] float64 sums buffer (entry at [...,0], total = sum over slots; the
    kernels accumulate with f64 atomics)."""
    return torch.zeros(lib.bh_bn_stats_doubles(groups, C), dtype=torch.float64, device=device)


def bn_fwd(x, gamma, beta, rmean, rvar, res, groups, eps, momentum, relu, training, stats=None, stats_ready=False, amax=None):
    """x [groups*rows..., C] NHWC (any leading shape); returns (y, stats).  stats: optional zeroed sums buffer (a slice of
    the caller's arena); stats_ready: the producing conv already accumulated the sums (conv_fwd(..., bn_sums=stats))."""
    _chk(x); _chk(res)
    # channels: the affine parameter's length (a one-channel NCHW tensor [N,1,h,w] IS its NHWC form: the Zhang extractor's last layer)
    C = gamma.numel() if gamma is not None else x.shape[-1]
    rows = x.numel() // C // groups
    y = torch.empty_like(x)
    if stats is None:
        stats = bn_stats_buffer(groups, C, x.device)
        stats_ready = False
    flags = (1 if relu else 0) | (2 if res is not None else 0) | (8 if stats_ready else 0) | (BN_DETERMINISTIC if deterministic() else 0)
    nb = 4.0 * x.numel() * ((2 if (training and not stats_ready) else 1) + 1 + (1 if res is not None else 0))
    with _Timed("bn_fwd(%d kernels)" % (2 if (training and not stats_ready) else 1) + (" g%d rows%d C%d" % (groups, rows, C) if TIMING_DETAIL else ""), 0.0, nb):
        check(lib.bh_bn_fwd_amax(_p(x), _p(gamma), _p(beta), _p(rmean), _p(rvar), _p(res), _p(y), _p(stats), groups, rows, C,
                                 float(eps), float(momentum), flags, 0 if training else 1, _p(amax), _stream()), "bh_bn_fwd")
    if amax is not None:
        y._bh_amax = amax           # zeroed magnitude record, now max |y| (measured by the apply kernel)
    return y, stats


def bn_bwd(gy, y, x, gamma, stats, rmean, rvar, groups, eps, relu, training, want_gres, ggamma=None, gbeta=None, beta=None,
           had_res=None, scratch=None, sums_ready=None, amax=None):
    """sums_ready: the gradient sums buffer filled by conv_dgrad(..., bn_reduce=...) - reduce / finalize are skipped."""
    _chk(gy); _chk(x)
    C = gamma.numel() if gamma is not None else x.shape[-1]
    rows = x.numel() // C // groups
    gx = torch.empty_like(x)
    gres = torch.empty_like(x) if want_gres else None
    if sums_ready is not None:
        scratch = sums_ready
    elif scratch is None:
        scratch = torch.empty(lib.bh_bn_scratch_doubles(groups, C), dtype=torch.float64, device=x.device)
    had_res = want_gres if had_res is None else had_res
    mask_from_x = relu and not had_res          # y = relu(x*scale+shift): the mask is recomputed, y is not read
    flags = ((1 if relu else 0) | (2 if want_gres else 0) | (4 if mask_from_x else 0) | (16 if sums_ready is not None else 0)
             | (BN_DETERMINISTIC if deterministic() else 0))
    passes = 1 if sums_ready is not None else 2
    nb = 4.0 * x.numel() * (passes * (2 + (1 if (relu and not mask_from_x) else 0)) + 1 + (1 if want_gres else 0))
    with _Timed("bn_bwd(%d kernels)" % (1 if sums_ready is not None else 3) + (" g%d rows%d C%d" % (groups, rows, C) if TIMING_DETAIL else ""), 0.0, nb):
        check(lib.bh_bn_bwd_amax(_p(gy), _p(y), _p(x), _p(gamma), _p(beta), _p(stats), _p(gx), _p(gres), _p(ggamma), _p(gbeta),
                                 _p(scratch), groups, rows, C, float(eps), flags, 0 if training else 1, _p(rmean), _p(rvar),
                                 _p(amax), _stream()), "bh_bn_bwd")
    if amax is not None:
        gx._bh_amax = amax          # zeroed magnitude record, now max |gx|
    return gx, gres


def bn_stats(x, stats, groups, C):
    """stats (zeroed slice of the caller's arena) += per-channel (sum, sum of squares) of x: the statistics pass alone."""
    _chk(x); _chk(stats, torch.float64)
    rows = x.numel() // C // groups
    with _Timed("bn_stats_kernel", 0.0, 4.0 * x.numel()):
        check(lib.bh_bn_stats(_p(x), _p(stats), groups, rows, C, BN_DETERMINISTIC if deterministic() else 0, _stream()), "bh_bn_stats")
    return stats


class BnPooled:
    """A BatchNorm(+ReLU) whose only consumer is MaxPool2d(3, 2, 1), both done by bn_maxpool_fwd: the BatchNorm's output does not exist;
    `pooled` / `idx` are what the pooling op returns, `shape` the shape the BatchNorm output would have had (for the adjoint)."""
    __slots__ = ("pooled", "idx", "shape")

    def __init__(self, pooled, idx, shape):
        self.pooled, self.idx, self.shape = pooled, idx, tuple(shape)


def bn_maxpool_fwd(x, gamma, beta, rmean, rvar, groups, eps, momentum, relu, training, stats, stats_ready, want_index=True, amax=None):
    """x NHWC -> BnPooled(maxpool3s2(act(bn(x))), idx, x.shape) in one pass; stats as in bn_fwd."""
    _chk(x)
    N, Hi, Wi, C = x.shape
    y = torch.empty((N, (Hi - 1) // 2 + 1, (Wi - 1) // 2 + 1, C), dtype=torch.float32, device=x.device)
    idx = torch.empty(y.shape, dtype=torch.uint8, device=x.device) if want_index else None
    flags = (1 if relu else 0) | (8 if stats_ready else 0) | (BN_DETERMINISTIC if deterministic() else 0)
    with _Timed("bn_maxpool_fwd" + (" N%d %dx%d C%d" % (N, Hi, Wi, C) if TIMING_DETAIL else ""), 0.0, 4.0 * (x.numel() + y.numel()) + y.numel()):
        check(lib.bh_bn_maxpool_fwd(_p(x), _p(gamma), _p(beta), _p(rmean), _p(rvar), _p(y), _p(idx), _p(stats), groups, N, Hi, Wi, C,
                                    float(eps), float(momentum), flags, 0 if training else 1, _p(amax), _stream()), "bh_bn_maxpool_fwd")
    if amax is not None:
        y._bh_amax = amax
    return BnPooled(y, idx, x.shape)


class BnJoinPending:
    """The lower-branch BatchNorm of a residual unit whose apply rides in the join (bn_join_fwd): `x` is its INPUT, `stats` its sums."""
    __slots__ = ("x", "stats", "mod")

    def __init__(self, x, stats, mod):
        self.x, self.stats, self.mod = x, stats, mod

    @property
    def shape(self):
        return self.x.shape


def bn_join_fwd(xa, xb, ma, mb, stats_a, stats_b, groups, relu, mom_a, mom_b, amax=None):
    """y = act(bn_a(xa) + bn_b(xb)) (training mode, both statistics tables already accumulated); ma / mb: the BatchNorm2d modules."""
    _chk(xa); _chk(xb); _chk(stats_a, torch.float64); _chk(stats_b, torch.float64)
    C = ma.num_features
    rows = xa.numel() // C // groups
    y = torch.empty_like(xa)
    flags = (1 if relu else 0) | (BN_DETERMINISTIC if deterministic() else 0)
    with _Timed("bn_join_fwd" + (" g%d rows%d C%d" % (groups, rows, C) if TIMING_DETAIL else ""), 0.0, 12.0 * xa.numel()):
        check(lib.bh_bn_join_fwd(_p(xa), _p(xb), _p(ma.weight), _p(ma.bias), _p(ma.running_mean), _p(ma.running_var), _p(mb.weight), _p(mb.bias),
                                 _p(mb.running_mean), _p(mb.running_var), _p(stats_a), _p(stats_b), _p(y), groups, rows, C, float(ma.eps),
                                 float(mb.eps), float(mom_a), float(mom_b), flags, _p(amax), _stream()), "bh_bn_join_fwd")
    if amax is not None:
        y._bh_amax = amax
    return y


def bn_join_bwd(gy, y, xa, xb, ma, mb, stats_a, stats_b, groups, relu, train_a, train_b, amax_a=None, amax_b=None):
    """-> (gxa, gxb); the affine parameters' gradients are added to .grad when train_a / train_b."""
    _chk(gy); _chk(xa); _chk(xb)
    C = ma.num_features
    rows = xa.numel() // C // groups
    gxa, gxb = torch.empty_like(xa), torch.empty_like(xb)
    scratch = torch.empty(lib.bh_bn_join_scratch_doubles(groups, C), dtype=torch.float64, device=xa.device)
    flags = (1 if relu else 0) | (BN_DETERMINISTIC if deterministic() else 0)
    # (round 5: the ReLU mask is recomputed from xa, xb - `y` is not read; BIHOME_JOIN_REMASK=0: the form that reads it)
    if os.environ.get("BIHOME_JOIN_REMASK", "1") != "0":
        with _Timed("bn_join_bwd(3 kernels)" + (" g%d rows%d C%d" % (groups, rows, C) if TIMING_DETAIL else ""), 0.0, 32.0 * xa.numel()):
            check(lib.bh_bn_join_bwd_remask(_p(gy), _p(xa), _p(xb), _p(ma.weight), _p(ma.bias), _p(mb.weight), _p(mb.bias), _p(stats_a),
                                            _p(stats_b), _p(gxa), _p(gxb),
                                            _p(ma.weight.grad) if train_a else None, _p(ma.bias.grad) if train_a else None,
                                            _p(mb.weight.grad) if train_b else None, _p(mb.bias.grad) if train_b else None, _p(scratch), groups,
                                            rows, C, float(ma.eps), float(mb.eps), flags, _p(amax_a), _p(amax_b), _stream()),
                  "bh_bn_join_bwd_remask")
    else:
      with _Timed("bn_join_bwd(3 kernels)" + (" g%d rows%d C%d" % (groups, rows, C) if TIMING_DETAIL else ""), 0.0, 40.0 * xa.numel()):
        check(lib.bh_bn_join_bwd(_p(gy), _p(y), _p(xa), _p(xb), _p(ma.weight), _p(mb.weight), _p(stats_a), _p(stats_b), _p(gxa), _p(gxb),
                                 _p(ma.weight.grad) if train_a else None, _p(ma.bias.grad) if train_a else None,
                                 _p(mb.weight.grad) if train_b else None, _p(mb.bias.grad) if train_b else None, _p(scratch), groups, rows, C,
                                 float(ma.eps), float(mb.eps), flags, _p(amax_a), _p(amax_b), _stream()), "bh_bn_join_bwd")
    if amax_a is not None:
        gxa._bh_amax = amax_a
    if amax_b is not None:
        gxb._bh_amax = amax_b
    return gxa, gxb


TAIL_ROUTE_VALU_FWD = 1     # include/bihome.h BH_TAIL_ROUTE_VALU_FWD
TAIL_ROUTE_LDS_MOMENTS = 2  # include/bihome.h BH_TAIL_ROUTE_LDS_MOMENTS


def tail_fwd(x, w1, b1, gamma, beta, rmean, rvar, w2, b2, groups, hw, eps, momentum, training, route=0):
    """x NHWC [N,h,w,Ci] -> out NCHW [N,Co,h,w] through conv1x1+BN+ReLU+conv1x1 (fused); returns (out, ws).
    route: per-call bits of bh_tail_fwd_route (TAIL_ROUTE_VALU_FWD keeps the per-pixel VALU kernel)."""
    _chk(x)
    N, h, w, Ci = x.shape
    Cm, Co = w1.shape[0], w2.shape[0]
    rows = N * h * w // groups
    out = torch.empty((N, Co, h, w), dtype=torch.float32, device=x.device)
    ws = torch.empty(lib.bh_tail_ws_doubles(groups, Ci, Cm), dtype=torch.float64, device=x.device)
    fl = 2.0 * N * h * w * Cm * (Ci + Co)
    with _Timed("tail_fwd(4 kernels)", fl, 4.0 * (x.numel() * (2 if training else 1) + out.numel())):
        check(lib.bh_tail_fwd_route(_p(x), _p(w1), _p(b1), _p(gamma), _p(beta), _p(rmean), _p(rvar), _p(w2), _p(b2), _p(out),
                                    _p(ws), groups, rows, hw, Ci, Cm, Co, float(eps), float(momentum), 0 if training else 1,
                                    int(route), _stream()), "bh_tail_fwd")
    return out, ws


def tail_bwd(gout, x, w1, b1, gamma, beta, w2, ws, rmean, rvar, groups, hw, eps, training, want_gx, gw1, ggamma, gbeta,
             gw2, gb2):
    _chk(gout); _chk(x)
    N, h, w, Ci = x.shape
    Cm, Co = w1.shape[0], w2.shape[0]
    rows = N * h * w // groups
    gx = torch.empty_like(x) if want_gx else None
    scratch = torch.empty(lib.bh_tail_scratch_floats(groups, Ci, Cm), dtype=torch.float32, device=x.device)
    fl = 2.0 * N * h * w * Cm * (3 * Ci + 2 * Co + Ci)
    with _Timed("tail_bwd(5 kernels)", fl, 4.0 * (x.numel() * 3 + gout.numel() * 3)):
        check(lib.bh_tail_bwd_f(_p(gout), _p(x), _p(w1), _p(b1), _p(gamma), _p(beta), _p(w2), _p(ws), _p(rmean), _p(rvar),
                                _p(gx), _p(gw1), _p(ggamma), _p(gbeta), _p(gw2), _p(gb2), _p(scratch), groups, rows, hw, Ci,
                                Cm, Co, float(eps), 0 if training else 1, _fdet(), _stream()), "bh_tail_bwd")
    return gx


def maxpool_fwd(x, want_index=True):
    _chk(x)
    N, Hi, Wi, C = x.shape
    y = torch.empty((N, (Hi - 1) // 2 + 1, (Wi - 1) // 2 + 1, C), dtype=torch.float32, device=x.device)
    idx = torch.empty(y.shape, dtype=torch.uint8, device=x.device) if want_index else None
    check(lib.bh_maxpool3s2_fwd(_p(x), _p(y), _p(idx), N, Hi, Wi, C, _stream()), "bh_maxpool3s2_fwd")
    rec = getattr(x, "_bh_amax", None)
    if rec is not None:
        y._bh_amax = rec            # a window maximum is bounded by the bound of its input
    return y, idx


class GradFrom1x1:
    """Gradient of a BatchNorm output that IS the dgrad of a 1x1 conv with few output channels: the conv's output gradient and its weight,
    handed to the BatchNorm's adjoint (bn_bwd_from_1x1) - the dgrad's result is never formed."""
    __slots__ = ("gs", "w")

    def __init__(self, gs, w):
        self.gs, self.w = gs, w


def bn_bwd_from_1x1(gf, x, gamma, beta, stats, groups, eps, relu, ggamma=None, gbeta=None, amax=None):
    """Adjoint of a training-mode BatchNorm (+ReLU, no residual) whose output feeds only a 1x1 conv: GradFrom1x1 + the BatchNorm input ->
    gx (bh_bn_bwd_from_1x1: conv_dgrad + bn_bwd without the tensor in between)."""
    gs, w = gf.gs, gf.w
    _chk(gs); _chk(w); _chk(x)
    C = x.shape[-1]
    KC = gs.shape[-1]
    rows = x.numel() // C // groups
    gx = torch.empty_like(x)
    scratch = torch.empty(lib.bh_bn_scratch_doubles(groups, C), dtype=torch.float64, device=x.device)
    flags = (1 if relu else 0) | (BN_DETERMINISTIC if deterministic() else 0)
    with _Timed("bn_bwd_from_1x1(3 kernels)" + (" g%d rows%d C%d<-%d" % (groups, rows, C, KC) if TIMING_DETAIL else ""), 4.0 * gs.numel() * C,
                4.0 * (3 * x.numel() + 2 * gs.numel())):
        check(lib.bh_bn_bwd_from_1x1(_p(gs), _p(w), KC, _p(x), _p(gamma), _p(beta), _p(stats), _p(gx), _p(ggamma), _p(gbeta), _p(scratch), groups,
                                     rows, C, float(eps), flags, _p(amax), _stream()), "bh_bn_bwd_from_1x1")
    if amax is not None:
        gx._bh_amax = amax
    return gx


class PooledGrad:
    """Gradient of a BnPooled slot: the pooled gradient and the arg-max positions, handed from the pooling op's adjoint to the BatchNorm's
    (bn_maxpool_bwd) - the full-resolution gradient in between is never formed."""
    __slots__ = ("gy", "idx")

    def __init__(self, gy, idx):
        self.gy, self.idx = gy, idx


def bn_maxpool_bwd(pg, x, gamma, beta, stats, rmean, rvar, groups, eps, relu, training, ggamma=None, gbeta=None, amax=None):
    """Adjoint of bn_maxpool_fwd: PooledGrad + the BatchNorm input -> gx (bh_bn_maxpool_bwd; bitwise what maxpool_bwd + bn_bwd give)."""
    gy, idx = pg.gy, pg.idx
    _chk(gy); _chk(idx, torch.uint8); _chk(x)
    N, Hi, Wi, C = x.shape
    gx = torch.empty_like(x)
    scratch = torch.empty(lib.bh_bn_scratch_doubles(groups, C), dtype=torch.float64, device=x.device)
    flags = (1 if relu else 0) | (BN_DETERMINISTIC if deterministic() else 0)
    with _Timed("bn_maxpool_bwd(3 kernels)" + (" g%d N%d %dx%d C%d" % (groups, N, Hi, Wi, C) if TIMING_DETAIL else ""), 0.0,
                4.0 * (3 * x.numel() + 2 * gy.numel()) + 2.0 * idx.numel()):
        check(lib.bh_bn_maxpool_bwd(_p(gy), _p(idx), _p(x), _p(gamma), _p(beta), _p(stats), _p(gx), _p(ggamma), _p(gbeta), _p(scratch), groups,
                                    N, Hi, Wi, C, float(eps), flags, 0 if training else 1, _p(rmean), _p(rvar), _p(amax), _stream()),
              "bh_bn_maxpool_bwd")
    if amax is not None:
        gx._bh_amax = amax
    return gx


def maxpool_bwd(idx, gy, in_shape):
    _chk(idx, torch.uint8); _chk(gy)
    N, Hi, Wi, C = in_shape
    gx = torch.empty(in_shape, dtype=torch.float32, device=gy.device)
    check(lib.bh_maxpool3s2_bwd(_p(idx), _p(gy), _p(gx), N, Hi, Wi, C, _stream()), "bh_maxpool3s2_bwd")
    return gx


def gap_fwd(x):
    _chk(x)
    N, H, W, C = x.shape
    y = torch.empty((N, 1, 1, C), dtype=torch.float32, device=x.device)
    check(lib.bh_gap_fwd(_p(x), _p(y), N, H * W, C, _stream()), "bh_gap_fwd")
    return y


def gap_bwd(gy, shape):
    _chk(gy)
    N, H, W, C = shape
    gx = torch.empty(shape, dtype=torch.float32, device=gy.device)
    check(lib.bh_gap_bwd(_p(gy), _p(gx), N, H * W, C, _stream()), "bh_gap_bwd")
    return gx


def add_(a, b):
    """a += b (same shape, contiguous)."""
    _chk(a); _chk(b)
    check(lib.bh_add(_p(a), _p(b), _p(a), a.numel(), _stream()), "bh_add")
    if getattr(a, "_bh_amax", None) is not None:
        a._bh_amax = None           # the magnitude record described the old contents
    return a
