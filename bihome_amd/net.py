"""Conv-stack executor: runs a module tree of Conv2d / ConvTranspose2d / BatchNorm2d / ReLU /
MaxPool2d leaves (the reference's `nn.Module` layout, kept only as the parameter container and
state-dict schema) as a flat program of HIP kernel launches on NHWC feature maps, forward and
backward, without autograd in between.

MI355X-first choices:
  * the reference runs the backbone once per direction (Rethinking.py:296-313) and the extractor once
    per patch (PerceptualHead.py:358-398); here those calls are stacked along the batch axis into ONE
    pass with `groups` independent BatchNorm statistics - half the launches, twice the GEMM M;
  * weights live in [Cout][kh][kw][Cin] order (torch channels_last), which is the K-contiguous B operand
    of the implicit GEMM, so no per-step repacking; gradients are written by the wgrad kernel straight
    into a flat fp32 buffer whose slices are the parameters' `.grad` views (one contiguous RCCL payload);
  * every launch goes to the current HIP stream with static shapes, so a whole step can be captured
    in a HIP graph.
"""
import os
import sys
import weakref

import torch
import torch.nn as nn

from . import kernels as K


# -----------------------------------------------------------------------------------------------
# parameter layout helpers
# -----------------------------------------------------------------------------------------------
def to_kernel_layout_(module):
    """Re-lay every conv weight in place as channels_last (values and state-dict shapes unchanged)."""
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            if not m.weight.data.is_contiguous(memory_format=torch.channels_last):
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    return module


def kview(w):
    """[O,I,kh,kw] channels_last parameter -> contiguous [O,kh,kw,I] view (no copy)."""
    if w.dim() == 2:
        return w
    v = w.permute(0, 2, 3, 1)
    if not v.is_contiguous():
        raise RuntimeError("conv weight is not in kernel (channels_last) layout; call net.to_kernel_layout_(module)")
    return v


class FlatGrads:
    """One flat fp32 buffer holding every trainable parameter's gradient; `.grad` of each parameter is a
    view into it laid out like the parameter (so wgrad kernels and the optimizer see the same bytes)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4          # keep every segment 16 B aligned
        self.numel = n
        self.flat = None
        self.views = None
        self.pflat = None             # round 4: the parameters themselves in a second flat buffer of the same layout (ensure_params)
        self.pviews = None

    def _views_of(self, flat):
        views = []
        for p, off in zip(self.params, self.offsets):
            seg = flat[off:off + p.numel()]
            if p.dim() == 4:
                O, I, kh, kw = p.shape
                v = seg.view(O, kh, kw, I).permute(0, 3, 1, 2)
            else:
                v = seg.view(p.shape)
            views.append(v)
        return views

    def ensure(self, device):
        if self.flat is None or self.flat.device != device:
            self.flat = torch.zeros(self.numel, dtype=torch.float32, device=device)
            self.views = self._views_of(self.flat)
        return self.flat

    def ensure_params(self, device):
        """The parameters as views into ONE flat buffer laid out like the gradient buffer, so that the optimizer updates one tensor
        instead of ~170 (step.build_optimizer: torch's fused Adam takes 265 us over the tensors of the Zeng backbone and 86 us over the
        same elements in one tensor).  Values are copied once; `p.data` of every parameter becomes its view (state_dict, load_state_dict,
        broadcast and the kernels see the same parameters as before).  Re-run (cheap pointer checks) before every optimizer step: a
        `model.to()`, a re-laid weight or a loader that REPLACES `p.data` is folded back in."""
        if self.pflat is not None and self.pflat.device == device and \
                all(p.data_ptr() == v.data_ptr() for p, v in zip(self.params, self.pviews)):
            return self.pflat
        pflat = torch.zeros(self.numel, dtype=torch.float32, device=device)
        pviews = self._views_of(pflat)
        with torch.no_grad():
            for p, v in zip(self.params, pviews):
                v.copy_(p.data)
                p.data = v
        self.pflat, self.pviews = pflat, pviews
        return pflat

    def attach(self, device):
        """Make sure every parameter's .grad is its view; zero the buffer if any was detached
        (optimizer.zero_grad(set_to_none=True) detaches them all - train.py:305)."""
        self.ensure(device)
        detached = False
        for p, v in zip(self.params, self.views):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                detached = True
                break
        if detached:
            self.flat.zero_()
            for p, v in zip(self.params, self.views):
                p.grad = v
        return self.flat


# -----------------------------------------------------------------------------------------------
# program
# -----------------------------------------------------------------------------------------------
class Op:
    __slots__ = ("kind", "src", "dst", "mod", "relu", "res", "extra")

    def __init__(self, kind, src, dst, mod=None, relu=False, res=None, extra=None):
        self.kind, self.src, self.dst, self.mod, self.relu, self.res, self.extra = kind, src, dst, mod, relu, res, extra


class Program:
    """Flat op list over numbered tensor slots. Slot 0 is the input."""

    def __init__(self):
        self.ops = []
        self.nslots = 1

    def _new(self):
        self.nslots += 1
        return self.nslots - 1

    def conv(self, src, mod, in_nchw=False, out_nchw=False, weight_fn=None):
        dst = self._new()
        self.ops.append(Op("conv", src, dst, mod, extra={"in_nchw": in_nchw, "out_nchw": out_nchw, "weight_fn": weight_fn}))
        return dst

    def bn(self, src, mod, relu=False, res=None):
        dst = self._new()
        self.ops.append(Op("bn", src, dst, mod, relu=relu, res=res))
        return dst

    def tail(self, src, conv1, bn, conv2):
        """Fused Conv1x1 + BatchNorm + ReLU + Conv1x1 -> NCHW (bh_tail_fwd/bwd)."""
        dst = self._new()
        self.ops.append(Op("tail", src, dst, (conv1, bn, conv2)))
        return dst

    def maxpool(self, src):
        dst = self._new()
        self.ops.append(Op("maxpool", src, dst))
        return dst

    def gap(self, src):
        dst = self._new()
        self.ops.append(Op("gap", src, dst))
        return dst

    # ---- builders for the reference's block types -------------------------------------------
    def sequential(self, src, seq):
        """nn.Sequential of Conv/ConvT/BN/ReLU/MaxPool leaves and residual blocks (in reference order)."""
        mods = list(seq)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, nn.Linear)):
                src = self.conv(src, m)
            elif isinstance(m, nn.BatchNorm2d):
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                src = self.bn(src, m, relu=relu)
                if relu:
                    i += 1
            elif isinstance(m, nn.MaxPool2d):
                src = self.maxpool(src)
            elif hasattr(m, "upper_branch"):
                src = self.residual(src, m)
            else:
                raise TypeError("unsupported leaf %r" % type(m))
            i += 1
        return src

    def residual(self, src, blk):
        """ReLU(upper_branch(x) + lower_branch(x) | x)  (src/backbones/utils.py:60-131): the last BN of the
        upper branch takes the lower result as fused residual input and applies the ReLU."""
        low = src
        if getattr(blk, "lower_branch", None) is not None and len(list(blk.lower_branch)) > 0:
            low = self.sequential(src, blk.lower_branch)
        up = list(blk.upper_branch)
        assert isinstance(up[-1], nn.BatchNorm2d)
        mid = self.sequential(src, up[:-1])
        return self.bn(mid, up[-1], relu=True, res=low)

    def basic_block(self, src, blk):
        """torchvision BasicBlock(conv1,bn1,relu,conv2,bn2,downsample)."""
        low = src
        if blk.downsample is not None:
            low = self.sequential(src, blk.downsample)
        t = self.bn(self.conv(src, blk.conv1), blk.bn1, relu=True)
        return self.bn(self.conv(t, blk.conv2), blk.bn2, relu=True, res=low)


class Ctx:
    """Saved tensors of one forward pass."""
    __slots__ = ("slots", "stats", "descs", "groups", "training", "weights", "wkeys", "wpacked", "precision", "joined")

    def __init__(self):
        self.slots, self.stats, self.descs, self.weights, self.wkeys, self.wpacked = {}, {}, {}, {}, {}, {}
        self.joined = {}          # join BatchNorm op index -> lower-branch BatchNorm op index (kernels.bn_join_fwd)


_GEOM_CACHE = {}
X3_WS_BYTES = 40 << 20          # Runner.x3_workspace: partial blocks of the f32x3 weight gradient
DET_WS_BYTES = 256 << 20        # deterministic mode: partial tiles / integer-limb shadow entries of ANY layer's weight gradient
                                # (32 bytes per weight: 75.5 MB for a 512 x 512 x 3 x 3 layer)


def _conv_geometry(mod, x_shape, in_nchw, out_nchw, precision=0):
    """Conv descriptor for `mod` on an input of `x_shape`, memoised together with what the library reports for it (halo
    3x3 kernel eligible for packed weights / for the fused BatchNorm reduce) and its packed-layout twin: building the
    ctypes struct and asking bh_conv_variant costs ~20 us of host time per launch otherwise, which is what bounds the
    shorter models (ResNet-34 regressor: 13.1 -> 15-17 ms/step when it was done per call)."""
    key = (id(mod), tuple(x_shape), bool(in_nchw), bool(out_nchw), int(precision), K.deterministic())   # (the workspace bytes depend on the mode)
    hit = _GEOM_CACHE.get(key)
    if hit is not None and hit[0] is mod:
        return hit[1]
    d = _conv_geometry_uncached(mod, x_shape, in_nchw, out_nchw, precision)
    d.bh_packs = bool(isinstance(mod, nn.Conv2d) and K.packs_3x3(d))
    d.bh_reduce_ok = bool(isinstance(mod, nn.Conv2d) and K.dgrad_bn_reduce_ok(d))
    d.bh_packed = K._with_layout(d, K.packed_layout(precision)) if d.bh_packs else None
    # f32x3 weight gradient (csrc/wgrad_x3.hip): reduces its split-K partial blocks through a workspace (fixed order, cheaper than atomics)
    d.bh_wx3 = bool(int(precision) in K.SPLIT_PIECES and isinstance(mod, nn.Conv2d) and K.conv_variant(d, "wgrad").startswith("wgrad_x3"))
    d.bh_wx3_bytes = K.wgrad_det_bytes(d) if d.bh_wx3 else 0
    _GEOM_CACHE[key] = (mod, d)
    return d


def _conv_geometry_uncached(mod, x_shape, in_nchw, out_nchw, precision=0):
    if in_nchw:
        N, Ci, Hi, Wi = x_shape
    else:
        N, Hi, Wi, Ci = x_shape
    if isinstance(mod, nn.Linear):
        return K.conv_desc(N, Hi, Wi, Ci, mod.out_features, 1, 1, 0, precision=precision)
    k, s, p = mod.kernel_size[0], mod.stride[0], mod.padding[0]
    tr = isinstance(mod, nn.ConvTranspose2d)
    Co = mod.out_channels
    return K.conv_desc(N, Hi, Wi, Ci, Co, k, s, p, transposed=tr, in_nchw=in_nchw, out_nchw=out_nchw, precision=precision)


def _folded(cache, conv, bn, weight_fn):
    """Eval-mode BatchNorm folded into the preceding conv: w' = w * s[co], b' = b * s + t with s = gamma / sqrt(var + eps),
    t = beta - mean * s (running statistics).  Cached per (conv, bn) pair until a parameter / buffer version changes."""
    key = (id(conv), id(bn))
    vers = tuple(t._version for t in (conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)
                 if t is not None)
    hit = cache.get(key)
    if hit is not None and hit[0] == vers:
        return hit[1], hit[2]
    with torch.no_grad():
        s = torch.rsqrt(bn.running_var + bn.eps)
        if bn.weight is not None:
            s = s * bn.weight
        t = -bn.running_mean * s
        if bn.bias is not None:
            t = t + bn.bias
        w = weight_fn(conv.weight) if weight_fn else conv.weight
        if isinstance(conv, nn.ConvTranspose2d):                  # [Ci, Co, kh, kw]
            wf = (w * s.view(1, -1, 1, 1)).contiguous(memory_format=torch.channels_last)
        elif w.dim() == 4:
            wf = (w * s.view(-1, 1, 1, 1)).contiguous(memory_format=torch.channels_last)
        else:
            wf = (w * s.view(-1, 1)).contiguous()
        bf = (conv.bias * s + t) if conv.bias is not None else t
        bf = bf.contiguous()
    cache[key] = (vers, wf, bf)
    return wf, bf


def _momentum(bn):
    """torch's momentum=None (cumulative moving average) is not built: the kernels replay one exponential update per
    statistics group.  No reference module uses it (every BatchNorm2d upstream keeps the default 0.1)."""
    if bn.momentum is None:
        raise NotImplementedError("BatchNorm2d(momentum=None) (cumulative average) is not supported by bihome_amd")
    return bn.momentum


def _zero_arenas(n_doubles, n_floats, device):
    """The two zeroed scratch arenas of a pass - float64 sums and float32 magnitude records - out of ONE zero-filled byte buffer (one fill
    launch instead of two, five times per training step).  Either may be empty (None)."""
    if not (n_doubles or n_floats):
        return None, None
    buf = torch.zeros(n_doubles * 8 + n_floats * 4, dtype=torch.uint8, device=device)
    a = buf[:n_doubles * 8].view(torch.float64) if n_doubles else None
    b = buf[n_doubles * 8:].view(torch.float32) if n_floats else None
    return a, b


def run_forward(prog, x, groups, training, save, precision=0, fold_cache=None, packer=None, input_source=None):
    """packer: kernels.WeightPacker holding fragment-ordered copies of the 3x3 weights (refreshed here, one launch, when
    a parameter changed): the halo-tiled 3x3 kernel then streams its B operand straight into registers.
    x: NHWC (or NCHW when the first conv is flagged in_nchw). Returns (out, ctx|None).
    fold_cache (inference only: not training, nothing saved): a dict - every conv whose only consumer is a BatchNorm
    runs with that BatchNorm folded into its weights and the ReLU / residual add fused into its epilogue.
    input_source (round 6): x is an unfilled buffer - the homography warp of input_source["src"], which the conv that reads the input
    makes on the way (kernels.conv_fwd warp_src) or has made in front of it."""
    ctx = Ctx() if save else None
    if int(precision) == K.F16X2 and (packer is None or not packer.f16):
        # the fp16-piece kernels exist for packed weights with magnitude records of their operands (the BatchNorm kernels of a
        # training-style pass leave them); the BatchNorm-folded inference pass runs the exact three-piece arithmetic
        precision = 2
    if packer is not None:
        # a forward that saves for backward is (probably) followed by an optimizer step, whatever module.training says
        # (frozen-BatchNorm fine-tuning runs the backbone in eval() mode): fused optimizers do not bump version counters
        packer.refresh(training or save)
    slots = {0: x}
    if save:
        ctx.groups, ctx.training, ctx.precision = groups, training, int(precision)
    # precision 4: one zeroed arena of magnitude records, one per BatchNorm output (the operands of the fp16-piece 3x3 kernels)
    amax_next = None
    nrec = 0
    if int(precision) == K.F16X2:
        nrec = max(1, sum(1 for op in prog.ops if op.kind == "bn" or (op.kind == "conv" and isinstance(op.mod, nn.ConvTranspose2d))))
    # BatchNorm sums: one zeroed float64 arena for the whole pass (the kernels accumulate with atomics); a conv whose
    # only consumer is a training-mode BatchNorm accumulates that layer's statistics in its own epilogue
    bn_off, total = {}, 0
    for i, op in enumerate(prog.ops):
        if op.kind == "bn":
            bn_off[i] = total
            total += K.bn_stats_doubles(groups, op.mod.num_features)
    arena, amax_arena = _zero_arenas(total, nrec * K.AMAX_FLOATS, x.device)          # (one fill launch for both)
    if nrec:
        amax_iter = iter(amax_arena.split(K.AMAX_FLOATS))
        amax_next = lambda: next(amax_iter)
    # the fusion plan below (which conv feeds which BatchNorm's sums, which BatchNorms are applied on load or joined) depends on the program, the
    # mode and the weight packer only: computed once per (mode, groups, arithmetic, packer) and kept on the program, as run_backward's
    # (the packer is named by its serial number, not by id(): an id can be reused by another packer after garbage collection)
    fw_key = (bool(training), int(groups), int(precision), packer.serial if packer is not None else 0,
              len(packer.entries) if packer is not None else 0, bool(packer.f16) if packer is not None else False,
              os.environ.get("BIHOME_BN_ON_LOAD_1X1", "1"), os.environ.get("BIHOME_BN_ON_LOAD", "1"), os.environ.get("BIHOME_BN_JOIN", "1"),
              tuple(op.mod.weight.requires_grad for op in prog.ops if op.kind == "conv"))
    fw_plans = prog.__dict__.setdefault("_fw_plans", {})
    fw_cached = fw_plans.get(fw_key) if os.environ.get("BIHOME_PLAN_CACHE", "1") != "0" else None
    consumer = None
    if fw_cached is None:
        fused_stats = {}                                      # conv op index -> bn op index
        if training:
            users = {}
            for op in prog.ops:
                users[op.src] = users.get(op.src, 0) + 1
                if op.res is not None:
                    users[op.res] = users.get(op.res, 0) + 1
            producer = {op.dst: j for j, op in enumerate(prog.ops)}
            for i, op in enumerate(prog.ops):
                j = producer.get(op.src)
                if (op.kind == "bn" and j is not None and prog.ops[j].kind == "conv" and users.get(op.src, 0) == 1
                        and not prog.ops[j].extra["out_nchw"] and prog.ops[j].mod.weight.dim() == 4):
                    fused_stats[j] = i
        # BatchNorm-on-load: a training-mode BatchNorm(+ReLU) without residual whose ONLY consumer is a 3x3 conv that runs the packed
        # f32x3 forward and the f32x3 weight gradient is not applied at all - the consumer transforms the BatchNorm's INPUT while
        # staging it (kernels.BnOnLoad): one launch and two tensor passes per such layer gone (the inner BatchNorm of every
        # residual unit).  Needs the sums from the producer's epilogue (fused_stats) and a table of <= 4 KB.
        bn_on_load = set()
        bn_on_load_1x1 = set()
        if training and os.environ.get("BIHOME_BN_ON_LOAD_1X1", "1") != "0" and groups <= 2 and int(precision) != 1:      # (not the bf16-operand mode)
            # round 4: the same for a 1x1 / stride-1 conv consumer with <= 32 channels on one side (the 1x1 conv of the decoder units behind
            # BatchNorm + ReLU at 64 x 64 and 128 x 128): generic forward kernel and small-channel weight-gradient kernel transform on load
            consumer_ = {}
            for j, op in enumerate(prog.ops):
                consumer_.setdefault(op.src, j)
            fused_bn_ = set(fused_stats.values())
            for i, op in enumerate(prog.ops):
                j = consumer_.get(op.dst)
                if (op.kind == "bn" and op.res is None and i in fused_bn_ and users.get(op.dst, 0) == 1 and j is not None
                        and prog.ops[j].kind == "conv" and prog.ops[j].src == op.dst and prog.ops[j].extra["weight_fn"] is None
                        and not prog.ops[j].extra["in_nchw"] and not prog.ops[j].extra["out_nchw"] and isinstance(prog.ops[j].mod, nn.Conv2d)
                        and prog.ops[j].mod.kernel_size == (1, 1) and prog.ops[j].mod.stride == (1, 1) and prog.ops[j].mod.padding == (0, 0)
                        and prog.ops[j].mod.in_channels % 32 == 0 and prog.ops[j].mod.out_channels % 4 == 0
                        and min(prog.ops[j].mod.in_channels, prog.ops[j].mod.out_channels) <= 32 and prog.ops[j].mod.weight.requires_grad):
                    bn_on_load_1x1.add(i)
        if training and packer is not None and int(precision) in K.SPLIT_PIECES and os.environ.get("BIHOME_BN_ON_LOAD", "1") != "0":
            consumer = {}
            for j, op in enumerate(prog.ops):
                consumer.setdefault(op.src, j)
            fused_bn = set(fused_stats.values())
            for i, op in enumerate(prog.ops):
                j = consumer.get(op.dst)
                if (op.kind == "bn" and op.res is None and i in fused_bn and users.get(op.dst, 0) == 1 and j is not None
                        and prog.ops[j].kind == "conv" and prog.ops[j].src == op.dst and prog.ops[j].extra["weight_fn"] is None
                        and not prog.ops[j].extra["in_nchw"] and id(prog.ops[j].mod.weight) in packer.entries
                        and groups * op.mod.num_features * 8 <= 4096):
                    bn_on_load.add(i)
        # Two-branch join (round 4): a training-mode BatchNorm whose residual input is itself the output of a training-mode BatchNorm that
        # nobody else reads (the lower branch of ResNet50DeconvBlock / the strided ResNet34ConvBlock): the lower BatchNorm is not applied
        # on its own - both are applied, added and rectified in ONE pass (kernels.bn_join_fwd), the adjoint is one reduce + one apply
        joins = {}                                            # join bn op index -> lower bn op index
        if training and os.environ.get("BIHOME_BN_JOIN", "1") != "0":
            producer_ = {op.dst: j for j, op in enumerate(prog.ops)}
            for i, op in enumerate(prog.ops):
                j = producer_.get(op.res) if (op.kind == "bn" and op.res is not None) else None
                if (j is not None and prog.ops[j].kind == "bn" and prog.ops[j].res is None and not prog.ops[j].relu
                        and users.get(op.res, 0) == 1 and j not in bn_on_load and i not in bn_on_load
                        and op.mod.num_features == prog.ops[j].mod.num_features and op.mod.num_features % 4 == 0
                        and op.mod.weight is not None and prog.ops[j].mod.weight is not None):
                    joins[i] = j
        join_lower = set(joins.values())
        if len(fw_plans) > 16:
            fw_plans.clear()
        fw_plans[fw_key] = (fused_stats, bn_on_load, bn_on_load_1x1, joins, join_lower, consumer)
    else:
        fused_stats, bn_on_load, bn_on_load_1x1, joins, join_lower, consumer = fw_cached
    folded = {}                                           # bn op index -> conv op index (conv deferred to the bn's position)
    if fold_cache is not None and not training and not save:
        users = {}
        for op in prog.ops:
            users[op.src] = users.get(op.src, 0) + 1
            if op.res is not None:
                users[op.res] = users.get(op.res, 0) + 1
        producer = {op.dst: j for j, op in enumerate(prog.ops)}
        for i, op in enumerate(prog.ops):
            j = producer.get(op.src)
            if (op.kind == "bn" and j is not None and prog.ops[j].kind == "conv" and users.get(op.src, 0) == 1
                    and not prog.ops[j].extra["out_nchw"]):
                folded[i] = j
    deferred = set(folded.values())
    ready = set()
    for i, op in enumerate(prog.ops):
        if i in deferred:
            continue
        src = slots.get(op.src)
        if i in folded:
            cop = prog.ops[folded[i]]
            e = cop.extra
            csrc = slots[cop.src]
            d = _conv_geometry(cop.mod, csrc.shape, e["in_nchw"], e["out_nchw"], precision)
            wf, bf = _folded(fold_cache, cop.mod, op.mod, e["weight_fn"])
            pf = None
            if int(precision) in K.SPLIT_PIECES and d.bh_packs and e["weight_fn"] is None:
                # f32x3 inference: the folded weights in cut fragment order (packed once per fold)
                ent = fold_cache[(id(cop.mod), id(op.mod))]
                if len(ent) < 4 or ent[3] is None:
                    pk = K.packer_for_precision(precision)
                    pf, _ = pk.get(wf, need_dgrad=False)
                    pk.refresh()
                    fold_cache[(id(cop.mod), id(op.mod))] = (ent[0], ent[1], ent[2], pf)
                else:
                    pf = ent[3]
            out = K.conv_fwd(csrc, kview(wf), bf, d, res=slots[op.res] if op.res is not None else None, relu=op.relu, wpacked=pf,
                             warp_src=input_source if cop.src == 0 else None)
        elif op.kind == "conv":
            e = op.extra
            d = _conv_geometry(op.mod, src.shape, e["in_nchw"], e["out_nchw"], precision)
            w = e["weight_fn"](op.mod.weight) if e["weight_fn"] else op.mod.weight
            wk = kview(w)
            pk = None
            if packer is not None and e["weight_fn"] is None and d.bh_packs and id(op.mod.weight) in packer.entries:
                pk = packer.entries[id(op.mod.weight)]
            wsrc = input_source if op.src == 0 else None
            if i in fused_stats and d.N % groups == 0:
                b = fused_stats[i]
                out = K.conv_fwd(src, wk, op.mod.bias, d,
                                 bn_sums=arena[bn_off[b]:bn_off[b] + K.bn_stats_doubles(groups, d.Co)], groups=groups,
                                 wpacked=pk[1] if pk else None, warp_src=wsrc)
                ready.add(b)
            elif amax_next and pk is None and isinstance(op.mod, nn.ConvTranspose2d) and not e["out_nchw"]:
                # a transposed conv in front of a packed fp16-piece 3x3 conv (the decoder units): the magnitude record of its output from
                # its own epilogue instead of a bh_absmax pass over the (up to 268 MB) tensor
                out = K.conv_fwd(src, wk, op.mod.bias, d, amax=amax_next(), warp_src=wsrc)
            else:
                out = K.conv_fwd(src, wk, op.mod.bias, d, wpacked=pk[1] if pk else None, warp_src=wsrc)
            if save:
                ctx.descs[i], ctx.weights[i] = d, wk
                ctx.wpacked[i] = pk[2] if pk else None
                ctx.wkeys[i] = (op.mod.weight, op.mod.weight._version)
        elif op.kind == "bn":
            m = op.mod
            res = slots[op.res] if op.res is not None else None
            lazy = False
            if i in join_lower and src.shape[0] % groups == 0:
                # the lower branch of a join: statistics only (from the producer's epilogue, else one pass); applied inside the join
                st = arena[bn_off[i]:bn_off[i] + K.bn_stats_doubles(groups, m.num_features)]
                if i not in ready:
                    K.bn_stats(src, st, groups, m.num_features)
                m._bh_pending_batches = getattr(m, "_bh_pending_batches", 0) + groups
                if save:
                    ctx.stats[i] = st
                slots[op.dst] = K.BnJoinPending(src, st, m)
                continue
            if i in joins and isinstance(res, K.BnJoinPending) and src.shape[0] % groups == 0:
                st = arena[bn_off[i]:bn_off[i] + K.bn_stats_doubles(groups, m.num_features)]
                if i not in ready:
                    K.bn_stats(src, st, groups, m.num_features)
                out = K.bn_join_fwd(src, res.x, m, res.mod, st, res.stats, groups, op.relu, _momentum(m), _momentum(res.mod),
                                    amax=amax_next() if amax_next else None)
                m._bh_pending_batches = getattr(m, "_bh_pending_batches", 0) + groups
                if save:
                    ctx.stats[i] = st
                    ctx.joined[i] = joins[i]
                slots[op.dst] = out
                continue
            if isinstance(res, K.BnJoinPending):          # (a join that could not be formed after all: apply the lower BatchNorm now)
                lo, _ = K.bn_fwd(res.x, res.mod.weight, res.mod.bias, res.mod.running_mean, res.mod.running_var, None, groups, res.mod.eps,
                                 _momentum(res.mod), False, training, stats=res.stats, stats_ready=True)
                res = slots[op.res] = lo
            nxt = prog.ops[i + 1] if i + 1 < len(prog.ops) else None
            if (res is None and nxt is not None and nxt.kind == "maxpool" and nxt.src == op.dst and src.dim() == 4 and src.shape[0] % groups == 0
                    and m.num_features % 4 == 0 and m.num_features > 1 and sum(1 for o_ in prog.ops if o_.src == op.dst or o_.res == op.dst) == 1
                    and os.environ.get("BIHOME_BN_POOL", "1") != "0"):
                # BatchNorm (+ReLU) -> MaxPool2d(3, 2, 1), its only consumer: one pass, the activation in between is never stored
                st = arena[bn_off[i]:bn_off[i] + K.bn_stats_doubles(groups, m.num_features)]
                slots[op.dst] = K.bn_maxpool_fwd(src, m.weight, m.bias, m.running_mean, m.running_var, groups, m.eps, _momentum(m), op.relu,
                                                 training, st, i in ready, want_index=save, amax=amax_next() if amax_next else None)
                if training:
                    m._bh_pending_batches = getattr(m, "_bh_pending_batches", 0) + groups
                if save:
                    ctx.stats[i] = st
                continue
            if i in bn_on_load_1x1 and i in ready and src.shape[0] % groups == 0 and (src.numel() // (m.num_features * groups)) % 128 == 0:
                lazy = True
            elif i in bn_on_load and i in ready and src.shape[0] % groups == 0:
                cop = prog.ops[consumer[op.dst]]
                cd = _conv_geometry(cop.mod, src.shape, False, cop.extra["out_nchw"], precision)
                # (the consumer's weight gradient must fit the fixed f32x3 workspace, else its backward could not take the
                #  un-materialised operand: decided here, before the BatchNorm output is elided)
                lazy = bool(cd.bh_packs and cd.bh_wx3 and 0 < cd.bh_wx3_bytes <= (DET_WS_BYTES if K.deterministic() else X3_WS_BYTES))
            if lazy:
                st = arena[bn_off[i]:bn_off[i] + K.bn_stats_doubles(groups, m.num_features)]
                rec = amax_next() if amax_next else None
                table = K.bn_fwd_coeffs(st, m.weight, m.bias, m.running_mean, m.running_var, groups,
                                        src.numel() // (m.num_features * groups), m.num_features, m.eps, _momentum(m), amax=rec)
                out = K.BnOnLoad(src, table, groups, op.relu, amax=rec)
            else:
                out, st = K.bn_fwd(src, m.weight, m.bias, m.running_mean, m.running_var, res, groups, m.eps,
                                   _momentum(m), op.relu, training,
                                   stats=arena[bn_off[i]:bn_off[i] + K.bn_stats_doubles(groups, m.num_features)],
                                   stats_ready=i in ready, amax=amax_next() if (amax_next and m.num_features > 1) else None)
            if training:        # flushed to the `num_batches_tracked` buffer lazily (flush_counters): no per-layer launch
                m._bh_pending_batches = getattr(m, "_bh_pending_batches", 0) + groups
            if save:
                ctx.stats[i] = st
        elif op.kind == "tail":
            c1, bn, c2 = op.mod
            N, h, w, _ = src.shape
            out, ws = K.tail_fwd(src, kview(c1.weight), c1.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                 kview(c2.weight), c2.bias, groups, h * w, bn.eps,
                                 _momentum(bn), training)
            if training:
                bn._bh_pending_batches = getattr(bn, "_bh_pending_batches", 0) + groups
            if save:
                ctx.stats[i] = ws
        elif op.kind == "maxpool":
            if isinstance(src, K.BnPooled):               # pooled by the fused BatchNorm kernel already
                out, idx = src.pooled, src.idx
            else:
                out, idx = K.maxpool_fwd(src, want_index=save)
            if save:
                ctx.stats[i] = idx
        elif op.kind == "gap":
            out = K.gap_fwd(src)
        else:
            raise RuntimeError(op.kind)
        slots[op.dst] = out
    if save:
        ctx.slots = slots
    return slots[prog.ops[-1].dst], ctx


_SIDE_STREAMS = {}


def side_stream(device):
    """The per-device HIP stream that carries the weight-gradient GEMMs (and the head's feature prefetch): they have no
    consumer inside the backward chain, so they run under the dgrad / BatchNorm kernels of the main stream."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        # (BIHOME_SIDE_PRIORITY: experiments - the stream's priority; lower number = higher priority, out-of-range values are clamped)
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=key, priority=int(os.environ.get("BIHOME_SIDE_PRIORITY", "0")))
    return _SIDE_STREAMS[key]


_FROM_1X1_KC = (16,)      # output channels of the 1x1 convs whose dgrad the BatchNorm in front rebuilds (run_backward)


def run_backward(prog, ctx, gout, want_wgrad, want_input_grad, on_param_grad=None, wgrad_stream=None, det_ws=None, x3_ws=None, input_sink=None):
    """Adjoint of run_forward. Parameter gradients are accumulated (+=) into each parameter's `.grad`
    tensor (which must already exist, see FlatGrads). Returns the input gradient or None.
    wgrad_stream: a second HIP stream for the conv weight-gradient launches.  The backward chain on the main stream is
    dgrad -> BatchNorm backward -> dgrad ...: the wgrad of a layer only needs that layer's output gradient, so it is
    enqueued behind an event and overlaps the chain (MFMA-bound wgrad under the HBM-bound BatchNorm kernels); the
    main stream joins the side stream before returning."""
    main = torch.cuda.current_stream() if wgrad_stream is not None else None
    if wgrad_stream is not None:
        wgrad_stream.wait_stream(main)          # activations, zeroed gradient buffer
    grads = {prog.ops[-1].dst: gout}
    slots = ctx.slots
    # (the fusion plan below is a pure function of the program, the mode and the shapes: made once and kept on the program - three
    #  backward walks per step used to rebuild it, ~0.4 ms of a host that has ~9 ms of enqueueing to do per 12 ms step)
    # (key: everything the plan reads that can differ between two walks of one program - mode, batch / map geometry via the output
    #  gradient's shape, the joins made by this forward, the precision, the switches)
    #  gradient's shape, the joins made by this forward and the convs it ran - their CONTENTS, not their counts - the precision, which conv
    #  weights / biases train (fuse_bias reads the flags: freezing a layer between two steps must not reuse the other plan), the switches)
    plan_key = (bool(ctx.training), bool(want_wgrad), int(ctx.groups), tuple(gout.shape), frozenset(ctx.joined.items()), frozenset(ctx.descs),
                getattr(ctx, "precision", 0),
                tuple((op.mod.weight.requires_grad, op.mod.bias is not None and op.mod.bias.requires_grad) for op in prog.ops if op.kind == "conv"),
                os.environ.get("BIHOME_FUSE_BN_REDUCE", "1"), os.environ.get("BIHOME_FUSE_BIAS_GRAD", "1"), os.environ.get("BIHOME_BN_FROM_1X1", "1"),
                os.environ.get("BIHOME_WGRAD_BNADJ", "0"))
    plans = prog.__dict__.setdefault("_bw_plans", {})
    cached = plans.get(plan_key) if os.environ.get("BIHOME_PLAN_CACHE", "1") != "0" else None
    consumed_by = {}
    for op in (prog.ops if cached is None else ()):
        consumed_by.setdefault(op.src, 0)
        consumed_by[op.src] += 1
        if op.res is not None:
            consumed_by[op.res] = consumed_by.get(op.res, 0) + 1

    def contribute(slot, g):
        if slot in grads:
            K.add_(grads[slot], g)
        else:
            grads[slot] = g

    # A conv dgrad that COMPLETES the gradient of a training-mode BatchNorm's output (it is the last consumer of that
    # slot in backward order) also accumulates that BatchNorm's backward sums in its epilogue (bh_conv_dgrad_bnreduce),
    # so the BatchNorm adjoint is one apply launch instead of reduce + finalize + apply.
    fuse_bn = {}                                          # conv op index -> bn op index
    if cached is None and ctx.training and os.environ.get("BIHOME_FUSE_BN_REDUCE", "1") != "0":
        producer = {op.dst: j for j, op in enumerate(prog.ops)}
        last_consumer = {}
        for j, op in enumerate(prog.ops):                 # the lowest-index consumer is processed last
            for sl in (op.src, op.res):
                if sl is not None and sl not in last_consumer:
                    last_consumer[sl] = j
        for j, op in enumerate(prog.ops):
            b = producer.get(op.src)
            if (op.kind == "conv" and b is not None and prog.ops[b].kind == "bn" and last_consumer.get(op.src) == j
                    and j in ctx.descs and ctx.descs[j].bh_reduce_ok and ctx.descs[j].N % ctx.groups == 0 and b not in ctx.joined):
                fuse_bn[j] = b
    # A 3x3 conv that is the ONLY consumer of a biased (transposed) conv's output: the column sums of its input gradient
    # are that layer's bias gradient - accumulated in the dgrad epilogue instead of a streaming pass over the gradient
    fuse_bias = {}                                        # conv op index -> producer op index
    if cached is None and want_wgrad and os.environ.get("BIHOME_FUSE_BIAS_GRAD", "1") != "0":
        producer = {op.dst: j for j, op in enumerate(prog.ops)}
        for j, op in enumerate(prog.ops):
            p = producer.get(op.src)
            if (op.kind == "conv" and j not in fuse_bn and p is not None and prog.ops[p].kind == "conv" and consumed_by.get(op.src, 0) == 1
                    and j in ctx.descs and p in ctx.descs and ctx.descs[j].bh_reduce_ok):
                pm = prog.ops[p].mod
                if pm.bias is not None and pm.bias.requires_grad and pm.weight.requires_grad and prog.ops[p].extra["weight_fn"] is None:
                    fuse_bias[j] = p
    # A 1x1 / stride-1 conv with 16 output channels that is the ONLY consumer of a training-mode BatchNorm (+ReLU, no residual, not joined):
    # its dgrad is rebuilt inside that BatchNorm's adjoint (bh_bn_bwd_from_1x1) - the full-resolution decoder unit's 268 MB gradient is
    # never written
    from_1x1 = set()
    if cached is None and ctx.training and os.environ.get("BIHOME_BN_FROM_1X1", "1") != "0":
        producer_ = {op.dst: j for j, op in enumerate(prog.ops)}
        for j, op in enumerate(prog.ops):
            b = producer_.get(op.src)
            m_ = op.mod
            if (op.kind == "conv" and b is not None and prog.ops[b].kind == "bn" and prog.ops[b].res is None and b not in ctx.joined
                    and consumed_by.get(op.src, 0) == 1 and j not in fuse_bn and isinstance(m_, nn.Conv2d) and m_.kernel_size == (1, 1)
                    and m_.stride == (1, 1) and m_.padding == (0, 0) and op.extra["weight_fn"] is None and m_.out_channels in _FROM_1X1_KC
                    and prog.ops[b].mod.num_features % 4 == 0 and 256 % (prog.ops[b].mod.num_features // 4) == 0
                    and not op.extra["in_nchw"] and not op.extra["out_nchw"]):
                from_1x1.add(j)
    # Round 6 (round-5 VERDICT item 2): a fused BatchNorm whose input comes from a 3x3 conv with an fp16-piece weight gradient - that
    # weight gradient takes the BatchNorm's adjoint ON LOAD (kernels.conv_wgrad_bnadj): it reads the gradient of the BatchNorm's output,
    # the BatchNorm's input and the sums the dgrad epilogue made, so it is enqueued IN FRONT of the BatchNorm's adjoint pass and runs next
    # to that HBM-bound pass instead of behind it.  bn op index -> conv op index
    # MEASURED AND NOT ADOPTED (profiles/r06g_wx3_bnadj_ab.txt, r06g_step_ab_bnadj.txt): the second operand stream and its transform cost
    # the weight-gradient kernel 6-15 us per launch alone (four waves 45.9 -> 61.4 us on 32 x 32 x 64, eight waves 43.8 -> 49.8) - as much as
    # the 12-22 us adjoint pass it no longer waits for - and the step gets 0.27 ms SLOWER (13.34 against 13.07 ms, three alternating runs on
    # one box).  The form stays behind BIHOME_WGRAD_BNADJ=1 (C ABI entry, parity test tests/test_f16x2_gpu.py); the default is off.
    bnadj = {}
    if (cached is None and want_wgrad and ctx.training and getattr(ctx, "precision", 0) == K.F16X2
            and os.environ.get("BIHOME_WGRAD_BNADJ", "0") != "0"):
        producer_c = {op.dst: j for j, op in enumerate(prog.ops)}
        for b in fuse_bn.values():
            c = producer_c.get(prog.ops[b].src)
            if c is None or prog.ops[c].kind != "conv" or c not in ctx.descs or consumed_by.get(prog.ops[b].src, 0) != 1:
                continue
            cm, cd = prog.ops[c].mod, ctx.descs[c]
            if (isinstance(cm, nn.Conv2d) and cm.kernel_size == (3, 3) and cm.stride == (1, 1) and cm.padding == (1, 1) and cm.weight.requires_grad
                    and prog.ops[c].extra["weight_fn"] is None and not (cm.bias is not None and cm.bias.requires_grad)
                    and getattr(cd, "bh_wx3", False) and cd.Ci % 64 == 0 and cd.Co % 64 == 0 and cd.Hi % 8 == 0 and cd.Wi % 8 == 0
                    and cd.N % ctx.groups == 0):
                bnadj[b] = c
    red_off, total = {}, 0
    bias_off = {}
    if cached is None:
        for j, p in fuse_bias.items():
            bias_off[p] = total
            total += K.bn_stats_doubles(1, ctx.descs[j].Ci)
        for b in fuse_bn.values():
            red_off[b] = total
            total += K.bn_stats_doubles(ctx.groups, prog.ops[b].mod.num_features)
        plans[plan_key] = (consumed_by, fuse_bn, fuse_bias, from_1x1, red_off, bias_off, total, bnadj)
    else:
        consumed_by, fuse_bn, fuse_bias, from_1x1, red_off, bias_off, total, bnadj = cached
    if os.environ.get("BIHOME_BN_PLAN") == "1" and not getattr(prog, "_bn_plan_printed", False):
        # tools/: which BatchNorm adjoints still run their own reduce pass, and what completes their output gradient
        prog._bn_plan_printed = True
        lc = {}
        for j, op in enumerate(prog.ops):
            for sl in (op.src, op.res):
                if sl is not None and sl not in lc:
                    lc[sl] = j
        for b, op in enumerate(prog.ops):
            if op.kind != "bn":
                continue
            j = lc.get(op.dst)
            cons = prog.ops[j] if j is not None else None
            what = "output" if cons is None else cons.kind
            if cons is not None and cons.kind == "conv":
                m = cons.mod
                what = "%s k%s s%s %d->%d" % (type(m).__name__, getattr(m, "kernel_size", "?"), getattr(m, "stride", "?"),
                                               getattr(m, "in_channels", getattr(m, "in_features", 0)), getattr(m, "out_channels", getattr(m, "out_features", 0)))
            elif cons is not None and cons.kind == "bn":
                what = "bn(res)" if cons.res == op.dst else "bn"
            print("BN_PLAN op %3d C%-4d %s last consumer: op %s %s | fused=%s joined=%s shape=%s" %
                  (b, op.mod.num_features, "relu" if op.relu else "    ", j, what, b in fuse_bn.values(), b in ctx.joined,
                   tuple(slots[op.src].shape) if hasattr(slots[op.src], "shape") else "?"), file=sys.stderr)
    bn_reduced = {}
    # precision 4: magnitude records of the BatchNorm input gradients (the gy operand of the fp16-piece dgrad / weight-gradient kernels)
    amax_next = None
    nrec = (max(1, sum(1 for op in prog.ops if op.kind == "bn")) + len(bnadj)) if getattr(ctx, "precision", 0) == K.F16X2 else 0
    red_arena, amax_arena = _zero_arenas(total, nrec * K.AMAX_FLOATS, gout.device)    # (one fill launch for both)
    bnadj_on = wgrad_stream is not None and bool(bnadj)     # (only where a second stream exists: what the form buys is the earlier start)
    amax_d_of, wgrad_done = {}, set()
    if nrec:
        amax_iter = iter(amax_arena.split(K.AMAX_FLOATS))
        amax_next = lambda: next(amax_iter)

    # Round 6: consecutive fp16-piece weight gradients of ONE geometry are queued and leave as one launch of up to `wbatch` layers
    # (kernels.conv_wgrad_batch: one set of split-K partial blocks and one kernel / reduce launch pair per BATCH instead of per layer);
    # a change of geometry, a full queue or the end of the walk flushes it.
    # MEASURED AND NOT ADOPTED (profiles/r06i_step_ab_wgrad_batch.txt, three alternating runs on one box): 12.43 ms per step with every
    # layer its own launch, 12.49 with pairs, 12.51 with batches of four - a batch starts when its LAST layer's gradient exists, and what
    # the later start costs the two-stream schedule exceeds the launches and the 12 MB of partial blocks per layer it saves.  The default
    # is 1 (every layer its own launch, as in rounds 2-5); BIHOME_WGRAD_BATCH=2..4 enables the queue.
    wbatch = max(1, min(4, int(os.environ.get("BIHOME_WGRAD_BATCH", "1")))) if wgrad_stream is not None else 1
    wpend = []

    def wflush():
        if not wpend:
            return
        items = list(wpend)
        wpend.clear()
        ev = torch.cuda.Event()
        ev.record(main)                                       # the last queued layer's gradient is final here
        for it in items:
            it[2].record_stream(wgrad_stream)
        with torch.cuda.stream(wgrad_stream):
            wgrad_stream.wait_event(ev)
            if len(items) == 1 or not K.conv_wgrad_batch([(it[1], it[2], it[3], it[4]) for it in items], x3_ws):
                for it in items:
                    K.conv_wgrad(it[1], it[2], it[3], None, it[4], det_ws=x3_ws)
        if on_param_grad is not None:                         # these layers' gradients are final (enqueued): their buckets may leave
            for it in items:
                on_param_grad(it[5].weight)

    for i in range(len(prog.ops) - 1, -1, -1):
        op = prog.ops[i]
        g = grads.pop(op.dst, None)
        if g is None:
            continue
        need_src_grad = (op.src != 0) or want_input_grad
        x = slots[op.src]
        if op.kind == "conv":
            d, wk = ctx.descs[i], ctx.weights[i]
            m = op.mod
            if want_wgrad and m.weight.requires_grad and op.extra["weight_fn"] is None and i not in wgrad_done:
                gw = m.weight.grad if m.weight.dim() == 2 else kview(m.weight.grad)
                gb = m.bias.grad if (m.bias is not None and m.bias.requires_grad) else None
                has_gb = gb is not None
                if i in bias_off and has_gb:              # column sums already taken by the consumer's dgrad epilogue
                    K.bias_grad_from_sums(red_arena[bias_off[i]:bias_off[i] + K.bn_stats_doubles(1, d.Co)], gb, 1, d.Co)
                    gb = None
                ws = det_ws if det_ws is not None else (x3_ws if getattr(d, "bh_wx3", False) else None)
                # the fp16-piece weight gradient's eight-wave form is the faster launch alone, the four-wave form the better neighbour: it
                # leaves ~200 registers per SIMD lane, so the main stream's BatchNorm kernels run ON the same CUs (same-box A/B, round 5:
                # two streams 13.30 ms with the four-wave form against 13.49; one stream 14.13 against 14.02)
                # and with 160 instead of 256 workgroups the side stream leaves CUs to the main stream's next launch (and writes fewer partial
                # blocks): two streams 12.60 against 12.79 ms (ROUTE_WX3_SHARED)
                if wgrad_stream is None:
                    d.route = (d.route | K.ROUTE_WX3_PC) & ~K.ROUTE_WX3_SHARED
                else:
                    d.route = (d.route & ~K.ROUTE_WX3_PC) | K.ROUTE_WX3_SHARED
                queued = False
                if wgrad_stream is None:
                    K.conv_wgrad(x, g, gw, gb, d, det_ws=ws)
                elif (wbatch > 1 and gb is None and getattr(ctx, "precision", 0) == K.F16X2 and getattr(d, "bh_wx3", False) and ws is not None
                      and d.Ci % 64 == 0 and d.Co % 64 == 0 and d.Hi >= 8):
                    # round 6: queued - up to `wbatch` consecutive layers of one geometry leave in ONE launch (kernels.conv_wgrad_batch)
                    key = (d.N, d.Hi, d.Wi, d.Ci, d.Co, (x.groups, bool(x.relu)) if isinstance(x, K.BnOnLoad) else None)
                    if wpend and wpend[0][0] != key:
                        wflush()
                    K.amax_of(g)                                      # (magnitude records: on the main stream, as below)
                    K.amax_of(x)
                    wpend.append((key, x, g, gw, d, m))
                    if len(wpend) >= wbatch:
                        wflush()
                    queued = True
                    if on_param_grad is not None and has_gb:      # (its column sums came from the consumer's dgrad epilogue: final)
                        on_param_grad(m.bias)
                else:
                    if getattr(ctx, "precision", 0) == K.F16X2 and getattr(d, "bh_wx3", False):
                        # magnitude records the fp16-piece weight gradient reads: made (if missing) on the MAIN stream, in front of the
                        # event - a record first measured on the side stream would be read by the main stream's dgrad without a wait
                        # (round-4 ADVICE: a still-zero record scales by 2^100)
                        K.amax_of(g)
                        K.amax_of(x)
                    ev = torch.cuda.Event()
                    ev.record(main)                                   # g is final here
                    g.record_stream(wgrad_stream)
                    with torch.cuda.stream(wgrad_stream):
                        wgrad_stream.wait_event(ev)
                        K.conv_wgrad(x, g, gw, gb, d, det_ws=ws if getattr(d, "bh_wx3", False) else None)     # (one stream: launches serialise on the workspace)
                if on_param_grad is not None and not queued:       # gradient of this layer is final: its bucket may leave
                    on_param_grad(m.weight)
                    if has_gb:
                        on_param_grad(m.bias)
            if need_src_grad and i in from_1x1 and op.src not in grads:
                # the BatchNorm in front rebuilds this dgrad per element in its own adjoint (bn_bwd_from_1x1): nothing is launched here
                grads[op.src] = K.GradFrom1x1(g, wk)
                continue
            if need_src_grad:
                red = None
                if i in fuse_bn:
                    b = fuse_bn[i]
                    bop, bm = prog.ops[b], prog.ops[b].mod
                    sums = red_arena[red_off[b]:red_off[b] + K.bn_stats_doubles(ctx.groups, bm.num_features)]
                    red = dict(z=slots[bop.src], y=slots[bop.dst] if (bop.relu and bop.res is not None) else None,
                               stats=ctx.stats[b], gamma=bm.weight, beta=bm.bias, eps=bm.eps, relu=bop.relu, sums=sums,
                               groups=ctx.groups)
                    bn_reduced[b] = sums
                    if bnadj_on and b in bnadj and amax_next is not None:
                        red["amax_d"] = amax_d_of[b] = amax_next()      # (max |mask(d)|: the bound of the on-load adjoint's fp16 scale)
                if i in fuse_bias and op.src not in grads and red is None:
                    p = fuse_bias[i]
                    grads[op.src] = K.conv_dgrad(g, wk, d, wpacked=ctx.wpacked.get(i),
                                                 colsum=red_arena[bias_off[p]:bias_off[p] + K.bn_stats_doubles(1, d.Ci)])
                elif op.src in grads:
                    K.conv_dgrad(g, wk, d, out=grads[op.src], bn_reduce=red, wkey=ctx.wkeys[i], wpacked=ctx.wpacked.get(i))
                else:
                    # (input_sink: the consumer of the INPUT's gradient folded into the first conv's dgrad - the warp's adjoint, round 6)
                    gsrc = K.conv_dgrad(g, wk, d, bn_reduce=red, wkey=ctx.wkeys[i], wpacked=ctx.wpacked.get(i),
                                        warp_sink=input_sink if (op.src == 0 and red is None) else None)
                    if gsrc is not None:
                        grads[op.src] = gsrc
        elif op.kind == "bn" and i in ctx.joined:
            j = ctx.joined[i]
            lop, m, lm = prog.ops[j], op.mod, prog.ops[j].mod
            train_a = want_wgrad and m.weight.requires_grad
            train_b = want_wgrad and lm.weight.requires_grad
            xb = slots[lop.src]
            gxa, gxb = K.bn_join_bwd(g, slots[op.dst], x, xb, m, lm, ctx.stats[i], ctx.stats[j], ctx.groups, op.relu, train_a, train_b,
                                     amax_a=amax_next() if amax_next else None, amax_b=amax_next() if amax_next else None)
            if on_param_grad is not None:
                for p_, tr_ in ((m.weight, train_a), (m.bias, train_a), (lm.weight, train_b), (lm.bias, train_b)):
                    if tr_:
                        on_param_grad(p_)
            if need_src_grad:
                contribute(op.src, gxa)
            if (lop.src != 0) or want_input_grad:
                contribute(lop.src, gxb)
        elif op.kind == "bn":
            m = op.mod
            train_w = want_wgrad and m.weight is not None and m.weight.requires_grad
            yb = slots[op.dst]
            if isinstance(yb, (K.BnOnLoad, K.BnPooled)):  # applied on load by its consumer / fused with the pooling: no output tensor (the mask comes from x)
                yb = None
            if isinstance(g, K.GradFrom1x1):             # the 1x1 conv behind this BatchNorm left its dgrad to this call
                gx = K.bn_bwd_from_1x1(g, x, m.weight, m.bias, ctx.stats[i], ctx.groups, m.eps, op.relu,
                                       m.weight.grad if train_w else None, m.bias.grad if train_w else None,
                                       amax=amax_next() if (amax_next and m.num_features > 1) else None)
                if train_w and on_param_grad is not None:
                    on_param_grad(m.weight)
                    on_param_grad(m.bias)
                if need_src_grad:
                    contribute(op.src, gx)
                continue
            if isinstance(g, K.PooledGrad):              # BatchNorm (+ReLU) + MaxPool in one pass: its adjoint in one call too
                gx = K.bn_maxpool_bwd(g, x, m.weight, m.bias, ctx.stats[i], m.running_mean, m.running_var, ctx.groups, m.eps, op.relu,
                                      ctx.training, m.weight.grad if train_w else None, m.bias.grad if train_w else None,
                                      amax=amax_next() if (amax_next and m.num_features > 1) else None)
                if train_w and on_param_grad is not None:
                    on_param_grad(m.weight)
                    on_param_grad(m.bias)
                if need_src_grad:
                    contribute(op.src, gx)
                continue
            if bnadj_on and i in amax_d_of and isinstance(g, torch.Tensor):
                c = bnadj[i]
                cop, cd, cm = prog.ops[c], ctx.descs[c], prog.ops[c].mod
                xin = slots[cop.src]
                if isinstance(xin, (torch.Tensor, K.BnOnLoad)) and x3_ws is not None:
                    cd.route = (cd.route & ~K.ROUTE_WX3_PC) | K.ROUTE_WX3_SHARED
                    K.amax_of(xin)                                    # (made - if missing - on the main stream, in front of the event)
                    ev = torch.cuda.Event()
                    ev.record(main)                                   # d and its sums are final here; the adjoint pass below has not started
                    g.record_stream(wgrad_stream)
                    with torch.cuda.stream(wgrad_stream):
                        wgrad_stream.wait_event(ev)
                        done = K.conv_wgrad_bnadj(xin, g, kview(cm.weight.grad), cd, x3_ws,
                                                  dict(z=x, y=yb if (op.relu and op.res is not None) else None, stats=ctx.stats[i],
                                                       sums=bn_reduced[i], gamma=m.weight, beta=m.bias, eps=m.eps, relu=op.relu, groups=ctx.groups),
                                                  amax_d_of[i])
                    if done:
                        wgrad_done.add(c)
                        if on_param_grad is not None:
                            on_param_grad(cm.weight)
            gx, gres = K.bn_bwd(g, yb, x, m.weight, ctx.stats[i], m.running_mean, m.running_var, ctx.groups,
                                m.eps, op.relu, ctx.training, op.res is not None and ((op.res != 0) or want_input_grad),
                                m.weight.grad if train_w else None, m.bias.grad if train_w else None, beta=m.bias,
                                had_res=op.res is not None, sums_ready=bn_reduced.get(i),
                                amax=amax_next() if (amax_next and m.num_features > 1) else None)
            if train_w and on_param_grad is not None:
                on_param_grad(m.weight)
                on_param_grad(m.bias)
            if need_src_grad:
                contribute(op.src, gx)
            if gres is not None:
                contribute(op.res, gres)
        elif op.kind == "tail":
            c1, bn, c2 = op.mod
            tr = want_wgrad and c1.weight.requires_grad
            N, h, w, _ = x.shape
            eval_b1 = tr and not ctx.training and c1.bias is not None and c1.bias.requires_grad
            gbeta0 = bn.bias.grad.clone() if eval_b1 else None
            gx = K.tail_bwd(g, x, kview(c1.weight), c1.bias, bn.weight, bn.bias, kview(c2.weight), ctx.stats[i],
                            bn.running_mean, bn.running_var, ctx.groups, h * w, bn.eps, ctx.training, need_src_grad,
                            kview(c1.weight.grad) if tr else None, bn.weight.grad if tr else None,
                            bn.bias.grad if tr else None, kview(c2.weight.grad) if tr else None,
                            c2.bias.grad if (tr and c2.bias is not None) else None)
            if eval_b1:
                # eval-mode BatchNorm is a fixed affine map, so layer8.0.bias has a gradient (with batch statistics it is
                # exactly zero and the fused kernel never forms it): g_b1 = g_beta * gamma / sqrt(running_var + eps)
                with torch.no_grad():
                    c1.bias.grad.add_((bn.bias.grad - gbeta0) * bn.weight * torch.rsqrt(bn.running_var + bn.eps))
            if tr and on_param_grad is not None:
                for p_ in (c2.weight, c2.bias, bn.weight, bn.bias, c1.weight, c1.bias):
                    if p_ is not None:
                        on_param_grad(p_)
            if need_src_grad:
                contribute(op.src, gx)
        elif op.kind == "maxpool":
            if need_src_grad:
                if isinstance(x, K.BnPooled) and op.src not in grads and os.environ.get("BIHOME_BN_POOL_BWD", "1") != "0":
                    grads[op.src] = K.PooledGrad(g, ctx.stats[i])          # (consumed by the BatchNorm's adjoint: bn_maxpool_bwd)
                else:
                    contribute(op.src, K.maxpool_bwd(ctx.stats[i], g, tuple(x.shape)))
        elif op.kind == "gap":
            if need_src_grad:
                contribute(op.src, K.gap_bwd(g, tuple(x.shape)))
    wflush()
    if wgrad_stream is not None:
        main.wait_stream(wgrad_stream)          # the optimiser (and the release of the activations) follows
    return grads.get(0)


def _drop_pending_counters(module, *unused):
    for m in module.modules():
        if getattr(m, "_bh_pending_batches", 0):
            m._bh_pending_batches = 0


def install_counter_hooks(module):
    """load_state_dict replaces `num_batches_tracked`: calls counted on the host before the load must not be added on
    top of the loaded value at the next state_dict()."""
    module.register_load_state_dict_pre_hook(lambda *a, **k: _drop_pending_counters(module))
    return module


def flush_counters(module):
    """Write the BatchNorm call counters accumulated on the host into the `num_batches_tracked`
    buffers (kept for state-dict compatibility with the reference's checkpoints)."""
    for m in module.modules():
        n = getattr(m, "_bh_pending_batches", 0)
        if n and getattr(m, "num_batches_tracked", None) is not None:
            m.num_batches_tracked += n
            m._bh_pending_batches = 0


class NetFunction(torch.autograd.Function):
    """One autograd node for a whole conv stack: forward = run_forward, backward = run_backward.
    `anchor` is any trainable parameter of the stack (or a dummy): it only tells autograd that the
    node has trainable state; parameter gradients are written by the kernels into `.grad` directly."""

    @staticmethod
    def forward(ctx, x, anchor, runner, groups, grad_mode=True):
        with K.det_scope(runner.det):               # every launch of this pass carries the Runner's own determinism bit
            return NetFunction._forward(ctx, x, anchor, runner, groups, grad_mode)

    @staticmethod
    def backward(ctx, g):
        with K.det_scope(ctx.runner.det):
            return NetFunction._backward(ctx, g)

    @staticmethod
    def _forward(ctx, x, anchor, runner, groups, grad_mode=True):
        # (needs_input_grad reflects requires_grad of the inputs even under torch.no_grad(): the caller passes the mode)
        need = grad_mode and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        training = runner.module.training
        if training or need:
            # running statistics (training) or the weights (a backward + fused optimizer step follows: no version bump) are
            # about to change under the folded copies
            runner._fold.clear()
        fold = runner._fold if (runner.fold_bn and not training and not need) else None
        source = getattr(runner, "input_source", None)
        out, saved = run_forward(runner.prog, x, groups, training, save=need, precision=runner.precision, fold_cache=fold,
                                 packer=runner.packer_for(x.device) if fold is None else None, input_source=source)
        if source is not None and not source.get("filled"):
            raise RuntimeError("input_source: no conv reads the network's input directly - nobody made the warped image")
        ctx.runner, ctx.saved, ctx.want_x = runner, saved, ctx.needs_input_grad[0]
        return out

    @staticmethod
    def _backward(ctx, g):
        r = ctx.runner
        if r.flat is not None:
            r.flat.attach(g.device)
        hook = r.reducer.param_ready if r.reducer is not None else None
        # (deterministic mode: one stream - the weight-gradient launches share one workspace)
        side = side_stream(g.device) if (r.wgrad_on_side_stream and r.flat is not None and not K.deterministic()) else None
        if r.reducer is not None:
            r.reducer.wait_streams = [side] if side is not None else []
        gin = run_backward(r.prog, ctx.saved, g.contiguous(), want_wgrad=r.flat is not None, want_input_grad=ctx.want_x,
                           on_param_grad=hook, wgrad_stream=side, det_ws=r.det_workspace(g.device) if side is None else None,
                           x3_ws=r.x3_workspace(g.device), input_sink=getattr(r, "input_sink", None))
        ctx.saved = None
        return gin, None, None, None, None


_RUNNERS = weakref.WeakSet()


def trainable_runners(model):
    """Every conv-stack executor of `model` that owns a flat gradient buffer (built now if the module has not run yet): the backbone's,
    and for the ContentAware backbone the feature extractor's and a trained mask predictor's."""
    out = []
    for m in model.modules():
        if not (hasattr(m, "_build") and hasattr(m, "__dict__") and "_runner" in m.__dict__):
            continue
        if getattr(m, "fix_mask", False):                 # (the all-ones mask predictor never runs)
            continue
        if m._runner is None:
            to_kernel_layout_(m)
            m._runner = m._build()
        if isinstance(m._runner, Runner) and m._runner.flat is not None:
            out.append(m._runner)
    return out


def set_stream_overlap(on, model=None):
    """Switch the side-stream overlap (weight gradients, the head's feature prefetch) of the Runners over `model`'s modules
    (all Runners when None) and of heads that hold a `prefetch_features` switch; returns nothing.  bench.py turns it off around
    its per-kernel timing leg."""
    mods = None if model is None else {id(m) for m in model.modules()}
    for r in list(_RUNNERS):
        if mods is None or id(r.module) in mods:
            r.wgrad_on_side_stream = bool(on)
    if model is not None:
        for m in model.modules():
            if hasattr(m, "prefetch_features"):
                m.prefetch_features = bool(on)


def invalidate_caches(model=None):
    """Drop every host-side copy derived from parameters / buffers (BatchNorm-folded weights, fragment-ordered packs, the
    transposed stem table) of the Runners over `model`'s modules (all Runners when None).  Needed wherever weights or
    running statistics change WITHOUT the eager forward's own bookkeeping seeing it: HIP-graph replays (no Python runs,
    fused Adam and the BatchNorm kernels bump no version counter) and the parameter broadcast of attach_reducer."""
    mods = None if model is None else {id(m) for m in model.modules()}
    for r in list(_RUNNERS):
        if mods is not None and id(r.module) not in mods:
            continue
        r._fold.clear()
        if r._packer is not None:
            r._packer.invalidate()
    K.invalidate_stem_tables()


class Runner:
    """Binds a Program to its nn.Module (parameter container) and, if trainable, a FlatGrads buffer."""

    def __init__(self, module, prog, trainable, precision="f32", fold_cache=None):
        _RUNNERS.add(self)
        self.module, self.prog = module, prog
        self.precision = K.PRECISION[str(precision).lower()]     # conv operand precision (0 fp32, 1 bf16 operands, 2 f32x3)
        # deterministic calls (include/bihome.h): a property of THIS Runner, fixed when it is built (kernels.set_deterministic /
        # BIHOME_DETERMINISTIC=1 give the default, `with kernels.det_scope(True): build_model(...)` a per-model choice) - the library has
        # no process-wide mode, two models of one process may differ
        self.det = K.deterministic()
        params = [p for p in module.parameters() if p.requires_grad]
        self.flat = FlatGrads(params) if (trainable and params) else None
        self.anchor = params[0] if (trainable and params) else None
        self._dummy = None
        self.reducer = None        # bihome_amd.ddp.FlatGradReducer when training data-parallel
        # eval-mode BatchNorm folding cache (run_forward / _folded); Runners over the SAME modules (the extractor's
        # 1- and 3-channel programs) share one dict so that a training forward through either invalidates both
        self._fold = fold_cache if fold_cache is not None else {}
        self.fold_bn = os.environ.get("BIHOME_FOLD_BN", "1") != "0"
        # second HIP stream for the weight-gradient launches (default since round 4: with three instead of six MFMA products per
        # product the 3x3 kernels are no longer matrix-pipe-bound and two streams fill each other's gaps: 15.5 -> 14.7 ms per step;
        # BIHOME_OVERLAP=0 or bench.py --no-overlap: one stream - per-kernel durations of rocprofv3 / the roofline leg are then those
        # of each kernel alone, which is how profiles/ and bench.py's roofline object are measured)
        self.wgrad_on_side_stream = os.environ.get("BIHOME_OVERLAP", "1") != "0"
        # fragment-ordered weight copies for the halo-tiled 3x3 kernel (csrc/conv3x3.hip PACKED; BIHOME_PACK_WEIGHTS=0: off)
        self.use_packer = os.environ.get("BIHOME_PACK_WEIGHTS", "1") != "0"
        self._packer = None
        self.deterministic_wgrad = os.environ.get("BIHOME_DETERMINISTIC_WGRAD", "0") == "1"
        self._det_ws = None

    def det_workspace(self, device):
        """BIHOME_DETERMINISTIC_WGRAD=1: one 40 MB workspace for the fixed-order split-K reduction of the 3x3 weight
        gradients (every launch of the stride-1 fast path stores 2048 x 16 KB partial tiles)."""
        det = K.deterministic()
        if not (self.deterministic_wgrad or det) or self.flat is None:
            return None
        size = DET_WS_BYTES if det else X3_WS_BYTES
        if self._det_ws is None or self._det_ws.device != device or self._det_ws.numel() * 4 < size:
            self._det_ws = torch.empty(size // 4, dtype=torch.float32, device=device)
        return self._det_ws

    def x3_workspace(self, device):
        """'f32' arithmetic (precision 2): the 40 MB workspace the f32x3 weight-gradient kernel stores its <= 256 partial blocks
        of 147 KB in (wgrad_x3_reduce_kernel adds them in split order: those layers' gradients are bitwise reproducible)."""
        if K.deterministic():
            return self.det_workspace(device)
        if self.precision not in K.SPLIT_PIECES or self.flat is None:
            return None
        if self._det_ws is None or self._det_ws.device != device:
            self._det_ws = torch.empty(X3_WS_BYTES // 4, dtype=torch.float32, device=device)
        return self._det_ws

    def packer_for(self, device):
        if not self.use_packer:
            return None
        if self._packer is None or self._packer_dev != device:
            pk = K.packer_for_precision(self.precision)
            for op in self.prog.ops:
                m = op.mod
                if (op.kind == "conv" and isinstance(m, nn.Conv2d) and op.extra["weight_fn"] is None and m.kernel_size == (3, 3)
                        and m.stride == (1, 1) and m.padding == (1, 1) and m.in_channels % 32 == 0 and m.out_channels % 32 == 0
                        and m.weight.device == device):
                    pk.get(m.weight)
            self._packer, self._packer_dev = pk, device
        return self._packer

    def __call__(self, x, groups):
        if not x.is_cuda:
            raise RuntimeError("bihome_amd runs on the MI355X only (input on %s): there is no CPU fallback; "
                               "use oracle/ for CPU checks" % x.device)
        anchor = self.anchor
        if anchor is None:
            if self._dummy is None or self._dummy.device != x.device:
                self._dummy = torch.zeros(1, device=x.device)
            anchor = self._dummy
        return NetFunction.apply(x, anchor, self, groups, torch.is_grad_enabled())
