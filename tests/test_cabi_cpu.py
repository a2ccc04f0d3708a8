"""CPU-side checks of the boundary: the shared library loads without a GPU and exports exactly the
symbols include/bihome.h declares; host-side program construction; the product never reaches into oracle/."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "bihome.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#ifdef BH_TUNING.*?#endif", "", text, flags=re.S)      # the ablation hook exists in the -DBH_TUNING build only
    return sorted(set(re.findall(r"\bint\s+(bh_\w+)\s*\(", text)))      # (status-returning entry points; bh_conv_wgrad_det_bytes returns a size)


def test_library_exports_every_declared_symbol():
    from bihome_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), "libbihome_hip.so does not export %s" % s
    assert sorted(_lib.SIGNATURES) == syms, (set(_lib.SIGNATURES) ^ set(syms))
    assert _lib.lib.bh_version() >= 1
    assert not hasattr(raw, "bh_debug_force_tile")      # no process-global tuning state in the product library


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from bihome_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.BihomeLibError, match="no CPU fallback"):
        _lib._load()


def test_cpu_tensors_are_refused():
    from bihome_amd import kernels as K
    with pytest.raises(RuntimeError, match="no CPU path"):
        K.h4pt_fwd(torch.zeros(2, 4, 2), 128)


def test_product_does_not_import_oracle():
    bad = []
    for base in ("bihome_amd", "src"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith(".py"):
                    t = open(os.path.join(dp, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", t, flags=re.M) or "/root/reference" in t:
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_zeng_program_shape():
    from bihome_amd import configs
    from bihome_amd.backbones.Rethinking import Model
    m = Model(**configs.get("zeng-bihome")["MODEL"]["BACKBONE"])
    r = m._build()
    kinds = [op.kind for op in r.prog.ops]
    # 54 BatchNorms / 59 convs in the backbone; the last conv+BN+ReLU+conv is one fused "tail" op
    assert kinds.count("bn") == 53 and kinds.count("tail") == 1
    assert kinds.count("conv") == 57 and kinds.count("maxpool") == 1
    assert r.prog.ops[0].extra["in_nchw"] and r.prog.ops[-1].kind == "tail"
    m.fuse_tail = False
    kinds = [op.kind for op in m._build().prog.ops]
    assert kinds.count("bn") == 54 and kinds.count("conv") == 59 and m._build().prog.ops[-1].extra["out_nchw"]
    assert sum(p.numel() for p in m.parameters()) == 10574178
    # every conv weight is in kernel layout and FlatGrads views alias it stride for stride
    for p in m.parameters():
        if p.dim() == 4:
            assert p.permute(0, 2, 3, 1).is_contiguous()
    r.flat.attach(torch.device("cpu"))
    for p in m.parameters():
        assert p.grad is not None
        if p.dim() == 4:        # same memory order as the parameter (strides of size-1 dims are free)
            assert p.grad.permute(0, 2, 3, 1).is_contiguous()


def test_conv_desc_and_variant():
    from bihome_amd import kernels as K
    d = K.conv_desc(128, 128, 128, 2, 64, 7, 2, 3, in_nchw=True)
    assert (d.Ho, d.Wo) == (64, 64) and K.conv_out_shape(d) == (128, 64, 64, 64)
    d = K.conv_desc(4, 8, 8, 256, 128, 2, 2, 0, transposed=True)
    assert (d.Ho, d.Wo) == (16, 16)
    assert K.conv_flops(d) == 2.0 * 4 * 64 * 256 * 128 * 4


def test_conv_variant_query_reports_the_dispatch():
    """bh_conv_variant runs the library's own dispatch with the launches replaced by a name record (no GPU needed): the
    kernel symbols of the BENCH shapes (BASELINE.json configs[1], 128 stacked images) and of the routing bits."""
    from bihome_amd import kernels as K
    from bihome_amd._lib import ROUTE_GENERIC_CONV, ROUTE_HALO_SMALL, ROUTE_NO_STEM7, ROUTE_WGRAD_3TAP, ROUTE_WGRAD_GENERIC
    d = K.conv_desc(128, 32, 32, 64, 64, 3, 1, 1)
    assert K.conv_variant(d, "fwd") == "conv3x3_halo_kernel<false,64,false,2,false,false,false,3,false,false>"
    assert K.conv_variant(d, "fwd", bn_groups=2) == "conv3x3_halo_kernel<false,64,false,2,false,false,false,3,false,false>"
    assert K.conv_variant(d, "dgrad") == "conv3x3_halo_kernel<true,64,false,2,false,false,false,3,false,false>"
    assert K.conv_variant(d, "wgrad") == "wgrad_s1_kernel<1,false>"
    assert K.dgrad_bn_reduce_ok(d)
    d8 = K.conv_desc(128, 8, 8, 256, 256, 3, 1, 1)
    assert K.conv_variant(d8, "fwd") == "conv3x3_halo_kernel<false,64,false,1,false,false,false,3,false,false>"          # one sub-tile per workgroup
    assert K.conv_variant(K.conv_desc(128, 128, 128, 32, 32, 3, 1, 1), "fwd") == "conv3x3_halo_kernel<false,32,false,2,false,false,false,3,false,false>"
    assert K.conv_variant(K.conv_desc(128, 32, 32, 64, 64, 3, 1, 1, precision=1), "wgrad") == "wgrad_s1_kernel<3,true>"
    # packed weights (w_layout 1: fp32 fragments, 2: three bf16 pieces with precision 2 = the default 'f32' arithmetic of the models)
    assert K.conv_variant(K._with_layout(d, 1), "fwd") == "conv3x3_halo_kernel<false,64,false,2,true,false,false,3,false,false>"
    assert K.PRECISION["f32"] == 4 and K.PRECISION["f32-mfma"] == 0          # round 4: the default 'f32' is the fp16-piece arithmetic
    d4 = K.conv_desc(128, 32, 32, 64, 64, 3, 1, 1, precision=K.PRECISION["f32"])
    assert K.packed_layout(4) == 4 and K.SPLIT_PIECES[4] == 2
    assert K.conv_variant(K._with_layout(d4, 4), "fwd") == "conv3x3_pc_kernel<false,false,0>"          # round 5: persistent producer / consumer workgroups
    assert K.conv_variant(K._with_layout(d4, 4), "dgrad") == "conv3x3_pc_kernel<true,false,0>"
    from bihome_amd._lib import ROUTE_C3_TILE_WG
    d4t = K.conv_desc(d4.N, d4.Hi, d4.Wi, d4.Ci, d4.Co, 3, 1, 1, precision=4, route=ROUTE_C3_TILE_WG)
    assert K.conv_variant(K._with_layout(d4t, 4), "fwd") == "conv3x3_halo_kernel<false,64,false,2,true,true,false,2,false,true>"
    assert K.conv_variant(K._with_layout(d4t, 4), "dgrad") == "conv3x3_halo_kernel<true,64,false,2,true,true,false,2,false,true>"
    dx = K.conv_desc(128, 32, 32, 64, 64, 3, 1, 1, precision=K.PRECISION["f32x3"])
    assert dx.precision == 2
    assert K.conv_variant(K._with_layout(dx, 2), "fwd") == "conv3x3_halo_kernel<false,64,false,2,true,true,false,3,false,false>"
    assert K.conv_variant(K._with_layout(dx, 2), "dgrad") == "conv3x3_halo_kernel<true,64,false,2,true,true,false,3,false,false>"
    assert K.conv_variant(dx, "fwd") == "conv3x3_halo_kernel<false,64,false,2,false,false,false,3,false,false>" and K.conv_variant(dx, "wgrad") == "wgrad_x3_kernel<64,false,3,false,false,false>"
    assert K.wgrad_det_bytes(dx) == 256 * 36864 * 4                       # 256 partial blocks of 64 x 9 x 64
    # 'f32x2' (precision 3, w_layout 3: two rounded bf16 pieces, three products) - the same kernels with NP = 2
    d2 = K.conv_desc(128, 32, 32, 64, 64, 3, 1, 1, precision=K.PRECISION["f32x2"])
    assert d2.precision == 3 and K.packed_layout(3) == 3 and K.SPLIT_PIECES[3] == 2
    assert K.conv_variant(K._with_layout(d2, 3), "fwd") == "conv3x3_halo_kernel<false,64,false,2,true,true,false,2,false,false>"
    assert K.conv_variant(K._with_layout(d2, 3), "dgrad") == "conv3x3_halo_kernel<true,64,false,2,true,true,false,2,false,false>"
    assert K.conv_variant(d2, "wgrad_det") == "wgrad_x3_kernel<64,false,2,false,false,false>+wgrad_x3_reduce_kernel<64>"
    for lay, prec in ((2, 3), (3, 2), (3, 0)):
        with pytest.raises(RuntimeError):
            K.conv_variant(K._with_layout(K.conv_desc(128, 32, 32, 64, 64, 3, 1, 1, precision=prec), lay), "fwd")   # layout <-> precision
    d32 = K.conv_desc(128, 64, 64, 32, 32, 3, 1, 1, precision=2)
    assert K.conv_variant(d32, "wgrad") == "wgrad_x3_kernel<32,false,3,false,false,false>" and K.conv_variant(d32, "wgrad_det") == "wgrad_x3_kernel<32,false,3,false,false,false>+wgrad_x3_reduce_kernel<32>"
    assert K.conv_variant(K.conv_desc(128, 128, 128, 16, 16, 3, 1, 1, precision=2), "wgrad").startswith("wgrad_small")           # channels % 32
    with pytest.raises(RuntimeError):
        K.conv_variant(K._with_layout(d, 2), "fwd")                  # split weights need precision 2
    # per-call routing bits (tests / benchmarks)
    assert K.conv_variant(K.conv_desc(128, 32, 32, 64, 64, 3, 1, 1, route=ROUTE_GENERIC_CONV), "fwd").startswith("conv_gemm_kernel<64,64,")
    small = K.conv_desc(2, 8, 8, 64, 64, 3, 1, 1)
    assert K.conv_variant(small, "fwd").startswith("conv_gemm_kernel") and not K.dgrad_bn_reduce_ok(small)
    assert K.conv_variant(K.conv_desc(2, 8, 8, 64, 64, 3, 1, 1, route=ROUTE_HALO_SMALL), "fwd").startswith("conv3x3_halo_kernel")
    assert K.conv_variant(K.conv_desc(128, 32, 32, 64, 64, 3, 1, 1, route=ROUTE_WGRAD_GENERIC), "wgrad") == "wgrad_kernel<true,false>"
    assert K.conv_variant(K.conv_desc(128, 32, 32, 64, 64, 3, 1, 1, route=ROUTE_WGRAD_3TAP), "wgrad") == "wgrad_s1_kernel<3,false>"
    stem = K.conv_desc(128, 128, 128, 2, 64, 7, 2, 3, in_nchw=True)
    assert K.conv_variant(stem, "fwd") == "stem7_fwd_kernel<2>"
    assert K.conv_variant(K.conv_desc(128, 128, 128, 2, 64, 7, 2, 3, in_nchw=True, route=ROUTE_NO_STEM7), "fwd").startswith("conv_gemm_kernel<128,")
    assert K.conv_variant(stem, "wgrad") == "wgrad_kernel<false,false>"                 # joint (tap, channel) columns: N = 98
    # generic implicit GEMM: 1x1, transposed, stride 2
    assert K.conv_variant(K.conv_desc(128, 16, 16, 256, 128, 1, 1, 0), "fwd") == "conv_gemm_kernel<64,64,64,true,false,true>"
    assert K.conv_variant(K.conv_desc(128, 8, 8, 256, 256, 2, 2, 0, transposed=True), "fwd") == "conv_gemm_kernel<64,128,32,true,false,true>"
    assert K.conv_variant(K.conv_desc(128, 128, 128, 128, 2, 1, 1, 0, out_nchw=True), "fwd") == "conv_gemm_kernel<128,32,32,true,false,false>"
    assert "conv_gemm_kernel" in K.conv_variant(K.conv_desc(128, 32, 32, 64, 128, 3, 2, 1), "dgrad")


def test_synthetic_pairs_are_consistent():
    """delta_gt really is the homography between the two patches (sanity of the synthetic generator)."""
    import numpy as np
    from bihome_amd import synth
    d = synth.make_pairs(3, seed=5)
    assert d["patch_1"].shape == (3, 1, 128, 128) and d["delta"].min() >= -32 and d["delta"].max() <= 31
    c = np.array([[0, 0], [128, 0], [128, 128], [0, 128]], np.float64)
    for b in range(3):
        H = synth.four_point_homography(c, c + d["delta"][b])
        w = synth.warp_bilinear(d["patch_1"][b].transpose(1, 2, 0).astype(np.float64), H, 128, 128)[..., 0]
        inside = synth.warp_bilinear(np.ones((128, 128, 1)), H, 128, 128)[..., 0] == 1
        assert np.abs(w - d["patch_2"][b, 0])[inside].mean() < 0.02


def test_imagenet_resnet34_key_placement():
    """PRETRAINED_RESNET: torchvision resnet34 keys land where Rethinking.py:189-282 / ResNet34.py:15-19
    put them.  Checked against the reference's torchvision-layout container built by the oracle (no network: random
    tensors of the right shapes stand for the ImageNet file)."""
    import torch
    from bihome_amd.weights import zeng_keys_from_torchvision
    from oracle import bihome_oracle as O
    tv = O._TVResNet34()                                  # torchvision.models.resnet34 layout
    g = torch.Generator().manual_seed(0)
    state = {k: torch.randn(v.shape, generator=g) if v.dtype.is_floating_point else v.clone() for k, v in tv.state_dict().items()}
    mapped = zeng_keys_from_torchvision(state)
    zeng = O.ZengBackbone(PATCH_KEYS=["patch_1", "patch_2"], TARGET_KEYS=["pf_hat_12"], RESNET_BLOCK="ResNet34",
                          VARIANT="OneLine", IMAGE_SIZE=128)
    own = zeng.state_dict()
    # every unit of layer2-4 is covered, nothing else is touched, shapes agree
    assert set(mapped) == {k for k in own if k.split(".")[0] in ("layer2", "layer3", "layer4")
                           and k.split(".")[1].isdigit() and int(k.split(".")[1]) < {"layer2": 3, "layer3": 4, "layer4": 6}[k.split(".")[0]]}
    for k, v in mapped.items():
        assert tuple(own[k].shape) == tuple(v.shape), k
    assert torch.equal(mapped["layer3.0.lower_branch.0.weight"], state["layer2.0.downsample.0.weight"])
    assert torch.equal(mapped["layer4.5.upper_branch.4.running_var"], state["layer3.5.bn2.running_var"])


def test_ctypes_structs_mirror_the_header():
    """The ctypes mirrors in bihome_amd/_lib.py (descriptor, pack job, BatchNorm reduce / on-load records) list the same fields,
    in the same order and with the same C types, as the structs of include/bihome.h - a field added on one side only would
    shift everything behind it silently."""
    import ctypes
    import re
    from bihome_amd import _lib
    src = open(os.path.join(os.path.dirname(__file__), "..", "include", "bihome.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    structs = {}
    for body, name in re.findall(r"typedef\s+struct\s*\w*\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"(const\s+)?(\w+)\s*(\*?)\s*(.*)", decl)
            ctype = "ptr" if m.group(3) or "*" in m.group(4) else m.group(2)
            for nm in m.group(4).replace("*", "").split(","):
                fields.append((nm.strip(), ctype))
        structs[name] = fields
    kinds = {ctypes.c_int: "int", ctypes.c_float: "float", ctypes.c_void_p: "ptr"}
    for cname, cls in (("bh_conv_desc", _lib.BhConvDesc), ("bh_pack3x3_job", _lib.BhPack3x3Job), ("bh_bn_reduce", _lib.BhBnReduce),
                       ("bh_bn_in", _lib.BhBnIn)):
        assert cname in structs, cname
        mirror = [(f, kinds[tp]) for f, tp in cls._fields_]
        assert mirror == structs[cname], (cname, mirror, structs[cname])


def test_zhang_plugins_are_discoverable_and_share_the_reference_state_dict_schema():
    """src.backbones.ContentAware / src.heads.TripletHead (round 3) resolve through the reference's importlib discovery (train.py:675-690)
    and carry the state-dict keys of the oracle's module tree (which the reference fixture pins); FIX_MASK False (round 4: built) keeps
    the same schema and, like every other path, has no CPU form."""
    import importlib
    import torch
    from bihome_amd import configs
    from bihome_amd.step import build_model
    from oracle import bihome_oracle as O
    cfg = configs.get("zhang-orig")
    model = build_model(cfg, "cpu")
    assert importlib.import_module("src.backbones.ContentAware").Model is type(model[0])
    assert importlib.import_module("src.heads.TripletHead").Model is type(model[1])
    bb, head = O.build(cfg)
    assert set(torch.nn.Sequential(bb, head).state_dict().keys()) == set(model.state_dict().keys())
    for k, v in bb.state_dict().items():
        assert tuple(model[0].state_dict()[k].shape) == tuple(v.shape), k
    import copy
    trained = copy.deepcopy(configs.get("zhang-orig"))
    trained["MODEL"]["BACKBONE"]["FIX_MASK"] = False
    tm = build_model(trained, "cpu")
    assert set(tm.state_dict().keys()) == set(model.state_dict().keys())
    for mdl in (model, tm):
        with pytest.raises(RuntimeError, match="no CPU"):
            mdl[0]({"patch_1": torch.zeros(1, 1, 128, 128), "patch_2": torch.zeros(1, 1, 128, 128)})
