"""ctypes binding of libbihome_hip.so (the C ABI declared in include/bihome.h).

The product path has no CPU fallback: if the shared library is missing or does not export a symbol
the import fails loudly.  `python __graft_entry__.py build` (or `make -C bihome_amd/csrc`) builds it.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# BIHOME_TUNING=1 (tools/ only): the -DBH_TUNING build with the bh_debug_force_tile ablation hooks (`make -C csrc tuning`)
TUNING = os.environ.get("BIHOME_TUNING") == "1"
# BIHOME_LIB_VARIANT=ab (tools/ only): an A/B build of the same sources (`make -C csrc ab ABFLAGS=...` -> libbihome_hip_ab.so)
_VARIANT = os.environ.get("BIHOME_LIB_VARIANT", "")
LIB_PATH = os.path.join(_HERE, "libbihome_hip_tuning.so" if TUNING else
                        ("libbihome_hip_%s.so" % _VARIANT if _VARIANT.isalnum() else "libbihome_hip.so"))


class BhConvDesc(Structure):
    _fields_ = [(n, c_int) for n in ("N", "Hi", "Wi", "Ci", "Ho", "Wo", "Co", "kh", "kw", "stride", "pad",
                                     "transposed", "in_nchw", "out_nchw", "precision", "w_layout", "route")] + \
               [("a_bound", c_void_p), ("b_bound", c_void_p)]      # per-call magnitude records (precision 4 only)


GEOMETRY_FIELDS = tuple(n for n, _ in BhConvDesc._fields_[:17])       # what kernel routing depends on (cache keys)
AMAX_FLOATS = 512                                                     # floats of a magnitude record (include/bihome.h BH_AMAX_FLOATS)


class BhPack3x3Job(Structure):
    _fields_ = [("w", c_void_p), ("pf", c_void_p), ("pd", c_void_p), ("Co", c_int), ("Ci", c_int), ("split", c_int),
                ("reserved", c_int)]


# bh_conv_desc.route bits (include/bihome.h): explicit per-call kernel routing for tests / benchmarks; 0 = automatic
ROUTE_GENERIC_CONV, ROUTE_HALO_SMALL, ROUTE_NO_STEM7, ROUTE_WGRAD_GENERIC, ROUTE_WGRAD_3TAP = 1, 2, 4, 8, 16


class BhBnReduce(Structure):
    _fields_ = [("z", c_void_p), ("y", c_void_p), ("stats", c_void_p), ("gamma", c_void_p), ("beta", c_void_p),
                ("eps", c_float), ("relu", c_int), ("amax_d", c_void_p)]


class BhBnAdj(Structure):
    _fields_ = [("z", c_void_p), ("y", c_void_p), ("stats", c_void_p), ("sums", c_void_p), ("gamma", c_void_p), ("beta", c_void_p),
                ("eps", c_float), ("relu", c_int), ("groups", c_int)]


class BhBnIn(Structure):
    _fields_ = [("table", c_void_p), ("groups", c_int), ("relu", c_int)]


P = c_void_p
# name -> argtypes (all return int).  Must list every symbol include/bihome.h declares.
SIGNATURES = {
    "bh_version": [],
    "bh_probe_mfma_bf16": [c_int, P, P, P],
    "bh_probe_mfma_f32": [c_int, P, P, P],
    "bh_device_arch": [c_char_p, c_int],
    "bh_h4pt_fwd": [P, c_int, c_float, c_float, P, P, P],
    "bh_h4pt_bwd": [P, P, P, c_int, c_float, c_float, P, P],
    "bh_dlt_fwd": [P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P],
    "bh_dlt_bwd": [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P],
    "bh_dlt_bwd_f": [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, c_int, P],
    "bh_dsac_scores_fwd": [P, c_int, c_int, P, P],
    "bh_dsac_scores_bwd": [P, P, P, P, c_int, c_int, c_int, c_int, P, P, P, P],
    "bh_dsac_scores_bwd_f": [P, P, P, P, c_int, c_int, c_int, c_int, P, P, P, c_int, P],
    "bh_scale_samples_fwd": [P, P, c_int, c_int64, c_int, P, P],
    "bh_scale_samples_bwd": [P, P, P, c_int, c_int64, c_int, P, P, P],
    "bh_scale_samples_bwd_f": [P, P, P, c_int, c_int64, c_int, P, P, c_int, P],
    "bh_dsac_score": [P, P, c_int, c_int, c_int, c_int, P, P, P],
    "bh_warp_fwd": [P, P, c_int, c_int, c_int, c_int, c_int, P, P, P],
    "bh_warp_fwd_f": [P, P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P],
    "bh_warp_bwd": [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P],
    "bh_warp_bwd_f": [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, c_int, P],
    "bh_triplet_l1_fwd": [P] * 8 + [c_int, c_int, c_int, P, P, P, P],
    "bh_triplet_l1_fwd_f": [P] * 8 + [c_int, c_int, c_int, P, P, P, c_int, P],
    "bh_bihome_loss_fwd": [P, P, P, c_int, c_float, P, P],
    "bh_bihome_loss_bwd": [P] * 14 + [c_int, c_int, c_int, c_float] + [P] * 6 + [P],
    "bh_oneline_loss_fwd": [P, P, P, P, P, c_int, c_int, c_int, c_float, c_int, P, P, P, P, P, P],
    "bh_oneline_loss_fwd_f": [P, P, P, P, P, c_int, c_int, c_int, c_float, c_int, P, P, P, P, P, c_int, P],
    "bh_oneline_loss_bwd": [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P, P, P, P],
    "bh_zhang_triplet_fwd": [P] * 8 + [c_int, c_int, c_float, c_int, P, P, P, P],
    "bh_zhang_triplet_bwd": [P] * 12 + [c_int, c_int, c_int] + [P] * 6 + [P],
    "bh_zhang_triplet_bwd_m": [P] * 12 + [c_int, c_int, c_int] + [P] * 8 + [P],
    "bh_warp_bwd_img_f": [P, P, c_int, c_int, c_int, c_int, P, P, c_int, P],
    "bh_mask_fwd": [P, P, c_int, c_int, c_float, P, P, P, P, P],
    "bh_mask_bwd": [P] * 7 + [c_int, c_int, c_float, P, P, P],
    "bh_conv3x3_pack": [P, c_int, P],
    "bh_conv3x3_pack_f16": [P, c_int, P],
    "bh_absmax": [P, c_int64, P, P],
    "bh_bn_fwd_coeffs_amax": [P, P, P, P, P, c_int, c_int, c_int, c_float, c_float, P, P, P],
    "bh_bn_fwd_amax": [P] * 8 + [c_int, c_int, c_int, c_float, c_float, c_int, c_int, P, P],
    "bh_bn_bwd_amax": [P] * 11 + [c_int, c_int, c_int, c_float, c_int, c_int, P, P, P, P],
    "bh_conv_variant": [POINTER(BhConvDesc), c_int, c_int, c_int, c_char_p, c_int],
    "bh_conv_fwd": [P, P, P, P, POINTER(BhConvDesc), P],
    "bh_conv_fwd_act": [P, P, P, P, P, POINTER(BhConvDesc), c_int, P],
    "bh_conv_fwd_amax": [P, P, P, P, POINTER(BhConvDesc), P, P],
    "bh_conv_fwd_bnstats": [P, P, P, P, POINTER(BhConvDesc), P, c_int, P],
    "bh_conv_fwd_bnin": [P, P, P, P, POINTER(BhConvDesc), P, c_int, POINTER(BhBnIn), P],
    "bh_conv_wgrad_bnin": [P, P, P, P, POINTER(BhConvDesc), P, c_int64, POINTER(BhBnIn), P],
    "bh_conv_wgrad_batch": [c_int, P, P, P, P, P, c_int64, P, P],
    "bh_conv_wgrad_bnadj": [P, P, P, POINTER(BhConvDesc), P, c_int64, POINTER(BhBnIn), POINTER(BhBnAdj), P],
    "bh_bn_fwd_coeffs": [P, P, P, P, P, c_int, c_int, c_int, c_float, c_float, P, P],
    "bh_conv_dgrad": [P, P, P, POINTER(BhConvDesc), c_int, P],
    "bh_conv_dgrad_colsum": [P, P, P, POINTER(BhConvDesc), P, P],
    "bh_bias_grad_from_sums": [P, P, c_int, c_int, P],
    "bh_conv_dgrad_s2": [P, P, P, POINTER(BhConvDesc), c_int, P, P],
    "bh_conv_dgrad_bnreduce": [P, P, P, POINTER(BhConvDesc), c_int, POINTER(BhBnReduce), P, c_int, P],
    "bh_col2im_c1": [P, P, POINTER(BhConvDesc), c_int, P],
    "bh_stem7_dgrad_c1": [P, P, P, POINTER(BhConvDesc), P],
    "bh_stem7_dgrad_c1_warp": [P, P, P, POINTER(BhConvDesc), P, P, P, c_int, P, P],
    "bh_stem7_fwd_warp": [P, P, c_int, P, P, P, POINTER(BhConvDesc), P, P, P, c_int, P],
    "bh_stem7_wgrad": [P, P, P, POINTER(BhConvDesc), P, ctypes.c_size_t, P],
    "bh_conv_wgrad": [P, P, P, P, POINTER(BhConvDesc), P],
    "bh_conv_wgrad_det": [P, P, P, P, POINTER(BhConvDesc), P, c_int64, P],
    "bh_conv_bias_grad": [P, P, POINTER(BhConvDesc), P],
    "bh_bn_stats_doubles": [c_int, c_int],
    "bh_bn_scratch_doubles": [c_int, c_int],
    "bh_bn_fwd": [P] * 8 + [c_int, c_int, c_int, c_float, c_float, c_int, c_int, P],
    "bh_bn_bwd": [P] * 11 + [c_int, c_int, c_int, c_float, c_int, c_int, P, P, P],
    "bh_bn_stats": [P, P, c_int, c_int, c_int, c_int, P],
    "bh_bn_maxpool_fwd": [P] * 8 + [c_int] * 5 + [c_float, c_float, c_int, c_int, P, P],
    "bh_bn_maxpool_bwd": [P] * 10 + [c_int] * 5 + [c_float, c_int, c_int, P, P, P, P],
    "bh_bn_bwd_from_1x1": [P, P, c_int] + [P] * 8 + [c_int, c_int, c_int, c_float, c_int, P, P],
    "bh_bn_join_scratch_doubles": [c_int, c_int],
    "bh_bn_join_fwd": [P] * 13 + [c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_int, P, P],
    "bh_bn_join_bwd": [P] * 15 + [c_int, c_int, c_int, c_float, c_float, c_int, P, P, P],
    "bh_bn_join_bwd_remask": [P] * 16 + [c_int, c_int, c_int, c_float, c_float, c_int, P, P, P],
    "bh_tail_ws_doubles": [c_int, c_int, c_int],
    "bh_tail_scratch_floats": [c_int, c_int, c_int],
    "bh_tail_fwd": [P] * 11 + [c_int] * 6 + [c_float, c_float, c_int, P],
    "bh_tail_fwd_route": [P] * 11 + [c_int] * 6 + [c_float, c_float, c_int, c_int, P],
    "bh_tail_bwd": [P] * 17 + [c_int] * 6 + [c_float, c_int, P],
    "bh_tail_bwd_f": [P] * 17 + [c_int] * 6 + [c_float, c_int, c_int, P],
    "bh_synth_pairs": [P] * 5 + [c_int] * 5 + [c_float, c_float, P, P, P],
    "bh_maxpool3s2_fwd": [P, P, P, c_int, c_int, c_int, c_int, P],
    "bh_maxpool3s2_bwd": [P, P, P, c_int, c_int, c_int, c_int, P],
    "bh_gap_fwd": [P, P, c_int, c_int, c_int, P],
    "bh_gap_bwd": [P, P, c_int, c_int, c_int, P],
    "bh_add": [P, P, P, c_int64, P],
}


class BihomeLibError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise BihomeLibError(
            "bihome_amd: %s not found - the HIP extension is not built. Run `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            if os.environ.get("BIHOME_DEV_PARTIAL") == "1":      # development only: library under construction
                continue
            raise BihomeLibError("bihome_amd: %s does not export %s (stale build?)" % (LIB_PATH, name)) from e
        fn.argtypes = argtypes
        fn.restype = c_int
    lib.bh_conv_wgrad_det_bytes.argtypes = [POINTER(BhConvDesc)]      # (the one entry point that does not return a status)
    lib.bh_conv_wgrad_det_bytes.restype = c_int64
    lib.bh_stem7_wgrad_ws_bytes.argtypes = [POINTER(BhConvDesc)]
    lib.bh_stem7_wgrad_ws_bytes.restype = ctypes.c_size_t
    lib.bh_warp_bwd_img_scratch_doubles.argtypes = [c_int, c_int, c_int, c_int, c_int]
    lib.bh_warp_bwd_img_scratch_doubles.restype = ctypes.c_size_t
    if TUNING:
        lib.bh_debug_force_tile.argtypes = [c_int, c_int]
        lib.bh_debug_force_tile.restype = c_int
    return lib


lib = _load()
# (BIHOME_DETERMINISTIC=1 is read by kernels.py: the library holds no mode, every call carries its own bit - include/bihome.h
#  "Deterministic calls")
ROUTE_DETERMINISTIC, BN_DETERMINISTIC, F_DETERMINISTIC = 128, 32, 1
ROUTE_C3_PC = 512
ROUTE_WX3_PC = 1024       # include/bihome.h BH_ROUTE_WX3_PC
ROUTE_WX3_SHARED = 2048   # include/bihome.h BH_ROUTE_WX3_SHARED             # ... the persistent kernel for every launch it supports (default: where it is the faster one)
ROUTE_GEMM_X3 = 4096         # include/bihome.h BH_ROUTE_GEMM_X3: the generic kernel in three bf16 pieces (precision 2 / 4), opt-in
ROUTE_C3_TILE_WG = 256        # precision-4 3x3 fwd / dgrad on the one-workgroup-per-tile halo kernel instead of the persistent one


def check(rc, what):
    if rc != 0:
        kind = {-1: "BH_E_BADARG", -2: "BH_E_UNSUPPORTED"}.get(rc, "hipError_t %d" % rc)
        raise BihomeLibError("%s failed: %s" % (what, kind))
