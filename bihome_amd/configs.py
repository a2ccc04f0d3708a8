"""The kwargs contract of the two shipped biHomE experiments this build covers.

Values (not text) of MODEL.BACKBONE / MODEL.HEAD / SOLVER from the reference's
config/s-coco/zeng-bihome-lr-1e-3.yaml and config/s-coco/detone-bihome-lr-5e-3.yaml; the reference
splats them as **kwargs into `Model.__init__` (train.py:679,690).  pds-coco differs only in the
data transform (photometric max_delta 32 instead of 0).
"""
import copy

ZENG_BIHOME = {
    "MODEL": {
        "BACKBONE": {
            "NAME": "Rethinking", "VARIANT": "DoubleLine", "IMAGE_SIZE": 128, "RESNET_BLOCK": "ResNet34",
            "PRETRAINED_RESNET": False,      # reference: True (ImageNet URL); no network here
            "IMAGE_KEY": ["image"], "PATCH_KEYS": ["patch_1", "patch_2"],
            "TARGET_KEYS": ["pf_hat_12", "pf_hat_21"],
        },
        "HEAD": {
            "NAME": "PerceptualHead", "PATCH_SIZE": 128, "PATCH_KEYS": ["patch_1", "patch_2"],
            "DELTA_HAT_KEYS": [], "PF_KEYS": ["pf_hat_12", "pf_hat_21"],
            "RANSAC_HYPOTHESIS_NO": 1, "POINTS_PER_HYPOTHESIS": 128,
            "AUXILIARY_RESNET": "resnet34", "AUXILIARY_RESNET_OUTPUT_LAYER": 1,
            "TRIPLET_LOSS": "double-line", "TRIPLET_AGGREGATION": "channel-agnostic",
            "TRIPLET_MARGIN": "inf", "TRIPLET_DISTANCE": "l1", "TRIPLET_MU": 0.01,
            "MASK_KEYS": [], "SAMPLING_STRATEGY": "downsample-mask",
        },
    },
    "SOLVER": {"OPTIMIZER": "Adam", "MOMENTUM_1": 0.9, "MOMENTUM_2": 0.999, "LR": 0.001,
               "MILESTONES": [30000, 60000, 90000], "LR_DECAY": 0.1, "LOSS": "biHomE"},
    "DATA": {"BATCH_SIZE": 64, "RHO": 32, "PATCH_SIZE": 128, "PHOTOMETRIC_MAX_DELTA": 0},
}

DETONE_BIHOME = {
    "MODEL": {
        "BACKBONE": {
            "NAME": "ResNet34", "VARIANT": "DoubleLine", "PRETRAINED_RESNET": False,
            "IMAGE_KEY": ["image"], "PATCH_KEYS": ["patch_1", "patch_2"],
            "TARGET_KEYS": ["delta_hat_12", "delta_hat_21"],
        },
        "HEAD": {
            "NAME": "PerceptualHead", "PATCH_SIZE": 128, "PATCH_KEYS": ["patch_1", "patch_2"],
            "DELTA_HAT_KEYS": ["delta_hat_12", "delta_hat_21"], "PF_KEYS": [],
            "RANSAC_HYPOTHESIS_NO": -1, "POINTS_PER_HYPOTHESIS": -1,
            "AUXILIARY_RESNET": "resnet34", "AUXILIARY_RESNET_OUTPUT_LAYER": 1,
            "TRIPLET_LOSS": "double-line", "TRIPLET_AGGREGATION": "channel-agnostic",
            "TRIPLET_MARGIN": "inf", "TRIPLET_DISTANCE": "l1", "TRIPLET_MU": 0.01,
            "MASK_KEYS": [], "SAMPLING_STRATEGY": "downsample-mask",
        },
    },
    "SOLVER": {"OPTIMIZER": "Adam", "MOMENTUM_1": 0.9, "MOMENTUM_2": 0.999, "LR": 0.005,
               "MILESTONES": [30000, 60000, 90000], "LR_DECAY": 0.1, "LOSS": "biHomE"},
    "DATA": {"BATCH_SIZE": 64, "RHO": 32, "PATCH_SIZE": 128, "PHOTOMETRIC_MAX_DELTA": 0},
}


# the supervised baselines of the same two backbones (config/s-coco/zeng-orig-lr-1e-3.yaml, detone-orig-lr-5e-3.yaml):
# one direction only (VARIANT OneLine), NoOpHead, a torch loss on (target, output) - train.py:318-322
ZENG_ORIG = {
    "MODEL": {
        "BACKBONE": {
            "NAME": "Rethinking", "VARIANT": "OneLine", "IMAGE_SIZE": 128, "RESNET_BLOCK": "ResNet34",
            "PRETRAINED_RESNET": False, "IMAGE_KEY": ["image"], "PATCH_KEYS": ["patch_1", "patch_2"],
            "TARGET_KEYS": ["pf_hat_12"],
        },
        "HEAD": {"NAME": "NoOpHead", "TARGET_GEN": "all_points",
                 "LEARNING_KEYS": ["target", "pf_hat_12", "delta", "pf_hat_12"]},
    },
    "SOLVER": {"OPTIMIZER": "Adam", "MOMENTUM_1": 0.9, "MOMENTUM_2": 0.999, "LR": 0.001,
               "MILESTONES": [30000, 60000, 90000], "LR_DECAY": 0.1, "LOSS": "SmoothL1Loss"},
    "DATA": {"BATCH_SIZE": 64, "RHO": 32, "PATCH_SIZE": 128, "PHOTOMETRIC_MAX_DELTA": 0, "TARGET_GEN": "all_points"},
}

DETONE_ORIG = {
    "MODEL": {
        "BACKBONE": {
            "NAME": "ResNet34", "VARIANT": "OneLine", "PRETRAINED_RESNET": False,
            "IMAGE_KEY": ["image"], "PATCH_KEYS": ["patch_1", "patch_2"], "TARGET_KEYS": ["delta_hat_12"],
        },
        "HEAD": {"NAME": "NoOpHead", "TARGET_GEN": "4_points",
                 "LEARNING_KEYS": ["delta", "delta_hat_12", "delta", "delta_hat_12"]},
    },
    "SOLVER": {"OPTIMIZER": "Adam", "MOMENTUM_1": 0.9, "MOMENTUM_2": 0.999, "LR": 0.005,
               "MILESTONES": [30000, 60000, 90000], "LR_DECAY": 0.1, "LOSS": "MSELoss"},
    "DATA": {"BATCH_SIZE": 64, "RHO": 32, "PATCH_SIZE": 128, "PHOTOMETRIC_MAX_DELTA": 0, "TARGET_GEN": "4_points"},
}


# Zhang et al. "Content-Aware" baselines (config/s-coco/zhang-orig-lr-1e-2.yaml: ContentAware backbone + TripletHead;
# zhang-bihome-lr-1e-2.yaml: the same backbone under the biHomE PerceptualHead with directly regressed offsets)
ZHANG_ORIG = {
    "MODEL": {
        "BACKBONE": {
            "NAME": "ContentAware", "VARIANT": "DoubleLine", "IMAGE_SIZE": 128, "PRETRAINED_RESNET": False,      # reference: True
            "IMAGE_KEY": ["image"], "PATCH_KEYS": ["patch_1", "patch_2"], "MASK_KEYS": ["mask_1", "mask_2"], "FIX_MASK": True,
            "FEATURE_KEYS": ["feature_1", "feature_2"], "TARGET_KEYS": ["delta_hat_12", "delta_hat_21"],
        },
        "HEAD": {
            "NAME": "TripletHead", "VARIANT": "DoubleLine", "PATCH_SIZE": 128, "PATCH_KEYS": ["patch_1", "patch_2"],
            "MASK_KEYS": ["mask_1", "mask_2"], "FEATURE_KEYS": ["feature_1", "feature_2"],
            "TARGET_KEYS": ["delta_hat_12", "delta_hat_21"], "LD": 2, "MU": 0.01, "TRIPLET_MARGIN": 1.0,
            "TRIPLET_AGGREGATION": "channel-agnostic",
        },
    },
    "SOLVER": {"OPTIMIZER": "Adam", "MOMENTUM_1": 0.9, "MOMENTUM_2": 0.999, "LR": 0.01,
               "MILESTONES": [30000, 60000, 90000], "LR_DECAY": 0.1, "LOSS": "TripletLoss"},
    "DATA": {"BATCH_SIZE": 64, "RHO": 32, "PATCH_SIZE": 128, "PHOTOMETRIC_MAX_DELTA": 0},
}
ZHANG_BIHOME = copy.deepcopy(ZHANG_ORIG)
ZHANG_BIHOME["MODEL"]["HEAD"] = copy.deepcopy(DETONE_BIHOME["MODEL"]["HEAD"])
ZHANG_BIHOME["SOLVER"]["LOSS"] = "biHomE"


def _ihome(base):
    """iHomE (one-line) variant of a biHomE config: one direction, hinge with a numeric margin (PerceptualHead.py:465-538).
    No yaml for it ships upstream (train.py:330 names the loss 'iHomE'); the kwargs follow the biHomE configs."""
    cfg = copy.deepcopy(base)
    cfg["MODEL"]["BACKBONE"]["VARIANT"] = "OneLine"
    cfg["MODEL"]["BACKBONE"]["TARGET_KEYS"] = cfg["MODEL"]["BACKBONE"]["TARGET_KEYS"][:1]
    h = cfg["MODEL"]["HEAD"]
    h["TRIPLET_LOSS"], h["TRIPLET_MARGIN"] = "one-line", 1.0
    h["DELTA_HAT_KEYS"], h["PF_KEYS"] = h["DELTA_HAT_KEYS"][:1], h["PF_KEYS"][:1]
    cfg["SOLVER"]["LOSS"] = "iHomE"
    return cfg


def _multihead(base):
    """multihead_resnet_loss variant (PerceptualHead.py:230-235,245-315): TRIPLET_LOSS '' - the head returns the extractor
    features of patch_2 and of the warped patch_1 and the driver applies a torch loss to them (train.py:318-322).  No
    yaml for it ships upstream; one direction, kwargs as in the biHomE configs."""
    cfg = _ihome(base)
    cfg["MODEL"]["HEAD"]["TRIPLET_LOSS"] = ""
    cfg["MODEL"]["HEAD"]["TRIPLET_MARGIN"] = "inf"
    cfg["SOLVER"]["LOSS"] = "L1Loss"
    return cfg


def get(name):
    """'zeng-bihome' / 'detone-bihome' = config/s-coco/*; the '-pds' variants = config/pds-coco/* (the two trees differ
    only in HomographyNetPrep's photometric max_delta, 0 vs 32, and the log dir).  'zeng-bihome-rgb256' is the
    build-side extension BASELINE.json configs[4] names (256x256 RGB patches, 6-channel stem; no upstream
    counterpart - SURVEY.md 0)."""
    if name == "zeng-bihome-rgb256":
        cfg = copy.deepcopy(ZENG_BIHOME)
        cfg["MODEL"]["BACKBONE"].update(IMAGE_SIZE=256, PATCH_CHANNELS=3)
        cfg["MODEL"]["HEAD"].update(PATCH_SIZE=256)
        cfg["DATA"].update(BATCH_SIZE=32, RHO=64, PATCH_SIZE=256, PATCH_CHANNELS=3)
        return cfg
    if name in ("zeng-multihead", "detone-multihead"):
        return _multihead(ZENG_BIHOME if name == "zeng-multihead" else DETONE_BIHOME)
    if name in ("zeng-ihome", "detone-ihome"):
        return _ihome(ZENG_BIHOME if name == "zeng-ihome" else DETONE_BIHOME)
    base = name[:-4] if name.endswith("-pds") else name
    cfg = copy.deepcopy({"zeng-bihome": ZENG_BIHOME, "detone-bihome": DETONE_BIHOME, "zeng-orig": ZENG_ORIG,
                         "detone-orig": DETONE_ORIG, "zhang-orig": ZHANG_ORIG, "zhang-bihome": ZHANG_BIHOME}[base])
    if name.endswith("-pds"):
        cfg["DATA"]["PHOTOMETRIC_MAX_DELTA"] = 32
    return cfg
