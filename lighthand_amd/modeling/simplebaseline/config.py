"""Model configuration of the SimpleBaseline path with the reference's field names and defaults
(src/modeling/simplebaseline/config.py:19-59): ``config.MODEL.EXTRA.NUM_LAYERS`` etc.  The reference builds
this with EasyDict; a small attribute dict stands in.  ``update_config`` merges a yaml file the same way."""
import yaml


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(d):
    return AttrDict({k: _wrap(v) if isinstance(v, dict) else v for k, v in d.items()})


def default_config(num_layers=50):
    return _wrap({
        "OUTPUT_DIR": "", "LOG_DIR": "", "DATA_DIR": "", "GPUS": "0", "WORKERS": 4, "PRINT_FREQ": 20,
        "MODEL": {
            "NAME": "pose_resnet", "INIT_WEIGHTS": True, "PRETRAINED": "", "NUM_JOINTS": 21,
            "IMAGE_SIZE": [256, 256], "STYLE": "pytorch",
            "EXTRA": {"NUM_LAYERS": num_layers, "DECONV_WITH_BIAS": False, "NUM_DECONV_LAYERS": 3,
                      "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_DECONV_KERNELS": [4, 4, 4], "FINAL_CONV_KERNEL": 1,
                      "TARGET_TYPE": "gaussian", "HEATMAP_SIZE": [64, 64], "SIGMA": 2},
        },
        "LOSS": {"USE_TARGET_WEIGHT": True},
        "TRAIN": {"LR": 0.001, "OPTIMIZER": "adam", "BATCH_SIZE": 32, "END_EPOCH": 140},
    })


config = default_config()


def update_config(config_file, cfg=None):
    cfg = config if cfg is None else cfg
    with open(config_file) as f:
        exp = yaml.safe_load(f) or {}

    def merge(dst, src):
        for k, v in src.items():
            if k not in dst:
                raise ValueError("{} not exist in config.py".format(k))
            if isinstance(v, dict) and isinstance(dst[k], dict):
                merge(dst[k], v)
            else:
                dst[k] = v
    merge(cfg, exp)
    return cfg


def get_model_name(cfg):
    extra = cfg.MODEL.EXTRA
    name = "{model}_{num_layers}".format(model=cfg.MODEL.NAME, num_layers=extra.NUM_LAYERS)
    deconv_suffix = "".join("d{}".format(nf) for nf in extra.NUM_DECONV_FILTERS)
    full = "{height}x{width}_{name}_{deconv_suffix}".format(height=cfg.MODEL.IMAGE_SIZE[1], width=cfg.MODEL.IMAGE_SIZE[0],
                                                           name=name, deconv_suffix=deconv_suffix)
    return name, full
