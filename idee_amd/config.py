"""Flag surface of the reference's config.py (names, types and defaults of config.py:18-189),
kept so that `config = read_arguments(...)` objects interchange with the reference's drivers.
Table-driven re-statement; the option printer/saver of the reference is not reproduced."""
from __future__ import annotations

import argparse

_years = lambda a, b: [str(y) for y in range(a, b)]

# (name, type, default)
_FLAGS = [
    # general
    ("seed", int, 0), ("n_workers", int, 8), ("pin_memory", bool, True), ("batch_size", int, 1), ("name", str, "test"),
    ("dir_log", str, r"./log"), ("root_CERRA", str, r"../CERRA"), ("root_NOAA_CERRA", str, r"../NOAA_CERRA"),
    ("root_ERA5_Land", str, r"../ERA5-Land"), ("root_NOAA", str, r"../NOAA_CORDEX"),
    ("root_synthetic", str, r"../Synthetic/synthetic_CERRA"),
    ("encoder", str, "Mamba"), ("classifier", str, "CNN_3D"), ("codebook", str, "LFQ"), ("gpu_id", str, "0, 1, 2, 3"),
    ("nan_fill", float, 0.),
    # encoder
    ("in_channels_dynamic", int, 6), ("in_channels", int, 2), ("en_embed_dim", int, [16, 16]), ("en_depths", int, [2, 1]),
    ("en_patch_size", int, (1, 1, 1)), ("en_window_size", int, [(2, 4, 4), (8, 1, 1)]), ("en_mlp_ratio", float, 4.),
    ("en_drop_rate", float, 0.), ("en_drop_path_rate", float, 0.), ("en_patch_norm", bool, False),
    ("en_use_checkpoint", bool, False),
    ("en_n_heads", int, [2, 2]), ("en_attn_drop_rate", float, 0.0), ("en_qkv_bias", bool, True), ("en_qk_scale", float, None),
    ("d_state", int, [1, 1]), ("d_conv", int, [3, 3]), ("expand", int, [1, 1]), ("dt_min", int, 0.01), ("dt_max", int, 0.1),
    # quantiser / classifier
    ("codebook_size", int, 2), ("codebook_dim", int, 16), ("cls_dim", int, 16), ("cls_drop_rate", float, 0.),
    ("en_de_pretrained", str, None),
    # data
    ("years_train", str, _years(1984, 2018)), ("years_val", str, ["2018", "2019", "2020"]),
    ("years_test", str, ["2021", "2022", "2023", "2024"]),
    ("times_train", tuple, (1, 52 * 34)), ("times_val", tuple, (52 * 34 + 1, 52 * 40)), ("times_test", tuple, (52 * 40 + 1, 52 * 46)),
    ("delta_t", int, 8), ("window_size", int, 1), ("threshold", float, 26.0), ("alpha", float, 0.5), ("region", str, "EUR-11"),
    ("x_min", int, 0), ("x_max", int, 200), ("y_min", int, 0), ("y_max", int, 200),
    ("is_shuffle", bool, False), ("is_aug", bool, True), ("is_norm", bool, True), ("is_clima_scale", bool, True),
    # optimisation
    ("n_epochs", int, 100), ("optimizer", str, "Adam"), ("lr", float, 1e-3), ("weight_decay", float, 0.003),
    ("beta1", float, 0.9), ("beta2", float, 0.999), ("lr_scheduler", str, "cosine"), ("lr_warmup", int, 1e-6),
    ("lr_warmup_epochs", int, 2), ("lr_min", float, 1e-5), ("lr_decay_step", int, 20), ("lr_decay_rate", float, 0.9),
    ("lambda_ortho", float, 10.), ("lambda_commitment", float, 3.0), ("lambda_anomaly", float, 100.0),
    ("lambda_entropy", float, 0.1), ("diversity_gamma", float, 0.1),
    ("variables", str, ["wdir10", "si10", "al", "hcc", "lcc", "msl"]), ("variables_static", str, ["latitude", "longitude"]),
]


def add_all_arguments(parser):
    for name, typ, default in _FLAGS:
        parser.add_argument("--" + name, type=typ, default=default)
    return parser


def read_arguments(train=True, print=True, save=True, argv=None):
    """Same call shape as the reference's read_arguments (config.py:192-203); the option
    printing / config.txt / config.pkl side effects are the caller's (drivers are out of scope)."""
    parser = add_all_arguments(argparse.ArgumentParser())
    parser.add_argument("--phase", type=str, default="train")
    config = parser.parse_args(argv)
    config.phase = "train" if train else "test"
    return config


# attributes models/build.py reads with getattr() that are NOT flags of the reference's config.py (its flag surface stays equal
# to the reference's): act_dtype "f32" | "bf16" (activation storage), codebook_impl "LFQ" | "VQ" (BASELINE config #4's variants)
EXTENSION_FIELDS = ("act_dtype", "codebook_impl")


def default_config(**overrides):
    """Defaults of the flag table with overrides; `encoder` defaults to the reference's 'Mamba',
    which this build rejects — pass encoder='Swin_3D' or 'CNN_3D' (and in_channels=1 for
    synthetic data), as every reference run must too."""
    config = add_all_arguments(argparse.ArgumentParser()).parse_args([])
    for k, v in overrides.items():
        if not hasattr(config, k) and k not in EXTENSION_FIELDS:
            raise AttributeError(f"unknown config field {k}")
        setattr(config, k, v)
    return config
