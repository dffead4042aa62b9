"""Configuration dictionaries for the iPOKE hot path.

The reference reads these from YAML (config/second_stage.yaml,
config/first_stage.yaml, config/poke_encoder.yaml, config/img_encoder.yaml and
config/pretrained_models/*.yaml).  Only the keys the hot path consumes are
restated here, with the reference's shipped values as defaults, so that the
benchmark configs c1-c5 of BASELINE.json can be built without the YAML files.
A user may equally pass dictionaries loaded from the reference's YAMLs: the
key names are the same.
"""
import copy

# second_stage.yaml:62 (num_steps), :63 (factor)
SHIPPED_NUM_STEPS = [10, 5, 5, 4, 4, 4, 3, 3, 3, 2, 2, 2, 1, 1, 1]


def flow_arch(z_dim, hidden=None, num_steps=None, factor=16, h_channels=128):
    """`architecture` section consumed by SupervisedMacowTransformer (reference INN.py:451-467).

    ``flow_mid_channels = flow_mid_channels_factor * z_dim`` (second_stage_video.py:107-108); the shipped
    factor is 64 for the z_dim=32 models and 32 for the z_dim=64 models (config/pretrained_models/*.yaml:15),
    i.e. 2048 hidden channels in every shipped flow.
    """
    mid_factor = 64 if z_dim <= 32 else 32
    return {
        "attention": False,
        "flow_attn_heads": 4,
        "kernel_size": [2, 3],
        "coupling_type": "conv",
        "num_steps": list(SHIPPED_NUM_STEPS if num_steps is None else num_steps),
        "factor": factor,
        "activation": "elu",
        "transform": "affine",
        "prior_transform": "affine",
        "condition_nice": False,
        "augmented_input": False,
        "multistack": False,
        "cond_conv": False,
        "cond_conv_hidden_channels": 256,
        "reshape": "none",
        "p_dropout": 0.0,
        "flow_mid_channels_factor": mid_factor,
        "flow_in_channels": z_dim,
        "flow_mid_channels": mid_factor * z_dim if hidden is None else hidden,
        "h_channels": h_channels,
    }


def reduced_flow_arch():
    """Small full-topology flow used by the parity tests."""
    return flow_arch(16, hidden=64, num_steps=[2, 1, 1], factor=4)


def first_stage_config(spatial_size=128, z_dim=32, n_frames=16):
    """first_stage.yaml restated; ``n_frames`` = frames per clip including x0 (max_frames = n_frames-1)."""
    enc = [64, 128, 256, 256, 256]
    dec = [256, 256, 256, 128, 64]
    if spatial_size == 64:                      # first_stage.yaml:51,60 comments
        enc, dec = enc[:-1], dec[1:]
    return {
        "data": {"spatial_size": (spatial_size, spatial_size), "max_frames": n_frames - 1, "batch_size": 20},
        "training": {"lr": 2e-4, "weight_decay": 1e-5, "w_kl": 1e-7, "w_l1": 10, "w_vgg": 10, "full_sequence": True},
        "logging": {"bs_i3d": 8, "n_samples_fvd": 1000},
        "architecture": {
            "ENC_M_channels": enc, "decoder_factor": 32, "z_dim": z_dim, "norm": "group", "CN_content": "spade",
            "CN_motion": "ADAIN", "spectral_norm": True, "running_stats": False, "n_gru_layers": 4,
            "dec_channels": dec, "min_spatial_size": 8, "motion_bias": True, "deterministic": False,
        },
    }


def encoder2d_config(spatial_size=128, nf_in=2, flow_ae=True):
    """poke_encoder.yaml (nf_in=2, flow_ae) / img_encoder.yaml (nf_in=3) restated."""
    arch = {"conv": True, "nf_in": nf_in, "nf_max": 64, "min_spatial_size": 8, "deterministic": True}
    if nf_in == 2:
        arch.update(flow_ae=flow_ae, poke_and_image=False)
    return {"data": {"spatial_size": (spatial_size, spatial_size)}, "architecture": arch}


def second_stage_config(spatial_size=128, z_dim=64, n_frames=16, batch_size=20, n_epochs=100, arch=None):
    """second_stage.yaml restated with the architecture derived as PokeMotionModel.__init__ does."""
    return {
        "general": {"experiment": "second_stage", "debug": False, "test": "none", "seed": 42},
        "data": {"spatial_size": (spatial_size, spatial_size), "max_frames": n_frames - 1, "batch_size": batch_size,
                 "n_pokes": 5},
        "architecture": copy.deepcopy(arch) if arch is not None else flow_arch(z_dim),
        "training": {"lr": 1e-3, "weight_decay": 1e-5, "n_epochs": n_epochs, "max_batches_per_epoch": 2000,
                     "lr_scaling": True, "lr_scaling_max_it": 500, "custom_lr_decrease": True, "mixed_prec": False,
                     "full_seq": True, "spatial_mean": False, "use_adabelief": False},
        "testing": {"n_samples_per_data_point": 5},
        "logging": {"log_train_prog_at": 200, "n_samples": 4, "n_log_images": 8, "n_fvd_samples": 1000},
        "conditioner": {"use": True},
        "first_stage": first_stage_config(spatial_size, z_dim, n_frames),
        "poke_embedder": encoder2d_config(spatial_size, 2),
        "conditioner_model": encoder2d_config(spatial_size, 3),
    }


# The five BASELINE.json configs
BENCH_CONFIGS = {
    "c1": dict(name="plants_64", spatial_size=64, z_dim=32, n_frames=16, batch_size=2),
    "c2": dict(name="plants_128", spatial_size=128, z_dim=64, n_frames=16, batch_size=20),
    "c3": dict(name="iper_128", spatial_size=128, z_dim=32, n_frames=16, batch_size=40),
    "c4": dict(name="first_stage_128", spatial_size=128, z_dim=32, n_frames=16, batch_size=20),
    "c4gan": dict(name="first_stage_128_gan", spatial_size=128, z_dim=32, n_frames=16, batch_size=20),
    "c5": dict(name="h36m_128", spatial_size=128, z_dim=64, n_frames=16, batch_size=32),
    "fvd": dict(name="fvd_i3d", spatial_size=128, z_dim=64, n_frames=16, batch_size=8),      # first_stage.yaml:92 bs_i3d
}
