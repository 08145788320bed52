"""Minimal stand-in for the OmegaConf DictConfig the reference passes to its constructors.

The reference reads its configs three ways: attribute access
(`model_config.use_voxel_embedding`, models/nerf_model.py:13), item access (`model_config["D"]`,
19-23) and `.get(key, default)` (26, 64; models/code_library.py:14-15).  A real DictConfig offers
all three; so does this dict subclass, which lets the drop-in types be constructed without
omegaconf installed.
"""


class AttrDict(dict):
    def __getattr__(self, key):
        try:
            v = self[key]
        except KeyError:
            raise AttributeError(key)
        return AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v

    def __setattr__(self, key, value):
        self[key] = value


# config/default_conf.yml:7-36
DEFAULT_MODEL_CONFIG = AttrDict(
    use_voxel_embedding=True,
    N_freq_xyz=10, N_freq_dir=4, N_freq_voxel=6,
    D=8, W=256, skips=[4], N_scn_voxel_size=16,
    inst_D=4, inst_W=128, inst_skips=[2], N_obj_voxel_size=8,
    N_samples=64, N_importance=64, frustum_bound=0.05, use_disp=False, perturb=1, noise_std=1,
    use_mask=True, N_vocab=1000, N_max_objs=64, N_obj_code_length=64, N_max_voxels=800000,
)


def default_model_config(**overrides):
    c = AttrDict(DEFAULT_MODEL_CONFIG)
    c.update(overrides)
    return c
