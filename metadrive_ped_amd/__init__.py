"""metadrive_ped_amd -- MI355X-native batched MetaDrive step() (one hot path)."""
from metadrive_ped_amd.config import make_config  # noqa: F401

__all__ = ["make_config", "BatchedMetaDriveEnv"]


def __getattr__(name):
    if name == "BatchedMetaDriveEnv":
        from metadrive_ped_amd.envs.metadrive_env import BatchedMetaDriveEnv
        return BatchedMetaDriveEnv
    raise AttributeError(name)
