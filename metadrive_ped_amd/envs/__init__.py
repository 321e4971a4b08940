from metadrive_ped_amd.envs.metadrive_env import BatchedMetaDriveEnv, BatchedSafeMetaDriveEnv  # noqa: F401
