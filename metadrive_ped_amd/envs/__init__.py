from metadrive_ped_amd.envs.metadrive_env import BatchedMetaDriveEnv, BatchedSafeMetaDriveEnv  # noqa: F401
from metadrive_ped_amd.envs.marl_env import BatchedMultiAgentIntersectionEnv, BatchedMultiAgentRoundaboutEnv  # noqa: F401
