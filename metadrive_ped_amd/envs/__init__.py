from metadrive_ped_amd.envs.metadrive_env import (BatchedMetaDriveEnv, BatchedSafeMetaDriveEnv,  # noqa: F401
                                                  BatchedVaryingDynamicsEnv)
from metadrive_ped_amd.envs.marl_env import (BatchedMultiAgentBidirectionEnv, BatchedMultiAgentBottleneckEnv, BatchedMultiAgentIntersectionEnv,  # noqa: F401
                                            BatchedMultiAgentTinyInter, BatchedMultiAgentRacingEnv,
                                             BatchedMultiAgentMetaDrive, BatchedMultiAgentParkingLotEnv,
                                             BatchedMultiAgentRoundaboutEnv, BatchedMultiAgentTollgateEnv)
