"""Sub-batches in flight on separate HIP streams: double-buffered stepping.

One md_step launch ends when its slowest environment ends; with 4096 environments on 256 CUs x 7 resident
workgroups the last third of a launch runs on a draining chip, and the next launch of the SAME batch cannot start
before it (it needs every observation to pick every action).  Split the batch into S sub-batches that are stepped
independently -- the policy acts on sub-batch k while the engine steps sub-batch k+1, the usual double-buffered
rollout loop -- and the tail of one launch overlaps the body of another: measured on the MI355X, 4096 envs x 240
beams, random actions: 1 x 4096: 76 us per 4096 agent-steps, 2 x 2048: 61 us, 4 x 1024: 58 us.

    envs = SubBatchedEnvs(BatchedMetaDriveEnv, dict(num_envs=4096, num_scenarios=4096), sub_batches=2)
    obs = [o for o, _ in envs.reset()]
    while training:
        for k, env in enumerate(envs.envs):
            with envs.on(k):                       # everything inside runs on sub-batch k's stream
                a = policy(obs[k])
                obs[k], r, term, trunc, info = env.step(a)
    envs.synchronize()

Sub-batch k holds the environments [k E/S, (k+1) E/S) of the whole batch: the same scenarios, seeds and results as
one env of E environments (scenario seed = start_seed + (env_seed_offset + e) % num_scenarios).  Nothing here is
new device code: BatchedEngine launches on torch's current stream, this class only owns the streams.
"""
import copy


class SubBatchedEnvs:
    def __init__(self, env_cls, config, sub_batches=2):
        config = dict(config or {})
        E = int(config.get("num_envs", 1))
        S = int(sub_batches)
        if S < 1 or E % S:
            raise ValueError("num_envs={} is not a multiple of sub_batches={}".format(E, S))
        base = int(config.get("env_seed_offset", 0))
        self.num_envs, self.sub_batches = E, S
        self.envs = []
        for k in range(S):
            c = copy.deepcopy(config)
            c["num_envs"] = E // S
            c["env_seed_offset"] = base + k * (E // S)
            self.envs.append(env_cls(c))
        self.streams = None
        self._hosts = None

    def build_host(self):
        """Generate every sub-batch's maps and scenes on the host (fork pool) without touching the GPU; the engines
        are created from them at the first reset()."""
        from metadrive_ped_amd.engine import HostScene
        self._hosts = [HostScene(e.config) for e in self.envs]
        return self._hosts

    def _ensure_streams(self):
        if self.streams is None:
            import torch
            self.streams = [torch.cuda.Stream(device=e.config["device"]) for e in self.envs]

    def on(self, k):
        """Context manager: torch's current stream becomes sub-batch k's stream."""
        import torch
        self._ensure_streams()
        return torch.cuda.stream(self.streams[k])

    def reset(self, seed=None):
        out = []
        for k, e in enumerate(self.envs):
            with self.on(k):
                if e.engine is None and self._hosts is not None and hasattr(e, "lazy_init"):
                    e.lazy_init(host=self._hosts[k])
                out.append(e.reset(seed) if seed is not None else e.reset())
        self._hosts = None
        return out

    def step(self, actions):
        """One step of every sub-batch, each on its own stream; `actions` is a list of S action batches.  Returns the
        list of the S step results.  The caller's stream is NOT made to wait: use `on(k)` / `synchronize()`."""
        out = []
        for k, e in enumerate(self.envs):
            with self.on(k):
                out.append(e.step(actions[k]))
        return out

    def synchronize(self):
        self._ensure_streams()
        for s in self.streams:
            s.synchronize()

    def close(self):
        for e in self.envs:
            e.close()
