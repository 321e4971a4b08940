"""createGymWrapper (metadrive/envs/gym_wrapper.py:37-127): the old 4-tuple gym API over a batched env.

`step` returns (obs, reward, done, info) with done = terminated | truncated (element-wise on the batch tensors),
`reset` returns the observation alone, `seed()` does nothing, every other attribute is the inner env's.  The
spaces are this package's own Box / Discrete / MultiDiscrete (envs/spaces.py: neither gym nor gymnasium is a
dependency), which carry the same fields in both APIs.
"""


def createGymWrapper(inner_class):
    class GymEnvWrapper:
        @classmethod
        def default_config(cls):
            return inner_class.default_config()

        def __init__(self, config=None):
            object.__setattr__(self, "_inner", inner_class(config))

        def step(self, actions):
            o, r, tm, tc, i = self._inner.step(actions)
            return o, r, tm | tc, i

        def reset(self, *, seed=None, options=None):
            obs, _ = self._inner.reset(seed) if seed is not None else self._inner.reset()
            return obs

        def close(self):
            self._inner.close()

        def seed(self, seed=None):
            pass

        def __getattr__(self, name):
            return getattr(self._inner, name)

        def __setattr__(self, name, value):
            if hasattr(self._inner, name):
                setattr(self._inner, name, value)
            else:
                object.__setattr__(self, name, value)

    GymEnvWrapper.__name__ = "Gym" + inner_class.__name__
    return GymEnvWrapper
