"""`gym` is an optional dependency: use it when importable, otherwise a 20-line stand-in with the
pieces the reference touches (gym.Env, spaces.Box, gym.make('pkg:Id'), envs.registration.register;
reference gym_ran_slice/__init__.py:1-8, ran_slice.py:12-28, scenario_creator.py:181)."""
try:  # pragma: no cover - gym is not installed in the build image
    import gym as _gym
    from gym import spaces
    from gym.envs.registration import register
    Env = _gym.Env
    make = _gym.make
    HAVE_GYM = True
except Exception:
    import importlib
    HAVE_GYM = False
    _registry = {}

    class Env:
        metadata = {}

        def reset(self):
            raise NotImplementedError

        def step(self, action):
            raise NotImplementedError

        def render(self):
            pass

        def close(self):
            pass

    class _Box:
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    class _Spaces:
        Box = _Box

    spaces = _Spaces()

    def register(id, entry_point, **kwargs):
        _registry[id] = entry_point

    def make(spec, **kwargs):
        pkg, _, env_id = spec.partition(':')
        if pkg and env_id:
            importlib.import_module(pkg)
        else:
            env_id = spec
        mod_name, _, cls_name = _registry[env_id].partition(':')
        return getattr(importlib.import_module(mod_name), cls_name)(**kwargs)
