"""Minimal stand-in for the `gym` package (not installed in this image, no network).

Only used by tools/gen_golden.py to import the reference in this container
(SURVEY.md §8c shim 2).  It is NOT part of the product and never travels to the
GPU box in any way that matters: nothing under tests/, bench.py or the package
imports it.
"""
import importlib
from . import spaces  # noqa: F401
from .envs import registration  # noqa: F401


class Env:
    metadata = {}

    def reset(self):
        raise NotImplementedError

    def step(self, action):
        raise NotImplementedError


class Wrapper(Env):
    def __init__(self, env):
        self.env = env


def make(spec, **kwargs):
    pkg, _, env_id = spec.partition(':')
    importlib.import_module(pkg)
    entry = registration.registry[env_id]
    mod_name, _, cls_name = entry.partition(':')
    cls = getattr(importlib.import_module(mod_name), cls_name)
    return cls(**kwargs)
