"""Minimal stand-in for gym 0.15 (see ../README.md): only what the reference touches at import/ctor time."""
import importlib

from . import spaces  # noqa: F401
from .envs import registration as _registration


class Env:
    metadata = {}
    reward_range = (-float("inf"), float("inf"))
    action_space = None
    observation_space = None

    def step(self, action):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def render(self, mode="human"):
        raise NotImplementedError

    def close(self):
        pass

    def seed(self, seed=None):
        return []


def make(env_id, **kwargs):
    entry = _registration.registry[env_id]
    mod_name, cls_name = entry.split(":")
    cls = getattr(importlib.import_module(mod_name), cls_name)
    return cls(**kwargs)
