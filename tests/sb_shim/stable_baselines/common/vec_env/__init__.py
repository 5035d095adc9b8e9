"""Interface-only stand-in of stable_baselines.common.vec_env (see tests/sb_shim/README.md)."""
from abc import ABC, abstractmethod


class VecEnv(ABC):
    """An abstract asynchronous, vectorized environment: num_envs, observation_space, action_space."""
    metadata = {'render.modes': ['human', 'rgb_array']}

    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.action_space = action_space

    @abstractmethod
    def reset(self):
        pass

    @abstractmethod
    def step_async(self, actions):
        pass

    @abstractmethod
    def step_wait(self):
        pass

    @abstractmethod
    def close(self):
        pass

    @abstractmethod
    def get_attr(self, attr_name, indices=None):
        pass

    @abstractmethod
    def set_attr(self, attr_name, value, indices=None):
        pass

    @abstractmethod
    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        pass

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def get_images(self, *args, **kwargs):
        raise NotImplementedError

    def render(self, *args, **kwargs):
        raise NotImplementedError

    @property
    def unwrapped(self):
        return self

    def getattr_depth_check(self, name, already_found):
        if hasattr(self, name) and already_found:
            return "{0}.{1}".format(type(self).__module__, type(self).__name__)
        return None

    def _get_indices(self, indices):
        if indices is None:
            indices = range(self.num_envs)
        elif isinstance(indices, int):
            indices = [indices]
        return indices


def wrap_like_base_rl_model(env):
    """What BaseRLModel.__init__ / set_env do with the env they are given (requires_vec_env policies): anything that is not a
    VecEnv is wrapped in DummyVecEnv([lambda: env]) — returned here as the string 'DummyVecEnv' instead of a wrapper."""
    return env if isinstance(env, VecEnv) else "DummyVecEnv"
