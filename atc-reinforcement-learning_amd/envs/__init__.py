"""Drop-in for the reference's `envs` package (envs/__init__.py:1-5): importing it registers 'AtcEnv-v0' with gym when
gym (or gymnasium) is installed; without gym the classes are still importable and usable directly."""
try:  # pragma: no cover - gym is not installed in the build image
    from gym.envs.registration import register
    register(id='AtcEnv-v0', entry_point='envs.atc.atc_gym:AtcGym')
except Exception:  # gym missing or id already registered
    try:
        from gymnasium.envs.registration import register as _register
        _register(id='AtcEnv-v0', entry_point='envs.atc.atc_gym:AtcGym')
    except Exception:
        pass
