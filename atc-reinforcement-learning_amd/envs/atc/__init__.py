from . import atc_gym  # noqa: F401  (mirrors the reference's envs/atc/__init__.py:1)
