"""Drop-in for the reference's `envs` package (envs/__init__.py:1-5): importing it registers 'AtcEnv-v0' with gym when
gym (or gymnasium) is installed; without gym the classes are still importable and usable directly.

Only two conditions are tolerated silently: the package is not installed (ImportError) and the id is already registered
(a re-import; gym raises gym.error.Error, gymnasium only warns).  Any other failure of `register` propagates — a broken
registration must not stay hidden until `gym.make` fails."""


def _register(module):
    try:
        registration = __import__(module + ".envs.registration", fromlist=["register"])
    except ImportError:
        return False
    try:
        registration.register(id='AtcEnv-v0', entry_point='envs.atc.atc_gym:AtcGym')
    except Exception as exc:
        err = getattr(__import__(module), "error", None)
        already = isinstance(exc, getattr(err, "Error", ())) and "re-register" in str(exc).lower()
        if not already:
            raise
    return True


if not _register("gym"):
    _register("gymnasium")
