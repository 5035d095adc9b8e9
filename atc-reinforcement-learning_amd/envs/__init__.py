"""Drop-in for the reference's `envs` package (envs/__init__.py:1-5): importing it registers 'AtcEnv-v0' with gym when
gym (or gymnasium) is installed; without gym the classes are still importable and usable directly.

What is tolerated: the package is not installed, or is installed but cannot be imported (an old gym on a new NumPy raises
AttributeError and the like at import time) — a warning, then gymnasium is tried, and the classes stay usable without either;
and the id is already registered (a re-import), which is asked of the registry itself before registering instead of read off
an error message.  A failure of `register` itself propagates — a broken registration must not stay hidden until `gym.make`
fails."""
import warnings

_ID = 'AtcEnv-v0'


def _registered(registration):
    """True if the registry of this gym flavour already holds the id (gym <= 0.21: registry.env_specs, later: a dict)."""
    reg = getattr(registration, "registry", None)
    specs = getattr(reg, "env_specs", reg)
    try:
        return specs is not None and _ID in specs
    except TypeError:
        return False


def _register(module):
    try:
        registration = __import__(module + ".envs.registration", fromlist=["register"])
    except ImportError:
        return False
    except Exception as exc:   # installed but broken: say so once, fall through to the next flavour
        warnings.warn("envs: %s is installed but failed to import (%s: %s); '%s' is not registered with it"
                      % (module, type(exc).__name__, exc, _ID))
        return False
    if not _registered(registration):
        registration.register(id=_ID, entry_point='envs.atc.atc_gym:AtcGym')
    return True


if not _register("gym"):
    _register("gymnasium")
