"""Stand-in for numba: `jit` is an identity decorator (see ../README.md)."""


def jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def wrap(fn):
        return fn

    return wrap
