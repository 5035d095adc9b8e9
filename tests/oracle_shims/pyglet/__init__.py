"""Stand-in for pyglet (render-only dependency of the reference): the GL calls the reference makes are no-ops."""
from . import gl  # noqa: F401
