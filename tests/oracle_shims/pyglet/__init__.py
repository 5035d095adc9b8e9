"""Stand-in for pyglet (render-only dependency of the reference; never exercised)."""
