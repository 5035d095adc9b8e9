"""Stand-in for Shapely 1.6 (see ../README.md)."""
