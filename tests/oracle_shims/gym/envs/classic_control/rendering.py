"""Recording stand-in for gym.envs.classic_control.rendering: keeps the geometry a caller builds (no window, no GL), so
that the golden generator can capture WHAT the reference's render() draws (tests/golden/g10_render_geometry.json)."""


class Geom:
    def __init__(self):
        self.attrs = []
        self.color = None
        self.linewidth = 1

    def set_color(self, r, g, b):
        self.color = (r, g, b)

    def set_linewidth(self, w):
        self.linewidth = w

    def add_attr(self, attr):
        self.attrs.append(attr)


class FilledPolygon(Geom):
    def __init__(self, v):
        super().__init__()
        self.v = v


class PolyLine(Geom):
    def __init__(self, v, close):
        super().__init__()
        self.v = v
        self.close = close


class Transform:
    def __init__(self, translation=(0.0, 0.0), rotation=0.0, scale=(1, 1)):
        self.translation = translation


def make_circle(radius=10, res=30, filled=True):
    g = FilledPolygon([])
    g.radius = radius
    return g


class Viewer:
    def __init__(self, width, height, display=None):
        self.width, self.height = width, height
        self.geoms, self.onetime_geoms = [], []
        self.last_frame = []

    def add_geom(self, geom):
        self.geoms.append(geom)

    def add_onetime(self, geom):
        self.onetime_geoms.append(geom)

    def render(self, return_rgb_array=False):
        self.last_frame = self.onetime_geoms
        self.onetime_geoms = []
        return None

    def close(self):
        pass
