class Geom:
    def __init__(self):
        self.attrs = []
