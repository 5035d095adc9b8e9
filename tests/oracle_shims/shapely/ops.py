class _Union:
    def __init__(self, bounds):
        self.bounds = bounds


def unary_union(polys):
    bs = [p.bounds for p in polys]
    return _Union((min(b[0] for b in bs), min(b[1] for b in bs), max(b[2] for b in bs), max(b[3] for b in bs)))
