GL_LINE_SMOOTH = GL_POLYGON_SMOOTH = 0


def glEnable(flag):
    pass
