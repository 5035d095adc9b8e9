import numpy as np


def np_random(seed=None):
    if seed is None:
        seed = 0
    return np.random.RandomState(seed % (2 ** 32)), seed
