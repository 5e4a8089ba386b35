import numpy as np


def np_random(seed=None):
    rng = np.random.RandomState()
    rng.seed(seed)
    return rng, seed
