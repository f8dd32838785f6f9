import numpy as np
import pytest


@pytest.fixture(scope="module")
def ctx():
    from inverted_index_2_amd import Context
    c = Context(0)
    yield c
    c.close()


def sorted_unique(rng, n, universe):
    if n == 0:
        return np.empty(0, np.uint32)
    if n > universe // 2:
        v = np.flatnonzero(rng.random(universe) < n / universe)
    else:
        v = np.unique(rng.integers(0, universe, int(n * 1.1) + 8, dtype=np.int64))[:n]
    return v.astype(np.uint32)
