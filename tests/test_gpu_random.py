"""GPU: the randomized GPU-vs-oracle runs of scripts/stress*.py (the generators that found the round-1 lane bug) as part
of the suite, seeded and time-boxed: codec round trips, intersections and unions over mixed densities with every
path-selecting option, merges with skewed / clustered / duplicated terms and tombstones, two-stage merges through a
device segment, the dense streaming kernel with random blocks per wave."""
import importlib.util
import os

import pytest

from tests.gpu_util import ctx  # noqa: F401

pytestmark = pytest.mark.gpu

_SCRIPTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts")


def _load(name):
    spec = importlib.util.spec_from_file_location("ii2_scripts_" + name, os.path.join(_SCRIPTS, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", [101, 102])
def test_random_intersect_union_codec(ctx, seed):
    assert _load("stress").main(budget=8.0, seed=seed, ctx=ctx, quiet=True) >= 5


@pytest.mark.parametrize("seed", [201, 202])
def test_random_merges(ctx, seed):
    assert _load("stress_merge").main(budget=8.0, seed=seed, ctx=ctx, quiet=True) >= 5


def test_random_dense_streaming(ctx):
    assert _load("stress_dense").main(iters=12, seed0=301, ctx=ctx, quiet=True) == 12
