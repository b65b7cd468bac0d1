"""PBRT_MI355_REORDER (the ray-reordering experiment of DESIGN.md 5.5, off by default): the closest-hit walk of every bounce
after the first goes through a sorted permutation of the queue.  Hits do not depend on the order rays are walked in, so the
film must be bit-identical with the sort on, whatever the key width."""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _film(pm):
    integ = pm.WavefrontIntegrator(5, True, True, pm.scenes.blob_sky_scene(64, (256, 192)), 4)
    integ.render()
    return integ.film().copy()


@pytest.mark.parametrize("env", [{"PBRT_MI355_REORDER": "5"}, {"PBRT_MI355_REORDER": "9", "PBRT_MI355_REORDER_OCT_HIGH": "1"}])
def test_film_bit_identical_with_sorted_queues(monkeypatch, env):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    pm = importlib.import_module("pbrt-rs_amd")
    monkeypatch.delenv("PBRT_MI355_REORDER", raising=False)
    monkeypatch.delenv("PBRT_MI355_REORDER_OCT_HIGH", raising=False)
    plain = _film(pm)
    assert plain[..., 3].min() > 0 and plain[..., :3].max() > 0
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    assert np.array_equal(_film(pm).view(np.uint64), plain.view(np.uint64))
