"""CPU: the series behind the analysis figures (src/figures.py) against the golden read back from the matplotlib artists
of the reference's own plot_* methods (oracle/make_golden.py:gen_figures), and the files the figure functions write."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, PKG

sys.path.insert(0, PKG)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "figures.npz"))


@pytest.mark.parametrize("dt", [1, 2, 6])
def test_leg_histogram_bins(gold, dt):
    from src import figures
    minutes, on, dep, arr = figures.leg_histogram_series(gold[f"leg{dt}__values"], dt)
    assert np.array_equal(minutes, gold[f"leg{dt}__minutes"])
    assert np.array_equal(on, gold[f"leg{dt}__on"])
    assert np.array_equal(dep, gold[f"leg{dt}__dep"])
    assert np.array_equal(arr, gold[f"leg{dt}__arr"])


def test_leg_histogram_bin_width_zero_is_an_error():
    from src import figures
    with pytest.raises(ZeroDivisionError):           # the reference's `i % (18 // timestep)` with timestep 20
        figures.leg_histogram_series(np.zeros((4, 4)), 20)


def test_road_optimality_per_road(gold):
    from src import figures
    recs = [(int(t), torch.from_numpy(v)) for t, v in zip(gold["opt__clocks"], gold["opt__dtt"])]
    hours, per_road = figures.road_optimality_series(recs, gold["opt__src"], 5)
    assert np.array_equal(hours, gold["opt__hours"])
    np.testing.assert_allclose(per_road, gold["opt__per_road"], rtol=1e-6, atol=1e-5)    # scatter-add order


def test_daily_counts(gold):
    from src import figures
    hist = [(int(t), torch.from_numpy(m)) for t, m in zip(gold["daily__clocks"], gold["daily__masks"])]
    expected = {int(k): float(v) for k, v in zip(gold["daily__expected_keys"], gold["daily__expected_vals"])}
    roads, sim, exp = figures.daily_counts_series(hist, expected)
    assert roads == sorted(expected)
    assert np.array_equal(exp, gold["daily__x"]) and np.array_equal(sim, gold["daily__y"])
    counts = figures.hourly_counts(hist)
    assert counts.shape == (6, 10) and int(counts.sum()) == int(gold["daily__masks"].sum())
    assert int(counts[:, :7].sum()) == 0             # the history starts at 07:00


def test_figure_files(tmp_path, gold):
    import matplotlib
    matplotlib.use("Agg")
    from src import figures
    out = str(tmp_path)
    assert figures.leg_histogram_figure(gold["leg2__values"], 2, out) is not None
    recs = [(int(t), torch.from_numpy(v)) for t, v in zip(gold["opt__clocks"], gold["opt__dtt"])]
    assert figures.road_optimality_figure(recs, gold["opt__src"], 5, [1, 3], out) is not None
    assert figures.computation_time_figure(0.5, 1.5, 2.0, 0.25, out) is not None
    with pytest.raises(ValueError):                  # a NaN timer becomes a -1 wedge, which matplotlib refuses (as there)
        figures.computation_time_figure(0.5, 1.5, 2.0, float("nan"), out)
    assert figures.computation_time_figure(0, 0, 0, 0, out) is None
    hist = [(int(t), torch.from_numpy(m)) for t, m in zip(gold["daily__clocks"], gold["daily__masks"])]
    assert figures.daily_counts_figure(hist, {0: 2.0, 2: 7.25}, out) is not None
    for name in ("leg_histogram.png", "road_optimality.png", "computation_time.png", "daily_counts.png",
                 "daily_counts.csv"):
        assert os.path.getsize(os.path.join(out, name)) > 0, name
    rows = open(os.path.join(out, "daily_counts.csv")).read().strip().splitlines()
    assert rows[0] == "link_id,simulated,expected,difference" and len(rows) == 3
    assert figures.leg_histogram_figure([], 1, out) is None and figures.road_optimality_figure([], [], 5) is None
    import matplotlib.pyplot as plt
    plt.close("all")
