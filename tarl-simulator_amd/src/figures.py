"""The simulator's analysis figures (reference: src/transportation_simulator.py:387-561, 672-745), drawn from the series
this build already keeps on the device (leg histogram, per-edge delta travel time, pop / withdraw masks, phase timers).

Every figure has a *series* function that does the arithmetic (binning, per-road aggregation, hourly counts) and returns
plain numpy arrays — that is what the tests pin against the reference's rules — and a *figure* function that only draws
them. matplotlib is imported on first use with the non-interactive Agg backend; a host without it gets an ImportError
that names the missing package, the tables (``node_metrics.csv``, ``daily_counts.csv``) do not depend on it.
"""
from __future__ import annotations

import os

import numpy as np
import torch


def _plt():
    try:
        import matplotlib
    except ImportError as exc:  # pragma: no cover - the image has it
        raise ImportError("the figures need matplotlib; the CSV tables are written without it") from exc
    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt
    return plt


def _save(fig, output_dir, filename):
    if output_dir is None:
        return None
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, filename)
    fig.savefig(path)
    return path


# ---- leg histogram (src/transportation_simulator.py:387-451) ----------------------------------------------------------
def leg_histogram_series(values, timestep):
    """values: (T, 4) rows [departures, arrivals, on the way, clock] per step. The reference opens a bin every
    ``18 // timestep`` steps: at step i with i % n == 0 it emits {agents on the way after step i-1, departures and
    arrivals summed since the previous emission, minute of step i-1's clock} — step 0 emits zeros at step 0's own
    minute — and only then adds step i. Steps after the last emission are never shown. Returns minutes, on_way,
    departures, arrivals (1-D arrays of equal length)."""
    v = np.asarray(values, dtype=np.float64).reshape(-1, 4)
    n = int(18 // timestep)
    if n < 1:
        raise ZeroDivisionError("leg histogram: 18 // timestep is 0 (the reference's bin width) for timestep > 18")
    T = v.shape[0]
    starts = np.arange(0, T, n)
    csum = np.concatenate([np.zeros((1, 2)), np.cumsum(v[:, :2], axis=0)])          # csum[i] = sum of steps < i
    prev = np.concatenate([[0], starts[:-1]])
    dep = csum[starts, 0] - csum[prev, 0]
    arr = csum[starts, 1] - csum[prev, 1]
    last = np.maximum(starts - 1, 0)
    on = np.where(starts == 0, 0.0, v[last, 2])
    minutes = (v[last, 3] // 60).astype(np.int64)
    return minutes, on, dep, arr


def leg_histogram_figure(values, timestep, output_dir="data/outputs"):
    if len(values) == 0:
        print("No data available for plotting.")
        return None
    minutes, on, dep, arr = leg_histogram_series(values, timestep)
    plt = _plt()
    fig, ax = plt.subplots(figsize=(12, 6))
    ax.step(minutes, on, label="On Way", color="green")
    ax.step(minutes, dep, label="Departure", color="red", linestyle="--", where="post")
    ax.step(minutes, arr, label="Arrival", color="blue", linestyle="-.", where="post")
    h0, h1 = int(minutes.min()) // 60, int(minutes.max()) // 60
    ax.set_xticks([60 * h for h in range(h0, h1 + 1)])
    ax.set_xticklabels([str(h) for h in range(h0, h1 + 1)])
    ax.set_xlabel("Hour of Day")
    ax.set_ylabel("Number of Agents")
    ax.set_title("Leg Histogram Over Time")
    ax.legend(loc="upper left")
    fig.tight_layout()
    if _save(fig, output_dir, "leg_histogram.png"):
        print("Leg histogram saved as ", "leg_histogram.png")
    return fig


# ---- road optimality (src/transportation_simulator.py:453-517) ------------------------------------------------------------
def road_optimality_series(records, edge_src, num_roads):
    """records: [(clock, delta_travel_time [E])] per step. Sum of every road's outgoing route edges per step, on the
    records' own device (one index_add over a (T, E) stack), returned as hours [T] and a (T, num_roads) array."""
    hours = np.asarray([float(t) for t, _ in records], dtype=np.float32) / np.float32(3600.0)
    per_edge = torch.stack([torch.as_tensor(v).reshape(-1) for _, v in records])            # (T, E)
    src = torch.as_tensor(edge_src, device=per_edge.device).reshape(-1).to(torch.long)
    per_road = torch.zeros((per_edge.size(0), int(num_roads)), dtype=per_edge.dtype, device=per_edge.device)
    per_road.index_add_(1, src, per_edge)
    return hours, per_road.cpu().numpy()


def road_optimality_figure(records, edge_src, num_roads, road_ids=(), output_dir="data/outputs", max_lines=64):
    if len(records) == 0:
        print("No road optimality data available for plotting.")
        return None
    hours, per_road = road_optimality_series(records, edge_src, num_roads)
    plt = _plt()
    fig, ax = plt.subplots(figsize=(12, 6))
    shown = list(road_ids) if len(road_ids) else list(range(per_road.shape[1]))
    for r in shown:
        ax.plot(hours, per_road[:, r], label=f"Node {r}")
    ax.set_xlabel("Time (h)")
    ax.set_ylabel("Delta Travel Time (s) — sum over outgoing edges")
    ax.set_title("Road Optimality (Aggregated by Source Node) Over Time")
    if len(shown) <= max_lines:          # a legend with one entry per road of a city network is larger than the figure
        ax.legend()
    fig.tight_layout()
    if _save(fig, output_dir, "road_optimality.png"):
        print("Road optimality plot saved as", "road_optimality.png")
    return fig


# ---- phase timers (src/transportation_simulator.py:519-561) -----------------------------------------------------------
def computation_time_series(inserting, choice, core, withdraw):
    """Wedge sizes in the reference's order; NaN timers become -1 there."""
    return ["Inserting", "Choice", "Core", "Withdraw"], [(-1 if np.isnan(t) else float(t))
                                                         for t in (inserting, choice, core, withdraw)]


def computation_time_figure(inserting, choice, core, withdraw, output_dir="data/outputs"):
    labels, sizes = computation_time_series(inserting, choice, core, withdraw)
    total = sum(sizes)
    if total == 0:
        print("No computation time data available for plotting.")
        return None
    plt = _plt()
    fig = plt.figure(figsize=(8, 8))
    ax = fig.add_subplot(111)

    def wedge_text(pct):
        return "{:.1f}%\n{:.2f} s".format(pct, pct * total / 100.0)
    _, _, autotexts = ax.pie(sizes, labels=labels, autopct=wedge_text, startangle=90,
                             textprops={"color": "black", "fontsize": 12})
    for txt in autotexts:
        txt.set_fontweight("bold")
    ax.set_title("Computation Time Distribution\nTotal Execution Time: {:.2f} s".format(total), fontsize=14)
    ax.axis("equal")
    if _save(fig, output_dir, "computation_time.png"):
        print("Computation time plot saved as", "computation_time.png")
    return fig


# ---- daily link counts (src/transportation_simulator.py:672-745) ----------------------------------------------------------
def hourly_counts(history):
    """history: [(clock, bool mask [R])] (Response pops followed by withdrawals). Departures per road and hour as an
    (R, H) int64 tensor on the masks' device: one (H, T) x (T, R) product (counts < 2^24: exact in fp32)."""
    dev = history[0][1].device
    clocks = torch.tensor([int(t) for t, _ in history], dtype=torch.long, device=dev)
    masks = torch.stack([m.reshape(-1) for _, m in history]).to(torch.float32)
    hours = (clocks // 3600).clamp(min=0)
    onehot = torch.nn.functional.one_hot(hours, num_classes=int(hours.max().item()) + 1).to(torch.float32)
    return (onehot.t() @ masks).t().contiguous().to(torch.long)


def daily_counts_series(history, expected_counts):
    """Simulated departures per road over the whole run against the expected flows of the MSA assignment: the roads the
    dictionary names (sorted), their simulated totals and their expected values; keys outside the graph count as 0
    expected and are an IndexError on the simulated side, as in the reference."""
    totals = hourly_counts(history).sum(dim=1).cpu()
    R = totals.numel()
    expected = torch.zeros(R, dtype=torch.float64)
    for road, flow in expected_counts.items():
        if 0 <= road < R:
            expected[road] = float(flow)
    roads = sorted(expected_counts.keys())
    return roads, totals[roads].numpy(), expected[roads].numpy()


def daily_counts_figure(history, expected_counts, output_dir="data/outputs"):
    if len(history) == 0:
        print("No update history available for computing node metrics.")
        return {}
    roads, simulated, expected = daily_counts_series(history, expected_counts)
    plt = _plt()
    fig, ax = plt.subplots()
    ax.scatter(expected, simulated, alpha=0.7)
    top = float(max(expected.max() if expected.size else 0.0, simulated.max() if simulated.size else 0.0))
    ax.plot([0, top], [0, top], "r--", linewidth=1)
    ax.set_xlabel("Expected daily count")
    ax.set_ylabel("Simulated daily count")
    ax.set_title("Daily Link Counts: Expected vs Simulated")
    fig.tight_layout()
    path = _save(fig, output_dir, "daily_counts.png")
    if path:
        import csv
        csv_path = os.path.join(output_dir, "daily_counts.csv")
        with open(csv_path, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["link_id", "simulated", "expected", "difference"])
            for r, s, e in zip(roads, simulated.tolist(), expected.tolist()):
                w.writerow([r, s, e, s - e])
        print(f"Daily counts plot saved as {path}")
        print(f"Daily counts CSV saved as {csv_path}")
    return fig
