"""MATSim XML ingestion for the MPNN+PPO path's inputs: the road-graph builder behind
``TransportationSimulator.config_network`` (reference: src/transportation_simulator.py:61-228) and the population
parser behind ``Agents.config_agents_from_xml`` (reference: src/agents/base.py:36-242).

Host-side, one-off preprocessing — it produces the tensors the HIP path consumes, so what matters is that they are
bit-identical to the reference's (tests/test_builders.py checks them against goldens generated from the reference).
Written over ``xml.etree`` + ``gzip`` with column-wise tensor arithmetic; no lxml, scikit-learn or tqdm needed.

Graph layout produced (the contract the kernels rely on):
  nodes   0 .. R-1            one per <link>, in file order                 ROAD_INDEX = link position
          R + 2k, R + 2k + 1  SRC / DEST pseudo-node of the k-th intersection (ids sorted as strings), ROAD_INDEX = -1
  edges   road -> road        for every link j and every link leaving j's head node, attr = cap_j / sum(cap_j ...)
          SRC(k) -> road      roads leaving intersection k, attr 0
          road -> DEST(k)     roads entering intersection k, attr 0
"""
from __future__ import annotations

import gzip
import os
import xml.etree.ElementTree as ET
from datetime import datetime

import numpy as np
import torch

from ._compat import Data
from .feature_helpers import FeatureHelpers

# dense N x N adjacency / SRC x R matrices are only materialised below this many nodes (the HIP path never reads them;
# the reference builds them unconditionally, which is 10 GB of bools for a 100k-node graph)
DENSE_ADJ_MAX_NODES = 16384


def resolve_xml(path_base: str) -> str:
    """``<base>.xml.gz`` if it exists, else ``<base>.xml`` (src/transportation_simulator.py:75-83)."""
    for ext in (".xml.gz", ".xml"):
        if os.path.exists(path_base + ext):
            return path_base + ext
    raise FileNotFoundError(f"Neither {path_base}.xml.gz nor {path_base}.xml exists.")


def parse_xml(path: str):
    if path.endswith(".gz"):
        with gzip.open(path, "rb") as f:
            return ET.parse(f).getroot()
    return ET.parse(path).getroot()


def _links_of(root):
    links = root.find("links")
    if links is None:
        raise ValueError("The XML file does not contain a 'links' element.")
    return links, [l for l in links if isinstance(l.tag, str)]


# ---- network ----------------------------------------------------------------------------------------------------------------
def build_network(path_base: str):
    """Returns ``(Data graph, Nmax)`` for ``<path_base>.xml[.gz]``; every tensor equals the reference's bit for bit
    (fp32 arithmetic in the same order: FIFO capacity = trunc(length * lanes / cell) + 1, free-flow time =
    length / freespeed, route-edge attribute accumulated in double then rounded once)."""
    root = parse_xml(resolve_xml(path_base))
    links_el, links = _links_of(root)
    try:
        cell = float(links_el.get("effectivecellsize"))
    except (TypeError, ValueError):
        cell = 7.5
    R = len(links)
    frm = [l.attrib["from"] for l in links]
    to = [l.attrib["to"] for l in links]
    f32 = lambda key: torch.tensor([float(l.attrib[key]) for l in links], dtype=torch.float32)
    length, cap32, speed, lanes = f32("length"), f32("capacity"), f32("freespeed"), f32("permlanes")
    cap = [float(l.attrib["capacity"]) for l in links]

    maxn = torch.trunc(length * lanes / cell) + 1.0
    Nmax = int(maxn.max().item() + 1)
    h = FeatureHelpers(Nmax=Nmax)

    inters = sorted(set(frm) | set(to))
    rank = {name: k for k, name in enumerate(inters)}
    I = len(inters)
    N = R + 2 * I
    x = torch.zeros((N, 3 * Nmax + 7), dtype=torch.float32)
    x[:R, h.MAX_NUMBER_OF_AGENT] = maxn
    x[:R, h.FREE_FLOW_TIME_TRAVEL] = length / speed
    x[:R, h.LENGHT_OF_ROAD] = length
    x[:R, h.MAX_FLOW] = cap32
    x[:R, h.ROAD_INDEX] = torch.arange(R, dtype=torch.float32)
    x[R:, h.ROAD_INDEX] = -1.0

    outgoing = [[] for _ in range(I)]
    incoming = [[] for _ in range(I)]
    for j in range(R):
        outgoing[rank[frm[j]]].append(j)
        incoming[rank[to[j]]].append(j)

    r_from, r_to, r_attr = [], [], []
    for j in range(R):
        down = outgoing[rank[to[j]]]
        total = 0.0
        for _ in down:
            total += cap[j]                 # the reference sums the UPSTREAM capacity once per downstream link
        share = cap[j] / (total if total > 0 else 1.0)
        r_from.extend([j] * len(down))
        r_to.extend(down)
        r_attr.extend([share] * len(down))
    e_from, e_to = list(r_from), list(r_to)
    for k in range(I):
        e_from.extend([R + 2 * k] * len(outgoing[k]))
        e_to.extend(outgoing[k])
    for k in range(I):
        e_from.extend(incoming[k])
        e_to.extend([R + 2 * k + 1] * len(incoming[k]))
    edge_index_routes = torch.tensor([r_from, r_to], dtype=torch.long).reshape(2, -1)
    edge_attr_routes = torch.tensor(r_attr, dtype=torch.float32).view(-1, 1)
    edge_index = torch.tensor([e_from, e_to], dtype=torch.long).reshape(2, -1)
    edge_attr = torch.cat([edge_attr_routes.view(-1), torch.zeros(len(e_from) - len(r_from))]).view(-1, 1)

    critical_number = x[:, h.MAX_FLOW] * x[:, h.FREE_FLOW_TIME_TRAVEL] / 3600
    congestion_constant = x[:, h.FREE_FLOW_TIME_TRAVEL] * (x[:, h.MAX_NUMBER_OF_AGENT] + 10 - critical_number)
    fields = dict(x=x, edge_index=edge_index, edge_attr=edge_attr, edge_index_routes=edge_index_routes,
                  edge_attr_routes=edge_attr_routes, num_roads=R, critical_number=critical_number,
                  congestion_constant=congestion_constant)
    if N <= DENSE_ADJ_MAX_NODES:
        adj = torch.zeros((N, N), dtype=torch.bool)
        adj[edge_index[0], edge_index[1]] = True
        src_adj = adj[R::2, :R].to(torch.float32)
        deg = src_adj.sum(dim=1, keepdim=True)
        fields.update(adj_matrix=adj, src_adj=torch.where(deg > 0, src_adj / deg, torch.zeros_like(src_adj)))
    return Data(**fields), Nmax


# ---- population -------------------------------------------------------------------------------------------------------------
def _end_time_seconds(act) -> int:
    s = act.get("end_time")
    if not s:
        return 0
    for fmt in ("%H:%M:%S", "%H:%M"):
        try:
            t = datetime.strptime(s, fmt)
        except ValueError:
            continue
        return t.hour * 3600 + t.minute * 60 + t.second
    return 0


def _person_attributes(person) -> dict:
    attrs = dict(person.attrib)
    nested = person.find("attributes")
    if nested is not None:
        for a in nested.findall("attribute"):
            if a.get("name") and a.text:
                attrs[a.get("name")] = a.text
    attrs.setdefault("car_avail", attrs.get("carAvail", "always"))
    attrs.setdefault("sex", "m")
    attrs.setdefault("employed", "no")
    attrs.setdefault("age", "20")
    return attrs


def build_population(population_base: str, network_base: str, *, log=None):
    """One row per TRIP (consecutive activity pair) of every car-owning person, row 0 the dummy agent. An activity's
    ``link`` attribute is looked up among the INTERSECTION ids (the reference's convention); when it is not one, the
    nearest intersection to the activity's x/y is used. Returns ``(rows float32 (A, 9), stats dict)``."""
    population = parse_xml(resolve_xml(population_base))
    network = parse_xml(resolve_xml(network_base))
    nodes = network.find("nodes")
    if nodes is None:
        raise ValueError("The XML file does not contain a 'nodes' element.")
    _, links = _links_of(network)
    pos = {n.get("id"): (float(n.get("x")), float(n.get("y"))) for n in nodes if isinstance(n.tag, str)}
    R = len(links)
    inters = sorted({l.get("from") for l in links} | {l.get("to") for l in links})
    rank = {name: k for k, name in enumerate(inters)}
    coords = np.array([pos[name] for name in inters], dtype=np.float64).reshape(-1, 2)

    def snap(act):
        """intersection rank of an activity, or None"""
        name = act.get("link")
        if name in rank:
            return rank[name]
        ax, ay = act.get("x"), act.get("y")
        if ax is None or ay is None or not len(coords):
            return None
        try:
            d = coords - np.array([float(ax), float(ay)])
        except ValueError:
            return None
        return int(np.argmin(np.einsum("ij,ij->i", d, d)))

    rows = [[0.0, 0.0, 25 * 3600, 0.0, 20.0, 0.0, 0.0, 0.0, 0.0]]
    stats = dict(total=0, selected=0, car_avail_not_always=0, no_plan=0, too_few_activities=0, no_valid_trip=0, trips=[])
    for person in population:
        if not isinstance(person.tag, str):
            continue
        stats["total"] += 1
        attrs = _person_attributes(person)
        if attrs.get("car_avail", attrs.get("carAvail", "")).lower() != "always":
            stats["car_avail_not_always"] += 1
            continue
        plan = person.find("plan")
        if plan is None:
            stats["no_plan"] += 1
            continue
        acts = plan.findall("act") or plan.findall("activity")
        if len(acts) < 2:
            stats["too_few_activities"] += 1
            continue
        sex = 1.0 if attrs.get("sex", "m").lower() == "f" else 0.0
        employed = 1.0 if attrs.get("employed", "no").lower() == "yes" else 0.0
        age = float(attrs.get("age", 0))
        snapped = [snap(a) for a in acts]
        trips = 0
        for i in range(len(acts) - 1):
            o, d = snapped[i], snapped[i + 1]
            if o is None or d is None:
                if log:
                    log(f"Could not create plan for person {person.get('id')}: Invalid trip : "
                        f"{acts[i].get('link')} -> {acts[i + 1].get('link')}")
                continue
            rows.append([float(R + 2 * o), float(R + 2 * d + 1), float(_end_time_seconds(acts[i])), 0.0, age, sex,
                         employed, 0.0, 0.0])
            trips += 1
        if trips:
            stats["selected"] += 1
            stats["trips"].append(trips)
        else:
            stats["no_valid_trip"] += 1
    return torch.tensor(rows, dtype=torch.float32), stats
