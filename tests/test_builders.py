"""MATSim network / population builders of the mirror (src/matsim_io.py) against goldens generated from the reference's
``TransportationSimulator.config_network`` (src/transportation_simulator.py:61-228) and
``Agents.config_agents_from_xml`` (src/agents/base.py:36-242) — oracle/make_golden.py:gen_builders. The two tiny XML
documents are the fixtures of the reference's own tests (tests/conftest.py:94-106,
tests/config_agents_from_xml_test.py:38-95); the torus documents come from tarl_hip.synth's writers.
Host-side code: runs without a GPU."""
import gzip
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
from make_golden_fixtures import EQUIL_NETWORK_XML, EQUIL_POPULATION_XML, SIMPLE_NETWORK_XML  # noqa: E402

from src.agents.base import Agents  # noqa: E402
from src.transportation_simulator import TransportationSimulator  # noqa: E402
from tarl_hip import synth  # noqa: E402

GRAPH_KEYS = ("x", "edge_index", "edge_attr", "edge_index_routes", "edge_attr_routes", "adj_matrix", "src_adj",
              "critical_number", "congestion_constant")


@pytest.fixture(scope="module")
def scenarios(tmp_path_factory):
    root = tmp_path_factory.mktemp("data")
    for tag, xml in (("simple", SIMPLE_NETWORK_XML), ("equil", EQUIL_NETWORK_XML)):
        os.makedirs(root / tag)
        (root / tag / "network.xml").write_text(xml)
    (root / "equil" / "population.xml").write_text(EQUIL_POPULATION_XML)
    os.makedirs(root / "torus")
    synth.write_matsim_network_xml(str(root / "torus" / "network.xml"), 3, 4, seed=5)
    synth.write_matsim_population_xml(str(root / "torus" / "population.xml"), 3, 4, 60, seed=6)
    return root


@pytest.mark.parametrize("tag", ["simple", "equil", "torus"])
def test_config_network_matches_reference(scenarios, tag):
    g = load_golden("builders")
    sim = TransportationSimulator("cpu")
    sim.config_network(str(scenarios / tag / "network"))
    assert sim.Nmax == int(g[f"{tag}__Nmax"])
    assert int(sim.graph.num_roads) == int(g[f"{tag}__num_roads"])
    for k in GRAPH_KEYS:
        got = getattr(sim.graph, k).numpy()
        want = np.asarray(g[f"{tag}__{k}"])
        assert got.dtype == want.dtype and got.shape == want.shape, k
        assert np.array_equal(got, want), k


def test_simple_network_facts(scenarios):
    """tests/transportation_simulator_test.py:8-14 of the reference: 2 links + 2 intersections -> 6 nodes, 6 edges,
    2 route edges."""
    sim = TransportationSimulator("cpu")
    sim.config_network(str(scenarios / "simple" / "network"))
    assert sim.graph.x.shape[0] == 6
    assert sim.graph.edge_index.shape[1] == 6
    assert sim.graph.edge_index_routes.shape[1] == 2


@pytest.mark.parametrize("tag", ["equil", "torus"])
def test_config_agents_from_xml_matches_reference(scenarios, tag):
    g = load_golden("builders")
    ag = Agents("cpu")
    ag.config_agents_from_xml(str(scenarios / tag), verbose=False)
    want = np.asarray(g[f"{tag}__agents"])
    assert tuple(ag.agent_features.shape) == want.shape
    assert np.array_equal(ag.agent_features.numpy(), want)


def test_equil_population_facts(scenarios):
    """The assertions of the reference's tests/config_agents_from_xml_test.py:99-124."""
    ag = Agents("cpu")
    ag.config_agents_from_xml(str(scenarios / "equil"), verbose=False)
    a = ag.agent_features
    assert a.shape[0] > 1
    trips = [tuple(int(v) for v in r) for r in a[1:, [ag.ORIGIN, ag.DESTINATION, ag.DEPARTURE_TIME]].tolist()]
    assert trips == [(3, 8, 21600), (7, 4, 25200), (3, 8, 28800), (3, 8, 21600), (7, 4, 23400)]
    assert (a[1:, ag.AGE] == 20).all() and (a[1:, ag.SEX] == 0).all() and (a[1:, ag.EMPLOYMENT_STATUS] == 0).all()


def test_gz_and_missing_cellsize(tmp_path):
    """``.xml.gz`` wins over ``.xml``; a missing ``effectivecellsize`` falls back to 7.5 (reference :75-83, :98-101)."""
    synth.write_matsim_network_xml(str(tmp_path / "a.xml"), 2, 2, seed=1, effectivecellsize=None)
    synth.write_matsim_network_xml(str(tmp_path / "b.xml"), 2, 2, seed=1, effectivecellsize=7.5)
    with open(tmp_path / "b.xml", "rb") as f, gzip.open(tmp_path / "c.xml.gz", "wb") as z:
        z.write(f.read())
    synth.write_matsim_network_xml(str(tmp_path / "c.xml"), 2, 3, seed=2)     # must be ignored
    sims = {}
    for n in "abc":
        sims[n] = TransportationSimulator("cpu")
        sims[n].config_network(str(tmp_path / n))
    for k in GRAPH_KEYS:
        assert torch.equal(getattr(sims["a"].graph, k), getattr(sims["b"].graph, k))
        assert torch.equal(getattr(sims["c"].graph, k), getattr(sims["b"].graph, k))
    with pytest.raises(FileNotFoundError):
        TransportationSimulator("cpu").config_network(str(tmp_path / "nope"))


def test_load_network_and_population_build_caches(scenarios, tmp_path, monkeypatch):
    """``load_network`` / ``Agents.load`` fall back from save/<scenario>/*.pt to data/<scenario>/*.xml and write the
    cache (reference: src/transportation_simulator.py:248-261, src/agents/base.py:420-444)."""
    monkeypatch.chdir(tmp_path)
    os.makedirs("data")
    os.symlink(str(scenarios / "torus"), os.path.join("data", "torus"))
    sim = TransportationSimulator("cpu")
    sim.load_network("torus")
    sim.agent.load("torus")
    assert os.path.exists("save/torus/network.pt") and os.path.exists("save/torus/population.pt")
    assert sim.agent.agent_features[0, sim.agent.DEPARTURE_TIME] == 48 * 3600
    again = TransportationSimulator("cpu")
    again.load_network("torus")
    again.agent.load("torus")
    assert again.Nmax == sim.Nmax and torch.equal(again.graph.x, sim.graph.x)
    assert torch.equal(again.graph.edge_index, sim.graph.edge_index)
    assert torch.equal(again.agent.agent_features, sim.agent.agent_features)
