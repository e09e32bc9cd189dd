"""Column maps = the memory-layout contract of the path (reference: src/feature_helpers.py:38-54,59-71,77-92)."""


class FeatureHelpers:
    """``graph.x`` row layout for FIFO depth ``Nmax``: three FIFO blocks (agent id, arrival time, departure time) of
    ``Nmax`` slots, then seven scalars; ``F = 3 * Nmax + 7`` columns. (``NODE_TYPE`` is declared one past the last
    column by the reference and never read.)"""

    def __init__(self, Nmax=100):
        self.Nmax = Nmax
        for k, name in enumerate(("AGENT_POSITION", "AGENT_TIME_ARRIVAL", "AGENT_TIME_DEPARTURE")):
            setattr(self, name, slice(k * Nmax, (k + 1) * Nmax))
        scalars = ("MAX_NUMBER_OF_AGENT", "NUMBER_OF_AGENT", "FREE_FLOW_TIME_TRAVEL", "LENGHT_OF_ROAD", "MAX_FLOW",
                   "SELECTED_ROAD", "ROAD_INDEX", "NODE_TYPE")
        for k, name in enumerate(scalars):
            setattr(self, name, 3 * Nmax + k)
        self.HEAD_FIFO, self.HEAD_FIFO_ARRIVAL_TIME, self.HEAD_FIFO_DEPARTURE_TIME = 0, Nmax, 2 * Nmax
        self.CONGESTION_FILE = 3


class AgentFeatureHelpers:
    """``agent_features`` row layout, 9 fp32 columns."""
    _COLUMNS = ("ORIGIN", "DESTINATION", "DEPARTURE_TIME", "ARRIVAL_TIME", "AGE", "SEX", "EMPLOYMENT_STATUS", "ON_WAY",
                "DONE")

    def __init__(self):
        for k, name in enumerate(self._COLUMNS):
            setattr(self, name, k)

    def __len__(self):
        return len(self._COLUMNS)


class ObservationFeatureHelpers:
    """Observation row = the 7 road scalars followed by the 9 agent columns (16 values)."""

    def __init__(self):
        road = ("MAX_NUMBER_OF_AGENT", "NUMBER_OF_AGENT", "FREE_FLOW_TIME_TRAVEL", "LENGHT_OF_ROAD", "MAX_FLOW",
                "SELECTED_ROAD", "ROAD_INDEX")
        for k, name in enumerate(road + AgentFeatureHelpers._COLUMNS):
            setattr(self, name, k)
