"""Stand-in for torch-geometric==2.5.0 (see oracle/refshim/README.md). Test infrastructure only."""
__version__ = "2.5.0-standin"
