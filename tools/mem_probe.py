#!/usr/bin/env python3
"""Developer tool: bench.py's own run (same flags) followed by torch's peak device-memory figures."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tarl-simulator_amd")]
import torch  # noqa: E402

import bench  # noqa: E402

bench.main()
print("max_memory_allocated GB %.1f   max_memory_reserved GB %.1f" % (torch.cuda.max_memory_allocated() / 1e9,
                                                                     torch.cuda.max_memory_reserved() / 1e9))
