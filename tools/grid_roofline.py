"""bench.py's roofline_gridencoder leg alone (stand-alone grid_encode, algorithmic bytes / event-timed duration)"""
import json
import os
import sys

import torch

sys.argv = sys.argv[:1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

r = bench.grid_roofline(torch.device("cuda", 0))
for k, v in r.items():
    print(k, v["ms"], "ms", v["frac"], flush=True)
print(json.dumps(r))
