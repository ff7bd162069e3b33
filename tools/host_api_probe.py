#!/usr/bin/env python3
"""bench.py's extras.host_api on its own (th::do_inference on the synthetic 7B: stochastic default sampler, unpipelined, greedy), for profiling runs:
    rocprofv3 --kernel-trace --stats ... -- python tools/host_api_probe.py [n_new]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402

thk = graft.load_package()
with thk.Context(0) as ctx:
    print(json.dumps(bench.extra_host_api(thk, ctx, int(sys.argv[1]) if len(sys.argv) > 1 else 128), indent=1))
