#!/usr/bin/env python3
"""Where the HOST spends a step's issue time (cProfile over a few steps of bench.py's workload)."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-roofline", "--steps", "10", "--warmup", "3"]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue())
