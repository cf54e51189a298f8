#!/usr/bin/env python3
"""One line per run of bench.py's main workload (no extras): python tools/bench_short.py [bench args]"""
import json, subprocess, sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--steps", os.environ.get("STEPS", "120"), "--warmup", "8", "--cpu-seconds", "0", "--no-latency", "--no-extras"] + sys.argv[1:],
                     capture_output=True, text=True)
try:
    d = json.loads(out.stdout.strip().splitlines()[-1]); k = d["kernel_ms_stats"]
    print(f"kernel mean {k['mean']*1e3:7.1f} us  median {k['median']*1e3:7.1f}  min {k['min']*1e3:7.1f}   step {d['ms_per_step']*1e3:7.1f} us  value {d['value']:.4g}")
except Exception as e:
    print("FAILED", e, out.stderr[-500:])
