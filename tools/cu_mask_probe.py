#!/usr/bin/env python3
"""Does a CU mask take effect in this process, and do the appended-workgroup hand-overs survive it?  (run on the GPU box)
    HSA_CU_MASK=0:0-31 python tools/cu_mask_probe.py     /     ROC_GLOBAL_CU_MASK=0xffffffff python tools/cu_mask_probe.py
Prints the time of a 2048-ego dense call with series (epilogue workgroups appended) and of a FISS+ call (search workgroups appended)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fiss_plus_planner_amd import synth  # noqa: E402
from fiss_plus_planner_amd.engine import FrenetEngine  # noqa: E402

B = int(os.environ.get("PROBE_EGOS", "2048"))
b3, b4 = synth.make_config(3, B=B), synth.make_config(4, B=B)
b3.tables_tag, b4.tables_tag = 4401, 4402
with FrenetEngine(0) as eng:
    for name, call in (("dense + series", lambda: eng.plan_dense(b3, tables=False, winner=True)), ("FISS+", lambda: eng.plan_fiss(b4, "FISS+"))):
        call()
        t = []
        for _ in range(5):
            t0 = time.perf_counter()
            out = call()
            t.append(time.perf_counter() - t0)
        print(f"{name}: {min(t) * 1e3:.3f} ms per call (host buffers, min of 5), handover_failed={eng.get_option('handover_failed')}, appended={eng.get_option('appended_workgroups')}",
              "HSA_CU_MASK=" + os.environ.get("HSA_CU_MASK", "-"), "ROC_GLOBAL_CU_MASK=" + os.environ.get("ROC_GLOBAL_CU_MASK", "-"), flush=True)
