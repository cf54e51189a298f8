import os, sys, shutil
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from fiss_plus_planner_amd import _abi
_abi.LIB_PATH = os.path.join(root, "tools", "_tmp", "libfrenetgpu_timing.so")
from fiss_plus_planner_amd import synth
from fiss_plus_planner_amd.engine import FrenetEngine
eng = FrenetEngine(0)
b = synth.make_config(3, B=2048)
eng.plan_dense(b, tables=False)
print("---- second launch")
eng.plan_dense(b, tables=False)
