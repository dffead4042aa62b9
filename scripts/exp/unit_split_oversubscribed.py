import sys, os
sys.path.insert(0, os.getcwd()); sys.argv = ["x"]
import importlib.util
spec = importlib.util.spec_from_file_location("pus", os.path.join(os.getcwd(), "scripts", "probe_unit_split.py")); P = importlib.util.module_from_spec(spec); spec.loader.exec_module(P)
import torch
for B in (70, 100, 160):
    ok = P.check_case(64, B, 64, reps=3, bwd=True)
    print("B", B, "ok" if ok else "FAIL")
