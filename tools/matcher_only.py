"""Run only the 2048 x 2048 matcher measurement (for rocprofv3 counter passes)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
b = importlib.util.module_from_spec(spec); sys.argv = ["bench.py"]; spec.loader.exec_module(b)
import torch
import uwimageproc_amd as uw
ctx = uw.Context(0)
print(json.dumps(b.matcher_report(ctx, torch.device("cuda", 0))))
