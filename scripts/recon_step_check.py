"""Print the measured agreement of the composed reconstruction step with its golden (tests/test_recon_step.py::step_metrics)."""
import os, pathlib, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_recon_step as t
with tempfile.TemporaryDirectory() as tmp:
    for k, v in t.step_metrics(pathlib.Path(tmp)).items():
        print(f"{k:40s} {v}")
