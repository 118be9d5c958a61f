import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


class GoldenCases:
    """the reference-generated Joystick.step records of tools/gen_golden.py: task_step*.npz keep one array per (case, key) as `c{i}_{key}`, the roll-out files
    (task_step_rollout*.npz) one stacked array per key; `cases(i, key)` reads either"""

    def __init__(self, path):
        import numpy as np
        self.g = np.load(path, allow_pickle=False)
        self.stacked = "layout" in self.g.files and str(self.g["layout"]) == "stacked"
        self.ncases = int(self.g["ncases"])

    def __call__(self, i, key):
        return self.g[key][i] if self.stacked else self.g[f"c{i}_{key}"]

    def __getitem__(self, name):          # the per-case spelling `g["c3_obs"]` keeps working for both layouts
        if self.stacked and name.startswith("c") and "_" in name and name[1:name.index("_")].isdigit():
            return self.g[name[name.index("_") + 1:]][int(name[1:name.index("_")])]
        return self.g[name]
