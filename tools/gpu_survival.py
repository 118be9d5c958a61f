"""survival / speed / duty / tilt of policy177 per level (deterministic actions, forward command)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_scan_orientation import run
for level in sys.argv[1:] or ["level4", "level10", "level13"]:
    print(level, run(level, "as_built", n=2048, steps=500))
