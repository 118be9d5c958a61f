"""N2: a policy TRAINED IN THE REFERENCE (policy_folder/policy177, 443 M samples on MJX) must walk in
this simulator.  It is the only end-to-end behavioural signal available without MJX: it exercises the observation
layout, the actuator / joint orderings, the contact model, the scan orientation and the phase clock at once."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_policy177_walks_on_flat_ground():
    from rollout_policy import rollout
    st = rollout("policy177", "flat", n=512, steps=400, cmd=(0.5, 0.0, 0.0))
    print(st)
    assert st["survival"] == 1.0
    assert st["track_lin"] > 0.85 and st["vx"] > 0.38           # reward term exp(-|v_cmd - v|^2 / 0.2), forward speed
    assert 0.35 < st["contact_duty"] < 0.55                      # policy's own statistics: <last_contact> = 0.44
    assert 9.6 < st["accel_z"] < 10.2                            # policy's own statistics: <accel_z> = 9.72
    st = rollout("policy177", "flat", n=512, steps=400, cmd=(0.0, 0.0, 0.8))
    assert st["survival"] == 1.0 and st["track_ang"] > 0.85 and abs(st["vx"]) < 0.05


def test_policy177_traverses_stairs():
    from rollout_policy import rollout
    st = rollout("policy177", "level4", n=512, steps=300, cmd=(0.5, 0.0, 0.0))
    print(st)
    assert st["survival"] > 0.6 and st["vx"] > 0.1
    assert 0.005 < st["obs_mean_scan"] < 0.08                    # normaliser mean of the scan block: 0.035
