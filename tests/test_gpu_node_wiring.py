"""gp_slip_node.gaussian_process() -- the rospy entry point of the node mirror (gp_slip_node.py:79-83) -- run against
in-process doubles of `rospy` and `core_nav.msg` (this image has no ROS): node name, topics and queue size as the
reference, and a GP_Input delivered through the registered subscriber comes out of the registered publisher as the
GP_Output the oracle computes."""
import sys
import types

import numpy as np
import pytest

from conftest import load_golden
from oracle import gp_oracle as go

pytestmark = pytest.mark.gpu


def test_gaussian_process_wiring(monkeypatch):
    import corenav_gp_amd.gp_slip_node as node
    g = load_golden("slipval_window_rbfbrownian")
    rec = {"published": [], "subs": [], "pubs": [], "node": None, "spins": 0}

    class RosIn:
        def __init__(self):
            self.time_array, self.slip_array = [], []

    class RosOut:
        def __init__(self):
            self.mean, self.sigma = [], []

    class Publisher:
        def __init__(self, topic, typ, queue_size=None):
            rec["pubs"].append((topic, typ, queue_size))

        def publish(self, m):
            rec["published"].append(m)

    rospy = types.ModuleType("rospy")
    rospy.Publisher = Publisher
    rospy.init_node = lambda name: rec.__setitem__("node", name)
    rospy.Subscriber = lambda topic, typ, cb: rec["subs"].append((topic, typ, cb))

    def spin():                       # the middleware delivers one window, then the node shuts down
        rec["spins"] += 1
        m = RosIn()
        m.time_array, m.slip_array = list(g["time_array"]), list(g["slip_array"])
        rec["subs"][0][2](m)

    rospy.spin = spin
    core_nav, msg = types.ModuleType("core_nav"), types.ModuleType("core_nav.msg")
    msg.GP_Input, msg.GP_Output = RosIn, RosOut
    core_nav.msg = msg
    for name, mod in (("rospy", rospy), ("core_nav", core_nav), ("core_nav.msg", msg)):
        monkeypatch.setitem(sys.modules, name, mod)

    node.gaussian_process()
    assert rec["node"] == "gp_slip_node" and rec["spins"] == 1                      # gp_slip_node.py:80
    assert rec["pubs"] == [("/core_nav/core_nav/gp_result", RosOut, 1)]             # :12
    assert [(t, ty) for t, ty, _ in rec["subs"]] == [("/core_nav/core_nav/gp_input", RosIn)]   # :81
    assert len(rec["published"]) == 1 and isinstance(rec["published"][0], RosOut)
    out = rec["published"][0]
    # the node optimises per window (gp_slip_node.py:36): evaluate the oracle at the theta it reports
    assert len(out.mean) == len(out.sigma) == 599
    import corenav_gp_amd.engine as engine
    ctx = engine.Context(max_n=256, max_m=1024, max_d=1)
    _, _, th = ctx.slip_node_callback_opt(g["time_array"], g["slip_array"], np.ones(4))
    em, es = go.slip_node_callback(g["time_array"], g["slip_array"], th)
    assert np.max(np.abs(np.asarray(out.mean) - em)) <= 1e-6 * np.max(np.abs(em))
    assert np.max(np.abs(np.asarray(out.sigma) - es) / es) < 1e-6
