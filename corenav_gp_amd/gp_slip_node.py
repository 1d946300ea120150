"""Drop-in for core_navigation/script/gp_slip_node.py: same node name, same topics, same callback
data flow -- the GPy calls are replaced by the HIP engine through the C ABI.

  reference                                      here
  ---------------------------------------------  ------------------------------------------------
  GPy.kern.RBF(1) * GPy.kern.Brownian(1)   :31   KERNEL_RBF_BROWNIAN + theta
  GPy.models.GPRegression(...)             :35   engine.Context.fit (cgp_fit)
  m.optimize()                             :36   engine.Context.optimize (cgp_optimize): L-BFGS on the
                                                 Logexp-transformed parameters, device gradients
  for x in X_: m.predict([[x]])         :45-49   engine.Context.predict (cgp_predict), all M at once
  msg_out.mean / .sigma                 :59-61   mean[n:], 2*sqrt(var[n:])

rospy is optional: without it `callback` is still usable in process (tests, replay harness).
"""
from __future__ import annotations

import numpy as np

from . import engine

NODE_NAME = "gp_slip_node"                               # gp_slip_node.py:80
SUB_TOPIC = "/core_nav/core_nav/gp_input"                # gp_slip_node.py:81
PUB_TOPIC = "/core_nav/core_nav/gp_result"               # gp_slip_node.py:12
PUB_QUEUE_SIZE = 1
# Hyper-parameters [sigma_rbf^2, ell, sigma_brownian^2, sigma_n^2].  The reference re-optimises them
# per window starting from GPy's defaults (every parameter 1.0); `optimize=False` evaluates at a
# fixed theta instead (the batched / benchmark mode).
GPY_START_THETA = (1.0, 1.0, 1.0, 1.0)
DEFAULT_THETA = (0.5, 30.0, 0.01, 0.002)
MAX_EVALS = 1000                                         # paramz 'lbfgsb' max_iters default


class GP_Input:                                          # core_navigation/msg/GP_Input.msg
    def __init__(self, time_array=(), slip_array=()):
        self.header = None
        self.time_array = list(time_array)
        self.slip_array = list(slip_array)


class GP_Output:                                         # core_navigation/msg/GP_Output.msg
    def __init__(self):
        self.header = None
        self.mean = []
        self.sigma = []


class GpSlipNode:
    def __init__(self, theta=None, device=0, publisher=None, optimize=True):
        self.optimize = optimize
        if theta is None:
            theta = GPY_START_THETA if optimize else DEFAULT_THETA
        self.theta = np.asarray(theta, dtype=np.float64)
        self.last_theta = self.theta.copy()
        self.ctx = engine.Context(device=device, max_n=256, max_m=1024, max_d=1, max_batch=1, dtype=engine.F64)
        self.publisher = publisher

    def callback(self, data):
        """gp_slip_node.py:16-63.  Returns the GP_Output it publishes."""
        if self.optimize:   # gp_slip_node.py:36: a fresh model per window, optimised from the start values
            mean, sigma, self.last_theta = self.ctx.slip_node_callback_opt(data.time_array, data.slip_array,
                                                                           self.theta, max_evals=MAX_EVALS)
        else:
            mean, sigma = self.ctx.slip_node_callback(data.time_array, data.slip_array, self.theta)
        msg_out = GP_Output()
        msg_out.mean = mean
        msg_out.sigma = sigma
        if self.publisher is not None:
            self.publisher(msg_out)
        return msg_out


def gaussian_process():                                  # gp_slip_node.py:79-83
    import rospy  # noqa: only on a ROS machine
    from core_nav.msg import GP_Input as RosIn, GP_Output as RosOut

    pub = rospy.Publisher(PUB_TOPIC, RosOut, queue_size=PUB_QUEUE_SIZE)

    def publish(m):
        out = RosOut()
        out.mean, out.sigma = list(m.mean), list(m.sigma)
        pub.publish(out)

    node = GpSlipNode(publisher=publish)
    rospy.init_node(NODE_NAME)
    rospy.Subscriber(SUB_TOPIC, RosIn, node.callback)
    rospy.spin()


if __name__ == "__main__":
    gaussian_process()
