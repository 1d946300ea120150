"""ROS-free closed-loop replay of the corenav-GP slip loop (the stand-in for
BASELINE configs[4]: the Pathfinder bag is not obtainable, so the sensor stream is synthetic).

    RoverSim (wheel + INS speeds, 10 Hz; IMU specific force + rate, 50 Hz --> FilterCovariance) --> SlipRecorder [CoreNav::Update, C++]
        --/core_nav/core_nav/gp_input-->   GP engine [gp_slip_node.py arithmetic, HIP]
        --/core_nav/core_nav/gp_result-->  GpPredictor [gp_predictor.cpp, C++]  <-- stopping_service
        --/core_nav/core_nav/stop_cmd-->   SlipRecorder.stopCallback + DriveStraightWithStop FSM
                                           (pathfinder_control/src/drive_straight_with_stop.cpp:28-66)
        rover stopped --> zero updates shrink P (FilterCovariance) --> the NEXT window's SetStopping answer

An ensemble of trajectories (Monte-Carlo) is stepped in lock-step; the windows published at a tick
are fitted together through the batched C ABI.
"""
from __future__ import annotations

import numpy as np

from . import engine, synth

DT_ODO = 0.1            # odometry period, parameters.yaml:46-47 (10 Hz)
STOP_DURATION = 5.0     # drive_straight_with_stop.cpp:7
DRIVE_SPEED = 0.8       # m/s; gp_predictor.cpp:73 assumes 0.8 / (1 - slip)


class RoverSim:
    """Straight drive with a slowly varying + impulsive slip process; wheel speeds are the commanded
    speed plus encoder noise, the INS forward speed is wheel * (1 - slip)."""

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.imu_rng = np.random.default_rng(seed + 7919)   # its own generator: the slip sequence does not depend on the IMU stream
        self.tick = 0
        self.v_prev = 0.0

    def step(self, driving):
        self.tick += 1
        if not driving:
            return (0.0, 0.0, 0.0, 0.0), 0.0
        t = float(self.tick)
        slip = 0.1 * np.sin(2.0 * np.pi * t / 40.0) + 0.05 * (self.rng.random() < 0.04) + self.rng.normal(0.0, 0.03)
        wheels = tuple(DRIVE_SPEED * (1.0 + 0.002 * self.rng.normal()) for _ in range(4))
        vlin = float(np.mean(wheels)) * (1.0 - float(np.clip(slip, -0.9, 0.9)))
        return wheels, vlin

    def imu(self, vlin):
        """The IMU_PER_ODO body-frame samples (specific force f_b [m/s^2], angular rate w_b [rad/s]) of one odometry period, 50 Hz
        (parameters.yaml:46: the reference's ADIS IMU rate): forward acceleration between two odometry speeds, gravity on z, a
        slowly wandering yaw rate, sensor noise at the levels of CoreNav's Q (sig_gyro_inRun / sig_accel_inRun)."""
        a_x = (vlin - self.v_prev) / DT_ODO
        self.v_prev = vlin
        out = []
        for _ in range(IMU_PER_ODO):
            f_b = np.array([a_x, 0.0, -9.80665]) + self.imu_rng.normal(0.0, 0.02, 3)
            w_b = np.array([0.0, 0.0, 0.01 * np.sin(self.tick / 50.0)]) + self.imu_rng.normal(0.0, 1e-3, 3)
            out.append((f_b, w_b))
        return out


class DriveStraightWithStop:
    """drive_straight_with_stop.cpp FSM: after stop_cmd, wait until time_to_stop, publish zero
    velocity for 5 s, then resume."""

    def __init__(self):
        self.state, self.stop_commanded, self.time_to_stop, self.stopped_time = "waiting", False, 0.0, 0.0

    def stop_callback(self, now, delta):          # :60-66
        self.time_to_stop = now + delta
        self.stop_commanded = True

    def cmd(self, now):                           # :28-51
        if self.state == "waiting":
            if self.stop_commanded and now > self.time_to_stop:
                self.stopped_time, self.state = now, "stopped"
        elif now - self.stopped_time > STOP_DURATION:
            self.state, self.stop_commanded = "waiting", False
        return 0.0 if self.state == "stopped" else 1.0


IMU_PER_ODO = 5         # parameters.yaml:46-47: IMU 50 Hz, odometry 10 Hz (gp_predictor.cpp:64 walks 5 steps per prediction too)
# CoreNav::Init, CoreNav.cpp:1026-1059: the zero-update measurement model (ZARU on the gyro-bias states 12..14, ZUPT on the
# velocity states 3..5) and its noise
_H_ZERO = np.zeros((6, 15))
for _r, _c in enumerate((12, 13, 14, 3, 4, 5)):
    _H_ZERO[_r, _c] = -1.0
_R_ZERO = np.diag([0.01 ** 2, 0.01 ** 2, 0.0025 ** 2, 0.02 ** 2, 0.02 ** 2, 1.0 ** 2])
# the odometry measurement noise at zero slip variance: gp_predictor.cpp:80-88 with chi_UT_est_cov = 0 (the floors 0.03^2,
# 0.03^2, 0.05^2, 0.05^2 through R_IP_1, times 25) -- what the look-ahead itself assumes when the GP predicts no slip
_R1 = np.array([[0.5, 0.5, 0.0, 0.0], [1 / 0.685, -1 / 0.685, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
_R_ODO = 25.0 * _R1 @ np.diag([0.03 ** 2, 0.03 ** 2, 0.05 ** 2, 0.05 ** 2]) @ _R1.T


class FilterCovariance:
    """The part of CoreNav's covariance bookkeeping the SetStopping answer depends on -- NOT the EKF (out of
    scope; no state, no mechanisation, fixed synthetic STM / Q / H per trajectory): per odometry tick the 15 x 15 error
    covariance is propagated IMU_PER_ODO times, `P = STM P STM' + Q` (CoreNav::Propagate, CoreNav.cpp:104); while the
    rover drives the tick ends with the odometry update in Joseph form (CoreNav::Update, :190-242 -- with the filter's own
    H and the zero-slip R above: the recursion the reference's look-ahead itself runs, gp_predictor.cpp:66-91); while it
    stands still every IMU step applies the zero-velocity / zero-angular-rate update instead (CoreNav.cpp:139-142,
    zeroUpdate :390-409).  `snapshot()` is what CoreNav::Update keeps at the stopRecording tick (`P_pred = P_`, :291-292)
    for setStopping_ to serve (:652-676).  So the uncertainty a look-ahead starts from grows over the mission, and a stop
    takes out what the zero updates can observe (velocity error and the position error correlated with it): the stop ->
    ZUPT -> smaller P -> later next stop feedback of the reference's loop."""

    def __init__(self, P, Q, STM, H, zero_updates_enabled=True):
        self.P = np.array(P, dtype=np.float64).reshape(15, 15)
        self.Q = np.asarray(Q, dtype=np.float64).reshape(15, 15)
        self.STM0 = np.asarray(STM, dtype=np.float64).reshape(15, 15)   # the trajectory's synthetic transition matrix at rest
        self.STM = self.STM0.copy()                                     # ... and of the latest IMU step (what setStopping_ serves)
        self.H = np.asarray(H, dtype=np.float64).reshape(4, 15)
        self.P_pred = self.P.copy()
        self.zero_updates = 0
        self.zero_updates_enabled = zero_updates_enabled   # False: the control of the feedback test (a stop that corrects nothing)

    @staticmethod
    def _joseph(P, H, R):
        K = P @ H.T @ np.linalg.inv(H @ P @ H.T + R)
        IKH = np.eye(15) - K @ H
        return IKH @ P @ IKH.T + K @ R @ K.T

    @staticmethod
    def _skew(v):
        return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])

    def imu_transition(self, f_b, w_b):
        """The IMU-dependent blocks of the error-state transition over one 20 ms IMU step (the first-order terms every INS
        error model has -- CoreNav builds its STM_ from the same measurements in insErrorStateModel_LNF, CoreNav.cpp:104 is where
        it is applied): attitude error rotates with the measured rate, velocity error picks up -[f x] times the attitude
        error; everything else stays the trajectory's synthetic matrix.  States: 0-2 attitude, 3-5 velocity (CoreNav's order)."""
        dt = DT_ODO / IMU_PER_ODO
        STM = self.STM0.copy()
        STM[0:3, 0:3] -= self._skew(w_b) * dt
        STM[3:6, 0:3] -= self._skew(f_b) * dt
        return STM

    def odometry_tick(self, stopped, imu=None):
        """One odometry period: IMU_PER_ODO propagations, each with the transition matrix of ITS IMU sample when a stream is
        given (imu = [(f_b, w_b)] * IMU_PER_ODO, RoverSim.imu), else with the fixed matrix (round 5's form)."""
        for j in range(IMU_PER_ODO):
            if imu is not None:
                self.STM = self.imu_transition(*imu[j])
            self.P = self.STM @ self.P @ self.STM.T + self.Q
            if stopped and self.zero_updates_enabled:   # |rearVel_| < 0.005 (CoreNav.cpp:139): the commanded stop holds the wheels
                self.P = self._joseph(self.P, _H_ZERO, _R_ZERO)
                self.zero_updates += 1
        if not stopped:
            self.P = self._joseph(self.P, self.H, _R_ODO)

    def snapshot(self):
        self.P_pred = self.P.copy()


def batched_odometry_tick(covs, stopped, imus):
    """FilterCovariance.odometry_tick for a whole ensemble at once (the per-trajectory form is 5 x 3 small matrix products and up to
    six 15 x 15 Joseph updates per tick in Python: 64 trajectories x 600 ticks spent 2.5 of the replay's 3 s of wall clock there).
    Stacked matmul / inv run the same routine per trajectory as the one-at-a-time form: the same numbers.  covs: FilterCovariance
    objects; stopped: bools; imus: per trajectory the IMU_PER_ODO (f_b, w_b) samples, or None (fixed matrix)."""
    n = len(covs)
    if n == 0:
        return
    P = np.stack([c.P for c in covs])
    Q = np.stack([c.Q for c in covs])
    STM0 = np.stack([c.STM0 for c in covs])
    STM = np.stack([c.STM for c in covs])
    stopped = np.asarray(stopped, dtype=bool)
    zu = np.array([c.zero_updates_enabled for c in covs], dtype=bool) & stopped
    has_imu = np.array([im is not None for im in imus], dtype=bool)
    dt = DT_ODO / IMU_PER_ODO
    I15 = np.eye(15)

    def skew(v):   # (m, 3) -> (m, 3, 3)
        z = np.zeros(len(v))
        return np.stack([np.stack([z, -v[:, 2], v[:, 1]], 1), np.stack([v[:, 2], z, -v[:, 0]], 1), np.stack([-v[:, 1], v[:, 0], z], 1)], 1)

    def joseph(Pm, H, R):
        Ht = np.swapaxes(H, -1, -2)
        K = Pm @ Ht @ np.linalg.inv(H @ Pm @ Ht + R)
        IKH = I15 - K @ H
        return IKH @ Pm @ np.swapaxes(IKH, -1, -2) + K @ R @ np.swapaxes(K, -1, -2)

    for j in range(IMU_PER_ODO):
        if has_imu.any():
            idx = np.nonzero(has_imu)[0]
            f_b = np.stack([imus[i][j][0] for i in idx])
            w_b = np.stack([imus[i][j][1] for i in idx])
            S = STM0[idx].copy()
            S[:, 0:3, 0:3] -= skew(w_b) * dt
            S[:, 3:6, 0:3] -= skew(f_b) * dt
            STM[idx] = S
        P = STM @ P @ np.swapaxes(STM, -1, -2) + Q
        if zu.any():
            P[zu] = joseph(P[zu], _H_ZERO[None], _R_ZERO[None])
    drv = ~stopped
    if drv.any():
        H = np.stack([covs[i].H for i in np.nonzero(drv)[0]])
        P[drv] = joseph(P[drv], H, _R_ODO[None])
    for i, c in enumerate(covs):
        c.P = P[i]
        c.STM = STM[i]
        if zu[i]:
            c.zero_updates += IMU_PER_ODO


class Trajectory:
    def __init__(self, seed, evolve_filter=True, zero_updates=True, imu_stream=True):
        self.sim = RoverSim(seed)
        self.imu_stream = imu_stream   # False: every propagation with the trajectory's fixed matrix (round 5)
        self.rec = engine.SlipRecorder()
        self.drv = DriveStraightWithStop()
        self.P, self.Q, self.STM, self.Hvec, self.pos, H = synth.filter_state(seed, with_H=True)
        # evolve_filter = False: the round-4 stand-in (every window answered from the same static snapshot)
        self.cov = FilterCovariance(self.P, self.Q, self.STM, H, zero_updates) if evolve_filter else None
        self.windows, self.results, self.stop_cmds, self.stops = [], [], [], 0
        self.served = []   # per published window: (the P_pred served, zero updates applied so far)
        self._was_stopped = False

    def tick(self, now):
        self.sense(now)
        if self.cov is not None:
            # the filter runs at the IMU rate: five 50 Hz samples per 10 Hz odometry tick, each with its own transition matrix
            self.cov.odometry_tick(self._cmd == 0.0, self._imu)
        return self.publish()

    def sense(self, now):
        """First half of a tick: the drive command, the wheel / INS speeds and the IMU samples of this odometry period."""
        cmd = self.drv.cmd(now)
        if cmd == 0.0 and not self._was_stopped:
            self.stops += 1
        self._was_stopped = cmd == 0.0
        self.rec.cmd_callback(cmd)
        self._wheels, self._vlin = self.sim.step(cmd != 0.0)
        self._cmd = cmd
        self._imu = self.sim.imu(self._vlin) if (self.cov is not None and self.imu_stream) else None

    def publish(self):
        """Second half (after the covariance update of the tick): the recorder, and the snapshot a published window is served from."""
        wheels, vlin, cmd = self._wheels, self._vlin, self._cmd
        win = self.rec.update(*wheels, vlin, cmd)
        if win is not None and self.cov is not None:
            # CoreNav.cpp:289-305: the snapshot is taken at the tick that publishes the window; setStopping_ (:652-676) serves
            # P_pred together with the filter's CURRENT Q_, STM_, H_
            self.cov.snapshot()
            self.P = self.cov.P_pred.reshape(225).copy()
            self.STM = self.cov.STM.reshape(225).copy()
            self.served.append((self.P.copy(), self.cov.zero_updates, self.STM.copy()))
        return win


class ClosedLoopEnsemble:
    def __init__(self, n_traj, theta=(0.5, 30.0, 0.01, 0.002), optimize=False, device=0, seed=synth.SEED_BASE + 5,
                 evolve_filter=True, zero_updates=True, imu_stream=True):
        self.traj = [Trajectory(seed + 31 * i, evolve_filter, zero_updates, imu_stream) for i in range(n_traj)]
        self.theta = np.asarray(theta, dtype=np.float64)
        self.optimize = optimize
        self.ctx = engine.Context(device=device, max_n=256, max_m=1024, max_d=1, max_batch=max(n_traj, 1))   # max_m >= N: optimiser needs it
        self.now = 0.0

    def _fit_windows(self, idx, wins):
        """gp_slip_node.py:16-63 for every window published this tick (batched when theta is fixed)."""
        outs = []
        same = len({len(t) for t, _ in wins}) == 1
        if same and len(wins) > 1:
            n = len(wins[0][0])
            ntr = int(0.9 * n)
            X = np.stack([t[:ntr, None] for t, _ in wins])
            y = np.stack([s[:ntr] for _, s in wins])
            Xs = np.stack([(t.min() + n + np.arange(int(np.ceil(t.max() + 600 - t.min())) - n))[:, None] for t, _ in wins])
            if self.optimize:   # gp_slip_node.py:36 for the whole ensemble: one batched L-BFGS from GPy's start values
                th, _, _ = self.ctx.optimize_batch(X, y, engine.KERNEL_RBF_BROWNIAN, np.ones(4))
            else:
                th = np.tile(self.theta, (len(wins), 1))
            rc, mean, var, _, info = self.ctx.fit_predict_batch(X, y, Xs, th, engine.KERNEL_RBF_BROWNIAN)
            assert rc == 0, info
            return [(mean[i], 2.0 * np.sqrt(var[i]), th[i]) for i in range(len(wins))]
        for t, s in wins:
            if self.optimize:
                m, sg, th = self.ctx.slip_node_callback_opt(t, s, np.ones(4))
            else:
                m, sg, th = (*self.ctx.slip_node_callback(t, s, self.theta), self.theta)
            outs.append((m, sg, th))
        return outs

    def step(self):
        self.now += DT_ODO
        for tr in self.traj:
            tr.sense(self.now)
        live = [tr for tr in self.traj if tr.cov is not None]
        batched_odometry_tick([tr.cov for tr in live], [tr._cmd == 0.0 for tr in live], [tr._imu for tr in live])   # the ensemble's filters in one go
        pubs = [(i, w) for i, tr in enumerate(self.traj) if (w := tr.publish()) is not None]
        if not pubs:
            return 0
        outs = self._fit_windows([i for i, _ in pubs], [w for _, w in pubs])
        for (i, w), (mean, sigma, th) in zip(pubs, outs):
            self.traj[i].windows.append(w)
            self.traj[i].results.append((mean, sigma, th))
        trs = [self.traj[i] for i, _ in pubs]
        if len(pubs) > 1 and len({len(o[0]) for o in outs}) == 1:
            # the ensemble's look-aheads in one launch (one wavefront per trajectory)
            fired, cmds, _, _ = self.ctx.predict_stop_batch(
                np.stack([o[0] for o in outs]), np.stack([o[1] for o in outs]), np.stack([t.P for t in trs]),
                np.stack([t.Q for t in trs]), np.stack([t.STM for t in trs]), np.stack([t.Hvec for t in trs]),
                np.stack([t.pos for t in trs]), self.now, self.now)
            answers = [(int(f), float(c)) for f, c in zip(fired, cmds)]
        else:
            # gp_predictor.cpp:17-132 through the C++ class, service answered from the filter state
            answers = [engine.gppredictor_callback(o[0], o[1], t.P, t.Q, t.STM, t.Hvec, t.pos, self.now, self.now)
                       for o, t in zip(outs, trs)]
        for tr, (npub, cmd) in zip(trs, answers):
            if npub:
                tr.stop_cmds.append(cmd)
                tr.rec.stop_callback(cmd)            # CoreNav::stopCallback
                tr.drv.stop_callback(self.now, cmd)  # drive_straight_with_stop stopCallback
        return len(pubs)

    def run(self, n_ticks):
        return sum(self.step() for _ in range(n_ticks))
