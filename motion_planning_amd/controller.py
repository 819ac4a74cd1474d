"""ROS-less mirror of the reference's MPPI node shell ``Controller``
(moribots/motion_planning control/src/mppi:296-389): the odom -> cmd_vel state machine for
waypoint following and parallel parking.  ROS itself is out of scope;
the boundary is the same: a pose comes in per odometry message, a (vx, wz) twist goes out.
"""
import math

import numpy as np

from .mppi import MPPI, WHEEL_BASE, WHEEL_RADIUS


def wheels_to_twist(wheel_vels):
    """Controller.wheelsToTwist, control/src/mppi:319-325."""
    ul, ur = wheel_vels[0], wheel_vels[1]
    vx = WHEEL_RADIUS * (ul + ur) / 2.0
    wz = WHEEL_RADIUS * (-ul + ur) / WHEEL_BASE
    return vx, wz


def yaw_from_quaternion(qx, qy, qz, qw):
    """yaw of tf.transformations.euler_from_quaternion(q)[2] (control/src/mppi:330-335)."""
    return math.atan2(2.0 * (qw * qz + qx * qy), 1.0 - 2.0 * (qy * qy + qz * qz))


class Controller(object):
    """``waypoints`` empty -> parallel park to [0, -1, 0] (control/src/mppi:305-309, :336-337);
    otherwise cycle through the [x, y] waypoints.  ``publish(vx, wz)`` replaces the cmd_vel
    publisher; every published twist is also appended to ``self.sent``."""

    def __init__(self, waypoints=(), mppi=None, publish=None, **mppi_kwargs):
        self.mppi = mppi if mppi is not None else MPPI(**mppi_kwargs)  # :298
        self.publish = publish
        self.sent = []
        self.done = False
        if not waypoints:  # :305-309
            self.parallel_park = True
            self.waypoints = []
        else:
            self.parallel_park = False
            self.waypoints = [list(w) for w in waypoints]
        self.idx = 0
        self.init = True
        self.state = self.mppi.start
        self._pub(0.0, 0.0)  # :312-316

    def _pub(self, vx, wz):
        self.sent.append((vx, wz))
        if self.publish is not None:
            self.publish(vx, wz)

    def _goal_from_waypoint(self):
        goal = self.waypoints[self.idx]  # goal only contains x, y (:347-352)
        theta = np.arctan2(goal[1] - self.mppi.start[1], goal[0] - self.mppi.start[0])
        self.mppi.goal = np.array([goal[0], goal[1], theta])
        self.state = self.mppi.start

    def pos_cb(self, x, y, theta):
        """One odometry callback (control/src/mppi:327-389) with the pose already reduced to
        (x, y, yaw).  Returns the published (vx, wz)."""
        self.mppi.start = np.array([x, y, theta])
        if self.parallel_park:
            self.mppi.goal = np.array([0.0, -1.0, 0.0])
        far = np.linalg.norm(self.mppi.start[:2] - self.mppi.goal[:2]) > self.mppi.thresh
        if far and not self.init:  # :339-343
            self.state = self.mppi.get_path(self.mppi.start, self.mppi.goal)
            self.done = False
        elif self.init:  # :344-355
            self.mppi.initialize()
            if not self.parallel_park:
                self._goal_from_waypoint()
            self.init = False
        else:  # :356-375
            if not self.parallel_park:
                self.idx = 0 if self.idx + 1 >= len(self.waypoints) else self.idx + 1
                self.mppi.initialize()
                self._goal_from_waypoint()
            else:
                self.done = True
        u = self.mppi.uvec[-1, :] if not self.done else np.array([0.0, 0.0])  # :377-381
        vx, wz = wheels_to_twist(u)
        self._pub(vx, wz)
        return vx, wz

    def odom_cb(self, px, py, qx, qy, qz, qw):
        """Same, from the raw odometry pose (position + orientation quaternion)."""
        return self.pos_cb(px, py, yaw_from_quaternion(qx, qy, qz, qw))
