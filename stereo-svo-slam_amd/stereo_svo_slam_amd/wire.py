"""Wire layout of the reference's websocket backend, so that a viewer written for
it reads this library's results unchanged: the three resources of
SvoSlamBackend::text_message_received (src/app/svo_slam_backend.cpp:18-110) as compact JSON text.

    keyframes  + "get" -> [{"pose": {x,y,z,rx,ry,rz}, "keypoints": [{x,y,z}..], "colors": [{r,g,b}..]}..]
    pose               -> {"pose": {x,y,z,rx,ry,rz}}
    trajectory         -> {"trajectory": [x,y,z,rx,ry,rz, x,y,z,...]}       (raw pose angles)

Poses of "keyframes" and "pose" carry the robot angles of PoseManager::get_robot_angles
(src/lib/pose_manager.cpp:45-59): Rodrigues(R_z (R_x R_y)). Transport (the socket itself) is out of
scope; `handle()` maps (resource name, message) to the reply text.
"""
import json

import numpy as np

from .replay import _rodrigues, _rodrigues_inv


def robot_angles(pose):
    rx = _rodrigues([pose[3], 0, 0])
    ry = _rodrigues([0, pose[4], 0])
    rz = _rodrigues([0, 0, pose[5]])
    return _rodrigues_inv(rz @ (rx @ ry))


def _pose_object(pose):
    a = robot_angles(pose)
    return {"x": float(pose[0]), "y": float(pose[1]), "z": float(pose[2]),
            "rx": float(a[0]), "ry": float(a[1]), "rz": float(a[2])}


def keyframes_message(slam, seq=0):
    out = []
    for kid in range(slam.num_keyframes(seq)):
        kf = slam.get_keyframe(kid, seq)
        col = kf.info["color"] if len(kf.info) else np.zeros((0, 3), np.uint8)
        out.append({"pose": _pose_object(kf.pose),
                    "keypoints": [{"x": float(p[0]), "y": float(p[1]), "z": float(p[2])} for p in kf.kps3d],
                    "colors": [{"r": int(c[0]), "g": int(c[1]), "b": int(c[2])} for c in col]})
    return json.dumps(out, separators=(",", ":"))


def pose_message(slam, seq=0):
    return json.dumps({"pose": _pose_object(slam.pose(seq))}, separators=(",", ":"))


def trajectory_message(slam, seq=0):
    flat = [float(v) for p in slam.get_trajectory(seq) for v in p]
    return json.dumps({"trajectory": flat}, separators=(",", ":"))


def handle(resource, message, slam, seq=0):
    """Reply text for a request on `resource` (the last path element of the socket URL), or None
    where the reference sends nothing (:26-27: keyframes answers only to "get")."""
    resource = resource.rstrip("/").rsplit("/", 1)[-1]
    if resource == "keyframes":
        return keyframes_message(slam, seq) if message == "get" else None
    if resource == "pose":
        return pose_message(slam, seq)
    if resource == "trajectory":
        return trajectory_message(slam, seq)
    return None
