"""Drop-in for the hot-path part of the reference's ``sunflower/utils/conversion.py``:
``procrustes_to_rotmat`` (:54-58), ``R2E``/``E2R`` (:45-51), ``get_pose_mat`` (:61-76).
The Procrustes projection runs on the GPU (flope_procrustes); the Euler helpers are
host-side scipy exactly as in the reference."""
import numpy as np
import torch
from scipy.spatial.transform import Rotation as _Rot

from flope_amd import engine as _engine


def procrustes_to_rotmat(inp: torch.Tensor) -> torch.Tensor:
    """[...,9] or [...,3,3] -> [N,3,3] closest rotations (special orthogonal Procrustes)."""
    return _engine.procrustes(inp)


def R2E(R):
    """rotation matrix/matrices -> 'zyx' Euler angles in degrees"""
    return _Rot.from_matrix(R).as_euler("zyx", degrees=True)


def E2R(E):
    """'zyx' Euler angles in degrees -> rotation matrix/matrices"""
    return _Rot.from_euler("zyx", E, degrees=True).as_matrix()


def get_pose_mat(trans_rot):
    """(N,12) rows [t(3), R.flatten()(9)] -> (N,4,4) homogeneous poses"""
    tr = np.asarray(trans_rot, dtype=np.float64).reshape(-1, 12)
    pose = np.tile(np.eye(4), (tr.shape[0], 1, 1))
    pose[:, :3, 3] = tr[:, :3]
    pose[:, :3, :3] = tr[:, 3:].reshape(-1, 3, 3)
    return pose


def qvec2rotmat(quat):
    """scalar-last quaternion(s) [x,y,z,w] -> rotation matrix/matrices (reference conversion.py:37-38)"""
    return _Rot.from_quat(quat).as_matrix()


def rotmat2qvec(rotmat):
    """rotation matrix/matrices -> scalar-last quaternion(s) [x,y,z,w] (reference conversion.py:41-42)"""
    return _Rot.from_matrix(rotmat).as_quat()
