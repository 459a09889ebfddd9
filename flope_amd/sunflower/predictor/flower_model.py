"""Drop-in for the reference's ``sunflower/predictor/flower_model.py`` (``FlowerModel``, :29-255;
``get_kalman_filter``, :18-26): the multi-frame step right after the pose hot path.

    add_data(rgb, depth, cam_pose[7]) -> per-frame flower poses in the camera frame (the hot path),
    moved to the world frame (cam_pose @ pose), turned into 7-vectors [x y z | qx qy qz qw] and
    associated to tracks by nearest translation (50 mm gate); every track is a 7-state
    identity-model Kalman filter whose quaternion part is re-normalised after each update.

Host-side control logic (tens of 7x7 operations per frame) and therefore plain numpy, as in the
reference.  What is deliberately kept from the reference, quirks included:
  * distances are measured to ``self.state`` -- the FIRST measurement of each track, which is never
    overwritten by the filter (flower_model.py:180-185, :211); the filtered values live in ``self.kfs``;
  * ``add_data(..., ignore=True)`` is what feeds the tracker; the default ``ignore=False`` only returns the
    per-frame poses (:244-245);
  * every measurement of a frame is matched against the tracks independently (two detections may update
    the same track in one frame), and an unmatched one opens a new track that later detections of the same
    frame cannot match (the distance matrix is computed once per frame, :181).
What is not reproduced: the matplotlib live plots (``get_plots``; display code) and the hard-coded
checkpoint / intrinsics paths -- the pose predictor is passed in.

The reference takes its filter from ``filterpy==1.4.5`` (not vendored).  ``KalmanFilter7`` restates
that class's published predict / update for the only configuration the reference uses
(F = H = I, Q = 1e-3 I, R = 0.1 I, P0 = I): predict  x <- F x, P <- F P F^T + Q;
update  y = z - H x, S = H P H^T + R, K = P H^T S^-1, x <- x + K y, P <- (I-KH) P (I-KH)^T + K R K^T.
"""
import numpy as np

from sunflower.utils.conversion import qvec2rotmat, rotmat2qvec
from sunflower.utils.mvg import pose_cam_to_world


class KalmanFilter7:
    """The subset of filterpy.kalman.KalmanFilter the reference touches: x, F, H, P, Q, R, predict(), update(z)."""

    def __init__(self, dim_x=7, dim_z=7):
        self.x = np.zeros(dim_x)
        self.F = np.eye(dim_x)
        self.H = np.eye(dim_z, dim_x)
        self.P = np.eye(dim_x)
        self.Q = np.eye(dim_x)
        self.R = np.eye(dim_z)

    def predict(self):
        self.x = self.F @ self.x
        self.P = self.F @ self.P @ self.F.T + self.Q

    def update(self, z):
        z = np.asarray(z, dtype=np.float64)
        y = z - self.H @ self.x
        PHT = self.P @ self.H.T
        S = self.H @ PHT + self.R
        K = PHT @ np.linalg.inv(S)
        self.x = self.x + K @ y
        I_KH = np.eye(self.P.shape[0]) - K @ self.H
        self.P = I_KH @ self.P @ I_KH.T + K @ self.R @ K.T


def get_kalman_filter(initial_value):
    kf = KalmanFilter7(dim_x=7, dim_z=7)
    kf.x = np.array(initial_value, dtype=np.float64)
    kf.F = np.eye(7)
    kf.H = np.eye(7)
    kf.P = np.eye(7)
    kf.Q = np.eye(7) * 0.001
    kf.R = np.eye(7) * 0.1
    return kf


def cam_pose_to_matrix(cam_pose):
    """[tx ty tz qx qy qz qw] -> 4x4 camera pose (flower_model.py:224-227)"""
    cam_pose = np.asarray(cam_pose, dtype=np.float64)
    m = np.eye(4)
    m[:3, :3] = qvec2rotmat(cam_pose[3:])
    m[:3, 3] = cam_pose[:3]
    return m


def poses_to_measurements(flower_pose_world):
    """(N,4,4) world poses -> (N,7) rows [translation, scalar-last quaternion] (flower_model.py:238-242)"""
    p = np.asarray(flower_pose_world)
    return np.hstack((p[:, :3, 3], rotmat2qvec(p[:, :3, :3])))


class FlowerModel:
    def __init__(self, dist_th=50, intrin_path=None, get_plots=False, pose_predictor=None):
        if get_plots:
            raise NotImplementedError("the live matplotlib plots of the reference are display code and are not reproduced")
        self.get_plots = False
        self.state = None
        self.scores = None
        self.kfs = []
        self.th = dist_th / 1000
        self.intrin_path = intrin_path
        self.pose_predictor = pose_predictor          # anything with get_flower_poses(rgb, depth)

    def assign_meas_to_state(self, meas):
        meas = np.asarray(meas, dtype=np.float64)
        if self.state is None:
            self.state = meas
            self.scores = np.ones(meas.shape[0])
            for each_meas in meas:
                self.kfs.append(get_kalman_filter(each_meas))
            return
        d = np.linalg.norm(meas[:, None, :3] - self.state[None, :, :3], axis=2)      # cdist(meas, state)
        min_idx = np.argmin(d, axis=1)
        good = np.min(d, axis=1) < self.th
        for i in range(meas.shape[0]):
            if good[i]:
                kf = self.kfs[min_idx[i]]
                kf.predict()
                kf.update(meas[i])
                kf.x[3:] /= np.linalg.norm(kf.x[3:])
                self.scores[min_idx[i]] += 1
            else:
                self.state = np.vstack((self.state, meas[i].reshape(1, 7)))
                self.scores = np.hstack((self.scores, np.array([1])))
                self.kfs.append(get_kalman_filter(meas[i]))

    def add_poses(self, flower_pose_cam, cam_pose, ignore=False):
        """flower_model.py:224-255 after the predictor call: camera-frame poses of one frame -> world frame -> tracker."""
        if flower_pose_cam is None:
            return None, None
        flower_pose = pose_cam_to_world(flower_pose_cam, cam_pose_to_matrix(cam_pose))
        if ignore:
            self.assign_meas_to_state(poses_to_measurements(flower_pose))
        return flower_pose_cam, flower_pose.astype(np.float32)

    def add_data(self, rgb, depth, cam_pose, ignore=False):
        if self.pose_predictor is None:
            raise RuntimeError("FlowerModel: pass pose_predictor= (PosePredictor / FastPosePredictor)")
        return self.add_poses(self.pose_predictor.get_flower_poses(rgb, depth), cam_pose, ignore)

    def add_stream(self, frames, ignore=True, detectors=2):
        """`add_data` over a stream of (rgb, depth, cam_pose) frames with the predictor's pipelined loop
        (`FastPosePredictor.iter_flower_poses`: several frames in flight on the GPU, results in frame order) instead of one
        blocking `get_flower_poses` per frame.  Same tracker updates in the same order as calling add_data frame by frame
        (build-defined convenience: the reference's loop, scripts/live_pose.py:31-41, is sequential).  Yields add_data's
        (poses in the camera frame, poses in the world frame) per frame."""
        if self.pose_predictor is None:
            raise RuntimeError("FlowerModel: pass pose_predictor= (PosePredictor / FastPosePredictor)")
        frames = list(frames)
        it = getattr(self.pose_predictor, "iter_flower_poses", None)
        poses = it(((f[0], f[1]) for f in frames), detectors=detectors) if it is not None else \
            (self.pose_predictor.get_flower_poses(f[0], f[1]) for f in frames)
        for f, pose_cam in zip(frames, poses):
            yield self.add_poses(pose_cam, f[2], ignore)

    def get_state(self):
        return self.state

    def filtered_state(self):
        """(T,7) current filter means, one row per track (the reference keeps these only inside ``kfs``)."""
        return np.array([kf.x for kf in self.kfs]).reshape(-1, 7)
