"""Host-side camera model of the hot path: K, W, H -> P[16] and the benchmark trajectory.

Mirrors the parts of the reference's CameraCalibration the projector uses
(reference: src/RTRenderer/include/CameraCalibration.h:8-54 -- K matrix, width,
height; distortion coefficients are parsed there but never used by the projector)
and the matrix composition of project_cloud.cu:318.
"""
import numpy as np


class CameraCalibration:
    """K (3x3 double), width, height -- same accessor names as the reference class
    (CameraCalibration.cpp:12-49)."""

    def __init__(self, K=None, width=640, height=480):
        # defaults follow CameraCalibration.cpp:5-10 (640x480)
        self.m_K = np.eye(3, dtype=np.float64) if K is None else np.asarray(K, dtype=np.float64).reshape(3, 3).copy()
        self.m_width, self.m_height = int(width), int(height)

    @classmethod
    def pinhole(cls, fx, fy, cx, cy, width, height):
        return cls(np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float64), width, height)

    def getIntrinsicsMatrix(self):
        return self.m_K

    def getWidth(self):
        return self.m_width

    def getHeight(self):
        return self.m_height

    def setWidth(self, w):
        self.m_width = int(w)

    def setHeight(self, h):
        self.m_height = int(h)


def compose_projection(K, E):
    """P = K4 * E in fp32, row-major float32[16] (project_cloud.cu:318 with
    project_cloud.h:50-59 and CameraCalibration.cpp:17-27: the two glm transposes
    cancel).  Every entry is the left-to-right sum of four separately rounded fp32
    products -- numpy float32 scalars round after each operation."""
    K = np.asarray(K, dtype=np.float64).reshape(3, 3)
    E = np.asarray(E, dtype=np.float64).reshape(4, 4)
    K4 = np.zeros((4, 4), np.float32)
    K4[:3, :3] = K.astype(np.float32)
    K4[3, 3] = np.float32(1)
    Ef = E.astype(np.float32)
    P = np.empty((4, 4), np.float32)
    with np.errstate(all="ignore"):
        for r in range(4):
            for c in range(4):
                s = np.float32(K4[r, 0] * Ef[0, c])
                s = np.float32(s + np.float32(K4[r, 1] * Ef[1, c]))
                s = np.float32(s + np.float32(K4[r, 2] * Ef[2, c]))
                s = np.float32(s + np.float32(K4[r, 3] * Ef[3, c]))
                P[r, c] = s
    return P.reshape(16)


def benchmark_calibration(width, height):
    """OPENCV pinhole, zero distortion, fx = fy = 0.8 W, cx = W/2, cy = H/2."""
    return CameraCalibration.pinhole(0.8 * width, 0.8 * width, width / 2.0, height / 2.0, width, height)


def orbit_pose(k, n_poses=1000, radius=1.5):
    """World->camera 4x4 (double) of pose k of the benchmark trajectory:
    camera centre on a circle of `radius` at y = 0, optical axis tangent to the circle,
    camera y axis = world y (OpenCV convention: x right, y down, z forward)."""
    a = 2.0 * np.pi * (k % n_poses) / n_poses
    c = np.array([radius * np.cos(a), 0.0, radius * np.sin(a)])
    fwd = np.array([-np.sin(a), 0.0, np.cos(a)])  # tangent
    down = np.array([0.0, 1.0, 0.0])
    right = np.cross(down, fwd)
    R = np.stack([right, down, fwd])  # rows = camera axes in world coordinates
    E = np.eye(4)
    E[:3, :3] = R
    E[:3, 3] = -R @ c
    return E


def orbit_projection(k, width, height, n_poses=1000):
    return compose_projection(benchmark_calibration(width, height).getIntrinsicsMatrix(), orbit_pose(k, n_poses))
