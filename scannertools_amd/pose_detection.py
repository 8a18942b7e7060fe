"""Python side of the OpenPose op: the ``Pose`` element type and the ``PoseList`` column reader.

Mirrors the public surface of the reference's ``scannertools_caffe/pose_detection.py`` (``Pose`` with its joint
indices, ``pose_keypoints`` / ``face_keypoints`` / ``hand_keypoints``, ``face_bbox`` / ``body_bbox``, ``draw``,
``distance_to``; ``PoseList`` = a list of fixed-size ``Pose`` records per row, pose_detection.py:3-158) for the element
layout the op writes (openpose_kernel.cpp:175-212): per person 1 pose score followed by (18 body + 70 face + 2 x 21 hand)
keypoints as (x, y, score) float32, coordinates in [0, 1] of the frame; a row without people is one float 0.
``draw`` needs no OpenCV here (segments are rasterised with numpy)."""
import math

import numpy as np

_JOINTS = ("Nose", "Neck", "RShoulder", "RElbow", "RWrist", "LShoulder", "LElbow", "LWrist", "RHip", "RKnee", "RAnkle",
           "LHip", "LKnee", "LAnkle", "REye", "LEye", "REar", "LEar", "Background")


class Pose:
    POSE_KEYPOINTS, POSE_SCORES, FACE_KEYPOINTS, HAND_KEYPOINTS = 18, 1, 70, 21
    # limbs drawn between body joints and their colours (OpenPose's COCO rendering table, as the reference lists it)
    DRAW_PAIRS = [[1, 2], [1, 5], [2, 3], [3, 4], [5, 6], [6, 7], [1, 8], [8, 9], [9, 10], [1, 11], [11, 12], [12, 13],
                  [1, 0], [0, 14], [14, 16], [0, 15], [15, 17]]
    DRAW_COLORS = [[255, 0, 85], [255, 0, 0], [255, 85, 0], [255, 170, 0], [255, 255, 0], [170, 255, 0], [85, 255, 0],
                   [0, 255, 0], [0, 255, 85], [0, 255, 170], [0, 255, 255], [0, 170, 255], [0, 85, 255], [0, 0, 255],
                   [255, 0, 170], [170, 0, 255], [255, 0, 255], [85, 0, 255]]

    def __init__(self, score, kp):
        self._score = float(score)
        self._kp = np.asarray(kp, np.float32).reshape(Pose.total_keypoints(), 3)

    @staticmethod
    def total_keypoints():
        return Pose.POSE_KEYPOINTS + Pose.FACE_KEYPOINTS + 2 * Pose.HAND_KEYPOINTS

    @staticmethod
    def kp_size():
        """floats per person record"""
        return Pose.total_keypoints() * 3 + Pose.POSE_SCORES

    @staticmethod
    def deserialize(buf):
        arr = np.frombuffer(buf, dtype="<f4", count=Pose.kp_size())
        return Pose(arr[0], arr[Pose.POSE_SCORES:])

    def serialize(self):
        return np.concatenate([[np.float32(self._score)], self._kp.reshape(-1)]).astype("<f4").tobytes()

    def score(self):
        return self._score

    def pose_keypoints(self):
        return self._kp[:Pose.POSE_KEYPOINTS]

    def face_keypoints(self):
        return self._kp[Pose.POSE_KEYPOINTS:Pose.POSE_KEYPOINTS + Pose.FACE_KEYPOINTS]

    def hand_keypoints(self):
        base = self._kp[Pose.POSE_KEYPOINTS + Pose.FACE_KEYPOINTS:]
        return [base[:Pose.HAND_KEYPOINTS], base[Pose.HAND_KEYPOINTS:]]

    def face_bbox(self, min_score=0.05):
        """[(xmin, ymin), (xmax, ymax), score] around eyes, ears and nose (a square-ish box one face-width tall)."""
        p = self.pose_keypoints()
        pts = np.array([p[i] for i in (Pose.REye, Pose.LEye, Pose.REar, Pose.LEar, Pose.Nose) if p[i, 2] > min_score], ndmin=2)
        if pts.size == 0:
            return [(0, 0), (0, 0), 0]
        xmin, xmax = pts[:, 0].min(), pts[:, 0].max()
        width = xmax - xmin
        yavg = pts[:, 1].mean()
        return [(xmin - 0.1 * width, yavg - width), (xmax + 0.1 * width, yavg + width), min(p[Pose.REar, 2], p[Pose.LEar, 2], p[Pose.Nose, 2])]

    def body_bbox(self):
        p = self.pose_keypoints()
        return [(p[:, 0].min(), p[:, 1].min()), (p[:, 0].max(), p[:, 1].max()), float(p[:, 2].mean())]

    def draw(self, img, thickness=5, draw_threshold=0.05):
        """Body limbs onto an (h, w, 3) uint8 image, in place; joints outside [0, 1) or below the threshold are skipped."""
        h, w = img.shape[:2]
        kp = self._kp

        def ok(i):
            return kp[i, 2] > draw_threshold and all(0 <= v < 1 for v in kp[i, :2])

        for (a, b), color in zip(Pose.DRAW_PAIRS, Pose.DRAW_COLORS):
            if not (ok(a) and ok(b)):
                continue
            x0, y0, x1, y1 = kp[a, 0] * w, kp[a, 1] * h, kp[b, 0] * w, kp[b, 1] * h
            steps = int(max(abs(x1 - x0), abs(y1 - y0))) + 1
            r = max(thickness // 2, 0)
            for t in np.linspace(0.0, 1.0, steps + 1):
                cx, cy = int(x0 + t * (x1 - x0)), int(y0 + t * (y1 - y0))
                img[max(cy - r, 0):min(cy + r + 1, h), max(cx - r, 0):min(cx + r + 1, w)] = color
        return img

    def distance_to(self, pose, confidence_threshold=0.2):
        """Median distance between the body joints both poses are confident about (inf when there is none)."""
        a, b = self.pose_keypoints(), pose.pose_keypoints()
        both = (a[:, 2] > confidence_threshold) & (b[:, 2] > confidence_threshold)
        if not both.any():
            return math.inf
        return float(np.median(np.hypot(a[both, 0] - b[both, 0], a[both, 1] - b[both, 1])))


for _i, _name in enumerate(_JOINTS):
    setattr(Pose, _name, _i)


def pose_list(buf):
    """Reader of the op's "PoseList" column: the row's bytes -> [Pose, ...] (a row of one float = nobody)."""
    size = Pose.kp_size() * 4
    if buf is None or len(buf) < size:
        return []
    return [Pose.deserialize(buf[i:i + size]) for i in range(0, len(buf) - size + 1, size)]
