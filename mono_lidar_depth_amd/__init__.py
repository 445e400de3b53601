"""MI355X-native DepthEstimator hot path (monolidar_fusion) behind the reference's interface.

Layout: csrc/ (HIP kernels + C-ABI, built to lib/libmld_hip.so), host/ (C++ shim class with the reference's
method names), depth_estimator.py (Python mirror used by tests and bench), synth.py (seeded synthetic frames),
sharding.py (sequence -> GPU assignment and the calibration broadcast).
"""
from .capi import MldCamera, MldParams, params_c0, params_default, params_from_file, RESULT_TYPE_NAMES
from .depth_estimator import (NO_PLANE, CameraPinhole, DepthEstimator, DepthEstimatorError, ExceptionPclInvalid,
                              GroundPlane, RansacPlane, SemanticPlane)

from .tracklets import TrackletBatch, TrackletDepthModule

__all__ = ["TrackletDepthModule", "MldCamera", "MldParams", "params_c0", "params_default", "params_from_file", "RESULT_TYPE_NAMES",
           "NO_PLANE", "RansacPlane", "SemanticPlane", "CameraPinhole", "DepthEstimator", "DepthEstimatorError", "ExceptionPclInvalid", "GroundPlane"]
