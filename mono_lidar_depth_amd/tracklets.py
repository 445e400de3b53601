"""Python mirror of tracklets_depth::TrackletDepthModule around the DepthEstimator boundary.

Reference: tracklets_depth/src/tracklet_depth_module.cpp — process :261-396, ExractNewTrackletFrames :23-61,
CalculateFeatureDepthsCurFrame :63-82, CalculateFeatureDepthsLastFrame :84-117, SaveFeatureDepths :119-169,
TidyUpTracklets :171-193.  The ROS message plumbing is out of scope; tracks arrive as arrays.

What runs on the GPU: the feature gather (newest feature of every track, previous feature of NEW tracks, truncated
to integer pixels), both CalculateDepth calls and the float32 scatter — one C-ABI call, `mld_tracklets_depth*`.
The previous frame is NOT re-projected as the reference does (:115 -> setInputCloud): its frame slot (cloud, pixel
map, ground plane) is kept and the two slots ping-pong.  The tracklet map itself (ids seen so far, per-track
history) is host bookkeeping, kept here in Python.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import capi
from .depth_estimator import (NO_PLANE, CameraPinhole, DepthEstimator, ExceptionPclInvalid, GroundPlane, RansacPlane,
                              SemanticPlane, _is_torch_cuda)


class TrackletDepthModule:
    def __init__(self, parameters, camera: CameraPinhole, transform_lidar_to_cam, device: int = 0,
                 keep_history: bool = True):
        self._est = DepthEstimator(device=device, max_frames=2)
        self._est.InitConfig(parameters)
        self._est.Initialize(camera, transform_lidar_to_cam)
        self._slot_cur = 0
        self._have_last = False  # _cloud_last_frame != nullptr
        self._keep_history = keep_history
        self.one_call = True  # process() = one C call (mld_tracklets_frame); False: setInputCloud + mld_tracklets_depth
        # _trackletMap: id -> list of (u, v, depth), newest first (feature_tracking::Tracklet::push_front)
        self._tracklet_map: Dict[int, List[Tuple[int, int, float]]] = {}

    @property
    def estimator(self) -> DepthEstimator:
        return self._est

    def known_ids(self):
        return self._tracklet_map.keys()

    def process(self, cloud, ids, u_new, v_new, u_old, v_old, ground_plane: Optional[GroundPlane], img=None):
        """One frame (tracklet_depth_module.cpp:261-396).  With a semantic label image `img` (the 4-argument
        overload, :261-284) the ground plane is a SemanticPlane with labels {6,7,8,9} and the inlier threshold
        ransac_plane_refinement_treshold, and `ground_plane` is ignored.

        ids: track ids; u_new/v_new: newest feature of every track (feature_points[0]); u_old/v_old: previous
        feature (feature_points[1]; only read for tracks that are new to the module).  Returns
        (d_cur, d_last, is_new): float32 depths of the newest features, float32 depths of the previous features
        (NaN where the track is not new) and the new-track mask.
        """
        est, lib = self._est, self._est._lib
        ids = np.asarray(ids, dtype=np.int64)
        n = int(ids.size)
        known = self._tracklet_map
        is_new = np.fromiter((int(i) not in known for i in ids), dtype=np.uint8, count=n)  # :31
        slot_cur = self._slot_cur
        slot_last = (1 - slot_cur) if self._have_last else -1
        import time
        t_abi = time.perf_counter()
        if img is not None:
            ground_plane = SemanticPlane(img, (6, 7, 8, 9), est.getParameters().ransac_plane_refinement_treshold)
        arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (u_new, v_new, u_old, v_old)]
        d_cur = np.empty(n, dtype=np.float32)
        d_last = np.full(n, np.nan, dtype=np.float32)
        t_cur = np.empty(n, dtype=np.int32)
        t_last = np.zeros(n, dtype=np.int32)
        n_new = C.c_int64(0)
        if self.one_call and not _is_torch_cuda(cloud) and self._frame_call(cloud, ground_plane, slot_cur, slot_last, arrs,
                                                                            is_new, n, d_cur, d_last, t_cur, t_last, n_new):
            pass  # the whole GPU side of process() was ONE C call (mld_tracklets_frame)
        else:
            est.setInputCloud(cloud, ground_plane, slot=slot_cur)  # the only projection of this frame
            est._check(lib.mld_tracklets_depth(est._ctx, slot_cur, slot_last, *[a.ctypes.data for a in arrs],
                                               is_new.ctypes.data, n, d_cur.ctypes.data, d_last.ctypes.data,
                                               t_cur.ctypes.data, t_last.ctypes.data, C.byref(n_new)))
        self.last_types = (t_cur, t_last)
        self.last_abi_seconds = time.perf_counter() - t_abi  # C-ABI calls only (copies, kernels, synchronise)
        # SaveFeatureDepths (:119-169) + TidyUpTracklets (:171-193)
        updated = {}
        ui0, vi0 = arrs[0].astype(np.int32), arrs[1].astype(np.int32)
        ui1, vi1 = arrs[2].astype(np.int32), arrs[3].astype(np.int32)
        for i in range(n):
            tid = int(ids[i])
            hist = known.get(tid)
            if hist is None:
                hist = [(int(ui1[i]), int(vi1[i]), float(d_last[i]))]
            elif not self._keep_history:
                hist = []
            hist.insert(0, (int(ui0[i]), int(vi0[i]), float(d_cur[i])))
            updated[tid] = hist
        self._tracklet_map = updated  # tracks without an update this frame are dropped
        # remember cloud / plane: the current slot becomes the previous one (:348, :357)
        self._have_last = True
        self._slot_cur = 1 - slot_cur
        return d_cur, d_last, is_new.astype(bool)

    def _frame_call(self, cloud, gp, slot_cur, slot_last, arrs, is_new, n, d_cur, d_last, t_cur, t_last, n_new) -> bool:
        """The GPU side of process() as one C call: upload, ground plane (estimated inside the call when it is not
        segmented yet - what the reference does every frame, tracklet_depth_module.cpp:269-284), projection, feature
        marshalling, both CalculateDepth calls, scatter.  False: not applicable (device inputs)."""
        est, lib = self._est, self._est._lib
        road = bool(est.getParameters().do_use_ransac_plane)
        if road and gp is None:
            gp = RansacPlane()
        ptr, npts, stride, keep = est._cloud_view(cloud)
        req = None
        coeffs = inl = None
        n_inl = 0
        hold = None
        if road and gp is not NO_PLANE:
            if isinstance(gp, RansacPlane) and not gp.isSegmented():
                req = capi.MldPlaneRequest()
                req.kind, req.seed = capi.MLD_PLANE_RANSAC, gp.seed & 0xFFFFFFFF
                if isinstance(gp, SemanticPlane):
                    if _is_torch_cuda(gp.img):
                        return False
                    im = np.ascontiguousarray(gp.img, dtype=np.uint8)
                    lab = np.ascontiguousarray(gp.groundplane_label, dtype=np.int32)
                    hold = (im, lab)
                    req.kind = capi.MLD_PLANE_SEMANTIC
                    req.label_image, req.rows, req.cols, req.row_stride_bytes = im.ctypes.data, im.shape[0], im.shape[1], im.strides[0]
                    req.ground_labels, req.n_labels, req.inlier_threshold = lab.ctypes.data, lab.size, gp.inlier_threshold
            else:
                if not isinstance(gp, GroundPlane) or _is_torch_cuda(gp.inliers):
                    return False
                ii = gp.getInlinersIndex()
                if ii is None:
                    return False
                inl = np.ascontiguousarray(ii, dtype=np.int32)
                coeffs = (C.c_float * 4)(*[float(x) for x in gp.coeffs])
                n_inl = int(inl.size)
        res = capi.MldPlaneResult()
        rc = lib.mld_tracklets_frame(est._ctx, slot_cur, slot_last, ptr, npts, stride, C.byref(req) if req is not None else None,
                                     coeffs, inl.ctypes.data if inl is not None else None, n_inl,
                                     *[a.ctypes.data for a in arrs], is_new.ctypes.data, n, d_cur.ctypes.data,
                                     d_last.ctypes.data, t_cur.ctypes.data, t_last.ctypes.data, C.byref(n_new), C.byref(res))
        del hold, keep
        if rc == capi.MLD_ERR_CLOUD_TOO_SMALL:
            # the reference's process() catches ExceptionPclInvalid per frame (:318-347): the previous frame's features
            # are answered, the current frame continues with invalid depths, cloud and plane are forgotten
            raise ExceptionPclInvalid(rc, "In GroundPlane: Input pointcloud is invalid")
        est._check(rc)
        if req is not None:
            gp.coeffs = np.array(list(res.coeffs), dtype=np.float32)
            gp.inliers = None
            gp.n_inliers = int(res.n_inliers)
            gp.iterations = int(res.iterations)
            gp._segmented = True
            gp._owner = (est, slot_cur)
        return True

    def tracklet(self, track_id: int):
        return list(self._tracklet_map[track_id])


class TrackletBatch:
    """The tracklet layer for S independent sequences at once (mld_set_clouds_planes_range_device +
    mld_tracklets_depths_device): the estimator's frame slots are two banks of S slots, sequence s keeps its current
    frame in slot bank * S + s and its previous frame, still resident, in the other bank.  Device tensors in, device
    tensors out; asynchronous.  (Track bookkeeping per sequence is the caller's, as in TrackletDepthModule.)"""

    def __init__(self, parameters, camera: CameraPinhole, transform_lidar_to_cam, n_seq: int, max_tracks: int,
                 device: int = 0, list_capacity=None):
        self.S = int(n_seq)
        self.est = DepthEstimator(device=device, max_frames=2 * self.S, max_features=max_tracks)
        self.est.InitConfig(parameters)
        self.est.Initialize(camera, transform_lidar_to_cam)
        if list_capacity:
            self.est.setListCapacity(*list_capacity)
        self.bank = 0
        self.have_last = False
        self._keep = None

    def prepare(self, clouds, coeffs, masks, u_new, v_new, u_old, v_old, is_new, d_cur, d_last, t_cur=None, t_last=None):
        """Pointer tables of one frame of every sequence (reusable: a steady-state caller prepares one per bank).
        clouds / masks: S torch CUDA tensors ([N,4] float32 / int32 inlier bitmask), coeffs [S,4] float32 (host);
        u_new ... is_new: S CUDA tensors each (float32 x4, uint8); outputs: S CUDA float32 tensors (d_cur, d_last) and
        optional int32 (t_cur, t_last)."""
        S = self.S
        vp = lambda ts: (C.c_void_p * S)(*[int(t.data_ptr()) for t in ts])  # noqa: E731
        co = np.ascontiguousarray(coeffs, dtype=np.float32).reshape(S, 4)
        return {"clouds": vp(clouds), "n": (C.c_int64 * S)(*[int(c.shape[0]) for c in clouds]), "coeffs": co,
                "masks": vp(masks), "u_new": vp(u_new), "v_new": vp(v_new), "u_old": vp(u_old), "v_old": vp(v_old),
                "is_new": vp(is_new), "nt": (C.c_int64 * S)(*[int(t.shape[0]) for t in u_new]), "d_cur": vp(d_cur),
                "d_last": vp(d_last), "t_cur": vp(t_cur) if t_cur is not None else None,
                "t_last": vp(t_last) if t_last is not None else None,
                "keep": (list(clouds), list(masks), list(u_new), list(v_new), list(u_old), list(v_old), list(is_new),
                         list(d_cur), list(d_last), t_cur, t_last)}

    def run(self, f, nxt: Optional["TrackletBatch"] = None, handover: str = "projection"):
        """One frame of every sequence from prepared tables: projection of the current bank, then the tracklet call.
        `nxt`: another TrackletBatch (other sequences) that takes the next step - it is released as soon as this
        projection has finished, so that its projection runs beside these feature kernels (include/mld.h "Two contexts")."""
        S, est, lib = self.S, self.est, self.est._lib
        est._check(lib.mld_set_clouds_planes_range_device(est._ctx, self.bank * S, S, f["clouds"], f["n"], 16,
                                                          f["coeffs"].ctypes.data_as(C.POINTER(C.c_float)), f["masks"]))
        if nxt is not None and nxt is not self:
            # (released at the end of this projection, or behind this step's classification kernel: include/mld.h)
            if handover == "classify":
                nxt.est.orderAfterClassify(est)
            else:
                nxt.est.orderAfter(est)
        est._check(lib.mld_tracklets_depths_device(est._ctx, S, self.bank, 1 if self.have_last else 0, f["u_new"], f["v_new"],
                                                   f["u_old"], f["v_old"], f["is_new"], f["nt"], f["d_cur"], f["d_last"],
                                                   f["t_cur"], f["t_last"]))
        # the arrays must outlive the asynchronous launches; the previous frame's clouds stay referenced by their slots
        self._keep = (self._keep[1] if self._keep else None, f)
        self.bank = 1 - self.bank
        self.have_last = True

    def frame(self, clouds, *args, **kw):
        self.est._after_torch(clouds[0])
        self.run(self.prepare(clouds, *args, **kw))

    def close(self):
        self.est.close()
