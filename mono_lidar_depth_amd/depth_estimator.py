"""Python mirror of the reference's DepthEstimator interface over the C-ABI.

Same method names, argument meaning and error behaviour as `Mono_Lidar::DepthEstimator`
(monolidar_fusion/include/monolidar_fusion/DepthEstimator.h:39-359) so that the parity tests read like
the reference's own call sites (tracklets_depth/src/tracklet_depth_module.cpp:80,115,401,413).  numpy arrays
take the host-pointer entry points; torch CUDA tensors take the zero-copy device entry points.  All compute
runs in libmld_hip.so on the GPU; nothing here falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import capi
from .capi import MldCamera, MldParams


class DepthEstimatorError(RuntimeError):
    """Counterpart of the reference's `throw "..."` usage errors; `.code` holds the mld_status."""

    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


class ExceptionPclInvalid(DepthEstimatorError):
    """GroundPlane::ExceptionPclInvalid (RansacPlane.h:46-50)."""


class CameraPinhole:
    """CameraPinhole(width, height, f, cu, cv) — camera_pinhole.h:29-34."""

    def __init__(self, width: int, height: int, focal_length: float, principal_point_x: float,
                 principal_point_y: float):
        self.width, self.height = int(width), int(height)
        self.focal_length = float(focal_length)
        self.principal_point_x = float(principal_point_x)
        self.principal_point_y = float(principal_point_y)

    def getImageSize(self):
        return self.width, self.height

    def as_struct(self) -> MldCamera:
        return MldCamera(self.focal_length, self.principal_point_x, self.principal_point_y, self.width, self.height)


class GroundPlane:
    """The GroundPlane object the path consumes (RansacPlane.h:38-123): coefficients + inlier index set.

    In this build the plane is an INPUT: `coeffs` a,b,c,d in the lidar frame, `inliers` the
    original-cloud indices that `CheckPointInPlane` would accept.
    """

    def __init__(self, coeffs: Sequence[float], inliers):
        self.coeffs = np.asarray(coeffs, dtype=np.float32).reshape(4)
        self.inliers = inliers  # numpy int32 array or torch cuda int32 tensor
        self._segmented = True

    def isSegmented(self) -> bool:
        return self._segmented

    def getModelCoeffs(self):
        return self.coeffs

    def getInlinersIndex(self):
        return self.inliers


class RansacPlane(GroundPlane):
    """RansacPlane(parameters) (RansacPlane.h:129-170): not segmented until setInputCloud estimates it on the GPU
    (DepthEstimator.cpp:275-283 -> RansacPlane::CalculateInliersPlane, RansacPlane.cpp:41-140).  `seed` fixes the
    random draws (the reference's pcl::RandomSample is time-seeded)."""

    def __init__(self, seed: int = 0):
        self.coeffs = None
        self.inliers = None
        self.seed = int(seed)
        self.n_inliers = None   # inlier count / RANSAC iterations reported by the one-call frame path
        self.iterations = None
        self._segmented = False
        self._owner = None

    def getInlinersIndex(self):
        if self.inliers is None and self._owner is not None:
            est, slot = self._owner
            self.inliers = est.getGroundPlaneInliers(slot)
        return self.inliers


class SemanticPlane(RansacPlane):
    """SemanticPlane(img, cam, groundplane_label, inlier_threshold) (RansacPlane.h:175-218): ground plane from a label
    image; estimated on the GPU by setInputCloud while not segmented (SemanticPlane::CalculateInliersPlane,
    RansacPlane.cpp:195-274).  The camera is the estimator's own calibration."""

    def __init__(self, img, groundplane_label=(6, 7, 8, 9), inlier_threshold: float = 0.1):
        super().__init__(0)
        self.img = img
        self.groundplane_label = tuple(int(x) for x in groundplane_label)
        self.inlier_threshold = float(inlier_threshold)


class _NoPlane:
    """`ransacPlane == nullptr` of the feature-only CalculateDepth overloads: the road fallback is skipped
    (DepthEstimator.cpp:580)."""

    def __repr__(self):
        return "NO_PLANE"


NO_PLANE = _NoPlane()


def _is_torch_cuda(x) -> bool:
    return hasattr(x, "is_cuda") and bool(x.is_cuda)


class DepthEstimator:
    """Drop-in mirror of Mono_Lidar::DepthEstimator for one GPU stream.

    `max_frames` > 1 exposes the frame slots of the C-ABI: `setInputClouds` / `CalculateDepths` process many
    independent frames per launch (frames of a sequence or a micro-batch of sequences).
    """

    def __init__(self, device: int = 0, max_frames: int = 1, max_points: int = 0, max_features: int = 0):
        self._lib = capi.load()
        self._device = int(device)
        self._max_frames = int(max_frames)
        self._max_points = int(max_points)
        self._max_features = int(max_features)
        self._parameters: Optional[MldParams] = None
        self._camera: Optional[CameraPinhole] = None
        self._T = None
        self._ctx = None
        self._isInitializedConfig = False
        self._isInitialized = False
        self._keepalive = {}
        self._debug = False
        self._debug_last = None
        self._last_types = None

    # ------------------------------------------------------------------ lifecycle
    def InitConfig(self, parameters=None, printparams: bool = False) -> bool:
        """InitConfig(path | parameters) — DepthEstimator.cpp:129-154."""
        if parameters is None:
            self._parameters = capi.params_default()
        elif isinstance(parameters, (str, bytes)) or hasattr(parameters, "__fspath__"):
            self._parameters = capi.params_from_file(str(parameters))
        else:
            self._parameters = parameters.copy()
        if printparams:
            for name, _ in MldParams._fields_:
                print(f"{name}: {getattr(self._parameters, name)}")
        self._isInitializedConfig = True
        return True

    def Initialize(self, camera: CameraPinhole, transform_lidar_to_cam) -> bool:
        """Initialize(camera, T_cam_lidar) — DepthEstimator.cpp:35-127.  T: 4x4 or 3x4 (lidar -> camera)."""
        if not self._isInitializedConfig:
            raise DepthEstimatorError(capi.MLD_ERR_NOT_INITIALIZED, "Call 'InitConfig' before calling 'Initialize'.")
        T = np.asarray(transform_lidar_to_cam, dtype=np.float64)
        if T.shape == (4, 4):
            T = T[:3, :]
        if T.shape != (3, 4):
            raise DepthEstimatorError(capi.MLD_ERR_INVALID_ARG, "transform must be 3x4 or 4x4")
        self._T = np.ascontiguousarray(T)
        self._camera = camera
        if self._ctx is not None:
            self._lib.mld_destroy(self._ctx)
            self._ctx = None
        status = C.c_int(0)
        cam = camera.as_struct()
        ctx = self._lib.mld_create(C.byref(self._parameters), C.byref(cam),
                                   self._T.ctypes.data_as(C.POINTER(C.c_double)), self._device, self._max_frames,
                                   self._max_points, self._max_features, C.byref(status))
        if not ctx:
            raise DepthEstimatorError(status.value, self._lib.mld_create_error().decode())
        self._ctx = C.c_void_p(ctx)
        self._isInitialized = True
        return True

    def close(self):
        if self._ctx is not None:
            self._lib.mld_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != capi.MLD_OK:
            msg = self._lib.mld_last_error(self._ctx).decode() if self._ctx else "no context"
            if rc == capi.MLD_ERR_CLOUD_TOO_SMALL:
                raise ExceptionPclInvalid(rc, "In GroundPlane: Input pointcloud is invalid")
            raise DepthEstimatorError(rc, msg)

    def _require_init(self, what: str):
        if not self._isInitialized:
            raise DepthEstimatorError(capi.MLD_ERR_NOT_INITIALIZED, f"call of '{what}' without 'initialize'")

    # ------------------------------------------------------------------ getters (DepthEstimator.h:98-122)
    def getParameters(self) -> MldParams:
        return self._parameters

    def getCamera(self) -> CameraPinhole:
        return self._camera

    def getTransformLidarToCam(self):
        return self._T.copy()

    @property
    def stream(self) -> int:
        return int(self._lib.mld_get_stream(self._ctx) or 0)

    @staticmethod
    def _after_torch(*tensors):
        """The context launches on its own non-blocking HIP stream.  Device tensors handed in may still be being
        written by work queued on torch's current stream (a kernel, a non_blocking copy): wait for that stream before
        launching on ours.  (Idle stream: a few microseconds.)"""
        import torch
        for t in tensors:
            if t is not None and _is_torch_cuda(t):
                torch.cuda.current_stream(t.device).synchronize()
                return

    def orderTorchAfter(self):
        """Makes torch's current stream wait for everything queued on the context's stream so far: outputs of the
        asynchronous batch calls (runBatch, CalculateDepths) may then be consumed by torch operations without a host
        synchronisation."""
        import torch
        dev = torch.device("cuda", self._device)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.ExternalStream(self.stream, device=dev))
        torch.cuda.current_stream(dev).wait_event(ev)

    def synchronize(self):
        self._check(self._lib.mld_synchronize(self._ctx))

    def orderAfter(self, other: "DepthEstimator"):
        """Everything submitted to this context from now on starts after everything submitted to `other` so far has
        finished (device-side; mld_order_after)."""
        self._check(self._lib.mld_order_after(self._ctx, other._ctx))

    def orderAfterClassify(self, other: "DepthEstimator"):
        """The same hand-over, released behind the classification kernel of `other`'s next batched CalculateDepth call
        (mld_order_after_classify): that kernel then has the GPU to itself, this context's projection still runs beside
        `other`'s feature kernels."""
        self._check(self._lib.mld_order_after_classify(self._ctx, other._ctx))

    def setListCapacity(self, wide: int, narrow: int):
        """Neighbour-list capacities of the lane-per-feature kernel (mld_set_list_capacity): 32 / 24 by default, 48 / 24
        for dense (128-beam) clouds."""
        self._check(self._lib.mld_set_list_capacity(self._ctx, int(wide), int(narrow)))

    def concurrentWith(self, other: "DepthEstimator") -> bool:
        """Whether this context's stream and `other`'s run side by side (distinct hardware queues of the HIP runtime;
        mld_contexts_concurrent).  mld_create already looks for a queue of its own; this repeats the probe."""
        r = self._lib.mld_contexts_concurrent(self._ctx, other._ctx)
        if r < 0:
            self._check(r)
        return r == 1

    def pathCounts(self, slot: int = 0):
        """(features queued for the lane-per-feature kernel, features the wave-cooperative kernel worked on) in the
        slot's last batched CalculateDepth call (mld_get_path_counts)."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.mld_get_path_counts(self._ctx, int(slot), C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def setListBudget(self, total: int):
        """LDS entries per feature shared by the wide and the narrow list (mld_set_list_budget): 40 for a new context,
        wide + narrow after setListCapacity; 0 = wide + narrow."""
        self._check(self._lib.mld_set_list_budget(self._ctx, int(total)))

    def pairWith(self, other: "DepthEstimator"):
        """The batched projections of this context and `other` run back to back on one stream (shared, reference
        counted: either context may be closed first), each context's feature kernels on its own (mld_pair_contexts)."""
        self._check(self._lib.mld_pair_contexts(self._ctx, other._ctx))
        self._pair, other._pair = other, self

    def setSharedGpu(self, shared: bool = True):
        """This context alternates with another one on the same GPU (see `runBatchesAlternating`): its feature kernel
        leaves room on every CU for the other context's projection (mld_set_shared_gpu)."""
        self._check(self._lib.mld_set_shared_gpu(self._ctx, int(shared)))  # bit 0: on; bits 8..15: feature wavefronts per CU

    # ------------------------------------------------------------------ setInputCloud
    @staticmethod
    def _cloud_view(cloud):
        """(pointer, n, stride, keepalive) of an [N,4] (packed xyzi) or [N,8] (pcl::PointXYZI) float32 cloud."""
        if _is_torch_cuda(cloud):
            import torch
            if cloud.dtype != torch.float32 or cloud.dim() != 2 or cloud.shape[1] not in (4, 8) or not cloud.is_contiguous():
                raise DepthEstimatorError(capi.MLD_ERR_INVALID_ARG, "cloud must be contiguous float32 [N,4] or [N,8]")
            return cloud.data_ptr(), int(cloud.shape[0]), int(cloud.shape[1]) * 4, cloud
        arr = np.ascontiguousarray(cloud, dtype=np.float32)
        if arr.ndim != 2 or arr.shape[1] not in (4, 8):
            raise DepthEstimatorError(capi.MLD_ERR_INVALID_ARG, "cloud must be float32 [N,4] or [N,8]")
        return arr.ctypes.data, int(arr.shape[0]), int(arr.shape[1]) * 4, arr

    def setInputCloud(self, cloud, groundPlane: Optional[GroundPlane] = None, slot: int = 0,
                      plane_given: bool = True):
        """setInputCloud(cloud, groundPlane) — DepthEstimator.cpp:220-312.

        groundPlane: a segmented GroundPlane (consumed as is), a RansacPlane (estimated here if not yet
        segmented), None (a RansacPlane is created and estimated, as the reference does for a null pointer,
        :275-278) or NO_PLANE (no plane at all: the road fallback is skipped, the `ransacPlane == nullptr` case of
        the feature-only overloads).  Without do_use_ransac_plane the argument is ignored.
        """
        self._require_init("setInputCloud")
        ptr, n, stride, keep = self._cloud_view(cloud)
        self._keepalive[("cloud", slot)] = keep
        if _is_torch_cuda(cloud):
            self._after_torch(cloud)
            self._check(self._lib.mld_set_cloud_device(self._ctx, slot, ptr, n, stride))
        else:
            self._check(self._lib.mld_set_cloud(self._ctx, slot, ptr, n, stride))
        if plane_given:
            self.setGroundPlane(groundPlane, slot)

    def setGroundPlane(self, groundPlane, slot: int = 0):
        """The ground-plane hook of setInputCloud (DepthEstimator.cpp:273-292)."""
        if groundPlane is NO_PLANE or not self._parameters.do_use_ransac_plane:
            self._check(self._lib.mld_set_ground_plane(self._ctx, slot, None, None, 0))
            return
        if groundPlane is None:
            groundPlane = RansacPlane()
        if isinstance(groundPlane, RansacPlane) and not groundPlane.isSegmented():
            if isinstance(groundPlane, SemanticPlane):
                coeffs, _ = self.estimateSemanticPlane(groundPlane.img, groundPlane.groundplane_label,
                                                       groundPlane.inlier_threshold, slot)
            else:
                coeffs, _ = self.estimateGroundPlane(slot, groundPlane.seed)
            groundPlane.coeffs = coeffs
            groundPlane.inliers = None
            groundPlane._segmented = True
            groundPlane._owner = (self, slot)
            return
        coeffs = (C.c_float * 4)(*[float(x) for x in groundPlane.coeffs])
        inl = groundPlane.inliers
        if _is_torch_cuda(inl):
            import torch
            if inl.dtype != torch.int32 or not inl.is_contiguous():
                raise DepthEstimatorError(capi.MLD_ERR_INVALID_ARG, "inliers must be contiguous int32")
            self._keepalive[("inl", slot)] = inl
            self._after_torch(inl)
            self._check(self._lib.mld_set_ground_plane_device(self._ctx, slot, coeffs, inl.data_ptr(), int(inl.numel())))
        else:
            arr = np.ascontiguousarray(inl, dtype=np.int32)
            self._check(self._lib.mld_set_ground_plane(self._ctx, slot, coeffs, arr.ctypes.data, int(arr.size)))

    def setGroundPlanesMask(self, coeffs, masks: Sequence):
        """Batched ground-plane hook: `coeffs` [B,4] float32 (host), `masks` B torch CUDA int32 bitmask tensors
        (bit i of word i/32 set = original point i is a plane inlier; the device form of `_pointIsInPlane`)."""
        n_slots = len(masks)
        c = np.ascontiguousarray(coeffs, dtype=np.float32).reshape(n_slots, 4)
        ptrs = (C.c_void_p * n_slots)(*[int(m.data_ptr()) for m in masks])
        self._keepalive["masks"] = list(masks)
        self._after_torch(*masks[:1])
        self._check(self._lib.mld_set_ground_planes_mask_device(
            self._ctx, n_slots, c.ctypes.data_as(C.POINTER(C.c_float)), ptrs))

    def prepareBatch(self, clouds: Sequence, uvs: Sequence, depths: Sequence, types: Sequence, coeffs, masks: Sequence,
                     stride_bytes: int = 16):
        """Pre-marshal the pointer tables of a fixed batch so that `runBatch` is three C calls per step."""
        n = len(clouds)
        b = {"n": n, "stride": stride_bytes}
        b["cloud_ptrs"] = (C.c_void_p * n)(*[int(c.data_ptr()) for c in clouds])
        b["cloud_n"] = (C.c_int64 * n)(*[int(c.shape[0]) for c in clouds])
        b["uv_ptrs"] = (C.c_void_p * n)(*[int(self._uv_device(u).data_ptr()) for u in uvs])
        b["F"] = (C.c_int64 * n)(*[int(u.shape[0]) for u in uvs])
        b["depth_ptrs"] = (C.c_void_p * n)(*[int(d.data_ptr()) for d in depths])
        b["type_ptrs"] = (C.c_void_p * n)(*[int(t.data_ptr()) for t in types])
        b["coeffs"] = np.ascontiguousarray(coeffs, dtype=np.float32).reshape(n, 4)
        b["mask_ptrs"] = (C.c_void_p * n)(*[int(m.data_ptr()) for m in masks])
        b["keep"] = (list(clouds), list(uvs), list(depths), list(types), list(masks))
        return b

    def runBatch(self, b):
        """One pass of the hot path over the batch: setInputCloud + ground-plane hook + CalculateDepth for every
        slot (asynchronous on `stream`; inputs still being produced on torch's current stream are waited for, use
        `synchronize()` or `orderTorchAfter()` before consuming the outputs)."""
        lib, ctx, n = self._lib, self._ctx, b["n"]
        self._after_torch(b["keep"][0][0])
        self._check(lib.mld_set_clouds_planes_device(ctx, n, b["cloud_ptrs"], b["cloud_n"], b["stride"],
                                                     b["coeffs"].ctypes.data_as(C.POINTER(C.c_float)), b["mask_ptrs"]))
        self._check(lib.mld_calculate_depths_device(ctx, n, b["uv_ptrs"], b["F"], b["depth_ptrs"], b["type_ptrs"]))

    def projectBatch(self, b):
        """setInputCloud(cloud, plane) of a prepared batch: the projection launch only."""
        lib, ctx, n = self._lib, self._ctx, b["n"]
        self._after_torch(b["keep"][0][0])
        self._check(lib.mld_set_clouds_planes_device(ctx, n, b["cloud_ptrs"], b["cloud_n"], b["stride"],
                                                     b["coeffs"].ctypes.data_as(C.POINTER(C.c_float)), b["mask_ptrs"]))

    def featuresBatchBeside(self, b, nxt: "DepthEstimator", handover: str = "classify"):
        """CalculateDepth of a prepared batch whose projection is already queued; the NEXT context's projection is
        released behind this batch's classification kernel (`handover="classify"`, mld_order_after_classify) or at once
        (`"projection"`, mld_order_after) - unless `nxt` is this context's pair partner: their projections are in call
        order on the shared stream; any other context - three or more in rotation - still needs the hand-over."""
        if nxt is not self and getattr(self, "_pair", None) is not nxt:
            if handover == "classify":
                nxt.orderAfterClassify(self)
            else:
                nxt.orderAfter(self)
        self._check(self._lib.mld_calculate_depths_device(self._ctx, b["n"], b["uv_ptrs"], b["F"], b["depth_ptrs"],
                                                          b["type_ptrs"]))

    def runBatchBeside(self, b, nxt: "DepthEstimator", handover: str = "classify"):
        """`runBatch` for two (or more) contexts used in turn: context `nxt`, which will take the next batch, is released
        as soon as THIS batch's projection has finished, so that its projection runs beside this batch's feature kernels
        (HBM-bound work beside gather-bound work: include/mld.h "Two contexts").  Call `setSharedGpu()` on both contexts
        once."""
        self.projectBatch(b)
        self.featuresBatchBeside(b, nxt, handover)

    def estimateGroundPlane(self, slot: int = 0, seed: int = 0):
        """RansacPlane::CalculateInliersPlane on the GPU for the slot's cloud; installs the plane.
        Returns (coefficients float32[4], inlier count)."""
        coeffs = (C.c_float * 4)()
        n_inl = C.c_int64(0)
        self._check(self._lib.mld_estimate_ground_plane(self._ctx, slot, int(seed) & 0xFFFFFFFF, coeffs, C.byref(n_inl)))
        return np.array(list(coeffs), dtype=np.float32), int(n_inl.value)

    def estimateSemanticPlane(self, label_image, groundplane_label=(6, 7, 8, 9), inlier_threshold: float = 0.1,
                              slot: int = 0):
        """SemanticPlane::CalculateInliersPlane for the slot's cloud (RansacPlane.cpp:195-274).  `label_image`:
        rows x cols uint8 (numpy, or a torch CUDA tensor for the device entry point).
        Returns (coefficients float32[4], inlier count)."""
        lab = np.ascontiguousarray(groundplane_label, dtype=np.int32)
        coeffs = (C.c_float * 4)()
        n_inl = C.c_int64(0)
        if _is_torch_cuda(label_image):
            import torch
            if label_image.dtype != torch.uint8 or label_image.dim() != 2 or label_image.stride(1) != 1:
                raise DepthEstimatorError(capi.MLD_ERR_INVALID_ARG, "label image must be a 2-D uint8 tensor")
            torch.cuda.current_stream(label_image.device).synchronize()
            self._check(self._lib.mld_estimate_semantic_plane_device(
                self._ctx, slot, label_image.data_ptr(), label_image.shape[0], label_image.shape[1],
                label_image.stride(0), lab.ctypes.data, lab.size, float(inlier_threshold), coeffs, C.byref(n_inl)))
        else:
            img = np.ascontiguousarray(label_image, dtype=np.uint8)
            if img.ndim != 2:
                raise DepthEstimatorError(capi.MLD_ERR_INVALID_ARG, "label image must be 2-D")
            self._check(self._lib.mld_estimate_semantic_plane(
                self._ctx, slot, img.ctypes.data, img.shape[0], img.shape[1], img.strides[0], lab.ctypes.data, lab.size,
                float(inlier_threshold), coeffs, C.byref(n_inl)))
        return np.array(list(coeffs), dtype=np.float32), int(n_inl.value)

    def getGroundPlaneInliers(self, slot: int = 0) -> np.ndarray:
        """GroundPlane::getInlinersIndex for the slot's current plane (ascending original indices)."""
        n = C.c_int64(0)
        self._check(self._lib.mld_get_ground_plane_inliers(self._ctx, slot, None, 0, C.byref(n)))
        out = np.empty(int(n.value), dtype=np.int32)
        self._check(self._lib.mld_get_ground_plane_inliers(self._ctx, slot, out.ctypes.data, out.size, C.byref(n)))
        return out

    def setInputClouds(self, clouds: Sequence, stride_bytes: int = 16):
        """Batched setInputCloud: torch CUDA clouds for slots 0..len-1 in one launch."""
        self._require_init("setInputCloud")
        n_slots = len(clouds)
        ptrs = (C.c_void_p * n_slots)()
        counts = (C.c_int64 * n_slots)()
        for i, cl in enumerate(clouds):
            p, n, stride, keep = self._cloud_view(cl)
            if stride != stride_bytes or not _is_torch_cuda(cl):
                raise DepthEstimatorError(capi.MLD_ERR_INVALID_ARG, "setInputClouds needs CUDA clouds of one stride")
            ptrs[i], counts[i] = p, n
            self._keepalive[("cloud", i)] = keep
        self._after_torch(*clouds[:1])
        self._check(self._lib.mld_set_clouds_device(self._ctx, n_slots, ptrs, counts, stride_bytes))

    def setInputCloudsEstimatePlanes(self, clouds: Sequence, seeds: Sequence[int], stride_bytes: int = 16):
        """Batched setInputCloud(cloud, RansacPlane()) — the reference's default call with a plane that is not segmented
        yet (DepthEstimator.cpp:275-283): the ground plane of every slot is estimated on the GPU ahead of the projection,
        asynchronously; nothing returns to the host (mld_set_clouds_estimate_planes_device)."""
        self._require_init("setInputCloud")
        n_slots = len(clouds)
        ptrs = (C.c_void_p * n_slots)()
        counts = (C.c_int64 * n_slots)()
        for i, cl in enumerate(clouds):
            p, n, stride, keep = self._cloud_view(cl)
            if stride != stride_bytes or not _is_torch_cuda(cl):
                raise DepthEstimatorError(capi.MLD_ERR_INVALID_ARG, "needs CUDA clouds of one stride")
            ptrs[i], counts[i] = p, n
            self._keepalive[("cloud", i)] = keep
        sd = (C.c_uint32 * n_slots)(*[int(x) & 0xFFFFFFFF for x in seeds])
        self._after_torch(*clouds[:1])
        self._check(self._lib.mld_set_clouds_estimate_planes_device(self._ctx, n_slots, ptrs, counts, stride_bytes, sd))

    def getEstimatedPlanes(self, n_slots: int):
        """(coefficients [n,4] float32, inlier counts, status 0 ok / 1 failed) of the last batched estimation."""
        co = np.empty((n_slots, 4), dtype=np.float32)
        ni = np.empty(n_slots, dtype=np.int64)
        st = np.empty(n_slots, dtype=np.int32)
        self._check(self._lib.mld_get_estimated_planes(self._ctx, n_slots, co.ctypes.data_as(C.POINTER(C.c_float)),
                                                       ni.ctypes.data_as(C.POINTER(C.c_int64)),
                                                       st.ctypes.data_as(C.POINTER(C.c_int32))))
        return co, ni, st

    # ------------------------------------------------------------------ CalculateDepth
    def CalculateDepth(self, *args, slot: int = 0, return_types: bool = True, uv_layout: Optional[str] = None):
        """The reference's overloads (DepthEstimator.cpp:404-488):

        CalculateDepth(cloud, uv, groundPlane)   -> setInputCloud + per-feature loop
        CalculateDepth(uv)                       -> per-feature loop on the current cloud
        `uv` is 2 x F (Eigen::Matrix2Xd layout, column i = (u,v)) or F x 2; a 2 x 2 array is ambiguous and needs
        uv_layout="2xF" (the reference's layout) or "Fx2".  Returns (depths, resultTypes) — the callee-resized
        VectorXd / VectorXi of the reference — as numpy arrays (host input) or torch CUDA tensors (device input;
        asynchronous inputs on torch's current stream are waited for, the results are complete on return).
        """
        if len(args) == 3:
            cloud, uv, gp = args
            fast = self._frame_call(cloud, uv, gp, slot, uv_layout)
            if fast is not None:
                return fast if return_types else fast[0]
            self.setInputCloud(cloud, gp, slot=slot)
        elif len(args) == 1:
            (uv,) = args
        else:
            raise TypeError("CalculateDepth(cloud, uv, groundPlane) or CalculateDepth(uv)")
        self._require_init("CalculateDepth")
        if _is_torch_cuda(uv):
            import torch
            uvd = self._uv_device(uv, uv_layout)
            F = int(uvd.numel() // 2)
            depth = torch.empty(F, dtype=torch.float64, device=uvd.device)
            types = torch.empty(F, dtype=torch.int32, device=uvd.device)
            torch.cuda.current_stream(uvd.device).synchronize()
            self._check(self._lib.mld_calculate_depth_device(self._ctx, slot, uvd.data_ptr(), F, depth.data_ptr(),
                                                             types.data_ptr()))
            self.synchronize()
            return (depth, types) if return_types else depth
        uvh = self._uv_host(uv, uv_layout)
        F = int(uvh.size // 2)
        depth = np.empty(F, dtype=np.float64)
        types = np.empty(F, dtype=np.int32)
        if self._debug:
            corners = np.empty((F, 9), dtype=np.float64)
            self._check(self._lib.mld_calculate_depth_debug(self._ctx, slot, uvh.ctypes.data, F, depth.ctypes.data,
                                                            types.ctypes.data, corners.ctypes.data))
            self._debug_last = (uvh.reshape(F, 2).copy(), depth.copy(), types.copy(), corners)
        else:
            self._check(self._lib.mld_calculate_depth(self._ctx, slot, uvh.ctypes.data, F, depth.ctypes.data,
                                                      types.ctypes.data))
        self._last_types = types
        return (depth, types) if return_types else depth

    def _frame_call(self, cloud, uv, gp, slot, uv_layout=None):
        """CalculateDepth(cloud, uv, groundPlane) entirely from host memory as ONE C call (one frame per call, the ROS
        usage): mld_calculate_depth_frame for a plane that needs no estimation, mld_calculate_depth_frame_estimate for a
        RansacPlane / SemanticPlane that is not segmented yet - the reference's production call
        (tracklet_depth_module.cpp:269-284): the plane is estimated on the GPU ahead of the projection, nothing returns
        to the host before the depths do, and the plane object's inlier list is fetched only when asked for.
        None: not applicable (debug mode, device inputs)."""
        self._require_init("CalculateDepth")
        if self._debug or _is_torch_cuda(cloud) or _is_torch_cuda(uv):
            return None
        road = bool(self._parameters.do_use_ransac_plane)
        estimate = False
        if road and gp is None:
            gp = RansacPlane()  # the reference creates one for a null pointer (DepthEstimator.cpp:275-278)
        if road and gp is not NO_PLANE:
            if isinstance(gp, RansacPlane) and not gp.isSegmented():
                estimate = True
                if isinstance(gp, SemanticPlane) and _is_torch_cuda(gp.img):
                    return None
            elif not isinstance(gp, GroundPlane) or not gp.isSegmented() or _is_torch_cuda(gp.inliers):
                return None
            elif isinstance(gp, RansacPlane) and gp.inliers is None:
                gp.getInlinersIndex()  # (estimated earlier, list not fetched yet)
                if gp.inliers is None:
                    return None
        ptr, n, stride, keep = self._cloud_view(cloud)
        uvh = self._uv_host(uv, uv_layout)
        F = int(uvh.size // 2)
        depth = np.empty(F, dtype=np.float64)
        types = np.empty(F, dtype=np.int32)
        if estimate:
            req = capi.MldPlaneRequest()
            req.kind, req.seed = capi.MLD_PLANE_RANSAC, gp.seed & 0xFFFFFFFF
            hold = None
            if isinstance(gp, SemanticPlane):
                img = np.ascontiguousarray(gp.img, dtype=np.uint8)
                if img.ndim != 2:
                    raise DepthEstimatorError(capi.MLD_ERR_INVALID_ARG, "label image must be 2-D")
                lab = np.ascontiguousarray(gp.groundplane_label, dtype=np.int32)
                hold = (img, lab)
                req.kind = capi.MLD_PLANE_SEMANTIC
                req.label_image, req.rows, req.cols, req.row_stride_bytes = img.ctypes.data, img.shape[0], img.shape[1], img.strides[0]
                req.ground_labels, req.n_labels = lab.ctypes.data, lab.size
                req.inlier_threshold = gp.inlier_threshold
            res = capi.MldPlaneResult()
            self._check(self._lib.mld_calculate_depth_frame_estimate(self._ctx, slot, ptr, n, stride, C.byref(req),
                                                                     uvh.ctypes.data, F, depth.ctypes.data,
                                                                     types.ctypes.data, C.byref(res)))
            del hold
            gp.coeffs = np.array(list(res.coeffs), dtype=np.float32)
            gp.inliers = None  # fetched on demand (getInlinersIndex)
            gp.n_inliers = int(res.n_inliers)
            gp.iterations = int(res.iterations)
            gp._segmented = True
            gp._owner = (self, slot)
            self._last_types = types
            return depth, types
        if road and gp is not NO_PLANE:
            coeffs = (C.c_float * 4)(*[float(x) for x in gp.coeffs])
            inl = np.ascontiguousarray(gp.inliers, dtype=np.int32)
            self._check(self._lib.mld_calculate_depth_frame(self._ctx, slot, ptr, n, stride, coeffs, inl.ctypes.data,
                                                            int(inl.size), uvh.ctypes.data, F, depth.ctypes.data,
                                                            types.ctypes.data))
        else:
            self._check(self._lib.mld_calculate_depth_frame(self._ctx, slot, ptr, n, stride, None, None, 0,
                                                            uvh.ctypes.data, F, depth.ctypes.data, types.ctypes.data))
        self._last_types = types
        return depth, types

    # ------------------------------------------------------------------ debug mode
    def ActivateDebugMode(self):
        """ActivateDebugMode (DepthEstimator.h:85-87): host-input CalculateDepth calls also record the debug
        vectors behind getCloudTriangleCorners / getCloudInterpolated."""
        self._debug = True

    def getTriangleCorners(self) -> np.ndarray:
        """F x 9 corners (corner1 xyz, corner2 xyz, corner3 xyz; NaN = no triangle) of the last debug-mode call."""
        if self._debug_last is None:
            return np.empty((0, 9))
        return self._debug_last[3]

    def getCloudTriangleCorners(self) -> np.ndarray:
        """getCloudTriangleCorners (DepthEstimator.cpp:347-349): 3 x 3m, the corners CalculatePlaneCorners
        published (PlaneEstimationCalcMaxSpanningTriangle.cpp:27-32), in feature order."""
        c = self.getTriangleCorners()
        c = c[np.isfinite(c[:, 0])] if c.size else c
        return c.reshape(-1, 3).T.copy()

    def getCloudInterpolated(self) -> np.ndarray:
        """getCloudInterpolated (DepthEstimator.cpp:339-341): 3 x m intersection points ray ∩ plane of the features
        with a valid depth.  (The reference's push at DepthEstimator.cpp:1032 is commented out, so its cloud is
        always empty; debug mode here returns what the member is documented to hold, DepthEstimator.h:328.)"""
        if self._debug_last is None:
            return np.empty((3, 0))
        uv, depth, types, _ = self._debug_last
        ok = depth >= 0
        f, cu, cv = self._camera.focal_length, self._camera.principal_point_x, self._camera.principal_point_y
        ray = np.stack([(uv[ok, 0] - cu) / f, (uv[ok, 1] - cv) / f, np.ones(int(ok.sum()))])
        return ray * depth[ok]

    def getCloudNeighbors(self) -> np.ndarray:
        """getCloudNeighbors (DepthEstimator.cpp:335-337): never filled by the reference (push commented out,
        DepthEstimator.cpp:663) — always empty."""
        return np.empty((3, 0))

    def getCloudRansacPlane(self, slot: int = 0) -> np.ndarray:
        """getCloudRansacPlane (DepthEstimator.cpp:396-398): 3 x n camera-frame ground-plane inliers
        (`_points_groundplane`, DepthEstimator.cpp:294-308)."""
        n = C.c_int64(0)
        self._check(self._lib.mld_get_ground_plane_cloud(self._ctx, slot, None, 0, C.byref(n)))
        out = np.empty((n.value, 3), dtype=np.float64)
        if n.value:
            self._check(self._lib.mld_get_ground_plane_cloud(self._ctx, slot, out.ctypes.data, n.value, C.byref(n)))
        return out.T

    @staticmethod
    def _uv_host(uv, layout: Optional[str] = None) -> np.ndarray:
        a = np.asarray(uv, dtype=np.float64)
        if a.ndim != 2:
            raise DepthEstimatorError(capi.MLD_ERR_INVALID_ARG, "uv must be 2xF or Fx2")
        if layout not in (None, "2xF", "Fx2"):
            raise DepthEstimatorError(capi.MLD_ERR_INVALID_ARG, "uv_layout must be '2xF' or 'Fx2'")
        if a.shape == (2, 2) and layout is None:
            # two features as Eigen::Matrix2Xd (columns) or as rows: silently guessing would transpose one of them
            raise DepthEstimatorError(capi.MLD_ERR_INVALID_ARG,
                                      "a 2x2 uv array is ambiguous: pass uv_layout='2xF' (Eigen::Matrix2Xd) or 'Fx2'")
        if layout == "2xF" or (layout is None and a.shape[0] == 2 and a.shape[1] != 2):
            if a.shape[0] != 2:
                raise DepthEstimatorError(capi.MLD_ERR_INVALID_ARG, "uv must be 2xF")
            a = a.T  # Matrix2Xd -> interleaved (u0,v0,u1,v1,...) == column-major 2xF
        elif a.shape[1] != 2:
            raise DepthEstimatorError(capi.MLD_ERR_INVALID_ARG, "uv must be 2xF or Fx2")
        return np.ascontiguousarray(a)

    @staticmethod
    def _uv_device(uv, layout: Optional[str] = None):
        """Device features as contiguous float64 [F,2] (= column-major 2 x F).  Same layout rule as the host path: a
        2 x 2 tensor needs `layout`; "2xF" (Eigen::Matrix2Xd) is transposed into a new tensor."""
        import torch
        if layout not in (None, "2xF", "Fx2"):
            raise DepthEstimatorError(capi.MLD_ERR_INVALID_ARG, "uv_layout must be '2xF' or 'Fx2'")
        if uv.dtype != torch.float64 or uv.dim() != 2:
            raise DepthEstimatorError(capi.MLD_ERR_INVALID_ARG, "device uv must be float64 [F,2] (or [2,F] with uv_layout='2xF')")
        if tuple(uv.shape) == (2, 2) and layout is None:
            raise DepthEstimatorError(capi.MLD_ERR_INVALID_ARG,
                                      "a 2x2 uv tensor is ambiguous: pass uv_layout='2xF' (Eigen::Matrix2Xd) or 'Fx2'")
        if layout == "2xF":
            if uv.shape[0] != 2:
                raise DepthEstimatorError(capi.MLD_ERR_INVALID_ARG, "uv must be 2xF")
            return uv.t().contiguous()
        if uv.shape[1] != 2 or not uv.is_contiguous():
            raise DepthEstimatorError(capi.MLD_ERR_INVALID_ARG, "device uv must be contiguous float64 [F,2]")
        return uv

    def CalculateDepths(self, uvs: Sequence, depths: Sequence, types: Optional[Sequence] = None):
        """Batched CalculateDepth over slots 0..len-1 (torch CUDA tensors, asynchronous on `stream`)."""
        n_slots = len(uvs)
        uvp = (C.c_void_p * n_slots)()
        dp = (C.c_void_p * n_slots)()
        tp = (C.c_void_p * n_slots)()
        Fs = (C.c_int64 * n_slots)()
        for i in range(n_slots):
            uvd = self._uv_device(uvs[i])
            uvp[i] = uvd.data_ptr()
            Fs[i] = int(uvd.shape[0])
            dp[i] = depths[i].data_ptr()
            tp[i] = types[i].data_ptr() if types is not None else None
        self._after_torch(*uvs[:1])
        self._check(self._lib.mld_calculate_depths_device(self._ctx, n_slots, uvp, Fs, dp, tp if types is not None else None))

    # ------------------------------------------------------------------ debug getters
    def getVisibleCount(self, slot: int = 0) -> int:
        n = C.c_int64(0)
        self._check(self._lib.mld_get_visible_count(self._ctx, slot, C.byref(n)))
        return int(n.value)

    def getPointsCloudImageCs(self, slot: int = 0) -> np.ndarray:
        """getPointsCloudImageCs -> 2 x Nvis (DepthEstimator.cpp:392-394)."""
        nvis = self.getVisibleCount(slot)
        out = np.empty((nvis, 2), dtype=np.float64)
        self._check(self._lib.mld_get_visible_image_points(self._ctx, slot, out.ctypes.data, nvis))
        return out.T

    def getPointIndex(self, slot: int = 0) -> np.ndarray:
        nvis = self.getVisibleCount(slot)
        out = np.empty(nvis, dtype=np.int32)
        self._check(self._lib.mld_get_point_index(self._ctx, slot, out.ctypes.data, nvis))
        return out

    def getCloudCameraCs(self, n_points: int, slot: int = 0) -> np.ndarray:
        """getCloudCameraCs -> 3 x N float64 (DepthEstimator.cpp:314-334)."""
        out = np.empty((n_points, 3), dtype=np.float64)
        self._check(self._lib.mld_get_cloud_camera_cs(self._ctx, slot, out.ctypes.data, n_points))
        return out.T

    def getPixelMap(self, slot: int = 0) -> np.ndarray:
        """NeighborFinderPixel::_img_points_lidar as [H, W] int32 (visible index or -1)."""
        W, H = self._camera.width, self._camera.height
        out = np.empty(W * H, dtype=np.int32)
        self._check(self._lib.mld_get_pixel_map(self._ctx, slot, out.ctypes.data, W * H))
        return out.reshape(H, W)

    def getPointDepthCamVisible(self, index: int, slot: int = 0) -> float:
        d = C.c_double(0)
        self._check(self._lib.mld_get_point_depth_cam_visible(self._ctx, slot, int(index), C.byref(d)))
        return float(d.value)

    @staticmethod
    def resultHistogram(types) -> np.ndarray:
        """DepthCalculationStatistics counterpart: counts per DepthResultType."""
        t = np.ascontiguousarray(np.asarray(types.cpu() if hasattr(types, "cpu") else types), dtype=np.int32)
        counts = (C.c_int64 * capi.MLD_RESULT_TYPE_COUNT)()
        capi.load().mld_result_histogram(t.ctypes.data, int(t.size), counts)
        return np.array(list(counts), dtype=np.int64)

    def getDepthCalcStats(self) -> dict:
        """getDepthCalcStats (DepthEstimator.cpp:400-402): result-type counters of the last host-input CalculateDepth
        call, keyed by the DepthResultType names, plus "PointCount"."""
        if self._last_types is None:
            return {"PointCount": 0}
        counts = self.resultHistogram(self._last_types)
        out = {capi.RESULT_TYPE_NAMES[i]: int(c) for i, c in enumerate(counts)}
        out["PointCount"] = int(self._last_types.size)
        return out

    # ------------------------------------------------------------------ measurement hooks
    def frameTiming(self) -> dict:
        """Phases of the last one-frame call in microseconds (mld_frame_timing): host-clock entries always, the GPU
        phases (h2d / plane / kernels / d2h / gpu) only with timingEnable(True)."""
        out = (C.c_double * 10)()
        self._check(self._lib.mld_frame_timing(self._ctx, out))
        keys = ("h2d_us", "plane_us", "kernels_us", "d2h_us", "api_us", "wait_us", "total_us", "gpu_us", "pre_us", "copycall_us")
        return {k: float(v) for k, v in zip(keys, out)}

    def timingEnable(self, on: bool = True):
        self._check(self._lib.mld_timing_enable(self._ctx, 1 if on else 0))

    def timingReset(self):
        self._check(self._lib.mld_timing_reset(self._ctx))

    def kernelTimeMs(self, which: int):
        avg = C.c_double(0)
        cnt = C.c_int64(0)
        self._check(self._lib.mld_kernel_time_ms(self._ctx, which, C.byref(avg), C.byref(cnt)))
        return float(avg.value), int(cnt.value)
