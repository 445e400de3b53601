"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/mld.h declares, mirrors the
reference's parameter sets, and refuses to compute without a GPU (no fallback)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from mono_lidar_depth_amd import capi, synth

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    header = (ROOT / "include" / "mld.h").read_text()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)  # drop comments
    declared = set(re.findall(r"\b(mld_[a-z0-9_]+)\s*\(", header))
    lib = capi.load()
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, f"libmld_hip.so lacks: {missing}"
    assert declared == set(capi.EXPORTED_SYMBOLS), declared ^ set(capi.EXPORTED_SYMBOLS)
    assert lib.mld_abi_version() == capi.MLD_ABI_VERSION == 8


def test_struct_layout_matches_header():
    # doubles first, then int32s: 15*8 + 26*4 = 224 bytes; camera 3*8 + 2*4 = 32
    assert C.sizeof(capi.MldParams) == 224
    assert C.sizeof(capi.MldCamera) == 32
    header = (ROOT / "include" / "mld.h").read_text()
    body = header[header.index("typedef struct mld_params {"):header.index("} mld_params;")]
    names = re.findall(r"^\s*(?:double|int32_t)\s+(\w+);", body, flags=re.M)
    assert names == [n for n, _ in capi.MldParams._fields_]


def test_parameter_sets():
    d = capi.params_default()  # DepthEstimatorParameters.h defaults
    assert (d.pixelarea_search_witdh, d.pixelarea_search_height, d.radiusSearch_count_min) == (12, 15, 3)
    assert d.histogram_segmentation_bin_witdh == 0.5 and d.viewray_plane_orthoganality_treshold == 1.0
    c0 = capi.params_c0()  # parameters.yaml with do_use_depth_segmentation 0
    assert (c0.pixelarea_search_witdh, c0.pixelarea_search_height, c0.radiusSearch_count_min) == (6, 9, 1)
    assert c0.histogram_segmentation_bin_witdh == 0.3 and c0.histogram_segmentation_min_pointcount == 3
    assert c0.viewray_plane_orthoganality_treshold == 0.03 and c0.triangleplanar_crossnorm_treshold == 0.1
    assert c0.treshold_depth_max == 100 and c0.treshold_depth_min == 0 and c0.treshold_depth_local_value == 0.5
    assert c0.do_use_ransac_plane == 1 and c0.plane_estimator_use_mestimator == 1 and c0.do_use_depth_segmentation == 0
    assert c0.ransac_plane_point_distance_treshold == 0.2 and c0.pca_treshold_2_1_rel_min == 1.5


def test_params_from_file(tmp_path):
    """OpenCV-FileStorage semantics: absent keys read as 0, (int) of a real rounds, bools collapse to 0/1."""
    y = tmp_path / "p.yaml"
    y.write_text("%YAML:1.0\n\n# comment\nneighbor_search_mode: 0 # trailing\npixelarea_search_witdh: 6\n"
                 "pixelarea_search_height: 9\nradiusSearch_count_min: 1\ndo_use_histogram_segmentation: 1\n"
                 "histogram_segmentation_bin_witdh: 0.3 # in meters\nhistogram_segmentation_min_pointcount: 3\n"
                 "do_use_depth_segmentation: 0\ntreshold_depth_enabled: 1\ntreshold_depth_mode: 0 \n"
                 "treshold_depth_max: 100\ntreshold_depth_min: 0\ntreshold_depth_local_enabled: 1\n"
                 "treshold_depth_local_mode: 0 \ntreshold_depth_local_valuetype: 1 \ntreshold_depth_local_value: 0.5\n"
                 "do_use_PCA: 0\npca_debug: 0.01\npca_treshold_3_abs_min: 0.005\npca_treshold_3_2_rel_max: 15\n"
                 "pca_treshold_2_1_rel_min: 1.5\ndo_use_ransac_plane: 1\nransac_plane_point_distance_treshold: 0.2\n"
                 "plane_estimator_use_triangle_maximation: 0 \nplane_estimator_use_leastsquares: 0\n"
                 "plane_estimator_use_mestimator: 1\nplane_estimator_z_x_min_relation: 0\n"
                 "do_use_cut_behind_camera: 1\ndo_use_triangle_size_maximation: 1\n"
                 "do_check_triangleplanar_condition: 1\ntriangleplanar_crossnorm_treshold: 0.1\n"
                 "viewray_plane_orthoganality_treshold: 0.03\nunknown_key: 7\nransac_plane_distance_treshold: 0.3\n"
                 "ransac_plane_max_iterations: 10000\nransac_plane_probability: 0.999\nransac_plane_use_refinement: 1\n"
                 "ransac_plane_refinement_treshold: 10.2\nransac_plane_use_camx_treshold: 0\n"
                 "ransac_plane_treshold_camx: 2.0\n")
    p = capi.params_from_file(str(y))
    c0 = capi.params_c0()
    for name, _ in capi.MldParams._fields_:
        if name in ("ransac_plane_min_z", "ransac_plane_max_z"):
            continue  # absent from the yaml: the loader reads 0 (cv::FileStorage), C0 keeps the header defaults
        assert getattr(p, name) == getattr(c0, name), name
    assert p.ransac_plane_min_z == 0 and p.ransac_plane_max_z == 0
    assert p.set_all_depths_to_zero == 0  # key absent in the yaml -> 0
    # the loader says which mirrored keys were absent (the reference's own parameters.yaml lacks exactly these: its
    # RansacPlane then keeps only points with z == 0 and the estimate fails - reproduced, and reported)
    assert p.absent_keys == ["ransac_plane_min_z", "ransac_plane_max_z", "set_all_depths_to_zero"]
    y2 = tmp_path / "q.yaml"
    y2.write_text("do_use_ransac_plane: 5\ntreshold_depth_max: 79.6\n")
    q = capi.params_from_file(str(y2))
    assert q.do_use_ransac_plane == 1 and q.treshold_depth_max == 80 and q.pixelarea_search_witdh == 0
    assert "pixelarea_search_witdh" in q.absent_keys and "treshold_depth_max" not in q.absent_keys
    y3 = tmp_path / "r.yaml"   # values beyond int: saturate (cvRound of such a value is undefined in the reference)
    y3.write_text("treshold_depth_max: 1e300\ntreshold_depth_min: -1e300\nradiusSearch_count_min: nan\n")
    r = capi.params_from_file(str(y3))
    assert r.treshold_depth_max == 2**31 - 1 and r.treshold_depth_min == -2**31 and r.radiusSearch_count_min == 0
    with pytest.raises(RuntimeError, match="Cant find settings file"):
        capi.params_from_file(str(tmp_path / "missing.yaml"))


def test_result_histogram():
    t = np.array([1, 1, 2, 16, 3, 16, 16, 99, -4], dtype=np.int32)
    counts = (C.c_int64 * capi.MLD_RESULT_TYPE_COUNT)()
    assert capi.load().mld_result_histogram(t.ctypes.data, t.size, counts) == 0
    c = list(counts)
    assert c[1] == 2 and c[2] == 1 and c[3] == 1 and c[16] == 3 and sum(c) == 7


def test_create_rejects_invalid_configurations_before_touching_the_gpu():
    lib = capi.load()
    cam = capi.MldCamera(synth.KITTI_F, synth.KITTI_CU, synth.KITTI_CV, synth.KITTI_W, synth.KITTI_H)
    T = np.ascontiguousarray(synth.T_CAM_LIDAR)
    for kw, code, text in ((dict(neighbor_search_mode=1), capi.MLD_ERR_UNSUPPORTED_MODE, "neighbor_search_mode"),
                           (dict(do_use_depth_segmentation=1), capi.MLD_ERR_UNSUPPORTED_MODE, "Region growing"),
                           (dict(plane_estimator_use_mestimator=0), capi.MLD_ERR_NO_ROAD_ESTIMATOR, "No road depth"),
                           (dict(plane_estimator_use_mestimator=0, plane_estimator_use_leastsquares=1),
                            capi.MLD_ERR_UNSUPPORTED_MODE, "leastsquares")):
        st = C.c_int(0)
        P = capi.params_c0().replace(**kw)
        ctx = lib.mld_create(C.byref(P), C.byref(cam), T.ctypes.data_as(C.POINTER(C.c_double)), 0, 1, 0, 0, C.byref(st))
        assert not ctx and st.value == code and text in lib.mld_create_error().decode()


def test_no_cpu_fallback_without_a_gpu():
    """On a box without a HIP device mld_create fails loudly; with one (the GPU box) it succeeds."""
    import torch
    lib = capi.load()
    cam = capi.MldCamera(synth.KITTI_F, synth.KITTI_CU, synth.KITTI_CV, synth.KITTI_W, synth.KITTI_H)
    T = np.ascontiguousarray(synth.T_CAM_LIDAR)
    P = capi.params_c0()
    st = C.c_int(0)
    ctx = lib.mld_create(C.byref(P), C.byref(cam), T.ctypes.data_as(C.POINTER(C.c_double)), 0, 1, 0, 0, C.byref(st))
    if torch.cuda.is_available():
        assert ctx
        lib.mld_destroy(C.c_void_p(ctx))
    else:
        assert not ctx and st.value == capi.MLD_ERR_HIP
        assert "no CPU fallback" in lib.mld_create_error().decode()


def test_product_package_never_imports_the_oracle():
    for path in (ROOT / "mono_lidar_depth_amd").rglob("*"):
        if path.suffix in (".py", ".hip", ".cpp", ".h", ".hpp") and path.is_file():
            text = path.read_text()
            assert "oracle" not in text.replace("TEST INFRASTRUCTURE", ""), f"{path} mentions the oracle"


def test_abi_version_constant_matches_the_header():
    import re
    from pathlib import Path
    hdr = (Path(__file__).resolve().parent.parent / "include" / "mld.h").read_text()
    assert int(re.search(r"#define\s+MLD_ABI_VERSION\s+(\d+)", hdr).group(1)) == capi.MLD_ABI_VERSION
    assert capi.load().mld_abi_version() == capi.MLD_ABI_VERSION


def test_params_from_file_tolerates_formatting(tmp_path):
    """CRLF line ends, tabs, missing space after the colon, scientific notation, signs, inline comments, a real value
    for an int field (cvRound), an unreadable file."""
    y = tmp_path / "p.yaml"
    y.write_bytes(b"%YAML:1.0\r\n---\r\npixelarea_search_witdh:8\r\n\tpixelarea_search_height:\t11 # rows\r\n"
                  b"histogram_segmentation_bin_witdh: 2.5e-1\r\ntreshold_depth_local_value: +0.75\r\n"
                  b"ransac_plane_min_z: -3.5\r\nradiusSearch_count_min: 2.6\r\ndo_use_PCA: 7\r\n"
                  b"plane_estimator_use_mestimator: 1\r\nnot a key value line\r\n: 5\r\nsome_key:\r\n")
    p = capi.params_from_file(str(y))
    assert p.pixelarea_search_witdh == 8 and p.pixelarea_search_height == 11
    assert p.histogram_segmentation_bin_witdh == 0.25 and p.treshold_depth_local_value == 0.75
    assert p.ransac_plane_min_z == -3.5
    assert p.radiusSearch_count_min == 3  # (int) of a real node rounds to nearest
    assert p.do_use_PCA == 1              # bool member: any non-zero int
    assert p.do_use_ransac_plane == 0     # absent key reads as 0
    with pytest.raises(Exception, match="Cant find settings file"):
        capi.params_from_file(str(tmp_path / "missing.yaml"))


def test_uv_layouts_of_the_python_mirror():
    """Eigen::Matrix2Xd (2 x F, columns are features) and F x 2 are both accepted; 2 x 2 is ambiguous and must be
    disambiguated instead of being silently transposed."""
    from mono_lidar_depth_amd.depth_estimator import DepthEstimator, DepthEstimatorError
    e = DepthEstimator.__new__(DepthEstimator)
    a = np.arange(10, dtype=np.float64).reshape(5, 2)
    assert np.array_equal(e._uv_host(a), a) and np.array_equal(e._uv_host(a.T), a)
    two = np.array([[1.0, 2.0], [3.0, 4.0]])
    with pytest.raises(DepthEstimatorError, match="ambiguous"):
        e._uv_host(two)
    assert np.array_equal(e._uv_host(two, "2xF"), [[1.0, 3.0], [2.0, 4.0]])  # columns (1,3) and (2,4) are the features
    assert np.array_equal(e._uv_host(two, "Fx2"), two)
    # the device path follows the same rule (CPU tensors suffice for the shape logic)
    import torch
    t2 = torch.tensor(two, dtype=torch.float64)
    with pytest.raises(DepthEstimatorError, match="ambiguous"):
        e._uv_device(t2)
    assert torch.equal(e._uv_device(t2, "2xF"), torch.tensor([[1.0, 3.0], [2.0, 4.0]], dtype=torch.float64))
    assert torch.equal(e._uv_device(t2, "Fx2"), t2)
    t5 = torch.tensor(a)
    assert torch.equal(e._uv_device(t5), t5) and torch.equal(e._uv_device(t5.t().contiguous(), "2xF"), t5)


def test_pack_points_host_repacks_pcl_records():
    """mld_pack_points_host: 32-byte pcl::PointXYZI records (x,y,z,pad | intensity,pad,pad,pad) -> packed {x,y,z,intensity};
    16-byte sources are copied; any thread count, aligned or unaligned destination, empty input; bad strides are refused."""
    lib = capi.load()
    rng = np.random.default_rng(3)
    for n in (0, 1, 5, 65536, 65537, 300001):
        src = rng.standard_normal((n, 8)).astype(np.float32)
        want = np.ascontiguousarray(src[:, [0, 1, 2, 4]])
        for threads in (1, 3, 64):
            for off in (0, 1):  # (a destination that is not 16-byte aligned takes the plain-store path)
                raw = np.full(4 * n + 8, np.float32(-7.0), dtype=np.float32)
                dst = raw[off:off + 4 * n]
                rc = lib.mld_pack_points_host(dst.ctypes.data, src.ctypes.data, n, 32, threads)
                assert rc == 0 and np.array_equal(dst.reshape(n, 4), want)
                assert (raw[:off] == -7.0).all() and (raw[off + 4 * n:] == -7.0).all()   # nothing written outside
        src16 = rng.standard_normal((n, 4)).astype(np.float32)
        dst16 = np.empty_like(src16)
        assert lib.mld_pack_points_host(dst16.ctypes.data, src16.ctypes.data, n, 16, 4) == 0
        assert np.array_equal(dst16, src16)
    a = np.zeros((4, 8), dtype=np.float32)
    assert lib.mld_pack_points_host(a.ctypes.data, a.ctypes.data, 4, 24, 1) == capi.MLD_ERR_INVALID_ARG
    assert lib.mld_pack_points_host(None, a.ctypes.data, 4, 32, 1) == capi.MLD_ERR_INVALID_ARG
    assert lib.mld_pack_points_host(None, None, 0, 32, 1) == 0
