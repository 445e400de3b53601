"""AddressSanitizer / UndefinedBehaviorSanitizer builds of the CPU-side C++ (GPU sanitizers are not available on this
pool): the oracle (test infrastructure), the parameter loader of the product library (csrc/mld_params.cpp) and the
shim headers as far as they run without a GPU.  Each is compiled with -fsanitize=address,undefined into a small
driver, run on seeded inputs and must exit clean (the sanitizers abort with a non-zero status on a finding).
Plus a hypothesis fuzz of mld_params_from_file through the shipped library."""
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

from mono_lidar_depth_amd import capi, synth

ROOT = Path(__file__).resolve().parent.parent
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = {**os.environ, "ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"}

ORACLE_DRIVER = r"""
// drives the oracle's C interface on a seeded frame: set cloud, plane (supplied, RANSAC, semantic), features
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <vector>
#include "mld.h"
extern "C" {
void* orc_create(const mld_params*, const mld_camera*, const double*, int*);
void orc_destroy(void*);
int orc_set_cloud(void*, const void*, long long, int);
int orc_set_ground_plane(void*, const float*, const int32_t*, long long);
int orc_estimate_ground_plane(void*, const void*, long long, int, unsigned, float*, long long*);
int orc_calculate_depth(void*, const double*, long long, double*, int32_t*, int);
}
template <typename T> static std::vector<T> slurp(const char* p) {
    std::ifstream f(p, std::ios::binary);
    std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::vector<T> out(raw.size() / sizeof(T));
    std::memcpy(out.data(), raw.data(), out.size() * sizeof(T));
    return out;
}
int main(int argc, char** argv) {
    if (argc != 4) return 2;
    auto cloud = slurp<float>(argv[1]);
    auto uv = slurp<double>(argv[2]);
    auto inl = slurp<int32_t>(argv[3]);
    mld_params P;
    mld_params_c0(&P);
    mld_camera cam{721.5377, 609.5593, 172.854, 1242, 375};
    const double T[12] = {0, -1, 0, 0.0, 0, 0, -1, -0.08, 1, 0, 0, -0.27};
    int status = 0;
    void* h = orc_create(&P, &cam, T, &status);
    if (!h) return 6;
    const long long n = (long long)(cloud.size() / 4), F = (long long)(uv.size() / 2);
    std::vector<double> d((size_t)F);
    std::vector<int32_t> t((size_t)F);
    long long ok = 0;
    for (int pass = 0; pass < 3; pass++) {
        if (orc_set_cloud(h, cloud.data(), pass == 2 ? 2 : n, 16) != 0) return 3;   // (last pass: a two-point cloud)
        const float co[4] = {0.f, 0.f, 1.f, 1.73f};
        if (pass == 0) orc_set_ground_plane(h, co, inl.data(), (long long)inl.size());
        if (pass == 1) {
            float c2[4];
            long long ni = 0;
            if (orc_estimate_ground_plane(h, cloud.data(), n, 16, 7u, c2, &ni) != 0) return 4;
        }
        if (pass == 2) orc_set_ground_plane(h, nullptr, nullptr, 0);
        if (orc_calculate_depth(h, uv.data(), F, d.data(), t.data(), 2) != 0) return 5;
        for (long long i = 0; i < F; i++) ok += t[(size_t)i] == 1 || t[(size_t)i] == 16;
    }
    orc_calculate_depth(h, uv.data(), 0, d.data(), t.data(), 1);   // empty feature set
    orc_destroy(h);
    std::printf("oracle_asan ok %lld\n", ok);
    return 0;
}
"""

PARAMS_DRIVER = r"""
#include <cstdio>
#include <cstring>
#include "mld.h"
int main(int argc, char** argv) {
    int bad = 0;
    for (int i = 1; i < argc; i++) {
        mld_params p;
        char err[64];   // deliberately short: the note / error text must be truncated, not overrun
        const int rc = mld_params_from_file(&p, argv[i], err, sizeof(err));
        if (rc != MLD_OK && rc != MLD_ERR_INVALID_ARG) bad++;
        if (std::strlen(err) >= sizeof(err)) bad++;
        char big[4096];
        mld_params_from_file(&p, argv[i], big, sizeof(big));
        mld_params_from_file(&p, argv[i], nullptr, 0);
    }
    mld_params d;
    mld_params_default(&d);
    mld_params_c0(&d);
    long long counts[MLD_RESULT_TYPE_COUNT];
    const int32_t types[5] = {1, 16, -3, 21, 2000000000};
    mld_result_histogram(types, 5, reinterpret_cast<int64_t*>(counts));
    std::printf("params_asan ok %d\n", bad);
    return bad;
}
"""


def _compile(tmp_path, name, source, extra_src, extra_flags=()):
    src = tmp_path / f"{name}.cpp"
    src.write_text(source)
    exe = tmp_path / name
    cmd = ["g++", "-std=c++17", *SAN, *extra_flags, f"-I{ROOT / 'include'}", "-o", str(exe), str(src), *map(str, extra_src)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_oracle_under_asan_ubsan(tmp_path):
    exe = _compile(tmp_path, "oracle_asan", ORACLE_DRIVER,
                   [ROOT / "oracle" / "mld_oracle.cpp", ROOT / "mono_lidar_depth_amd" / "csrc" / "mld_params.cpp"],
                   ["-fopenmp", "-ffp-contract=off"])
    cloud = synth.make_cloud(synth.Scanner(64, 512, 2.0, -24.9), seed=4, frame=1)
    uv = synth.make_features(600, seed=4)
    uv[:5] = [[-3.0, 10.0], [1241.9, 374.9], [np.nan, 5.0], [0.0, 0.0], [1e9, 1e9]]   # edge features
    _, inl = synth.make_ground_plane(cloud)
    (tmp_path / "cloud.bin").write_bytes(cloud.tobytes())
    (tmp_path / "uv.bin").write_bytes(uv.tobytes())
    (tmp_path / "inl.bin").write_bytes(inl.tobytes())
    r = subprocess.run([str(exe), str(tmp_path / "cloud.bin"), str(tmp_path / "uv.bin"), str(tmp_path / "inl.bin")],
                       capture_output=True, text=True, timeout=600, env={**ENV, "OMP_NUM_THREADS": "2"})
    assert r.returncode == 0 and "oracle_asan ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def test_params_loader_under_asan_ubsan(tmp_path):
    exe = _compile(tmp_path, "params_asan", PARAMS_DRIVER, [ROOT / "mono_lidar_depth_amd" / "csrc" / "mld_params.cpp"])
    files = []
    cases = {
        "huge.yaml": "ransac_plane_max_iterations: 1e300\ntreshold_depth_max: -1e300\nradiusSearch_count_min: nan\n"
                     "pixelarea_search_witdh: inf\npixelarea_search_height: -inf\ndo_use_PCA: 4e18\n",
        "long.yaml": "histogram_segmentation_bin_witdh: " + "9" * 5000 + "\n" + "x" * 3000 + ": 1\n",
        "binary.yaml": "\x00\x01\x02:\x03\n:::\n: \n#\n%\nkey:\nkey: value\n",
        "empty.yaml": "",
        "ref.yaml": "pixelarea_search_witdh: 6\npixelarea_search_height: 9\ndo_use_ransac_plane: 1\n",
    }
    for name, text in cases.items():
        (tmp_path / name).write_text(text)
        files.append(str(tmp_path / name))
    files.append(str(tmp_path / "does_not_exist.yaml"))
    r = subprocess.run([str(exe), *files], capture_output=True, text=True, timeout=300, env=ENV)
    assert r.returncode == 0 and "params_asan ok 0" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def test_shim_host_logic_under_asan_ubsan(tmp_path):
    """The shim's host-side pieces that need no GPU: GroundPlane's bitmask lookup, SemanticPlane's image copies (strided
    source, cv::Mat stand-in), DepthEstimatorParameters::fromFile with its absent-key note, the statistics counters."""
    src = r"""
#include <cstdio>
#include <set>
#include <thread>
#include "monolidar_fusion/DepthEstimator.h"
// a foreign plane that rewrites its inlier list in place: same size, same storage (ADVICE r5: the lookup cache was keyed on
// size + data pointer only).  `tell` = whether it reports the rewrite through inliersChanged().
struct Rewriter : Mono_Lidar::GroundPlane {
    Rewriter() : GroundPlane({0.f, 0.f, 1.f, 0.f}, {10, 20, 30, 40}) {}
    void rewrite(bool tell) {
        const std::vector<int> other{11, 21, 31, 41};
        _inliersIndex = other;   // copy-assign into the existing storage
        if (tell) inliersChanged();
    }
};
int main(int argc, char** argv) {
    using namespace Mono_Lidar;
    int bad = 0;
    for (bool tell : {true, false}) {
        Rewriter rw;
        bad += (rw.CheckPointInPlane(20) && !rw.CheckPointInPlane(21)) ? 0 : 1;
        rw.rewrite(tell);
        bad += (!rw.CheckPointInPlane(20) && rw.CheckPointInPlane(21) && rw.CheckPointInPlane(41)) ? 0 : 1;
    }
    {   // concurrent lookups on a fresh plane: every thread may find the bitmask missing and build its own
        std::vector<int> many;
        for (int i = 0; i < 50000; i++) many.push_back(3 * i);
        GroundPlane big({0.f, 0.f, 1.f, 0.f}, many);
        int wrong[8] = {0};
        std::vector<std::thread> th;
        for (int t = 0; t < 8; t++)
            th.emplace_back([&big, &wrong, t] {
                for (int i = 0; i < 150000; i++) wrong[t] += (big.CheckPointInPlane(i) != (i % 3 == 0)) ? 1 : 0;
            });
        for (auto& x : th) x.join();
        for (int t = 0; t < 8; t++) bad += wrong[t] ? 1 : 0;
    }
    GroundPlane gp({0.f, 0.f, 1.f, 1.73f}, {5, 1, 900000, 64, 63, 0});
    for (int i : {5, 1, 900000, 64, 63, 0}) bad += gp.CheckPointInPlane(i) ? 0 : 1;
    for (int i : {-1, 2, 65, 899999, 900001, 2147483647}) bad += gp.CheckPointInPlane(i) ? 1 : 0;
    GroundPlane none;
    bad += none.CheckPointInPlane(0) ? 1 : 0;
    std::vector<uint8_t> img(7 * 16, 3);   // 5 columns used of a 16-byte row stride
    img[2 * 16 + 4] = 7;
    SemanticPlane sp(img.data(), 7, 5, 16, std::set<int>{6, 7}, 0.25);
    bad += (sp.image().size() == 35 && sp.image()[2 * 5 + 4] == 7 && sp.labels().size() == 2 && !sp.hasCamera()) ? 0 : 1;
    const cv::Mat m(7, 5, CV_8UC1, img.data(), 16);
    SemanticPlane::Camera cam;
    cam.f = 1;
    SemanticPlane sp2(m, cam, std::set<int>{6, 7, 8, 9}, 0.1);
    bad += (sp2.image() == sp.image() && sp2.hasCamera()) ? 0 : 1;
    DepthEstimatorParameters P;
    if (argc > 1) {
        P.fromFile(argv[1]);
        bad += (P.absentKeys.find("ransac_plane_min_z") != std::string::npos && P.pixelarea_search_witdh == 6) ? 0 : 1;
    }
    bool threw = false;
    try {
        P.fromFile("/nonexistent/file.yaml");
    } catch (const std::string&) {
        threw = true;
    }
    bad += threw ? 0 : 1;
    DepthCalculationStatistics st;
    const int32_t types[6] = {1, 1, 16, 2, 3, 99};
    st.SetFromTypes(types, 6);
    bad += (st.getSuccess() == 2 && st.getSuccessRoad() == 1 && st.getPointCount() == 6) ? 0 : 1;
    std::printf("shim_asan ok %d\n", bad);
    return bad;
}
"""
    exe = _compile(tmp_path, "shim_asan", src, [ROOT / "mono_lidar_depth_amd" / "csrc" / "mld_params.cpp"],
                   [f"-I{ROOT / 'tests' / 'stubs'}", f"-I{ROOT / 'mono_lidar_depth_amd' / 'host'}", "-DMLD_SHIM_NO_GPU_LINK",
                    "-Wl,--unresolved-symbols=ignore-all", "-pthread"])
    y = tmp_path / "ref.yaml"
    y.write_text("pixelarea_search_witdh: 6\npixelarea_search_height: 9\ndo_use_ransac_plane: 1\n")
    r = subprocess.run([str(exe), str(y)], capture_output=True, text=True, timeout=300, env=ENV)
    assert r.returncode == 0 and "shim_asan ok 0" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "no ransac_plane_min_z" in r.stderr   # the shim says what the reference does silently
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


_KEYS = [name for name, _ in capi.MldParams._fields_]


@settings(max_examples=150, deadline=None)
@given(st.lists(st.tuples(st.sampled_from(_KEYS + ["unknown_key", "", " ", "#"]),
                          st.one_of(st.floats(allow_nan=True, allow_infinity=True), st.integers(-2**70, 2**70),
                                    st.text(alphabet=" \t:#%-+.eE0123456789abcxyz", max_size=24))),
                max_size=40),
       st.sampled_from(["\n", "\r\n"]))
def test_params_from_file_fuzz(tmp_path_factory, lines, eol):
    """Whatever the file holds, the loader returns MLD_OK with every int field inside int32 and the bool fields 0 / 1."""
    d = tmp_path_factory.mktemp("fz")
    y = d / "p.yaml"
    y.write_text(eol.join(f"{k}: {v}" for k, v in lines), newline="")
    p = capi.params_from_file(str(y))
    for name, ctype in capi.MldParams._fields_:
        v = getattr(p, name)
        if isinstance(v, int):
            assert -2**31 <= v <= 2**31 - 1
    for name in ("do_use_PCA", "do_use_ransac_plane", "treshold_depth_enabled", "set_all_depths_to_zero"):
        assert getattr(p, name) in (0, 1)
    written = {k for k, _ in lines}
    assert all(k in written or k in p.absent_keys for k in _KEYS)


PACK_DRIVER = r"""
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "mld.h"
int main() {
    int bad = 0;
    for (long long n : {0LL, 1LL, 7LL, 65536LL, 200003LL}) {
        std::vector<float> src((size_t)n * 8 + 1), dst((size_t)n * 4 + 5, -7.f);
        for (size_t i = 0; i < src.size(); i++) src[i] = (float)(i % 977) * 0.25f;
        for (int threads : {1, 5, 64})
            for (int off : {0, 1}) {   // exact-size heap blocks: a record read or written past the end trips ASAN
                float* d = dst.data() + off;
                if (mld_pack_points_host(d, src.data(), n, 32, threads) != MLD_OK) bad++;
                for (long long i = 0; i < n; i++)
                    if (d[4 * i] != src[8 * i] || d[4 * i + 1] != src[8 * i + 1] || d[4 * i + 2] != src[8 * i + 2] ||
                        d[4 * i + 3] != src[8 * i + 4]) { bad++; break; }
            }
    }
    if (mld_pack_points_host(nullptr, nullptr, 3, 32, 1) != MLD_ERR_INVALID_ARG) bad++;
    std::printf("pack_asan ok %d\n", bad);
    return bad;
}
"""


def test_pack_points_host_under_asan_ubsan(tmp_path):
    """mld_pack_points_host reads exactly 20 bytes of the last 32-byte record's 32 and writes exactly 16 n bytes."""
    exe = _compile(tmp_path, "pack_asan", PACK_DRIVER, [ROOT / "mono_lidar_depth_amd" / "csrc" / "mld_host.cpp"], ["-pthread"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=ENV)
    assert r.returncode == 0 and "pack_asan ok 0" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
