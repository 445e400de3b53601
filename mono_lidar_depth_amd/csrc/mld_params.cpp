// Parameter sets for the DepthEstimator hot path (host only, no HIP).
//
// Mirrors Mono_Lidar::DepthEstimatorParameters (monolidar_fusion/include/monolidar_fusion/
// DepthEstimatorParameters.h:7-173) and its loader DepthEstimatorParameters::fromFile
// (monolidar_fusion/src/DepthEstimatorParameters.cpp:16-114) for the fields the path reads.
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/mld.h"

extern "C" {

void mld_params_default(mld_params* p) {
    std::memset(p, 0, sizeof(*p));
    p->neighbor_search_mode = 0;
    p->pixelarea_search_witdh = 12;
    p->pixelarea_search_height = 15;
    p->radiusSearch_count_min = 3;
    p->do_use_histogram_segmentation = 1;
    p->histogram_segmentation_bin_witdh = 0.5;
    p->histogram_segmentation_min_pointcount = 3;
    p->do_use_depth_segmentation = 0;
    p->treshold_depth_enabled = 1;
    p->treshold_depth_mode = 0;
    p->treshold_depth_max = 100;
    p->treshold_depth_min = 0;
    p->treshold_depth_local_enabled = 1;
    p->treshold_depth_local_mode = 0;
    p->treshold_depth_local_valuetype = 1;
    p->treshold_depth_local_value = 0.5;
    p->do_use_PCA = 0;
    p->pca_treshold_3_abs_min = 0.005;
    p->pca_treshold_3_2_rel_max = 15;
    p->pca_treshold_2_1_rel_min = 0.5;
    p->do_use_ransac_plane = 1;
    p->ransac_plane_point_distance_treshold = 0.2;
    p->ransac_plane_distance_treshold = 0.2;
    p->ransac_plane_min_z = -10000;
    p->ransac_plane_max_z = 10000;
    p->ransac_plane_max_iterations = 10000;
    p->ransac_plane_use_refinement = 1;
    p->ransac_plane_refinement_treshold = 10.2;
    p->ransac_plane_probability = 0.999;
    p->ransac_plane_use_camx_treshold = 0;
    p->ransac_plane_treshold_camx = 2.0;
    p->plane_estimator_use_triangle_maximation = 0;
    p->plane_estimator_z_x_min_relation = 0;
    p->plane_estimator_use_leastsquares = 0;
    p->plane_estimator_use_mestimator = 1;
    p->do_use_cut_behind_camera = 1;
    p->do_use_triangle_size_maximation = 1;
    p->do_check_triangleplanar_condition = 1;
    p->triangleplanar_crossnorm_treshold = 0.1;
    p->viewray_plane_orthoganality_treshold = 01;  // octal literal 1, as in the reference (:155)
    p->set_all_depths_to_zero = 0;
}

// monolidar_fusion/parameters.yaml, with do_use_depth_segmentation forced to 0 (the shipped value 1
// makes every feature throw "Region growing not supported!", DepthEstimator.cpp:608).
void mld_params_c0(mld_params* p) {
    mld_params_default(p);
    p->pixelarea_search_witdh = 6;
    p->pixelarea_search_height = 9;
    p->radiusSearch_count_min = 1;
    p->histogram_segmentation_bin_witdh = 0.3;
    p->histogram_segmentation_min_pointcount = 3;
    p->pca_treshold_2_1_rel_min = 1.5;
    p->viewray_plane_orthoganality_treshold = 0.03;
    p->ransac_plane_distance_treshold = 0.3;
    // ransac_plane_min_z / max_z are absent from the yaml: fromFile reads them as 0 (see mld_params_from_file);
    // C0 keeps the header defaults so that the z pass-through stays off as in DepthEstimator.cpp:282
}

namespace {

struct Field {
    const char* key;
    int is_double;
    size_t offset;
};

#define FD(name) {#name, 1, offsetof(mld_params, name)}
#define FI(name) {#name, 0, offsetof(mld_params, name)}
const Field kFields[] = {
    FI(neighbor_search_mode),
    FI(pixelarea_search_witdh),
    FI(pixelarea_search_height),
    FI(radiusSearch_count_min),
    FI(do_use_histogram_segmentation),
    FD(histogram_segmentation_bin_witdh),
    FI(histogram_segmentation_min_pointcount),
    FI(do_use_depth_segmentation),
    FI(treshold_depth_enabled),
    FI(treshold_depth_mode),
    FI(treshold_depth_max),
    FI(treshold_depth_min),
    FI(treshold_depth_local_enabled),
    FI(treshold_depth_local_mode),
    FI(treshold_depth_local_valuetype),
    FD(treshold_depth_local_value),
    FI(do_use_PCA),
    FD(pca_treshold_3_abs_min),
    FD(pca_treshold_3_2_rel_max),
    FD(pca_treshold_2_1_rel_min),
    FI(do_use_ransac_plane),
    FD(ransac_plane_point_distance_treshold),
    FD(ransac_plane_distance_treshold),
    FD(ransac_plane_min_z),
    FD(ransac_plane_max_z),
    FI(ransac_plane_max_iterations),
    FI(ransac_plane_use_refinement),
    FD(ransac_plane_refinement_treshold),
    FD(ransac_plane_probability),
    FI(ransac_plane_use_camx_treshold),
    FD(ransac_plane_treshold_camx),
    FI(plane_estimator_use_triangle_maximation),
    FD(plane_estimator_z_x_min_relation),
    FI(plane_estimator_use_leastsquares),
    FI(plane_estimator_use_mestimator),
    FI(do_use_cut_behind_camera),
    FI(do_use_triangle_size_maximation),
    FI(do_check_triangleplanar_condition),
    FD(triangleplanar_crossnorm_treshold),
    FD(viewray_plane_orthoganality_treshold),
    FI(set_all_depths_to_zero),
};
#undef FD
#undef FI

// Flags the reference stores in `bool` members: any non-zero int becomes 1.
bool is_bool_field(const char* key) {
    static const char* kBool[] = {"do_use_histogram_segmentation", "do_use_depth_segmentation",
                                  "treshold_depth_enabled", "treshold_depth_local_enabled", "do_use_PCA",
                                  "do_use_ransac_plane", "plane_estimator_use_triangle_maximation",
                                  "plane_estimator_use_leastsquares", "plane_estimator_use_mestimator",
                                  "do_use_cut_behind_camera", "do_use_triangle_size_maximation",
                                  "do_check_triangleplanar_condition", "set_all_depths_to_zero",
                                  "ransac_plane_use_refinement", "ransac_plane_use_camx_treshold"};
    for (const char* k : kBool)
        if (std::strcmp(k, key) == 0) return true;
    return false;
}

}  // namespace

// cv::FileStorage semantics kept: a key that is absent reads as 0 (empty FileNode), so every mirrored
// field is zeroed first; `(int)node` of a real value rounds to nearest (cvRound).
int mld_params_from_file(mld_params* p, const char* path, char* err, int err_len) {
    auto fail = [&](const std::string& m) {
        if (err && err_len > 0) std::snprintf(err, static_cast<size_t>(err_len), "%s", m.c_str());
        return static_cast<int>(MLD_ERR_INVALID_ARG);
    };
    if (!p || !path) return fail("null argument");
    FILE* f = std::fopen(path, "r");
    if (!f) return fail(std::string("Cant find settings file: ") + path);  // DepthEstimatorParameters.cpp:24
    std::memset(p, 0, sizeof(*p));
    bool seen[sizeof(kFields) / sizeof(kFields[0])] = {};
    char line[1024];
    while (std::fgets(line, sizeof(line), f)) {
        std::string s(line);
        size_t hash = s.find('#');
        if (hash != std::string::npos) s.erase(hash);
        if (s.empty() || s[0] == '%') continue;
        size_t colon = s.find(':');
        if (colon == std::string::npos) continue;
        std::string key = s.substr(0, colon), val = s.substr(colon + 1);
        auto trim = [](std::string& t) {
            size_t a = t.find_first_not_of(" \t\r\n"), b = t.find_last_not_of(" \t\r\n");
            t = (a == std::string::npos) ? std::string() : t.substr(a, b - a + 1);
        };
        trim(key);
        trim(val);
        if (key.empty() || val.empty()) continue;
        char* end = nullptr;
        double d = std::strtod(val.c_str(), &end);
        if (end == val.c_str()) continue;  // non-numeric node: reads as 0
        for (size_t fi = 0; fi < sizeof(kFields) / sizeof(kFields[0]); fi++) {
            const Field& fd = kFields[fi];
            if (key != fd.key) continue;
            seen[fi] = true;
            char* base = reinterpret_cast<char*>(p);
            if (fd.is_double) {
                *reinterpret_cast<double*>(base + fd.offset) = d;
            } else {
                // cvRound of the node's value; out-of-range / NaN values saturate instead of being undefined
                const double r = std::nearbyint(d);
                int iv = 0;
                if (r >= 2147483647.0)
                    iv = 2147483647;
                else if (r <= -2147483648.0)
                    iv = -2147483647 - 1;
                else if (r == r)
                    iv = static_cast<int>(r);
                if (is_bool_field(fd.key)) iv = (iv != 0) ? 1 : 0;
                *reinterpret_cast<int32_t*>(base + fd.offset) = iv;
            }
        }
    }
    std::fclose(f);
    // On success `err` carries a NOTE: the mirrored keys the file does not hold, which therefore read as 0 exactly as
    // the reference's cv::FileStorage reads them.  The reference's own parameters.yaml lacks ransac_plane_min_z /
    // ransac_plane_max_z: they load as 0 / 0, the z pass-through (RansacPlane.cpp:58-64) then keeps only points with
    // z == 0 and the RANSAC plane estimate fails - a caller that wants the header's +-10000 must say so in the file.
    if (err && err_len > 0) {
        std::string note;
        for (size_t fi = 0; fi < sizeof(kFields) / sizeof(kFields[0]); fi++)
            if (!seen[fi]) note += (note.empty() ? "absent (read as 0): " : ", ") + std::string(kFields[fi].key);
        std::snprintf(err, static_cast<size_t>(err_len), "%s", note.c_str());
    }
    return MLD_OK;
}

int mld_abi_version(void) { return MLD_ABI_VERSION; }

int mld_result_histogram(const int32_t* types, int64_t F, int64_t counts[MLD_RESULT_TYPE_COUNT]) {
    if (!types || !counts || F < 0) return MLD_ERR_INVALID_ARG;
    for (int i = 0; i < MLD_RESULT_TYPE_COUNT; i++) counts[i] = 0;
    for (int64_t i = 0; i < F; i++)
        if (types[i] >= 0 && types[i] < MLD_RESULT_TYPE_COUNT) counts[types[i]]++;
    return MLD_OK;
}

}  // extern "C"
