/*
 * retain_best.cpp -- TEST INFRASTRUCTURE (see oracle.h).  cv::KeyPointsFilter::retainBest
 * (OpenCV features2d/src/keypoint.cpp) restated on top of the REAL std::nth_element and
 * std::partition of this container's libstdc++.
 *
 * Why a C++ file in a C oracle: the keypoint ORDER cv2 leaves behind after
 *     std::nth_element(kp.begin(), kp.begin() + n - 1, kp.end(), KeypointResponseGreater());
 *     new_end = std::partition(kp.begin() + n, kp.end(), response >= kp[n - 1].response);
 * is defined by libstdc++'s introselect / Hoare partition, not by the C++ standard.  The
 * opencv-python wheels the reference installs (requirements.txt:1, manylinux, GCC) inline those
 * header algorithms, whose code has not changed across GCC 5 .. 13, so calling them here on
 * the same input sequence (FAST's raster emission) gives the same permutation.  ORB calls
 * retainBest twice per level (orb.cpp computeKeyPoints: 2 * quota on the FAST score, then quota
 * on the Harris response); descriptor row order = keypoint order, which decides matcher ties,
 * the stable sort before the top-500 cut (reference src/core/pose_estimator.py:147-151) and so
 * which points the fixed-seed RANSAC samples.
 */
#include <algorithm>
#include <cstdint>
#include <vector>

namespace {
struct Kp { float response; int32_t id; };          // the comparator only reads .response; moves are whole-element, as for cv::KeyPoint
struct ResponseGreater { bool operator()(const Kp &a, const Kp &b) const { return a.response > b.response; } };
struct ResponseGE { float v; bool operator()(const Kp &k) const { return k.response >= v; } };
}

/* In: n responses in emission order, ids[i] = caller's handle of element i.  Out: the retained
 * elements' ids in cv2's resulting order; returns their number. */
extern "C" int orc_retain_best(const float *resp, int32_t *ids, int n, int n_points)
{
    std::vector<Kp> kp((size_t)n);
    for (int i = 0; i < n; ++i) { kp[(size_t)i].response = resp[i]; kp[(size_t)i].id = ids[i]; }
    if (n_points >= 0 && kp.size() > (size_t)n_points) {
        if (n_points == 0) return 0;
        std::nth_element(kp.begin(), kp.begin() + n_points - 1, kp.end(), ResponseGreater());
        const float ambiguous = kp[(size_t)n_points - 1].response;
        std::vector<Kp>::iterator new_end = std::partition(kp.begin() + n_points, kp.end(), ResponseGE{ambiguous});
        kp.resize((size_t)(new_end - kp.begin()));
    }
    for (size_t i = 0; i < kp.size(); ++i) ids[i] = kp[i].id;
    return (int)kp.size();
}
