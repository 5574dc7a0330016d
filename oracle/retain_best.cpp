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

/* ---------------------------------------------------------------------------------------------------------
 * The same retainBest on libc++'s nth_element (LLVM <algorithm>: __nth_element, __sort3, __selection_sort),
 * restated from its published source -- experiment only (knob 1 = 4): opencv-python wheels for macOS are built
 * against libc++, whose selection algorithm leaves a different permutation behind than libstdc++'s introselect.
 * Used to find out which C++ runtime produced each of the reference's three result files. */
namespace {
template <class C> unsigned sort3_llvm(Kp *x, Kp *y, Kp *z, C c)
{
    unsigned r = 0;
    if (!c(*y, *x)) {
        if (!c(*z, *y)) return r;
        std::swap(*y, *z); r = 1;
        if (c(*y, *x)) { std::swap(*x, *y); r = 2; }
        return r;
    }
    if (c(*z, *y)) { std::swap(*x, *z); return 1; }
    std::swap(*x, *y); r = 1;
    if (c(*z, *y)) { std::swap(*y, *z); r = 2; }
    return r;
}
template <class C> void selection_sort_llvm(Kp *first, Kp *last, C comp)
{
    Kp *lm1 = last;
    for (--lm1; first != lm1; ++first) {
        Kp *m = first;
        for (Kp *i = first; ++i != last;) if (comp(*i, *m)) m = i;
        if (m != first) std::swap(*first, *m);
    }
}
template <class C> void nth_element_llvm(Kp *first, Kp *nth, Kp *last, C comp)
{
    const std::ptrdiff_t limit = 7;
    while (true) {
    restart:
        if (nth == last) return;
        std::ptrdiff_t len = last - first;
        switch (len) {
        case 0: case 1: return;
        case 2: if (comp(*--last, *first)) std::swap(*first, *last); return;
        case 3: { Kp *m = first; sort3_llvm(first, ++m, --last, comp); return; }
        }
        if (len <= limit) { selection_sort_llvm(first, last, comp); return; }
        Kp *m = first + len / 2;
        Kp *lm1 = last;
        unsigned n_swaps = sort3_llvm(first, m, --lm1, comp);
        Kp *i = first, *j = lm1;
        if (!comp(*i, *m)) {
            while (true) {
                if (i == --j) {
                    ++i; j = last;
                    if (!comp(*first, *--j)) {
                        while (true) {
                            if (i == j) return;
                            if (comp(*first, *i)) { std::swap(*i, *j); ++n_swaps; ++i; break; }
                            ++i;
                        }
                    }
                    if (i == j) return;
                    while (true) {
                        while (!comp(*first, *i)) ++i;
                        while (comp(*first, *--j)) ;
                        if (i >= j) break;
                        std::swap(*i, *j); ++n_swaps; ++i;
                    }
                    if (nth < i) return;
                    first = i;
                    goto restart;
                }
                if (comp(*j, *m)) { std::swap(*i, *j); ++n_swaps; break; }
            }
        }
        ++i;
        if (i < j) {
            while (true) {
                while (comp(*i, *m)) ++i;
                while (!comp(*--j, *m)) ;
                if (i >= j) break;
                std::swap(*i, *j); ++n_swaps;
                if (m == i) m = j;
                ++i;
            }
        }
        if (i != m && comp(*m, *i)) { std::swap(*i, *m); ++n_swaps; }
        if (nth == i) return;
        if (n_swaps == 0) {
            bool sorted = true;
            if (nth < i) {
                j = m = first;
                while (++j != i) { if (comp(*j, *m)) { sorted = false; break; } m = j; }
            } else {
                j = m = i;
                while (++j != last) { if (comp(*j, *m)) { sorted = false; break; } m = j; }
            }
            if (sorted) return;
        }
        if (nth < i) last = i; else first = ++i;
    }
}
}

extern "C" int orc_retain_best_llvm(const float *resp, int32_t *ids, int n, int n_points)
{
    std::vector<Kp> kp((size_t)n);
    for (int i = 0; i < n; ++i) { kp[(size_t)i].response = resp[i]; kp[(size_t)i].id = ids[i]; }
    if (n_points >= 0 && kp.size() > (size_t)n_points) {
        if (n_points == 0) return 0;
        nth_element_llvm(kp.data(), kp.data() + n_points - 1, kp.data() + kp.size(), ResponseGreater());
        const float ambiguous = kp[(size_t)n_points - 1].response;
        std::vector<Kp>::iterator new_end = std::partition(kp.begin() + n_points, kp.end(), ResponseGE{ambiguous});
        kp.resize((size_t)(new_end - kp.begin()));
    }
    for (size_t i = 0; i < kp.size(); ++i) ids[i] = kp[i].id;
    return (int)kp.size();
}

/* ---------------------------------------------------------------------------------------------------------
 * ... and on the MSVC STL's nth_element (<algorithm>: _Partition_by_median_guess_unchecked with Tukey's ninther,
 * _ISORT_MAX = 32, fat pivot), restated from its published source -- experiment only (knob 1 = 5): the
 * opencv-python wheels for Windows are built with MSVC. */
namespace {
template <class C> void med3_msvc(Kp *first, Kp *mid, Kp *last, C pred)
{
    if (pred(*mid, *first)) std::swap(*mid, *first);
    if (pred(*last, *mid)) {
        std::swap(*last, *mid);
        if (pred(*mid, *first)) std::swap(*mid, *first);
    }
}
template <class C> void guess_median_msvc(Kp *first, Kp *mid, Kp *last, C pred)
{
    const std::ptrdiff_t count = last - first;
    if (40 < count) {
        const std::ptrdiff_t step = (count + 1) >> 3, two_step = step << 1;
        med3_msvc(first, first + step, first + two_step, pred);
        med3_msvc(mid - step, mid, mid + step, pred);
        med3_msvc(last - two_step, last - step, last, pred);
        med3_msvc(first + step, mid, last - step, pred);
    } else
        med3_msvc(first, mid, last, pred);
}
template <class C> std::pair<Kp *, Kp *> partition_by_median_guess_msvc(Kp *first, Kp *last, C pred)
{
    Kp *mid = first + ((last - first) >> 1);
    guess_median_msvc(first, mid, last - 1, pred);
    Kp *pfirst = mid, *plast = pfirst + 1;
    while (first < pfirst && !pred(*(pfirst - 1), *pfirst) && !pred(*pfirst, *(pfirst - 1))) --pfirst;
    while (plast < last && !pred(*plast, *pfirst) && !pred(*pfirst, *plast)) ++plast;
    Kp *gfirst = plast, *glast = pfirst;
    for (;;) {
        for (; gfirst < last; ++gfirst) {
            if (pred(*pfirst, *gfirst)) continue;
            else if (pred(*gfirst, *pfirst)) break;
            else if (plast != gfirst) { std::swap(*plast, *gfirst); ++plast; }
            else ++plast;
        }
        for (; first < glast; --glast) {
            if (pred(*(glast - 1), *pfirst)) continue;
            else if (pred(*pfirst, *(glast - 1))) break;
            else if (--pfirst != glast - 1) std::swap(*pfirst, *(glast - 1));
        }
        if (glast == first && gfirst == last) return std::pair<Kp *, Kp *>(pfirst, plast);
        if (glast == first) {
            if (plast != gfirst) std::swap(*pfirst, *plast);
            ++plast;
            std::swap(*pfirst, *gfirst);
            ++pfirst; ++gfirst;
        } else if (gfirst == last) {
            if (--glast != --pfirst) std::swap(*glast, *pfirst);
            std::swap(*pfirst, *--plast);
        } else {
            std::swap(*gfirst, *--glast);
            ++gfirst;
        }
    }
}
template <class C> void insertion_sort_msvc(Kp *first, Kp *last, C pred)
{
    if (first == last) return;
    for (Kp *mid = first; ++mid != last;) {
        Kp *hole = mid;
        Kp val = *mid;
        if (pred(val, *first)) {
            ++hole;
            std::move_backward(first, mid, hole);
            *first = val;
        } else {
            for (Kp *prev = hole; pred(val, *--prev); hole = prev) *hole = *prev;
            *hole = val;
        }
    }
}
template <class C> void nth_element_msvc(Kp *first, Kp *nth, Kp *last, C pred)
{
    if (nth == last) return;
    while (32 < last - first) {
        std::pair<Kp *, Kp *> mid = partition_by_median_guess_msvc(first, last, pred);
        if (mid.second <= nth) first = mid.second;
        else if (mid.first <= nth) return;
        else last = mid.first;
    }
    insertion_sort_msvc(first, last, pred);
}
}

extern "C" int orc_retain_best_msvc(const float *resp, int32_t *ids, int n, int n_points)
{
    std::vector<Kp> kp((size_t)n);
    for (int i = 0; i < n; ++i) { kp[(size_t)i].response = resp[i]; kp[(size_t)i].id = ids[i]; }
    if (n_points >= 0 && kp.size() > (size_t)n_points) {
        if (n_points == 0) return 0;
        nth_element_msvc(kp.data(), kp.data() + n_points - 1, kp.data() + kp.size(), ResponseGreater());
        const float ambiguous = kp[(size_t)n_points - 1].response;
        std::vector<Kp>::iterator new_end = std::partition(kp.begin() + n_points, kp.end(), ResponseGE{ambiguous});
        kp.resize((size_t)(new_end - kp.begin()));
    }
    for (size_t i = 0; i < kp.size(); ++i) ids[i] = kp[i].id;
    return (int)kp.size();
}
