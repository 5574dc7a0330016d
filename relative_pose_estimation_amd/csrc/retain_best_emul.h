// retain_best_emul.h -- cv::KeyPointsFilter::retainBest (OpenCV features2d/src/keypoint.cpp) as ONE sequential
// procedure, written so that a single GPU lane (or the host, in the unit test) can run it.
//
// Replaces (inside cv2.ORB.detectAndCompute, reference src/core/pose_estimator.py:85-91, :108)
//     std::nth_element(kp.begin(), kp.begin() + n - 1, kp.end(), KeypointResponseGreater());
//     new_end = std::partition(kp.begin() + n, kp.end(), response >= kp[n - 1].response);
// The ORDER these two calls leave behind is not defined by the C++ standard but by the runtime library, and it is
// visible in the reference's results: descriptor rows follow keypoint order, the matcher and the stable sort before
// the top-500 cut (pose_estimator.py:147-151) break ties by row index, and the fixed-seed RANSAC then samples other
// points.  The reference's committed result rows pin it (tests/test_reference_rows_cpu.py): the Salah and phone files
// come from a libstdc++ build of cv2 (Linux wheels, the reference's Dockerfile), the simulator file from an MSVC build
// (Windows wheels).  Both selection algorithms are restated here from the libraries' published sources
// (GCC bits/stl_algo.h __introselect / __unguarded_partition_pivot / __insertion_sort / __heap_select;
// MSVC <algorithm> nth_element / _Partition_by_median_guess_unchecked / _Guess_median_unchecked / _Insertion_sort_unchecked)
// on a plain array of elements; tests/test_retain_best_cpu.py checks the libstdc++ one against the real std::nth_element
// of this container on tie-heavy inputs, and the MSVC one through the simulator rows.
//
// An element is an opaque E; `GT(a, b)` is KeypointResponseGreater (response of a > response of b).  Elements are moved
// whole, like cv::KeyPoint.  Everything is index arithmetic on `a` -- no recursion, no allocation.
#pragma once

#ifdef __HIPCC__
#define RB_FN __device__ __forceinline__
#else
#define RB_FN static inline
#endif

namespace rb {

template <class E> RB_FN void swp(E *a, int i, int j) { const E t = a[i]; a[i] = a[j]; a[j] = t; }

// ------------------------------------------------------------------------------------------------ libstdc++
template <class E, class GT> RB_FN void gnu_move_median_to_first(E *a, int result, int ia, int ib, int ic, GT gt)
{
    if (gt(a[ia], a[ib])) {
        if (gt(a[ib], a[ic])) swp(a, result, ib);
        else if (gt(a[ia], a[ic])) swp(a, result, ic);
        else swp(a, result, ia);
    } else if (gt(a[ia], a[ic])) swp(a, result, ia);
    else if (gt(a[ib], a[ic])) swp(a, result, ic);
    else swp(a, result, ib);
}

template <class E, class GT> RB_FN int gnu_unguarded_partition(E *a, int first, int last, int pivot, GT gt)
{
    const E pv = a[pivot];                       // the pivot sits below `first` and is never moved by the loop
    while (true) {
        while (gt(a[first], pv)) ++first;
        --last;
        while (gt(pv, a[last])) --last;
        if (!(first < last)) return first;
        swp(a, first, last);
        ++first;
    }
}

template <class E, class GT> RB_FN void gnu_adjust_heap(E *a, int first, int hole, int len, E value, GT gt)
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (gt(a[first + child], a[first + child - 1])) --child;
        a[first + hole] = a[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a[first + hole] = a[first + child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;                 // __push_heap
    while (hole > top && gt(a[first + parent], value)) {
        a[first + hole] = a[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[first + hole] = value;
}

template <class E, class GT> RB_FN void gnu_heap_select(E *a, int first, int middle, int last, GT gt)
{
    const int len = middle - first;
    if (len >= 2) {                              // __make_heap
        int parent = (len - 2) / 2;
        while (true) {
            const E v = a[first + parent];
            gnu_adjust_heap(a, first, parent, len, v, gt);
            if (parent == 0) break;
            --parent;
        }
    }
    for (int i = middle; i < last; ++i)
        if (gt(a[i], a[first])) {                // __pop_heap(first, middle, i)
            const E v = a[i];
            a[i] = a[first];
            gnu_adjust_heap(a, first, 0, len, v, gt);
        }
}

template <class E, class GT> RB_FN void gnu_insertion_sort(E *a, int first, int last, GT gt)
{
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        const E v = a[i];
        if (gt(v, a[first])) {
            for (int k = i; k > first; --k) a[k] = a[k - 1];
            a[first] = v;
        } else {                                  // __unguarded_linear_insert
            int lastp = i, next = i - 1;
            while (gt(v, a[next])) { a[lastp] = a[next]; lastp = next; --next; }
            a[lastp] = v;
        }
    }
}

// std::nth_element(a + 0, a + nth, a + n, gt); depth_limit_override < 0: the library's 2 * floor(log2(n))
template <class E, class GT> RB_FN void gnu_nth_element(E *a, int n, int nth, GT gt, int depth_limit_override = -1)
{
    int first = 0, last = n;
    if (first == last || nth == last) return;
    int depth = 0;
    for (int m = n; m > 1; m >>= 1) ++depth;
    depth *= 2;
    if (depth_limit_override >= 0) depth = depth_limit_override;
    while (last - first > 3) {
        if (depth == 0) {
            gnu_heap_select(a, first, nth + 1, last, gt);
            swp(a, first, nth);
            return;
        }
        --depth;
        const int mid = first + (last - first) / 2;
        gnu_move_median_to_first(a, first, first + 1, mid, last - 1, gt);
        const int cut = gnu_unguarded_partition(a, first + 1, last, first, gt);
        if (cut <= nth) first = cut; else last = cut;
    }
    gnu_insertion_sort(a, first, last, gt);
}

// ------------------------------------------------------------------------------------------------ MSVC STL
template <class E, class GT> RB_FN void msvc_med3(E *a, int first, int mid, int last, GT gt)
{
    if (gt(a[mid], a[first])) swp(a, mid, first);
    if (gt(a[last], a[mid])) {
        swp(a, last, mid);
        if (gt(a[mid], a[first])) swp(a, mid, first);
    }
}

template <class E, class GT> RB_FN void msvc_guess_median(E *a, int first, int mid, int last, GT gt)
{
    const int count = last - first;
    if (40 < count) {                            // Tukey's ninther
        const int step = (count + 1) >> 3, two_step = step << 1;
        msvc_med3(a, first, first + step, first + two_step, gt);
        msvc_med3(a, mid - step, mid, mid + step, gt);
        msvc_med3(a, last - two_step, last - step, last, gt);
        msvc_med3(a, first + step, mid, last - step, gt);
    } else
        msvc_med3(a, first, mid, last, gt);
}

// returns the fat pivot [pfirst, plast) through the two references
template <class E, class GT> RB_FN void msvc_partition_by_median_guess(E *a, int first, int last, GT gt, int &out_first, int &out_last)
{
    const int mid = first + ((last - first) >> 1);
    msvc_guess_median(a, first, mid, last - 1, gt);
    int pfirst = mid, plast = pfirst + 1;
    while (first < pfirst && !gt(a[pfirst - 1], a[pfirst]) && !gt(a[pfirst], a[pfirst - 1])) --pfirst;
    while (plast < last && !gt(a[plast], a[pfirst]) && !gt(a[pfirst], a[plast])) ++plast;
    int gfirst = plast, glast = pfirst;
    for (;;) {
        for (; gfirst < last; ++gfirst) {
            if (gt(a[pfirst], a[gfirst])) continue;
            else if (gt(a[gfirst], a[pfirst])) break;
            else if (plast != gfirst) { swp(a, plast, gfirst); ++plast; }
            else ++plast;
        }
        for (; first < glast; --glast) {
            if (gt(a[glast - 1], a[pfirst])) continue;
            else if (gt(a[pfirst], a[glast - 1])) break;
            else if (--pfirst != glast - 1) swp(a, pfirst, glast - 1);
        }
        if (glast == first && gfirst == last) { out_first = pfirst; out_last = plast; return; }
        if (glast == first) {                    // no room at bottom, rotate pivot upward
            if (plast != gfirst) swp(a, pfirst, plast);
            ++plast;
            swp(a, pfirst, gfirst);
            ++pfirst; ++gfirst;
        } else if (gfirst == last) {             // no room at top, rotate pivot downward
            if (--glast != --pfirst) swp(a, glast, pfirst);
            swp(a, pfirst, --plast);
        } else {
            swp(a, gfirst, --glast);
            ++gfirst;
        }
    }
}

template <class E, class GT> RB_FN void msvc_insertion_sort(E *a, int first, int last, GT gt)
{
    if (first == last) return;
    for (int mid = first; ++mid != last;) {
        int hole = mid;
        const E v = a[mid];
        if (gt(v, a[first])) {
            for (int k = mid; k > first; --k) a[k] = a[k - 1];
            a[first] = v;
        } else {
            for (int prev = hole; gt(v, a[--prev]); hole = prev) a[hole] = a[prev];
            a[hole] = v;
        }
    }
}

template <class E, class GT> RB_FN void msvc_nth_element(E *a, int n, int nth, GT gt)
{
    int first = 0, last = n;
    if (nth == last) return;
    while (32 < last - first) {                  // _ISORT_MAX
        int pf, pl;
        msvc_partition_by_median_guess(a, first, last, gt, pf, pl);
        if (pl <= nth) first = pl;
        else if (pf <= nth) return;              // nth inside the fat pivot
        else last = pf;
    }
    msvc_insertion_sort(a, first, last, gt);
}

// ------------------------------------------------------------------------------------------------ retainBest
// std::partition(a + first, a + last, pred) for bidirectional iterators (same element moves in libstdc++ and MSVC)
template <class E, class P> RB_FN int partition_pred(E *a, int first, int last, P pred)
{
    while (true) {
        while (true) {
            if (first == last) return first;
            else if (pred(a[first])) ++first;
            else break;
        }
        --last;
        while (true) {
            if (first == last) return first;
            else if (!pred(a[last])) --last;
            else break;
        }
        swp(a, first, last);
        ++first;
    }
}

enum { RT_LIBSTDCXX = 0, RT_MSVC = 1 };

// KeyPointsFilter::retainBest(keypoints, n_points) on a[0..n): returns the new size; GE(x, y) = response(x) >= response(y)
template <class E, class GT, class GE> RB_FN int retain_best(E *a, int n, int n_points, int runtime, GT gt, GE ge)
{
    if (!(n_points >= 0 && n > n_points)) return n;
    if (n_points == 0) return 0;
    if (runtime == RT_MSVC) msvc_nth_element(a, n, n_points - 1, gt);
    else gnu_nth_element(a, n, n_points - 1, gt);
    const E amb = a[n_points - 1];               // the boundary response, ambiguous for FAST scores
    return partition_pred(a, n_points, n, [&](const E &x) { return ge(x, amb); });
}

}   // namespace rb
