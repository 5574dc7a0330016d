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

// ================================================================================================ wave-parallel form
// The two linear passes of the libstdc++ procedure -- __unguarded_partition and std::partition -- are Hoare-style
// two-pointer scans, and their net effect has a closed form: let A_1 < A_2 < ... be the positions (ascending) where the
// LEFT scan stops, B_1 > B_2 > ... the positions (descending) where the RIGHT scan stops; the k-th swap exchanges A_k and
// B_k, for as long as A_k < B_k.  (Each scan only ever looks at positions no swap has touched yet, so A and B can be read
// off the array as it is before the pass; A_k <= A_K < B_K <= B_k' keeps the swapped sets disjoint even though an element
// equal to the pivot stops both scans.)  With K = #{k : A_k < B_k}: the pass = those K swaps, and __unguarded_partition
// returns min(A_{K+1}, B_K) (the left scan rests on the next stopper of the untouched middle or on the element the last swap
// put at B_K), std::partition returns first + #{pred true}.  K follows from ranks alone: the left stopper at i with rank
// r is swapped iff at least r right stoppers lie beyond i.  That is a few ballot / popcount sweeps for a wave instead of a
// chain of dependent LDS round trips for one lane.
//
// pair_swap_model: the same arithmetic as the device routine below, lane by lane, on the host -- the unit test compares it
// with the sequential scans above (tests/test_retain_best_cpu.py), the GPU parity tests compare the device routine with the
// oracle.  LS(e): the left scan stops at e;  RS(e): the right scan stops at e.
template <class E, class LS, class RS> RB_FN int pair_swap_model(E *a, int lo, int hi, LS ls, RS rs, int *cut /* A_{K+1}, or hi */)
{
    const int m = hi - lo;
    int totalR = 0;
    for (int x = lo; x < hi; ++x) totalR += rs(a[x]) ? 1 : 0;
    // ranks from the ORIGINAL contents
    int K = 0, rankL = 0, seenR = 0, capK = m / 2 + 1;
    int *posA = new int[capK + 1], *posB = new int[capK + 1];
    for (int x = lo; x < hi; ++x) {
        const bool l = ls(a[x]), r = rs(a[x]);
        if (r) { ++seenR; const int rr = totalR - seenR + 1; if (rr <= capK) posB[rr - 1] = x; }       // rank from the right
        if (l) {
            ++rankL;
            if (rankL <= capK) posA[rankL - 1] = x;
            if (totalR - seenR >= rankL) ++K;                                                          // right stoppers strictly beyond x
        }
    }
    for (int k = 0; k < K; ++k) swp(a, posA[k], posB[k]);
    // where the left scan comes to rest: at the next left stopper of the untouched middle, A_{K+1}, or -- the scans cross
    // there -- on the element the last swap put at B_K, whichever comes first
    *cut = K < rankL && K < capK ? posA[K] : hi;
    if (K >= 1 && posB[K - 1] < *cut) *cut = posB[K - 1];
    delete[] posA; delete[] posB;
    return K;
}

#ifdef __HIPCC__
// One wave (all 64 lanes, uniform arguments); a[] in LDS; s_mask: 2 x ceil(m / 64) u64 of LDS; s_pos: 2 x (m / 2 + 1) u16 of LDS.
template <class E, class LS, class RS>
__device__ __forceinline__ int wave_pair_swap(E *a, int lo, int hi, LS ls, RS rs, unsigned long long *s_mask, unsigned short *s_pos, int &cut)
{
    const int lane = threadIdx.x & 63, m = hi - lo, nc = (m + 63) >> 6, capK = m / 2 + 1;
    unsigned long long *mL = s_mask, *mR = s_mask + nc;
    unsigned short *posA = s_pos, *posB = s_pos + capK;
    int totalR = 0, totalL = 0;
    for (int c = 0; c < nc; ++c) {
        const int x = lo + 64 * c + lane;
        bool l = false, r = false;
        if (x < hi) { const E e = a[x]; l = ls(e); r = rs(e); }
        const unsigned long long bl = __ballot(l), br = __ballot(r);
        if (lane == 0) { mL[c] = bl; mR[c] = br; }
        totalL += __popcll(bl); totalR += __popcll(br);
    }
    __syncthreads();
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;        // lanes < lane
    int K = 0, cumL = 0, cumR = 0;
    for (int c = 0; c < nc; ++c) {
        const unsigned long long bl = mL[c], br = mR[c];
        const int x = lo + 64 * c + lane;
        const bool l = (bl >> lane) & 1ull, r = (br >> lane) & 1ull;
        const int rl = cumL + __popcll(bl & below) + 1;                            // rank of this left stopper
        const int r_le = cumR + __popcll(br & below) + (r ? 1 : 0);                // right stoppers at positions <= x
        if (r) { const int rr = totalR - r_le + 1; if (rr <= capK) posB[rr - 1] = (unsigned short)x; }
        bool sw = false;
        if (l) { if (rl <= capK) posA[rl - 1] = (unsigned short)x; sw = (totalR - r_le) >= rl; }
        K += __popcll(__ballot(sw));
        cumL += __popcll(bl); cumR += __popcll(br);
    }
    __syncthreads();
    cut = (K < totalL && K < capK) ? (int)posA[K] : hi;
    if (K >= 1) cut = min(cut, (int)posB[K - 1]);              // see pair_swap_model
    for (int k = lane; k < K; k += 64) { const int i = posA[k], j = posB[k]; const E t = a[i]; a[i] = a[j]; a[j] = t; }
    __syncthreads();
    return K;
}

// KeyPointsFilter::retainBest on libstdc++, one wave: the partition passes through wave_pair_swap, the few-element steps
// (median of three, the <= 3-element insertion sort, the never-taken heap fallback) by lane 0.  Returns the new size (uniform).
template <class E, class GT, class GE>
__device__ __forceinline__ int wave_retain_best_gnu(E *a, int n, int n_points, GT gt, GE ge, unsigned long long *s_mask, unsigned short *s_pos, int *s_ctl)
{
    if (!(n_points >= 0 && n > n_points)) return n;
    if (n_points == 0) return 0;
    const int lane = threadIdx.x & 63, nth = n_points - 1;
    int first = 0, last = n, depth = 0;
    for (int m = n; m > 1; m >>= 1) ++depth;
    depth *= 2;
    while (last - first > 3) {
        if (depth == 0) {
            if (lane == 0) { gnu_heap_select(a, first, nth + 1, last, gt); swp(a, first, nth); }
            __syncthreads();
            first = last;                                      // done (the library returns here)
            break;
        }
        --depth;
        const int mid = first + (last - first) / 2;
        if (lane == 0) gnu_move_median_to_first(a, first, first + 1, mid, last - 1, gt);
        __syncthreads();
        const E pv = a[first];
        int cut;
        wave_pair_swap(a, first + 1, last, [&](const E &e) { return !gt(e, pv); }, [&](const E &e) { return !gt(pv, e); }, s_mask, s_pos, cut);
        if (cut <= nth) first = cut; else last = cut;
    }
    if (lane == 0 && last > first) gnu_insertion_sort(a, first, last, gt);
    __syncthreads();
    const E amb = a[nth];
    int cut2, cnt = 0;
    // std::partition(a + n_points, a + n, e >= amb): the left scan stops at !pred, the right scan at pred
    for (int x = n_points + lane; x < n; x += 64) cnt += ge(a[x], amb) ? 1 : 0;
    cnt = wave_sum(cnt);
    wave_pair_swap(a, n_points, n, [&](const E &e) { return !ge(e, amb); }, [&](const E &e) { return ge(e, amb); }, s_mask, s_pos, cut2);
    (void)s_ctl;
    return n_points + cnt;
}
#endif

}   // namespace rb
