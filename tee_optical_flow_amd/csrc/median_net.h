// median_net.h -- min/max selection networks for the 3x3 and 5x5 float median (cv::medianBlur on
// CV_32F uses the same idea; the median VALUE is unique, so any correct network gives upstream's
// result).  Shared between the HIP kernels and tests/csrc/verify_median_net.c, which proves both
// networks with the zero-one principle (all 2^9 / 2^25 binary inputs).
#ifndef TEEFLOW_MEDIAN_NET_H
#define TEEFLOW_MEDIAN_NET_H

#include <type_traits>

#ifndef TF_HD
#define TF_HD
#endif

#define TF_CE(a, b) { const T lo_ = tf_min(p[a], p[b]); const T hi_ = tf_max(p[a], p[b]); p[a] = lo_; p[b] = hi_; }

// On the GPU a float min/max is ONE v_min_f32 / v_max_f32 instead of compare + select (no NaNs reach the median and
// +-0 compare equal downstream); the generic form serves the host-side zero-one proof.
template <typename T> TF_HD inline T tf_min(T a, T b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (std::is_same<T, float>::value) return __builtin_fminf(a, b);
#endif
    return b < a ? b : a;
}
template <typename T> TF_HD inline T tf_max(T a, T b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (std::is_same<T, float>::value) return __builtin_fmaxf(a, b);
#endif
    return b < a ? a : b;
}

// 19 compare-exchanges; median lands in p[4]
template <typename T> TF_HD inline T tf_median9(T* p)
{
    TF_CE(1, 2) TF_CE(4, 5) TF_CE(7, 8) TF_CE(0, 1) TF_CE(3, 4) TF_CE(6, 7)
    TF_CE(1, 2) TF_CE(4, 5) TF_CE(7, 8) TF_CE(0, 3) TF_CE(5, 8) TF_CE(4, 7)
    TF_CE(3, 6) TF_CE(1, 4) TF_CE(2, 5) TF_CE(4, 7) TF_CE(4, 2) TF_CE(6, 4)
    TF_CE(4, 2)
    return p[4];
}

// 99 compare-exchanges; median lands in p[12]
template <typename T> TF_HD inline T tf_median25(T* p)
{
    TF_CE(0, 1) TF_CE(3, 4) TF_CE(2, 4) TF_CE(2, 3) TF_CE(6, 7) TF_CE(5, 7) TF_CE(5, 6) TF_CE(9, 10)
    TF_CE(8, 10) TF_CE(8, 9) TF_CE(12, 13) TF_CE(11, 13) TF_CE(11, 12) TF_CE(15, 16) TF_CE(14, 16)
    TF_CE(14, 15) TF_CE(18, 19) TF_CE(17, 19) TF_CE(17, 18) TF_CE(21, 22) TF_CE(20, 22) TF_CE(20, 21)
    TF_CE(23, 24) TF_CE(2, 5) TF_CE(3, 6) TF_CE(0, 6) TF_CE(0, 3) TF_CE(4, 7) TF_CE(1, 7) TF_CE(1, 4)
    TF_CE(11, 14) TF_CE(8, 14) TF_CE(8, 11) TF_CE(12, 15) TF_CE(9, 15) TF_CE(9, 12) TF_CE(13, 16)
    TF_CE(10, 16) TF_CE(10, 13) TF_CE(20, 23) TF_CE(17, 23) TF_CE(17, 20) TF_CE(21, 24) TF_CE(18, 24)
    TF_CE(18, 21) TF_CE(19, 22) TF_CE(8, 17) TF_CE(9, 18) TF_CE(0, 18) TF_CE(0, 9) TF_CE(10, 19)
    TF_CE(1, 19) TF_CE(1, 10) TF_CE(11, 20) TF_CE(2, 20) TF_CE(2, 11) TF_CE(12, 21) TF_CE(3, 21)
    TF_CE(3, 12) TF_CE(13, 22) TF_CE(4, 22) TF_CE(4, 13) TF_CE(14, 23) TF_CE(5, 23) TF_CE(5, 14)
    TF_CE(15, 24) TF_CE(6, 24) TF_CE(6, 15) TF_CE(7, 16) TF_CE(7, 19) TF_CE(13, 21) TF_CE(15, 23)
    TF_CE(7, 13) TF_CE(7, 15) TF_CE(1, 9) TF_CE(3, 11) TF_CE(5, 17) TF_CE(11, 17) TF_CE(9, 17)
    TF_CE(4, 10) TF_CE(6, 12) TF_CE(7, 14) TF_CE(4, 6) TF_CE(4, 7) TF_CE(12, 14) TF_CE(10, 14)
    TF_CE(6, 7) TF_CE(10, 12) TF_CE(6, 10) TF_CE(6, 17) TF_CE(12, 17) TF_CE(7, 17) TF_CE(7, 10)
    TF_CE(12, 18) TF_CE(7, 12) TF_CE(10, 18) TF_CE(12, 20) TF_CE(10, 20) TF_CE(10, 12)
    return p[12];
}

#endif
