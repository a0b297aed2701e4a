/* saliency_oracle.c -- CPU restatement of cv2.saliency.StaticSaliencyFineGrained.computeSaliency(), the preprocessing the
 * reference applies to every frame when no_saliency=False (/root/reference/optical_flow/calculate_optical_flow.py:559-560
 * create, :586 computeSaliency; the map then takes the place of img2uint8(rgb2gray(frame)) as the solver's input).
 * OUTPUT TYPE: orc_saliency_fine_grained() is the 8-bit map the algorithm produces; orc_saliency_fine_grained_f32() (end of file) is
 * what computeSaliency() returns in opencv-contrib 4.x -- that map as CV_32F in [0,1] -- and is the DEFAULT hand-over of the
 * product path (DenseFlow.calc_study_saliency(map_dtype="f32")): DualTVL1 then multiplies by 255 in float (close to, not equal to,
 * the integers), DeepFlow takes [0,1] frames unscaled.  The 8-bit hand-over stays selectable (map_dtype="u8").
 *
 * TEST INFRASTRUCTURE ONLY: tests/ compare the HIP kernels (teeflow_saliency.hip.h) against this file; the product never
 * links or loads it.
 *
 * PARITY UNPINNED.  The algorithm lives in opencv-contrib (requirements.txt:7, `opencv-contrib-python>=4.5.0`), module
 * `saliency`, src/staticSaliencyFineGrained.cpp (Montabone & Soto, "Human detection using a mobile platform and novel
 * features derived from a visual saliency mechanism", IVC 2010).  Neither cv2 nor its sources are in /root/reference or in
 * this image and the reference holds no saliency fixture, so nothing here is checked against OpenCV: the steps below are
 * that file's as published, restated from its structure:
 *   calcIntensityChannel: cvtColor(BGR2GRAY) [the reference hands over RGB frames, so the "blue" weight lands on R],
 *     GaussianBlur 3x3 (sigma 0) twice, integral(CV_32F) once, six neighbourhoods {12,24,48,28,56,112}:
 *     getIntensityScaled (on = gray - mean of the surround, off = mean - gray, truncated to uchar when positive),
 *     mixScales (sum of the six maps, each sum scaled to 0..255 by its own maximum), mixOnOff ((on+off) scaled by
 *     max(max on, max off)).
 * Arithmetic types follow the C++ expressions: float for the integral image and the means, `255.` (double) in the two
 * scalings, (uchar) casts truncate.  8-bit building blocks are OpenCV 4.5+'s: BGR2GRAY = (c0*3735 + c1*19235 + c2*9798 +
 * 2^14) >> 15, GaussianBlur 3x3 on CV_8U = fixed point, (S + 8) >> 4 with S the 1-2-1 x 1-2-1 sum, BORDER_REFLECT_101,
 * integral: row prefix (exact) added to the row above, one float rounding per element.
 * Conversions C leaves undefined (NaN or out-of-range float -> uchar, reached only by flat images: 0/0) follow x86:
 * cvtt* to int32 (0x80000000 when not representable), low byte. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

static inline int refl101(int i, int n)
{
    if (n == 1) return 0;
    if (i < 0) return -i;
    if (i >= n) return 2 * n - 2 - i;
    return i;
}

static inline uint8_t u8_from_f64(double v)
{
    if (!(v == v) || v >= 2147483648.0 || v <= -2147483649.0) return 0;      /* cvttsd2si -> 0x80000000 -> low byte 0 */
    return (uint8_t)(uint32_t)(int32_t)v;
}

ORC_API void orc_sal_gray(const uint8_t* src, int H, int W, int channels, uint8_t* gray)
{
    const size_t n = (size_t)H * W;
    if (channels == 1) { memcpy(gray, src, n); return; }
    for (size_t i = 0; i < n; ++i) {
        const int c0 = src[i * 3], c1 = src[i * 3 + 1], c2 = src[i * 3 + 2];
        gray[i] = (uint8_t)((c0 * 3735 + c1 * 19235 + c2 * 9798 + (1 << 14)) >> 15);
    }
}

ORC_API void orc_sal_blur3(const uint8_t* src, int H, int W, uint8_t* dst)
{
    for (int y = 0; y < H; ++y) {
        const uint8_t* r0 = src + (size_t)refl101(y - 1, H) * W;
        const uint8_t* r1 = src + (size_t)y * W;
        const uint8_t* r2 = src + (size_t)refl101(y + 1, H) * W;
        for (int x = 0; x < W; ++x) {
            const int xl = refl101(x - 1, W), xr = refl101(x + 1, W);
            const int h0 = r0[xl] + 2 * r0[x] + r0[xr], h1 = r1[xl] + 2 * r1[x] + r1[xr], h2 = r2[xl] + 2 * r2[x] + r2[xr];
            dst[(size_t)y * W + x] = (uint8_t)((h0 + 2 * h1 + h2 + 8) >> 4);
        }
    }
}

/* integral(gray, sum, CV_32F): sum is (H+1) x (W+1) */
ORC_API void orc_sal_integral(const uint8_t* gray, int H, int W, float* sum)
{
    const int SW = W + 1;
    for (int x = 0; x < SW; ++x) sum[x] = 0.f;
    for (int y = 0; y < H; ++y) {
        float s = 0.f;
        sum[(size_t)(y + 1) * SW] = 0.f;
        for (int x = 0; x < W; ++x) {
            s += (float)gray[(size_t)y * W + x];
            sum[(size_t)(y + 1) * SW + x + 1] = sum[(size_t)y * SW + x + 1] + s;
        }
    }
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static inline float get_mean(const float* I, int SW, int SH, int x, int y, int nb, int center)
{
    const int x1 = clampi(x - nb + 1, 0, SW - 1), y1 = clampi(y - nb + 1, 0, SH - 1);
    const int x2 = clampi(x + nb + 1, 0, SW - 1), y2 = clampi(y + nb + 1, 0, SH - 1);
    float v = I[(size_t)y2 * SW + x2] + I[(size_t)y1 * SW + x1] - I[(size_t)y2 * SW + x1] - I[(size_t)y1 * SW + x2];
    v = (v - (float)center) / (float)((x2 - x1) * (y2 - y1) - 1);
    return v;
}

static const int k_neighbourhoods[6] = {3 * 4, 3 * 4 * 2, 3 * 4 * 2 * 2, 7 * 4, 7 * 4 * 2, 7 * 4 * 2 * 2};

/* src: H x W x channels (1 or 3) uint8, out: H x W uint8.  Returns 0, -1 on a bad argument / allocation failure. */
ORC_API int orc_saliency_fine_grained(const uint8_t* src, int H, int W, int channels, uint8_t* out)
{
    if (!src || !out || H < 1 || W < 1 || (channels != 1 && channels != 3)) return -1;
    const size_t n = (size_t)H * W;
    uint8_t* gray = (uint8_t*)malloc(n); uint8_t* tmp = (uint8_t*)malloc(n);
    float* I = (float*)malloc((size_t)(H + 1) * (W + 1) * sizeof(float));
    uint16_t* mon = (uint16_t*)calloc(n, sizeof(uint16_t)); uint16_t* moff = (uint16_t*)calloc(n, sizeof(uint16_t));
    uint8_t* ion = (uint8_t*)malloc(n); uint8_t* ioff = (uint8_t*)malloc(n);
    int rc = -1;
    if (gray && tmp && I && mon && moff && ion && ioff) {
        orc_sal_gray(src, H, W, channels, gray);
        orc_sal_blur3(gray, H, W, tmp);
        orc_sal_blur3(tmp, H, W, gray);
        orc_sal_integral(gray, H, W, I);
        for (int s = 0; s < 6; ++s)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    const int g = gray[(size_t)y * W + x];
                    const float value = get_mean(I, W + 1, H + 1, x, y, k_neighbourhoods[s], g);
                    const float on = (float)g - value, off = value - (float)g;
                    if (on > 0) mon[(size_t)y * W + x] += u8_from_f64((double)on);
                    if (off > 0) moff[(size_t)y * W + x] += u8_from_f64((double)off);
                }
        short max_on = 0, max_off = 0;
        for (size_t i = 0; i < n; ++i) {
            if ((short)mon[i] > max_on) max_on = (short)mon[i];
            if ((short)moff[i] > max_off) max_off = (short)moff[i];
        }
        int mx_on = 0, mx_off = 0;
        for (size_t i = 0; i < n; ++i) {
            ion[i] = u8_from_f64(255. * (double)(float)((float)mon[i] / (float)max_on));
            ioff[i] = u8_from_f64(255. * (double)(float)((float)moff[i] / (float)max_off));
            if (ion[i] > mx_on) mx_on = ion[i];
            if (ioff[i] > mx_off) mx_off = ioff[i];
        }
        const int mx = mx_on > mx_off ? mx_on : mx_off;
        for (size_t i = 0; i < n; ++i)
            out[i] = u8_from_f64(255. * (double)(float)(ion[i] + ioff[i]) / (double)(float)mx);
        rc = 0;
    }
    free(gray); free(tmp); free(I); free(mon); free(moff); free(ion); free(ioff);
    return rc;
}

/* What computeSaliency() hands back in opencv-contrib 4.x: computeSaliencyImpl ends with
 *     dst.convertTo(saliencyMap, CV_32F, 1.0f / 255.0f);      // values are in range [0; 1]
 * i.e. the map above times (1/255) in float, one rounding per pixel (cvtScale 8u -> 32f: (float)src * (float)alpha + 0).
 * [UPSTREAM-FROM-MEMORY like the rest of this file: the 3.x module returned the 8-bit map itself, which is why both forms exist;
 * the reference demands opencv-contrib >= 4.5.0 (requirements.txt:7), so THIS is the map its OF_model.calc receives (:586, :631).] */
ORC_API int orc_saliency_fine_grained_f32(const uint8_t* src, int H, int W, int channels, float* out)
{
    if (!out || H < 1 || W < 1) return -1;
    const size_t n = (size_t)H * W;
    uint8_t* m = (uint8_t*)malloc(n);
    if (!m) return -1;
    const int rc = orc_saliency_fine_grained(src, H, W, channels, m);
    if (rc == 0) for (size_t i = 0; i < n; ++i) out[i] = (float)m[i] * (1.0f / 255.0f);
    free(m);
    return rc;
}
