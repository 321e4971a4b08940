/*
 * md_math.h -- deterministic float32 scalar math shared by the HIP kernels (device) and the CPU
 * oracle (host, gcc).
 *
 * Why this exists: the parity bar for this path is BIT-EXACT booleans (crash / out-of-road / done)
 * and tight float32 obs.  glibc's sinf/atan2f and the ROCm device library's do not round the same
 * way, so a boolean that sits next to a threshold would flip between the CPU checker and the GPU.
 * Every transcendental used on the path is therefore written here ONCE with +,-,*,/ and sqrt only
 * (IEEE-754 correctly rounded on both sides; both builds use -ffp-contract=off so no FMA fusion),
 * which makes the GPU result reproducible on the host to the last bit.  The oracle itself is pinned
 * against the reference's float64 Python results (tests/golden) with a stated tolerance, so an error
 * in this header is visible there.
 *
 * Replaces (reference, float64): math.sin/cos/atan2/acos and
 *   metadrive/utils/math.py:29-41  wrap_to_pi
 *   metadrive/utils/math.py:50-55  norm / clip
 *
 * Polynomials: classic Cephes single-precision kernels (public domain algorithm constants).
 */
#ifndef MD_MATH_H
#define MD_MATH_H

#if defined(__HIPCC__)
#define MD_HD __host__ __device__ static inline
#else
#define MD_HD static inline
#endif

#define MD_PI_F 3.14159265358979323846f
#define MD_TWO_PI_F 6.28318530717958647692f
#define MD_HALF_PI_F 1.57079632679489661923f
#define MD_QUARTER_PI_F 0.78539816339744830962f

MD_HD float md_fabs(float x) { return x < 0.0f ? -x : x; }
MD_HD float md_min(float a, float b) { return a < b ? a : b; }
MD_HD float md_max(float a, float b) { return a > b ? a : b; }
/* metadrive/utils/math.py:54-55  clip(a, low, high) = min(max(a, low), high) */
MD_HD float md_clip(float a, float lo, float hi) { return md_min(md_max(a, lo), hi); }

/* Correctly rounded on both sides: gcc -> sqrtss; hipcc -> IEEE sqrt expansion
 * (-fhip-fp32-correctly-rounded-divide-sqrt is the default).  NOT __fsqrt_rn: on ROCm 7.2 that is
 * __ocml_native_sqrt_f32, a ~1 ulp approximation (found by tests/test_gpu_math.py). */
MD_HD float md_sqrt(float x) { return __builtin_sqrtf(x); }

/* metadrive/utils/math.py:50-51  norm(x, y) */
MD_HD float md_norm(float x, float y) { return md_sqrt(x * x + y * y); }

MD_HD float md_floor(float x) {
    /* exact for |x| < 2^23; larger values are already integral */
    if (!(md_fabs(x) < 8388608.0f)) return x;
    float t = (float)(int)x;
    return (t > x) ? t - 1.0f : t;
}

/* sin & cos of x (radians), |x| up to ~1e4 keeps ~1e-7 abs error. Cephes sinf/cosf scheme. */
MD_HD void md_sincos(float xx, float* s_out, float* c_out) {
    const float FOPI = 1.27323954473516f; /* 4/pi */
    const float DP1 = 0.78515625f;
    const float DP2 = 2.4187564849853515625e-4f;
    const float DP3 = 3.77489497744594108e-8f;
    float x = xx;
    int sign_s = 1;
    if (x < 0.0f) {
        x = -x;
        sign_s = -1;
    }
    int j = (int)(FOPI * x);
    float y = (float)j;
    if (j & 1) {
        j += 1;
        y += 1.0f;
    }
    j &= 7;
    int sign_c = 1;
    if (j > 3) {
        sign_s = -sign_s;
        sign_c = -sign_c;
        j -= 4;
    }
    if (j > 1) sign_c = -sign_c;
    x = ((x - y * DP1) - y * DP2) - y * DP3;
    float z = x * x;
    float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * x + x;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.0f;
    float s, c;
    if (j == 1 || j == 2) {
        s = pc;
        c = ps;
    } else {
        s = ps;
        c = pc;
    }
    *s_out = (sign_s < 0) ? -s : s;
    *c_out = (sign_c < 0) ? -c : c;
}

MD_HD float md_atan(float xx) {
    float x = xx;
    int neg = 0;
    if (x < 0.0f) {
        neg = 1;
        x = -x;
    }
    float y;
    if (x > 2.414213562373095f) { /* tan(3pi/8) */
        y = MD_HALF_PI_F;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) { /* tan(pi/8) */
        y = MD_QUARTER_PI_F;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    float z = x * x;
    y += (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
    return neg ? -y : y;
}

/* atan2(y, x) with the usual quadrant rules; atan2(0,0) = 0 */
MD_HD float md_atan2(float y, float x) {
    if (x == 0.0f) {
        if (y > 0.0f) return MD_HALF_PI_F;
        if (y < 0.0f) return -MD_HALF_PI_F;
        return 0.0f;
    }
    if (y == 0.0f) return (x > 0.0f) ? 0.0f : MD_PI_F;
    float z = md_atan(y / x);
    if (x < 0.0f) z = (y < 0.0f) ? z - MD_PI_F : z + MD_PI_F;
    return z;
}

MD_HD float md_asin(float xx) {
    float x = xx;
    int neg = 0;
    if (x < 0.0f) {
        neg = 1;
        x = -x;
    }
    if (x > 1.0f) x = 1.0f;
    int flag = 0;
    float z;
    if (x > 0.5f) {
        z = 0.5f * (1.0f - x);
        x = md_sqrt(z);
        flag = 1;
    } else {
        z = x * x;
    }
    float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z
               + 1.6666752422e-1f) * z * x + x;
    if (flag) p = MD_HALF_PI_F - (p + p);
    return neg ? -p : p;
}

MD_HD float md_acos(float x) {
    if (x < -1.0f) x = -1.0f;
    if (x > 1.0f) x = 1.0f;
    if (x < -0.5f) return MD_PI_F - 2.0f * md_asin(md_sqrt(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * md_asin(md_sqrt(0.5f * (1.0f - x)));
    return MD_HALF_PI_F - md_asin(x);
}

/* metadrive/utils/math.py:29-41: wrap to (-pi, pi] via python-style modulo */
MD_HD float md_wrap_to_pi(float x) {
    /* multiply by 1/(2 pi) instead of an IEEE division (10+ dependent instructions on the GPU); an
     * off-by-one of the floor at an exact multiple only moves the result by one ulp-sized step
     * across the +-pi seam, which the two guards below absorb */
    float a = x - MD_TWO_PI_F * md_floor(x * 0.15915494309189535f);
    if (a < 0.0f) a += MD_TWO_PI_F; /* guard rounding */
    if (a >= MD_TWO_PI_F) a -= MD_TWO_PI_F;
    if (a > MD_PI_F) a -= MD_TWO_PI_F;
    return a;
}

/* exp(x) for modest |x| (energy model, metadrive base_vehicle.py:263) -- Cephes expf */
MD_HD float md_exp(float xx) {
    float x = xx;
    if (x > 88.0f) x = 88.0f;
    if (x < -88.0f) return 0.0f;
    const float LOG2EF = 1.44269504088896341f;
    const float C1 = 0.693359375f;
    const float C2 = -2.12194440e-4f;
    float z = md_floor(LOG2EF * x + 0.5f);
    x -= z * C1;
    x -= z * C2;
    int n = (int)z;
    z = x * x;
    z = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x
          + 1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
    /* scale by 2^n exactly */
    union { float f; unsigned int u; } sc;
    int e = n + 127;
    if (e < 1) return 0.0f;
    if (e > 254) e = 254;
    sc.u = ((unsigned int)e) << 23;
    return z * sc.f;
}

#endif /* MD_MATH_H */
