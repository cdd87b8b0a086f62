// The FM discriminator's arctangents (demod_fm.py:40-49: np.angle of the conj-lagged product), shared by the M = 1 chain kernels.
#pragma once
#include <hip/hip_runtime.h>

// full range: odd degree-15 minimax polynomial on [0, 1] + octant fix-up; atan2(0, 0) = 0 like np.angle(0)
__device__ __forceinline__ float dd_atan2_poly(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float t = mn * __builtin_amdgcn_rcpf(mx);
    const float z = t * t;
    float p = -4.054567120e-03f;
    p = fmaf(p, z, 2.186295773e-02f);
    p = fmaf(p, z, -5.591232695e-02f);
    p = fmaf(p, z, 9.642197381e-02f);
    p = fmaf(p, z, -1.390862959e-01f);
    p = fmaf(p, z, 1.994656567e-01f);
    p = fmaf(p, z, -3.332986079e-01f);
    p = fmaf(p, z, 9.999993356e-01f);
    float r = p * t;
    r = (mx == 0.f) ? 0.f : r;
    r = (ay > ax) ? 1.5707963267948966f - r : r;
    r = (x < 0.f) ? 3.141592653589793f - r : r;
    return copysignf(r, y);
}

// atan(y / x) for x > 0, |y| <= tan(pi/8) x (minimax fit, 2.3e-8 rad evaluated in f32), no octant logic
__device__ __forceinline__ float dd_atan_small(float y, float x) {
    const float t = y * __builtin_amdgcn_rcpf(x);
    const float z = t * t;
    float p = fmaf(7.902598251e-02f, z, -1.382445378e-01f);
    p = fmaf(p, z, 1.997187931e-01f);
    p = fmaf(p, z, -3.333275667e-01f);
    return fmaf(t, z * p, t);
}
