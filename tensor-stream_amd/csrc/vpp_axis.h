// vpp_axis.h -- source coordinate / weight of one output index along one axis, shared by the
// kernels and by the host (which uses them to recognise requests whose weights are all zero).
// Plain IEEE with explicit fused multiply-adds only where the reference's binary has them: host and device evaluate the
// same correctly rounded operations.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#pragma clang fp contract(off)

namespace tsvpp {

// Source coordinate + weight of one axis for BILINEAR (src/Resize.cu:276-303).  `(j + 0.5f) * xRatio - 0.5f` is ONE fused
// multiply-add in the reference as nvcc compiles it (fmad is on by default): pinned by the reference's CRC goldens
// (tests/test_reference_crcs.py; oracle/vpp_oracle.c, "contraction") -- the only floating-point contractions in this
// library are the ones those goldens demand.
__host__ __device__ inline void bilinear_axis(int idx, float ratio, int limit, int &p, float &w) {
    float f = __builtin_fmaf((float)idx + 0.5f, ratio, -0.5f);
    p = (int)floorf(f);
    w = f - (float)p;
    if (p < 0) { p = 0; w = 0.f; }
    if (p > limit - 1) { p = limit - 1; w = 0.f; }
}
// ... for the AREA up-scale variant (src/Resize.cu:221-234).
__host__ __device__ inline void areaup_axis(int idx, float ratio, int &p, float &w) {
    p = (int)floorf(ratio * (float)idx);
    float q = (float)(p + 1) / ratio;
    float f = (float)(idx + 1) - q;
    if (f <= 0.f) f = 0.f; else f = f - floorf(f);
    w = f;
}
// ... for BICUBIC (src/Resize.cu:321-347): fp32 coordinate widened to double.
__host__ __device__ inline void bicubic_axis(int idx, float ratio, int limit, int &p, double &w) {
    float ff = __builtin_fmaf((float)idx + 0.5f, ratio, -0.5f); // the same source expression as BILINEAR's: fused alike
    double f = (double)ff;
    p = (int)floor(f);
    w = f - (double)p;
    if (p < 0) { p = 0; w = 0.0; }
    if (p > limit - 1) { p = limit - 1; w = 0.0; }
}

// Source sample of a pure point sampler: NEAREST (src/Resize.cu:249-250), or BILINEAR / BICUBIC
// requests whose weights are zero for every output index (odd integer ratios: (j + .5) r - .5 is an
// integer), where the interpolation formulas reduce exactly to the centre tap.
enum PointKind : int { PK_NONE = -1, PK_NEAREST = 0, PK_BILINEAR0 = 1, PK_BICUBIC0 = 2 };
template <int KIND>
__host__ __device__ inline int point_coord(int idx, float ratio, int limit) {
    int p;
    if constexpr (KIND == PK_NEAREST) {
        p = (int)(ratio * (float)idx);
    } else if constexpr (KIND == PK_BILINEAR0) {
        float w;
        bilinear_axis(idx, ratio, limit, p, w);
    } else {
        double w;
        bicubic_axis(idx, ratio, limit, p, w);
    }
    return p;
}

} // namespace tsvpp
