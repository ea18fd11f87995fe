// Device-side helpers shared by the fused render kernels (gfx950 only).
// Geometry follows the reference's fp32 operation order exactly (the library is built with
// -ffp-contract=off); citations are into /root/reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/n3dt.h"

#define N3DT_HID 384
#define N3DT_C 256
#define N3DT_G 192       // RGB_layer_1 width (hidden/2)
#define N3DT_PE_ROWS 64  // 63 PE channels + one zero row
#define N3DT_PART_STRIDE (N3DT_G + 4)  // per (ray, sample-block) partial: G[192], wsum, depth, Tprod, pad

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct N3dtFrameCam {  // per-frame camera, folded into scalars once per wave
    float R[9], K[9], T[3];
};

// torch.linspace(0,1,steps) (CPU/CUDA kernels agree): symmetric evaluation around the midpoint.
// Call site: NetWorks/utils.py:140.
__device__ __forceinline__ float n3dt_linspace01(int i, int steps) {
    float step = 1.0f / (float)(steps - 1);
    int half = steps / 2;
    return (i < half) ? step * (float)i : 1.0f - step * (float)(steps - 1 - i);
}

// GenSamplePoints.forward, NetWorks/utils.py:149-155: d = normalize(R (Kinv [x,y,1])), l = -1/d_z
__device__ __forceinline__ void n3dt_ray_setup(const float* __restrict__ R, const float* __restrict__ K, float x, float y,
                                               float d[3], float& l) {
    float c[3], w[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) c[i] = K[i * 3 + 0] * x + K[i * 3 + 1] * y + K[i * 3 + 2] * 1.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) w[i] = R[i * 3 + 0] * c[0] + R[i * 3 + 1] * c[1] + R[i * 3 + 2] * c[2];
    float n = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    d[0] = w[0] / n;
    d[1] = w[1] / n;
    d[2] = w[2] / n;
    l = -1.0f / d[2];
}

// Edge j (0..Ns) of the sample planes; stratified jitter when tr != nullptr
// (NetWorks/utils.py:142 and :73-78).
__device__ __forceinline__ float n3dt_edge_z(float rz1, float rz2, int j, int Ns, const float* __restrict__ tr) {
    float t = n3dt_linspace01(j, Ns + 1);
    float zv = rz1 * (1.0f - t) + rz2 * t;
    if (!tr) return zv;
    float lower, upper;
    if (j == 0) {
        lower = zv;
    } else {
        float tp = n3dt_linspace01(j - 1, Ns + 1);
        float zp = rz1 * (1.0f - tp) + rz2 * tp;
        lower = 0.5f * (zv + zp);
    }
    if (j == Ns) {
        upper = zv;
    } else {
        float tn = n3dt_linspace01(j + 1, Ns + 1);
        float zn = rz1 * (1.0f - tn) + rz2 * tn;
        upper = 0.5f * (zn + zv);
    }
    return lower + (upper - lower) * tr[j];
}

// One sample point of one ray: position, plane distance and z value (utils.py:80-86).
// Lanes whose sample index is past N_s get a harmless point and dist = 0 (alpha = 0).
__device__ __forceinline__ void n3dt_sample_point(const N3dtGeom& g, const float* __restrict__ xy, const float* __restrict__ R,
                                                  const float* __restrict__ T, const float* __restrict__ Kinv,
                                                  const float* __restrict__ t_rand, int b, int ray, int s, float p[3],
                                                  float& dist, float& zval) {
    const float* Rb = R + b * 9;
    const float* Kb = Kinv + b * 9;
    const float* Tb = T + b * 3;
    float x = xy[(int64_t)b * g.xy_stride_b + 0 * g.xy_stride_c + (int64_t)ray * g.xy_stride_r];
    float y = xy[(int64_t)b * g.xy_stride_b + 1 * g.xy_stride_c + (int64_t)ray * g.xy_stride_r];
    float d[3], l;
    n3dt_ray_setup(Rb, Kb, x, y, d, l);
    float rz1 = Tb[2] - g.world_z1, rz2 = Tb[2] - g.world_z2;  // utils.py:125-126
    const int Ns = g.n_samples;
    if (s < Ns) {
        const float* tr = t_rand ? t_rand + ((int64_t)b * g.n_rays + ray) * (Ns + 1) : nullptr;
        float z_lo, z_hi;
        if (g.z_planes_given) {  // hierarchical pass: the planes come from n3dt_fine_sample (utils.py:173-183)
            z_lo = tr[s];
            z_hi = tr[s + 1];
        } else {
            z_lo = n3dt_edge_z(rz1, rz2, s, Ns, tr);
            z_hi = n3dt_edge_z(rz1, rz2, s + 1, Ns, tr);
        }
        dist = (z_hi - z_lo) * l;
        zval = z_lo;
#pragma unroll
        for (int i = 0; i < 3; ++i) p[i] = Tb[i] + (d[i] * l) * z_lo;
    } else {
        dist = 0.0f;
        zval = 0.0f;
        p[0] = p[1] = p[2] = 0.0f;
    }
}

// Positional-encoding channel `row` (0..63) of point p, accurate sin/cos (Embedder, utils.py:20-51):
// [p, sin(2^0 p), cos(2^0 p), ..., sin(2^9 p), cos(2^9 p)], row 63 = 0 padding.
__device__ __forceinline__ float n3dt_pe_row_accurate(const float p[3], int row) {
    if (row < 3) return row == 0 ? p[0] : (row == 1 ? p[1] : p[2]);
    if (row >= N3DT_PE_DIM) return 0.0f;
    int k = (row - 3) / 6, w = (row - 3) % 6;
    int dim = w % 3;
    float v = dim == 0 ? p[0] : (dim == 1 ? p[1] : p[2]);
    float a = v * (float)(1 << k);
    return w < 3 ? sinf(a) : cosf(a);
}

// Division of an element index by an extent that is a power of two in every shipped configuration (channel counts,
// map sizes): shift and mask when it is, the general division otherwise.  A runtime 64-bit division costs ~100 VALU
// operations -- per-element kernels that decompose their index with five of them are bound by that, not by memory.
struct N3dtDiv {
    unsigned d;
    int sh;  // log2(d) when d is a power of two, else -1
};
__device__ __forceinline__ N3dtDiv n3dt_div(const unsigned d) {
    N3dtDiv v;
    v.d = d;
    v.sh = (d != 0 && (d & (d - 1)) == 0) ? 31 - __builtin_clz(d) : -1;
    return v;
}
__device__ __forceinline__ size_t n3dt_quot(const size_t x, const N3dtDiv v) { return v.sh >= 0 ? x >> v.sh : x / v.d; }
__device__ __forceinline__ int n3dt_rem(const size_t x, const N3dtDiv v) { return v.sh >= 0 ? (int)(x & (v.d - 1)) : (int)(x % v.d); }

// rgb_upsample = bilinear x2 (align_corners=False) followed by the reflect-border [1,2,1]/4 blur (neural_renderer.py:54-55),
// along one axis, in closed form: output o of the 2n-long axis is a combination of the inputs base-1, base, base+1
// (base = o >> 1).  The blur reads the upsampled values at reflect(o-1), o, reflect(o+1); an even upsampled index 2m is
// 0.25 x[m-1] + 0.75 x[m], an odd one 2m+1 is 0.75 x[m] + 0.25 x[m+1], indices clamped to the axis -- all inside the window.
// Slots whose input index falls off the axis keep weight 0.  (Evaluating the nine bilinear samples separately costs 36 loads
// per output pixel; this is 9.)
__device__ __forceinline__ void n3dt_up_blur_w3(const int o, const int n, float (&w3)[3]) {
    if (n >= 3) {
        // The loop below in closed form (every weight is a dyadic fraction, so this is bit-identical): interior outputs take
        // [5, 10, 1] / 16 (even) or [1, 10, 5] / 16 (odd); only the two outputs at either end of the axis differ.  The loop costs
        // ~60 VALU operations per call and the fused blur kernel calls it four times per thread: 28 of its 125 us.
        const bool ev = (o & 1) == 0;
        const bool first = o == 0, second = o == 1, last = o == 2 * n - 1, last2 = o == 2 * n - 2;
        w3[0] = (first || second) ? 0.0f : (last ? 0.125f : (ev ? 0.3125f : 0.0625f));
        w3[1] = (first || last) ? 0.875f : ((second || last2) ? 0.6875f : 0.625f);
        w3[2] = (last || last2) ? 0.0f : (first ? 0.125f : (ev ? 0.0625f : 0.3125f));
        return;
    }
    const int base = o >> 1;
    w3[0] = w3[1] = w3[2] = 0.0f;
#pragma unroll
    for (int d = -1; d <= 1; ++d) {
        int i = o + d;
        i = i < 0 ? -i : (i >= 2 * n ? 4 * n - 2 - i : i);  // reflect on the 2n-long axis
        const float kd = d == 0 ? 0.5f : 0.25f;
        const int m = i >> 1;
        int a0, a1;
        float f0, f1;
        if ((i & 1) == 0) {
            a0 = max(m - 1, 0); f0 = 0.25f;
            a1 = m; f1 = 0.75f;
        } else {
            a0 = m; f0 = 0.75f;
            a1 = min(m + 1, n - 1); f1 = 0.25f;
        }
        const int s0 = a0 - base + 1, s1 = a1 - base + 1;
        w3[0] += (s0 == 0 ? kd * f0 : 0.0f) + (s1 == 0 ? kd * f1 : 0.0f);
        w3[1] += (s0 == 1 ? kd * f0 : 0.0f) + (s1 == 1 ? kd * f1 : 0.0f);
        w3[2] += (s0 == 2 ? kd * f0 : 0.0f) + (s1 == 2 ? kd * f1 : 0.0f);
    }
}

// exclusive prefix product over `width` consecutive lanes (width 16 or 32, power of two)
template <int WIDTH>
__device__ __forceinline__ float n3dt_exclusive_prod(float x, int lane_in_group) {
    float incl = x;
#pragma unroll
    for (int off = 1; off < WIDTH; off <<= 1) {
        float o = __shfl_up(incl, off, WIDTH);
        if (lane_in_group >= off) incl *= o;
    }
    float excl = __shfl_up(incl, 1, WIDTH);
    return lane_in_group == 0 ? 1.0f : excl;
}
