// gsr_math.h -- per-gaussian device math for the preprocess kernels.
// Translation units that include this are built with -ffp-contract=off so that every +,* is one IEEE rounding,
// division and sqrt are correctly rounded (hipcc default): integer outputs derived from them (radii, tile rects,
// sort keys) are then reproducible bit-for-bit against a CPU evaluation in the same expression order.
// The expression ORDER follows the reference kernels (cited per function) because the order is part of the
// numerical contract; the code structure (plain structs, no glm) is ours.
#pragma once
#include <hip/hip_runtime.h>

namespace gsr {

struct M3 { float c[3][3]; };   // c[col][row]

__device__ __forceinline__ M3 m3_cols(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1, float c2)
{
    M3 m;
    m.c[0][0] = a0; m.c[0][1] = a1; m.c[0][2] = a2;
    m.c[1][0] = b0; m.c[1][1] = b1; m.c[1][2] = b2;
    m.c[2][0] = c0; m.c[2][1] = c1; m.c[2][2] = c2;
    return m;
}
__device__ __forceinline__ M3 m3_mul(const M3& a, const M3& b)
{
    M3 r;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++)
            r.c[j][i] = a.c[0][i] * b.c[j][0] + a.c[1][i] * b.c[j][1] + a.c[2][i] * b.c[j][2];
    return r;
}
__device__ __forceinline__ M3 m3_t(const M3& a)
{
    M3 r;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++) r.c[j][i] = a.c[i][j];
    return r;
}
__device__ __forceinline__ float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// row-vector convention transforms (3DGS auxiliary.h:58-100)
__device__ __forceinline__ float3 xform_point4x3(float3 p, const float* m)
{
    return make_float3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
                       m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
__device__ __forceinline__ float4 xform_point4x4(float3 p, const float* m)
{
    return make_float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
                       m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14],
                       m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
}
__device__ __forceinline__ float3 xform_vec4x3(float3 p, const float* m)
{
    return make_float3(m[0] * p.x + m[4] * p.y + m[8] * p.z,
                       m[1] * p.x + m[5] * p.y + m[9] * p.z,
                       m[2] * p.x + m[6] * p.y + m[10] * p.z);
}
__device__ __forceinline__ float3 xform_vec4x3_t(float3 p, const float* m)
{
    return make_float3(m[0] * p.x + m[1] * p.y + m[2] * p.z,
                       m[4] * p.x + m[5] * p.y + m[6] * p.z,
                       m[8] * p.x + m[9] * p.y + m[10] * p.z);
}
// 3DGS auxiliary.h:41-44: the literals there are double
__device__ __forceinline__ float ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// 3DGS auxiliary.h:46-56
__device__ __forceinline__ void tile_rect(float px, float py, int max_radius, int gx, int gy, int& x0, int& y0, int& x1, int& y1)
{
    x0 = min(gx, max(0, (int)((px - max_radius) / GSR_TILE)));
    y0 = min(gy, max(0, (int)((py - max_radius) / GSR_TILE)));
    x1 = min(gx, max(0, (int)((px + max_radius + GSR_TILE - 1) / GSR_TILE)));
    y1 = min(gy, max(0, (int)((py + max_radius + GSR_TILE - 1) / GSR_TILE)));
}

// Sigma = R S^2 R^T with the UN-normalised quaternion (3DGS forward.cu:118-152)
__device__ __forceinline__ void cov3d_from_scale_rot(const float* sc, float mod, const float* q, float* cov3D)
{
    M3 S = m3_cols(mod * sc[0], 0, 0, 0, mod * sc[1], 0, 0, 0, mod * sc[2]);
    float r = q[0], x = q[1], y = q[2], z = q[3];
    M3 R = m3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                   2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                   2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    M3 Mm = m3_mul(S, R);
    M3 Sg = m3_mul(m3_t(Mm), Mm);
    cov3D[0] = Sg.c[0][0]; cov3D[1] = Sg.c[0][1]; cov3D[2] = Sg.c[0][2];
    cov3D[3] = Sg.c[1][1]; cov3D[4] = Sg.c[1][2]; cov3D[5] = Sg.c[2][2];
}

struct Cov2D { float a, b, c; M3 T; M3 Vrk; float3 t; float xgm, ygm; };

// EWA projection of the 3D covariance (3DGS forward.cu:74-113; the backward at backward.cu:144-195 recomputes it)
__device__ __forceinline__ Cov2D cov2d_project(float3 mean, float fx, float fy, float tan_fovx, float tan_fovy,
                                              const float* cov3D, const float* view)
{
    Cov2D o;
    float3 t = xform_point4x3(mean, view);
    const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    const float txtz = t.x / t.z, tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
    o.xgm = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    o.ygm = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    M3 J = m3_cols(fx / t.z, 0.0f, -(fx * t.x) / (t.z * t.z), 0.0f, fy / t.z, -(fy * t.y) / (t.z * t.z), 0, 0, 0);
    M3 Wm = m3_cols(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    o.T = m3_mul(Wm, J);
    o.Vrk = m3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    M3 c2 = m3_mul(m3_mul(m3_t(o.T), m3_t(o.Vrk)), o.T);
    o.a = c2.c[0][0] + 0.3f;
    o.b = c2.c[0][1];
    o.c = c2.c[1][1] + 0.3f;
    o.t = t;
    return o;
}

// SH constants (3DGS auxiliary.h:21-39)
#define GSR_SH_C0 0.28209479177387814f
#define GSR_SH_C1 0.4886025119029199f
__device__ static const float GSR_SH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                                -1.0925484305920792f, 0.5462742152960396f };
__device__ static const float GSR_SH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                                0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                                -0.5900435899266435f };

// SH -> RGB (+0.5, clamp at 0 with per-channel flag), 3DGS forward.cu:20-71
__device__ __forceinline__ float3 sh_to_rgb(int deg, const float* sh /*[M,3]*/, float3 mean, float3 campos, uint32_t& clamped)
{
    float dx = mean.x - campos.x, dy = mean.y - campos.y, dz = mean.z - campos.z;
    float len = sqrtf(dx * dx + dy * dy + dz * dz);
    float x = dx / len, y = dy / len, z = dz / len;
    float out[3];
    clamped = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float r = GSR_SH_C0 * sh[0 * 3 + c];
        if (deg > 0) {
            r = r - GSR_SH_C1 * y * sh[1 * 3 + c] + GSR_SH_C1 * z * sh[2 * 3 + c] - GSR_SH_C1 * x * sh[3 * 3 + c];
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + GSR_SH_C2[0] * xy * sh[4 * 3 + c] + GSR_SH_C2[1] * yz * sh[5 * 3 + c] +
                    GSR_SH_C2[2] * (2.0f * zz - xx - yy) * sh[6 * 3 + c] + GSR_SH_C2[3] * xz * sh[7 * 3 + c] +
                    GSR_SH_C2[4] * (xx - yy) * sh[8 * 3 + c];
                if (deg > 2) {
                    r = r + GSR_SH_C3[0] * y * (3.0f * xx - yy) * sh[9 * 3 + c] + GSR_SH_C3[1] * xy * z * sh[10 * 3 + c] +
                        GSR_SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[11 * 3 + c] +
                        GSR_SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12 * 3 + c] +
                        GSR_SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[13 * 3 + c] +
                        GSR_SH_C3[5] * z * (xx - yy) * sh[14 * 3 + c] + GSR_SH_C3[6] * x * (xx - 3.0f * yy) * sh[15 * 3 + c];
                }
            }
        }
        r += 0.5f;
        if (r < 0) clamped |= (1u << c);
        out[c] = r > 0.0f ? r : 0.0f;
    }
    return make_float3(out[0], out[1], out[2]);
}

// normalised quaternion -> rotation, columns = local axes (SURFEL auxiliary.h:215-238).  1/sqrt, not v_rsq, so the
// result is correctly rounded and reproducible.
__device__ __forceinline__ M3 quat_to_rotmat(const float* q)
{
    float s = 1.0f / sqrtf(q[3] * q[3] + q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    float w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
    return m3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y + w * z), 2.f * (x * z - w * y),
                   2.f * (x * y - w * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z + w * x),
                   2.f * (x * z + w * y), 2.f * (y * z - w * x), 1.f - 2.f * (x * x + y * y));
}

// surfel splat->pixel homography rows Tu,Tv,Tw (SURFEL forward.cu:75-115); T[0..2]=Tu, T[3..5]=Tv, T[6..8]=Tw
__device__ __forceinline__ void surfel_transmat(float3 p, const float* sc2, float mod, const float* q, const float* proj,
                                                const float* view, int W, int H, float* T, float3& normal, M3& R)
{
    R = quat_to_rotmat(q);
    float sx = mod * sc2[0], sy = mod * sc2[1];
    float rows[3][4] = { { R.c[0][0] * sx, R.c[0][1] * sx, R.c[0][2] * sx, 0.0f },
                         { R.c[1][0] * sy, R.c[1][1] * sy, R.c[1][2] * sy, 0.0f },
                         { p.x, p.y, p.z, 1.0f } };
    float n00 = (float)((float)W / 2.0), n03 = (float)((float)(W - 1) / 2.0);
    float n11 = (float)((float)H / 2.0), n13 = (float)((float)(H - 1) / 2.0);
#pragma unroll
    for (int r = 0; r < 3; r++) {
        float h[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
            h[j] = rows[r][0] * proj[0 * 4 + j] + rows[r][1] * proj[1 * 4 + j] + rows[r][2] * proj[2 * 4 + j] + rows[r][3] * proj[3 * 4 + j];
        T[0 + r] = h[0] * n00 + h[1] * 0.0f + h[2] * 0.0f + h[3] * n03;
        T[3 + r] = h[0] * 0.0f + h[1] * n11 + h[2] * 0.0f + h[3] * n13;
        T[6 + r] = h[0] * 0.0f + h[1] * 0.0f + h[2] * 0.0f + h[3] * 1.0f;
    }
    normal = xform_vec4x3(make_float3(R.c[2][0], R.c[2][1], R.c[2][2]), view);
}

// bounding box of the projected c-sigma ellipse (SURFEL forward.cu:119-145)
__device__ __forceinline__ bool surfel_aabb(const float* T, float cutoff2, float& px, float& py, float& ex, float& ey, float* d_out = nullptr)
{
    const float* Tu = T; const float* Tv = T + 3; const float* Tw = T + 6;
    float t[3] = { cutoff2, cutoff2, -1.0f };
    float ww[3] = { Tw[0] * Tw[0], Tw[1] * Tw[1], Tw[2] * Tw[2] };
    float d = dot3(t, ww);
    if (d_out) *d_out = d;
    if (d == 0.0f) return false;
    float inv = 1 / d;
    float f[3] = { inv * t[0], inv * t[1], inv * t[2] };
    float uw[3] = { Tu[0] * Tw[0], Tu[1] * Tw[1], Tu[2] * Tw[2] };
    float vw[3] = { Tv[0] * Tw[0], Tv[1] * Tw[1], Tv[2] * Tw[2] };
    float uu[3] = { Tu[0] * Tu[0], Tu[1] * Tu[1], Tu[2] * Tu[2] };
    float vv[3] = { Tv[0] * Tv[0], Tv[1] * Tv[1], Tv[2] * Tv[2] };
    px = dot3(f, uw); py = dot3(f, vw);
    float h0x = px * px - dot3(f, uu), h0y = py * py - dot3(f, vv);
    ex = sqrtf(fmaxf(1e-4f, h0x)); ey = sqrtf(fmaxf(1e-4f, h0y));
    return true;
}

// 3DGS auxiliary.h:110-120
__device__ __forceinline__ float3 dnormvdv(float3 v, float3 dv)
{
    float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    float3 r;
    r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return r;
}

}  // namespace gsr
