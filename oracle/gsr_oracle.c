/*
 * gsr_oracle.c -- plain-C CPU restatement of the GS-SR differentiable rasterizer hot path.
 * TEST INFRASTRUCTURE ONLY (see gsr_oracle.h).  float32 arithmetic throughout, built with
 * -ffp-contract=off so every +,* is a single IEEE rounding (the reference is built without --use_fast_math).
 *
 * Each function cites the reference file:line it follows.  Shorthand for the reference trees:
 *   3DGS   = submodules/diff-gaussian-rasterization/cuda_rasterizer
 *   SURFEL = submodules/diff-surfel-rasterization/cuda_rasterizer
 *   PLANE  = submodules/diff-plane-rasterization/cuda_rasterizer
 *   FILTER = submodules/scaffold-filter/cuda_rasterizer
 */
#include "gsr_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Arithmetic type.  The default build is float32 (the reference's type); -DGSR_REAL=double builds the SAME statements in float64
   (libgsr_oracle_f64.so), the truth the float32 evaluations -- this oracle's and the HIP library's -- are measured against.  Literals keep their
   float suffix, so both builds use bit-identical constants and thresholds. */
#define RS sizeof(real)
#if GSR_REAL_IS_DOUBLE
#define R_sqrt sqrt
#define R_exp exp
#define R_fmin fmin
#define R_fmax fmax
#define R_ceil ceil
#define R_fabs fabs
#define R_floor floor
#else
#define R_sqrt sqrtf
#define R_exp expf
#define R_fmin fminf
#define R_fmax fmaxf
#define R_ceil ceilf
#define R_fabs fabsf
#define R_floor floorf
#endif

#define BLOCK_X 16
#define BLOCK_Y 16

/* ------------------------------------------------------------------ small linear algebra */
/* m3 mimics a column-major 3x3 (element c[col][row]) so expressions can be restated index-for-index
   from sources that use that convention. */
typedef struct { real c[3][3]; } m3;
typedef struct { real x, y, z; } f3;
typedef struct { real x, y; } f2;

static m3 m3_cols(real a0, real a1, real a2, real b0, real b1, real b2, real c0, real c1, real c2)
{
    m3 m; m.c[0][0]=a0; m.c[0][1]=a1; m.c[0][2]=a2; m.c[1][0]=b0; m.c[1][1]=b1; m.c[1][2]=b2;
    m.c[2][0]=c0; m.c[2][1]=c1; m.c[2][2]=c2; return m;
}
/* r = a*b: r.col[j] = a * b.col[j];  element (col j,row i) = sum_k a[k][i]*b[j][k], evaluated k=0,1,2 */
static m3 m3_mul(m3 a, m3 b)
{
    m3 r;
    for (int j = 0; j < 3; j++)
        for (int i = 0; i < 3; i++)
            r.c[j][i] = a.c[0][i] * b.c[j][0] + a.c[1][i] * b.c[j][1] + a.c[2][i] * b.c[j][2];
    return r;
}
static m3 m3_t(m3 a)
{
    m3 r;
    for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) r.c[j][i] = a.c[i][j];
    return r;
}
static real dot3(const real* a, const real* b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }

/* 3DGS auxiliary.h:58-100 */
static f3 transformPoint4x3(f3 p, const real* m)
{
    f3 t = { m[0]*p.x + m[4]*p.y + m[8]*p.z + m[12],
             m[1]*p.x + m[5]*p.y + m[9]*p.z + m[13],
             m[2]*p.x + m[6]*p.y + m[10]*p.z + m[14] };
    return t;
}
static void transformPoint4x4(f3 p, const real* m, real o[4])
{
    o[0] = m[0]*p.x + m[4]*p.y + m[8]*p.z + m[12];
    o[1] = m[1]*p.x + m[5]*p.y + m[9]*p.z + m[13];
    o[2] = m[2]*p.x + m[6]*p.y + m[10]*p.z + m[14];
    o[3] = m[3]*p.x + m[7]*p.y + m[11]*p.z + m[15];
}
static f3 transformVec4x3(f3 p, const real* m)
{
    f3 t = { m[0]*p.x + m[4]*p.y + m[8]*p.z,
             m[1]*p.x + m[5]*p.y + m[9]*p.z,
             m[2]*p.x + m[6]*p.y + m[10]*p.z };
    return t;
}
static f3 transformVec4x3Transpose(f3 p, const real* m)
{
    f3 t = { m[0]*p.x + m[1]*p.y + m[2]*p.z,
             m[4]*p.x + m[5]*p.y + m[6]*p.z,
             m[8]*p.x + m[9]*p.y + m[10]*p.z };
    return t;
}
/* 3DGS auxiliary.h:41-44 -- NB the literals are double, so this is evaluated in double */
static real ndc2Pix(real v, int S) { return (real)(((v + 1.0) * S - 1.0) * 0.5); }

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* 3DGS auxiliary.h:46-56 */
static void getRect(f2 p, int max_radius, int gx, int gy, uint32_t rmin[2], uint32_t rmax[2])
{
    rmin[0] = (uint32_t)imin(gx, imax(0, (int)((p.x - max_radius) / BLOCK_X)));
    rmin[1] = (uint32_t)imin(gy, imax(0, (int)((p.y - max_radius) / BLOCK_Y)));
    rmax[0] = (uint32_t)imin(gx, imax(0, (int)((p.x + max_radius + BLOCK_X - 1) / BLOCK_X)));
    rmax[1] = (uint32_t)imin(gy, imax(0, (int)((p.y + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

/* 3DGS auxiliary.h:139-164 (the prefiltered trap is not restated: it aborts the reference) */
static int in_frustum(int idx, const real* pts, const real* view, f3* p_view)
{
    f3 p = { pts[3*idx], pts[3*idx+1], pts[3*idx+2] };
    *p_view = transformPoint4x3(p, view);
    return !(p_view->z <= 0.2f);
}

/* ------------------------------------------------------------------ spherical harmonics */
static const real SH_C0 = 0.28209479177387814f;
static const real SH_C1 = 0.4886025119029199f;
static const real SH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                -1.0925484305920792f, 0.5462742152960396f };
static const real SH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                -0.5900435899266435f };

/* 3DGS forward.cu:20-71 */
static void computeColorFromSH(int idx, int deg, int M, const real* means, const real* campos,
                               const real* shs, uint8_t* clamped, real rgb[3])
{
    real dx = means[3*idx] - campos[0], dy = means[3*idx+1] - campos[1], dz = means[3*idx+2] - campos[2];
    real len = R_sqrt(dx*dx + dy*dy + dz*dz);
    real x = dx / len, y = dy / len, z = dz / len;
    const real* sh = shs + (size_t)idx * M * 3;
    for (int c = 0; c < 3; c++) {
        real r = SH_C0 * sh[0*3+c];
        if (deg > 0) {
            r = r - SH_C1 * y * sh[1*3+c] + SH_C1 * z * sh[2*3+c] - SH_C1 * x * sh[3*3+c];
            if (deg > 1) {
                real xx = x*x, yy = y*y, zz = z*z, xy = x*y, yz = y*z, xz = x*z;
                r = r + SH_C2[0] * xy * sh[4*3+c] + SH_C2[1] * yz * sh[5*3+c]
                      + SH_C2[2] * (2.0f*zz - xx - yy) * sh[6*3+c]
                      + SH_C2[3] * xz * sh[7*3+c] + SH_C2[4] * (xx - yy) * sh[8*3+c];
                if (deg > 2) {
                    r = r + SH_C3[0] * y * (3.0f*xx - yy) * sh[9*3+c]
                          + SH_C3[1] * xy * z * sh[10*3+c]
                          + SH_C3[2] * y * (4.0f*zz - xx - yy) * sh[11*3+c]
                          + SH_C3[3] * z * (2.0f*zz - 3.0f*xx - 3.0f*yy) * sh[12*3+c]
                          + SH_C3[4] * x * (4.0f*zz - xx - yy) * sh[13*3+c]
                          + SH_C3[5] * z * (xx - yy) * sh[14*3+c]
                          + SH_C3[6] * x * (xx - 3.0f*yy) * sh[15*3+c];
                }
            }
        }
        r += 0.5f;
        clamped[3*idx + c] = (r < 0);
        rgb[c] = r > 0.0f ? r : 0.0f;
    }
}

/* 3DGS auxiliary.h:110-120 */
static f3 dnormvdv(f3 v, f3 dv)
{
    real sum2 = v.x*v.x + v.y*v.y + v.z*v.z;
    real invsum32 = 1.0f / R_sqrt(sum2 * sum2 * sum2);
    f3 r;
    r.x = ((+sum2 - v.x*v.x) * dv.x - v.y*v.x*dv.y - v.z*v.x*dv.z) * invsum32;
    r.y = (-v.x*v.y*dv.x + (sum2 - v.y*v.y) * dv.y - v.z*v.y*dv.z) * invsum32;
    r.z = (-v.x*v.z*dv.x - v.y*v.z*dv.y + (sum2 - v.z*v.z) * dv.z) * invsum32;
    return r;
}

/* 3DGS backward.cu:20-139 */
static void computeColorFromSH_bwd(int idx, int deg, int M, const real* means, const real* campos,
                                   const real* shs, const uint8_t* clamped, const real* dL_dcolor,
                                   real* dL_dmeans, real* dL_dshs)
{
    f3 dir_orig = { means[3*idx] - campos[0], means[3*idx+1] - campos[1], means[3*idx+2] - campos[2] };
    real len = R_sqrt(dir_orig.x*dir_orig.x + dir_orig.y*dir_orig.y + dir_orig.z*dir_orig.z);
    real x = dir_orig.x / len, y = dir_orig.y / len, z = dir_orig.z / len;
    const real* sh = shs + (size_t)idx * M * 3;
    real* dL_dsh = dL_dshs + (size_t)idx * M * 3;
    real dRGB[3];
    for (int c = 0; c < 3; c++) dRGB[c] = dL_dcolor[3*idx+c] * (clamped[3*idx+c] ? 0.0f : 1.0f);

    real dRGBdx[3] = {0,0,0}, dRGBdy[3] = {0,0,0}, dRGBdz[3] = {0,0,0};
#define SHSET(k, coef) do { real cf_ = (coef); for (int c = 0; c < 3; c++) dL_dsh[(k)*3+c] = cf_ * dRGB[c]; } while (0)
    SHSET(0, SH_C0);
    if (deg > 0) {
        SHSET(1, -SH_C1 * y); SHSET(2, SH_C1 * z); SHSET(3, -SH_C1 * x);
        for (int c = 0; c < 3; c++) {
            dRGBdx[c] = -SH_C1 * sh[3*3+c]; dRGBdy[c] = -SH_C1 * sh[1*3+c]; dRGBdz[c] = SH_C1 * sh[2*3+c];
        }
        if (deg > 1) {
            real xx = x*x, yy = y*y, zz = z*z, xy = x*y, yz = y*z, xz = x*z;
            SHSET(4, SH_C2[0] * xy); SHSET(5, SH_C2[1] * yz); SHSET(6, SH_C2[2] * (2.f*zz - xx - yy));
            SHSET(7, SH_C2[3] * xz); SHSET(8, SH_C2[4] * (xx - yy));
            for (int c = 0; c < 3; c++) {
                dRGBdx[c] += SH_C2[0] * y * sh[4*3+c] + SH_C2[2] * 2.f * -x * sh[6*3+c] + SH_C2[3] * z * sh[7*3+c] + SH_C2[4] * 2.f * x * sh[8*3+c];
                dRGBdy[c] += SH_C2[0] * x * sh[4*3+c] + SH_C2[1] * z * sh[5*3+c] + SH_C2[2] * 2.f * -y * sh[6*3+c] + SH_C2[4] * 2.f * -y * sh[8*3+c];
                dRGBdz[c] += SH_C2[1] * y * sh[5*3+c] + SH_C2[2] * 2.f * 2.f * z * sh[6*3+c] + SH_C2[3] * x * sh[7*3+c];
            }
            if (deg > 2) {
                SHSET(9,  SH_C3[0] * y * (3.f*xx - yy));
                SHSET(10, SH_C3[1] * xy * z);
                SHSET(11, SH_C3[2] * y * (4.f*zz - xx - yy));
                SHSET(12, SH_C3[3] * z * (2.f*zz - 3.f*xx - 3.f*yy));
                SHSET(13, SH_C3[4] * x * (4.f*zz - xx - yy));
                SHSET(14, SH_C3[5] * z * (xx - yy));
                SHSET(15, SH_C3[6] * x * (xx - 3.f*yy));
                for (int c = 0; c < 3; c++) {
                    dRGBdx[c] += (SH_C3[0] * sh[9*3+c] * 3.f * 2.f * xy + SH_C3[1] * sh[10*3+c] * yz
                                + SH_C3[2] * sh[11*3+c] * -2.f * xy + SH_C3[3] * sh[12*3+c] * -3.f * 2.f * xz
                                + SH_C3[4] * sh[13*3+c] * (-3.f*xx + 4.f*zz - yy) + SH_C3[5] * sh[14*3+c] * 2.f * xz
                                + SH_C3[6] * sh[15*3+c] * 3.f * (xx - yy));
                    dRGBdy[c] += (SH_C3[0] * sh[9*3+c] * 3.f * (xx - yy) + SH_C3[1] * sh[10*3+c] * xz
                                + SH_C3[2] * sh[11*3+c] * (-3.f*yy + 4.f*zz - xx) + SH_C3[3] * sh[12*3+c] * -3.f * 2.f * yz
                                + SH_C3[4] * sh[13*3+c] * -2.f * xy + SH_C3[5] * sh[14*3+c] * -2.f * yz
                                + SH_C3[6] * sh[15*3+c] * -3.f * 2.f * xy);
                    dRGBdz[c] += (SH_C3[1] * sh[10*3+c] * xy + SH_C3[2] * sh[11*3+c] * 4.f * 2.f * yz
                                + SH_C3[3] * sh[12*3+c] * 3.f * (2.f*zz - xx - yy) + SH_C3[4] * sh[13*3+c] * 4.f * 2.f * xz
                                + SH_C3[5] * sh[14*3+c] * (xx - yy));
                }
            }
        }
    }
#undef SHSET
    f3 dL_ddir = { dot3(dRGBdx, dRGB), dot3(dRGBdy, dRGB), dot3(dRGBdz, dRGB) };
    f3 dm = dnormvdv(dir_orig, dL_ddir);
    dL_dmeans[3*idx+0] += dm.x; dL_dmeans[3*idx+1] += dm.y; dL_dmeans[3*idx+2] += dm.z;
}

/* ------------------------------------------------------------------ EWA preprocess helpers */
/* 3DGS forward.cu:118-152 (quaternion deliberately NOT normalised) */
static void computeCov3D(const real* scale, real mod, const real* rot, real* cov3D)
{
    m3 S = m3_cols(mod*scale[0],0,0, 0,mod*scale[1],0, 0,0,mod*scale[2]);
    real r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    m3 R = m3_cols(1.f - 2.f*(y*y + z*z), 2.f*(x*y - r*z), 2.f*(x*z + r*y),
                   2.f*(x*y + r*z), 1.f - 2.f*(x*x + z*z), 2.f*(y*z - r*x),
                   2.f*(x*z - r*y), 2.f*(y*z + r*x), 1.f - 2.f*(x*x + y*y));
    m3 Mm = m3_mul(S, R);
    m3 Sigma = m3_mul(m3_t(Mm), Mm);
    cov3D[0] = Sigma.c[0][0]; cov3D[1] = Sigma.c[0][1]; cov3D[2] = Sigma.c[0][2];
    cov3D[3] = Sigma.c[1][1]; cov3D[4] = Sigma.c[1][2]; cov3D[5] = Sigma.c[2][2];
}

/* 3DGS forward.cu:74-113; also returns T for the backward (backward.cu:144-195 recomputes the same) */
static void computeCov2D(f3 mean, real fx, real fy, real tan_fovx, real tan_fovy, const real* cov3D,
                         const real* view, real cov[3], m3* T_out, m3* Vrk_out, f3* t_out,
                         real* xgm, real* ygm)
{
    f3 t = transformPoint4x3(mean, view);
    const real limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    const real txtz = t.x / t.z, tytz = t.y / t.z;
    t.x = R_fmin(limx, R_fmax(-limx, txtz)) * t.z;
    t.y = R_fmin(limy, R_fmax(-limy, tytz)) * t.z;
    if (xgm) *xgm = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    if (ygm) *ygm = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    m3 J = m3_cols(fx / t.z, 0.0f, -(fx * t.x) / (t.z * t.z),
                   0.0f, fy / t.z, -(fy * t.y) / (t.z * t.z),
                   0, 0, 0);
    m3 Wm = m3_cols(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    m3 T = m3_mul(Wm, J);
    m3 Vrk = m3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    m3 c2 = m3_mul(m3_mul(m3_t(T), m3_t(Vrk)), T);
    cov[0] = c2.c[0][0] + 0.3f;   /* low-pass: >= one pixel */
    cov[1] = c2.c[0][1];
    cov[2] = c2.c[1][1] + 0.3f;
    if (T_out) *T_out = T;
    if (Vrk_out) *Vrk_out = Vrk;
    if (t_out) *t_out = t;
}

/* ------------------------------------------------------------------ surfel preprocess helpers */
/* SURFEL auxiliary.h:215-238 */
static m3 quat_to_rotmat(const real* q)
{
    real s = 1.0f / R_sqrt(q[3]*q[3] + q[0]*q[0] + q[1]*q[1] + q[2]*q[2]);
    real w = q[0]*s, x = q[1]*s, y = q[2]*s, z = q[3]*s;
    return m3_cols(1.f - 2.f*(y*y + z*z), 2.f*(x*y + w*z), 2.f*(x*z - w*y),
                   2.f*(x*y - w*z), 1.f - 2.f*(x*x + z*z), 2.f*(y*z + w*x),
                   2.f*(x*z + w*y), 2.f*(y*z - w*x), 1.f - 2.f*(x*x + y*y));
}
/* SURFEL auxiliary.h:241-284; v_R indexed [col][row] */
static void quat_to_rotmat_vjp(const real* q, m3 v_R, real out[4])
{
    real s = 1.0f / R_sqrt(q[3]*q[3] + q[0]*q[0] + q[1]*q[1] + q[2]*q[2]);
    real w = q[0]*s, x = q[1]*s, y = q[2]*s, z = q[3]*s;
    out[0] = 2.f * (x * (v_R.c[1][2] - v_R.c[2][1]) + y * (v_R.c[2][0] - v_R.c[0][2]) + z * (v_R.c[0][1] - v_R.c[1][0]));
    out[1] = 2.f * (-2.f * x * (v_R.c[1][1] + v_R.c[2][2]) + y * (v_R.c[0][1] + v_R.c[1][0]) +
                    z * (v_R.c[0][2] + v_R.c[2][0]) + w * (v_R.c[1][2] - v_R.c[2][1]));
    out[2] = 2.f * (x * (v_R.c[0][1] + v_R.c[1][0]) - 2.f * y * (v_R.c[0][0] + v_R.c[2][2]) +
                    z * (v_R.c[1][2] + v_R.c[2][1]) + w * (v_R.c[2][0] - v_R.c[0][2]));
    out[3] = 2.f * (x * (v_R.c[0][2] + v_R.c[2][0]) + y * (v_R.c[1][2] + v_R.c[2][1]) -
                    2.f * z * (v_R.c[0][0] + v_R.c[1][1]) + w * (v_R.c[0][1] - v_R.c[1][0]));
}

/* Pm[k][c] = (world2ndc * ndc2pix) as a 4x3 matrix: column c of the pixel-space homogeneous coordinate.
   SURFEL forward.cu:99-112 / backward.cu:498-513 */
static void surfel_P(const real* proj, int W, int H, real Pm[4][3])
{
    real n00 = (real)((real)W / 2.0), n03 = (real)((real)(W - 1) / 2.0);
    real n11 = (real)((real)H / 2.0), n13 = (real)((real)(H - 1) / 2.0);
    for (int k = 0; k < 4; k++) {
        /* world2ndc element (row k, col j) = proj[4k + j] */
        real a0 = proj[4*k+0], a1 = proj[4*k+1], a3 = proj[4*k+3];
        real a2 = proj[4*k+2];
        Pm[k][0] = a0 * n00 + a1 * 0.0f + a2 * 0.0f + a3 * n03;
        Pm[k][1] = a0 * 0.0f + a1 * n11 + a2 * 0.0f + a3 * n13;
        Pm[k][2] = a0 * 0.0f + a1 * 0.0f + a2 * 0.0f + a3 * 1.0f;
    }
}

/* SURFEL forward.cu:75-115.  T stored as three float3: Tu (x coeffs of u,v,1), Tv (y coeffs), Tw (w coeffs) */
static void compute_transmat(f3 p, const real* scale2, real mod, const real* rot, const real* proj,
                             const real* view, int W, int H, real T[9], f3* normal)
{
    m3 R = quat_to_rotmat(rot);
    real sx = mod * scale2[0], sy = mod * scale2[1];
    real L0[3] = { R.c[0][0]*sx, R.c[0][1]*sx, R.c[0][2]*sx };
    real L1[3] = { R.c[1][0]*sy, R.c[1][1]*sy, R.c[1][2]*sy };
    real L2[3] = { R.c[2][0], R.c[2][1], R.c[2][2] };
    real rows[3][4] = { { L0[0], L0[1], L0[2], 0.0f }, { L1[0], L1[1], L1[2], 0.0f }, { p.x, p.y, p.z, 1.0f } };
    /* (splat2world^T * world2ndc) first, then * ndc2pix -- left-to-right like the source expression */
    real n00 = (real)((real)W / 2.0), n03 = (real)((real)(W - 1) / 2.0);
    real n11 = (real)((real)H / 2.0), n13 = (real)((real)(H - 1) / 2.0);
    for (int r = 0; r < 3; r++) {
        real h[4];
        for (int j = 0; j < 4; j++)
            h[j] = rows[r][0]*proj[0*4+j] + rows[r][1]*proj[1*4+j] + rows[r][2]*proj[2*4+j] + rows[r][3]*proj[3*4+j];
        real tx = h[0]*n00 + h[1]*0.0f + h[2]*0.0f + h[3]*n03;
        real ty = h[0]*0.0f + h[1]*n11 + h[2]*0.0f + h[3]*n13;
        real tw = h[0]*0.0f + h[1]*0.0f + h[2]*0.0f + h[3]*1.0f;
        T[0 + r] = tx; T[3 + r] = ty; T[6 + r] = tw;
    }
    f3 l2 = { L2[0], L2[1], L2[2] };
    *normal = transformVec4x3(l2, view);
}

/* SURFEL forward.cu:119-145 */
static int compute_aabb(const real T[9], real cutoff, f2* point_image, f2* extent)
{
    const real* Tu = T; const real* Tv = T + 3; const real* Tw = T + 6;
    real t[3] = { cutoff*cutoff, cutoff*cutoff, -1.0f };
    real ww[3] = { Tw[0]*Tw[0], Tw[1]*Tw[1], Tw[2]*Tw[2] };
    real d = dot3(t, ww);
    if (d == 0.0f) return 0;
    real inv = 1 / d;
    real f[3] = { inv*t[0], inv*t[1], inv*t[2] };
    real uw[3] = { Tu[0]*Tw[0], Tu[1]*Tw[1], Tu[2]*Tw[2] };
    real vw[3] = { Tv[0]*Tw[0], Tv[1]*Tw[1], Tv[2]*Tw[2] };
    real uu[3] = { Tu[0]*Tu[0], Tu[1]*Tu[1], Tu[2]*Tu[2] };
    real vv[3] = { Tv[0]*Tv[0], Tv[1]*Tv[1], Tv[2]*Tv[2] };
    real px = dot3(f, uw), py = dot3(f, vw);
    real h0x = px*px - dot3(f, uu), h0y = py*py - dot3(f, vv);
    point_image->x = px; point_image->y = py;
    extent->x = R_sqrt(R_fmax(1e-4f, h0x)); extent->y = R_sqrt(R_fmax(1e-4f, h0y));
    return 1;
}

/* every stride-th tile only in the blend loops (forward and backward): the bounded 1-thread sample of bench.py's cpu_baseline */
static int g_tile_stride = 1;

/* ------------------------------------------------------------------ state */
struct ref_state {
    int variant, P, W, H, gx, gy, T, N, R;
    /* geometry state (3DGS rasterizer_impl.cu:155-170, SURFEL :162-163) */
    real* depths; uint8_t* clamped; real* means2D; real* cov3D /* or transMat (9) */;
    real* conic_opacity /* or normal_opacity */; real* rgb; uint32_t* tiles_touched; uint32_t* point_offsets;
    int32_t* radii;
    /* binning */
    uint64_t* keys; uint32_t* point_list;
    /* image */
    real* final_T; uint32_t* n_contrib; uint32_t* ranges;
    real* out_all_map; /* PLANE: kept for backward (all_map_pixels) */
    uint64_t pairs_evaluated, pairs_contributing;   /* workload statistics of the blend forward ("contributing pairs C") */
    real* kappa;       /* truth runs: conditioning of the per-gaussian quantities (EWA/PLANE: (ac + b^2)/det of the 2D covariance) */
};

static void* xcalloc(size_t n, size_t s) { void* p = calloc(n ? n : 1, s); if (!p) { fprintf(stderr, "oracle: OOM\n"); abort(); } return p; }

void ref_free(ref_state* st)
{
    if (!st) return;
    free(st->depths); free(st->clamped); free(st->means2D); free(st->cov3D); free(st->conic_opacity);
    free(st->rgb); free(st->tiles_touched); free(st->point_offsets); free(st->radii);
    free(st->keys); free(st->point_list); free(st->final_T); free(st->n_contrib); free(st->ranges);
    free(st->out_all_map); free(st->kappa);
    free(st);
}

/* 3DGS rasterizer_impl.cu:35-50 */
static uint32_t getHigherMsb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) { step /= 2; if (n >> msb) msb += step; else msb -= step; }
    if (n >> msb) msb++;
    return msb;
}

/* stable LSD radix sort of (key,value) pairs on bits [0,end_bit): semantics of cub::DeviceRadixSort::SortPairs
   as called at 3DGS rasterizer_impl.cu:303-308 */
static void radix_sort_pairs(uint64_t* keys, uint32_t* vals, size_t n, int end_bit)
{
    uint64_t* k2 = (uint64_t*)xcalloc(n, sizeof(uint64_t));
    uint32_t* v2 = (uint32_t*)xcalloc(n, sizeof(uint32_t));
    for (int shift = 0; shift < end_bit; shift += 8) {
        int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        uint32_t mask = (1u << bits) - 1u;
        size_t cnt[257]; memset(cnt, 0, sizeof(cnt));
        for (size_t i = 0; i < n; i++) cnt[((keys[i] >> shift) & mask) + 1]++;
        for (int b = 0; b < 256; b++) cnt[b + 1] += cnt[b];
        for (size_t i = 0; i < n; i++) { size_t d = (keys[i] >> shift) & mask; size_t o = cnt[d]++; k2[o] = keys[i]; v2[o] = vals[i]; }
        uint64_t* tk = keys; keys = k2; k2 = tk; uint32_t* tv = vals; vals = v2; v2 = tv;
    }
    /* after an odd number of passes the result lives in the scratch arrays; copy back */
    int passes = (end_bit + 7) / 8;
    if (passes & 1) { memcpy(k2, keys, n * sizeof(uint64_t)); memcpy(v2, vals, n * sizeof(uint32_t)); free(keys); free(vals); }
    else { free(k2); free(v2); }
}

/* ------------------------------------------------------------------ forward */
/* per-gaussian preprocess: EWA = 3DGS forward.cu:156-256 (PLANE forward.cu:156-268 identical);
   SURFEL = forward.cu:149-251 */
static void preprocess_one(ref_state* st, const ref_inputs* in, int idx, real fx, real fy)
{
    const int W = in->W, H = in->H;
    st->radii[idx] = 0; st->tiles_touched[idx] = 0;
    /* truth runs: the culling decisions and the radius come from the float32 run (in->ov_radii); own gates are not applied */
    const int ov = in->ov_radii != NULL;
    if (ov && in->ov_radii[idx] <= 0) return;
    f3 p_view;
    if (!in_frustum(idx, in->means3D, in->viewmatrix, &p_view) && !ov) return;
    f3 p_orig = { in->means3D[3*idx], in->means3D[3*idx+1], in->means3D[3*idx+2] };

    f2 point_image; real my_radius;
    uint32_t rmin[2], rmax[2];
    if (st->variant != REF_SURFEL) {
        real ph[4]; transformPoint4x4(p_orig, in->projmatrix, ph);
        real p_w = 1.0f / (ph[3] + 0.0000001f);
        real projx = ph[0] * p_w, projy = ph[1] * p_w;
        const real* cov3D;
        if (in->cov3D_precomp) cov3D = in->cov3D_precomp + 6*idx;
        else { computeCov3D(in->scales + 3*idx, in->scale_modifier, in->rotations + 4*idx, st->cov3D + 6*idx); cov3D = st->cov3D + 6*idx; }
        real cov[3];
        computeCov2D(p_orig, fx, fy, in->tanfovx, in->tanfovy, cov3D, in->viewmatrix, cov, NULL, NULL, NULL, NULL, NULL);
        real det = (cov[0] * cov[2] - cov[1] * cov[1]);
        if (det == 0.0f && !ov) return;
        if (st->kappa) st->kappa[idx] = (cov[0] * cov[2] + cov[1] * cov[1]) / R_fabs(det);
        real det_inv = 1.f / det;
        real conic[3] = { cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv };
        real mid = 0.5f * (cov[0] + cov[2]);
        real lambda1 = mid + R_sqrt(R_fmax(0.1f, mid * mid - det));
        real lambda2 = mid - R_sqrt(R_fmax(0.1f, mid * mid - det));
        my_radius = R_ceil(3.f * R_sqrt(R_fmax(lambda1, lambda2)));
        point_image.x = ndc2Pix(projx, W); point_image.y = ndc2Pix(projy, H);
        getRect(point_image, (int)my_radius, st->gx, st->gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0 && !ov) return;
        st->conic_opacity[4*idx+0] = conic[0]; st->conic_opacity[4*idx+1] = conic[1];
        st->conic_opacity[4*idx+2] = conic[2]; st->conic_opacity[4*idx+3] = in->opacities[idx];
    } else {
        real T[9]; f3 normal;
        if (!in->cov3D_precomp) {
            compute_transmat(p_orig, in->scales + 2*idx, in->scale_modifier, in->rotations + 4*idx,
                             in->projmatrix, in->viewmatrix, W, H, T, &normal);
            memcpy(st->cov3D + 9*idx, T, sizeof(T));
        } else {
            memcpy(T, in->cov3D_precomp + 9*idx, sizeof(T));
            normal.x = 0.0f; normal.y = 0.0f; normal.z = 1.0f;
        }
        /* DUAL_VISIABLE, SURFEL forward.cu:209-214 */
        real cosv = -(p_view.x*normal.x + p_view.y*normal.y + p_view.z*normal.z);
        if (cosv == 0 && !ov) return;
        real mult = cosv > 0 ? 1.f : -1.f;
        normal.x = mult*normal.x; normal.y = mult*normal.y; normal.z = mult*normal.z;
        const real cutoff = 3.0f;
        f2 extent = { 0, 0 };
        point_image.x = 0; point_image.y = 0;
        if (!compute_aabb(T, cutoff, &point_image, &extent) && !ov) return;
        my_radius = R_ceil(R_fmax(R_fmax(extent.x, extent.y), cutoff * 0.707106f));
        getRect(point_image, (int)my_radius, st->gx, st->gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0 && !ov) return;
        st->conic_opacity[4*idx+0] = normal.x; st->conic_opacity[4*idx+1] = normal.y;
        st->conic_opacity[4*idx+2] = normal.z; st->conic_opacity[4*idx+3] = in->opacities[idx];
    }
    if (!in->colors_precomp) {
        real rgb[3];
        computeColorFromSH(idx, in->D, in->M, in->means3D, in->campos, in->shs, st->clamped, rgb);
        st->rgb[3*idx] = rgb[0]; st->rgb[3*idx+1] = rgb[1]; st->rgb[3*idx+2] = rgb[2];
    }
    st->depths[idx] = p_view.z;
    st->radii[idx] = ov ? in->ov_radii[idx] : (int)my_radius;
    st->means2D[2*idx] = point_image.x; st->means2D[2*idx+1] = point_image.y;
    st->tiles_touched[idx] = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
}

/* lock-free max into a non-negative real (bit patterns of non-negative IEEE numbers order like unsigned integers) */
static void atomic_max_real(real* p, real v)
{
#if GSR_REAL_IS_DOUBLE
    uint64_t* q = (uint64_t*)p; uint64_t nv; memcpy(&nv, &v, 8);
    uint64_t cur = __atomic_load_n(q, __ATOMIC_RELAXED);
    while (cur < nv && !__atomic_compare_exchange_n(q, &cur, nv, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { }
#else
    uint32_t* q = (uint32_t*)p; uint32_t nv; memcpy(&nv, &v, 4);
    uint32_t cur = __atomic_load_n(q, __ATOMIC_RELAXED);
    while (cur < nv && !__atomic_compare_exchange_n(q, &cur, nv, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { }
#endif
}

static const real near_n = 0.2f, far_n = 100.0f, FilterInvSquare = 2.0f;
/* truth runs: float32 unit roundoff and the safety factor on the first-order error bounds of the gate bookkeeping */
static const real U32 = 5.9604644775390625e-08f, GATE_SAFETY = 4.0f;

/* blend forward for one pixel of one tile.  EWA: 3DGS forward.cu:261-374.  PLANE: forward.cu:273-407.
   SURFEL: forward.cu:256-448. */
static void blend_pixel_fwd(const ref_state* st, const ref_inputs* in, const real* feat, real fx, real fy,
                            uint32_t px, uint32_t py, uint32_t r0, uint32_t r1,
                            real* out_color, real* out_others, int32_t* out_observe,
                            real* out_all_map, real* out_plane_depth)
{
    const int W = st->W, H = st->H; const size_t HW = (size_t)H * W;
    const uint32_t pix_id = (uint32_t)W * py + px;
    const real pixfx = (real)px, pixfy = (real)py;
    real T = 1.0f; uint32_t contributor = 0, last_contributor = 0;
    real C[3] = {0,0,0};
    /* surfel aux */
    real Nn[3] = {0,0,0}, Dd = 0, M1 = 0, M2 = 0, distortion = 0, median_depth = 0, median_contributor = -1;
    int surf_idx = -1; real median_normal[3] = {0,0,0};
    /* plane aux */
    real All_map[5] = {0,0,0,0,0};
    const real rayx = (pixfx - (real)(W * 0.5f)) / fx, rayy = (pixfy - (real)(H * 0.5f)) / fy;

    /* gate bookkeeping of truth runs (in->gate_margin != NULL), see gsr_oracle.h: GM = smallest margin / bound so far */
    const int gates = in->gate_margin != NULL;
    real GM = 1e30f, nT = 0; int GI = REF_GATE_NONE, GS = -1;
    uint32_t n_eval = 0, n_contrib_pairs = 0;
#define GATE(which, value, thr, bound) do { if (gates) { const real m_ = R_fabs((value) - (thr)) / ((bound) + 1e-300); \
        if (m_ < GM) { GM = m_; GI = (which); GS = (int)id; } } } while (0)
    for (uint32_t k = r0; k < r1; k++) {
        const uint32_t id = st->point_list[k];
        contributor++; n_eval++;
        real alpha, depth = 0; const real* nor_o = st->conic_opacity + 4*id;
        real e_pow = 0;     /* first-order bound of the float32 error of `power` (absolute) = of alpha (relative) */
        if (st->variant != REF_SURFEL) {
            real dx = st->means2D[2*id] - pixfx, dy = st->means2D[2*id+1] - pixfy;
            const real* con_o = nor_o;
            real power = -0.5f * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
            if (gates) {
                /* conic entries carry eta_c = 4 U32 kappa (inverse of an elongated 2D covariance), the projected centre eta_m = 2 U32
                   relative to its pixel coordinate; the arithmetic of `power` itself 4 roundings of its terms */
                const real S = 0.5f * (R_fabs(con_o[0]) * dx * dx + R_fabs(con_o[2]) * dy * dy) + R_fabs(con_o[1] * dx * dy);
                const real eta_c = U32 * 4 * st->kappa[id], eta_m = U32 * 2;
                e_pow = GATE_SAFETY * ((eta_c + 2 * U32) * S + eta_m * (R_fabs(con_o[0] * dx + con_o[1] * dy) * R_fabs(st->means2D[2*id])
                                                                   + R_fabs(con_o[2] * dy + con_o[1] * dx) * R_fabs(st->means2D[2*id+1])));
                GATE(REF_GATE_POWER, power, 0, e_pow);
            }
            if (power > 0.0f) continue;
            alpha = R_fmin(0.99f, con_o[3] * R_exp(power));
        } else {
            const real* Tm = (in->cov3D_precomp ? in->cov3D_precomp : st->cov3D) + 9*id;
            const real* Tu = Tm; const real* Tv = Tm + 3; const real* Tw = Tm + 6;
            real kx = pixfx*Tw[0] - Tu[0], ky = pixfx*Tw[1] - Tu[1], kz = pixfx*Tw[2] - Tu[2];
            real lx = pixfy*Tw[0] - Tv[0], ly = pixfy*Tw[1] - Tv[1], lz = pixfy*Tw[2] - Tv[2];
            real ppx = ky*lz - kz*ly, ppy = kz*lx - kx*lz, ppz = kx*ly - ky*lx;
            if (ppz == 0.0f) continue;
            real sx = ppx / ppz, sy = ppy / ppz;
            real rho3d = (sx*sx + sy*sy);
            real dx = st->means2D[2*id] - pixfx, dy = st->means2D[2*id+1] - pixfy;
            real rho2d = FilterInvSquare * (dx*dx + dy*dy);
            real rho = R_fmin(rho3d, rho2d);
            depth = (rho3d <= rho2d) ? (sx*Tw[0] + sy*Tw[1]) + Tw[2] : Tw[2];
            if (gates) {
                /* running error of the ray-splat intersection: T entries carry eta_T = 2 U32 (products of three matrices), then
                   k = px Tw - Tu, l = py Tw - Tv, p = k x l, s = p.xy / p.z -- the amplifier is 1 / |p.z| for edge-on splats */
                const real eta_T = U32 * 2;
                const real ek[3] = { eta_T * (R_fabs(pixfx*Tw[0]) + R_fabs(Tu[0])), eta_T * (R_fabs(pixfx*Tw[1]) + R_fabs(Tu[1])), eta_T * (R_fabs(pixfx*Tw[2]) + R_fabs(Tu[2])) };
                const real el[3] = { eta_T * (R_fabs(pixfy*Tw[0]) + R_fabs(Tv[0])), eta_T * (R_fabs(pixfy*Tw[1]) + R_fabs(Tv[1])), eta_T * (R_fabs(pixfy*Tw[2]) + R_fabs(Tv[2])) };
                const real epx = ek[1]*R_fabs(lz) + ek[2]*R_fabs(ly) + R_fabs(ky)*el[2] + R_fabs(kz)*el[1] + 2*U32*(R_fabs(ky*lz) + R_fabs(kz*ly));
                const real epy = ek[2]*R_fabs(lx) + ek[0]*R_fabs(lz) + R_fabs(kz)*el[0] + R_fabs(kx)*el[2] + 2*U32*(R_fabs(kz*lx) + R_fabs(kx*lz));
                const real epz = ek[0]*R_fabs(ly) + ek[1]*R_fabs(lx) + R_fabs(kx)*el[1] + R_fabs(ky)*el[0] + 2*U32*(R_fabs(kx*ly) + R_fabs(ky*lx));
                const real apz = R_fabs(ppz);
                const real esx = (epx + R_fabs(sx) * epz) / apz + 2*U32*R_fabs(sx), esy = (epy + R_fabs(sy) * epz) / apz + 2*U32*R_fabs(sy);
                const real e3 = GATE_SAFETY * (2 * (R_fabs(sx) * esx + R_fabs(sy) * esy) + 2*U32*rho3d);
                const real e2 = GATE_SAFETY * (2 * FilterInvSquare * U32 * 2 * (R_fabs(dx) * R_fabs(st->means2D[2*id]) + R_fabs(dy) * R_fabs(st->means2D[2*id+1])) + 2*U32*rho2d);
                GATE(REF_GATE_RHO, rho3d, rho2d, e3 + e2);
                const real ed = (rho3d <= rho2d) ? GATE_SAFETY * (esx * R_fabs(Tw[0]) + esy * R_fabs(Tw[1]) + (eta_T + 2*U32) * (R_fabs(sx*Tw[0]) + R_fabs(sy*Tw[1]) + R_fabs(Tw[2])))
                                                 : GATE_SAFETY * eta_T * R_fabs(Tw[2]);
                GATE(REF_GATE_NEAR, depth, near_n, ed);
                e_pow = 0.5f * ((rho3d <= rho2d) ? e3 : e2);
                GATE(REF_GATE_POWER, -0.5f * rho, 0, e_pow);
            }
            if (depth < near_n) continue;
            real power = -0.5f * rho;
            if (power > 0.0f) continue;
            alpha = R_fmin(0.99f, nor_o[3] * R_exp(power));
        }
        real e_alpha = 0;       /* absolute float32 error bound of alpha */
        if (gates) {
            e_alpha = (alpha < 0.99f) ? alpha * (e_pow + GATE_SAFETY * 2 * U32) : 0;
            GATE(REF_GATE_ALPHA, alpha, 1.0f / 255.0f, e_alpha);
            if (in->splat_noise && alpha >= 0.5f / 255.0f) {
                const real rel = e_pow + GATE_SAFETY * 2 * U32;
                atomic_max_real(&in->splat_noise[id], rel);
            }
        }
        if (alpha < 1.0f / 255.0f) continue;
        real test_T = T * (1 - alpha);
        if (gates) {
            GATE(REF_GATE_TERMINATE, test_T, 0.0001f, test_T * (nT + e_alpha / (1 - alpha) + GATE_SAFETY * 2 * U32));
            if (st->variant != REF_EWA && test_T >= 0.0001f) GATE(REF_GATE_HALF, T, 0.5f, T * (nT + GATE_SAFETY * U32));
        }
        if (test_T < 0.0001f) break;   /* done = true: nothing after this can contribute */
        n_contrib_pairs++;
        nT += e_alpha / (1 - alpha) + GATE_SAFETY * 2 * U32;
        real w = alpha * T;
        if (st->variant == REF_SURFEL) {
            real A = 1 - T;
            real m = far_n / (far_n - near_n) * (1 - near_n / depth);
            distortion += (m * m * A + M2 - 2 * m * M1) * w;
            Dd += depth * w; M1 += m * w; M2 += m * m * w;
            if (T > 0.5f) {
                median_depth = depth; surf_idx = (int)id;
                for (int ch = 0; ch < 3; ch++) median_normal[ch] = nor_o[ch];
                median_contributor = (real)contributor;
            }
            for (int ch = 0; ch < 3; ch++) Nn[ch] += nor_o[ch] * w;
        }
        for (int ch = 0; ch < 3; ch++) C[ch] += feat[3*id + ch] * alpha * T;
        if (st->variant == REF_PLANE) {
            if (in->render_geo)
                for (int ch = 0; ch < 5; ch++) All_map[ch] += in->all_map[5*id + ch] * alpha * T;
            if (T > 0.5f) {
#ifdef _OPENMP
#pragma omp atomic
#endif
                out_observe[id] += 1;
            }
        }
        T = test_T;
        last_contributor = contributor;
    }
#undef GATE
    __atomic_fetch_add(&((ref_state*)st)->pairs_evaluated, (uint64_t)n_eval, __ATOMIC_RELAXED);
    __atomic_fetch_add(&((ref_state*)st)->pairs_contributing, (uint64_t)n_contrib_pairs, __ATOMIC_RELAXED);
    if (gates) { in->gate_margin[pix_id] = GM; in->gate_id[pix_id] = GI; in->gate_splat[pix_id] = GS; }
    st->final_T[pix_id] = T;
    st->n_contrib[pix_id] = last_contributor;
    for (int ch = 0; ch < 3; ch++) out_color[ch*HW + pix_id] = C[ch] + T * in->bg[ch];
    if (st->variant == REF_SURFEL) {
        /* real -> uint32 conversion of -1 saturates to 0 on the GPU; restated explicitly */
        st->n_contrib[pix_id + HW] = median_contributor < 0 ? 0u : (uint32_t)median_contributor;
        st->final_T[pix_id + HW] = M1;
        st->final_T[pix_id + 2*HW] = M2;
        out_others[pix_id + 0*HW] = Dd;
        out_others[pix_id + 1*HW] = 1 - T;
        for (int ch = 0; ch < 3; ch++) out_others[pix_id + (2+ch)*HW] = Nn[ch];
        out_others[pix_id + 5*HW] = median_depth;
        out_others[pix_id + 6*HW] = distortion;
        out_others[pix_id + 7*HW] = (real)surf_idx;
        for (int ch = 0; ch < 3; ch++) out_others[pix_id + (8+ch)*HW] = median_normal[ch];
    }
    if (st->variant == REF_PLANE && in->render_geo) {
        for (int ch = 0; ch < 5; ch++) out_all_map[ch*HW + pix_id] = All_map[ch];
        out_plane_depth[pix_id] = (real)(All_map[4] / -(All_map[0] * rayx + All_map[1] * rayy + All_map[2] + 1.0e-8));
    }
}

ref_state* ref_forward(int variant, const ref_inputs* in, real* out_color, int32_t* radii, real* out_others,
                       int32_t* out_observe, real* out_all_map, real* out_plane_depth)
{
    ref_state* st = (ref_state*)xcalloc(1, sizeof(ref_state));
    const int P = in->P, W = in->W, H = in->H;
    st->variant = variant; st->P = P; st->W = W; st->H = H;
    st->gx = (W + BLOCK_X - 1) / BLOCK_X; st->gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    st->T = st->gx * st->gy; st->N = W * H;
    const size_t HW = (size_t)W * H;
    const int tm = (variant == REF_SURFEL) ? 9 : 6;
    st->depths = (real*)xcalloc(P, RS); st->clamped = (uint8_t*)xcalloc((size_t)P*3, 1);
    st->means2D = (real*)xcalloc((size_t)P*2, RS); st->cov3D = (real*)xcalloc((size_t)P*tm, RS);
    st->conic_opacity = (real*)xcalloc((size_t)P*4, RS); st->rgb = (real*)xcalloc((size_t)P*3, RS);
    st->tiles_touched = (uint32_t*)xcalloc(P, 4); st->point_offsets = (uint32_t*)xcalloc(P, 4);
    st->radii = (int32_t*)xcalloc(P, 4);
    st->final_T = (real*)xcalloc(HW * (variant == REF_SURFEL ? 3 : 1), RS);
    st->n_contrib = (uint32_t*)xcalloc(HW * (variant == REF_SURFEL ? 2 : 1), 4);
    st->ranges = (uint32_t*)xcalloc((size_t)st->T * 2, 4);
    if (in->gate_margin) st->kappa = (real*)xcalloc(P, RS);

    /* 3DGS rasterizer_impl.cu:220-221 */
    const real focal_y = H / (2.0f * in->tanfovy);
    const real focal_x = W / (2.0f * in->tanfovx);

    memset(out_color, 0, HW * 3 * RS);
    if (variant == REF_SURFEL) memset(out_others, 0, HW * 11 * RS);
    if (variant == REF_PLANE) { memset(out_observe, 0, (size_t)P * 4); memset(out_all_map, 0, HW * 5 * RS); memset(out_plane_depth, 0, HW * RS); }

#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int i = 0; i < P; i++) preprocess_one(st, in, i, focal_x, focal_y);
    if (radii) memcpy(radii, st->radii, (size_t)P * 4);
    /* float32-geometry runs (gsr_oracle.h ov_cov ...): blend what a float32 preprocess produced */
    if (in->ov_cov) memcpy(st->cov3D, in->ov_cov, (size_t)P * tm * RS);
    if (in->ov_conic_opacity) memcpy(st->conic_opacity, in->ov_conic_opacity, (size_t)P * 4 * RS);
    if (in->ov_means2D) memcpy(st->means2D, in->ov_means2D, (size_t)P * 2 * RS);
    if (in->ov_depths) memcpy(st->depths, in->ov_depths, (size_t)P * RS);
    if (in->ov_rgb && !in->colors_precomp) memcpy(st->rgb, in->ov_rgb, (size_t)P * 3 * RS);

    if (in->ov_radii) {
        /* truth run: sorted instance list and tile ranges of the float32 run */
        st->R = in->ov_R;
        st->point_list = (uint32_t*)xcalloc(st->R, 4);
        memcpy(st->point_list, in->ov_point_list, (size_t)st->R * 4);
        memcpy(st->ranges, in->ov_ranges, (size_t)st->T * 8);
        goto blend;
    }
    /* inclusive scan, 3DGS rasterizer_impl.cu:277-281 */
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) { acc += st->tiles_touched[i]; st->point_offsets[i] = acc; }
    st->R = (int)acc;
    st->keys = (uint64_t*)xcalloc(st->R, 8); st->point_list = (uint32_t*)xcalloc(st->R, 4);

    /* duplicateWithKeys, 3DGS rasterizer_impl.cu:70-111 */
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int i = 0; i < P; i++) {
        if (st->radii[i] > 0) {
            uint32_t off = (i == 0) ? 0 : st->point_offsets[i - 1];
            uint32_t rmin[2], rmax[2];
            f2 pxy = { st->means2D[2*i], st->means2D[2*i+1] };
            getRect(pxy, st->radii[i], st->gx, st->gy, rmin, rmax);
            const float d32 = (float)st->depths[i]; uint32_t dbits; memcpy(&dbits, &d32, 4);   /* the key is the FLOAT32 bit pattern of the depth */
            for (uint32_t y = rmin[1]; y < rmax[1]; y++)
                for (uint32_t x = rmin[0]; x < rmax[0]; x++) {
                    uint64_t key = (uint64_t)(y * (uint32_t)st->gx + x);
                    key <<= 32; key |= dbits;
                    st->keys[off] = key; st->point_list[off] = (uint32_t)i; off++;
                }
        }
    }
    /* sort on bits [0, 32 + msb(T)), 3DGS rasterizer_impl.cu:300-308 */
    int bit = (int)getHigherMsb((uint32_t)st->T);
    if (st->R > 0) {
        /* radix_sort_pairs may swap buffers internally; it guarantees the result ends up in the arrays passed */
        radix_sort_pairs(st->keys, st->point_list, (size_t)st->R, 32 + bit);
    }
    /* identifyTileRanges, 3DGS rasterizer_impl.cu:116-138 */
    for (int i = 0; i < st->R; i++) {
        uint32_t cur = (uint32_t)(st->keys[i] >> 32);
        if (i == 0) st->ranges[2*cur] = 0;
        else {
            uint32_t prev = (uint32_t)(st->keys[i-1] >> 32);
            if (cur != prev) { st->ranges[2*prev+1] = (uint32_t)i; st->ranges[2*cur] = (uint32_t)i; }
        }
        if (i == st->R - 1) st->ranges[2*cur+1] = (uint32_t)st->R;
    }

blend:;
    const int tile_stride = g_tile_stride;
    const real* feat = in->colors_precomp ? in->colors_precomp : st->rgb;
    if (in->splat_noise) memset(in->splat_noise, 0, (size_t)P * RS);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int tile = 0; tile < st->T; tile++) {
        if (tile % tile_stride) continue;
        uint32_t tx = (uint32_t)(tile % st->gx), ty = (uint32_t)(tile / st->gx);
        uint32_t r0 = st->ranges[2*tile], r1 = st->ranges[2*tile+1];
        for (uint32_t ly = 0; ly < BLOCK_Y; ly++)
            for (uint32_t lx = 0; lx < BLOCK_X; lx++) {
                uint32_t px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (px < (uint32_t)W && py < (uint32_t)H)
                    blend_pixel_fwd(st, in, feat, focal_x, focal_y, px, py, r0, r1, out_color, out_others,
                                    out_observe, out_all_map, out_plane_depth);
            }
    }
    if (variant == REF_PLANE) {
        st->out_all_map = (real*)xcalloc(HW * 5, RS);
        memcpy(st->out_all_map, out_all_map, HW * 5 * RS);
    }
    return st;
}

/* ------------------------------------------------------------------ which tile instances can contribute at all
   For every instance k of the sorted list (tile = the range it lies in) the largest alpha the reference's per-pixel gates let through on any
   pixel of that tile that lies inside the image -- the same statements as blend_pixel_fwd above (EWA 3DGS forward.cu:330-345, SURFEL
   forward.cu:351-392: power > 0, p.z == 0, depth < near skip the pair), without the transmittance state: an instance whose value stays below
   1/255 fails the alpha gate (3DGS forward.cu:346, SURFEL :393) on every pixel, whatever lies in front of it.  The HIP library drops such
   instances when it emits the list (gsr_tile_cull.h); tests hold its filtered list against this. */
static real pair_alpha(const ref_state* st, const ref_inputs* in, uint32_t id, real pixfx, real pixfy)
{
    const real* nor_o = st->conic_opacity + 4*id;
    if (st->variant != REF_SURFEL) {
        real dx = st->means2D[2*id] - pixfx, dy = st->means2D[2*id+1] - pixfy;
        real power = -0.5f * (nor_o[0] * dx * dx + nor_o[2] * dy * dy) - nor_o[1] * dx * dy;
        if (power > 0.0f) return 0;
        return R_fmin(0.99f, nor_o[3] * R_exp(power));
    }
    const real* Tm = (in->cov3D_precomp ? in->cov3D_precomp : st->cov3D) + 9*id;
    const real* Tu = Tm; const real* Tv = Tm + 3; const real* Tw = Tm + 6;
    real kx = pixfx*Tw[0] - Tu[0], ky = pixfx*Tw[1] - Tu[1], kz = pixfx*Tw[2] - Tu[2];
    real lx = pixfy*Tw[0] - Tv[0], ly = pixfy*Tw[1] - Tv[1], lz = pixfy*Tw[2] - Tv[2];
    real ppx = ky*lz - kz*ly, ppy = kz*lx - kx*lz, ppz = kx*ly - ky*lx;
    if (ppz == 0.0f) return 0;
    real sx = ppx / ppz, sy = ppy / ppz;
    real rho3d = (sx*sx + sy*sy);
    real dx = st->means2D[2*id] - pixfx, dy = st->means2D[2*id+1] - pixfy;
    real rho2d = FilterInvSquare * (dx*dx + dy*dy);
    real rho = R_fmin(rho3d, rho2d);
    real depth = (rho3d <= rho2d) ? (sx*Tw[0] + sy*Tw[1]) + Tw[2] : Tw[2];
    if (depth < near_n) return 0;
    real power = -0.5f * rho;
    if (power > 0.0f) return 0;
    return R_fmin(0.99f, nor_o[3] * R_exp(power));
}
void ref_instance_max_alpha(const ref_state* st, const ref_inputs* in, real* out)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int tile = 0; tile < st->T; tile++) {
        const uint32_t tx = (uint32_t)(tile % st->gx), ty = (uint32_t)(tile / st->gx);
        const uint32_t r0 = st->ranges[2*tile], r1 = st->ranges[2*tile+1];
        for (uint32_t k = r0; k < r1; k++) {
            const uint32_t id = st->point_list[k];
            real best = 0;
            for (uint32_t ly = 0; ly < BLOCK_Y; ly++)
                for (uint32_t lx = 0; lx < BLOCK_X; lx++) {
                    const uint32_t px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                    if (px < (uint32_t)st->W && py < (uint32_t)st->H) best = R_fmax(best, pair_alpha(st, in, id, (real)px, (real)py));
                }
            out[k] = best;
        }
    }
}

/* ------------------------------------------------------------------ backward */
static void atomic_addf(real* p, real v)
{
#ifdef _OPENMP
#pragma omp atomic
#endif
    *p += v;
}

/* blend backward, one pixel.  EWA: 3DGS backward.cu:399-557.  PLANE: backward.cu:399-614.
   SURFEL: backward.cu:143-447. */
static void blend_pixel_bwd(const ref_state* st, const ref_inputs* in, const real* colors, real fx, real fy,
                            uint32_t px, uint32_t py, uint32_t r0, uint32_t r1, const ref_out_grads* og,
                            real* dL_dmean2D /*[P,3]*/, real* dL_dmean2D_abs, real* dL_dconic /*[P,4]*/,
                            real* dL_dnormal3D /*[P,3]*/, real* dL_dtransMat /*[P,9]*/, real* dL_dopacity,
                            real* dL_dcolors, real* dL_dall_map)
{
    const int W = st->W, H = st->H; const size_t HW = (size_t)H * W;
    const uint32_t pix_id = (uint32_t)W * py + px;
    const real pixfx = (real)px, pixfy = (real)py;
    const int toDo = (int)(r1 - r0);
    const real T_final = st->final_T[pix_id];
    real T = T_final;
    uint32_t contributor = (uint32_t)toDo;
    const int last_contributor = (int)st->n_contrib[pix_id];
    real accum_rec[3] = {0,0,0}, dL_dpixel[3], last_color[3] = {0,0,0};
    for (int i = 0; i < 3; i++) dL_dpixel[i] = og->dL_dcolor ? og->dL_dcolor[i*HW + pix_id] : 0.0f;
    real last_alpha = 0;
    const real ddelx_dx = (real)(0.5 * W), ddely_dy = (real)(0.5 * H);

    /* PLANE extras, backward.cu:433,460-490 */
    real accum_all_map[5] = {0,0,0,0,0}, last_all_map[5] = {0,0,0,0,0}, dL_dout_all_map[5] = {0,0,0,0,0};
    const int geo = (st->variant == REF_PLANE) && in->render_geo;
    if (geo) {
        const real rayx = (real)((pixfx - W * 0.5) / fx), rayy = (real)((pixfy - H * 0.5) / fy);
        for (int i = 0; i < 5; i++) dL_dout_all_map[i] = og->dL_dout_all_map ? og->dL_dout_all_map[i*HW + pix_id] : 0.0f;
        const real nx = st->out_all_map[pix_id], ny = st->out_all_map[HW + pix_id], nz = st->out_all_map[2*HW + pix_id];
        const real distance = st->out_all_map[4*HW + pix_id];
        const real tmp = (real)(nx * rayx + ny * rayy + nz + 1.0e-8);
        const real dpd = og->dL_dplane_depth ? og->dL_dplane_depth[pix_id] : 0.0f;
        dL_dout_all_map[4] += (-dpd / tmp);
        dL_dout_all_map[0] += dpd * (distance / (tmp * tmp) * rayx);
        dL_dout_all_map[1] += dpd * (distance / (tmp * tmp) * rayy);
        dL_dout_all_map[2] += dpd * (distance / (tmp * tmp));
    }
    /* SURFEL extras, backward.cu:205-243 */
    real dL_dreg = 0, dL_ddepth = 0, dL_daccum = 0, dL_dnormal2D[3] = {0,0,0}, dL_dmedian_depth = 0;
    real dL_dmedian_normal2D[3] = {0,0,0};
    int median_contributor = 0;
    real last_depth = 0, last_normal[3] = {0,0,0}, accum_depth_rec = 0, accum_alpha_rec = 0, accum_normal_rec[3] = {0,0,0};
    real final_D = 0, final_D2 = 0, final_A = 0, last_dL_dT = 0;
    if (st->variant == REF_SURFEL) {
        median_contributor = (int)st->n_contrib[pix_id + HW];
        if (og->dL_dothers) {
            const real* g = og->dL_dothers;
            dL_ddepth = g[0*HW + pix_id]; dL_daccum = g[1*HW + pix_id]; dL_dreg = g[6*HW + pix_id];
            for (int i = 0; i < 3; i++) dL_dnormal2D[i] = g[(2+i)*HW + pix_id];
            dL_dmedian_depth = g[5*HW + pix_id];
            for (int i = 0; i < 3; i++) dL_dmedian_normal2D[i] = g[(8+i)*HW + pix_id];
        }
        final_D = st->final_T[pix_id + HW]; final_D2 = st->final_T[pix_id + 2*HW]; final_A = 1 - T_final;
    }

    for (int j = 0; j < toDo; j++) {
        const uint32_t id = st->point_list[r1 - 1 - (uint32_t)j];
        contributor--;
        if ((int)contributor >= last_contributor) continue;   /* uint32 vs int compare in the source: both non-negative */
        const real* nor_o = st->conic_opacity + 4*id;
        const real dx = st->means2D[2*id] - pixfx, dy = st->means2D[2*id+1] - pixfy;
        real G, alpha;
        /* surfel intersection temporaries */
        real kx=0,ky=0,kz=0,lx=0,ly=0,lz=0,ppz=0,sx=0,sy=0,rho3d=0,rho2d=0,c_d=0; const real* Tw = NULL;
        if (st->variant != REF_SURFEL) {
            const real power = -0.5f * (nor_o[0] * dx * dx + nor_o[2] * dy * dy) - nor_o[1] * dx * dy;
            if (power > 0.0f) continue;
            G = R_exp(power);
        } else {
            const real* Tm = (in->cov3D_precomp ? in->cov3D_precomp : st->cov3D) + 9*id;
            const real* Tu = Tm; const real* Tv = Tm + 3; Tw = Tm + 6;
            kx = pixfx*Tw[0] - Tu[0]; ky = pixfx*Tw[1] - Tu[1]; kz = pixfx*Tw[2] - Tu[2];
            lx = pixfy*Tw[0] - Tv[0]; ly = pixfy*Tw[1] - Tv[1]; lz = pixfy*Tw[2] - Tv[2];
            real ppx = ky*lz - kz*ly, ppy = kz*lx - kx*lz; ppz = kx*ly - ky*lx;
            if (ppz == 0.0f) continue;
            sx = ppx / ppz; sy = ppy / ppz;
            rho3d = (sx*sx + sy*sy);
            rho2d = FilterInvSquare * (dx*dx + dy*dy);
            real rho = R_fmin(rho3d, rho2d);
            c_d = (rho3d <= rho2d) ? (sx*Tw[0] + sy*Tw[1]) + Tw[2] : Tw[2];
            if (c_d < near_n) continue;
            real power = -0.5f * rho;
            if (power > 0.0f) continue;
            G = R_exp(power);
        }
        alpha = R_fmin(0.99f, nor_o[3] * G);
        if (alpha < 1.0f / 255.0f) continue;

        T = T / (1.f - alpha);
        const real dchannel_dcolor = alpha * T;
        real dL_dalpha = 0.0f;
        for (int ch = 0; ch < 3; ch++) {
            const real c = colors[3*id + ch];
            accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
            last_color[ch] = c;
            const real dL_dchannel = dL_dpixel[ch];
            dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
            atomic_addf(&dL_dcolors[3*id + ch], dchannel_dcolor * dL_dchannel);
        }
        if (geo) {
            for (int ch = 0; ch < 5; ch++) {
                const real c = in->all_map[5*id + ch];
                accum_all_map[ch] = last_alpha * last_all_map[ch] + (1.f - last_alpha) * accum_all_map[ch];
                last_all_map[ch] = c;
                const real dL_dchannel = dL_dout_all_map[ch];
                dL_dalpha += (c - accum_all_map[ch]) * dL_dchannel;
                atomic_addf(&dL_dall_map[5*id + ch], dchannel_dcolor * dL_dchannel);
            }
        }
        real dL_dz = 0.0f;
        if (st->variant == REF_SURFEL) {
            real dL_dweight = 0;
            const real m_d = far_n / (far_n - near_n) * (1 - near_n / c_d);
            const real dmd_dd = (far_n * near_n) / ((far_n - near_n) * c_d * c_d);
            if ((int)contributor == median_contributor - 1) dL_dz += dL_dmedian_depth;
            dL_dweight += (final_D2 + m_d * m_d * final_A - 2 * m_d * final_D) * dL_dreg;
            dL_dalpha += dL_dweight - last_dL_dT;
            last_dL_dT = dL_dweight * alpha + (1 - alpha) * last_dL_dT;
            const real dL_dmd = 2.0f * (T * alpha) * (m_d * final_A - final_D) * dL_dreg;
            dL_dz += dL_dmd * dmd_dd;
            accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
            last_depth = c_d;
            dL_dalpha += (c_d - accum_depth_rec) * dL_ddepth;
            accum_alpha_rec = (real)(last_alpha * 1.0 + (1.f - last_alpha) * accum_alpha_rec);
            dL_dalpha += (1 - accum_alpha_rec) * dL_daccum;
            for (int ch = 0; ch < 3; ch++) {
                accum_normal_rec[ch] = last_alpha * last_normal[ch] + (1.f - last_alpha) * accum_normal_rec[ch];
                last_normal[ch] = nor_o[ch];
                dL_dalpha += (nor_o[ch] - accum_normal_rec[ch]) * dL_dnormal2D[ch];
                atomic_addf(&dL_dnormal3D[3*id + ch], alpha * T * dL_dnormal2D[ch]);
                /* fork quirk (SURFEL backward.cu:381): median-normal grad goes to EVERY contributing splat */
                atomic_addf(&dL_dnormal3D[3*id + ch], dL_dmedian_normal2D[ch]);
            }
        }
        dL_dalpha *= T;
        last_alpha = alpha;
        real bg_dot_dpixel = 0;
        for (int i = 0; i < 3; i++) bg_dot_dpixel += in->bg[i] * dL_dpixel[i];
        dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
        const real dL_dG = nor_o[3] * dL_dalpha;

        if (st->variant != REF_SURFEL) {
            const real gdx = G * dx, gdy = G * dy;
            const real dG_ddelx = -gdx * nor_o[0] - gdy * nor_o[1];
            const real dG_ddely = -gdy * nor_o[2] - gdx * nor_o[1];
            atomic_addf(&dL_dmean2D[3*id + 0], dL_dG * dG_ddelx * ddelx_dx);
            atomic_addf(&dL_dmean2D[3*id + 1], dL_dG * dG_ddely * ddely_dy);
            if (st->variant == REF_PLANE) {
                atomic_addf(&dL_dmean2D_abs[3*id + 0], R_fabs(dL_dG * dG_ddelx * ddelx_dx));
                atomic_addf(&dL_dmean2D_abs[3*id + 1], R_fabs(dL_dG * dG_ddely * ddely_dy));
            }
            atomic_addf(&dL_dconic[4*id + 0], -0.5f * gdx * dx * dL_dG);
            atomic_addf(&dL_dconic[4*id + 1], -0.5f * gdx * dy * dL_dG);
            atomic_addf(&dL_dconic[4*id + 3], -0.5f * gdy * dy * dL_dG);
        } else {
            dL_dz += alpha * T * dL_ddepth;
            if (rho3d <= rho2d) {
                const real dL_dsx = dL_dG * -G * sx + dL_dz * Tw[0];
                const real dL_dsy = dL_dG * -G * sy + dL_dz * Tw[1];
                const real dsx_pz = dL_dsx / ppz, dsy_pz = dL_dsy / ppz;
                const real dpx = dsx_pz, dpy = dsy_pz, dpz = -(dsx_pz * sx + dsy_pz * sy);
                /* dL_dk = cross(l, dL_dp); dL_dl = cross(dL_dp, k) */
                const real dkx = ly*dpz - lz*dpy, dky = lz*dpx - lx*dpz, dkz = lx*dpy - ly*dpx;
                const real dlx = dpy*kz - dpz*ky, dly = dpz*kx - dpx*kz, dlz = dpx*ky - dpy*kx;
                real* g = dL_dtransMat + 9*id;
                atomic_addf(&g[0], -dkx); atomic_addf(&g[1], -dky); atomic_addf(&g[2], -dkz);
                atomic_addf(&g[3], -dlx); atomic_addf(&g[4], -dly); atomic_addf(&g[5], -dlz);
                atomic_addf(&g[6], pixfx * dkx + pixfy * dlx + dL_dz * sx);
                atomic_addf(&g[7], pixfx * dky + pixfy * dly + dL_dz * sy);
                atomic_addf(&g[8], pixfx * dkz + pixfy * dlz + dL_dz * 1.0f);
            } else {
                const real dG_ddelx = -G * FilterInvSquare * dx;
                const real dG_ddely = -G * FilterInvSquare * dy;
                atomic_addf(&dL_dmean2D[3*id + 0], dL_dG * dG_ddelx);
                atomic_addf(&dL_dmean2D[3*id + 1], dL_dG * dG_ddely);
                atomic_addf(&dL_dtransMat[9*id + 8], dL_dz);
            }
        }
        atomic_addf(&dL_dopacity[id], G * dL_dalpha);
    }
}

/* 3DGS backward.cu:144-274 */
static void computeCov2D_bwd(const ref_inputs* in, int idx, const real* cov3Ds, real h_x, real h_y,
                             const real* dL_dconics, real* dL_dmeans, real* dL_dcov)
{
    const real* cov3D = cov3Ds + 6*idx;
    f3 mean = { in->means3D[3*idx], in->means3D[3*idx+1], in->means3D[3*idx+2] };
    real dcx = dL_dconics[4*idx], dcy = dL_dconics[4*idx+1], dcz = dL_dconics[4*idx+3];
    real cov[3]; m3 T, Vrk; f3 t; real x_grad_mul, y_grad_mul;
    computeCov2D(mean, h_x, h_y, in->tanfovx, in->tanfovy, cov3D, in->viewmatrix, cov, &T, &Vrk, &t, &x_grad_mul, &y_grad_mul);
    const real* vm = in->viewmatrix;
    m3 Wm = m3_cols(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
    real a = cov[0], b = cov[1], c = cov[2];
    real denom = a * c - b * b;
    real dL_da = 0, dL_db = 0, dL_dc = 0;
    real denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    if (denom2inv != 0) {
        dL_da = denom2inv * (-c * c * dcx + 2 * b * c * dcy + (denom - a * c) * dcz);
        dL_dc = denom2inv * (-a * a * dcz + 2 * a * b * dcy + (denom - a * c) * dcx);
        dL_db = denom2inv * 2 * (b * c * dcx - (denom + 2 * b * b) * dcy + a * b * dcz);
        dL_dcov[6*idx+0] = (T.c[0][0]*T.c[0][0]*dL_da + T.c[0][0]*T.c[1][0]*dL_db + T.c[1][0]*T.c[1][0]*dL_dc);
        dL_dcov[6*idx+3] = (T.c[0][1]*T.c[0][1]*dL_da + T.c[0][1]*T.c[1][1]*dL_db + T.c[1][1]*T.c[1][1]*dL_dc);
        dL_dcov[6*idx+5] = (T.c[0][2]*T.c[0][2]*dL_da + T.c[0][2]*T.c[1][2]*dL_db + T.c[1][2]*T.c[1][2]*dL_dc);
        dL_dcov[6*idx+1] = 2*T.c[0][0]*T.c[0][1]*dL_da + (T.c[0][0]*T.c[1][1] + T.c[0][1]*T.c[1][0])*dL_db + 2*T.c[1][0]*T.c[1][1]*dL_dc;
        dL_dcov[6*idx+2] = 2*T.c[0][0]*T.c[0][2]*dL_da + (T.c[0][0]*T.c[1][2] + T.c[0][2]*T.c[1][0])*dL_db + 2*T.c[1][0]*T.c[1][2]*dL_dc;
        dL_dcov[6*idx+4] = 2*T.c[0][2]*T.c[0][1]*dL_da + (T.c[0][1]*T.c[1][2] + T.c[0][2]*T.c[1][1])*dL_db + 2*T.c[1][1]*T.c[1][2]*dL_dc;
    } else {
        for (int i = 0; i < 6; i++) dL_dcov[6*idx+i] = 0;
    }
    real dL_dT00 = 2*(T.c[0][0]*Vrk.c[0][0] + T.c[0][1]*Vrk.c[0][1] + T.c[0][2]*Vrk.c[0][2])*dL_da + (T.c[1][0]*Vrk.c[0][0] + T.c[1][1]*Vrk.c[0][1] + T.c[1][2]*Vrk.c[0][2])*dL_db;
    real dL_dT01 = 2*(T.c[0][0]*Vrk.c[1][0] + T.c[0][1]*Vrk.c[1][1] + T.c[0][2]*Vrk.c[1][2])*dL_da + (T.c[1][0]*Vrk.c[1][0] + T.c[1][1]*Vrk.c[1][1] + T.c[1][2]*Vrk.c[1][2])*dL_db;
    real dL_dT02 = 2*(T.c[0][0]*Vrk.c[2][0] + T.c[0][1]*Vrk.c[2][1] + T.c[0][2]*Vrk.c[2][2])*dL_da + (T.c[1][0]*Vrk.c[2][0] + T.c[1][1]*Vrk.c[2][1] + T.c[1][2]*Vrk.c[2][2])*dL_db;
    real dL_dT10 = 2*(T.c[1][0]*Vrk.c[0][0] + T.c[1][1]*Vrk.c[0][1] + T.c[1][2]*Vrk.c[0][2])*dL_dc + (T.c[0][0]*Vrk.c[0][0] + T.c[0][1]*Vrk.c[0][1] + T.c[0][2]*Vrk.c[0][2])*dL_db;
    real dL_dT11 = 2*(T.c[1][0]*Vrk.c[1][0] + T.c[1][1]*Vrk.c[1][1] + T.c[1][2]*Vrk.c[1][2])*dL_dc + (T.c[0][0]*Vrk.c[1][0] + T.c[0][1]*Vrk.c[1][1] + T.c[0][2]*Vrk.c[1][2])*dL_db;
    real dL_dT12 = 2*(T.c[1][0]*Vrk.c[2][0] + T.c[1][1]*Vrk.c[2][1] + T.c[1][2]*Vrk.c[2][2])*dL_dc + (T.c[0][0]*Vrk.c[2][0] + T.c[0][1]*Vrk.c[2][1] + T.c[0][2]*Vrk.c[2][2])*dL_db;
    real dL_dJ00 = Wm.c[0][0]*dL_dT00 + Wm.c[0][1]*dL_dT01 + Wm.c[0][2]*dL_dT02;
    real dL_dJ02 = Wm.c[2][0]*dL_dT00 + Wm.c[2][1]*dL_dT01 + Wm.c[2][2]*dL_dT02;
    real dL_dJ11 = Wm.c[1][0]*dL_dT10 + Wm.c[1][1]*dL_dT11 + Wm.c[1][2]*dL_dT12;
    real dL_dJ12 = Wm.c[2][0]*dL_dT10 + Wm.c[2][1]*dL_dT11 + Wm.c[2][2]*dL_dT12;
    real tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
    real dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
    real dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
    real dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12;
    f3 dt = { dL_dtx, dL_dty, dL_dtz };
    f3 dm = transformVec4x3Transpose(dt, vm);
    dL_dmeans[3*idx] = dm.x; dL_dmeans[3*idx+1] = dm.y; dL_dmeans[3*idx+2] = dm.z;   /* assignment */
}

/* 3DGS backward.cu:278-341 (no quaternion-normalisation Jacobian) */
static void computeCov3D_bwd(int idx, const real* scale, real mod, const real* rot, const real* dL_dcov3Ds,
                             real* dL_dscales, real* dL_drots)
{
    real r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    m3 R = m3_cols(1.f - 2.f*(y*y + z*z), 2.f*(x*y - r*z), 2.f*(x*z + r*y),
                   2.f*(x*y + r*z), 1.f - 2.f*(x*x + z*z), 2.f*(y*z - r*x),
                   2.f*(x*z - r*y), 2.f*(y*z + r*x), 1.f - 2.f*(x*x + y*y));
    real s[3] = { mod*scale[0], mod*scale[1], mod*scale[2] };
    m3 S = m3_cols(s[0],0,0, 0,s[1],0, 0,0,s[2]);
    m3 Mm = m3_mul(S, R);
    const real* g = dL_dcov3Ds + 6*idx;
    m3 dL_dSigma = m3_cols(g[0], 0.5f*g[1], 0.5f*g[2], 0.5f*g[1], g[3], 0.5f*g[4], 0.5f*g[2], 0.5f*g[4], g[5]);
    m3 M2 = Mm; for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) M2.c[j][i] = 2.0f * Mm.c[j][i];
    m3 dL_dM = m3_mul(M2, dL_dSigma);
    m3 Rt = m3_t(R), dL_dMt = m3_t(dL_dM);
    dL_dscales[3*idx+0] = dot3(Rt.c[0], dL_dMt.c[0]);
    dL_dscales[3*idx+1] = dot3(Rt.c[1], dL_dMt.c[1]);
    dL_dscales[3*idx+2] = dot3(Rt.c[2], dL_dMt.c[2]);
    for (int i = 0; i < 3; i++) { dL_dMt.c[0][i] *= s[0]; dL_dMt.c[1][i] *= s[1]; dL_dMt.c[2][i] *= s[2]; }
    real* q = dL_drots + 4*idx;
    q[0] = 2*z*(dL_dMt.c[0][1] - dL_dMt.c[1][0]) + 2*y*(dL_dMt.c[2][0] - dL_dMt.c[0][2]) + 2*x*(dL_dMt.c[1][2] - dL_dMt.c[2][1]);
    q[1] = 2*y*(dL_dMt.c[1][0] + dL_dMt.c[0][1]) + 2*z*(dL_dMt.c[2][0] + dL_dMt.c[0][2]) + 2*r*(dL_dMt.c[1][2] - dL_dMt.c[2][1]) - 4*x*(dL_dMt.c[2][2] + dL_dMt.c[1][1]);
    q[2] = 2*x*(dL_dMt.c[1][0] + dL_dMt.c[0][1]) + 2*r*(dL_dMt.c[2][0] - dL_dMt.c[0][2]) + 2*z*(dL_dMt.c[1][2] + dL_dMt.c[2][1]) - 4*y*(dL_dMt.c[2][2] + dL_dMt.c[0][0]);
    q[3] = 2*r*(dL_dMt.c[0][1] - dL_dMt.c[1][0]) + 2*x*(dL_dMt.c[2][0] + dL_dMt.c[0][2]) + 2*y*(dL_dMt.c[1][2] + dL_dMt.c[2][1]) - 4*z*(dL_dMt.c[1][1] + dL_dMt.c[0][0]);
}

/* SURFEL backward.cu:450-580 */
static void compute_transmat_aabb_bwd(const ref_state* st, const ref_inputs* in, int idx, int W, int H,
                                      const real* dL_dnormals, const real* dL_dmean2Ds, real* dL_dTs,
                                      real* dL_dmeans, real* dL_dscales, real* dL_drots)
{
    const int precomp = (in->scales == NULL);
    real T[9]; f3 normal = {0,0,0}; real Pm[4][3]; m3 R; f3 p_orig = {0,0,0};
    const real* rot = NULL; const real* scale = NULL;
    if (precomp) {
        memcpy(T, in->cov3D_precomp + 9*idx, sizeof(T));
    } else {
        p_orig.x = in->means3D[3*idx]; p_orig.y = in->means3D[3*idx+1]; p_orig.z = in->means3D[3*idx+2];
        rot = in->rotations + 4*idx; scale = in->scales + 2*idx;
        R = quat_to_rotmat(rot);
        /* fork quirk (SURFEL backward.cu:488): scale_modifier is NOT applied in the backward recompute */
        compute_transmat(p_orig, scale, 1.0f, rot, in->projmatrix, in->viewmatrix, W, H, T, &normal);
        surfel_P(in->projmatrix, W, H, Pm);
    }
    (void)st;
    real dT[3][3];   /* dT[j] = dL/d(T_j), j: 0=Tu 1=Tv 2=Tw */
    for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) dT[j][i] = dL_dTs[9*idx + 3*j + i];
    const real gmx = dL_dmean2Ds[3*idx], gmy = dL_dmean2Ds[3*idx+1];
    if (gmx != 0 || gmy != 0) {
        const real* Tu = T; const real* Tv = T + 3; const real* Tw = T + 6;
        real tv[3] = { 9.0f, 9.0f, -1.0f };
        real ww[3] = { Tw[0]*Tw[0], Tw[1]*Tw[1], Tw[2]*Tw[2] };
        real d = dot3(tv, ww);
        real inv = 1.0f / d;
        real f[3] = { tv[0]*inv, tv[1]*inv, tv[2]*inv };
        real dT0[3], dT1[3], dT3[3], dL_df[3];
        for (int i = 0; i < 3; i++) {
            dT0[i] = gmx * f[i] * Tw[i];
            dT1[i] = gmy * f[i] * Tw[i];
            dT3[i] = gmx * f[i] * Tu[i] + gmy * f[i] * Tv[i];
            dL_df[i] = gmx * Tu[i] * Tw[i] + gmy * Tv[i] * Tw[i];
        }
        real dL_dd = (real)(dot3(dL_df, f) * (-1.0 / d));
        for (int i = 0; i < 3; i++) {
            real dd_dT3 = tv[i] * Tw[i] * 2.0f;
            dT3[i] += dL_dd * dd_dT3;
            dT[0][i] += dT0[i]; dT[1][i] += dT1[i]; dT[2][i] += dT3[i];
        }
        if (precomp) {
            for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) dL_dTs[9*idx + 3*j + i] = dT[j][i];
            return;
        }
    }
    if (precomp) return;
    /* dL_dM = P * transpose(dL_dT): dM[j][k] = sum_c Pm[k][c] * dT[c][j]   (j: 0=L0 row,1=L1 row,2=centre row) */
    real dM[3][4];
    for (int j = 0; j < 3; j++)
        for (int k = 0; k < 4; k++)
            dM[j][k] = Pm[k][0]*dT[0][j] + Pm[k][1]*dT[1][j] + Pm[k][2]*dT[2][j];
    f3 dn = { dL_dnormals[3*idx], dL_dnormals[3*idx+1], dL_dnormals[3*idx+2] };
    f3 dL_dtn = transformVec4x3Transpose(dn, in->viewmatrix);
    f3 p_view = transformPoint4x3(p_orig, in->viewmatrix);
    real cosv = -(p_view.x*normal.x + p_view.y*normal.y + p_view.z*normal.z);
    real mult = cosv > 0 ? 1.f : -1.f;
    dL_dtn.x *= mult; dL_dtn.y *= mult; dL_dtn.z *= mult;
    m3 dL_dRS = m3_cols(dM[0][0], dM[0][1], dM[0][2], dM[1][0], dM[1][1], dM[1][2], dL_dtn.x, dL_dtn.y, dL_dtn.z);
    m3 dL_dR = m3_cols(dL_dRS.c[0][0]*scale[0], dL_dRS.c[0][1]*scale[0], dL_dRS.c[0][2]*scale[0],
                       dL_dRS.c[1][0]*scale[1], dL_dRS.c[1][1]*scale[1], dL_dRS.c[1][2]*scale[1],
                       dL_dRS.c[2][0], dL_dRS.c[2][1], dL_dRS.c[2][2]);
    quat_to_rotmat_vjp(rot, dL_dR, dL_drots + 4*idx);
    dL_dscales[2*idx+0] = dot3(dL_dRS.c[0], R.c[0]);
    dL_dscales[2*idx+1] = dot3(dL_dRS.c[1], R.c[1]);
    dL_dmeans[3*idx+0] = dM[2][0]; dL_dmeans[3*idx+1] = dM[2][1]; dL_dmeans[3*idx+2] = dM[2][2];
}

void ref_backward(ref_state* st, const ref_inputs* in, const ref_out_grads* og, ref_in_grads* ig)
{
    const int P = st->P, W = st->W, H = st->H, M = in->M;
    const int surf = st->variant == REF_SURFEL;
    const int tm = surf ? 9 : 6;
    memset(ig->dL_dmeans3D, 0, (size_t)P*3*RS); memset(ig->dL_dmeans2D, 0, (size_t)P*3*RS);
    if (ig->dL_dmeans2D_abs) memset(ig->dL_dmeans2D_abs, 0, (size_t)P*3*RS);
    memset(ig->dL_dcolors, 0, (size_t)P*3*RS); memset(ig->dL_dopacity, 0, (size_t)P*RS);
    memset(ig->dL_dcov3D, 0, (size_t)P*tm*RS);
    if (ig->dL_dsh && M > 0) memset(ig->dL_dsh, 0, (size_t)P*M*3*RS);
    memset(ig->dL_dscales, 0, (size_t)P*(surf ? 2 : 3)*RS); memset(ig->dL_drotations, 0, (size_t)P*4*RS);
    if (ig->dL_dall_map) memset(ig->dL_dall_map, 0, (size_t)P*5*RS);
    memset(ig->dL_dconic, 0, (size_t)P*(surf ? 3 : 4)*RS);

    const real focal_y = H / (2.0f * in->tanfovy);
    const real focal_x = W / (2.0f * in->tanfovx);
    const real* colors = in->colors_precomp ? in->colors_precomp : st->rgb;

#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int tile = 0; tile < st->T; tile++) {
        if (tile % g_tile_stride) continue;
        uint32_t tx = (uint32_t)(tile % st->gx), ty = (uint32_t)(tile / st->gx);
        uint32_t r0 = st->ranges[2*tile], r1 = st->ranges[2*tile+1];
        for (uint32_t ly = 0; ly < BLOCK_Y; ly++)
            for (uint32_t lx = 0; lx < BLOCK_X; lx++) {
                uint32_t px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (px < (uint32_t)W && py < (uint32_t)H)
                    blend_pixel_bwd(st, in, colors, focal_x, focal_y, px, py, r0, r1, og,
                                    ig->dL_dmeans2D, ig->dL_dmeans2D_abs, surf ? NULL : ig->dL_dconic,
                                    surf ? ig->dL_dconic : NULL, surf ? ig->dL_dcov3D : NULL, ig->dL_dopacity,
                                    ig->dL_dcolors, ig->dL_dall_map);
            }
    }

    if (!surf) {
        /* BACKWARD::preprocess, 3DGS backward.cu:559-625 */
        const real* cov3D_ptr = in->cov3D_precomp ? in->cov3D_precomp : st->cov3D;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
        for (int idx = 0; idx < P; idx++) {
            if (!(st->radii[idx] > 0)) continue;
            computeCov2D_bwd(in, idx, cov3D_ptr, focal_x, focal_y, ig->dL_dconic, ig->dL_dmeans3D, ig->dL_dcov3D);
            /* preprocessCUDA bwd, 3DGS backward.cu:346-396 */
            f3 m = { in->means3D[3*idx], in->means3D[3*idx+1], in->means3D[3*idx+2] };
            const real* proj = in->projmatrix;
            real mh[4]; transformPoint4x4(m, proj, mh);
            real m_w = 1.0f / (mh[3] + 0.0000001f);
            real mul1 = (proj[0]*m.x + proj[4]*m.y + proj[8]*m.z + proj[12]) * m_w * m_w;
            real mul2 = (proj[1]*m.x + proj[5]*m.y + proj[9]*m.z + proj[13]) * m_w * m_w;
            const real gx_ = ig->dL_dmeans2D[3*idx], gy_ = ig->dL_dmeans2D[3*idx+1];
            real dmx = (proj[0]*m_w - proj[3]*mul1) * gx_ + (proj[1]*m_w - proj[3]*mul2) * gy_;
            real dmy = (proj[4]*m_w - proj[7]*mul1) * gx_ + (proj[5]*m_w - proj[7]*mul2) * gy_;
            real dmz = (proj[8]*m_w - proj[11]*mul1) * gx_ + (proj[9]*m_w - proj[11]*mul2) * gy_;
            ig->dL_dmeans3D[3*idx] += dmx; ig->dL_dmeans3D[3*idx+1] += dmy; ig->dL_dmeans3D[3*idx+2] += dmz;
            if (in->shs)
                computeColorFromSH_bwd(idx, in->D, M, in->means3D, in->campos, in->shs, st->clamped,
                                       ig->dL_dcolors, ig->dL_dmeans3D, ig->dL_dsh);
            if (in->scales)
                computeCov3D_bwd(idx, in->scales + 3*idx, in->scale_modifier, in->rotations + 4*idx,
                                 ig->dL_dcov3D, ig->dL_dscales, ig->dL_drotations);
        }
    } else {
        /* SURFEL backward.cu:582-637.  fork quirk: W,H are re-derived from focal*tan*2 in float32 */
        const int Wb = (int)(focal_x * in->tanfovx * 2);
        const int Hb = (int)(focal_y * in->tanfovy * 2);
        const real* transMats = in->cov3D_precomp ? in->cov3D_precomp : st->cov3D;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
        for (int idx = 0; idx < P; idx++) {
            if (!(st->radii[idx] > 0)) continue;
            compute_transmat_aabb_bwd(st, in, idx, Wb, Hb, ig->dL_dconic, ig->dL_dmeans2D, ig->dL_dcov3D,
                                      ig->dL_dmeans3D, ig->dL_dscales, ig->dL_drotations);
            if (in->shs)
                computeColorFromSH_bwd(idx, in->D, M, in->means3D, in->campos, in->shs, st->clamped,
                                       ig->dL_dcolors, ig->dL_dmeans3D, ig->dL_dsh);
            /* densification proxy overwrites dL_dmean2D, SURFEL backward.cu:633-636 */
            real depth = transMats[9*idx + 8];
            ig->dL_dmeans2D[3*idx+0] = (real)(ig->dL_dcov3D[9*idx + 2] * depth * 0.5 * (real)Wb);
            ig->dL_dmeans2D[3*idx+1] = (real)(ig->dL_dcov3D[9*idx + 5] * depth * 0.5 * (real)Hb);
        }
    }
}

/* ------------------------------------------------------------------ introspection */
int32_t ref_num_rendered(const ref_state* st) { return st->R; }
int32_t ref_num_tiles(const ref_state* st) { return st->T; }
void ref_get_pair_counts(const ref_state* st, uint64_t* evaluated, uint64_t* contributing) { *evaluated = st->pairs_evaluated; *contributing = st->pairs_contributing; }
void ref_set_tile_stride(int32_t n) { g_tile_stride = n > 0 ? n : 1; }
void ref_set_threads(int32_t n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
void ref_get_point_list(const ref_state* st, uint32_t* out) { memcpy(out, st->point_list, (size_t)st->R * 4); }
void ref_get_keys(const ref_state* st, uint64_t* out) { memcpy(out, st->keys, (size_t)st->R * 8); }
void ref_get_ranges(const ref_state* st, uint32_t* out) { memcpy(out, st->ranges, (size_t)st->T * 8); }
void ref_get_tiles_touched(const ref_state* st, uint32_t* out) { memcpy(out, st->tiles_touched, (size_t)st->P * 4); }
void ref_get_geom(const ref_state* st, real* depths, real* means2D, real* conic_opacity, real* rgb, real* cov)
{
    size_t P = (size_t)st->P;
    if (depths) memcpy(depths, st->depths, P*RS);
    if (means2D) memcpy(means2D, st->means2D, P*2*RS);
    if (conic_opacity) memcpy(conic_opacity, st->conic_opacity, P*4*RS);
    if (rgb) memcpy(rgb, st->rgb, P*3*RS);
    if (cov) memcpy(cov, st->cov3D, P*(st->variant == REF_SURFEL ? 9 : 6)*RS);
}
void ref_get_image_state(const ref_state* st, real* final_T, uint32_t* n_contrib)
{
    size_t N = (size_t)st->N;
    if (final_T) memcpy(final_T, st->final_T, N * (st->variant == REF_SURFEL ? 3 : 1) * RS);
    if (n_contrib) memcpy(n_contrib, st->n_contrib, N * (st->variant == REF_SURFEL ? 2 : 1) * 4);
}

/* ------------------------------------------------------------------ scaffold-filter & markVisible */
/* FILTER forward.cu:268-340 via rasterizer_impl.cu:340-396 */
void ref_visible_filter(const ref_inputs* in, int32_t* radii)
{
    const int P = in->P, W = in->W, H = in->H;
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    const real focal_y = H / (2.0f * in->tanfovy);
    const real focal_x = W / (2.0f * in->tanfovx);
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int idx = 0; idx < P; idx++) {
        radii[idx] = 0;
        f3 p_view;
        if (!in_frustum(idx, in->means3D, in->viewmatrix, &p_view)) continue;
        f3 p_orig = { in->means3D[3*idx], in->means3D[3*idx+1], in->means3D[3*idx+2] };
        real ph[4]; transformPoint4x4(p_orig, in->projmatrix, ph);
        real p_w = 1.0f / (ph[3] + 0.0000001f);
        real cov3Dl[6]; const real* cov3D;
        if (in->cov3D_precomp) cov3D = in->cov3D_precomp + 6*idx;
        else { computeCov3D(in->scales + 3*idx, in->scale_modifier, in->rotations + 4*idx, cov3Dl); cov3D = cov3Dl; }
        real cov[3];
        computeCov2D(p_orig, focal_x, focal_y, in->tanfovx, in->tanfovy, cov3D, in->viewmatrix, cov, NULL, NULL, NULL, NULL, NULL);
        real det = (cov[0] * cov[2] - cov[1] * cov[1]);
        if (det == 0.0f) continue;
        real mid = 0.5f * (cov[0] + cov[2]);
        real lambda1 = mid + R_sqrt(R_fmax(0.1f, mid * mid - det));
        real lambda2 = mid - R_sqrt(R_fmax(0.1f, mid * mid - det));
        real my_radius = R_ceil(3.f * R_sqrt(R_fmax(lambda1, lambda2)));
        f2 pi = { ndc2Pix(ph[0] * p_w, W), ndc2Pix(ph[1] * p_w, H) };
        uint32_t rmin[2], rmax[2];
        getRect(pi, (int)my_radius, gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
        radii[idx] = (int)my_radius;
    }
}

/* 3DGS rasterizer_impl.cu:54-66,141-153 */
void ref_mark_visible(int32_t P, const real* means3D, const real* viewmatrix, const real* projmatrix, uint8_t* present)
{
    (void)projmatrix;
    for (int i = 0; i < P; i++) { f3 pv; present[i] = (uint8_t)in_frustum(i, means3D, viewmatrix, &pv); }
}

/* ------------------------------------------------------------------ TSDF (gssr/utils/mesh_utils.py:195-246) */
/* torch.nn.functional.grid_sample(mode='bilinear', padding_mode='border', align_corners=True) for one sample */
static float bilinear_border(const float* img, int W, int H, float u, float v)
{
    /* align_corners=True: x = (u+1)/2*(W-1); border padding clamps the coordinate into [0, size-1] */
    float x = ((u + 1.f) / 2.f) * (float)(W - 1);
    float y = ((v + 1.f) / 2.f) * (float)(H - 1);
    x = fminf(fmaxf(x, 0.f), (float)(W - 1));
    y = fminf(fmaxf(y, 0.f), (float)(H - 1));
    int x0 = (int)floorf(x), y0 = (int)floorf(y);
    int x1 = x0 + 1, y1 = y0 + 1;
    /* weights as ATen's grid_sampler: (x1 - x), (x - x0) */
    float wx1 = x - (float)x0, wy1 = y - (float)y0, wx0 = (float)x1 - x, wy0 = (float)y1 - y;
    float acc = 0.f;
    /* out-of-range corners contribute zero (their weight is zero after clamping, restated like the torch kernel) */
    if (x0 >= 0 && x0 < W && y0 >= 0 && y0 < H) acc += img[(size_t)y0*W + x0] * (wx0 * wy0);
    if (x1 >= 0 && x1 < W && y0 >= 0 && y0 < H) acc += img[(size_t)y0*W + x1] * (wx1 * wy0);
    if (x0 >= 0 && x0 < W && y1 >= 0 && y1 < H) acc += img[(size_t)y1*W + x0] * (wx0 * wy1);
    if (x1 >= 0 && x1 < W && y1 >= 0 && y1 < H) acc += img[(size_t)y1*W + x1] * (wx1 * wy1);
    return acc;
}

void ref_tsdf_integrate(int64_t V, const float* points, const float* F, int32_t W, int32_t H, const float* depth,
                        const float* rgb, float sdf_trunc, const float* trunc_pp, float* tsdf, float* weight,
                        float* rgb_acc)
{
    const size_t HW = (size_t)W * H;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < V; i++) {
        float x = points[3*i], y = points[3*i+1], z3 = points[3*i+2];
        /* [x y z 1] @ full_proj_transform (row-vector convention) */
        float qx = x*F[0] + y*F[4] + z3*F[8] + F[12];
        float qy = x*F[1] + y*F[5] + z3*F[9] + F[13];
        float qw = x*F[3] + y*F[7] + z3*F[11] + F[15];
        float z = qw;
        float u = qx / qw, v = qy / qw;
        int mask = (u > -1.f) && (u < 1.f) && (v > -1.f) && (v < 1.f) && (z > 0);
        float d = bilinear_border(depth, W, H, u, v);
        float sdf = d - z;
        float tr = trunc_pp ? trunc_pp[i] : sdf_trunc;
        mask = mask && (sdf > -tr);
        if (!mask) continue;
        float s = sdf / tr; s = fminf(fmaxf(s, -1.0f), 1.0f);
        float w = weight[i], wp = w + 1;
        tsdf[i] = (tsdf[i] * w + s) / wp;
        for (int c = 0; c < 3; c++) {
            float col = bilinear_border(rgb + c*HW, W, H, u, v);
            rgb_acc[3*i + c] = (rgb_acc[3*i + c] * w + col) / wp;
        }
        weight[i] = wp;
    }
}

/* Open3D 0.18.0 (pinned at requirements.txt:6, absent from /root/reference): UniformTSDFVolume::Integrate restated from
   its published algorithm; call sites gssr/utils/mesh_utils.py:154-178, extract_mesh_split.py:91-119.  PARITY UNPINNED. */
void ref_tsdf_integrate_dense(int32_t nx, int32_t ny, int32_t nz, const float* origin, float vl, float trunc, float dtrunc,
                              int32_t W, int32_t H, const float* depth, const float* rgb, float fx, float fy, float cx, float cy,
                              const float* E, float* tsdf, float* weight, float* color)
{
    const int64_t V = (int64_t)nx * ny * nz;
    const size_t HW = (size_t)W * H;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < V; i++) {
        const int iz = (int)(i % nz), iy = (int)((i / nz) % ny), ix = (int)(i / ((int64_t)nz * ny));
        const float x = origin[0] + vl * ((float)ix + 0.5f), y = origin[1] + vl * ((float)iy + 0.5f), z = origin[2] + vl * ((float)iz + 0.5f);
        const float xc = E[0]*x + E[1]*y + E[2]*z + E[3];
        const float yc = E[4]*x + E[5]*y + E[6]*z + E[7];
        const float zc = E[8]*x + E[9]*y + E[10]*z + E[11];
        if (!(zc > 0.f)) continue;
        const float uf = xc * fx / zc + cx + 0.5f, vf = yc * fy / zc + cy + 0.5f;
        if (!(uf >= 0.f && uf < (float)W && vf >= 0.f && vf < (float)H)) continue;
        const int u = (int)uf, v = (int)vf;
        const float d = depth[(size_t)v * W + u];
        if (!(d > 0.f) || d > dtrunc) continue;
        const float rx = ((float)u - cx) / fx, ry = ((float)v - cy) / fy;
        const float sdf = (d - zc) * sqrtf(rx*rx + ry*ry + 1.0f);
        if (!(sdf > -trunc)) continue;
        const float t = fminf(1.0f, sdf / trunc);
        const float w = weight[i], wp = w + 1.0f;
        tsdf[i] = (tsdf[i] * w + t) / wp;
        for (int c = 0; c < 3; c++) color[3*i + c] = (color[3*i + c] * w + rgb[c*HW + (size_t)v * W + u]) / wp;
        weight[i] = wp;
    }
}

/* ---- block-sparse TSDF volume: CPU restatement of csrc/gsr_tsdf_sparse.hip, i.e. of Open3D 0.18's ScalableTSDFVolume::Integrate
   (cpp/open3d/pipelines/integration/ScalableTSDFVolume.cpp) as the reference drives it (gssr/utils/mesh_utils.py:154-178).
   PARITY UNPINNED: Open3D is a pip dependency of the reference (requirements.txt:6), absent here, and no reference test holds vectors.
   Units of 16^3 voxels keyed by floor(p / unit_len); per frame every `stride`-th valid depth pixel opens the units overlapping
   [p - trunc, p + trunc]; each opened unit is integrated once with the uniform-volume voxel rule (ref_tsdf_integrate_dense). */
#include <stdlib.h>
#include <string.h>
struct RefSparse {
    float vl, trunc;
    int n, cap;
    int32_t* coord;     /* [n][3] in first-touch order */
    float *tsdf, *weight, *color;
};
typedef struct RefSparse RefSparse;
RefSparse* ref_tsdf_sparse_new(float voxel_length, float sdf_trunc)
{
    RefSparse* v = (RefSparse*)calloc(1, sizeof(RefSparse));
    v->vl = voxel_length; v->trunc = sdf_trunc;
    return v;
}
void ref_tsdf_sparse_free(RefSparse* v)
{
    if (!v) return;
    free(v->coord); free(v->tsdf); free(v->weight); free(v->color); free(v);
}
static int sparse_find(const RefSparse* v, int x, int y, int z)
{
    for (int i = v->n - 1; i >= 0; i--)      /* newest first: neighbouring pixels hit the units opened a moment ago */
        if (v->coord[3*i] == x && v->coord[3*i+1] == y && v->coord[3*i+2] == z) return i;
    return -1;
}
static int sparse_open(RefSparse* v, int x, int y, int z)
{
    int i = sparse_find(v, x, y, z);
    if (i >= 0) return i;
    if (v->n == v->cap) {
        const int nc = v->cap ? 2 * v->cap : 256;
        v->coord = (int32_t*)realloc(v->coord, (size_t)nc * 3 * sizeof(int32_t));
        v->tsdf = (float*)realloc(v->tsdf, (size_t)nc * 4096 * sizeof(float));
        v->weight = (float*)realloc(v->weight, (size_t)nc * 4096 * sizeof(float));
        v->color = (float*)realloc(v->color, (size_t)nc * 4096 * 3 * sizeof(float));
        v->cap = nc;
    }
    i = v->n++;
    v->coord[3*i] = x; v->coord[3*i+1] = y; v->coord[3*i+2] = z;
    memset(v->tsdf + (size_t)i * 4096, 0, 4096 * sizeof(float));
    memset(v->weight + (size_t)i * 4096, 0, 4096 * sizeof(float));
    memset(v->color + (size_t)i * 4096 * 3, 0, 4096 * 3 * sizeof(float));
    return i;
}
void ref_tsdf_sparse_integrate(RefSparse* v, int32_t W, int32_t H, const float* depth, const float* rgb, float fx, float fy, float cx, float cy,
                               const float* E /*[12] world->camera*/, const float* P /*[12] camera->world*/, float dtrunc, int32_t stride)
{
    const float unit_len = v->vl * 16, inv_unit = 1.0f / unit_len, rfx = 1.0f / fx, rfy = 1.0f / fy;
    int ntouched = 0, tcap = 1024;
    int* touched = (int*)malloc((size_t)tcap * sizeof(int));
    const int n0 = v->n;
    unsigned char* seen = (unsigned char*)calloc((size_t)(n0 > 0 ? n0 : 1), 1);      /* units that existed before this frame */
    for (int vv = 0; vv < H; vv += stride)
        for (int u = 0; u < W; u += stride) {
            const float d = depth[(size_t)vv * W + u];
            if (!(d > 0.f) || d > dtrunc) continue;
            const float xc = ((float)u - cx) * d * rfx, yc = ((float)vv - cy) * d * rfy;
            const float p[3] = { P[0]*xc + P[1]*yc + P[2]*d + P[3], P[4]*xc + P[5]*yc + P[6]*d + P[7], P[8]*xc + P[9]*yc + P[10]*d + P[11] };
            int lo[3], hi[3], ok = 1;
            for (int a = 0; a < 3; a++) {
                lo[a] = (int)floorf((p[a] - v->trunc) * inv_unit);
                hi[a] = (int)floorf((p[a] + v->trunc) * inv_unit);
                if (lo[a] < -(1 << 20) + 1 || hi[a] > (1 << 20) - 2 || hi[a] - lo[a] > 3) ok = 0;
            }
            if (!ok) continue;
            for (int x = lo[0]; x <= hi[0]; x++)
                for (int y = lo[1]; y <= hi[1]; y++)
                    for (int z = lo[2]; z <= hi[2]; z++) {
                        const int before = v->n;
                        const int i = sparse_open(v, x, y, z);
                        const int is_new = (v->n != before);
                        if (is_new || (i < n0 && !seen[i])) {
                            if (i < n0) seen[i] = 1;
                            if (ntouched == tcap) { tcap *= 2; touched = (int*)realloc(touched, (size_t)tcap * sizeof(int)); }
                            touched[ntouched++] = i;
                        }
                    }
        }
    float E16[16];
    for (int k = 0; k < 12; k++) E16[k] = E[k];
    E16[12] = E16[13] = E16[14] = 0.f; E16[15] = 1.f;
    for (int t = 0; t < ntouched; t++) {
        const int i = touched[t];
        const float origin[3] = { (float)v->coord[3*i] * unit_len, (float)v->coord[3*i+1] * unit_len, (float)v->coord[3*i+2] * unit_len };
        ref_tsdf_integrate_dense(16, 16, 16, origin, v->vl, v->trunc, dtrunc, W, H, depth, rgb, fx, fy, cx, cy, E16,
                                 v->tsdf + (size_t)i * 4096, v->weight + (size_t)i * 4096, v->color + (size_t)i * 4096 * 3);
    }
    free(touched); free(seen);
}
int32_t ref_tsdf_sparse_num_units(const RefSparse* v) { return v->n; }
void ref_tsdf_sparse_get(const RefSparse* v, int32_t* coord, float* tsdf, float* weight, float* color)
{
    memcpy(coord, v->coord, (size_t)v->n * 3 * sizeof(int32_t));
    memcpy(tsdf, v->tsdf, (size_t)v->n * 4096 * sizeof(float));
    memcpy(weight, v->weight, (size_t)v->n * 4096 * sizeof(float));
    memcpy(color, v->color, (size_t)v->n * 4096 * 3 * sizeof(float));
}

/* simple-knn (submodules/simple-knn/simple_knn.cu:148-184): mean of the squared distances to the 3 nearest
   neighbours.  Brute force; the Morton/box pruning of the source is an acceleration structure only. */
void ref_dist2(int32_t P, const float* pts, float* out)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int i = 0; i < P; i++) {
        float best[3] = { 3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f };
        for (int j = 0; j < P; j++) {
            if (j == i) continue;
            float dx = pts[3*i] - pts[3*j], dy = pts[3*i+1] - pts[3*j+1], dz = pts[3*i+2] - pts[3*j+2];
            float d = dx*dx + dy*dy + dz*dz;
            for (int k = 0; k < 3; k++) if (d < best[k]) { float t = best[k]; best[k] = d; d = t; }
        }
        out[i] = (best[0] + best[1] + best[2]) / 3;
    }
}

int32_t ref_omp_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
