/*
 * gsl_oracle.c -- CPU restatement of the photometric loss of the training loop.  TEST INFRASTRUCTURE ONLY (see gsr_oracle.h).
 *   loss = (1-lambda)*mean|img-gt| + lambda*(1-SSIM)      gssr/scene/vanilla_scene.py:29-69 (l1_loss, _ssim, _gaussian, ssim, get_loss_dict)
 * Direct (non-separable) 11x11 correlation with zero padding, like F.conv2d(padding=5, groups=C) with the outer-product window; the
 * gradient is the chain rule of those ops, checked against torch autograd of the reference's formula in tests/.
 * PARITY STATUS: PINNED against the reference itself run in the authoring container (tests/golden/make_golden_ref.py calls
 * VanillaScene.get_loss_dict, TwoDGSScene.render post-processing + get_loss_dict, PGSRScene.render_normal/_get_img_grad_weight/erode on CPU
 * torch; vectors in tests/golden/ref_loss_*.npz, checked by tests/test_golden_ref_cpu.py); randomized cases in tests/test_loss_cpu.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static void window(float w2[11][11])
{
    float g[11], s = 0.f;
    for (int i = 0; i < 11; i++) { g[i] = (float)exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); s += g[i]; }
    for (int i = 0; i < 11; i++) g[i] /= s;                       /* vanilla_scene.py:49-51 */
    for (int i = 0; i < 11; i++) for (int j = 0; j < 11; j++) w2[i][j] = g[i] * g[j];   /* _1D_window.mm(_1D_window.t()) :55-56 */
}

/* loss_out[3] = {mean|img-gt|, mean SSIM, loss}; dL_dimg [C,H,W] */
void ref_loss_l1_ssim(int32_t C, int32_t H, int32_t W, const float* img, const float* gt, float lambda, float* loss_out, float* dL_dimg)
{
    float w2[11][11];
    window(w2);
    const size_t N = (size_t)C * H * W;
    float* dmu = malloc(N * sizeof(float)); float* d11 = malloc(N * sizeof(float)); float* d12 = malloc(N * sizeof(float));
    double sum_l1 = 0.0, sum_ss = 0.0;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
#pragma omp parallel for collapse(2) reduction(+ : sum_l1, sum_ss) schedule(static)
    for (int c = 0; c < C; c++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const float* ip = img + (size_t)c * H * W; const float* gp = gt + (size_t)c * H * W;
                double m1 = 0, m2 = 0, e11 = 0, e22 = 0, e12 = 0;
                for (int i = 0; i < 11; i++) {
                    const int yy = y + i - 5;
                    if (yy < 0 || yy >= H) continue;
                    for (int j = 0; j < 11; j++) {
                        const int xx = x + j - 5;
                        if (xx < 0 || xx >= W) continue;
                        const double w = w2[i][j], u = ip[(size_t)yy * W + xx], v = gp[(size_t)yy * W + xx];
                        m1 += w * u; m2 += w * v; e11 += w * u * u; e22 += w * v * v; e12 += w * u * v;
                    }
                }
                const float mu1 = (float)m1, mu2 = (float)m2;
                const float s1 = (float)e11 - mu1 * mu1, s2 = (float)e22 - mu2 * mu2, s12 = (float)e12 - mu1 * mu2;
                const float A1 = 2.f * mu1 * mu2 + C1, A2 = 2.f * s12 + C2, B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s1 + s2 + C2;
                const float S = (A1 * A2) / (B1 * B2);                                      /* :43 */
                const size_t o = ((size_t)c * H + y) * W + x;
                dmu[o] = 2.f * mu2 * (A2 - A1) / (B1 * B2) - S * 2.f * mu1 * (1.f / B1 - 1.f / B2);
                d11[o] = -S / B2;
                d12[o] = 2.f * A1 / (B1 * B2);
                sum_ss += S;
                sum_l1 += fabs((double)ip[(size_t)y * W + x] - (double)gp[(size_t)y * W + x]);
            }
    const float inv = 1.0f / (float)N;
    loss_out[0] = (float)(sum_l1 / (double)N); loss_out[1] = (float)(sum_ss / (double)N);
    loss_out[2] = (1.f - lambda) * loss_out[0] + lambda * (1.f - loss_out[1]);
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; c++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const size_t pl = (size_t)c * H * W, o = pl + (size_t)y * W + x;
                double a = 0, b = 0, d = 0;
                for (int i = 0; i < 11; i++) {
                    const int yy = y + i - 5;
                    if (yy < 0 || yy >= H) continue;
                    for (int j = 0; j < 11; j++) {
                        const int xx = x + j - 5;
                        if (xx < 0 || xx >= W) continue;
                        const double w = w2[i][j];                    /* symmetric window: adjoint == same correlation */
                        const size_t q = pl + (size_t)yy * W + xx;
                        a += w * dmu[q]; b += w * d11[q]; d += w * d12[q];
                    }
                }
                const float u = img[o], v = gt[o], df = u - v;
                const float sgn = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
                dL_dimg[o] = (1.f - lambda) * inv * sgn - lambda * inv * (float)(a + 2.0 * u * b + v * d);
            }
    free(dmu); free(d11); free(d12);
}

/* ------------------------------------------------------------------------------------------------------------------------
 * 2DGS geometric regularisers: the post-processing of TwoDGSScene.render (gssr/scene/twodgs_scene.py:88-115), depth_to_normal /
 * depths_to_points (gssr/utils/point_utils.py:9-37) and the normal / distortion losses (twodgs_scene.py:25-35).
 *   alpha = allmap[1]; d_exp = nan_to_num(allmap[0]/alpha, 0, 0); d_med = nan_to_num(allmap[5], 0, 0); depth = d_exp (1-r) + r d_med
 *   P(y,x) = depth * ([x y 1] * ray_mat)          (+ rays_o, which cancels in the differences)
 *   interior: n = normalize(cross(P(y+1,x)-P(y-1,x), P(y,x+1)-P(y,x-1))), border: 0;  surf_normal = n * alpha (alpha detached)
 *   normal_world = allmap[2:5] * normal_rot (row vector);  normal_error = 1 - <normal_world, surf_normal>
 *   loss = lambda_normal * mean(normal_error) + lambda_dist * mean(allmap[6])
 * loss_out[3] = {mean normal_error, mean allmap[6], loss}.  dL_dallmap [11,H,W].  Optional outputs may be NULL.
 */
static float nan0(float v) { return (isnan(v) || isinf(v)) ? 0.f : v; }

void ref_loss_surfel_geo(int32_t H, int32_t W, const float* allmap, const float* ray_mat, const float* normal_rot, float depth_ratio,
                         float lambda_normal, float lambda_dist, float* loss_out, float* dL_dallmap, float* out_surf_depth,
                         float* out_normal_world, float* out_surf_normal)
{
    const size_t N = (size_t)H * W;
    float* P = malloc(3 * N * sizeof(float));
    float* dP = calloc(3 * N, sizeof(float));
    float* depth = malloc(N * sizeof(float));
    memset(dL_dallmap, 0, 11 * N * sizeof(float));
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t o = (size_t)y * W + x;
            const float a = allmap[N + o];
            const float de = nan0(allmap[o] / a), dm = nan0(allmap[5 * N + o]);
            const float d = de * (1.f - depth_ratio) + depth_ratio * dm;
            depth[o] = d;
            for (int c = 0; c < 3; c++) P[3 * o + c] = d * ((float)x * ray_mat[c] + (float)y * ray_mat[3 + c] + ray_mat[6 + c]);
            if (out_surf_depth) out_surf_depth[o] = d;
        }
    double s_err = 0.0, s_dist = 0.0;
    const float wn = lambda_normal / (float)N, wd = lambda_dist / (float)N;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t o = (size_t)y * W + x;
            const float a = allmap[N + o];
            float nv[3] = {allmap[2 * N + o], allmap[3 * N + o], allmap[4 * N + o]}, nw[3], n[3] = {0, 0, 0};
            for (int c = 0; c < 3; c++) nw[c] = nv[0] * normal_rot[c] + nv[1] * normal_rot[3 + c] + nv[2] * normal_rot[6 + c];
            const int interior = y >= 1 && y <= H - 2 && x >= 1 && x <= W - 2;
            float dx[3], dy[3], cr[3], len = 0.f;
            if (interior) {
                for (int c = 0; c < 3; c++) {
                    dx[c] = P[3 * (o + W) + c] - P[3 * (o - W) + c];
                    dy[c] = P[3 * (o + 1) + c] - P[3 * (o - 1) + c];
                }
                cr[0] = dx[1] * dy[2] - dx[2] * dy[1]; cr[1] = dx[2] * dy[0] - dx[0] * dy[2]; cr[2] = dx[0] * dy[1] - dx[1] * dy[0];
                len = sqrtf(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
                const float den = len > 1e-12f ? len : 1e-12f;
                for (int c = 0; c < 3; c++) n[c] = cr[c] / den;
            }
            float dot = 0.f;
            for (int c = 0; c < 3; c++) dot += nw[c] * n[c] * a;
            s_err += 1.0 - (double)dot;
            s_dist += allmap[6 * N + o];
            dL_dallmap[6 * N + o] = wd;
            /* d/d normal (view space): -wn * surf_normal rotated back */
            for (int i = 0; i < 3; i++) {
                float g = 0.f;
                for (int c = 0; c < 3; c++) g += normal_rot[3 * i + c] * (-wn * n[c] * a);
                dL_dallmap[(2 + i) * N + o] = g;
            }
            for (int c = 0; c < 3; c++) {
                if (out_normal_world) out_normal_world[c * N + o] = nw[c];
                if (out_surf_normal) out_surf_normal[c * N + o] = n[c] * a;
            }
            if (interior) {
                float dn[3], dc[3], nd = 0.f;
                for (int c = 0; c < 3; c++) { dn[c] = -wn * a * nw[c]; nd += n[c] * dn[c]; }
                if (len > 1e-12f) for (int c = 0; c < 3; c++) dc[c] = (dn[c] - n[c] * nd) / len;
                else for (int c = 0; c < 3; c++) dc[c] = dn[c] / 1e-12f;
                /* c = dx x dy: d dx = dy x dc, d dy = dc x dx */
                const float gdx[3] = {dy[1] * dc[2] - dy[2] * dc[1], dy[2] * dc[0] - dy[0] * dc[2], dy[0] * dc[1] - dy[1] * dc[0]};
                const float gdy[3] = {dc[1] * dx[2] - dc[2] * dx[1], dc[2] * dx[0] - dc[0] * dx[2], dc[0] * dx[1] - dc[1] * dx[0]};
                for (int c = 0; c < 3; c++) {
                    dP[3 * (o + W) + c] += gdx[c]; dP[3 * (o - W) + c] -= gdx[c];
                    dP[3 * (o + 1) + c] += gdy[c]; dP[3 * (o - 1) + c] -= gdy[c];
                }
            }
        }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t o = (size_t)y * W + x;
            float dd = 0.f;
            for (int c = 0; c < 3; c++) dd += dP[3 * o + c] * ((float)x * ray_mat[c] + (float)y * ray_mat[3 + c] + ray_mat[6 + c]);
            const float a = allmap[N + o], q = allmap[o] / a;
            if (!(isnan(q) || isinf(q))) {
                dL_dallmap[o] += dd * (1.f - depth_ratio) / a;
                dL_dallmap[N + o] += -dd * (1.f - depth_ratio) * allmap[o] / (a * a);
            }
            const float m = allmap[5 * N + o];
            if (!(isnan(m) || isinf(m))) dL_dallmap[5 * N + o] += dd * depth_ratio;
        }
    loss_out[0] = (float)(s_err / (double)N); loss_out[1] = (float)(s_dist / (double)N);
    loss_out[2] = lambda_normal * loss_out[0] + lambda_dist * loss_out[1];
    free(P); free(dP); free(depth);
}

/* ------------------------------------------------------------------------------------------------------------------------
 * PGSR single-view normal regulariser: normal_from_depth_image / depth_pcd2normal / depth2point_cam / ndc_2_cam
 * (gssr/utils/graphics_utils.py:80-146), PGSRScene.render (gssr/scene/pgsr_scene.py:320) and get_loss_dict (:105-112).
 *   P = depth * ([x y 1] * ray_mat), ray_mat = inverse(K^T);  n = normalize(cross(P(y,x+1)-P(y,x-1), P(y-1,x)-P(y+1,x))), 0 on the border
 *   depth_normal = n * alpha (detached);  loss = lambda * mean(weight * sum_c |depth_normal_c - normal_c|)
 * loss_out[3] = {mean weighted L1, 0, loss}.
 */
void ref_loss_plane_geo(int32_t H, int32_t W, const float* depth, const float* alpha, const float* normal, const float* weight,
                        const float* ray_mat, float lambda, float* loss_out, float* dL_ddepth, float* dL_dnormal, float* out_dn)
{
    const size_t N = (size_t)H * W;
    float* P = malloc(3 * N * sizeof(float));
    float* dP = calloc(3 * N, sizeof(float));
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t o = (size_t)y * W + x;
            for (int c = 0; c < 3; c++) P[3 * o + c] = depth[o] * ((float)x * ray_mat[c] + (float)y * ray_mat[3 + c] + ray_mat[6 + c]);
        }
    double s_err = 0.0;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t o = (size_t)y * W + x;
            const float a = alpha[o], w = weight ? weight[o] : 1.f, wl = lambda * w / (float)N;
            const int interior = y >= 1 && y <= H - 2 && x >= 1 && x <= W - 2;
            float lr[3], bt[3], cr[3], n[3] = {0, 0, 0}, len = 0.f;
            if (interior) {
                for (int c = 0; c < 3; c++) {
                    lr[c] = P[3 * (o + 1) + c] - P[3 * (o - 1) + c];             /* left_to_right */
                    bt[c] = P[3 * (o - W) + c] - P[3 * (o + W) + c];             /* bottom_to_top */
                }
                cr[0] = lr[1] * bt[2] - lr[2] * bt[1]; cr[1] = lr[2] * bt[0] - lr[0] * bt[2]; cr[2] = lr[0] * bt[1] - lr[1] * bt[0];
                len = sqrtf(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
                const float den = len > 1e-12f ? len : 1e-12f;
                for (int c = 0; c < 3; c++) n[c] = cr[c] / den;
            }
            float dn[3], e = 0.f;
            for (int c = 0; c < 3; c++) {
                const float df = n[c] * a - normal[c * N + o];
                const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
                e += fabsf(df);
                dL_dnormal[c * N + o] = -wl * sg;
                dn[c] = wl * a * sg;
                if (out_dn) out_dn[c * N + o] = n[c] * a;
            }
            s_err += (double)(w * e);
            if (interior) {
                float dc[3], nd = 0.f;
                for (int c = 0; c < 3; c++) nd += n[c] * dn[c];
                if (len > 1e-12f) for (int c = 0; c < 3; c++) dc[c] = (dn[c] - n[c] * nd) / len;
                else for (int c = 0; c < 3; c++) dc[c] = dn[c] / 1e-12f;
                /* c = lr x bt: d lr = bt x dc, d bt = dc x lr */
                const float glr[3] = {bt[1] * dc[2] - bt[2] * dc[1], bt[2] * dc[0] - bt[0] * dc[2], bt[0] * dc[1] - bt[1] * dc[0]};
                const float gbt[3] = {dc[1] * lr[2] - dc[2] * lr[1], dc[2] * lr[0] - dc[0] * lr[2], dc[0] * lr[1] - dc[1] * lr[0]};
                for (int c = 0; c < 3; c++) {
                    dP[3 * (o + 1) + c] += glr[c]; dP[3 * (o - 1) + c] -= glr[c];
                    dP[3 * (o - W) + c] += gbt[c]; dP[3 * (o + W) + c] -= gbt[c];
                }
            }
        }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t o = (size_t)y * W + x;
            float dd = 0.f;
            for (int c = 0; c < 3; c++) dd += dP[3 * o + c] * ((float)x * ray_mat[c] + (float)y * ray_mat[3 + c] + ray_mat[6 + c]);
            dL_ddepth[o] = dd;
        }
    loss_out[0] = (float)(s_err / (double)N); loss_out[1] = 0.f; loss_out[2] = lambda * loss_out[0];
    free(P); free(dP);
}
