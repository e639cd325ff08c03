/*
 * gsl_oracle.c -- CPU restatement of the photometric loss of the training loop.  TEST INFRASTRUCTURE ONLY (see gsr_oracle.h).
 *   loss = (1-lambda)*mean|img-gt| + lambda*(1-SSIM)      gssr/scene/vanilla_scene.py:29-69 (l1_loss, _ssim, _gaussian, ssim, get_loss_dict)
 * Direct (non-separable) 11x11 correlation with zero padding, like F.conv2d(padding=5, groups=C) with the outer-product window; the
 * gradient is the chain rule of those ops, checked against torch autograd of the reference's formula in tests/.
 * PARITY STATUS: the reference ships no fixture for it -- "parity unpinned by the reference"; pinned by tests/test_loss_cpu.py.
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
