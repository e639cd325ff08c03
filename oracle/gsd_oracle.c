/*
 * gsd_oracle.c -- see gsd_oracle.h.  Plain C, float32 per-anchor arithmetic in the order the torch ops imply (dot products
 * accumulated left to right), float64 only for the sums over anchors in the parameter gradients.
 * Follows gssr/scene/scaffold_scene.py:27-120 and gssr/scene/octree_scene.py:26-133 (forward); the backward is the chain rule of
 * those ops (autograd in the reference), checked against a float64 torch-autograd transcription in tests/.
 */
#include "gsd_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define F 32
#define HID 32
#define KMAX 16
#define XMAX (F + 3 + 1 + 1 + 64)

typedef struct mlp_t {
    const float *W1, *b1, *W2, *b2;
    int in, out;
} mlp_t;

static int in_dim(const refd_cfg* c, int dist, int app) { return F + 3 + (dist ? 1 : 0) + (c->level ? 1 : 0) + (app ? c->A : 0); }

/* x = [feat, view, (dist), (level), (appearance)] -- scaffold_scene.py:57-58,79-81, octree_scene.py:60-65 */
static int build_x(const refd_cfg* c, const refd_inputs* in, const refd_params* p, int a, const float v[3], float d, int dist, int app,
                   float* x)
{
    int n = 0;
    for (int i = 0; i < F; i++) x[n++] = in->feat[(size_t)a * F + i];
    x[n++] = v[0]; x[n++] = v[1]; x[n++] = v[2];
    if (dist) x[n++] = d;
    if (c->level) x[n++] = in->level[a];
    if (app) for (int i = 0; i < c->A; i++) x[n++] = p->app[i];
    return n;
}

static void mlp_fwd(const mlp_t* m, const float* x, float* pre1, float* h, float* pre2)
{
    for (int j = 0; j < HID; j++) {
        float s = 0.f;
        for (int i = 0; i < m->in; i++) s += m->W1[j * m->in + i] * x[i];
        s += m->b1[j];
        pre1[j] = s;
        h[j] = s > 0.f ? s : 0.f;
    }
    for (int j = 0; j < m->out; j++) {
        float s = 0.f;
        for (int i = 0; i < HID; i++) s += m->W2[j * HID + i] * h[i];
        pre2[j] = s + m->b2[j];
    }
}

static void view_of(const refd_inputs* in, int a, float v[3], float* d)
{
    float r0 = in->anchor[3 * a] - in->campos[0], r1 = in->anchor[3 * a + 1] - in->campos[1], r2 = in->anchor[3 * a + 2] - in->campos[2];
    float n = sqrtf(r0 * r0 + r1 * r1 + r2 * r2);
    v[0] = r0 / n; v[1] = r1 / n; v[2] = r2 / n; *d = n;
}

static float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

int64_t refd_forward(const refd_cfg* c, const refd_inputs* in, const refd_params* p, float* neural_opacity, uint8_t* mask,
                     float* xyz, float* color, float* opacity, float* scaling, float* rot)
{
    const int k = c->k;
    mlp_t mo = {p->W1o, p->b1o, p->W2o, p->b2o, in_dim(c, c->dist_o, 0), k};
    mlp_t mc = {p->W1c, p->b1c, p->W2c, p->b2c, in_dim(c, c->dist_c, 0), 7 * k};
    mlp_t mk = {p->W1k, p->b1k, p->W2k, p->b2k, in_dim(c, c->dist_k, 1), 3 * k};
    int64_t P = 0;
    for (int vi = 0; vi < c->Nv; vi++) {
        const int a = in->vis_idx[vi];
        float v[3], d, x[XMAX], pre1[HID], h[HID], o[KMAX], sr[7 * KMAX], col[3 * KMAX];
        view_of(in, a, v, &d);
        build_x(c, in, p, a, v, d, c->dist_o, 0, x); mlp_fwd(&mo, x, pre1, h, o);
        build_x(c, in, p, a, v, d, c->dist_c, 0, x); mlp_fwd(&mc, x, pre1, h, sr);
        build_x(c, in, p, a, v, d, c->dist_k, 1, x); mlp_fwd(&mk, x, pre1, h, col);
        const float s = in->opacity_scale ? in->opacity_scale[a] : 1.f;
        const float* S = in->scaling + (size_t)a * 6;
        for (int j = 0; j < k; j++) {
            float oj = tanhf(o[j]);
            if (in->opacity_scale) oj = oj * s;                       /* octree_scene.py:80-84 */
            neural_opacity[(size_t)vi * k + j] = oj;
            const int m = oj > 0.0f;                                   /* scaffold_scene.py:70 */
            mask[(size_t)vi * k + j] = (uint8_t)m;
            if (!m) continue;
            const float* off = in->offset + ((size_t)a * k + j) * 3;
            for (int i = 0; i < 3; i++) {
                xyz[3 * P + i] = in->anchor[3 * a + i] + off[i] * S[i];                    /* :113-114 */
                color[3 * P + i] = sigmoidf_(col[3 * j + i]);
                scaling[3 * P + i] = S[3 + i] * sigmoidf_(sr[7 * j + i]);                  /* :109 */
            }
            const float* q = sr + 7 * j + 3;
            float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            n = n > 1e-12f ? n : 1e-12f;                                                   /* F.normalize eps */
            for (int i = 0; i < 4; i++) rot[4 * P + i] = q[i] / n;
            opacity[P] = oj;
            P++;
        }
    }
    return P;
}

typedef struct acc_t {      /* float64 parameter-gradient accumulators of one MLP */
    double *W1, *b1, *W2, *b2;
} acc_t;

static void acc_alloc(acc_t* a, const mlp_t* m)
{
    a->W1 = calloc((size_t)HID * m->in, sizeof(double)); a->b1 = calloc(HID, sizeof(double));
    a->W2 = calloc((size_t)m->out * HID, sizeof(double)); a->b2 = calloc(m->out, sizeof(double));
}
static void acc_store_free(acc_t* a, const mlp_t* m, float* W1, float* b1, float* W2, float* b2)
{
    for (int i = 0; i < HID * m->in; i++) W1[i] = (float)a->W1[i];
    for (int i = 0; i < HID; i++) b1[i] = (float)a->b1[i];
    for (int i = 0; i < m->out * HID; i++) W2[i] = (float)a->W2[i];
    for (int i = 0; i < m->out; i++) b2[i] = (float)a->b2[i];
    free(a->W1); free(a->b1); free(a->W2); free(a->b2);
}

/* backward of one MLP for one anchor: dpre2 given; accumulates parameter grads, returns dx */
static void mlp_bwd(const mlp_t* m, acc_t* acc, const float* x, const float* pre1, const float* h, const float* dpre2, float* dx)
{
    float dh[HID];
    for (int i = 0; i < HID; i++) dh[i] = 0.f;
    for (int j = 0; j < m->out; j++) {
        const float g = dpre2[j];
        acc->b2[j] += g;
        for (int i = 0; i < HID; i++) { acc->W2[j * HID + i] += (double)(g * h[i]); dh[i] += m->W2[j * HID + i] * g; }
    }
    for (int i = 0; i < m->in; i++) dx[i] = 0.f;
    for (int j = 0; j < HID; j++) {
        const float g = pre1[j] > 0.f ? dh[j] : 0.f;
        acc->b1[j] += g;
        for (int i = 0; i < m->in; i++) { acc->W1[j * m->in + i] += (double)(g * x[i]); dx[i] += m->W1[j * m->in + i] * g; }
    }
}

void refd_backward(const refd_cfg* c, const refd_inputs* in, const refd_params* p, const uint8_t* mask,
                   const float* dL_dxyz, const float* dL_dcolor, const float* dL_dopacity, const float* dL_dscaling, const float* dL_drot,
                   float* d_anchor, float* d_feat, float* d_offset, float* d_scaling, refd_params* g)
{
    const int k = c->k;
    mlp_t mo = {p->W1o, p->b1o, p->W2o, p->b2o, in_dim(c, c->dist_o, 0), k};
    mlp_t mc = {p->W1c, p->b1c, p->W2c, p->b2c, in_dim(c, c->dist_c, 0), 7 * k};
    mlp_t mk = {p->W1k, p->b1k, p->W2k, p->b2k, in_dim(c, c->dist_k, 1), 3 * k};
    acc_t ao, ac, ak;
    acc_alloc(&ao, &mo); acc_alloc(&ac, &mc); acc_alloc(&ak, &mk);
    double dapp[64];
    for (int i = 0; i < 64; i++) dapp[i] = 0.0;
    memset(d_anchor, 0, sizeof(float) * 3 * (size_t)c->Na);
    memset(d_feat, 0, sizeof(float) * F * (size_t)c->Na);
    memset(d_offset, 0, sizeof(float) * 3 * (size_t)c->Na * k);
    memset(d_scaling, 0, sizeof(float) * 6 * (size_t)c->Na);
    int64_t row = 0;
    for (int vi = 0; vi < c->Nv; vi++) {
        const int a = in->vis_idx[vi];
        float v[3], d, xo[XMAX], xc[XMAX], xk[XMAX];
        float p1o[HID], ho[HID], o[KMAX], p1c[HID], hc[HID], sr[7 * KMAX], p1k[HID], hk[HID], col[3 * KMAX];
        view_of(in, a, v, &d);
        build_x(c, in, p, a, v, d, c->dist_o, 0, xo); mlp_fwd(&mo, xo, p1o, ho, o);
        build_x(c, in, p, a, v, d, c->dist_c, 0, xc); mlp_fwd(&mc, xc, p1c, hc, sr);
        build_x(c, in, p, a, v, d, c->dist_k, 1, xk); mlp_fwd(&mk, xk, p1k, hk, col);
        const float s = in->opacity_scale ? in->opacity_scale[a] : 1.f;
        const float* S = in->scaling + (size_t)a * 6;
        float dpo[KMAX], dpc[7 * KMAX], dpk[3 * KMAX], dS[6] = {0, 0, 0, 0, 0, 0}, dA[3] = {0, 0, 0};
        for (int j = 0; j < k; j++) {
            dpo[j] = 0.f;
            for (int i = 0; i < 7; i++) dpc[7 * j + i] = 0.f;
            for (int i = 0; i < 3; i++) dpk[3 * j + i] = 0.f;
            if (!mask[(size_t)vi * k + j]) continue;
            const float t = tanhf(o[j]);
            dpo[j] = dL_dopacity[row] * s * (1.f - t * t);
            const float* off = in->offset + ((size_t)a * k + j) * 3;
            for (int i = 0; i < 3; i++) {
                const float gx = dL_dxyz[3 * row + i];
                dA[i] += gx;
                d_offset[((size_t)a * k + j) * 3 + i] = gx * S[i];
                dS[i] += gx * off[i];
                const float cc = sigmoidf_(col[3 * j + i]);
                dpk[3 * j + i] = dL_dcolor[3 * row + i] * cc * (1.f - cc);
                const float sg = sigmoidf_(sr[7 * j + i]);
                const float gs = dL_dscaling[3 * row + i];
                dpc[7 * j + i] = gs * S[3 + i] * sg * (1.f - sg);
                dS[3 + i] += gs * sg;
            }
            const float* q = sr + 7 * j + 3;
            const float* gr = dL_drot + 4 * row;
            float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            if (n > 1e-12f) {
                float rh[4], dot = 0.f;
                for (int i = 0; i < 4; i++) { rh[i] = q[i] / n; dot += rh[i] * gr[i]; }
                for (int i = 0; i < 4; i++) dpc[7 * j + 3 + i] = (gr[i] - rh[i] * dot) / n;
            } else {
                for (int i = 0; i < 4; i++) dpc[7 * j + 3 + i] = gr[i] / 1e-12f;
            }
            row++;
        }
        float dxo[XMAX], dxc[XMAX], dxk[XMAX];
        mlp_bwd(&mo, &ao, xo, p1o, ho, dpo, dxo);
        mlp_bwd(&mc, &ac, xc, p1c, hc, dpc, dxc);
        mlp_bwd(&mk, &ak, xk, p1k, hk, dpk, dxk);
        for (int i = 0; i < F; i++) d_feat[(size_t)a * F + i] = dxo[i] + dxc[i] + dxk[i];
        float dv[3], dd = 0.f;
        for (int i = 0; i < 3; i++) dv[i] = dxo[F + i] + dxc[F + i] + dxk[F + i];
        if (c->dist_o) dd += dxo[F + 3];
        if (c->dist_c) dd += dxc[F + 3];
        if (c->dist_k) dd += dxk[F + 3];
        if (c->A) {
            const int base = F + 3 + (c->dist_k ? 1 : 0) + (c->level ? 1 : 0);
            for (int i = 0; i < c->A; i++) dapp[i] += (double)dxk[base + i];
        }
        const float vd = v[0] * dv[0] + v[1] * dv[1] + v[2] * dv[2];
        for (int i = 0; i < 3; i++) {
            dA[i] += (dv[i] - v[i] * vd) / d + dd * v[i];
            d_anchor[3 * a + i] = dA[i];
        }
        for (int i = 0; i < 6; i++) d_scaling[(size_t)a * 6 + i] = dS[i];
    }
    acc_store_free(&ao, &mo, g->W1o, g->b1o, g->W2o, g->b2o);
    acc_store_free(&ac, &mc, g->W1c, g->b1c, g->W2c, g->b2c);
    acc_store_free(&ak, &mk, g->W1k, g->b1k, g->W2k, g->b2k);
    if (c->A && g->app) for (int i = 0; i < c->A; i++) g->app[i] = (float)dapp[i];
}

/* Octree-GS level-of-detail mask: OctreeGaussianModel.set_anchor_mask + map_to_int_level (gssr/gaussian/octree_gaussian.py:184-203,255-267).
 * mode: 0 floor, 1 round (half to even, torch.round), 2 ceil, 3 progressive.  prog_ratio / transition may be NULL. */
void refd_lod_mask(int32_t Na, const float* anchor, const int32_t* level, const float* extra_level, const float* campos, float voxel_size,
                   float fork, float standard_dist, float resolution_scale, int32_t coarse_index, int32_t mode, uint8_t* anchor_mask,
                   float* prog_ratio, uint8_t* transition)
{
    const int cur = coarse_index - 1;
    for (int i = 0; i < Na; i++) {
        const float half = (voxel_size / 2) / powf(fork, (float)level[i]);
        float d2 = 0.f;
        for (int c = 0; c < 3; c++) { const float d = (anchor[3 * i + c] + half) - campos[c]; d2 += d * d; }
        const float dist = sqrtf(d2) * resolution_scale;
        float pred = log2f(standard_dist / dist) / log2f(fork) + (extra_level ? extra_level[i] : 0.f);
        int il;
        if (mode == 3) {
            pred = pred + 1.0f;
            if (pred < 0.9999f) pred = 0.9999f;
            if (pred > (float)cur + 0.9999f) pred = (float)cur + 0.9999f;
            il = (int)floorf(pred);
            if (prog_ratio) prog_ratio[i] = pred - truncf(pred);
            if (transition) transition[i] = level[i] == il;
        } else {
            const float r = mode == 0 ? floorf(pred) : (mode == 1 ? rintf(pred) : ceilf(pred));
            il = (int)r;
            il = il < 0 ? 0 : (il > cur ? cur : il);
        }
        anchor_mask[i] = level[i] <= il;
    }
}

/*
 * Per-iteration densification statistics of the Scaffold / Octree methods (gssr/gaussian/scaffold_gaussian.py:488-508 training_statis):
 *   for every visible anchor a = vis_idx[v]:  opacity_accum[a] += sum_j max(neural_opacity[v,j], 0);  anchor_demon[a] += 1
 *   for every generated Gaussian p (the p-th set entry of mask, i.e. slot (v,j)) with update_filter[p]:
 *       offset_gradient_accum[a*k+j] += |viewspace_grad[p, 0:2]|;  offset_denom[a*k+j] += 1
 * (the reference expresses the slot -> Gaussian map through three boolean-mask assignments.)
 */
void refd_training_stats(int32_t Nv, int32_t k, const int32_t* vis_idx, const float* neural_opacity, const uint8_t* mask, const uint8_t* update_filter,
                         const float* grad, int32_t grad_stride, float* opacity_accum, float* anchor_demon, float* offset_gradient_accum,
                         float* offset_denom)
{
    int64_t p = 0;
    for (int v = 0; v < Nv; ++v) {
        const int a = vis_idx[v];
        float s = 0.f;
        for (int j = 0; j < k; ++j) { const float o = neural_opacity[(size_t)v * k + j]; s += o < 0.f ? 0.f : o; }
        opacity_accum[a] += s; anchor_demon[a] += 1.f;
        for (int j = 0; j < k; ++j) {
            if (!mask[(size_t)v * k + j]) continue;
            if (update_filter[p]) {
                const float gx = grad[(size_t)p * grad_stride], gy = grad[(size_t)p * grad_stride + 1];
                offset_gradient_accum[(size_t)a * k + j] += sqrtf(gx * gx + gy * gy);
                offset_denom[(size_t)a * k + j] += 1.f;
            }
            ++p;
        }
    }
}
