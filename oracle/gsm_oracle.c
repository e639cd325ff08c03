/*
 * gsm_oracle.c -- CPU restatement of PGSR's multi-view regularisers.  TEST INFRASTRUCTURE ONLY (see gsr_oracle.h).
 *   gssr/scene/pgsr_scene.py:113-204 (get_loss_dict, "multi-view loss"), :60-95 (lncc)
 *   gssr/utils/point_utils.py:38-75 (get_points_from_depth, get_points_depth_in_depth_map)
 *   gssr/utils/graphics_utils.py:185-198 (patch_offsets, patch_warp);  gssr/cameras/__init__.py:96-121 (get_rays, get_k, get_inv_k)
 * PARITY STATUS: PINNED against the reference itself run in the authoring container (tests/golden/make_golden_ref.py ->
 * tests/golden/ref_loss_plane_multiview.npz, checked by tests/test_golden_ref_cpu.py).
 *
 * The two rigid transforms are handed in composed (row-vector convention, X' = X A + b, A row-major then b):
 *   v2n: view camera -> near camera  (A = Rv^T Rn, b = Tn - Tv A);  n2v its inverse.  The reference walks through world coordinates.
 * Part 1 (geometric consistency), per pixel (x,y):
 *   pc = ((x-cx)/fx, (y-cy)/fy, 1) * depth;  q = pc v2n;  (u,v) = (q.x nfx/q.z + ncx, q.y nfy/q.z + ncy)
 *   mask = 0<u<Wn & 0<v<Hn & q.z>0.1;  mz = bilinear(near_depth, u, v) with border clamp (grid_sample align_corners, padding 'border')
 *   r = (q/q.z * mz) n2v;  e = (r.x fx/r.z + cx - x, r.y fy/r.z + cy - y);  noise = |e|;  d_mask = mask & noise < th
 *   weight = exp(-noise) (detached), 0 outside d_mask;  stats = {sum_{d_mask} weight*noise, |d_mask|}
 *   g_depth / g_near = d stats[0] / d depth, d near_depth (weight held constant) -- the caller scales by lambda_geo / stats[1].
 * Part 2 (patch NCC), per sampled pixel p = idx[i]:
 *   ref_j  = bilinear0(gray, x/s + ox, y/s + oy), (ox,oy) in [-h,h]^2      (grid_sample zeros padding, no gradient)
 *   Hm = A^T - b n^T / dist;  Hk = K_near(s) Hm Kinv_view(s);  g = Hk (u_j, v_j, 1);  nea_j = bilinear0(near_gray, g.x/(g.z+1e-10), g.y/(g.z+1e-10))
 *   cc = cross^2 / (ref_var nea_var + 1e-8) from the patch sums (lncc);  ncc = clamp(1-cc, 0, 2);  mask = ncc < 0.9
 *   stats = {sum_{mask} ncc*weight[p], |mask|};  g_normal[:,p], g_dist[p] = d stats[0] / d normal, dist.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct {
    int32_t W, H, Wn, Hn, Wg, Hg;
    float fx, fy, cx, cy, nfx, nfy, ncx, ncy;
    float v2n[12], n2v[12];
    float ncc_scale, noise_th;
    int32_t patch;
} refm_cfg;

static void xform(const float* M, const float* p, float* o) {
    for (int j = 0; j < 3; ++j) o[j] = p[0] * M[0 * 3 + j] + p[1] * M[1 * 3 + j] + p[2] * M[2 * 3 + j] + M[9 + j];
}

/* grid_sample(align_corners=True, padding_mode='border') at pixel coordinates (u,v): value, d/du, d/dv, taps */
static float bilerp_border(const float* img, int W, int H, float u, float v, float* du, float* dv, int* idx, float* wt) {
    float mu = 1.f, mv = 1.f;
    if (u < 0.f) { u = 0.f; mu = 0.f; } else if (u > (float)(W - 1)) { u = (float)(W - 1); mu = 0.f; }
    if (v < 0.f) { v = 0.f; mv = 0.f; } else if (v > (float)(H - 1)) { v = (float)(H - 1); mv = 0.f; }
    int x0 = (int)floorf(u), y0 = (int)floorf(v);
    float ax = u - (float)x0, ay = v - (float)y0;
    int xs[2] = {x0, x0 + 1}, ys[2] = {y0, y0 + 1};
    float wx[2] = {1.f - ax, ax}, wy[2] = {1.f - ay, ay};
    float val = 0.f, gx = 0.f, gy = 0.f;
    for (int b = 0; b < 2; ++b)
        for (int a = 0; a < 2; ++a) {
            int k = b * 2 + a;
            int in = xs[a] >= 0 && xs[a] < W && ys[b] >= 0 && ys[b] < H;
            idx[k] = in ? ys[b] * W + xs[a] : -1;
            wt[k] = in ? wx[a] * wy[b] : 0.f;
            float t = in ? img[idx[k]] : 0.f;
            val += t * wx[a] * wy[b];
            gx += t * (a ? 1.f : -1.f) * wy[b];
            gy += t * (b ? 1.f : -1.f) * wx[a];
        }
    *du = gx * mu; *dv = gy * mv;
    return val;
}

/* grid_sample(align_corners=True, padding_mode='zeros') */
static float bilerp_zeros(const float* img, int W, int H, float u, float v, float* du, float* dv) {
    int x0 = (int)floorf(u), y0 = (int)floorf(v);
    float ax = u - (float)x0, ay = v - (float)y0;
    float val = 0.f, gx = 0.f, gy = 0.f;
    for (int b = 0; b < 2; ++b)
        for (int a = 0; a < 2; ++a) {
            int xx = x0 + a, yy = y0 + b;
            if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
            float t = img[yy * W + xx], wx = a ? ax : 1.f - ax, wy = b ? ay : 1.f - ay;
            val += t * wx * wy;
            gx += t * (a ? 1.f : -1.f) * wy;
            gy += t * (b ? 1.f : -1.f) * wx;
        }
    if (du) { *du = gx; *dv = gy; }
    return val;
}

void refm_multiview_geo(const refm_cfg* c, const float* depth, const float* near_depth, float* noise_out, uint8_t* dmask_out, float* weight_out,
                        double* stats, float* g_depth, float* g_near) {
    const int W = c->W, H = c->H, Wn = c->Wn, Hn = c->Hn;
    memset(g_near, 0, sizeof(float) * (size_t)Wn * Hn);
    double sum = 0.0; int64_t cnt = 0;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int p = y * W + x;
            const float d = depth[p];
            const float rx = ((float)x - c->cx) / c->fx, ry = ((float)y - c->cy) / c->fy;
            float pc[3] = {rx * d, ry * d, d}, q[3];
            xform(c->v2n, pc, q);
            const float u = q[0] * c->nfx / q[2] + c->ncx, v = q[1] * c->nfy / q[2] + c->ncy;
            const int mask = u > 0.f && u < (float)Wn && v > 0.f && v < (float)Hn && q[2] > 0.1f;
            float du, dv, wt[4]; int idx[4];
            const float mz = bilerp_border(near_depth, Wn, Hn, u, v, &du, &dv, idx, wt);
            float qp[3] = {q[0] / q[2] * mz, q[1] / q[2] * mz, mz}, r[3];
            xform(c->n2v, qp, r);
            const float ex = r[0] * c->fx / r[2] + c->cx - (float)x, ey = r[1] * c->fy / r[2] + c->cy - (float)y;
            const float noise = sqrtf(ex * ex + ey * ey);
            const int dm = mask && noise < c->noise_th;
            const float w = dm ? 1.0f / expf(noise) : 0.f;
            noise_out[p] = noise; dmask_out[p] = (uint8_t)dm; weight_out[p] = w;
            g_depth[p] = 0.f;
            if (!dm) continue;
            sum += (double)(w * noise); ++cnt;
            if (noise == 0.f) continue;
            const float dex = w * ex / noise, dey = w * ey / noise;
            float dr[3] = {dex * c->fx / r[2], dey * c->fy / r[2], -(dex * r[0] * c->fx + dey * r[1] * c->fy) / (r[2] * r[2])};
            float dqp[3];
            for (int i = 0; i < 3; ++i) dqp[i] = dr[0] * c->n2v[i * 3 + 0] + dr[1] * c->n2v[i * 3 + 1] + dr[2] * c->n2v[i * 3 + 2];
            const float dmz = dqp[0] * q[0] / q[2] + dqp[1] * q[1] / q[2] + dqp[2];
            float dq[3] = {dqp[0] * mz / q[2], dqp[1] * mz / q[2], -(dqp[0] * q[0] + dqp[1] * q[1]) * mz / (q[2] * q[2])};
            for (int k = 0; k < 4; ++k)
                if (idx[k] >= 0) g_near[idx[k]] += dmz * wt[k];
            const float gu = dmz * du, gv = dmz * dv;
            dq[0] += gu * c->nfx / q[2]; dq[1] += gv * c->nfy / q[2];
            dq[2] += -(gu * q[0] * c->nfx + gv * q[1] * c->nfy) / (q[2] * q[2]);
            float dpc[3];
            for (int i = 0; i < 3; ++i) dpc[i] = dq[0] * c->v2n[i * 3 + 0] + dq[1] * c->v2n[i * 3 + 1] + dq[2] * c->v2n[i * 3 + 2];
            g_depth[p] = dpc[0] * rx + dpc[1] * ry + dpc[2];
        }
    stats[0] = sum; stats[1] = (double)cnt;
}

void refm_multiview_ncc(const refm_cfg* c, int32_t N, const int32_t* idx, const float* weight, const float* normal, const float* dist,
                        const float* gray, const float* near_gray, float* ncc_out, uint8_t* mask_out, double* stats, float* g_normal, float* g_dist) {
    const int W = c->W, H = c->H, Wg = c->Wg, Hg = c->Hg, h = c->patch;
    const size_t HW = (size_t)W * H;
    const float s = c->ncc_scale, tps = (float)((2 * h + 1) * (2 * h + 1));
    memset(g_normal, 0, sizeof(float) * 3 * HW); memset(g_dist, 0, sizeof(float) * HW);
    const float Kn[9] = {c->nfx / s, 0, c->ncx / s, 0, c->nfy / s, c->ncy / s, 0, 0, 1};
    const float Kvi[9] = {s / c->fx, 0, -c->cx / c->fx, 0, s / c->fy, -c->cy / c->fy, 0, 0, 1};
    double sum = 0.0; int64_t cnt = 0;
    for (int i = 0; i < N; ++i) {
        const int p = idx[i];
        if (p < 0) { ncc_out[i] = 0.f; mask_out[i] = 0; continue; }
        const int x = p % W, y = p / W;
        const float n[3] = {normal[p], normal[HW + p], normal[2 * HW + p]}, dd = dist[p];
        float Hm[9], T1[9], Hk[9];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) Hm[a * 3 + b] = c->v2n[b * 3 + a] - c->v2n[9 + a] * n[b] / dd;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) T1[a * 3 + b] = Kn[a * 3 + 0] * Hm[0 * 3 + b] + Kn[a * 3 + 1] * Hm[1 * 3 + b] + Kn[a * 3 + 2] * Hm[2 * 3 + b];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) Hk[a * 3 + b] = T1[a * 3 + 0] * Kvi[0 * 3 + b] + T1[a * 3 + 1] * Kvi[1 * 3 + b] + T1[a * 3 + 2] * Kvi[2 * 3 + b];
        const float px = (float)x / s, py = (float)y / s;
        float Sr = 0, Sn = 0, Srr = 0, Snn = 0, Srn = 0;
        for (int oy = -h; oy <= h; ++oy)
            for (int ox = -h; ox <= h; ++ox) {
                const float uu = px + (float)ox, vv = py + (float)oy;
                const float rj = bilerp_zeros(gray, Wg, Hg, uu, vv, 0, 0);
                const float g0 = Hk[0] * uu + Hk[1] * vv + Hk[2], g1 = Hk[3] * uu + Hk[4] * vv + Hk[5], g2 = Hk[6] * uu + Hk[7] * vv + Hk[8] + 1e-10f;
                const float nj = bilerp_zeros(near_gray, Wg, Hg, g0 / g2, g1 / g2, 0, 0);
                Sr += rj; Sn += nj; Srr += rj * rj; Snn += nj * nj; Srn += rj * nj;
            }
        const float ravg = Sr / tps, navg = Sn / tps;
        const float cross = Srn - navg * Sr, rvar = Srr - ravg * Sr, nvar = Snn - navg * Sn;
        const float den = rvar * nvar + 1e-8f;
        const float cc = cross * cross / den;
        float ncc = 1.f - cc;
        const int clamped = ncc < 0.f || ncc > 2.f;
        ncc = ncc < 0.f ? 0.f : (ncc > 2.f ? 2.f : ncc);
        const int m = ncc < 0.9f;
        ncc_out[i] = ncc; mask_out[i] = (uint8_t)m;
        if (!m) continue;
        const float w = weight[p];
        sum += (double)(ncc * w); ++cnt;
        if (clamped || w == 0.f) continue;
        const float dcc = -w;                                       /* d(ncc w)/dcc */
        const float dcross = dcc * 2.f * cross / den, dnvar = -dcc * cross * cross * rvar / (den * den);
        float dH[9] = {0};
        for (int oy = -h; oy <= h; ++oy)
            for (int ox = -h; ox <= h; ++ox) {
                const float uu = px + (float)ox, vv = py + (float)oy;
                const float rj = bilerp_zeros(gray, Wg, Hg, uu, vv, 0, 0);
                const float g0 = Hk[0] * uu + Hk[1] * vv + Hk[2], g1 = Hk[3] * uu + Hk[4] * vv + Hk[5], g2 = Hk[6] * uu + Hk[7] * vv + Hk[8] + 1e-10f;
                float du, dv;
                const float nj = bilerp_zeros(near_gray, Wg, Hg, g0 / g2, g1 / g2, &du, &dv);
                const float dn = dcross * (rj - Sr / tps) + dnvar * (2.f * nj - 2.f * navg);
                const float dgx = dn * du, dgy = dn * dv;
                const float dg[3] = {dgx / g2, dgy / g2, -(dgx * g0 + dgy * g1) / (g2 * g2)};
                const float uv1[3] = {uu, vv, 1.f};
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) dH[a * 3 + b] += dg[a] * uv1[b];
            }
        float T2[9], dHm[9];                                        /* dHm = Kn^T dH Kvi^T */
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) T2[a * 3 + b] = Kn[0 * 3 + a] * dH[0 * 3 + b] + Kn[1 * 3 + a] * dH[1 * 3 + b] + Kn[2 * 3 + a] * dH[2 * 3 + b];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) dHm[a * 3 + b] = T2[a * 3 + 0] * Kvi[b * 3 + 0] + T2[a * 3 + 1] * Kvi[b * 3 + 1] + T2[a * 3 + 2] * Kvi[b * 3 + 2];
        float gd = 0.f;
        for (int b = 0; b < 3; ++b) {
            float gn = 0.f;
            for (int a = 0; a < 3; ++a) { gn -= dHm[a * 3 + b] * c->v2n[9 + a] / dd; gd += dHm[a * 3 + b] * c->v2n[9 + a] * n[b] / (dd * dd); }
            g_normal[(size_t)b * HW + p] = gn;
        }
        g_dist[p] = gd;
    }
    stats[0] = sum; stats[1] = (double)cnt;
}

/*
 * Per-Gaussian input of the plane rasterizer (gssr/scene/pgsr_scene.py:241-257 get_rotation_matrix / get_smallest_axis / get_normal and
 * :297-304 in render()):  all_map[i] = {local_normal(3), 1, local_distance}.
 *   R = quaternion_to_matrix(q)   -- pytorch3d.transforms (third-party, absent from /root/reference; requirements.txt names pytorch3d
 *                                    without a pin).  Published algorithm: (r,i,j,k) = q, two_s = 2/(q.q), R = I + two_s * [...] (below).
 *   n = R[:, argmin(scale)] (first minimum);  n = -n where n . (campos - xyz) < 0
 *   ln = n Wv[:3,:3];  pc = xyz Wv[:3,:3] + Wv[3,:3];  dist = |ln . pc|
 * Backward: dL/dxyz (through pc only: the flip mask is not differentiable), dL/dq (through R incl. the two_s normalisation).
 * viewmatrix: the reference's world_view_transform, 16 floats row-major.
 */
static void q2m(const float* q, float* R) {
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    const float two_s = 2.0f / (r * r + i * i + j * j + k * k);
    R[0] = 1 - two_s * (j * j + k * k); R[1] = two_s * (i * j - k * r); R[2] = two_s * (i * k + j * r);
    R[3] = two_s * (i * j + k * r); R[4] = 1 - two_s * (i * i + k * k); R[5] = two_s * (j * k - i * r);
    R[6] = two_s * (i * k - j * r); R[7] = two_s * (j * k + i * r); R[8] = 1 - two_s * (i * i + j * j);
}

static int argmin3(const float* s) { int k = 0; if (s[1] < s[k]) k = 1; if (s[2] < s[k]) k = 2; return k; }

void refm_plane_allmap(int32_t P, const float* xyz, const float* rot, const float* scale, const float* V, const float* campos, float* all_map) {
    for (int p = 0; p < P; ++p) {
        float R[9]; q2m(rot + 4 * p, R);
        const int k = argmin3(scale + 3 * p);
        float n[3] = {R[k], R[3 + k], R[6 + k]};
        const float* x = xyz + 3 * p;
        const float dot = n[0] * (campos[0] - x[0]) + n[1] * (campos[1] - x[1]) + n[2] * (campos[2] - x[2]);
        if (dot < 0.f) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
        float ln[3], pc[3];
        for (int c = 0; c < 3; ++c) {
            ln[c] = n[0] * V[0 * 4 + c] + n[1] * V[1 * 4 + c] + n[2] * V[2 * 4 + c];
            pc[c] = x[0] * V[0 * 4 + c] + x[1] * V[1 * 4 + c] + x[2] * V[2 * 4 + c] + V[3 * 4 + c];
        }
        float* o = all_map + 5 * p;
        o[0] = ln[0]; o[1] = ln[1]; o[2] = ln[2]; o[3] = 1.0f;
        o[4] = fabsf(ln[0] * pc[0] + ln[1] * pc[1] + ln[2] * pc[2]);
    }
}

void refm_plane_allmap_bwd(int32_t P, const float* xyz, const float* rot, const float* scale, const float* V, const float* campos,
                           const float* d_all_map, float* d_xyz, float* d_rot) {
    for (int p = 0; p < P; ++p) {
        const float* q = rot + 4 * p;
        float R[9]; q2m(q, R);
        const int k = argmin3(scale + 3 * p);
        float n[3] = {R[k], R[3 + k], R[6 + k]};
        const float* x = xyz + 3 * p;
        const float dot = n[0] * (campos[0] - x[0]) + n[1] * (campos[1] - x[1]) + n[2] * (campos[2] - x[2]);
        const float flip = dot < 0.f ? -1.f : 1.f;
        n[0] *= flip; n[1] *= flip; n[2] *= flip;
        float ln[3], pc[3];
        for (int c = 0; c < 3; ++c) {
            ln[c] = n[0] * V[0 * 4 + c] + n[1] * V[1 * 4 + c] + n[2] * V[2 * 4 + c];
            pc[c] = x[0] * V[0 * 4 + c] + x[1] * V[1 * 4 + c] + x[2] * V[2 * 4 + c] + V[3 * 4 + c];
        }
        const float* g = d_all_map + 5 * p;
        const float sd = ln[0] * pc[0] + ln[1] * pc[1] + ln[2] * pc[2];
        const float sg = sd > 0.f ? 1.f : (sd < 0.f ? -1.f : 0.f);                  /* d|x|/dx, 0 at 0 like torch.abs */
        float dln[3], dpc[3], dn[3];
        for (int c = 0; c < 3; ++c) { dln[c] = g[c] + sg * g[4] * pc[c]; dpc[c] = sg * g[4] * ln[c]; }
        for (int a = 0; a < 3; ++a) {
            d_xyz[3 * p + a] = dpc[0] * V[a * 4 + 0] + dpc[1] * V[a * 4 + 1] + dpc[2] * V[a * 4 + 2];
            dn[a] = flip * (dln[0] * V[a * 4 + 0] + dln[1] * V[a * 4 + 1] + dln[2] * V[a * 4 + 2]);
        }
        /* dR: only column k receives dn.  R = I + two_s * M(q) with M the bracketed terms: dR = dtwo_s * M + two_s * dM */
        const float r = q[0], i = q[1], j = q[2], kk = q[3];
        const float s2 = r * r + i * i + j * j + kk * kk, two_s = 2.0f / s2;
        float dR[9] = {0};
        dR[k] = dn[0]; dR[3 + k] = dn[1]; dR[6 + k] = dn[2];
        const float M[9] = {-(j * j + kk * kk), i * j - kk * r, i * kk + j * r, i * j + kk * r, -(i * i + kk * kk), j * kk - i * r,
                            i * kk - j * r, j * kk + i * r, -(i * i + j * j)};
        float dts = 0.f;
        for (int e = 0; e < 9; ++e) dts += dR[e] * M[e];
        float dM[9];
        for (int e = 0; e < 9; ++e) dM[e] = dR[e] * two_s;
        float dq[4];
        dq[0] = -kk * dM[1] + j * dM[2] + kk * dM[3] - i * dM[5] - j * dM[6] + i * dM[7];
        dq[1] = j * dM[1] + kk * dM[2] + j * dM[3] - 2 * i * dM[4] - r * dM[5] + kk * dM[6] + r * dM[7] - 2 * i * dM[8];
        dq[2] = -2 * j * dM[0] + i * dM[1] + r * dM[2] + i * dM[3] + kk * dM[5] - r * dM[6] + kk * dM[7] - 2 * j * dM[8];
        dq[3] = -2 * kk * dM[0] - r * dM[1] + i * dM[2] + r * dM[3] - 2 * kk * dM[4] + j * dM[5] + i * dM[6] + j * dM[7];
        const float dtds2 = -2.0f / (s2 * s2);                                     /* d two_s / d (q.q) */
        for (int e = 0; e < 4; ++e) d_rot[4 * p + e] = dq[e] + dts * dtds2 * 2.0f * q[e];
    }
}

/*
 * Per-iteration densification statistics of the explicit-Gaussian methods (3DGS / 2DGS / PGSR):
 *   gssr/gaussian/vanilla_gaussian.py:467-472 (densify: max_radii2D) + :428-430 (add_densification_stats)
 *   gssr/gaussian/pgsr_gaussian.py:164-172 + :157-161 (mask = (out_observe > 0) & visibility_filter for max_radii2D; abs-gradient accumulators)
 * out_observe / grad_abs / accum_abs / denom_abs may be NULL (vanilla).
 */
void refm_densify_stats(int32_t P, const uint8_t* filter, const int32_t* radii, const int32_t* out_observe, const float* grad, int32_t gs,
                        const float* grad_abs, float* max_radii2D, float* accum, float* denom, float* accum_abs, float* denom_abs)
{
    for (int p = 0; p < P; ++p) {
        if (!filter[p]) continue;
        if (!out_observe || out_observe[p] > 0) { const float r = (float)radii[p]; if (r > max_radii2D[p]) max_radii2D[p] = r; }
        const float gx = grad[(size_t)p * gs], gy = grad[(size_t)p * gs + 1];
        accum[p] += sqrtf(gx * gx + gy * gy); denom[p] += 1.f;
        if (grad_abs) {
            const float ax = grad_abs[(size_t)p * gs], ay = grad_abs[(size_t)p * gs + 1];
            accum_abs[p] += sqrtf(ax * ax + ay * ay); denom_abs[p] += 1.f;
        }
    }
}
