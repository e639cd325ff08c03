/*
 * gsd_oracle.h -- CPU restatement of the neural-Gaussian decode that feeds the rasterizer in the Scaffold/Octree methods
 * (gssr/scene/scaffold_scene.py:27-120, gssr/scene/octree_scene.py:26-133).  TEST INFRASTRUCTURE ONLY (see gsr_oracle.h).
 *
 * PARITY STATUS: PINNED against the reference itself run in the authoring container -- tests/golden/make_golden_ref.py instantiates the
 * reference's ScaffoldGaussian / OctreeGaussian (CPU torch; the CUDA-extension imports of those modules satisfied by inert stand-ins)
 * and calls ScaffoldScene / OctreeScene.generate_neural_gaussians and OctreeGaussian.set_anchor_mask; outputs and autograd gradients are
 * committed as tests/golden/ref_decode_*.npz, ref_lod_*.npz and checked by tests/test_golden_ref_cpu.py (oracle) and
 * tests/test_gpu_golden_ref.py (HIP).  A float64 transcription (tests/ref_decode_torch.py) adds randomized cases beyond the fixtures.
 *
 * Per visible anchor a (feat_dim = hidden = 32, k = n_offsets):
 *   view = anchor - campos; dist = |view|; view /= dist
 *   x    = [feat(32), view(3), (dist), (level)]                 (dist/level present per MLP according to the flags)
 *   o    = tanh(W2o relu(W1o x + b1o) + b2o) (* opacity_scale[a])   -> neural_opacity (k), mask = o > 0
 *   sr   = W2c relu(W1c x + b1c) + b2c                              (7k)
 *   col  = sigmoid(W2k relu(W1k [x, appearance] + b1k) + b2k)       (3k)
 *   for each offset j with mask: xyz = anchor + offset[j]*scaling[0:3]; scale = scaling[3:6]*sigmoid(sr[7j..7j+2]);
 *                                rot = sr[7j+3..7j+6]/max(|.|,1e-12); color = col[3j..3j+2]; opacity = o[j]
 * Outputs are compacted in (anchor, offset) order, exactly like boolean-mask indexing.
 */
#ifndef GSD_ORACLE_H
#define GSD_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct refd_cfg {
    int32_t Na, Nv, k, A;                 /* anchors, visible anchors, offsets per anchor, appearance dims (0 = none) */
    int32_t dist_o, dist_c, dist_k;       /* add_opacity_dist / add_cov_dist / add_color_dist */
    int32_t level;                        /* add_level */
} refd_cfg;

typedef struct refd_params {              /* torch nn.Linear layout: W[out][in], row-major */
    float *W1o, *b1o, *W2o, *b2o;         /* [32][35+dist_o+level], [32], [k][32], [k] */
    float *W1c, *b1c, *W2c, *b2c;         /* [32][35+dist_c+level], [32], [7k][32], [7k] */
    float *W1k, *b1k, *W2k, *b2k;         /* [32][35+dist_k+level+A], [32], [3k][32], [3k] */
    float *app;                           /* [A] appearance embedding row of this camera */
} refd_params;

typedef struct refd_inputs {
    const float* anchor;        /* [Na,3] */
    const float* feat;          /* [Na,32] */
    const float* offset;        /* [Na,k,3] */
    const float* scaling;       /* [Na,6] (activated) */
    const float* level;         /* [Na] or NULL */
    const float* opacity_scale; /* [Na] or NULL (Octree progressive ratio) */
    const int32_t* vis_idx;     /* [Nv] ascending anchor indices */
    const float* campos;        /* [3] */
} refd_inputs;

/* returns P; outputs have capacity Nv*k rows */
int64_t refd_forward(const refd_cfg* c, const refd_inputs* in, const refd_params* p,
                     float* neural_opacity /*[Nv*k]*/, uint8_t* mask /*[Nv*k]*/,
                     float* xyz /*[P,3]*/, float* color /*[P,3]*/, float* opacity /*[P]*/, float* scaling /*[P,3]*/, float* rot /*[P,4]*/);

/* gradients: per-anchor arrays are [Na,...] zero-filled by the callee; g holds the parameter gradients (same shapes as p) */
void refd_backward(const refd_cfg* c, const refd_inputs* in, const refd_params* p, const uint8_t* mask,
                   const float* dL_dxyz, const float* dL_dcolor, const float* dL_dopacity, const float* dL_dscaling, const float* dL_drot,
                   float* d_anchor /*[Na,3]*/, float* d_feat /*[Na,32]*/, float* d_offset /*[Na,k,3]*/, float* d_scaling /*[Na,6]*/,
                   refd_params* g);

#ifdef __cplusplus
}
#endif
#endif
