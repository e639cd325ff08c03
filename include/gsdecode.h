/*
 * gsdecode.h -- C ABI of the fused neural-Gaussian decode (the step that feeds the rasterizer in the Scaffold-GS / Octree-GS methods).
 * Part of libgsrast_hip.so (gfx950).  Plain C, device pointers + sizes, no torch types.
 *
 * Replaces the chain of ~60 torch ops (and their autograd graph) in
 *     ScaffoldScene.generate_neural_gaussians   gssr/scene/scaffold_scene.py:27-120
 *     OctreeScene.generate_neural_gaussians     gssr/scene/octree_scene.py:26-133
 * with the default head layout of gssr/gaussian/scaffold_gaussian.py:141-159 / octree_gaussian.py:77-95:
 *     feat_dim = hidden = 32, Sequential(Linear(in,32), ReLU, Linear(32,out), act), out = k | 7k | 3k, act = Tanh | none | Sigmoid.
 * use_feat_bank=True (scaffold_gaussian.py:133-139) is not covered: the shipped wrapper raises NotImplementedError for it.
 *
 * Per visible anchor a = vis_idx[v]:
 *     view = anchor[a]-campos; dist = |view|; view /= dist;   x = [feat[a](32), view(3), (dist), (level[a])]
 *     o   = tanh(W2o relu(W1o x + b1o) + b2o) (* opacity_scale[a])      neural_opacity[v*k+j], mask = o > 0      (scaffold :62-70)
 *     sr  = W2c relu(W1c x + b1c) + b2c                                  (7 per offset)                            (:88-92)
 *     col = sigmoid(W2k relu(W1k [x, appearance] + b1k) + b2k)           (3 per offset)                            (:76-85)
 *     masked offsets j, compacted in (v, j) order (== boolean-mask indexing :100-102):
 *       xyz = anchor + offset[a][j]*scaling[a][0:3];  scaling_out = scaling[a][3:6]*sigmoid(sr[0:3]);  rot = normalize(sr[3:7]);
 *       color = col;  opacity = o[j]                                                                               (:104-114)
 *
 * Like the rasterizer, the forward is split at the one point where the host must learn a count (P = number of emitted Gaussians,
 * needed to size the outputs; the reference's `[mask]` indexing synchronises at the same place).
 * All pointers are device pointers unless marked host.  Return 0 on success; message via gsr_last_error().
 */
#ifndef GSDECODE_H
#define GSDECODE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GSD_FEAT 32          /* feat_dim == hidden width (scaffold_gaussian.py:27) */
#define GSD_MAX_K 16         /* n_offsets (default 10, scaffold_gaussian.py:28) */
#define GSD_MAX_APP 64       /* appearance_dim (default 32, scaffold_gaussian.py:41) */

typedef struct gsd_cfg {
    int32_t Na;              /* anchors */
    int32_t Nv;              /* visible anchors = rows of vis_idx */
    int32_t k;               /* n_offsets */
    int32_t A;               /* appearance_dim, 0 = no appearance embedding */
    int32_t dist_o, dist_c, dist_k;   /* add_opacity_dist, add_cov_dist, add_color_dist (scaffold_gaussian.py:47-49) */
    int32_t level;           /* add_level (octree_gaussian.py:29) */
} gsd_cfg;

typedef struct gsd_inputs {
    const float* anchor;        /* [Na,3]   get_anchor */
    const float* feat;          /* [Na,32]  _anchor_feat */
    const float* offset;        /* [Na,k,3] _offset */
    const float* scaling;       /* [Na,6]   get_scaling (activated) */
    const float* level;         /* [Na] or NULL */
    const float* opacity_scale; /* [Na] or NULL: Octree 'progressive' ratio with prog[~transition_mask]=1 applied (octree_scene.py:80-84) */
    const int32_t* vis_idx;     /* [Nv] ascending indices of the visible anchors (== nonzero(visible_mask)) */
    const float* campos;        /* [3] */
} gsd_inputs;

/* nn.Linear layout W[out][in] row-major.  Used for the parameters (read) and for their gradients (written). */
typedef struct gsd_params {
    float *W1o, *b1o, *W2o, *b2o;     /* [32][35+dist_o+level], [32], [k][32],  [k]  */
    float *W1c, *b1c, *W2c, *b2c;     /* [32][35+dist_c+level], [32], [7k][32], [7k] */
    float *W1k, *b1k, *W2k, *b2k;     /* [32][35+dist_k+level+A], [32], [3k][32], [3k] */
    float *app;                       /* [A] embedding row of the current camera (embedding_appearance(uid)), or NULL */
} gsd_params;

typedef struct gsd_outputs {          /* compacted, P rows */
    float* xyz;       /* [P,3] */
    float* color;     /* [P,3] */
    float* opacity;   /* [P]   */
    float* scaling;   /* [P,3] */
    float* rot;       /* [P,4] */
} gsd_outputs;

typedef struct gsd_out_grads {        /* dL/d(outputs), P rows, all required */
    const float *xyz, *color, *opacity, *scaling, *rot;
} gsd_out_grads;

typedef struct gsd_in_grads {
    float* anchor;    /* [Na,3]   rows of invisible anchors are left untouched: pass zero-filled buffers */
    float* feat;      /* [Na,32]  */
    float* offset;    /* [Na,k,3] */
    float* scaling;   /* [Na,6]   */
    gsd_params params;/* parameter gradients, fully overwritten */
} gsd_in_grads;

/* nonzero(visible_mask): vis_idx[0..count) ascending; *count_host written after an internal stream synchronisation.
   scratch: >= gsd_compact_scratch_bytes(Na). */
size_t gsd_compact_scratch_bytes(int32_t Na);
int gsd_compact_visible(const uint8_t* mask, int32_t Na, int32_t* vis_idx, uint32_t* count_host, void* scratch, size_t scratch_bytes,
                        void* stream);
/* The same compaction without the host synchronisation: vis_idx has Na entries, the visible anchors' indices first (ascending) and -1 behind
 * them ("padding rows"); run the decode with cfg.Nv = Na -- every gsd_* kernel skips rows whose index is negative (they emit nothing, their
 * neural_opacity / mask rows are written as 0).  *count_dev (DEVICE, may be NULL) receives the number of visible anchors, or 0xFFFFFFFF when
 * the compaction's bounded look-back timed out (vis_idx then holds in-range indices and -1 only; the synchronous entry points turn the same condition into an error). */
int gsd_compact_visible_padded(const uint8_t* mask, int32_t Na, int32_t* vis_idx, uint32_t* count_dev, void* scratch, size_t scratch_bytes,
                               void* stream);

/* stage 1: opacity head -> neural_opacity [Nv*k], mask [Nv*k] (u8), row_offset [Nv] (exclusive prefix of per-anchor emitted counts);
   *P_host = total, valid on return (one stream synchronisation).  scratch >= gsd_forward_scratch_bytes(Nv). */
size_t gsd_forward_scratch_bytes(int32_t Nv);
int gsd_forward_stage1(const gsd_cfg* cfg, const gsd_inputs* in, const gsd_params* p, float* neural_opacity, uint8_t* mask,
                       uint32_t* row_offset, uint32_t* P_host, void* scratch, size_t scratch_bytes, void* stream);
/* stage 2: cov + colour heads, assembly of the P compacted Gaussians.  scratch: the SAME buffer stage 1 was given (it holds the
   repacked layer-1 weights). */
int gsd_forward_stage2(const gsd_cfg* cfg, const gsd_inputs* in, const gsd_params* p, const float* neural_opacity,
                       const uint32_t* row_offset, uint32_t P, const gsd_outputs* out, void* scratch, size_t scratch_bytes, void* stream);

/* single call: stage 1 and stage 2 enqueued back to back, no host round trip in between.  `out` must have room for the worst case
   Nv*k rows; the first *P_host rows are valid on return (one stream synchronisation at the end).  scratch as for stage 1. */
int gsd_forward(const gsd_cfg* cfg, const gsd_inputs* in, const gsd_params* p, float* neural_opacity, uint8_t* mask, uint32_t* row_offset,
                const gsd_outputs* out, uint32_t* P_host, void* scratch, size_t scratch_bytes, void* stream);

/* static-shape forward (round 3): gsd_forward without its host synchronisation -- recordable into a HIP graph.  `out` keeps all Nv*k rows: the P
   emitted Gaussians first, the remaining rows PARKED at the camera centre (xyz = campos, opacity 0, scaling 0, rot identity, colour 0) so that the
   rasterizers of this library cull them in preprocess (radii 0, no tile instance, zero gradients); the caller hands all Nv*k rows on.
   *count_dev (DEVICE word, may be NULL) <- P (0xFFFFFFFF: the scan timed out and ALL rows are parked).  gsd_backward is unchanged: it never reads rows behind P. */
int gsd_forward_static(const gsd_cfg* cfg, const gsd_inputs* in, const gsd_params* p, float* neural_opacity, uint8_t* mask, uint32_t* row_offset,
                       const gsd_outputs* out, uint32_t* count_dev, void* scratch, size_t scratch_bytes, void* stream);

/* deferred-count forward (round 4): stage 1, then an asynchronous copy of P to *count_host (PINNED host memory) with `event` (a hipEvent_t the
   caller created) recorded behind it, then stage 2 against the worst-case-sized `out` -- no host synchronisation in the call.  The caller waits on
   the event when it needs P (the first *count_host rows of `out` are valid once the stream has passed stage 2): P is known as soon as the opacity
   head and its scan are done, while the device is still emitting the Gaussians or decoding another camera.  Rows behind P are left unwritten. */
int gsd_forward_deferred(const gsd_cfg* cfg, const gsd_inputs* in, const gsd_params* p, float* neural_opacity, uint8_t* mask, uint32_t* row_offset,
                         const gsd_outputs* out, uint32_t* count_host, void* event, void* scratch, size_t scratch_bytes, void* stream);

/* backward: recomputes the heads, writes the per-anchor gradients, and contracts the weight gradients over the anchors with MFMA.
   scratch >= gsd_backward_scratch_bytes(cfg).  fwd_scratch: the forward's scratch buffer if the caller kept it and the parameters are
   unchanged since (its repacked weights are reused), else NULL. */
size_t gsd_backward_scratch_bytes(const gsd_cfg* cfg);
int gsd_backward(const gsd_cfg* cfg, const gsd_inputs* in, const gsd_params* p, const float* neural_opacity, const uint32_t* row_offset,
                 uint32_t P, const gsd_out_grads* og, const gsd_in_grads* ig, const void* fwd_scratch, void* scratch, size_t scratch_bytes,
                 void* stream);

/* Per-iteration densification statistics of the Scaffold / Octree methods (gssr/gaussian/scaffold_gaussian.py:488-508 training_statis,
 * called from densify() :707-712 every iteration between start_stat and densify_until_iter):
 *   for each visible anchor a = vis_idx[v]:  opacity_accum[a] += sum_j max(neural_opacity[v*k+j], 0);  anchor_demon[a] += 1
 *   for each generated Gaussian p (= the p-th set byte of mask, slot (v,j)) with update_filter[p] != 0:
 *       offset_gradient_accum[a*k+j] += |viewspace_grad[p, 0:2]|;  offset_denom[a*k+j] += 1
 * neural_opacity / mask [Nv*k]: the decode's outputs (`neural_opacity`, `selection_mask`); update_filter [P] u8 = `visibility_filter`
 * (radii > 0); viewspace_grad [P, grad_stride] = `viewspace_points.grad`.  The four accumulators ([Na], [Na], [Na*k], [Na*k] floats) are
 * updated in place.  No host synchronisation (the reference's boolean-mask indexing has six). */
size_t gsd_training_stats_scratch_bytes(int32_t Nv);
int gsd_training_stats(int32_t Nv, int32_t k, const int32_t* vis_idx, const float* neural_opacity, const uint8_t* mask,
                       const uint8_t* update_filter, const float* viewspace_grad, int32_t grad_stride, float* opacity_accum,
                       float* anchor_demon, float* offset_gradient_accum, float* offset_denom, void* scratch, size_t scratch_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
