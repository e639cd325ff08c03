/*
 * gsr_oracle.h -- CPU restatement of the GS-SR rasterizer hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (gs-sr_amd/) never links, imports or falls back to it.
 *
 * PARITY STATUS: "parity unpinned by the reference" -- /root/reference ships no golden vectors, tests or
 * fixtures for the rasterizers and its CUDA sources cannot be built here (no nvcc, no GPU).  This oracle is
 * pinned instead by (1) golden vectors generated from the importable reference Python helpers
 * (tests/golden/make_golden.py: camera matrices, SH evaluation), (2) a float64 torch-autograd restatement of the
 * forward maths that checks every analytic backward (tests/ref_torch.py), (3) closed-form cases and invariants.
 * Exception: ref_tsdf_integrate restates PYTHON (gssr/utils/mesh_utils.py:195-246) and IS pinned by a reference run
 * (tests/golden/ref_tsdf_unbounded.npz, tests/test_golden_ref_cpu.py).
 *
 * Variants: 0 = EWA   (submodules/diff-gaussian-rasterization)
 *           1 = SURFEL(submodules/diff-surfel-rasterization)
 *           2 = PLANE (submodules/diff-plane-rasterization)
 */
#ifndef GSR_ORACLE_H
#define GSR_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* arithmetic type of the rasterizer entry points: float (default build, libgsr_oracle.so / _fma.so) or double
   (-DGSR_REAL_IS_DOUBLE=1, libgsr_oracle_f64.so: every `real` array below is then float64) */
#if GSR_REAL_IS_DOUBLE
typedef double real;
#else
typedef float real;
#endif

enum { REF_EWA = 0, REF_SURFEL = 1, REF_PLANE = 2 };

typedef struct ref_inputs {
    int32_t P, D, M;          /* #gaussians, active SH degree, SH coefficients per gaussian (0 = none) */
    int32_t W, H;
    real tanfovx, tanfovy;
    real scale_modifier;
    int32_t prefiltered;
    int32_t render_geo;       /* PLANE only */
    const real* bg;          /* [3] */
    const real* viewmatrix;  /* [16] row-vector convention (p_view = [p 1] * V) */
    const real* projmatrix;  /* [16] */
    const real* campos;      /* [3] */
    const real* means3D;     /* [P,3] */
    const real* shs;         /* [P,M,3] or NULL */
    const real* colors_precomp; /* [P,3] or NULL */
    const real* opacities;   /* [P] */
    const real* scales;      /* [P,3] (SURFEL [P,2]) or NULL */
    const real* rotations;   /* [P,4] wxyz or NULL */
    const real* cov3D_precomp; /* [P,6] (SURFEL: transMat [P,9]) or NULL */
    const real* all_map;     /* PLANE [P,5] or NULL */
    /* ---- truth-run controls; all NULL / 0 in ordinary runs --------------------------------------------------------------------
       Integer stages taken from another (float32) run, so that the float64 build walks the SAME culling decisions, radii, sorted
       instance list and tile ranges and its images / gradients are comparable element by element: */
    const int32_t* ov_radii;         /* [P]   culling + radii */
    const uint32_t* ov_point_list;   /* [R]   sorted instance list */
    const uint32_t* ov_ranges;       /* [T,2] tile ranges */
    int32_t ov_R;
    /* Gate robustness of every pixel (outputs): a running first-order float32 rounding-error bound is carried beside every quantity
       that feeds a discrete decision of the blend loop; gate_margin = min over the pixel's decisions of |value - threshold| / bound.
       > 1: every correct float32 evaluation of the reference's formulas takes the same decisions at this pixel (robust);
       <= 1: a 1-ulp-level difference may flip the decision named by gate_id on splat gate_splat (fragile). */
    real* gate_margin;               /* [H*W] or NULL */
    int32_t* gate_id;                /* [H*W] REF_GATE_* */
    int32_t* gate_splat;             /* [H*W] */
    real* splat_noise;               /* [P] or NULL: max over the splat's evaluated pairs of the relative float32 error bound of alpha */
    /* "float32 geometry, exact blend" runs (the floor diagnostic of tests/parity_truth.py): the per-gaussian state the preprocess hands the blend
       kernels -- what the reference keeps in its float32 geomBuffer -- is REPLACED by these arrays (values of a float32 run, widened) after the
       preprocess, so that blend forward / backward are evaluated in `real` on exactly the numbers a float32 implementation blends.  NULL = keep own. */
    const real* ov_cov;              /* [P,9] transMat (SURFEL) / [P,6] cov3D */
    const real* ov_conic_opacity;    /* [P,4] conic + opacity (SURFEL: normal + opacity) */
    const real* ov_means2D;          /* [P,2] */
    const real* ov_depths;           /* [P] */
    const real* ov_rgb;              /* [P,3] (used when colours come from SH) */
} ref_inputs;

enum { REF_GATE_NONE = 0, REF_GATE_POWER = 1,   /* power > 0 */
       REF_GATE_ALPHA = 2,                      /* alpha < 1/255 */
       REF_GATE_TERMINATE = 3,                  /* T (1 - alpha) < 1e-4 */
       REF_GATE_HALF = 4,                       /* T > 0.5 (median depth / observe count) */
       REF_GATE_RHO = 5,                        /* SURFEL rho3d <= rho2d */
       REF_GATE_NEAR = 6 };                     /* SURFEL depth < near */

typedef struct ref_out_grads {      /* dL/d(outputs); NULL pointers are treated as zeros */
    const real* dL_dcolor;         /* [3,H,W] */
    const real* dL_dothers;        /* SURFEL [11,H,W] */
    const real* dL_dout_all_map;   /* PLANE [5,H,W] */
    const real* dL_dplane_depth;   /* PLANE [1,H,W] */
} ref_out_grads;

typedef struct ref_in_grads {       /* all caller-allocated, zero-filled by the callee first */
    real* dL_dmeans3D;     /* [P,3] */
    real* dL_dmeans2D;     /* [P,3] */
    real* dL_dmeans2D_abs; /* PLANE [P,3] or NULL */
    real* dL_dcolors;      /* [P,3] */
    real* dL_dopacity;     /* [P] */
    real* dL_dcov3D;       /* [P,6]  (SURFEL: dL_dtransMat [P,9]) */
    real* dL_dsh;          /* [P,M,3] or NULL when M==0 */
    real* dL_dscales;      /* [P,3] (SURFEL [P,2]) */
    real* dL_drotations;   /* [P,4] */
    real* dL_dall_map;     /* PLANE [P,5] or NULL */
    real* dL_dconic;       /* EWA/PLANE [P,4]; SURFEL: dL_dnormal [P,3] -- intermediate, exposed for tests */
} ref_in_grads;

typedef struct ref_state ref_state;

/* forward: returns an opaque state (owned by the oracle; free with ref_free) holding geometry/binning/image state */
ref_state* ref_forward(int variant, const ref_inputs* in,
                       real* out_color /*[3,H,W]*/, int32_t* radii /*[P]*/,
                       real* out_others /*SURFEL [11,H,W]*/,
                       int32_t* out_observe /*PLANE [P]*/, real* out_all_map /*PLANE [5,H,W]*/,
                       real* out_plane_depth /*PLANE [1,H,W]*/);
void ref_backward(ref_state* st, const ref_inputs* in, const ref_out_grads* og, ref_in_grads* ig);
void ref_free(ref_state* st);

/* introspection of intermediate (integer, bit-exact comparable) stages */
int32_t  ref_num_rendered(const ref_state* st);
int32_t  ref_num_tiles(const ref_state* st);
/* workload statistics of the forward: (pixel, splat) pairs evaluated before termination / passing every gate (the "contributing pairs C" of SURVEY 8d) */
void     ref_get_pair_counts(const ref_state* st, uint64_t* evaluated, uint64_t* contributing);
/* measurement controls of bench.py's cpu_baseline leg: OpenMP thread count, and "every n-th tile only" in the blend loops */
void     ref_set_threads(int32_t n);
void     ref_set_tile_stride(int32_t n);
/* out[k], k < R: the largest alpha any in-image pixel of instance k's tile sees from that gaussian under the reference's per-pixel gates (0: none) */
void     ref_instance_max_alpha(const ref_state* st, const ref_inputs* in, real* out);
void     ref_get_point_list(const ref_state* st, uint32_t* out /*[R]*/);
void     ref_get_keys(const ref_state* st, uint64_t* out /*[R] sorted keys*/);
void     ref_get_ranges(const ref_state* st, uint32_t* out /*[T,2]*/);
void     ref_get_tiles_touched(const ref_state* st, uint32_t* out /*[P]*/);
void     ref_get_geom(const ref_state* st, real* depths /*[P]*/, real* means2D /*[P,2]*/,
                      real* conic_opacity /*[P,4] (SURFEL normal_opacity)*/, real* rgb /*[P,3]*/,
                      real* cov3D_or_transmat /*[P,6] or [P,9]*/);
void     ref_get_image_state(const ref_state* st, real* final_T /*[N] (SURFEL [3N])*/,
                             uint32_t* n_contrib /*[N] (SURFEL [2N])*/);

void ref_visible_filter(const ref_inputs* in, int32_t* radii);   /* scaffold-filter */
void ref_mark_visible(int32_t P, const real* means3D, const real* viewmatrix, const real* projmatrix,
                      uint8_t* present);

/* in-repo TSDF definition (gssr/utils/mesh_utils.py:195-246), one frame, updates tsdf/weight/rgb in place */
void ref_tsdf_integrate(int64_t V, const float* points /*[V,3]*/, const float* full_proj /*[16]*/,
                        int32_t W, int32_t H, const float* depth /*[H,W]*/, const float* rgb /*[3,H,W]*/,
                        float sdf_trunc, const float* sdf_trunc_per_point /*[V] or NULL*/,
                        float* tsdf /*[V]*/, float* weight /*[V]*/, float* rgb_acc /*[V,3]*/);

/* dense-grid Open3D-style integration (Open3D 0.18.0 UniformTSDFVolume algorithm restated; NOT in /root/reference:
   parity unpinned) -- see include/gsrast.h gsr_tsdf_integrate_dense for the definition */
void ref_tsdf_integrate_dense(int32_t nx, int32_t ny, int32_t nz, const float* origin, float voxel_length, float sdf_trunc,
                              float depth_trunc, int32_t W, int32_t H, const float* depth, const float* rgb, float fx, float fy,
                              float cx, float cy, const float* extrinsic, float* tsdf, float* weight, float* color);

/* simple-knn distCUDA2: mean squared distance to the 3 nearest neighbours (brute-force restatement) */
void ref_dist2(int32_t P, const float* points, float* out);

int32_t ref_omp_threads(void);

#ifdef __cplusplus
}
#endif
#endif
