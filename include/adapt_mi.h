/*
 * adapt_mi.h — C-ABI of libadapt_mi.so: the MI355X (gfx950) wavefront path tracer that
 * stands in for AdaPT's Taichi `pt` renderer (Renderer, renderer/vanilla_renderer.py) and, with apt_render_cfg.volumetric = 1,
 * for its `vpt` renderer (VolumeRenderer, renderer/vpt.py: homogeneous media, null surfaces, one grid volume).
 *
 * What each entry point replaces in the reference (paths under /root/reference):
 *   apt_bvh_build_linear / apt_linear_bvh_*   tracer/bvh/bvh.cpp:274-296  bvh_cpp.bvh_build(...) (pybind11 module), called from
 *                                   tracer/path_tracer.py:143-179 (bvh_process): the same four arrays in the same layout
 *   apt_bvh_build / apt_bvh_*      the same builder's role for this library's own kernels (binary SAH tree -> 8-wide quantised tree)
 *   apt_flat_records               tracer/tracer_base.py:117-134,184-212: the data of the brute-force intersector, as the flat sweep wants it
 *   apt_scene_create               tracer/tracer_base.py:117-134 (load_primitives) +
 *                                   tracer/path_tracer.py:245-274 (initialze): numpy -> device fields
 *   apt_renderer_create            renderer/vanilla_renderer.py:26-30 / tracer_base.py:36-102 (film, crop, camera,
 *                                   sampling flags) — the constructor half that is not scene data
 *   apt_render                     renderer/vanilla_renderer.py:32-120  Renderer.render, or renderer/vpt.py:145-258
 *                                   VolumeRenderer.render (one launch == one spp there; here `n_spp` samples per call, cnt += n_spp)
 *   apt_read_pixels                `rdr.pixels.to_numpy()` (utils/watermark.py:23): color / cnt, layout [x][y][rgb]
 *   apt_get_accum / apt_set_accum  tracer/path_tracer.py:181-211  get_check_point / load_check_point
 *   apt_get_stats                  (none; the reference only has ti.profiler, render.py:154-160)
 *   apt_device_ptr                 (none; hands the tile framebuffer to RCCL for the multi-GPU gather)
 *
 * Conventions: every function returns 0 on success, a negative APT_E_* code on failure, and
 * apt_last_error() then returns a thread-local message.  Input pointers are borrowed for the
 * duration of the call and copied; outputs are caller-allocated; handles are opaque; one host
 * thread per handle.  All floats are IEEE binary32, ints are 32-bit.  There is no CPU fallback:
 * without a HIP device apt_scene_create / apt_renderer_create fail with APT_E_NO_DEVICE.
 */
#ifndef ADAPT_MI_H
#define ADAPT_MI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APT_OK            0
#define APT_E_INVALID    -1   /* bad argument */
#define APT_E_NO_DEVICE  -2   /* no usable HIP device */
#define APT_E_HIP        -3   /* HIP runtime error (message has the hipError string) */
#define APT_E_NOMEM      -4
#define APT_E_STATE      -5   /* call not valid in this state */

typedef struct apt_bvh apt_bvh;
typedef struct apt_scene apt_scene;
typedef struct apt_renderer apt_renderer;

/* Flat scene description — the arrays of SURVEY.md §A.2 (what the reference keeps in Taichi fields). */
typedef struct apt_scene_desc {
    int32_t n_prims, n_objects, n_sources, has_vertex_normal;
    const float*   prims;       /* n_prims*9   triangle (v0,v1,v2) | sphere (centre, r r r, 0 0 0) */
    const float*   normals;     /* n_prims*3   geometric normals */
    const float*   v_normals;   /* n_prims*9   per-vertex shading normals (zeros when a mesh has none) */
    const int32_t* obj_info;    /* n_objects*3 first prim, prim count, 0 = mesh | 1 = sphere */
    const float*   obj_aabb;    /* n_objects*6 min xyz, max xyz */
    const int32_t* emitter_id;  /* n_objects   attached emitter index or -1 */
    const int32_t* bxdf_i;      /* n_objects*4 type, is_delta, is_bsdf, 0     (bxdf/brdf.py:152-158, bsdf.py:68-73).  BRDF type 3 (microfacet) shades as the
                                 * reference's Trowbridge-Reitz model, i.e. as the reference does with `__ENABLE_MICROFACET__ = True` (brdf.py:8,428-484); with its
                                 * default False the reference's parser never emits type 3 (it rewrites such a BRDF to Lambertian, brdf.py:60-65): so does the host side here */
    const float*   bxdf_f;      /* n_objects*13 k_d k_s k_g mean, medium ior */
    const int32_t* src_i;       /* n_sources*4 type, bool_bits, obj_ref_id, 0 (emitters/abtract_source.py:44-54) */
    const float*   src_f;       /* n_sources*11 intensity dir pos inv_area r */
    float          world_ior;   /* free-space medium ior */
    /* Image textures on meshes (bxdf/texture.py:99-139, tracer/path_tracer.py:84-126,261-307): all NULL / 0 when the scene has
     * none.  Maps: 0 albedo (replaces k_d, vanilla_renderer.py:66), 1 normal and 2 bump (camera-ray hit only, vanilla_renderer.py:42). */
    const float*   uvs;         /* n_prims*6     per-vertex (u, v) of every triangle */
    const int32_t* tex_i;       /* n_objects*3*5 per object and map: type (-255 = none, 0 = image), off_x, off_y, w, h inside the atlas */
    const float*   tex_f;       /* n_objects*3*2 scale_u, scale_v */
    const float*   atlas[3];    /* per map: atlas_h * atlas_w * 3 floats, row-major [y][x][rgb], or NULL */
    int32_t        atlas_w[3], atlas_h[3];
    /* Participating media for the volumetric path tracer (bxdf/medium.py:24-125, parsers/world.py): both NULL when the scene has
     * none.  Row o < n_objects is the medium attached to object o's BSDF, row n_objects the world's. */
    const int32_t* med_i;       /* (n_objects+1)    type: -1 transparent, 0 hg, 1 multi-hg, 2 rayleigh, 3 mie */
    const float*   med_f;       /* (n_objects+1)*16 ior, u_s rgb, u_a rgb, u_e rgb, par[3], pdf[3] */
    /* Grid volume for the volumetric path tracer (bxdf/volume.py:36-246; what GridVolume_np.export() hands to the kernels): all NULL
     * when the scene declares none. */
    const int32_t* vol_i;       /* 5   type (2 = RGB; the only kind the reference can export), xres, yres, zres, phase-function type */
    const float*   vol_f;       /* 33  albedo rgb, inv_T[9] row-major, trans, mini, maxi (world box), majorant rgb, majorant pdf rgb, phase par[3], lobe weights[3] */
    const float*   vol_grid;    /* zres*yres*xres*3 extinction per channel, [z][y][x][rgb] */
} apt_scene_desc;

/* Per-renderer configuration: film, camera, sampling flags, tile ownership, batching. */
typedef struct apt_render_cfg {
    int32_t width, height;                        /* full film size */
    int32_t do_crop, start_x, end_x, start_y, end_y;
    int32_t max_bounce, num_shadow_ray;
    int32_t use_rr, use_mis, anti_alias, stratified, brdf_two_sides;
    int32_t rr_bounce_th;
    float   rr_threshold;
    float   cam_r[9];                             /* row-major camera rotation */
    float   cam_t[3];
    float   inv_focal, half_w, half_h;
    uint32_t seed;                                /* Philox key word 1 (key word 0 = global pixel index x*H+y) */
    /* tile ownership: columns x with (x / band_width) % world_size == rank belong to this renderer */
    int32_t band_width, rank, world_size;
    int32_t spp_per_batch;                        /* samples per pixel in flight per wavefront batch; 0 = auto: ~32 Mi paths per render lane,
                                                     fitted under the class queues' 32-bit slot addressing and a third of the free device memory.
                                                     A render call splits its samples into equal batches of at most this size, a whole number
                                                     per lane; the image does not depend on the split */
    int32_t device;                               /* HIP device ordinal */
    int32_t profile;                              /* 1 = time every kernel launch with HIP events */
    int32_t volumetric;                           /* 0 = Renderer.render (renderer/vanilla_renderer.py:32-120); 1 = VolumeRenderer.render
                                                     (renderer/vpt.py:145-258): free-path sampling in homogeneous media, null surfaces,
                                                     transmittance-tracked light samples */
} apt_render_cfg;

#define APT_N_KERNELS 5   /* generate, extend, shade, shadow, finalize */
typedef struct apt_stats {
    int64_t n_samples;        /* pixel-samples generated */
    int64_t n_extend;         /* closest-hit rays traced */
    int64_t n_shade;          /* bounce-loop iterations that reached shading (after miss / RR / cut-off) */
    int64_t n_shadow;         /* shadow rays the reference would cast (valid emitter sample) */
    int64_t n_shadow_traced;  /* of those, rays with a non-zero contribution that were actually traced */
    int64_t n_lit;            /* traced shadow rays found unoccluded */
    int64_t n_draws;          /* RNG draws consumed */
    int64_t n_poisoned;       /* light samples whose MIS weight was NaN (sample zeroed, as upstream) */
    int64_t launches[APT_N_KERNELS];
    double  kernel_ms[APT_N_KERNELS];   /* summed HIP-event time per kernel (profile=1 only) */
    double  render_ms;                  /* HIP-event time of all apt_render calls so far */
    int64_t n_track;                    /* volumetric: closest-hit queries made by the transmittance walk of the light samples */
} apt_stats;

/* ---- BVH build (host, own layout; replaces bvh_cpp.bvh_build) */
int apt_bvh_build(const float* prims /* n_prims*9 */, int32_t n_prims,
                  const int32_t* obj_info /* n_objects*3 */, int32_t n_objects, apt_bvh** out);
int apt_bvh_counts(const apt_bvh*, int32_t* n_nodes, int32_t* n_leaf_prims, int32_t* max_depth);
/* nodes: n_nodes*16 floats (two child boxes + two child links); prim_order: n_prims ints (BVH order -> original prim) */
int apt_bvh_export(const apt_bvh*, float* nodes, int32_t* prim_order);
/* The tree the traversal kernels walk: the binary tree (rebuilt with single-primitive leaves) collapsed to 8-wide nodes with
 * 8-bit quantised child boxes, 64 bytes = 16 dwords per node, breadth-first, node 0 = root (layout: csrc/bvh_wide.cpp);
 * prim_order: leaf-order slot -> original primitive.  n_levels = depth in 8-wide nodes.  apt_bvh_wide_frame: the global grid the
 * 16-bit node corners live on (world = gmin + gstep * grid, power-of-two steps). */
int apt_bvh_wide_counts(const apt_bvh*, int32_t* n_nodes, int32_t* n_levels);
int apt_bvh_wide_export(const apt_bvh*, uint32_t* nodes /* n_nodes*16 */, int32_t* prim_order /* n_prims */);
int apt_bvh_wide_frame(const apt_bvh*, float gmin[3], float gstep[3]);
void apt_bvh_free(apt_bvh*);

/* ---- records of the flat sweep (host; csrc/flat_build.cpp).  What the product build's small-scene intersector reads instead of the
 * reference's `prims` / `precom_vec` (tracer/tracer_base.py:117-134,184-212): per planar primitive its corner and the rows of
 * [e1 e2 n]^-1; two coplanar triangles of an object that share an edge and have a convex outline are one record.  apt_scene_create
 * builds them internally; this entry exposes the builder so that it can be checked without a device.
 * counts[7]: parallelograms, parallelograms in coplanar groups, convex quads, convex quads in groups, triangles, triangles in groups,
 * spheres.  stream (may be NULL): counts-ordered records, 12 floats each (corner p0, rows U, V, T), 18 for convex quads (+ the two far
 * edges as a u + b v + c), 4 per sphere (centre, r^2); tab (may be NULL): 28 floats per record (U, p0.x | V, p0.y | prim_a prim_b
 * class_a class_b as int32 | map_a[6] | map_b[6] | p0.z ...).  n_stream / n_tab: floats needed (always written). */
int apt_flat_records(const float* prims /* n_prims*9 */, int32_t n_prims, const int32_t* obj_info /* n_objects*3 */, int32_t n_objects,
                     int32_t counts[7], float* stream, int32_t stream_cap, float* tab, int32_t tab_cap, int32_t* n_stream, int32_t* n_tab);

/* ---- BVH build, reference layout: the drop-in for the pybind11 module itself.
 * Replaces bvh_cpp.bvh_build(obj_array, obj_info, world_min, world_max) (tracer/bvh/bvh.cpp:274-296), whose four flat arrays
 * PathTracer.bvh_process reshapes and loads into the LinearBVH / LinearNode fields (tracer/path_tracer.py:155-170,
 * tracer/ti_bvh.py:10-53) for AdaPT's own stackless preorder walk.  obj_prim_cnt / obj_is_sphere are the two rows of the
 * reference's (2, n_obj) obj_info table (tracer/path_tracer.py:222-230).  Outputs of apt_linear_bvh_export, caller-allocated:
 *   bvh_minmax  n_prims*6  per-primitive box (min xyz, max xyz) in tree order      node_minmax n_nodes*6  per-node box
 *   bvh_info    n_prims*2  (object, original primitive)                            node_info   n_nodes*3  (first, count, subtree size)
 * Nodes are in preorder; node 0 is the root with box = the world box and subtree size = n_nodes; a leaf has subtree size 1.
 * Host only (no device needed).  adapt_amd/bvh_cpp.py wraps these three calls as a module named like the reference's. */
typedef struct apt_linear_bvh apt_linear_bvh;
int apt_bvh_build_linear(const float* prims /* n_prims*9 */, int32_t n_prims, const int32_t* obj_prim_cnt, const int32_t* obj_is_sphere,
                         int32_t n_objects, const float* world_min /* 3 */, const float* world_max /* 3 */, apt_linear_bvh** out);
int apt_linear_bvh_counts(const apt_linear_bvh*, int32_t* n_nodes, int32_t* n_prims);
int apt_linear_bvh_export(const apt_linear_bvh*, float* bvh_minmax, float* node_minmax, int32_t* bvh_info, int32_t* node_info);
void apt_linear_bvh_free(apt_linear_bvh*);

/* ---- scene / renderer lifetime */
int apt_scene_create(const apt_scene_desc* desc, int32_t device, apt_scene** out);
void apt_scene_destroy(apt_scene*);
int apt_renderer_create(const apt_scene*, const apt_render_cfg* cfg, apt_renderer** out);
void apt_renderer_destroy(apt_renderer*);

/* ---- rendering */
int apt_render(apt_renderer*, int32_t n_spp);             /* accumulates n_spp more samples per owned pixel */
int apt_synchronize(apt_renderer*);
int apt_tile_shape(const apt_renderer*, int32_t* n_cols, int32_t* height);   /* owned framebuffer = n_cols*height*3 */
int apt_read_pixels(apt_renderer*, float* out);           /* owned tile, [local col][y][rgb], color / cnt */
int apt_get_accum(apt_renderer*, float* out, int32_t* cnt);
int apt_set_accum(apt_renderer*, const float* in, int32_t cnt);
int apt_reset(apt_renderer*);                             /* color = 0, cnt = 0, stats = 0 */
int apt_get_stats(apt_renderer*, apt_stats* out);
int apt_device_ptr(apt_renderer*, void** accum_dev, int32_t* cnt); /* device float[n_cols*height*3] accumulation buffer */
int apt_stream(apt_renderer*, void** hip_stream);         /* the hipStream_t every kernel of this renderer runs on */

/* ---- unit entry points used by the parity tests (same device code paths as apt_render) */
int apt_intersect(apt_renderer*, int32_t n, const float* o, const float* d,
                  int32_t* prim_out, float* t_out, float* uv_out);
int apt_occluded(apt_renderer*, int32_t n, const float* o, const float* d, const float* tmax, int32_t* occ_out);
int apt_rng_stream(int32_t device, uint32_t pixel, uint32_t seed, uint32_t sample, int32_t n, uint32_t* out);
/* Surface-model probe: test k uses material (bxdf_i[4k..], bxdf_f[13k..]) and dirs12[12k..] = n_s, n_g, incid, out.
 * do_sample = 0: out[4k..] = eval rgb (f*cos), pdf(out | incid).   [BRDF.eval/get_pdf, BSDF.eval_surf/get_pdf]
 * do_sample = 1: out[9k..] = dir xyz, f*cos rgb, pdf, is_specular, draws; RNG = Philox(key=(k, seed), sample 1). */
int apt_bxdf_probe(int32_t device, int32_t n, const int32_t* bxdf_i, const float* bxdf_f, const float* dirs12,
                   float world_ior, int32_t do_sample, uint32_t seed, float* out);
/* Texture probe: map_obj[2k..] = map (0 albedo, 1 normal, 2 bump), object; uv[2k..]; out3[3k..] = Texture.query. */
int apt_texture_probe(const apt_scene*, int32_t n, const int32_t* map_obj, const float* uv, float* out3);
/* Medium probe (volumetric tracer; bxdf/medium.py:84-125, bxdf/phase.py): test k uses medium (med_i[k], med_f[16k..]) and in7[7k..].
 * mode 0: Medium.sample_mfp, in = max_depth            -> out8[8k..] = is_mi, t, beta rgb, draws
 * mode 1: Medium.sample_new_rays, in = incid xyz       -> dir xyz, phase value x3, pdf, draws
 * mode 2: Medium.eval + transmittance, in = incid xyz, out xyz, depth -> phase value, transmittance rgb.  RNG as above. */
int apt_medium_probe(int32_t device, int32_t n, const int32_t* med_i, const float* med_f, int32_t mode, const float* in7,
                     uint32_t seed, float* out8);
/* Emitter probe: in11[11k..] = source index, hit_pos, normal, ray_d, min_depth;
 * out12[12k..] = sampled pos, intensity (/pdf), pdf, draws, eval_le rgb, solid_angle_pdf; RNG as above. */
int apt_emitter_probe(const apt_scene*, int32_t n, const float* in11, uint32_t seed, float* out12);
/* trace_mode: 0 = BVH traversal, 1 = wave-uniform sweep (scenes of <= 96 primitives; env APT_TRAVERSAL=bvh|sweep overrides) */
int apt_renderer_info(const apt_renderer*, int32_t* spp_batch, int32_t* n_subqueues, int64_t* queue_bytes,
                      int32_t* lds_bytes, const char** shade_variant, int32_t* trace_mode);

/* Shader clock (MHz) with every CU busy: cycle counter against the 100 MHz wall clock over a full-grid FMA chain (~1 ms).
 * bench.py prices the VALU roofline of the trace kernels with it.  (No reference counterpart.) */
int apt_measure_sclk_mhz(int32_t device, float* mhz);

const char* apt_last_error(void);
const char* apt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ADAPT_MI_H */
