/*
 * pt_oracle.c — CPU restatement of AdaPT's unidirectional path tracer (`--type pt`).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under adapt_amd/ may include, link or call
 * this file; it is the checker that tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg compare the HIP path against.
 *
 * What it restates (citations are file:line under /root/reference):
 *   renderer/vanilla_renderer.py:32-120   Renderer.render (one spp for all pixels)
 *   tracer/tracer_base.py:136-278         pix2ray, aabb_test, ray_intersect, does_intersect
 *   tracer/path_tracer.py:309-554         BVH traversal, BxDF dispatch, sample_light
 *   tracer/ti_bvh.py:10-53                LinearBVH / LinearNode slab tests
 *   tracer/bvh/bvh.cpp:19-212, bvh_helper.h:18-120   SAH BVH build + preorder linearise
 *   bxdf/brdf.py:147-601, bxdf/bsdf.py:61-262        BRDF / BSDF models
 *   emitters/abtract_source.py:35-232     TaichiSource.sample_hit / eval_le / solid_angle_pdf
 *   sampler/general_sampling.py:16-123    direction / triangle samplers, balance heuristic
 *   la/cam_transform.py:51-105, la/geo_optics.py:14-74   frames and optics
 *
 * Third-party arithmetic that is NOT in the reference tree: taichi==1.6.0
 * (requirements.txt:1) supplies vector ops, the 3x3 inverse and the RNG.  Restated
 * here from Taichi's published matrix code: normalized(v) = (1/sqrt(v.v)) * v,
 * sum/dot accumulate left to right, 3x3 inverse = adjugate * (1/det) with
 * det expanded along column 0, pow(x, constant int) = multiplication chain by squaring.  The RNG (ti.random) is replaced by a counter-based
 * Philox-4x32-10 stream — key (pixel, seed), counter (sample, draw/4) — the same
 * stream the HIP path uses; draw ORDER follows the reference source.
 *
 * Pinning: the reference has no tests for this path.  The oracle is
 * pinned by fixtures under tests/golden/ produced by running the reference's own
 * Python source under a float32 stand-in for the absent `taichi` package
 * (tests/golden/gen/), i.e. pinned at source level, unpinned at the real-Taichi
 * boundary (RNG stream, fast-math rounding).
 *
 * All arithmetic is float32, compiled with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ vectors */
typedef struct { float x, y, z; } v3;
typedef struct { float m[3][3]; } m3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vdiv(v3 a, v3 b) { return V(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 vadds(v3 a, float s) { return V(a.x + s, a.y + s, a.z + s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float vnorm2(v3 a) { return vdot(a, a); }
static inline float vnorm(v3 a) { return sqrtf(vnorm2(a)); }
static inline v3 vnormalized(v3 a) { float inv = 1.0f / vnorm(a); return vscale(a, inv); }
static inline v3 vcross(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float vmax(v3 a) { return fmaxf(fmaxf(a.x, a.y), a.z); }
static inline float vmin(v3 a) { return fminf(fminf(a.x, a.y), a.z); }
static inline v3 vabs(v3 a) { return V(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
static inline v3 vminv(v3 a, v3 b) { return V(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
static inline v3 vmaxv(v3 a, v3 b) { return V(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }
static inline v3 vpow_sv(float b, v3 e) { return V(powf(b, e.x), powf(b, e.y), powf(b, e.z)); }
static inline v3 m3mulv(const m3* M, v3 a) {
    return V((M->m[0][0] * a.x + M->m[0][1] * a.y) + M->m[0][2] * a.z,
             (M->m[1][0] * a.x + M->m[1][1] * a.y) + M->m[1][2] * a.z,
             (M->m[2][0] * a.x + M->m[2][1] * a.y) + M->m[2][2] * a.z);
}
/* Integer powers: Taichi lowers pow(x, n) with a constant integer n to exponentiation by squaring
 * (x^2 = x*x, x^5 = x * ((x*x)*(x*x))), not to a libm call. */
static inline float sq(float x) { return x * x; }
static inline float pow5i(float x) { float x2 = x * x; float x4 = x2 * x2; return x * x4; }
static inline float signf_(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

static const v3 ZERO3 = {0.f, 0.f, 0.f};
#define F_PI      ((float)3.14159265358979323846)
#define F_INV_PI  ((float)(1.0 / 3.14159265358979323846))
#define F_INV_2PI ((float)((1.0 / 3.14159265358979323846) * 0.5))
#define F_PI2     ((float)(2.0 * 3.14159265358979323846))
#define F_PI_DIV2 ((float)(3.14159265358979323846 / 2.0))
#define BRDF_EPS  1e-7f

/* taichi 3x3 inverse (adjugate / determinant), columns = (c0, c1, c2) */
static inline void inverse_cols(v3 c0, v3 c1, v3 c2, m3* out) {
    float a[3][3] = {{c0.x, c1.x, c2.x}, {c0.y, c1.y, c2.y}, {c0.z, c1.z, c2.z}};
    float det = a[0][0] * (a[1][1] * a[2][2] - a[2][1] * a[1][2])
              - a[1][0] * (a[0][1] * a[2][2] - a[2][1] * a[0][2])
              + a[2][0] * (a[0][1] * a[1][2] - a[1][1] * a[0][2]);
    float inv_det = 1.0f / det;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            out->m[j][i] = inv_det * (a[(i + 1) % 3][(j + 1) % 3] * a[(i + 2) % 3][(j + 2) % 3]
                                    - a[(i + 2) % 3][(j + 1) % 3] * a[(i + 1) % 3][(j + 2) % 3]);
}

/* ---------------------------------------------------------------------- RNG */
typedef struct {
    int mode;                 /* 0 = Philox counter stream, 1 = scripted values */
    uint32_t key0, key1, ctr0;
    uint32_t draw;            /* draws consumed so far on this path */
    uint32_t cache[4];
    uint32_t cache_blk;       /* block index held in cache (0xffffffff = none) */
    const double* script;
    int script_n, script_pos;
} rng_t;

static void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                          uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static void rng_seed(rng_t* r, uint32_t pixel, uint32_t seed, uint32_t sample) {
    memset(r, 0, sizeof(*r));
    r->key0 = pixel; r->key1 = seed; r->ctr0 = sample; r->cache_blk = 0xffffffffu;
}
static uint32_t rng_u32(rng_t* r) {
    uint32_t d = r->draw++;
    uint32_t blk = d >> 2;
    if (blk != r->cache_blk) {
        philox4x32_10(r->ctr0, blk, 0u, 0u, r->key0, r->key1, r->cache);
        r->cache_blk = blk;
    }
    return r->cache[d & 3u];
}
/* ti.random(float): uniform in [0,1) — 24 high bits */
static float rng_float(rng_t* r) {
    if (r->mode == 1) {
        double v = (r->script_pos < r->script_n) ? r->script[r->script_pos] : 0.5;
        r->script_pos++; r->draw++;
        return (float)v;
    }
    return (float)(rng_u32(r) >> 8) * (1.0f / 16777216.0f);
}
/* ti.random(int): full-range int32 */
static int32_t rng_int(rng_t* r) {
    if (r->mode == 1) {
        double v = (r->script_pos < r->script_n) ? r->script[r->script_pos] : 0.0;
        r->script_pos++; r->draw++;
        return (int32_t)v;
    }
    return (int32_t)rng_u32(r);
}
/* Python-style modulo (Taichi integer % is floor-mod) */
static inline int pymod(int a, int n) { int m = a % n; return (m < 0) ? m + n : m; }

/* --------------------------------------------------------------- scene data */
typedef struct {
    int type, is_delta, is_bsdf;
    v3 k_d, k_s, k_g, mean;
    float ior;                /* attached medium ior (BSDF only) */
} bxdf_t;

typedef struct {
    int type, bool_bits, obj_ref_id;
    v3 intensity, dir, pos;
    float inv_area, r;
} src_t;

typedef struct {              /* tracer/interaction.py:11-39 */
    int obj_id, prim_id;
    v3 n_s, n_g, tex;
    float u, v, min_depth;
} isect_t;

typedef struct { v3 mini, maxi; int base, prim_cnt, all_offset; } lin_node_t;   /* ti_bvh.py:30-53 */
typedef struct { v3 mini, maxi; int obj_idx, prim_idx; } lin_bvh_t;            /* ti_bvh.py:10-28 */

/* flat description handed over by the test harness (ctypes) */
typedef struct {
    int n_prims, n_objects, n_sources, has_vertex_normal;
    const float* prims;       /* n_prims*9 */
    const float* normals;     /* n_prims*3 */
    const float* v_normals;   /* n_prims*9 */
    const int*   obj_info;    /* n_objects*3: start, count, is_sphere */
    const float* obj_aabb;    /* n_objects*6 */
    const int*   emitter_id;  /* n_objects */
    const int*   bxdf_i;      /* n_objects*4: type, is_delta, is_bsdf, 0 */
    const float* bxdf_f;      /* n_objects*13: k_d k_s k_g mean ior */
    const int*   src_i;       /* n_sources*4: type, bool_bits, obj_ref_id, 0 */
    const float* src_f;       /* n_sources*11: intensity dir pos inv_area r */
    float world_ior;
    /* image textures (all NULL / 0 when the scene has none).  Maps: 0 albedo, 1 normal, 2 bump */
    const float* uvs;         /* n_prims*6: per-vertex (u, v) of every triangle */
    const int*   tex_i;       /* n_objects*3*5: per object and map: type (-255 = none), off_x, off_y, w, h */
    const float* tex_f;       /* n_objects*3*2: scale_u, scale_v */
    const float* atlas[3];    /* atlas_h * atlas_w * 3 floats, row-major [y][x] */
    int atlas_w[3], atlas_h[3];
    /* participating media (volumetric path tracer only; NULL = every medium transparent).  Row o < n_objects: the medium attached to
       object o's BSDF; row n_objects: the world medium */
    const int*   med_i;       /* (n_objects+1): type  -1 transparent, 0 hg, 1 multi-hg, 2 rayleigh, 3 mie */
    const float* med_f;       /* (n_objects+1)*16: ior, u_s[3], u_a[3], u_e[3], par[3], pdf[3] */
    /* grid volume (volumetric path tracer; all NULL when the scene declares none): bxdf/volume.py:36-218 */
    const int*   vol_i;       /* 5: type (2 = RGB), xres, yres, zres, phase type */
    const float* vol_f;       /* 33: albedo, inv_T (row-major), trans, mini, maxi, majorant, majorant pdf, phase par, phase lobe weights */
    const float* vol_grid;    /* zres*yres*xres*3 extinction per channel, [z][y][x][c] */
} orc_scene_desc;

typedef struct {
    int width, height;
    int do_crop, start_x, end_x, start_y, end_y;
    int max_bounce, num_shadow_ray;
    int use_rr, use_mis, anti_alias, stratified, brdf_two_sides, use_bvh;
    int rr_bounce_th;
    float rr_threshold;
    float cam_r[9], cam_t[3];
    float inv_focal, half_w, half_h;
    uint32_t seed;
    int volumetric;           /* 0: Renderer.render (vanilla_renderer.py), 1: VolumeRenderer.render (vpt.py) */
} orc_cfg;

typedef struct {
    long long n_samples, n_shade, n_shadow, n_lit, n_draws;
    long long n_extend, n_track;          /* closest-hit queries by the main loop / by track_ray (vpt) */
} orc_stats;

typedef struct {              /* bxdf/medium.py:71-78 + bxdf/phase.py:33-37 */
    int type;
    float ior;
    v3 u_s, u_a, u_e, par, pdf;
} medium_t;

typedef struct {
    int n_prims, n_objects, n_sources, has_vn;
    v3 (*prims)[3];
    v3 (*precom)[3];
    v3 (*vnorm)[3];
    v3* normals;
    int (*obj_info)[3];
    v3 (*aabbs)[2];
    int* emitter_id;
    bxdf_t* bxdf;
    src_t* src;
    float world_ior;
    medium_t* med;            /* n_objects + 1 rows, the last one is the world's */
    /* grid volume, bxdf/volume.py:221-246 */
    int vol_type, vol_res[3];
    v3 vol_albedo, vol_trans, vol_mini, vol_maxi, vol_majorant, vol_pdf;
    m3 vol_inv_T;
    medium_t vol_ph;          /* phase function of the volume (type, par, pdf) */
    float* vol_grid;
    v3 w_aabb_min, w_aabb_max;    /* path_tracer.py:130-138 */
    /* textures: bxdf/texture.py:99-139, path_tracer.py:84-126,261-266 */
    float (*uvs)[3][2];
    int (*tex_i)[3][5];
    float (*tex_f)[3][2];
    float* atlas[3];
    int atlas_w[3], atlas_h[3];
    /* BVH (reference layout) */
    int node_num, bvh_num;
    lin_node_t* nodes;
    lin_bvh_t* bvhs;
} scene_t;

typedef struct {
    const scene_t* sc;
    const orc_cfg* cfg;
    float inv_num_shadow_ray;
    m3 cam_r; v3 cam_t;
} ctx_t;

/* ------------------------------------------------ la/cam_transform.py:51-105 */
static void rotation_between(v3 fixed, v3 target, m3* R) {
    v3 axis = vcross(fixed, target);
    float cos_theta = vdot(fixed, target);
    if (fabsf(cos_theta) < 1.0f - 1e-5f) {
        v3 n = vnormalized(axis);
        float k = 1.0f - cos_theta;
        float nn[3] = {n.x, n.y, n.z};
        float skew[3][3] = {{0.f, -axis.z, axis.y}, {axis.z, 0.f, -axis.x}, {-axis.y, axis.x, 0.f}};
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                float d = (i == j) ? cos_theta : 0.0f;
                R->m[i][j] = (d + (k * nn[i]) * nn[j]) + skew[i][j];
            }
    } else {
        float s = signf_(cos_theta);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) R->m[i][j] = (i == j) ? s : 0.0f;
    }
}
static v3 delocalize_rotate(v3 anchor, v3 local_dir, m3* R_out) {
    m3 R; rotation_between(V(0.f, 1.f, 0.f), anchor, &R);
    if (R_out) *R_out = R;
    return m3mulv(&R, local_dir);
}
static v3 localize_rotate(v3 anchor, v3 global_dir) {
    m3 R; rotation_between(anchor, V(0.f, 1.f, 0.f), &R);
    return m3mulv(&R, global_dir);
}
/* convert_to_raw (cam_transform.py:71-88): (cos_theta, sin_theta, cos_phi, sin_phi) */
static void raw_of_local(v3 l, float raw[4]);
static void convert_to_raw(v3 d_in, v3 normal, float raw[4]) { raw_of_local(localize_rotate(normal, d_in), raw); }
static void raw_of_local(v3 l, float raw[4]) {              /* convert_to_raw(..., localize = False) */
    float cos_theta = l.y;
    float sin_theta = sqrtf(fmaxf(0.f, 1.f - cos_theta * cos_theta));
    float cos_phi = 1.f, sin_phi = 0.f;
    if (sin_theta > 1e-5f) { cos_phi = l.x / sin_theta; sin_phi = l.z / sin_theta; }
    raw[0] = cos_theta; raw[1] = sin_theta; raw[2] = cos_phi; raw[3] = sin_phi;
}

/* ------------------------------------------------------ la/geo_optics.py:14-74 */
static v3 inci_reflect_dir(v3 ray, v3 normal, float* dot_out) {
    float d = vdot(normal, ray);
    if (dot_out) *dot_out = d;
    return vnormalized(vsub(ray, vscale(vscale(normal, 2.f), d)));
}
static v3 schlick_fresnel(v3 r_s, float dot_val) {
    float p = pow5i(1.f - dot_val);
    return vadd(r_s, vscale(V(1.f - r_s.x, 1.f - r_s.y, 1.f - r_s.z), p));
}
static float fresnel_equation(float n_in, float n_out, float cos_inc, float cos_ref) {
    float n1cos_i = n_in * cos_inc, n2cos_i = n_out * cos_inc;
    float n1cos_r = n_in * cos_ref, n2cos_r = n_out * cos_ref;
    float rs = (n1cos_i - n2cos_r) / (n1cos_i + n2cos_r);
    float rp = (n1cos_r - n2cos_i) / (n1cos_r + n2cos_i);
    return 0.5f * (rs * rs + rp * rp);
}
static float fresnel_eval(float cos_v, float n_in, float n_tr) {          /* geo_optics.py:29-44 */
    int neg = cos_v < 0.f;
    float cos_value = neg ? -cos_v : cos_v;
    float ior_in = neg ? n_tr : n_in, ior_tr = neg ? n_in : n_tr;
    float sin_v = sqrtf(fmaxf(0.f, 1.f - cos_value * cos_value));
    float sin_t = ior_in / ior_tr * sin_v;
    float cos_tr = sqrtf(fmaxf(0.f, 1.f - sin_t * sin_t));
    return fresnel_equation(ior_in, ior_tr, cos_value, cos_tr);
}
static int is_total_reflection(float dot_normal, float ni, float nr) {
    return (1.f - sq(ni / nr) * (1.f - sq(dot_normal))) < 0.f;
}
static v3 snell_refraction(v3 incid, v3 normal, float dot_n, float ni, float nr, float* cos_r2_out) {
    float exiting = signf_(dot_n);
    float ratio = ni / nr;
    float cos_r2 = 1.f - sq(ratio) * (1.f - sq(dot_n));
    *cos_r2_out = cos_r2;
    if (cos_r2 > 0.f) {
        v3 a = vscale(incid, ratio);
        v3 b = vscale(normal, ratio * dot_n);
        v3 c = vscale(normal, exiting * sqrtf(cos_r2));
        return vnormalized(vadd(vsub(a, b), c));
    }
    return ZERO3;
}

/* -------------------------------------- sampler/general_sampling.py:16-123 */
static v3 cosine_hemisphere(rng_t* r, float* pdf) {
    float eps = rng_float(r);
    float cos_theta = sqrtf(eps);
    float sin_theta = sqrtf(1.f - eps);
    float phi = F_PI2 * rng_float(r);
    *pdf = cos_theta * F_INV_PI;
    return V(cosf(phi) * sin_theta, cos_theta, sinf(phi) * sin_theta);
}
static v3 mod_phong_hemisphere(rng_t* r, float alpha, float* pdf) {
    float cos_theta = powf(rng_float(r), 1.f / (alpha + 1.f));
    float sin_theta = sqrtf(1.f - cos_theta * cos_theta);
    float phi = F_PI2 * rng_float(r);
    *pdf = 0.5f * (1.f + alpha) * powf(cos_theta, alpha) * F_INV_PI;
    return V(cosf(phi) * sin_theta, cos_theta, sinf(phi) * sin_theta);
}
static v3 uniform_sphere(rng_t* r, float* pdf) {
    float cos_theta = 2.f * rng_float(r) - 1.f;
    float sin_theta = sqrtf(1.f - cos_theta * cos_theta);
    float phi = F_PI2 * rng_float(r);
    *pdf = F_INV_2PI * 0.5f;
    return V(cosf(phi) * sin_theta, cos_theta, sinf(phi) * sin_theta);
}
static v3 fresnel_hemisphere(rng_t* r, float nu, float nv, float* power_coeff) {
    float eps1 = rng_float(r) * 4.f;
    float inner_angle = eps1 - floorf(eps1);
    float tan_phi = sqrtf((nu + 1.f) / (nv + 1.f)) * tanf(F_PI_DIV2 * inner_angle);
    float cos_phi2 = 1.f / (1.f + tan_phi * tan_phi);
    float sin_phi2 = 1.f - cos_phi2;
    float cos_phi = sqrtf(cos_phi2);
    if (eps1 > 1.f && eps1 <= 3.f) cos_phi *= -1.f;
    float sin_phi = sqrtf(sin_phi2) * signf_(2.f - eps1);
    float pc = nu * cos_phi2 + nv * sin_phi2;
    float cos_theta = powf(1.f - rng_float(r), 1.f / (pc + 1.f));
    float sin_theta = sqrtf(1.f - cos_theta * cos_theta);
    *power_coeff = pc;
    return V(cos_phi * sin_theta, cos_theta, sin_phi * sin_theta);
}
static v3 sample_triangle(rng_t* r, v3 dv1, v3 dv2) {
    float u1 = rng_float(r), u2 = rng_float(r);
    v3 pt = vadd(vscale(dv1, u1), vscale(dv2, u2));
    if (u1 + u2 > 1.0f) pt = vsub(vadd(dv1, dv2), pt);
    return pt;
}
static float balance_heuristic(float a, float b) { return (a > 1e-7f) ? a / (a + b) : 0.f; }

/* ------------------------------------------------------- bxdf/brdf.py:147-601 */
static inline int tex_invalid(const isect_t* it) { return it->tex.x < 0.f; }
static inline v3 diffuse_color(const bxdf_t* b, const isect_t* it) { return tex_invalid(it) ? b->k_d : it->tex; }

static v3 eval_lambertian(const bxdf_t* b, const isect_t* it, v3 normal, v3 ray_out) {
    float cosine = fmaxf(0.f, vdot(normal, ray_out));
    return vscale(vscale(diffuse_color(b, it), F_INV_PI), cosine);
}
static v3 sample_lambertian(const bxdf_t* b, const isect_t* it, v3 normal, rng_t* r, v3* spec, float* pdf) {
    v3 local = cosine_hemisphere(r, pdf);
    v3 out = delocalize_rotate(normal, local, NULL);
    *spec = eval_lambertian(b, it, normal, out);
    return out;
}
static v3 eval_phong(const bxdf_t* b, const isect_t* it, v3 ray_in, v3 ray_out) {
    v3 half = vsub(ray_out, ray_in);
    if (vmax(vabs(half)) > BRDF_EPS) half = vnormalized(half); else half = ZERO3;
    float dot_clamp = fmaxf(0.f, vdot(half, it->n_s));
    v3 glossy = vpow_sv(dot_clamp, b->k_g);
    float cosine = fmaxf(0.f, vdot(it->n_s, ray_out));
    v3 spec_part = vmul(b->k_s, vmul(vscale(vadds(b->k_g, 2.0f), 0.5f), glossy));
    return vscale(vscale(vadd(diffuse_color(b, it), spec_part), F_INV_PI), cosine);
}
static v3 eval_mod_phong(const bxdf_t* b, const isect_t* it, v3 ray_in, v3 ray_out) {
    float dot_normal = vdot(it->n_s, ray_out);
    v3 spec = ZERO3;
    if (dot_normal > 0.f) {
        v3 reflect_d = vnormalized(vsub(vscale(vscale(it->n_s, 2.f), dot_normal), ray_out));
        float dot_view = fmaxf(0.f, -vdot(ray_in, reflect_d));
        v3 glossy = vmul(vpow_sv(dot_view, b->k_g), b->k_s);
        spec = vscale(vscale(vmul(vscale(vadds(b->k_g, 2.f), 0.5f), glossy), F_INV_PI), dot_normal);
        spec = vadd(spec, eval_lambertian(b, it, it->n_s, ray_out));
    }
    return spec;
}
static v3 sample_mod_phong(const bxdf_t* b, const isect_t* it, v3 incid, rng_t* r, v3* spec_out, float* pdf_out) {
    float eps = rng_float(r);
    v3 out = V(0.f, 1.f, 0.f);
    v3 spec = ZERO3;
    float pdf = vmax(diffuse_color(b, it));
    float ks_max = vmax(b->k_s);
    if (eps < pdf) {
        float lp;
        out = sample_lambertian(b, it, it->n_s, r, &spec, &lp);
        pdf *= lp;
    } else if (eps < pdf + ks_max) {
        v3 local = mod_phong_hemisphere(r, b->mean.z, &pdf);
        v3 normal = delocalize_rotate(it->n_s, local, NULL);
        out = vnormalized(vadd(vscale(vscale(normal, -2.f), vdot(incid, normal)), incid));
        spec = eval_mod_phong(b, it, incid, out);
        pdf *= ks_max;
    } else {
        pdf = 1.f - pdf - ks_max;
    }
    *spec_out = spec; *pdf_out = pdf;
    return out;
}
/* Fresnel blend (Ashikhmin-Shirley), brdf.py:237-286 */
static void fresnel_cos2_sin2(v3 half_vec, v3 normal, const m3* R, float dot_half, float* c2, float* s2) {
    v3 tx = m3mulv(R, V(1.f, 0.f, 0.f));
    float d = vdot(tx, vnormalized(vsub(half_vec, vscale(normal, dot_half))));
    *c2 = d * d; *s2 = 1.f - *c2;
}
static v3 eval_fresnel_blend(const bxdf_t* b, const isect_t* it, v3 ray_in, v3 ray_out, const m3* R) {
    v3 half_vec = vsub(ray_out, ray_in);
    float dot_out = vdot(it->n_s, ray_out);
    v3 spec = ZERO3;
    if (dot_out > 0.f && vmax(vabs(half_vec)) > 1e-4f) {
        half_vec = vnormalized(half_vec);
        float dot_in = -vdot(it->n_s, ray_in);
        float dot_half = fabsf(vdot(it->n_s, half_vec));
        float dot_hk = fabsf(vdot(half_vec, ray_out));
        v3 fresnel = schlick_fresnel(b->k_s, dot_hk);
        float c2, s2; fresnel_cos2_sin2(half_vec, it->n_s, R, dot_half, &c2, &s2);
        float denom = dot_hk * fmaxf(dot_in, dot_out);
        float lobe = b->k_g.z * powf(dot_half, b->k_g.x * c2 + b->k_g.y * s2);
        v3 specular = vdivs(vscale(fresnel, lobe), denom);
        v3 kd = diffuse_color(b, it);
        v3 diffuse = vmul(vscale(kd, (float)(28. / (23. * 3.14159265358979323846))),
                          V(1.f - b->k_s.x, 1.f - b->k_s.y, 1.f - b->k_s.z));
        float pow5_in = pow5i(1.f - dot_in / 2.f);
        float pow5_out = pow5i(1.f - dot_out / 2.f);
        diffuse = vscale(diffuse, (1.f - pow5_in) * (1.f - pow5_out));
        spec = vscale(vadd(specular, diffuse), dot_out);
    }
    return spec;
}
static v3 sample_fresnel_blend(const bxdf_t* b, const isect_t* it, v3 incid, rng_t* r, v3* spec_out, float* pdf_out) {
    float pc;
    v3 local = fresnel_hemisphere(r, b->k_g.x, b->k_g.y, &pc);
    m3 R;
    v3 ray_half = delocalize_rotate(it->n_s, local, &R);
    /* fresnel_blend_dir, brdf.py:237-244 */
    float dot_incid;
    v3 out = inci_reflect_dir(incid, ray_half, &dot_incid);
    float half_pdf = b->k_g.z * powf(vdot(ray_half, it->n_s), pc);
    float pdf = half_pdf / fmaxf(fabsf(dot_incid), BRDF_EPS);
    int is_valid = vdot(it->n_s, out) > 0.f;
    if (rng_float(r) > 0.5f) {
        v3 s_; float p_;
        out = sample_lambertian(b, it, it->n_s, r, &s_, &p_);
    }
    pdf = 0.5f * (pdf + fabsf(vdot(out, it->n_s)) * F_INV_PI);
    *spec_out = is_valid ? eval_fresnel_blend(b, it, incid, out, &R) : ZERO3;
    *pdf_out = pdf;
    return out;
}
/* Oren-Nayar, brdf.py:312-342 */
static v3 eval_oren_nayar(const bxdf_t* b, const isect_t* it, v3 ray_in, v3 ray_out) {
    float wi[4], wo[4];
    convert_to_raw(vneg(ray_in), it->n_s, wi);
    convert_to_raw(ray_out, it->n_s, wo);
    float sin_i = wi[1], sin_o = wo[1];
    float max_cos = 0.f;
    if (sin_i > 1e-5f && sin_o > 1e-5f) {
        float d_cos = wi[2] * wo[2] + wi[3] * wo[3];
        max_cos = fmaxf(0.f, d_cos);
    }
    float sin_alpha, tan_beta;
    float aci = fabsf(wi[0]), aco = fabsf(wo[0]);
    if (aci > aco) { sin_alpha = sin_o; tan_beta = sin_i / aci; }
    else           { sin_alpha = sin_i; tan_beta = sin_o / aco; }
    float f = b->k_g.x + b->k_g.y * max_cos * sin_alpha * tan_beta;
    return vscale(vscale(vscale(diffuse_color(b, it), F_INV_PI), f), fabsf(wo[0]));
}
/* Thin coat, brdf.py:348-422 */
static v3 sample_thin_coat(const bxdf_t* b, const isect_t* it, v3 incid, rng_t* r, v3* spec_out, float* pdf_out, int* is_specular) {
    float pdf = 1.0f; v3 spec = ZERO3; v3 out = V(0.f, 1.f, 0.f);
    float dot_normal = vdot(incid, it->n_s);
    float cos_r2;
    v3 refra_in = snell_refraction(incid, it->n_s, dot_normal, 1.0f, b->k_g.z, &cos_r2);
    float in_ref_F = fresnel_equation(1.f, b->k_g.x, fabsf(dot_normal), sqrtf(cos_r2));
    *is_specular = 0;
    if (rng_float(r) > in_ref_F) {
        v3 local = cosine_hemisphere(r, &pdf);
        out = delocalize_rotate(it->n_s, local, NULL);
        float dot_out = vdot(out, it->n_s);
        if (!is_total_reflection(dot_out, b->k_g.z, 1.0f)) {
            v3 refra_out = snell_refraction(out, it->n_s, dot_out, b->k_g.z, 1.0f, &cos_r2);
            float out_ref_F = fresnel_equation(b->k_g.z, 1.f, fabsf(dot_out), sqrtf(cos_r2));
            pdf *= (1.f - in_ref_F);
            out = refra_out;
            spec = eval_oren_nayar(b, it, refra_in, out);
            spec = vscale(spec, (1.f - in_ref_F) * (1.f - out_ref_F));
        }
    } else {
        spec = vscale(b->k_s, in_ref_F);
        out = inci_reflect_dir(incid, it->n_s, NULL);
        pdf = in_ref_F;
        *is_specular = 1;
    }
    *spec_out = spec; *pdf_out = pdf;
    return out;
}
static v3 eval_thin_coating(const bxdf_t* b, const isect_t* it, v3 ray_in, v3 ray_out) {
    v3 ret;
    v3 reflect = inci_reflect_dir(ray_in, it->n_s, NULL);
    float dot_in = vdot(ray_in, it->n_s);
    float cos_r2;
    v3 refra_in = snell_refraction(ray_in, it->n_s, dot_in, 1.0f, b->k_g.z, &cos_r2);
    float in_ref_F = fresnel_equation(1.f, b->k_g.z, fabsf(dot_in), sqrtf(cos_r2));
    if (fabsf(vdot(ray_out, reflect)) > (1.f - 1e-4f)) {
        ret = vscale(b->k_s, in_ref_F);
    } else {
        float dot_out = vdot(ray_out, it->n_s);
        v3 refra_out = snell_refraction(ray_out, it->n_s, dot_out, 1.0f, b->k_g.z, &cos_r2);
        float out_ref_F = fresnel_equation(1.0f, b->k_g.z, fabsf(dot_out), sqrtf(cos_r2));
        ret = vscale(eval_oren_nayar(b, it, refra_in, refra_out), 1.f - fmaxf(in_ref_F, out_ref_F));
    }
    return ret;
}
static float thin_coat_fresnel(const bxdf_t* b, const isect_t* it, v3 ray_in) {
    float dot_in = vdot(ray_in, it->n_s);
    float ratio = 1.0f / b->k_g.z;
    float cos_r2 = 1.f - sq(ratio) * (1.f - sq(dot_in));
    return fresnel_equation(1.f, b->k_g.z, fabsf(dot_in), sqrtf(cos_r2));
}

/* BRDF.eval, brdf.py:503-526 */
/* ------------------------------------------------ sampler/microfacet.py:27-176 + bxdf/brdf.py:428-484
 * Trowbridge-Reitz microfacet BRDF.  Upstream ships it switched off (`__ENABLE_MICROFACET__ = False`, brdf.py:8) and its parser
 * then turns a microfacet BRDF into a Lambertian one (brdf.py:60-65), so BRDF type 3 only ever reaches the device with the switch
 * on: that is what type 3 means here.  Pinned by tests/golden/microfacet_*.npz, generated with the switch flipped. */
static float trow_reitz_D(const float raw[4], v3 alphas) {
    float pdf = 0.f;
    if (raw[0] > 0.f) {
        float wh_dot2 = raw[0] * raw[0];
        float wh_dot4 = wh_dot2 * wh_dot2;
        float tan_theta2 = raw[1] * raw[1] / wh_dot2;
        float ax = alphas.x, ay = alphas.y;
        float e = (raw[2] * raw[2] / (ax * ax) + raw[3] * raw[3] / (ay * ay)) * tan_theta2;
        pdf = 1.f / (F_PI * ax * ay * wh_dot4 * (1.f + e) * (1.f + e));
    }
    return pdf;
}
static float trow_reitz_lambda(v3 dir_vec, v3 alphas, v3 normal) {
    float value = 0.f, raw[4];
    convert_to_raw(dir_vec, normal, raw);
    float abs_cos = fabsf(raw[0]);
    if (abs_cos > 1e-5f) {
        float abs_tan = raw[1] / abs_cos;
        float alpha = sqrtf(raw[2] * raw[2] * alphas.x * alphas.x + raw[3] * raw[3] * alphas.y * alphas.y);
        float at2 = alpha * abs_tan;
        at2 *= at2;
        value = (-1.f + sqrtf(1.f + at2)) * 0.5f;
    }
    return value;
}
static float trow_reitz_G1(v3 d, v3 alphas, v3 normal) { return 1.f / (1.f + trow_reitz_lambda(d, alphas, normal)); }
static float trow_reitz_G(v3 incid, v3 outdir, v3 alphas, v3 normal) {
    return 1.f / (1.f + trow_reitz_lambda(incid, alphas, normal) + trow_reitz_lambda(outdir, alphas, normal));
}
static void trow_reitz_slopes(float cos_theta, rng_t* r, float* sx, float* sy) {       /* microfacet.py:65-99 */
    float u1 = rng_float(r), u2 = rng_float(r);
    if (cos_theta > (float)(1.0 - 1e-5)) {
        float rad = sqrtf(u1 / (1.f - u1));
        float phi = 6.28318530718f * u2;
        *sx = rad * cosf(phi); *sy = rad * sinf(phi);
        return;
    }
    float sin_theta = sqrtf(fmaxf(0.f, 1.f - cos_theta * cos_theta));
    float tan_theta = sin_theta / cos_theta;
    float G1 = 2.f / (1.f + sqrtf(1.f + tan_theta * tan_theta));
    float A = 2.f * u1 / G1 - 1.f;
    float tmp = fminf(1e10f, 1.f / (A * A - 1.f));
    float D = sqrtf(fmaxf(tan_theta * tan_theta * tmp * tmp - (A * A - tan_theta * tan_theta) * tmp, 0.f));
    float s1 = tan_theta * tmp - D;
    float s2 = s1 + D * 2.f;
    float slope_x = ((A < 0.f) || (s2 > 1.f / tan_theta)) ? s1 : s2;
    float S;
    if (u2 > 0.5f) { S = 1.f; u2 = 2.0f * (u2 - 0.5f); }
    else { S = -1.f; u2 = 2.f * (0.5f - u2); }
    float z = (u2 * (u2 * (u2 * 0.27385f - 0.73369f) + 0.46341f)) / (u2 * (u2 * (u2 * 0.093073f + 0.309420f) - 1.0f) + 0.597999f);
    *sx = slope_x; *sy = S * z * sqrtf(1.f + slope_x * slope_x);
}
static v3 trow_reitz_sample(v3 incid, v3 normal, float ax, float ay, rng_t* r) {       /* microfacet.py:101-124: a LOCAL direction */
    v3 stretch = vnormalized(vmul(incid, V(ax, 1.f, ay)));
    float raw[4], sx, sy;
    convert_to_raw(stretch, normal, raw);
    trow_reitz_slopes(raw[0], r, &sx, &sy);
    float tmp = raw[2] * sx - raw[3] * sy;
    sy = raw[3] * sx + raw[2] * sy;
    sx = tmp;
    sx = ax * sx; sy = ay * sy;
    return vnormalized(V(-sx, 1.f, -sy));
}
static v3 trow_reitz_sample_wh(v3 incid, v3 normal, float ax, float ay, rng_t* r, float raw[4]) {     /* microfacet.py:161-169 */
    int flip = vdot(incid, normal) > 0.f;
    v3 wh = trow_reitz_sample(flip ? incid : vneg(incid), normal, ax, ay, r);
    if (flip) wh = vneg(wh);
    raw_of_local(wh, raw);
    return wh;
}
static float trow_reitz_pdf(v3 incid, v3 wh, v3 alphas, v3 normal) {                  /* microfacet.py:171-176 */
    float raw[4];
    convert_to_raw(wh, normal, raw);
    return trow_reitz_D(raw, alphas) * trow_reitz_G1(incid, alphas, normal) * fabsf(vdot(wh, incid)) / fabsf(vdot(normal, incid));
}
static v3 eval_microfacet_with_raw(const bxdf_t* b, const isect_t* it, v3 wh, const float raw[4], v3 ray_in, v3 ray_out) {   /* brdf.py:457-471 */
    v3 ret = ZERO3;
    if (fabsf(wh.x) > 1e-7f || fabsf(wh.y) > 1e-7f || fabsf(wh.z) > 1e-7f) {
        wh = vnormalized(wh);
        float dot_hk = vdot(wh, ray_out);
        float fresnel = fresnel_eval(dot_hk, b->k_s.x, b->k_s.y);
        float cosine_term = fabsf(vdot(it->n_s, ray_out));
        ret = vscale(vscale(vscale(vscale(diffuse_color(b, it), trow_reitz_D(raw, b->k_g)), trow_reitz_G(vneg(ray_in), ray_out, b->k_g, it->n_s)), fresnel), cosine_term);
    }
    return ret;
}
static v3 eval_microfacet(const bxdf_t* b, const isect_t* it, v3 ray_in, v3 ray_out) {                /* brdf.py:473-484 */
    v3 ret = ZERO3;
    float cos_mult = vdot(it->n_s, ray_out) * vdot(it->n_s, ray_in);
    if (cos_mult < 0.f) {
        v3 wh = vnormalized(vsub(ray_out, ray_in));
        float raw[4];
        convert_to_raw(wh, it->n_s, raw);
        ret = vdivs(eval_microfacet_with_raw(b, it, wh, raw, ray_in, ray_out), -4.f * cos_mult);
    }
    return ret;
}
static v3 sample_microfacet(const bxdf_t* b, const isect_t* it, v3 incid, rng_t* r, v3* spec_out, float* pdf_out) {   /* brdf.py:429-455 */
    float raw[4];
    v3 local_wh = trow_reitz_sample_wh(incid, it->n_s, b->k_g.x, b->k_g.y, r, raw);
    v3 half_vector = delocalize_rotate(it->n_s, local_wh, NULL);
    float dot_val = -vdot(incid, half_vector);
    v3 spec = ZERO3, out_d = V(0.f, 1.f, 0.f);
    float pdf = 1.0f;
    if (dot_val > 0.f) {
        out_d = inci_reflect_dir(incid, half_vector, NULL);
        float cos_o = vdot(it->n_s, out_d), cos_i = vdot(it->n_s, incid);
        if (cos_o * cos_i < 0.f) {
            cos_i = fabsf(cos_i); cos_o = fabsf(cos_o);
            if (cos_o > 1e-7f && cos_i > 1e-7f) {
                spec = vdivs(eval_microfacet_with_raw(b, it, half_vector, raw, incid, out_d), 4.f * cos_o * cos_i);
                pdf = trow_reitz_pdf(vneg(incid), half_vector, b->k_g, it->n_s);
                pdf /= 4.f * dot_val;
            }
        }
    }
    *spec_out = spec; *pdf_out = pdf;
    return out_d;
}

static v3 brdf_eval(const bxdf_t* b, const isect_t* it, v3 incid, v3 out) {
    v3 ret = ZERO3;
    if (vdot(incid, it->n_g) * vdot(out, it->n_g) < 0.f) {
        switch (b->type) {
        case 0: ret = eval_phong(b, it, incid, out); break;
        case 1: ret = eval_lambertian(b, it, it->n_s, out); break;
        case 4: ret = eval_mod_phong(b, it, incid, out); break;
        case 5: { m3 R; rotation_between(V(0.f, 1.f, 0.f), it->n_s, &R); ret = eval_fresnel_blend(b, it, incid, out, &R); break; }
        case 6: ret = eval_oren_nayar(b, it, incid, out); break;
        case 7: ret = eval_thin_coating(b, it, incid, out); break;
        case 3: ret = eval_microfacet(b, it, incid, out); break;
        default: break;            /* specular(2) -> 0 */
        }
    }
    return ret;
}
/* BRDF.sample_new_rays, brdf.py:528-560 */
static v3 brdf_sample(const bxdf_t* b, const isect_t* it, v3 incid, rng_t* r, v3* spec_out, float* pdf_out, int* is_specular) {
    v3 dir = V(0.f, 1.f, 0.f); v3 spec = V(1.f, 1.f, 1.f); float pdf = 1.0f;
    *is_specular = 0;
    switch (b->type) {
    case 0: {
        v3 local = cosine_hemisphere(r, &pdf);
        dir = delocalize_rotate(it->n_s, local, NULL);
        spec = eval_phong(b, it, incid, dir);
        break; }
    case 1: case 6: dir = sample_lambertian(b, it, it->n_s, r, &spec, &pdf); break;
    case 2: dir = inci_reflect_dir(incid, it->n_s, NULL); spec = diffuse_color(b, it); pdf = 1.0f; break;
    case 7: dir = sample_thin_coat(b, it, incid, r, &spec, &pdf, is_specular); break;
    case 4: dir = sample_mod_phong(b, it, incid, r, &spec, &pdf); break;
    case 5: dir = sample_fresnel_blend(b, it, incid, r, &spec, &pdf); break;
    case 3: dir = sample_microfacet(b, it, incid, r, &spec, &pdf); break;
    default: break;
    }
    if (!(vdot(dir, it->n_g) > 0.f)) spec = ZERO3;     /* brdf.py:558-559 */
    *spec_out = spec; *pdf_out = pdf;
    return dir;
}
/* BRDF.get_pdf, brdf.py:562-601 */
static float brdf_pdf(const bxdf_t* b, const isect_t* it, v3 outdir, v3 incid) {
    float pdf = 0.f;
    float dot_outdir = vdot(it->n_s, outdir);
    float dot_indir = vdot(it->n_s, incid);
    if (dot_outdir * dot_indir < 0.f) {
        switch (b->type) {
        case 0: case 1: case 6: pdf = dot_outdir * F_INV_PI; break;
        case 4: {
            float gloss = b->mean.z;
            v3 reflect_view = inci_reflect_dir(incid, it->n_s, NULL);
            float dot_ref_out = fmaxf(0.f, vdot(reflect_view, outdir));
            float diffuse_pdf = dot_outdir * F_INV_PI;
            float specular_pdf = 0.5f * (gloss + 1.f) * F_INV_PI * powf(dot_ref_out, gloss);
            pdf = vmax(diffuse_color(b, it)) * diffuse_pdf + vmax(b->k_s) * specular_pdf;
            break; }
        case 7: {
            v3 reflect = inci_reflect_dir(incid, it->n_s, NULL);
            float F = thin_coat_fresnel(b, it, incid);
            pdf = (fabsf(vdot(outdir, reflect)) > (1.f - 1e-3f)) ? F : (1.f - F) * dot_outdir * F_INV_PI;
            break; }
        case 5: {
            v3 half_vec = vnormalized(vsub(outdir, incid));
            float dot_half = vdot(half_vec, it->n_s);
            m3 R; rotation_between(V(0.f, 1.f, 0.f), it->n_s, &R);
            float c2, s2; fresnel_cos2_sin2(half_vec, it->n_s, &R, dot_half, &c2, &s2);
            pdf = b->k_g.z * powf(dot_half, b->k_g.x * c2 + b->k_g.y * s2) / fabsf(vdot(incid, half_vec));
            pdf = 0.5f * (pdf + dot_outdir * F_INV_PI);
            break; }
        case 3: {                                                    /* brdf.py:597-600 */
            v3 wh = vnormalized(vsub(outdir, incid));
            pdf = trow_reitz_pdf(vneg(incid), wh, b->k_g, it->n_s) / (-4.f * vdot(wh, incid));
            break; }
        default: break;
        }
    }
    return pdf;
}

/* ------------------------------------------------------- bxdf/bsdf.py:61-262 */
static v3 sample_det_refraction(const bxdf_t* b, const isect_t* it, v3 incid, float world_ior, rng_t* r, v3* spec_out, float* pdf_out) {
    float dot_normal = vdot(incid, it->n_s);
    int entering = dot_normal < 0.f;
    float ni = entering ? world_ior : b->ior;
    float nr = entering ? b->ior : world_ior;
    float ret_pdf = 1.0f;
    v3 ret_dir;
    v3 ret_int = diffuse_color(b, it);
    if (is_total_reflection(dot_normal, ni, nr)) {
        ret_dir = vnormalized(vsub(incid, vscale(vscale(it->n_s, 2.f), dot_normal)));
    } else {
        float cos_r2;
        v3 refra = snell_refraction(incid, it->n_s, dot_normal, ni, nr, &cos_r2);
        float reflect_ratio = fresnel_equation(ni, nr, fabsf(dot_normal), sqrtf(cos_r2));
        if (rng_float(r) > reflect_ratio) {
            ret_pdf = 1.f - reflect_ratio;
            ret_dir = refra;              /* mode == TRANSPORT_UNI: no (ni/nr)^2 factor */
        } else {
            ret_dir = vnormalized(vsub(incid, vscale(vscale(it->n_s, 2.f), dot_normal)));
            ret_pdf = reflect_ratio;
        }
    }
    *spec_out = vscale(ret_int, ret_pdf); *pdf_out = ret_pdf;
    return ret_dir;
}
static v3 eval_det_refraction(const bxdf_t* b, const isect_t* it, v3 ray_in, v3 ray_out, float world_ior) {
    float dot_out = vdot(ray_out, it->n_s);
    int entering = dot_out < 0.f;
    float ni = entering ? world_ior : b->ior;
    float nr = entering ? b->ior : world_ior;
    v3 ret = ZERO3;
    v3 kd = diffuse_color(b, it);
    v3 ref_dir = vnormalized(vsub(ray_out, vscale(vscale(it->n_s, 2.f), dot_out)));
    if (is_total_reflection(dot_out, ni, nr)) {
        if (vdot(ref_dir, ray_in) > 1.f - 5e-5f) ret = kd;
    } else {
        float cos_r2;
        v3 refra = snell_refraction(ray_out, it->n_s, dot_out, ni, nr, &cos_r2);
        if (cos_r2 > 0.f) {
            float rr = fresnel_equation(ni, nr, fabsf(dot_out), sqrtf(cos_r2));
            if (vdot(refra, ray_in) > 1.f - 1e-4f) ret = vscale(kd, 1.f - rr);
            else if (vdot(ref_dir, ray_in) > 1.f - 1e-4f) ret = vscale(kd, rr);
        } else {
            if (vdot(ref_dir, ray_in) > 1.f - 1e-4f) ret = kd;
        }
    }
    return ret;
}
static v3 sample_lambertian_trans(const bxdf_t* b, const isect_t* it, v3 incid, float world_ior, rng_t* r, v3* spec_out, float* pdf_out, int* is_delta) {
    float dot_normal = vdot(incid, it->n_s);
    int entering = dot_normal < 0.f;
    float ni = entering ? world_ior : b->ior;
    float nr = entering ? b->ior : world_ior;
    float ret_pdf = 1.0f, fresnel = 1.0f;
    *is_delta = 1;
    v3 ret_dir;
    v3 ret_int = diffuse_color(b, it);
    if (is_total_reflection(dot_normal, ni, nr)) {
        ret_dir = vnormalized(vsub(incid, vscale(vscale(it->n_s, 2.f), dot_normal)));
    } else {
        float ratio = ni / nr;
        float cos_r2 = 1.f - sq(ratio) * (1.f - sq(dot_normal));
        float reflect_ratio = fresnel_equation(ni, nr, fabsf(dot_normal), sqrtf(cos_r2));
        if (rng_float(r) > reflect_ratio) {
            fresnel = 1.f - reflect_ratio;
            v3 local = cosine_hemisphere(r, &ret_pdf);
            ret_pdf *= fresnel;
            v3 normal = vscale(it->n_s, signf_(dot_normal));
            ret_dir = delocalize_rotate(normal, local, NULL);
            float cosine = fmaxf(0.f, vdot(normal, ret_dir));
            ret_int = vscale(ret_int, F_INV_PI * cosine);
            *is_delta = 0;
        } else {
            ret_dir = vnormalized(vsub(incid, vscale(vscale(it->n_s, 2.f), dot_normal)));
            fresnel = reflect_ratio;
            ret_pdf = reflect_ratio;
        }
    }
    *spec_out = vscale(ret_int, fresnel); *pdf_out = ret_pdf;
    return ret_dir;
}
static v3 eval_lambertian_trans(const bxdf_t* b, const isect_t* it, v3 ray_in, v3 ray_out, float world_ior) {
    float dot_out = vdot(ray_out, it->n_s);
    int entering = dot_out < 0.f;
    float ni = entering ? world_ior : b->ior;
    float nr = entering ? b->ior : world_ior;
    v3 ret = ZERO3;
    v3 kd = diffuse_color(b, it);
    v3 ref_dir = vnormalized(vsub(ray_out, vscale(vscale(it->n_s, 2.f), dot_out)));
    if (is_total_reflection(dot_out, ni, nr)) {
        if (vdot(ref_dir, ray_in) > 1.f - 1e-4f) ret = kd;
    } else {
        float ratio = ni / nr;
        float cos_r2 = 1.f - sq(ratio) * (1.f - sq(dot_out));
        float dot_in = vdot(ray_in, it->n_s);
        if (cos_r2 > 0.f) {
            float rr = fresnel_equation(ni, nr, fabsf(dot_out), sqrtf(cos_r2));
            if (dot_in * dot_out < 0.f) {
                if (vdot(ref_dir, ray_in) > 1.f - 1e-4f) ret = vscale(kd, rr);
            } else {
                ret = vscale(kd, (1.f - rr) * F_INV_PI * fabsf(dot_out));
            }
        } else {
            if (vdot(ref_dir, ray_in) > 1.f - 1e-4f) ret = kd;
        }
    }
    return ret;
}
/* BSDF.get_pdf, bsdf.py:211-236 */
static float bsdf_pdf(const bxdf_t* b, const isect_t* it, v3 outdir, v3 incid, float world_ior) {
    float pdf = 0.f;
    if (b->type == -1) {
        pdf = (vdot(incid, outdir) > 1.f - 1e-4f) ? 1.f : 0.f;
    } else {
        float dot_out = vdot(outdir, it->n_s);
        int entering = dot_out < 0.f;
        float ni = entering ? world_ior : b->ior;
        float nr = entering ? b->ior : world_ior;
        v3 ref_dir = vnormalized(vsub(outdir, vscale(vscale(it->n_s, 2.f), dot_out)));
        float cos_r2;
        v3 refra = snell_refraction(outdir, it->n_s, dot_out, ni, nr, &cos_r2);
        if (cos_r2 > 0.0f) {
            float rr = fresnel_equation(ni, nr, fabsf(dot_out), sqrtf(cos_r2));
            if (vdot(ref_dir, incid) > 1.f - 1e-4f) pdf = rr;
            else {
                if (b->type == 0 && vdot(refra, incid) > 1.f - 1e-4f) pdf = 1.f - rr;
                else if (b->type == 1 && (vdot(incid, it->n_s) * dot_out > 0.f)) pdf = (1.f - rr) * fabsf(dot_out) * F_INV_PI;
            }
        } else {
            if (vdot(ref_dir, incid) > 1.f - 1e-4f) pdf = 1.f;
        }
    }
    return pdf;
}

/* ---------------------------- PathTracer dispatch, path_tracer.py:424-526 */
static v3 pt_sample_new_ray(const ctx_t* c, isect_t* it, v3 incid, rng_t* r, v3* spec, float* pdf, int* is_specular) {
    const bxdf_t* b = &c->sc->bxdf[it->obj_id];
    if (!b->is_bsdf) {
        if (c->cfg->brdf_two_sides && vdot(incid, it->n_s) > 0.f) { it->n_s = vneg(it->n_s); it->n_g = vneg(it->n_g); }
        return brdf_sample(b, it, incid, r, spec, pdf, is_specular);
    }
    v3 dir = ZERO3; *spec = ZERO3; *pdf = 0.f; *is_specular = 0;
    if (b->type == 0) dir = sample_det_refraction(b, it, incid, c->sc->world_ior, r, spec, pdf);
    else if (b->type == 1) dir = sample_lambertian_trans(b, it, incid, c->sc->world_ior, r, spec, pdf, is_specular);
    return dir;
}
static v3 pt_eval(const ctx_t* c, isect_t* it, v3 incid, v3 out) {
    const bxdf_t* b = &c->sc->bxdf[it->obj_id];
    if (!b->is_bsdf) {
        if (c->cfg->brdf_two_sides && vdot(incid, it->n_s) > 0.f) { it->n_s = vneg(it->n_s); it->n_g = vneg(it->n_g); }
        return brdf_eval(b, it, incid, out);
    }
    if (b->type == 0) return eval_det_refraction(b, it, incid, out, c->sc->world_ior);
    if (b->type == 1) return eval_lambertian_trans(b, it, incid, out, c->sc->world_ior);
    return ZERO3;
}
static float pt_surface_pdf(const ctx_t* c, isect_t* it, v3 outdir, v3 incid) {
    const bxdf_t* b = &c->sc->bxdf[it->obj_id];
    if (!b->is_bsdf) {
        if (c->cfg->brdf_two_sides && vdot(incid, it->n_s) > 0.f) { it->n_s = vneg(it->n_s); it->n_g = vneg(it->n_g); }
        return brdf_pdf(b, it, outdir, incid);
    }
    return bsdf_pdf(b, it, outdir, incid, c->sc->world_ior);
}
static int pt_is_delta(const ctx_t* c, int idx) { return (idx >= 0) ? c->sc->bxdf[idx].is_delta : 0; }

/* --------------------------------- emitters/abtract_source.py:35-232 */
static float distance_attenuate(v3 x) { return fminf(1.0f / fmaxf(vnorm2(x), 1e-5f), 1.0f); }

static v3 src_sample_hit(const ctx_t* c, const src_t* s, v3 hit_pos, rng_t* r, v3* ret_int_out, float* ret_pdf_out) {
    const scene_t* sc = c->sc;
    v3 ret_int = s->intensity, ret_pos = s->pos;
    float ret_pdf = 1.0f;
    v3 normal = ZERO3;
    if (s->type == 0) {
        ret_int = vscale(ret_int, distance_attenuate(vsub(hit_pos, ret_pos)));
    } else if (s->type == 1) {
        ret_pdf = s->inv_area;
        int is_sphere = sc->obj_info[s->obj_ref_id][2];
        if (is_sphere) {
            int tri_id = sc->obj_info[s->obj_ref_id][0];
            v3 center = sc->precom[tri_id][0];
            float radius = sc->precom[tri_id][1].x;
            v3 to_hit = vnormalized(vsub(hit_pos, center));
            float pdf;
            v3 local = uniform_sphere(r, &pdf);
            normal = delocalize_rotate(to_hit, local, NULL);
            ret_pos = vadd(center, vscale(normal, radius));
            ret_pdf = pdf / (radius * radius);
        } else {
            int mesh_num = sc->obj_info[s->obj_ref_id][1];
            int tri_id = pymod(rng_int(r), mesh_num) + sc->obj_info[s->obj_ref_id][0];
            normal = sc->normals[tri_id];
            v3 dv1 = sc->precom[tri_id][0], dv2 = sc->precom[tri_id][1];
            ret_pos = vadd(sample_triangle(r, dv1, dv2), sc->precom[tri_id][2]);
        }
        v3 diff = vsub(hit_pos, ret_pos);
        float dot_light = vdot(vnormalized(diff), normal);
        if (dot_light <= 0.0f) {
            ret_int = ZERO3; ret_pdf = 1.0f;
        } else {
            float diff_norm2 = vnorm2(diff);
            ret_pdf *= diff_norm2 / dot_light;
            ret_int = (ret_pdf > 0.0f) ? vdivs(ret_int, ret_pdf) : ZERO3;
        }
    } else if (s->type == 2) {
        v3 to_hit = vsub(hit_pos, ret_pos);
        float depth = fmaxf(vnorm(to_hit), 1e-5f);
        to_hit = vdivs(to_hit, depth);
        float cos_val = vdot(to_hit, s->dir);
        if (cos_val > s->r) ret_int = vdivs(ret_int, depth * depth);
        else ret_int = ZERO3;
    } else if (s->type == 4) {
        ret_pdf = 0.f;
        if (s->r > 0.f) {
            v3 to_hit = vsub(hit_pos, s->pos);
            float proj_d = vdot(to_hit, s->dir);
            if (proj_d > 0.0f) {
                float dist = sqrtf(vnorm2(to_hit) - proj_d * proj_d);
                if (dist < s->r) { ret_pos = vsub(hit_pos, vscale(s->dir, proj_d)); normal = s->dir; }
                else ret_int = ZERO3;
            }
        } else ret_int = ZERO3;
    }
    (void)normal;
    *ret_int_out = ret_int; *ret_pdf_out = ret_pdf;
    return ret_pos;
}
static v3 src_eval_le(const src_t* s, v3 inci_dir, v3 normal) {
    v3 ret = ZERO3;
    if (s->type == 1) {
        float dot_light = -vdot(vnormalized(inci_dir), normal);
        if (dot_light > 0.f) ret = s->intensity;
    }
    return ret;
}
static float src_solid_angle_pdf(const src_t* s, const isect_t* it, v3 incid_dir) {
    float dot_res = fabsf(vdot(incid_dir, it->n_s));
    float area_pdf = (s->type == 1) ? s->inv_area : 0.f;
    return (dot_res > 0.0f) ? area_pdf * sq(it->min_depth) / dot_res : 0.0f;
}

/* sample_light, path_tracer.py:537-554 */
static const src_t* pt_sample_light(const ctx_t* c, int no_sample, rng_t* r, float* pdf, int* valid) {
    int n = c->sc->n_sources;
    int idx = pymod(rng_int(r), n);
    *pdf = 1.f / (float)n;
    *valid = 1;
    if (no_sample >= 0) {
        if (n <= 1) *valid = 0;
        else {
            idx = pymod(rng_int(r), n - 1);
            if (idx >= no_sample) idx += 1;
            *pdf = 1.f / (float)(n - 1);
        }
    }
    return &c->sc->src[idx];
}

/* ----------------------------------------------------------- intersection */
/* Interaction tail shared by both intersectors (tracer_base.py:209-237, path_tracer.py:375-394) */
static void finish_isect(const scene_t* sc, isect_t* it, int sphere_flag, v3 ray, v3 start_p) {
    it->n_g = V(1.f, 0.f, 0.f); it->n_s = V(1.f, 0.f, 0.f);
    it->tex = V(-1.f, -1.f, -1.f);
    if (it->obj_id >= 0) {
        if (sphere_flag) {
            v3 center = sc->prims[it->prim_id][0];
            it->n_g = vnormalized(vsub(vadd(start_p, vscale(ray, it->min_depth)), center));
            it->u = (atan2f(it->n_g.y, it->n_g.x) + F_PI) * F_INV_2PI;
            it->v = acosf(it->n_g.z) * F_INV_PI;
            it->n_s = it->n_g;
        } else {
            it->n_g = sc->normals[it->prim_id];
            if (sc->has_vn) {
                const v3* vn = sc->vnorm[it->prim_id];
                it->n_s = vadd(vadd(vscale(vn[0], 1.f - it->u - it->v), vscale(vn[1], it->u)), vscale(vn[2], it->v));
            } else it->n_s = it->n_g;
        }
    }
}
/* sphere test shared (tracer_base.py:184-199); returns ray_t or -1 */
static float sphere_t(const scene_t* sc, int prim, v3 ray, v3 start_p) {
    v3 center = sc->prims[prim][0];
    float r = sc->prims[prim][1].x;
    float radius2 = r * r;
    v3 s2c = vsub(center, start_p);
    float center_norm2 = vnorm2(s2c);
    float proj = vdot(ray, s2c);
    float c2ray = center_norm2 - proj * proj;
    if (c2ray >= radius2) return -1.f;
    float cut = sqrtf(radius2 - c2ray);
    return proj + ((center_norm2 > radius2 + 1e-4f) ? -cut : cut);
}
static inline void tri_uvt(const scene_t* sc, int prim, v3 ray, v3 start_p, float* u, float* v, float* t) {
    v3 p1 = sc->prims[prim][0];
    v3 v1 = sc->precom[prim][0], v2 = sc->precom[prim][1];
    m3 inv; inverse_cols(v1, v2, vneg(ray), &inv);
    v3 s = vsub(start_p, p1);
    v3 uvt = m3mulv(&inv, s);
    *u = uvt.x; *v = uvt.y; *t = uvt.z;
}
/* TracerBase.aabb_test, tracer_base.py:159-166 (division by the ray, not inv_ray) */
static int obj_aabb_test(const scene_t* sc, int idx, v3 ray, v3 ray_o, float* t_near) {
    v3 tmin = vdiv(vsub(sc->aabbs[idx][0], ray_o), ray);
    v3 tmax = vdiv(vsub(sc->aabbs[idx][1], ray_o), ray);
    float tn = vmax(vminv(tmin, tmax));
    float tf = vmin(vmaxv(tmin, tmax));
    *t_near = tn;
    return (tn < tf) && tf > 0.f;
}
/* TracerBase.ray_intersect, tracer_base.py:168-237 */
static void ray_intersect_brute(const scene_t* sc, v3 ray, v3 start_p, float min_depth_in, isect_t* it) {
    int obj_id = -1, prm_id = -1, sphere_flag = 0;
    float cu = 0.f, cv = 0.f;
    float min_depth = (min_depth_in > 0.0f) ? min_depth_in - 1e-4f : 1e7f;
    for (int o = 0; o < sc->n_objects; o++) {
        float t_near;
        if (!obj_aabb_test(sc, o, ray, start_p, &t_near)) continue;
        if (t_near > min_depth) continue;
        int start_id = sc->obj_info[o][0];
        if (sc->obj_info[o][2]) {
            float rt = sphere_t(sc, start_id, ray, start_p);
            /* `continue` on a miss: rt = -1 fails the > 1e-4 test */
            if (rt > 1e-4f && rt < min_depth) { min_depth = rt; obj_id = o; prm_id = start_id; sphere_flag = 1; }
        } else {
            int tri_num = sc->obj_info[o][1];
            for (int m = start_id; m < tri_num + start_id; m++) {
                float u, v, t; tri_uvt(sc, m, ray, start_p, &u, &v, &t);
                if (u >= 0.f && v >= 0.f && u + v <= 1.0f)
                    if (t > 1e-4f && t < min_depth) { min_depth = t; obj_id = o; prm_id = m; cu = u; cv = v; sphere_flag = 0; }
            }
        }
    }
    it->obj_id = obj_id; it->prim_id = prm_id; it->u = cu; it->v = cv; it->min_depth = min_depth;
    finish_isect(sc, it, sphere_flag, ray, start_p);
}
/* TracerBase.does_intersect, tracer_base.py:239-278 */
static int does_intersect_brute(const scene_t* sc, v3 ray, v3 start_p, float min_depth_in) {
    float min_depth = (min_depth_in > 0.0f) ? min_depth_in - 1e-4f : 1e7f;
    for (int o = 0; o < sc->n_objects; o++) {
        float t_near;
        if (!obj_aabb_test(sc, o, ray, start_p, &t_near)) continue;
        if (t_near > min_depth) continue;
        int start_id = sc->obj_info[o][0];
        if (sc->obj_info[o][2]) {
            float rt = sphere_t(sc, start_id, ray, start_p);
            if (rt > 1e-4f && rt < min_depth) return 1;
        } else {
            int tri_num = sc->obj_info[o][1];
            for (int m = start_id; m < tri_num + start_id; m++) {
                float u, v, t; tri_uvt(sc, m, ray, start_p, &u, &v, &t);
                if (u >= 0.f && v >= 0.f && u + v <= 1.0f)
                    if (t > 1e-4f && t < min_depth) return 1;
            }
        }
    }
    return 0;
}
/* LinearNode/LinearBVH.aabb_test, ti_bvh.py:17-24,38-45 */
static int slab_test(v3 mini, v3 maxi, v3 inv_ray, v3 ray_o, float* t_near) {
    v3 tmin = vmul(vsub(mini, ray_o), inv_ray);
    v3 tmax = vmul(vsub(maxi, ray_o), inv_ray);
    float tn = vmax(vminv(tmin, tmax));
    float tf = vmin(vmaxv(tmin, tmax));
    *t_near = tn;
    return (tn < tf) && tf > 0.f;
}
/* PathTracer.bvh_intersect, path_tracer.py:309-336 */
static float bvh_prim_t(const scene_t* sc, int bvh_id, v3 ray, v3 start_p, int* obj_idx, int* prim_idx, int* is_sphere, float* u, float* v) {
    *obj_idx = sc->bvhs[bvh_id].obj_idx; *prim_idx = sc->bvhs[bvh_id].prim_idx;
    *is_sphere = sc->obj_info[*obj_idx][2];
    float ray_t = -1.f; *u = 0.f; *v = 0.f;
    if (*is_sphere > 0) {
        ray_t = sphere_t(sc, *prim_idx, ray, start_p);
    } else {
        float t; tri_uvt(sc, *prim_idx, ray, start_p, u, v, &t);
        if (*u >= 0.f && *v >= 0.f && *u + *v <= 1.0f) ray_t = t;
    }
    return ray_t;
}
/* PathTracer.ray_intersect_bvh, path_tracer.py:338-394 */
static void ray_intersect_bvh(const scene_t* sc, v3 ray, v3 start_p, float min_depth_in, isect_t* it) {
    int obj_id = -1, prim_id = -1, sphere_flag = 0;
    float min_depth = (min_depth_in > 0.0f) ? min_depth_in - 1e-4f : 1e7f;
    int node_idx = 0;
    v3 inv_ray = V(1.f / ray.x, 1.f / ray.y, 1.f / ray.z);
    float cu = 0.f, cv = 0.f;
    while (node_idx < sc->node_num) {
        const lin_node_t* nd = &sc->nodes[node_idx];
        float t_near;
        int hit = slab_test(nd->mini, nd->maxi, inv_ray, start_p, &t_near);
        if (!hit || t_near > min_depth) { node_idx += nd->all_offset; continue; }
        if (nd->all_offset == 1) {
            for (int bi = nd->base; bi < nd->base + nd->prim_cnt; bi++) {
                const lin_bvh_t* lb = &sc->bvhs[bi];
                hit = slab_test(lb->mini, lb->maxi, inv_ray, start_p, &t_near);
                if (!hit || t_near > min_depth) continue;
                int oi, pi, sph; float u, v;
                float rt = bvh_prim_t(sc, bi, ray, start_p, &oi, &pi, &sph, &u, &v);
                if (rt > 1e-4f && rt < min_depth) { min_depth = rt; obj_id = oi; prim_id = pi; sphere_flag = sph; cu = u; cv = v; }
            }
        }
        node_idx += 1;
    }
    it->obj_id = obj_id; it->prim_id = prim_id; it->u = cu; it->v = cv; it->min_depth = min_depth;
    finish_isect(sc, it, sphere_flag, ray, start_p);
}
/* PathTracer.does_intersect_bvh, path_tracer.py:396-422 */
static int does_intersect_bvh(const scene_t* sc, v3 ray, v3 start_p, float min_depth_in) {
    float min_depth = (min_depth_in > 0.0f) ? min_depth_in - 1e-4f : 1e7f;
    int node_idx = 0;
    v3 inv_ray = V(1.f / ray.x, 1.f / ray.y, 1.f / ray.z);
    while (node_idx < sc->node_num) {
        const lin_node_t* nd = &sc->nodes[node_idx];
        float t_near;
        int hit = slab_test(nd->mini, nd->maxi, inv_ray, start_p, &t_near);
        if (!hit || t_near > min_depth) { node_idx += nd->all_offset; continue; }
        if (nd->all_offset == 1) {
            for (int bi = nd->base; bi < nd->base + nd->prim_cnt; bi++) {
                const lin_bvh_t* lb = &sc->bvhs[bi];
                hit = slab_test(lb->mini, lb->maxi, inv_ray, start_p, &t_near);
                if (!hit || t_near > min_depth) continue;
                int oi, pi, sph; float u, v;
                float rt = bvh_prim_t(sc, bi, ray, start_p, &oi, &pi, &sph, &u, &v);
                if (rt > 1e-4f && rt < min_depth) return 1;
            }
        }
        node_idx += 1;
    }
    return 0;
}
static void pt_ray_intersect(const ctx_t* c, v3 ray, v3 o, isect_t* it) {
    if (c->cfg->use_bvh && c->sc->node_num > 0) ray_intersect_bvh(c->sc, ray, o, -1.0f, it);
    else ray_intersect_brute(c->sc, ray, o, -1.0f, it);
}
static int pt_does_intersect(const ctx_t* c, v3 ray, v3 o, float dist) {
    if (c->cfg->use_bvh && c->sc->node_num > 0) return does_intersect_bvh(c->sc, ray, o, dist);
    return does_intersect_brute(c->sc, ray, o, dist);
}

/* -------------------------------- SAH BVH build: tracer/bvh/bvh.cpp:19-212 */
typedef struct { v3 mini, maxi; } aabb_t;
typedef struct { aabb_t bound; v3 centroid; int prim_idx, obj_idx; } bvh_info_t;
typedef struct bnode { int base, prim_num; aabb_t bound; struct bnode *l, *r; } bnode_t;
#define NUM_BINS 12
static const float TRAVERSE_COST = 0.1f;

static void aabb_clear(aabb_t* a) { a->mini = V(1e4f, 1e4f, 1e4f); a->maxi = V(-1e4f, -1e4f, -1e4f); }
static void aabb_grow(aabb_t* a, const aabb_t* b) { a->mini = vminv(b->mini, a->mini); a->maxi = vmaxv(b->maxi, a->maxi); }
static float aabb_area(const aabb_t* a) {
    v3 d = vsub(a->maxi, a->mini);
    return (float)(2. * (double)(d.x * d.y + d.y * d.z + d.x * d.z));   /* bvh_helper.h:61-64: `2. *` is double */
}
static inline float comp(v3 a, int ax) { return ax == 0 ? a.x : (ax == 1 ? a.y : a.z); }

static void prim_bound(const v3 p[3], int is_sphere, bvh_info_t* out) {     /* bvh_helper.h:30-45,75-85 */
    if (is_sphere) {
        out->bound.mini = vsub(p[0], p[1]); out->bound.maxi = vadd(p[0], p[1]);
        out->centroid = p[0];
    } else {
        v3 lo = vminv(vminv(p[0], p[1]), p[2]), hi = vmaxv(vmaxv(p[0], p[1]), p[2]);
        float* l = &lo.x; float* h = &hi.x;
        /* bvh_helper.h:38-42: `diff(i) < 1e-4` compares the float difference with a double literal, `mini(i) -= 1e-4` moves the bound in double */
        for (int i = 0; i < 3; i++) if ((double)(h[i] - l[i]) < 1e-4) { l[i] = (float)((double)l[i] - 1e-4); h[i] = (float)((double)h[i] + 1e-4); }
        out->bound.mini = lo; out->bound.maxi = hi;
        /* Eigen 3.4 rowwise().mean() = sum() / 3 (VectorwiseOp.h), and a fixed-size sum of three coefficients is unrolled as
         * x0 + (x1 + x2) (Redux.h, redux_novec_unroller: the range is halved recursively, 3 -> 1 + 2) */
        out->centroid = V((p[0].x + (p[1].x + p[2].x)) / 3.f, (p[0].y + (p[1].y + p[2].y)) / 3.f, (p[0].z + (p[1].z + p[2].z)) / 3.f);
    }
}
/* The reference is built against libstdc++ (tracer/setup.py: g++ -O3), and the standard leaves the element order that
 * std::partition / std::nth_element produce unspecified - but that order decides later splits wherever centroids tie, which is
 * everywhere in axis-aligned geometry (the Cornell box comes out with 61 nodes, not 57 as with a stable partition + sort).
 * So both are restated from libstdc++'s published algorithms (bits/stl_algo.h):
 *   std::partition, bidirectional iterators: Hoare scheme - advance `first` over elements satisfying the predicate, retreat `last`
 *     over elements that do not, swap, repeat;
 *   std::nth_element = introselect: while the range is longer than 3, median-of-three (first+1, middle, last-1) moved to the front,
 *     unguarded Hoare partition about it, continue in the half that holds nth; then insertion sort of the rest.  Ranges here
 *     have at most 4 elements (bvh.cpp:153-158): at most one partition round, so the depth limit never triggers. */
static void info_swap(bvh_info_t* a, bvh_info_t* b) { bvh_info_t t = *a; *a = *b; *b = t; }
static void stl_partition_lt(bvh_info_t* first, bvh_info_t* last, int axis, float pivot) {
    for (;;) {
        for (;;) { if (first == last) return; else if (comp(first->centroid, axis) < pivot) ++first; else break; }
        --last;
        for (;;) { if (first == last) return; else if (!(comp(last->centroid, axis) < pivot)) --last; else break; }
        info_swap(first, last);
        ++first;
    }
}
static void stl_insertion_sort(bvh_info_t* first, bvh_info_t* last, int axis) {
    if (first == last) return;
    for (bvh_info_t* i = first + 1; i != last; ++i) {
        bvh_info_t val = *i;
        bvh_info_t* j = i;
        while (j != first && comp(val.centroid, axis) < comp((j - 1)->centroid, axis)) { *j = *(j - 1); --j; }
        *j = val;
    }
}
static void stl_nth_element(bvh_info_t* first, bvh_info_t* nth, bvh_info_t* last, int axis) {
#define LT(a, b) (comp((a)->centroid, axis) < comp((b)->centroid, axis))
    while (last - first > 3) {
        bvh_info_t *mid = first + (last - first) / 2, *a = first + 1, *b = mid, *c = last - 1;
        if (LT(a, b)) { if (LT(b, c)) info_swap(first, b); else if (LT(a, c)) info_swap(first, c); else info_swap(first, a); }
        else if (LT(a, c)) info_swap(first, a);
        else if (LT(b, c)) info_swap(first, c);
        else info_swap(first, b);
        bvh_info_t *lo = first + 1, *hi = last;
        for (;;) {
            while (LT(lo, first)) ++lo;
            --hi;
            while (LT(first, hi)) --hi;
            if (!(lo < hi)) break;
            info_swap(lo, hi);
            ++lo;
        }
        if (lo <= nth) first = lo; else last = lo;
    }
#undef LT
    stl_insertion_sort(first, last, axis);
}
static int build_sah(bnode_t* cur, bvh_info_t* infos) {
    aabb_t fwd, bwd; aabb_clear(&fwd); aabb_clear(&bwd);
    int child_cnt = 0;
    const int prim_num = cur->prim_num, base = cur->base, max_pos = base + prim_num;
    float min_cost = 5e9f, node_prim_cnt = (float)prim_num;
    float node_inv_area = (float)(1. / (double)aabb_area(&cur->bound));
    /* max_extent_axis, bvh.cpp:19-41 */
    v3 min_c = infos[base].centroid, max_c = infos[base].centroid;
    for (int i = 1; i < prim_num; i++) { min_c = vminv(min_c, infos[base + i].centroid); max_c = vmaxv(max_c, infos[base + i].centroid); }
    v3 diff = vsub(max_c, min_c);
    float max_diff = diff.x; int axis = 0;
    if (diff.y > max_diff) { max_diff = diff.y; axis = 1; }
    if (diff.z > max_diff) { max_diff = diff.z; axis = 2; }
    float bins[NUM_BINS];
    float min_r = comp(min_c, axis) - 0.001f, interval = (max_diff + 0.002f) / (float)NUM_BINS;
    for (int i = 0; i < NUM_BINS; i++) bins[i] = min_r + interval * (float)(i + 1);
    if (prim_num > 4) {
        aabb_t bin_bound[NUM_BINS]; int bin_cnt[NUM_BINS];
        for (int i = 0; i < NUM_BINS; i++) { aabb_clear(&bin_bound[i]); bin_cnt[i] = 0; }
        for (int i = base; i < max_pos; i++) {
            float cv = comp(infos[i].centroid, axis);
            int idx = 0; while (idx < NUM_BINS && bins[idx] < cv) idx++;      /* std::lower_bound */
            if (idx >= NUM_BINS) idx = NUM_BINS - 1;                         /* cannot happen (bins padded) */
            aabb_grow(&bin_bound[idx], &infos[i].bound); bin_cnt[idx]++;
        }
        int prim_cnts[NUM_BINS]; float fwd_areas[NUM_BINS], bwd_areas[NUM_BINS];
        for (int i = 0; i < NUM_BINS; i++) {
            aabb_grow(&fwd, &bin_bound[i]);
            prim_cnts[i] = bin_cnt[i];
            fwd_areas[i] = aabb_area(&fwd);
            if (i > 0) { aabb_grow(&bwd, &bin_bound[NUM_BINS - i]); bwd_areas[NUM_BINS - 1 - i] = aabb_area(&bwd); }
        }
        for (int i = 1; i < NUM_BINS; i++) prim_cnts[i] += prim_cnts[i - 1];
        int seg = 0;
        for (int i = 0; i < NUM_BINS - 1; i++) {
            float cost = TRAVERSE_COST + node_inv_area * ((float)prim_cnts[i] * fwd_areas[i] + (node_prim_cnt - (float)prim_cnts[i]) * bwd_areas[i]);
            if (cost < min_cost) { min_cost = cost; seg = i; }
        }
        if (min_cost < node_prim_cnt) {
            stl_partition_lt(infos + base, infos + max_pos, axis, bins[seg]);      /* bvh.cpp:138-141 */
            child_cnt = prim_cnts[seg];
        }
        aabb_clear(&fwd); aabb_clear(&bwd);
        for (int i = 0; i <= seg; i++) aabb_grow(&fwd, &bin_bound[i]);
        for (int i = NUM_BINS - 1; i > seg; i--) aabb_grow(&bwd, &bin_bound[i]);
    } else {
        int seg_idx = (base + max_pos) >> 1;
        stl_nth_element(infos + base, infos + seg_idx, infos + max_pos, axis);       /* bvh.cpp:153-158 */
        for (int i = base; i < seg_idx; i++) aabb_grow(&fwd, &infos[i].bound);
        for (int i = seg_idx; i < max_pos; i++) aabb_grow(&bwd, &infos[i].bound);
        child_cnt = seg_idx - base;
        float split_cost = TRAVERSE_COST + node_inv_area * (aabb_area(&fwd) * (float)child_cnt + aabb_area(&bwd) * (node_prim_cnt - (float)child_cnt));
        if (split_cost >= node_prim_cnt) child_cnt = 0;
    }
    if (child_cnt > 0) {
        cur->l = (bnode_t*)calloc(1, sizeof(bnode_t)); cur->r = (bnode_t*)calloc(1, sizeof(bnode_t));
        cur->l->base = base; cur->l->prim_num = child_cnt; cur->l->bound = fwd;
        cur->r->base = base + child_cnt; cur->r->prim_num = prim_num - child_cnt; cur->r->bound = bwd;
        int n = 1;
        n += (cur->l->prim_num > 1) ? build_sah(cur->l, infos) : 1;
        n += (cur->r->prim_num > 1) ? build_sah(cur->r, infos) : 1;
        return n;
    }
    return 1;
}
static int linearize(bnode_t* cur, lin_node_t* out, int* n) {       /* bvh.cpp:195-212 */
    int me = (*n)++;
    out[me].mini = cur->bound.mini; out[me].maxi = cur->bound.maxi;
    out[me].base = cur->base; out[me].prim_cnt = cur->prim_num;
    if (cur->l) {
        int cnt = linearize(cur->l, out, n);
        cnt += linearize(cur->r, out, n);
        out[me].all_offset = cnt + 1;
        return cnt + 1;
    }
    out[me].all_offset = 1;
    return 1;
}
static void free_tree(bnode_t* n) { if (!n) return; free_tree(n->l); free_tree(n->r); free(n); }

static void build_reference_bvh(scene_t* sc, v3 wmin, v3 wmax) {
    int N = sc->n_prims;
    bvh_info_t* infos = (bvh_info_t*)malloc(sizeof(bvh_info_t) * (size_t)N);
    for (int o = 0; o < sc->n_objects; o++) {
        int start = sc->obj_info[o][0], cnt = sc->obj_info[o][1], sph = sc->obj_info[o][2];
        for (int p = start; p < start + cnt; p++) { prim_bound(sc->prims[p], sph > 0, &infos[p]); infos[p].prim_idx = p; infos[p].obj_idx = o; }
    }
    bnode_t* root = (bnode_t*)calloc(1, sizeof(bnode_t));
    root->base = 0; root->prim_num = N; root->bound.mini = wmin; root->bound.maxi = wmax;
    int node_num = build_sah(root, infos);
    sc->nodes = (lin_node_t*)malloc(sizeof(lin_node_t) * (size_t)node_num);
    int n = 0; linearize(root, sc->nodes, &n);
    sc->node_num = n;
    sc->bvhs = (lin_bvh_t*)malloc(sizeof(lin_bvh_t) * (size_t)N);
    for (int i = 0; i < N; i++) { sc->bvhs[i].mini = infos[i].bound.mini; sc->bvhs[i].maxi = infos[i].bound.maxi; sc->bvhs[i].obj_idx = infos[i].obj_idx; sc->bvhs[i].prim_idx = infos[i].prim_idx; }
    sc->bvh_num = N;
    free(infos); free_tree(root);
}

/* ----------------------------------------------------- scene construction */
static const v3* as_v3(const float* p) { return (const v3*)p; }
#define LD3(p) V((p)[0], (p)[1], (p)[2])

ORC_API scene_t* orc_scene_create(const orc_scene_desc* d, const float cam_t[3], int build_bvh) {
    scene_t* sc = (scene_t*)calloc(1, sizeof(scene_t));
    int N = d->n_prims, O = d->n_objects, S = d->n_sources;
    sc->n_prims = N; sc->n_objects = O; sc->n_sources = S; sc->has_vn = d->has_vertex_normal;
    sc->world_ior = d->world_ior;
    sc->prims = malloc(sizeof(v3) * 3 * (size_t)N); sc->precom = malloc(sizeof(v3) * 3 * (size_t)N);
    sc->vnorm = malloc(sizeof(v3) * 3 * (size_t)N); sc->normals = malloc(sizeof(v3) * (size_t)N);
    memcpy(sc->prims, d->prims, sizeof(v3) * 3 * (size_t)N);
    memcpy(sc->normals, d->normals, sizeof(v3) * (size_t)N);
    if (d->v_normals) memcpy(sc->vnorm, d->v_normals, sizeof(v3) * 3 * (size_t)N); else memset(sc->vnorm, 0, sizeof(v3) * 3 * (size_t)N);
    sc->obj_info = malloc(sizeof(int) * 3 * (size_t)O); memcpy(sc->obj_info, d->obj_info, sizeof(int) * 3 * (size_t)O);
    sc->aabbs = malloc(sizeof(v3) * 2 * (size_t)O); memcpy(sc->aabbs, d->obj_aabb, sizeof(v3) * 2 * (size_t)O);
    sc->emitter_id = malloc(sizeof(int) * (size_t)O); memcpy(sc->emitter_id, d->emitter_id, sizeof(int) * (size_t)O);
    /* load_primitives, tracer_base.py:117-134: precom = (v1-v0, v2-v0, v0); sphere rows keep (centre, rrr) */
    for (int p = 0; p < N; p++) {
        sc->precom[p][0] = vsub(sc->prims[p][1], sc->prims[p][0]);
        sc->precom[p][1] = vsub(sc->prims[p][2], sc->prims[p][0]);
        sc->precom[p][2] = sc->prims[p][0];
    }
    for (int o = 0; o < O; o++) if (sc->obj_info[o][2]) {
        int p = sc->obj_info[o][0];
        sc->precom[p][0] = sc->prims[p][0]; sc->precom[p][1] = sc->prims[p][1];
    }
    if (d->tex_i && d->uvs) {
        sc->uvs = malloc(sizeof(float) * 6 * (size_t)N); memcpy(sc->uvs, d->uvs, sizeof(float) * 6 * (size_t)N);
        sc->tex_i = malloc(sizeof(int) * 15 * (size_t)O); memcpy(sc->tex_i, d->tex_i, sizeof(int) * 15 * (size_t)O);
        sc->tex_f = malloc(sizeof(float) * 6 * (size_t)O); memcpy(sc->tex_f, d->tex_f, sizeof(float) * 6 * (size_t)O);
        for (int m = 0; m < 3; m++) if (d->atlas[m]) {
            size_t n = (size_t)d->atlas_w[m] * (size_t)d->atlas_h[m] * 3;
            sc->atlas[m] = malloc(sizeof(float) * n); memcpy(sc->atlas[m], d->atlas[m], sizeof(float) * n);
            sc->atlas_w[m] = d->atlas_w[m]; sc->atlas_h[m] = d->atlas_h[m];
        }
    }
    sc->bxdf = calloc((size_t)O, sizeof(bxdf_t));
    for (int o = 0; o < O; o++) {
        const int* bi = d->bxdf_i + 4 * o; const float* bf = d->bxdf_f + 13 * o;
        bxdf_t* b = &sc->bxdf[o];
        b->type = bi[0]; b->is_delta = bi[1]; b->is_bsdf = bi[2];
        b->k_d = as_v3(bf)[0]; b->k_s = as_v3(bf)[1]; b->k_g = as_v3(bf)[2]; b->mean = as_v3(bf)[3]; b->ior = bf[12];
    }
    sc->src = calloc((size_t)(S > 0 ? S : 1), sizeof(src_t));
    for (int s = 0; s < S; s++) {
        const int* si = d->src_i + 4 * s; const float* sf = d->src_f + 11 * s;
        src_t* e = &sc->src[s];
        e->type = si[0]; e->bool_bits = si[1]; e->obj_ref_id = si[2];
        e->intensity = as_v3(sf)[0]; e->dir = as_v3(sf)[1]; e->pos = as_v3(sf)[2]; e->inv_area = sf[9]; e->r = sf[10];
    }
    sc->med = calloc((size_t)O + 1, sizeof(medium_t));
    for (int o = 0; o <= O; o++) {
        medium_t* m = &sc->med[o];
        if (d->med_i && d->med_f) {
            const float* f = d->med_f + 16 * o;
            m->type = d->med_i[o]; m->ior = f[0];
            m->u_s = as_v3(f + 1)[0]; m->u_a = as_v3(f + 1)[1]; m->u_e = as_v3(f + 1)[2]; m->par = as_v3(f + 1)[3]; m->pdf = as_v3(f + 1)[4];
        } else {
            m->type = -1; m->ior = (o < O) ? sc->bxdf[o].ior : sc->world_ior; m->pdf = V(1.f, 0.f, 0.f);
        }
    }
    if (d->vol_i && d->vol_f && d->vol_grid && d->vol_i[0] > 0) {
        const float* f = d->vol_f;
        sc->vol_type = d->vol_i[0]; sc->vol_res[0] = d->vol_i[1]; sc->vol_res[1] = d->vol_i[2]; sc->vol_res[2] = d->vol_i[3];
        sc->vol_albedo = LD3(f);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) sc->vol_inv_T.m[i][j] = f[3 + 3 * i + j];
        sc->vol_trans = LD3(f + 12); sc->vol_mini = LD3(f + 15); sc->vol_maxi = LD3(f + 18);
        sc->vol_majorant = LD3(f + 21); sc->vol_pdf = LD3(f + 24);
        memset(&sc->vol_ph, 0, sizeof(sc->vol_ph));
        sc->vol_ph.type = d->vol_i[4]; sc->vol_ph.par = LD3(f + 27); sc->vol_ph.pdf = LD3(f + 30);
        size_t n = (size_t)d->vol_i[1] * (size_t)d->vol_i[2] * (size_t)d->vol_i[3] * 3;
        sc->vol_grid = malloc(sizeof(float) * n); memcpy(sc->vol_grid, d->vol_grid, sizeof(float) * n);
    }
    {   /* world AABB, path_tracer.py:130-138 */
        v3 lo = V(1e3f, 1e3f, 1e3f), hi = V(-1e3f, -1e3f, -1e3f);
        for (int o = 0; o < O; o++) { lo = vminv(lo, sc->aabbs[o][0]); hi = vmaxv(hi, sc->aabbs[o][1]); }
        v3 ct = V(cam_t[0], cam_t[1], cam_t[2]);
        sc->w_aabb_min = vadds(vminv(ct, lo), -0.1f); sc->w_aabb_max = vadds(vmaxv(ct, hi), 0.1f);
        if (build_bvh) build_reference_bvh(sc, sc->w_aabb_min, sc->w_aabb_max);
    }
    return sc;
}
ORC_API void orc_scene_destroy(scene_t* sc) {
    if (!sc) return;
    free(sc->prims); free(sc->precom); free(sc->vnorm); free(sc->normals); free(sc->obj_info); free(sc->aabbs);
    free(sc->emitter_id); free(sc->bxdf); free(sc->src); free(sc->med); free(sc->vol_grid); free(sc->nodes); free(sc->bvhs);
    free(sc->uvs); free(sc->tex_i); free(sc->tex_f); for (int m = 0; m < 3; m++) free(sc->atlas[m]);
    free(sc);
}
static v3 texture_query(const scene_t* sc, int map, int obj, float u, float v);
/* Texture.query on explicit coordinates (test entry): map 0 albedo, 1 normal, 2 bump */
ORC_API int orc_texture_query(const scene_t* sc, int n, const int* map_obj, const float* uv, float* out3) {
    for (int k = 0; k < n; k++) {
        int m = map_obj[2 * k], o = map_obj[2 * k + 1];
        if (!sc->tex_i || m < 0 || m > 2 || !sc->atlas[m] || o < 0 || o >= sc->n_objects || !(sc->tex_i[o][m][0] > -255)) return -1;
        v3 r = texture_query(sc, m, o, uv[2 * k], uv[2 * k + 1]);
        out3[3 * k] = r.x; out3[3 * k + 1] = r.y; out3[3 * k + 2] = r.z;
    }
    return 0;
}
ORC_API int orc_bvh_counts(const scene_t* sc, int* node_num, int* bvh_num) { *node_num = sc->node_num; *bvh_num = sc->bvh_num; return 0; }
/* node_minmax[M*6], node_info[M*3] (base,cnt,all_offset), bvh_minmax[N*6], bvh_info[N*2] (obj,prim): bvh.cpp:215-251 */
ORC_API int orc_bvh_export(const scene_t* sc, float* node_minmax, int* node_info, float* bvh_minmax, int* bvh_info) {
    for (int i = 0; i < sc->node_num; i++) {
        memcpy(node_minmax + 6 * i, &sc->nodes[i].mini, 12); memcpy(node_minmax + 6 * i + 3, &sc->nodes[i].maxi, 12);
        node_info[3 * i] = sc->nodes[i].base; node_info[3 * i + 1] = sc->nodes[i].prim_cnt; node_info[3 * i + 2] = sc->nodes[i].all_offset;
    }
    for (int i = 0; i < sc->bvh_num; i++) {
        memcpy(bvh_minmax + 6 * i, &sc->bvhs[i].mini, 12); memcpy(bvh_minmax + 6 * i + 3, &sc->bvhs[i].maxi, 12);
        bvh_info[2 * i] = sc->bvhs[i].obj_idx; bvh_info[2 * i + 1] = sc->bvhs[i].prim_idx;
    }
    return 0;
}

/* The restated builder on bare arrays, with bvh_cpp.bvh_build's own signature (bvh.cpp:274-296): prims[N*9], the (2, n_obj) table of
 * primitive counts and sphere flags, the world box.  Two calls: with the output pointers NULL it only returns the node count.
 * Used by the golden generator's `bvh_cpp` stand-in so that the reference's own traversal code walks this tree. */
ORC_API int orc_bvh_build_raw(const float* prims, int n_prims, const int* obj_cnt, const int* obj_flag, int n_obj, const float wmin[3], const float wmax[3],
                              float* bvh_minmax, float* node_minmax, int* bvh_info, int* node_info) {
    scene_t tmp; memset(&tmp, 0, sizeof(tmp));
    tmp.n_prims = n_prims; tmp.n_objects = n_obj;
    tmp.prims = (v3(*)[3])prims;
    tmp.obj_info = malloc(sizeof(int) * 3 * (size_t)n_obj);
    int start = 0;
    for (int o = 0; o < n_obj; o++) { tmp.obj_info[o][0] = start; tmp.obj_info[o][1] = obj_cnt[o]; tmp.obj_info[o][2] = obj_flag[o]; start += obj_cnt[o]; }
    if (start != n_prims) { free(tmp.obj_info); return -1; }
    build_reference_bvh(&tmp, V(wmin[0], wmin[1], wmin[2]), V(wmax[0], wmax[1], wmax[2]));
    const int n = tmp.node_num;
    if (bvh_minmax && node_minmax && bvh_info && node_info) orc_bvh_export(&tmp, node_minmax, node_info, bvh_minmax, bvh_info);
    free(tmp.nodes); free(tmp.bvhs); free(tmp.obj_info);
    return n;
}

/* ------------------------------------- TracerBase.pix2ray, tracer_base.py:136-157 */
static v3 pix2ray(const ctx_t* c, int i, int j, int cnt, rng_t* r) {
    const orc_cfg* g = c->cfg;
    float pi = (float)i, pj = (float)j, vx = 0.5f, vy = 0.5f;
    if (g->anti_alias) {
        if (g->stratified) {
            int mod_val = pymod(cnt, 16);
            vx = (float)(mod_val % 4) * 0.25f + rng_float(r) * 0.25f;
            vy = (float)(mod_val / 4) * 0.25f + rng_float(r) * 0.25f;
        } else {
            const float eps = 1e-4f, inv_eps = (float)(1 - 1e-4 * 2.);
            vx = rng_float(r) * inv_eps + eps;
            vy = rng_float(r) * inv_eps + eps;
        }
    }
    v3 cam_dir = V((g->half_w + vx - pi) * g->inv_focal, (pj - g->half_h - vy) * g->inv_focal, 1.f);
    return vnormalized(m3mulv(&c->cam_r, cam_dir));
}

/* One pixel-sample of Renderer.render, vanilla_renderer.py:39-119.  Returns the
 * sample's colour with NaN components zeroed (line 119). */
typedef struct {
    int max_events; int n_events;
    float* ev;       /* per bounce: [obj_id, prim_id, min_depth, direct_int xyz, emit*w xyz, contribution xyz, next ray o xyz, d xyz] = 18 floats */
} trace_t;

/* Taichi's float `a % b` is a - b * floor(a / b) (python/taichi/lang/ops.py, mod) */
static inline float ti_fmod(float a, float b) { float q = floorf(a / b); return a - b * q; }
static inline v3 vmix(v3 a, v3 b, float t) { return vadd(vscale(a, 1.0f - t), vscale(b, t)); }       /* taichi.math.mix: x * (1 - a) + y * a */
/* Texture.query, bxdf/texture.py:111-139: bilinear lookup inside the texture's rectangle of the atlas */
static v3 texture_query(const scene_t* sc, int map, int obj, float u, float v) {
    const int* ti_ = sc->tex_i[obj][map]; const float* tf = sc->tex_f[obj][map];
    float w = (float)ti_[3], h = (float)ti_[4];
    float scaled_u = ti_fmod((u * tf[0]) * w, w - 1.f);
    float scaled_v = ti_fmod((v * tf[1]) * h, h - 1.f);
    float floor_u = floorf(scaled_u), floor_v = floorf(scaled_v);
    float ratio_u = scaled_u - floor_u, ratio_v = scaled_v - floor_v;
    floor_u = floor_u + (float)ti_[1]; floor_v = floor_v + (float)ti_[2];
    int fu = (int)floor_u, fv = (int)floor_v, cu = fu + 1, cv = fv + 1;
    const float* img = sc->atlas[map]; int W = sc->atlas_w[map];
#define TEXEL(y, x) V(img[((size_t)(y) * W + (x)) * 3], img[((size_t)(y) * W + (x)) * 3 + 1], img[((size_t)(y) * W + (x)) * 3 + 2])
    v3 q_ff = TEXEL(fv, fu), q_cf = TEXEL(fv, cu), q_fc = TEXEL(cv, fu), q_cc = TEXEL(cv, cu);
#undef TEXEL
    return vmix(vmix(q_ff, q_cf, ratio_u), vmix(q_fc, q_cc, ratio_u), ratio_v);
}
/* PathTracer.get_uv_item, path_tracer.py:276-289 */
static int get_uv_item(const scene_t* sc, int map, const isect_t* it, v3* out) {
    *out = V(-1.f, -1.f, -1.f);
    if (!sc->tex_i || !sc->atlas[map] || it->obj_id < 0 || !(sc->tex_i[it->obj_id][map][0] > -255)) return 0;
    float u = it->u, v = it->v;
    if (sc->obj_info[it->obj_id][2] == 0) {
        const float (*uv)[2] = sc->uvs[it->prim_id];
        float w0 = 1.f - it->u - it->v;
        float gu = (uv[1][0] * it->u + uv[2][0] * it->v) + uv[0][0] * w0;
        float gv = (uv[1][1] * it->u + uv[2][1] * it->v) + uv[0][1] * w0;
        u = gu; v = gv;
    }
    *out = texture_query(sc, map, it->obj_id, u, v);
    return 1;
}
/* PathTracer.process_ns, path_tracer.py:291-307: applied to the camera ray's hit only (vanilla_renderer.py:42) */
static void process_ns(const scene_t* sc, isect_t* it) {
    v3 t;
    if (get_uv_item(sc, 1, it, &t)) { m3 R; rotation_between(V(0.f, 1.f, 0.f), it->n_g, &R); it->n_s = m3mulv(&R, t); }
    if (get_uv_item(sc, 2, it, &t)) { m3 R; it->n_s = delocalize_rotate(it->n_s, t, &R); }
}

static v3 render_sample(const ctx_t* c, int i, int j, int cnt, rng_t* rng, orc_stats* st, trace_t* tr) {
    const orc_cfg* g = c->cfg; const scene_t* sc = c->sc;
    v3 ray_d = pix2ray(c, i, j, cnt, rng);
    v3 ray_o = c->cam_t;
    isect_t it; pt_ray_intersect(c, ray_d, ray_o, &it);
    process_ns(sc, &it);
    int hit_light = sc->emitter_id[it.obj_id > 0 ? it.obj_id : 0];
    v3 color = ZERO3, contribution = V(1.f, 1.f, 1.f);
    float emission_weight = 1.0f;
    st->n_samples++;
    for (int bounce = 0; bounce < g->max_bounce; bounce++) {
        if (it.obj_id < 0) break;
        if (g->use_rr) {
            float max_value = vmax(contribution);
            if (max_value < g->rr_threshold && bounce >= g->rr_bounce_th) {
                if (rng_float(rng) > max_value) break;
                else contribution = vscale(contribution, 1.f / (max_value + 1e-7f));
            }
        } else {
            if (vmax(contribution) < 1e-4f) break;
        }
        st->n_shade++;
        v3 hit_point = vadd(vscale(ray_d, it.min_depth), ray_o);
        float direct_pdf = 1.0f, emitter_pdf = 1.0f;
        int break_flag = 0;
        v3 shadow_int = ZERO3, direct_int = ZERO3, direct_spec = V(1.f, 1.f, 1.f);
        get_uv_item(sc, 0, &it, &it.tex);          /* vanilla_renderer.py:66 */
        for (int s = 0; s < g->num_shadow_ray; s++) {
            int emitter_valid;
            const src_t* emitter = pt_sample_light(c, hit_light, rng, &emitter_pdf, &emitter_valid);
            v3 light_dir = ZERO3;
            if (emitter_valid) {
                v3 emit_pos = src_sample_hit(c, emitter, hit_point, rng, &shadow_int, &direct_pdf);
                v3 to_emitter = vsub(emit_pos, hit_point);
                float emitter_d = vnorm(to_emitter);
                light_dir = vdivs(to_emitter, emitter_d);
                st->n_shadow++;
                if (pt_does_intersect(c, light_dir, hit_point, emitter_d)) shadow_int = ZERO3;
                else { direct_spec = pt_eval(c, &it, ray_d, light_dir); st->n_lit++; }
            } else { break_flag = 1; break; }
            float light_pdf = emitter_pdf * direct_pdf;
            if (g->use_mis) {
                float mis_w = 1.0f;
                if (!(emitter->bool_bits & 0x01)) {
                    float bsdf_pdf_v = pt_surface_pdf(c, &it, light_dir, ray_d);
                    mis_w = balance_heuristic(light_pdf, bsdf_pdf_v);
                }
                direct_int = vadd(direct_int, vdivs(vscale(vmul(direct_spec, shadow_int), mis_w), emitter_pdf));
            } else {
                direct_int = vadd(direct_int, vdivs(vmul(direct_spec, shadow_int), emitter_pdf));
            }
        }
        if (!break_flag) direct_int = vscale(direct_int, c->inv_num_shadow_ray);
        v3 emit_int = ZERO3;
        if (hit_light >= 0) emit_int = src_eval_le(&sc->src[hit_light], vsub(hit_point, ray_o), it.n_s);
        v3 indirect_spec; float ray_pdf; int is_specular;
        v3 new_d = pt_sample_new_ray(c, &it, ray_d, rng, &indirect_spec, &ray_pdf, &is_specular);
        if (tr && tr->n_events < tr->max_events) {
            float* e = tr->ev + 18 * tr->n_events++;
            v3 ew = vscale(emit_int, emission_weight);
            e[0] = (float)it.obj_id; e[1] = (float)it.prim_id; e[2] = it.min_depth;
            e[3] = direct_int.x; e[4] = direct_int.y; e[5] = direct_int.z;
            e[6] = ew.x; e[7] = ew.y; e[8] = ew.z;
            e[9] = contribution.x; e[10] = contribution.y; e[11] = contribution.z;
            e[12] = hit_point.x; e[13] = hit_point.y; e[14] = hit_point.z; e[15] = new_d.x; e[16] = new_d.y; e[17] = new_d.z;
        }
        ray_d = new_d;
        ray_o = hit_point;
        color = vadd(color, vmul(vadd(direct_int, vscale(emit_int, emission_weight)), contribution));
        contribution = vmul(contribution, vdivs(indirect_spec, ray_pdf));
        pt_ray_intersect(c, ray_d, ray_o, &it);
        if (it.obj_id >= 0) {
            hit_light = sc->emitter_id[it.obj_id];
            if (g->use_mis) {
                float e_pdf = 0.0f;
                if (hit_light >= 0 && pt_is_delta(c, it.obj_id) == 0 && !is_specular)
                    e_pdf = src_solid_angle_pdf(&sc->src[hit_light], &it, ray_d);
                emission_weight = balance_heuristic(ray_pdf, e_pdf);
            }
        }
    }
    st->n_draws += rng->draw;
    if (isnan(color.x)) color.x = 0.f;
    if (isnan(color.y)) color.y = 0.f;
    if (isnan(color.z)) color.z = 0.f;
    return color;
}

/* ===================================================================== volumetric path tracer
 * VolumeRenderer.render (renderer/vpt.py:145-258) with homogeneous media: the world medium and media attached to BSDF objects
 * (bxdf/medium.py:71-125), their phase functions (bxdf/phase.py, sampler/phase_sampling.py), null surfaces (bsdf.py:214-216) and
 * the transmittance walk track_ray (vpt.py:99-138).  Grid volumes (bxdf/volume.py) follow further down ("grid volume" section). */
static float random_rgb(rng_t* r, v3 v) {                          /* general_sampling.py:17-27 */
    int idx = pymod(rng_int(r), 3);
    float res = (idx == 0) ? v.x : ((idx == 1) ? v.y : v.z);
    return fmaxf(res, 1e-5f);
}
static inline v3 vexp_neg(v3 u_e, float d) { return V(expf(-u_e.x * d), expf(-u_e.y * d), expf(-u_e.z * d)); }
static inline float vsum(v3 a) { return (a.x + a.y) + a.z; }
static v3 medium_transmittance(const medium_t* m, float depth) { return vexp_neg(m->u_e, depth); }     /* medium.py:84-87 */
/* Medium.sample_mfp, medium.py:89-108 */
static int medium_sample_mfp(const medium_t* m, float max_depth, rng_t* r, float* t_out, v3* beta) {
    float random_ue = random_rgb(r, m->u_e);
    float sample_t = -logf(1.f - rng_float(r)) / random_ue;
    int is_mi = 0;
    if (sample_t >= max_depth) {
        sample_t = max_depth;
        v3 tr = vexp_neg(m->u_e, max_depth);
        float pdf = vsum(tr) / 3.f;
        pdf = (pdf > 0.f) ? pdf : 1.f;
        *beta = vdivs(tr, pdf);
    } else {
        is_mi = 1;
        v3 tr = vexp_neg(m->u_e, sample_t);
        float pdf = vsum(vmul(m->u_e, tr)) / 3.f;
        pdf = (pdf > 0.f) ? pdf : 1.f;
        *beta = vdivs(vmul(tr, m->u_s), pdf);
    }
    *t_out = sample_t;
    return is_mi;
}
/* bxdf/phase.py:21-31 */
static float phase_hg(float cos_theta, float g) {
    float g2 = g * g;
    float denom = (1.f + g2) - (2.f * g) * cos_theta;
    return (((1.f - g2) / (sqrtf(denom) * denom)) * 0.5f) * F_INV_2PI;
}
static float phase_rayleigh(float cos_theta) { return (float)(0.375 * ((1.0 / 3.14159265358979323846) * 0.5)) * (1.f + cos_theta * cos_theta); }
/* sampler/phase_sampling.py:16-42 */
static v3 sample_hg(rng_t* r, float g, float* cos_out) {
    float cos_theta;
    if (fabsf(g) < 1e-4f) cos_theta = 1.f - 2.f * rng_float(r);
    else {
        float g2 = g * g;
        float sqr_term = (1.f - g2) / ((1.f + g) - (2.f * g) * rng_float(r));
        cos_theta = ((1.f + g2) - sqr_term * sqr_term) / (2.f * g);
    }
    float sin_theta = sqrtf(fmaxf(0.f, 1.f - cos_theta * cos_theta));
    float phi = F_PI2 * rng_float(r);
    *cos_out = cos_theta;
    return V(cosf(phi) * sin_theta, cos_theta, sinf(phi) * sin_theta);
}
static v3 sample_rayleigh(rng_t* r, float* cos_out) {
    float rd = 2.f * rng_float(r) - 1.f;
    float u = -powf(2.f * rd + sqrtf((4.f * rd) * rd + 1.f), (float)(1.0 / 3.0));
    float cos_theta = fminf(fmaxf(u - 1.f / u, -1.f), 1.f);
    float sin_theta = sqrtf(fmaxf(0.f, 1.f - cos_theta * cos_theta));
    float phi = F_PI2 * rng_float(r);
    *cos_out = cos_theta;
    return V(cosf(phi) * sin_theta, cos_theta, sinf(phi) * sin_theta);
}
/* PhaseFunction.sample_p / eval_p, phase.py:39-84 */
static v3 phase_sample_p(const medium_t* m, v3 incid, rng_t* r, float* p_out) {
    v3 dir = incid; float p = 1.f, cos_t = 0.f;
    if (m->type == 0) { float g = m->par.x; dir = sample_hg(r, g, &cos_t); p = phase_hg(cos_t, g); }
    else if (m->type == 1) {
        float eps = rng_float(r), g;
        if (eps < m->pdf.x) g = m->par.x;
        else if (eps < m->pdf.x + m->pdf.y) g = m->par.y;
        else g = m->par.z;
        dir = sample_hg(r, g, &cos_t); p = phase_hg(cos_t, g);
    } else if (m->type == 2) { dir = sample_rayleigh(r, &cos_t); p = phase_rayleigh(cos_t); }
    *p_out = p;
    return dir;
}
static float phase_eval_p(const medium_t* m, v3 ray_in, v3 ray_out) {
    float p = 1.f, cos_theta = -vdot(ray_in, ray_out);
    if (m->type == 0) p = phase_hg(cos_theta, m->par.x);
    else if (m->type == 1) {
        p = phase_hg(cos_theta, m->par.x) * m->pdf.x + phase_hg(cos_theta, m->par.y) * m->pdf.y;
        if (m->pdf.y > 1e-4f) p += phase_hg(cos_theta, m->par.z) * m->pdf.z;
    } else if (m->type == 2) p = phase_rayleigh(cos_theta);
    return p;
}
/* Medium.sample_new_rays, medium.py:112-121 */
static v3 medium_sample_new_rays(const medium_t* m, v3 incid, rng_t* r, v3* spec, float* pdf) {
    *spec = V(1.f, 1.f, 1.f); *pdf = 1.f;
    if (m->type < 0) return incid;
    float p; v3 local = phase_sample_p(m, incid, r, &p);
    m3 R; v3 dir = delocalize_rotate(incid, local, &R);
    *pdf = p; *spec = V(p, p, p);
    return dir;
}
static inline int vpt_world_scattering(const ctx_t* c) { return c->sc->med[c->sc->n_objects].type >= 0; }
static inline int vpt_is_scattering(const ctx_t* c, int idx) {                 /* path_tracer.py:528-535 */
    return idx >= 0 && c->sc->bxdf[idx].is_bsdf && c->sc->med[idx].type >= 0;
}
static inline int vpt_non_null_surface(const ctx_t* c, int idx) {             /* vpt.py:64-70 */
    return !(idx >= 0 && c->sc->bxdf[idx].is_bsdf) || c->sc->bxdf[idx].type >= 0;
}
static v3 vpt_get_transmittance(const ctx_t* c, int idx, int in_free_space, float depth) {     /* vpt.py:52-62 */
    v3 tr = V(1.f, 1.f, 1.f);
    int world_valid = in_free_space && vpt_world_scattering(c);
    if (world_valid || vpt_is_scattering(c, idx)) {
        if (world_valid) tr = medium_transmittance(&c->sc->med[c->sc->n_objects], depth);
        else if (!in_free_space) tr = medium_transmittance(&c->sc->med[idx], depth);
    }
    return tr;
}
/* ------------------------------------------------------------------ grid volume, bxdf/volume.py:248-463 (RGB volumes) */
static float vmax_np(v3 a) { if (isnan(a.x) || isnan(a.y) || isnan(a.z)) return NAN; return vmax(a); }      /* Vector.max() / .min(): NaN-propagating */
static float vmin_np(v3 a) { if (isnan(a.x) || isnan(a.y) || isnan(a.z)) return NAN; return vmin(a); }
static int vol_intersect(const scene_t* sc, v3 o, v3 d, float max_t, float* near_t, float* far_t) {             /* volume.py:271-285 */
    v3 inv_dir = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    v3 t1s = vmul(vsub(sc->vol_mini, o), inv_dir), t2s = vmul(vsub(sc->vol_maxi, o), inv_dir);
    v3 tmin = vminv(t1s, t2s), tmax = vmaxv(t1s, t2s);
    *near_t = fmaxf(0.f, vmax_np(tmin)) + 1e-5f;
    *far_t = fminf(max_t, vmin_np(tmax)) - 1e-5f;
    return *near_t < *far_t && *far_t > 0.f;
}
static v3 vol_density_lookup(const scene_t* sc, v3 index, v3 u_offset) {                                       /* volume.py:307-314 */
    v3 f = V(floorf(index.x + (u_offset.x - 0.5f)), floorf(index.y + (u_offset.y - 0.5f)), floorf(index.z + (u_offset.z - 0.5f)));
    int ix = (int)f.x, iy = (int)f.y, iz = (int)f.z;
    if (ix >= 0 && iy >= 0 && iz >= 0 && ix <= sc->vol_res[0] - 1 && iy <= sc->vol_res[1] - 1 && iz <= sc->vol_res[2] - 1) {
        const float* p = sc->vol_grid + 3 * (((size_t)iz * sc->vol_res[1] + iy) * sc->vol_res[0] + ix);
        return V(p[0], p[1], p[2]);
    }
    return ZERO3;
}
static inline float rgb_select(v3 a, int ch) { return ch == 0 ? a.x : (ch == 1 ? a.y : a.z); }
/* wavelength channel by throughput x majorant pdf (volume.py:352-377, 410-432) */
static int vol_pick_channel(const scene_t* sc, v3 thp, rng_t* r, float* pdf) {
    v3 pdfs = vmul(thp, sc->vol_pdf);
    pdfs = vdivs(pdfs, vsum(pdfs));
    float val = rng_float(r);
    if (val <= pdfs.x) { *pdf = pdfs.x; return 0; }
    if (val <= pdfs.x + pdfs.y) { *pdf = pdfs.y; return 1; }
    *pdf = pdfs.z; return 2;
}
static inline v3 vol_channel_vec(int ch, float v) { return ch == 0 ? V(v, 0.f, 0.f) : (ch == 1 ? V(0.f, v, 0.f) : V(0.f, 0.f, v)); }
/* GridVolume.sample_mfp -> delta tracking (volume.py:295-305, 346-397): returns the collision distance or -1, beta in *out */
static float vol_sample_mfp(const scene_t* sc, v3 ray_o, v3 ray_d, v3 thp, float max_t, rng_t* r, v3* out) {
    *out = V(1.f, 1.f, 1.f);
    float near_t, far_t;
    if (!sc->vol_type || !vol_intersect(sc, ray_o, ray_d, max_t, &near_t, &far_t)) return -1.f;
    v3 ol = m3mulv(&sc->vol_inv_T, vsub(ray_o, sc->vol_trans)), dl = m3mulv(&sc->vol_inv_T, ray_d);
    float pdf; int ch = vol_pick_channel(sc, thp, r, &pdf);
    float albedo = rgb_select(sc->vol_albedo, ch), inv_maj = 1.0f / rgb_select(sc->vol_majorant, ch);
    float Tr = 1.0f, hit_t = -1.f;
    float t = near_t - logf(1.0f - rng_float(r)) * inv_maj;
    while (t < far_t) {
        float u0 = rng_float(r), u1 = rng_float(r), u2 = rng_float(r);
        v3 dns = vol_density_lookup(sc, vadd(ol, vscale(dl, t)), V(u0, u1, u2));
        float n_t = rgb_select(dns, ch);
        if (rng_float(r) < n_t * inv_maj) { Tr *= albedo; hit_t = t; break; }
        t -= logf(1.0f - rng_float(r)) * inv_maj;
    }
    *out = (sc->vol_type == 2) ? vol_channel_vec(ch, Tr / pdf) : V(Tr, Tr, Tr);
    return hit_t;
}
/* GridVolume.transmittance -> ratio tracking with roulette (volume.py:283-293, 399-463) */
static v3 vol_transmittance(const scene_t* sc, v3 ray_o, v3 ray_d, v3 thp, float max_t, rng_t* r) {
    float near_t, far_t;
    if (!sc->vol_type || !vol_intersect(sc, ray_o, ray_d, max_t, &near_t, &far_t)) return V(1.f, 1.f, 1.f);
    v3 ol = m3mulv(&sc->vol_inv_T, vsub(ray_o, sc->vol_trans)), dl = m3mulv(&sc->vol_inv_T, ray_d);
    float pdf; int ch = vol_pick_channel(sc, thp, r, &pdf);
    float inv_maj = 1.0f / rgb_select(sc->vol_majorant, ch);
    float Tr = 1.0f, t = near_t;
    for (;;) {
        t -= logf(1.0f - rng_float(r)) * inv_maj;
        if (t >= far_t) break;
        float u0 = rng_float(r), u1 = rng_float(r), u2 = rng_float(r);
        v3 dns = vol_density_lookup(sc, vadd(ol, vscale(dl, t)), V(u0, u1, u2));
        Tr *= fmaxf(0.0f, 1.0f - rgb_select(dns, ch) * inv_maj);
        if (Tr < 0.1f) {
            if (rng_float(r) >= Tr) { Tr = 0.0f; break; }
            Tr = 1.0f;
        }
    }
    return (sc->vol_type == 2) ? vol_channel_vec(ch, Tr / pdf) : V(Tr, Tr, Tr);
}
static int vpt_sample_mfp(const ctx_t* c, v3 ray_o, v3 ray_d, v3 thp, int idx, int in_free_space, float depth, rng_t* r, float* mfp, v3* beta) {   /* vpt.py:72-97 */
    int is_mi = 0; *mfp = depth; *beta = V(1.f, 1.f, 1.f);
    int world_valid = in_free_space && vpt_world_scattering(c);
    if (world_valid || vpt_is_scattering(c, idx)) {
        if (world_valid) is_mi = medium_sample_mfp(&c->sc->med[c->sc->n_objects], depth, r, mfp, beta);
        else if (!in_free_space) is_mi = medium_sample_mfp(&c->sc->med[idx], depth, r, mfp, beta);
    }
    if (c->sc->vol_type) {                  /* a grid-volume event overrides the homogeneous one (vpt.py:91-96) */
        v3 vb; float vt = vol_sample_mfp(c->sc, ray_o, ray_d, thp, depth, r, &vb);
        if (vt > 0.f) { is_mi = 2; *mfp = vt; *beta = vb; }
    }
    return is_mi;
}
static void vpt_ray_intersect(const ctx_t* c, v3 ray, v3 o, float min_depth, isect_t* it) {
    if (c->cfg->use_bvh && c->sc->node_num > 0) ray_intersect_bvh(c->sc, ray, o, min_depth, it);
    else ray_intersect_brute(c->sc, ray, o, min_depth, it);
}
/* VolumeRenderer.track_ray, vpt.py:99-138 (the accumulated optical length it also returns is unused by render) */
static v3 vpt_track_ray(const ctx_t* c, v3 cur_ray, v3 cur_point, v3 thp, float depth, rng_t* r, orc_stats* st) {
    v3 tr = V(1.f, 1.f, 1.f);
    if (c->sc->vol_type) tr = vol_transmittance(c->sc, cur_point, cur_ray, thp, depth, r);      /* vpt.py:107-108: draws from the path's stream */
    int in_free_space = 1;
    for (int k = 0; k < 7; k++) {
        isect_t it; vpt_ray_intersect(c, cur_ray, cur_point, depth, &it);
        st->n_track++;
        if (it.obj_id < 0) {
            if (!vpt_world_scattering(c)) break;
            it.min_depth = depth; in_free_space = 1; it.obj_id = -1;
        } else {
            if (vpt_non_null_surface(c, it.obj_id)) { tr = ZERO3; break; }
            in_free_space = vdot(it.n_g, cur_ray) < 0.f;
        }
        tr = vmul(tr, vpt_get_transmittance(c, it.obj_id, in_free_space, it.min_depth));
        cur_point = vadd(cur_point, vscale(cur_ray, it.min_depth));
        depth -= it.min_depth;
        if (depth <= 5e-5f) break;
    }
    return tr;
}
static float vpt_world_bound_time(const scene_t* sc, v3 o, v3 d) {             /* vpt.py:140-143 */
    v3 t_min = vdiv(vsub(sc->w_aabb_min, o), d), t_max = vdiv(vsub(sc->w_aabb_max, o), d);
    v3 m = vmaxv(t_min, t_max);
    float r = m.x;                       /* Vector.min(): NaN-propagating, like the fixture generator's */
    if (!isnan(r) && (isnan(m.y) || m.y < r)) r = m.y;
    if (!isnan(r) && (isnan(m.z) || m.z < r)) r = m.z;
    return r;
}
/* PathTracer.eval / sample_new_ray with the medium-interaction flag, path_tracer.py:424-480 */
static v3 vpt_eval(const ctx_t* c, isect_t* it, v3 incid, v3 out, int is_mi, int in_free_space) {
    if (is_mi) {
        const medium_t* m = in_free_space ? &c->sc->med[c->sc->n_objects] : &c->sc->med[it->obj_id];
        float p = phase_eval_p(m, incid, out);
        return V(p, p, p);
    }
    return pt_eval(c, it, incid, out);
}
static v3 vpt_sample_new_ray(const ctx_t* c, isect_t* it, v3 incid, int is_mi, int in_free_space, rng_t* r, v3* spec, float* pdf, int* is_specular) {
    if (is_mi) {
        /* is_mi == 2: the grid volume's own phase function (path_tracer.py:440-441); eval above keeps using the medium the
           flags point at, as upstream */
        const medium_t* m = (is_mi > 1) ? &c->sc->vol_ph : (in_free_space ? &c->sc->med[c->sc->n_objects] : &c->sc->med[it->obj_id]);
        *is_specular = 0;
        if (is_mi > 1) {                    /* GridVolume.sample_new_rays: scattering iff _type >= 1, whatever the phase type */
            *spec = V(1.f, 1.f, 1.f); *pdf = 1.f;
            float p; v3 local = phase_sample_p(m, incid, r, &p);
            m3 R; v3 dir = delocalize_rotate(incid, local, &R);
            *pdf = p; *spec = V(p, p, p);
            return dir;
        }
        return medium_sample_new_rays(m, incid, r, spec, pdf);
    }
    return pt_sample_new_ray(c, it, incid, r, spec, pdf, is_specular);
}

static v3 render_sample_vpt(const ctx_t* c, int i, int j, int cnt, rng_t* rng, orc_stats* st, trace_t* tr) {
    const orc_cfg* g = c->cfg; const scene_t* sc = c->sc;
    v3 ray_d = pix2ray(c, i, j, cnt, rng);
    v3 ray_o = c->cam_t;
    v3 color = ZERO3, throughput = V(1.f, 1.f, 1.f);
    float emission_weight = 1.0f;
    int in_free_space = 1, bounce = 0;
    const int world_scat = vpt_world_scattering(c);
    st->n_samples++;
    for (;;) {
        if (g->use_rr) {                                                        /* vpt.py:164-172 */
            float max_value = vmax(throughput);
            if (max_value < g->rr_threshold && bounce >= g->rr_bounce_th) {
                if (rng_float(rng) > max_value) break;
                else throughput = vscale(throughput, 1.f / (max_value + 1e-7f));
            }
        } else {
            if (vmax(throughput) < 1e-5f) break;
        }
        isect_t it; vpt_ray_intersect(c, ray_d, ray_o, -1.0f, &it);
        st->n_extend++;
        if (it.obj_id < 0) {
            if (!world_scat && !sc->vol_type) break;
            it.min_depth = vpt_world_bound_time(sc, ray_o, ray_d);
            in_free_space = 1; it.obj_id = -1;
        } else {
            in_free_space = vdot(it.n_g, ray_d) < 0.f;
        }
        v3 path_beta; float mfp;
        int is_mi = vpt_sample_mfp(c, ray_o, ray_d, throughput, it.obj_id, in_free_space, it.min_depth, rng, &mfp, &path_beta);
        it.min_depth = mfp;
        if (it.obj_id < 0 && !is_mi) break;
        v3 hit_point = vadd(vscale(ray_d, it.min_depth), ray_o);
        throughput = vmul(throughput, path_beta);
        if (!is_mi && !vpt_non_null_surface(c, it.obj_id)) { ray_o = hit_point; continue; }
        int hit_light = is_mi ? -1 : sc->emitter_id[it.obj_id];
        st->n_shade++;
        float direct_pdf = 1.0f, emitter_pdf = 1.0f;
        int break_flag = 0;
        v3 shadow_int = ZERO3, direct_int = ZERO3, direct_spec = V(1.f, 1.f, 1.f);
        get_uv_item(sc, 0, &it, &it.tex);                                       /* vpt.py:199 */
        for (int s = 0; s < g->num_shadow_ray; s++) {
            int emitter_valid;
            const src_t* emitter = pt_sample_light(c, hit_light, rng, &emitter_pdf, &emitter_valid);
            v3 light_dir = ZERO3;
            if (emitter_valid) {
                v3 emit_pos = src_sample_hit(c, emitter, hit_point, rng, &shadow_int, &direct_pdf);
                v3 to_emitter = vsub(emit_pos, hit_point);
                float emitter_d = vnorm(to_emitter);
                light_dir = vdivs(to_emitter, emitter_d);
                st->n_shadow++;
                v3 trn = vpt_track_ray(c, light_dir, hit_point, throughput, emitter_d, rng, st);
                if (trn.x != 0.f || trn.y != 0.f || trn.z != 0.f) st->n_lit++;
                shadow_int = vmul(shadow_int, trn);
                direct_spec = vpt_eval(c, &it, ray_d, light_dir, is_mi, in_free_space);
            } else { break_flag = 1; break; }
            float light_pdf = emitter_pdf * direct_pdf;
            if (g->use_mis) {
                float mis_w = 1.0f;
                if (!(emitter->bool_bits & 0x01)) {
                    float bsdf_pdf_v = is_mi ? direct_spec.x : pt_surface_pdf(c, &it, light_dir, ray_d);
                    mis_w = balance_heuristic(light_pdf, bsdf_pdf_v);
                }
                direct_int = vadd(direct_int, vdivs(vscale(vmul(direct_spec, shadow_int), mis_w), emitter_pdf));
            } else {
                direct_int = vadd(direct_int, vdivs(vmul(direct_spec, shadow_int), emitter_pdf));
            }
        }
        if (!break_flag) direct_int = vscale(direct_int, c->inv_num_shadow_ray);
        v3 emit_int = ZERO3;
        if (hit_light >= 0) emit_int = src_eval_le(&sc->src[hit_light], vsub(hit_point, ray_o), it.n_g);     /* n_g here, vpt.py:233 */
        v3 indirect_spec; float ray_pdf; int is_specular;
        v3 new_d = vpt_sample_new_ray(c, &it, ray_d, is_mi, in_free_space, rng, &indirect_spec, &ray_pdf, &is_specular);
        if (tr && tr->n_events < tr->max_events) {
            float* e = tr->ev + 18 * tr->n_events++;
            v3 ew = vscale(emit_int, emission_weight);
            e[0] = (float)it.obj_id; e[1] = is_mi ? -2.f : (float)it.prim_id; e[2] = it.min_depth;
            e[3] = direct_int.x; e[4] = direct_int.y; e[5] = direct_int.z;
            e[6] = ew.x; e[7] = ew.y; e[8] = ew.z;
            e[9] = throughput.x; e[10] = throughput.y; e[11] = throughput.z;
            e[12] = hit_point.x; e[13] = hit_point.y; e[14] = hit_point.z; e[15] = new_d.x; e[16] = new_d.y; e[17] = new_d.z;
        }
        ray_d = new_d;
        ray_o = hit_point;
        color = vadd(color, vmul(vadd(direct_int, vscale(emit_int, emission_weight)), throughput));
        if (!is_mi) {
            if (vmax(indirect_spec) == 0.f || ray_pdf == 0.f) break;
            throughput = vmul(throughput, vdivs(indirect_spec, ray_pdf));
        }
        bounce++;
        if (bounce >= g->max_bounce) break;
        if (it.obj_id >= 0) {                                                   /* vpt.py:247-253: weights with THIS interaction */
            hit_light = sc->emitter_id[it.obj_id];
            if (g->use_mis) {
                float e_pdf = 0.0f;
                if (hit_light >= 0 && pt_is_delta(c, it.obj_id) == 0 && !is_specular)
                    e_pdf = src_solid_angle_pdf(&sc->src[hit_light], &it, ray_d);
                emission_weight = balance_heuristic(ray_pdf, e_pdf);
            }
        }
    }
    st->n_draws += rng->draw;
    if (isnan(color.x)) color.x = 0.f;
    if (isnan(color.y)) color.y = 0.f;
    if (isnan(color.z)) color.z = 0.f;
    return color;
}

static void make_ctx(ctx_t* c, const scene_t* sc, const orc_cfg* cfg) {
    c->sc = sc; c->cfg = cfg;
    c->inv_num_shadow_ray = (cfg->num_shadow_ray > 0) ? 1.f / (float)cfg->num_shadow_ray : 1.f;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c->cam_r.m[i][j] = cfg->cam_r[3 * i + j];
    c->cam_t = V(cfg->cam_t[0], cfg->cam_t[1], cfg->cam_t[2]);
}

/* Renderer.render x n_spp.  accum is color[w][h][3] ([i=x][j=y]), *cnt the sample counter. */
ORC_API int orc_render(const scene_t* sc, const orc_cfg* cfg, float* accum, int* cnt, int n_spp, int n_threads, orc_stats* stats) {
    ctx_t c; make_ctx(&c, sc, cfg);
    const int W = cfg->width, H = cfg->height;
    orc_stats total; memset(&total, 0, sizeof(total));
    for (int s = 0; s < n_spp; s++) {
        *cnt += 1;                                  /* vanilla_renderer.py:34 */
        const int cur = *cnt;
#ifdef _OPENMP
        if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
        long long a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : a0, a1, a2, a3, a4, a5, a6)
        for (int p = 0; p < W * H; p++) {
            int i = p / H, j = p % H;
            if (cfg->do_crop && !(i >= cfg->start_x && i < cfg->end_x && j >= cfg->start_y && j < cfg->end_y)) continue;
            rng_t rng; rng_seed(&rng, (uint32_t)p, cfg->seed, (uint32_t)cur);
            orc_stats st; memset(&st, 0, sizeof(st));
            v3 col = cfg->volumetric ? render_sample_vpt(&c, i, j, cur, &rng, &st, NULL) : render_sample(&c, i, j, cur, &rng, &st, NULL);
            float* px = accum + 3 * (size_t)p;
            px[0] += col.x; px[1] += col.y; px[2] += col.z;
            a0 += st.n_samples; a1 += st.n_shade; a2 += st.n_shadow; a3 += st.n_lit; a4 += st.n_draws; a5 += st.n_extend; a6 += st.n_track;
        }
        total.n_samples += a0; total.n_shade += a1; total.n_shadow += a2; total.n_lit += a3; total.n_draws += a4;
        total.n_extend += a5; total.n_track += a6;
    }
    if (stats) *stats = total;
    (void)n_threads;
    return 0;
}

/* One pixel-sample with a per-bounce event trace and either RNG mode (script != NULL -> scripted). */
ORC_API int orc_trace_sample(const scene_t* sc, const orc_cfg* cfg, int i, int j, int cnt, const double* script, int script_n,
                             float color_out[3], float* events, int max_events, int* n_events, int* n_draws) {
    ctx_t c; make_ctx(&c, sc, cfg);
    rng_t rng; rng_seed(&rng, (uint32_t)(i * cfg->height + j), cfg->seed, (uint32_t)cnt);
    if (script) { rng.mode = 1; rng.script = script; rng.script_n = script_n; }
    orc_stats st; memset(&st, 0, sizeof(st));
    trace_t tr = {max_events, 0, events};
    v3 col = cfg->volumetric ? render_sample_vpt(&c, i, j, cnt, &rng, &st, events ? &tr : NULL)
                             : render_sample(&c, i, j, cnt, &rng, &st, events ? &tr : NULL);
    color_out[0] = col.x; color_out[1] = col.y; color_out[2] = col.z;
    if (n_events) *n_events = tr.n_events;
    if (n_draws) *n_draws = (int)rng.draw;
    return 0;
}

/* ------------------------------------------------ unit entry points (goldens) */
static void load_it(isect_t* it, const float n_s[3], const float n_g[3], float min_depth) {
    memset(it, 0, sizeof(*it));
    it->n_s = V(n_s[0], n_s[1], n_s[2]); it->n_g = V(n_g[0], n_g[1], n_g[2]);
    it->tex = V(-1.f, -1.f, -1.f); it->min_depth = min_depth;
}
static void load_bxdf(bxdf_t* b, const int bi[4], const float bf[13]) {
    b->type = bi[0]; b->is_delta = bi[1]; b->is_bsdf = bi[2];
    b->k_d = as_v3(bf)[0]; b->k_s = as_v3(bf)[1]; b->k_g = as_v3(bf)[2]; b->mean = as_v3(bf)[3]; b->ior = bf[12];
}
static void st3(float* o, v3 a) { o[0] = a.x; o[1] = a.y; o[2] = a.z; }

/* eval (f*cos), pdf for given directions; BRDF or BSDF by bi[2] */
ORC_API void orc_bxdf_eval_pdf(const int bi[4], const float bf[13], float world_ior, const float n_s[3], const float n_g[3],
                               const float incid[3], const float out[3], float eval_out[3], float* pdf_out) {
    bxdf_t b; load_bxdf(&b, bi, bf); isect_t it; load_it(&it, n_s, n_g, 1.f);
    v3 wi = LD3(incid), wo = LD3(out);
    if (!b.is_bsdf) { st3(eval_out, brdf_eval(&b, &it, wi, wo)); *pdf_out = brdf_pdf(&b, &it, wo, wi); }
    else {
        v3 e = ZERO3;
        if (b.type == 0) e = eval_det_refraction(&b, &it, wi, wo, world_ior);
        else if (b.type == 1) e = eval_lambertian_trans(&b, &it, wi, wo, world_ior);
        st3(eval_out, e); *pdf_out = bsdf_pdf(&b, &it, wo, wi, world_ior);
    }
}
/* sample with a scripted RNG */
static void unit_rng(rng_t* r, const double* script, int script_n, uint32_t key, uint32_t seed) {
    if (script) { memset(r, 0, sizeof(*r)); r->mode = 1; r->script = script; r->script_n = script_n; }
    else rng_seed(r, key, seed, 1u);          /* Philox stream keyed like a pixel-sample: (key, seed), sample 1 */
}
ORC_API void orc_bxdf_sample(const int bi[4], const float bf[13], float world_ior, const float n_s[3], const float n_g[3],
                             const float incid[3], const double* script, int script_n, uint32_t key, uint32_t seed,
                             float dir_out[3], float spec_out[3], float* pdf_out, int* is_specular, int* n_draws) {
    bxdf_t b; load_bxdf(&b, bi, bf); isect_t it; load_it(&it, n_s, n_g, 1.f);
    rng_t r; unit_rng(&r, script, script_n, key, seed);
    v3 spec; float pdf; int sp = 0; v3 dir;
    if (!b.is_bsdf) dir = brdf_sample(&b, &it, LD3(incid), &r, &spec, &pdf, &sp);
    else {
        dir = ZERO3; spec = ZERO3; pdf = 0.f;
        if (b.type == 0) dir = sample_det_refraction(&b, &it, LD3(incid), world_ior, &r, &spec, &pdf);
        else if (b.type == 1) dir = sample_lambertian_trans(&b, &it, LD3(incid), world_ior, &r, &spec, &pdf, &sp);
    }
    st3(dir_out, dir); st3(spec_out, spec); *pdf_out = pdf; *is_specular = sp; *n_draws = (int)r.draw;
}
/* Medium functions on explicit inputs (bxdf/medium.py:84-125); RNG = Philox stream (key, seed), sample 1.
 * mode 0  sample_mfp       in = max_depth                  out = is_mi, t, beta rgb, draws
 * mode 1  sample_new_rays  in = incid xyz                  out = dir xyz, spec rgb, pdf, draws
 * mode 2  eval + transmit  in = incid xyz, out xyz, depth  out = phase value, transmittance rgb */
ORC_API void orc_medium_probe(int type, const float f[16], int mode, const float* in, uint32_t key, uint32_t seed, float* out) {
    medium_t m; m.type = type; m.ior = f[0];
    m.u_s = LD3(f + 1); m.u_a = LD3(f + 4); m.u_e = LD3(f + 7); m.par = LD3(f + 10); m.pdf = LD3(f + 13);
    rng_t r; unit_rng(&r, NULL, 0, key, seed);
    if (mode == 0) {
        float t; v3 beta; int is_mi = medium_sample_mfp(&m, in[0], &r, &t, &beta);
        out[0] = (float)is_mi; out[1] = t; st3(out + 2, beta); out[5] = (float)r.draw;
    } else if (mode == 1) {
        v3 spec; float pdf; v3 d = medium_sample_new_rays(&m, LD3(in), &r, &spec, &pdf);
        st3(out, d); st3(out + 3, spec); out[6] = pdf; out[7] = (float)r.draw;
    } else {
        out[0] = (m.type >= 0) ? phase_eval_p(&m, LD3(in), LD3(in + 3)) : 1.f;
        st3(out + 1, medium_transmittance(&m, in[6]));
    }
}
ORC_API void orc_rotation_between(const float a[3], const float b[3], float R_out[9]) {
    m3 R; rotation_between(LD3(a), LD3(b), &R);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R_out[3 * i + j] = R.m[i][j];
}
ORC_API float orc_fresnel_equation(float a, float b, float c, float d) { return fresnel_equation(a, b, c, d); }
ORC_API void orc_snell_refraction(const float incid[3], const float normal[3], float dot_n, float ni, float nr, float out[3], float* cos_r2) {
    st3(out, snell_refraction(LD3(incid), LD3(normal), dot_n, ni, nr, cos_r2));
}
/* emitter sample_hit with scripted RNG; needs a scene for attached geometry */
ORC_API void orc_src_sample_hit(const scene_t* sc, int src_idx, const float hit_pos[3], const double* script, int script_n,
                                uint32_t key, uint32_t seed, float pos_out[3], float int_out[3], float* pdf_out, int* n_draws) {
    ctx_t c; memset(&c, 0, sizeof(c)); c.sc = sc;
    rng_t r; unit_rng(&r, script, script_n, key, seed);
    v3 ri; float rp;
    v3 p = src_sample_hit(&c, &sc->src[src_idx], LD3(hit_pos), &r, &ri, &rp);
    st3(pos_out, p); st3(int_out, ri); *pdf_out = rp; *n_draws = (int)r.draw;
}
ORC_API void orc_src_eval(const scene_t* sc, int src_idx, const float inci_dir[3], const float normal[3], float min_depth,
                          const float ray_d[3], float le_out[3], float* sa_pdf_out) {
    isect_t it; load_it(&it, normal, normal, min_depth);
    st3(le_out, src_eval_le(&sc->src[src_idx], LD3(inci_dir), LD3(normal)));
    *sa_pdf_out = src_solid_angle_pdf(&sc->src[src_idx], &it, LD3(ray_d));
}
/* closest hit / occlusion for a batch of rays: hit_out[n*4] = (obj, prim, u|sphere-u, v), t_out[n] */
ORC_API void orc_intersect_batch(const scene_t* sc, int use_bvh, int n, const float* o, const float* d,
                                 int* obj_out, int* prim_out, float* t_out, float* uv_out, float* ns_out) {
    for (int k = 0; k < n; k++) {
        isect_t it;
        if (use_bvh && sc->node_num > 0) ray_intersect_bvh(sc, LD3(d + 3 * k), LD3(o + 3 * k), -1.f, &it);
        else ray_intersect_brute(sc, LD3(d + 3 * k), LD3(o + 3 * k), -1.f, &it);
        obj_out[k] = it.obj_id; prim_out[k] = it.prim_id; t_out[k] = it.min_depth;
        uv_out[2 * k] = it.u; uv_out[2 * k + 1] = it.v;
        if (ns_out) st3(ns_out + 3 * k, it.n_s);
    }
}
ORC_API void orc_occluded_batch(const scene_t* sc, int use_bvh, int n, const float* o, const float* d, const float* tmax, int* occ_out) {
    for (int k = 0; k < n; k++)
        occ_out[k] = (use_bvh && sc->node_num > 0) ? does_intersect_bvh(sc, LD3(d + 3 * k), LD3(o + 3 * k), tmax[k])
                                                   : does_intersect_brute(sc, LD3(d + 3 * k), LD3(o + 3 * k), tmax[k]);
}
ORC_API void orc_pix2ray(const scene_t* sc, const orc_cfg* cfg, int i, int j, int cnt, const double* script, int script_n, float out[3]) {
    ctx_t c; make_ctx(&c, sc, cfg);
    rng_t r; memset(&r, 0, sizeof(r)); r.mode = 1; r.script = script; r.script_n = script_n;
    st3(out, pix2ray(&c, i, j, cnt, &r));
}
/* raw RNG stream, for cross-checking the HIP generator and the golden generator's shim */
ORC_API void orc_rng_stream(uint32_t pixel, uint32_t seed, uint32_t sample, int n, uint32_t* out) {
    rng_t r; rng_seed(&r, pixel, seed, sample);
    for (int k = 0; k < n; k++) out[k] = rng_u32(&r);
}
ORC_API void orc_philox(const uint32_t c[4], const uint32_t k[2], uint32_t out[4]) { philox4x32_10(c[0], c[1], c[2], c[3], k[0], k[1], out); }
ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
