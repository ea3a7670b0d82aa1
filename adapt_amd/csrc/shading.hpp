// shading.hpp — device-side surface models, emitters, frames and optics for the gfx950 path tracer.
//
// Behavioural contract = AdaPT's `pt` renderer (citations under /root/reference):
//   bxdf/brdf.py:147-601        BRDF: Blinn-Phong(0) Lambertian(1) Specular(2) Microfacet(3: upstream's opt-in switch)
//                               Mod-Phong(4) Fresnel-Blend(5) Oren-Nayar(6) Thin-coat(7)
//   bxdf/bsdf.py:61-262         BSDF: det-refraction(0) Lambertian-transmission(1) null(-1)
//   emitters/abtract_source.py:81-232   sample_hit / eval_le / solid_angle_pdf
//   sampler/general_sampling.py:16-123  direction samplers, sample_triangle, balance heuristic
//   la/cam_transform.py:51-105, la/geo_optics.py:14-74   Rodrigues frames, reflection/refraction/Fresnel
// including the quirks SURVEY.md A.3 lists (they are what "same image as AdaPT" means).
// Textures are out of scope: the albedo is always k_d (the reference's `it.tex` is INVALID).
#pragma once
#include "rng.hpp"
#include "vec.hpp"

#define BRDF_EPS 1e-7f

struct DevBxdf {          // one per object: bxdf/brdf.py:152-158 | bxdf/bsdf.py:68-73 (+ medium ior)
    int type, is_delta, is_bsdf, _pad;
    f3 k_d, k_s, k_g, mean;
    float ior;
    float _pad2[3];
};
struct DevSrc {           // emitters/abtract_source.py:44-54
    int type, bool_bits, obj_ref_id;
    int prim_first;       // area emitters: first primitive of the attached object (copied from obj_info at scene creation, so that
    f3 intensity, dir, pos;   // sampling the emitter does not chase obj_info -> primitive through two dependent loads)
    float inv_area, r;
    int prim_count;       // area emitters: triangles of the attached mesh, or -1 for a sphere
};
struct Hit {              // tracer/interaction.py:11-39 minus texture/uv members
    int obj_id, prim_id;
    f3 n_s, n_g;
    float min_depth;
};

// ------------------------------------------------------------------- frames
// Rodrigues rotation taking `fixed` onto `target`; diag(sign(cos)) when (anti)parallel
APT_D void rotation_between(f3 fixed, f3 target, m33& R) {
    f3 axis = cross(fixed, target);
    float c = dot(fixed, target);
    if (fabsf(c) < 1.0f - 1e-5f) {
        f3 n = fnormalize(axis);
        float k = 1.0f - c;
        float kx = k * n.x, ky = k * n.y, kz = k * n.z;
        R.m[0][0] = (c + kx * n.x) + 0.0f;     R.m[0][1] = (0.0f + kx * n.y) + (-axis.z); R.m[0][2] = (0.0f + kx * n.z) + axis.y;
        R.m[1][0] = (0.0f + ky * n.x) + axis.z; R.m[1][1] = (c + ky * n.y) + 0.0f;        R.m[1][2] = (0.0f + ky * n.z) + (-axis.x);
        R.m[2][0] = (0.0f + kz * n.x) + (-axis.y); R.m[2][1] = (0.0f + kz * n.y) + axis.x; R.m[2][2] = (c + kz * n.z) + 0.0f;
    } else {
        float s = sgn(c);
        R.m[0][0] = s; R.m[0][1] = 0.f; R.m[0][2] = 0.f;
        R.m[1][0] = 0.f; R.m[1][1] = s; R.m[1][2] = 0.f;
        R.m[2][0] = 0.f; R.m[2][1] = 0.f; R.m[2][2] = s;
    }
}
// local frame has +y along `anchor`
APT_D f3 delocalize(f3 anchor, f3 local_dir) { m33 R; rotation_between(mk3(0.f, 1.f, 0.f), anchor, R); return mul(R, local_dir); }
APT_D f3 delocalize(f3 anchor, f3 local_dir, m33& R) { rotation_between(mk3(0.f, 1.f, 0.f), anchor, R); return mul(R, local_dir); }
APT_D f3 localize(f3 anchor, f3 global_dir) { m33 R; rotation_between(anchor, mk3(0.f, 1.f, 0.f), R); return mul(R, global_dir); }

struct RawAngles { float cos_t, sin_t, cos_p, sin_p; };
APT_D RawAngles raw_of_local(f3 l) {               // convert_to_raw(..., localize = False)
    RawAngles a;
    a.cos_t = l.y;
    a.sin_t = ssqrt(fmaxf(0.f, 1.f - a.cos_t * a.cos_t));
    a.cos_p = 1.f; a.sin_p = 0.f;
    if (a.sin_t > 1e-5f) { a.cos_p = sdiv(l.x, a.sin_t); a.sin_p = sdiv(l.z, a.sin_t); }
    return a;
}
APT_D RawAngles to_raw(f3 d_in, f3 normal) { return raw_of_local(localize(normal, d_in)); }

// ------------------------------------------------------------------- optics
APT_D f3 reflect_in(f3 ray, f3 normal) {           // inci_reflect_dir
    float d = dot(normal, ray);
    return normalize(ray - (normal * 2.f) * d);
}
APT_D f3 reflect_in(f3 ray, f3 normal, float& d) {
    d = dot(normal, ray);
    return normalize(ray - (normal * 2.f) * d);
}
// pow(x, 5) with a constant integer exponent: multiplication chain by squaring, x * ((x*x)*(x*x)) (float parity)
APT_D float pow5(float x) { float x2 = x * x; float x4 = x2 * x2; return x * x4; }
APT_D f3 schlick(f3 r_s, float dot_val) {
    float p = pow5(1.f - dot_val);
    return r_s + mk3(1.f - r_s.x, 1.f - r_s.y, 1.f - r_s.z) * p;
}
APT_D float fresnel_dielectric(float n_in, float n_out, float cos_inc, float cos_ref) {
    float a = n_in * cos_inc, b = n_out * cos_inc, c = n_in * cos_ref, d = n_out * cos_ref;
    float rs = (a - d) / (a + d);
    float rp = (c - b) / (c + b);
    return 0.5f * (rs * rs + rp * rp);
}
APT_D bool total_reflection(float dot_normal, float ni, float nr) {
    return (1.f - sqr(ni / nr) * (1.f - sqr(dot_normal))) < 0.f;
}
APT_D f3 refract_snell(f3 incid, f3 normal, float dot_n, float ni, float nr, float& cos_r2) {
    float exiting = sgn(dot_n);
    float ratio = ni / nr;
    cos_r2 = 1.f - sqr(ratio) * (1.f - sqr(dot_n));
    if (cos_r2 > 0.f) return normalize((incid * ratio - normal * (ratio * dot_n)) + normal * (exiting * sqrtf(cos_r2)));
    return mk3(0.f, 0.f, 0.f);
}

// ----------------------------------------------------------------- samplers
APT_D f3 polar_dir(float cos_t, float sin_t, float phi) {
    float s, c; apt_sincos(phi, &s, &c);
    return mk3(c * sin_t, cos_t, s * sin_t);
}
APT_D f3 sample_cosine_hemisphere(Philox& r, float& pdf) {
    float eps = rng_float(r);
    float cos_t = ssqrt(eps), sin_t = ssqrt(1.f - eps);
    float phi = APT_2PI * rng_float(r);
    pdf = cos_t * APT_INV_PI;
    return polar_dir(cos_t, sin_t, phi);
}
APT_D f3 sample_mod_phong_lobe(Philox& r, float alpha, float& pdf) {
    float cos_t = apt_pow(rng_float(r), 1.f / (alpha + 1.f));
    float sin_t = ssqrt(1.f - cos_t * cos_t);
    float phi = APT_2PI * rng_float(r);
    pdf = 0.5f * (1.f + alpha) * apt_pow(cos_t, alpha) * APT_INV_PI;
    return polar_dir(cos_t, sin_t, phi);
}
APT_D f3 sample_uniform_sphere(Philox& r, float& pdf) {
    float cos_t = 2.f * rng_float(r) - 1.f;
    float sin_t = ssqrt(1.f - cos_t * cos_t);
    float phi = APT_2PI * rng_float(r);
    pdf = APT_INV_2PI * 0.5f;
    return polar_dir(cos_t, sin_t, phi);
}
APT_D f3 sample_fresnel_half(Philox& r, float nu, float nv, float& power_coeff) {
    float eps1 = rng_float(r) * 4.f;
    float inner = eps1 - floorf(eps1);
    float tan_phi = sqrtf((nu + 1.f) / (nv + 1.f)) * apt_tan(APT_PI_2 * inner);
    float cos_phi2 = 1.f / (1.f + tan_phi * tan_phi);
    float sin_phi2 = 1.f - cos_phi2;
    float cos_phi = sqrtf(cos_phi2);
    if (eps1 > 1.f && eps1 <= 3.f) cos_phi *= -1.f;
    float sin_phi = sqrtf(sin_phi2) * sgn(2.f - eps1);
    power_coeff = nu * cos_phi2 + nv * sin_phi2;
    float cos_t = apt_pow(1.f - rng_float(r), 1.f / (power_coeff + 1.f));
    float sin_t = ssqrt(1.f - cos_t * cos_t);
    return mk3(cos_phi * sin_t, cos_t, sin_phi * sin_t);
}
APT_D f3 sample_on_triangle(Philox& r, f3 dv1, f3 dv2) {
    float u1 = rng_float(r), u2 = rng_float(r);
    f3 pt = dv1 * u1 + dv2 * u2;
    if (u1 + u2 > 1.0f) pt = (dv1 + dv2) - pt;
    return pt;
}
APT_D float balance(float a, float b) { return (a > 1e-7f) ? sdiv(a, a + b) : 0.f; }

// -------------------------------------------------------------------- BRDFs
APT_D f3 lambert_eval(const DevBxdf& b, f3 normal, f3 out) {
    float c = fmaxf(0.f, dot(normal, out));
    return (b.k_d * APT_INV_PI) * c;
}
APT_D f3 lambert_sample(const DevBxdf& b, f3 normal, Philox& r, f3& spec, float& pdf) {
    f3 local = sample_cosine_hemisphere(r, pdf);
    f3 out = delocalize(normal, local);
    spec = lambert_eval(b, normal, out);
    return out;
}
// Mask bit 11 ("no specular lobe"): every Blinn-Phong material of the scene has k_s = (0, 0, 0) and finite k_g >= 0, e.g. the
// "phong" walls of BASELINE C3.  The lobe is then k_s * (finite) = +0 and k_d + 0 = k_d exactly, so it is left out together with
// its pow - which is what costs the class kernel a third of its registers (187 -> 4 waves per SIMD without it).
template <int BM>
APT_D f3 blinn_phong_eval(const DevBxdf& b, const Hit& it, f3 in, f3 out) {
    float c = fmaxf(0.f, dot(it.n_s, out));
    if ((BM >> 11) & 1) return (b.k_d * APT_INV_PI) * c;
    f3 h = out - in;
    if (max3(abs3(h)) > BRDF_EPS) h = normalize(h); else h = splat3(0.f);
    float dc = fmaxf(0.f, dot(h, it.n_s));
    f3 glossy = pow_sv(dc, b.k_g);
    return ((b.k_d + b.k_s * (((b.k_g + 2.0f) * 0.5f) * glossy)) * APT_INV_PI) * c;
}
APT_D f3 mod_phong_eval(const DevBxdf& b, const Hit& it, f3 in, f3 out) {
    float dn = dot(it.n_s, out);
    f3 spec = splat3(0.f);
    if (dn > 0.f) {
        f3 refl = normalize((it.n_s * 2.f) * dn - out);
        float dv = fmaxf(0.f, -dot(in, refl));
        f3 glossy = pow_sv(dv, b.k_g) * b.k_s;
        spec = ((((b.k_g + 2.f) * 0.5f) * glossy) * APT_INV_PI) * dn;
        spec = spec + lambert_eval(b, it.n_s, out);
    }
    return spec;
}
APT_D f3 mod_phong_sample(const DevBxdf& b, const Hit& it, f3 incid, Philox& r, f3& spec, float& pdf) {
    float eps = rng_float(r);
    f3 out = mk3(0.f, 1.f, 0.f);
    spec = splat3(0.f);
    pdf = max3(b.k_d);
    float ks_max = max3(b.k_s);
    if (eps < pdf) {
        float lp;
        out = lambert_sample(b, it.n_s, r, spec, lp);
        pdf *= lp;
    } else if (eps < pdf + ks_max) {
        f3 local = sample_mod_phong_lobe(r, b.mean.z, pdf);
        f3 n = delocalize(it.n_s, local);
        out = normalize((n * -2.f) * dot(incid, n) + incid);
        spec = mod_phong_eval(b, it, incid, out);
        pdf *= ks_max;
    } else {
        pdf = 1.f - pdf - ks_max;
    }
    return out;
}
APT_D void fb_cos2_sin2(f3 half_vec, f3 normal, const m33& R, float dot_half, float& c2, float& s2) {
    f3 tx = mul(R, mk3(1.f, 0.f, 0.f));
    float d = dot(tx, normalize(half_vec - normal * dot_half));
    c2 = d * d; s2 = 1.f - c2;
}
APT_D f3 fresnel_blend_eval(const DevBxdf& b, const Hit& it, f3 in, f3 out, const m33& R) {
    f3 h = out - in;
    float d_out = dot(it.n_s, out);
    f3 spec = splat3(0.f);
    if (d_out > 0.f && max3(abs3(h)) > 1e-4f) {
        h = normalize(h);
        float d_in = -dot(it.n_s, in);
        float d_half = fabsf(dot(it.n_s, h));
        float d_hk = fabsf(dot(h, out));
        f3 F = schlick(b.k_s, d_hk);
        float c2, s2; fb_cos2_sin2(h, it.n_s, R, d_half, c2, s2);
        float denom = d_hk * fmaxf(d_in, d_out);
        float lobe = b.k_g.z * apt_pow(d_half, b.k_g.x * c2 + b.k_g.y * s2);
        f3 specular = (F * lobe) / denom;
        f3 diffuse = (b.k_d * (float)(28. / (23. * 3.14159265358979323846))) * mk3(1.f - b.k_s.x, 1.f - b.k_s.y, 1.f - b.k_s.z);
        float p_in = pow5(1.f - d_in / 2.f);
        float p_out = pow5(1.f - d_out / 2.f);
        diffuse = diffuse * ((1.f - p_in) * (1.f - p_out));
        spec = (specular + diffuse) * d_out;
    }
    return spec;
}
APT_D f3 fresnel_blend_sample(const DevBxdf& b, const Hit& it, f3 incid, Philox& r, f3& spec, float& pdf) {
    float pc;
    f3 local = sample_fresnel_half(r, b.k_g.x, b.k_g.y, pc);
    m33 R;
    f3 half = delocalize(it.n_s, local, R);
    float d_inc;
    f3 out = reflect_in(incid, half, d_inc);
    float half_pdf = b.k_g.z * apt_pow(dot(half, it.n_s), pc);
    pdf = sdiv(half_pdf, fmaxf(fabsf(d_inc), BRDF_EPS));
    bool valid = dot(it.n_s, out) > 0.f;
    if (rng_float(r) > 0.5f) {
        f3 s_; float p_;
        out = lambert_sample(b, it.n_s, r, s_, p_);
    }
    pdf = 0.5f * (pdf + fabsf(dot(out, it.n_s)) * APT_INV_PI);
    spec = valid ? fresnel_blend_eval(b, it, incid, out, R) : splat3(0.f);
    return out;
}
APT_D f3 oren_nayar_eval(const DevBxdf& b, const Hit& it, f3 in, f3 out) {
    RawAngles wi = to_raw(-in, it.n_s), wo = to_raw(out, it.n_s);
    float max_cos = 0.f;
    if (wi.sin_t > 1e-5f && wo.sin_t > 1e-5f) max_cos = fmaxf(0.f, wi.cos_p * wo.cos_p + wi.sin_p * wo.sin_p);
    float sin_alpha, tan_beta;
    float aci = fabsf(wi.cos_t), aco = fabsf(wo.cos_t);
    if (aci > aco) { sin_alpha = wo.sin_t; tan_beta = wi.sin_t / aci; }
    else           { sin_alpha = wi.sin_t; tan_beta = wo.sin_t / aco; }
    float f = b.k_g.x + b.k_g.y * max_cos * sin_alpha * tan_beta;
    return ((b.k_d * APT_INV_PI) * f) * fabsf(wo.cos_t);
}
APT_D f3 thin_coat_sample(const DevBxdf& b, const Hit& it, f3 incid, Philox& r, f3& spec, float& pdf, bool& is_specular) {
    pdf = 1.0f; spec = splat3(0.f);
    f3 out = mk3(0.f, 1.f, 0.f);
    float dn = dot(incid, it.n_s);
    float cos_r2;
    f3 refra_in = refract_snell(incid, it.n_s, dn, 1.0f, b.k_g.z, cos_r2);
    float F_in = fresnel_dielectric(1.f, b.k_g.x, fabsf(dn), sqrtf(cos_r2));     // k_g[0] here, as upstream (brdf.py:361)
    is_specular = false;
    if (rng_float(r) > F_in) {
        f3 local = sample_cosine_hemisphere(r, pdf);
        out = delocalize(it.n_s, local);
        float d_out = dot(out, it.n_s);
        if (!total_reflection(d_out, b.k_g.z, 1.0f)) {
            f3 refra_out = refract_snell(out, it.n_s, d_out, b.k_g.z, 1.0f, cos_r2);
            float F_out = fresnel_dielectric(b.k_g.z, 1.f, fabsf(d_out), sqrtf(cos_r2));
            pdf *= (1.f - F_in);
            out = refra_out;
            spec = oren_nayar_eval(b, it, refra_in, out);
            spec = spec * ((1.f - F_in) * (1.f - F_out));
        }
    } else {
        spec = b.k_s * F_in;
        out = reflect_in(incid, it.n_s);
        pdf = F_in;
        is_specular = true;
    }
    return out;
}
APT_D f3 thin_coat_eval(const DevBxdf& b, const Hit& it, f3 in, f3 out) {
    f3 refl = reflect_in(in, it.n_s);
    float d_in = dot(in, it.n_s);
    float cos_r2;
    f3 refra_in = refract_snell(in, it.n_s, d_in, 1.0f, b.k_g.z, cos_r2);
    float F_in = fresnel_dielectric(1.f, b.k_g.z, fabsf(d_in), sqrtf(cos_r2));
    if (fabsf(dot(out, refl)) > (1.f - 1e-4f)) return b.k_s * F_in;
    float d_out = dot(out, it.n_s);
    f3 refra_out = refract_snell(out, it.n_s, d_out, 1.0f, b.k_g.z, cos_r2);
    float F_out = fresnel_dielectric(1.0f, b.k_g.z, fabsf(d_out), sqrtf(cos_r2));
    return oren_nayar_eval(b, it, refra_in, refra_out) * (1.f - fmaxf(F_in, F_out));
}
APT_D float thin_coat_fresnel(const DevBxdf& b, const Hit& it, f3 in) {
    float d_in = dot(in, it.n_s);
    float ratio = 1.0f / b.k_g.z;
    float cos_r2 = 1.f - sqr(ratio) * (1.f - sqr(d_in));
    return fresnel_dielectric(1.f, b.k_g.z, fabsf(d_in), sqrtf(cos_r2));
}

// ---------------------------------------------------- Trowbridge-Reitz microfacet BRDF (type 3)
// sampler/microfacet.py:27-176 + bxdf/brdf.py:428-484.  Upstream ships it switched off (`__ENABLE_MICROFACET__`, brdf.py:8) and its
// parser then turns a microfacet BRDF into a Lambertian one, so type 3 reaches the device only with the switch on
// (adapt_amd/materials.py ENABLE_MICROFACET): that is what type 3 means here.  k_g = (alpha_x, alpha_y, .), k_s = (ior outside, ior inside, .).
APT_D float fresnel_eval(float cos_v, float n_in, float n_tr) {            // geo_optics.py:29-44
    const bool neg = cos_v < 0.f;
    const float cv = neg ? -cos_v : cos_v;
    const float ior_in = neg ? n_tr : n_in, ior_tr = neg ? n_in : n_tr;
    const float sin_v = sqrtf(fmaxf(0.f, 1.f - cv * cv));
    const float sin_t = ior_in / ior_tr * sin_v;
    const float cos_tr = sqrtf(fmaxf(0.f, 1.f - sin_t * sin_t));
    return fresnel_dielectric(ior_in, ior_tr, cv, cos_tr);
}
APT_D float trow_reitz_D(const RawAngles& w, f3 alphas) {
    float pdf = 0.f;
    if (w.cos_t > 0.f) {
        const float d2 = w.cos_t * w.cos_t, d4 = d2 * d2;
        const float tan2 = w.sin_t * w.sin_t / d2;
        const float e = (w.cos_p * w.cos_p / (alphas.x * alphas.x) + w.sin_p * w.sin_p / (alphas.y * alphas.y)) * tan2;
        pdf = 1.f / (APT_PI * alphas.x * alphas.y * d4 * (1.f + e) * (1.f + e));
    }
    return pdf;
}
APT_D float trow_reitz_lambda(f3 dir_vec, f3 alphas, f3 normal) {
    float value = 0.f;
    const RawAngles w = to_raw(dir_vec, normal);
    const float abs_cos = fabsf(w.cos_t);
    if (abs_cos > 1e-5f) {
        const float abs_tan = w.sin_t / abs_cos;
        const float alpha = sqrtf(w.cos_p * w.cos_p * alphas.x * alphas.x + w.sin_p * w.sin_p * alphas.y * alphas.y);
        float at2 = alpha * abs_tan;
        at2 *= at2;
        value = (-1.f + sqrtf(1.f + at2)) * 0.5f;
    }
    return value;
}
APT_D float trow_reitz_G1(f3 d, f3 alphas, f3 normal) { return 1.f / (1.f + trow_reitz_lambda(d, alphas, normal)); }
APT_D float trow_reitz_G(f3 in, f3 out, f3 alphas, f3 normal) { return 1.f / (1.f + trow_reitz_lambda(in, alphas, normal) + trow_reitz_lambda(out, alphas, normal)); }
APT_D void trow_reitz_slopes(float cos_theta, Philox& r, float& sx, float& sy) {          // microfacet.py:65-99 (two draws, always)
    const float u1 = rng_float(r);
    float u2 = rng_float(r);
    if (cos_theta > (float)(1.0 - 1e-5)) {
        const float rad = sqrtf(u1 / (1.f - u1)), phi = 6.28318530718f * u2;
        float sn, cs; apt_sincos(phi, &sn, &cs);
        sx = rad * cs; sy = rad * sn;
        return;
    }
    const float sin_theta = sqrtf(fmaxf(0.f, 1.f - cos_theta * cos_theta));
    const float tan_theta = sin_theta / cos_theta;
    const float G1 = 2.f / (1.f + sqrtf(1.f + tan_theta * tan_theta));
    const float A = 2.f * u1 / G1 - 1.f;
    const float tmp = fminf(1e10f, 1.f / (A * A - 1.f));
    const float D = sqrtf(fmaxf(tan_theta * tan_theta * tmp * tmp - (A * A - tan_theta * tan_theta) * tmp, 0.f));
    const float s1 = tan_theta * tmp - D, s2 = s1 + D * 2.f;
    const float slope_x = ((A < 0.f) || (s2 > 1.f / tan_theta)) ? s1 : s2;
    float S;
    if (u2 > 0.5f) { S = 1.f; u2 = 2.0f * (u2 - 0.5f); }
    else { S = -1.f; u2 = 2.f * (0.5f - u2); }
    const float z = (u2 * (u2 * (u2 * 0.27385f - 0.73369f) + 0.46341f)) / (u2 * (u2 * (u2 * 0.093073f + 0.309420f) - 1.0f) + 0.597999f);
    sx = slope_x; sy = S * z * sqrtf(1.f + slope_x * slope_x);
}
// half vector in the LOCAL frame (+y = normal) and its angles; note upstream stretches the WORLD-frame direction by (alpha_x, 1, alpha_y)
APT_D f3 trow_reitz_sample_wh(f3 incid, f3 normal, float ax, float ay, Philox& r, RawAngles& raw) {
    const bool flip = dot(incid, normal) > 0.f;
    const f3 wi = flip ? incid : -incid;
    const RawAngles w = to_raw(normalize(wi * mk3(ax, 1.f, ay)), normal);
    float sx, sy;
    trow_reitz_slopes(w.cos_t, r, sx, sy);
    const float tmp = w.cos_p * sx - w.sin_p * sy;
    sy = w.sin_p * sx + w.cos_p * sy;
    sx = tmp;
    sx = ax * sx; sy = ay * sy;
    f3 wh = normalize(mk3(-sx, 1.f, -sy));
    if (flip) wh = -wh;
    raw = raw_of_local(wh);
    return wh;
}
APT_D float trow_reitz_pdf(f3 incid, f3 wh, f3 alphas, f3 normal) {
    return trow_reitz_D(to_raw(wh, normal), alphas) * trow_reitz_G1(incid, alphas, normal) * fabsf(dot(wh, incid)) / fabsf(dot(normal, incid));
}
APT_D f3 microfacet_eval_raw(const DevBxdf& b, const Hit& it, f3 wh, const RawAngles& raw, f3 in, f3 out) {        // not yet / (4 cos_i cos_o)
    f3 ret = splat3(0.f);
    if (fabsf(wh.x) > BRDF_EPS || fabsf(wh.y) > BRDF_EPS || fabsf(wh.z) > BRDF_EPS) {
        wh = normalize(wh);
        const float fresnel = fresnel_eval(dot(wh, out), b.k_s.x, b.k_s.y);
        const float cosine_term = fabsf(dot(it.n_s, out));
        ret = (((b.k_d * trow_reitz_D(raw, b.k_g)) * trow_reitz_G(-in, out, b.k_g, it.n_s)) * fresnel) * cosine_term;
    }
    return ret;
}
APT_D f3 microfacet_eval(const DevBxdf& b, const Hit& it, f3 in, f3 out) {
    f3 ret = splat3(0.f);
    const float cos_mult = dot(it.n_s, out) * dot(it.n_s, in);
    if (cos_mult < 0.f) {
        const f3 wh = normalize(out - in);
        ret = microfacet_eval_raw(b, it, wh, to_raw(wh, it.n_s), in, out) / (-4.f * cos_mult);
    }
    return ret;
}
APT_D f3 microfacet_sample(const DevBxdf& b, const Hit& it, f3 incid, Philox& r, f3& spec, float& pdf) {
    RawAngles raw;
    const f3 local_wh = trow_reitz_sample_wh(incid, it.n_s, b.k_g.x, b.k_g.y, r, raw);
    const f3 half_vector = delocalize(it.n_s, local_wh);
    const float dot_val = -dot(incid, half_vector);
    f3 out_d = mk3(0.f, 1.f, 0.f);
    spec = splat3(0.f); pdf = 1.0f;
    if (dot_val > 0.f) {
        out_d = reflect_in(incid, half_vector);
        float cos_o = dot(it.n_s, out_d), cos_i = dot(it.n_s, incid);
        if (cos_o * cos_i < 0.f) {
            cos_i = fabsf(cos_i); cos_o = fabsf(cos_o);
            if (cos_o > BRDF_EPS && cos_i > BRDF_EPS) {
                spec = microfacet_eval_raw(b, it, half_vector, raw, incid, out_d) / (4.f * cos_o * cos_i);
                pdf = trow_reitz_pdf(-incid, half_vector, b.k_g, it.n_s);
                pdf /= 4.f * dot_val;
            }
        }
    }
    return out_d;
}

// Material-set masks: bit t (0..7) = BRDF type t present, bit 8 = det-refraction BSDF, bit 9 = Lambertian
// transmission BSDF, bit 10 = null BSDF.  Emitter mask: bit = emitter type (0 point, 1 area, 2 spot, 4 collimated).
// The dispatchers take the scene's mask as a template argument so that a shade kernel specialised for,
// say, {Lambertian} x {point} carries none of the other models' registers or code.
#define APT_BX_ALL 0x7ff
#define APT_SRC_ALL 0x17
#define BXHAS(M, bit) (((M) >> (bit)) & 1)

// BRDF.eval: f * cos, zero unless incid/out are on opposite sides of the GEOMETRIC normal
template <int BM>
APT_D f3 brdf_eval(const DevBxdf& b, const Hit& it, f3 incid, f3 out) {
    f3 ret = splat3(0.f);
    if (dot(incid, it.n_g) * dot(out, it.n_g) < 0.f) {
        switch (b.type) {
            case 0: if (BXHAS(BM, 0)) ret = blinn_phong_eval<BM>(b, it, incid, out); break;
            case 1: if (BXHAS(BM, 1)) ret = lambert_eval(b, it.n_s, out); break;
            case 4: if (BXHAS(BM, 4)) ret = mod_phong_eval(b, it, incid, out); break;
            case 5: if (BXHAS(BM, 5)) { m33 R; rotation_between(mk3(0.f, 1.f, 0.f), it.n_s, R); ret = fresnel_blend_eval(b, it, incid, out, R); } break;
            case 6: if (BXHAS(BM, 6)) ret = oren_nayar_eval(b, it, incid, out); break;
            case 7: if (BXHAS(BM, 7)) ret = thin_coat_eval(b, it, incid, out); break;
            case 3: if (BXHAS(BM, 3)) ret = microfacet_eval(b, it, incid, out); break;
            default: break;
        }
    }
    return ret;
}
template <int BM>
APT_D f3 brdf_sample(const DevBxdf& b, const Hit& it, f3 incid, Philox& r, f3& spec, float& pdf, bool& is_specular) {
    f3 dir = mk3(0.f, 1.f, 0.f);
    spec = splat3(1.f); pdf = 1.0f; is_specular = false;
    switch (b.type) {
        case 0: if (BXHAS(BM, 0)) {
            f3 local = sample_cosine_hemisphere(r, pdf);
            dir = delocalize(it.n_s, local);
            spec = blinn_phong_eval<BM>(b, it, incid, dir);
        } break;
        case 1: case 6: if (BXHAS(BM, 1) || BXHAS(BM, 6)) dir = lambert_sample(b, it.n_s, r, spec, pdf); break;
        case 2: if (BXHAS(BM, 2)) { dir = reflect_in(incid, it.n_s); spec = b.k_d; pdf = 1.0f; } break;
        case 7: if (BXHAS(BM, 7)) dir = thin_coat_sample(b, it, incid, r, spec, pdf, is_specular); break;
        case 4: if (BXHAS(BM, 4)) dir = mod_phong_sample(b, it, incid, r, spec, pdf); break;
        case 5: if (BXHAS(BM, 5)) dir = fresnel_blend_sample(b, it, incid, r, spec, pdf); break;
        case 3: if (BXHAS(BM, 3)) dir = microfacet_sample(b, it, incid, r, spec, pdf); break;
        default: break;
    }
    if (!(dot(dir, it.n_g) > 0.f)) spec = splat3(0.f);
    return dir;
}
template <int BM>
APT_D float brdf_pdf(const DevBxdf& b, const Hit& it, f3 outdir, f3 incid) {
    float pdf = 0.f;
    float d_out = dot(it.n_s, outdir);
    float d_in = dot(it.n_s, incid);
    if (d_out * d_in < 0.f) {
        switch (b.type) {
            case 0: case 1: case 6: pdf = d_out * APT_INV_PI; break;
            case 4: if (BXHAS(BM, 4)) {
                float g = b.mean.z;
                f3 rv = reflect_in(incid, it.n_s);
                float dro = fmaxf(0.f, dot(rv, outdir));
                float dpdf = d_out * APT_INV_PI;
                float spdf = 0.5f * (g + 1.f) * APT_INV_PI * apt_pow(dro, g);
                pdf = max3(b.k_d) * dpdf + max3(b.k_s) * spdf;
            } break;
            case 7: if (BXHAS(BM, 7)) {
                f3 refl = reflect_in(incid, it.n_s);
                float F = thin_coat_fresnel(b, it, incid);
                pdf = (fabsf(dot(outdir, refl)) > (1.f - 1e-3f)) ? F : (1.f - F) * d_out * APT_INV_PI;
            } break;
            case 5: if (BXHAS(BM, 5)) {
                f3 h = normalize(outdir - incid);
                float d_half = dot(h, it.n_s);
                m33 R; rotation_between(mk3(0.f, 1.f, 0.f), it.n_s, R);
                float c2, s2; fb_cos2_sin2(h, it.n_s, R, d_half, c2, s2);
                pdf = b.k_g.z * apt_pow(d_half, b.k_g.x * c2 + b.k_g.y * s2) / fabsf(dot(incid, h));
                pdf = 0.5f * (pdf + d_out * APT_INV_PI);
            } break;
            case 3: if (BXHAS(BM, 3)) {
                const f3 wh = normalize(outdir - incid);
                pdf = trow_reitz_pdf(-incid, wh, b.k_g, it.n_s) / (-4.f * dot(wh, incid));
            } break;
            default: break;
        }
    }
    return pdf;
}

// -------------------------------------------------------------------- BSDFs
APT_D f3 glass_sample(const DevBxdf& b, const Hit& it, f3 incid, float world_ior, Philox& r, f3& spec, float& pdf) {
    float dn = dot(incid, it.n_s);
    bool entering = dn < 0.f;
    float ni = entering ? world_ior : b.ior, nr = entering ? b.ior : world_ior;
    pdf = 1.0f;
    f3 dir;
    if (total_reflection(dn, ni, nr)) {
        dir = normalize(incid - (it.n_s * 2.f) * dn);
    } else {
        float cos_r2;
        f3 refra = refract_snell(incid, it.n_s, dn, ni, nr, cos_r2);
        float F = fresnel_dielectric(ni, nr, fabsf(dn), sqrtf(cos_r2));
        if (rng_float(r) > F) { pdf = 1.f - F; dir = refra; }
        else { dir = normalize(incid - (it.n_s * 2.f) * dn); pdf = F; }
    }
    spec = b.k_d * pdf;
    return dir;
}
APT_D f3 glass_eval(const DevBxdf& b, const Hit& it, f3 in, f3 out, float world_ior) {
    float d_out = dot(out, it.n_s);
    bool entering = d_out < 0.f;
    float ni = entering ? world_ior : b.ior, nr = entering ? b.ior : world_ior;
    f3 ret = splat3(0.f);
    f3 ref_dir = normalize(out - (it.n_s * 2.f) * d_out);
    if (total_reflection(d_out, ni, nr)) {
        if (dot(ref_dir, in) > 1.f - 5e-5f) ret = b.k_d;
    } else {
        float cos_r2;
        f3 refra = refract_snell(out, it.n_s, d_out, ni, nr, cos_r2);
        if (cos_r2 > 0.f) {
            float F = fresnel_dielectric(ni, nr, fabsf(d_out), sqrtf(cos_r2));
            if (dot(refra, in) > 1.f - 1e-4f) ret = b.k_d * (1.f - F);
            else if (dot(ref_dir, in) > 1.f - 1e-4f) ret = b.k_d * F;
        } else if (dot(ref_dir, in) > 1.f - 1e-4f) ret = b.k_d;
    }
    return ret;
}
APT_D f3 lambert_trans_sample(const DevBxdf& b, const Hit& it, f3 incid, float world_ior, Philox& r, f3& spec, float& pdf, bool& is_delta) {
    float dn = dot(incid, it.n_s);
    bool entering = dn < 0.f;
    float ni = entering ? world_ior : b.ior, nr = entering ? b.ior : world_ior;
    pdf = 1.0f;
    float fres = 1.0f;
    is_delta = true;
    f3 dir;
    f3 inten = b.k_d;
    if (total_reflection(dn, ni, nr)) {
        dir = normalize(incid - (it.n_s * 2.f) * dn);
    } else {
        float ratio = ni / nr;
        float cos_r2 = 1.f - sqr(ratio) * (1.f - sqr(dn));
        float F = fresnel_dielectric(ni, nr, fabsf(dn), sqrtf(cos_r2));
        if (rng_float(r) > F) {
            fres = 1.f - F;
            f3 local = sample_cosine_hemisphere(r, pdf);
            pdf *= fres;
            f3 n = it.n_s * sgn(dn);
            dir = delocalize(n, local);
            float c = fmaxf(0.f, dot(n, dir));
            inten = inten * (APT_INV_PI * c);
            is_delta = false;
        } else {
            dir = normalize(incid - (it.n_s * 2.f) * dn);
            fres = F; pdf = F;
        }
    }
    spec = inten * fres;
    return dir;
}
APT_D f3 lambert_trans_eval(const DevBxdf& b, const Hit& it, f3 in, f3 out, float world_ior) {
    float d_out = dot(out, it.n_s);
    bool entering = d_out < 0.f;
    float ni = entering ? world_ior : b.ior, nr = entering ? b.ior : world_ior;
    f3 ret = splat3(0.f);
    f3 ref_dir = normalize(out - (it.n_s * 2.f) * d_out);
    if (total_reflection(d_out, ni, nr)) {
        if (dot(ref_dir, in) > 1.f - 1e-4f) ret = b.k_d;
    } else {
        float ratio = ni / nr;
        float cos_r2 = 1.f - sqr(ratio) * (1.f - sqr(d_out));
        float d_in = dot(in, it.n_s);
        if (cos_r2 > 0.f) {
            float F = fresnel_dielectric(ni, nr, fabsf(d_out), sqrtf(cos_r2));
            if (d_in * d_out < 0.f) { if (dot(ref_dir, in) > 1.f - 1e-4f) ret = b.k_d * F; }
            else ret = b.k_d * ((1.f - F) * APT_INV_PI * fabsf(d_out));
        } else if (dot(ref_dir, in) > 1.f - 1e-4f) ret = b.k_d;
    }
    return ret;
}
APT_D float bsdf_pdf(const DevBxdf& b, const Hit& it, f3 outdir, f3 incid, float world_ior) {
    float pdf = 0.f;
    if (b.type == -1) return (dot(incid, outdir) > 1.f - 1e-4f) ? 1.f : 0.f;
    float d_out = dot(outdir, it.n_s);
    bool entering = d_out < 0.f;
    float ni = entering ? world_ior : b.ior, nr = entering ? b.ior : world_ior;
    f3 ref_dir = normalize(outdir - (it.n_s * 2.f) * d_out);
    float cos_r2;
    f3 refra = refract_snell(outdir, it.n_s, d_out, ni, nr, cos_r2);
    if (cos_r2 > 0.0f) {
        float F = fresnel_dielectric(ni, nr, fabsf(d_out), sqrtf(cos_r2));
        if (dot(ref_dir, incid) > 1.f - 1e-4f) pdf = F;
        else if (b.type == 0 && dot(refra, incid) > 1.f - 1e-4f) pdf = 1.f - F;
        else if (b.type == 1 && (dot(incid, it.n_s) * d_out > 0.f)) pdf = (1.f - F) * fabsf(d_out) * APT_INV_PI;
    } else if (dot(ref_dir, incid) > 1.f - 1e-4f) pdf = 1.f;
    return pdf;
}

// ---------------------------------------- surface dispatch (path_tracer.py:424-526)
APT_D void flip_if_two_sided(Hit& it, f3 incid, int two_sides) {
    if (two_sides && dot(incid, it.n_s) > 0.f) { it.n_s = -it.n_s; it.n_g = -it.n_g; }
}
template <int BM>
APT_D f3 surface_sample(const DevBxdf& b, Hit& it, f3 incid, float world_ior, int two_sides, Philox& r, f3& spec, float& pdf, bool& is_specular) {
    if (!(BM & 0x700) || !b.is_bsdf) { flip_if_two_sided(it, incid, two_sides); return brdf_sample<BM>(b, it, incid, r, spec, pdf, is_specular); }
    spec = splat3(0.f); pdf = 0.f; is_specular = false;
    if (BXHAS(BM, 8) && b.type == 0) return glass_sample(b, it, incid, world_ior, r, spec, pdf);
    if (BXHAS(BM, 9) && b.type == 1) return lambert_trans_sample(b, it, incid, world_ior, r, spec, pdf, is_specular);
    return splat3(0.f);
}
template <int BM>
APT_D f3 surface_eval(const DevBxdf& b, Hit& it, f3 incid, f3 out, float world_ior, int two_sides) {
    if (!(BM & 0x700) || !b.is_bsdf) { flip_if_two_sided(it, incid, two_sides); return brdf_eval<BM>(b, it, incid, out); }
    if (BXHAS(BM, 8) && b.type == 0) return glass_eval(b, it, incid, out, world_ior);
    if (BXHAS(BM, 9) && b.type == 1) return lambert_trans_eval(b, it, incid, out, world_ior);
    return splat3(0.f);
}
template <int BM>
APT_D float surface_pdf(const DevBxdf& b, Hit& it, f3 outdir, f3 incid, float world_ior, int two_sides) {
    if (!(BM & 0x700) || !b.is_bsdf) { flip_if_two_sided(it, incid, two_sides); return brdf_pdf<BM>(b, it, outdir, incid); }
    return bsdf_pdf(b, it, outdir, incid, world_ior);
}

// ------------------------------------------------------------------ emitters
struct EmitterGeom {        // what sample_hit needs from the attached object
    const float* precom;    // n_prims*9: (v1-v0, v2-v0, v0) | sphere (centre, rrr, centre)
    const float* normals;   // n_prims*3
    const int* obj_info;    // n_objects*3
};
APT_D f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }

// NEE sample: returns the point on the emitter; `inten` is already divided by the
// area->solid-angle pdf for area emitters (abtract_source.py:129-132)
template <int SM>
APT_D f3 emitter_sample_hit(const DevSrc& s, const EmitterGeom& g, f3 hit_pos, Philox& r, f3& inten, float& pdf) {
    inten = s.intensity;
    f3 pos = s.pos;
    pdf = 1.0f;
    if (BXHAS(SM, 0) && s.type == 0) {
        f3 x = hit_pos - pos;
        inten = inten * fminf(srcp(fmaxf(norm2(x), 1e-5f)), 1.0f);
    } else if (BXHAS(SM, 1) && s.type == 1) {
        pdf = s.inv_area;
        f3 normal;
        if (s.prim_count < 0) {
            const float* pc = g.precom + 9 * s.prim_first;
            f3 center = ld3(pc);
            float radius = pc[3];
            f3 to_hit = normalize(hit_pos - center);
            float p;
            f3 local = sample_uniform_sphere(r, p);
            normal = delocalize(to_hit, local);
            pos = center + normal * radius;
            pdf = sdiv(p, radius * radius);
        } else {
            int tri = pymod(rng_int(r), s.prim_count) + s.prim_first;
            normal = ld3(g.normals + 3 * tri);
            const float* pc = g.precom + 9 * tri;
            pos = sample_on_triangle(r, ld3(pc), ld3(pc + 3)) + ld3(pc + 6);
        }
        f3 diff = hit_pos - pos;
        float dl = dot(fnormalize(diff), normal);
        if (dl <= 0.0f) { inten = splat3(0.f); pdf = 1.0f; }
        else {
            pdf *= sdiv(norm2(diff), dl);
            inten = (pdf > 0.0f) ? fdiv3(inten, pdf) : splat3(0.f);
        }
    } else if (BXHAS(SM, 2) && s.type == 2) {
        f3 to_hit = hit_pos - pos;
        float depth = fmaxf(fnorm(to_hit), 1e-5f);
        to_hit = fdiv3(to_hit, depth);
        if (dot(to_hit, s.dir) > s.r) inten = fdiv3(inten, depth * depth);
        else inten = splat3(0.f);
    } else if (BXHAS(SM, 4) && s.type == 4) {
        pdf = 0.f;
        if (s.r > 0.f) {
            f3 to_hit = hit_pos - s.pos;
            float proj = dot(to_hit, s.dir);
            if (proj > 0.0f) {
                float dist = ssqrt(norm2(to_hit) - proj * proj);
                if (dist < s.r) pos = hit_pos - s.dir * proj;
                else inten = splat3(0.f);
            }
        } else inten = splat3(0.f);
    }
    return pos;
}
APT_D f3 emitter_eval_le(const DevSrc& s, f3 inci_dir, f3 normal) {
    if (s.type == 1 && -dot(normalize(inci_dir), normal) > 0.f) return s.intensity;
    return splat3(0.f);
}
APT_D float emitter_solid_angle_pdf(const DevSrc& s, const Hit& it, f3 incid_dir) {
    float d = fabsf(dot(incid_dir, it.n_s));
    float area_pdf = (s.type == 1) ? s.inv_area : 0.f;
    return (d > 0.0f) ? sdiv(area_pdf * sqr(it.min_depth), d) : 0.0f;
}
