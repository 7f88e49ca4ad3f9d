/*
 * rt_oracle.c -- CPU restatement of the trace/shade path of jrraymond/ray-tracer.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (ray-tracer_amd/, the
 * C-ABI library) may include, link or call this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY UNPINNED.  The reference is Haskell; no GHC exists in this image, the
 * reference ships no tests, no golden vectors and no recorded seeds
 * (SURVEY.md F2/F3), so this restatement is pinned only by (a) known-answer
 * tests derived by hand from the source (tests/test_oracle_kat.py), (b) the
 * reference's own debugging cross-check noBVH == sahBVH, and (c) a loose visual
 * comparison with the PNGs the reference ships.  Every function below cites the
 * reference lines it restates (paths relative to /root/reference).
 *
 * Arithmetic: IEEE binary32, one rounding per source-level operation, in the
 * source's association order; build with -O2 -ffp-contract=off -fno-fast-math
 * (GHC's x86-64 NCG emits SSE mulss/addss/divss/sqrtss, never FMA).  powf/expf
 * are the platform libm's, as GHC's `**`/`exp` on Float are.
 */
#include "rt_mi355x.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RTO_FLAG_CULL_ZERO_WEIGHT 0x1u /* mirror of the GPU's default ray culling: zero-weight bounce rays, shadow rays of lights with zero factors */

typedef struct rto_stats {
  uint64_t rays_primary, rays_shadow, rays_reflection, rays_refraction;
  uint64_t box_tests, sphere_tests, triangle_tests;
  uint64_t primary_hits;
  uint64_t nan_hits;          /* hit records whose t is NaN (tie rule differs) */
  uint64_t nonfinite_samples; /* samples whose colour has an inf/NaN channel   */
  uint64_t exit_branch;       /* getIndirectColor took the leaving-the-medium branch (dn >= 0, :140-145) */
  uint64_t tir;               /* ... and refract reported total internal reflection there (:144)          */
} rto_stats;

typedef struct { float x, y, z; } vec3;
typedef struct { float r, g, b; } color;

/* ---- Geometry3.hs:6-40 ---- */
static inline vec3 v3(float x, float y, float z) { vec3 v = {x, y, z}; return v; }
static inline vec3 add(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 subt(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 multiply(vec3 a, float f) { return v3(f * a.x, f * a.y, f * a.z); }
static inline vec3 divide(vec3 a, float f) { return v3(a.x / f, a.y / f, a.z / f); }
static inline float magnitude(vec3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
static inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline vec3 cross(vec3 a, vec3 b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline vec3 normalize(vec3 a) { return divide(a, magnitude(a)); }
static inline vec3 ld3(const float *p) { return v3(p[0], p[1], p[2]); }

/* ---- Surfaces.hs:7-13 ---- */
static inline color col(float r, float g, float b) { color c = {r, g, b}; return c; }
static inline color cadd(color a, color b) { return col(a.r + b.r, a.g + b.g, a.b + b.b); }
static inline color cmul(color a, color b) { return col(a.r * b.r, a.g * b.g, a.b * b.b); }
static inline color cscale(color a, float s) { return col(a.r * s, a.g * s, a.b * s); }
static inline color ldc(const float *p) { return col(p[0], p[1], p[2]); }

/* Ord Float: max x y = if x <= y then y else x ; min x y = if x <= y then x else y */
static inline float hs_max(float x, float y) { return (x <= y) ? y : x; }
static inline float hs_min(float x, float y) { return (x <= y) ? x : y; }

static const float EPSILON = 0.00000001f; /* RayTracer.hs:238-239 */

typedef struct {
  const rt_scene *sc;
  const rt_camera *cam;
  uint32_t flags;
  rto_stats st;
} octx;

/* ---- hitsBox, BoundingVolumeHierarchy.hs:191-204 ---- */
static int hits_box(vec3 b, vec3 d, const rt_bvh_node *n) {
  float l = n->xmin, r = n->xmax, bo = n->ymin, t = n->ymax, ne = n->zmin, f = n->zmax;
  float tx0, tx1, ty0, ty1, tz0, tz1;
  if (d.x >= 0) { tx0 = (l - b.x) / d.x; tx1 = (r - b.x) / d.x; }
  else          { tx0 = (r - b.x) / d.x; tx1 = (l - b.x) / d.x; }
  if (d.y >= 0) { ty0 = (bo - b.y) / d.y; ty1 = (t - b.y) / d.y; }
  else          { ty0 = (t - b.y) / d.y; ty1 = (bo - b.y) / d.y; }
  if (d.z >= 0) { tz0 = (ne - b.z) / d.z; tz1 = (f - b.z) / d.z; }
  else          { tz0 = (f - b.z) / d.z; tz1 = (ne - b.z) / d.z; }
  return !(tx0 > ty1 || tx0 > tz1 || tx1 < ty0 || tx1 < tz0 || ty0 > tz1 || ty1 < tz0);
}

/* ---- hit (sphere), RayTracer.hs:339-353.  Returns 1 and *t if Just. ---- */
static int hit_sphere(vec3 base, vec3 dir, const rt_sphere *s, float *t_out) {
  vec3 ec = subt(base, ld3(s->center));
  float radius = s->radius;
  float dec = dot(dir, ec);
  float dirdir = dot(dir, dir);
  float ecec = dot(ec, ec);
  float discriminant = dec * dec - dirdir * (ecec - radius * radius);
  float disc_val = sqrtf(discriminant);
  vec3 neg_dir = multiply(dir, -1.0f);
  float neg_dir_ec = dot(neg_dir, ec);
  float t_a = (neg_dir_ec + disc_val) / dirdir;
  float t_b = (neg_dir_ec - disc_val) / dirdir;
  float t = hs_min(t_a, t_b);
  if (discriminant > 0 && t > EPSILON) { *t_out = t; return 1; }
  return 0;
}

/* ---- hit (triangle), RayTracer.hs:354-380 ---- */
static int hit_triangle(vec3 rb, vec3 dir, const rt_triangle *tr, float *t_out) {
  float ax = tr->a[0], ay = tr->a[1], az = tr->a[2];
  float bx = tr->b[0], by = tr->b[1], bz = tr->b[2];
  float cx = tr->c[0], cy = tr->c[1], cz = tr->c[2];
  float g = dir.x, h = dir.y, i = dir.z;
  float theta = dot(dir, ld3(tr->n));
  float a = ax - bx, b = ay - by, c = az - bz;
  float d = ax - cx, e = ay - cy, f = az - cz;
  float j = ax - rb.x, k = ay - rb.y, l = az - rb.z;
  float ei = e * i, hf = h * f, gf = g * f, di = d * i, dh = d * h, eg = e * g;
  float ak = a * k, jb = j * b, jc = j * c, al = a * l, bl = b * l, kc = k * c;
  float ei_hf = ei - hf, gf_di = gf - di, dh_eg = dh - eg;
  float ak_jb = ak - jb, jc_al = jc - al, bl_kc = bl - kc;
  float m = a * ei_hf + b * gf_di + c * dh_eg;
  float beta = (j * ei_hf + k * gf_di + l * dh_eg) / m;
  float gamma = (i * ak_jb + h * jc_al + g * bl_kc) / m;
  float t = -((f * ak_jb + e * jc_al + d * bl_kc) / m);
  if (theta >= 0 || gamma < 0 || gamma > 1 || beta < 0 || beta + gamma > 1 || t < EPSILON) return 0;
  *t_out = t;
  return 1;
}

/* HitRec (RayTracer.hs:383-387) inside a Maybe */
typedef struct { int has; float t; uint32_t ref; } hitrec;

static int hit_ref(octx *o, vec3 base, vec3 dir, uint32_t ref, float *t) {
  if (ref & RT_REF_TRIANGLE) {
    o->st.triangle_tests++;
    return hit_triangle(base, dir, &o->sc->triangles[RT_REF_INDEX(ref)], t);
  }
  o->st.sphere_tests++;
  return hit_sphere(base, dir, &o->sc->spheres[ref], t);
}

/* ---- hits, RayTracer.hs:389-402: first Just seeds, strict `<` replaces ---- */
static hitrec hits(octx *o, vec3 base, vec3 dir, const uint32_t *refs, int n) {
  hitrec r = {0, 0.0f, 0};
  for (int q = 0; q < n; ++q) {
    float t;
    if (!hit_ref(o, base, dir, refs[q], &t)) continue;
    if (!r.has) { r.has = 1; r.t = t; r.ref = refs[q]; }
    else if (t < r.t) { r.t = t; r.ref = refs[q]; }
  }
  return r;
}

/* max on Maybe HitRec: Nothing < Just; Just a <= Just b  <=>  compare t_b t_a /= GT */
static hitrec max_hit(hitrec x, hitrec y) {
  int le;
  if (!x.has) le = 1;
  else if (!y.has) le = 0;
  else le = (y.t < x.t) || (y.t == x.t);
  return le ? y : x;
}

/* ---- hitBVH, RayTracer.hs:404-412 ---- */
static hitrec hit_bvh_node(octx *o, vec3 base, vec3 dir, int idx) {
  const rt_bvh_node *n = &o->sc->nodes[idx];
  hitrec none = {0, 0.0f, 0};
  o->st.box_tests++;
  if (!hits_box(base, dir, n)) return none;
  if (n->left < 0) return hits(o, base, dir, o->sc->leaf_refs + (-1 - n->left), n->right);
  hitrec a = hit_bvh_node(o, base, dir, n->left);
  hitrec b = hit_bvh_node(o, base, dir, n->right);
  return max_hit(a, b);
}
static hitrec hit_bvh(octx *o, vec3 base, vec3 dir) {
  hitrec none = {0, 0.0f, 0};
  if (o->sc->n_nodes == 0) return none; /* Empty */
  hitrec r = hit_bvh_node(o, base, dir, 0);
  if (r.has && r.t != r.t) o->st.nan_hits++;
  return r;
}

/* ---- inShadow, RayTracer.hs:231-235 ---- */
static int in_shadow(octx *o, vec3 pt, vec3 dir, float max_d) {
  o->st.rays_shadow++;
  hitrec r = hit_bvh(o, pt, dir);
  return r.has && r.t < max_d;
}

static const rt_material *material_of(const octx *o, uint32_t ref) {
  int m = (ref & RT_REF_TRIANGLE) ? o->sc->triangles[RT_REF_INDEX(ref)].material
                                  : o->sc->spheres[ref].material;
  return &o->sc->materials[m];
}

/* ---- getNormal, RayTracer.hs:75-78 ---- */
static vec3 get_normal(const octx *o, uint32_t ref, vec3 pt) {
  if (ref & RT_REF_TRIANGLE) return ld3(o->sc->triangles[RT_REF_INDEX(ref)].n);
  const rt_sphere *s = &o->sc->spheres[ref];
  return divide(subt(pt, ld3(s->center)), s->radius);
}

/* ---- getDirectColor, RayTracer.hs:81-106 (recursion over the light list) ---- */
static color direct_color(octx *o, int li, float ra, float rb, const rt_material *mt, vec3 pt,
                          vec3 n, vec3 v) {
  if (li == o->sc->n_lights) return cmul(ldc(o->cam->ambient), ldc(mt->diffuse)); /* :91 */
  const rt_light *L = &o->sc->lights[li];
  vec3 l_a = ld3(L->edge_a), l_b = ld3(L->edge_b);
  vec3 l_pt = add(ld3(L->corner), add(v3(l_a.x * ra, l_a.y * ra, l_a.z * ra),
                                      v3(l_b.x * rb, l_b.y * rb, l_b.z * rb)));
  vec3 l = subt(l_pt, pt);
  vec3 l_dir = normalize(l);
  float p = mt->phong;
  /* GPU default policy (flag only): a light whose diffuse AND specular factors are both exactly
   * zero -- max 0 (l.n) = 0 and max 0 (n.h) = 0 with a positive Phong exponent -- adds (0 + 0) to
   * the colour whether or not it is occluded, so its inShadow ray is neither traced nor counted.
   * The decision uses only the reference's own single-rounded dot products. */
  if ((o->flags & RTO_FLAG_CULL_ZERO_WEIGHT) && dot(l_dir, n) <= 0 &&
      dot(n, normalize(add(v, l_dir))) <= 0 && p > 0)
    return direct_color(o, li + 1, ra, rb, mt, pt, n, v);
  int shadowed = in_shadow(o, pt, l_dir, magnitude(l));
  color rest = direct_color(o, li + 1, ra, rb, mt, pt, n, v);
  if (shadowed) return rest;
  color l_c = ldc(L->color);
  float s = hs_max(0.0f, dot(l_dir, n));
  color diffuse_c = cscale(cmul(ldc(mt->diffuse), l_c), s);
  vec3 h = normalize(add(v, l_dir));
  float pw = powf(hs_max(0.0f, dot(n, h)), p);
  color spec_c = cscale(cmul(l_c, ldc(mt->specular)), pw);
  color c = cadd(diffuse_c, spec_c);
  return cadd(c, rest);
}

/* ---- reflect / refract / orthonormal, RayTracer.hs:162-179, 453-459 ---- */
static vec3 reflect(vec3 dir, vec3 n) { return subt(dir, multiply(n, 2 * dot(dir, n))); }

static int refract(vec3 dir, vec3 normal, float nt, vec3 *t_out) {
  const float n = 1;
  float dn = dot(dir, normal);
  float x = 1 - n * n * (1 - dn * dn) / nt / nt;
  if (x < 0) return 0;
  vec3 y = multiply(subt(dir, multiply(normal, dn)), n / nt);
  *t_out = normalize(subt(y, multiply(normal, sqrtf(x))));
  return 1;
}

static void orthonormal(vec3 w, vec3 *u, vec3 *v) {
  vec3 t = (w.x == 1 && w.y == 0 && w.z == 0) ? v3(0, 1, 0) : v3(1, 0, 0);
  *u = normalize(cross(t, w));
  *v = cross(w, *u);
}

enum { KIND_PRIMARY = 0, KIND_REFLECTION = 1, KIND_REFRACTION = 2 };

static color raytrace(octx *o, int depth, float s0, float s1, vec3 base, vec3 dir, int kind,
                      color wt, uint32_t *hit_id);

/* ---- getIndirectColor, RayTracer.hs:109-156 ---- */
static color indirect_color(octx *o, int depth, float s0, float s1, vec3 dir, vec3 pt, vec3 normal,
                            const rt_material *mt, color wt) {
  int cull = (o->flags & RTO_FLAG_CULL_ZERO_WEIGHT) != 0;
  vec3 refl0 = reflect(dir, normal);
  vec3 u, v;
  orthonormal(refl0, &u, &v);
  float a = 1 / mt->phong;
  float ju = -(a / 2) + s0 * a;
  float jv = -(a / 2) + s1 * a;
  vec3 refl_v = add(refl0, add(multiply(u, ju), multiply(v, jv)));
  float nt = mt->refr_ix;
  color spec = ldc(mt->specular);

  /* The accumulated weights (wt) are tracked ONLY to decide which rays the GPU's
   * default policy culls; they never enter the colour arithmetic below.  Their
   * formulas mirror ray-tracer_amd/csrc/rt_kernels.inl (shade step). */
  color f_refl = cscale(spec, mt->refl);
  color k = col(1, 1, 1);
  float c = 0, bigR = 0;
  vec3 tdir = v3(0, 0, 0);
  int has_t = 0;
  if (nt != 0) {
    float dn = dot(dir, normal);
    if (dn < 0) {
      c = -dn;
      has_t = refract(dir, normal, nt, &tdir);
    } else {
      has_t = refract(dir, multiply(normal, -1), 1 / nt, &tdir);
      o->st.exit_branch++;
      if (!has_t) o->st.tir++;
      const float *at = mt->log_atten;
      k = col(expf(-at[0]), expf(-at[1]), expf(-at[2]));
      c = has_t ? dot(tdir, normal) : 0;
    }
    float r_0 = (nt - 1) * (nt - 1) / ((nt + 1) * (nt + 1));
    float om = 1 - c;
    bigR = r_0 + (1 - r_0) * (om * om * om * om * om); /* ^(5::Int), GHC.Real rule "^5/Int" */
    if (has_t) f_refl = cmul(cscale(f_refl, bigR), k);
    else f_refl = cmul(f_refl, k);
  }
  color w_refl = cmul(wt, f_refl);
  color child = col(0, 0, 0);
  if (!(cull && w_refl.r == 0 && w_refl.g == 0 && w_refl.b == 0))
    child = raytrace(o, depth - 1, s0, s1, pt, refl_v, KIND_REFLECTION, w_refl, NULL);
  color refl_c = cscale(cmul(spec, child), mt->refl); /* :129-131 */
  if (nt == 0) return refl_c;
  if (!has_t) return cmul(k, refl_c); /* :151 */
  color w_refr = cmul(wt, cscale(k, 1 - bigR));
  color rchild = col(0, 0, 0);
  if (!(cull && w_refr.r == 0 && w_refr.g == 0 && w_refr.b == 0))
    rchild = raytrace(o, depth - 1, s0, s1, pt, tdir, KIND_REFRACTION, w_refr, NULL);
  color refl_c2 = cscale(refl_c, bigR);
  color refr_c2 = cscale(rchild, 1 - bigR);
  return cmul(k, cadd(refl_c2, refr_c2)); /* :155 */
}

/* ---- raytrace, RayTracer.hs:55-71 ---- */
static color raytrace(octx *o, int depth, float s0, float s1, vec3 base, vec3 dir, int kind,
                      color wt, uint32_t *hit_id) {
  if (hit_id) *hit_id = RT_HIT_NONE;
  if (depth == 0) return col(0, 0, 0);
  if (kind == KIND_PRIMARY) o->st.rays_primary++;
  else if (kind == KIND_REFLECTION) o->st.rays_reflection++;
  else o->st.rays_refraction++;
  hitrec hr = hit_bvh(o, base, dir);
  if (!hr.has) return col(0, 0, 0);
  if (hit_id) {
    *hit_id = (hr.ref & RT_REF_TRIANGLE) ? (uint32_t)o->sc->triangles[RT_REF_INDEX(hr.ref)].object_id
                                          : (uint32_t)o->sc->spheres[hr.ref].object_id;
    o->st.primary_hits++;
  }
  const rt_material *mt = material_of(o, hr.ref);
  vec3 pt = add(base, multiply(dir, hr.t));
  vec3 n = get_normal(o, hr.ref, pt);
  vec3 v = v3(-dir.x, -dir.y, -dir.z);
  color direct_c = direct_color(o, 0, s0, s1, mt, pt, n, v);
  color indirect_c = indirect_color(o, depth, s0, s1, dir, pt, n, mt, wt);
  return cadd(direct_c, indirect_c);
}

/* ---- getRay, RayTracer.hs:183-205 ---- */
static void get_ray(const rt_camera *w, float i, float j, float r1, float r2, vec3 *base_out,
                    vec3 *dir_out) {
  float lens = w->lens;
  vec3 eye = ld3(w->eye), u = ld3(w->u), v = ld3(w->v), ww = ld3(w->w);
  float r1l = r1 * lens, r2l = r2 * lens, l2 = lens / 2;
  vec3 ru = subt(v3(u.x * r1l, u.y * r1l, u.z * r1l), v3(u.x * l2, u.y * l2, u.z * l2));
  vec3 rv = subt(v3(v.x * r2l, v.y * r2l, v.z * r2l), v3(v.x * l2, v.y * l2, v.z * l2));
  vec3 base = add(eye, add(ru, rv));
  float vWd = w->view_w, vHt = w->view_h, vDt = w->view_d;
  float iWd = (float)w->img_w, iHt = (float)w->img_h;
  float u_world = i * vWd / iWd - vWd / 2.0f;
  float v_world = j * vHt / iHt - vHt / 2.0f;
  float w_world = -vDt;
  vec3 u_dir = multiply(u, u_world);
  vec3 v_dir = multiply(v, v_world);
  vec3 w_dir = multiply(ww, w_world);
  vec3 dir = add(w_dir, add(u_dir, v_dir));
  vec3 fdir = subt(add(dir, eye), base);
  *base_out = base;
  *dir_out = normalize(fdir);
}

static void stats_add(rto_stats *a, const rto_stats *b) {
  uint64_t *pa = (uint64_t *)a;
  const uint64_t *pb = (const uint64_t *)b;
  for (size_t q = 0; q < sizeof(rto_stats) / sizeof(uint64_t); ++q) pa[q] += pb[q];
}

/* ======================================================================
 * Exported API (called through ctypes by tests / bench cpu_baseline).
 * ====================================================================== */

/* render + colorPixel + avgColors (RayTracer.hs:18-23, 47-51; Surfaces.hs:15-18) for
 * the output rows [row0,row1) (row 0 = top = j = H-1).  rgb / primary_hit_ids are
 * whole-frame buffers; only the selected rows are written. */
int rto_render(const rt_scene *sc, const rt_camera *cam, const float *f6, int n_grids, int spp,
               uint32_t flags, int row0, int row1, int n_threads, float *rgb,
               uint32_t *primary_hit_ids, rto_stats *stats) {
  if (!sc || !cam || !f6 || !rgb || n_grids <= 0 || spp <= 0) return RT_ERR_INVALID;
  int W = cam->img_w, H = cam->img_h;
  if (row0 < 0 || row1 > H || row0 > row1) return RT_ERR_INVALID;
  rto_stats total;
  memset(&total, 0, sizeof total);
  if (n_threads < 1) n_threads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads)
#endif
  {
    octx o;
    o.sc = sc;
    o.cam = cam;
    o.flags = flags;
    memset(&o.st, 0, sizeof o.st);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (int row = row0; row < row1; ++row) {
      float j = (float)(H - 1 - row); /* j <- reverse [0..H-1] */
      for (int x = 0; x < W; ++x) {
        float i = (float)x;
        size_t kpix = (size_t)row * W + x; /* position in render's list */
        const float *grid = f6 + (kpix % (size_t)n_grids) * (size_t)spp * 6; /* zip ps (cycle grids) */
        color sum = col(0, 0, 0);
        for (int s = 0; s < spp; ++s) {
          const float *F = grid + s * 6; /* F6 p q r0 r1 s0 s1 */
          vec3 base, dir;
          get_ray(cam, i + F[0], j + F[1], F[2], F[3], &base, &dir);
          uint32_t hid;
          color c = raytrace(&o, cam->max_depth, F[4], F[5], base, dir, KIND_PRIMARY, col(1, 1, 1),
                             &hid);
          if (primary_hit_ids) primary_hit_ids[kpix * spp + s] = hid;
          if (!isfinite(c.r) || !isfinite(c.g) || !isfinite(c.b)) o.st.nonfinite_samples++;
          sum = cadd(sum, c);
        }
        float nf = (float)spp;
        rgb[kpix * 3 + 0] = sum.r / nf;
        rgb[kpix * 3 + 1] = sum.g / nf;
        rgb[kpix * 3 + 2] = sum.b / nf;
      }
    }
#ifdef _OPENMP
#pragma omp critical
#endif
    stats_add(&total, &o.st);
  }
  if (stats) *stats = total;
  return RT_OK;
}

/* ---- single-function probes for the known-answer tests ---- */
int rto_hit_sphere(const float base[3], const float dir[3], const rt_sphere *s, float *t) {
  return hit_sphere(ld3(base), ld3(dir), s, t);
}
int rto_hit_triangle(const float base[3], const float dir[3], const rt_triangle *tr, float *t) {
  return hit_triangle(ld3(base), ld3(dir), tr, t);
}
int rto_hits_box(const float base[3], const float dir[3], const rt_bvh_node *n) {
  return hits_box(ld3(base), ld3(dir), n);
}
void rto_reflect(const float dir[3], const float n[3], float out[3]) {
  vec3 r = reflect(ld3(dir), ld3(n));
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
int rto_refract(const float dir[3], const float n[3], float nt, float out[3]) {
  vec3 t = v3(0, 0, 0);
  int ok = refract(ld3(dir), ld3(n), nt, &t);
  out[0] = t.x; out[1] = t.y; out[2] = t.z;
  return ok;
}
void rto_orthonormal(const float w[3], float u[3], float v[3]) {
  vec3 uu, vv;
  orthonormal(ld3(w), &uu, &vv);
  u[0] = uu.x; u[1] = uu.y; u[2] = uu.z;
  v[0] = vv.x; v[1] = vv.y; v[2] = vv.z;
}
void rto_get_ray(const rt_camera *cam, float i, float j, float r1, float r2, float base[3],
                 float dir[3]) {
  vec3 b, d;
  get_ray(cam, i, j, r1, r2, &b, &d);
  base[0] = b.x; base[1] = b.y; base[2] = b.z;
  dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
}
/* closest hit through the tree: returns 1 + (*t, *object_id) or 0 */
int rto_hit_bvh(const rt_scene *sc, const float base[3], const float dir[3], float *t,
                uint32_t *object_id) {
  octx o;
  o.sc = sc; o.cam = NULL; o.flags = 0;
  memset(&o.st, 0, sizeof o.st);
  hitrec r = hit_bvh(&o, ld3(base), ld3(dir));
  if (!r.has) return 0;
  *t = r.t;
  *object_id = (r.ref & RT_REF_TRIANGLE) ? (uint32_t)sc->triangles[RT_REF_INDEX(r.ref)].object_id
                                         : (uint32_t)sc->spheres[r.ref].object_id;
  return 1;
}
/* colour of one ray (raytrace), for shading KATs */
void rto_raytrace(const rt_scene *sc, const rt_camera *cam, int depth, float s0, float s1,
                  const float base[3], const float dir[3], uint32_t flags, float rgb[3]) {
  octx o;
  o.sc = sc; o.cam = cam; o.flags = flags;
  memset(&o.st, 0, sizeof o.st);
  color c = raytrace(&o, depth, s0, s1, ld3(base), ld3(dir), KIND_PRIMARY, col(1, 1, 1), NULL);
  rgb[0] = c.r; rgb[1] = c.g; rgb[2] = c.b;
}
int rto_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
