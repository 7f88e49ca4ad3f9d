// rt_kernels.inl -- the trace/shade path as one persistent, wave-autonomous HIP kernel
// for gfx950.  Included by rt_kernels_strict.hip (-ffp-contract=off) and
// rt_kernels_fma.hip (-ffp-contract=fast) with RT_NS set.
//
// Execution model (DESIGN.md has the long form):
//   * one 64-lane wave owns one image tile (~1024 samples) at a time and pulls tile
//     indices from a global atomic counter until none are left;
//   * lane = ray.  Primary rays are generated in registers; secondary rays
//     (reflection / refraction, RayTracer.hs:109-156) are pushed, wave-compacted with
//     ballot + popcount, into per-level ray queues in the wave's own HBM scratch
//     (SoA, coalesced), and popped 64 at a time, deepest full level first, so every
//     bounce runs with (almost) full waves and queue memory stays bounded
//     (<= 191 rays per level);
//   * BVH traversal (hitBVH/hits/hitsBox) keeps its per-lane stack in LDS;
//   * recursion is flattened to weight-carrying rays: colour(sample) =
//     sum over its rays of weight * direct colour; per-sample sums live in the wave's
//     scratch and are resolved to the pixel average (avgColors) at tile end.
//
// Every arithmetic expression that decides a hit is written in the reference's
// operand order; see the file:line notes.  Paths are relative to /root/reference.

#include "rt_device.h"

namespace rtdev {
namespace RT_NS {

struct V3 {
  float x, y, z;
};
#define DEVI __device__ __forceinline__

// Geometry3.hs:6-40
DEVI V3 mk(float x, float y, float z) { V3 v = {x, y, z}; return v; }
DEVI V3 add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DEVI V3 subt(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DEVI V3 multiply(V3 a, float f) { return mk(f * a.x, f * a.y, f * a.z); }
DEVI V3 divide(V3 a, float f) { return mk(a.x / f, a.y / f, a.z / f); }
DEVI float magnitude(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
DEVI float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEVI V3 cross(V3 a, V3 b) {
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
DEVI V3 normalize(V3 a) { return divide(a, magnitude(a)); }
DEVI V3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }
// Surfaces.hs:7-13 (colours reuse V3)
DEVI V3 cmul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
DEVI V3 cscale(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
// Ord Float
DEVI float hs_max(float x, float y) { return (x <= y) ? y : x; }
DEVI float hs_min(float x, float y) { return (x <= y) ? x : y; }

constexpr float EPSILON = 0.00000001f;  // RayTracer.hs:238-239
constexpr int STOP = INT32_MIN + 1;     // traversal finished (distinct from REF_NONE)

struct Counters {
  unsigned box, sph, tri;
};

// hitsBox (BoundingVolumeHierarchy.hs:191-204): six IEEE divisions, slab ends
// swapped by the sign test `d >= 0`, six interval comparisons.  t0/t1 out for the
// (result-preserving) range pruning.
DEVI bool hits_box(V3 b, V3 d, float l, float r, float bo, float t, float n, float f, float &e0,
                   float &e1, float &e2, float &x0, float &x1, float &x2) {
  float tx0 = ((d.x >= 0 ? l : r) - b.x) / d.x;
  float tx1 = ((d.x >= 0 ? r : l) - b.x) / d.x;
  float ty0 = ((d.y >= 0 ? bo : t) - b.y) / d.y;
  float ty1 = ((d.y >= 0 ? t : bo) - b.y) / d.y;
  float tz0 = ((d.z >= 0 ? n : f) - b.z) / d.z;
  float tz1 = ((d.z >= 0 ? f : n) - b.z) / d.z;
  e0 = tx0; e1 = ty0; e2 = tz0;
  x0 = tx1; x1 = ty1; x2 = tz1;
  return !(tx0 > ty1 || tx0 > tz1 || tx1 < ty0 || tx1 < tz0 || ty0 > tz1 || ty1 < tz0);
}

// hit, sphere (RayTracer.hs:339-353); q0 = (C, r), r2 = r*r
DEVI bool hit_sphere(V3 base, V3 dir, float4 q0, float r2, float &t_out) {
  V3 ec = subt(base, mk(q0.x, q0.y, q0.z));
  float dec = dot(dir, ec);
  float dirdir = dot(dir, dir);
  float ecec = dot(ec, ec);
  float discriminant = dec * dec - dirdir * (ecec - r2);
  float disc_val = sqrtf(discriminant);
  V3 neg_dir = multiply(dir, -1.0f);
  float neg_dir_ec = dot(neg_dir, ec);
  float t_a = (neg_dir_ec + disc_val) / dirdir;
  float t_b = (neg_dir_ec - disc_val) / dirdir;
  float t = hs_min(t_a, t_b);
  t_out = t;
  return discriminant > 0 && t > EPSILON;
}

// hit, triangle (RayTracer.hs:354-380); A, A-B, A-C, N precomputed at upload
// (the same single-rounded subtractions the reference performs per test).
DEVI bool hit_triangle(V3 rb, V3 dir, float4 q0, float4 q1, float4 q2, float &t_out) {
  float ax = q0.x, ay = q0.y, az = q0.z;
  float a = q0.w, b = q1.x, c = q1.y;
  float d = q1.z, e = q1.w, f = q2.x;
  V3 tn = mk(q2.y, q2.z, q2.w);
  float g = dir.x, h = dir.y, i = dir.z;
  float theta = dot(dir, tn);
  float j = ax - rb.x, k = ay - rb.y, l = az - rb.z;
  float ei = e * i, hf = h * f, gf = g * f, di = d * i, dh = d * h, eg = e * g;
  float ak = a * k, jb = j * b, jc = j * c, al = a * l, bl = b * l, kc = k * c;
  float ei_hf = ei - hf, gf_di = gf - di, dh_eg = dh - eg;
  float ak_jb = ak - jb, jc_al = jc - al, bl_kc = bl - kc;
  float m = a * ei_hf + b * gf_di + c * dh_eg;
  float beta = (j * ei_hf + k * gf_di + l * dh_eg) / m;
  float gamma = (i * ak_jb + h * jc_al + g * bl_kc) / m;
  float t = -((f * ak_jb + e * jc_al + d * bl_kc) / m);
  t_out = t;
  return !(theta >= 0 || gamma < 0 || gamma > 1 || beta < 0 || beta + gamma > 1 || t < EPSILON);
}

// Range-pruning margins (only with PRUNE).  A child is skipped when its slab entry
// lies beyond the current closest hit (or the light, for shadow rays) by more than
// this, or when the box ends behind the origin by more than this.  The boxes bound
// their primitives exactly, so this can only change a result whose hit distance is
// itself off by more than the margin (ill-conditioned grazing hits); the parity
// tests count such cases.
DEVI float prune_limit(float t) { return t + (fabsf(t) * 0.000244140625f + 0.000244140625f); }
constexpr float BEHIND = -0.000244140625f;

// hitBVH + hits (RayTracer.hs:389-412), stack-based.  ANY = inShadow's question
// "is the closest hit nearer than tlimit" answered by the first hit with t < tlimit.
// Tie rule of the closest-hit form: t < best, or t == best and a larger record
// index (see rt_device.h).  NaN distances never win (documented deviation: the
// reference's Ord instance lets a NaN stick in two corner cases).
template <bool ANY, bool PRUNE, bool COUNT>
DEVI void traverse(const RenderParams &P, unsigned *stack, int lane, bool active, V3 o, V3 d,
                   float tlimit, float &best_t, int &best_p, Counters &cn) {
  best_t = ANY ? tlimit : INFINITY;
  best_p = -1;
  int sp = 0;
  int cur = (active && P.has_root) ? 0 : STOP;
  while (cur != STOP) {
    if (cur >= 0) {
      const float4 *np = P.nodes + (size_t)cur * NODE_F4;
      float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
      int lref = __float_as_int(n3.x), rref = __float_as_int(n3.y);
      float le0, le1, le2, lx0, lx1, lx2, re0, re1, re2, rx0, rx1, rx2;
      bool hl = hits_box(o, d, n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, le0, le1, le2, lx0, lx1, lx2);
      bool hr = false;
      if (COUNT) cn.box += 1;
      if (rref != REF_NONE) {
        hr = hits_box(o, d, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, re0, re1, re2, rx0, rx1, rx2);
        if (COUNT) cn.box += 1;
      }
      bool left_first = true;
      if (PRUNE) {
        float lim = prune_limit(best_t);
        if (hl && (le0 > lim || le1 > lim || le2 > lim || lx0 < BEHIND || lx1 < BEHIND || lx2 < BEHIND))
          hl = false;
        if (hr && (re0 > lim || re1 > lim || re2 > lim || rx0 < BEHIND || rx1 < BEHIND || rx2 < BEHIND))
          hr = false;
        if (hl && hr) {
          float el = fmaxf(fmaxf(le0, le1), le2), er = fmaxf(fmaxf(re0, re1), re2);
          left_first = !(er < el);
        }
      }
      if (hl && hr) {
        int first = left_first ? lref : rref, second = left_first ? rref : lref;
        stack[sp * 64 + lane] = (unsigned)second;
        ++sp;
        cur = first;
      } else if (hl) {
        cur = lref;
      } else if (hr) {
        cur = rref;
      } else if (sp > 0) {
        --sp;
        cur = (int)stack[sp * 64 + lane];
      } else {
        cur = STOP;
      }
    } else {
      // leaf: records first .. first+count-1 (reversed list order)
      int first = -1 - cur;
      const float4 *pp = P.prims + (size_t)first * PRIM_F4;
      int count = __float_as_int(pp[3].w);
      bool done = false;
      for (int k = 0; k < count; ++k) {
        const float4 *q = pp + (size_t)k * PRIM_F4;
        float4 q0 = q[0], q1 = q[1], q3 = q[3];
        float t;
        bool ok;
        if (__float_as_int(q3.x) != 0) {
          float4 q2 = q[2];
          ok = hit_triangle(o, d, q0, q1, q2, t);
          if (COUNT) cn.tri += 1;
        } else {
          ok = hit_sphere(o, d, q0, q1.x, t);
          if (COUNT) cn.sph += 1;
        }
        if (ok) {
          if (ANY) {
            if (t < tlimit) {
              best_t = t;
              best_p = first + k;
              done = true;
              break;
            }
          } else if (t < best_t || (t == best_t && first + k > best_p)) {
            best_t = t;
            best_p = first + k;
          }
        }
      }
      if (done) {
        cur = STOP;
      } else if (sp > 0) {
        --sp;
        cur = (int)stack[sp * 64 + lane];
      } else {
        cur = STOP;
      }
    }
  }
}

// reflect / refract / orthonormal (RayTracer.hs:162-179, 453-459)
DEVI V3 reflect(V3 dir, V3 n) { return subt(dir, multiply(n, 2 * dot(dir, n))); }
DEVI bool refract(V3 dir, V3 normal, float nt, V3 &t_out) {
  const float n = 1;
  float dn = dot(dir, normal);
  float x = 1 - n * n * (1 - dn * dn) / nt / nt;
  if (x < 0) return false;
  V3 y = multiply(subt(dir, multiply(normal, dn)), n / nt);
  t_out = normalize(subt(y, multiply(normal, sqrtf(x))));
  return true;
}
DEVI void orthonormal(V3 w, V3 &u, V3 &v) {
  V3 t = (w.x == 1 && w.y == 0 && w.z == 0) ? mk(0, 1, 0) : mk(1, 0, 0);
  u = normalize(cross(t, w));
  v = cross(w, u);
}

// getRay (RayTracer.hs:183-205)
DEVI void get_ray(const RenderParams &P, float i, float j, float r1, float r2, V3 &base, V3 &dirn) {
  float lens = P.lens;
  V3 eye = ld3(P.eye), u = ld3(P.u), v = ld3(P.v), ww = ld3(P.w);
  float r1l = r1 * lens, r2l = r2 * lens, l2 = lens / 2;
  V3 ru = subt(mk(u.x * r1l, u.y * r1l, u.z * r1l), mk(u.x * l2, u.y * l2, u.z * l2));
  V3 rv = subt(mk(v.x * r2l, v.y * r2l, v.z * r2l), mk(v.x * l2, v.y * l2, v.z * l2));
  base = add(eye, add(ru, rv));
  float vWd = P.view_w, vHt = P.view_h, vDt = P.view_d;
  float iWd = (float)P.W, iHt = (float)P.H;
  float u_world = i * vWd / iWd - vWd / 2.0f;
  float v_world = j * vHt / iHt - vHt / 2.0f;
  float w_world = -vDt;
  V3 u_dir = multiply(u, u_world);
  V3 v_dir = multiply(v, v_world);
  V3 w_dir = multiply(ww, w_world);
  V3 dir = add(w_dir, add(u_dir, v_dir));
  V3 fdir = subt(add(dir, eye), base);
  dirn = normalize(fdir);
}

DEVI int wave_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }

constexpr unsigned SLOT_REFRACTION = 0x80000000u;

template <bool PRUNE, bool COUNT>
__global__ __launch_bounds__(256) void render_kernel(const RenderParams P) {
  extern __shared__ unsigned lds[];
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  const unsigned long long gwave = (unsigned long long)blockIdx.x * WAVES_PER_BLOCK + wib;
  unsigned *stack = lds + (size_t)wib * (P.stack_depth * 64 + TAG_SLOTS + 32);
  unsigned *tag = stack + P.stack_depth * 64;
  unsigned *cnt = tag + TAG_SLOTS;  // cnt[L], L = 2..max_depth : rays queued at level L

  float *Q = P.scratch + gwave * P.scratch_stride;
  const int n_qlevels = P.max_depth > 1 ? P.max_depth - 1 : 0;
  float *acc = Q + (size_t)n_qlevels * QFIELDS * QCAP;  // [3][tile_samples]
  const int TS = P.tile_samples;
  const int tile_px = P.tile_w * P.tile_h;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

  unsigned long long n_primary = 0, n_shadow = 0, n_refl = 0, n_refr = 0, n_phits = 0, n_hits = 0;
  Counters cn = {0, 0, 0};

  for (;;) {
    // ---- fetch a tile -------------------------------------------------------
    int ltile = 0;
    if (lane == 0) ltile = (int)atomicAdd(P.tile_counter, 1u);
    ltile = wave_uniform(ltile);
    if (ltile >= P.n_local_tiles) break;
    const int gtile = P.rank + ltile * P.nranks;
    const int tx0 = (gtile % P.tiles_x) * P.tile_w;
    const int ty0 = (gtile / P.tiles_x) * P.tile_h;

    for (int q = lane; q < 3 * TS; q += 64) acc[q] = 0.0f;
    for (int q = lane; q < TAG_SLOTS; q += 64) tag[q] = 0u;
    if (lane < 32) cnt[lane] = 0u;
    unsigned epoch = 0;
    int next_sample = 0;

    for (;;) {
      __threadfence_block();  // own-wave queue/acc stores -> visible to the loads below
      // ---- choose the level to run: deepest full queue, else new primaries, else
      //      deepest non-empty queue --------------------------------------------
      int level = 0;
      for (int L = P.max_depth; L >= 2; --L)
        if ((int)cnt[L] >= 64) { level = L; break; }
      if (level == 0) {
        if (next_sample < TS && P.max_depth >= 1) level = 1;
        else
          for (int L = P.max_depth; L >= 2; --L)
            if ((int)cnt[L] > 0) { level = L; break; }
      }
      level = wave_uniform(level);
      if (level == 0) break;

      // ---- obtain 64 (or fewer) rays ---------------------------------------------
      bool active;
      V3 o, d, W;
      float s0, s1;
      unsigned slot;
      if (level == 1) {
        int q = next_sample + lane;
        next_sample += 64;
        int pit = q / P.spp, s = q - pit * P.spp;
        int px = tx0 + pit % P.tile_w, py = ty0 + pit / P.tile_w;
        active = (q < TS) && (px < P.W) && (py < P.H);
        slot = (unsigned)q;
        W = mk(1.0f, 1.0f, 1.0f);
        o = mk(0, 0, 0); d = mk(0, 0, 1); s0 = 0; s1 = 0;
        if (active) {
          // render (RayTracer.hs:18-23): list position k uses grid k mod n_grids
          size_t kpix = (size_t)py * P.W + px;
          const float *F = P.f6 + ((kpix % (size_t)P.n_grids) * (size_t)P.spp + s) * 6;
          float fi = (float)px, fj = (float)(P.H - 1 - py);
          get_ray(P, fi + F[0], fj + F[1], F[2], F[3], o, d);  // colorPixel :47-51
          s0 = F[4];
          s1 = F[5];
        }
        n_primary += __popcll(__ballot(active));
      } else {
        int have = (int)cnt[level];
        int take = have < 64 ? have : 64;
        int idx = have - take + lane;
        active = lane < take;
        const float *ql = Q + (size_t)(level - 2) * QFIELDS * QCAP;
        o = mk(0, 0, 0); d = mk(0, 0, 1); W = mk(0, 0, 0); s0 = 0; s1 = 0; slot = 0;
        if (active) {
          o = mk(ql[0 * QCAP + idx], ql[1 * QCAP + idx], ql[2 * QCAP + idx]);
          d = mk(ql[3 * QCAP + idx], ql[4 * QCAP + idx], ql[5 * QCAP + idx]);
          W = mk(ql[6 * QCAP + idx], ql[7 * QCAP + idx], ql[8 * QCAP + idx]);
          s0 = ql[9 * QCAP + idx];
          s1 = ql[10 * QCAP + idx];
          slot = __float_as_uint(ql[11 * QCAP + idx]);
        }
        if (lane == 0) cnt[level] = (unsigned)(have - take);
        unsigned long long am = __ballot(active);
        unsigned long long rm = __ballot(active && (slot & SLOT_REFRACTION));
        n_refr += __popcll(rm);
        n_refl += __popcll(am) - __popcll(rm);
        slot &= ~SLOT_REFRACTION;
      }

      // ---- raytrace (RayTracer.hs:55-71): closest hit ------------------------------
      float t;
      int p;
      traverse<false, PRUNE, COUNT>(P, stack, lane, active, o, d, 0.0f, t, p, cn);
      const bool hit = active && p >= 0;
      n_hits += __popcll(__ballot(hit));

      if (level == 1) {
        n_phits += __popcll(__ballot(hit));
        if (P.hit_ids && active) {
          int q = (int)slot;
          int pit = q / P.spp, s = q - pit * P.spp;
          int px = tx0 + pit % P.tile_w, py = ty0 + pit / P.tile_w;
          uint32_t id = RT_HIT_NONE;
          if (hit) id = (uint32_t)__float_as_int(P.prims[(size_t)p * PRIM_F4 + 3].z);
          P.hit_ids[((size_t)py * P.W + px) * P.spp + s] = id;
        }
      }

      // ---- shade -------------------------------------------------------------------
      V3 pt = mk(0, 0, 0), n = mk(0, 1, 0), clr = mk(0, 0, 0);
      int mat = 0;
      if (hit) {
        const float4 *q = P.prims + (size_t)p * PRIM_F4;
        float4 q3 = q[3];
        mat = __float_as_int(q3.y);
        pt = add(o, multiply(d, t));  // :63
        if (__float_as_int(q3.x) != 0) {
          float4 q2 = q[2];
          n = mk(q2.y, q2.z, q2.w);   // getNormal, triangle (:77)
        } else {
          float4 q0 = q[0];
          n = divide(subt(pt, mk(q0.x, q0.y, q0.z)), q0.w);  // getNormal, sphere (:76)
        }
      }
      const float *M = P.materials + (size_t)mat * 12;
      const V3 v = mk(-d.x, -d.y, -d.z);  // :65

      // getDirectColor (:81-106).  The reference folds the light list from the right:
      // c1 + (c2 + (... + ambient*diffuse)); iterate the lights backwards to add in
      // exactly that order.
      if (hit) clr = cmul(ld3(P.ambient), ld3(M + 0));  // :91
      for (int li = P.n_lights - 1; li >= 0; --li) {
        const float *Lp = P.lights + (size_t)li * 12;
        V3 l_a = ld3(Lp + 3), l_b = ld3(Lp + 6);
        V3 l_pt = add(ld3(Lp + 0), add(mk(l_a.x * s0, l_a.y * s0, l_a.z * s0),
                                       mk(l_b.x * s1, l_b.y * s1, l_b.z * s1)));  // :96
        V3 l = subt(l_pt, pt);
        V3 l_dir = normalize(l);
        float max_d = magnitude(l);
        float st;
        int sprim;
        traverse<true, PRUNE, COUNT>(P, stack, lane, hit, pt, l_dir, max_d, st, sprim, cn);  // inShadow :231-235
        n_shadow += __popcll(__ballot(hit));
        if (hit && sprim < 0) {
          V3 l_c = ld3(Lp + 9);
          float s = hs_max(0.0f, dot(l_dir, n));
          V3 diffuse_c = cscale(cmul(ld3(M + 0), l_c), s);
          V3 h = normalize(add(v, l_dir));
          float pw = powf(hs_max(0.0f, dot(n, h)), M[6]);
          V3 spec_c = cscale(cmul(l_c, ld3(M + 3)), pw);
          clr = add(add(diffuse_c, spec_c), clr);
        }
      }

      // ---- accumulate weight * direct colour into the sample (conflict-free) -------
      {
        V3 c = cmul(W, clr);
        bool pending = hit;
        const unsigned h = (slot * 0x9E3779B1u) >> 24;  // TAG_SLOTS = 256
        while (__ballot(pending)) {
          ++epoch;
          unsigned key = (epoch << 6) | (unsigned)(63 - lane);
          if (pending) atomicMax(&tag[h], key);
          bool win = pending && tag[h] == key;
          if (win) {
            acc[0 * TS + slot] += c.x;
            acc[1 * TS + slot] += c.y;
            acc[2 * TS + slot] += c.z;
            pending = false;
          }
          if (__ballot(pending)) __threadfence_block();
        }
      }

      // ---- getIndirectColor (:109-156): emit the child rays -----------------------
      if (level < P.max_depth) {
        bool e_refl = false, e_refr = false;
        V3 refl_v = mk(0, 0, 1), tdir = mk(0, 0, 1), w_refl = mk(0, 0, 0), w_refr = mk(0, 0, 0);
        if (hit) {
          V3 spec = ld3(M + 3);
          float phong = M[6], reflx = M[7], nt = M[8];
          V3 refl0 = reflect(d, n);
          V3 ou, ov;
          orthonormal(refl0, ou, ov);
          float a = 1 / phong;
          float ju = -(a / 2) + s0 * a;
          float jv = -(a / 2) + s1 * a;
          refl_v = add(refl0, add(multiply(ou, ju), multiply(ov, jv)));
          V3 f_refl = cscale(spec, reflx);
          e_refl = true;
          if (nt != 0) {
            V3 k = mk(1, 1, 1);
            float c;
            bool has_t;
            float dn = dot(d, n);
            if (dn < 0) {
              c = -dn;
              has_t = refract(d, n, nt, tdir);
            } else {
              has_t = refract(d, multiply(n, -1), 1 / nt, tdir);
              k = mk(expf(-M[9]), expf(-M[10]), expf(-M[11]));
              c = has_t ? dot(tdir, n) : 0.0f;
            }
            float r_0 = (nt - 1) * (nt - 1) / ((nt + 1) * (nt + 1));
            float om = 1 - c;
            float bigR = r_0 + (1 - r_0) * (om * om * om * om * om);
            if (has_t) {
              f_refl = cmul(cscale(f_refl, bigR), k);
              w_refr = cmul(W, cscale(k, 1 - bigR));
              e_refr = true;
            } else {
              f_refl = cmul(f_refl, k);
            }
          }
          w_refl = cmul(W, f_refl);
          if (!P.trace_zero_weight) {
            if (w_refl.x == 0 && w_refl.y == 0 && w_refl.z == 0) e_refl = false;
            if (e_refr && w_refr.x == 0 && w_refr.y == 0 && w_refr.z == 0) e_refr = false;
          }
        }
        unsigned long long m1 = __ballot(e_refl), m2 = __ballot(e_refr);
        if (m1 | m2) {
          const int nl = level + 1;
          int base = (int)cnt[nl];
          int off = base + __popcll(m1 & lt_mask) + __popcll(m2 & lt_mask);
          float *qn = Q + (size_t)(nl - 2) * QFIELDS * QCAP;
          if (e_refl) {
            qn[0 * QCAP + off] = pt.x; qn[1 * QCAP + off] = pt.y; qn[2 * QCAP + off] = pt.z;
            qn[3 * QCAP + off] = refl_v.x; qn[4 * QCAP + off] = refl_v.y; qn[5 * QCAP + off] = refl_v.z;
            qn[6 * QCAP + off] = w_refl.x; qn[7 * QCAP + off] = w_refl.y; qn[8 * QCAP + off] = w_refl.z;
            qn[9 * QCAP + off] = s0; qn[10 * QCAP + off] = s1;
            qn[11 * QCAP + off] = __uint_as_float(slot);
            ++off;
          }
          if (e_refr) {
            qn[0 * QCAP + off] = pt.x; qn[1 * QCAP + off] = pt.y; qn[2 * QCAP + off] = pt.z;
            qn[3 * QCAP + off] = tdir.x; qn[4 * QCAP + off] = tdir.y; qn[5 * QCAP + off] = tdir.z;
            qn[6 * QCAP + off] = w_refr.x; qn[7 * QCAP + off] = w_refr.y; qn[8 * QCAP + off] = w_refr.z;
            qn[9 * QCAP + off] = s0; qn[10 * QCAP + off] = s1;
            qn[11 * QCAP + off] = __uint_as_float(slot | SLOT_REFRACTION);
          }
          if (lane == 0) cnt[nl] = (unsigned)(base + __popcll(m1) + __popcll(m2));
        }
      }
    }  // level loop

    // ---- avgColors (Surfaces.hs:15-18): left fold from 0, then divide by n --------
    __threadfence_block();
    for (int pit = lane; pit < tile_px; pit += 64) {
      int px = tx0 + pit % P.tile_w, py = ty0 + pit / P.tile_w;
      if (px >= P.W || py >= P.H) continue;
      float r = 0.0f, g = 0.0f, b = 0.0f;
      for (int s = 0; s < P.spp; ++s) {
        int q = pit * P.spp + s;
        r = r + acc[0 * TS + q];
        g = g + acc[1 * TS + q];
        b = b + acc[2 * TS + q];
      }
      float nf = (float)P.spp;
      size_t o3 = P.compact ? ((size_t)ltile * tile_px + pit) * 3 : ((size_t)py * P.W + px) * 3;
      P.rgb[o3 + 0] = r / nf;
      P.rgb[o3 + 1] = g / nf;
      P.rgb[o3 + 2] = b / nf;
    }
  }  // tile loop

  // ---- statistics ------------------------------------------------------------------
  if (lane == 0) {
    atomicAdd(&P.stats[0], n_primary);
    atomicAdd(&P.stats[1], n_shadow);
    atomicAdd(&P.stats[2], n_refl);
    atomicAdd(&P.stats[3], n_refr);
    atomicAdd(&P.stats[7], n_phits);
    atomicAdd(&P.stats[8], n_hits);
  }
  if (COUNT) {
    atomicAdd(&P.stats[4], (unsigned long long)cn.box);
    atomicAdd(&P.stats[5], (unsigned long long)cn.sph);
    atomicAdd(&P.stats[6], (unsigned long long)cn.tri);
  }
}

template <bool PRUNE, bool COUNT>
static void launch_t(const RenderParams &p, const LaunchCfg &cfg, hipStream_t s) {
  hipLaunchKernelGGL((render_kernel<PRUNE, COUNT>), dim3(cfg.grid), dim3(cfg.block), cfg.lds_bytes, s,
                     p);
}

static void launch(const RenderParams &p, const LaunchCfg &cfg, hipStream_t s) {
  if (cfg.prune) {
    if (cfg.count) launch_t<true, true>(p, cfg, s);
    else launch_t<true, false>(p, cfg, s);
  } else {
    if (cfg.count) launch_t<false, true>(p, cfg, s);
    else launch_t<false, false>(p, cfg, s);
  }
}

static int occupancy(bool prune, bool count, size_t lds_bytes) {
  int n = 0;
  hipError_t e;
  if (prune) {
    if (count) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, render_kernel<true, true>, 256, lds_bytes);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, render_kernel<true, false>, 256, lds_bytes);
  } else {
    if (count) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, render_kernel<false, true>, 256, lds_bytes);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, render_kernel<false, false>, 256, lds_bytes);
  }
  return e == hipSuccess ? n : 0;
}

}  // namespace RT_NS
}  // namespace rtdev
