// rt_kernels.inl -- the trace/shade path as one persistent, wave-autonomous HIP kernel
// for gfx950.  Included by rt_kernels_strict.hip (-ffp-contract=off) and
// rt_kernels_fma.hip (-ffp-contract=fast) with RT_NS set.
//
// Execution model (DESIGN.md has the long form):
//   * every 64-lane wave is autonomous: it pulls packets of whole pixels (<= 64 primary
//     samples, e.g. 4 pixels of an 8x8-pixel tile at 16 spp) from sharded global counters
//     until none are left; there is no block-level synchronisation;
//   * lane = ray.  Primary rays are generated in registers; secondary rays
//     (reflection / refraction, RayTracer.hs:109-156) are pushed, wave-compacted with
//     ballot + popcount, into per-level ray queues in the wave's own HBM scratch
//     (SoA, coalesced), and popped 64 at a time, deepest full level first, so every
//     bounce runs with (almost) full waves and queue memory stays bounded
//     (<= 191 rays per level);
//   * BVH traversal (hitBVH/hits/hitsBox) keeps its per-lane stack in LDS;
//   * recursion is flattened to weight-carrying rays: colour(sample) =
//     sum over its rays of weight * direct colour, accumulated EXACTLY (2^-34 fixed
//     point) in a per-pixel buffer (L2-resident), so the frame is bitwise independent of the
//     scheduling; resolve_kernel then forms the pixel average (avgColors).
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
#ifdef RT_PROFILE_PHASES
  // per lane: loop iterations it took part in, and its 2^20 / (lanes taking part) share of each
  // (summed over the lanes: lane-iterations and 2^20 x wave-iterations).  [0] node steps,
  // [1] leaf steps of closest-hit traversals, [2], [3] the same for shadow traversals
  unsigned long long iters[4], lanes[4];
  // deepest stack of the traversal in progress (entries), and this lane's wide-node visits with a
  // record index below 64 << k (k = 0..6) / in all
  unsigned spmax;
  unsigned topk[8];
  // shader-clock cycles of the wave inside [0] node / [1] leaf iterations of closest-hit traversals, [2], [3] shadow;
  // [4] cycles inside the shadow-ray pool's re-arm blocks, [5] their number, [6] rays they handed out
  unsigned long long tcyc[7];
  // where the lanes are, summed over the loop iterations (wave-uniform counts, kept by every lane):
  // [0] iterations of the shadow-ray pool, lanes [1] at a node, [2] at a leaf, [3] idle with rays left in the
  // pool (waiting for a re-arm), [4] idle with the pool handed out (the tail);
  // [5] iterations of closest-hit traversals (scenes in memory), lanes [6] at a node, [7] at a leaf, [8] done
  unsigned long long where[9];
#endif
};
#ifdef RT_PROFILE_PHASES
#define DIV_COUNT(cn, i) do { (cn).iters[i] += 1; (cn).lanes[i] += (1u << 20) / (unsigned)__popcll(__ballot(true)); } while (0)
#define IT_T0 const unsigned long long it_t0_ = __builtin_amdgcn_s_memtime()
#define IT_T1(cn, i) do { (cn).tcyc[i] += __builtin_amdgcn_s_memtime() - it_t0_; } while (0)
#else
#define DIV_COUNT(cn, i)
#define IT_T0
#define IT_T1(cn, i)
#endif
#if defined(RT_PROFILE_PHASES) && defined(RT_PROFILE_STACK)
// (a second switch: one atomic per lane and traversal makes the frame 50x slower and the timers meaningless)
#define SP_COUNT(cn, sp) do { if ((unsigned)(sp) > (cn).spmax) (cn).spmax = (unsigned)(sp); } while (0)
#define TOPK_COUNT(cn, idx) do { for (int k_ = 0; k_ < 7; ++k_) (cn).topk[k_] += (idx) < (64 << k_); (cn).topk[7] += 1; } while (0)
// one traversal is over: its deepest stack goes into histogram `which` (0 closest hit, 1 shadow-ray pool)
#define SP_FLUSH(P, cn, which) do { atomicAdd(&(P).stats[STATS_DIAG_BASE + 32 + 40 * (which) + ((cn).spmax < 39u ? (cn).spmax : 39u)], 1ull); (cn).spmax = 0; } while (0)
#else
#define SP_COUNT(cn, sp)
#define TOPK_COUNT(cn, idx)
#define SP_FLUSH(P, cn, which)
#endif

// ---- exact division by a per-ray constant ---------------------------------------------
// hitsBox divides six box-minus-origin differences by the three direction components.
// The divisors are fixed for a ray, so 1/d is computed ONCE per ray with a true IEEE
// division and every quotient is then formed with Markstein's correction steps:
//   q0 = RN(a*r);  e = a - d*q0 (exact, FMA);  q1 = RN(q0 + e*r)   -> faithful
//   e2 = a - d*q1 (exact);                    q  = RN(q1 + e2*r)  -> RN(a/d)
// With r = RN(1/d) and q1 within one ulp of a/d the last step returns the correctly
// rounded quotient (Markstein 1990; Muller et al., Handbook of FP Arithmetic, Thm 4.8),
// i.e. bit-for-bit what `/` returns -- 5 VALU ops instead of the 11-op (one of them
// quarter-rate) IEEE division sequence.  The theorem needs no overflow/underflow
// anywhere, which holds when
//   |d| in [2^-40, 2^40],  origin and box coordinates are 0 or in [2^-36, 2^59]
// (then a = c - o is 0 or a multiple of 2^-59 below 2^60, and all residuals are normal
// numbers).  Rays or scenes outside that range take the plain-division path; +-0
// quotients may differ in sign only, which no comparison can see.
// tools/check_exact_div.c hammers the identity on the CPU.
struct RayDiv {
  float rx, ry, rz;
  bool safe;
};
DEVI bool coord_safe(float c) {
  float a = fabsf(c);
  return c == 0.0f || (a >= 1.4551915228366852e-11f /*2^-36*/ && a <= 5.764607523034235e17f /*2^59*/);
}
DEVI bool dir_safe(float d) {
  float a = fabsf(d);
  return a >= 9.094947017729282e-13f /*2^-40*/ && a <= 1099511627776.0f /*2^40*/;
}
DEVI RayDiv make_raydiv(const RenderParams &P, V3 o, V3 d) {
  RayDiv r;
  r.rx = 1.0f / d.x;
  r.ry = 1.0f / d.y;
  r.rz = 1.0f / d.z;
  r.safe = P.boxes_safe && dir_safe(d.x) && dir_safe(d.y) && dir_safe(d.z) && coord_safe(o.x) &&
           coord_safe(o.y) && coord_safe(o.z);
  return r;
}
#ifdef RT_FAST_MATH
// FMA / fast build: plain reciprocal multiply (not bit-identical to the reference)
DEVI float mdiv(float a, float d, float r) { (void)d; return a * r; }
#else
DEVI float mdiv(float a, float d, float r) {
  float q0 = a * r;
  float e = __builtin_fmaf(-d, q0, a);
  float q1 = __builtin_fmaf(e, r, q0);
  float e2 = __builtin_fmaf(-d, q1, a);
  return __builtin_fmaf(e2, r, q1);
}
#endif

// hitsBox (BoundingVolumeHierarchy.hs:191-204): six IEEE divisions, slab ends
// swapped by the sign test `d >= 0`, six interval comparisons.  t0/t1 out for the
// (result-preserving) range pruning.  SAFE selects the exact-reciprocal quotients.
// entry = max of the three slab entries, exit = min of the three slab exits.  On the SAFE
// path no quotient is NaN (finite operands, |d| >= 2^-40) and every box has min <= max
// (checked at upload), so t_i0 <= t_i1 on each axis by monotonicity of rounding and the six
// cross-axis comparisons of the reference collapse to ONE: some t_i0 > t_j1  <=>  entry > exit.
template <bool SAFE>
DEVI bool hits_box(V3 b, V3 d, const RayDiv &rd, float l, float r, float bo, float t, float n, float f,
                   float &entry, float &exit_) {
  float ax0 = (d.x >= 0 ? l : r) - b.x, ax1 = (d.x >= 0 ? r : l) - b.x;
  float ay0 = (d.y >= 0 ? bo : t) - b.y, ay1 = (d.y >= 0 ? t : bo) - b.y;
  float az0 = (d.z >= 0 ? n : f) - b.z, az1 = (d.z >= 0 ? f : n) - b.z;
  float tx0, tx1, ty0, ty1, tz0, tz1;
  if (SAFE) {
    tx0 = mdiv(ax0, d.x, rd.rx); tx1 = mdiv(ax1, d.x, rd.rx);
    ty0 = mdiv(ay0, d.y, rd.ry); ty1 = mdiv(ay1, d.y, rd.ry);
    tz0 = mdiv(az0, d.z, rd.rz); tz1 = mdiv(az1, d.z, rd.rz);
  } else {
    tx0 = ax0 / d.x; tx1 = ax1 / d.x;
    ty0 = ay0 / d.y; ty1 = ay1 / d.y;
    tz0 = az0 / d.z; tz1 = az1 / d.z;
  }
  entry = fmaxf(fmaxf(tx0, ty0), tz0);
  exit_ = fminf(fminf(tx1, ty1), tz1);
  if (SAFE) return !(entry > exit_);
  return !(tx0 > ty1 || tx0 > tz1 || tx1 < ty0 || tx1 < tz0 || ty0 > tz1 || ty1 < tz0);
}

// hit, sphere (RayTracer.hs:339-353); q0 = (C, r), r2 = r*r
DEVI bool hit_sphere(V3 base, V3 dir, float4 q0, float r2, float &t_out) {
  V3 ec = subt(base, mk(q0.x, q0.y, q0.z));
  float dec = dot(dir, ec);
  float dirdir = dot(dir, dir);
  float ecec = dot(ec, ec);
  float discriminant = dec * dec - dirdir * (ecec - r2);
  if (!(discriminant > 0)) { t_out = 0.0f; return false; }  // same verdict, skips sqrt + 2 divisions
  float disc_val = sqrtf(discriminant);
  // neg_dir_ec = dot (dir * (-1)) ec (:346-347) is -dec bit for bit: negation commutes with every
  // rounding of the products and sums (a zero sum may differ in SIGN only, and disc_val > 0 is
  // added to it).
  float neg_dir_ec = -dec;
  // t = min t_a t_b (:350) is t_b: disc_val > 0 and dirdir > 0 here (discriminant > 0 rules out a
  // zero or NaN direction), so RN(n - s) <= RN(n + s) and division by dirdir keeps the order; when
  // the two are equal `min` returns the same number.  One division instead of two.
  float t = (neg_dir_ec - disc_val) / dirdir;
  t_out = t;
  return t > EPSILON;
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
  if (theta >= 0) { t_out = 0.0f; return false; }  // back face: same verdict, skips the 3 divisions
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
// Scene records: HBM (through L1/L2) or, when the whole scene fits, the block's LDS copy.
// LDS records are LDS_REC_F4 = 5 float4 (80 B) apart: 20 dwords mod 64 banks walks all 16
// four-bank groups, so a 16-lane ds_read_b128 group reading 16 different records is
// conflict-free (64-B spacing would fold them onto 4 groups).
template <bool LDS>
DEVI const float4 *node_rec(const RenderParams &P, const float4 *sc, int idx) {
  return LDS ? sc + (size_t)idx * LDS_REC_F4 : P.nodes + (size_t)idx * NODE_F4;
}
template <bool LDS>
DEVI const float4 *prim_rec(const RenderParams &P, const float4 *sc, int idx) {
  return LDS ? sc + (P.lds_prim_base + idx * LDS_REC_F4) : P.prims + (size_t)idx * PRIM_F4;
}

// One inner node: both children's hitsBox tests, the (result-preserving) pruning and ordering,
// then the move -- descend, or pop, or stop.
template <bool PRUNE, bool COUNT, bool LDS>
DEVI void node_step(const RenderParams &P, const float4 *sc, unsigned *stack, int lane, V3 o, V3 d, const RayDiv &rd,
                    bool all_safe, float best_t, int &cur, int &sp, Counters &cn) {
  const float4 *np = node_rec<LDS>(P, sc, cur);
  float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
  int lref = __float_as_int(n3.x), rref = __float_as_int(n3.y);
  float le, lx, re = 0.0f, rx = 0.0f;
  bool hl, hr = false;
  if (all_safe) {
    hl = hits_box<true>(o, d, rd, n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, le, lx);
    if (rref != REF_NONE) hr = hits_box<true>(o, d, rd, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, re, rx);
  } else {
    hl = hits_box<false>(o, d, rd, n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, le, lx);
    if (rref != REF_NONE) hr = hits_box<false>(o, d, rd, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, re, rx);
  }
  if (COUNT) cn.box += (rref != REF_NONE) ? 2 : 1;
  bool left_first = true;
  if (PRUNE) {
    // a slab entry beyond the limit on any axis <=> entry > lim; an exit behind the origin
    // on any axis <=> exit < BEHIND.  (A NaN quotient -- plain-division path only: zero
    // direction component with the origin on a box plane -- is ignored by max/min; the
    // prune then rests on a real axis and is just as valid.)
    const float lim = prune_limit(best_t);
    if (hl && (le > lim || lx < BEHIND)) hl = false;
    if (hr && (re > lim || rx < BEHIND)) hr = false;
    left_first = !(re < le);
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
}

// ---- wide (4-child) node step of the pruned traversal (record layout: rt_device.h) ----------
// Per ray: sx, sy, sz = 1 where the direction component is negative (the reference's slab swap,
// `d >= 0` is true for -0.0: BoundingVolumeHierarchy.hs:194-199), else 0.
struct RaySign {
  int x, y, z;
};
DEVI RaySign make_sign(V3 d) {
  RaySign s;
  s.x = d.x >= 0 ? 0 : 1;
  s.y = d.y >= 0 ? 0 : 1;
  s.z = d.z >= 0 ? 0 : 1;
  return s;
}
template <bool LDS>
DEVI const float4 *wnode_rec(const RenderParams &P, const float4 *sc, int idx) {
  return LDS ? sc + idx * LDS_WNODE_F4 : P.wnodes + (size_t)idx * WNODE_F4;
}
// compare-exchange of (key, ref) pairs: afterwards ka <= kb
DEVI void cswap(float &ka, int &ra, float &kb, int &rb) {
  const bool sw = kb < ka;
  const float k0 = sw ? kb : ka, k1 = sw ? ka : kb;
  const int r0 = sw ? rb : ra, r1 = sw ? ra : rb;
  ka = k0; kb = k1; ra = r0; rb = r1;
}
// ---- conservative box tests for the inner levels, exact verdicts where they matter ------------
// The reference tests a primitive iff hitsBox accepts the box of every BVH node from the root down
// to the primitive's leaf (RayTracer.hs:404-412).  On the exact-quotient path a box's verdict
// implies its ancestors' (a parent's planes are the min / max of its children's and every rounding
// is monotone: parent entry <= child entry, parent exit >= child exit on each axis), so the
// primitives the reference tests are exactly those whose LEAF box it accepts.  The pruned traversal
// therefore only needs the exact verdict of a leaf box, and only when it matters: when a primitive
// of that leaf has become the ray's hit (leaf_box_clear below; once per traversal).  Every
// box test on the way down may err on the side of "hit": it costs 6 fused multiply-adds per box
// against a per-ray reciprocal, no divisions and no corrections.
//
// Per ray and axis: r~ = v_rcp_f32(d) (1 ulp), or~ = RN(o * r~); a slab distance is one FMA,
// t~ = RN(c * r~ - or~), against the reference's t = RN(RN(c - o) / d).  With tau = (c - o) / d:
// the reciprocal (2 u), the rounding inside or~ (u |or~|), the final rounding (u) and t's own two
// roundings (2.1 u |tau|) give  |t~ - t| <= 5.2 u |t~| + 1.02 u |or~|   (u = 2^-24; the cancellation
// c * r~ - or~ is exact inside the FMA).  Two ways to stay on the safe side of that, E~ = max of the
// three entries, X~ = min of the three exits:
//  * scenes held in LDS (few node visits per ray: the per-ray set-up counts): the reference
//    accepting (E <= X) implies E~ - X~ <= 5.2 u (|E~| + |X~|) + 2.04 u max |or~| <
//    thr := 2^-21 (|E~| + |X~|) + 2^-22 max |or~|; a box is skipped only when E~ - X~ > thr, range
//    pruning uses E~ - thr and X~ + thr;
//  * scenes in memory (tens of node visits per ray): the slop is folded into the FMA's addend once
//    per ray.  delta = 2^-20 (S + |o|) |r~| per axis, S = the largest |coordinate| of any box of the
//    scene; entries are RN(c * r~ + a0), a0 = RN(-or~ - delta), exits RN(c * r~ + a1), a1 =
//    RN(-or~ + delta).  |tau| <= (S + |o|) |1/d|, the addend's own rounding adds u (|or~| + delta):
//    all errors together stay below 7.3 u (S + |o|) |r~| + u delta < delta = 16 u (S + |o|) |r~|, so
//    every computed entry is <= the reference's and every computed exit >= the reference's, and
//    the reference accepting implies E~ <= X~: 6 FMAs, max3, min3 and three comparisons per box.
// No overflow or NaN on this path: |d| in [2^-40, 2^40], |o| and box coordinates <= 2^59 (other
// rays and scenes take the plain-division path below, which is exact at every node).
// The slop of the conservative tests: 2^-21 relative (scenes in LDS), 2^-20 (scenes in memory).  The
// RT_DEBUG_KNOBS build (librt_mi355x_dbg.so, tests only) takes them from the launch parameters so
// that RT_DEBUG_BOX_SLOP can make the conservative tests err on purpose.
#ifdef RT_DEBUG_KNOBS
#define RT_SLOP_LDS(P) ((P).slop_lds)
#define RT_SLOP_MEM(P) ((P).slop_mem)
#else
#define RT_SLOP_LDS(P) 4.76837158203125e-07f
#define RT_SLOP_MEM(P) 9.5367431640625e-07f
#endif
struct RayBox {
  float rx, ry, rz;     // ~ 1 / d
  float nx, ny, nz;     // addend of the near planes: -(o * r~) [- delta]
  float fx, fy, fz;     // addend of the far planes (scenes in memory): -(o * r~) + delta
  float b0;             // scenes in LDS: 2^-22 max |or~|
  bool ok;              // the conservative path is valid for this ray
};
template <bool LDS>
DEVI RayBox make_raybox(const RenderParams &P, V3 o, V3 d) {
  RayBox b;
  b.rx = __builtin_amdgcn_rcpf(d.x);
  b.ry = __builtin_amdgcn_rcpf(d.y);
  b.rz = __builtin_amdgcn_rcpf(d.z);
  const float ox = o.x * b.rx, oy = o.y * b.ry, oz = o.z * b.rz;
  if (LDS) {
    b.nx = b.fx = -ox;
    b.ny = b.fy = -oy;
    b.nz = b.fz = -oz;
    b.b0 = (0.5f * RT_SLOP_LDS(P)) * fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz));   // 2^-22 max |or~|
  } else {
    const float S = P.box_absmax;
    const float dx = (RT_SLOP_MEM(P) * (S + fabsf(o.x))) * fabsf(b.rx);
    const float dy = (RT_SLOP_MEM(P) * (S + fabsf(o.y))) * fabsf(b.ry);
    const float dz = (RT_SLOP_MEM(P) * (S + fabsf(o.z))) * fabsf(b.rz);
    b.nx = -ox - dx; b.ny = -oy - dy; b.nz = -oz - dz;
    b.fx = dx - ox;  b.fy = dy - oy;  b.fz = dz - oz;
    b.b0 = 0.0f;
  }
  constexpr float DLO = 9.094947017729282e-13f /*2^-40*/, DHI = 1099511627776.0f /*2^40*/,
                  OHI = 5.764607523034235e17f /*2^59*/;
  // (every comparison is false for a NaN)
  b.ok = P.boxes_safe && fabsf(d.x) >= DLO && fabsf(d.y) >= DLO && fabsf(d.z) >= DLO && fabsf(d.x) <= DHI &&
         fabsf(d.y) <= DHI && fabsf(d.z) <= DHI && fabsf(o.x) <= OHI && fabsf(o.y) <= OHI && fabsf(o.z) <= OHI;
  return b;
}

// Leaf boxes (exact verdicts): two float4 per primitive record, the box of the record's LEAF:
// (xmin, xmax, ymin, ymax) (zmin, zmax, -, -)
template <bool LDS>
DEVI const float4 *leafbox_rec(const RenderParams &P, const float4 *sc, int prim) {
  return LDS ? sc + (P.lds_leafbox_base + prim * 2) : P.leafbox + (size_t)prim * 2;
}
// Does the reference's hitsBox (BoundingVolumeHierarchy.hs:191-204) CERTAINLY accept the box of the
// leaf that holds record `prim`?  (Rays on the conservative path: rb.ok.)  Quotients by the ray's
// 1-ulp reciprocal: q~ = RN(a * r~) is within 4.1 u |q~| of the correctly rounded RN(a / d).  The
// verdict "every slab entry <= every exit of ANOTHER axis" is the sign of the smallest cross-axis
// gap g = min_j (exit_j - max_{i != j} entry_i), and with D = 2^-20 (|E~| + |X~|) (eps = 2^-21 >=
// 4.1 u each): g~ > D => the reference accepts (had the order of a cross pair flipped under the
// roundings although |g~| > D, its two members would lie between X~ and E~ and their error would
// be below D).  A slab's entry never exceeds its OWN exit here (box min <= max, monotone rounding),
// and for a flat box -- an axis-aligned triangle's -- the two are one number, so the own-axis gap
// is no yardstick.  Anything else (a ray grazing a box edge to 7 digits, a miss, a NaN) is not
// decided here: the caller walks the tree again on the plain-division path, which is the
// reference's own arithmetic at every node.
template <bool LDS>
DEVI bool leaf_box_clear(const RenderParams &P, const float4 *sc, int prim, V3 o, V3 d, const RayBox &rb) {
  const float4 *bp = leafbox_rec<LDS>(P, sc, prim);
  const float4 b0 = bp[0], b1 = bp[1];
  const float l = b0.x, r = b0.y, bo = b0.z, t = b0.w, n = b1.x, f = b1.y;
  const float tx0 = ((d.x >= 0 ? l : r) - o.x) * rb.rx, tx1 = ((d.x >= 0 ? r : l) - o.x) * rb.rx;
  const float ty0 = ((d.y >= 0 ? bo : t) - o.y) * rb.ry, ty1 = ((d.y >= 0 ? t : bo) - o.y) * rb.ry;
  const float tz0 = ((d.z >= 0 ? n : f) - o.z) * rb.rz, tz1 = ((d.z >= 0 ? f : n) - o.z) * rb.rz;
  const float mx = fmaxf(ty0, tz0), my = fmaxf(tx0, tz0), mz = fmaxf(tx0, ty0);
  const float E = fmaxf(tx0, mx), X = fminf(fminf(tx1, ty1), tz1);
  const float g = fminf(fminf(tx1 - mx, ty1 - my), tz1 - mz);
  return g > P.clear_band * (fabsf(E) + fabsf(X));
}

// Four box tests from one wide record, range pruning, then the children that remain go on the
// stack farthest first and the nearest becomes current (or pop, or stop).  The stack writes are
// unconditional -- a slot above the top is scratch -- so the step has no divergent pushes.
// all_ok: every lane of the wave is on the conservative path (RayBox::ok); otherwise the whole
// wave takes plain divisions and the reference's comparisons (exact at every node).
// A slot without a child holds an inverted box (min = 2^58, max = -2^58; not counted in S):
// E~ - X~ is about 2^59 |r~|, far beyond any slop, so it is never entered on the conservative path.
// (RT_DEBUG_KNOBS build, RT_DEBUG_NO_RANGE_PRUNE: no range pruning -- a test tree whose leaf boxes do not
//  bound their objects breaks what the pruning rests on, not what the exact verdicts rest on)
#ifdef RT_DEBUG_KNOBS
#define RT_LIM(P, t) ((P).debug_no_range ? INFINITY : prune_limit(t))
#define RT_BEHIND(P) ((P).debug_no_range ? -INFINITY : BEHIND)
#else
#define RT_LIM(P, t) prune_limit(t)
#define RT_BEHIND(P) BEHIND
#endif
template <bool ANY, bool COUNT, bool LDS>
DEVI void wide_step(const RenderParams &P, const float4 *sc, unsigned *stack, int lane, V3 o, V3 d, const RayBox &rb,
                    const RaySign &sg, bool all_ok, float best_t, int &cur, int &sp, Counters &cn) {
  const float4 *np = wnode_rec<LDS>(P, sc, cur);
  TOPK_COUNT(cn, cur);
  const float4 rf = np[6];
  int r0 = __float_as_int(rf.x), r1 = __float_as_int(rf.y), r2 = __float_as_int(rf.z), r3 = __float_as_int(rf.w);
  if (COUNT) cn.box += (r0 != REF_NONE) + (r1 != REF_NONE) + (r2 != REF_NONE) + (r3 != REF_NONE);
  // range pruning as in node_step
  const float lim = RT_LIM(P, best_t);
  const float behind = RT_BEHIND(P);
  float e0, e1, e2, e3;
  bool h0, h1, h2, h3;
  if (all_ok) {
    const float4 nx = np[sg.x], fx = np[sg.x ^ 1], ny = np[2 + sg.y], fy = np[2 + (sg.y ^ 1)], nz = np[4 + sg.z],
                 fz = np[4 + (sg.z ^ 1)];
#define RT_WIDE_CHILD(C_, E_, H_)                                                                           \
  {                                                                                                         \
    const float tx0 = __builtin_fmaf(nx.C_, rb.rx, rb.nx), tx1 = __builtin_fmaf(fx.C_, rb.rx, rb.fx);       \
    const float ty0 = __builtin_fmaf(ny.C_, rb.ry, rb.ny), ty1 = __builtin_fmaf(fy.C_, rb.ry, rb.fy);       \
    const float tz0 = __builtin_fmaf(nz.C_, rb.rz, rb.nz), tz1 = __builtin_fmaf(fz.C_, rb.rz, rb.fz);       \
    E_ = fmaxf(fmaxf(tx0, ty0), tz0);                                                                       \
    const float X_ = fminf(fminf(tx1, ty1), tz1);                                                           \
    if (LDS) {                                                                                              \
      const float thr = __builtin_fmaf(fabsf(E_) + fabsf(X_), RT_SLOP_LDS(P), rb.b0);                       \
      H_ = (E_ - X_ <= thr) && !(E_ - thr > lim || X_ + thr < behind);                                      \
    } else {                                                                                                \
      H_ = (E_ <= X_) && !(E_ > lim || X_ < behind);                                                        \
    }                                                                                                       \
  }
    RT_WIDE_CHILD(x, e0, h0)
    RT_WIDE_CHILD(y, e1, h1)
    RT_WIDE_CHILD(z, e2, h2)
    RT_WIDE_CHILD(w, e3, h3)
#undef RT_WIDE_CHILD
  } else {
    // plain divisions (a zero / denormal / huge direction component, out-of-range origin or scene);
    // a NaN entry or exit is ignored by max / min and never prunes
    const float4 a0 = np[0], a1 = np[1], a2 = np[2], a3 = np[3], a4 = np[4], a5 = np[5];
    RayDiv rd;
    rd.rx = rd.ry = rd.rz = 1.0f;
    rd.safe = false;
    float x0, x1, x2, x3;
    h0 = hits_box<false>(o, d, rd, a0.x, a1.x, a2.x, a3.x, a4.x, a5.x, e0, x0);
    h1 = hits_box<false>(o, d, rd, a0.y, a1.y, a2.y, a3.y, a4.y, a5.y, e1, x1);
    h2 = hits_box<false>(o, d, rd, a0.z, a1.z, a2.z, a3.z, a4.z, a5.z, e2, x2);
    h3 = hits_box<false>(o, d, rd, a0.w, a1.w, a2.w, a3.w, a4.w, a5.w, e3, x3);
    h0 = h0 && r0 != REF_NONE && !(e0 > lim || x0 < behind);
    h1 = h1 && r1 != REF_NONE && !(e1 > lim || x1 < behind);
    h2 = h2 && r2 != REF_NONE && !(e2 > lim || x2 < behind);
    h3 = h3 && r3 != REF_NONE && !(e3 > lim || x3 < behind);
  }
  if (ANY) {
    // "is anything in the way" does not care in which order the boxes are searched: the first
    // child that passed becomes current, the others go on the stack as they come (no sort)
    const bool p1 = h1 && h0, p2 = h2 && (h0 || h1), p3 = h3 && (h0 || h1 || h2);
    unsigned *sl = stack + sp * 64 + lane;
    sl[0] = (unsigned)r3;
    sl += p3 ? 64 : 0;
    sl[0] = (unsigned)r2;
    sl += p2 ? 64 : 0;
    sl[0] = (unsigned)r1;
    sp += (int)p1 + (int)p2 + (int)p3;
    SP_COUNT(cn, sp);
    if (h0 || h1 || h2 || h3) {
      cur = h0 ? r0 : (h1 ? r1 : (h2 ? r2 : r3));
    } else if (sp > 0) {
      --sp;
      cur = (int)stack[sp * 64 + lane];
    } else {
      cur = STOP;
    }
    return;
  }
  const int n = (int)h0 + (int)h1 + (int)h2 + (int)h3;
  // sort by entry distance, misses last (5 compare-exchanges)
  // (a hit's key is capped below a miss's +inf -- also a NaN or +inf entry, which fminf turns into
  //  FLT_MAX -- so that the n hits always end up in the first n places)
  constexpr float KMAX = 3.4028234663852886e38f;
  float k0 = h0 ? fminf(e0, KMAX) : INFINITY, k1 = h1 ? fminf(e1, KMAX) : INFINITY,
        k2 = h2 ? fminf(e2, KMAX) : INFINITY, k3 = h3 ? fminf(e3, KMAX) : INFINITY;
  cswap(k0, r0, k1, r1);
  cswap(k2, r2, k3, r3);
  cswap(k0, r0, k2, r2);
  cswap(k1, r1, k3, r3);
  cswap(k1, r1, k2, r2);
  unsigned *sl = stack + sp * 64 + lane;
  sl[0] = (unsigned)r3;
  sl += (n >= 4) ? 64 : 0;
  sl[0] = (unsigned)r2;
  sl += (n >= 3) ? 64 : 0;
  sl[0] = (unsigned)r1;
  sp += (n >= 2 ? n - 1 : 0);
  SP_COUNT(cn, sp);
  if (n >= 1) {
    cur = r0;
  } else if (sp > 0) {
    --sp;
    cur = (int)stack[sp * 64 + lane];
  } else {
    cur = STOP;
  }
}

// One leaf: records first .. first+count-1 (reversed list order), then pop or stop.
template <bool ANY, bool PRUNE, bool COUNT, bool LDS>
DEVI void leaf_step(const RenderParams &P, const float4 *sc, unsigned *stack, int lane, V3 o, V3 d, const RayBox &rb,
                    bool verify_now, bool &redo, float tlimit, float &best_t, int &best_p, int &cur, int &sp,
                    Counters &cn) {
  int first = -1 - cur;
  const float4 *pp = prim_rec<LDS>(P, sc, first);
  int count = __float_as_int(pp[3].w);
  // the leaf's candidate: what the ray's hit would become if the reference visits this leaf
  float lt = best_t;
  int lp = best_p;
  for (int k = 0; k < count; ++k) {
    const float4 *q = pp + (size_t)k * (LDS ? LDS_REC_F4 : PRIM_F4);
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
    if (ok && (ANY ? (t < tlimit) : (t < lt || (t == lt && first + k > lp)))) {
      lt = t;
      lp = first + k;
      if (ANY) break;
    }
  }
  bool got = lp != best_p;
  // PRUNE: only now does it matter whether the reference visits this leaf at all.  The closest-hit
  // and single-light shadow traversals put the question off to their end (traverse); the
  // shadow-ray pool asks here, and a ray without a clear answer starts again on the exact path.
  redo = false;
  if (PRUNE && verify_now && got && !leaf_box_clear<LDS>(P, sc, lp, o, d, rb)) {
    got = false;
    redo = true;
  }
  if (got) {
    best_t = lt;
    best_p = lp;
  }
  if (ANY && got) {
    cur = STOP;
  } else if (sp > 0) {
    --sp;
    cur = (int)stack[sp * 64 + lane];
  } else {
    cur = STOP;
  }
}

// One inner-node step of either tree: the pruned traversal walks the wide records.
template <bool ANY, bool PRUNE, bool COUNT, bool LDS>
DEVI void inner_step(const RenderParams &P, const float4 *sc, unsigned *stack, int lane, V3 o, V3 d, const RayDiv &rd,
                     const RayBox &rb, const RaySign &sg, bool all_safe, float best_t, int &cur, int &sp, Counters &cn) {
  if (PRUNE) wide_step<ANY, COUNT, LDS>(P, sc, stack, lane, o, d, rb, sg, all_safe, best_t, cur, sp, cn);
  else node_step<PRUNE, COUNT, LDS>(P, sc, stack, lane, o, d, rd, all_safe, best_t, cur, sp, cn);
}

// Per-ray constants of the box tests: the exact reciprocals of the reference-visit mode, or the
// conservative test's (pruned mode); `safe` = this ray may take the mode's fast path.
template <bool PRUNE, bool LDS>
DEVI void ray_setup(const RenderParams &P, V3 o, V3 d, RayDiv &rd, RayBox &rb, bool &safe) {
  rd.rx = rd.ry = rd.rz = 1.0f;
  rd.safe = true;
  rb.rx = rb.ry = rb.rz = 1.0f;
  rb.nx = rb.ny = rb.nz = rb.fx = rb.fy = rb.fz = rb.b0 = 0.0f;
  rb.ok = true;
  if (PRUNE) {
    rb = make_raybox<LDS>(P, o, d);
    safe = rb.ok;
  } else {
    rd = make_raydiv(P, o, d);
    safe = rd.safe;
  }
}

template <bool ANY, bool PRUNE, bool COUNT, bool LDS>
DEVI void traverse(const RenderParams &P, const float4 *sc, unsigned *stack, int lane, bool active, V3 o, V3 d,
                   float tlimit, float &best_t, int &best_p, Counters &cn) {
  best_t = ANY ? tlimit : INFINITY;
  best_p = -1;
  int sp = 0;
  int cur = (active && P.has_root) ? 0 : STOP;
  RayDiv rd;
  RayBox rb;
  bool safe;
  ray_setup<PRUNE, LDS>(P, o, d, rd, rb, safe);
#ifdef RT_DEBUG_KNOBS
  if (P.debug_unsafe_mod && lane % P.debug_unsafe_mod == 0) safe = false;  // (tests)
#endif
  const RaySign sg = make_sign(d);
  // one unsafe lane sends the whole wave down the plain-division path (rare: a zero or
  // denormal direction component, or an out-of-range scene)
  bool all_safe = __ballot(cur != STOP && !safe) == 0ull;
  // Pruned mode: whether the reference accepts the winning primitive's leaf box is asked once, when
  // the traversal is over (leaf_box_clear: ~30 instructions per traversal).  A lane without a clear
  // "yes" -- about one ray in a million grazes a box edge that closely -- walks the tree again on
  // the plain-division path, where every box test is the reference's own.
  bool redo_unused;
  for (;;) {
    if (!LDS) {
      // scenes too big for LDS (deep trees, divergent rays): each iteration runs whichever of the
      // two bodies more lanes are waiting for, instead of descending until every lane stands on
      // a leaf (mesh workload 6 % faster single-frame; the small LDS scenes are faster while-while)
      for (;;) {
        const int at_node = __popcll(__ballot(cur >= 0)), at_leaf = __popcll(__ballot(cur < 0 && cur != STOP));
        if ((at_node | at_leaf) == 0) break;
#ifdef RT_PROFILE_PHASES
        if (!ANY) { cn.where[5] += 1; cn.where[6] += at_node; cn.where[7] += at_leaf; cn.where[8] += 64 - at_node - at_leaf; }
#endif
        IT_T0;
        if (at_node >= at_leaf) {
          if (cur >= 0) {
            DIV_COUNT(cn, ANY ? 2 : 0);
            inner_step<ANY, PRUNE, COUNT, LDS>(P, sc, stack, lane, o, d, rd, rb, sg, all_safe, best_t, cur, sp, cn);
          }
          IT_T1(cn, ANY ? 2 : 0);
        } else {
          if (cur < 0 && cur != STOP) {
            DIV_COUNT(cn, ANY ? 3 : 1);
            leaf_step<ANY, PRUNE, COUNT, LDS>(P, sc, stack, lane, o, d, rb, false, redo_unused, tlimit, best_t, best_p, cur, sp, cn);
          }
          IT_T1(cn, ANY ? 3 : 1);
        }
      }
    } else {
      // "while-while" form: all lanes first descend until each stands on a leaf (or is done),
      // then the leaves are processed together -- node code and primitive code are not
      // interleaved per iteration, which keeps more lanes busy in both.
      while (cur != STOP) {
        while (cur >= 0) {
          DIV_COUNT(cn, ANY ? 2 : 0);
          inner_step<ANY, PRUNE, COUNT, LDS>(P, sc, stack, lane, o, d, rd, rb, sg, all_safe, best_t, cur, sp, cn);
        }
        if (cur != STOP) {
          DIV_COUNT(cn, ANY ? 3 : 1);
          leaf_step<ANY, PRUNE, COUNT, LDS>(P, sc, stack, lane, o, d, rb, false, redo_unused, tlimit, best_t, best_p, cur, sp, cn);
        }
      }
    }
    // (waves on the plain-division path have tested every box exactly on the way down)
    if (!PRUNE || !all_safe) break;
    const bool bad = best_p >= 0 && !leaf_box_clear<LDS>(P, sc, best_p, o, d, rb);
    if (__ballot(bad) == 0ull) break;
    all_safe = false;
    if (bad) {
      best_t = ANY ? tlimit : INFINITY;
      best_p = -1;
      sp = 0;
      cur = 0;
    }
  }
  if (active) SP_FLUSH(P, cn, ANY ? 1 : 0);
}

// The sampled point on light li and the shadow ray towards it (getDirectColor :96-100).
DEVI void light_ray(const float *Lp, V3 pt, float s0, float s1, V3 &l_dir, float &max_d) {
  V3 l_a = ld3(Lp + 3), l_b = ld3(Lp + 6);
  V3 l_pt = add(ld3(Lp + 0), add(mk(l_a.x * s0, l_a.y * s0, l_a.z * s0),
                                 mk(l_b.x * s1, l_b.y * s1, l_b.z * s1)));  // :96
  V3 l = subt(l_pt, pt);
  l_dir = normalize(l);
  max_d = magnitude(l);
}

// x ** p of the Phong term (:104).  pow(x, 1) is x for every x (NaN, 0 and inf included), so a
// wave whose lit lanes all carry exponent 1 (the matte materials of the shipped scenes) does no
// power at all; mixed waves select per lane.
// Otherwise x ** p = 2 ** (p * log2 x) on the hardware's log2 / exp2 (v_log_f32, v_exp_f32: 1 ulp
// each), 3 instructions instead of the ~100 of the powf expansion.  x is max 0 (n . h) in [0, 1];
// the error of the log, 2^-23 |log2 x|, is multiplied by p, so the result is off by about
// (p |log2 x|) 2^-23 ln 2 relative: measured against double-precision pow for p = 0.5 .. 1e5
// (tools/micro/pow_probe.hip, profiles/r03_pow_probe.txt) at most 1.4e-6 relative wherever
// x ** p >= 1e-4 and 8.5e-8 absolute everywhere (the device's powf: 1.4e-7; the oracle's is
// glibc's).  A colour only: no hit decision and no ray count depends on it.  0 ** p = 0 (p > 0)
// and NaN ** p = NaN come out of the hardware ops; anything ** 0 = 1 is selected.
// -DRT_LIBM_POW keeps powf.
#ifdef RT_LIBM_POW
DEVI float pow_x_p(float x, float p) { return powf(x, p); }
#else
DEVI float pow_x_p(float x, float p) {
  const float w = __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(x));
  return p == 0.0f ? 1.0f : w;
}
#endif
DEVI float phong_pow(float x, float p) {
  if (__ballot(p != 1.0f) == 0ull) return x;
  const float w = pow_x_p(x, p);
  return p == 1.0f ? x : w;
}

// Default ray policy (not with RT_FLAG_TRACE_ZERO_WEIGHT): a light whose diffuse AND specular
// factors are both exactly zero for this point -- max 0 (l.n) = 0 and max 0 (n.h) = 0 with a
// positive Phong exponent (getDirectColor :101-105) -- adds (0 + 0) to the colour whether or not
// it is occluded, so its inShadow ray is neither traced nor counted.  The decision uses only the
// reference's own single-rounded dot products (the oracle takes it the same way).
DEVI bool light_matters(V3 l_dir, V3 n, V3 v, float phong) {
  return !(dot(l_dir, n) <= 0 && dot(n, normalize(add(v, l_dir))) <= 0 && phong > 0);
}

// inShadow (:231-235) for lights lo .. hi-1 (hi - lo <= 32) of the wave's shaded points, one
// point per `hit` lane.  `need` (in): bit (li - lo) set = this lane's point needs the shadow
// ray of light li.  Returns the same layout with a bit set where the light is occluded.
// The needed (point, light) pairs form ONE pool of shadow rays for the wave, light by light
// so that rays in flight together mostly aim at the same light.  The points are parked in LDS
// and every lane -- also those whose own ray missed -- takes the next ray of the pool whenever
// its current one is decided ("dynamic fetch"), instead of the wave walking the light list in
// lockstep with only the `hit` lanes and waiting for the slowest ray of every light.  An
// occluded ray ORs its bit into the owner's mask in LDS.  The answers are per-ray facts: who
// computes them, and in which order, cannot change them (the light ray is recomputed from
// the same operands by the same operations).
// Lanes are re-armed only when P.rearm_min of them are idle (or nothing else is running): the
// set-up code is issued for the whole wave whoever needs it.
// LDS, 64 words each: x, y, z, s0, s1, occlusion mask (indexed by the owner's lane), hand-out
// table, per-light 64-bit masks of the points that need the light.
template <bool PRUNE, bool COUNT, bool LDS>
DEVI unsigned shadow_lights(const RenderParams &P, const float4 *sc, unsigned *stack, unsigned *pts, int lane,
                            unsigned long long lt_mask, bool hit, V3 pt, float s0, float s1, unsigned need,
                            int hi, int lo, Counters &cn) {
  if (!P.has_root) return 0;
  unsigned *occl = pts + 5 * 64, *handout = pts + 6 * 64, *lmask = pts + 7 * 64;
  int remaining = 0;  // rays of the pool not handed out yet (wave-uniform)
  for (int b = 0; b < hi - lo; ++b) {
    const unsigned long long m = __ballot(hit && ((need >> b) & 1u));
    remaining += __popcll(m);
    if (lane == 0) {
      lmask[2 * b] = (unsigned)m;
      lmask[2 * b + 1] = (unsigned)(m >> 32);
    }
  }
  if (remaining == 0) return 0;
  if (hit) {
    pts[0 * 64 + lane] = __float_as_uint(pt.x);
    pts[1 * 64 + lane] = __float_as_uint(pt.y);
    pts[2 * 64 + lane] = __float_as_uint(pt.z);
    pts[3 * 64 + lane] = __float_as_uint(s0);
    pts[4 * 64 + lane] = __float_as_uint(s1);
    occl[lane] = 0u;
  }
  __threadfence_block();  // other lanes of this wave read them
  int cl = hi - lo;                // current light (bit index); its unserved points in `rm`
  unsigned long long rm = 0ull;    // both wave-uniform
  int cur = STOP, sp = 0, owner = 0, bit = 0;
  bool armed = false;  // (profiling build: this lane has taken a ray of the pool)
  (void)armed;
  V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
  float tlimit = 0.0f;
  RayDiv rd;
  RayBox rb;
  bool safe;
  ray_setup<PRUNE, LDS>(P, mk(0, 0, 0), mk(1, 1, 1), rd, rb, safe);
  RaySign sg = make_sign(d);
  for (;;) {
    const unsigned long long idle = __ballot(cur == STOP);
    if (remaining > 0 && (~idle == 0ull || __popcll(idle) >= P.rearm_min)) {
      IT_T0;
      while (rm == 0ull) {  // next light with work (there is one: remaining > 0)
        --cl;
        rm = ((unsigned long long)lmask[2 * cl + 1] << 32) | lmask[2 * cl];
        rm = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(rm >> 32)) << 32) |
             (unsigned)__builtin_amdgcn_readfirstlane((int)rm);
      }
      const int n_idle = __popcll(idle), n_left = __popcll(rm);
      const int n_take = n_idle < n_left ? n_idle : n_left;
      // the first n_take points of rm publish their lane; the first n_take idle lanes take them
      const bool served = ((rm >> lane) & 1ull) && __popcll(rm & lt_mask) < n_take;
      if (served) handout[__popcll(rm & lt_mask)] = (unsigned)lane;
      rm &= ~__ballot(served);
      remaining -= n_take;
      __threadfence_block();
      const int rank = __popcll(idle & lt_mask);
      if (cur == STOP && rank < n_take) {
        if (armed) SP_FLUSH(P, cn, 1);
        armed = true;
        owner = (int)handout[rank];
        bit = cl;
        o = mk(__uint_as_float(pts[0 * 64 + owner]), __uint_as_float(pts[1 * 64 + owner]),
               __uint_as_float(pts[2 * 64 + owner]));
        light_ray(P.lights + (size_t)(lo + bit) * 12, o, __uint_as_float(pts[3 * 64 + owner]), __uint_as_float(pts[4 * 64 + owner]), d,
                  tlimit);
        ray_setup<PRUNE, LDS>(P, o, d, rd, rb, safe);
#ifdef RT_DEBUG_KNOBS
        if (P.debug_unsafe_mod && (owner + bit) % P.debug_unsafe_mod == 0) safe = false;  // (tests: exact and conservative rays in one wave)
#endif
        sg = make_sign(d);
        sp = 0;
        cur = 0;
      }
      __threadfence_block();  // the hand-out table is rewritten by the next re-arm
      IT_T1(cn, 4);
#ifdef RT_PROFILE_PHASES
      cn.tcyc[5] += 1;
      cn.tcyc[6] += n_take;
#endif
    }
    const int at_node = __popcll(__ballot(cur >= 0)), at_leaf = __popcll(__ballot(cur < 0 && cur != STOP));
    if ((at_node | at_leaf) == 0) break;
#ifdef RT_PROFILE_PHASES
    cn.where[0] += 1; cn.where[1] += at_node; cn.where[2] += at_leaf;
    cn.where[remaining > 0 ? 3 : 4] += 64 - at_node - at_leaf;
#endif
    const bool all_safe = __ballot(cur != STOP && !safe) == 0ull;
    IT_T0;
    if (at_node >= at_leaf) {
      if (cur >= 0) {
        DIV_COUNT(cn, 2);
        inner_step<true, PRUNE, COUNT, LDS>(P, sc, stack, lane, o, d, rd, rb, sg, all_safe, tlimit, cur, sp, cn);
      }
      IT_T1(cn, 2);
    } else if (cur < 0 && cur != STOP) {
      DIV_COUNT(cn, 3);
      float bt = tlimit;
      int bp = -1;
      bool redo;
      // verify_now is PER LANE: a ray that set out on the conservative path asks for the exact verdict of
      // its occluder's leaf box even after a neighbour (re-armed later with a zero direction component,
      // or redoing itself) has sent the whole wave to the plain-division node tests -- the leaves it
      // holds were entered on conservative verdicts.  A ray that has been exact from the root
      // (safe == false when it was armed or redone) never asks.
      leaf_step<true, PRUNE, COUNT, LDS>(P, sc, stack, lane, o, d, rb, safe, redo, tlimit, bt, bp, cur, sp, cn);
      if (bp >= 0) atomicOr(&occl[owner], 1u << bit);
      if (redo) {  // no clear verdict for the occluder's leaf box: this ray again, on the exact path
        safe = false;
        sp = 0;
        cur = 0;
      }
      IT_T1(cn, 3);
    }
  }
  if (armed) SP_FLUSH(P, cn, 1);
  __threadfence_block();
  return hit ? occl[lane] : 0u;
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
// C: the camera constants in RenderParams::hot order (the block's LDS copy, or the kernel argument)
DEVI void get_ray(const float *C, float i, float j, float r1, float r2, V3 &base, V3 &dirn) {
  float lens = C[16];
  V3 eye = ld3(C + 0), u = ld3(C + 3), v = ld3(C + 6);
  if (lens == 0) {
    // pinhole: ru = u * (r1 * 0) - u * (0 / 2) and rv likewise are +0 per component (x * 0 - x * 0
    // for finite x; a negative r1 could make it -0, which eye + (ru + rv) absorbs the same way
    // unless an eye coordinate is itself -0 -- a zero origin's sign reaches no quotient), so
    // base = eye + 0: 3 additions for 24 operations
    base = add(eye, mk(0.0f, 0.0f, 0.0f));
  } else {
    float r1l = r1 * lens, r2l = r2 * lens, l2 = C[17];  // lens / 2, divided on the host
    V3 ru = subt(mk(u.x * r1l, u.y * r1l, u.z * r1l), mk(u.x * l2, u.y * l2, u.z * l2));
    V3 rv = subt(mk(v.x * r2l, v.y * r2l, v.z * r2l), mk(v.x * l2, v.y * l2, v.z * l2));
    base = add(eye, add(ru, rv));
  }
  float vWd = C[12], vHt = C[13];
  float iWd = C[18], iHt = C[19];
  float u_world = i * vWd / iWd - C[14];  // vWd / 2.0 and vHt / 2.0: same IEEE quotients,
  float v_world = j * vHt / iHt - C[15];  // formed once on the host
  V3 u_dir = multiply(u, u_world);
  V3 v_dir = multiply(v, v_world);
  V3 w_dir = ld3(C + 9);  // multiply w (-vDt), formed once on the host (the same three products)
  V3 dir = add(w_dir, add(u_dir, v_dir));
  V3 fdir = subt(add(dir, eye), base);
  dirn = normalize(fdir);
}

// n / d with the host's magic numbers (rt_device.h make_fastdiv); exact for all n < 2^32
DEVI unsigned fdiv(unsigned n, const FastDiv &f) {
  const unsigned q = __umulhi(n, f.magic);
  const unsigned t = ((n - q) >> 1) + q;
  return f.d == 1u ? n : (t >> f.shift);
}

DEVI int wave_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }

constexpr unsigned SLOT_REFRACTION = 0x80000000u;

// Diagnostic build (-DRT_PROFILE_PHASES, never the shipped library): shader-clock stamps
// around the phases of the scheduling loop, summed per wave into the rt_wave_diag area.
#ifdef RT_PROFILE_PHASES
#define PH_DECL unsigned long long ph_start = __builtin_amdgcn_s_memtime(), ph_t = ph_start, ph_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_steps = 0, ph_lanes = 0
#define PH_MARK(i) do { unsigned long long n_ = __builtin_amdgcn_s_memtime(); ph_acc[i] += n_ - ph_t; ph_t = n_; } while (0)
#define PH_MARK_LIT PH_MARK(2)
#elif defined(RT_TRACE_STEPS)
// (time line of single steps: tools/step_trace.py)
#define PH_DECL
#define PH_MARK(i) do { if (gwave < (unsigned long long)STATS_TRACE_WAVES && n_steps <= (unsigned)STATS_TRACE_STEPS && lane == 0) \
    P.stats[STATS_TRACE_BASE + ((size_t)gwave * STATS_TRACE_STEPS + (n_steps - 1)) * STATS_TRACE_MARKS + (i)] = wall_clock64(); } while (0)
#define PH_MARK_LIT PH_MARK(10)
#elif defined(RT_ASM_MARKS)
// (static instruction counts per phase: markers in the assembly listing, tools/phase_static.py)
#define PH_DECL
#define PH_STR2(x) #x
#define PH_STR(x) PH_STR2(x)
#define PH_MARK(i) asm volatile("; PHASE_MARK " #i " line " PH_STR(__LINE__))
#define PH_MARK_LIT PH_MARK(2)
#else
#define PH_DECL
#define PH_MARK(i)
#define PH_MARK_LIT
#endif

#ifndef RT_WAVES_PER_SIMD
#define RT_WAVES_PER_SIMD 4  // register budget: 512 / 4 = 128 VGPRs per lane
#endif

// Exact, order-independent accumulation: a contribution is converted to 2^-34 fixed
// point (int64); integer addition is associative, so a pixel's sum does not depend
// on the order in which the scheduler happens to process its rays (bitwise
// reproducible frames, identical for every tile partition).  A contribution is limited to
// +-2^12, i.e. +-2^46 in fixed point: a pixel's at most 2^16 contributions (1024 samples x
// 63 rays of a depth-6 tree) cannot overflow the 64-bit sum, and one saturated sample still
// drives the pixel average to >= 4 > 1 at 1024 spp.  A non-finite contribution therefore saturates:
// +-inf (the reference can produce k = +inf, RayTracer.hs:141) to +-2^12, still clamped to 1 / 0
// by writePPM like the reference's inf; NaN (that weight times a zero colour: inf * 0) to +2^12 --
// fmaxf/fminf would DROP a NaN operand and turn it into -2^12, cancelling a sibling's +2^12.  The
// reference's pixel is NaN in that case and `truncate NaN` is garbage (Main.hs:131), so there is no
// value to match; NaN pixels are excluded from parity and counted by the oracle.
// The fixed-point number (long long)(c * 2^34) is formed as two 32-bit halves, hi * 2^24 + lo with
// |hi| <= 2^22, |lo| < 2^24 and both of c's sign: x = c * 2^34 is exact (a power of two), h = trunc(x * 2^-24) and the
// remainder x - h * 2^24 are exact floats, and truncating the remainder truncates x.  Up to 64 such
// pairs are summed half by half in 32-bit registers (|sums| < 2^30) -- one DPP add per value and
// scan stage instead of two moves and a 64-bit add -- and joined where a sum leaves the wave.
DEVI void to_fixed2(float c, int &hi, int &lo) {
  c = (c != c) ? 4096.0f : fminf(fmaxf(c, -4096.0f), 4096.0f);
  const float x = c * 17179869184.0f;                              // 2^34
  const float h = __builtin_truncf(x * 5.9604644775390625e-08f);   // 2^-24
  const float l = __builtin_fmaf(-h, 16777216.0f, x);
  hi = (int)h;
  lo = (int)l;
}
DEVI long long join_fixed(int hi, int lo) { return (long long)hi * 16777216ll + (long long)lo; }

// Data-parallel-primitive add: lane i adds the value of the lane that `CTRL` names (row_shr:n =
// 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143), 0 when that lane does not exist or the
// lane's row is not in ROW_MASK.  All 64 lanes must be active.
template <int CTRL, int ROW_MASK = 0xF>
DEVI int dpp_get(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, true);
}

// Packet fetch.  PACKET_SHARDS ticket counters (one 128-B line each); ticket i of shard
// sh is entry i * PACKET_SHARDS + sh of the cost-sorted packet order.  A wave takes a BATCH of ids per atomic (one L2
// round trip per batch instead of per 64 rays); the batch shrinks towards the end of
// the shard ("guided" scheduling) so the tail stays one packet long.
struct PacketSrc {
  int next, end, shard;   // ids next .. end-1 of `shard` are owned by this wave
  int tries;              // shards found exhausted so far
  // a ticket request in flight: the atomic's return value is NOT waited for when it is
  // issued; it is consumed (first use = s_waitcnt) when the current batch has run out, a
  // whole scheduling step later
  bool pending;
  int pend_want;
  unsigned pend_got;      // valid in lane 0
  // order_kernel's totals, read once per wave
  long long n_entries;
  int per_shard;
  long long refr_entries, refl_end;
};
// Order entries are (packet << 7) | code: code 127 = the whole packet, otherwise ONE pixel
// of it (split_refractive).  class_totals[4] = number of entries, [5] = entries of the
// refractive class, both written by order_kernel.
constexpr int ENTRY_WHOLE = 127;
DEVI void packet_src_init(const RenderParams &P, PacketSrc &ps, int shard) {
  ps.next = ps.end = 0;
  ps.shard = shard;
  ps.tries = 0;
  ps.pending = false;
  ps.pend_want = 0;
  ps.pend_got = 0;
  ps.n_entries = P.n_packets > 0 ? (long long)P.class_totals[4] : 0;  // (no packets: order_kernel did not run)
  ps.per_shard = (int)((ps.n_entries + PACKET_SHARDS - 1) / PACKET_SHARDS);
  ps.refr_entries = (long long)P.class_totals[5];
  ps.refl_end = ps.refr_entries + (long long)P.class_totals[2];
}
// Ask ps.shard for the next batch of tickets ("guided": the batch shrinks towards the end
// of the shard so the tail stays one packet long).  Does not wait for the answer.
DEVI void issue_ticket(const RenderParams &P, int lane, PacketSrc &ps) {
  // estimate of what is left from the last ticket we saw; first call: everything
  int left = ps.per_shard - ps.end;
  int want = left / P.batch_div;
  // tickets are cost-sorted: two at a time while in the refractive class (a packet there
  // can cost 100x the average), at most 4 in the reflective class
  long long g = (long long)ps.end * PACKET_SHARDS;
  int cap = g < ps.refr_entries ? P.refr_cap : (g < ps.refl_end ? P.refl_cap : P.batch_cap);
  want = want < 1 ? 1 : (want > cap ? cap : want);
  unsigned got = 0;
  if (lane == 0) got = atomicAdd(P.tile_counter + ps.shard * 32, (unsigned)want);
  ps.pend_got = got;
  ps.pend_want = want;
  ps.pending = true;
}
DEVI int fetch_packet(const RenderParams &P, int lane, PacketSrc &ps) {
  for (;;) {
    if (ps.next < ps.end) {
      // ticket -> packet through the cost-sorted order (probe_kernel / order_kernel):
      // expensive packets (refractive, then reflective first hits) are handed out first,
      // the frame ends on cheap ones, so no wave is left alone with a 100x packet
      long long g = (long long)ps.next * PACKET_SHARDS + ps.shard;
      ++ps.next;
      // last ticket of the batch: request the next batch now, use it next time
      if (ps.next == ps.end && !ps.pending && ps.tries < PACKET_SHARDS) issue_ticket(P, lane, ps);
      if (g < ps.n_entries) return P.order[g];
      continue;
    }
    if (ps.tries >= PACKET_SHARDS) return -1;
    if (!ps.pending) issue_ticket(P, lane, ps);
    const int got = wave_uniform((int)ps.pend_got);  // the wait for the atomic is here
    const int want = ps.pend_want;
    ps.pending = false;
    if (got < ps.per_shard) {
      ps.next = got;
      ps.end = got + want < ps.per_shard ? got + want : ps.per_shard;
    } else {
      // Shard exhausted.  Do NOT probe the other shards with atomics (at the end of a frame
      // every wave would: thousands of serialised same-address atomics per counter): read
      // all counters with one coherent load, lane s <- shard s, and move to a shard that
      // still has tickets, if any.  Counters only grow, so a stale value can only send us to
      // a shard that turns out exhausted, which brings us back here.
      unsigned v = 0xffffffffu;
      if (lane < PACKET_SHARDS)
        v = __hip_atomic_load(P.tile_counter + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned alive = (unsigned)__ballot(lane < PACKET_SHARDS && v < (unsigned)ps.per_shard);
      ps.next = ps.end = 0;
      if (alive == 0) {
        ps.tries = PACKET_SHARDS;  // nothing left anywhere
      } else {
        // first live shard after our own (rotation spreads the waves over the live shards)
        const unsigned rot = (alive >> ps.shard) | (alive << (PACKET_SHARDS - ps.shard));
        ps.shard = (ps.shard + __builtin_ctz(rot & 0xffffu)) % PACKET_SHARDS;
      }
    }
  }
}

// WPB waves per block: 4, or 8 for scenes whose LDS copy only fits when two (not four) blocks
// share a CU's 160 KB -- the waves never synchronise after the staging barrier, so the block
// size is only a unit of LDS sharing.
// WPS waves per SIMD the registers are budgeted for: 4 (128 VGPRs), or 5 (96 VGPRs, a dozen
// spilled) for the small-scene single-light build, where the fifth wave buys more than the
// spills cost once frames are pipelined (c2 2.39 -> 2.34 ms); everywhere else 5 measured slower.
template <bool PRUNE, bool COUNT, bool LDS, bool MLIGHT, int WPB = WAVES_PER_BLOCK, int WPS = RT_WAVES_PER_SIMD>
__global__ __launch_bounds__(64 * WPB, WPS) void render_kernel(const RenderParams P) {
  extern __shared__ __align__(16) unsigned lds[];
  const int lane = threadIdx.x & 63;
  // The wave's index in the block is the same in all its lanes.  Saying so moves every pointer
  // derived from it (stack, queues, counters) into scalar registers: 15 VGPRs fewer, which the
  // 96-register build needs; the 128-register builds are faster without (more scalar spills).
  const int wib = (WPS > 4 || MLIGHT) ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : (int)(threadIdx.x >> 6);
  const unsigned long long gwave = (unsigned long long)blockIdx.x * WPB + wib;
  // LDS: [scene copy (LDS builds only)] [per-wave: traversal stack | shaded points (multi-light builds) | level counters]
  const float4 *sc = reinterpret_cast<const float4 *>(lds);
  // (LDS builds also hold the material table behind the records: per-lane gathers of the
  //  12 material floats then cost an LDS read instead of a trip through TA/L1)
  const int rec_words = LDS ? P.lds_rec_f4 * 4 : 0;  // wide or binary records, primitive records, leaf boxes
  const int mat_words = LDS ? P.n_materials * MAT_F : 0;
  const int scene_words = rec_words + mat_words;
  const float *ldsM = reinterpret_cast<const float *>(lds + rec_words);
  if (LDS) {
    // the block's one and only barrier: stage the scene (a few KB) once per persistent block --
    // the inner nodes of the tree this build walks (wide records when pruning), then the primitives
    float4 *dst = reinterpret_cast<float4 *>(lds);
    if (PRUNE) {
      for (int i = threadIdx.x; i < P.n_wnode_recs * 7; i += 64 * WPB) {
        const int rec = i / 7, part = i - rec * 7;
        dst[rec * LDS_WNODE_F4 + part] = P.wnodes[(size_t)rec * WNODE_F4 + part];
      }
    } else {
      for (int i = threadIdx.x; i < P.n_node_recs * 4; i += 64 * WPB) {
        const int rec = i >> 2, part = i & 3;
        dst[rec * LDS_REC_F4 + part] = P.nodes[(size_t)rec * NODE_F4 + part];
      }
    }
    for (int i = threadIdx.x; i < P.n_prim_recs * 4; i += 64 * WPB) {
      const int rec = i >> 2, part = i & 3;
      dst[P.lds_prim_base + rec * LDS_REC_F4 + part] = P.prims[(size_t)rec * PRIM_F4 + part];
    }
    if (PRUNE)
      for (int i = threadIdx.x; i < P.n_prim_recs * 2; i += 64 * WPB) dst[P.lds_leafbox_base + i] = P.leafbox[i];
    for (int i = threadIdx.x; i < P.n_materials * MAT_F; i += 64 * WPB) lds[rec_words + i] = __float_as_uint(P.materials[i]);
    if (threadIdx.x < 24) lds[scene_words + threadIdx.x] = __float_as_uint(P.hot[threadIdx.x]);
    if (threadIdx.x >= 32 && threadIdx.x < 44 && P.n_lights > 0)
      lds[scene_words + threadIdx.x] = __float_as_uint(P.lights[threadIdx.x - 32]);
    __syncthreads();
  }
  // the block's copy of the per-frame vector constants and of the first light (rt_device.h)
  const float *hotL = reinterpret_cast<const float *>(lds + scene_words);
  if (!LDS) {
    if (threadIdx.x < 24) lds[scene_words + threadIdx.x] = __float_as_uint(P.hot[threadIdx.x]);
    if (threadIdx.x >= 32 && threadIdx.x < 44 && P.n_lights > 0)
      lds[scene_words + threadIdx.x] = __float_as_uint(P.lights[threadIdx.x - 32]);
    __syncthreads();
  }
  unsigned *stack = lds + scene_words + HOT_WORDS + (size_t)wib * (P.stack_depth * 64 + P.point_words + 32);
  unsigned *pts = stack + P.stack_depth * 64;  // shaded points of the step (shadow_lights; MLIGHT builds)
  unsigned *cnt = pts + P.point_words;           // cnt[L], L = 2..max_depth : rays queued at level L

  float *Q = P.scratch + gwave * P.scratch_stride;
  long long *acc = P.acc;  // [n_local_pixels][3] fixed point
  const int tile_px = P.tile_w * P.tile_h;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

  // per-wave ray counters: 32 bits are plenty for one wave's share of a frame (and half the
  // scalar registers); they are widened when added to the frame's totals
  unsigned n_primary = 0, n_shadow = 0, n_refl = 0, n_refr = 0, n_phits = 0, n_hits = 0;
  Counters cn = {0, 0, 0};
  // diagnostics (rt_wave_diag): the wave's start time goes straight to memory, not into
  // registers that would stay live for the whole frame
  if (lane == 0) {
    const unsigned long long t0 = wall_clock64();
    P.stats[16 + gwave * 4] = t0;
    P.stats[19 + gwave * 4] = t0;
  }
  unsigned n_steps = 0;
  PH_DECL;

  if (lane < 32) cnt[lane] = 0u;
  // wave-uniform summaries of cnt[] (bit L): which levels hold >= 64 rays / any ray; the
  // level choice then needs no LDS round trips
  unsigned full_mask = 0, nonempty_mask = 0;
  // The NEXT packet is always fetched one step ahead (its ticket atomic and order[] load
  // overlap the current step instead of heading the next one).
  PacketSrc psrc;
  packet_src_init(P, psrc, (int)(gwave % PACKET_SHARDS));
  int cur_packet = P.max_depth >= 1 ? fetch_packet(P, lane, psrc) : -1;
  int cur_chunk = 0;  // a packet = whole pixels; spp > 64 needs several chunks

  for (;;) {
    // ---- choose what to run: deepest full queue, else the next chunk of primaries,
    //      else (drain, no packets left) the SHALLOWEST non-empty queue: its children top
    //      up the next level before that level gets its one partial step, so a drain costs
    //      at most one partial step per level (deepest-first revisits every deeper level
    //      after each shallower one).  No queue can overflow: without a full level every
    //      level holds < 64 rays and a step adds at most 128 to the next (QCAP = 192).
    int level = 0;
    if (full_mask) level = 31 - __clz((int)full_mask);
    else if (cur_packet >= 0) level = 1;
    else if (nonempty_mask) level = __ffs((int)nonempty_mask) - 1;
    level = wave_uniform(level);
    if (level == 0) break;
    ++n_steps;
#ifdef RT_WAVE_PRIO
    // A frame ends on a few deep ray trees (a glass packet: tens of DEPENDENT steps on one wave).
    // Waves that are out of packets, or that run a bounce level, are the ones the frame waits for:
    // the SIMD's arbiter takes their instructions first.
    if (cur_packet < 0 || level >= RT_WAVE_PRIO) __builtin_amdgcn_s_setprio(3);
    else __builtin_amdgcn_s_setprio(0);
#endif
    PH_MARK(6);

    // ---- obtain 64 (or fewer) rays -------------------------------------------------
    bool active;
    V3 o, d, W;
    float s0, s1;
    unsigned pix;  // local pixel index: ltile * tile_px + (pixel in tile)
    int px = 0, py = 0, smp = 0;
    if (level == 1) {
      // (the order[] word is the same in every lane: saying so keeps the tile arithmetic scalar)
      const int packet_word = wave_uniform(cur_packet);
      const int packet_id = packet_word >> 7, code = packet_word & 127;
      const int ltile = (int)fdiv((unsigned)packet_id, P.fd_groups_per_tile), grp = packet_id - ltile * P.groups_per_tile;
      const int gtile = P.rank + ltile * P.nranks;
      const int tyi = (int)fdiv((unsigned)gtile, P.fd_tiles_x);
      const int tx0 = (gtile - tyi * P.tiles_x) * P.tile_w;
      const int ty0 = tyi * P.tile_h;
      int pit;
      bool in_packet;
      if (P.spp <= 64 && code != ENTRY_WHOLE) {  // single-pixel entry
        smp = lane;
        pit = grp * P.px_per_packet + code;
        in_packet = lane < P.spp;
      } else if (P.spp <= 64) {
        int sub = (int)fdiv((unsigned)lane, P.fd_spp);
        smp = lane - sub * P.spp;
        pit = grp * P.px_per_packet + sub;
        in_packet = sub < P.px_per_packet;
      } else {
        smp = cur_chunk * 64 + lane;
        pit = grp;
        in_packet = smp < P.spp;
      }
      const int prow = (int)fdiv((unsigned)pit, P.fd_tile_w);
      px = tx0 + (pit - prow * P.tile_w);
      py = ty0 + prow;
      active = in_packet && (pit < tile_px) && (px < P.W) && (py < P.H);
      pix = (unsigned)(ltile * tile_px + pit);
      W = mk(1.0f, 1.0f, 1.0f);
      o = mk(0, 0, 0); d = mk(0, 0, 1); s0 = 0; s1 = 0;
      PH_MARK(7);
      if (active) {
        // render (RayTracer.hs:18-23): list position k uses grid k mod n_grids
        const unsigned kpix = (unsigned)py * (unsigned)P.W + (unsigned)px;  // < 2^32 (checked on the host)
        const unsigned grid = kpix - fdiv(kpix, P.fd_n_grids) * (unsigned)P.n_grids;  // kpix mod n_grids
        const float *F = P.f6 + ((size_t)grid * (size_t)P.spp + smp) * 6;
        float fi = (float)px, fj = (float)(P.H - 1 - py);
        get_ray(hotL, fi + F[0], fj + F[1], F[2], F[3], o, d);  // colorPixel :47-51
        s0 = F[4];
        s1 = F[5];
      }
      n_primary += __popcll(__ballot(active));
      PH_MARK(8);
#ifdef RT_PROFILE_PHASES
      if (n_steps == 1) ph_acc[10] = __builtin_amdgcn_s_memtime() - ph_start;
#endif
      // Fetch the NEXT packet now, after this packet's sample loads have been consumed: the
      // ticket atomic and the order[] load are then waited for a whole step later, not by
      // the s_waitcnt of the sample loads (vector memory returns in order).
      if (++cur_chunk >= P.chunks_per_packet) {
        cur_packet = fetch_packet(P, lane, psrc);
        cur_chunk = 0;
        if (cur_packet < 0 && lane == 0) P.stats[19 + gwave * 4] = wall_clock64();  // supply ran dry
      }
      PH_MARK(9);
    } else {
      __threadfence_block();  // this wave's earlier queue stores -> visible to these loads
      int have = wave_uniform((int)cnt[level]);
      int take = have < 64 ? have : 64;
      int idx = have - take + lane;
      active = lane < take;
      const float *ql = Q + (size_t)(level - 2) * QFIELDS * QCAP;
      o = mk(0, 0, 0); d = mk(0, 0, 1); W = mk(0, 0, 0); s0 = 0; s1 = 0; pix = 0;
      if (active) {
        o = mk(ql[0 * QCAP + idx], ql[1 * QCAP + idx], ql[2 * QCAP + idx]);
        d = mk(ql[3 * QCAP + idx], ql[4 * QCAP + idx], ql[5 * QCAP + idx]);
        W = mk(ql[6 * QCAP + idx], ql[7 * QCAP + idx], ql[8 * QCAP + idx]);
        s0 = ql[9 * QCAP + idx];
        s1 = ql[10 * QCAP + idx];
        pix = __float_as_uint(ql[11 * QCAP + idx]);
      }
      if (lane == 0) cnt[level] = (unsigned)(have - take);
      if (have - take < 64) full_mask &= ~(1u << level);
      if (have - take == 0) nonempty_mask &= ~(1u << level);
      unsigned long long am = __ballot(active);
      unsigned long long rm = __ballot(active && (pix & SLOT_REFRACTION));
      n_refr += __popcll(rm);
      n_refl += __popcll(am) - __popcll(rm);
      pix &= ~SLOT_REFRACTION;
    }

    PH_MARK(0);
    // ---- raytrace (RayTracer.hs:55-71): closest hit ----------------------------------
    float t;
    int p;
    traverse<false, PRUNE, COUNT, LDS>(P, sc, stack, lane, active, o, d, 0.0f, t, p, cn);
    const bool hit = active && p >= 0;
    PH_MARK(1);
    n_hits += __popcll(__ballot(hit));

    if (level == 1) {
      n_phits += __popcll(__ballot(hit));
      if (P.hit_ids && active) {
        uint32_t id = RT_HIT_NONE;
        if (hit) id = (uint32_t)__float_as_int(prim_rec<LDS>(P, sc, p)[3].z);
        P.hit_ids[((size_t)py * P.W + px) * P.spp + smp] = id;
      }
    }

    // ---- shade -----------------------------------------------------------------------
    V3 pt = mk(0, 0, 0), n = mk(0, 1, 0), clr = mk(0, 0, 0);
    int mat = 0;
    if (hit) {
      const float4 *q = prim_rec<LDS>(P, sc, p);
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
    const float *M = LDS ? ldsM + mat * MAT_F : P.materials + (size_t)mat * MAT_F;
    const V3 v = mk(-d.x, -d.y, -d.z);  // :65

    // getDirectColor (:81-106).  The reference folds the light list from the right:
    // c1 + (c2 + (... + ambient*diffuse)); iterate the lights backwards to add in
    // exactly that order.
    if (hit) clr = cmul(ld3(hotL + 20), ld3(M + 0));  // :91
    const bool cull_unlit = !P.trace_zero_weight;
    if (MLIGHT) {
      // several lights (the host picks this build for them): which lights matter to the point,
      // then all their shadow rays as one pool, then the lit terms in the reference's order
      for (int hi = P.n_lights; hi > 0; hi -= 32) {
        const int lo = hi > 32 ? hi - 32 : 0;
        unsigned need = 0;
        if (hit) {
          need = (hi - lo == 32) ? ~0u : ((1u << (hi - lo)) - 1u);
          if (cull_unlit)
            for (int li = hi - 1; li >= lo; --li) {
              V3 l_dir;
              float max_d;
              light_ray(P.lights + (size_t)li * 12, pt, s0, s1, l_dir, max_d);
              if (!light_matters(l_dir, n, v, M[6])) need &= ~(1u << (li - lo));
            }
        }
        PH_MARK(2);
        const unsigned occluded =
            shadow_lights<PRUNE, COUNT, LDS>(P, sc, stack, pts, lane, lt_mask, hit, pt, s0, s1, need, hi, lo, cn);
        PH_MARK(3);
        if (need) {
          // the point and its light-sample coordinates were parked in LDS for the pool: take them
          // back from there, so that they do not occupy five registers while the pool runs
          pt = mk(__uint_as_float(pts[0 * 64 + lane]), __uint_as_float(pts[1 * 64 + lane]),
                  __uint_as_float(pts[2 * 64 + lane]));
          s0 = __uint_as_float(pts[3 * 64 + lane]);
          s1 = __uint_as_float(pts[4 * 64 + lane]);
        }
        {
          unsigned cnt_rays = (unsigned)__popc(need);  // wave sum below
          for (int dlt = 32; dlt >= 1; dlt >>= 1) cnt_rays += __shfl_xor(cnt_rays, dlt);
          n_shadow += cnt_rays;
        }
        const unsigned lit = need & ~occluded;
        for (int li = hi - 1; li >= lo; --li) {
          if ((lit >> (li - lo)) & 1u) {
            V3 l_dir;
            float max_d;
            light_ray(P.lights + (size_t)li * 12, pt, s0, s1, l_dir, max_d);
            V3 l_c = ld3(P.lights + (size_t)li * 12 + 9);
            float s = hs_max(0.0f, dot(l_dir, n));
            V3 diffuse_c = cscale(cmul(ld3(M + 0), l_c), s);
            V3 h = normalize(add(v, l_dir));
            float pw = phong_pow(hs_max(0.0f, dot(n, h)), M[6]);
            V3 spec_c = cscale(cmul(l_c, ld3(M + 3)), pw);
            clr = add(add(diffuse_c, spec_c), clr);
          }
        }
      }
    } else
    for (int li = P.n_lights - 1; li >= 0; --li) {
      // Light 0 has a copy in LDS.  The pruned, uncounted instances of this build only ever run with at
      // most one light (with_render_kernel: several lights take the MLIGHT instances), and for them the
      // pointer must be the LDS one and nothing else: a select between it and P.lights makes every load
      // of the light a FLAT load, whose s_waitcnt vmcnt(0) then waits, twice per step, for the
      // acknowledgement of every queue store and pixel atomic the wave has in flight.  The counting and
      // the reference-visit instances do run this loop with several lights.
      const float *Lp = (PRUNE && !COUNT) ? hotL + 32 : (li == 0 ? hotL + 32 : P.lights + (size_t)li * 12);
      V3 l_dir;
      float max_d;
      light_ray(Lp, pt, s0, s1, l_dir, max_d);
      // (the half vector is only formed here for the few lanes whose light is behind the surface;
      //  nothing but `need` stays live across the traversal)
      bool need = hit;
      if (cull_unlit && hit && dot(l_dir, n) <= 0) need = light_matters(l_dir, n, v, M[6]);
      float st;
      int sprim;
      PH_MARK(2);
      traverse<true, PRUNE, COUNT, LDS>(P, sc, stack, lane, need, pt, l_dir, max_d, st, sprim, cn);  // inShadow :231-235
      PH_MARK(3);
      n_shadow += __popcll(__ballot(need));
      if (need && sprim < 0) {
        V3 l_c = ld3(Lp + 9);
        float s = hs_max(0.0f, dot(l_dir, n));
        V3 diffuse_c = cscale(cmul(ld3(M + 0), l_c), s);
        V3 h = normalize(add(v, l_dir));
        float pw = phong_pow(hs_max(0.0f, dot(n, h)), M[6]);
        V3 spec_c = cscale(cmul(l_c, ld3(M + 3)), pw);
        clr = add(add(diffuse_c, spec_c), clr);
      }
    }

    PH_MARK_LIT;
    // ---- accumulate weight * direct colour into the PIXEL -----------------------------
    // Inside the chunk, lanes of one pixel are first summed with a segmented wave scan
    // over contiguous runs; the last lane of each run adds the run's sum to the pixel's
    // fixed-point accumulators (zeroed by the host before the launch).
    {
      V3 c = cmul(W, clr);
      const bool contrib = (level == 1) ? active : hit;
      int v[6] = {0, 0, 0, 0, 0, 0};  // (hi, lo) of red, green, blue
      if (contrib) {
        to_fixed2(c.x, v[0], v[1]);
        to_fixed2(c.y, v[2], v[3]);
        to_fixed2(c.z, v[4], v[5]);
      }
      bool tail;
      if (level == 1 && (P.spp == 4 || P.spp == 16 || P.spp == 64)) {
        // primaries: a pixel's samples are spp consecutive lanes starting at a multiple of spp,
        // so the sums run inside (spp 4, 16) or over (spp 64) the 16-lane DPP rows: row shifts
        // (out-of-row sources read 0) instead of LDS-routed shuffles
        const int in_run = lane & (P.spp - 1);
        tail = active && in_run == P.spp - 1;
        if (P.spp == 4) {
          // runs of 4 inside a row: the shifted value must come from the lane's own run
#pragma unroll
          for (int k = 0; k < 6; ++k) { const int u = dpp_get<0x111>(v[k]); v[k] += in_run >= 1 ? u : 0; }
#pragma unroll
          for (int k = 0; k < 6; ++k) { const int u = dpp_get<0x112>(v[k]); v[k] += in_run >= 2 ? u : 0; }
        } else {
          // a run is a whole row (or four): what a row shift brings in from outside the row is 0
#pragma unroll
          for (int k = 0; k < 6; ++k) v[k] += dpp_get<0x111>(v[k]);
#pragma unroll
          for (int k = 0; k < 6; ++k) v[k] += dpp_get<0x112>(v[k]);
#pragma unroll
          for (int k = 0; k < 6; ++k) v[k] += dpp_get<0x114>(v[k]);
#pragma unroll
          for (int k = 0; k < 6; ++k) v[k] += dpp_get<0x118>(v[k]);
        }
        if (P.spp == 64) {
          // lane 15 of each row holds the row's sum: rows 1, 3 add their predecessor's
          // (row_bcast:15), then rows 2, 3 add lane 31's (row_bcast:31); lane 63 = wave sum
#pragma unroll
          for (int k = 0; k < 6; ++k) v[k] += dpp_get<0x142, 0xA>(v[k]);
#pragma unroll
          for (int k = 0; k < 6; ++k) v[k] += dpp_get<0x143, 0xC>(v[k]);
        }
      } else {
        const unsigned key = contrib ? pix : (0xC0000000u | (unsigned)lane);  // unique when idle
        const unsigned key_prev = __shfl_up(key, 1);
        const unsigned key_next = __shfl_down(key, 1);
        const bool head = (lane == 0) || key_prev != key;
        tail = contrib && ((lane == 63) || key_next != key);
        const unsigned long long heads = __ballot(head);
        // level 1: a pixel's lanes are its spp consecutive samples, runs cannot be longer
        const int scan_span = (level == 1 && P.spp < 64) ? P.spp : 64;
        if (~heads) {  // some run is longer than one lane
          const unsigned long long upto = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
          const int start = 63 - __clzll(upto);  // first lane of this lane's run
#pragma unroll
          for (int dlt = 1; dlt < scan_span; dlt <<= 1) {
            int u[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) u[k] = __shfl_up(v[k], dlt);
            if (lane - dlt >= start) {
#pragma unroll
              for (int k = 0; k < 6; ++k) v[k] += u[k];
            }
          }
        }
      }
      // the run's last lane adds the run's sum to the pixel: 64-bit integer atomics without
      // return value (fire and forget; associative, so still order-independent and exact)
#ifndef RT_DEBUG_NO_ACC
      if (tail) {
        const long long cr = join_fixed(v[0], v[1]), cg = join_fixed(v[2], v[3]), cb = join_fixed(v[4], v[5]);
        unsigned long long *a = reinterpret_cast<unsigned long long *>(acc + (size_t)pix * 3);
        if (cr) __hip_atomic_fetch_add(a + 0, (unsigned long long)cr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cg) __hip_atomic_fetch_add(a + 1, (unsigned long long)cg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cb) __hip_atomic_fetch_add(a + 2, (unsigned long long)cb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#else
      if (tail && v[0] == 0x7fffffff) acc[0] = v[1] + v[2];
#endif
    }

    PH_MARK(4);
    // ---- getIndirectColor (:109-156): emit the child rays -----------------------------
    if (level < P.max_depth) {
      bool e_refl = false, e_refr = false;
      V3 refl_v = mk(0, 0, 1), tdir = mk(0, 0, 1), w_refl = mk(0, 0, 0), w_refr = mk(0, 0, 0);
      if (hit) {
        V3 spec = ld3(M + 3);
        float reflx = M[7], nt = M[8];
        V3 f_refl = cscale(spec, reflx);
        e_refl = true;
        if (nt != 0) {
          // Entering (dn < 0): refract dir n nt, k = 1, c = -dn.  Leaving: refract dir (-n) (1 / nt),
          // k = exp(-attenuation), c = tdir . n (:134-143).  One body serves both: -n is exact, so
          // dot dir (-n) = -dn and (-n) * (-dn) = n * dn bit for bit; what differs is the index, its
          // reciprocal (material constants, MAT_F) and the side the normal is taken from.
          const float dn = dot(d, n);
          const bool entering = dn < 0;
          const float ntv = entering ? nt : M[14];
          const float n_over_nt = entering ? M[14] : M[15];
          // refract (:162-170), n = 1
          const float x = 1 - (1 - dn * dn) / ntv / ntv;
          const bool has_t = !(x < 0);
          V3 k = mk(1, 1, 1);
          float c = -dn;
          if (has_t) {
            const V3 y = multiply(subt(d, multiply(n, dn)), n_over_nt);
            const float sx = entering ? sqrtf(x) : -sqrtf(x);   // (-n) * sqrt x = n * (-sqrt x)
            tdir = normalize(subt(y, multiply(n, sx)));
          }
          if (!entering) {
            k = mk(M[16], M[17], M[18]);
            c = has_t ? dot(tdir, n) : 0.0f;
          }
          const float r_0 = M[13];
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
        // the jittered reflection direction (:113-121) only where a reflection ray is emitted:
        // waves whose hits are all on non-reflective materials skip it altogether
        if (e_refl) {
          V3 refl0 = reflect(d, n);
          V3 ou, ov;
          orthonormal(refl0, ou, ov);
          float a = M[12];  // 1 / phong
          float ju = -(a / 2) + s0 * a;
          float jv = -(a / 2) + s1 * a;
          refl_v = add(refl0, add(multiply(ou, ju), multiply(ov, jv)));
        }
      }
      unsigned long long m1 = __ballot(e_refl), m2 = __ballot(e_refr);
      if (m1 | m2) {
        const int nl = level + 1;
        int base = wave_uniform((int)cnt[nl]);
        int off = base + __popcll(m1 & lt_mask) + __popcll(m2 & lt_mask);
        float *qn = Q + (size_t)(nl - 2) * QFIELDS * QCAP;
        if (e_refl) {
          qn[0 * QCAP + off] = pt.x; qn[1 * QCAP + off] = pt.y; qn[2 * QCAP + off] = pt.z;
          qn[3 * QCAP + off] = refl_v.x; qn[4 * QCAP + off] = refl_v.y; qn[5 * QCAP + off] = refl_v.z;
          qn[6 * QCAP + off] = w_refl.x; qn[7 * QCAP + off] = w_refl.y; qn[8 * QCAP + off] = w_refl.z;
          qn[9 * QCAP + off] = s0; qn[10 * QCAP + off] = s1;
          qn[11 * QCAP + off] = __uint_as_float(pix);
          ++off;
        }
        if (e_refr) {
          qn[0 * QCAP + off] = pt.x; qn[1 * QCAP + off] = pt.y; qn[2 * QCAP + off] = pt.z;
          qn[3 * QCAP + off] = tdir.x; qn[4 * QCAP + off] = tdir.y; qn[5 * QCAP + off] = tdir.z;
          qn[6 * QCAP + off] = w_refr.x; qn[7 * QCAP + off] = w_refr.y; qn[8 * QCAP + off] = w_refr.z;
          qn[9 * QCAP + off] = s0; qn[10 * QCAP + off] = s1;
          qn[11 * QCAP + off] = __uint_as_float(pix | SLOT_REFRACTION);
        }
        const int now = base + __popcll(m1) + __popcll(m2);
        if (lane == 0) cnt[nl] = (unsigned)now;
        nonempty_mask |= 1u << nl;
        if (now >= 64) full_mask |= 1u << nl;
      }
    }
    PH_MARK(5);
#ifdef RT_PROFILE_PHASES
    if (n_steps == 1) ph_acc[11] = __builtin_amdgcn_s_memtime() - ph_start;
    ph_steps += 1;
    ph_lanes += __popcll(__ballot(active));
#endif
  }  // scheduling loop

  // ---- statistics --------------------------------------------------------------------
#ifdef RT_PROFILE_PHASES
  for (int i = 0; i < 4; ++i) {
    atomicAdd(&P.stats[STATS_DIAG_BASE + 16 + i], cn.iters[i]);
    atomicAdd(&P.stats[STATS_DIAG_BASE + 20 + i], cn.lanes[i]);
  }
  for (int i = 0; i < 8; ++i) atomicAdd(&P.stats[STATS_DIAG_BASE + 112 + i], (unsigned long long)cn.topk[i]);
  if (lane == 0)
    for (int i = 0; i < 7; ++i) atomicAdd(&P.stats[STATS_DIAG_BASE + 24 + i], cn.tcyc[i]);
  if (lane == 0)
    for (int i = 0; i < 9; ++i) atomicAdd(&P.stats[STATS_DIAG_BASE + 120 + i], cn.where[i]);
#endif
  if (lane == 0) {
    const unsigned long long wave_t0 = P.stats[16 + gwave * 4];
    const unsigned long long t_dry = P.stats[19 + gwave * 4];
    unsigned long long life = wall_clock64() - wave_t0;  // 100 MHz ticks (diagnostics)
    atomicAdd(&P.stats[0], (unsigned long long)n_primary);
    atomicAdd(&P.stats[1], (unsigned long long)n_shadow);
    atomicAdd(&P.stats[2], (unsigned long long)n_refl);
    atomicAdd(&P.stats[3], (unsigned long long)n_refr);
    atomicAdd(&P.stats[7], (unsigned long long)n_phits);
    atomicAdd(&P.stats[8], (unsigned long long)n_hits);
    atomicAdd(&P.stats[9], life);
    atomicMax(&P.stats[10], life);
    atomicAdd(&P.stats[11], 1ull);
    // per-wave diagnostics (rt_wave_diag): lifetime ticks, rays processed
    P.stats[16 + gwave * 4] = life;
    P.stats[17 + gwave * 4] = (unsigned long long)n_primary + n_shadow + n_refl + n_refr;
    P.stats[18 + gwave * 4] = ((unsigned long long)n_steps << 40) | (wave_t0 & 0xFFFFFFFFFFull);
    P.stats[19 + gwave * 4] = t_dry - wave_t0;  // when the packet supply ran dry for this wave
#ifdef RT_PROFILE_PHASES
    for (int i = 0; i < 12; ++i) atomicAdd(&P.stats[STATS_DIAG_BASE + i], ph_acc[i]);
    atomicAdd(&P.stats[STATS_DIAG_BASE + 12], ph_steps);
    atomicAdd(&P.stats[STATS_DIAG_BASE + 13], ph_lanes);

#endif
  }
  if (COUNT) {
    atomicAdd(&P.stats[4], (unsigned long long)cn.box);
    atomicAdd(&P.stats[5], (unsigned long long)cn.sph);
    atomicAdd(&P.stats[6], (unsigned long long)cn.tri);
  }
}

// ---- packet cost classes (scheduling only; never affects a pixel value) ---------------
// One packet costs between ~2 rays per sample (diffuse floor) and ~100 (glass: a binary
// ray tree of depth wMaxDepth).  probe_kernel shoots one ray per packet (or per four single-pixel packets) and classifies
// it by the material of the first hit; order_kernel then counting-sorts the packets,
// most expensive class first (stable, no atomics).
//   class 3 refractive | 2 reflective (non-zero reflection weight) | 1 other hit | 0 miss
constexpr int PROBE_CLASSES = 4;

DEVI void decode_packet(const RenderParams &P, int packet, int sub, int &px, int &py, bool &inside) {
  const int tile_px = P.tile_w * P.tile_h;
  const int ltile = packet / P.groups_per_tile, grp = packet - ltile * P.groups_per_tile;
  const int gtile = P.rank + ltile * P.nranks;
  int pit = (P.spp <= 64) ? grp * P.px_per_packet + sub % P.px_per_packet : grp;
  px = (gtile % P.tiles_x) * P.tile_w + pit % P.tile_w;
  py = (gtile / P.tiles_x) * P.tile_h + pit / P.tile_w;
  inside = pit < tile_px && px < P.W && py < P.H;
}

template <bool PRUNE>
__global__ __launch_bounds__(256) void probe_kernel(const RenderParams P) {
  extern __shared__ unsigned lds[];
  __shared__ unsigned hist[PROBE_CLASSES];
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  unsigned *stack = lds + (size_t)wib * (P.stack_depth * 64 + P.point_words + 32);
  if (threadIdx.x < PROBE_CLASSES) hist[threadIdx.x] = 0;
  __syncthreads();
  const int span = P.probe_span;  // packets per block (multiple of 256)
  const int first = blockIdx.x * span;
  Counters cn = {0, 0, 0};
  // probe rays: one through the centre of the packet's middle pixel (P.probe_two: two, through
  // opposite quadrants of two of its pixels); single-pixel packets (64 spp and more) are classified
  // P.probe_group at a time by one ray
  const int two = (P.px_per_packet >= 2 && P.probe_two) ? 1 : 0;
  // single-pixel packets: one ray may stand for `grp` consecutive packets (pixels of one tile row)
  const int grp = P.px_per_packet >= 2 ? 1 : P.probe_group;
  for (int base = first; base < first + span; base += (256 >> two) * grp) {  // uniform trip count per block
    const int packet = base + (int)(threadIdx.x >> two) * grp;
    const int sub = two ? (int)(threadIdx.x & 1) * 3 : (P.px_per_packet >= 2 ? P.px_per_packet / 2 : 0);
    bool inside = false;
    int px = 0, py = 0;
    const bool valid = packet < P.n_packets && packet < first + span;
    // (the ray goes through the group's second packet when there is one)
    const int shot = (grp > 1 && packet + 1 < P.n_packets && packet + 1 < first + span) ? packet + 1 : packet;
    if (valid) decode_packet(P, shot, sub, px, py, inside);
    V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
    if (inside) {
      float fi = (float)px + (two ? 0.25f + 0.5f * (float)(sub & 1) : 0.5f);
      float fj = (float)(P.H - 1 - py) + (two ? 0.25f + 0.5f * (float)(sub >> 1) : 0.5f);
      get_ray(P.hot, fi, fj, 0.5f, 0.5f, o, d);
    }
    float t;
    int p;
    traverse<false, PRUNE, false, false>(P, nullptr, stack, lane, inside, o, d, 0.0f, t, p, cn);
    int cls = 0;
    if (inside && p >= 0) {
      int mat = __float_as_int(P.prims[(size_t)p * PRIM_F4 + 3].y);
      const float *M = P.materials + (size_t)mat * MAT_F;
      float refl = M[7];
      bool mirror = (M[3] * refl != 0) || (M[4] * refl != 0) || (M[5] * refl != 0);
      cls = (M[8] != 0) ? 3 : (mirror ? 2 : 1);
    }
    if (two) cls = max(cls, __shfl_xor(cls, 1));
    if (valid && (!two || (threadIdx.x & 1) == 0)) {
      unsigned n_in = 0;
      for (int j = 0; j < grp; ++j)
        if (packet + j < P.n_packets && packet + j < first + span) {
          P.pclass[packet + j] = (unsigned char)cls;
          ++n_in;
        }
      atomicAdd(&hist[cls], n_in);
    }
  }
  __syncthreads();
  if (threadIdx.x < PROBE_CLASSES) P.block_hist[blockIdx.x * PROBE_CLASSES + threadIdx.x] = hist[threadIdx.x];
}

// Stable counting sort of the packets by class, heaviest first.  Block b owns the same
// packet span as in probe_kernel; its output offsets come from the per-block histograms.
__global__ __launch_bounds__(256) void order_kernel(const RenderParams P) {
  __shared__ unsigned base[PROBE_CLASSES];      // running output offset per class
  __shared__ unsigned red[PROBE_CLASSES][4];    // per-wave partials
  __shared__ unsigned wcount[4][PROBE_CLASSES];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int nb = gridDim.x;
  // the frame's ticket counters and statistics start at zero: cleared here, in a kernel that
  // runs before render_kernel anyway, instead of by three more operations on the stream
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < PACKET_SHARDS * 32; i += 256) P.tile_counter[i] = 0u;
    if (threadIdx.x < 16) P.stats[threadIdx.x] = 0ull;
    if (threadIdx.x < STATS_DIAG_WORDS) P.stats[STATS_DIAG_BASE + threadIdx.x] = 0ull;
  }
  // totals of every class and the counts of the blocks before this one
  unsigned tot[PROBE_CLASSES] = {0, 0, 0, 0}, before[PROBE_CLASSES] = {0, 0, 0, 0};
  for (int b = threadIdx.x; b < nb; b += 256)
    for (int c = 0; c < PROBE_CLASSES; ++c) {
      unsigned v = P.block_hist[b * PROBE_CLASSES + c];
      tot[c] += v;
      if (b < (int)blockIdx.x) before[c] += v;
    }
  // block reduce (8 values): wave shuffles then LDS
  unsigned vals[2 * PROBE_CLASSES];
  for (int c = 0; c < PROBE_CLASSES; ++c) { vals[c] = tot[c]; vals[PROBE_CLASSES + c] = before[c]; }
  __shared__ unsigned part[4][2 * PROBE_CLASSES];
  for (int k = 0; k < 2 * PROBE_CLASSES; ++k) {
    unsigned v = vals[k];
    for (int dlt = 32; dlt >= 1; dlt >>= 1) v += __shfl_xor(v, dlt);
    if (lane == 0) part[wib][k] = v;
  }
  __syncthreads();
  // a refractive packet becomes one entry per pixel (EXP entries), every other packet one
  // (measured: per-pixel entries cost more in 4..16-lane primary chunks than they save in
  //  tail on every shipped workload, so the split is off unless split_refractive is set)
  const unsigned EXP = (P.split_refractive && P.spp <= 64 && P.px_per_packet > 1) ? (unsigned)P.px_per_packet : 1u;
  if (blockIdx.x == 0 && threadIdx.x < PROBE_CLASSES)
    P.class_totals[threadIdx.x] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] +
                                  part[3][threadIdx.x];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    unsigned t3 = part[0][3] + part[1][3] + part[2][3] + part[3][3];
    P.class_totals[4] = (unsigned)P.n_packets + (EXP - 1) * t3;
    P.class_totals[5] = EXP * t3;
  }
  if (threadIdx.x < PROBE_CLASSES) {
    int c = threadIdx.x;
    unsigned heavier = 0;
    for (int h = c + 1; h < PROBE_CLASSES; ++h)
      heavier += (part[0][h] + part[1][h] + part[2][h] + part[3][h]) * (h == 3 ? EXP : 1u);
    base[c] = heavier + (part[0][PROBE_CLASSES + c] + part[1][PROBE_CLASSES + c] + part[2][PROBE_CLASSES + c] +
                         part[3][PROBE_CLASSES + c]) * (c == 3 ? EXP : 1u);
  }
  __syncthreads();
  (void)red;
  const int span = P.probe_span;
  const int first = blockIdx.x * span;
  for (int b0 = first; b0 < first + span; b0 += 256) {
    const int packet = b0 + (int)threadIdx.x;
    const bool valid = packet < P.n_packets && packet < first + span;
    const int cls = valid ? (int)P.pclass[packet] : -1;
    unsigned rank = 0;
    for (int c = 0; c < PROBE_CLASSES; ++c) {
      unsigned long long m = __ballot(cls == c);
      if (cls == c) rank = (unsigned)__popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
      if (lane == 0) wcount[wib][c] = (unsigned)__popcll(m);
    }
    __syncthreads();
    if (valid) {
      unsigned r = rank;
      for (int w = 0; w < wib; ++w) r += wcount[w][cls];
      if (cls == 3 && EXP > 1) {
        unsigned off = base[3] + r * EXP;
        for (unsigned k = 0; k < EXP; ++k) P.order[off + k] = (packet << 7) | (int)k;
      } else {
        P.order[base[cls] + r] = (packet << 7) | ENTRY_WHOLE;
      }
    }
    __syncthreads();
    if (threadIdx.x < PROBE_CLASSES)
      base[threadIdx.x] += (wcount[0][threadIdx.x] + wcount[1][threadIdx.x] + wcount[2][threadIdx.x] +
                            wcount[3][threadIdx.x]) * (threadIdx.x == 3 ? EXP : 1u);
    __syncthreads();
  }
}

// Per-material constants of getIndirectColor / refract (rt_device.h MAT_F), with the operations the
// reference performs per hit (RayTracer.hs:113, 138-141, 145, 166); once per uploaded scene.
__global__ void material_prep_kernel(const float *__restrict__ in12, float *__restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *m = in12 + (size_t)i * 12;
  float *o = out + (size_t)i * MAT_F;
  for (int k = 0; k < 12; ++k) o[k] = m[k];
  const float phong = m[6], nt = m[8];
  o[12] = 1 / phong;
  o[13] = (nt - 1) * (nt - 1) / ((nt + 1) * (nt + 1));
  const float inv_nt = 1 / nt;
  o[14] = inv_nt;
  o[15] = 1 / inv_nt;
  o[16] = expf(-m[9]);
  o[17] = expf(-m[10]);
  o[18] = expf(-m[11]);
  o[19] = 0.0f;
}

// avgColors (Surfaces.hs:15-18) per pixel: sum of the sample colours divided by the
// count.  One thread per local pixel; the sum is the exact fixed-point total rounded
// once to fp32 (the reference rounds after every addition: <= a few ulp apart).
__global__ void resolve_kernel(const RenderParams P) {
  const int tile_px = P.tile_w * P.tile_h;
  const size_t total = (size_t)P.n_local_tiles * tile_px;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int ltile = (int)(i / tile_px), pit = (int)(i % tile_px);
    int gtile = P.rank + ltile * P.nranks;
    int px = (gtile % P.tiles_x) * P.tile_w + pit % P.tile_w;
    int py = (gtile / P.tiles_x) * P.tile_h + pit / P.tile_w;
    if (px >= P.W || py >= P.H) continue;
    // exact sum of all the pixel's contributions, rounded once to fp32
    long long *a = P.acc + i * 3;
    float r = (float)((double)a[0] * 5.820766091346741e-11);  // 2^-34
    float g = (float)((double)a[1] * 5.820766091346741e-11);
    float b = (float)((double)a[2] * 5.820766091346741e-11);
    a[0] = a[1] = a[2] = 0;  // the next frame's sums start from zero (no separate 50 MB clear)
    float nf = (float)P.spp;
    size_t o3 = P.compact ? i * 3 : ((size_t)py * P.W + px) * 3;
    P.rgb[o3 + 0] = r / nf;
    P.rgb[o3 + 1] = g / nf;
    P.rgb[o3 + 2] = b / nf;
  }
}

// More than 64 KB of dynamic LDS (deep trees: the per-lane traversal stacks; the 8-wave blocks
// that share one scene copy) has to be asked for, per kernel and device; a cheap host-side call.
template <class K>
static void allow_lds(K kernel, size_t lds_bytes) {
  if (lds_bytes > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_bytes);
}
template <class K>
static void launch_k(K kernel, int grid, int block, size_t lds_bytes, hipStream_t s, const RenderParams &p) {
  allow_lds(kernel, lds_bytes);
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds_bytes, s, p);
}
template <class K>
static int occupancy_k(K kernel, int block, size_t lds_bytes) {
  int n = 0;
  allow_lds(kernel, lds_bytes);
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, block, lds_bytes) == hipSuccess ? n : 0;
}

// The instantiation the host's choices (rt_api.hip launch()) select; F is called with the kernel
// and the dynamic LDS it needs, so that launch and occupancy query cannot disagree.
template <class F>
static auto with_render_kernel(bool prune, bool count, bool lds_scene, bool mlight, bool five_waves, int block,
                               size_t lds_bytes, size_t lds_scene_bytes, F f) {
  const size_t no_scene = lds_bytes - lds_scene_bytes;
  if (block == 512) {  // LDS scene shared by two 8-wave blocks per CU; pruned, uncounted (host only picks it then)
    if (mlight) return f(&render_kernel<true, false, true, true, 8>, lds_bytes);
    return f(&render_kernel<true, false, true, false, 8>, lds_bytes);
  }
  if (mlight && prune && !count) {
    if (lds_scene) return f(&render_kernel<true, false, true, true>, lds_bytes);
    return f(&render_kernel<true, false, false, true>, no_scene);
  }
  if (five_waves) return f(&render_kernel<true, false, true, false, WAVES_PER_BLOCK, 5>, lds_bytes);
  if (lds_scene && !count) {
    if (prune) return f(&render_kernel<true, false, true, false>, lds_bytes);
    return f(&render_kernel<false, false, true, false>, lds_bytes);
  }
  if (prune) {
    if (count) return f(&render_kernel<true, true, false, false>, no_scene);
    return f(&render_kernel<true, false, false, false>, no_scene);
  }
  if (count) return f(&render_kernel<false, true, false, false>, no_scene);
  return f(&render_kernel<false, false, false, false>, no_scene);
}

static void launch(const RenderParams &p, const LaunchCfg &cfg, hipStream_t s) {
  if (p.n_packets > 0) {
    // probe_kernel: 4 waves per block, traversal stacks only (it reads the scene from memory)
    if (cfg.prune) launch_k(&probe_kernel<true>, p.probe_blocks, 256, cfg.probe_lds_bytes, s, p);
    else launch_k(&probe_kernel<false>, p.probe_blocks, 256, cfg.probe_lds_bytes, s, p);
    hipLaunchKernelGGL(order_kernel, dim3(p.probe_blocks), dim3(256), 0, s, p);
  }
  if (cfg.ev_k0) (void)hipEventRecord(cfg.ev_k0, s);  // around the trace/shade kernel alone
  with_render_kernel(cfg.prune, cfg.count, cfg.lds_scene, cfg.mlight, cfg.five_waves, cfg.block, cfg.lds_bytes,
                     cfg.lds_scene_bytes, [&](auto kernel, size_t lds) {
                       launch_k(kernel, cfg.grid, cfg.block, lds, s, p);
                       return 0;
                     });
  if (cfg.ev_k1) (void)hipEventRecord(cfg.ev_k1, s);
  size_t total = (size_t)p.n_local_tiles * p.tile_w * p.tile_h;
  if (total) {
    // eight pixels per thread: a small grid gets its wave slots sooner beside other contexts' persistent waves
    int grid = (int)((total + 2047) / 2048);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(resolve_kernel, dim3(grid), dim3(256), 0, s, p);
  }
}

static int occupancy(bool prune, bool count, bool lds_scene, bool mlight, bool five_waves, int block, size_t lds_bytes,
                     size_t lds_scene_bytes) {
  return with_render_kernel(prune, count, lds_scene, mlight, five_waves, block, lds_bytes, lds_scene_bytes,
                            [&](auto kernel, size_t lds) { return occupancy_k(kernel, block, lds); });
}

}  // namespace RT_NS
}  // namespace rtdev
