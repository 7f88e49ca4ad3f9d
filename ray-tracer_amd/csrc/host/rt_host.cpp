// rt_host.cpp -- host-side producer of the inputs of the trace/shade path
// (see include/rt_host.h).  Plain C++17, no GPU code.  Compiled with
// -ffp-contract=off so every float operation rounds exactly as GHC's SSE code does.
//
// Reference citations are relative to /root/reference.

#include "rt_host.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <sstream>
#include <random>
#include <string>
#include <vector>

namespace {

struct V3 {
  float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
// Geometry3.hs:6-40
inline V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 subt(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 divide(V3 a, float f) { return v3(a.x / f, a.y / f, a.z / f); }
inline float magnitude(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
inline V3 cross(V3 a, V3 b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline V3 normalize(V3 a) { return divide(a, magnitude(a)); }
inline void put(float o[3], V3 a) { o[0] = a.x; o[1] = a.y; o[2] = a.z; }
inline V3 get(const float o[3]) { return v3(o[0], o[1], o[2]); }

// Haskell's Ord Float min/max: `if x <= y then ...` (GHC.Classes)
inline float hmin(float x, float y) { return (x <= y) ? x : y; }
inline float hmax(float x, float y) { return (x <= y) ? y : x; }

const float kInf = std::numeric_limits<float>::infinity();

// ---------------------------------------------------------------- materials
struct NamedMaterial {
  const char *name;
  float d[3], s[3], phong, refl, refr, atten[3];
};
// Materials.hs:6-91 (colours inlined).
#define WHITE 1, 1, 1
#define RED 1, 0.1f, 0.1f
#define GREEN 0.1f, 1, 0.1f
#define BLUE 0.1f, 0.1f, 1
#define GREY 0.5f, 0.5f, 0.5f
#define DARKGREY 0.1f, 0.1f, 0.1f
#define BLACK 0, 0, 0
#define YELLOW 1, 1, 0.1f
#define GOLD 1, 0.842f, 0.1f
#define PURPLE 0.5f, 0.2f, 0.5f
#define OPAQUE 99, 99, 99
const NamedMaterial kMaterials[] = {
    {"whiteDull", {WHITE}, {WHITE}, 1, 0, 0, {OPAQUE}},
    {"whiteShiny", {WHITE}, {WHITE}, 100, 0.3f, 0, {OPAQUE}},
    {"whiteGlossy", {WHITE}, {WHITE}, 1000, 0.6f, 0, {OPAQUE}},
    {"redDull", {RED}, {RED}, 10, 0, 0, {RED}},
    {"greenDull", {GREEN}, {GREEN}, 10, 0, 0, {GREEN}},
    {"greenShiny", {GREEN}, {GREEN}, 1000, 0.01f, 0, {GREEN}},
    {"blueDull", {BLUE}, {BLUE}, 10, 0, 0, {BLUE}},
    {"whiteMirror", {WHITE}, {WHITE}, 10000, 1, 0, {OPAQUE}},
    {"purpleM", {PURPLE}, {PURPLE}, 10000, 1, 0, {WHITE}},
    {"purpleDull", {PURPLE}, {PURPLE}, 10, 0, 0, {WHITE}},
    {"redM", {RED}, {RED}, 10000, 1.0f, 0, {OPAQUE}},
    {"greenM", {GREEN}, {GREEN}, 10000, 1.0f, 0, {OPAQUE}},
    {"blueM", {BLUE}, {BLUE}, 10000, 1, 0, {OPAQUE}},
    {"blackM", {BLACK}, {GREY}, 10000, 1, 0, {OPAQUE}},
    {"darkgreyM", {DARKGREY}, {WHITE}, 10, 1, 0, {OPAQUE}},
    {"greyM", {GREY}, {GREY}, 10000, 1, 0, {OPAQUE}},
    {"yellowM", {YELLOW}, {YELLOW}, 10, 1, 0, {OPAQUE}},
    {"goldM", {GOLD}, {GOLD}, 10000, 1, 0, {OPAQUE}},
    {"goldD", {GOLD}, {GOLD}, 10, 0, 0, {OPAQUE}},
    {"glass", {BLACK}, {WHITE}, 10000, 1, 1.5f, {WHITE}},
    {"diamond", {BLACK}, {WHITE}, 10000, 1, 2.4f, {WHITE}},
    {"greenGlass", {0, 1, 0}, {0, 1, 0}, 10000, 1, 1.5f, {99, 0, 99}},
    {"greenDiamond", {0, 0.9f, 0}, {0, 0.9f, 0}, 10000, 1, 2.5f, {99, 0, 99}},
};

// makeMaterial (Surfaces.hs:22-24): attenuation is stored as its log.
rt_material make_material(const float d[3], const float s[3], float phong, float refl, float refr,
                          const float atten[3]) {
  rt_material m;
  for (int i = 0; i < 3; ++i) {
    m.diffuse[i] = d[i];
    m.specular[i] = s[i];
    m.log_atten[i] = logf(atten[i]);
  }
  m.phong = phong;
  m.refl = refl;
  m.refr_ix = refr;
  return m;
}

bool named_material(const char *name, rt_material *out) {
  for (const NamedMaterial &nm : kMaterials)
    if (std::strcmp(nm.name, name) == 0) {
      *out = make_material(nm.d, nm.s, nm.phong, nm.refl, nm.refr, nm.atten);
      return true;
    }
  return false;
}

}  // namespace

// ------------------------------------------------------------------- scene
struct rth_scene {
  std::vector<rt_sphere> spheres;
  std::vector<rt_triangle> triangles;
  std::vector<uint32_t> order;  // [Object] list order
  std::vector<rt_material> materials;
  std::vector<std::string> material_names;
  std::vector<rt_light> lights;
  // tree
  std::vector<rt_bvh_node> nodes;
  std::vector<uint32_t> leaf_refs;
  int depth = 0;

  int material(const char *name) {
    for (size_t i = 0; i < material_names.size(); ++i)
      if (material_names[i] == name) return (int)i;
    rt_material m;
    if (!named_material(name, &m)) std::abort();
    materials.push_back(m);
    material_names.push_back(name);
    return (int)materials.size() - 1;
  }
  int material(const rt_material &m, const std::string &tag) {
    for (size_t i = 0; i < material_names.size(); ++i)
      if (material_names[i] == tag) return (int)i;
    materials.push_back(m);
    material_names.push_back(tag);
    return (int)materials.size() - 1;
  }
  void sphere(V3 c, float r, int mat) {
    rt_sphere s;
    put(s.center, c);
    s.radius = r;
    s.material = mat;
    s.object_id = (int32_t)order.size();
    order.push_back((uint32_t)spheres.size());
    spheres.push_back(s);
  }
  void triangle_n(V3 a, V3 b, V3 c, V3 n, int mat) {
    rt_triangle t;
    put(t.a, a);
    put(t.b, b);
    put(t.c, c);
    put(t.n, n);
    t.material = mat;
    t.object_id = (int32_t)order.size();
    order.push_back(RT_REF_TRIANGLE | (uint32_t)triangles.size());
    triangles.push_back(t);
  }
  // calcNormal / makeTriangle (Objects.hs:13-19)
  static V3 calc_normal(V3 a, V3 b, V3 c) { return normalize(cross(subt(b, a), subt(c, a))); }
  void triangle(V3 a, V3 b, V3 c, int mat) { triangle_n(a, b, c, calc_normal(a, b, c), mat); }
  // fromMesh's filter (Convert.hs:20): drop degenerate faces
  bool mesh_triangle(V3 a, V3 b, V3 c, int mat) {
    auto eq = [](V3 p, V3 q) { return p.x == q.x && p.y == q.y && p.z == q.z; };
    V3 n = calc_normal(a, b, c);
    if (eq(a, b) || eq(b, c) || eq(a, c) || std::isnan(n.x) || std::isnan(n.y) || std::isnan(n.z))
      return false;
    triangle_n(a, b, c, n, mat);
    return true;
  }
  // makeParallelPiped (Objects.hs:26-54): 12 triangles, this vertex order.
  void parallelepiped(V3 p0, V3 wd, V3 ht, V3 dp, int m) {
    V3 p1 = add(p0, wd), p2 = add(p0, ht), p3 = add(p1, ht);
    V3 p4 = add(p0, dp), p5 = add(p1, dp), p6 = add(p2, dp), p7 = add(p3, dp);
    triangle(p0, p2, p1, m);
    triangle(p1, p2, p3, m);
    triangle(p0, p1, p5, m);
    triangle(p0, p5, p4, m);
    triangle(p0, p4, p2, m);
    triangle(p2, p4, p6, m);
    triangle(p2, p6, p3, m);
    triangle(p3, p6, p7, m);
    triangle(p3, p7, p5, m);
    triangle(p3, p5, p1, m);
    triangle(p4, p5, p6, m);
    triangle(p5, p7, p6, m);
  }
  void light(V3 corner, V3 a, V3 b, V3 c) {
    rt_light l;
    put(l.corner, corner);
    put(l.edge_a, a);
    put(l.edge_b, b);
    put(l.color, c);
    lights.push_back(l);
  }
};

namespace {

// ------------------------------------------------------------ SAH BVH build
// BoundingVolumeHierarchy.hs:34-94, 135-142, 171-189.
struct Box {
  bool empty;
  float l, r, b, t, n, f;
};
const Box kEmptyBox = {true, 0, 0, 0, 0, 0, 0};

Box surround(const Box &p, const Box &q) {  // :135-142
  if (p.empty) return q;
  if (q.empty) return p;
  return Box{false,         hmin(p.l, q.l), hmax(p.r, q.r), hmin(p.b, q.b),
             hmax(p.t, q.t), hmin(p.n, q.n), hmax(p.f, q.f)};
}
float surface_area(const Box &x) {  // :89-94, including `abs r - l` (NOT abs (r - l))
  if (x.empty) return kInf;
  float w = fabsf(x.r) - x.l;
  float h = fabsf(x.t) - x.b;
  float d = fabsf(x.f) - x.n;
  return 2 * (w * h + w * d + h * d);
}
float mid_box(int axis, const Box &x) {  // :171-175
  if (axis == 0) return (x.l + x.r) / 2;
  if (axis == 1) return (x.b + x.t) / 2;
  return (x.n + x.f) / 2;
}
float min3(float a, float b, float c) { return hmin(hmin(a, b), c); }  // minimum = foldl1 min
float max3(float a, float b, float c) { return hmax(hmax(a, b), c); }

Box to_box(const rth_scene &s, uint32_t ref) {  // :178-189
  if (ref & RT_REF_TRIANGLE) {
    const rt_triangle &t = s.triangles[RT_REF_INDEX(ref)];
    return Box{false,
               min3(t.a[0], t.b[0], t.c[0]),
               max3(t.a[0], t.b[0], t.c[0]),
               min3(t.a[1], t.b[1], t.c[1]),
               max3(t.a[1], t.b[1], t.c[1]),
               min3(t.a[2], t.b[2], t.c[2]),
               max3(t.a[2], t.b[2], t.c[2])};
  }
  const rt_sphere &p = s.spheres[ref];
  float x = p.center[0], y = p.center[1], z = p.center[2], r = p.radius;
  return Box{false, x - r, x + r, y - r, y + r, z - r, z + r};
}

// sahCost (:81-87): 2*t_aabb + t_tri*isa*(sa1*n1 + sa2*n2), t_aabb = 0.3, t_tri = 1
float sah_cost(float isa, int n1, float sa1, int n2, float sa2) {
  const float t_aabb = 0.3f, t_tri = 1.0f;
  return 2 * t_aabb + (t_tri * isa) * (sa1 * (float)n1 + sa2 * (float)n2);
}

struct Item {
  Box box;
  uint32_t ref;
};

// sortOn (midBox axis): stable, ascending.
void sort_on_mid(std::vector<Item> &xs, int axis) {
  std::vector<std::pair<float, Item>> keyed;
  keyed.reserve(xs.size());
  for (const Item &it : xs) keyed.emplace_back(mid_box(axis, it.box), it);
  std::stable_sort(keyed.begin(), keyed.end(),
                   [](const std::pair<float, Item> &a, const std::pair<float, Item> &b) {
                     return a.first < b.first;
                   });
  for (size_t i = 0; i < xs.size(); ++i) xs[i] = keyed[i].second;
}

// bestSplit (:55-72).  Returns (index, cost).
std::pair<int, float> best_split(const std::vector<Item> &xs0, int axis) {
  int n = (int)xs0.size();
  if (n == 1) return {0, kInf};
  std::vector<Item> xs = xs0;
  sort_on_mid(xs, axis);
  Box all = kEmptyBox;
  for (const Item &it : xs) all = surround(all, it.box);  // foldl' surround EmptyBox
  float isa = 1 / surface_area(all);
  // lbs = scanl' surround EmptyBox boxes ; rbs = scanr surround EmptyBox boxes
  std::vector<Box> lbs(n + 1), rbs(n + 1);
  lbs[0] = kEmptyBox;
  for (int i = 0; i < n; ++i) lbs[i + 1] = surround(lbs[i], xs[i].box);
  rbs[n] = kEmptyBox;
  for (int i = n - 1; i >= 0; --i) rbs[i] = surround(xs[i].box, rbs[i + 1]);
  bool have = false;
  int best_i = 0;
  float best_c = kInf;
  for (int i = 0; i <= n; ++i) {
    float c = sah_cost(isa, i, surface_area(lbs[i]), n - i, surface_area(rbs[i]));
    if (std::isnan(c)) continue;  // filter (not . isNaN . snd)
    // minimumBy (comparing snd): keeps the EARLIER element unless compare says GT
    if (!have) {
      have = true;
      best_i = i;
      best_c = c;
    } else if (!(best_c < c) && !(best_c == c)) {  // compare best c == GT
      best_i = i;
      best_c = c;
    }
  }
  if (!have) return {0, kInf};
  return {best_i, best_c};
}

struct Builder {
  rth_scene &s;
  int max_depth = 0;
  explicit Builder(rth_scene &sc) : s(sc) {}

  int make_leaf(int idx, const std::vector<Item> &xs) {
    Box bx = kEmptyBox;
    for (const Item &it : xs) bx = surround(bx, it.box);
    rt_bvh_node &nd = s.nodes[idx];
    nd = rt_bvh_node{bx.l, bx.r, bx.b, bx.t, bx.n, bx.f, -1 - (int32_t)s.leaf_refs.size(),
                     (int32_t)xs.size()};
    for (const Item &it : xs) s.leaf_refs.push_back(it.ref);
    return idx;
  }

  // sahBVH's `go` (:36-51)
  int go(std::vector<Item> &xs, int depth) {
    if (depth > max_depth) max_depth = depth;
    int idx = (int)s.nodes.size();
    s.nodes.push_back(rt_bvh_node{});
    int n = (int)xs.size();
    if (n < 3) return make_leaf(idx, xs);
    float leaf_cost = 1.0f * (float)n;
    int best_axis = 0, best_ix = 0;
    float best_cost = 0;
    for (int a = 0; a < 3; ++a) {
      std::pair<int, float> r = best_split(xs, a);
      if (a == 0 || (!(best_cost < r.second) && !(best_cost == r.second))) {
        best_axis = a;
        best_ix = r.first;
        best_cost = r.second;
      }
    }
    if (best_cost > leaf_cost) return make_leaf(idx, xs);
    sort_on_mid(xs, best_axis);
    std::vector<Item> ls(xs.begin(), xs.begin() + best_ix), rs(xs.begin() + best_ix, xs.end());
    xs.clear();
    xs.shrink_to_fit();
    int l = go(ls, depth + 1);
    int r = go(rs, depth + 1);
    const rt_bvh_node &ln = s.nodes[l], &rn = s.nodes[r];
    Box lb{false, ln.xmin, ln.xmax, ln.ymin, ln.ymax, ln.zmin, ln.zmax};
    Box rb{false, rn.xmin, rn.xmax, rn.ymin, rn.ymax, rn.zmin, rn.zmax};
    Box bx = surround(lb, rb);
    s.nodes[idx] = rt_bvh_node{bx.l, bx.r, bx.b, bx.t, bx.n, bx.f, l, r};
    return idx;
  }
};

// ------------------------------------------------------------------ scenes
void scene_common_tris(rth_scene &s) {  // BenchmarkScene.hs:14-37 / BenchmarkScene2.hs:14-37
  V3 t0a = v3(8, 0, -4), t0b = v3(0, 3, 0), t0c = v3(0, 0, 0);
  V3 t1a = v3(8, 0, -4), t1b = v3(8, 3, -4), t1c = v3(0, 3, 0);
  V3 t2a = v3(-4, 0, 8), t2b = v3(0, 0, 0), t2c = v3(0, 3, 0);
  V3 t3a = v3(-4, 0, 8), t3b = v3(0, 3, 0), t3c = v3(-4, 3, 8);
  int wd = s.material("whiteDull"), gd = s.material("goldD");
  s.triangle_n(v3(-20, 0, -20), v3(20, 0, 20), v3(20, 0, -20), v3(0, 1, 0), wd);
  s.triangle_n(v3(-20, 0, -20), v3(-20, 0, 20), v3(20, 0, 20), v3(0, 1, 0), wd);
  s.triangle(t0a, t0b, t0c, gd);
  s.triangle(t1a, t1b, t1c, gd);
  s.triangle(t2a, t2b, t2c, gd);
  s.triangle(t3a, t3b, t3c, gd);
}

void build_bench1(rth_scene &s) {  // BenchmarkScene.hs:7-79
  s.light(v3(10, 10, 20), v3(5, 0, 0), v3(0, 5, 0), v3(0.5f, 0.5f, 0.5f));
  s.parallelepiped(v3(0, 0, 13), v3(4, 0, 0), v3(0, 4, 0), v3(0, 0, -1), s.material("greenGlass"));
  s.sphere(v3(0, 0, 0), 2, s.material("purpleDull"));
  s.sphere(v3(2, 1, 10), 1, s.material("redM"));
  s.sphere(v3(9, 0.75f, 13), 1, s.material("greenShiny"));
  s.sphere(v3(1, 7.5f, 1), 5, s.material("blueM"));
  s.sphere(v3(10, 2, 4), 1, s.material("glass"));
  scene_common_tris(s);
}

void build_bench2(rth_scene &s) {  // BenchmarkScene2.hs:7-114
  s.light(v3(10, 10, 20), v3(5, 0, 0), v3(0, 5, 0), v3(0.5f, 0.5f, 0.5f));
  int glass = s.material("glass");
  s.parallelepiped(v3(0, 0, 14), v3(4, 0, 0), v3(0, 4, 0), v3(0, 0, -1), glass);
  s.parallelepiped(v3(4, 0, 10), v3(4, 0, 0), v3(0, 4, 0), v3(0, 0, -1), glass);
  s.parallelepiped(v3(8, 0, 6), v3(4, 0, 0), v3(0, 4, 0), v3(0, 0, -1), glass);
  s.parallelepiped(v3(12, 0, 2), v3(4, 0, 0), v3(0, 4, 0), v3(0, 0, -1), s.material("greenGlass"));
  int diamond = s.material("diamond");
  s.parallelepiped(v3(10, 0, 16), v3(1, 0, 0), v3(0, 2, 0), v3(0, 0, -1), diamond);
  s.parallelepiped(v3(12, 0, 14), v3(1, 0, 0), v3(0, 2, 0), v3(0, 0, -1), diamond);
  s.parallelepiped(v3(14, 0, 12), v3(1, 0, 0), v3(0, 2, 0), v3(0, 0, -1), diamond);
  s.parallelepiped(v3(16, 0, 10), v3(1, 0, 0), v3(0, 2, 0), v3(0, 0, -1),
                   s.material("greenDiamond"));
  s.sphere(v3(0, 0, 0), 2, s.material("purpleDull"));
  s.sphere(v3(2, 1, 10), 1, s.material("redM"));
  s.sphere(v3(1, 7.5f, 1), 5, s.material("blueM"));
  s.sphere(v3(10, 2, 4), 1, glass);
  scene_common_tris(s);
}

void build_bench3(rth_scene &s) {  // BenchmarkScene3.hs:7-44
  s.light(v3(25, 0, 25), v3(1, 0, 0), v3(0, 1, 0), v3(0.5f, 0.5f, 0.5f));
  V3 bt0 = v3(-5, -20, 20), bt1 = v3(-5, 20, 20), bt2 = v3(-5, -20, -5);
  V3 bt3 = v3(-5, 20, -5), bt4 = v3(20, -20, -5), bt5 = v3(20, 20, -5);
  int wd = s.material("whiteDull");
  s.triangle(bt0, bt2, bt1, wd);
  s.triangle(bt2, bt3, bt1, wd);
  s.triangle(bt2, bt4, bt3, wd);
  s.triangle(bt4, bt5, bt3, wd);
  const float lo[3] = {-4, 0, 4};      // [-4,0 .. 4]
  const float hi[3] = {12, 14, 16};    // [12,14 .. 16]
  const float neg[3] = {-16, -14, -12};  // [-16,-14 .. -12]
  V3 ex = v3(1, 0, 0), ey = v3(0, 1, 0), ez = v3(0, 0, 1);
  struct G {
    const float *xs, *ys, *zs;
    const char *mat;
  } groups[5] = {{lo, lo, hi, "greenGlass"},
                 {lo, hi, lo, "redM"},
                 {hi, lo, lo, "blueM"},
                 {lo, lo, lo, "greenDiamond"},
                 {lo, neg, lo, "goldM"}};
  for (const G &g : groups) {
    int m = s.material(g.mat);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        for (int k = 0; k < 3; ++k) s.parallelepiped(v3(g.xs[i], g.ys[j], g.zs[k]), ex, ey, ez, m);
  }
}

void build_bench4(rth_scene &s) {  // BenchmarkScene4.hs:9-35
  s.light(v3(20, 20, -20), v3(-1, 1, 1), v3(4, 0, 1), v3(0.8f, 0.8f, 0.8f));
  V3 t0 = v3(50, 0, 50), t1 = v3(50, 0, -50), t2 = v3(-50, 0, 50), t3 = v3(-50, 0, -50);
  int wd = s.material("whiteDull"), gm = s.material("goldM"), rd = s.material("redDull");
  s.triangle(t0, t3, t2, wd);
  s.triangle(t0, t1, t3, wd);
  s.sphere(v3(-40, 7, -20), 1, gm);
  s.sphere(v3(0, 7, -5), 1, gm);
  s.sphere(v3(40, 7, 3), 1, gm);
  // x <- [-50,-45 .. 50], z <- [-20,-18 .. 20], 5 * z / 2 == x
  for (int xi = 0; xi <= 20; ++xi) {
    float x = -50.0f + 5.0f * (float)xi;
    for (int zi = 0; zi <= 20; ++zi) {
      float z = -20.0f + 2.0f * (float)zi;
      if (5 * z / 2 == x) s.parallelepiped(v3(x, 0, z), v3(2, 0, 0), v3(0, 5, 0), v3(0, 0, 1), rd);
    }
  }
}

void build_bench5(rth_scene &s) {  // BenchmarkScene5.hs:9-23
  s.light(v3(20, 0, 0), v3(0, 1, 0), v3(0, 0, 1), v3(0.8f, 0.8f, 0.8f));
  s.sphere(v3(1, 0, 0), 3, s.material("whiteMirror"));
  s.sphere(v3(5, -4, 4), 3, s.material("redM"));
  s.sphere(v3(5, -4, -4), 3, s.material("darkgreyM"));
  s.sphere(v3(5, 4, 4), 3, s.material("greenM"));
  s.sphere(v3(5, 4, -4), 3, s.material("greyM"));
}

void build_kat_tris(rth_scene &s) {  // benchmarks/Main.hs:10-13
  int m = s.material("redM");
  s.triangle(v3(0, 1, 0), v3(1, 0, 0), v3(0, 0, 1), m);
  s.triangle(v3(1, 0, 0), v3(0, 1, 0), v3(0, 0, 1), m);
  s.triangle(v3(4, 0, 0), v3(4, 0, 1), v3(4, 1, 0), m);
  s.light(v3(10, 10, 20), v3(5, 0, 0), v3(0, 5, 0), v3(0.5f, 0.5f, 0.5f));
}

// convertMaterial (Convert.hs:55-78) for the illum models the generated mesh uses.
rt_material mtl_material(int illum, const float kd[3], const float ks[3], float ns, float ni) {
  const float xx[3] = {0, 0, 0}, white[3] = {1, 1, 1}, opaque[3] = {99, 99, 99};
  float phong = (ns == 0) ? 10.0f : ns * 10;
  switch (illum) {
    case 2: return make_material(kd, ks, phong, 0, 0, xx);
    case 3: return make_material(kd, ks, phong, 1, 0, opaque);
    case 4: return make_material(kd, ks, phong, 0, ni, white);
    case 6: return make_material(kd, ks, phong, 1, ni, opaque);
    default: std::abort();
  }
}

// Outward-facing UV sphere made of triangles (no degenerate faces at the poles).
void mesh_uv_sphere(rth_scene &s, V3 c, float r, int nu, int nv, int mat) {
  const float pi = 3.14159265358979323846f;
  auto P = [&](int iu, int iv) {
    float th = pi * (float)iv / (float)nv;            // 0 = +y pole
    float ph = 2 * pi * (float)(iu % nu) / (float)nu;
    return add(c, v3(r * sinf(th) * cosf(ph), r * cosf(th), r * sinf(th) * sinf(ph)));
  };
  for (int iv = 0; iv < nv; ++iv)
    for (int iu = 0; iu < nu; ++iu) {
      V3 p00 = P(iu, iv), p10 = P(iu + 1, iv), p01 = P(iu, iv + 1), p11 = P(iu + 1, iv + 1);
      if (iv != 0) s.mesh_triangle(p00, p10, p11, mat);
      if (iv != nv - 1) s.mesh_triangle(p00, p11, p01, mat);
    }
}

// Torus around the y axis.
void mesh_torus(rth_scene &s, V3 c, float R, float r, int nu, int nv, int mat) {
  const float pi = 3.14159265358979323846f;
  auto P = [&](int iu, int iv) {
    float a = 2 * pi * (float)(iu % nu) / (float)nu;
    float b = 2 * pi * (float)(iv % nv) / (float)nv;
    float rr = R + r * cosf(b);
    return add(c, v3(rr * cosf(a), r * sinf(b), rr * sinf(a)));
  };
  for (int iu = 0; iu < nu; ++iu)
    for (int iv = 0; iv < nv; ++iv) {
      V3 p00 = P(iu, iv), p10 = P(iu + 1, iv), p01 = P(iu, iv + 1), p11 = P(iu + 1, iv + 1);
      s.mesh_triangle(p00, p01, p11, mat);
      s.mesh_triangle(p00, p11, p10, mat);
    }
}

// Tessellated floor, normals +y.
void mesh_floor(rth_scene &s, V3 O, float y, float half, int n, int mat_a, int mat_b) {
  float step = 2 * half / (float)n;
  y += O.y;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      float x0 = O.x - half + step * (float)i, z0 = O.z - half + step * (float)j;
      float x1 = x0 + step, z1 = z0 + step;
      int m = ((i + j) & 1) ? mat_a : mat_b;
      s.mesh_triangle(v3(x0, y, z0), v3(x0, y, z1), v3(x1, y, z1), m);
      s.mesh_triangle(v3(x0, y, z0), v3(x1, y, z1), v3(x1, y, z0), m);
    }
}

// Station-style triangle mesh (stand-in for the missing station.obj, Main.hs:235):
// reflective hub (illum 3), torus ring (illum 3), refractive pods (illum 6 and 4),
// spokes and a checkered floor (illum 2); lights of bench6World (Main.hs:239-246).
//
// The whole scene is translated by (+20,+20,+20) into the positive octant: the
// reference's surfaceArea computes `abs r - l` (BoundingVolumeHierarchy.hs:91-93), which
// is the true extent only when the upper bounds are non-negative; around the origin the
// faithful sahBVH degenerates into leaves of thousands of triangles (measured: 3939),
// which would benchmark the quirk rather than the traversal.  Config "mesh" moves the
// camera of bench6Config by the same offset.
void build_mesh(rth_scene &s, int target_tris) {
  const V3 O = v3(20, 20, 20);
  const float lc[3] = {0.5f, 0.5f, 0.5f};
  const float lp[8][3] = {{10, 20, 0},  {-10, 20, 0},  {0, 20, 10},  {0, 20, -10},
                          {0, -20, 10}, {0, -20, -10}, {10, -20, 0}, {-10, -20, 0}};
  for (int i = 0; i < 8; ++i) s.light(add(O, get(lp[i])), v3(0, 0, 0), v3(0, 0, 0), get(lc));
  const float grey[3] = {0.6f, 0.6f, 0.65f}, white[3] = {1, 1, 1}, blue[3] = {0.2f, 0.3f, 0.9f};
  const float gold[3] = {1, 0.842f, 0.1f}, dark[3] = {0.15f, 0.15f, 0.2f}, black[3] = {0, 0, 0};
  const float red[3] = {0.9f, 0.15f, 0.1f}, half[3] = {0.5f, 0.5f, 0.5f};
  int m_hub = s.material(mtl_material(3, grey, white, 100, 1), "mtl_hub");
  int m_ring = s.material(mtl_material(3, gold, gold, 50, 1), "mtl_ring");
  int m_pod = s.material(mtl_material(6, black, white, 1000, 1.5f), "mtl_pod");
  int m_pod2 = s.material(mtl_material(4, blue, white, 100, 1.3f), "mtl_pod2");
  int m_spoke = s.material(mtl_material(2, red, half, 10, 1), "mtl_spoke");
  int m_fa = s.material(mtl_material(2, white, half, 5, 1), "mtl_floor_a");
  int m_fb = s.material(mtl_material(2, dark, half, 5, 1), "mtl_floor_b");
  // budget: hub 20 %, ring 40 %, 4 pods 20 %, floor 20 %
  int nh = std::max(4, (int)sqrtf(0.20f * (float)target_tris / 4.0f));   // 2*nu*nv, nu = 2 nv
  int nr = std::max(4, (int)sqrtf(0.40f * (float)target_tris / 8.0f));   // 2*nu*nv, nu = 4 nv
  int np = std::max(4, (int)sqrtf(0.05f * (float)target_tris / 4.0f));
  int nf = std::max(2, (int)sqrtf(0.20f * (float)target_tris / 2.0f));
  mesh_uv_sphere(s, add(O, v3(0, 0, 0)), 3, 2 * nh, nh, m_hub);
  mesh_torus(s, add(O, v3(0, 0, 0)), 8, 1, 4 * nr, nr, m_ring);
  mesh_uv_sphere(s, add(O, v3(8, 2.5f, 0)), 1.5f, 2 * np, np, m_pod);
  mesh_uv_sphere(s, add(O, v3(-8, 2.5f, 0)), 1.5f, 2 * np, np, m_pod2);
  mesh_uv_sphere(s, add(O, v3(0, 2.5f, 8)), 1.5f, 2 * np, np, m_pod);
  mesh_uv_sphere(s, add(O, v3(0, 2.5f, -8)), 1.5f, 2 * np, np, m_pod2);
  s.parallelepiped(add(O, v3(-7.5f, -0.25f, -0.25f)), v3(15, 0, 0), v3(0, 0.5f, 0), v3(0, 0, 0.5f), m_spoke);
  s.parallelepiped(add(O, v3(-0.25f, -0.25f, -7.5f)), v3(0.5f, 0, 0), v3(0, 0.5f, 0), v3(0, 0, 15), m_spoke);
  mesh_floor(s, O, -3, 16, nf, m_fa, m_fb);
}


// --------------------------------------------------------- .obj / .mtl import
// HaObj.hs (grammar) + Convert.hs (Mesh -> [Object]).  Line oriented, same dialect:
//   .obj : comments / blank lines, `mtllib <file>`, then objects `o <name>` each made of
//          `v x y z`*, `vt ..`*, `vn x y z`*, `vp ..`*, groups (`usemtl <name>`, optional
//          `s off|<n>`, `f i[/t][/n] j.. k..`+), `l a b`*          (HaObj.hs:76-130)
//   .mtl : `newmtl <name>`, `Ns`, `Ka`, `Kd`, `Ks`, optional `Ni` (default 1.0), `d`,
//          `illum`                                                  (HaObj.hs:148-162)
// makeScene (HaObj.hs:44-54): vertex indices are global over the file (1-based); the
// groups with s == 0 come first, then the smooth ones, each in file order.  convertMesh
// (Convert.hs:29-39) gives every GROUP its own material index; fromMesh (Convert.hs:15-27)
// drops faces with two equal corners or a NaN normal and uses the flat calcNormal.
struct ObjError {
  std::string msg;
};

struct MtlEntry {
  float ns = 0, ka[3] = {0, 0, 0}, kd[3] = {0, 0, 0}, ks[3] = {0, 0, 0}, ni = 1.0f, d = 1.0f;
  int illum = -1;
};

// convertMaterial (Convert.hs:55-78), all eleven illumination models.
rt_material convert_material(const MtlEntry &m) {
  const float xx[3] = {0, 0, 0}, white[3] = {1, 1, 1}, opaque[3] = {99, 99, 99};
  const float phong = (m.ns == 0) ? 10.0f : m.ns * 10;
  const float refr = m.ni;
  switch (m.illum) {
    case 0:
    case 1: return make_material(m.kd, xx, 0, 0, 0, xx);
    case 2: return make_material(m.kd, m.ks, phong, 0, 0, xx);
    case 3: return make_material(m.kd, m.ks, phong, 1, 0, opaque);
    case 4: return make_material(m.kd, m.ks, phong, 0, refr, white);
    case 5: return make_material(m.kd, m.ks, phong, 1, 0, opaque);
    case 6:
    case 7:
    case 8: return make_material(m.kd, m.ks, phong, 1, refr, opaque);
    case 9: return make_material(xx, white, phong, 1, refr, white);
    case 10: return make_material(m.kd, m.ks, phong, 0, 0, xx);
    default: throw ObjError{"invalid illum value " + std::to_string(m.illum)};
  }
}

std::vector<std::string> split_ws(const std::string &line) {
  std::vector<std::string> out;
  std::istringstream is(line);
  std::string t;
  while (is >> t) out.push_back(t);
  return out;
}

// `float` of HaObj.hs:212-215: [+-]digits[.digits][(e|E)[+-]digits]
float parse_float(const std::string &t, const std::string &where) {
  size_t i = 0, n = t.size();
  auto digits = [&]() { size_t k = i; while (i < n && std::isdigit((unsigned char)t[i])) ++i; return i > k; };
  if (i < n && (t[i] == '+' || t[i] == '-')) ++i;
  bool ok = digits();
  if (ok && i < n && t[i] == '.') { ++i; ok = digits(); }
  if (ok && i < n && (t[i] == 'e' || t[i] == 'E')) {
    ++i;
    if (i < n && (t[i] == '+' || t[i] == '-')) ++i;
    ok = digits();
  }
  if (!ok || i != n) throw ObjError{where + ": bad number '" + t + "'"};
  return strtof(t.c_str(), nullptr);
}
int parse_uint(const std::string &t, const std::string &where) {
  if (t.empty() || t.size() > 9) throw ObjError{where + ": bad index '" + t + "'"};
  for (char c : t)
    if (!std::isdigit((unsigned char)c)) throw ObjError{where + ": bad index '" + t + "'"};
  return std::atoi(t.c_str());
}
void parse_v3(const std::vector<std::string> &tok, float out[3], const std::string &where) {
  if (tok.size() != 4) throw ObjError{where + ": unexpected number of floats"};
  for (int i = 0; i < 3; ++i) out[i] = parse_float(tok[1 + i], where);
}

std::map<std::string, MtlEntry> read_mtl(const std::string &path) {
  std::ifstream f(path);
  if (!f) throw ObjError{"cannot open material library '" + path + "'"};
  std::map<std::string, MtlEntry> out;
  std::string line, name;
  MtlEntry cur;
  int have = 0, ln = 0;  // bits: Ns Ka Kd Ks d illum
  auto flush = [&]() {
    if (name.empty()) return;
    if ((have & 0x3f) != 0x3f) throw ObjError{path + ": material '" + name + "' is missing Ns/Ka/Kd/Ks/d/illum"};
    if (!out.count(name)) out[name] = cur;  // Data.Map.fromList keeps the LAST duplicate ...
    else out[name] = cur;                   // ... so later definitions replace earlier ones
  };
  while (std::getline(f, line)) {
    ++ln;
    const std::string where = path + ":" + std::to_string(ln);
    std::vector<std::string> tok = split_ws(line);
    if (tok.empty() || tok[0][0] == '#') continue;
    if (tok[0] == "newmtl") {
      flush();
      size_t at = line.find("newmtl") + 6;
      while (at < line.size() && std::isspace((unsigned char)line[at])) ++at;
      name = line.substr(at);
      while (!name.empty() && (name.back() == '\r' || name.back() == '\n')) name.pop_back();
      cur = MtlEntry();
      have = 0;
    } else if (name.empty()) {
      throw ObjError{where + ": expected newmtl"};
    } else if (tok[0] == "Ns" && tok.size() == 2) { cur.ns = parse_float(tok[1], where); have |= 1; }
    else if (tok[0] == "Ka") { parse_v3(tok, cur.ka, where); have |= 2; }
    else if (tok[0] == "Kd") { parse_v3(tok, cur.kd, where); have |= 4; }
    else if (tok[0] == "Ks") { parse_v3(tok, cur.ks, where); have |= 8; }
    else if (tok[0] == "Ni" && tok.size() == 2) { cur.ni = parse_float(tok[1], where); }
    else if (tok[0] == "d" && tok.size() == 2) { cur.d = parse_float(tok[1], where); have |= 16; }
    else if (tok[0] == "illum" && tok.size() == 2) { cur.illum = parse_uint(tok[1], where); have |= 32; }
    else throw ObjError{where + ": unexpected '" + tok[0] + "'"};
  }
  flush();
  return out;
}

struct FaceGroup {
  std::string mtl;
  int smooth = 0;
  std::vector<int> corners;  // 3 vertex indices (1-based) per face
};

rth_scene *load_obj(const std::string &path) {
  std::ifstream f(path);
  if (!f) throw ObjError{"cannot open '" + path + "'"};
  std::vector<float> verts;  // global vertex list (concat vs)
  std::vector<FaceGroup> groups;
  std::string line, mtllib;
  bool in_object = false, group_open = false;
  int ln = 0;
  while (std::getline(f, line)) {
    ++ln;
    const std::string where = path + ":" + std::to_string(ln);
    std::vector<std::string> tok = split_ws(line);
    if (tok.empty() || tok[0][0] == '#') continue;
    if (mtllib.empty()) {
      if (tok[0] != "mtllib" || tok.size() < 2) throw ObjError{where + ": expected mtllib (HaObj.hs:78)"};
      size_t at = line.find("mtllib") + 6;
      while (at < line.size() && std::isspace((unsigned char)line[at])) ++at;
      mtllib = line.substr(at);
      while (!mtllib.empty() && std::isspace((unsigned char)mtllib.back())) mtllib.pop_back();
      continue;
    }
    if (tok[0] == "o") { in_object = true; group_open = false; continue; }
    if (!in_object) throw ObjError{where + ": expected 'o <name>' (HaObj.hs:83)"};
    if (tok[0] == "v") {
      float v[3];
      parse_v3(tok, v, where);
      verts.insert(verts.end(), v, v + 3);
    } else if (tok[0] == "vn" || tok[0] == "vt" || tok[0] == "vp") {
      for (size_t i = 1; i < tok.size(); ++i) parse_float(tok[i], where);  // parsed, unused (Convert.hs:11)
    } else if (tok[0] == "usemtl") {
      size_t at = line.find("usemtl") + 6;
      while (at < line.size() && std::isspace((unsigned char)line[at])) ++at;
      FaceGroup g;
      g.mtl = line.substr(at);
      while (!g.mtl.empty() && std::isspace((unsigned char)g.mtl.back())) g.mtl.pop_back();
      groups.push_back(g);
      group_open = true;
    } else if (tok[0] == "s") {
      if (!group_open || !groups.back().corners.empty() || tok.size() != 2)
        throw ObjError{where + ": 's' must follow usemtl (HaObj.hs:112)"};
      groups.back().smooth = (tok[1] == "off") ? 0 : parse_uint(tok[1], where);
    } else if (tok[0] == "f") {
      if (!group_open) throw ObjError{where + ": face before usemtl (HaObj.hs:110-115)"};
      if (tok.size() != 4) throw ObjError{where + ": non-triangle face " + std::to_string(tok.size() - 1)};  // Convert.hs:50
      for (int i = 1; i <= 3; ++i) {
        std::string c = tok[i].substr(0, tok[i].find('/'));  // v[/vt][/vn] or v//vn: only v is used
        groups.back().corners.push_back(parse_uint(c, where));
      }
    } else if (tok[0] == "l") {
      // lines are parsed and ignored (HaObj.hs:100-105)
    } else {
      throw ObjError{where + ": unexpected '" + tok[0] + "'"};
    }
  }
  if (mtllib.empty()) throw ObjError{path + ": no mtllib"};
  // the reference opens mtllib relative to the working directory; fall back to the .obj's
  std::string mpath = mtllib;
  if (!std::ifstream(mpath)) {
    size_t slash = path.find_last_of('/');
    if (slash != std::string::npos) mpath = path.substr(0, slash + 1) + mtllib;
  }
  std::map<std::string, MtlEntry> mtls = read_mtl(mpath);
  for (const FaceGroup &g : groups)
    if (g.corners.empty()) throw ObjError{path + ": usemtl '" + g.mtl + "' without faces (many1 ngonFace)"};

  rth_scene *s = new rth_scene();
  try {
    const int nv = (int)(verts.size() / 3);
    int group_index = 0;
    for (int pass = 0; pass < 2; ++pass)          // partition: s == 0 first, then smooth
      for (const FaceGroup &g : groups) {
        if ((g.smooth == 0) != (pass == 0)) continue;
        auto it = mtls.find(g.mtl);
        if (it == mtls.end()) throw ObjError{"\"" + g.mtl + "\" not found"};  // findMaterial
        int mi = s->material(convert_material(it->second), "group" + std::to_string(group_index++));
        for (size_t k = 0; k + 2 < g.corners.size(); k += 3) {
          int ix[3] = {g.corners[k] - 1, g.corners[k + 1] - 1, g.corners[k + 2] - 1};  // convertFaceV
          for (int q : ix)
            if (q < 0 || q >= nv) throw ObjError{path + ": vertex index " + std::to_string(q + 1) + " out of range"};
          V3 a = get(&verts[3 * ix[0]]), b = get(&verts[3 * ix[1]]), c = get(&verts[3 * ix[2]]);
          s->mesh_triangle(a, b, c, mi);  // drops degenerate faces (Convert.hs:20)
        }
      }
  } catch (...) {
    delete s;
    throw;
  }
  return s;
}

const struct {
  const char *name;
  rth_config cfg;
} kConfigs[] = {
    // Main.hs:151-235
    {"bench1", {800, 600, 8, 6, 7, 6, 25, 0.0f, {0, 1, 0}, {20, 5, 20}, {0, 0, 0}}},
    {"bench2", {800, 600, 8, 6, 7, 6, 25, 0.0f, {0, 1, 0}, {25, 10, 25}, {0, 0, 0}}},
    {"bench3", {800, 600, 8, 6, 7, 6, 25, 0.0f, {0, 1, 0}, {25, 0, 25}, {0, 0, 0}}},
    {"bench4", {800, 600, 8, 6, 7, 6, 25, 0.05f, {0, 1, 0}, {50, 5, 0}, {0, 0, 0}}},
    {"bench5", {800, 600, 8, 6, 7, 6, 25, 0.0f, {0, 1, 0}, {20, 0, 0}, {0, 0, 0}}},
    {"bench6", {80, 60, 8, 6, 8, 6, 4, 0.0f, {0, 1, 0}, {18, 12, 18}, {0, 0, 0}}},
    // bench6Config moved with the generated mesh (see build_mesh)
    {"mesh", {80, 60, 8, 6, 8, 6, 4, 0.0f, {0, 1, 0}, {38, 32, 38}, {20, 20, 20}}},
    // scene1C (Main.hs:141-142)
    {"scene1", {800, 600, 8, 6, 7, 6, 25, 0.0f, {0, 1, 0}, {10, 0, 0}, {0, 0, 0}}},
};

}  // namespace

// ------------------------------------------------------------------ C ABI
extern "C" {

rth_scene *rth_scene_builtin(const char *name) {
  if (!name) return nullptr;
  rth_scene *s = new rth_scene();
  std::string n(name);
  if (n == "bench1") build_bench1(*s);
  else if (n == "bench2") build_bench2(*s);
  else if (n == "bench3") build_bench3(*s);
  else if (n == "bench4") build_bench4(*s);
  else if (n == "bench5") build_bench5(*s);
  else if (n == "kat_tris") build_kat_tris(*s);
  else if (n.rfind("mesh:", 0) == 0) {
    int t = std::atoi(n.c_str() + 5);
    if (t < 64 || t > 50000000) {
      delete s;
      return nullptr;
    }
    build_mesh(*s, t);
  } else {
    delete s;
    return nullptr;
  }
  return s;
}

rth_scene *rth_scene_create(const rt_sphere *spheres, int32_t n_spheres,
                            const rt_triangle *triangles, int32_t n_triangles,
                            const uint32_t *order, int32_t n_objects,
                            const rt_material *materials, int32_t n_materials,
                            const rt_light *lights, int32_t n_lights) {
  if (n_spheres < 0 || n_triangles < 0 || n_objects != n_spheres + n_triangles || n_materials < 0 ||
      n_lights < 0)
    return nullptr;
  rth_scene *s = new rth_scene();
  s->spheres.assign(spheres, spheres + n_spheres);
  s->triangles.assign(triangles, triangles + n_triangles);
  s->order.assign(order, order + n_objects);
  s->materials.assign(materials, materials + n_materials);
  s->material_names.resize(n_materials);
  for (int i = 0; i < n_materials; ++i) s->material_names[i] = "user" + std::to_string(i);
  s->lights.assign(lights, lights + n_lights);
  std::vector<char> seen_s(n_spheres, 0), seen_t(n_triangles, 0);
  for (int i = 0; i < n_objects; ++i) {
    uint32_t r = s->order[i];
    uint32_t ix = RT_REF_INDEX(r);
    bool tri = (r & RT_REF_TRIANGLE) != 0;
    if ((tri && (ix >= (uint32_t)n_triangles || seen_t[ix])) ||
        (!tri && (ix >= (uint32_t)n_spheres || seen_s[ix]))) {
      delete s;
      return nullptr;
    }
    int mat;
    if (tri) {
      seen_t[ix] = 1;
      s->triangles[ix].object_id = i;
      mat = s->triangles[ix].material;
    } else {
      seen_s[ix] = 1;
      s->spheres[ix].object_id = i;
      mat = s->spheres[ix].material;
    }
    if (mat < 0 || mat >= n_materials) {
      delete s;
      return nullptr;
    }
  }
  return s;
}

void rth_scene_free(rth_scene *s) { delete s; }

rth_scene *rth_scene_load_obj(const char *obj_path, char *err, size_t err_len) {
  if (err && err_len) err[0] = 0;
  if (!obj_path) return nullptr;
  try {
    return load_obj(obj_path);
  } catch (const ObjError &e) {
    if (err && err_len) std::snprintf(err, err_len, "%s", e.msg.c_str());
  } catch (const std::exception &e) {
    if (err && err_len) std::snprintf(err, err_len, "%s", e.what());
  }
  return nullptr;
}

int rth_scene_add_light(rth_scene *s, const rt_light *light) {
  if (!s || !light) return RT_ERR_INVALID;
  s->lights.push_back(*light);
  return RT_OK;
}

int rth_scene_add_bench6_lights(rth_scene *s) {  // bench6World, Main.hs:239-246
  if (!s) return RT_ERR_INVALID;
  const float lp[8][3] = {{10, 20, 0},  {-10, 20, 0},  {0, 20, 10},  {0, 20, -10},
                          {0, -20, 10}, {0, -20, -10}, {10, -20, 0}, {-10, -20, 0}};
  for (int i = 0; i < 8; ++i) s->light(get(lp[i]), v3(0, 0, 0), v3(0, 0, 0), v3(0.5f, 0.5f, 0.5f));
  return RT_OK;
}

int rth_scene_build_bvh(rth_scene *s, int builder) {
  if (!s) return RT_ERR_INVALID;
  s->nodes.clear();
  s->leaf_refs.clear();
  s->depth = 0;
  if (s->order.empty()) return RT_OK;  // Empty
  std::vector<Item> xs;
  xs.reserve(s->order.size());
  for (uint32_t r : s->order) xs.push_back(Item{to_box(*s, r), r});
  Builder b(*s);
  if (builder == 0) {
    b.go(xs, 1);
  } else if (builder == 1) {  // noBVH
    s->nodes.push_back(rt_bvh_node{});
    b.make_leaf(0, xs);
    b.max_depth = 1;
  } else {
    return RT_ERR_INVALID;
  }
  s->depth = b.max_depth;
  return RT_OK;
}

int rth_scene_view(const rth_scene *s, rt_scene *out) {
  if (!s || !out) return RT_ERR_INVALID;
  out->nodes = s->nodes.data();
  out->n_nodes = (int32_t)s->nodes.size();
  out->leaf_refs = s->leaf_refs.data();
  out->n_refs = (int32_t)s->leaf_refs.size();
  out->spheres = s->spheres.data();
  out->n_spheres = (int32_t)s->spheres.size();
  out->triangles = s->triangles.data();
  out->n_triangles = (int32_t)s->triangles.size();
  out->materials = s->materials.data();
  out->n_materials = (int32_t)s->materials.size();
  out->lights = s->lights.data();
  out->n_lights = (int32_t)s->lights.size();
  return RT_OK;
}

int32_t rth_scene_n_objects(const rth_scene *s) { return s ? (int32_t)s->order.size() : 0; }
int32_t rth_scene_bvh_depth(const rth_scene *s) { return s ? s->depth : 0; }

int rth_default_config(const char *name, rth_config *out) {
  if (!name || !out) return RT_ERR_INVALID;
  for (const auto &c : kConfigs)
    if (std::strcmp(c.name, name) == 0) {
      *out = c.cfg;
      return RT_OK;
    }
  return RT_ERR_INVALID;
}

int rth_config_to_camera(const rth_config *c, rt_camera *out) {
  if (!c || !out) return RT_ERR_INVALID;
  out->img_w = c->image_w;
  out->img_h = c->image_h;
  out->view_w = (float)c->view_w;
  out->view_h = (float)c->view_h;
  out->view_d = (float)c->view_d;
  out->aa = (int32_t)floorf(sqrtf((float)c->anti_aliasing));  // Main.hs:111
  out->lens = c->lens;
  // getCam (Main.hs:122-126)
  V3 e = get(c->eye), lat = get(c->look_at), up = get(c->up);
  V3 w = normalize(subt(e, lat));
  V3 u = normalize(cross(up, w));
  V3 v = cross(w, u);
  put(out->up, up);
  put(out->eye, e);
  put(out->u, u);
  put(out->v, v);
  put(out->w, w);
  out->ambient[0] = out->ambient[1] = out->ambient[2] = 0.1f;  // Main.hs:117
  out->max_depth = c->reflection_depth;
  return RT_OK;
}

int rth_generate_grids(uint64_t seed, int32_t n_grids, int32_t aa, float *out) {
  if (n_grids <= 0 || aa <= 0 || aa > 64 || !out) return RT_ERR_INVALID;
  const int nn = aa * aa;
  std::mt19937_64 stream(seed);                       // plays randomFloats (RayTracer.hs:478-481)
  std::mt19937_64 perm_rng(seed ^ 0x9E3779B97F4A7C15ull);  // plays the per-grid `rngs` (:464)
  auto next_float = [&]() {
    // randomDouble: 53 random bits / 2^53, then double2Float
    double d = (double)(stream() >> 11) / 9007199254740992.0;
    return (float)d;
  };
  std::vector<float> rs(2 * nn);
  std::vector<float> px(nn), py(nn);
  std::vector<int> perm(nn);
  const float nf = (float)aa;
  for (int g = 0; g < n_grids; ++g) {
    for (int i = 0; i < 2 * nn; ++i) rs[i] = next_float();  // chunksOf (2*aa*aa)  (:465)
    // getGridR (:215-222): irs = first nn, jrs = next nn; cells p-major
    for (int p = 0; p < aa; ++p)
      for (int q = 0; q < aa; ++q) {
        int k = p * aa + q;
        px[k] = ((float)p + rs[k]) / nf;
        py[k] = ((float)q + rs[nn + k]) / nf;
      }
    // one permutation pi per grid, applied twice with the SAME generator
    // (getGrids :471-475): dof = pi(aa), ss = pi(dof)
    for (int i = 0; i < nn; ++i) perm[i] = i;
    for (int i = nn - 1; i > 0; --i) {
      int j = (int)(perm_rng() % (uint64_t)(i + 1));
      std::swap(perm[i], perm[j]);
    }
    float *o = out + (size_t)g * nn * 6;
    for (int k = 0; k < nn; ++k) {
      int d = perm[k], s2 = perm[perm[k]];
      o[k * 6 + 0] = px[k];
      o[k * 6 + 1] = py[k];
      o[k * 6 + 2] = px[d];
      o[k * 6 + 3] = py[d];
      o[k * 6 + 4] = px[s2];
      o[k * 6 + 5] = py[s2];
    }
  }
  return RT_OK;
}

// ---- seed-level generateGrids (see include/rt_host.h for the package algorithms followed) ----
namespace {
// MT19937-64 (Matsumoto & Nishimura, mt19937-64.c 2004/9/29), written out rather than taken
// from <random> so that every constant of the published algorithm is visible and testable.
struct PureMT64 {
  static constexpr int NN = 312, MM = 156;
  uint64_t mt[NN];
  int mti;
  explicit PureMT64(uint64_t seed) {                     // init_genrand64
    mt[0] = seed;
    for (mti = 1; mti < NN; ++mti) mt[mti] = 6364136223846793005ull * (mt[mti - 1] ^ (mt[mti - 1] >> 62)) + (uint64_t)mti;
  }
  uint64_t next() {                                      // genrand64_int64
    static const uint64_t mag01[2] = {0ull, 0xB5026F5AA96619E9ull};
    const uint64_t UM = 0xFFFFFFFF80000000ull, LM = 0x7FFFFFFFull;
    if (mti >= NN) {
      int i;
      uint64_t x;
      for (i = 0; i < NN - MM; ++i) {
        x = (mt[i] & UM) | (mt[i + 1] & LM);
        mt[i] = mt[i + MM] ^ (x >> 1) ^ mag01[(int)(x & 1ull)];
      }
      for (; i < NN - 1; ++i) {
        x = (mt[i] & UM) | (mt[i + 1] & LM);
        mt[i] = mt[i + (MM - NN)] ^ (x >> 1) ^ mag01[(int)(x & 1ull)];
      }
      x = (mt[NN - 1] & UM) | (mt[0] & LM);
      mt[NN - 1] = mt[MM - 1] ^ (x >> 1) ^ mag01[(int)(x & 1ull)];
      mti = 0;
    }
    uint64_t x = mt[mti++];
    x ^= (x >> 29) & 0x5555555555555555ull;
    x ^= (x << 17) & 0x71D67FFFEDA60000ull;
    x ^= (x << 37) & 0xFFF7EEE000000000ull;
    x ^= (x >> 43);
    return x;
  }
};
}  // namespace

int rth_mt19937_64(uint64_t seed, int32_t n, uint64_t *out) {
  if (n < 0 || !out) return RT_ERR_INVALID;
  PureMT64 g(seed);
  for (int i = 0; i < n; ++i) out[i] = g.next();
  return RT_OK;
}

int rth_generate_grids_pure64(uint64_t seed, int32_t n_grids, int32_t aa, float *out) {
  if (n_grids <= 0 || aa <= 0 || aa > 64 || !out) return RT_ERR_INVALID;
  const int nn = aa * aa;
  // every draw of the run comes from ONE stream: word i is what the generator reached after i steps
  const size_t n_words = std::max((size_t)2 * nn * n_grids, (size_t)n_grids + nn);
  std::vector<uint64_t> w(n_words);
  PureMT64 g(seed);
  for (uint64_t &x : w) x = g.next();
  std::vector<float> px(nn), py(nn);
  std::vector<int> remaining, perm(nn);
  const float nf = (float)aa;
  for (int k = 0; k < n_grids; ++k) {
    // randomFloats (:478-481): double2Float (fromIntegral (w `div` 2048) / 9007199254740992)
    const uint64_t *f = &w[(size_t)2 * nn * k];
    auto rf = [&](int i) { return (float)((double)(f[i] >> 11) / 9007199254740992.0); };
    for (int p = 0; p < aa; ++p)                        // getGridR (:215-222), cells p-major
      for (int q = 0; q < aa; ++q) {
        const int c = p * aa + q;
        px[c] = ((float)p + rf(c)) / nf;
        py[c] = ((float)q + rf(nn + c)) / nf;
      }
    // shuffle' xs nn gen_k: r_j = (x_j - minBound) mod (nn - j + 1), x_j = the j-th word from position k on;
    // output element j-1 = the r_j-th of what remains; the last remaining element closes the list
    remaining.resize(nn);
    for (int i = 0; i < nn; ++i) remaining[i] = i;
    for (int j = 1; j < nn; ++j) {
      const uint64_t x = w[(size_t)k + (j - 1)] ^ 0x8000000000000000ull;   // Int x - minBound, as an unsigned number
      const int r = (int)(x % (uint64_t)(nn - j + 1));
      perm[j - 1] = remaining[r];
      remaining.erase(remaining.begin() + r);
    }
    perm[nn - 1] = remaining[0];
    // dof = shuffle' aa .. gen, ss = shuffle' dof .. gen with the same gen (:471-475): position i of
    // dof holds aa[perm[i]], position i of ss holds dof[perm[i]] = aa[perm[perm[i]]]
    float *o = out + (size_t)k * nn * 6;
    for (int i = 0; i < nn; ++i) {
      const int d = perm[i], s2 = perm[perm[i]];
      o[i * 6 + 0] = px[i];
      o[i * 6 + 1] = py[i];
      o[i * 6 + 2] = px[d];
      o[i * 6 + 3] = py[d];
      o[i * 6 + 4] = px[s2];
      o[i * 6 + 5] = py[s2];
    }
  }
  return RT_OK;
}

// clamp (Main.hs:99-101) then truncate (255*)
static inline uint8_t quant(float x) {
  float c = (x < 0) ? 0.0f : ((x > 1) ? 1.0f : x);
  float y = 255 * c;
  if (!(y == y)) return 0;  // NaN: the reference's `truncate NaN` is garbage; we emit 0
  return (uint8_t)(int)y;
}

void rth_quantize(const float *rgb, size_t n, uint8_t *out) {
  for (size_t i = 0; i < n; ++i) out[i] = quant(rgb[i]);
}

int rth_write_ppm(const char *path, int32_t w, int32_t h, const float *rgb) {
  if (!path || !rgb || w <= 0 || h <= 0) return RT_ERR_INVALID;
  FILE *f = std::fopen(path, "w");
  if (!f) return RT_ERR_INVALID;
  std::fprintf(f, "P3\n%d %d 255\n", w, h);
  size_t n = (size_t)w * h * 3;
  for (size_t i = 0; i < n; ++i) std::fprintf(f, "%d ", (int)quant(rgb[i]));
  std::fclose(f);
  return RT_OK;
}

int rth_write_ppm_binary(const char *path, int32_t w, int32_t h, const uint8_t *rgb8) {
  if (!path || !rgb8 || w <= 0 || h <= 0) return RT_ERR_INVALID;
  FILE *f = std::fopen(path, "wb");
  if (!f) return RT_ERR_INVALID;
  std::fprintf(f, "P6\n%d %d\n255\n", w, h);
  size_t n = (size_t)w * h * 3;
  bool ok = std::fwrite(rgb8, 1, n, f) == n;
  std::fclose(f);
  return ok ? RT_OK : RT_ERR_INVALID;
}

int rth_material_by_name(const char *name, rt_material *out) {
  if (!name || !out) return RT_ERR_INVALID;
  return named_material(name, out) ? RT_OK : RT_ERR_INVALID;
}

void rth_make_triangle(const float a[3], const float b[3], const float c[3], int32_t material,
                       rt_triangle *out) {
  V3 n = rth_scene::calc_normal(get(a), get(b), get(c));
  std::memcpy(out->a, a, 12);
  std::memcpy(out->b, b, 12);
  std::memcpy(out->c, c, 12);
  put(out->n, n);
  out->material = material;
  out->object_id = -1;
}

}  // extern "C"
