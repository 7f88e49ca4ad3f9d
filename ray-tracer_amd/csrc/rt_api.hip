// rt_api.hip -- host side of librt_mi355x.so: the C ABI of include/rt_mi355x.h.
// Context, scene validation + repacking into the device layout (rt_device.h), launch
// of the persistent trace/shade kernel, tile assembly and 8-bit quantisation kernels.
// No torch, no CPU fallback: every entry point needs a HIP device.

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "rt_device.h"

using namespace rtdev;

// work memory of rt_build_scene (rt_build.inl), kept between builds
struct BuildBuffers {
  float *box_in = nullptr, *leaf_box6 = nullptr, *node_box = nullptr;   // object boxes (input order), leaf boxes (sorted), internal nodes' boxes
  unsigned long long *keys = nullptr;                                   // Morton keys, unsorted | sorted
  int2 *child = nullptr;                                                // Karras tree: children of internal node i
  int *ints = nullptr;                                                  // parents of internal nodes | of leaves | heights | arrival counters
  int3 *queue = nullptr;                                                // the wide collapse's two level queues
  void *scalars_mem = nullptr, *sort_tmp = nullptr;
  rt_bvh_node *export_nodes = nullptr;
  size_t cap_box_in = 0, cap_leaf_box6 = 0, cap_node_box = 0, cap_keys = 0, cap_child = 0, cap_ints = 0, cap_queue = 0,
         cap_scalars = 0, cap_sort_tmp = 0, cap_export_nodes = 0;
};

struct rt_ctx {
  int device = 0;
  int n_cus = 0;
  hipStream_t stream = nullptr;
  hipStream_t last_launch_stream = nullptr;  // where the newest frame was enqueued (caller's or ours)
  hipEvent_t ev0 = nullptr, ev1 = nullptr, evk0 = nullptr, evk1 = nullptr;
  hipEvent_t evt0 = nullptr, evt1 = nullptr;  // rt_render's whole-call bracket (ms_total)
  bool timed = false;
  std::string err;

  // scene
  float4 *d_nodes = nullptr, *d_prims = nullptr, *d_wnodes = nullptr, *d_leafbox = nullptr;
  float *d_materials12 = nullptr;  // rt_material as uploaded; d_materials holds the MAT_F-float device records
  size_t cap_materials12 = 0;
  float *d_materials = nullptr, *d_lights = nullptr;
  size_t cap_nodes = 0, cap_prims = 0, cap_wnodes = 0, cap_leafbox = 0, cap_materials = 0, cap_lights = 0, cap_f6 = 0;  // bytes allocated
  void *h_stage = nullptr;  // pinned staging buffer of the uploads
  size_t stage_cap = 0;
  int n_lights = 0, has_root = 0, tree_depth = 0, boxes_safe = 0;
  int n_node_recs = 0, n_prim_recs = 0, n_materials = 0;
  float box_absmax = 0.0f;  // largest |coordinate| of any node box
  bool nested = true;       // every child box lies inside its parent's (what the pruned traversal rests on)
  float clear_band = 9.5367431640625e-07f;  // leaf_box_clear's band, 2^-20 (RT_DEBUG_CLEAR_BAND: tests force the re-walk)
  int n_wnode_recs = 0, wide_stack = 0;  // wide records; most stack entries a wide traversal can hold at once
  // refit (rt_refit_async): which object a primitive record came from, which binary record and side a
  // wide slot copies, and the binary records bottom-up by depth; kept on the host by rt_upload_scene,
  // uploaded by the first refit after it
  std::vector<uint32_t> h_prim_src;
  std::vector<int> h_wide_src, h_level_nodes, h_level_start;
  int n_spheres = 0, n_triangles = 0;
  bool refit_ready = false;
  bool device_built = false;   // the tree came from rt_build_scene: no refit tables (rebuild instead), rt_download_tree works
  int device_built_n = 0;      // objects in that tree
  BuildBuffers build;
  uint32_t *d_prim_src = nullptr;
  int *d_wide_src = nullptr, *d_level_nodes = nullptr;
  float *d_prim_box = nullptr;
  rt_sphere *d_spheres = nullptr;
  rt_triangle *d_triangles = nullptr;
  size_t cap_prim_src = 0, cap_wide_src = 0, cap_level_nodes = 0, cap_prim_box = 0, cap_spheres = 0, cap_triangles = 0;
  int lds_scene_mode = -1;  // RT_LDS_SCENE: 0 never, 1/-1 when it fits (tuning knob)
  int five_waves_mode = -1; // RT_FIVE_WAVES: 0 never use the 96-VGPR build
  bool have_scene = false;
  // camera / samples
  rt_camera cam{};
  bool have_cam = false;
  float *d_f6 = nullptr;
  int n_grids = 0, spp = 0;
  bool have_samples = false;
  // partition / tiling
  int rank = 0, nranks = 1;
  rt_tiling tiling{};
  // work memory
  float *d_scratch = nullptr;
  size_t scratch_floats = 0;
  unsigned long long scratch_stride = 0;
  int scratch_waves = 0;
  bool timing = true;      // record the events behind rt_stats.ms_launch / ms_kernel (rt_set_timing)
  bool acc_clean = false;  // every pixel sum of d_acc is zero (left so by resolve_kernel)
  unsigned *d_tile_counter = nullptr;
  unsigned long long *d_stats = nullptr;
  unsigned long long *last_stats = nullptr;  // statistics block of the newest launch (own or static work memory)
  long long *d_acc = nullptr;
  size_t acc_samples = 0;
  unsigned char *d_pclass = nullptr;
  unsigned *d_block_hist = nullptr;
  int *d_order = nullptr;
  size_t order_cap = 0;
  // staging for the host-pointer form
  float *d_rgb = nullptr;
  size_t d_rgb_floats = 0;
  uint32_t *d_ids = nullptr;
  size_t d_ids_count = 0;
  bool prepared = false;
  int stack_depth = 8;
  int grid_div = 1;                // RT_GRID_DIV: launch 1/grid_div of the resident capacity
  int batches = 0;                 // RT_BATCHES
  int refr_cap = 0, refl_cap = 0;  // RT_REFR_CAP / RT_REFL_CAP (0 = default)
  int rearm_min = 0;               // RT_REARM_MIN (0 = default)
  // probe_kernel's sampling density (scheduling only): one ray per multi-pixel packet (two cost c2 1-2 %
  // of the frame and bought no better order), one ray per FOUR single-pixel packets (64 spp: the
  // DOF frame 4.80 -> 4.71 ms; 8 and 16 are no better).  RT_PROBE_TWO, RT_PROBE_GROUP override.
  int probe_two = 0, probe_group = 4;
  int split_mode = 0;   // RT_SPLIT_REFRACTIVE: 0 off (default: measured no gain), 1 on, -1 auto; never changes a pixel
  float box_slop_scale = 1.0f;  // RT_DEBUG_BOX_SLOP: inflates the conservative box tests' slop (tests; >= 1)
  int debug_unsafe_mod = 0;     // RT_DEBUG_UNSAFE_MOD
  int debug_no_range = 0;       // RT_DEBUG_NO_RANGE_PRUNE
  int trust_tree = 0;           // RT_DEBUG_TRUST_TREE: skip the check that leaf boxes bound their objects (tests)
};

// ---- the library's static work memory (rt_render_static_async) --------------------------
// One array per GPU the code object is loaded on, held until the process ends.  In HOST code
// the array's name is only a stand-in (passing it to a kernel compiles and then faults on the
// GPU): its device address is looked up once per device in rt_init (hipGetSymbolAddress) and
// only that pointer is used below.  Layout (all offsets multiples of 256 B):
//   stats | ticket counters | block histograms | packet classes | packet order | pixel sums | ray queues
constexpr size_t ST_MAX_PIXELS = 1u << 18;   // local pixels incl. tile padding (e.g. 640x360, 512x512)
constexpr int ST_MAX_WAVES = 512;            // persistent waves of the launch (the normal path uses up to 5120)
constexpr int ST_MAX_DEPTH = 6;              // wMaxDepth (every shipped config uses 6)
constexpr size_t ST_STATS_BYTES = STATS_WORDS * sizeof(unsigned long long);
constexpr size_t ST_COUNTER_BYTES = PACKET_SHARDS * 32 * sizeof(unsigned);
constexpr size_t ST_HIST_BYTES = (1024 * 4 + 64) * sizeof(unsigned);
constexpr size_t ST_ORDER_ENTRIES = ST_MAX_PIXELS;  // one entry per packet, or per pixel of a refractive packet
constexpr size_t ST_QUEUE_FLOATS = (size_t)ST_MAX_WAVES * (((size_t)(ST_MAX_DEPTH - 1) * QFIELDS * QCAP + 63) & ~(size_t)63);
constexpr size_t st_align(size_t x) { return (x + 255) & ~(size_t)255; }
constexpr size_t ST_OFF_STATS = 0;
constexpr size_t ST_OFF_COUNTER = ST_OFF_STATS + st_align(ST_STATS_BYTES);
constexpr size_t ST_OFF_HIST = ST_OFF_COUNTER + st_align(ST_COUNTER_BYTES);
constexpr size_t ST_OFF_PCLASS = ST_OFF_HIST + st_align(ST_HIST_BYTES);
constexpr size_t ST_OFF_ORDER = ST_OFF_PCLASS + st_align(ST_ORDER_ENTRIES);
constexpr size_t ST_OFF_ACC = ST_OFF_ORDER + st_align(ST_ORDER_ENTRIES * sizeof(int));
constexpr size_t ST_OFF_QUEUES = ST_OFF_ACC + st_align(ST_MAX_PIXELS * 3 * sizeof(long long));
constexpr size_t ST_BYTES = ST_OFF_QUEUES + st_align(ST_QUEUE_FLOATS * sizeof(float));
static_assert(ST_BYTES <= 32u * 1024 * 1024, "static work memory is documented as at most 32 MiB per GPU");

static __device__ __align__(256) unsigned char scratch[ST_BYTES];

namespace {

constexpr int ST_MAX_DEVICES = 64;
unsigned char *g_static_ptr[ST_MAX_DEVICES] = {nullptr};  // device address of `scratch` on each GPU
std::mutex g_static_mutex;

// once per device, with that device current (rt_init)
hipError_t lookup_static_scratch(int device) {
  if (device < 0 || device >= ST_MAX_DEVICES) return hipSuccess;  // such a device simply has no static path
  std::lock_guard<std::mutex> lock(g_static_mutex);
  if (g_static_ptr[device]) return hipSuccess;
  void *p = nullptr;
  hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
  if (e != hipSuccess) return e;
  g_static_ptr[device] = static_cast<unsigned char *>(p);
  return hipSuccess;
}

// The per-frame work memory of one launch: the context's own allocations (prepare) or the
// carved-up static array.
struct WorkMem {
  float *queues = nullptr;
  int queue_waves = 0;
  long long *acc = nullptr;
  size_t acc_pixels = 0;
  unsigned *tile_counter = nullptr;
  unsigned long long *stats = nullptr;
  unsigned char *pclass = nullptr;
  unsigned *block_hist = nullptr;
  int *order = nullptr;
  size_t order_cap = 0;
  bool acc_known_clean = false;  // every pixel sum is zero (left so by the previous resolve_kernel)
  int max_waves = 1 << 30;       // cap of the launch's persistent waves
};

int fail(rt_ctx *c, int code, const char *fmt, ...) {
  if (c) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    c->err = buf;
  }
  return code;
}

#define HIP_TRY(c, expr)                                                                     \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail((c), RT_ERR_DEVICE, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                  __LINE__);                                                                 \
  } while (0)

// Host -> device copies of the context's input tables, batched: every table of one call goes
// through the context's pinned staging buffer and is copied on the CONTEXT's stream, which is
// waited for ONCE (an animation uploads five tables per frame).  The allocations are kept while
// they are large enough, with headroom (an animated scene's tree changes shape from frame to
// frame; growing by a few records must not mean a hipFree -- a device-wide wait of tens of
// milliseconds).  Ordered on the context's stream, the copies cannot overtake the context's frame
// in flight, which still reads the old tables, and do not stall the other contexts of the GPU the
// way a null-stream copy (device-wide synchronisation) would.
struct UploadBatch {
  struct Item { void **dst; size_t *cap; const void *src; size_t bytes; };
  std::vector<Item> items;
  template <class T>
  void add(T **dst, size_t *cap, const void *src, size_t bytes) {
    items.push_back({reinterpret_cast<void **>(dst), cap, bytes ? src : nullptr, bytes ? bytes : 16});  // (empty table: allocate, copy nothing)
  }
  int run(rt_ctx *c) {
    size_t total = 0;
    for (Item &it : items) {
      if (!*it.dst || *it.cap < it.bytes) {
        if (*it.dst) {
          (void)hipFree(*it.dst);
          *it.dst = nullptr;
          *it.cap = 0;
        }
        const size_t want = std::max<size_t>(it.bytes + it.bytes / 2, 4096);
        HIP_TRY(c, hipMalloc(it.dst, want));
        *it.cap = want;
      }
      if (it.src) total += (it.bytes + 255) & ~(size_t)255;
    }
    if (!total) return RT_OK;
    // a frame enqueued on a caller's stream is not ordered before c->stream: wait for it first
    if (c->last_launch_stream && c->last_launch_stream != c->stream)
      HIP_TRY(c, hipStreamSynchronize(c->last_launch_stream));
    if (c->stage_cap < total) {
      if (c->h_stage) (void)hipHostFree(c->h_stage);
      c->h_stage = nullptr;
      c->stage_cap = 0;
      size_t cap = std::max<size_t>(total * 2, 1 << 16);
      HIP_TRY(c, hipHostMalloc(&c->h_stage, cap, hipHostMallocDefault));
      c->stage_cap = cap;
    }
    size_t off = 0;
    for (Item &it : items) {
      if (!it.src) continue;
      char *stage = static_cast<char *>(c->h_stage) + off;
      std::memcpy(stage, it.src, it.bytes);
      HIP_TRY(c, hipMemcpyAsync(*it.dst, stage, it.bytes, hipMemcpyHostToDevice, c->stream));
      off += (it.bytes + 255) & ~(size_t)255;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // the staging buffer is free again
    return RT_OK;
  }
};

void compute_tiling(rt_ctx *c) {
  // ~1024 samples per tile, power-of-two sides, width >= height (mirrored by
  // tile_shape() in ray-tracer_amd/__init__.py for the CPU-side partition tests)
  int spp = std::max(1, c->spp);
  int px = std::max(1, 1024 / spp);
  int lg = 0;
  while ((2 << lg) <= px) ++lg;
  rt_tiling &t = c->tiling;
  t.tile_w = 1 << ((lg + 1) / 2);
  t.tile_h = 1 << (lg / 2);
  t.tiles_x = (c->cam.img_w + t.tile_w - 1) / t.tile_w;
  t.tiles_y = (c->cam.img_h + t.tile_h - 1) / t.tile_h;
  t.n_tiles = t.tiles_x * t.tiles_y;
  t.n_local_tiles = (t.n_tiles > c->rank) ? (t.n_tiles - c->rank + c->nranks - 1) / c->nranks : 0;
}

struct Repack {
  const rt_scene *sc;
  std::vector<float> nodes;  // 16 floats per device node
  std::vector<float> prims;  // 16 floats per record
  std::vector<char> seen;
  std::vector<uint32_t> prim_src;  // leaf_refs entry each primitive record was made from
  std::vector<float> leafbox;      // 8 floats per primitive record: the box of the record's leaf
  int max_depth = 0;
  std::string err;

  static void put_i(float *p, int v) { std::memcpy(p, &v, 4); }
  static float as_f(int v) { float f; std::memcpy(&f, &v, 4); return f; }

  // emits the records of a leaf (reversed list order), returns the leaf ref
  int emit_leaf(const rt_bvh_node &n) {
    int first_ref = -1 - n.left, count = n.right;
    int first_rec = (int)(prims.size() / 16);
    for (int k = count - 1; k >= 0; --k) {
      uint32_t r = sc->leaf_refs[first_ref + k];
      float rec[16] = {0};
      if (r & RT_REF_TRIANGLE) {
        const rt_triangle &t = sc->triangles[RT_REF_INDEX(r)];
        for (int a = 0; a < 3; ++a) {
          rec[0 + a] = t.a[a];
          rec[3 + a] = t.a[a] - t.b[a];  // (a,b,c) = A - B   RayTracer.hs:362
          rec[6 + a] = t.a[a] - t.c[a];  // (d,e,f) = A - C   RayTracer.hs:363
          rec[9 + a] = t.n[a];
        }
        put_i(rec + 12, 1);
        put_i(rec + 13, t.material);
        put_i(rec + 14, t.object_id);
      } else {
        const rt_sphere &s = sc->spheres[r];
        rec[0] = s.center[0];
        rec[1] = s.center[1];
        rec[2] = s.center[2];
        rec[3] = s.radius;
        rec[4] = s.radius * s.radius;  // radius*radius   RayTracer.hs:344
        put_i(rec + 12, 0);
        put_i(rec + 13, s.material);
        put_i(rec + 14, s.object_id);
      }
      put_i(rec + 15, 0);
      prims.insert(prims.end(), rec, rec + 16);
      prim_src.push_back(r);
    }
    put_i(&prims[(size_t)first_rec * 16 + 15], count);
    leafbox.resize(prims.size() / 2, 0.0f);
    for (int k = 0; k < count; ++k) std::memcpy(&leafbox[(size_t)(first_rec + k) * 8], &n.xmin, 24);
    return -1 - first_rec;
  }

  // returns the device ref of host node `idx` (inner -> device node index, leaf -> leaf ref)
  int visit(int idx, int depth) {
    if (!err.empty()) return REF_NONE;
    if (idx < 0 || idx >= sc->n_nodes) { err = "child index out of range"; return REF_NONE; }
    if (seen[idx]) { err = "node reachable twice (not a tree)"; return REF_NONE; }
    seen[idx] = 1;
    if (depth > max_depth) max_depth = depth;
    if (depth > 4096) { err = "tree too deep"; return REF_NONE; }
    const rt_bvh_node &n = sc->nodes[idx];
    if (n.left < 0) {
      long first = -1L - (long)n.left;
      if (n.right < 1 || first + n.right > sc->n_refs) { err = "leaf range out of bounds"; return REF_NONE; }
      for (int k = 0; k < n.right; ++k) {
        uint32_t r = sc->leaf_refs[first + k];
        uint32_t ix = RT_REF_INDEX(r);
        int mat;
        if (r & RT_REF_TRIANGLE) {
          if ((int)ix >= sc->n_triangles) { err = "triangle ref out of range"; return REF_NONE; }
          mat = sc->triangles[ix].material;
        } else {
          if ((int)ix >= sc->n_spheres) { err = "sphere ref out of range"; return REF_NONE; }
          mat = sc->spheres[ix].material;
        }
        if (mat < 0 || mat >= sc->n_materials) { err = "material index out of range"; return REF_NONE; }
      }
      return emit_leaf(n);
    }
    if (n.right < 0) { err = "inner node with negative right child"; return REF_NONE; }
    int me = (int)(nodes.size() / 16);
    nodes.resize(nodes.size() + 16, 0.0f);
    int l = visit(n.left, depth + 1);
    int r = visit(n.right, depth + 1);
    if (!err.empty()) return REF_NONE;
    float *d = &nodes[(size_t)me * 16];
    std::memcpy(d + 0, &sc->nodes[n.left].xmin, 24);
    std::memcpy(d + 6, &sc->nodes[n.right].xmin, 24);
    put_i(d + 12, l);
    put_i(d + 13, r);
    return me;
  }
};

// Collapse of the binary device records (Repack::nodes) into wide records (rt_device.h).
struct WideBuild {
  const std::vector<float> &bin;  // 16 floats per binary record
  std::vector<float> wide;        // 32 floats per wide record
  std::vector<int> src;           // per wide slot: (binary record << 1) | side whose box it copies, -1 = none
  int max_stack = 0;
  // Most entries the traversal stack may ever hold.  A wide node leaves up to 3 siblings on the
  // stack per level instead of the binary tree's 1, and the per-lane stacks are LDS: the cap
  // (set by rt_upload_scene from the LDS budget) is what still lets four blocks share a CU.  Where a
  // path would exceed the cap the collapse stops early (fewer children per node); a binary
  // continuation needs one entry per remaining level, so cap >= the binary tree's depth suffices.
  int cap = 27;
  std::vector<int> height;        // of the binary subtree under each binary record (leaf children = 0)

  int height_of(int bidx) {
    if (height[bidx] >= 0) return height[bidx];
    int h = 0;
    const float *r = &bin[(size_t)bidx * 16];
    for (int k = 0; k < 2; ++k) {
      int ref = get_i(r + 12 + k);
      if (ref >= 0) h = std::max(h, 1 + height_of(ref));
    }
    return height[bidx] = h;
  }

  struct Child { float box[6]; int ref; int from; };
  static constexpr float ABSENT_PLANE = 288230376151711744.0f;  // 2^58
  static int get_i(const float *p) { int v; std::memcpy(&v, p, 4); return v; }
  static float area(const float *b) {
    float w = b[1] - b[0], h = b[3] - b[2], d = b[5] - b[4];
    return 2 * (w * h + w * d + h * d);
  }
  void children_of(int bidx, std::vector<Child> &out) const {
    const float *r = &bin[(size_t)bidx * 16];
    for (int k = 0; k < 2; ++k) {
      int ref = get_i(r + 12 + k);
      if (ref == REF_NONE) continue;
      Child c;
      std::memcpy(c.box, r + 6 * k, 24);
      c.ref = ref;
      c.from = (bidx << 1) | k;
      out.push_back(c);
    }
  }
  // returns the wide index built for binary record `bidx`; `stack_above` = entries that can sit on
  // the traversal stack when this node is entered
  int build(int bidx, int stack_above) {
    std::vector<Child> ch;
    children_of(bidx, ch);
    while (ch.size() < 4) {
      // one more child = one more sibling on the stack for every path below: allowed while the
      // deepest path below (one entry per remaining binary level) still fits under the cap
      int below = 0;
      for (const Child &c : ch)
        if (c.ref >= 0) below = std::max(below, 1 + height_of(c.ref));
      if (stack_above + (int)ch.size() + below > cap) break;
      int best = -1;
      float best_a = -1.0f;
      for (size_t i = 0; i < ch.size(); ++i)
        if (ch[i].ref >= 0) {
          float a = area(ch[i].box);
          if (!(a <= best_a)) { best_a = a; best = (int)i; }  // (NaN areas still get picked)
        }
      if (best < 0) break;
      std::vector<Child> two;
      children_of(ch[best].ref, two);
      ch.erase(ch.begin() + best);
      ch.insert(ch.begin() + best, two.begin(), two.end());
    }
    const int me = (int)(wide.size() / 32);
    wide.resize(wide.size() + 32, 0.0f);
    src.resize(src.size() + 4, -1);
    const int k = (int)ch.size();
    // entering a child leaves at most k - 1 siblings on the stack
    if (stack_above + k - 1 > max_stack) max_stack = stack_above + k - 1;
    std::vector<int> refs(4, REF_NONE);
    for (int i = 0; i < k; ++i)
      refs[i] = ch[i].ref >= 0 ? build(ch[i].ref, stack_above + k - 1) : ch[i].ref;
    float *w = &wide[(size_t)me * 32];
    for (int i = 0; i < 4; ++i) {
      if (i < k) {
        for (int pl = 0; pl < 6; ++pl) w[pl * 4 + i] = ch[i].box[pl];
        src[(size_t)me * 4 + i] = ch[i].from;
      } else {
        // no such child: an inverted box, which the conservative test never enters (rt_kernels.inl);
        // refit leaves it alone (src = -1)
        for (int pl = 0; pl < 6; ++pl) w[pl * 4 + i] = (pl & 1) ? -ABSENT_PLANE : ABSENT_PLANE;
      }
      std::memcpy(w + 24 + i, &refs[i], 4);
    }
    return me;
  }

  // Renumber the records so that the first `top` of them are the top of the tree in BREADTH-first
  // order (the kernel keeps a prefix of the table on chip: "record index < n" must mean "one of the n
  // records nearest the root"); the others keep their depth-first order (a subtree's records stay
  // together in memory).  Record 0 stays the root.
  void breadth_first_top(int top) {
    const int n = (int)(wide.size() / 32);
    if (n <= 2) return;
    std::vector<int> order;   // new index -> old index
    std::vector<char> taken((size_t)n, 0);
    order.reserve((size_t)n);
    order.push_back(0);
    taken[0] = 1;
    for (size_t head = 0; head < order.size() && (int)order.size() < top; ++head) {
      const float *w = &wide[(size_t)order[head] * 32];
      for (int i = 0; i < 4 && (int)order.size() < top; ++i) {
        const int ref = get_i(w + 24 + i);
        if (ref >= 0 && !taken[ref]) { taken[ref] = 1; order.push_back(ref); }
      }
    }
    for (int i = 0; i < n; ++i)
      if (!taken[i]) order.push_back(i);
    std::vector<int> new_of((size_t)n);
    for (int i = 0; i < n; ++i) new_of[order[i]] = i;
    std::vector<float> w2(wide.size());
    std::vector<int> s2(src.size());
    for (int i = 0; i < n; ++i) {
      std::memcpy(&w2[(size_t)i * 32], &wide[(size_t)order[i] * 32], 128);
      for (int k = 0; k < 4; ++k) {
        s2[(size_t)i * 4 + k] = src[(size_t)order[i] * 4 + k];
        int ref = get_i(&w2[(size_t)i * 32 + 24 + k]);
        if (ref >= 0) {
          ref = new_of[ref];
          std::memcpy(&w2[(size_t)i * 32 + 24 + k], &ref, 4);
        }
      }
    }
    wide.swap(w2);
    src.swap(s2);
  }
};

// ---- refit (rt_refit_async): boxes of an unchanged tree from moved objects, on the GPU ----------
// toBox (BoundingVolumeHierarchy.hs:178-189), surround (:135-142) with Haskell's min / max
// (min x y = if x <= y then x else y), minimum / maximum = left folds.
__device__ inline float hmin_d(float x, float y) { return x <= y ? x : y; }
__device__ inline float hmax_d(float x, float y) { return x <= y ? y : x; }

// one thread per primitive record: the record's geometry (as Repack::emit_leaf packs it) and its toBox
__global__ void refit_prims_kernel(float *__restrict__ prims, const uint32_t *__restrict__ prim_src,
                                   const rt_sphere *__restrict__ sph, const rt_triangle *__restrict__ tri,
                                   float *__restrict__ prim_box, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float *rec = prims + (size_t)i * 16;
  float *bx = prim_box + (size_t)i * 6;
  const uint32_t r = prim_src[i];
  if (r & RT_REF_TRIANGLE) {
    const rt_triangle t = tri[RT_REF_INDEX(r)];
    for (int a = 0; a < 3; ++a) {
      rec[0 + a] = t.a[a];
      rec[3 + a] = t.a[a] - t.b[a];
      rec[6 + a] = t.a[a] - t.c[a];
      rec[9 + a] = t.n[a];
      bx[2 * a] = hmin_d(hmin_d(t.a[a], t.b[a]), t.c[a]);
      bx[2 * a + 1] = hmax_d(hmax_d(t.a[a], t.b[a]), t.c[a]);
    }
  } else {
    const rt_sphere p = sph[r];
    for (int a = 0; a < 3; ++a) {
      rec[a] = p.center[a];
      bx[2 * a] = p.center[a] - p.radius;
      bx[2 * a + 1] = p.center[a] + p.radius;
    }
    rec[3] = p.radius;
    rec[4] = p.radius * p.radius;
  }
}

// one thread per (binary record of one depth level, side): the child's box -- a leaf's is the left
// fold of its objects' boxes in LIST order (records are stored reversed), an inner node's the
// union of the two boxes its own record holds (already refitted: deeper levels run first)
__global__ void refit_nodes_kernel(float *__restrict__ nodes, const float *__restrict__ prims,
                                   const float *__restrict__ prim_box, float *__restrict__ leafbox,
                                   const int *__restrict__ level_nodes, int lo, int hi) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= 2 * (hi - lo)) return;
  const int rec = level_nodes[lo + (k >> 1)], side = k & 1;
  float *me = nodes + (size_t)rec * 16;
  const int ref = __float_as_int(me[12 + side]);
  if (ref == REF_NONE) return;
  float b[6];
  if (ref < 0) {
    const int first = -1 - ref, count = __float_as_int(prims[(size_t)first * 16 + 15]);
    const float *p = prim_box + (size_t)(first + count - 1) * 6;
    for (int a = 0; a < 6; ++a) b[a] = p[a];
    for (int j = count - 2; j >= 0; --j) {
      p = prim_box + (size_t)(first + j) * 6;
      for (int a = 0; a < 3; ++a) {
        b[2 * a] = hmin_d(b[2 * a], p[2 * a]);
        b[2 * a + 1] = hmax_d(b[2 * a + 1], p[2 * a + 1]);
      }
    }
    for (int j = 0; j < count; ++j)   // the leaf's own box, with each of its records (exact verdicts)
      for (int a = 0; a < 6; ++a) leafbox[(size_t)(first + j) * 8 + a] = b[a];
  } else {
    const float *ch = nodes + (size_t)ref * 16;
    for (int a = 0; a < 3; ++a) {
      b[2 * a] = hmin_d(ch[2 * a], ch[6 + 2 * a]);
      b[2 * a + 1] = hmax_d(ch[2 * a + 1], ch[6 + 2 * a + 1]);
    }
  }
  for (int a = 0; a < 6; ++a) me[6 * side + a] = b[a];
}

// one thread per wide slot: the box of the binary (record, side) it was collapsed from
__global__ void refit_wide_kernel(float *__restrict__ wide, const float *__restrict__ nodes,
                                  const int *__restrict__ wide_src, int n_slots) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_slots) return;
  const int from = wide_src[k];
  if (from < 0) return;
  const float *b = nodes + (size_t)(from >> 1) * 16 + 6 * (from & 1);
  float *w = wide + (size_t)(k >> 2) * 32 + (k & 3);
  for (int pl = 0; pl < 6; ++pl) w[pl * 4] = b[pl];
}

template <class T>
__global__ void assemble_kernel(const T *__restrict__ gathered, int nranks, size_t stride,
                                T *__restrict__ rgb, int W, int H, int tile_w, int tile_h,
                                int tiles_x, int n_tiles) {
  const int tile_px = tile_w * tile_h;
  size_t total = (size_t)n_tiles * tile_px;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int gt = (int)(i / tile_px), pit = (int)(i % tile_px);
    int px = (gt % tiles_x) * tile_w + pit % tile_w;
    int py = (gt / tiles_x) * tile_h + pit / tile_w;
    if (px >= W || py >= H) continue;
    int rank = gt % nranks, l = gt / nranks;
    const T *s = gathered + (size_t)rank * stride + ((size_t)l * tile_px + pit) * 3;
    T *d = rgb + ((size_t)py * W + px) * 3;
    d[0] = s[0];
    d[1] = s[1];
    d[2] = s[2];
  }
}

// clamp (Main.hs:99-101) then truncate (255*) (Main.hs:131).  NaN -> 0 (the
// reference's `truncate NaN` is undefined garbage).
__global__ void quantize_kernel(const float *__restrict__ in, uint8_t *__restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    float x = in[i];
    float c = (x < 0) ? 0.0f : ((x > 1) ? 1.0f : x);
    float y = 255 * c;
    out[i] = (y == y) ? (uint8_t)(int)y : (uint8_t)0;
  }
}

// State checks and the launch geometry that follows from camera / samples / partition.  No
// device call, no allocation.
int validate(rt_ctx *c) {
  if (!c->have_scene) return fail(c, RT_ERR_STATE, "rt_upload_scene has not been called");
  if (!c->have_cam) return fail(c, RT_ERR_STATE, "rt_set_camera has not been called");
  if (!c->have_samples) return fail(c, RT_ERR_STATE, "rt_set_samples has not been called");
  if (c->spp != c->cam.aa * c->cam.aa)
    return fail(c, RT_ERR_INVALID, "samples have spp=%d but camera aa=%d needs %d", c->spp, c->cam.aa,
                c->cam.aa * c->cam.aa);
  compute_tiling(c);
  c->stack_depth = std::max(4, (c->tree_depth + 3) & ~3);
  c->scratch_stride = (scratch_floats_per_wave(c->cam.max_depth) + 63) & ~63ull;
  const int ppp = std::max(1, 64 / c->spp);
  const size_t n_packets = (size_t)c->tiling.n_local_tiles * ((size_t)(c->tiling.tile_w * c->tiling.tile_h + ppp - 1) / ppp);
  if (n_packets >= (1u << 24)) return fail(c, RT_ERR_LIMIT, "too many packets (%zu)", n_packets);
  return RT_OK;
}
size_t local_pixels(const rt_ctx *c) { return (size_t)c->tiling.n_local_tiles * c->tiling.tile_w * c->tiling.tile_h; }
size_t order_entries(const rt_ctx *c) {
  const int ppp = std::max(1, 64 / c->spp);
  // a refractive packet may expand to one entry per pixel
  return (size_t)c->tiling.n_local_tiles * ((size_t)(c->tiling.tile_w * c->tiling.tile_h + ppp - 1) / ppp) * (size_t)ppp;
}

// The context's own work memory (allocated on first use and whenever it has to grow).
int prepare(rt_ctx *c) {
  int rc = validate(c);
  if (rc) return rc;
  if (c->prepared) return RT_OK;
  size_t need_samples = local_pixels(c);
  if (need_samples > c->acc_samples) {
    if (c->d_acc) (void)hipFree(c->d_acc);
    c->d_acc = nullptr;
    c->acc_samples = 0;
    HIP_TRY(c, hipMalloc((void **)&c->d_acc, need_samples * 3 * sizeof(long long)));
    c->acc_samples = need_samples;
    c->acc_clean = false;
  }
  int waves = c->n_cus * 8 * WAVES_PER_BLOCK;  // upper bound on resident waves
  size_t need = std::max<size_t>(64, (size_t)waves * c->scratch_stride);
  if (need > c->scratch_floats) {
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    c->d_scratch = nullptr;
    c->scratch_floats = 0;
    HIP_TRY(c, hipMalloc((void **)&c->d_scratch, need * sizeof(float)));
    c->scratch_floats = need;
  }
  c->scratch_waves = waves;
  {
    size_t n_entries = order_entries(c);
    if (n_entries > c->order_cap || !c->d_order) {
      if (c->d_order) (void)hipFree(c->d_order);
      if (c->d_pclass) (void)hipFree(c->d_pclass);
      c->d_order = nullptr;
      c->d_pclass = nullptr;
      c->order_cap = 0;
      size_t cap = std::max<size_t>(n_entries, 256);
      HIP_TRY(c, hipMalloc((void **)&c->d_order, cap * sizeof(int)));
      HIP_TRY(c, hipMalloc((void **)&c->d_pclass, cap));
      c->order_cap = cap;
    }
    if (!c->d_block_hist) HIP_TRY(c, hipMalloc((void **)&c->d_block_hist, (1024 * 4 + 64) * sizeof(unsigned)));
  }
  c->prepared = true;
  return RT_OK;
}

WorkMem own_work_memory(const rt_ctx *c) {
  WorkMem m;
  m.queues = c->d_scratch;
  m.queue_waves = c->scratch_waves;
  m.acc = c->d_acc;
  m.acc_pixels = c->acc_samples;
  m.tile_counter = c->d_tile_counter;
  m.stats = c->d_stats;
  m.pclass = c->d_pclass;
  m.block_hist = c->d_block_hist;
  m.order = c->d_order;
  m.order_cap = c->order_cap;
  m.acc_known_clean = c->acc_clean;
  return m;
}

// Enqueue one frame (probe, order, render, resolve kernels) on `s` with the work memory `m`.
// The caller has validated the context (prepare / validate) and sized `m` for it.
int launch(rt_ctx *c, uint32_t flags, float *d_out, int compact, uint32_t *d_ids, hipStream_t s, const WorkMem &m) {
  HIP_TRY(c, hipSetDevice(c->device));
  // A context owns ONE set of per-frame work buffers (pixel sums, ticket counters, packet order,
  // ray queues): it has one frame in flight.  Frames on the same stream are ordered by the stream;
  // a frame enqueued on ANOTHER stream than the previous one first waits for that one (rare: a
  // caller alternating streams on one context -- use one context per stream to overlap frames).
  if (c->last_launch_stream && c->last_launch_stream != s) HIP_TRY(c, hipStreamSynchronize(c->last_launch_stream));
  c->last_launch_stream = s;
  const rt_camera &cam = c->cam;
  RenderParams p{};
  p.nodes = c->d_nodes;
  p.prims = c->d_prims;
  p.materials = c->d_materials;
  p.lights = c->d_lights;
  p.f6 = c->d_f6;
  p.n_lights = c->n_lights;
  p.n_grids = c->n_grids;
  p.spp = c->spp;
  p.has_root = c->has_root;
  p.boxes_safe = c->boxes_safe;
  p.box_absmax = c->box_absmax;
  p.clear_band = c->clear_band;
  p.slop_lds = 4.76837158203125e-07f * c->box_slop_scale;   // 2^-21
  p.slop_mem = 9.5367431640625e-07f * c->box_slop_scale;    // 2^-20
  p.debug_unsafe_mod = c->debug_unsafe_mod;
  p.debug_no_range = c->debug_no_range;
  p.W = cam.img_w;
  p.H = cam.img_h;
  p.max_depth = cam.max_depth;
  p.view_w = cam.view_w;
  p.view_h = cam.view_h;
  p.view_d = cam.view_d;
  p.lens = cam.lens;
  p.half_view_w = cam.view_w / 2.0f;
  p.half_view_h = cam.view_h / 2.0f;
  p.half_lens = cam.lens / 2;
  for (int i = 0; i < 3; ++i) {
    p.eye[i] = cam.eye[i];
    p.u[i] = cam.u[i];
    p.v[i] = cam.v[i];
    p.w[i] = cam.w[i];
    p.ambient[i] = cam.ambient[i];
    p.w_dir[i] = (-cam.view_d) * cam.w[i];  // multiply ww (-vDt) (RayTracer.hs:197-199): f * a.x
  }
  for (int i = 0; i < 3; ++i) {
    p.hot[0 + i] = p.eye[i];
    p.hot[3 + i] = p.u[i];
    p.hot[6 + i] = p.v[i];
    p.hot[9 + i] = p.w_dir[i];
    p.hot[20 + i] = p.ambient[i];
  }
  p.hot[12] = p.view_w;
  p.hot[13] = p.view_h;
  p.hot[14] = p.half_view_w;
  p.hot[15] = p.half_view_h;
  p.hot[16] = p.lens;
  p.hot[17] = p.half_lens;
  p.hot[18] = (float)p.W;
  p.hot[19] = (float)p.H;
  p.hot[23] = 0.0f;
  p.tile_w = c->tiling.tile_w;
  p.tile_h = c->tiling.tile_h;
  p.tiles_x = c->tiling.tiles_x;
  p.tile_samples = c->tiling.tile_w * c->tiling.tile_h * c->spp;
  p.rank = c->rank;
  p.nranks = c->nranks;
  p.n_local_tiles = c->tiling.n_local_tiles;
  p.compact = compact;
  p.rgb = d_out;
  p.hit_ids = d_ids;
  p.scratch = m.queues;
  p.scratch_stride = c->scratch_stride;
  p.acc = m.acc;
  p.px_per_packet = std::max(1, 64 / c->spp);
  p.chunks_per_packet = (c->spp + 63) / 64;
  p.groups_per_tile = (c->tiling.tile_w * c->tiling.tile_h + p.px_per_packet - 1) / p.px_per_packet;
  p.n_packets = c->tiling.n_local_tiles * p.groups_per_tile;
  p.fd_groups_per_tile = make_fastdiv((unsigned)p.groups_per_tile);
  p.fd_tiles_x = make_fastdiv((unsigned)p.tiles_x);
  p.fd_spp = make_fastdiv((unsigned)p.spp);
  p.fd_tile_w = make_fastdiv((unsigned)p.tile_w);
  p.fd_n_grids = make_fastdiv((unsigned)p.n_grids);
  p.tile_counter = m.tile_counter;
  p.stats = m.stats;
  // pruned traversal walks the wide records (its stack holds up to wide_stack siblings, and the
  // branch-free pushes may write one slot above the top); the reference-visit mode the binary tree
  // (a tree whose boxes do not nest -- a child sticking out of its parent -- is walked the
  //  reference's way, node by node: the pruned traversal's shortcuts rest on the nesting)
  const bool prune = !(flags & RT_FLAG_NO_PRUNE) && c->nested;
  const int stack_depth = prune ? std::max(4, (c->wide_stack + 1 + 3) & ~3) : c->stack_depth;
  p.stack_depth = stack_depth;
  p.wnodes = c->d_wnodes;
  p.n_wnode_recs = c->n_wnode_recs;
  p.trace_zero_weight = (flags & RT_FLAG_TRACE_ZERO_WEIGHT) ? 1 : 0;

  LaunchCfg cfg{};
  cfg.block = 64 * WAVES_PER_BLOCK;
  cfg.mlight = c->n_lights > 1;
  p.point_words = point_words_for(c->n_lights);
  cfg.lds_bytes = lds_bytes_for(stack_depth, p.point_words);
  cfg.ev_k0 = c->timing ? c->evk0 : nullptr;
  cfg.ev_k1 = c->timing ? c->evk1 : nullptr;
  cfg.prune = prune;
  cfg.count = (flags & RT_FLAG_COUNT_TESTS) != 0;
  // Small scenes (every shipped benchmark scene but bench3) are staged into each block's LDS:
  // node/primitive fetches then cost an LDS read instead of a trip through TA/L1.  The copy
  // must leave room for 4 blocks (16 waves) per CU: 160 KiB / 4 = 40 KiB per block.
  p.n_node_recs = c->n_node_recs;
  p.n_prim_recs = c->n_prim_recs;
  p.n_materials = c->n_materials;
  p.lds_prim_base = prune ? c->n_wnode_recs * LDS_WNODE_F4 : c->n_node_recs * LDS_REC_F4;
  p.lds_leafbox_base = p.lds_prim_base + c->n_prim_recs * LDS_REC_F4;
  p.lds_rec_f4 = p.lds_leafbox_base + (prune ? c->n_prim_recs * 2 : 0);  // leaf boxes: pruned traversal only
  p.leafbox = c->d_leafbox;
  cfg.lds_scene_bytes = (size_t)p.lds_rec_f4 * 16 + (size_t)c->n_materials * MAT_F * 4;
  cfg.lds_scene = c->lds_scene_mode != 0 && !cfg.count && cfg.lds_bytes + cfg.lds_scene_bytes <= 40 * 1024;
  int wpb = WAVES_PER_BLOCK;
  if (!cfg.lds_scene && c->lds_scene_mode != 0 && !cfg.count && cfg.prune &&
      lds_bytes_for(stack_depth, p.point_words, 8) + cfg.lds_scene_bytes <= 80 * 1024) {
    // too big beside four blocks' stacks, but two 8-wave blocks per CU can share one copy each
    wpb = 8;
    cfg.block = 512;
    cfg.lds_bytes = lds_bytes_for(stack_depth, p.point_words, 8);
    cfg.lds_scene = true;
  }
  if (!cfg.lds_scene) cfg.lds_scene_bytes = 0;
  cfg.lds_bytes += cfg.lds_scene_bytes + HOT_WORDS * 4;   // + the block's copy of RenderParams::hot
  // The 96-VGPR build (five waves per SIMD): small single-light LDS scenes, and only when a wave
  // has many packets to work through -- on a small share of a frame the extra waves only add
  // start-up and queue-drain steps (1/8 of c2: 0.71 -> 0.79 ms; whole frame, pipelined: 2.42 -> 2.36).
  cfg.five_waves = cfg.lds_scene && cfg.block == 64 * WAVES_PER_BLOCK && cfg.prune && !cfg.count && !cfg.mlight &&
                   cfg.lds_bytes <= 32 * 1024 &&  // five 4-wave blocks share a CU's 160 KB
                   (c->five_waves_mode > 0 ||
                    (c->five_waves_mode < 0 && (long long)p.n_packets >= 64ll * c->n_cus * 20));
  const bool fma = (flags & RT_FLAG_FMA) != 0;
  cfg.probe_lds_bytes = lds_bytes_for(stack_depth, p.point_words, WAVES_PER_BLOCK);
  int per_cu = fma ? occupancy_fma(cfg) : occupancy_strict(cfg);
  if (per_cu < 1) return fail(c, RT_ERR_DEVICE, "kernel does not fit on a CU (lds=%zu)", cfg.lds_bytes);
  per_cu = std::min(per_cu, 8);
  int want = (p.n_packets + wpb - 1) / wpb;
  cfg.grid = std::max(1, std::min(c->n_cus * per_cu / std::max(1, c->grid_div), want));
  cfg.grid = std::max(1, std::min(cfg.grid, m.max_waves / wpb));
  if (cfg.grid * wpb > m.queue_waves)
    return fail(c, RT_ERR_LIMIT, "grid exceeds the ray-queue memory");
  if (std::getenv("RT_DEBUG_LAUNCH"))
    std::fprintf(stderr, "[rt launch] grid %d x %d threads, %d blocks/CU, lds %zu B (scene part %zu), stack rows %d, "
                 "lds_scene=%d mlight=%d five=%d prune=%d count=%d packets %d\n", cfg.grid, cfg.block, per_cu, cfg.lds_bytes,
                 cfg.lds_scene_bytes, p.stack_depth, (int)cfg.lds_scene, (int)cfg.mlight,
                 (int)cfg.five_waves, (int)cfg.prune, (int)cfg.count, p.n_packets);

  // Refractive packets (a binary ray tree per sample) are handed out pixel by pixel when the
  // rank has few packets per wave: the frame then ends on quarter-size work items.
  p.split_refractive = c->split_mode >= 0 ? c->split_mode
                                           : (p.n_packets < 48 * cfg.grid * wpb ? 1 : 0);
  {
    // a wave should come back for tickets at least ~RT_BATCHES times per frame: its estimate of
    // what is left in the shard is one batch old, and an oversized last batch is the frame's tail
    int waves_per_shard = std::max(1, cfg.grid * wpb / PACKET_SHARDS);
    int per_shard = (p.n_packets + PACKET_SHARDS - 1) / PACKET_SHARDS;
    int batches = c->batches > 0 ? c->batches : 8;
    p.batch_cap = std::min(16, std::max(1, per_shard / (waves_per_shard * batches)));
  }
  p.refr_cap = c->refr_cap > 0 ? c->refr_cap : 1;  // measured: 1/1 never worse on the small scenes, c3 (mesh) 28.8 -> 21.7 ms
  p.refl_cap = c->refl_cap > 0 ? c->refl_cap : 1;
  p.rearm_min = c->rearm_min > 0 ? std::min(c->rearm_min, 64) : 12;
  p.probe_two = c->probe_two;
  p.probe_group = std::max(1, std::min(c->probe_group, 16));
  p.batch_div = std::max(1, 2 * cfg.grid * wpb / PACKET_SHARDS);  // 2 x waves per shard
  p.packets_per_shard = (p.n_packets + PACKET_SHARDS - 1) / PACKET_SHARDS;
  p.probe_span = std::max(256, ((p.n_packets + 1023) / 1024 + 255) / 256 * 256);  // <= 1024 blocks
  p.probe_blocks = std::max(1, (p.n_packets + p.probe_span - 1) / p.probe_span);
  p.pclass = m.pclass;
  p.block_hist = m.block_hist;
  p.class_totals = m.block_hist + 1024 * 4;
  p.order = m.order;
  // Ticket counters and statistics are cleared by order_kernel (which precedes render_kernel),
  // the pixel sums by the previous frame's resolve_kernel after it has read them; explicit
  // clears only where those have not run: no packets, or sums not known to be zero (new or
  // resized buffer, a frame that failed half way).
  if (p.n_packets == 0) {
    HIP_TRY(c, hipMemsetAsync(m.tile_counter, 0, PACKET_SHARDS * 32 * sizeof(unsigned), s));
    HIP_TRY(c, hipMemsetAsync(m.stats, 0, 16 * sizeof(unsigned long long), s));
    HIP_TRY(c, hipMemsetAsync(m.stats + STATS_DIAG_BASE, 0, STATS_DIAG_WORDS * sizeof(unsigned long long), s));
  }
  if (!m.acc_known_clean && m.acc_pixels)
    HIP_TRY(c, hipMemsetAsync(m.acc, 0, m.acc_pixels * 3 * sizeof(long long), s));
  c->last_stats = m.stats;
  if (STATS_TRACE_WORDS > 0)
    HIP_TRY(c, hipMemsetAsync(m.stats + STATS_TRACE_BASE, 0, (size_t)STATS_TRACE_WORDS * sizeof(unsigned long long), s));
  if (c->timing) HIP_TRY(c, hipEventRecord(c->ev0, s));
  (void)hipGetLastError();  // the check below is for THESE launches, not for a stale error of the process
  if (fma) launch_render_fma(p, cfg, s);
  else launch_render_strict(p, cfg, s);
  HIP_TRY(c, hipGetLastError());
  if (c->timing) HIP_TRY(c, hipEventRecord(c->ev1, s));
  c->timed = c->timing;
  return RT_OK;
}

// A frame on the context's own work memory (every entry point but the static one).
int launch_own(rt_ctx *c, uint32_t flags, float *d_out, int compact, uint32_t *d_ids, hipStream_t s) {
  int rc = prepare(c);
  if (rc) return rc;
  const WorkMem m = own_work_memory(c);
  c->acc_clean = false;  // until resolve_kernel is known to be enqueued
  rc = launch(c, flags, d_out, compact, d_ids, s, m);
  if (rc) return rc;
  c->acc_clean = true;   // resolve_kernel is enqueued: it leaves every sum it reads at zero
  return RT_OK;
}

int collect(rt_ctx *c, hipStream_t s, rt_stats *st) {
  HIP_TRY(c, hipStreamSynchronize(s));
  if (!st) return RT_OK;
  std::memset(st, 0, sizeof *st);
  unsigned long long h[12];
  HIP_TRY(c, hipMemcpy(h, c->last_stats ? c->last_stats : c->d_stats, sizeof h, hipMemcpyDeviceToHost));
  st->rays_primary = h[0];
  st->rays_shadow = h[1];
  st->rays_reflection = h[2];
  st->rays_refraction = h[3];
  st->box_tests = h[4];
  st->sphere_tests = h[5];
  st->triangle_tests = h[6];
  st->primary_hits = h[7];
  st->shaded_hits = h[8];
  st->wave_ticks_sum = h[9];
  st->wave_ticks_max = h[10];
  st->n_waves = h[11];
  if (c->timed) {
    float ms = 0;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    st->ms_launch = ms;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->evk0, c->evk1));
    st->ms_kernel = ms;
  }
  return RT_OK;
}

}  // namespace

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }

int rt_init(int device, rt_ctx **out) {
  if (!out) return RT_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return RT_ERR_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return RT_ERR_DEVICE;
  rt_ctx *c = new rt_ctx();
  c->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
    delete c;
    return RT_ERR_DEVICE;
  }
  c->n_cus = prop.multiProcessorCount;
  // device address of the library's static work memory on THIS GPU (rt_render_static_async)
  if (hipError_t e = lookup_static_scratch(device); e != hipSuccess) {
    std::fprintf(stderr, "rt_init: hipGetSymbolAddress(scratch) failed on device %d: %s\n", device, hipGetErrorString(e));
    delete c;
    return RT_ERR_DEVICE;
  }
  if (const char *e = std::getenv("RT_SPLIT_REFRACTIVE")) c->split_mode = std::atoi(e);
  if (const char *e = std::getenv("RT_LDS_SCENE")) c->lds_scene_mode = std::atoi(e);
  if (const char *e = std::getenv("RT_GRID_DIV")) c->grid_div = std::max(1, std::atoi(e));
  if (const char *e = std::getenv("RT_BATCHES")) c->batches = std::atoi(e);
  if (const char *e = std::getenv("RT_REFR_CAP")) c->refr_cap = std::atoi(e);
  if (const char *e = std::getenv("RT_REFL_CAP")) c->refl_cap = std::atoi(e);
  if (const char *e = std::getenv("RT_REARM_MIN")) c->rearm_min = std::atoi(e);
  if (const char *e = std::getenv("RT_PROBE_TWO")) c->probe_two = std::atoi(e) != 0;
  if (const char *e = std::getenv("RT_PROBE_GROUP")) c->probe_group = std::atoi(e);
  if (const char *e = std::getenv("RT_FIVE_WAVES")) c->five_waves_mode = std::atoi(e);
  if (const char *e = std::getenv("RT_DEBUG_CLEAR_BAND")) c->clear_band = (float)std::atof(e);
  if (const char *e = std::getenv("RT_DEBUG_BOX_SLOP")) c->box_slop_scale = std::max(1.0f, (float)std::atof(e));
  if (const char *e = std::getenv("RT_DEBUG_UNSAFE_MOD")) c->debug_unsafe_mod = std::max(0, std::atoi(e));
  if (const char *e = std::getenv("RT_DEBUG_NO_RANGE_PRUNE")) c->debug_no_range = std::atoi(e) != 0;
  if (const char *e = std::getenv("RT_DEBUG_TRUST_TREE")) c->trust_tree = std::atoi(e) != 0;
  bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess &&
            hipEventCreate(&c->evk0) == hipSuccess && hipEventCreate(&c->evk1) == hipSuccess &&
            hipEventCreate(&c->evt0) == hipSuccess && hipEventCreate(&c->evt1) == hipSuccess &&
            hipMalloc((void **)&c->d_tile_counter, PACKET_SHARDS * 32 * sizeof(unsigned)) == hipSuccess &&
            hipMalloc((void **)&c->d_stats, STATS_WORDS * sizeof(unsigned long long)) == hipSuccess;
  if (!ok) {
    rt_shutdown(c);
    return RT_ERR_DEVICE;
  }
  *out = c;
  return RT_OK;
}

void rt_shutdown(rt_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  void *bufs[] = {c->d_nodes, c->d_prims, c->d_wnodes, c->d_leafbox, c->d_materials12, c->d_prim_src, c->d_wide_src, c->d_level_nodes, c->d_prim_box,
                  c->d_spheres, c->d_triangles, c->d_materials, c->d_lights, c->d_f6, c->d_scratch,
                  c->d_tile_counter, c->d_stats, c->d_rgb, c->d_ids, c->d_acc,
                  c->d_pclass, c->d_block_hist, c->d_order};
  for (void *b : bufs)
    if (b) (void)hipFree(b);
  void *build_bufs[] = {c->build.box_in, c->build.leaf_box6, c->build.node_box, c->build.keys, c->build.child, c->build.ints,
                        c->build.queue, c->build.scalars_mem, c->build.sort_tmp, c->build.export_nodes};
  for (void *b : build_bufs)
    if (b) (void)hipFree(b);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->evk0) (void)hipEventDestroy(c->evk0);
  if (c->evk1) (void)hipEventDestroy(c->evk1);
  if (c->evt0) (void)hipEventDestroy(c->evt0);
  if (c->evt1) (void)hipEventDestroy(c->evt1);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char *rt_last_error(rt_ctx *c) { return c ? c->err.c_str() : "null context"; }

int rt_upload_scene(rt_ctx *c, const rt_scene *sc) {
  if (!c || !sc) return RT_ERR_INVALID;
  if (sc->n_nodes < 0 || sc->n_refs < 0 || sc->n_spheres < 0 || sc->n_triangles < 0 ||
      sc->n_materials < 0 || sc->n_lights < 0)
    return fail(c, RT_ERR_INVALID, "negative count in rt_scene");
  if ((sc->n_nodes && !sc->nodes) || (sc->n_refs && !sc->leaf_refs) || (sc->n_spheres && !sc->spheres) ||
      (sc->n_triangles && !sc->triangles) || (sc->n_materials && !sc->materials) ||
      (sc->n_lights && !sc->lights))
    return fail(c, RT_ERR_INVALID, "null array with non-zero count in rt_scene");
  HIP_TRY(c, hipSetDevice(c->device));
  c->have_scene = false;
  c->prepared = false;
  c->device_built = false;
  Repack rp;
  rp.sc = sc;
  rp.seen.assign((size_t)sc->n_nodes, 0);
  // pseudo root (rt_device.h): left = real root, right = none
  rp.nodes.assign(16, 0.0f);
  int root_ref = REF_NONE;
  if (sc->n_nodes > 0) {
    root_ref = rp.visit(0, 1);
    if (!rp.err.empty()) return fail(c, RT_ERR_INVALID, "invalid BVH: %s", rp.err.c_str());
    std::memcpy(&rp.nodes[0], &sc->nodes[0].xmin, 24);
  }
  Repack::put_i(&rp.nodes[12], root_ref);
  Repack::put_i(&rp.nodes[13], REF_NONE);
  if (rp.max_depth > 64)
    return fail(c, RT_ERR_LIMIT, "BVH depth %d exceeds the traversal stack limit of 64", rp.max_depth);
  int rc;
  UploadBatch up;
  up.add(&c->d_nodes, &c->cap_nodes, rp.nodes.data(), rp.nodes.size() * 4);
  up.add(&c->d_prims, &c->cap_prims, rp.prims.data(), rp.prims.size() * 4);
  WideBuild wb{rp.nodes};
  // The deepest traversal stack with which four 4-wave blocks still share a CU's 160 KB of LDS
  // beside the block's constants and, with several lights, the shadow-ray pool (c3: 31 entries):
  // the collapse widens nodes up to that need and no further (a wider tree is a faster tree, c3
  // 5.15 ms at a cap of 22, 4.40 at 27; a third block per CU less costs more than it gains).
  {
    const int per_wave = ((160 * 1024 / 4) - HOT_WORDS * 4) / WAVES_PER_BLOCK - (point_words_for(sc->n_lights) + 32) * 4;
    const int depth_max = std::min(40, (per_wave / 256) & ~3);   // stack_depth is a multiple of 4 and holds one scratch slot
    wb.cap = std::max(std::max(3, depth_max - 1), rp.max_depth);
  }
  wb.height.assign(rp.nodes.size() / 16, -1);
  if (sc->n_nodes > 0) {
    wb.build(0, 0);   // binary record 0 is the pseudo root: its only child is the real root
    wb.breadth_first_top(WIDE_BFS_TOP);
  } else {
    wb.wide.assign(32, 0.0f);
    for (int i = 0; i < 24; ++i) wb.wide[i] = ((i >> 2) & 1) ? -WideBuild::ABSENT_PLANE : WideBuild::ABSENT_PLANE;
    std::fill(wb.wide.begin() + 24, wb.wide.begin() + 28, Repack::as_f(REF_NONE));
  }
  up.add(&c->d_wnodes, &c->cap_wnodes, wb.wide.data(), wb.wide.size() * 4);
  rp.leafbox.resize(rp.prims.size() / 2, 0.0f);
  up.add(&c->d_leafbox, &c->cap_leafbox, rp.leafbox.data(), rp.leafbox.size() * 4);
  c->n_wnode_recs = (int)(wb.wide.size() / 32);
  c->wide_stack = wb.max_stack;
  // refit tables (host copies; uploaded by the first rt_refit_async)
  c->h_prim_src.swap(rp.prim_src);
  c->h_wide_src.swap(wb.src);
  {
    const int nb = (int)(rp.nodes.size() / 16);
    std::vector<int> depth(nb, 0);
    int deepest = 0;
    for (int i = 0; i < nb; ++i)   // records are in pre-order: a parent comes before its children
      for (int k = 0; k < 2; ++k) {
        int ref = WideBuild::get_i(&rp.nodes[(size_t)i * 16 + 12 + k]);
        if (ref >= 0 && ref != i) { depth[ref] = depth[i] + 1; deepest = std::max(deepest, depth[ref]); }
      }
    c->h_level_nodes.clear();
    c->h_level_start.assign(1, 0);
    for (int d = deepest; d >= 0; --d) {   // deepest level first
      for (int i = 0; i < nb; ++i)
        if (depth[i] == d) c->h_level_nodes.push_back(i);
      c->h_level_start.push_back((int)c->h_level_nodes.size());
    }
  }
  c->n_spheres = sc->n_spheres;
  c->n_triangles = sc->n_triangles;
  c->refit_ready = false;
  if (std::getenv("RT_DEBUG_UPLOAD"))
    std::fprintf(stderr, "[rt_upload_scene] binary records %zu (depth %d), wide records %d, wide stack need %d (cap %d), primitives %zu\n",
                 rp.nodes.size() / 16, rp.max_depth, c->n_wnode_recs, wb.max_stack, wb.cap, rp.prims.size() / 16);
  up.add(&c->d_materials12, &c->cap_materials12, sc->materials, (size_t)sc->n_materials * sizeof(rt_material));
  up.add(&c->d_materials, &c->cap_materials, nullptr, (size_t)std::max(1, sc->n_materials) * MAT_F * sizeof(float));
  up.add(&c->d_lights, &c->cap_lights, sc->lights, (size_t)sc->n_lights * sizeof(rt_light));
  if ((rc = up.run(c))) return rc;
  (void)hipGetLastError();
  launch_material_prep(c->d_materials12, c->d_materials, sc->n_materials, c->stream);
  HIP_TRY(c, hipGetLastError());
  c->n_lights = sc->n_lights;
  c->n_materials = sc->n_materials;
  c->n_node_recs = (int)(rp.nodes.size() / 16);
  c->n_prim_recs = (int)(rp.prims.size() / 16);
  c->has_root = sc->n_nodes > 0;
  c->tree_depth = rp.max_depth;
  c->boxes_safe = 1;
  c->box_absmax = 0.0f;
  for (int i = 0; i < sc->n_nodes && c->boxes_safe; ++i) {
    if (!rp.seen[i]) continue;
    const float *b = &sc->nodes[i].xmin;
    for (int k = 0; k < 6; ++k) {
      float a = std::fabs(b[k]);
      if (!(b[k] == 0.0f || (a >= 1.4551915228366852e-11f && a <= 5.764607523034235e17f))) c->boxes_safe = 0;
      if (a > c->box_absmax) c->box_absmax = a;
    }
    if (!(b[0] <= b[1] && b[2] <= b[3] && b[4] <= b[5])) c->boxes_safe = 0;  // min <= max (entry/exit form)
  }
  // Do the boxes nest?  (sahBVH's do: nodeBox = surround left right, BoundingVolumeHierarchy.hs:50.)
  c->nested = true;
  for (int i = 0; i < sc->n_nodes && c->nested; ++i) {
    const rt_bvh_node &n = sc->nodes[i];
    // (only nodes the tree reaches: Repack::visit has range-checked their children; an unreachable
    //  node may hold anything)
    if (!rp.seen[i] || n.left < 0) continue;
    const float *pb = &n.xmin;
    for (int k = 0; k < 2; ++k) {
      const float *cb = &sc->nodes[k ? n.right : n.left].xmin;
      for (int a = 0; a < 3; ++a)
        if (!(cb[2 * a] >= pb[2 * a] && cb[2 * a + 1] <= pb[2 * a + 1])) c->nested = false;
    }
  }
  // Does every leaf box bound its objects?  (sahBVH's do: a leaf's box is the fold of its objects'
  // toBox, BoundingVolumeHierarchy.hs:50-51,178-189.)  hitsBox is a LINE test: with a leaf box that cuts its
  // object short the reference tests the object for every ray whose line meets the box, wherever the
  // hit lies -- range pruning (entry beyond the closest hit, exit behind the origin) would lose such
  // hits.  Such a tree is walked the reference's way as well.
  for (int i = 0; i < sc->n_nodes && c->nested && !c->trust_tree; ++i) {
    const rt_bvh_node &n = sc->nodes[i];
    if (!rp.seen[i] || n.left >= 0) continue;
    const float *lb = &n.xmin;
    const long first = -1L - (long)n.left;
    for (int k = 0; k < n.right && c->nested; ++k) {
      const uint32_t r = sc->leaf_refs[first + k];
      for (int a = 0; a < 3 && c->nested; ++a) {
        float lo, hi;
        if (r & RT_REF_TRIANGLE) {
          const rt_triangle &t = sc->triangles[RT_REF_INDEX(r)];
          lo = std::min(std::min(t.a[a], t.b[a]), t.c[a]);
          hi = std::max(std::max(t.a[a], t.b[a]), t.c[a]);
        } else {
          const rt_sphere &s = sc->spheres[r];
          lo = s.center[a] - s.radius;
          hi = s.center[a] + s.radius;
        }
        if (!(lb[2 * a] <= lo && hi <= lb[2 * a + 1])) c->nested = false;
      }
    }
  }
  c->have_scene = true;
  return RT_OK;
}

int rt_set_camera(rt_ctx *c, const rt_camera *cam) {
  if (!c || !cam) return RT_ERR_INVALID;
  if (cam->img_w < 1 || cam->img_h < 1 || cam->img_w > 65536 || cam->img_h > 65536)
    return fail(c, RT_ERR_INVALID, "image size %dx%d out of range", cam->img_w, cam->img_h);
  if ((uint64_t)cam->img_w * (uint64_t)cam->img_h >= (1ull << 32))
    return fail(c, RT_ERR_LIMIT, "image of %dx%d pixels exceeds 2^32", cam->img_w, cam->img_h);
  if (cam->aa < 1 || cam->aa > 32) return fail(c, RT_ERR_INVALID, "aa=%d out of range [1,32]", cam->aa);
  if (cam->max_depth < 0 || cam->max_depth > MAX_DEPTH)
    return fail(c, RT_ERR_LIMIT, "max_depth=%d out of range [0,%d]", cam->max_depth, MAX_DEPTH);
  c->cam = *cam;
  c->have_cam = true;
  c->prepared = false;
  return RT_OK;
}

int rt_set_samples(rt_ctx *c, const float *f6, int32_t n_grids, int32_t spp) {
  if (!c || !f6) return RT_ERR_INVALID;
  if (n_grids < 1 || spp < 1 || spp > 1024) return fail(c, RT_ERR_INVALID, "bad sample table shape");
  HIP_TRY(c, hipSetDevice(c->device));
  UploadBatch up;
  up.add(&c->d_f6, &c->cap_f6, f6, (size_t)n_grids * spp * 6 * sizeof(float));
  int rc = up.run(c);
  if (rc) return rc;
  c->n_grids = n_grids;
  c->spp = spp;
  c->have_samples = true;
  c->prepared = false;
  return RT_OK;
}

int rt_refit_async(rt_ctx *c, const rt_sphere *spheres, int32_t n_spheres, const rt_triangle *triangles,
                   int32_t n_triangles, void *stream) {
  if (!c) return RT_ERR_INVALID;
  if (!c->have_scene) return fail(c, RT_ERR_STATE, "rt_upload_scene has not been called");
  if (c->device_built)
    return fail(c, RT_ERR_STATE, "rt_refit_async: the tree was built by rt_build_scene, which keeps no refit tables: build again");
  if (n_spheres != c->n_spheres || n_triangles != c->n_triangles)
    return fail(c, RT_ERR_INVALID, "rt_refit_async: %d spheres / %d triangles, the uploaded scene has %d / %d", n_spheres,
                n_triangles, c->n_spheres, c->n_triangles);
  if ((n_spheres && !spheres) || (n_triangles && !triangles)) return fail(c, RT_ERR_INVALID, "null object array");
  HIP_TRY(c, hipSetDevice(c->device));
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  if (s != c->stream) return fail(c, RT_ERR_INVALID, "rt_refit_async runs on the context's own stream (pass NULL)");
  const int n_prims = c->n_prim_recs, n_bin = c->n_node_recs;
  UploadBatch up;
  if (!c->refit_ready) {
    up.add(&c->d_prim_src, &c->cap_prim_src, c->h_prim_src.data(), c->h_prim_src.size() * sizeof(uint32_t));
    up.add(&c->d_wide_src, &c->cap_wide_src, c->h_wide_src.data(), c->h_wide_src.size() * sizeof(int));
    up.add(&c->d_level_nodes, &c->cap_level_nodes, c->h_level_nodes.data(), c->h_level_nodes.size() * sizeof(int));
    up.add(&c->d_prim_box, &c->cap_prim_box, nullptr, (size_t)std::max(1, n_prims) * 6 * sizeof(float));
  }
  up.add(&c->d_spheres, &c->cap_spheres, spheres, (size_t)n_spheres * sizeof(rt_sphere));
  up.add(&c->d_triangles, &c->cap_triangles, triangles, (size_t)n_triangles * sizeof(rt_triangle));
  int rc = up.run(c);   // ordered on the context's stream, behind its frame in flight
  if (rc) return rc;
  c->refit_ready = true;
  (void)hipGetLastError();
  if (n_prims > 0) {
    hipLaunchKernelGGL(refit_prims_kernel, dim3((n_prims + 255) / 256), dim3(256), 0, s, reinterpret_cast<float *>(c->d_prims),
                       c->d_prim_src, c->d_spheres, c->d_triangles, c->d_prim_box, n_prims);
    for (size_t lv = 0; lv + 1 < c->h_level_start.size(); ++lv) {
      const int lo = c->h_level_start[lv], hi = c->h_level_start[lv + 1];
      if (hi > lo)
        hipLaunchKernelGGL(refit_nodes_kernel, dim3((2 * (hi - lo) + 255) / 256), dim3(256), 0, s,
                           reinterpret_cast<float *>(c->d_nodes), reinterpret_cast<const float *>(c->d_prims),
                           c->d_prim_box, reinterpret_cast<float *>(c->d_leafbox), c->d_level_nodes, lo, hi);
    }
    const int n_slots = c->n_wnode_recs * 4;
    hipLaunchKernelGGL(refit_wide_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, s, reinterpret_cast<float *>(c->d_wnodes),
                       reinterpret_cast<const float *>(c->d_nodes), c->d_wide_src, n_slots);
    HIP_TRY(c, hipGetLastError());
  }
  (void)n_bin;
  // the exact-reciprocal box path needs every box coordinate 0 or in [2^-36, 2^59] and min <= max:
  // box planes are object coordinates (triangles) or centre -+ radius (spheres)
  c->nested = true;   // the refitted boxes are unions of their children's
  c->boxes_safe = 1;
  c->box_absmax = 0.0f;
  auto ok = [c](float v) {
    if (std::fabs(v) > c->box_absmax) c->box_absmax = std::fabs(v);
    float a = std::fabs(v);
    return v == 0.0f || (a >= 1.4551915228366852e-11f && a <= 5.764607523034235e17f);
  };
  for (int i = 0; i < n_triangles && c->boxes_safe; ++i)
    for (int a = 0; a < 3; ++a)
      if (!ok(triangles[i].a[a]) || !ok(triangles[i].b[a]) || !ok(triangles[i].c[a])) c->boxes_safe = 0;
  for (int i = 0; i < n_spheres && c->boxes_safe; ++i)
    for (int a = 0; a < 3; ++a) {
      const float lo = spheres[i].center[a] - spheres[i].radius, hi = spheres[i].center[a] + spheres[i].radius;
      if (!ok(lo) || !ok(hi) || !(lo <= hi)) c->boxes_safe = 0;
    }
  return RT_OK;
}

int rt_set_partition(rt_ctx *c, int32_t rank, int32_t nranks) {
  if (!c) return RT_ERR_INVALID;
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(c, RT_ERR_INVALID, "bad partition %d/%d", rank, nranks);
  c->rank = rank;
  c->nranks = nranks;
  c->prepared = false;
  return RT_OK;
}

int rt_get_tiling(rt_ctx *c, rt_tiling *out) {
  if (!c || !out) return RT_ERR_INVALID;
  if (!c->have_cam || !c->have_samples) return fail(c, RT_ERR_STATE, "camera and samples must be set first");
  compute_tiling(c);
  *out = c->tiling;
  return RT_OK;
}

int rt_render_async(rt_ctx *c, uint32_t flags, float *d_rgb, uint32_t *d_ids, void *stream) {
  if (!c || !d_rgb) return RT_ERR_INVALID;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  return launch_own(c, flags, d_rgb, 0, d_ids, s);
}

int rt_render_tiles_async(rt_ctx *c, uint32_t flags, float *d_tiles, void *stream) {
  if (!c || !d_tiles) return RT_ERR_INVALID;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  return launch_own(c, flags, d_tiles, 1, nullptr, s);
}

int rt_render_static_async(rt_ctx *c, uint32_t flags, float *d_rgb, uint32_t *d_ids, void *stream) {
  if (!c || !d_rgb) return RT_ERR_INVALID;
  int rc = validate(c);  // state checks and launch geometry only: no allocation
  if (rc) return rc;
  if (c->rank != 0 || c->nranks != 1)
    return fail(c, RT_ERR_STATE, "rt_render_static_async renders whole frames: reset the partition to (0, 1)");
  unsigned char *base = (c->device >= 0 && c->device < ST_MAX_DEVICES) ? g_static_ptr[c->device] : nullptr;
  if (!base) return fail(c, RT_ERR_DEVICE, "no static work memory on device %d", c->device);
  if (local_pixels(c) > ST_MAX_PIXELS)
    return fail(c, RT_ERR_LIMIT, "rt_render_static_async: %zu pixels (image %dx%d rounded up to whole %dx%d tiles) exceed "
                "the static work memory's limit of %zu", local_pixels(c), c->cam.img_w, c->cam.img_h, c->tiling.tile_w,
                c->tiling.tile_h, ST_MAX_PIXELS);
  if (c->cam.max_depth > ST_MAX_DEPTH)
    return fail(c, RT_ERR_LIMIT, "rt_render_static_async: max_depth %d exceeds the static work memory's limit of %d",
                c->cam.max_depth, ST_MAX_DEPTH);
  if (order_entries(c) > ST_ORDER_ENTRIES)
    return fail(c, RT_ERR_LIMIT, "rt_render_static_async: %zu packet-order entries exceed the limit of %zu",
                order_entries(c), ST_ORDER_ENTRIES);
  WorkMem m;
  m.stats = reinterpret_cast<unsigned long long *>(base + ST_OFF_STATS);
  m.tile_counter = reinterpret_cast<unsigned *>(base + ST_OFF_COUNTER);
  m.block_hist = reinterpret_cast<unsigned *>(base + ST_OFF_HIST);
  m.pclass = base + ST_OFF_PCLASS;
  m.order = reinterpret_cast<int *>(base + ST_OFF_ORDER);
  m.order_cap = ST_ORDER_ENTRIES;
  m.acc = reinterpret_cast<long long *>(base + ST_OFF_ACC);
  m.acc_pixels = local_pixels(c);   // the array keeps the last call's contents (any context's): always cleared
  m.acc_known_clean = false;
  m.queues = reinterpret_cast<float *>(base + ST_OFF_QUEUES);
  m.queue_waves = ST_MAX_WAVES;
  m.max_waves = ST_MAX_WAVES;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  return launch(c, flags, d_rgb, 0, d_ids, s, m);
}

}  // extern "C"

template <class T>
static int assemble_any(rt_ctx *c, const T *d_gathered, int32_t nranks, size_t stride, T *d_rgb, void *stream) {
  if (!c || !d_gathered || !d_rgb || nranks < 1) return RT_ERR_INVALID;
  if (!c->have_cam || !c->have_samples) return fail(c, RT_ERR_STATE, "camera and samples must be set first");
  HIP_TRY(c, hipSetDevice(c->device));
  compute_tiling(c);
  const rt_tiling &t = c->tiling;
  size_t tile_elems = (size_t)t.tile_w * t.tile_h * 3;
  size_t max_local = (size_t)(t.n_tiles + nranks - 1) / nranks;
  if (stride < max_local * tile_elems)
    return fail(c, RT_ERR_INVALID, "stride %zu smaller than a rank's tile buffer %zu", stride, max_local * tile_elems);
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  size_t total = (size_t)t.n_tiles * t.tile_w * t.tile_h;
  int grid = (int)std::min<size_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(assemble_kernel<T>, dim3(grid), dim3(256), 0, s, d_gathered, nranks, stride, d_rgb,
                     c->cam.img_w, c->cam.img_h, t.tile_w, t.tile_h, t.tiles_x, t.n_tiles);
  HIP_TRY(c, hipGetLastError());
  return RT_OK;
}

extern "C" {

int rt_assemble_async(rt_ctx *c, const float *d_gathered, int32_t nranks, size_t stride_floats, float *d_rgb,
                      void *stream) {
  return assemble_any<float>(c, d_gathered, nranks, stride_floats, d_rgb, stream);
}

int rt_assemble8_async(rt_ctx *c, const uint8_t *d_gathered8, int32_t nranks, size_t stride_bytes, uint8_t *d_rgb8,
                       void *stream) {
  return assemble_any<uint8_t>(c, d_gathered8, nranks, stride_bytes, d_rgb8, stream);
}

int rt_quantize_async(rt_ctx *c, const float *d_rgb, uint8_t *d_rgb8, size_t n, void *stream) {
  if (!c || !d_rgb || !d_rgb8) return RT_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  if (n == 0) return RT_OK;
  int grid = (int)std::min<size_t>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(quantize_kernel, dim3(grid), dim3(256), 0, s, d_rgb, d_rgb8, n);
  HIP_TRY(c, hipGetLastError());
  return RT_OK;
}

int rt_set_timing(rt_ctx *c, int32_t on) {
  if (!c) return RT_ERR_INVALID;
  c->timing = on != 0;
  return RT_OK;
}

int rt_set_concurrency(rt_ctx *c, int32_t n) {
  if (!c) return RT_ERR_INVALID;
  if (n < 1 || n > 16) return fail(c, RT_ERR_INVALID, "n_concurrent=%d out of range [1,16]", n);
  c->grid_div = n;
  return RT_OK;
}

int rt_get_stream(rt_ctx *c, void **out) {
  if (!c || !out) return RT_ERR_INVALID;
  *out = (void *)c->stream;
  return RT_OK;
}

/* diagnostics: per-wave (lifetime in 100 MHz ticks, rays) pairs of the last render */
int rt_wave_diag(rt_ctx *c, uint64_t *out, int32_t max_waves) {
  if (!c || !out || max_waves < 1) return RT_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipDeviceSynchronize());
  int n = std::min<int>(max_waves, STATS_MAX_WAVES + (STATS_DIAG_WORDS + STATS_TRACE_WORDS) / 4);  // the last rows: sums / step traces of diagnostic builds
  HIP_TRY(c, hipMemcpy(out, (c->last_stats ? c->last_stats : c->d_stats) + 16, (size_t)n * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return RT_OK;
}

int rt_collect(rt_ctx *c, void *stream, rt_stats *st) {
  if (!c) return RT_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  return collect(c, stream ? (hipStream_t)stream : c->stream, st);
}

int rt_render(rt_ctx *c, uint32_t flags, float *rgb_out, uint32_t *ids_out, rt_stats *st) {
  if (!c || !rgb_out) return RT_ERR_INVALID;
  int rc = prepare(c);
  if (rc) return rc;
  HIP_TRY(c, hipSetDevice(c->device));
  // (the two events of the whole-call bracket live in the context: nothing to leak on the
  //  error returns below)
  hipEvent_t t0 = c->evt0, t1 = c->evt1;
  size_t npx = (size_t)c->cam.img_w * c->cam.img_h;
  if (c->d_rgb_floats < npx * 3) {
    if (c->d_rgb) (void)hipFree(c->d_rgb);
    c->d_rgb = nullptr;
    c->d_rgb_floats = 0;
    HIP_TRY(c, hipMalloc((void **)&c->d_rgb, npx * 3 * sizeof(float)));
    c->d_rgb_floats = npx * 3;
  }
  if (ids_out && c->d_ids_count < npx * c->spp) {
    if (c->d_ids) (void)hipFree(c->d_ids);
    c->d_ids = nullptr;
    c->d_ids_count = 0;
    HIP_TRY(c, hipMalloc((void **)&c->d_ids, npx * c->spp * sizeof(uint32_t)));
    c->d_ids_count = npx * c->spp;
  }
  HIP_TRY(c, hipEventRecord(t0, c->stream));
  HIP_TRY(c, hipMemsetAsync(c->d_rgb, 0, npx * 3 * sizeof(float), c->stream));
  if (ids_out) HIP_TRY(c, hipMemsetAsync(c->d_ids, 0xff, npx * c->spp * sizeof(uint32_t), c->stream));
  rc = launch_own(c, flags, c->d_rgb, 0, ids_out ? c->d_ids : nullptr, c->stream);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(rgb_out, c->d_rgb, npx * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (ids_out)
    HIP_TRY(c, hipMemcpyAsync(ids_out, c->d_ids, npx * c->spp * sizeof(uint32_t), hipMemcpyDeviceToHost,
                              c->stream));
  HIP_TRY(c, hipEventRecord(t1, c->stream));
  rc = collect(c, c->stream, st);
  if (rc) return rc;
  if (st) {
    float ms = 0;
    HIP_TRY(c, hipEventElapsedTime(&ms, t0, t1));
    st->ms_total = ms;
  }
  return RT_OK;
}

}  // extern "C"

#include "rt_build.inl"
