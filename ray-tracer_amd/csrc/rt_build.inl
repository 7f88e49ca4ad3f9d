// rt_build.inl -- the tree built ON THE GPU (rt_build_scene), for scenes whose objects move so much
// from frame to frame that a refit of the old tree will not do and the host's sahBVH
// (BoundingVolumeHierarchy.hs:34-94: O(n log^2 n), 0.2 s for 100 000 triangles) would dwarf the frame.
// Included by rt_api.hip (it needs rt_ctx and the upload helpers).
//
// This is a TREE-CHANGING mode: the reference builds its tree with sahBVH, this file builds a
// linear BVH (Morton order + Karras' parallel radix tree, HPG 2012), one object per leaf.  What the
// reference's hitBVH returns depends on the tree only where hits tie (equal t: the right subtree
// wins, RayTracer.hs:383-387,410) and where its un-pruned line tests visit a leaf at all; both are
// functions of the tree handed to it.  The parity statement is therefore: the GPU frame equals the
// reference's algorithm (the oracle) walking THE SAME device-built tree, which rt_download_tree
// hands back in the input format of rt_upload_scene (tests/test_device_build.py); tie winners may
// differ from a frame rendered with a host sahBVH tree.
//
// Steps, all on the context's stream: object boxes (toBox, BoundingVolumeHierarchy.hs:178-189) and the
// scene's bounds -> 30-bit Morton codes of the box centres, made unique by the object's index ->
// radix sort (rocPRIM) -> leaves in sorted order = the depth-first leaf order the device records
// use -> Karras' internal nodes (one thread each) -> boxes and heights bottom-up (the second
// thread to arrive at a node does the work) -> binary records (rt_device.h) -> the 4-wide records
// level by level with the same widening rule and stack cap as the host collapse (WideBuild).

#include <rocprim/device/device_radix_sort.hpp>

namespace {

struct BuildScalars {       // device-side results the host reads back
  unsigned lo[3], hi[3];    // scene bounds, order-preserving encoding of floats
  unsigned absmax_bits;     // largest |coordinate| of any box
  unsigned unsafe;          // some coordinate is not 0 and outside [2^-36, 2^59] (or not finite)
  int root_height;          // WideBuild::height_of(pseudo root)
  int n_wide, max_stack;
  int queue_n[2];           // entries in the two level queues
  int pad;
};

__device__ inline unsigned enc_f(float f) {   // monotone float -> unsigned
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float dec_f(unsigned e) {
  const unsigned u = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// One thread per object (spheres first, then triangles): toBox, and the scene's bounds / checks.
__global__ void build_boxes_kernel(const rt_sphere *__restrict__ sph, int n_s, const rt_triangle *__restrict__ tri, int n_t,
                                   float *__restrict__ box_in, BuildScalars *__restrict__ sc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = n_s + n_t;
  float b[6] = {INFINITY, -INFINITY, INFINITY, -INFINITY, INFINITY, -INFINITY};
  if (i < n) {
    if (i < n_s) {
      const rt_sphere p = sph[i];
      for (int a = 0; a < 3; ++a) {
        b[2 * a] = p.center[a] - p.radius;
        b[2 * a + 1] = p.center[a] + p.radius;
      }
    } else {
      const rt_triangle t = tri[i - n_s];
      for (int a = 0; a < 3; ++a) {
        b[2 * a] = hmin_d(hmin_d(t.a[a], t.b[a]), t.c[a]);
        b[2 * a + 1] = hmax_d(hmax_d(t.a[a], t.b[a]), t.c[a]);
      }
    }
    for (int a = 0; a < 6; ++a) box_in[(size_t)i * 6 + a] = b[a];
  }
  unsigned lo[3], hi[3], amax = 0, bad = 0;
  for (int a = 0; a < 3; ++a) {
    lo[a] = enc_f(b[2 * a]);
    hi[a] = enc_f(b[2 * a + 1]);
    if (i < n) {
      for (int k = 0; k < 2; ++k) {
        const float v = b[2 * a + k], av = fabsf(v);
        amax = max(amax, __float_as_uint(av));   // (non-negative floats order like their bits; NaN sorts above inf)
        if (!(v == 0.0f || (av >= 1.4551915228366852e-11f && av <= 5.764607523034235e17f))) bad = 1;
      }
      if (!(b[2 * a] <= b[2 * a + 1])) bad = 1;
    }
  }
  for (int d = 32; d >= 1; d >>= 1) {
    for (int a = 0; a < 3; ++a) {
      lo[a] = min(lo[a], (unsigned)__shfl_xor((int)lo[a], d));
      hi[a] = max(hi[a], (unsigned)__shfl_xor((int)hi[a], d));
    }
    amax = max(amax, (unsigned)__shfl_xor((int)amax, d));
    bad |= (unsigned)__shfl_xor((int)bad, d);
  }
  if ((threadIdx.x & 63) == 0) {
    for (int a = 0; a < 3; ++a) {
      atomicMin(&sc->lo[a], lo[a]);
      atomicMax(&sc->hi[a], hi[a]);
    }
    atomicMax(&sc->absmax_bits, amax);
    if (bad) atomicOr(&sc->unsafe, 1u);
  }
}

__device__ inline unsigned spread10(unsigned v) {   // 10 bits -> every third bit
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}
// key = (Morton code of the box centre, 30 bits) << 32 | object index: unique, so that the radix tree
// is well defined for coincident centres as well
__global__ void build_keys_kernel(const float *__restrict__ box_in, int n, const BuildScalars *__restrict__ sc,
                                  unsigned long long *__restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned q[3];
  for (int a = 0; a < 3; ++a) {
    const float lo = dec_f(sc->lo[a]), hi = dec_f(sc->hi[a]);
    const float c = 0.5f * box_in[(size_t)i * 6 + 2 * a] + 0.5f * box_in[(size_t)i * 6 + 2 * a + 1];
    const float ext = hi - lo;
    float u = ext > 0 ? (c - lo) / ext : 0.0f;
    u = u == u ? fminf(fmaxf(u * 1024.0f, 0.0f), 1023.0f) : 0.0f;   // (not finite: anywhere will do)
    q[a] = (unsigned)u;
  }
  const unsigned m = (spread10(q[0]) << 2) | (spread10(q[1]) << 1) | spread10(q[2]);
  keys[i] = ((unsigned long long)m << 32) | (unsigned)i;
}

// One thread per leaf j (sorted position): the object's device record (as Repack::emit_leaf packs it,
// one object per leaf), its box as the leaf's box, and where it came from.
__global__ void build_leaves_kernel(const unsigned long long *__restrict__ keys, int n, const rt_sphere *__restrict__ sph,
                                    int n_s, const rt_triangle *__restrict__ tri, const float *__restrict__ box_in,
                                    float *__restrict__ prims, float *__restrict__ leafbox, float *__restrict__ leaf_box6,
                                    uint32_t *__restrict__ prim_src) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int i = (int)(unsigned)keys[j];
  float rec[16];
  for (int k = 0; k < 16; ++k) rec[k] = 0.0f;
  if (i < n_s) {
    const rt_sphere p = sph[i];
    rec[0] = p.center[0]; rec[1] = p.center[1]; rec[2] = p.center[2];
    rec[3] = p.radius;
    rec[4] = p.radius * p.radius;           // radius*radius   RayTracer.hs:344
    rec[12] = __int_as_float(0);
    rec[13] = __int_as_float(p.material);
    rec[14] = __int_as_float(p.object_id);
    prim_src[j] = (uint32_t)i;
  } else {
    const rt_triangle t = tri[i - n_s];
    for (int a = 0; a < 3; ++a) {
      rec[0 + a] = t.a[a];
      rec[3 + a] = t.a[a] - t.b[a];         // (a,b,c) = A - B   RayTracer.hs:362
      rec[6 + a] = t.a[a] - t.c[a];         // (d,e,f) = A - C   RayTracer.hs:363
      rec[9 + a] = t.n[a];
    }
    rec[12] = __int_as_float(1);
    rec[13] = __int_as_float(t.material);
    rec[14] = __int_as_float(t.object_id);
    prim_src[j] = RT_REF_TRIANGLE | (uint32_t)(i - n_s);
  }
  rec[15] = __int_as_float(1);              // records in this leaf
  for (int k = 0; k < 16; ++k) prims[(size_t)j * 16 + k] = rec[k];
  for (int a = 0; a < 6; ++a) {
    const float v = box_in[(size_t)i * 6 + a];
    leafbox[(size_t)j * 8 + a] = v;
    leaf_box6[(size_t)j * 6 + a] = v;
  }
  leafbox[(size_t)j * 8 + 6] = leafbox[(size_t)j * 8 + 7] = 0.0f;
}

// Karras 2012: internal node i of the binary radix tree over the n sorted, distinct keys.
// child >= 0: internal node; child < 0: leaf -1 - child.  parent_* : the node above.
__device__ inline int key_delta(const unsigned long long *keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  return __clzll((long long)(keys[i] ^ keys[j]));
}
__global__ void build_karras_kernel(const unsigned long long *__restrict__ keys, int n, int2 *__restrict__ child,
                                    int *__restrict__ parent_of_inner, int *__restrict__ parent_of_leaf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const int d = key_delta(keys, n, i, i + 1) - key_delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
  const int dmin = key_delta(keys, n, i, i - d);
  int lmax = 2;
  while (key_delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (key_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = key_delta(keys, n, i, j);
  int s = 0;
  for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
    if (key_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    if (t == 1) break;
  }
  const int gamma = i + s * d + min(d, 0);
  const int lo = min(i, j), hi = max(i, j);
  const int left = (lo == gamma) ? -1 - gamma : gamma;
  const int right = (hi == gamma + 1) ? -1 - (gamma + 1) : gamma + 1;
  child[i] = make_int2(left, right);
  if (left >= 0) parent_of_inner[left] = i; else parent_of_leaf[-1 - left] = i;
  if (right >= 0) parent_of_inner[right] = i; else parent_of_leaf[-1 - right] = i;
  if (i == 0) parent_of_inner[0] = -1;
}

// Bottom-up: boxes (surround, BoundingVolumeHierarchy.hs:135-142: Haskell's min / max) and heights
// (WideBuild::height_of: leaf children count 0).  One thread per leaf climbs; at every node the first
// thread to arrive leaves, the second finds both children done.
__global__ void build_fit_kernel(int n, const int2 *__restrict__ child, const int *__restrict__ parent_of_inner,
                                 const int *__restrict__ parent_of_leaf, const float *__restrict__ leaf_box6,
                                 float *__restrict__ node_box, int *__restrict__ height, unsigned *__restrict__ arrived,
                                 BuildScalars *__restrict__ sc) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  int node = parent_of_leaf[j];
  while (node >= 0) {
    __threadfence();   // this thread's box below is visible before the counter says so
    if (atomicAdd(&arrived[node], 1u) == 0u) return;
    __threadfence();
    const int2 ch = child[node];
    float b[6];
    int h = 0;
    for (int k = 0; k < 2; ++k) {
      const int c = k ? ch.y : ch.x;
      const float *cb = c >= 0 ? node_box + (size_t)c * 6 : leaf_box6 + (size_t)(-1 - c) * 6;
      // (plain loads could be served by a stale line of this CU's L1: read through to L2)
      float v[6];
      for (int a = 0; a < 6; ++a) v[a] = __hip_atomic_load(cb + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (k == 0) for (int a = 0; a < 6; ++a) b[a] = v[a];
      else for (int a = 0; a < 3; ++a) {
        b[2 * a] = hmin_d(b[2 * a], v[2 * a]);
        b[2 * a + 1] = hmax_d(b[2 * a + 1], v[2 * a + 1]);
      }
      if (c >= 0) h = max(h, 1 + __hip_atomic_load(height + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    for (int a = 0; a < 6; ++a) __hip_atomic_store(node_box + (size_t)node * 6 + a, b[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(height + node, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (node == 0) sc->root_height = 1 + h;   // the pseudo root's height: one more level
    node = parent_of_inner[node];
  }
}

// Binary device records (rt_device.h): record 0 = pseudo root, record i + 1 = internal node i.
__global__ void build_records_kernel(int n, const int2 *__restrict__ child, const float *__restrict__ leaf_box6,
                                     const float *__restrict__ node_box, float *__restrict__ nodes) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;   // record index: max(1, n) records
  if (r >= max(1, n)) return;
  float rec[16];
  for (int k = 0; k < 16; ++k) rec[k] = 0.0f;
  auto ref_of = [](int c) { return c >= 0 ? c + 1 : c; };   // internal -> its record; leaf ref as it is (-1 - first record)
  if (r == 0) {
    const float *rb = n >= 2 ? node_box : leaf_box6;
    for (int a = 0; a < 6; ++a) rec[a] = n >= 1 ? rb[a] : 0.0f;
    rec[12] = __int_as_float(n >= 2 ? 1 : (n == 1 ? -1 : REF_NONE));
    rec[13] = __int_as_float(REF_NONE);
  } else {
    const int2 ch = child[r - 1];
    for (int k = 0; k < 2; ++k) {
      const int c = k ? ch.y : ch.x;
      const float *cb = c >= 0 ? node_box + (size_t)c * 6 : leaf_box6 + (size_t)(-1 - c) * 6;
      for (int a = 0; a < 6; ++a) rec[6 * k + a] = cb[a];
      rec[12 + k] = __int_as_float(ref_of(c));
    }
  }
  for (int k = 0; k < 16; ++k) nodes[(size_t)r * 16 + k] = rec[k];
}

// One level of the wide collapse (WideBuild::build): entry = (binary record, stack entries above, wide index).
// The child with the largest box is replaced by its own two children until there are four, as long as
// the deepest path below still fits under the stack cap.  Inner children go into the next level's queue
// with their wide index (records come out in breadth-first order).
__global__ void build_wide_level_kernel(const int3 *__restrict__ q_in, int3 *__restrict__ q_out, int level_parity,
                                        const float *__restrict__ nodes, const int *__restrict__ height, int cap,
                                        float *__restrict__ wide, BuildScalars *__restrict__ sc) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= sc->queue_n[level_parity]) return;
  const int3 e = q_in[k];
  const int bidx = e.x, above = e.y, me = e.z;
  float box[4][6];
  int ref[4];
  int cnt = 0;
  {
    const float *r = nodes + (size_t)bidx * 16;
    for (int s = 0; s < 2; ++s) {
      const int c = __float_as_int(r[12 + s]);
      if (c == REF_NONE) continue;
      for (int a = 0; a < 6; ++a) box[cnt][a] = r[6 * s + a];
      ref[cnt++] = c;
    }
  }
  auto h_of = [&](int rec) { return height[rec - 1]; };   // (records >= 1 are internal nodes)
  while (cnt < 4) {
    int below = 0;
    for (int i = 0; i < cnt; ++i)
      if (ref[i] >= 0) below = max(below, 1 + h_of(ref[i]));
    if (above + cnt + below > cap) break;
    int best = -1;
    float best_a = -1.0f;
    for (int i = 0; i < cnt; ++i)
      if (ref[i] >= 0) {
        const float w = box[i][1] - box[i][0], h = box[i][3] - box[i][2], d = box[i][5] - box[i][4];
        const float a = 2 * (w * h + w * d + h * d);
        if (!(a <= best_a)) { best_a = a; best = i; }
      }
    if (best < 0) break;
    const float *r = nodes + (size_t)ref[best] * 16;
    // its two children take its place (an internal node always has two)
    for (int i = cnt; i > best + 1; --i) {
      for (int a = 0; a < 6; ++a) box[i][a] = box[i - 1][a];
      ref[i] = ref[i - 1];
    }
    for (int s = 0; s < 2; ++s) {
      for (int a = 0; a < 6; ++a) box[best + s][a] = r[6 * s + a];
      ref[best + s] = __float_as_int(r[12 + s]);
    }
    ++cnt;
  }
  atomicMax(&sc->max_stack, above + cnt - 1);
  float *w = wide + (size_t)me * 32;
  for (int i = 0; i < 4; ++i) {
    int out_ref = REF_NONE;
    if (i < cnt) {
      for (int pl = 0; pl < 6; ++pl) w[pl * 4 + i] = box[i][pl];
      if (ref[i] >= 0) {
        const int idx = atomicAdd(&sc->n_wide, 1);
        const int slot = atomicAdd(&sc->queue_n[level_parity ^ 1], 1);
        q_out[slot] = make_int3(ref[i], above + cnt - 1, idx);
        out_ref = idx;
      } else {
        out_ref = ref[i];
      }
    } else {
      for (int pl = 0; pl < 6; ++pl) w[pl * 4 + i] = (pl & 1) ? -WideBuild::ABSENT_PLANE : WideBuild::ABSENT_PLANE;
    }
    w[24 + i] = __int_as_float(out_ref);
  }
  for (int i = 28; i < 32; ++i) w[i] = 0.0f;
}
__global__ void build_queue_reset_kernel(BuildScalars *sc, int parity) { sc->queue_n[parity] = 0; }

// The tree in rt_upload_scene's input format: internal node i -> node i (the root is node 0), leaf j -> node
// n - 1 + j with the one object of the leaf as its list; leaf_refs[j] = that object.
__global__ void build_export_kernel(int n, const int2 *__restrict__ child, const float *__restrict__ leaf_box6,
                                    const float *__restrict__ node_box, rt_bvh_node *__restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_inner = n - 1;
  if (k >= n_inner + n) return;
  rt_bvh_node o;
  const float *b = k < n_inner ? node_box + (size_t)k * 6 : leaf_box6 + (size_t)(k - n_inner) * 6;
  o.xmin = b[0]; o.xmax = b[1]; o.ymin = b[2]; o.ymax = b[3]; o.zmin = b[4]; o.zmax = b[5];
  if (k < n_inner) {
    const int2 ch = child[k];
    o.left = ch.x >= 0 ? ch.x : n_inner + (-1 - ch.x);
    o.right = ch.y >= 0 ? ch.y : n_inner + (-1 - ch.y);
  } else {
    o.left = -1 - (k - n_inner);
    o.right = 1;
  }
  out[k] = o;
}

template <class T>
int ensure_dev(rt_ctx *c, T **p, size_t *cap, size_t bytes) {
  if (*p && *cap >= bytes) return RT_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  const size_t want = std::max<size_t>(bytes + bytes / 4, 4096);
  HIP_TRY(c, hipMalloc((void **)p, want));
  *cap = want;
  return RT_OK;
}

}  // namespace

extern "C" {

int rt_build_scene(rt_ctx *c, const rt_sphere *spheres, int32_t n_spheres, const rt_triangle *triangles, int32_t n_triangles,
                   const rt_material *materials, int32_t n_materials, const rt_light *lights, int32_t n_lights) {
  if (!c) return RT_ERR_INVALID;
  if (n_spheres < 0 || n_triangles < 0 || n_materials < 0 || n_lights < 0) return fail(c, RT_ERR_INVALID, "negative count");
  if ((n_spheres && !spheres) || (n_triangles && !triangles) || (n_materials && !materials) || (n_lights && !lights))
    return fail(c, RT_ERR_INVALID, "null array with non-zero count");
  const long long n_ll = (long long)n_spheres + n_triangles;
  if (n_ll > (1ll << 24)) return fail(c, RT_ERR_LIMIT, "rt_build_scene: %lld objects exceed the limit of 2^24", n_ll);
  const int n = (int)n_ll;
  for (int i = 0; i < n_spheres; ++i)
    if (spheres[i].material < 0 || spheres[i].material >= n_materials) return fail(c, RT_ERR_INVALID, "sphere %d: material index out of range", i);
  for (int i = 0; i < n_triangles; ++i)
    if (triangles[i].material < 0 || triangles[i].material >= n_materials) return fail(c, RT_ERR_INVALID, "triangle %d: material index out of range", i);
  HIP_TRY(c, hipSetDevice(c->device));
  c->have_scene = false;
  c->prepared = false;
  hipStream_t s = c->stream;
  BuildBuffers &B = c->build;
  int rc;
  UploadBatch up;
  up.add(&c->d_spheres, &c->cap_spheres, spheres, (size_t)n_spheres * sizeof(rt_sphere));
  up.add(&c->d_triangles, &c->cap_triangles, triangles, (size_t)n_triangles * sizeof(rt_triangle));
  up.add(&c->d_materials12, &c->cap_materials12, materials, (size_t)n_materials * sizeof(rt_material));
  up.add(&c->d_materials, &c->cap_materials, nullptr, (size_t)std::max(1, n_materials) * MAT_F * sizeof(float));
  up.add(&c->d_lights, &c->cap_lights, lights, (size_t)n_lights * sizeof(rt_light));
  const size_t n1 = (size_t)std::max(1, n);
  up.add(&c->d_nodes, &c->cap_nodes, nullptr, n1 * 64);
  up.add(&c->d_prims, &c->cap_prims, nullptr, n1 * 64);
  up.add(&c->d_wnodes, &c->cap_wnodes, nullptr, n1 * 128);
  up.add(&c->d_leafbox, &c->cap_leafbox, nullptr, n1 * 32);
  up.add(&c->d_prim_src, &c->cap_prim_src, nullptr, n1 * sizeof(uint32_t));
  if ((rc = up.run(c))) return rc;   // (waits for the context's frame in flight before anything is overwritten)
  HIP_TRY(c, hipStreamSynchronize(s));
  if ((rc = ensure_dev(c, &B.box_in, &B.cap_box_in, n1 * 24)) || (rc = ensure_dev(c, &B.leaf_box6, &B.cap_leaf_box6, n1 * 24)) ||
      (rc = ensure_dev(c, &B.node_box, &B.cap_node_box, n1 * 24)) || (rc = ensure_dev(c, &B.keys, &B.cap_keys, n1 * 16)) ||
      (rc = ensure_dev(c, &B.child, &B.cap_child, n1 * sizeof(int2))) || (rc = ensure_dev(c, &B.ints, &B.cap_ints, n1 * 4 * sizeof(int))) ||
      (rc = ensure_dev(c, &B.queue, &B.cap_queue, n1 * 2 * sizeof(int3))) || (rc = ensure_dev(c, &B.scalars_mem, &B.cap_scalars, sizeof(BuildScalars))))
    return rc;
  BuildScalars *const d_scalars = static_cast<BuildScalars *>(B.scalars_mem);
  int *parent_of_inner = B.ints, *parent_of_leaf = B.ints + n1, *height = B.ints + 2 * n1;
  unsigned *arrived = reinterpret_cast<unsigned *>(B.ints + 3 * n1);
  unsigned long long *keys_in = B.keys, *keys_sorted = B.keys + n1;
  BuildScalars h{};
  for (int a = 0; a < 3; ++a) { h.lo[a] = 0xffffffffu; h.hi[a] = 0u; }
  HIP_TRY(c, hipMemcpyAsync(d_scalars, &h, sizeof h, hipMemcpyHostToDevice, s));
  (void)hipGetLastError();
  const int T = 256, G = (n + T - 1) / T;
  c->n_lights = n_lights;
  c->n_materials = n_materials;
  c->n_spheres = n_spheres;
  c->n_triangles = n_triangles;
  c->refit_ready = false;
  c->device_built = true;
  c->device_built_n = n;
  launch_material_prep(c->d_materials12, c->d_materials, n_materials, s);
  int root_height = 0;
  if (n > 0) {
    hipLaunchKernelGGL(build_boxes_kernel, dim3(G), dim3(T), 0, s, c->d_spheres, n_spheres, c->d_triangles, n_triangles, B.box_in, d_scalars);
    hipLaunchKernelGGL(build_keys_kernel, dim3(G), dim3(T), 0, s, B.box_in, n, d_scalars, keys_in);
    size_t tmp_bytes = 0;
    if (rocprim::radix_sort_keys(nullptr, tmp_bytes, keys_in, keys_sorted, (size_t)n, 0, 62, s) != hipSuccess)
      return fail(c, RT_ERR_DEVICE, "radix sort: size query failed");
    if ((rc = ensure_dev(c, &B.sort_tmp, &B.cap_sort_tmp, std::max<size_t>(tmp_bytes, 16)))) return rc;
    if (rocprim::radix_sort_keys(B.sort_tmp, tmp_bytes, keys_in, keys_sorted, (size_t)n, 0, 62, s) != hipSuccess)
      return fail(c, RT_ERR_DEVICE, "radix sort failed");
    hipLaunchKernelGGL(build_leaves_kernel, dim3(G), dim3(T), 0, s, keys_sorted, n, c->d_spheres, n_spheres, c->d_triangles, B.box_in,
                       reinterpret_cast<float *>(c->d_prims), reinterpret_cast<float *>(c->d_leafbox), B.leaf_box6, c->d_prim_src);
    if (n >= 2) {
      HIP_TRY(c, hipMemsetAsync(arrived, 0, (size_t)n * sizeof(unsigned), s));
      hipLaunchKernelGGL(build_karras_kernel, dim3(G), dim3(T), 0, s, keys_sorted, n, B.child, parent_of_inner, parent_of_leaf);
      hipLaunchKernelGGL(build_fit_kernel, dim3(G), dim3(T), 0, s, n, B.child, parent_of_inner, parent_of_leaf, B.leaf_box6, B.node_box,
                         height, arrived, d_scalars);
    }
    hipLaunchKernelGGL(build_records_kernel, dim3(G), dim3(T), 0, s, n, B.child, B.leaf_box6, B.node_box, reinterpret_cast<float *>(c->d_nodes));
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(&h, d_scalars, sizeof h, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    root_height = n >= 2 ? h.root_height : 0;
  } else {
    hipLaunchKernelGGL(build_records_kernel, dim3(1), dim3(T), 0, s, 0, B.child, B.leaf_box6, B.node_box, reinterpret_cast<float *>(c->d_nodes));
  }
  // depth of the deepest node with the root at 1 and leaves counted (Repack::max_depth)
  const int max_depth = n == 0 ? 0 : (n == 1 ? 1 : root_height + 1);
  if (max_depth > 64) return fail(c, RT_ERR_LIMIT, "device-built BVH depth %d exceeds the traversal stack limit of 64", max_depth);
  // the wide records, with the host collapse's stack cap (rt_upload_scene)
  const int per_wave = ((160 * 1024 / 4) - HOT_WORDS * 4) / WAVES_PER_BLOCK - (point_words_for(n_lights) + 32) * 4;
  const int depth_max = std::min(40, (per_wave / 256) & ~3);
  const int cap = std::max(std::max(3, depth_max - 1), max_depth);
  if (n > 0) {
    const int3 first = make_int3(0, 0, 0);
    int3 *q0 = B.queue, *q1 = B.queue + n1;
    HIP_TRY(c, hipMemcpyAsync(q0, &first, sizeof first, hipMemcpyHostToDevice, s));
    h.n_wide = 1;
    h.max_stack = 0;
    h.queue_n[0] = 1;
    h.queue_n[1] = 0;
    HIP_TRY(c, hipMemcpyAsync(d_scalars, &h, sizeof h, hipMemcpyHostToDevice, s));
    // a wide level consumes at least one binary level: at most root_height + 1 levels (grids sized for the
    // widest possible level; threads beyond the queue's length leave at once)
    const int gw = (std::max(1, n - 1) + T - 1) / T;
    for (int lv = 0; lv <= root_height; ++lv) {
      const int par = lv & 1;
      hipLaunchKernelGGL(build_wide_level_kernel, dim3(lv == 0 ? 1 : gw), dim3(T), 0, s, par ? q1 : q0, par ? q0 : q1, par,
                         reinterpret_cast<const float *>(c->d_nodes), height, cap, reinterpret_cast<float *>(c->d_wnodes), d_scalars);
      hipLaunchKernelGGL(build_queue_reset_kernel, dim3(1), dim3(1), 0, s, d_scalars, par);
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(&h, d_scalars, sizeof h, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (h.queue_n[0] != 0 || h.queue_n[1] != 0) return fail(c, RT_ERR_DEVICE, "wide collapse did not finish (internal error)");
    c->n_wnode_recs = h.n_wide;
    c->wide_stack = h.max_stack;
  } else {
    std::vector<float> w(32, 0.0f);
    for (int i = 0; i < 24; ++i) w[i] = ((i >> 2) & 1) ? -WideBuild::ABSENT_PLANE : WideBuild::ABSENT_PLANE;
    std::fill(w.begin() + 24, w.begin() + 28, Repack::as_f(REF_NONE));
    HIP_TRY(c, hipMemcpyAsync(c->d_wnodes, w.data(), 128, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    c->n_wnode_recs = 1;
    c->wide_stack = 0;
  }
  c->n_node_recs = std::max(1, n);     // pseudo root + internal nodes
  c->n_prim_recs = n;
  c->has_root = n > 0;
  c->tree_depth = max_depth;
  c->boxes_safe = n > 0 ? (h.unsafe == 0) : 1;
  {
    float am;
    memcpy(&am, &h.absmax_bits, 4);
    c->box_absmax = n > 0 ? am : 0.0f;
  }
  c->nested = true;            // unions of the children's boxes; leaf boxes are their object's toBox
  c->h_prim_src.clear();
  c->h_wide_src.clear();
  c->h_level_nodes.clear();
  c->h_level_start.assign(1, 0);
  c->have_scene = true;
  return RT_OK;
}

int rt_download_tree(rt_ctx *c, rt_bvh_node *nodes_out, int32_t cap_nodes, uint32_t *leaf_refs_out, int32_t cap_refs,
                     int32_t *n_nodes_out, int32_t *n_refs_out) {
  if (!c || !n_nodes_out || !n_refs_out) return RT_ERR_INVALID;
  if (!c->have_scene || !c->device_built) return fail(c, RT_ERR_STATE, "rt_download_tree: the context's tree was not built by rt_build_scene");
  const int n = c->device_built_n;
  const int n_nodes = n == 0 ? 0 : 2 * n - 1;
  *n_nodes_out = n_nodes;
  *n_refs_out = n;
  if (!nodes_out && !leaf_refs_out) return RT_OK;   // size query
  if (cap_nodes < n_nodes || cap_refs < n || (n_nodes && !nodes_out) || (n && !leaf_refs_out))
    return fail(c, RT_ERR_INVALID, "rt_download_tree: room for %d nodes / %d refs needed", n_nodes, n);
  if (n == 0) return RT_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  BuildBuffers &B = c->build;
  int rc = ensure_dev(c, &B.export_nodes, &B.cap_export_nodes, (size_t)n_nodes * sizeof(rt_bvh_node));
  if (rc) return rc;
  (void)hipGetLastError();
  hipLaunchKernelGGL(build_export_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, c->stream, n, B.child, B.leaf_box6, B.node_box, B.export_nodes);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(nodes_out, B.export_nodes, (size_t)n_nodes * sizeof(rt_bvh_node), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(leaf_refs_out, c->d_prim_src, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return RT_OK;
}

}  // extern "C"
