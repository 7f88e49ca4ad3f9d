/*
 * rt_mi355x.h -- C ABI of the MI355X trace/shade library (librt_mi355x.so).
 *
 * This is the drop-in boundary for ONE path of jrraymond/ray-tracer: everything
 * that happens underneath
 *
 *     render :: PureMT -> World -> [Color]              (src/RayTracer.hs:18-23)
 *
 * i.e. colorPixel / getRay / raytrace / hitBVH / hits / hit / hitsBox /
 * getDirectColor / inShadow / getIndirectColor / reflect / refract / orthonormal
 * / avgColors.  The reference has no FFI of its own (it is pure Haskell); this
 * header is what a `foreign import ccall safe` shim placed under `render` binds
 * (see INTEGRATION.md for the Haskell side).  The Haskell host keeps building
 * the scene, the SAH BVH (src/BoundingVolumeHierarchy.hs:34-51) and the sample
 * grids (src/RayTracer.hs:462-475); it flattens them into the PODs below.
 *
 * Conventions: plain C, no torch types; every call returns 0 on success or a
 * negative rt_status; nothing throws or aborts across the boundary; the library
 * copies host inputs on upload and never keeps a host pointer; all outputs are
 * caller-allocated.  A context is bound to one GPU, is not thread-safe, and all
 * calls are synchronous unless the name ends in _async.  A context has ONE frame in
 * flight (one set of work buffers): frames enqueued on the same stream queue up behind
 * each other; a frame enqueued on another stream than the context's previous one first
 * waits (on the host) for that one -- use one context per stream to overlap frames.  A
 * caller's stream must stay alive until the context's next call.
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 1

typedef enum rt_status {
  RT_OK = 0,
  RT_ERR_INVALID = -1,   /* bad argument / inconsistent scene              */
  RT_ERR_STATE = -2,     /* call order (render before upload, ...)         */
  RT_ERR_DEVICE = -3,    /* HIP runtime error (text in rt_last_error)      */
  RT_ERR_LIMIT = -4,     /* a stated limit exceeded (tree depth, sizes...) */
  RT_ERR_NOMEM = -5
} rt_status;

/* ---- scene PODs ------------------------------------------------------- */

/* One BVH tree node.  Mirrors `data BVH = Node !BVH !BVH !Box | Leaf ![Object] !Box`
 * (src/Types.hs:97-100); the six floats are the fields of `Box` in declaration
 * order (src/Types.hs:88-93).
 *   inner node: left >= 0 and right >= 0 are node indices.
 *   leaf      : left  = -1 - first   (first = offset into leaf_refs)
 *               right = count        (>= 1; the leaf's object list, in list order)
 * Node 0 is the root.  `Empty` (src/Types.hs:99) is n_nodes == 0. */
typedef struct rt_bvh_node {
  float xmin, xmax, ymin, ymax, zmin, zmax;
  int32_t left, right;
} rt_bvh_node;

/* leaf_refs entry: bit 31 set = triangle, clear = sphere; low 31 bits = index
 * into the triangle / sphere array. */
#define RT_REF_TRIANGLE 0x80000000u
#define RT_REF_INDEX(r) ((r) & 0x7fffffffu)

/* `Sphere pos rad mat` (src/Types.hs:74-76).  object_id = position of the object in
 * the host's [Object] list; it is what primary_hit_ids reports. */
typedef struct rt_sphere {
  float center[3];
  float radius;
  int32_t material;
  int32_t object_id;
} rt_sphere;

/* `Triangle a b c n mat` (src/Types.hs:78-82); n is the STORED normal (it is not
 * recomputed: src/RayTracer.hs:77, and scenes hand-write it, BenchmarkScene.hs:50-58). */
typedef struct rt_triangle {
  float a[3], b[3], c[3], n[3];
  int32_t material;
  int32_t object_id;
} rt_triangle;

/* `Material` (src/Types.hs:27-37).  log_atten is what makeMaterial stores:
 * log of the attenuation colour (src/Surfaces.hs:22-24). */
typedef struct rt_material {
  float diffuse[3];
  float specular[3];
  float phong;
  float refl;
  float refr_ix;
  float log_atten[3];
} rt_material;

/* `Light corner edgeA edgeB color` (src/Types.hs:102-106). */
typedef struct rt_light {
  float corner[3], edge_a[3], edge_b[3], color[3];
} rt_light;

typedef struct rt_scene {
  const rt_bvh_node *nodes;     int32_t n_nodes;
  const uint32_t *leaf_refs;    int32_t n_refs;
  const rt_sphere *spheres;     int32_t n_spheres;
  const rt_triangle *triangles; int32_t n_triangles;
  const rt_material *materials; int32_t n_materials;
  const rt_light *lights;       int32_t n_lights;
} rt_scene;

/* The scalar fields of `World` (src/Types.hs:132-146) as configToWorld fills
 * them (src/Main.hs:104-120): aa = wAntiAliasing = floor(sqrt(rays per pixel)),
 * (u,v,w) = getCam eye lookAt up (src/Main.hs:122-126). */
typedef struct rt_camera {
  int32_t img_w, img_h;          /* wImgWd, wImgHt (whole numbers)          */
  float view_w, view_h, view_d;  /* wViewWd, wViewHt, wViewDt               */
  int32_t aa;                    /* wAntiAliasing; samples per pixel = aa*aa */
  float lens;                    /* wLens                                    */
  float up[3], eye[3];           /* wUp (unused by the path), wEye           */
  float u[3], v[3], w[3];        /* wCamera                                  */
  float ambient[3];              /* wAmbient                                 */
  int32_t max_depth;             /* wMaxDepth                                */
} rt_camera;

/* ---- render flags ----------------------------------------------------- */

/* Arithmetic mode.  STRICT (default, 0): every fp32 add/sub/mul/div/sqrt is a
 * separate IEEE operation in the reference's order (GHC's x86-64 code generator
 * emits no FMA) -> hit decisions are bit-identical to the CPU oracle.
 * FMA: same kernels compiled with -ffp-contract=fast (Geometry3 math contracted
 * to v_fma_f32); hit decisions on ill-conditioned self-intersections may differ,
 * the mismatch rate is reported by the parity tests. */
#define RT_FLAG_FMA            0x1u
/* Trace every ray the reference traces: also reflection rays whose weight is
 * exactly (0,0,0) (src/RayTracer.hs:129-131 always recurses) and inShadow rays
 * towards lights whose diffuse AND specular factors are both exactly zero for
 * the point (max 0 (l.n) = 0 and max 0 (n.h) = 0, src/RayTracer.hs:101-105: the
 * light adds 0 + 0 whether occluded or not).  Off by default: such rays can only
 * change the result if they produce inf/NaN; they are then neither traced nor
 * counted in rt_stats. */
#define RT_FLAG_TRACE_ZERO_WEIGHT 0x2u
/* Visit every node whose ancestors pass hitsBox, like hitBVH
 * (src/RayTracer.hs:404-412): no front-to-back order, no t-range pruning.
 * Without this flag the traversal is ordered and pruned and tests the inner boxes
 * conservatively; it returns the same hits as long as every child box lies inside its
 * parent's (sahBVH's do: BoundingVolumeHierarchy.hs:50).  rt_upload_scene checks that and
 * walks a tree whose boxes do not nest as if this flag were set. */
#define RT_FLAG_NO_PRUNE       0x4u
/* Count box / sphere / triangle tests (slower kernels; for roofline numbers). */
#define RT_FLAG_COUNT_TESTS    0x8u

#define RT_HIT_NONE 0xffffffffu

typedef struct rt_stats {
  uint64_t rays_primary;      /* rays actually traversed, by kind          */
  uint64_t rays_shadow;
  uint64_t rays_reflection;
  uint64_t rays_refraction;
  uint64_t box_tests;         /* only with RT_FLAG_COUNT_TESTS             */
  uint64_t sphere_tests;
  uint64_t triangle_tests;
  uint64_t primary_hits;
  uint64_t shaded_hits;       /* closest hits of any generation (shaded points) */
  uint64_t wave_ticks_sum;    /* diagnostics, 100 MHz ticks: sum / max of the lifetimes  */
  uint64_t wave_ticks_max;    /* of the persistent waves, and their number (load balance) */
  uint64_t n_waves;
  double ms_kernel;           /* render_kernel alone, HIP events on its stream      */
  double ms_launch;           /* probe + order + render + resolve kernels            */
  double ms_total;            /* whole call incl. copies                      */
} rt_stats;

typedef struct rt_ctx rt_ctx;

/* ---- entry points ----------------------------------------------------- */

/* Create a context on HIP device `device`.  No counterpart in the reference
 * (which has no device); it stands where Main.run builds the World
 * (src/Main.hs:41-44). */
int rt_init(int device, rt_ctx **out);
void rt_shutdown(rt_ctx *ctx);

/* Upload wObjects (the BVH with its objects and their materials) and wLights:
 * replaces the in-memory `World` fields of src/Types.hs:142-144. */
int rt_upload_scene(rt_ctx *ctx, const rt_scene *scene);

/* Animation (BASELINE config 5): the objects of the uploaded scene have MOVED --
 * same counts, same order in the arrays, same materials -- and the tree keeps the topology of the
 * last rt_upload_scene.  Every box is refitted bottom-up ON THE GPU: toBox per object, a leaf's box
 * = left fold of `surround` over its objects, a node's = `surround` of its children's
 * (BoundingVolumeHierarchy.hs:50-51,135-142,178-189; Haskell's min/max), i.e. exactly the boxes
 * sahBVH would attach to this topology; the primitive records and the wide records follow.  No
 * host rebuild, one staged copy of the two object arrays.  This is a TREE-CHANGING mode with
 * respect to a host that rebuilds sahBVH every frame: closest hits are the same, but among
 * objects hit at EQUAL distance the winner follows the stale topology's leaf order
 * (src/RayTracer.hs:383-402,410).  Rebuild (rt_upload_scene) every few frames to keep the tree
 * tight.  Enqueued on the context's own stream (hip_stream must be NULL), behind its frame in
 * flight; host arrays are copied before the call returns. */
int rt_refit_async(rt_ctx *ctx, const rt_sphere *spheres, int32_t n_spheres, const rt_triangle *triangles,
                   int32_t n_triangles, void *hip_stream);

/* Animation whose objects move so far (or change in number) that the old tree will not do: BUILD the
 * tree on the GPU instead of handing one over (the reference's builder,
 * src/BoundingVolumeHierarchy.hs:34-94, stays the host's for still scenes).  The objects are uploaded
 * and a linear BVH is built over them on the context's stream: Morton order of the box centres,
 * Karras' parallel radix tree, one object per leaf, boxes bottom-up with toBox / surround
 * (BoundingVolumeHierarchy.hs:135-142,178-189), then the device records as rt_upload_scene would
 * make them (100 000 triangles: about a millisecond, against 0.2 s for the host sahBVH and 9 ms for
 * its upload).  Replaces rt_upload_scene for that frame; camera and samples are kept.
 * This is a TREE-CHANGING mode: the tree is not sahBVH's, so (i) among hits at EQUAL distance the
 * winner may differ from a frame rendered with the host's tree (the right subtree wins:
 * src/RayTracer.hs:383-387,410), (ii) the reference's un-pruned visits differ.  The frame equals what
 * the reference's algorithm returns when it walks THIS tree: rt_download_tree hands it back in
 * rt_scene's format for exactly that check (the objects keep their positions in the two arrays:
 * leaf_refs index them as in rt_upload_scene).  rt_refit_async does not work on such a tree (build
 * again).  Blocking (two small read-backs); host arrays are copied before it returns.  Limit: 2^24
 * objects, depth <= 64 (RT_ERR_LIMIT). */
int rt_build_scene(rt_ctx *ctx, const rt_sphere *spheres, int32_t n_spheres, const rt_triangle *triangles,
                   int32_t n_triangles, const rt_material *materials, int32_t n_materials, const rt_light *lights,
                   int32_t n_lights);
/* The tree rt_build_scene made, as rt_upload_scene's input: 2 * n - 1 nodes (node 0 = the root; every
 * leaf holds one object) and n leaf_refs.  nodes_out = leaf_refs_out = NULL: only the two counts. */
int rt_download_tree(rt_ctx *ctx, rt_bvh_node *nodes_out, int32_t cap_nodes, uint32_t *leaf_refs_out, int32_t cap_refs,
                     int32_t *n_nodes_out, int32_t *n_refs_out);

/* The remaining World fields (src/Types.hs:132-141,143,145). */
int rt_set_camera(rt_ctx *ctx, const rt_camera *cam);

/* The output of generateGrids (src/RayTracer.hs:462-467) before `cycle`:
 * n_grids (= img_w + 10, src/RayTracer.hs:20) grids of spp (= aa*aa) samples of
 * `F6 p q r0 r1 s0 s1` (src/Types.hs:108-114); f6 is [n_grids][spp][6] floats.
 * Pixel k in render's list order uses grid k mod n_grids. */
int rt_set_samples(rt_ctx *ctx, const float *f6, int32_t n_grids, int32_t spp);

/* Static tile partition for data-parallel rendering: this context renders the
 * tiles t with t % nranks == rank.  (0,1) = whole frame (default). */
int rt_set_partition(rt_ctx *ctx, int32_t rank, int32_t nranks);

/* Geometry of the tile decomposition for the current camera/samples. */
typedef struct rt_tiling {
  int32_t tile_w, tile_h;      /* pixels                                  */
  int32_t tiles_x, tiles_y;    /* tiles across / down the image           */
  int32_t n_tiles;             /* tiles_x * tiles_y                       */
  int32_t n_local_tiles;       /* tiles of this rank                      */
} rt_tiling;
int rt_get_tiling(rt_ctx *ctx, rt_tiling *out);

/* render (src/RayTracer.hs:18-23).  rgb_out: img_w*img_h*3 floats, pixel order
 * of render's list (top row first, j = H-1..0, i = 0..W-1).  With a partition,
 * pixels of other ranks' tiles are set to 0.  primary_hit_ids (may be
 * NULL): img_w*img_h*spp object ids of the first-generation hit (RT_HIT_NONE =
 * miss), sample-major within the pixel.  stats may be NULL.  Host pointers. */
int rt_render(rt_ctx *ctx, uint32_t flags, float *rgb_out, uint32_t *primary_hit_ids,
              rt_stats *stats);

/* Same, device pointers, enqueued on `hip_stream` (a hipStream_t, NULL = the
 * context's own stream) without a host sync; inputs stay resident in HBM.
 * d_rgb as rgb_out above.  With nranks > 1 use rt_render_tiles_async instead. */
int rt_render_async(rt_ctx *ctx, uint32_t flags, float *d_rgb, uint32_t *d_primary_hit_ids,
                    void *hip_stream);

/* rt_render_async for callers that cannot allocate device memory: tight serving loops, harnesses
 * that forbid hipMalloc, code that must not trigger an allocation at this point.  NO counterpart
 * in the reference.  Same inputs, outputs and results (bit for bit) as rt_render_async, but every
 * per-frame work buffer (pixel sums, ticket counters, packet order, per-wave ray queues,
 * statistics) lives in a static device array of the library, at most 32 MiB per GPU, held
 * from library load to process end, instead of the context's hipMalloc'ed buffers: the call
 * performs no hipMalloc / hipFree, nor any other allocating routine.  Limits (RT_ERR_LIMIT with
 * a message beyond them): img_w x img_h rounded up to whole tiles <= 262 144 pixels (e.g.
 * 640x360, 512x512), max_depth <= 6, whole frames only (partition (0,1)); the launch uses at
 * most 512 persistent waves, so it is slower than rt_render_async on large frames.  The
 * scene, camera and samples are set up beforehand with the usual calls (which do allocate).
 * ALL calls on one GPU -- from any context -- share that GPU's array: calls that overlap on a
 * GPU race; serialise them (one stream, or events).  rt_collect works as usual afterwards, but
 * the statistics block is part of the shared array: it is valid until the NEXT static call on
 * that GPU, whichever context makes it.  Allocating nothing is all this call promises: it still
 * waits for the context's previous launch when that went to another stream, and with timing on
 * (rt_set_timing) it records events -- neither is capture-safe, and it has not been run under
 * stream capture. */
int rt_render_static_async(rt_ctx *ctx, uint32_t flags, float *d_rgb, uint32_t *d_primary_hit_ids,
                           void *hip_stream);

/* Partitioned form: writes this rank's tiles compactly,
 * d_tiles[n_local_tiles][tile_h][tile_w][3] floats (local tile l is global tile
 * rank + l*nranks); the buffers of all ranks are gathered (RCCL) and passed to
 * rt_assemble_async on the root. */
int rt_render_tiles_async(rt_ctx *ctx, uint32_t flags, float *d_tiles, void *hip_stream);

/* Root side of the gather: d_gathered holds the nranks compact tile buffers
 * back to back, each padded to `stride_floats` floats; scatters them into
 * d_rgb (img_w*img_h*3 floats, render's order). */
int rt_assemble_async(rt_ctx *ctx, const float *d_gathered, int32_t nranks,
                      size_t stride_floats, float *d_rgb, void *hip_stream);

/* The same for frames quantised BEFORE the exchange (rt_quantize_async on each rank's compact
 * tile buffer, then a gather of bytes: a quarter of the traffic into the root): 8-bit compact
 * buffers, `stride_bytes` bytes apart, into d_rgb8 (img_w*img_h*3 bytes). */
int rt_assemble8_async(rt_ctx *ctx, const uint8_t *d_gathered8, int32_t nranks, size_t stride_bytes,
                       uint8_t *d_rgb8, void *hip_stream);

/* Wait for the stream of the last render call and fill stats (ray counters and
 * ms_kernel of that call). */
int rt_collect(rt_ctx *ctx, void *hip_stream, rt_stats *stats);

/* The 8-bit quantisation of writePPM (src/Main.hs:99-101,128-138):
 * out = truncate(255 * clamp x).  n = number of floats.  Device pointers. */
int rt_quantize_async(rt_ctx *ctx, const float *d_rgb, uint8_t *d_rgb8, size_t n,
                      void *hip_stream);

/* The context's own HIP stream (a hipStream_t, created non-blocking by rt_init): what the
 * _async calls use when hip_stream is NULL.  Two contexts on one GPU render on two hardware
 * queues, so alternate frames overlap (the next frame's blocks fill the wave slots the
 * current frame's tail leaves empty).  No counterpart in the reference. */
int rt_get_stream(rt_ctx *ctx, void **hip_stream_out);

/* Tell the context that `n_concurrent` contexts render on this GPU at the same time (frames in
 * flight): its persistent kernel then launches 1/n_concurrent of the resident wave capacity,
 * so the frames run side by side from the start instead of queueing for wave slots
 * (measured, 4 contexts: 1/8-frame share 0.55 -> 0.39 ms per frame).  1 = whole GPU (default). */
int rt_set_concurrency(rt_ctx *ctx, int32_t n_concurrent);

/* Frame timing on (default) / off.  On: every launch records the four events behind
 * rt_stats.ms_launch and ms_kernel.  Off: those fields read 0 and a frame is four stream
 * operations instead of eight -- worth 3 % when frames take a few hundred microseconds (a
 * 1/8 share of the headline frame: 0.40 -> 0.39 ms).  No counterpart in the reference. */
int rt_set_timing(rt_ctx *ctx, int32_t on);

/* Diagnostics of the last render, four words per persistent wave w (100 MHz ticks):
 * out[4w] lifetime, out[4w+1] rays it traversed, out[4w+2] = (scheduling steps << 40) | (start tick mod 2^40),
 * out[4w+3] ticks after its start at which the packet supply ran dry.  max_waves <= 16384. */
int rt_wave_diag(rt_ctx *ctx, uint64_t *out, int32_t max_waves);

const char *rt_last_error(rt_ctx *ctx);
int rt_abi_version(void);

/* ---- one host process, N GPUs ------------------------------------------ */

/* What a single-process host (the Haskell program is one) needs to render ONE frame on the N
 * GPUs of a node: SURVEY.md 8(b)'s `rt_init(n_gpus)` and BASELINE.json's "image tiles split
 * statically across the GPUs, framebuffer assembled with an RCCL gather over xGMI".  No
 * counterpart in the reference (render is one sequential map, src/RayTracer.hs:23).  An rt_multi
 * owns one rt_ctx per GPU; scene, camera and samples are replicated; tile t of the frame
 * (rt_get_tiling) is rendered by rank t % N; the ranks' compact tile buffers are gathered on rank
 * 0 with grouped RCCL point-to-point transfers (every peer has its own xGMI link into rank 0) and
 * scattered into the frame.  The frame is bitwise identical for every N.
 * devices: n_gpus HIP device ids, NULL = 0..n_gpus-1.  Ids may repeat (a rehearsal of the N-rank
 * data path on fewer GPUs): the exchange then uses device-to-device copies instead of RCCL, which
 * refuses duplicate devices.  RCCL is loaded at run time by this call only.
 * On failure *out is still set when possible, so that rt_multi_last_error can be read; free it
 * with rt_multi_shutdown.  A lone GPU (n_gpus = 1) works without RCCL installed (it has nobody to
 * exchange with); with RCCL it sends itself a few bytes per frame so that the binding is exercised.
 * State: rt_multi_upload_scene / _refit / _set_camera / _set_samples / _set_concurrency return
 * RT_ERR_STATE while a frame is in flight (between rt_multi_submit and rt_multi_wait).
 * NOTE: the N > 1 RCCL exchange (distinct devices) has been exercised by the builders on one GPU only
 * (N = 1 through RCCL; N = 3 and 8 ranks sharing the GPU through the device-copy branch): on real
 * multi-GPU nodes it is unverified. */
typedef struct rt_multi rt_multi;
int rt_multi_init(int32_t n_gpus, const int32_t *devices, rt_multi **out);
void rt_multi_shutdown(rt_multi *m);
int rt_multi_upload_scene(rt_multi *m, const rt_scene *scene);
int rt_multi_refit(rt_multi *m, const rt_sphere *spheres, int32_t n_spheres, const rt_triangle *triangles,
                   int32_t n_triangles);   /* rt_refit_async on every GPU */
int rt_multi_build_scene(rt_multi *m, const rt_sphere *spheres, int32_t n_spheres, const rt_triangle *triangles,
                         int32_t n_triangles, const rt_material *materials, int32_t n_materials, const rt_light *lights,
                         int32_t n_lights);   /* rt_build_scene on every GPU (the same tree on each) */
int rt_multi_set_camera(rt_multi *m, const rt_camera *cam);
int rt_multi_set_samples(rt_multi *m, const float *f6, int32_t n_grids, int32_t spp);
/* render on all GPUs (blocking).  rgb_out: img_w*img_h*3 floats as rt_render, or NULL;
 * rgb8_out: the frame as writePPM quantises it (src/Main.hs:99-101,131), img_w*img_h*3 bytes, or
 * NULL; at least one of them.  With rgb8_out alone every rank quantises its own tiles and BYTES are
 * gathered.  stats (may be NULL): sums over the ranks; times = the slowest rank's.  Host pointers. */
int rt_multi_render(rt_multi *m, uint32_t flags, float *rgb_out, uint8_t *rgb8_out, rt_stats *stats);
/* The two halves of rt_multi_render, for hosts that keep frames in flight.  rt_multi_submit
 * enqueues the frame on every GPU (render, exchange, assembly, copy to the host pointers) and
 * returns without waiting; rt_multi_wait blocks until that frame is in rgb_out / rgb8_out (which
 * must stay valid until then) and returns the statistics.  One rt_multi has one frame in flight:
 * a host that wants K frames in flight makes K rt_multi on the same devices and goes round them,
 *     submit(m[0]) submit(m[1]) .. wait(m[0]) submit(m[0]) wait(m[1]) submit(m[1]) ..
 * -- the tail of a frame is a few deep ray trees, and the next frame's work fills the wave slots
 * it leaves idle (c2 on one GPU, 8-bit frame delivered to host memory: 2.47 ms per frame one at a
 * time, 1.74 ms with four in flight on a third of the wave slots each, rt_multi_set_concurrency;
 * tools/multi_pipeline.py).  The frame travels through pinned staging memory of the library, so
 * that rt_multi_submit does not wait for it; rt_multi_wait copies it to the caller's pointers.
 * Frames do not depend on how many are in flight.  The exchanges of all rt_multi of a process run in
 * ONE order on every GPU, the order of the rt_multi_submit calls (each is enqueued behind an event
 * recorded after the previous exchange on that GPU; submits are serialised by a mutex), so that two
 * communicators' transfers can never be scheduled in opposite orders on two GPUs.  rt_stats.ms_total
 * stays 0 (it is rt_render's host-side bracket).  No counterpart in the reference. */
int rt_multi_submit(rt_multi *m, uint32_t flags, float *rgb_out, uint8_t *rgb8_out);
int rt_multi_wait(rt_multi *m, rt_stats *stats);
/* rt_set_concurrency on every GPU: with K rt_multi in flight, each launching 1/K of the wave slots
 * is usually the fastest split (the choice never changes a pixel). */
int rt_multi_set_concurrency(rt_multi *m, int32_t n);
const char *rt_multi_last_error(rt_multi *m);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355X_H */
