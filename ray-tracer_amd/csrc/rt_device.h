// rt_device.h -- private interface between the host side of librt_mi355x.so
// (rt_api.cpp) and its HIP kernels (rt_kernels.inl, compiled twice: strict / fma).
#pragma once

#include <hip/hip_runtime.h>
#include "rt_fastdiv.h"
#include <stdint.h>

#include "rt_mi355x.h"

namespace rtdev {

// ---- device-side scene layout (HBM; tiny scenes live in L1/L2) -------------
//
// Node record, 64 B = 4 x float4.  An INNER node carries the boxes of BOTH its
// children, so one 64-B fetch decides both hitsBox tests
// (BoundingVolumeHierarchy.hs:191-204) that hitBVH (RayTracer.hs:404-412) performs
// on entering the children:
//   f[0..5]   left  child box  xmin xmax ymin ymax zmin zmax
//   f[6..11]  right child box
//   i[12]     left  child ref      i[13] right child ref
//   ref >= 0: inner node index;  ref < 0, ref != REF_NONE: leaf, first prim = -1-ref
// Record 0 is a pseudo root whose left child is the real root (so the root's own
// box is tested like every other) and whose right child is REF_NONE.
constexpr int REF_NONE = INT32_MIN;
constexpr int NODE_F4 = 4;

// Primitive record, 64 B = 4 x float4, stored in depth-first leaf order; inside a
// leaf the object list is stored REVERSED, so that among hits with equal t the
// larger record index is the one hitBVH's `max` / hits' strict `<` would keep
// (right subtree wins, first object of a leaf wins: RayTracer.hs:383-402,410).
//   triangle: f[0..2] A   f[3..5] A-B   f[6..8] A-C   f[9..11] N
//   sphere  : f[0..2] C   f[3] r        f[4] r*r
//   i[12] kind (0 sphere, 1 triangle)  i[13] material  i[14] object id
//   i[15] number of records in this leaf (valid in the leaf's first record)
constexpr int PRIM_F4 = 4;
// WIDE node record (pruned traversal), 128 B = 8 x float4: up to FOUR children, collapsed at upload
// from two levels of the binary tree (the child with the largest box is replaced by its own two
// children until there are four).  One fetch decides four hitsBox tests and the dependent chain
// pop -> fetch -> test -> descend is walked half as often.  Boxes are stored plane by plane,
//   f4[0] xmin of the 4 children  f4[1] xmax   f4[2] ymin  f4[3] ymax   f4[4] zmin  f4[5] zmax
//   f4[6] the 4 child refs (as in the binary record; REF_NONE = no such child, with an inverted box)
// so that a ray whose direction sign on an axis is s reads its NEAR planes at f4[2a + s] and its
// FAR planes at f4[2a + (s ^ 1)]: hitsBox's six sign selects per box (BoundingVolumeHierarchy.hs:
// 194-199) become three per-ray address offsets.  In the reference's arithmetic a box's verdict
// implies its ancestors' (their planes are min / max of the children's, rounding is monotone), so
// the primitives the reference tests are those whose LEAF box hitsBox accepts: skipping the
// collapsed levels' own boxes changes nothing, the inner tests may err towards "hit", and the leaf
// box of the primitive that wins is what gets the exact question (rt_kernels.inl, RayBox /
// leaf_box_clear; leaf boxes: RenderParams::leafbox).  The closest hit is order-independent under
// the total order of the tie rule.  Record 0 is the root.
constexpr int WNODE_F4 = 8;
constexpr int WIDE_BFS_TOP = 1024;  // the first this-many wide records are the top of the tree in breadth-first order (rt_api.hip)
constexpr int LDS_WNODE_F4 = 9;  // LDS stride of a wide record: 36 dwords, walks all sixteen 4-bank groups
constexpr int LDS_REC_F4 = 5;  // stride of a node/primitive record in the LDS scene copy (80 B, see rt_kernels.inl)

// Material record on the device, MAT_F floats: [0..11] rt_material (diffuse rgb, specular rgb, phong,
// reflectivity, refractive index nt, attenuation rgb), then what getIndirectColor / refract
// (RayTracer.hs:109-179) compute from the material alone, formed ONCE per scene by
// material_prep_kernel with the very operations the reference performs per hit:
//   [12] 1 / phong                      (:113, the glossy jitter's a)
//   [13] (nt-1)*(nt-1) / ((nt+1)*(nt+1))  (:145, Schlick's r_0)
//   [14] 1 / nt                         (:138 the index handed to refract on leaving; :166 n / nt, n = 1, on entering)
//   [15] 1 / (1 / nt)                   (:166 n / nt on leaving)
//   [16..18] exp(-attenuation)          (:139-141, k on leaving)      [19] unused
constexpr int MAT_F = 20;
constexpr int QCAP = 192;       // ray-queue capacity per level: 63 left over + 2*64 pushed
constexpr int QFIELDS = 12;     // o.xyz d.xyz w.rgb s0 s1 slot
// LDS words per wave for the shadow-ray pool (multi-light builds): x, y, z, s0, s1, occlusion mask of
// the step's points, hand-out table (64 words each), then two mask words per light of a 32-light round
inline int point_words_for(int n_lights) { return n_lights > 1 ? 7 * 64 + ((2 * (n_lights < 32 ? n_lights : 32) + 3) & ~3) : 0; }
constexpr int MAX_DEPTH = 30;   // wMaxDepth limit (levels kept in an LDS counter array)
constexpr int WAVES_PER_BLOCK = 4;
constexpr int HOT_WORDS = 64;    // LDS words per block for RenderParams::hot and the first light
constexpr int PACKET_SHARDS = 16;  // packet counters, one 128-B line each (<= ~88 atomics/us per word)
// Statistics block (unsigned long long words): [0..15] frame totals (rt_stats order), [16 .. 16 + 4 * 16384)
// four words per wave (rt_wave_diag), then STATS_DIAG_WORDS of a profiling build's sums: [0..11] phase
// cycles, [12] steps, [13] active lanes, [16..23] traversal iterations / lane shares, [24..30] iteration and re-arm cycles, [32..71] deepest
// stack per closest-hit traversal, [72..111] per shadow ray of the pool, [112..119] wide-node visits
// with record index < 64 << k (k = 0..6) and all, [120..128] where the lanes are per loop iteration (Counters::where)
constexpr int STATS_WAVE_BASE = 16, STATS_MAX_WAVES = 16384;
constexpr int STATS_DIAG_BASE = STATS_WAVE_BASE + 4 * STATS_MAX_WAVES, STATS_DIAG_WORDS = 160;
// -DRT_TRACE_STEPS (tools/step_trace.py, never the shipped library): wall-clock stamps (100 MHz) of the
// phase marks of the first STATS_TRACE_STEPS scheduling steps of the first STATS_TRACE_WAVES waves
#ifdef RT_TRACE_STEPS
constexpr int STATS_TRACE_WAVES = 64, STATS_TRACE_STEPS = 48, STATS_TRACE_MARKS = 12;
#else
constexpr int STATS_TRACE_WAVES = 0, STATS_TRACE_STEPS = 0, STATS_TRACE_MARKS = 12;
#endif
constexpr int STATS_TRACE_BASE = STATS_DIAG_BASE + STATS_DIAG_WORDS;
constexpr int STATS_TRACE_WORDS = STATS_TRACE_WAVES * STATS_TRACE_STEPS * STATS_TRACE_MARKS;
constexpr int STATS_WORDS = STATS_TRACE_BASE + STATS_TRACE_WORDS;

struct RenderParams {
  const float4 *nodes;
  const float4 *prims;
  const float *materials;  // MAT_F floats each: rt_material's 12, then the per-material constants of getIndirectColor (below)
  const float *lights;     // 12 floats each, rt_light order
  const float *f6;         // [n_grids][spp][6]
  int n_lights, n_grids, spp, has_root;
  // camera (rt_camera)
  int W, H, max_depth, pad0;
  float view_w, view_h, view_d, lens;
  float eye[3], u[3], v[3], w[3], ambient[3];
  float half_view_w, half_view_h, half_lens;  // view_w / 2, view_h / 2, lens / 2 (IEEE, host)
  float w_dir[3];                             // (-view_d) * w, the camera's w term of getRay (host, same products)
  // The per-frame constants the step loop uses as VECTOR operands, copied to LDS once per block and
  // read from there (kept in scalar registers they crowd out the loop's own scalar state, which
  // the compiler then parks in vector-register lanes: two VALU moves per use):
  //   [0..2] eye [3..5] u [6..8] v [9..11] w_dir [12] view_w [13] view_h [14] half_view_w
  //   [15] half_view_h [16] lens [17] half_lens [18] W [19] H (as floats) [20..22] ambient
  // The LDS copy continues with [32..43] the first light (single-light builds).
  float hot[24];
  // tiling / partition
  int tile_w, tile_h, tiles_x, tile_samples;
  int rank, nranks, n_local_tiles, compact;
  // outputs
  float *rgb;         // full frame [H*W*3] or compact [n_local_tiles][tile_h][tile_w][3]
  uint32_t *hit_ids;  // may be null
  // per-wave scratch in HBM: the ray queues
  float *scratch;
  unsigned long long scratch_stride;  // floats per wave
  long long *acc;                     // [n_local_tiles * tile_w * tile_h][3] fixed-point pixel sums
  unsigned *tile_counter;             // PACKET_SHARDS counters, 32 words apart
  unsigned long long *stats;          // 12 counters (rt_stats order)
  int stack_depth, trace_zero_weight;
  int n_packets, groups_per_tile;     // packet = group of whole pixels; n_local_tiles * groups_per_tile
  int px_per_packet, chunks_per_packet;  // max(1, 64 / spp) pixels; ceil(spp / 64) chunks of 64 samples
  float clear_band;                      // leaf_box_clear: relative band of a clear "yes" (2^-20)
  float box_absmax;                      // S: the largest |coordinate| of any node box (conservative box tests, rt_kernels.inl)
  int boxes_safe, split_refractive;                  // all node-box coordinates are 0 or in [2^-36, 2^59] (exact-reciprocal path)
  int batch_div, packets_per_shard;                   // guided packet batches: left_in_shard / batch_div, clamped to [1,16]
  // cost-sorted packet order (probe_kernel -> order_kernel)
  unsigned char *pclass;                 // [n_packets] cost class
  unsigned *block_hist;                  // [probe_blocks][4]
  unsigned *class_totals;                // [0..3] packets per class, [4] order entries, [5] refractive entries
  int *order;                            // ticket -> (packet << 7) | code, cost-sorted
  int probe_blocks, probe_span;          // blocks and packets per block (multiple of 256)
  int n_node_recs, n_prim_recs;          // device records (for the LDS scene copy)
  const float4 *wnodes;                  // wide node records (pruned traversal)
  int n_wnode_recs, lds_prim_base;       // lds_prim_base: float4 offset of the primitive records in the LDS scene copy
  const float4 *leafbox;                 // 2 float4 per primitive record: the box of the record's leaf (exact verdicts)
  int lds_leafbox_base, lds_rec_f4;      // float4 offset of the leaf boxes in the LDS copy; float4s of all its records
  int n_materials, batch_cap;            // batch_cap: most tickets per request in the cheap classes
  int refr_cap, refl_cap;                // ticket batch caps in the refractive / reflective cost class
  FastDiv fd_groups_per_tile, fd_tiles_x, fd_spp, fd_tile_w, fd_n_grids;  // the hot loop's divisors
  int point_words;                       // LDS words per wave for the step's shaded points (point_words_for; 0 with one light)
  int rearm_min;                         // shadow-ray pool: idle lanes that trigger a re-arm (RT_REARM_MIN)
  int probe_two, probe_group;            // probe_kernel: two rays per multi-pixel packet (else one); single-pixel packets classified per ray
  float slop_lds, slop_mem;              // relative slop of the conservative box tests: 2^-21 (scenes in LDS), 2^-20 (in memory); RT_DEBUG_BOX_SLOP scales them
  int debug_unsafe_mod, debug_no_range;  // RT_DEBUG_UNSAFE_MOD = k: every k-th ray is sent down the plain-division path; RT_DEBUG_NO_RANGE_PRUNE (tests)
};

struct LaunchCfg {
  int grid, block;
  size_t lds_bytes;
  bool prune, count;
  hipEvent_t ev_k0, ev_k1;  // recorded right before / after render_kernel (may be null)
  bool lds_scene;           // render from the block's LDS copy of the scene
  bool five_waves;          // the 96-VGPR build: LDS scene of <= 32 KB per block, one light, pruned, uncounted
  bool mlight;              // more than one light: the build with the shadow-ray pool (pruned, uncounted launches)
  size_t lds_scene_bytes;   // part of lds_bytes that holds that copy (0 if !lds_scene)
  size_t probe_lds_bytes;   // probe_kernel: the traversal stacks of its own four waves
};

// one per arithmetic mode (separate translation units, different -ffp-contract)
void launch_render_strict(const RenderParams &p, const LaunchCfg &cfg, hipStream_t s);
void launch_render_fma(const RenderParams &p, const LaunchCfg &cfg, hipStream_t s);
void launch_material_prep(const float *materials12, float *materials_dev, int n, hipStream_t s);  // strict arithmetic
int occupancy_strict(const LaunchCfg &cfg);  // resident blocks per CU of the kernel cfg selects
int occupancy_fma(const LaunchCfg &cfg);

inline size_t lds_bytes_for(int stack_depth, int point_words, int waves_per_block = WAVES_PER_BLOCK) {
  return (size_t)waves_per_block * (size_t)(stack_depth * 64 + point_words + 32) * 4;
}
inline unsigned long long scratch_floats_per_wave(int max_depth) {
  int levels = max_depth > 1 ? max_depth - 1 : 0;  // level 1 (primary) is never queued
  return (unsigned long long)levels * QFIELDS * QCAP;
}

}  // namespace rtdev
