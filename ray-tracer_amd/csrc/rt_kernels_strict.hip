// Strict arithmetic build of the trace/shade kernels: compile with
// -ffp-contract=off so no multiply-add is fused (the reference's GHC/SSE code never
// fuses) and hit decisions are bit-identical to the CPU oracle.
#define RT_NS strict
#include "rt_kernels.inl"

namespace rtdev {
void launch_render_strict(const RenderParams &p, const LaunchCfg &cfg, hipStream_t s) {
  strict::launch(p, cfg, s);
}
int occupancy_strict(bool prune, bool count, bool lds_scene, bool mlight, bool five_waves, int block, size_t lds_bytes) {
  return strict::occupancy(prune, count, lds_scene, mlight, five_waves, block, lds_bytes);
}
}  // namespace rtdev
