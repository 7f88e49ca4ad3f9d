// rt_multi.hip -- the N-GPU form of the C ABI (include/rt_mi355x.h, rt_multi_*): ONE host process
// (the Haskell host is one process), one rt_ctx per GPU, the frame's tiles dealt statically to the
// GPUs (tile t -> rank t % N), ONE exchange step per frame: the ranks' compact tile
// buffers gathered on rank 0 with RCCL point-to-point transfers over xGMI (every peer has its own
// link into rank 0: grouped ncclSend/ncclRecv, no ring), then rt_assemble_async on rank 0.
//
// RCCL is bound at run time (dlopen/dlsym in rt_multi_init): librt_mi355x.so has no link-time
// dependency on it, and single-GPU users never load it.  Everything is enqueued from the calling
// thread; the per-GPU work is ordered by the contexts' own streams.

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "rt_mi355x.h"

namespace {

struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;

  bool load(std::string &err) {
    if (handle) return true;
    // a copy the process already holds (a framework's bundled RCCL) first, then ROCm's
    const char *names[] = {"librccl.so", "librccl.so.1"};
    for (const char *n : names)
      if ((handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!handle)
      for (const char *n : names)
        if ((handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!handle) {
      err = std::string("cannot load RCCL: ") + dlerror();
      return false;
    }
#define RT_SYM(field, name)                                              \
  field = reinterpret_cast<decltype(field)>(dlsym(handle, name));         \
  if (!field) {                                                           \
    err = std::string("RCCL symbol missing: ") + name;                    \
    return false;                                                         \
  }
    RT_SYM(CommInitAll, "ncclCommInitAll")
    RT_SYM(CommDestroy, "ncclCommDestroy")
    RT_SYM(GroupStart, "ncclGroupStart")
    RT_SYM(GroupEnd, "ncclGroupEnd")
    RT_SYM(Send, "ncclSend")
    RT_SYM(Recv, "ncclRecv")
    RT_SYM(GetErrorString, "ncclGetErrorString")
#undef RT_SYM
    return true;
  }
};

}  // namespace

struct rt_multi {
  int n = 0;
  std::vector<int> devices;
  std::vector<rt_ctx *> ctx;
  std::vector<hipStream_t> stream;       // the contexts' own streams
  std::vector<ncclComm_t> comm;          // empty when the ranks share a device (test mode, below)
  bool use_rccl = false;
  Rccl rccl;
  std::string err;
  // per-frame buffers (sized on first use / when the geometry changes)
  rt_tiling tiling{};
  size_t stride = 0;                     // elements per rank in the gathered layout
  std::vector<float *> d_tiles;          // rank r's compact tiles (rank 0: a slice of d_gathered)
  std::vector<uint8_t *> d_tiles8;
  float *d_gathered = nullptr, *d_rgb = nullptr;
  uint8_t *d_gathered8 = nullptr, *d_rgb8 = nullptr;
  size_t cap_stride = 0, cap_pixels = 0;
  std::vector<hipEvent_t> done;          // rank r's tiles are in its buffer
  std::vector<hipEvent_t> xchg;          // rank r's side of this rt_multi's newest exchange has been enqueued up to here
  size_t n3_in_flight = 0;               // floats (bytes) of the frame in flight: what rt_multi_wait hands over
  bool have_cam = false, have_samples = false;
  bool in_flight = false;                // a frame has been submitted and not yet waited for
  // pinned staging for the frame on its way to the caller's (pageable) memory: a device-to-host
  // copy into pageable memory would hold rt_multi_submit until the frame is done
  float *h_rgb = nullptr;
  uint8_t *h_rgb8 = nullptr;
  size_t cap_host = 0;                   // pixels
  float *out_rgb = nullptr;              // where rt_multi_wait delivers the frame in flight
  uint8_t *out_rgb8 = nullptr;
  rt_camera cam{};
};

namespace {

// Exchanges of DIFFERENT rt_multi on the same GPUs (K of them keep K frames in flight) run on
// different streams and communicators.  Nothing in RCCL orders two communicators' transfer kernels
// on a device, so two devices could schedule them in opposite orders, each kernel waiting for a
// peer that is queued behind the other one.  Every device therefore runs its exchanges in ONE
// process-wide order, the order of the rt_multi_submit calls: an exchange is enqueued behind an
// event recorded after the previous exchange on that device, whoever owned it, and the whole
// enqueue section is serialised by a mutex (host threads driving different rt_multi).
std::mutex g_exchange_mutex;
struct LastExchange { hipEvent_t ev; const rt_multi *owner; };
std::map<int, LastExchange> g_last_exchange;   // by device id

int mfail(rt_multi *m, int code, const char *fmt, ...) {
  char buf[640];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (m) m->err = buf;
  return code;
}

#define M_HIP(m, expr)                                                                                  \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) return mfail((m), RT_ERR_DEVICE, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                       __FILE__, __LINE__);                                             \
  } while (0)
#define M_NCCL(m, expr)                                                                                 \
  do {                                                                                                  \
    ncclResult_t r_ = (expr);                                                                           \
    if (r_ != ncclSuccess) return mfail((m), RT_ERR_DEVICE, "%s: %s (%s:%d)", #expr,                     \
                                        (m)->rccl.GetErrorString(r_), __FILE__, __LINE__);              \
  } while (0)
#define M_CTX(m, r, expr)                                                                               \
  do {                                                                                                  \
    int rc_ = (expr);                                                                                   \
    if (rc_ != RT_OK) return mfail((m), rc_, "GPU %d: %s", (m)->devices[(r)], rt_last_error((m)->ctx[(r)])); \
  } while (0)

void free_frame_buffers(rt_multi *m) {
  for (int r = 0; r < m->n; ++r) {
    if (!m->ctx[r]) continue;  // (a rank whose rt_init failed: its device id may not even exist)
    (void)hipSetDevice(m->devices[r]);
    if (r > 0 && r < (int)m->d_tiles.size() && m->d_tiles[r]) (void)hipFree(m->d_tiles[r]);
    if (r > 0 && r < (int)m->d_tiles8.size() && m->d_tiles8[r]) (void)hipFree(m->d_tiles8[r]);
  }
  if (m->n && m->ctx[0]) (void)hipSetDevice(m->devices[0]);
  for (void *p : {(void *)m->d_gathered, (void *)m->d_rgb, (void *)m->d_gathered8, (void *)m->d_rgb8})
    if (p) (void)hipFree(p);
  m->d_gathered = m->d_rgb = nullptr;
  m->d_gathered8 = m->d_rgb8 = nullptr;
  if (m->h_rgb) (void)hipHostFree(m->h_rgb);
  if (m->h_rgb8) (void)hipHostFree(m->h_rgb8);
  m->h_rgb = nullptr;
  m->h_rgb8 = nullptr;
  m->cap_host = 0;
  m->d_tiles.assign(m->n, nullptr);
  m->d_tiles8.assign(m->n, nullptr);
  m->cap_stride = m->cap_pixels = 0;
}

// (re)size the gathered / per-rank buffers for the current camera and samples
int ensure_buffers(rt_multi *m) {
  rt_tiling t;
  M_CTX(m, 0, rt_set_partition(m->ctx[0], 0, m->n));
  M_CTX(m, 0, rt_get_tiling(m->ctx[0], &t));
  const size_t tile_px = (size_t)t.tile_w * t.tile_h;
  const size_t stride = (size_t)((t.n_tiles + m->n - 1) / m->n) * tile_px * 3;
  const size_t pixels = (size_t)m->cam.img_w * m->cam.img_h;
  m->tiling = t;
  m->stride = stride;
  if (stride <= m->cap_stride && pixels <= m->cap_pixels) return RT_OK;
  for (int r = 0; r < m->n; ++r) {  // nothing of the old frame may still be in flight
    M_HIP(m, hipSetDevice(m->devices[r]));
    M_HIP(m, hipStreamSynchronize(m->stream[r]));
  }
  free_frame_buffers(m);
  M_HIP(m, hipSetDevice(m->devices[0]));
  M_HIP(m, hipMalloc((void **)&m->d_gathered, std::max<size_t>(1, m->n * stride) * sizeof(float)));
  M_HIP(m, hipMalloc((void **)&m->d_gathered8, std::max<size_t>(1, m->n * stride)));
  M_HIP(m, hipMalloc((void **)&m->d_rgb, std::max<size_t>(1, pixels * 3) * sizeof(float)));
  M_HIP(m, hipMalloc((void **)&m->d_rgb8, std::max<size_t>(1, pixels * 3)));
  M_HIP(m, hipHostMalloc((void **)&m->h_rgb, std::max<size_t>(1, pixels * 3) * sizeof(float), hipHostMallocDefault));
  M_HIP(m, hipHostMalloc((void **)&m->h_rgb8, std::max<size_t>(1, pixels * 3), hipHostMallocDefault));
  m->cap_host = pixels;
  M_HIP(m, hipMemset(m->d_gathered, 0, std::max<size_t>(1, m->n * stride) * sizeof(float)));  // (tile padding is never written)
  m->d_tiles[0] = m->d_gathered;    // rank 0 renders straight into its slice
  m->d_tiles8[0] = m->d_gathered8;
  for (int r = 1; r < m->n; ++r) {
    M_HIP(m, hipSetDevice(m->devices[r]));
    M_HIP(m, hipMalloc((void **)&m->d_tiles[r], std::max<size_t>(1, stride) * sizeof(float)));
    M_HIP(m, hipMalloc((void **)&m->d_tiles8[r], std::max<size_t>(1, stride)));
    M_HIP(m, hipMemset(m->d_tiles[r], 0, std::max<size_t>(1, stride) * sizeof(float)));
  }
  m->cap_stride = stride;
  m->cap_pixels = pixels;
  return RT_OK;
}

}  // namespace

extern "C" {

int rt_multi_init(int32_t n_gpus, const int32_t *devices, rt_multi **out) {
  if (!out) return RT_ERR_INVALID;
  *out = nullptr;
  if (n_gpus < 1 || n_gpus > 64) return RT_ERR_INVALID;
  rt_multi *m = new rt_multi();
  m->n = n_gpus;
  for (int r = 0; r < n_gpus; ++r) m->devices.push_back(devices ? devices[r] : r);
  m->ctx.assign(n_gpus, nullptr);
  m->stream.assign(n_gpus, nullptr);
  m->done.assign(n_gpus, nullptr);
  m->xchg.assign(n_gpus, nullptr);
  m->d_tiles.assign(n_gpus, nullptr);
  m->d_tiles8.assign(n_gpus, nullptr);
  *out = m;  // returned even on failure so that rt_multi_last_error can be read; rt_multi_shutdown frees it
  std::vector<int> sorted = m->devices;
  std::sort(sorted.begin(), sorted.end());
  const bool distinct = std::adjacent_find(sorted.begin(), sorted.end()) == sorted.end();
  for (int r = 0; r < n_gpus; ++r) {
    int rc = rt_init(m->devices[r], &m->ctx[r]);
    if (rc != RT_OK) return mfail(m, rc, "rt_init(device %d) failed", m->devices[r]);
    void *s = nullptr;
    M_CTX(m, r, rt_get_stream(m->ctx[r], &s));
    m->stream[r] = (hipStream_t)s;
    M_HIP(m, hipSetDevice(m->devices[r]));
    M_HIP(m, hipEventCreateWithFlags(&m->done[r], hipEventDisableTiming));
    M_HIP(m, hipEventCreateWithFlags(&m->xchg[r], hipEventDisableTiming));
  }
  // Ranks on distinct GPUs exchange through RCCL.  Ranks that SHARE a device (a rehearsal of the
  // N-rank data path on a smaller box; RCCL refuses duplicate devices) exchange with
  // device-to-device copies instead: same buffers, same layout, same ordering.
  m->use_rccl = distinct;
  if (m->use_rccl) {
    if (!m->rccl.load(m->err)) {
      // a lone GPU has nobody to exchange with: without RCCL it simply assembles its own tiles
      // (with RCCL it sends itself a few bytes so that the binding is exercised on a one-GPU box)
      if (n_gpus > 1) return RT_ERR_DEVICE;
      m->use_rccl = false;
      m->err.clear();
      return RT_OK;
    }
    m->comm.assign(n_gpus, nullptr);
    M_NCCL(m, m->rccl.CommInitAll(m->comm.data(), n_gpus, m->devices.data()));
  }
  return RT_OK;
}

void rt_multi_shutdown(rt_multi *m) {
  if (!m) return;
  for (int r = 0; r < m->n; ++r)
    if (m->ctx[r] && m->stream[r]) {
      (void)hipSetDevice(m->devices[r]);
      (void)hipStreamSynchronize(m->stream[r]);
    }
  {
    std::lock_guard<std::mutex> lock(g_exchange_mutex);   // (the streams are idle: nobody waits for these events any more)
    for (auto it = g_last_exchange.begin(); it != g_last_exchange.end();)
      it = it->second.owner == m ? g_last_exchange.erase(it) : std::next(it);
  }
  for (ncclComm_t c : m->comm)
    if (c) (void)m->rccl.CommDestroy(c);
  free_frame_buffers(m);
  for (int r = 0; r < m->n; ++r) {
    if (!m->ctx[r]) continue;
    (void)hipSetDevice(m->devices[r]);
    if (m->done[r]) (void)hipEventDestroy(m->done[r]);
    if (m->xchg[r]) (void)hipEventDestroy(m->xchg[r]);
    rt_shutdown(m->ctx[r]);
  }
  (void)hipGetLastError();  // leave no stale error behind for the process's next HIP call
  delete m;
}

const char *rt_multi_last_error(rt_multi *m) { return m ? m->err.c_str() : "null rt_multi"; }

// The inputs of a frame in flight are not to be changed under it: rt_multi_wait sizes its copies from
// the camera, and new buffers would free the ones the frame is still using.
#define M_IDLE(m, what)                                                                                   \
  if ((m)->in_flight) return mfail((m), RT_ERR_STATE, what ": a frame is in flight (rt_multi_wait first)")

int rt_multi_upload_scene(rt_multi *m, const rt_scene *scene) {
  if (!m || !scene) return RT_ERR_INVALID;
  M_IDLE(m, "rt_multi_upload_scene");
  for (int r = 0; r < m->n; ++r) M_CTX(m, r, rt_upload_scene(m->ctx[r], scene));  // replicated (KB..MB)
  return RT_OK;
}

int rt_multi_refit(rt_multi *m, const rt_sphere *spheres, int32_t n_spheres, const rt_triangle *triangles,
                   int32_t n_triangles) {
  if (!m) return RT_ERR_INVALID;
  M_IDLE(m, "rt_multi_refit");
  for (int r = 0; r < m->n; ++r)
    M_CTX(m, r, rt_refit_async(m->ctx[r], spheres, n_spheres, triangles, n_triangles, nullptr));
  return RT_OK;
}

int rt_multi_build_scene(rt_multi *m, const rt_sphere *spheres, int32_t n_spheres, const rt_triangle *triangles,
                         int32_t n_triangles, const rt_material *materials, int32_t n_materials, const rt_light *lights,
                         int32_t n_lights) {
  if (!m) return RT_ERR_INVALID;
  M_IDLE(m, "rt_multi_build_scene");
  // every GPU builds the same tree from the same objects (the build is deterministic: sorted unique keys)
  for (int r = 0; r < m->n; ++r)
    M_CTX(m, r, rt_build_scene(m->ctx[r], spheres, n_spheres, triangles, n_triangles, materials, n_materials, lights, n_lights));
  return RT_OK;
}

int rt_multi_set_camera(rt_multi *m, const rt_camera *cam) {
  if (!m || !cam) return RT_ERR_INVALID;
  M_IDLE(m, "rt_multi_set_camera");
  for (int r = 0; r < m->n; ++r) M_CTX(m, r, rt_set_camera(m->ctx[r], cam));
  m->cam = *cam;
  m->have_cam = true;
  return RT_OK;
}

int rt_multi_set_concurrency(rt_multi *m, int32_t n) {
  if (!m) return RT_ERR_INVALID;
  M_IDLE(m, "rt_multi_set_concurrency");
  for (int r = 0; r < m->n; ++r) M_CTX(m, r, rt_set_concurrency(m->ctx[r], n));
  return RT_OK;
}

int rt_multi_set_samples(rt_multi *m, const float *f6, int32_t n_grids, int32_t spp) {
  if (!m || !f6) return RT_ERR_INVALID;
  M_IDLE(m, "rt_multi_set_samples");
  for (int r = 0; r < m->n; ++r) M_CTX(m, r, rt_set_samples(m->ctx[r], f6, n_grids, spp));
  m->have_samples = true;
  return RT_OK;
}

int rt_multi_submit(rt_multi *m, uint32_t flags, float *rgb_out, uint8_t *rgb8_out) {
  if (!m || (!rgb_out && !rgb8_out)) return RT_ERR_INVALID;
  if (m->in_flight) return mfail(m, RT_ERR_STATE, "rt_multi_submit: the previous frame has not been waited for (rt_multi_wait)");
  if (!m->have_cam || !m->have_samples) return mfail(m, RT_ERR_STATE, "camera and samples must be set first");
  int rc = ensure_buffers(m);
  if (rc) return rc;
  const bool want8 = rgb8_out != nullptr && rgb_out == nullptr;  // gather bytes: a quarter of the traffic into rank 0
  const size_t stride = m->stride;
  // 1. every rank renders its tiles (compact layout) on its own stream; with 8-bit output the
  //    quantisation of writePPM runs on the rank, before the exchange
  for (int r = 0; r < m->n; ++r) {
    M_CTX(m, r, rt_set_partition(m->ctx[r], r, m->n));
    M_CTX(m, r, rt_render_tiles_async(m->ctx[r], flags, m->d_tiles[r], nullptr));
    if (want8) M_CTX(m, r, rt_quantize_async(m->ctx[r], m->d_tiles[r], m->d_tiles8[r], stride, nullptr));
  }
  // 2. the one exchange step: gather on rank 0
  if (m->use_rccl) {
    std::lock_guard<std::mutex> lock(g_exchange_mutex);
    // behind the previous exchange on each of the devices, whichever rt_multi enqueued it
    for (int r = 0; r < m->n; ++r) {
      auto it = g_last_exchange.find(m->devices[r]);
      if (it != g_last_exchange.end() && it->second.owner != m) {
        M_HIP(m, hipSetDevice(m->devices[r]));
        M_HIP(m, hipStreamWaitEvent(m->stream[r], it->second.ev, 0));
      }
    }
    M_NCCL(m, m->rccl.GroupStart());
    // (a transfer that fails to enqueue must not leave the group open: close it, then report)
    ncclResult_t bad = ncclSuccess;
    auto queued = [&](ncclResult_t r) {
      if (r != ncclSuccess && bad == ncclSuccess) bad = r;
      return bad == ncclSuccess;
    };
    for (int r = 1; r < m->n && bad == ncclSuccess; ++r) {
      if (want8) {
        if (queued(m->rccl.Send(m->d_tiles8[r], stride, ncclUint8, 0, m->comm[r], m->stream[r])))
          (void)queued(m->rccl.Recv(m->d_gathered8 + r * stride, stride, ncclUint8, r, m->comm[0], m->stream[0]));
      } else {
        if (queued(m->rccl.Send(m->d_tiles[r], stride, ncclFloat32, 0, m->comm[r], m->stream[r])))
          (void)queued(m->rccl.Recv(m->d_gathered + r * stride, stride, ncclFloat32, r, m->comm[0], m->stream[0]));
      }
    }
    if (m->n == 1) {
      // a lone rank still goes through RCCL once (send to self), so that this path -- library
      // binding, communicator, grouped transfers on the render stream -- is exercised on a one-GPU box
      void *buf = want8 ? (void *)m->d_gathered8 : (void *)m->d_gathered;
      const ncclDataType_t ty = want8 ? ncclUint8 : ncclFloat32;
      const size_t head = std::min<size_t>({stride, (size_t)m->cam.img_w * m->cam.img_h * 3, (size_t)256});
      void *tmp = want8 ? (void *)m->d_rgb8 : (void *)m->d_rgb;  // scratch: overwritten by the assembly below
      if (head && queued(m->rccl.Send(buf, head, ty, 0, m->comm[0], m->stream[0])))
        (void)queued(m->rccl.Recv(tmp, head, ty, 0, m->comm[0], m->stream[0]));
    }
    const ncclResult_t closed = m->rccl.GroupEnd();
    if (bad != ncclSuccess)
      return mfail(m, RT_ERR_DEVICE, "RCCL gather (send/recv): %s", m->rccl.GetErrorString(bad));
    M_NCCL(m, closed);
    for (int r = 0; r < m->n; ++r) {
      M_HIP(m, hipSetDevice(m->devices[r]));
      M_HIP(m, hipEventRecord(m->xchg[r], m->stream[r]));
      g_last_exchange[m->devices[r]] = LastExchange{m->xchg[r], m};
    }
  } else {
    for (int r = 1; r < m->n; ++r) {
      M_HIP(m, hipSetDevice(m->devices[r]));
      M_HIP(m, hipEventRecord(m->done[r], m->stream[r]));
      M_HIP(m, hipSetDevice(m->devices[0]));
      M_HIP(m, hipStreamWaitEvent(m->stream[0], m->done[r], 0));
      if (want8)
        M_HIP(m, hipMemcpyAsync(m->d_gathered8 + r * stride, m->d_tiles8[r], stride, hipMemcpyDeviceToDevice, m->stream[0]));
      else
        M_HIP(m, hipMemcpyAsync(m->d_gathered + r * stride, m->d_tiles[r], stride * sizeof(float),
                                hipMemcpyDeviceToDevice, m->stream[0]));
    }
  }
  // 3. rank 0 scatters the tiles into the frame and hands it to the caller
  M_HIP(m, hipSetDevice(m->devices[0]));
  const size_t n3 = (size_t)m->cam.img_w * m->cam.img_h * 3;
  if (want8) {
    M_CTX(m, 0, rt_assemble8_async(m->ctx[0], m->d_gathered8, m->n, stride, m->d_rgb8, nullptr));
    M_HIP(m, hipMemcpyAsync(m->h_rgb8, m->d_rgb8, n3, hipMemcpyDeviceToHost, m->stream[0]));
  } else {
    M_CTX(m, 0, rt_assemble_async(m->ctx[0], m->d_gathered, m->n, stride, m->d_rgb, nullptr));
    M_HIP(m, hipMemcpyAsync(m->h_rgb, m->d_rgb, n3 * sizeof(float), hipMemcpyDeviceToHost, m->stream[0]));
    if (rgb8_out) {
      M_CTX(m, 0, rt_quantize_async(m->ctx[0], m->d_rgb, m->d_rgb8, n3, nullptr));
      M_HIP(m, hipMemcpyAsync(m->h_rgb8, m->d_rgb8, n3, hipMemcpyDeviceToHost, m->stream[0]));
    }
  }
  m->out_rgb = rgb_out;
  m->out_rgb8 = rgb8_out;
  m->n3_in_flight = n3;
  m->in_flight = true;
  return RT_OK;
}

int rt_multi_wait(rt_multi *m, rt_stats *stats) {
  if (!m) return RT_ERR_INVALID;
  if (!m->in_flight) return mfail(m, RT_ERR_STATE, "rt_multi_wait: no frame has been submitted");
  m->in_flight = false;
  // wait for every rank; the statistics are the sums over the ranks (times: the slowest)
  rt_stats total;
  std::memset(&total, 0, sizeof total);
  for (int r = m->n - 1; r >= 0; --r) {
    rt_stats st;
    M_CTX(m, r, rt_collect(m->ctx[r], nullptr, &st));
    total.rays_primary += st.rays_primary;
    total.rays_shadow += st.rays_shadow;
    total.rays_reflection += st.rays_reflection;
    total.rays_refraction += st.rays_refraction;
    total.box_tests += st.box_tests;
    total.sphere_tests += st.sphere_tests;
    total.triangle_tests += st.triangle_tests;
    total.primary_hits += st.primary_hits;
    total.shaded_hits += st.shaded_hits;
    total.wave_ticks_sum += st.wave_ticks_sum;
    total.wave_ticks_max = std::max(total.wave_ticks_max, st.wave_ticks_max);
    total.n_waves += st.n_waves;
    total.ms_kernel = std::max(total.ms_kernel, st.ms_kernel);
    total.ms_launch = std::max(total.ms_launch, st.ms_launch);
  }
  // (rt_collect has waited for rank 0's stream: the frame is in the staging buffers; ms_total,
  //  the blocking single-GPU call's host-side bracket, stays 0 here)
  const size_t n3 = m->n3_in_flight;
  if (m->out_rgb) std::memcpy(m->out_rgb, m->h_rgb, n3 * sizeof(float));
  if (m->out_rgb8) std::memcpy(m->out_rgb8, m->h_rgb8, n3);
  if (stats) *stats = total;
  return RT_OK;
}

int rt_multi_render(rt_multi *m, uint32_t flags, float *rgb_out, uint8_t *rgb8_out, rt_stats *stats) {
  int rc = rt_multi_submit(m, flags, rgb_out, rgb8_out);
  if (rc) return rc;
  return rt_multi_wait(m, stats);
}

}  // extern "C"
