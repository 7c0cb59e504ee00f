// pmx_alloc.cpp — placing the prediction matrix where the row-strided write stream runs fastest.
//
// On MI355X the rate of the prediction stream depends on WHERE in device memory the matrix sits (the same kernel:
// 0.81-0.90 ms in some 5.6 GB windows, 1.04-1.12 ms in most; linear fills do not care; tools/experiments/store_pattern_probe.hip
// `arena`).  Nothing in the HIP API says which memory is the fast kind, so this helper measures: it maps
// an arena out of separately allocated physical chunks (HIP virtual-memory API) window by window, times the real kernel
// into each, stops inside the first plateau of the fast kind (or at the arena's size limit, with the best window seen),
// keeps the chunks under the chosen window and gives every other chunk back.  A plain allocation is a candidate too
// (timed first); when it beats the chosen window it is what the caller gets.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "pmx.h"

namespace {

struct Arena {
  int dev = 0;            // the device the buffer lives on (destroy switches to it: VMM calls and the sync act on the CURRENT device)
  void* plain = nullptr;  // the buffer is a plain hipMalloc allocation (it beat every window): nothing else is set
  void* va = nullptr;
  size_t total = 0, chunk = 0;
  std::vector<hipMemGenericAllocationHandle_t> handles;  // one per chunk
  std::vector<char> mapped;                              // chunk still mapped?
};
std::mutex g_mu;
std::map<void*, Arena> g_arenas;  // by the pointer handed to the caller

void release(Arena& a) {
  for (size_t i = 0; i < a.handles.size(); ++i) {
    if (a.mapped[i]) (void)hipMemUnmap(static_cast<char*>(a.va) + i * a.chunk, a.chunk);
    (void)hipMemRelease(a.handles[i]);
  }
  if (a.va) (void)hipMemAddressFree(a.va, a.total);
  a = Arena{};
}

}  // namespace

extern "C" {

int32_t pmx_prediction_buffer_create_pitched(const pmx_model* model, const pmx_population* pop, const double* d_theta,
                                     int64_t n_support, int64_t ld, int64_t search_bytes, void* stream, double** d_pred,
                                     double* ms_per_pass) {
  if (!model || !pop || !d_theta || !d_pred || n_support <= 0 || ld < n_support) return PMX_ERR_INVALID_ARGUMENT;
  *d_pred = nullptr;
  const size_t need = static_cast<size_t>(pmx_population_n_observations(pop)) * static_cast<size_t>(ld) * sizeof(double);
  if (need == 0) return PMX_ERR_INVALID_ARGUMENT;
  // the arena lives on the population's device, whatever the calling thread's current device is (restored on return)
  const int dev = pmx_population_device(pop);
  int prev = 0;
  if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(dev) != hipSuccess) return PMX_ERR_NO_DEVICE;
  struct Restore {
    int d;
    ~Restore() { (void)hipSetDevice(d); }
  } restore{prev};
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  // physical chunks of 1/8 of the matrix (at least 2 MiB, a multiple of 2 MiB): windows start on chunk boundaries
  size_t chunk = ((need / 8 + (2u << 20) - 1) / (2u << 20)) * (2u << 20);
  if (chunk < (2u << 20)) chunk = 2u << 20;
  const size_t win_chunks = (need + chunk - 1) / chunk;
  const bool exhaustive = search_bytes < 0;  // -bytes: time every window of the arena, keep the best
  if (exhaustive) search_bytes = -search_bytes;
  size_t n_chunks = search_bytes > 0 ? static_cast<size_t>(search_bytes) / chunk : 0;
  if (n_chunks < win_chunks) n_chunks = win_chunks;
  Arena a;
  a.dev = dev;
  a.chunk = chunk;
  a.total = n_chunks * chunk;
  if (hipMemAddressReserve(&a.va, a.total, 0, nullptr, 0) != hipSuccess) return PMX_ERR_OUT_OF_MEMORY;
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  // chunks are created and mapped as the search reaches them: a search that finds a fast plateau early never touches
  // the rest of the arena
  bool exhausted = false;
  auto map_up_to = [&](size_t want) {  // chunks [0, want) mapped; false when the device gave out first
    while (!exhausted && a.handles.size() < want && a.handles.size() < n_chunks) {
      hipMemGenericAllocationHandle_t h;
      char* at = static_cast<char*>(a.va) + a.handles.size() * chunk;
      if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) {
        exhausted = true;
        break;
      }
      if (hipMemMap(at, chunk, 0, h, 0) != hipSuccess) {
        (void)hipMemRelease(h);
        exhausted = true;
        break;
      }
      if (hipMemSetAccess(at, chunk, &acc, 1) != hipSuccess) {
        (void)hipMemUnmap(at, chunk);
        (void)hipMemRelease(h);
        exhausted = true;
        break;
      }
      a.handles.push_back(h);
      a.mapped.push_back(1);
    }
    return a.handles.size() >= want;
  };
  if (!map_up_to(win_chunks)) {
    release(a);
    return PMX_ERR_OUT_OF_MEMORY;
  }
  // clocks up, then a plain allocation as the first candidate (boxes exist whose arenas hold no fast window while an
  // ordinary allocation runs at the medium speed), then the kernel into window after window
  double ms = 0.0;
  double* w0 = static_cast<double*>(a.va);
  int32_t rc = pmx_time_predict_device(model, pop, d_theta, n_support, w0, ld, 30, stream, &ms);
  void* plain = nullptr;
  double plain_ms = 1e300;
  // (only where placement matters at all: a pass that moves less than ~1.5 TB/s of predictions is not write-bound)
  const bool plain_matters = rc == PMX_OK && ms > 0.0 && static_cast<double>(need) / (ms * 1.0e-3) > 1.5e12;
  if (plain_matters && search_bytes > 0 && hipMalloc(&plain, need) == hipSuccess) {
    if (pmx_time_predict_device(model, pop, d_theta, n_support, static_cast<double*>(plain), ld, 6, stream, &plain_ms) != PMX_OK) {
      (void)hipFree(plain);
      plain = nullptr;
      plain_ms = 1e300;
    }
  } else {
    plain = nullptr;
    (void)hipGetLastError();
  }
  std::vector<double> t;
  // a pass that moves less than ~1.5 TB/s of predictions is not write-bound: where the matrix sits is moot, take window 0
  const bool write_bound = rc == PMX_OK && ms > 0.0 && static_cast<double>(need) / (ms * 1.0e-3) > 1.5e12;
  const bool debug = std::getenv("PMX_DEBUG_PLACEMENT") != nullptr;
  // PMX_TUNE_PLACE_WINDOW=i: take window i without searching (profiling runs: the landscape of an arena repeats from
  // process to process on one box, so a traced run can sit where an untraced one found the best window and the trace
  // holds no search dispatches).  PMX_TUNE_PLACE_FULL=1: time every window of the arena (landscape studies).
  long forced = -1;
  if (const char* e = std::getenv("PMX_TUNE_PLACE_WINDOW")) forced = std::atol(e);
  if (forced >= 0 && static_cast<size_t>(forced) + win_chunks > n_chunks) forced = -1;  // (past the arena: search instead of mapping all of it first)
  if (forced >= 0 && !map_up_to(static_cast<size_t>(forced) + win_chunks)) forced = -1;
  const bool full = exhaustive || std::getenv("PMX_TUNE_PLACE_FULL") != nullptr;
  size_t best = 0;
  double best_ms = 1e300;
  // fast memory comes in plateaus several windows wide: a window is scored by the worst of itself and its two
  // neighbours, i.e. windows inside a plateau win over windows on its edge
  auto score_of = [&](size_t i) {
    double sc = t[i];
    if (i > 0 && t[i - 1] > sc) sc = t[i - 1];
    if (i + 1 < t.size() && t[i + 1] > sc) sc = t[i + 1];
    return sc;
  };
  if (forced >= 0 && rc == PMX_OK) {
    best = static_cast<size_t>(forced);
    double* w = reinterpret_cast<double*>(static_cast<char*>(a.va) + best * chunk);
    rc = pmx_time_predict_device(model, pop, d_theta, n_support, w, ld, 4, stream, &ms);
    t.assign(best + 1, ms);
  } else if (!write_bound) {
    if (rc == PMX_OK) t.push_back(ms);
  } else {
    double slowest = 0.0;
    long stopped_at = -1;
    for (size_t i = 0; rc == PMX_OK && map_up_to(i + win_chunks); ++i) {
      double* w = reinterpret_cast<double*>(static_cast<char*>(a.va) + i * chunk);
      rc = pmx_time_predict_device(model, pop, d_theta, n_support, w, ld, 4, stream, &ms);
      if (debug) std::fprintf(stderr, "[pmx] window at chunk %zu (%.2f GiB): %.4f ms\n", i, i * chunk / 1073741824.0, ms);
      t.push_back(ms);
      if (ms > slowest) slowest = ms;
      // stop inside the first plateau that is clearly the fast kind: its middle window and both neighbours write at
      // >= 6.4 TB/s, or 15 % faster than the slowest window met so far (the two kinds differ by 20-25 %)
      if (!full && i >= 2) {
        const double sc = score_of(i - 1);
        if (static_cast<double>(need) / (sc * 1.0e-3) >= 6.4e12 || sc < 0.85 * slowest) {
          stopped_at = static_cast<long>(i) - 1;  // the plateau's middle window, not its last-timed edge
          break;
        }
      }
    }
    if (stopped_at >= 0) {
      best = static_cast<size_t>(stopped_at);
      best_ms = t[best];
    }
    for (size_t i = 0; stopped_at < 0 && i < t.size(); ++i) {
      const double sc = t.size() >= 3 ? score_of(i) : t[i];
      if (sc < best_ms) {
        best_ms = sc;
        best = i;
      }
    }
  }
  if (debug) std::fprintf(stderr, "[pmx] chosen window %zu\n", best);
  if (!t.empty()) best_ms = t[best];
  if (rc != PMX_OK) {
    if (plain) (void)hipFree(plain);
    release(a);
    return rc;
  }
  if (debug && plain) std::fprintf(stderr, "[pmx] plain allocation: %.4f ms\n", plain_ms);
  if (plain && forced < 0 && plain_ms < 0.99 * best_ms) {  // the plain allocation wins: the whole arena goes back
    release(a);
    Arena p;
    p.dev = dev;
    p.plain = plain;
    {
      std::lock_guard<std::mutex> lock(g_mu);
      g_arenas[plain] = std::move(p);
    }
    *d_pred = static_cast<double*>(plain);
    if (ms_per_pass) *ms_per_pass = plain_ms;
    return PMX_OK;
  }
  if (plain) (void)hipFree(plain);
  // give back every chunk outside the chosen window
  for (size_t i = 0; i < a.handles.size(); ++i) {
    if (i >= best && i < best + win_chunks) continue;
    (void)hipMemUnmap(static_cast<char*>(a.va) + i * chunk, chunk);
    a.mapped[i] = 0;
    (void)hipMemRelease(a.handles[i]);
  }
  {  // what stays: the reservation and the window's chunks (released handles are marked unmapped)
    double* out = reinterpret_cast<double*>(static_cast<char*>(a.va) + best * chunk);
    std::lock_guard<std::mutex> lock(g_mu);
    g_arenas[out] = std::move(a);
    *d_pred = out;
  }
  if (ms_per_pass) *ms_per_pass = best_ms;
  return PMX_OK;
}

int32_t pmx_prediction_buffer_create(const pmx_model* model, const pmx_population* pop, const double* d_theta,
                                     int64_t n_support, int64_t search_bytes, void* stream, double** d_pred,
                                     double* ms_per_pass) {
  return pmx_prediction_buffer_create_pitched(model, pop, d_theta, n_support, n_support, search_bytes, stream, d_pred, ms_per_pass);
}

void pmx_prediction_buffer_destroy(double* d_pred) {
  if (!d_pred) return;
  Arena a;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_arenas.find(d_pred);
    if (it == g_arenas.end()) return;
    a = std::move(it->second);
    g_arenas.erase(it);
  }
  // on the buffer's own device: kernels still writing the window must be waited for THERE before the memory goes away,
  // and hipMemUnmap / hipMemRelease / hipFree act on the current device (a multi-GPU process calls this from anywhere)
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != a.dev) (void)hipSetDevice(a.dev);
  struct Restore {
    int from, to;
    ~Restore() {
      if (to >= 0 && to != from) (void)hipSetDevice(to);
    }
  } restore{a.dev, prev};
  (void)hipDeviceSynchronize();
  if (a.plain) {
    (void)hipFree(a.plain);
    return;
  }
  for (size_t i = 0; i < a.handles.size(); ++i) {
    if (!a.mapped[i]) continue;
    (void)hipMemUnmap(static_cast<char*>(a.va) + i * a.chunk, a.chunk);
    (void)hipMemRelease(a.handles[i]);
  }
  if (a.va) (void)hipMemAddressFree(a.va, a.total);
}

}  // extern "C"
