// vmm_probe.hip — build device buffers with a chosen virtual alignment / physical chunk size / mapping order through the
// HIP virtual-memory API, for the placement experiments of tools/experiments/vmm_probe.py (which times the real kernel into them).
// hipcc -shared -fPIC --offload-arch=gfx950 -o tools/bin/libvmmprobe.so tools/experiments/vmm_probe.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

namespace {
struct Buf {
  void* va;
  size_t total, chunk;
  std::vector<hipMemGenericAllocationHandle_t> h;
};
std::map<void*, Buf> g;
}  // namespace

extern "C" {
// bytes rounded up to whole chunks; va_align = alignment of the reservation; va_shift = extra offset of the returned
// pointer inside a larger reservation (to misalign VA against the chunks' physical alignment); order: 0 = chunk i at
// slot i, 1 = reversed, 2 = pseudo-random permutation (seeded)
int vmm_alloc(size_t bytes, size_t chunk, size_t va_align, size_t va_shift, int order, unsigned seed, void** out) {
  int dev = 0;
  hipGetDevice(&dev);
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
  if (chunk % gran) return -10;
  const size_t n = (bytes + chunk - 1) / chunk;
  Buf b;
  b.chunk = chunk;
  b.total = n * chunk + va_shift;
  void* base = nullptr;
  hipError_t e = hipMemAddressReserve(&base, b.total, va_align, nullptr, 0);
  if (e != hipSuccess) return -1;
  b.va = base;
  char* p = static_cast<char*>(base) + va_shift;
  std::vector<size_t> slot(n);
  for (size_t i = 0; i < n; ++i) slot[i] = i;
  if (order == 1)
    for (size_t i = 0; i < n / 2; ++i) std::swap(slot[i], slot[n - 1 - i]);
  if (order == 2) {
    uint64_t s = seed * 0x9E3779B97F4A7C15ULL + 1;
    for (size_t i = n; i > 1; --i) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      std::swap(slot[i - 1], slot[s % i]);
    }
  }
  for (size_t i = 0; i < n; ++i) {
    hipMemGenericAllocationHandle_t h;
    e = hipMemCreate(&h, chunk, &prop, 0);
    if (e != hipSuccess) return -2;
    e = hipMemMap(p + slot[i] * chunk, chunk, 0, h, 0);
    if (e != hipSuccess) return -3;
    b.h.push_back(h);
  }
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  e = hipMemSetAccess(p, n * chunk, &acc, 1);
  if (e != hipSuccess) return -4;
  g[p] = b;
  *out = p;
  return 0;
}
void vmm_free(void* p) {
  auto it = g.find(p);
  if (it == g.end()) return;
  hipDeviceSynchronize();
  Buf& b = it->second;
  for (size_t i = 0; i < b.h.size(); ++i) (void)hipMemUnmap(static_cast<char*>(p) + i * b.chunk, b.chunk);
  for (auto h : b.h) (void)hipMemRelease(h);
  hipMemAddressFree(b.va, b.total);
  g.erase(it);
}
size_t vmm_granularity(int recommended) {
  int dev = 0;
  hipGetDevice(&dev);
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  hipMemGetAllocationGranularity(&gran, &prop, recommended ? hipMemAllocationGranularityRecommended : hipMemAllocationGranularityMinimum);
  return gran;
}
}
