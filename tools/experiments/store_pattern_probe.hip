// store_pattern_probe.hip — which geometry of the prediction write stream does MI355X like?
// Stand-alone micro-benchmark (not part of the library): writes the C3 prediction matrix
// (700 000 rows x 1000 doubles = 5.6 GB) with different wave->address maps and prints TB/s for each.
//   build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/store_probe tools/experiments/store_pattern_probe.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));             \
      std::exit(1);                                                            \
    }                                                                          \
  } while (0)

typedef double dbl2 __attribute__((ext_vector_type(2)));

constexpr int64_t S = 100000, O = 7, P = 1000, G = 8;
constexpr int64_t ROWS = S * O;

template <bool NT>
__device__ __forceinline__ void st16(double* p, double a, double b) {
  dbl2 v;
  v.x = a;
  v.y = b;
  if (NT)  // "nt" rows below = cache policy sc1 nt (what the library's kernel uses; plain nt is in the P rows)
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
  else
    *reinterpret_cast<dbl2*>(p) = v;
}

// the classed map with explicit cache-policy bits on the store (inline asm): 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt,
// 5 sc0 sc1 nt, 6 sc0
template <int POLICY>
__device__ __forceinline__ void st16_policy(double* p, double a, double b) {
  dbl2 v;
  v.x = a;
  v.y = b;
  if constexpr (POLICY == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  if constexpr (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if constexpr (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  if constexpr (POLICY == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
  if constexpr (POLICY == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
  if constexpr (POLICY == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
}
template <int POLICY>
__global__ __launch_bounds__(256) void k_classed_policy(double* out, int64_t n_chunks, int cpb, int n_ptiles) {
  const int64_t b = blockIdx.x;
  const int64_t group = b / (8 * n_ptiles);
  const int local = static_cast<int>(b % (8 * n_ptiles));
  const int ptile = local / 8;
  const int64_t cblock = group * 8 + (local % 8);
  const unsigned lane = threadIdx.x & 63u;
  const bool upper = lane >= 32u;
  const int64_t p_even = static_cast<int64_t>(ptile) * 256 + (threadIdx.x & ~63u) + 2u * (lane & 31u);
  const bool ok = p_even + 1 < P;
  for (int64_t c = cblock * cpb; c < (cblock + 1) * cpb && c < n_chunks; ++c) {
    for (int k = 0; k < O; ++k) {
#pragma unroll
      for (int h = 0; h < G / 2; ++h) {
        const int64_t subj = c * G + 2 * h + (upper ? 1 : 0);
        const int64_t row = subj * O + k;
        if (ok) st16_policy<POLICY>(out + row * P + p_even, 1.0, 2.0);
      }
    }
  }
}

// A: linear fill, 16 B per lane, grid-stride
template <bool NT>
__global__ __launch_bounds__(256) void k_linear(double* out, int64_t n2) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n2; i += gridDim.x * 256ll) st16<NT>(out + 2 * i, 1.0, 2.0);
}

// A2: one 16 B store per lane, no loop, huge grid
template <bool NT>
__global__ __launch_bounds__(256) void k_linear1(double* out, int64_t n2) {
  const int64_t i = blockIdx.x * 256ll + threadIdx.x;
  if (i < n2) st16<NT>(out + 2 * i, 1.0, 2.0);
}
// A3: 4 x 16 B per lane, each wave-store 1 KB contiguous, block writes 16 KB contiguous, no loop
template <bool NT>
__global__ __launch_bounds__(256) void k_linear4(double* out, int64_t n2) {
  const int64_t base = blockIdx.x * 1024ll;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t i = base + j * 256 + threadIdx.x;
    if (i < n2) st16<NT>(out + 2 * i, 1.0, 2.0);
  }
}
// A4: 32 B per lane adjacent (two dwordx4 back to back per lane, the elementwise-kernel shape)
template <bool NT>
__global__ __launch_bounds__(256) void k_linear32(double* out, int64_t n2) {
  const int64_t i = (blockIdx.x * 256ll + threadIdx.x) * 2;
  if (i + 1 < n2) {
    st16<NT>(out + 2 * i, 1.0, 2.0);
    st16<NT>(out + 2 * i + 2, 1.0, 2.0);
  }
}

// B: the classed kernel's map.  block -> (chunk-block, ptile of 256 p); per chunk: 7 obs x 4 member pairs;
// lanes 0-31 write member 2h's row, lanes 32-63 member 2h+1's row, 16 B per lane (two adjacent p).
// ROWMAJOR_SUBJ: row = subject*7 + obs (the ABI's order).  else obs-major: row = obs*S + subject.
template <bool NT, bool OBS_MAJOR, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_classed(double* out, int64_t n_chunks, int cpb, int n_ptiles, int delay) {
  const int64_t b = blockIdx.x;
  const int64_t group = b / (8 * n_ptiles);
  const int local = static_cast<int>(b % (8 * n_ptiles));
  const int ptile = local / 8;
  const int64_t cblock = group * 8 + (local % 8);
  const unsigned lane = threadIdx.x & 63u;
  const bool upper = lane >= 32u;
  const int64_t p_even = static_cast<int64_t>(ptile) * BLOCK + (threadIdx.x & ~63u) + 2u * (lane & 31u);
  const bool ok = p_even + 1 < P;
  double acc = static_cast<double>(threadIdx.x);
  for (int64_t c = cblock * cpb; c < (cblock + 1) * cpb && c < n_chunks; ++c) {
    for (int k = 0; k < O; ++k) {
      for (int d = 0; d < delay; ++d) acc = fma(acc, 1.0000001, 0.5);  // stand-in for the propagator math
#pragma unroll
      for (int h = 0; h < G / 2; ++h) {
        const int64_t subj = c * G + 2 * h + (upper ? 1 : 0);
        const int64_t row = OBS_MAJOR ? (k * S + subj) : (subj * O + k);
        if (ok) st16<NT>(out + row * P + p_even, acc, acc);
      }
    }
  }
}

// F/G: subject-major rows (the ABI's order) but KB rows of a member are written back to back
// (KB = 2: pairs of observations; KB = 7: a member's whole block of rows) — what staging predictions of
// several steps before storing them would produce.
template <bool NT, int KB>
__global__ __launch_bounds__(256) void k_classed_kb(double* out, int64_t n_chunks, int cpb, int n_ptiles) {
  const int64_t b = blockIdx.x;
  const int64_t group = b / (8 * n_ptiles);
  const int local = static_cast<int>(b % (8 * n_ptiles));
  const int ptile = local / 8;
  const int64_t cblock = group * 8 + (local % 8);
  const unsigned lane = threadIdx.x & 63u;
  const bool upper = lane >= 32u;
  const int64_t p_even = static_cast<int64_t>(ptile) * 256 + (threadIdx.x & ~63u) + 2u * (lane & 31u);
  const bool ok = p_even + 1 < P;
  for (int64_t c = cblock * cpb; c < (cblock + 1) * cpb && c < n_chunks; ++c) {
    for (int k0 = 0; k0 < O; k0 += KB) {
#pragma unroll
      for (int h = 0; h < G / 2; ++h) {
        const int64_t subj = c * G + 2 * h + (upper ? 1 : 0);
        for (int k = k0; k < k0 + KB && k < O; ++k) {
          const int64_t row = subj * O + k;
          if (ok) st16<NT>(out + row * P + p_even, 1.0, 2.0);
        }
      }
    }
  }
}

// H: the classed map with chunk members far apart (subject = j * S/G + c) instead of consecutive
template <bool NT>
__global__ __launch_bounds__(256) void k_classed_spread(double* out, int64_t n_chunks, int cpb, int n_ptiles) {
  const int64_t b = blockIdx.x;
  const int64_t group = b / (8 * n_ptiles);
  const int local = static_cast<int>(b % (8 * n_ptiles));
  const int ptile = local / 8;
  const int64_t cblock = group * 8 + (local % 8);
  const unsigned lane = threadIdx.x & 63u;
  const bool upper = lane >= 32u;
  const int64_t p_even = static_cast<int64_t>(ptile) * 256 + (threadIdx.x & ~63u) + 2u * (lane & 31u);
  const bool ok = p_even + 1 < P;
  for (int64_t c = cblock * cpb; c < (cblock + 1) * cpb && c < n_chunks; ++c) {
    for (int k = 0; k < O; ++k) {
#pragma unroll
      for (int h = 0; h < G / 2; ++h) {
        const int64_t subj = (2 * h + (upper ? 1 : 0)) * (S / G) + c;
        const int64_t row = subj * O + k;
        if (ok) st16<NT>(out + row * P + p_even, 1.0, 2.0);
      }
    }
  }
}

// S: member stride study.  Chunk c of a group of (stride*G) subjects: member j = base + j*stride
// (stride 1 = B consecutive members, stride n_chunks = H).
template <bool NT>
__global__ __launch_bounds__(256) void k_classed_stride(double* out, int64_t n_chunks, int cpb, int n_ptiles, int64_t stride) {
  const int64_t b = blockIdx.x;
  const int64_t group = b / (8 * n_ptiles);
  const int local = static_cast<int>(b % (8 * n_ptiles));
  const int ptile = local / 8;
  const int64_t cblock = group * 8 + (local % 8);
  const unsigned lane = threadIdx.x & 63u;
  const bool upper = lane >= 32u;
  const int64_t p_even = static_cast<int64_t>(ptile) * 256 + (threadIdx.x & ~63u) + 2u * (lane & 31u);
  const bool ok = p_even + 1 < P;
  for (int64_t c = cblock * cpb; c < (cblock + 1) * cpb && c < n_chunks; ++c) {
    const int64_t base = (c / stride) * (stride * G) + (c % stride);
    for (int k = 0; k < O; ++k) {
#pragma unroll
      for (int h = 0; h < G / 2; ++h) {
        const int64_t subj = base + (2 * h + (upper ? 1 : 0)) * stride;
        const int64_t row = subj * O + k;
        if (ok && subj < S) st16<NT>(out + row * P + p_even, 1.0, 2.0);
      }
    }
  }
}

// E: one wave-store = 1 KB of ONE row (64 lanes x 16 B = 128 adjacent p): what an LDS transpose would allow.
// block of 256 threads covers 512 p of a row -> 2 ptiles for P=1000; members are walked one row at a time.
template <bool NT>
__global__ __launch_bounds__(256) void k_rowwide(double* out, int64_t n_chunks, int cpb, int n_ptiles) {
  const int64_t b = blockIdx.x;
  const int64_t group = b / (8 * n_ptiles);
  const int local = static_cast<int>(b % (8 * n_ptiles));
  const int ptile = local / 8;
  const int64_t cblock = group * 8 + (local % 8);
  const int64_t p_even = static_cast<int64_t>(ptile) * 512 + 2 * threadIdx.x;
  const bool ok = p_even + 1 < P;
  for (int64_t c = cblock * cpb; c < (cblock + 1) * cpb && c < n_chunks; ++c) {
    for (int k = 0; k < O; ++k) {
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int64_t row = (c * G + j) * O + k;
        if (ok) st16<NT>(out + row * P + p_even, 1.0, 2.0);
      }
    }
  }
}

template <typename F>
static void run(const char* name, F launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  const int reps = 10;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  std::printf("%-44s %7.4f ms  %6.3f TB/s\n", name, ms, ROWS * P * 8.0 / (ms * 1e-3) / 1e12);
  std::fflush(stdout);
}

// Does the rate of a scattered pattern depend on WHICH allocation it writes?  (argv[1] = "alloc")
static int alloc_study() {
  const int64_t n_chunks = S / G;
  const int cpb = 6;
  const int64_t cb = ((n_chunks + cpb - 1) / cpb + 7) / 8 * 8;
  void* pad[8] = {};
  for (int trial = 0; trial < 8; ++trial) {
    if (trial % 2 == 1) CK(hipMalloc(&pad[trial], (64ull + 37ull * trial) << 20));  // perturb the allocator between trials
    double* out = nullptr;
    CK(hipMalloc(&out, ROWS * P * 8));
    CK(hipMemset(out, 0, ROWS * P * 8));
    std::printf("allocation %d at %p\n", trial, static_cast<void*>(out));
    run("  B classed map (sc1 nt)", [&] { hipLaunchKernelGGL((k_classed<true, false, 256>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, cpb, 4, 0); });
    run("  H members spread (sc1 nt)", [&] { hipLaunchKernelGGL((k_classed_spread<true>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, cpb, 4); });
    run("  D obs-major (sc1 nt)", [&] { hipLaunchKernelGGL((k_classed<true, true, 256>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, cpb, 4, 0); });
    CK(hipFree(out));
  }
  for (void* p : pad)
    if (p) CK(hipFree(p));
  return 0;
}

static int stride_study() {
  const int64_t n_chunks = S / G;
  void* pad = nullptr;
  double* out = nullptr;
  for (int trial = 0; trial < 3; ++trial) {  // a few allocations: the effect is allocation dependent
    if (trial) CK(hipMalloc(&pad, 97u << 20));
    CK(hipMalloc(&out, ROWS * P * 8));
    CK(hipMemset(out, 0, ROWS * P * 8));
    std::printf("allocation %d\n", trial);
    for (int cpb : {1, 6}) {
      const int64_t cb = ((n_chunks + cpb - 1) / cpb + 7) / 8 * 8;
      for (int64_t stride : {1ll, 2ll, 5ll, 20ll, 100ll, 500ll, 2500ll, 12500ll}) {
        char nm[96];
        std::snprintf(nm, sizeof nm, "  member stride %lld cpb=%d", static_cast<long long>(stride), cpb);
        run(nm, [&] { hipLaunchKernelGGL((k_classed_stride<true>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, cpb, 4, stride); });
      }
    }
    CK(hipFree(out));
    if (pad) CK(hipFree(pad));
    pad = nullptr;
  }
  return 0;
}

// Which kind of allocation does the spread pattern like?  (argv[1] = "alloc3")
static int alloc3_study() {
  const int64_t n_chunks = S / G;
  const size_t big = ROWS * P * 8;
  const int64_t cb = (n_chunks + 7) / 8 * 8;
  auto measure = [&](const char* what, double* out) {
    CK(hipMemset(out, 0, big));
    run(what, [&] { hipLaunchKernelGGL((k_classed_stride<true>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, 1, 4, n_chunks); });
  };
  for (int rep = 0; rep < 4; ++rep) {
    double* a = nullptr;
    void* pad = nullptr;
    if (rep % 2) CK(hipMalloc(&pad, 97u << 20));
    CK(hipMalloc(&a, big)); measure("hipMalloc", a); CK(hipFree(a));
    CK(hipMalloc(&a, 6ull << 30)); measure("hipMalloc 6 GiB", a); CK(hipFree(a));
    CK(hipMalloc(&a, 8ull << 30)); measure("hipMalloc 8 GiB", a); CK(hipFree(a));
    if (hipExtMallocWithFlags(reinterpret_cast<void**>(&a), big, hipDeviceMallocUncached) == hipSuccess) { measure("hipExtMallocWithFlags uncached", a); CK(hipFree(a)); }
    if (hipExtMallocWithFlags(reinterpret_cast<void**>(&a), big, hipDeviceMallocFinegrained) == hipSuccess) { measure("hipExtMallocWithFlags finegrained", a); CK(hipFree(a)); }
    {
      hipMemPool_t pool; hipStream_t st; CK(hipStreamCreate(&st));
      CK(hipDeviceGetDefaultMemPool(&pool, 0));
      if (hipMallocAsync(reinterpret_cast<void**>(&a), big, st) == hipSuccess) { CK(hipStreamSynchronize(st)); measure("hipMallocAsync (pool)", a); CK(hipFreeAsync(a, st)); CK(hipStreamSynchronize(st)); }
      CK(hipStreamDestroy(st));
    }
    if (pad) CK(hipFree(pad));
    std::printf("--\n");
  }
  return 0;
}

// Virtual-memory API: the matrix mapped from physical handles of a chosen size (argv[1] = "vmm")
static int vmm_study() {
  const int64_t n_chunks = S / G;
  const size_t big = ROWS * P * 8;
  const int64_t cb = (n_chunks + 7) / 8 * 8;
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  std::printf("recommended granularity %zu\n", gran);
  for (int rep = 0; rep < 3; ++rep) {
    for (size_t chunk : {size_t(2) << 20, size_t(64) << 20, size_t(1) << 30, size_t(6) << 30}) {
      const size_t total = (big + chunk - 1) / chunk * chunk;
      void* va = nullptr;
      if (hipMemAddressReserve(&va, total, chunk < (size_t(1) << 30) ? 0 : 0, nullptr, 0) != hipSuccess) { std::printf("reserve failed\n"); continue; }
      std::vector<hipMemGenericAllocationHandle_t> hs;
      bool ok = true;
      for (size_t off = 0; off < total && ok; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { ok = false; break; }
        hs.push_back(h);
        if (hipMemMap(static_cast<char*>(va) + off, chunk, 0, h, 0) != hipSuccess) ok = false;
      }
      hipMemAccessDesc acc{};
      acc.location = prop.location;
      acc.flags = hipMemAccessFlagsProtReadWrite;
      if (ok && hipMemSetAccess(va, total, &acc, 1) != hipSuccess) ok = false;
      if (ok) {
        double* out = static_cast<double*>(va);
        CK(hipMemset(out, 0, big));
        char nm[96];
        std::snprintf(nm, sizeof nm, "VMM, %zu MiB physical handles", chunk >> 20);
        run(nm, [&] { hipLaunchKernelGGL((k_classed_stride<true>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, 1, 4, n_chunks); });
      } else {
        std::printf("VMM with %zu MiB handles: failed\n", chunk >> 20);
      }
      (void)hipMemUnmap(va, total);
      for (auto h : hs) (void)hipMemRelease(h);
      (void)hipMemAddressFree(va, total);
    }
    double* a = nullptr;
    CK(hipMalloc(&a, big));
    CK(hipMemset(a, 0, big));
    run("hipMalloc", [&] { hipLaunchKernelGGL((k_classed_stride<true>), dim3(cb * 4), dim3(256), 0, 0, a, n_chunks, 1, 4, n_chunks); });
    CK(hipFree(a));
    std::printf("--\n");
  }
  return 0;
}

// How many write fronts?  Spread members, one chunk per block, GG members per chunk (argv[1] = "fronts")
template <int GG>
__global__ __launch_bounds__(256) void k_fronts(double* out, int64_t n_chunks, int n_ptiles) {
  const int64_t b = blockIdx.x;
  const int64_t group = b / (8 * n_ptiles);
  const int local = static_cast<int>(b % (8 * n_ptiles));
  const int ptile = local / 8;
  const int64_t c = group * 8 + (local % 8);
  if (c >= n_chunks) return;
  const unsigned lane = threadIdx.x & 63u;
  const bool upper = lane >= 32u;
  const int64_t p_even = static_cast<int64_t>(ptile) * 256 + (threadIdx.x & ~63u) + 2u * (lane & 31u);
  const bool ok = p_even + 1 < P;
  for (int k = 0; k < O; ++k) {
#pragma unroll
    for (int h = 0; h < (GG + 1) / 2; ++h) {
      const int j = 2 * h + (upper ? 1 : 0);
      const int64_t subj = c + static_cast<int64_t>(j) * n_chunks;
      const int64_t row = subj * O + k;
      if (ok && j < GG && subj < S) st16<true>(out + row * P + p_even, 1.0, 2.0);
    }
  }
}

static int fronts_study() {
  const size_t big = ROWS * P * 8;
  void* pad = nullptr;
  for (int trial = 0; trial < 4; ++trial) {
    if (trial % 2) CK(hipMalloc(&pad, 97u << 20));
    double* out = nullptr;
    CK(hipMalloc(&out, big));
    CK(hipMemset(out, 0, big));
    std::printf("allocation %d\n", trial);
#define FRONTS(GG)                                                                                              \
    {                                                                                                            \
      const int64_t nc = (S + GG - 1) / GG;                                                                      \
      const int64_t cb = (nc + 7) / 8 * 8;                                                                       \
      run("  " #GG " fronts", [&] { hipLaunchKernelGGL((k_fronts<GG>), dim3(cb * 4), dim3(256), 0, 0, out, nc, 4); }); \
    }
    FRONTS(1) FRONTS(2) FRONTS(4) FRONTS(8) FRONTS(16) FRONTS(32)
    CK(hipFree(out));
    if (pad) CK(hipFree(pad));
    pad = nullptr;
  }
  return 0;
}

// One big arena, the matrix placed at successive offsets inside it (argv[1] = "arena")
static int arena_study() {
  const int64_t n_chunks = S / G;
  const size_t big = ROWS * P * 8;
  const int64_t cb = (n_chunks + 7) / 8 * 8;
  const int n_slots = 10;
  char* base = nullptr;
  CK(hipMalloc(&base, big * n_slots));
  CK(hipMemset(base, 0, big * n_slots));
  for (int rep = 0; rep < 2; ++rep)
    for (int i = 0; i < n_slots; ++i) {
      double* out = reinterpret_cast<double*>(base + big * i);
      char nm[64];
      std::snprintf(nm, sizeof nm, "arena slot %d (spread, 1 chunk/block)", i);
      run(nm, [&] { hipLaunchKernelGGL((k_classed_stride<true>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, 1, 4, n_chunks); });
    }
  // finer map: the matrix start moved in quarter-matrix steps through the same arena
  std::printf("start_GiB  ms\n");
  for (size_t off = 0; off + big <= big * n_slots; off += big / 4) {
    double* out = reinterpret_cast<double*>(base + off);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_classed_stride<true>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, 1, 4, n_chunks);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_classed_stride<true>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, 1, 4, n_chunks);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("%8.2f  %.4f\n", off / 1073741824.0, ms / 5);
  }
  CK(hipFree(base));
  return 0;
}

static int alloc2_study() {
  const int64_t n_chunks = S / G;
  const int cpb = 6;
  const int64_t cb = ((n_chunks + cpb - 1) / cpb + 7) / 8 * 8;
  auto measure = [&](const char* what, double* out) {
    CK(hipMemset(out, 0, ROWS * P * 8));
    run(what, [&] { hipLaunchKernelGGL((k_classed<true, false, 256>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, cpb, 4, 0); });
  };
  const size_t big = ROWS * P * 8;
  for (int rep = 0; rep < 2; ++rep) {
    double *a = nullptr, *b = nullptr;
    void* pad = nullptr;
    CK(hipMalloc(&a, big)); measure("v0 nothing before", a); CK(hipFree(a));
    CK(hipMalloc(&pad, 4096)); CK(hipMalloc(&a, big)); measure("v1 4 KB pad before", a); CK(hipFree(a)); CK(hipFree(pad));
    CK(hipMalloc(&pad, 2u << 20)); CK(hipMalloc(&a, big)); measure("v2 2 MB pad before", a); CK(hipFree(a)); CK(hipFree(pad));
    CK(hipMalloc(&pad, 64u << 20)); CK(hipMalloc(&a, big)); measure("v3 64 MB pad before", a); CK(hipFree(a)); CK(hipFree(pad));
    CK(hipMalloc(&pad, 64u << 20)); CK(hipFree(pad)); CK(hipMalloc(&a, big)); measure("v4 64 MB pad allocated+freed before", a); CK(hipFree(a));
    CK(hipMalloc(&a, big)); CK(hipMalloc(&b, big)); measure("v5 first of two", a); measure("v5 second of two", b); CK(hipFree(a)); CK(hipFree(b));
    CK(hipMalloc(&a, big + (64u << 20))); measure("v6 one allocation 64 MB larger", a); CK(hipFree(a));
    CK(hipMalloc(&a, big + 4096)); measure("v7 one allocation 4 KB larger", a); CK(hipFree(a));
  }
  return 0;
}

// ... or on where inside one allocation the matrix starts?  (argv[1] = "offset")
static int offset_study() {
  const int64_t n_chunks = S / G;
  const int cpb = 6;
  const int64_t cb = ((n_chunks + cpb - 1) / cpb + 7) / 8 * 8;
  char* base = nullptr;
  const size_t extra = 1ull << 30;
  CK(hipMalloc(&base, ROWS * P * 8 + extra));
  CK(hipMemset(base, 0, ROWS * P * 8 + extra));
  const size_t offs[] = {0, 4096, 1ull << 20, 2ull << 20, 37ull << 20, 64ull << 20, 101ull << 20, 128ull << 20, 256ull << 20, 512ull << 20, (512ull << 20) + 8000};
  for (size_t off : offs) {
    double* out = reinterpret_cast<double*>(base + off);
    std::printf("offset %zu MiB (+%zu B)\n", off >> 20, off & ((1u << 20) - 1));
    run("  B classed map (sc1 nt)", [&] { hipLaunchKernelGGL((k_classed<true, false, 256>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, cpb, 4, 0); });
    run("  H members spread (sc1 nt)", [&] { hipLaunchKernelGGL((k_classed_spread<true>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, cpb, 4); });
  }
  CK(hipFree(base));
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && std::string(argv[1]) == "alloc") return alloc_study();
  if (argc > 1 && std::string(argv[1]) == "offset") return offset_study();
  if (argc > 1 && std::string(argv[1]) == "alloc2") return alloc2_study();
  if (argc > 1 && std::string(argv[1]) == "stride") return stride_study();
  if (argc > 1 && std::string(argv[1]) == "alloc3") return alloc3_study();
  if (argc > 1 && std::string(argv[1]) == "vmm") return vmm_study();
  if (argc > 1 && std::string(argv[1]) == "fronts") return fronts_study();
  if (argc > 1 && std::string(argv[1]) == "arena") return arena_study();
  double* out = nullptr;
  CK(hipMalloc(&out, ROWS * P * 8));
  CK(hipMemset(out, 0, ROWS * P * 8));
  const int64_t n2 = ROWS * P / 2;
  const int64_t n_chunks = S / G;
  for (int rep = 0; rep < 2; ++rep) {
    run("A linear fill", [&] { hipLaunchKernelGGL(k_linear<false>, dim3(256 * 16), dim3(256), 0, 0, out, n2); });
    run("A linear fill nt", [&] { hipLaunchKernelGGL(k_linear<true>, dim3(256 * 16), dim3(256), 0, 0, out, n2); });
    run("A linear fill grid 256*64", [&] { hipLaunchKernelGGL(k_linear<false>, dim3(256 * 64), dim3(256), 0, 0, out, n2); });
    run("A2 one store per lane", [&] { hipLaunchKernelGGL(k_linear1<false>, dim3((n2 + 255) / 256), dim3(256), 0, 0, out, n2); });
    run("A2 one store per lane nt", [&] { hipLaunchKernelGGL(k_linear1<true>, dim3((n2 + 255) / 256), dim3(256), 0, 0, out, n2); });
    run("A3 4 stores per lane", [&] { hipLaunchKernelGGL(k_linear4<false>, dim3((n2 + 1023) / 1024), dim3(256), 0, 0, out, n2); });
    run("A3 4 stores per lane nt", [&] { hipLaunchKernelGGL(k_linear4<true>, dim3((n2 + 1023) / 1024), dim3(256), 0, 0, out, n2); });
    run("A4 32 B per lane", [&] { hipLaunchKernelGGL(k_linear32<false>, dim3((n2 / 2 + 255) / 256), dim3(256), 0, 0, out, n2); });
    run("A5 hipMemsetAsync", [&] { CK(hipMemsetAsync(out, 0, ROWS * P * 8, 0)); });
    for (int cpb : {1, 6}) {
      const int64_t cb = ((n_chunks + cpb - 1) / cpb + 7) / 8 * 8;
      char nm[96];
      std::snprintf(nm, sizeof nm, "B classed map cpb=%d", cpb);
      run(nm, [&] { hipLaunchKernelGGL((k_classed<false, false, 256>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, cpb, 4, 0); });
      std::snprintf(nm, sizeof nm, "B classed map nt cpb=%d", cpb);
      run(nm, [&] { hipLaunchKernelGGL((k_classed<true, false, 256>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, cpb, 4, 0); });
      std::snprintf(nm, sizeof nm, "B classed map nt +delay64 cpb=%d", cpb);
      run(nm, [&] { hipLaunchKernelGGL((k_classed<true, false, 256>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, cpb, 4, 64); });
      if (cpb == 6) {
        run("P classed map, store policy plain", [&] { hipLaunchKernelGGL((k_classed_policy<0>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, cpb, 4); });
        run("P classed map, store policy nt", [&] { hipLaunchKernelGGL((k_classed_policy<1>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, cpb, 4); });
        run("P classed map, store policy sc1", [&] { hipLaunchKernelGGL((k_classed_policy<2>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, cpb, 4); });
        run("P classed map, store policy sc0 sc1", [&] { hipLaunchKernelGGL((k_classed_policy<3>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, cpb, 4); });
        run("P classed map, store policy sc1 nt", [&] { hipLaunchKernelGGL((k_classed_policy<4>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, cpb, 4); });
        run("P classed map, store policy sc0 sc1 nt", [&] { hipLaunchKernelGGL((k_classed_policy<5>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, cpb, 4); });
        run("P classed map, store policy sc0", [&] { hipLaunchKernelGGL((k_classed_policy<6>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, cpb, 4); });
      }
      std::snprintf(nm, sizeof nm, "C full-row blocks (1024 thr) nt cpb=%d", cpb);
      run(nm, [&] { hipLaunchKernelGGL((k_classed<true, false, 1024>), dim3(cb), dim3(1024), 0, 0, out, n_chunks, cpb, 1, 0); });
      std::snprintf(nm, sizeof nm, "D obs-major rows nt cpb=%d", cpb);
      run(nm, [&] { hipLaunchKernelGGL((k_classed<true, true, 256>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, cpb, 4, 0); });
      std::snprintf(nm, sizeof nm, "F 2 obs rows back to back nt cpb=%d", cpb);
      run(nm, [&] { hipLaunchKernelGGL((k_classed_kb<true, 2>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, cpb, 4); });
      std::snprintf(nm, sizeof nm, "G 7 obs rows back to back nt cpb=%d", cpb);
      run(nm, [&] { hipLaunchKernelGGL((k_classed_kb<true, 7>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, cpb, 4); });
      std::snprintf(nm, sizeof nm, "H members spread nt cpb=%d", cpb);
      run(nm, [&] { hipLaunchKernelGGL((k_classed_spread<true>), dim3(cb * 4), dim3(256), 0, 0, out, n_chunks, cpb, 4); });
      std::snprintf(nm, sizeof nm, "E 1KB-per-store rows nt cpb=%d", cpb);
      run(nm, [&] { hipLaunchKernelGGL((k_rowwide<true>), dim3(cb * 2), dim3(256), 0, 0, out, n_chunks, cpb, 2); });
      std::snprintf(nm, sizeof nm, "E 1KB-per-store rows cpb=%d", cpb);
      run(nm, [&] { hipLaunchKernelGGL((k_rowwide<false>), dim3(cb * 2), dim3(256), 0, 0, out, n_chunks, cpb, 2); });
    }
  }
  CK(hipFree(out));
  return 0;
}
