// Which store shape fills HBM fastest on this box?  (feeds pmx_measure_write_ceiling's choice of shapes)
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/fill_probe.hip -o /tmp/fill_probe && /tmp/fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef double dbl2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void fill_stride16(double* d, int64_t n_pairs) {
  dbl2 v; v.x = 0.0; v.y = 0.0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_pairs; i += stride) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<dbl2*>(d) + i); else reinterpret_cast<dbl2*>(d)[i] = v;
  }
}
template <bool NT>
__global__ __launch_bounds__(256) void fill_stride8(double* d, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    if (NT) __builtin_nontemporal_store(0.0, d + i); else d[i] = 0.0;
  }
}
// the prediction kernels' shape: block = 256 columns x a chunk of rows; lanes across columns
template <bool NT>
__global__ __launch_bounds__(256) void fill_rows(double* d, int64_t rows, int64_t P, int64_t ld, int32_t n_ptiles, int32_t rows_per_block) {
  const int64_t b = blockIdx.x;
  const int32_t ptile = (int32_t)(b % n_ptiles);
  const int64_t chunk = b / n_ptiles;
  const int64_t p = (int64_t)ptile * 256 + threadIdx.x;
  const int64_t r0 = chunk * rows_per_block, r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
  if (p >= P) return;
  for (int64_t r = r0; r < r1; ++r) {
    if (NT) __builtin_nontemporal_store(0.0, d + r * ld + p); else d[r * ld + p] = 0.0;
  }
}
int main() {
  const int64_t rows = 700000, P = 1000, ld = 1000;  // the C3 prediction matrix: 5.6 GB
  const int64_t n = rows * ld;
  double* d = nullptr;
  CK(hipMalloc(&d, n * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char* name, auto launch) {
    launch();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::printf("%-40s %8.3f ms  %8.1f GB/s\n", name, ms / 10, n * 8.0 / (ms / 10 * 1e-3) / 1e9);
    std::fflush(stdout);
  };
  char nm[96];
  for (int blocks : {2048, 4096, 8192, 16384, 65536, 262144}) {
    std::snprintf(nm, sizeof nm, "stride16 nt blocks=%d", blocks);
    time(nm, [&] { hipLaunchKernelGGL(fill_stride16<true>, dim3(blocks), dim3(256), 0, 0, d, n / 2); });
    std::snprintf(nm, sizeof nm, "stride16 plain blocks=%d", blocks);
    time(nm, [&] { hipLaunchKernelGGL(fill_stride16<false>, dim3(blocks), dim3(256), 0, 0, d, n / 2); });
    std::snprintf(nm, sizeof nm, "stride8 nt blocks=%d", blocks);
    time(nm, [&] { hipLaunchKernelGGL(fill_stride8<true>, dim3(blocks), dim3(256), 0, 0, d, n); });
  }
  {
    const int64_t exact = (n / 2 + 255) / 256;
    time("one 16B store per lane (exact grid) nt", [&] { hipLaunchKernelGGL(fill_stride16<true>, dim3((uint32_t)exact), dim3(256), 0, 0, d, n / 2); });
  }
  for (int rpb : {7, 28, 56, 112, 448}) {
    const int32_t n_ptiles = (int32_t)((P + 255) / 256);
    const int64_t chunks = (rows + rpb - 1) / rpb;
    std::snprintf(nm, sizeof nm, "rows nt rows/block=%d", rpb);
    time(nm, [&] { hipLaunchKernelGGL(fill_rows<true>, dim3((uint32_t)(chunks * n_ptiles)), dim3(256), 0, 0, d, rows, P, ld, n_ptiles, rpb); });
    std::snprintf(nm, sizeof nm, "rows plain rows/block=%d", rpb);
    time(nm, [&] { hipLaunchKernelGGL(fill_rows<false>, dim3((uint32_t)(chunks * n_ptiles)), dim3(256), 0, 0, d, rows, P, ld, n_ptiles, rpb); });
  }
  time("hipMemsetAsync", [&] { hipMemsetAsync(d, 0, n * 8, 0); });
  CK(hipDeviceSynchronize());
  CK(hipFree(d));
  return 0;
}
