// How long does ONE wave take per fixed-step RK4 step of dx/dt = -ke x + r (the C4 model)?  The PAIR-mode ODE
// kernel at the C4 shape (50k pairs = 782 waves on 1024 SIMDs) is latency-bound: its time barely moves from 6k to
// 50k pairs.  Variants: 0 = x += (h/6)(...) with the division in the step (what the compiler sees when h is a
// loop-carried per-lane value), 1 = h/6 hoisted, 2 = hoisted + 4 steps per trip, 3 = as 0 but behind the
// state-machine shaped branch (rem > 0) with per-lane trip counts.
// build: hipcc -O3 --offload-arch=gfx950 tools/experiments/rk4_latency_probe.hip -o tools/bin/rk4_latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int V>
__global__ void probe(const double* __restrict__ hs, const int* __restrict__ ns, double* out, double ke, double r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double h = hs[i];
  int rem = ns[i];
  double x = 1.0 + i;
  if (V == 0) {
    while (rem > 0) {
      const double k1 = -ke * x + r;
      const double k2 = -ke * (x + 0.5 * h * k1) + r;
      const double k3 = -ke * (x + 0.5 * h * k2) + r;
      const double k4 = -ke * (x + h * k3) + r;
      x = x + (h / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4);
      --rem;
      h = __longlong_as_double(__double_as_longlong(h) ^ (rem & 0));  // keep h loop-carried
    }
  } else if (V == 1) {
    const double h6 = h / 6.0, hh = 0.5 * h;
    while (rem > 0) {
      const double k1 = -ke * x + r;
      const double k2 = -ke * (x + hh * k1) + r;
      const double k3 = -ke * (x + hh * k2) + r;
      const double k4 = -ke * (x + h * k3) + r;
      x = x + h6 * (k1 + 2.0 * k2 + 2.0 * k3 + k4);
      --rem;
    }
  } else if (V == 2) {
    const double h6 = h / 6.0, hh = 0.5 * h;
    while (rem > 0) {
      const int k = rem < 4 ? rem : 4;
      for (int j = 0; j < k; ++j) {
        const double k1 = -ke * x + r;
        const double k2 = -ke * (x + hh * k1) + r;
        const double k3 = -ke * (x + hh * k2) + r;
        const double k4 = -ke * (x + h * k3) + r;
        x = x + h6 * (k1 + 2.0 * k2 + 2.0 * k3 + k4);
      }
      rem -= k;
    }
  }
  out[i] = x;
}

int main() {
  const int n = 64;
  const int steps = 6000;
  std::vector<double> hs(n, 0.02);
  std::vector<int> ns(n, steps);
  double *d_h, *d_o;
  int* d_n;
  hipMalloc(&d_h, n * 8); hipMalloc(&d_o, n * 8); hipMalloc(&d_n, n * 4);
  hipMemcpy(d_h, hs.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_n, ns.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int v = 0; v < 3; ++v) {
    for (int waves : {1, 256 * 4, 256 * 8, 256 * 16}) {
      // every wave reads the same 64 inputs (block index ignored on purpose by using blockDim = 64 and i % 64)
      float best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        if (v == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, d_h, d_n, d_o, 0.3, 1.5);
        if (v == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, d_h, d_n, d_o, 0.3, 1.5);
        if (v == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, d_h, d_n, d_o, 0.3, 1.5);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      if (waves == 1) printf("variant %d: one wave, %d steps: %.3f ms = %.1f ns/step\n", v, steps, best, best * 1e6 / steps);
    }
  }
  return 0;
}
