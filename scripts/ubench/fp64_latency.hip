// Micro-benchmark (diagnostic): cycles per fp64 VALU instruction on gfx950 for a dependent chain vs K independent
// chains, one wavefront per SIMD.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off fp64_latency.hip -o fp64_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int K, int OP>
__global__ void __launch_bounds__(64) chain(double *out, long long *cyc, double seed) {
  double a[K];
#pragma unroll
  for (int k = 0; k < K; k++) a[k] = seed + threadIdx.x * 1e-3 + k;
  const double m = seed * 0.5 + 1.0;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 64; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
#pragma unroll
      for (int k = 0; k < K; k++) {
        if (OP == 0) a[k] = a[k] + m;
        else if (OP == 1) a[k] = a[k] * m;
        else if (OP == 2) a[k] = a[k] + a[k] * m;         // mul then dependent add (no contraction)
        else a[k] = __builtin_fma(a[k], m, m);
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int k = 0; k < K; k++) s += a[k];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K, int OP>
static void run(const char *name, int nblk) {
  double *out; long long *cyc;
  hipMalloc(&out, nblk * 64 * sizeof(double));
  hipMalloc(&cyc, nblk * sizeof(long long));
  for (int rep = 0; rep < 2; rep++) chain<K, OP><<<nblk, 64>>>(out, cyc, 1.0 + 1e-9);
  hipDeviceSynchronize();
  std::vector<long long> h(nblk);
  hipMemcpy(h.data(), cyc, nblk * sizeof(long long), hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto v : h) mean += v;
  mean /= nblk;
  const double nins = 64.0 * 16 * K * (OP == 2 ? 2 : 1);
  printf("%-14s K=%2d blocks=%5d: %.2f cycles (s_memtime) per instruction\n", name, K, nblk, mean / nins);
  hipFree(out); hipFree(cyc);
}

int main() {
  // measured on MI355X: dependent add / mul / fma 7.9-8.2 cycles per instruction, 4 independent chains 4.3-4.6
  for (int nblk : {1, 1024}) {
    run<1, 0>("add dep", nblk); run<2, 0>("add", nblk); run<4, 0>("add", nblk); run<8, 0>("add", nblk);
    run<1, 1>("mul dep", nblk); run<4, 1>("mul", nblk); run<8, 1>("mul", nblk);
    run<1, 2>("mul+add dep", nblk); run<2, 2>("mul+add", nblk); run<4, 2>("mul+add", nblk); run<8, 2>("mul+add", nblk);
    run<1, 3>("fma dep", nblk); run<4, 3>("fma", nblk); run<8, 3>("fma", nblk);
  }
  return 0;
}
