// Micro-benchmark (diagnostic): what it costs on gfx950 to hand ONE fp64 value from the lane that computed
// it to the lane that needs it next, the way a barrier-separated step of the LDS machine does it and the ways a
// register-resident critical chain could:
//   0  LDS + workgroup barrier (W wavefronts): ds_write, s_waitcnt, s_barrier, ds_read of another WAVEFRONT's slot, use
//   1  LDS inside one wavefront, no barrier: ds_write, ds_read of another lane's slot (a wavefront's LDS operations are in order)
//   2  ds_bpermute_b32 x 2: a register of another lane, per-lane source index
//   3  v_readlane x 2: a register of one lane broadcast to all (uniform source)
//   4  the dependent fp64 division alone (correctly rounded, as the evaluators divide)
//   5  a dependent fp64 add alone (reference: 8 cycles)
// Every variant is a dependent chain (the value handed over feeds the next hand-over), so cycles / iteration is the LATENCY of
// one hand-over (+ one add).  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off lds_handoff.hip -o lds_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE, int W>
__global__ void __launch_bounds__(64 * W) handoff(double *out, long long *cyc, double seed, int iters) {
  __shared__ double buf[64 * W];
  const int t = threadIdx.x, lane = t & 63;
  const int other = (MODE == 0) ? (t + 64) % (64 * W) : (t & ~63) | ((lane + 17) & 63);     // a lane of the NEXT wavefront / of this one
  double x = seed + t * 1e-3;
  const double d = 1.0 + seed * 1e-3;
  buf[t] = x;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
      buf[t] = x;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
      x = buf[other] + d;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");      // (the read must be over before the next write)
      __builtin_amdgcn_s_barrier();
    } else if (MODE == 1) {
      buf[t] = x;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      x = buf[other] + d;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    } else if (MODE == 2) {
      const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
      const int lo = __builtin_amdgcn_ds_bpermute(((lane + 17) & 63) << 2, (int)(unsigned)u);
      const int hi = __builtin_amdgcn_ds_bpermute(((lane + 17) & 63) << 2, (int)(unsigned)(u >> 32));
      x = __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo) + d;
    } else if (MODE == 3) {
      const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
      const int lo = __builtin_amdgcn_readlane((int)(unsigned)u, 17);
      const int hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 17);
      x = __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo) + d + lane * 1e-9;
    } else if (MODE == 4) {
      x = x / d;
    } else {
      x = x + d;
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 * W + t] = x;
  if (t == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int W>
static void run(const char *name, int nblk) {
  double *out; long long *cyc;
  const int iters = 2000;
  hipMalloc(&out, (size_t)nblk * 64 * W * sizeof(double));
  hipMalloc(&cyc, nblk * sizeof(long long));
  for (int rep = 0; rep < 2; rep++) handoff<MODE, W><<<nblk, 64 * W>>>(out, cyc, 1.0 + 1e-9, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(nblk);
  hipMemcpy(h.data(), cyc, nblk * sizeof(long long), hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto v : h) mean += v;
  mean /= nblk;
  // s_memtime counts at 100 MHz on this part?  No: it is the shader clock's constant-rate counter used by every other
  // script of this repo (scripts/level_trace.py); the fp64 add line calibrates it (8 cycles)
  printf("%-46s W=%d blocks=%4d: %7.1f cycles (s_memtime) per hand-over\n", name, W, nblk, mean / iters);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int nblk : {1, 256}) {
    run<5, 1>("fp64 add, dependent (calibration)", nblk);
    run<4, 1>("fp64 division, dependent", nblk);
    run<3, 1>("v_readlane broadcast + add", nblk);
    run<2, 1>("ds_bpermute + add", nblk);
    run<1, 1>("LDS write -> read, one wavefront + add", nblk);
    run<0, 2>("LDS write -> barrier -> read -> barrier + add", nblk);
    run<0, 4>("LDS write -> barrier -> read -> barrier + add", nblk);
    run<0, 8>("LDS write -> barrier -> read -> barrier + add", nblk);
  }
  return 0;
}
