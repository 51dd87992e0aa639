// Micro-benchmark (diagnostic): the HBM bandwidth the ACCESS PATTERN of the tile machine can reach on gfx950,
// without any of its arithmetic or decoding.  The tile machine keeps the state of 64 instances in one tile of `rows` rows of
// 512 bytes (one double per lane and row); a record loads up to 32 operand rows (all loads in flight, then the use), stores up to
// 16 result rows; the rows a record names are scattered over the tile (26 MB for l1l2) and a row is read a few times, far apart.
// Here: `tiles` workgroups of W wavefronts each own a tile; every wavefront repeats {NL row loads in flight -> add -> NS row
// stores} over rows drawn by a hash, optionally with the streaming hint the walk uses (`nt`).  Reported: bytes moved / time.
// Modes:  r = NL loads only     rw = NL loads + NS stores     seq = rows in ascending order instead of hashed
//   hipcc --offload-arch=gfx950 -O3 tile_rows.hip -o tile_rows ; ./tile_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int NL, int NS, bool NT, bool SEQ, int G = 1, int D = 1>
__global__ void __launch_bounds__(1024) walk(double *state, size_t rows, int iters, double *sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
  double *tile = state + (size_t)blockIdx.x * rows * 64;
  unsigned h = 0x9e3779b9u * (blockIdx.x * 16 + wave + 1);
  size_t seq = (size_t)wave * (rows / W);
  double acc = 0;
  for (int it = 0; it < iters; it++) {
    double v[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) {
      size_t r;
      if (SEQ) { r = seq++; if (seq >= rows) seq = 0; }
      else if (D > 1) {      // the two rows of a pair are issued D loads apart: k = blk * 2D + member * D + j
        const int blk = k / (2 * D), member = (k / D) & 1, j = k % D;
        const unsigned hp = (h + 0x9e3779b9u * (unsigned)(it * 64 + blk * D + j + 1)) * 1664525u + 1013904223u;
        r = ((size_t)(__builtin_amdgcn_readfirstlane(hp) >> 8) % (rows / 2)) * 2 + member;
      } else {
        if (k % G == 0) h = h * 1664525u + 1013904223u;
        r = ((size_t)(__builtin_amdgcn_readfirstlane(h) >> 8) % (rows / G)) * G + k % G;
      }
      const double *p = tile + r * 64 + lane;
      v[k] = NT ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int k = 0; k < NL; k++) acc += v[k];
#pragma unroll
    for (int k = 0; k < NS; k++) {
      size_t r;
      if (SEQ) { r = seq++; if (seq >= rows) seq = 0; }
      else {
        if (k % G == 0) h = h * 1664525u + 1013904223u;
        r = ((size_t)(__builtin_amdgcn_readfirstlane(h) >> 8) % (rows / G)) * G + k % G;
      }
      double *p = tile + r * 64 + lane;
      if (NT) __builtin_nontemporal_store(acc + k, p); else *p = acc + k;
    }
  }
  if (acc == 12345.678) sink[0] = acc;       // (keeps the loads alive)
}

template <int NL, int NS, bool NT, bool SEQ, int G = 1, int D = 1>
static void run(const char *name, double *state, size_t rows, int tiles, int W, double *sink) {
  const int iters = 4000 / (NL + NS) * 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  walk<NL, NS, NT, SEQ, G, D><<<tiles, 64 * W>>>(state, rows, iters / 8, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  walk<NL, NS, NT, SEQ, G, D><<<tiles, 64 * W>>>(state, rows, iters, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)tiles * W * iters * (NL + NS) * 512.0;
  printf("%-44s tiles=%4d W=%2d: %7.2f ms  %6.2f TB/s\n", name, tiles, W, ms, bytes / ms * 1e-9);
  fflush(stdout);
}

int main(int argc, char **argv) {
  const size_t rows = 51200;                 // 26 MB per tile, as l1l2's
  const int tiles = 256;
  double *state, *sink;
  if (hipMalloc(&state, rows * 512 * tiles) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  hipMalloc(&sink, 8);
  hipMemset(state, 0, rows * 512 * tiles);
  for (int W : {8, 16}) {
    run<32, 0, false, true>("sequential rows, loads only", state, rows, tiles, W, sink);
    run<32, 0, false, false>("hashed rows, 32 loads", state, rows, tiles, W, sink);
    run<32, 0, true, false>("hashed rows, 32 loads, nt", state, rows, tiles, W, sink);
    run<24, 8, false, false>("hashed rows, 24 loads + 8 stores", state, rows, tiles, W, sink);
    run<24, 8, true, false>("hashed rows, 24 loads + 8 stores, nt", state, rows, tiles, W, sink);
    run<24, 8, true, true>("sequential rows, 24 loads + 8 stores, nt", state, rows, tiles, W, sink);
    run<12, 4, true, false>("hashed rows, 12 loads + 4 stores, nt", state, rows, tiles, W, sink);
  }
  // runs of G consecutive rows at a hashed place: what a storage map that puts the operands of a record side by side would see
  run<24, 8, true, false, 2>("hashed runs of 2 rows, 24 loads + 8 stores, nt", state, rows, tiles, 8, sink);
  run<24, 8, true, false, 4>("hashed runs of 4 rows, 24 loads + 8 stores, nt", state, rows, tiles, 8, sink);
  run<24, 8, true, false, 8>("hashed runs of 8 rows, 24 loads + 8 stores, nt", state, rows, tiles, 8, sink);
  // ... and when the two rows of a pair are not issued back to back but D loads apart (loads only)
  run<32, 0, true, false, 2, 1>("hashed pairs, 32 loads, back to back", state, rows, tiles, 8, sink);
  run<32, 0, true, false, 2, 2>("hashed pairs, 32 loads, 2 loads apart", state, rows, tiles, 8, sink);
  run<32, 0, true, false, 2, 4>("hashed pairs, 32 loads, 4 loads apart", state, rows, tiles, 8, sink);
  run<32, 0, true, false, 2, 16>("hashed pairs, 32 loads, 16 loads apart", state, rows, tiles, 8, sink);
  // the tile machine's other knob: more tiles than CUs (a batch of 32 768 / 65 536)
  run<24, 8, true, false>("hashed rows, 24 loads + 8 stores, nt", state, rows / 2, tiles * 2, 8, sink);
  hipFree(state); hipFree(sink);
  return 0;
}
