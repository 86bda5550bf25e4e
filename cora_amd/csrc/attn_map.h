// attn_map.h -- what the fp32 (attn.hip) and bf16 (attn_bf16.hip) attention kernels share: the
// workgroup -> (row block, head, image) map, the choice of waves per workgroup, small lane helpers.
#pragma once
#include <stdlib.h>
#include "common.h"

// Workgroup -> (row block, head, image).  The hardware deals consecutive workgroup ids round-robin over the
// 8 XCDs, each with its own L2; the row blocks of one (image, head) all stream the same K / V (or Q / dO)
// rows, so the 1-D grid is cut into 8 contiguous runs of logical ids, one per XCD, with the row block
// fastest: the blocks that share operands run on one XCD at about the same time and all but the first
// find them in that L2.  The grid is rounded up to a multiple of 8; surplus workgroups return at once.
__device__ __forceinline__ bool attn_coords(int nrb, int H, int nimg, int& rb, int& h, int& img) {
  const int total = nrb * H * nimg;
  const int per_xcd = (total + 7) >> 3;
  const int id = (int)blockIdx.x;
  const int logical = (id & 7) * per_xcd + (id >> 3);
  if (logical >= total) return false;
  rb = logical % nrb;
  const int t = logical / nrb;
  h = t % H;
  img = t / H;
  return true;
}
static inline unsigned attn_grid(int nrb, int H, int nimg) {
  const long long total = (long long)nrb * H * nimg;
  return (unsigned)(8 * ((total + 7) / 8));
}

template <int V> struct IntC { static constexpr int value = V; };
// run body(IntC<nvt>) for nvt = 1..4 valid 16-row tiles in the streamed chunk (uniform over the workgroup)
template <typename F>
__device__ __forceinline__ void for_valid_tiles(int nvt, F&& body) {
  if (nvt >= 4) body(IntC<4>{});
  else if (nvt == 3) body(IntC<3>{});
  else if (nvt == 2) body(IntC<2>{});
  else body(IntC<1>{});
}

__device__ __forceinline__ float grp_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}
__device__ __forceinline__ float grp_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// Waves per workgroup: 8 (128 own rows, the streamed chunks staged half as often per own row, and the next
// chunk prefetched into registers under the compute) when the sequence is longer than a chunk and the launch
// still gives every CU two such workgroups, else 4.  Measured (fp32 kernels): video16 (N = 1024) forward
// 253 -> 230 us, backward 886 -> 796 us; cfg 4 (N = 196, two ragged row blocks of 128) forward 322 -> 303 us but
// backward 1 033 -> 1 094 us, so short sequences keep 4 waves in the backward.
static inline int attn_waves(int nimg, int N, int H, bool bwd) {
  if (const char* e = getenv("CORA_ATTN_WAVES")) { int v = atoi(e); if (v == 4 || v == 8) return v; }
  if (N <= 64) return 4;
  const long long wg8 = (long long)((N + 127) / 128) * H * nimg;
  if (wg8 < 512) return 4;
  if (bwd && N < 256) return 4;
  return 8;
}
