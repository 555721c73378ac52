// linear.hip — K4: small-batch Linear (+LeakyReLU) for the FC head
// 32(n/32)^3 → 800 → 256 → L.  Weight-streaming GEMV-like kernel: HBM-bound on
// the weight read (52 MB for FC1 at 256^3); a block owns OT=4 output neurons so
// the L2-resident activations are re-read 4x less, its 4 wavefronts split K.
//
// Replaces (reference file:line)
//   src/liftreg/layers/layers.py:413-439  FullyConnectBlock
//   src/liftreg/models/LiftRegDeformSubspaceBackproj.py:34-39
#include "lr_common.h"

namespace {

constexpr int OT4 = 4;

// OT output neurons per block: 4 (the activations are re-read 4x less), or 1 when 4 would leave CUs idle (FC1: 800 neurons =
// 200 blocks of 4 on a 256-CU chip; the 52 MB weight read ran at 1.4 TB/s).  Same accumulation order per neuron either way.
template <int BT, int OT>
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ w,
                                                     const float* __restrict__ bias,
                                                     float* __restrict__ y, int B, int b_lo, int K,
                                                     int O, float slope, int vec_ok) {
  const int o0 = blockIdx.x * OT;
  b_lo += (int)blockIdx.y * BT;   // batches above BT rows: the row chunks are blockIdx.y of ONE launch (round 6: B = 30 was four launches per layer)
  float acc[OT][BT];
#pragma unroll
  for (int o = 0; o < OT; ++o)
#pragma unroll
    for (int b = 0; b < BT; ++b) acc[o][b] = 0.0f;

  const float* wr[OT];
  const float* xr[BT];
#pragma unroll
  for (int o = 0; o < OT; ++o) wr[o] = w + (int64_t)min(o0 + o, O - 1) * K;
#pragma unroll
  for (int b = 0; b < BT; ++b) xr[b] = x + (int64_t)min(b_lo + b, B - 1) * K;

  int k_done = 0;
  if (vec_ok) {
    const int K4 = K >> 2;
#pragma unroll 4
    for (int k4 = threadIdx.x; k4 < K4; k4 += 256) {  // several weight loads in flight per thread
      float4 wv[OT];
#pragma unroll
      for (int o = 0; o < OT; ++o) wv[o] = reinterpret_cast<const float4*>(wr[o])[k4];
#pragma unroll
      for (int b = 0; b < BT; ++b) {
        const float4 xv = reinterpret_cast<const float4*>(xr[b])[k4];
#pragma unroll
        for (int o = 0; o < OT; ++o) {
          float a = acc[o][b];
          a = fmaf(wv[o].x, xv.x, a);
          a = fmaf(wv[o].y, xv.y, a);
          a = fmaf(wv[o].z, xv.z, a);
          a = fmaf(wv[o].w, xv.w, a);
          acc[o][b] = a;
        }
      }
    }
    k_done = K4 << 2;
  }
  for (int k = k_done + threadIdx.x; k < K; k += 256) {
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      const float xv = xr[b][k];
#pragma unroll
      for (int o = 0; o < OT; ++o) acc[o][b] = fmaf(wr[o][k], xv, acc[o][b]);
    }
  }

  __shared__ float red[4][OT * BT];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 0; o < OT; ++o)
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      const float s = lr_wave_sum(acc[o][b]);
      if (lane == 0) red[wave][o * BT + b] = s;
    }
  __syncthreads();
  if (threadIdx.x < OT * BT) {
    const int o = threadIdx.x / BT, b = threadIdx.x % BT;
    if (o0 + o < O && b_lo + b < B) {
      float s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
      if (bias) s = s + bias[o0 + o];
      s = s >= 0.0f ? s : s * slope;  // LeakyReLU(slope); slope = 1 → identity
      y[(int64_t)(b_lo + b) * O + o0 + o] = s;
    }
  }
}

}  // namespace

extern "C" int lr_linear_lrelu_f32(const float* x, const float* w, const float* bias, float* y,
                                   int B, int K, int O, float negative_slope, void* stream) {
  if (!x || !w || !y) return LR_ENULL;
  if (B < 1 || B > 32 || K < 1 || O < 1) return LR_EINVAL;
  const int vec_ok = ((K & 3) == 0) &&
                     (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15u) == 0);
  const bool wide = (O + OT4 - 1) / OT4 < 512;  // few neurons: one per block
  const unsigned nblk = wide ? (unsigned)O : (unsigned)((O + OT4 - 1) / OT4);
  hipStream_t st = lr_stream(stream);
  dim3 grid(nblk);
#define LR_LIN(BTV)                                                                                                          \
  do {                                                                                                                      \
    if (wide) hipLaunchKernelGGL((linear_kernel<BTV, 1>), grid, dim3(256), 0, st, x, w, bias, y, B, b_lo, K, O, negative_slope, vec_ok); \
    else hipLaunchKernelGGL((linear_kernel<BTV, OT4>), grid, dim3(256), 0, st, x, w, bias, y, B, b_lo, K, O, negative_slope, vec_ok);    \
  } while (0)
  {
    const int b_lo = 0;
    if (B > 4) { grid.y = (unsigned)((B + 7) / 8); LR_LIN(8); }   // rows past B are clamped on the way in and not written
    else if (B > 1) LR_LIN(4);
    else LR_LIN(1);
  }
#undef LR_LIN
  return lr_launch_status();
}
