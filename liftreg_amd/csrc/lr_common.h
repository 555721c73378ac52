// lr_common.h — shared device/host helpers for libliftreg_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <atomic>
#include "../../include/liftreg_hip.h"

#define LR_WAVE 64

// Launch check: report, never throw (C ABI).
static inline int lr_launch_status() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess && getenv("LIFTREG_HIP_DEBUG"))
    fprintf(stderr, "liftreg_hip: launch failed: %s (%s)\n", hipGetErrorName(e), hipGetErrorString(e));
  return e == hipSuccess ? LR_OK : LR_ELAUNCH;
}

// Every launch site converts its `void* stream` through here first, which also drops any stale
// non-sticky error an earlier runtime call of the host application left behind (hipGetLastError
// is per-thread state shared with PyTorch), so lr_launch_status() reports THIS launch only.
static inline hipStream_t lr_stream(void* s) {
  (void)hipGetLastError();
  return reinterpret_cast<hipStream_t>(s);
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: `done` (one static per kernel
// instantiation at the launch site) remembers the devices it was raised on, so a process that drives several GPUs — or
// several threads launching at once — raises it on each of them.  Returns LR_OK or LR_ELAUNCH (checked by every caller).
static inline int lr_raise_dyn_lds(const void* kernel, size_t bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return LR_ELAUNCH;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return LR_OK;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return LR_ELAUNCH;
  done.fetch_or(bit, std::memory_order_release);
  return LR_OK;
}

// Emitter poses travel by value (kernel-argument segment → SGPR loads).
struct LrPoses {
  float e[LR_MAX_VIEWS][3];
};

// Bijective XCD-aware block remap: hardware deals consecutive block ids
// round-robin over the 8 XCDs; give each XCD one contiguous chunk of the
// logical grid so neighbouring tiles share an L2 (speed only, never
// correctness).
__device__ __forceinline__ unsigned lr_xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned xcd = bid & 7u, q = nblk >> 3, r = nblk & 7u;
  const unsigned base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
  return base + (bid >> 3);
}

// ATen grid_sample un-normalise, align_corners=True:
// ((g + 1) / 2) * (size - 1)   — aten/src/ATen/native/GridSampler.h
__device__ __forceinline__ float lr_unnormalize(float g, int size) {
  return ((g + 1.0f) * 0.5f) * (float)(size - 1);
}

__device__ __forceinline__ float lr_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double lr_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// ncc.hip: fixed-order sum of per-block moment partials [R][nblk][5] -> moments [R][5] (one block per row)
int lr_internal_ncc_reduce(const double* partial, double* moments, int R, int nblk, hipStream_t st);

// conv3d_rows.hip: a stride-2 block (parity-split channels-last input, Cin = 16 | 32) as the persistent Winograd-along-W
// rows kernel with its weight fragments in LDS (z_phase: parity of the global output plane of local plane 0).
// LR_EUNSUPPORTED -> use conv3d.hip's kernels.
int lr_internal_conv_rows_wlds(const float* in, const float* packed_w, const float* bias, float* out, int B, int Cin,
                               int Cout, int D, int W, int H, int out_layout, float slope, int z_phase, long long out_bs,
                               hipStream_t st);

// conv0_pc.hip: the first block as a producer/consumer kernel (double-buffered LDS brick; proj != NULL: channels 1..P
// are the backprojection of the views, computed by the producers).  LR_EUNSUPPORTED -> use the single-buffer kernel.
int lr_internal_conv0_pc(const float* in0, int64_t bs0, const float* in_rest, int64_t bsr, const float* packed_w,
                         const float* bias, float* out, int B, int Cin, int D, int W, int H, int out_layout, float slope,
                         const float* proj, const float* poses, int P, int Pw, int Ph, hipStream_t st);

// conv0_cl_bf16.hip: the first block with 4 < Cin <= 16 input channels on the bf16 MFMA (all channels of a brick staged once,
// channels-last in LDS).  Its operand packing is appended to the channel-pass packing of lr_conv3d_pack_weights_bf16_planar.
// LR_EUNSUPPORTED -> use conv0_bf16_kernel's channel passes.
int64_t lr_internal_conv0_cl_bf16_packed_bytes(int Cin, int Cout);
int lr_internal_conv0_cl_bf16_pack(const float* weight, void* packed, int Cin, int Cout, hipStream_t st);
int lr_internal_conv0_cl_bf16(const float* in, const void* packed, const float* bias, void* out, int B, int Cin, int Cout, int D,
                              int W, int H, int out_layout, float slope, long long out_bs, int clin, hipStream_t st);

// conv0_split_f32.hip: the first block (Cin <= 4, fp32 in / fp32 channels-last out) on the bf16 MFMA with exact three-way
// operand splits (6 of the 9 partial products).  LR_EUNSUPPORTED -> conv3d.hip's fp32-MFMA kernels.
int64_t lr_internal_conv0_split_packed_floats(int Cin, int Cout);
int lr_internal_conv0_split_pack(const float* weight, float* packed, int Cin, int Cout, hipStream_t st);
int lr_internal_conv0_split_f32(const float* in0, long long bs0, const float* in_rest, long long bsr, const float* packed,
                                const float* bias, float* out, int B, int Cin, int D, int W, int H, int out_layout, float slope,
                                long long out_bs, hipStream_t st);
// the same march under the bf16 storage contract (one rounding of the operands, bf16 channels-last output): the 3-channel
// first block of the bf16 variant; `packed` as written by lr_internal_conv0_split_pack
int lr_internal_conv0_march_bf16(const float* in, const void* packed, const float* bias, void* out, int B, int Cin, int D, int W,
                                 int H, int out_layout, float slope, long long out_bs, unsigned char* mask_out, hipStream_t st);
