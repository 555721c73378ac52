// lr_common.h — shared device/host helpers for libliftreg_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <atomic>
#include "../../include/liftreg_hip.h"

#define LR_WAVE 64

// ---- run-time switches.  The library reads its LIFTREG_* environment variables ONCE per process (first launch; thread-safe)
// into this table — a launcher looks a value up, it never calls getenv().  lr_reload_switches() (C ABI) re-reads them: tests and
// A/B tools that flip a switch between two calls of one process call it after changing the environment.  The list (with the
// meaning of each variable) is part of the header: include/liftreg_hip.h.
enum LrSwitch {
  LR_SW_HIP_DEBUG,   // LIFTREG_HIP_DEBUG: print the HIP error text of a failed launch to stderr
  LR_SW_CONV_DIRECT,   // LIFTREG_CONV_DIRECT: stride-2 fp32 blocks: the direct row walk = the oracle's fmaf chain bit for bit (default: Winograd F(2,2) rows kernel)
  LR_SW_CONV0_DIRECT,   // LIFTREG_CONV0_DIRECT: first fp32 block: the direct sweep = the oracle's fmaf chain (default: Winograd F(2,3) along H)
  LR_SW_CONV_TAPMAJOR,   // LIFTREG_CONV_TAPMAJOR: stride-2 blocks: the tap-major kernel instead of the row kernels (same bits as the direct walk)
  LR_SW_CONV_ROWS_ALWAYS,   // LIFTREG_CONV_ROWS_ALWAYS: persistent Winograd rows kernel also on planes below 64 x 64 outputs (tests)
  LR_SW_CONV0_BF16_CL,   // LIFTREG_CONV0_BF16_CL: bf16 first block: the all-channels brick kernel also for <= 3 channels
  LR_SW_CONV0_BF16_PASSES,   // LIFTREG_CONV0_BF16_PASSES: bf16 first block: the channel-pass kernel (round-2 path)
  LR_SW_WARP_GENERAL,   // LIFTREG_WARP_GENERAL: trilinear warp / its gradient: the general kernels instead of the fast ones (same bits)
  LR_SW_DRR_GENERAL,   // LIFTREG_DRR_GENERAL: projector: the general kernel instead of the fast one (same bits)
  LR_SW_REG_NOMARCH,   // LIFTREG_REG_NOMARCH: displacement regulariser: the generic kernels instead of the marching ones
  LR_SW_DGRAD_OLD,   // LIFTREG_DGRAD_OLD: data gradient: the per-tile kernels instead of the persistent weights-in-LDS ones (tests cross-check both)
  LR_SW_WGRAD_SPLIT,   // LIFTREG_WGRAD_SPLIT=0: block 1's weight gradient on the fp32 MFMA (default: exact 3-way bf16 splits)
  LR_SW_WGRAD_ROWS,   // LIFTREG_WGRAD_ROWS: weight gradient: bricks of 1 instead of 2 rows
  LR_SW_WGRAD0_COPIES,   // LIFTREG_WGRAD0_COPIES: bf16 training: first block's weight gradient through the three-copies kernel
  LR_SW_CONV0_BLOCKS,   // LIFTREG_CONV0_BLOCKS: persistent blocks of the fp32 first-block kernels
  LR_SW_CONV0_SPLIT_BLOCKS,   // LIFTREG_CONV0_SPLIT_BLOCKS: persistent blocks of conv0_split_f32.hip
  LR_SW_CONV0_SPLIT_CHUNKS,   // LIFTREG_CONV0_SPLIT_CHUNKS: z chunks per column of conv0_split_f32.hip (tests: chunk boundaries)
  LR_SW_CONV0_CL_BLOCKS,   // LIFTREG_CONV0_CL_BLOCKS: persistent blocks of conv0_cl_bf16.hip
  LR_SW_C0CL_SHAPE,   // LIFTREG_C0CL_SHAPE: brick shape of conv0_cl_bf16.hip
  LR_SW_C0CL_CHUNKS,   // LIFTREG_C0CL_CHUNKS: z chunks per column of conv0_cl_bf16.hip
  LR_SW_CONV_LDS,   // LIFTREG_CONV_LDS: dynamic LDS bytes that cap the resident blocks of the channels-last conv kernels
  LR_SW_CONV_ROWS_MT1_BELOW,   // LIFTREG_CONV_ROWS_MT1_BELOW: block count below which the 32->32 blocks take one output row per wave
  LR_SW_CONV_ROWS_BLOCKS,   // LIFTREG_CONV_ROWS_BLOCKS: persistent blocks of conv3d_rows.hip
  LR_SW_CONV_ROWS_XMAP,   // LIFTREG_CONV_ROWS_XMAP: 0: plain strided tile order instead of the XCD-aware one
  LR_SW_BF16_MT,   // LIFTREG_BF16_MT: output rows per tile of the bf16 row kernels (4 | 8)
  LR_SW_PAIR01_BLOCKS,   // LIFTREG_PAIR01_BLOCKS: persistent blocks of the fused pair kernel (default: one per CU)
  LR_SW_PAIR01_DENSE,   // LIFTREG_PAIR01_DENSE: 0: three-channel pair kernel with the padded K of block 0 (24 MFMAs per tile; A/B aid) instead of the dense 17
  LR_SW_BF16_NO_MARCH,   // LIFTREG_BF16_NO_MARCH: bf16 16->32 block: the row kernel instead of the z-marching one (A/B aid)
  LR_SW_BF16_MARCH_TY8,   // LIFTREG_BF16_MARCH_TY8: bf16 16->32 z-march: columns of 8 x 16 outputs (512 threads, one block per CU; A/B aid)
  LR_SW_BF16_MARCH_ZC,   // LIFTREG_BF16_MARCH_ZC: output planes per z chunk of the bf16 z-marching kernel (tests: chunk boundaries)
  LR_SW_DGRAD_BLOCKS,   // LIFTREG_DGRAD_BLOCKS: persistent blocks of the data-gradient kernels
  LR_SW_FUSED_BWD_BLOCKS,   // LIFTREG_FUSED_BWD_BLOCKS: persistent blocks of the fused dgrad1 + wgrad0 kernel
  LR_SW_REG_BWD_BLOCKS,   // LIFTREG_REG_BWD_BLOCKS: block cap of the regulariser's gradient kernel
  LR_SW_FUSED_BWD_NZ,   // LIFTREG_FUSED_BWD_NZ: 4: the fused dgrad1 + wgrad0 kernel's 4-plane tile form (two waves per quotient plane) for <= 3 input channels too (default 8; 4 / 5 channels always 4)
  LR_SW_BP_TOUCH,   // LIFTREG_BP_TOUCH: 0: no streaming pass over the views in front of the tiled backprojection (default 1)
  LR_SW_BP_CHUNK,   // LIFTREG_BP_CHUNK: batch elements per block of the tiled backprojection (default: chosen from the grid size; 0 = the whole batch)
  LR_SW_BP_JP,   // LIFTREG_BP_JP: planes a block of the tiled backprojection works on side by side (1 | 2 | 4; default: by row length)
#ifdef LR_EXPERIMENTAL   // (make exp)
  LR_SW_CONV0_SPLIT,   // LIFTREG_CONV0_SPLIT: first fp32 block alone on the bf16 MFMA with exact 3-way operand splits (conv0_split_f32.hip; A/B aid — the model's default is the fused pair kernel)
  LR_SW_CONV0_PC,   // LIFTREG_CONV0_PC: first fp32 block as the producer/consumer kernel (conv0_pc.hip; same bits)
#endif
  LR_SW_COUNT
};
#define LR_SW_UNSET (-2147483647 - 1)
int lr_sw_raw(int id);   // LR_SW_UNSET, or atoi() of the variable's value
static inline bool lr_sw_set(int id) { return lr_sw_raw(id) != LR_SW_UNSET; }                          // variable present
static inline bool lr_sw_on(int id) { const int v = lr_sw_raw(id); return v != LR_SW_UNSET && v != 0; }  // present and non-zero
static inline int lr_sw_int(int id, int dflt) { const int v = lr_sw_raw(id); return v == LR_SW_UNSET ? dflt : v; }

// Launch check: report, never throw (C ABI).
static inline int lr_launch_status() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess && lr_sw_set(LR_SW_HIP_DEBUG))
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
