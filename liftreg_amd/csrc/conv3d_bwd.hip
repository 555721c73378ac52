// conv3d_bwd.hip — backward of convBlock (Conv3d k3 p1 + LeakyReLU) for the training step.
// The reference obtains these from ATen/cuDNN autograd (RegistrationNet.py:401); here:
//
//   lr_lrelu_bwd_f32      gpre = gy * (y > 0 ? 1 : slope), any activation layout -> plain NDHWC (the last block only:
//                         the other blocks get their mask from the consumer's data-gradient epilogue)
//   lr_conv3d_dgrad_f32   data gradient of a stride-2 block (blocks 1..5), one parity class of gx at a time so no
//                         MFMA multiplies a structural zero; conv3d_dgrad_lds_kernel stages the gpre tile in LDS
//                         (Cg in {16,32}), conv3d_dgrad_kernel loads it directly (any Cg % 4 == 0); optional fused
//                         LeakyReLU mask of the producer block (fp32 or bf16-stored activation)
//   lr_conv3d_wgrad_f32   weight (+ bias) gradient: D[cout][(tap,cin)] += gpre^T · X over voxels on the MFMA,
//                         persistent blocks, per-block partials, fixed-order double reduce (wgrad_finish_kernel):
//                           conv3d_wgrad_cl_kernel        channels-last x, stride 2 (fp32 or bf16-stored x / gpre)
//                           conv3d_wgrad_planar_kernel    the first block (planar x, stride 1)
//                           conv3d_wgrad_cl_bf16_kernel   } bf16-gradient variant: v_mfma_f32_16x16x32_bf16 with
//                           conv3d_wgrad_planar_bf16_kernel } transposing LDS reads (ds_read_b64_tr_b16)
//                           conv3d_wgrad_kernel           generic fallback (odd shapes, unaligned rows)
//   (the bf16-gradient data gradient, lr_conv3d_dgrad_bf16, lives in conv3d_bf16.hip)
//
// Replaces: autograd of src/liftreg/layers/layers.py:365-369 as wired at …Backproj.py:29-33,95-100.
#include "lr_common.h"
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOR = 0x80000000u;
typedef unsigned short u16;

// 4 bf16 (8 bytes) -> 4 floats (exact)
__device__ __forceinline__ f32x4 bf16x4_to_f32(uint2 p) {
  f32x4 v;
  v[0] = __builtin_bit_cast(float, p.x << 16);
  v[1] = __builtin_bit_cast(float, p.x & 0xffff0000u);
  v[2] = __builtin_bit_cast(float, p.y << 16);
  v[3] = __builtin_bit_cast(float, p.y & 0xffff0000u);
  return v;
}
__device__ __forceinline__ float round_bf16(float v) { return (float)(__bf16)v; }  // nearest even, like the forward

// element offset of (b, c, z, y, x) in a tensor of the given layout
__device__ __forceinline__ int64_t act_off(int layout, int b, int c, int z, int y, int x, int C, int D, int W, int H) {
  if (layout == LR_LAYOUT_NCDHW) return ((((int64_t)b * C + c) * D + z) * W + y) * H + x;
  const int64_t row = (((int64_t)b * D + z) * W + y) * H * C;
  if (layout == LR_LAYOUT_NDHWC) return row + (int64_t)x * C + c;
  const int hp = (x & 1) * (H >> 1) + (x >> 1);  // LR_LAYOUT_NDHWC_HPS: [C/16][parity][H/2][16]
  return row + ((int64_t)(c >> 4) * H + hp) * 16 + (c & 15);
}

// ------------------------------------------------------------------------------------------ LeakyReLU bwd
// One thread per (voxel, 4 channels) of the NDHWC output; per-block channel sums for the bias gradient.
__global__ __launch_bounds__(256) void lrelu_bwd_kernel(const float* __restrict__ gy, int gy_layout,
                                                        const float* __restrict__ y, int y_layout,
                                                        float* __restrict__ gpre, float* __restrict__ gb_partial,
                                                        int B, int C, int D, int W, int H, float slope) {
  __shared__ float bsum[32];
  if (threadIdx.x < 32) bsum[threadIdx.x] = 0.0f;
  __syncthreads();
  const int C4 = C >> 2;
  const int64_t total = (int64_t)B * D * W * H * C4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // blockDim (256) is a multiple of C4 (4 | 8): a thread keeps the same channel quad in every iteration
  const int c0 = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) % C4) * 4;
  float local[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    int64_t v = idx / C4;
    const int x = (int)(v % H); v /= H;
    const int yy = (int)(v % W); v /= W;
    const int z = (int)(v % D);
    const int b = (int)(v / D);
    f32x4 g;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gv = gy[act_off(gy_layout, b, c0 + r, z, yy, x, C, D, W, H)];
      const float yv = y[act_off(y_layout, b, c0 + r, z, yy, x, C, D, W, H)];
      g[r] = yv > 0.0f ? gv : gv * slope;
      local[r] += g[r];
    }
    *reinterpret_cast<f32x4*>(gpre + ((((int64_t)b * D + z) * W + yy) * H + x) * C + c0) = g;
  }
  if (gb_partial) {
#pragma unroll
    for (int r = 0; r < 4; ++r) atomicAdd(&bsum[c0 + r], local[r]);  // LDS atomics, <= 32 channels
    __syncthreads();
    if (threadIdx.x < C) gb_partial[(int64_t)blockIdx.x * C + threadIdx.x] = bsum[threadIdx.x];
  }
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                           int nblk, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int k = 0; k < nblk; ++k) s += (double)partial[(int64_t)k * n + i];
  out[i] = (float)s;
}

// ------------------------------------------------------------------------------------------ dgrad (stride 2)
// gx[b, x, ci] = sum_{tap, co} gpre[b, (x + 1 - tap)/2, co] * W[co][ci][tap]   over taps with (x+1-tap) even per axis
// GEMM view as the forward: rows = ci (the block's input channels, 16|32 -> NT tiles), cols = 16 voxels of gx,
// k = (tap, co).  A block owns ONE parity class (pz,py,px) of gx: along an axis an even coordinate takes tap 1
// only and an odd one taps 0 and 2, so the class fixes its 1|2|4|8 taps and no MFMA multiplies a structural zero.
// Its 16 columns are 16 same-parity voxels (2 apart along H; contiguous in the parity-split layout); brick per
// block: 4 planes x 4 rows x 16 voxels of the class.
// 4 channels (nt*16 + kq*4 ..) of the block's saved input at one voxel, for the LeakyReLU mask of the fused epilogue.
// x: the voxel's column, hp: its position in a parity-split row; fp32 layouts 1|2, bf16 storage 3|4.
__device__ __forceinline__ f32x4 load_mask_src(const float* xsave, int layout, int64_t row, int x, int hp, int nt, int kq,
                                               int H, int Cx) {
  const int c = nt * 16 + kq * 4;
  if (layout == LR_LAYOUT_SIGN4) {  // (B,D,W,H,Cx/4) uint8 written by the producer's forward: bit r = channel 4q+r > 0
    // row = voxel-row index * H * Cx, Cx in {16,32}: the divisions are shifts
    const int64_t vox = (Cx == 16 ? row >> 4 : row >> 5) + x;
    const unsigned m = reinterpret_cast<const unsigned char*>(xsave)[vox * (Cx >> 2) + nt * 4 + kq];
    return (f32x4){(float)(m & 1u), (float)((m >> 1) & 1u), (float)((m >> 2) & 1u), (float)((m >> 3) & 1u)};
  }
  if (layout == LR_LAYOUT_NDHWC) return *reinterpret_cast<const f32x4*>(xsave + row + (int64_t)x * Cx + c);
  if (layout == LR_LAYOUT_NDHWC_HPS) return *reinterpret_cast<const f32x4*>(xsave + row + ((int64_t)nt * H + hp) * 16 + kq * 4);
  const u16* xb = reinterpret_cast<const u16*>(xsave);  // bf16: rows [H][C] or [parity][H/2][C]
  const int64_t o = row + (int64_t)(layout == LR_LAYOUT_BF16_NDHWC ? x : hp) * Cx + c;
  return bf16x4_to_f32(*reinterpret_cast<const uint2*>(xb + o));
}

struct DgDims {
  int B, Cg, Cx, D, W, H, Do, Wo, Ho;  // gx is (B,D,W,H,Cx); gpre is (B,Do,Wo,Ho,Cg)
  int nHq, nWq, nDq;
  int gx_layout;  // LR_LAYOUT_NDHWC or LR_LAYOUT_NDHWC_HPS (the layout of the block's saved input)
  int xs_layout;  // layout of xsave (the block's saved input = the producer block's activation)
  float slope;    // the producer's LeakyReLU slope (used when xsave != nullptr)
};
constexpr int DMT = 4;

template <int NT>
__global__ __launch_bounds__(256) void conv3d_dgrad_kernel(const float* __restrict__ gpre,
                                                           const float4* __restrict__ wp,
                                                           float* __restrict__ gx,
                                                           const float* __restrict__ xsave, DgDims d) {
  // a block walks the 8 parity classes of its tile back to back (they read the same gpre tile and together fill
  // one 8 x 8 x 32 brick of gx), the 8-tap class first
  unsigned t = lr_xcd_remap(blockIdx.x, gridDim.x);
  const int hq = t % d.nHq; t /= d.nHq;
  const int wq = t % d.nWq; t /= d.nWq;
  const int dq = t % d.nDq;
  const int b = t / d.nDq;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int zq = dq * 4 + wave;
  const int yq0 = wq * DMT;
  const int col = lane & 15, kq = lane >> 4;
  const int xq = hq * 16 + col;
  const int CB = (d.Cg + 15) >> 4;
  const float* base = gpre + (int64_t)b * d.Do * d.Wo * d.Ho * d.Cg;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, 0x7fffffff, 0x00020000);
  const unsigned lv = (unsigned)((xq * d.Cg + kq * 4) * 4);
  float4 a0[DMT], a1[DMT], b0[NT], b1[NT];
  for (int cls = 7; cls >= 0; --cls) {
  const int px = cls & 1, py = (cls >> 1) & 1, pz = cls >> 2;
  const int z = 2 * zq + pz, x = 2 * xq + px;
  if (z >= d.D) continue;  // wave-uniform
  f32x4 acc[DMT][NT];
#pragma unroll
  for (int mt = 0; mt < DMT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // lane mask per source column shift ox: OOR when the gx voxel or its source column does not exist
  const unsigned xinv0 = (x < d.H && xq < d.Ho) ? 0u : OOR;
  const unsigned xinv1 = (x < d.H && xq + 1 < d.Ho) ? 0u : OOR;

  // the producer's activation at this class's voxels (mask of the fused epilogue), in flight during the k-loop
  f32x4 xv[DMT][NT];
  if (xsave) {
#pragma unroll
    for (int mt = 0; mt < DMT; ++mt) {
      const int y = 2 * (yq0 + mt) + py;
      const bool ok = x < d.H && y < d.W;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int64_t row = (((int64_t)b * d.D + z) * d.W + (ok ? y : 0)) * d.H * d.Cx;
        xv[mt][nt] = load_mask_src(xsave, d.xs_layout, row, ok ? x : 0, ok ? px * (d.H >> 1) + xq : 0, nt, kq, d.H, d.Cx);
      }
    }
  }
  const int NS = (1 << (px + py + pz)) * CB;
  auto load_step = [&](int s, float4 (&a)[DMT], float4 (&bw)[NT]) {
    const int tapi = s / CB, cb = s - tapi * CB;
    // class-local tap -> (tap index of the 3x3x3 kernel, source shift): parity 0: tap 1, shift 0;
    // parity 1: taps 0 (source q+1) and 2 (source q)
    const int ix = tapi & px, r1 = tapi >> px, iy = r1 & py, iz = (r1 >> py) & pz;
    const int tx = px ? 2 * ix : 1, ox = px ? 1 - ix : 0;
    const int ty = py ? 2 * iy : 1, oy = py ? 1 - iy : 0;
    const int tz = pz ? 2 * iz : 1, oz = pz ? 1 - iz : 0;
    const int sfull = ((tz * 3 + ty) * 3 + tx) * CB + cb;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bw[nt] = wp[((int64_t)sfull * NT + nt) * 64 + lane];
    const unsigned coor = (cb * 16 + kq * 4 < d.Cg) ? 0u : OOR;
    const unsigned lvx = lv | coor | (ox ? xinv1 : xinv0);
    const int zs = zq + oz;
#pragma unroll
    for (int mt = 0; mt < DMT; ++mt) {
      const int ys = yq0 + mt + oy;
      const bool ok = zs < d.Do && ys < d.Wo && 2 * (yq0 + mt) + py < d.W;  // wave-uniform
      const unsigned soff = ok ? (unsigned)(((((zs * d.Wo) + ys) * d.Ho + ox) * d.Cg + cb * 16) * 4) : 0u;
      a[mt] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lvx | (ok ? 0u : OOR), soff, 0));
    }
  };
  auto mfma_step = [&](const float4 (&a)[DMT], const float4 (&bw)[NT]) {
#pragma unroll
    for (int mt = 0; mt < DMT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[nt].x, a[mt].x, acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[nt].y, a[mt].y, acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[nt].z, a[mt].z, acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[nt].w, a[mt].w, acc[mt][nt], 0, 0, 0);
      }
  };
  load_step(0, a0, b0);
  for (int s = 0; s + 1 < NS; s += 2) {
    load_step(s + 1, a1, b1);
    mfma_step(a0, b0);
    load_step(min(s + 2, NS - 1), a0, b0);
    mfma_step(a1, b1);
  }
  if (NS & 1) mfma_step(a0, b0);
  if (x < d.H) {
#pragma unroll
    for (int mt = 0; mt < DMT; ++mt) {
      const int y = 2 * (yq0 + mt) + py;
      if (y < d.W)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int64_t row = (((int64_t)b * d.D + z) * d.W + y) * d.H * d.Cx;
          const int64_t o = d.gx_layout == LR_LAYOUT_NDHWC
                                ? row + (int64_t)x * d.Cx + nt * 16 + kq * 4
                                : row + ((int64_t)nt * d.H + (px * (d.H >> 1) + xq)) * 16 + kq * 4;
          f32x4 v = acc[mt][nt];
          if (xsave) {  // gradient through the producer's LeakyReLU: its output is this block's saved input
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = xv[mt][nt][r] > 0.0f ? v[r] : v[r] * d.slope;
          }
          *reinterpret_cast<f32x4*>(gx + o) = v;
        }
    }
  }
  }  // parity classes
}

// LDS-staged variant (Cg in {16,32}): the (4+1) x (4+1) x (16+1) gpre voxels a tile needs are loaded ONCE
// (bounds-checked: rows/columns past the gradient read 0) and every (class, tap) operand is a conflict-free
// ds_read_b128 — the direct-load kernel above re-fetches each gpre voxel up to 27 times through the L1.
// The two classes that differ in px are computed back to back and stored together, so a plain-NDHWC gx receives
// both 64-byte halves of a 128-byte line from the same wave.
template <int NT, int CB>
__global__ __launch_bounds__(256, 2) void conv3d_dgrad_lds_kernel(const float* __restrict__ gpre,
                                                                  const float4* __restrict__ wp,
                                                                  float* __restrict__ gx,
                                                                  const float* __restrict__ xsave, DgDims d) {
  constexpr int CG = CB * 16, VS = CG + 4, C4 = CG / 4, NVOX = 5 * 5 * 17, NCH = NVOX * C4;
  constexpr int NIT = (NCH + 255) / 256;
  __shared__ __attribute__((aligned(16))) float ts[NVOX * VS];
  unsigned t = lr_xcd_remap(blockIdx.x, gridDim.x);
  const int hq = t % d.nHq; t /= d.nHq;
  const int wq = t % d.nWq; t /= d.nWq;
  const int dq = t % d.nDq;
  const int b = t / d.nDq;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int zq0 = dq * 4, yq0 = wq * DMT, xq0 = hq * 16;
  {
    // resource = the tile's own origin: offsets stay inside five planes whatever the volume size
    const float* base = gpre + ((((int64_t)b * d.Do + zq0) * d.Wo + yq0) * d.Ho + xq0) * CG;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, 0x7fffffff, 0x00020000);
    float4 st[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = it * 256 + tid;
      const int vox = q / C4, c4 = q % C4;
      const int xx = vox % 17, r = vox / 17, yy = r % 5, zz = r / 5;
      const int zs = zq0 + zz, ys = yq0 + yy, xs = xq0 + xx;
      const bool ok = q < NCH && zs < d.Do && ys < d.Wo && xs < d.Ho;
      const unsigned voff = ok ? (unsigned)(((((zz * d.Wo) + yy) * d.Ho + xx) * CG + c4 * 4) * 4) : OOR;
      st[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = it * 256 + tid;
      if (q < NCH) *reinterpret_cast<float4*>(ts + (q / C4) * VS + (q % C4) * 4) = st[it];
    }
  }
  __syncthreads();
  const int zq = zq0 + wave;
  const int col = lane & 15, kq = lane >> 4;
  const int xq = xq0 + col;
  const float* lts = ts + col * VS + kq * 4;
  float4 a0[DMT], a1[DMT], b0[NT], b1[NT];
  for (int pp = 3; pp >= 0; --pp) {
    const int py = pp & 1, pz = pp >> 1;
    const int z = 2 * zq + pz;
    if (z >= d.D) continue;  // wave-uniform; no barrier below
    f32x4 accp[2][DMT][NT];
    f32x4 xv[2][DMT][NT];
#pragma unroll
    for (int px = 1; px >= 0; --px) {
      const int x = 2 * xq + px;
      if (xsave) {
#pragma unroll
        for (int mt = 0; mt < DMT; ++mt) {
          const int y = 2 * (yq0 + mt) + py;
          const bool ok = x < d.H && y < d.W;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int64_t row = (((int64_t)b * d.D + z) * d.W + (ok ? y : 0)) * d.H * d.Cx;
            xv[px][mt][nt] = load_mask_src(xsave, d.xs_layout, row, ok ? x : 0, ok ? px * (d.H >> 1) + xq : 0, nt, kq, d.H, d.Cx);
          }
        }
      }
#pragma unroll
      for (int mt = 0; mt < DMT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) accp[px][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int NS = (1 << (px + py + pz)) * CB;
      auto load_step = [&](int s, float4 (&a)[DMT], float4 (&bw)[NT]) {
        const int tapi = s / CB, cb = s - tapi * CB;
        const int ix = tapi & px, r1 = tapi >> px, iy = r1 & py, iz = (r1 >> py) & pz;
        const int tx = px ? 2 * ix : 1, ox = px ? 1 - ix : 0;
        const int ty = py ? 2 * iy : 1, oy = py ? 1 - iy : 0;
        const int tz = pz ? 2 * iz : 1, oz = pz ? 1 - iz : 0;
        const int sfull = ((tz * 3 + ty) * 3 + tx) * CB + cb;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bw[nt] = wp[((int64_t)sfull * NT + nt) * 64 + lane];
        const float* src = lts + (((wave + oz) * 5 + oy) * 17 + ox) * VS + cb * 16;
#pragma unroll
        for (int mt = 0; mt < DMT; ++mt) a[mt] = *reinterpret_cast<const float4*>(src + mt * 17 * VS);
      };
      auto mfma_step = [&](const float4 (&a)[DMT], const float4 (&bw)[NT]) {
#pragma unroll
        for (int mt = 0; mt < DMT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            f32x4 c = accp[px][mt][nt];
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[nt].x, a[mt].x, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[nt].y, a[mt].y, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[nt].z, a[mt].z, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[nt].w, a[mt].w, c, 0, 0, 0);
            accp[px][mt][nt] = c;
          }
      };
      load_step(0, a0, b0);
      for (int s2 = 0; s2 + 1 < NS; s2 += 2) {
        load_step(s2 + 1, a1, b1);
        mfma_step(a0, b0);
        load_step(min(s2 + 2, NS - 1), a0, b0);
        mfma_step(a1, b1);
      }
      if (NS & 1) mfma_step(a0, b0);
    }
    // store both px classes of this (pz, py) together
#pragma unroll
    for (int mt = 0; mt < DMT; ++mt) {
      const int y = 2 * (yq0 + mt) + py;
      if (y < d.W) {
        const int64_t row = (((int64_t)b * d.D + z) * d.W + y) * d.H * d.Cx;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            const int x = 2 * xq + px;
            if (x < d.H) {
              const int64_t o = d.gx_layout == LR_LAYOUT_NDHWC
                                    ? row + (int64_t)x * d.Cx + nt * 16 + kq * 4
                                    : row + ((int64_t)nt * d.H + (px * (d.H >> 1) + xq)) * 16 + kq * 4;
              f32x4 v = accp[px][mt][nt];
              if (xsave) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = xv[px][mt][nt][r] > 0.0f ? v[r] : v[r] * d.slope;
              }
              *reinterpret_cast<f32x4*>(gx + o) = v;
            }
          }
      }
    }
  }
}

// Weights-in-LDS variant for the 16-channel data gradient (NT = 1: the encoder's block 1, 464 of the step's GFLOP).
// conv3d_dgrad_lds_kernel fetches its weight fragments from global memory inside the k-loop, one step ahead: a step is
// 16 MFMAs (512 cycles), an L2 hit under load is longer, the loop restarts for each of the 8 parity classes (2..16 steps
// each), and with 61 KB of LDS per block only two waves share a SIMD — the matrix pipe idled half the time (43 % of the
// fp32 MFMA peak).  Here ALL packed weights of the layer live in LDS for the life of a PERSISTENT block (27*CB KiB,
// loaded once), a block is 8 wavefronts = 8 quotient planes of one 8 x 2 x 16 tile (two waves per SIMD around ONE tile
// and ONE weight copy: 54 + 66 KB), the k-loop touches LDS only (three ds_read_b128 per 8 MFMAs), and the next tile's
// gpre voxels are already in flight (bounds-checked buffer loads into registers) while the current one is computed.
// Same parity-class scheme, same tap/channel order per accumulator as conv3d_dgrad_lds_kernel -> same bits.
// MK = how the producer's LeakyReLU mask arrives (compile time: a runtime switch around the mask loads makes hipcc drain
// the memory queue at every join — each load then costs a full latency with the matrix pipe idle): 0 none, 1 the fp32
// activation in plain NDHWC, 2 in NDHWC_HPS, 3 the LR_LAYOUT_SIGN4 byte mask (kept RAW until the epilogue).
constexpr int WMT = 2;  // quotient rows per wave
template <int CB, int MK>
__global__ __launch_bounds__(512, 2) void conv3d_dgrad_wlds_kernel(const float* __restrict__ gpre,
                                                                   const float4* __restrict__ wp,
                                                                   float* __restrict__ gx,
                                                                   const float* __restrict__ xsave, DgDims d, int ntiles, int dbg) {
  constexpr int CG = CB * 16, VS = CG + 4, C4 = CG / 4, NVOX = 9 * (WMT + 1) * 17, NCH = NVOX * C4;
  constexpr int NIT = (NCH + 511) / 512;
  constexpr int NWF = 27 * CB * 64;  // float4 weight fragments
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  float4* wl = reinterpret_cast<float4*>(dsm);           // [27][CB][64 lanes]
  float* ts = dsm + NWF * 4;                              // [9][WMT+1][17][VS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < NWF; i += 512) wl[i] = wp[i];
  const int nWq = (d.Wo + WMT - 1) / WMT, nDq = (d.Do + 7) / 8;
  auto tile_coords = [&](int t, int& b, int& zq0, int& yq0, int& xq0) {
    const int hq = t % d.nHq; t /= d.nHq;
    const int wq = t % nWq; t /= nWq;
    const int dq = t % nDq;
    b = t / nDq;
    zq0 = dq * 8; yq0 = wq * WMT; xq0 = hq * 16;
  };
  float4 st[NIT];
  auto prefetch = [&](int t) {
    int b, zq0, yq0, xq0;
    tile_coords(t, b, zq0, yq0, xq0);
    // resource = the tile's own origin: offsets stay inside nine planes whatever the volume size
    const float* base = gpre + ((((int64_t)b * d.Do + zq0) * d.Wo + yq0) * d.Ho + xq0) * CG;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = it * 512 + tid;
      const int vox = q / C4, c4 = q % C4;
      const int xx = vox % 17, r = vox / 17, yy = r % (WMT + 1), zz = r / (WMT + 1);
      const bool ok = (int)(q < NCH) & (int)(zq0 + zz < d.Do) & (int)(yq0 + yy < d.Wo) & (int)(xq0 + xx < d.Ho);  // bitwise: no exec-mask branches
      const unsigned voff = ok ? (unsigned)(((((zz * d.Wo) + yy) * d.Ho + xx) * CG + c4 * 4) * 4) : OOR;
      st[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
    }
  };
  const int col = lane & 15, kq = lane >> 4;
  const float* lts = ts + col * VS + kq * 4;
  int t = (int)blockIdx.x;
  if (t < ntiles) prefetch(t);
  for (; t < ntiles; t += (int)gridDim.x) {
    __syncthreads();  // every wave is done with the previous tile (first pass: the weights are in LDS after the next one)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = it * 512 + tid;
      if (q < NCH) *reinterpret_cast<float4*>(ts + (q / C4) * VS + (q % C4) * 4) = st[it];
    }
    __syncthreads();
    int b, zq0, yq0, xq0;
    tile_coords(t, b, zq0, yq0, xq0);
    if (t + (int)gridDim.x < ntiles) prefetch(t + (int)gridDim.x);  // lands while this tile runs on the matrix pipe
    const int zq = zq0 + wave, xq = xq0 + col;
    float4 a0[WMT], a1[WMT], b0, b1;
    for (int pp = 3; pp >= 0; --pp) {
      const int py = pp & 1, pz = pp >> 1;
      const int z = 2 * zq + pz;
      if (z >= d.D) continue;  // wave-uniform; no barrier inside the class loop
      f32x4 accp[2][WMT];
      f32x4 xv[2][WMT];        // MK 1|2: the producer's activation at this lane's voxel (4 channels)
      unsigned mraw[2][WMT];   // MK 3: its sign byte
      // all mask loads of the class pair first, unconditionally (clamped addresses), a whole pair ahead of their use
      if constexpr (MK != 0) {
#pragma unroll
        for (int px = 1; px >= 0; --px)
#pragma unroll
          for (int mt = 0; mt < WMT; ++mt) {
            const int x = 2 * xq + px, y = 2 * (yq0 + mt) + py;
            const bool ok = x < d.H && y < d.W;
            const int64_t rowv = (((int64_t)b * d.D + z) * d.W + (ok ? y : 0)) * d.H;  // voxel index of the row's start
            if constexpr (MK == 1) xv[px][mt] = *reinterpret_cast<const f32x4*>(xsave + (rowv + (ok ? x : 0)) * 16 + kq * 4);
            if constexpr (MK == 2) xv[px][mt] = *reinterpret_cast<const f32x4*>(xsave + (rowv + (ok ? px * (d.H >> 1) + xq : 0)) * 16 + kq * 4);
            if constexpr (MK == 3) mraw[px][mt] = reinterpret_cast<const unsigned char*>(xsave)[(rowv + (ok ? x : 0)) * 4 + kq];
          }
        __builtin_amdgcn_sched_barrier(0);  // the scheduler otherwise sinks these loads to the store epilogue
      }
#pragma unroll
      for (int px = 1; px >= 0; --px) {
        const int x = 2 * xq + px;
        (void)x;
#pragma unroll
        for (int mt = 0; mt < WMT; ++mt) accp[px][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int NS = (1 << (px + py + pz)) * CB;
        auto load_step = [&](int s, float4 (&a)[WMT], float4& bw) {
          const int tapi = s / CB, cb = s - tapi * CB;
          const int ix = tapi & px, r1 = tapi >> px, iy = r1 & py, iz = (r1 >> py) & pz;
          const int tx = px ? 2 * ix : 1, ox = px ? 1 - ix : 0;
          const int ty = py ? 2 * iy : 1, oy = py ? 1 - iy : 0;
          const int tz = pz ? 2 * iz : 1, oz = pz ? 1 - iz : 0;
          const int sfull = ((tz * 3 + ty) * 3 + tx) * CB + cb;
          bw = wl[sfull * 64 + lane];
          const float* src = lts + (((wave + oz) * (WMT + 1) + oy) * 17 + ox) * VS + cb * 16;
#pragma unroll
          for (int mt = 0; mt < WMT; ++mt) a[mt] = *reinterpret_cast<const float4*>(src + mt * 17 * VS);
        };
        auto mfma_step = [&](const float4 (&a)[WMT], const float4& bw) {
#pragma unroll
          for (int mt = 0; mt < WMT; ++mt) {
            f32x4 c = accp[px][mt];
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(bw.x, a[mt].x, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(bw.y, a[mt].y, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(bw.z, a[mt].z, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(bw.w, a[mt].w, c, 0, 0, 0);
            accp[px][mt] = c;
          }
        };
        load_step(0, a0, b0);
        for (int s2 = 0; s2 + 1 < NS; s2 += 2) {
          load_step(s2 + 1, a1, b1);
          mfma_step(a0, b0);
          load_step(min(s2 + 2, NS - 1), a0, b0);
          mfma_step(a1, b1);
        }
        if (NS & 1) mfma_step(a0, b0);
      }
      // Mask first, for all four tiles of the pair, THEN the stores: with a mask load still pending when a store is issued,
      // hipcc (loads and stores share one wait counter, which it treats as unordered) makes every later use of a mask
      // wait for the just-issued stores as well — one store latency per class pair with the matrix pipe idle.
#pragma unroll
      for (int mt = 0; mt < WMT; ++mt)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          f32x4 v = accp[px][mt];
          if constexpr (MK == 1 || MK == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = xv[px][mt][r] > 0.0f ? v[r] : v[r] * d.slope;
          }
          if constexpr (MK == 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = ((mraw[px][mt] >> r) & 1u) ? v[r] : v[r] * d.slope;
          }
          accp[px][mt] = v;
        }
      __builtin_amdgcn_sched_barrier(0);
      // store both px classes of this (pz, py) together
#pragma unroll
      for (int mt = 0; mt < WMT; ++mt) {
        const int y = 2 * (yq0 + mt) + py;
        if (y < d.W) {
          const int64_t row = (((int64_t)b * d.D + z) * d.W + y) * d.H * d.Cx;
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            const int x = 2 * xq + px;
            if (x < d.H && !(dbg & 2)) {
              const int64_t o = d.gx_layout == LR_LAYOUT_NDHWC ? row + (int64_t)x * d.Cx + kq * 4
                                                               : row + (int64_t)(px * (d.H >> 1) + xq) * 16 + kq * 4;
              *reinterpret_cast<f32x4*>(gx + o) = accp[px][mt];
            }
          }
        }
      }
    }
  }
}

// The same persistent weights-in-LDS scheme for the 32 -> 32 blocks (gx has 32 channels: blocks 2..5, 116 GFLOP at block 2):
// 27 taps x 2 channel blocks x 2 gx tiles of fragments = 108 KB of LDS, which leaves room for a window of ONE quotient row
// (9 x 2 x 17 voxels x 36 floats = 43 KB): a tile is 8 x 1 x 16 quotient voxels, a wave again one quotient plane; per step
// one window read feeds the two gx tiles (8 MFMAs per 3 LDS reads, as above), the two tiles alternate on the matrix pipe.
// conv3d_dgrad_lds_kernel<2,2> (fragments from global memory inside the k-loop, 2 x 16 KB of LDS per block) ran block 2 at
// 43 % of the fp32 MFMA peak.  MK: 0 no mask, 1 the producer's fp32 activation in NDHWC, 2 in NDHWC_HPS.
template <int MK>
__global__ __launch_bounds__(512, 1) void conv3d_dgrad_wlds32_kernel(const float* __restrict__ gpre,
                                                                     const float4* __restrict__ wp, float* __restrict__ gx,
                                                                     const float* __restrict__ xsave, DgDims d, int ntiles) {
  constexpr int CBv = 2, NT = 2, CG = 32, VS = CG + 4, C4 = CG / 4, NVOX = 9 * 2 * 17, NCH = NVOX * C4;
  constexpr int NIT = (NCH + 511) / 512;
  constexpr int NWF = 27 * CBv * NT * 64;  // float4 weight fragments
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  float4* wl = reinterpret_cast<float4*>(dsm);  // [27][CB][NT][64 lanes]
  float* ts = dsm + NWF * 4;                     // [9][2][17][VS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    constexpr int K = (NWF + 511) / 512;  // all of a thread's fragment loads in flight at once
    float4 t4[K];
#pragma unroll
    for (int k = 0; k < K; ++k) t4[k] = wp[min(tid + k * 512, NWF - 1)];
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (tid + k * 512 < NWF) wl[tid + k * 512] = t4[k];
  }
  const int nDq = (d.Do + 7) / 8;
  auto tile_coords = [&](int t, int& b, int& zq0, int& yq, int& xq0) {
    const int hq = t % d.nHq; t /= d.nHq;
    yq = t % d.Wo; t /= d.Wo;
    const int dq = t % nDq;
    b = t / nDq;
    zq0 = dq * 8; xq0 = hq * 16;
  };
  float4 st[NIT];
  auto prefetch = [&](int t) {
    int b, zq0, yq, xq0;
    tile_coords(t, b, zq0, yq, xq0);
    const float* base = gpre + ((((int64_t)b * d.Do + zq0) * d.Wo + yq) * d.Ho + xq0) * CG;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = it * 512 + tid;
      const int vox = q / C4, c4 = q % C4;
      const int xx = vox % 17, r = vox / 17, yy = r % 2, zz = r / 2;
      const bool ok = (int)(q < NCH) & (int)(zq0 + zz < d.Do) & (int)(yq + yy < d.Wo) & (int)(xq0 + xx < d.Ho);  // bitwise: no exec-mask branches
      const unsigned voff = ok ? (unsigned)(((((zz * d.Wo) + yy) * d.Ho + xx) * CG + c4 * 4) * 4) : OOR;
      st[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
    }
  };
  const int col = lane & 15, kq = lane >> 4;
  const float* lts = ts + col * VS + kq * 4;
  int t = (int)blockIdx.x;
  if (t < ntiles) prefetch(t);
  for (; t < ntiles; t += (int)gridDim.x) {
    __syncthreads();  // every wave is done with the previous tile (first pass: the weights are in LDS after the next one)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = it * 512 + tid;
      if (q < NCH) *reinterpret_cast<float4*>(ts + (q / C4) * VS + (q % C4) * 4) = st[it];
    }
    __syncthreads();
    int b, zq0, yq, xq0;
    tile_coords(t, b, zq0, yq, xq0);
    if (t + (int)gridDim.x < ntiles) prefetch(t + (int)gridDim.x);  // lands while this tile runs on the matrix pipe
    const int zq = zq0 + wave, xq = xq0 + col;
    float4 a0, a1, b0[NT], b1[NT];
    for (int pp = 3; pp >= 0; --pp) {
      const int py = pp & 1, pz = pp >> 1;
      const int z = 2 * zq + pz, y = 2 * yq + py;
      if (z >= d.D || y >= d.W) continue;  // wave-uniform; no barrier inside the class loop
      f32x4 accp[2][NT];
      f32x4 xv[2][NT];  // the producer's activation at this lane's voxel (4 channels per gx tile)
      const int64_t rowv = (((int64_t)b * d.D + z) * d.W + y) * d.H;  // voxel index of the row's start
      if constexpr (MK != 0) {
#pragma unroll
        for (int px = 1; px >= 0; --px)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int x = 2 * xq + px;
            const bool ok = x < d.H;
            if constexpr (MK == 1) xv[px][nt] = *reinterpret_cast<const f32x4*>(xsave + (rowv + (ok ? x : 0)) * 32 + nt * 16 + kq * 4);
            if constexpr (MK == 2)
              xv[px][nt] = *reinterpret_cast<const f32x4*>(xsave + rowv * 32 + ((int64_t)nt * d.H + (ok ? px * (d.H >> 1) + xq : 0)) * 16 + kq * 4);
          }
        __builtin_amdgcn_sched_barrier(0);  // the scheduler otherwise sinks these loads to the store epilogue
      }
#pragma unroll
      for (int px = 1; px >= 0; --px) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) accp[px][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int NS = (1 << (px + py + pz)) * CBv;
        auto load_step = [&](int s, float4& a, float4 (&bw)[NT]) {
          const int tapi = s / CBv, cb = s - tapi * CBv;
          const int ix = tapi & px, r1 = tapi >> px, iy = r1 & py, iz = (r1 >> py) & pz;
          const int tx = px ? 2 * ix : 1, ox = px ? 1 - ix : 0;
          const int ty = py ? 2 * iy : 1, oy = py ? 1 - iy : 0;
          const int tz = pz ? 2 * iz : 1, oz = pz ? 1 - iz : 0;
          const int sfull = ((tz * 3 + ty) * 3 + tx) * CBv + cb;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bw[nt] = wl[(sfull * NT + nt) * 64 + lane];
          a = *reinterpret_cast<const float4*>(lts + (((wave + oz) * 2 + oy) * 17 + ox) * VS + cb * 16);
        };
        auto mfma_step = [&](const float4& a, const float4 (&bw)[NT]) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) accp[px][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[nt].x, a.x, accp[px][nt], 0, 0, 0);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) accp[px][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[nt].y, a.y, accp[px][nt], 0, 0, 0);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) accp[px][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[nt].z, a.z, accp[px][nt], 0, 0, 0);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) accp[px][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[nt].w, a.w, accp[px][nt], 0, 0, 0);
        };
        load_step(0, a0, b0);
        for (int s2 = 0; s2 + 1 < NS; s2 += 2) {
          load_step(s2 + 1, a1, b1);
          mfma_step(a0, b0);
          load_step(min(s2 + 2, NS - 1), a0, b0);
          mfma_step(a1, b1);
        }
      }
      // mask first, for all four tiles of the pair, THEN the stores (see conv3d_dgrad_wlds_kernel)
      if constexpr (MK != 0) {
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            f32x4 v = accp[px][nt];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = xv[px][nt][r] > 0.0f ? v[r] : v[r] * d.slope;
            accp[px][nt] = v;
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      const int64_t row = rowv * 32;
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        const int x = 2 * xq + px;
        if (x < d.H) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int64_t o = d.gx_layout == LR_LAYOUT_NDHWC ? row + (int64_t)x * 32 + nt * 16 + kq * 4
                                                             : row + ((int64_t)nt * d.H + (px * (d.H >> 1) + xq)) * 16 + kq * 4;
            *reinterpret_cast<f32x4*>(gx + o) = accp[px][nt];
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ wgrad
// gw[co][ci][tap] = sum_{b, o} gpre[b, o, co] * X[b, ci, s*o + tap - 1]
// MFMA: rows = co (NTC tiles of 16), cols = 16 "columns" n of an N-tile, k = 4 consecutive output voxels
// along Ho.  Column n of N-tile j stands for one (tap, ci): channels-last X: j = (tap, 16-channel block),
// n = ci in the block; planar X: j*16+n = ci*27 + tap.  Each lane carries, per N-tile it owns, the packed
// descriptor of its column (channel offset and tap); the 4 waves of a block split the N-tiles.
struct WgDims {
  int B, Cin, Cout, D, W, H, Do, Wo, Ho, stride, x_layout;
  int ntiles;  // N-tiles in total
};
constexpr int WG_MAXT = 14;  // N-tiles per wave (27 taps * 2 channel blocks / 4 waves, rounded up)

template <int NTC>
__global__ __launch_bounds__(256) void conv3d_wgrad_kernel(const float* __restrict__ xin,
                                                           const float* __restrict__ gpre,
                                                           float* __restrict__ partial, WgDims d, int64_t ngroups,
                                                           int round_x /* x rounded to bf16 like the bf16 forward did */,
                                                           int gbf /* gpre is bf16 storage */) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 15, kq = lane >> 4;
  // this lane's column descriptor for each N-tile of the wave
  int64_t coff[WG_MAXT];
  int ctap[WG_MAXT];  // tz | ty<<2 | tx<<4 | valid<<6
  const int64_t V = (int64_t)d.D * d.W * d.H;
#pragma unroll
  for (int t = 0; t < WG_MAXT; ++t) {
    const int j = wave + 4 * t;
    int tap = 0, ci = 0;
    bool ok = j < d.ntiles;
    if (d.x_layout == LR_LAYOUT_NCDHW) {
      const int n = j * 16 + col;
      ci = n / 27; tap = n - ci * 27;
      ok = ok && ci < d.Cin;
    } else {
      const int cbn = (d.Cin + 15) >> 4;
      tap = j / cbn;
      ci = (j - tap * cbn) * 16 + col;
      ok = ok && ci < d.Cin && tap < 27;
    }
    ctap[t] = (tap / 9) | (((tap / 3) % 3) << 2) | ((tap % 3) << 4) | (ok ? 64 : 0);
    if (d.x_layout == LR_LAYOUT_NCDHW) coff[t] = (int64_t)ci * V;
    else if (d.x_layout == LR_LAYOUT_NDHWC) coff[t] = ci;
    else coff[t] = (int64_t)(ci >> 4) * d.H * 16 + (ci & 15);  // HPS: channel block offset inside a row
  }
  f32x4 acc[WG_MAXT][NTC];
#pragma unroll
  for (int t = 0; t < WG_MAXT; ++t)
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int hog = (d.Ho + 3) >> 2;  // groups of 4 output voxels per row
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    int64_t r = grp;
    const int hg = (int)(r % hog); r /= hog;
    const int wo = (int)(r % d.Wo); r /= d.Wo;
    const int dz = (int)(r % d.Do);
    const int b = (int)(r / d.Do);
    const int ho = hg * 4 + kq;
    const bool vok = ho < d.Ho;
    // A operand: gpre[b, dz, wo, ho, co]
    float a[NTC];
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt)
    {
      const int64_t go = ((((int64_t)b * d.Do + dz) * d.Wo + wo) * d.Ho + ho) * d.Cout + nt * 16 + col;
      a[nt] = !vok ? 0.0f
                   : gbf ? __builtin_bit_cast(float, (unsigned)reinterpret_cast<const u16*>(gpre)[go] << 16) : gpre[go];
    }
    const int zi0 = dz * d.stride - 1, yi0 = wo * d.stride - 1, xi0 = ho * d.stride - 1;
    const int64_t xb = (int64_t)b * d.Cin * V;
#pragma unroll
    for (int t = 0; t < WG_MAXT; ++t) {
      if (wave + 4 * t >= d.ntiles) break;  // wave-uniform: this wave owns no further N-tile
      const int zi = zi0 + (ctap[t] & 3), yi = yi0 + ((ctap[t] >> 2) & 3), xi = xi0 + ((ctap[t] >> 4) & 3);
      const bool ok = vok && (ctap[t] & 64) && zi >= 0 && zi < d.D && yi >= 0 && yi < d.W && xi >= 0 && xi < d.H;
      int64_t off;
      if (d.x_layout == LR_LAYOUT_NCDHW) off = xb + coff[t] + ((int64_t)zi * d.W + yi) * d.H + xi;
      else if (d.x_layout == LR_LAYOUT_NDHWC) off = xb + (((int64_t)zi * d.W + yi) * d.H + xi) * d.Cin + coff[t];
      else off = xb + ((int64_t)zi * d.W + yi) * d.H * d.Cin + coff[t] + (int64_t)((xi & 1) * (d.H >> 1) + (xi >> 1)) * 16;
      float xv = ok ? xin[off] : 0.0f;
      if (round_x) xv = round_bf16(xv);
#pragma unroll
      for (int nt = 0; nt < NTC; ++nt)
        acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nt], xv, acc[t][nt], 0, 0, 0);
    }
  }
  // partial[block][co][n]: lane holds co = nt*16 + kq*4 + r of column (j*16 + col)
  const int ncols = d.ntiles * 16;
#pragma unroll
  for (int t = 0; t < WG_MAXT; ++t) {
    const int j = wave + 4 * t;
    if (j < d.ntiles) {
#pragma unroll
      for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          partial[((int64_t)blockIdx.x * d.Cout + nt * 16 + kq * 4 + r) * ncols + j * 16 + col] = acc[t][nt][r];
    }
  }
}

// ------------------------------------------------------------------- wgrad, channels-last X, stride 2 (blocks 1..5)
// Same GEMM view (rows = co, cols = (tap, 16-channel block), k = 4 consecutive output voxels along Ho), but the
// operands come from LDS: a persistent block walks bricks = one output row segment of HB voxels, stages the
// 3x3x(2HB+1) input window (bounds-checked buffer loads: the zero halo is the conv's padding) and the HB x Cout
// gradient segment, and the window of brick i+1 is in flight while brick i is on the matrix pipe.
// LDS window layout [pz*3+py][cb][pos][16] with the columns parity-split (even window columns first): the 4
// voxels of a k-step are then 16 floats apart for every tap — bank-conflict-free 256-byte wavefront reads.
template <int CB, int NTC, int ROWS>
struct WclGeom {
  static constexpr int HB = 32 / CB;             // output voxels per brick row
  static constexpr int NCP = 2 * HB + 1;         // window columns
  static constexpr int NR = 2 * ROWS + 1;        // window rows (y) of ROWS consecutive output rows
  static constexpr int ROWF = CB * NCP * 16;     // floats of one window row (all channel blocks)
  static constexpr int XF4 = 3 * NR * CB * NCP * 4;  // float4 chunks of the window
  static constexpr int NW = 4 * CB;              // waves per block: 27*CB N-tiles, 7 per wave
  static constexpr int NTH = NW * 64;
  static constexpr int XIT = (XF4 + NTH - 1) / NTH;
  static constexpr int GF4 = HB * NTC * 4;       // float4 chunks of ONE row's gradient segment (<= 256)
  static constexpr int T = (27 * CB + NW - 1) / NW;  // N-tiles per wave
  static constexpr int XS_FLOATS = XF4 * 4 + HB * 16;  // + a tile of ones (bias gradient)
  static constexpr int GS_FLOATS = ROWS * HB * NTC * 16;
  static constexpr size_t LDS_BYTES = (size_t)(XS_FLOATS + GS_FLOATS) * sizeof(float);
};

// XB: the saved input is bf16 storage (rows [H][C] or, HPS, [parity][H/2][C]) — 8-byte loads expanded to fp32 on the
// way into the same LDS image (the bf16-forward training variant: fp32 gradient math on the bf16-rounded activations)
// ROWS: output rows (along Wo) per brick.  Two rows share the middle window row (15 staged rows instead of 18) and halve
// the barriers and staging instructions per MFMA: with one row a brick is 112 MFMAs per wave against 10 loads, 10 LDS
// stores and two barriers (57 % of the fp32 MFMA peak on block 1's 464 GFLOP).
template <int CB, int NTC, bool HPS, bool XB, int ROWS>
__global__ __launch_bounds__(256 * CB, (CB == 1 ? 2 : 1)) void conv3d_wgrad_cl_kernel(const float* __restrict__ xin,
                                                                 const float* __restrict__ gpre,
                                                                 float* __restrict__ partial, WgDims d, int nbricks,
                                                                 int gbf /* gpre is bf16 storage */) {
  using G = WclGeom<CB, NTC, ROWS>;
  constexpr int Cin = CB * 16, Cout = NTC * 16, HB = G::HB, NR = G::NR;
  extern __shared__ __attribute__((aligned(16))) float wsm[];
  float* xs = wsm;                    // [pz*NR+py][cb][pos][16] + a tile of ones
  float* gs = wsm + G::XS_FLOATS;     // [row][nt][HB][16]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nseg = (d.Ho + HB - 1) / HB;
  const int nwo = (d.Wo + ROWS - 1) / ROWS;

  // staging slots: chunk q = it*NTH + tid of the window, decoded once (brick-independent)
  // xdec: pc | py << 8 | pz << 16 in 7-bit fields with a guard bit each, bits 24..26 = "pc / py / pz is 0", bit 27 = the slot
  // is past the window.  Per brick one add of a bias vector makes every field that is past the volume overflow into its guard
  // bit, so a chunk's validity is add + and + compare (the chain of short-circuit range tests it replaces compiled into ~20
  // instructions and two exec-mask branches per chunk, issued by all waves at once right after the barrier).
  unsigned xrel[G::XIT];
  unsigned xdec[G::XIT];
#pragma unroll
  for (int it = 0; it < G::XIT; ++it) {
    const int q = it * G::NTH + tid;
    const bool used = q < G::XF4;
    const int c4 = q & 3, pos = (q >> 2) % G::NCP, rc = (q >> 2) / G::NCP;
    const int cb = rc % CB, rowi = used ? rc / CB : 0;
    const int pz = rowi / NR, py = rowi % NR;
    const int pc = pos <= HB ? 2 * pos : 2 * (pos - HB - 1) + 1;
    xdec[it] = used ? (unsigned)(pc | (py << 8) | (pz << 16) | ((pc == 0) << 24) | ((py == 0) << 25) | ((pz == 0) << 26)) : 0x08000000u;
    const int hprel = (pc & 1) ? (pc - 1) / 2 : (d.H >> 1) - 1 + pc / 2;  // parity-split position relative to ho0
    if (XB)
      xrel[it] = (unsigned)(((((pz * d.W + py) * d.H + (HPS ? hprel : pc)) * Cin) + cb * 16 + c4 * 4) * 2);
    else if (HPS)
      xrel[it] = (unsigned)((((pz * d.W + py) * d.H * Cin) + cb * d.H * 16 + hprel * 16 + c4 * 4) * 4);
    else
      xrel[it] = (unsigned)(((((pz * d.W + py) * d.H + pc) * Cin) + cb * 16 + c4 * 4) * 4);
  }
  // N-tiles of this wave: j = wave + NW*t -> (tap, cb); LDS float offset of the tile's first voxel (output row 0)
  // (the bias gradient = column sums of gpre used to be a 28th tile multiplying by ones: a wave-uniform select per LDS read
  // of every tile — 1.4 scalar instructions per MFMA by PMC — and 1/28 of the MFMAs; it is two vector adds per k-step now)
  int boff[G::T];
  // (a wave whose last slot is past the 27*CB tiles recomputes its previous tile there and does not write it: no branch in the k-loop)
#pragma unroll
  for (int t = 0; t < G::T; ++t) {
    const int j = min(wave + G::NW * t, 27 * CB - 1);
    const int tap = j / CB, cb = j % CB;
    const int tz = tap / 9, ty = (tap / 3) % 3, tx = tap % 3;
    boff[t] = (((tz * NR + ty) * CB + cb) * G::NCP + (tx == 1 ? HB + 1 : (tx >> 1))) * 16 + lane;
  }
  float bsum[NTC];
#pragma unroll
  for (int nt = 0; nt < NTC; ++nt) bsum[nt] = 0.0f;
  f32x4 acc[G::T][NTC];
#pragma unroll
  for (int t = 0; t < G::T; ++t)
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 xst[G::XIT], gst[ROWS];
  // Brick order: dz FASTEST, and a block owns a contiguous run of it: consecutive bricks of a block are the same (wo, ho) tile
  // one output plane further — one of the three window planes was read by this CU an iteration ago.  (Dealt round-robin
  // over the blocks with ho fastest, PMC showed a TCC hit rate of 0.4 %: 18.6 GB from HBM for 10.7 GB of tensors on block 1.)
  const int per_blk = (nbricks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int run = (int)lr_xcd_remap(blockIdx.x, gridDim.x);   // neighbouring runs (the next h segment / the next output rows) on the SAME XCD: they walk dz in step and share halo rows in its L2
  const int brick_begin = min(nbricks, run * per_blk), brick_end = min(nbricks, brick_begin + per_blk);
  auto prefetch = [&](int brick) {
    const bool live = brick < brick_end;
    int r = live ? brick : 0;
    const int dz = r % d.Do; r /= d.Do;
    const int hseg = r % nseg; r /= nseg;
    const int wo = (r % nwo) * ROWS;
    const int b = r / nwo;
    const int ho0 = hseg * HB;
    const int zi0 = 2 * dz - 1, yi0 = 2 * wo - 1, xi0 = 2 * ho0 - 1;
    // resource = the window's own origin (it may lie before the tensor: never dereferenced there), so the byte
    // offsets stay inside three planes whatever the volume size
    const int64_t xorg = (int64_t)b * d.D * d.W * d.H * Cin +
                         (XB ? (((int64_t)zi0 * d.W + yi0) * d.H + (HPS ? ho0 : xi0)) * Cin
                             : HPS ? ((int64_t)zi0 * d.W + yi0) * d.H * Cin + (int64_t)ho0 * 16
                                   : (((int64_t)zi0 * d.W + yi0) * d.H + xi0) * Cin);
    const void* xb = XB ? (const void*)(reinterpret_cast<const u16*>(xin) + xorg) : (const void*)(xin + xorg);
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(xb), (short)0, 0x7fffffff, 0x00020000);
    auto bias7 = [](int rem) { return (unsigned)(128 - (rem < 0 ? 0 : rem > 128 ? 128 : rem)); };  // field >= rem -> guard bit
    const unsigned vbias = bias7(d.H - xi0) | bias7(d.W - yi0) << 8 | bias7(d.D - zi0) << 16 | (live ? 0u : 0x08000000u);
    const unsigned vmask = 0x18808080u | (unsigned)(xi0 < 0) << 24 | (unsigned)(yi0 < 0) << 25 | (unsigned)(zi0 < 0) << 26;
#pragma unroll
    for (int it = 0; it < G::XIT; ++it) {
      const bool ok = ((xdec[it] + vbias) & vmask) == 0u;
      const unsigned voff = ok ? xrel[it] : OOR;
      if (XB) {
        const f32x4 v = bf16x4_to_f32(__builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rx, voff, 0, 0)));
        xst[it] = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        xst[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, 0, 0));
      }
    }
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
      const int64_t gorg = ((((int64_t)b * d.Do + dz) * d.Wo + min(wo + rr, d.Wo - 1)) * d.Ho + ho0) * Cout;
      const void* gb = gbf ? (const void*)(reinterpret_cast<const u16*>(gpre) + gorg) : (const void*)(gpre + gorg);
      const __amdgpu_buffer_rsrc_t rg =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(gb), (short)0, 0x7fffffff, 0x00020000);
      const bool gok = (int)live & (int)(tid < G::GF4) & (int)(ho0 + tid / (NTC * 4) < d.Ho) & (int)(wo + rr < d.Wo);
      if (gbf) {
        const f32x4 v = bf16x4_to_f32(__builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rg, gok ? (unsigned)tid * 8u : OOR, 0, 0)));
        gst[rr] = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        gst[rr] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rg, gok ? (unsigned)tid * 16u : OOR, 0, 0));
      }
    }
  };

  int brick = brick_begin;
  prefetch(brick);
  for (; brick < brick_end; ++brick) {
    __syncthreads();  // the previous brick's reads are done
#pragma unroll
    for (int it = 0; it < G::XIT; ++it)
      if (it * G::NTH + tid < G::XF4) *reinterpret_cast<float4*>(xs + (it * G::NTH + tid) * 4) = xst[it];  // compile-time true but for the last slot
    if (tid < G::GF4) {
      const int i = tid / (NTC * 4), c4 = tid % (NTC * 4);
#pragma unroll
      for (int rr = 0; rr < ROWS; ++rr)
        *reinterpret_cast<float4*>(gs + rr * HB * NTC * 16 + ((c4 >> 2) * HB + i) * 16 + (c4 & 3) * 4) = gst[rr];
    }
    __syncthreads();
    prefetch(brick + 1);
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
#pragma unroll
      for (int ks = 0; ks < HB / 4; ++ks) {
        float a[NTC];
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) {
          a[nt] = gs[rr * HB * NTC * 16 + nt * HB * 16 + ks * 64 + lane];
          bsum[nt] += a[nt];  // lane = (co, voxel 4ks + kq): the bias gradient's share of this lane
        }
#pragma unroll
        for (int t = 0; t < G::T; ++t) {
          const float bv = xs[boff[t] + rr * 2 * G::ROWF + ks * 64];
#pragma unroll
          for (int nt = 0; nt < NTC; ++nt)
            acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nt], bv, acc[t][nt], 0, 0, 0);
        }
      }
    }
  }
  const int col = lane & 15, kq = lane >> 4;
  const int ncols = (27 * CB + 1) * 16;
#pragma unroll
  for (int t = 0; t < G::T; ++t) {
    const int j = wave + G::NW * t;
    if (j < 27 * CB) {
#pragma unroll
      for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          partial[((int64_t)blockIdx.x * Cout + nt * 16 + kq * 4 + r) * ncols + j * 16 + col] = acc[t][nt][r];
    }
  }
  // bias gradient: every wave summed the same gradient values (co = lane & 15 of tile nt, its own voxels kq): fold the four
  // lane groups, wave 0 writes column 27*CB*16 (= gb_col of wgrad_finish_kernel)
#pragma unroll
  for (int nt = 0; nt < NTC; ++nt) {
    float v = bsum[nt];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (wave == 0 && kq == 0) partial[((int64_t)blockIdx.x * Cout + nt * 16 + col) * ncols + 27 * CB * 16] = v;
  }
}

// ------------------------------------------------------------------- wgrad on the bf16 MFMA (bf16-gradient variant)
// x (bf16 channels-last) and gpre (bf16 plain) both enter v_mfma_f32_16x16x32_bf16 with K = 32 output voxels of one
// row segment: 8x the fp32 kernel's k per instruction at 2x its rate.  Both operands are K-STRIDED in their
// channels-last LDS images ([voxel][16 channels], 32-byte rows), which is what gfx950's transposing LDS read is
// for: ds_read_b64_tr_b16 hands lane i of a 16-lane group column (= channel) i of a 4-row (= 4-voxel) block, so two
// reads give a lane its 8 consecutive voxels of one channel.  Same bricks, same window image (parity-split columns:
// a tap's voxels are consecutive rows), same N-tile split and ones tile (bias gradient) as conv3d_wgrad_cl_kernel.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

template <int CB, int NTC, bool HPS>
__global__ __launch_bounds__(256 * CB, 2) void conv3d_wgrad_cl_bf16_kernel(const u16* __restrict__ xin,
                                                                           const u16* __restrict__ gpre,
                                                                           float* __restrict__ partial, WgDims d,
                                                                           int nbricks) {
  constexpr int Cin = CB * 16, Cout = NTC * 16, HB = 32, NCP = 2 * HB + 1, NW = 4 * CB, NTH = NW * 64;
  constexpr int T = (27 * CB + NW - 1) / NW;
  constexpr int XCH = 9 * NCP * (Cin / 8);  // 16-byte chunks of the window
  constexpr int XIT = (XCH + NTH - 1) / NTH;
  constexpr int XBYTES = 9 * CB * NCP * 32, ONES = XBYTES, GOFF = XBYTES + HB * 32;  // window | ones tile | gradient segment
  __shared__ __attribute__((aligned(16))) unsigned char lds[GOFF + NTC * HB * 32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nseg = (d.Ho + HB - 1) / HB;
  for (int i = tid; i < HB * 16; i += NTH) reinterpret_cast<u16*>(lds + ONES)[i] = 0x3f80;  // bf16 1.0
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;

  // staging slots (brick-independent decode): chunk q -> (row9, pos, c8)
  unsigned xrel[XIT], xdst[XIT];
  int xdec[XIT];  // pz | py<<2 | pc<<4 | used<<12
#pragma unroll
  for (int it = 0; it < XIT; ++it) {
    const int q = it * NTH + tid;
    const bool used = q < XCH;
    const int c8 = q % (Cin / 8), vox = q / (Cin / 8);
    const int pos = vox % NCP, row9 = used ? vox / NCP : 0;
    const int pz = row9 / 3, py = row9 % 3;
    const int pc = pos <= HB ? 2 * pos : 2 * (pos - HB - 1) + 1;
    const int hprel = (pc & 1) ? (pc - 1) / 2 : (d.H >> 1) - 1 + pc / 2;
    xdec[it] = pz | (py << 2) | (pc << 4) | (used ? 1 << 12 : 0);
    xrel[it] = (unsigned)(((((pz * d.W + py) * d.H + (HPS ? hprel : pc)) * Cin) + c8 * 8) * 2);
    xdst[it] = (unsigned)((((row9 * CB + (c8 >> 1)) * NCP + pos) * 32) + (c8 & 1) * 16);
  }
  // transposed-read lane address inside a 32-row (voxel) x 16-channel tile: lane 4q+p of group kq = lane>>4 supplies row q of
  // the group's 4-row block, columns 4p..4p+3 (8 bytes); second read: +16 rows
  // (K order: group kq owns voxels 4kq..4kq+3 and 16+4kq..16+4kq+3 — one read's 64 lanes cover 512 contiguous bytes; see
  // conv3d_wgrad_cl_split_kernel)
  const int kq = lane >> 4, lq = (lane >> 2) & 3, lp = lane & 3;
  const unsigned ltr = (unsigned)((kq * 4 + lq) * 32 + lp * 8);
  unsigned boff[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int jr = wave + NW * t;
    const int j = min(jr, 27 * CB - 1);
    const int tap = j / CB, cb = j % CB;
    const int tz = tap / 9, ty = (tap / 3) % 3, tx = tap % 3;
    boff[t] = lds0 + ltr + (jr == 27 * CB ? (unsigned)ONES
                                           : (unsigned)((((tz * 3 + ty) * CB + cb) * NCP + (tx == 1 ? HB + 1 : (tx >> 1))) * 32));
  }
  const unsigned aoff = lds0 + GOFF + ltr;
  f32x4 acc[T][NTC];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u32x4_t xst[XIT], gst;
  // brick order: dz fastest, a contiguous run per block (see conv3d_wgrad_cl_kernel: the window plane two consecutive
  // bricks share comes from the XCD's L2, not from HBM)
  const int per_blk = (nbricks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int run = (int)lr_xcd_remap(blockIdx.x, gridDim.x);   // neighbouring runs (the next h segment / the next output rows) on the SAME XCD: they walk dz in step and share halo rows in its L2
  const int brick_begin = min(nbricks, run * per_blk), brick_end = min(nbricks, brick_begin + per_blk);
  auto prefetch = [&](int brick) {
    const bool live = brick < brick_end;
    int r = live ? brick : 0;
    const int dz = r % d.Do; r /= d.Do;
    const int hseg = r % nseg; r /= nseg;
    const int wo = r % d.Wo;
    const int b = r / d.Wo;
    const int ho0 = hseg * HB;
    const int zi0 = 2 * dz - 1, yi0 = 2 * wo - 1, xi0 = 2 * ho0 - 1;
    const u16* xb = xin + (int64_t)b * d.D * d.W * d.H * Cin + (((int64_t)zi0 * d.W + yi0) * d.H + (HPS ? ho0 : xi0)) * Cin;
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(xb), (short)0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int zi = zi0 + (xdec[it] & 3), yi = yi0 + ((xdec[it] >> 2) & 3), xi = xi0 + ((xdec[it] >> 4) & 255);
      const bool ok = (int)(live) & (int)(((xdec[it] >> 12) & 1)) & (int)(zi >= 0) & (int)(zi < d.D) & (int)(yi >= 0) & (int)(yi < d.W) & (int)(xi >= 0) & (int)(xi < d.H);  // bitwise: no exec-mask branches around the loads' address math
      xst[it] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? xrel[it] : OOR, 0, 0);
    }
    const u16* gb = gpre + ((((int64_t)b * d.Do + dz) * d.Wo + wo) * d.Ho + ho0) * Cout;
    const __amdgpu_buffer_rsrc_t rg =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(gb), (short)0, 0x7fffffff, 0x00020000);
    const bool gok = live && tid < HB * (Cout / 8) && ho0 + tid / (Cout / 8) < d.Ho;
    gst = __builtin_amdgcn_raw_buffer_load_b128(rg, gok ? (unsigned)tid * 16u : OOR, 0, 0);
  };

  int brick = brick_begin;
  prefetch(brick);
  for (; brick < brick_end; ++brick) {
    __syncthreads();  // the previous brick's reads are done
#pragma unroll
    for (int it = 0; it < XIT; ++it)
      if ((xdec[it] >> 12) & 1) *reinterpret_cast<u32x4_t*>(lds + xdst[it]) = xst[it];
    if (tid < HB * (Cout / 8)) {
      const int i = tid / (Cout / 8), c8 = tid % (Cout / 8);
      *reinterpret_cast<u32x4_t*>(lds + GOFF + ((c8 >> 1) * HB + i) * 32 + (c8 & 1) * 16) = gst;
    }
    __syncthreads();
    prefetch(brick + 1);
    // operands: 2 transposed reads each (rows +0..3 and +4..7 of the lane group's 8 voxels); all lanes take part
    unsigned long long ar[NTC][2], br[T][2];
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ar[nt][h]) : "v"(aoff), "n"(nt * HB * 32 + h * 512) : "memory");
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(br[t][h]) : "v"(boff[t]), "n"(h * 512) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt) asm volatile("" : "+v"(ar[nt][0]), "+v"(ar[nt][1]));
#pragma unroll
    for (int t = 0; t < T; ++t) {
      asm volatile("" : "+v"(br[t][0]), "+v"(br[t][1]));
      const u32x4_t bv = {(unsigned)br[t][0], (unsigned)(br[t][0] >> 32), (unsigned)br[t][1], (unsigned)(br[t][1] >> 32)};
#pragma unroll
      for (int nt = 0; nt < NTC; ++nt) {
        const u32x4_t av = {(unsigned)ar[nt][0], (unsigned)(ar[nt][0] >> 32), (unsigned)ar[nt][1], (unsigned)(ar[nt][1] >> 32)};
        acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av), __builtin_bit_cast(bf16x8_t, bv),
                                                             acc[t][nt], 0, 0, 0);
      }
    }
  }
  const int col = lane & 15;
  const int ncols = (27 * CB + 1) * 16;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int j = wave + NW * t;
    if (j <= 27 * CB) {
#pragma unroll
      for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          partial[((int64_t)blockIdx.x * Cout + nt * 16 + kq * 4 + r) * ncols + j * 16 + col] = acc[t][nt][r];
    }
  }
}

// ------------------------------------------------------------------- wgrad, fp32 operands on the bf16 MFMA (exact splits)
// The fp32 weight gradient of blocks 1..5 is bound by the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: 107 TFLOP/s on block 1's
// 464 GFLOP).  Every fp32 value is EXACTLY the sum of three bf16 values (x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 -
// x1): 3 x 8 significant bits), so x·g = sum of nine bf16 products, each exact in the fp32 accumulator; the three smallest
// (x1 g2, x2 g1, x2 g2 <= 2^-26 |x g|) are below the rounding of the fp32 product itself and are dropped — the same split as
// the forward's conv0_split_f32.hip.  Six v_mfma_f32_16x16x32_bf16 (K = 32 voxels, 16 cycles) replace eight
// v_mfma_f32_16x16x4_f32 (K = 4 voxels, 32 cycles): 96 instead of 256 matrix-pipe cycles per 32 voxels and tile.
// Same bricks, window image, transposing operand reads and partial layout as conv3d_wgrad_cl_bf16_kernel; the operands are
// split on their way into LDS (three images of the window, three of the gradient segment).  The bias gradient's ones tile
// is 1.0 in the first image and 0 in the other two.
typedef float wsf32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 wsbf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void wg_split3(float a, float b, unsigned (&p)[3]) {
  wsf32x2_t v = {a, b};
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const wsbf16x2_t h = __builtin_convertvector(v, wsbf16x2_t);  // round to nearest even
    const unsigned u = __builtin_bit_cast(unsigned, h);
    p[s] = u;
    if (s < 2) {
      const wsf32x2_t f = {__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
      v = v - f;  // exact
    }
  }
}

template <int NTC, bool HPS>
__global__ __launch_bounds__(256, 2) void conv3d_wgrad_cl_split_kernel(const float* __restrict__ xin,
                                                                       const float* __restrict__ gpre,
                                                                       float* __restrict__ partial, WgDims d, int nbricks) {
  constexpr int CB = 1, Cin = 16, Cout = NTC * 16, HB = 32, NCP = 2 * HB + 1, NW = 4, NTH = 256;
  constexpr int T = (27 * CB + 1 + NW - 1) / NW;  // 27 taps + the ones tile over 4 waves
  constexpr int XCH = 9 * NCP * 2;                // 8-channel chunks of the window (two 16-byte loads each)
  constexpr int XIT = (XCH + NTH - 1) / NTH;
  constexpr int XBYTES = 9 * CB * NCP * 32, XSPL = XBYTES + HB * 32;  // one image: window | special tile (ones / zeros)
  constexpr int GOFF = 3 * XSPL, GSPL = NTC * HB * 32;
  static_assert(GOFF + 3 * GSPL <= 65536 && 2 * XSPL + 128 < 65536, "LDS image / DS offset range");
  __shared__ __attribute__((aligned(16))) unsigned char lds[GOFF + 3 * GSPL];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nseg = (d.Ho + HB - 1) / HB;
  for (int i = tid; i < 3 * HB * 16; i += NTH)
    reinterpret_cast<u16*>(lds + (i / (HB * 16)) * XSPL + XBYTES)[i % (HB * 16)] = i < HB * 16 ? 0x3f80 : 0;  // bf16 1.0 | 0
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;

  unsigned xrel[XIT], xdst[XIT];
  int xdec[XIT];  // pz | py<<2 | pc<<4 | used<<12
#pragma unroll
  for (int it = 0; it < XIT; ++it) {
    const int q = it * NTH + tid;
    const bool used = q < XCH;
    const int c8 = q & 1, vox = q >> 1;
    const int pos = vox % NCP, row9 = used ? vox / NCP : 0;
    const int pz = row9 / 3, py = row9 % 3;
    const int pc = pos <= HB ? 2 * pos : 2 * (pos - HB - 1) + 1;
    const int hprel = (pc & 1) ? (pc - 1) / 2 : (d.H >> 1) - 1 + pc / 2;
    xdec[it] = pz | (py << 2) | (pc << 4) | (used ? 1 << 12 : 0);
    xrel[it] = (unsigned)(((((pz * d.W + py) * d.H + (HPS ? hprel : pc)) * Cin) + c8 * 8) * 4);
    xdst[it] = (unsigned)(((row9 * NCP + pos) * 32) + c8 * 16);
  }
  // K order: lane group kq owns voxels 4kq..4kq+3 (first read) and 16+4kq..16+4kq+3 (second read), the same for both operands:
  // the 64 lanes of ONE read then cover 16 consecutive 32-byte rows = 512 contiguous bytes, conflict-free.  (With 8
  // consecutive voxels per group the groups sit 256 bytes apart — the same banks: PMC showed 480 conflict cycles per wave and
  // brick, the LDS pipe 70 % busy.)
  const int kq = lane >> 4, lq = (lane >> 2) & 3, lp = lane & 3;
  const unsigned ltr = (unsigned)((kq * 4 + lq) * 32 + lp * 8);
  unsigned boff[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int jr = wave + NW * t;
    const int tap = min(jr, 26);
    const int tz = tap / 9, ty = (tap / 3) % 3, tx = tap % 3;
    boff[t] = lds0 + ltr + (jr >= 27 ? (unsigned)XBYTES
                                     : (unsigned)(((tz * 3 + ty) * NCP + (tx == 1 ? HB + 1 : (tx >> 1))) * 32));
  }
  const unsigned aoff = lds0 + GOFF + ltr;
  f32x4 acc[T][NTC];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 xst[XIT][2], gst;
  const int per_blk = (nbricks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int run = (int)lr_xcd_remap(blockIdx.x, gridDim.x);
  const int brick_begin = min(nbricks, run * per_blk), brick_end = min(nbricks, brick_begin + per_blk);
  const int gi = tid / (Cout / 4), gc4 = tid % (Cout / 4);   // this thread's gradient chunk: voxel, 4-channel group
  auto prefetch = [&](int brick) {
    const bool live = brick < brick_end;
    int r = live ? brick : 0;
    const int dz = r % d.Do; r /= d.Do;
    const int hseg = r % nseg; r /= nseg;
    const int wo = r % d.Wo;
    const int b = r / d.Wo;
    const int ho0 = hseg * HB;
    const int zi0 = 2 * dz - 1, yi0 = 2 * wo - 1, xi0 = 2 * ho0 - 1;
    const float* xb = xin + (int64_t)b * d.D * d.W * d.H * Cin + (((int64_t)zi0 * d.W + yi0) * d.H + (HPS ? ho0 : xi0)) * Cin;
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), (short)0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int zi = zi0 + (xdec[it] & 3), yi = yi0 + ((xdec[it] >> 2) & 3), xi = xi0 + ((xdec[it] >> 4) & 255);
      const bool ok = (int)(live) & (int)(((xdec[it] >> 12) & 1)) & (int)(zi >= 0) & (int)(zi < d.D) & (int)(yi >= 0) & (int)(yi < d.W) & (int)(xi >= 0) & (int)(xi < d.H);
      const unsigned o = ok ? xrel[it] : OOR;
      xst[it][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, o, 0, 0));
      xst[it][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, o, 16, 0));
    }
    const float* gb = gpre + ((((int64_t)b * d.Do + dz) * d.Wo + wo) * d.Ho + ho0) * Cout;
    const __amdgpu_buffer_rsrc_t rg =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gb), (short)0, 0x7fffffff, 0x00020000);
    const bool gok = live && gi < HB && ho0 + gi < d.Ho;
    gst = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, gok ? (unsigned)tid * 16u : OOR, 0, 0));
  };

  int brick = brick_begin;
  prefetch(brick);
  for (; brick < brick_end; ++brick) {
    __syncthreads();  // the previous brick's reads are done
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      unsigned p[4][3];
#pragma unroll
      for (int j = 0; j < 4; ++j) wg_split3(xst[it][j >> 1][(j & 1) * 2], xst[it][j >> 1][(j & 1) * 2 + 1], p[j]);
      if ((xdec[it] >> 12) & 1) {
#pragma unroll
        for (int sp = 0; sp < 3; ++sp)
          *reinterpret_cast<u32x4_t*>(lds + sp * XSPL + xdst[it]) = (u32x4_t){p[0][sp], p[1][sp], p[2][sp], p[3][sp]};
      }
    }
    if (gi < HB) {
      unsigned p[2][3];
      wg_split3(gst[0], gst[1], p[0]);
      wg_split3(gst[2], gst[3], p[1]);
#pragma unroll
      for (int sp = 0; sp < 3; ++sp)
        *reinterpret_cast<u32x2_t*>(lds + GOFF + sp * GSPL + ((gc4 >> 2) * HB + gi) * 32 + (gc4 & 3) * 8) = (u32x2_t){p[0][sp], p[1][sp]};
    }
    __syncthreads();
    prefetch(brick + 1);
    unsigned long long ar[3][NTC][2], br[3][T][2];
#pragma unroll
    for (int sp = 0; sp < 3; ++sp)
#pragma unroll
      for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ar[sp][nt][h]) : "v"(aoff), "n"(sp * GSPL + nt * HB * 32 + h * 512) : "memory");
#pragma unroll
    for (int sp = 0; sp < 3; ++sp)
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(br[sp][t][h]) : "v"(boff[t]), "n"(sp * XSPL + h * 512) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) {
#pragma unroll
      for (int nt = 0; nt < NTC; ++nt) asm volatile("" : "+v"(ar[sp][nt][0]), "+v"(ar[sp][nt][1]));
#pragma unroll
      for (int t = 0; t < T; ++t) asm volatile("" : "+v"(br[sp][t][0]), "+v"(br[sp][t][1]));
    }
    // the six products, smallest first: (g split, x split) with the splits' indices summing to <= 2
    constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int pr = 0; pr < 6; ++pr) {
        const unsigned long long b0 = br[PB[pr]][t][0], b1 = br[PB[pr]][t][1];
        const u32x4_t bv = {(unsigned)b0, (unsigned)(b0 >> 32), (unsigned)b1, (unsigned)(b1 >> 32)};
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) {
          const unsigned long long a0 = ar[PA[pr]][nt][0], a1 = ar[PA[pr]][nt][1];
          const u32x4_t av = {(unsigned)a0, (unsigned)(a0 >> 32), (unsigned)a1, (unsigned)(a1 >> 32)};
          acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av), __builtin_bit_cast(bf16x8_t, bv),
                                                               acc[t][nt], 0, 0, 0);
        }
      }
  }
  const int col = lane & 15;
  const int ncols = (27 * CB + 1) * 16;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int j = wave + NW * t;
    if (j <= 27 * CB) {
#pragma unroll
      for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          partial[((int64_t)blockIdx.x * Cout + nt * 16 + kq * 4 + r) * ncols + j * 16 + col] = acc[t][nt][r];
    }
  }
}

// ------------------------------------------------------------------- wgrad, planar X, stride 1 (the first block)
// Columns c = ci*27 + tap (NTL tiles of 16); the 4 waves of a block split the VOXELS (one brick row each) and
// every wave carries all NTL tiles.  Brick = 4 rows x 64 voxels of one plane; k-group g of a k-step owns voxels
// 16q + 4g + s (s = the k-step within a quad), so a lane's B operands of a quad are 4 consecutive floats of one
// window row.  The gradient rows sit in LDS as [voxel][16] with 16 floats of padding after every 4 voxels
// (conflict-free A reads).
template <int NTL>
struct WplGeom {
  static constexpr int CMAX = NTL * 16 / 27;     // 3 | 12 input channels
  static constexpr int RS = 72;                  // window row: x = h0-4 .. h0+67
  static constexpr int CS = 18 * RS;             // channel stride (3 planes x 6 rows)
  static constexpr int XF4 = CMAX * 18 * 18;     // float4 chunks of the window
  static constexpr int XIT = (XF4 + 255) / 256;
  static constexpr int GP = 16 * 80;             // padded gradient row
};

template <int NTL>
__global__ __launch_bounds__(256, (NTL > 6 ? 1 : 2)) void conv3d_wgrad_planar_kernel(const float* __restrict__ xin,
                                                                     const float* __restrict__ gpre,
                                                                     float* __restrict__ partial, WgDims d,
                                                                     int nbricks, int round_x, int gbf) {
  using G = WplGeom<NTL>;
  __shared__ __attribute__((aligned(16))) float xs[G::XF4 * 4 + 80];  // + ones (the bias-gradient column)
  __shared__ __attribute__((aligned(16))) float gs[4 * G::GP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kq = lane >> 4;
  if (tid < 80) xs[G::XF4 * 4 + tid] = 1.0f;
  const int64_t V = (int64_t)d.D * d.W * d.H;
  const int nH = (d.H + 63) / 64, nW = (d.W + 3) / 4;

  unsigned xrel[G::XIT];
  int xdec[G::XIT];  // rz | ry<<2 | cc<<5 | f4<<9 | used<<14
#pragma unroll
  for (int it = 0; it < G::XIT; ++it) {
    const int q = it * 256 + tid;
    const bool used = q < G::XF4;
    const int row = used ? q / 18 : 0, f4 = q % 18;
    const int cc = row / 18, rz = (row / 6) % 3, ry = row % 6;
    xdec[it] = rz | (ry << 2) | (cc << 5) | (f4 << 9) | ((used && cc < d.Cin) ? 1 << 14 : 0);
    xrel[it] = (unsigned)(((int64_t)cc * V + ((int64_t)rz * d.W + ry) * d.H + f4 * 4) * 4);
  }
  int bbase[NTL];
#pragma unroll
  for (int j = 0; j < NTL; ++j) {
    int c = j * 16 + col;
    const bool ones = c == 27 * d.Cin;  // the first spare column multiplies by ones: sum of gpre = gb
    if (c >= 27 * d.Cin) c = 0;         // other unused columns: computed, never read
    const int ci = c / 27, tap = c % 27;
    const int tz = tap / 9, ty = (tap / 3) % 3, tx = tap % 3;
    bbase[j] = ones ? G::XF4 * 4 + 4 * kq : ci * G::CS + (tz * 6 + ty + wave) * G::RS + tx + 3 + 4 * kq;
  }
  f32x4 acc[NTL];
#pragma unroll
  for (int j = 0; j < NTL; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 xst[G::XIT], gst[4];
  auto prefetch = [&](int brick) {
    const bool live = brick < nbricks;
    int r = live ? brick : 0;
    const int hq = r % nH; r /= nH;
    const int wq = r % nW; r /= nW;
    const int z = r % d.D;
    const int b = r / d.D;
    const int h0 = hq * 64, y0 = wq * 4;
    const float* xb = xin + (int64_t)b * d.Cin * V;
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), (short)0, 0x7fffffff, 0x00020000);
    const unsigned org = (unsigned)(((((int64_t)(z - 1) * d.W + (y0 - 1)) * d.H) + h0 - 4) * 4);
#pragma unroll
    for (int it = 0; it < G::XIT; ++it) {
      const int zi = z - 1 + (xdec[it] & 3), yi = y0 - 1 + ((xdec[it] >> 2) & 7), xi = h0 - 4 + ((xdec[it] >> 9) & 31) * 4;
      const bool ok = (int)(live) & (int)(((xdec[it] >> 14) & 1)) & (int)(zi >= 0) & (int)(zi < d.D) & (int)(yi >= 0) & (int)(yi < d.W) & (int)(xi >= 0) & (int)(xi < d.H);  // bitwise: no exec-mask branches around the loads' address math
      xst[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? org + xrel[it] : OOR, 0, 0));
    }
    const int64_t gorg = ((((int64_t)b * d.D + z) * d.W + y0) * d.H + h0) * 16;
    const void* gb = gbf ? (const void*)(reinterpret_cast<const u16*>(gpre) + gorg) : (const void*)(gpre + gorg);
    const __amdgpu_buffer_rsrc_t rg =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(gb), (short)0, 0x7fffffff, 0x00020000);
    const bool vok = live && h0 + (tid >> 2) < d.H;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const bool ok = vok && y0 + rr < d.W;
      const unsigned el = (unsigned)(rr * d.H * 16 + tid * 4);  // element offset of this thread's 4 channels
      if (gbf) {
        const f32x4 v = bf16x4_to_f32(__builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rg, ok ? el * 2u : OOR, 0, 0)));
        gst[rr] = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        gst[rr] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? el * 4u : OOR, 0, 0));
      }
    }
  };

  int brick = blockIdx.x;
  prefetch(brick);
  for (; brick < nbricks; brick += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < G::XIT; ++it)
      if (it * 256 + tid < G::XF4) {
        float4 v = xst[it];
        if (round_x) {  // the forward rounded this input to bf16 on its way into the MFMA (lr_conv3d_first_bf16)
          v.x = round_bf16(v.x); v.y = round_bf16(v.y); v.z = round_bf16(v.z); v.w = round_bf16(v.w);
        }
        *reinterpret_cast<float4*>(xs + (it * 256 + tid) * 4) = v;
      }
    {
      const int v = tid >> 2, c4 = tid & 3;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        *reinterpret_cast<float4*>(gs + rr * G::GP + (v + (v >> 2)) * 16 + c4 * 4) = gst[rr];
    }
    __syncthreads();
    prefetch(brick + (int)gridDim.x);
    const float* ga = gs + wave * G::GP + kq * 80 + col;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float a = ga[(20 * q + s) * 16];
#pragma unroll
        for (int j = 0; j < NTL; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xs[bbase[j] + 16 * q + s], acc[j], 0, 0, 0);
      }
  }
  const int ncols = d.ntiles * 16;
  const int64_t pb = (int64_t)blockIdx.x * 4 + wave;
#pragma unroll
  for (int j = 0; j < NTL; ++j)
    if (j < d.ntiles) {
#pragma unroll
      for (int r = 0; r < 4; ++r) partial[(pb * 16 + kq * 4 + r) * ncols + j * 16 + col] = acc[j][r];
    }
}

// ------------------------------------------------------------------- wgrad of the first block on the bf16 MFMA
// (bf16-gradient variant).  K = 32 voxels of a row per v_mfma_f32_16x16x32_bf16.  A = gpre^T: transposing LDS reads of
// the [voxel][16 co] image, as in conv3d_wgrad_cl_bf16_kernel.  B = X: a lane's column is one (channel, tap) and its
// 8 k-values are 8 consecutive voxels of a window row starting at 32ks + 8kq + tx + 3 — a 16-byte read only if that
// start is a multiple of 8 elements, so the bf16 window is kept in THREE copies shifted by tx (copy_tx[j] =
// window[j + tx + 3]); the lane picks the copy of its tap and every B operand is one aligned ds_read_b128.
// x is rounded to bf16 while staged — the rounding lr_conv3d_first_bf16 applied in the forward.
template <int NTL>
__global__ __launch_bounds__(256, (NTL > 6 ? 1 : 2)) void conv3d_wgrad_planar_bf16_kernel(const float* __restrict__ xin,
                                                                                           const u16* __restrict__ gpre,
                                                                                           float* __restrict__ partial,
                                                                                           WgDims d, int nbricks) {
  constexpr int CMAX = NTL * 16 / 27;          // 3 | 12 input channels
  constexpr int ROWS = CMAX * 18;              // window rows (channel, plane, row)
  constexpr int RS = 72;                       // elements per LDS row: 64 + 8 of padding — with 64 (128 bytes) the 16 rows a
                                               // B-operand read touches sat on the same four banks (5.8 conflict cycles per
                                               // LDS instruction by PMC); 144 bytes put them 36 banks apart
  constexpr int CPY = ROWS * RS;               // elements of one shifted copy
  constexpr int XF4 = ROWS * 18;               // float4 chunks of the fp32 window (72 columns)
  constexpr int XIT = (XF4 + 255) / 256;
  constexpr int ONES = 3 * CPY, GOFF = ONES + 64;  // element offsets: copies | ones row | gradient rows
  __shared__ __attribute__((aligned(16))) u16 lds[GOFF + 4 * 64 * 16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kq = lane >> 4;
  const int64_t V = (int64_t)d.D * d.W * d.H;
  const int nH = (d.H + 63) / 64, nW = (d.W + 3) / 4;
  if (tid < 64) lds[ONES + tid] = 0x3f80;  // bf16 1.0
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) u16*)lds;

  unsigned xrel[XIT];
  int xdec[XIT];  // rz | ry<<2 | cc<<5 | f4<<9 | used<<14
#pragma unroll
  for (int it = 0; it < XIT; ++it) {
    const int q = it * 256 + tid;
    const bool used = q < XF4;
    const int row = used ? q / 18 : 0, f4 = q % 18;
    const int cc = row / 18, rz = (row / 6) % 3, ry = row % 6;
    xdec[it] = rz | (ry << 2) | (cc << 5) | (f4 << 9) | ((used && cc < d.Cin) ? 1 << 14 : 0);
    xrel[it] = (unsigned)(((int64_t)cc * V + ((int64_t)rz * d.W + ry) * d.H + f4 * 4) * 4);
  }
  // B operand: per N-tile the lane's byte address of (copy tx)[ci][tz][ty + wave][8kq]
  unsigned bbase[NTL];
#pragma unroll
  for (int j = 0; j < NTL; ++j) {
    int c = j * 16 + col;
    const bool ones = c == 27 * d.Cin;  // the first spare column multiplies by ones: sum of gpre = gb
    if (c >= 27 * d.Cin) c = 0;
    const int ci = c / 27, tap = c % 27;
    const int tz = tap / 9, ty = (tap / 3) % 3, tx = tap % 3;
    bbase[j] = lds0 + 2u * (unsigned)(ones ? ONES + 8 * kq : tx * CPY + (ci * 18 + tz * 6 + ty + wave) * RS + 8 * kq);
  }
  // A operand: transposing reads of this wave's gradient row image [64 voxels][16 co]
  const unsigned aoff = lds0 + 2u * (unsigned)(GOFF + wave * 64 * 16) + (unsigned)((kq * 8 + ((lane >> 2) & 3)) * 32 + (lane & 3) * 8);
  f32x4 acc[NTL];
#pragma unroll
  for (int j = 0; j < NTL; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // A brick is 12 MFMAs per wave (NTL = 6): nowhere near a memory latency.  With the next brick requested one iteration
  // ahead every brick waited for its loads in the open (C5: 4.4 ms for 10 GB = 2.3 TB/s).  The requests now run DEPTH
  // bricks ahead in DEPTH register sets; they and their waits are inline asm with static counts (a brick is NL loads and
  // the loop holds no other vector-memory operation) because hipcc's wait-count pass drains every request that is in flight
  // across a loop's back edge.
  // this block's contiguous run of bricks (z fastest, see prefetch)
  const int per_blk = (nbricks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int run = (int)lr_xcd_remap(blockIdx.x, gridDim.x);   // neighbouring runs (the next h segment / the next output rows) on the SAME XCD: they walk dz in step and share halo rows in its L2
  const int brick_begin = min(nbricks, run * per_blk), brick_end = min(nbricks, brick_begin + per_blk);
  constexpr int DEPTH = NTL > 6 ? 1 : 3;
  constexpr int NL = XIT + 2;
  typedef int i32x4_t __attribute__((ext_vector_type(4)));
  f32x4 xst[DEPTH][XIT];
  u32x4_t gst[DEPTH][2];
  auto make_srd = [](const void* p) __attribute__((always_inline)) -> i32x4_t {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    i32x4_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(a >> 32)) & 0xffff;
    r[2] = 0x7fffffff;
    r[3] = 0x00020000;
    return r;
  };
  auto prefetch = [&](int brick, auto setc) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value;
    // Brick order: z FASTEST, and a block owns a contiguous run of it — consecutive bricks of a block are the same (w, h)
    // tile one plane further, so two of the three window planes were read by this CU an iteration ago and come from its
    // XCD's L2.  (With the bricks dealt round-robin over the blocks, h fastest, every window came from HBM: 16.9 GB of
    // L2 misses by PMC for 5.9 GB of tensors, TCC hit rate 0.5 % — the kernel sat on the HBM rate at 2.7 ms.)
    const bool live = brick < brick_end;
    int r = live ? brick : 0;
    const int z = r % d.D; r /= d.D;
    const int hq = r % nH; r /= nH;
    const int wq = r % nW;
    const int b = r / nW;
    const int h0 = hq * 64, y0 = wq * 4;
    const i32x4_t rx = make_srd(xin + (int64_t)b * d.Cin * V);
    const unsigned org = (unsigned)(((((int64_t)(z - 1) * d.W + (y0 - 1)) * d.H) + h0 - 4) * 4);
    f32x4 (&X)[XIT] = xst[SET];
    u32x4_t (&Gs)[2] = gst[SET];
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int zi = z - 1 + (xdec[it] & 3), yi = y0 - 1 + ((xdec[it] >> 2) & 7), xi = h0 - 4 + ((xdec[it] >> 9) & 31) * 4;
      const bool ok = (int)(live) & (int)(((xdec[it] >> 14) & 1)) & (int)(zi >= 0) & (int)(zi < d.D) & (int)(yi >= 0) & (int)(yi < d.W) & (int)(xi >= 0) & (int)(xi < d.H);  // bitwise: no exec-mask branches around the loads' address math
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(X[it]) : "v"(ok ? org + xrel[it] : OOR), "s"(rx));
    }
    // gradient rows: 4 rows x 64 voxels x 16 co bf16 = 512 chunks of 16 bytes, 2 per thread
    const i32x4_t rg = make_srd(gpre + ((((int64_t)b * d.D + z) * d.W + y0) * d.H + h0) * 16);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int ch = k * 256 + tid, rr = ch >> 7, v = (ch >> 1) & 63, half = ch & 1;
      const bool ok = (int)live & (int)(y0 + rr < d.W) & (int)(h0 + v < d.H);
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(Gs[k]) : "v"(ok ? (unsigned)(((rr * d.H + v) * 16 + half * 8) * 2) : OOR), "s"(rg));
    }
  };
  // wait until at most (DEPTH - 1) * NL younger loads are outstanding: set SET has landed
  auto wait_set = [&](auto setc) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value, N = (DEPTH - 1) * NL;
    static_assert(N <= 63, "vmcnt is a 6-bit counter");
    f32x4 (&X)[XIT] = xst[SET];
    u32x4_t (&Gs)[2] = gst[SET];
    if constexpr (XIT == 4)
      asm volatile("s_waitcnt vmcnt(%6)" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[XIT > 3 ? 3 : 0]), "+v"(Gs[0]), "+v"(Gs[1]) : "n"(N));
    else {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(Gs[0]), "+v"(Gs[1]));
#pragma unroll
      for (int it = 0; it < XIT; ++it) asm volatile("" : "+v"(X[it]));
    }
  };
  auto stage = [&](auto setc) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value;
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int q = it * 256 + tid;
      if (q < XF4) {
        const int row = q / 18, f4 = q - row * 18;
        // copy_t[j] = window[j + t + 3]: this chunk (window 4f4..4f4+3) lands at j0 = 4f4 - t - 3 of copy t.
        const f32x4 xv = xst[SET][it];
        const unsigned h0 = __builtin_bit_cast(u16, (__bf16)xv[0]), h1 = __builtin_bit_cast(u16, (__bf16)xv[1]);
        const unsigned h2 = __builtin_bit_cast(u16, (__bf16)xv[2]), h3 = __builtin_bit_cast(u16, (__bf16)xv[3]);
        u16* r0 = lds + row * RS;
        // t = 1: j0 = 4(f4-1), an aligned 8-byte store, wholly inside or outside the 64 columns
        if (f4 >= 1 && f4 <= 16) *reinterpret_cast<uint2*>(r0 + CPY + 4 * (f4 - 1)) = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
        // t = 0: j0 = 4(f4-1) + 1 -> 2 + 4 + 2 bytes; t = 2: j0 = 4(f4-2) + 3 -> 2 + 4 + 2 bytes (the 4-byte piece is aligned)
        {
          const int j0 = 4 * (f4 - 1) + 1;
          if (j0 >= 0 && j0 < 64) r0[j0] = (u16)h0;
          if (j0 + 1 >= 0 && j0 + 2 < 64) *reinterpret_cast<unsigned*>(r0 + j0 + 1) = h1 | (h2 << 16);
          if (j0 + 3 >= 0 && j0 + 3 < 64) r0[j0 + 3] = (u16)h3;
        }
        {
          u16* r2 = r0 + 2 * CPY;
          const int j0 = 4 * (f4 - 2) + 3;
          if (j0 >= 0 && j0 < 64) r2[j0] = (u16)h0;
          if (j0 + 1 >= 0 && j0 + 2 < 64) *reinterpret_cast<unsigned*>(r2 + j0 + 1) = h1 | (h2 << 16);
          if (j0 + 3 >= 0 && j0 + 3 < 64) r2[j0 + 3] = (u16)h3;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int ch = k * 256 + tid, rr = ch >> 7, v = (ch >> 1) & 63, half = ch & 1;
      *reinterpret_cast<u32x4_t*>(lds + GOFF + (rr * 64 + v) * 16 + half * 8) = gst[SET][k];
    }
  };
  auto sweep = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      unsigned long long ar[2];
#pragma unroll
      for (int h = 0; h < 2; ++h)
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ar[h]) : "v"(aoff), "n"(ks * 32 * 32 + h * 128) : "memory");
      u32x4_t bv[NTL];
#pragma unroll
      for (int j = 0; j < NTL; ++j)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bv[j]) : "v"(bbase[j]), "n"(ks * 64) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("" : "+v"(ar[0]), "+v"(ar[1]));
      const u32x4_t av = {(unsigned)ar[0], (unsigned)(ar[0] >> 32), (unsigned)ar[1], (unsigned)(ar[1] >> 32)};
#pragma unroll
      for (int j = 0; j < NTL; ++j) {
        asm volatile("" : "+v"(bv[j]));
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av), __builtin_bit_cast(bf16x8_t, bv[j]),
                                                         acc[j], 0, 0, 0);
      }
    }
  };
  int brick = brick_begin;
  prefetch(brick, std::integral_constant<int, 0>{});
  if constexpr (DEPTH >= 2) prefetch(brick + 1, std::integral_constant<int, DEPTH >= 2 ? 1 : 0>{});
  if constexpr (DEPTH >= 3) prefetch(brick + 2, std::integral_constant<int, DEPTH >= 3 ? 2 : 0>{});
  auto iteration = [&](auto setc) __attribute__((always_inline)) -> bool {
    if (brick >= brick_end) return false;
    __syncthreads();   // the previous brick's LDS reads are done
    wait_set(setc);
    stage(setc);
    __syncthreads();
    prefetch(brick + DEPTH, setc);   // into the set just consumed (past the end: NL loads that return zeros)
    sweep();
    brick += 1;
    return true;
  };
  while (true) {
    if (!iteration(std::integral_constant<int, 0>{})) break;
    if constexpr (DEPTH >= 2) { if (!iteration(std::integral_constant<int, DEPTH >= 2 ? 1 : 0>{})) break; }
    if constexpr (DEPTH >= 3) { if (!iteration(std::integral_constant<int, DEPTH >= 3 ? 2 : 0>{})) break; }
  }
  // the look-ahead past the last brick must have landed before its registers are reused (the operands keep them alive)
  if constexpr (XIT == 4 && DEPTH == 3)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(xst[0][0]), "+v"(xst[0][1]), "+v"(xst[0][2]), "+v"(xst[0][3]), "+v"(gst[0][0]), "+v"(gst[0][1]),
                 "+v"(xst[DEPTH > 1 ? 1 : 0][0]), "+v"(xst[DEPTH > 1 ? 1 : 0][1]), "+v"(xst[DEPTH > 1 ? 1 : 0][2]), "+v"(xst[DEPTH > 1 ? 1 : 0][3]),
                 "+v"(gst[DEPTH > 1 ? 1 : 0][0]), "+v"(gst[DEPTH > 1 ? 1 : 0][1]),
                 "+v"(xst[DEPTH > 2 ? 2 : 0][0]), "+v"(xst[DEPTH > 2 ? 2 : 0][1]), "+v"(xst[DEPTH > 2 ? 2 : 0][2]), "+v"(xst[DEPTH > 2 ? 2 : 0][3]),
                 "+v"(gst[DEPTH > 2 ? 2 : 0][0]), "+v"(gst[DEPTH > 2 ? 2 : 0][1]));
  else
    wait_set(std::integral_constant<int, 0>{});   // DEPTH 1: vmcnt(0) over the only set
  const int ncols = d.ntiles * 16;
  const int64_t pb = (int64_t)blockIdx.x * 4 + wave;
#pragma unroll
  for (int j = 0; j < NTL; ++j)
    if (j < d.ntiles) {
#pragma unroll
      for (int r = 0; r < 4; ++r) partial[(pb * 16 + kq * 4 + r) * ncols + j * 16 + col] = acc[j][r];
    }
}

// ------------------------------------------------------------------- the same weight gradient, SHIFTING THE GRADIENT
// instead of the window (first block, Cin <= 3, bf16 gradients; round 3).  The kernel above keeps the bf16 window in three
// copies shifted by the tap's tx so that every B operand is an aligned 16-byte read — 28 predicated LDS stores of 2 / 4 / 8
// bytes per thread and brick for 12 MFMAs per wave: it was bound by its staging (PMC: 46 LDS instructions and 120 vector
// instructions per wave and brick, 2 conflict cycles per LDS instruction).  A weight gradient sums over ALL voxels, so it
// does not matter which brick adds a product x[u] * g[v]: here a brick owns the 64 ALIGNED window columns u = h0 .. h0+63
// (no x halo, ONE unshifted copy, one aligned 8-byte store per staged chunk) and meets them, for tap tx, with the gradient
// at v = u - tx + 1 — a shift of the A operand by whole voxel rows of its [voxel][16 co] image, which the transposing read
// takes as an immediate.  The gradient row is staged with one voxel of halo on each side (zero outside the volume).
// Columns are grouped by tx: N-tile j = (tx = j / 2, half = j % 2), column m = half * 16 + col = (ci, tz, ty) for m < 27;
// column 27 of group tx = 1 (unshifted gradient = exactly the brick's own outputs) multiplies by ones: the bias gradient.
// wgrad_finish_kernel maps this order with x_layout = LR_WG_PLANAR_TX.
constexpr int LR_WG_PLANAR_TX = -7;
__global__ __launch_bounds__(256, 2) void conv3d_wgrad_planar_bf16s_kernel(const float* __restrict__ xin, const u16* __restrict__ gpre,
                                                                            float* __restrict__ partial, WgDims d, int nbricks) {
  constexpr int NTL = 6, CMAX = 3;
  constexpr int ROWS = CMAX * 18;              // window rows (channel, plane, row)
  constexpr int RS = 72;                       // elements per LDS row (64 + 8 of padding: rows 36 banks apart)
  constexpr int XF4 = ROWS * 16;               // float4 chunks of the fp32 window (64 columns)
  constexpr int XIT = (XF4 + 255) / 256;       // 4
  constexpr int GV = 66;                       // gradient voxels per row: 64 + one of halo each side
  constexpr int GCH = 4 * GV * 2;              // 16-byte chunks of the four gradient rows
  constexpr int GIT = (GCH + 255) / 256;       // 3
  constexpr int ONES = ROWS * RS, GOFF = ONES + 64;  // element offsets: window | ones row | gradient rows
  __shared__ __attribute__((aligned(16))) u16 lds[GOFF + 4 * GV * 16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kq = lane >> 4;
  const int64_t V = (int64_t)d.D * d.W * d.H;
  const int nH = (d.H + 63) / 64, nW = (d.W + 3) / 4;
  if (tid < 64) lds[ONES + tid] = 0x3f80;  // bf16 1.0
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) u16*)lds;

  unsigned xrel[XIT];
  int xdec[XIT];  // rz | ry<<2 | cc<<5 | f4<<9 | used<<14
#pragma unroll
  for (int it = 0; it < XIT; ++it) {
    const int q = it * 256 + tid;
    const bool used = q < XF4;
    const int row = used ? q / 16 : 0, f4 = q % 16;
    const int cc = row / 18, rz = (row / 6) % 3, ry = row % 6;
    xdec[it] = rz | (ry << 2) | (cc << 5) | (f4 << 9) | ((used && cc < d.Cin) ? 1 << 14 : 0);
    xrel[it] = (unsigned)(((int64_t)cc * V + ((int64_t)rz * d.W + ry) * d.H + f4 * 4) * 4);
  }
  // B operand: per N-tile the lane's byte address of window row (ci, tz, ty + wave), columns 8kq ..
  unsigned bbase[NTL];
#pragma unroll
  for (int j = 0; j < NTL; ++j) {
    const int tx = j >> 1, m = (j & 1) * 16 + col;
    const bool ones = tx == 1 && m == 27;
    const int mm = m < 27 ? m : 0;
    const int ci = mm / 9, tz = (mm / 3) % 3, ty = mm % 3;
    bbase[j] = lds0 + 2u * (unsigned)(ones ? ONES + 8 * kq : (ci * 18 + tz * 6 + ty + wave) * RS + 8 * kq);
  }
  // A operand: transposing reads of this wave's gradient row image [66 voxels][16 co]; image row 0 = voxel h0 - 1
  const unsigned aoff = lds0 + 2u * (unsigned)(GOFF + wave * GV * 16) + (unsigned)((kq * 8 + ((lane >> 2) & 3)) * 32 + (lane & 3) * 8);
  f32x4 acc[NTL];
#pragma unroll
  for (int j = 0; j < NTL; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int per_blk = (nbricks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int run = (int)lr_xcd_remap(blockIdx.x, gridDim.x);
  const int brick_begin = min(nbricks, run * per_blk), brick_end = min(nbricks, brick_begin + per_blk);
  constexpr int DEPTH = 3, NL = XIT + GIT;
  typedef int i32x4_t __attribute__((ext_vector_type(4)));
  f32x4 xst[DEPTH][XIT];
  u32x4_t gst[DEPTH][GIT];
  auto make_srd = [](const void* p) __attribute__((always_inline)) -> i32x4_t {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    i32x4_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(a >> 32)) & 0xffff;
    r[2] = 0x7fffffff;
    r[3] = 0x00020000;
    return r;
  };
  auto prefetch = [&](int brick, auto setc) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value;
    const bool live = brick < brick_end;
    int r = live ? brick : 0;
    const int z = r % d.D; r /= d.D;   // z fastest: consecutive bricks of a block share two of their three window planes (L2)
    const int hq = r % nH; r /= nH;
    const int wq = r % nW;
    const int b = r / nW;
    const int h0 = hq * 64, y0 = wq * 4;
    const i32x4_t rx = make_srd(xin + (int64_t)b * d.Cin * V);
    const unsigned org = (unsigned)(((((int64_t)(z - 1) * d.W + (y0 - 1)) * d.H) + h0) * 4);
    f32x4 (&X)[XIT] = xst[SET];
    u32x4_t (&Gs)[GIT] = gst[SET];
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int zi = z - 1 + (xdec[it] & 3), yi = y0 - 1 + ((xdec[it] >> 2) & 7), xi = h0 + ((xdec[it] >> 9) & 31) * 4;
      const bool ok = (int)(live) & (int)(((xdec[it] >> 14) & 1)) & (int)(zi >= 0) & (int)(zi < d.D) & (int)(yi >= 0) & (int)(yi < d.W) & (int)(xi < d.H);
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(X[it]) : "v"(ok ? org + xrel[it] : OOR), "s"(rx));
    }
    // gradient rows y0 .. y0+3, voxels h0-1 .. h0+64 (16 co bf16 = two 16-byte chunks per voxel)
    const i32x4_t rg = make_srd(gpre + ((((int64_t)b * d.D + z) * d.W + y0) * d.H) * 16);
#pragma unroll
    for (int k = 0; k < GIT; ++k) {
      const int ch = k * 256 + tid, rr = ch / (GV * 2), rem = ch - rr * (GV * 2), vi = rem >> 1, half = rem & 1;
      const int v = h0 - 1 + vi;
      const bool ok = (int)live & (int)(ch < GCH) & (int)(y0 + rr < d.W) & (int)(v >= 0) & (int)(v < d.H);
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(Gs[k]) : "v"(ok ? (unsigned)(((rr * d.H + v) * 16 + half * 8) * 2) : OOR), "s"(rg));
    }
  };
  auto wait_set = [&](auto setc) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value, N = (DEPTH - 1) * NL;
    static_assert(N <= 63 && XIT == 4 && GIT == 3, "static wait counts");
    f32x4 (&X)[XIT] = xst[SET];
    u32x4_t (&Gs)[GIT] = gst[SET];
    asm volatile("s_waitcnt vmcnt(%7)" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(Gs[0]), "+v"(Gs[1]), "+v"(Gs[2]) : "n"(N));
  };
  auto stage = [&](auto setc) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value;
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int q = it * 256 + tid;
      if (q < XF4) {   // compile-time true but for the last slot
        const int row = q / 16, f4 = q - row * 16;
        const f32x4 xv = xst[SET][it];
        const unsigned h0 = __builtin_bit_cast(u16, (__bf16)xv[0]), h1 = __builtin_bit_cast(u16, (__bf16)xv[1]);
        const unsigned h2 = __builtin_bit_cast(u16, (__bf16)xv[2]), h3 = __builtin_bit_cast(u16, (__bf16)xv[3]);
        *reinterpret_cast<uint2*>(lds + row * RS + 4 * f4) = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
      }
    }
#pragma unroll
    for (int k = 0; k < GIT; ++k) {
      const int ch = k * 256 + tid, rr = ch / (GV * 2), rem = ch - rr * (GV * 2), vi = rem >> 1, half = rem & 1;
      if (ch < GCH) *reinterpret_cast<u32x4_t*>(lds + GOFF + (rr * GV + vi) * 16 + half * 8) = gst[SET][k];
    }
  };
  auto sweep = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      unsigned long long ar[3][2];
#pragma unroll
      for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int h = 0; h < 2; ++h)   // image row = K index + 2 - tx
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ar[tx][h]) : "v"(aoff), "n"(ks * 32 * 32 + h * 128 + (2 - tx) * 32) : "memory");
      u32x4_t bv[NTL];
#pragma unroll
      for (int j = 0; j < NTL; ++j)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bv[j]) : "v"(bbase[j]), "n"(ks * 64) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        asm volatile("" : "+v"(ar[tx][0]), "+v"(ar[tx][1]));
        const u32x4_t av = {(unsigned)ar[tx][0], (unsigned)(ar[tx][0] >> 32), (unsigned)ar[tx][1], (unsigned)(ar[tx][1] >> 32)};
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int j = 2 * tx + hh;
          asm volatile("" : "+v"(bv[j]));
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av), __builtin_bit_cast(bf16x8_t, bv[j]),
                                                           acc[j], 0, 0, 0);
        }
      }
    }
  };
  int brick = brick_begin;
  prefetch(brick, std::integral_constant<int, 0>{});
  prefetch(brick + 1, std::integral_constant<int, 1>{});
  prefetch(brick + 2, std::integral_constant<int, 2>{});
  auto iteration = [&](auto setc) __attribute__((always_inline)) -> bool {
    if (brick >= brick_end) return false;
    __syncthreads();   // the previous brick's LDS reads are done
    wait_set(setc);
    stage(setc);
    __syncthreads();
    prefetch(brick + DEPTH, setc);   // into the set just consumed (past the end: NL loads that return zeros)
    sweep();
    brick += 1;
    return true;
  };
  while (true) {
    if (!iteration(std::integral_constant<int, 0>{})) break;
    if (!iteration(std::integral_constant<int, 1>{})) break;
    if (!iteration(std::integral_constant<int, 2>{})) break;
  }
  // the look-ahead past the last brick must have landed before its registers are reused (the operands keep them alive)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(xst[0][0]), "+v"(xst[0][1]), "+v"(xst[0][2]), "+v"(xst[0][3]), "+v"(gst[0][0]), "+v"(gst[0][1]), "+v"(gst[0][2]),
               "+v"(xst[1][0]), "+v"(xst[1][1]), "+v"(xst[1][2]), "+v"(xst[1][3]), "+v"(gst[1][0]), "+v"(gst[1][1]), "+v"(gst[1][2]),
               "+v"(xst[2][0]), "+v"(xst[2][1]), "+v"(xst[2][2]), "+v"(xst[2][3]), "+v"(gst[2][0]), "+v"(gst[2][1]), "+v"(gst[2][2]));
  const int ncols = NTL * 16;
  const int64_t pb = (int64_t)blockIdx.x * 4 + wave;
#pragma unroll
  for (int j = 0; j < NTL; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) partial[(pb * 16 + kq * 4 + r) * ncols + j * 16 + col] = acc[j][r];
}

// partial[k][co][n] summed over the nblk partials in double (fixed order), column n -> (ci, tap) of gw.
// Block = 64 columns x 16 slices of k: consecutive threads read consecutive floats, the 16 slice sums meet in LDS.
__global__ __launch_bounds__(1024) void wgrad_finish_kernel(const float* __restrict__ partial, float* __restrict__ gw,
                                                            float* __restrict__ gb, int nblk, int Cout, int Cin,
                                                            int ncols, int x_layout, int gb_col) {
  __shared__ double red[16][64];
  const int tx = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + tx;  // (co, n)
  const bool live = t < Cout * ncols;
  double s = 0.0;
  if (live) {
    const int64_t step = (int64_t)Cout * ncols;
    const int per = (nblk + 15) / 16, k0 = sl * per, k1 = min(nblk, k0 + per);
    // four independent chains: the loads of a chain wait for nothing but each other's issue (one chain = per dependent
    // HBM/L2 round trips, 0.1 ms at 256 partials per slice)
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = k0;
    for (; k + 3 < k1; k += 4) {
      s += (double)partial[k * step + t];
      s1 += (double)partial[(k + 1) * step + t];
      s2 += (double)partial[(k + 2) * step + t];
      s3 += (double)partial[(k + 3) * step + t];
    }
    for (; k < k1; ++k) s += (double)partial[k * step + t];
    s = (s + s1) + (s2 + s3);
  }
  red[sl][tx] = s;
  __syncthreads();
  if (sl != 0 || !live) return;
#pragma unroll
  for (int i = 1; i < 16; ++i) s += red[i][tx];
  const int co = t / ncols, n = t - co * ncols;
  int ci, tap;
  if (x_layout == LR_WG_PLANAR_TX) {   // conv3d_wgrad_planar_bf16s_kernel: columns grouped by tx, 32 per group
    const int tx = n >> 5, m = n & 31;
    ci = m < 27 ? m / 9 : Cin;
    tap = m < 27 ? ((m / 3) % 3) * 9 + (m % 3) * 3 + tx : 27;
  } else if (x_layout == LR_LAYOUT_NCDHW) {
    ci = n / 27; tap = n - ci * 27;
  } else {
    const int cbn = (Cin + 15) >> 4, j = n >> 4;
    tap = j / cbn; ci = (j - tap * cbn) * 16 + (n & 15);
  }
  if (gb != nullptr && n == gb_col) gb[co] = (float)s;  // the ones column of the fast paths
  else if (ci < Cin && tap < 27) gw[((int64_t)co * Cin + ci) * 27 + tap] = (float)s;
}

// bias gradient on the generic path: per-block channel sums of gpre (B*V, C) -> partial -> sum_partials_kernel
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ gpre, float* __restrict__ partial,
                                                          int64_t nvox, int C, int gbf) {
  __shared__ float red[256];
  const int c = threadIdx.x % C, lanes = 256 / C;  // C in {16, 32}
  double s = 0.0;
  for (int64_t v = (int64_t)blockIdx.x * lanes + threadIdx.x / C; v < nvox; v += (int64_t)gridDim.x * lanes)
    s += gbf ? (double)__builtin_bit_cast(float, (unsigned)reinterpret_cast<const u16*>(gpre)[v * C + c] << 16)
             : (double)gpre[v * C + c];
  red[threadIdx.x] = (float)s;
  __syncthreads();
  if (threadIdx.x < C) {
    float t = 0.0f;
    for (int i = threadIdx.x; i < 256; i += C) t += red[i];
    partial[(int64_t)blockIdx.x * C + threadIdx.x] = t;
  }
}

}  // namespace

extern "C" int lr_lrelu_bwd_f32(const float* gy, int gy_layout, const float* y, int y_layout, float* gpre,
                                float* gb_partial, float* gb, int B, int C, int D, int W, int H,
                                float negative_slope, int nblk, void* stream) {
  if (!gy || !y || !gpre) return LR_ENULL;
  if (B < 1 || D < 1 || W < 1 || H < 1 || nblk < 1 || nblk > 65535) return LR_EINVAL;
  if (C != 16 && C != 32) return LR_EUNSUPPORTED;
  if ((gy_layout == LR_LAYOUT_NDHWC_HPS || y_layout == LR_LAYOUT_NDHWC_HPS) && (H & 1)) return LR_EUNSUPPORTED;
  if ((gb != nullptr) != (gb_partial != nullptr)) return LR_ENULL;
  hipStream_t st = lr_stream(stream);
  hipLaunchKernelGGL(lrelu_bwd_kernel, dim3((unsigned)nblk), dim3(256), 0, st, gy, gy_layout, y, y_layout, gpre,
                     gb_partial, B, C, D, W, H, negative_slope);
  if (int e = lr_launch_status()) return e;
  if (gb) {
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, gb_partial, gb, nblk, C);
    return lr_launch_status();
  }
  return LR_OK;
}

extern "C" int lr_conv3d_dgrad_f32(const float* gpre, const float* packed_wT, float* gx, int B, int Cg, int Cx,
                                   int D, int W, int H, int stride, int gx_layout, const float* x_saved,
                                   int x_layout, float negative_slope, void* stream) {
  if (!gpre || !packed_wT || !gx) return LR_ENULL;
  if (stride != 2) return LR_EUNSUPPORTED;  // blocks 1..5; block 0's input needs no gradient
  if (Cx != 16 && Cx != 32) return LR_EUNSUPPORTED;
  if (Cg % 4 || B < 1 || D < 1 || W < 1 || H < 1) return LR_EINVAL;
  if (gx_layout != LR_LAYOUT_NDHWC && gx_layout != LR_LAYOUT_NDHWC_HPS) return LR_EINVAL;
  if (gx_layout == LR_LAYOUT_NDHWC_HPS && (H & 1)) return LR_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(gpre) | reinterpret_cast<uintptr_t>(gx)) & 15u) return LR_EALIGN;
  DgDims d;
  d.B = B; d.Cg = Cg; d.Cx = Cx; d.D = D; d.W = W; d.H = H;
  d.Do = (D - 1) / 2 + 1; d.Wo = (W - 1) / 2 + 1; d.Ho = (H - 1) / 2 + 1;
  const bool lds_path = Cg == 16 || Cg == 32;  // tile-relative offsets (five planes); the direct kernel: per batch element
  if ((lds_path ? (int64_t)6 * d.Wo * d.Ho * Cg * 4 : (int64_t)d.Do * d.Wo * d.Ho * Cg * 4) >= 0x7fffffffLL) return LR_EINVAL;
  d.nHq = ((H + 1) / 2 + 15) / 16; d.nWq = ((W + 1) / 2 + DMT - 1) / DMT; d.nDq = ((D + 1) / 2 + 3) / 4;
  d.gx_layout = gx_layout;
  if (x_saved && x_layout != LR_LAYOUT_NDHWC && x_layout != LR_LAYOUT_NDHWC_HPS && x_layout != LR_LAYOUT_BF16_NDHWC &&
      x_layout != LR_LAYOUT_BF16_NDHWC_HPS && x_layout != LR_LAYOUT_SIGN4)
    return LR_EINVAL;
  if (x_saved && (x_layout == LR_LAYOUT_NDHWC_HPS || x_layout == LR_LAYOUT_BF16_NDHWC_HPS) && (H & 1)) return LR_EUNSUPPORTED;
  if (x_saved && (reinterpret_cast<uintptr_t>(x_saved) & (x_layout == LR_LAYOUT_SIGN4 ? 0u : 15u))) return LR_EALIGN;
  d.xs_layout = x_layout; d.slope = negative_slope;
  const int64_t nblk = (int64_t)B * d.nDq * d.nWq * d.nHq;  // tiles of 4 x 4 x 16 voxels per parity class
  if (nblk > 0x7fffffffLL) return LR_EINVAL;
  const float4* wt = reinterpret_cast<const float4*>(packed_wT);
  hipStream_t st = lr_stream(stream);
  const dim3 grid((unsigned)nblk), blk(256);
#ifdef LR_DIAG_ABLATIONS   // diagnostic build only (make -B EXTRA=-DLR_DIAG_ABLATIONS): timing ablations, WRONG results
  const int dbgv = getenv("LIFTREG_DGRAD_DBG") ? atoi(getenv("LIFTREG_DGRAD_DBG")) : 0;
#else
  const int dbgv = 0;
#endif
  const float* xs_eff = (dbgv & 1) ? nullptr : x_saved;
  const int mk = !xs_eff ? 0 : x_layout == LR_LAYOUT_NDHWC ? 1 : x_layout == LR_LAYOUT_NDHWC_HPS ? 2 : x_layout == LR_LAYOUT_SIGN4 ? 3 : -1;
  if (Cx == 16 && (Cg == 32 || Cg == 16) && mk >= 0 && !lr_sw_set(LR_SW_DGRAD_OLD)) {
    // 16-channel gx (the encoder's block 1): persistent 8-wave blocks, all weights in LDS (conv3d_dgrad_wlds_kernel)
    const int CBv = Cg / 16;
    const int nWq2 = (d.Wo + WMT - 1) / WMT, nDq2 = (d.Do + 7) / 8;
    const int64_t nt = (int64_t)B * nDq2 * nWq2 * d.nHq;
    if (nt <= 0x7fffffffLL && (int64_t)10 * d.Wo * d.Ho * Cg * 4 < 0x7fffffffLL) {
      const size_t ldsb = ((size_t)27 * CBv * 64 * 4 + (size_t)9 * (WMT + 1) * 17 * (Cg + 4)) * sizeof(float);
      int resident = 256;  // one 8-wave block per CU
      resident = lr_sw_int(LR_SW_DGRAD_BLOCKS, resident);  // tuning aid
      const dim3 g2((unsigned)(nt < resident ? nt : resident)), b2(512);
#define LR_DGW(CBV, MKV)                                                                                                  \
  do {                                                                                                                    \
    static std::atomic<uint64_t> attr_done{0}; /* one bit per device */                                                 \
    if (lr_raise_dyn_lds(reinterpret_cast<const void*>(&conv3d_dgrad_wlds_kernel<CBV, MKV>), ldsb, attr_done) != LR_OK) return LR_ELAUNCH;\
    hipLaunchKernelGGL((conv3d_dgrad_wlds_kernel<CBV, MKV>), g2, b2, ldsb, st, gpre, wt, gx, xs_eff, d, (int)nt, dbgv);   \
  } while (0)
#define LR_DGW_MK(CBV)                                                                                                    \
  do {                                                                                                                    \
    if (mk == 0) LR_DGW(CBV, 0); else if (mk == 1) LR_DGW(CBV, 1); else if (mk == 2) LR_DGW(CBV, 2); else LR_DGW(CBV, 3);  \
  } while (0)
      if (CBv == 2) LR_DGW_MK(2); else LR_DGW_MK(1);
#undef LR_DGW_MK
#undef LR_DGW
      return lr_launch_status();
    }
  }
  if (Cx == 32 && Cg == 32 && mk >= 0 && mk <= 2 && !lr_sw_set(LR_SW_DGRAD_OLD)) {
    // 32-channel gx (blocks 2..5): persistent 8-wave blocks, all fragments in LDS (conv3d_dgrad_wlds32_kernel)
    const int nDq2 = (d.Do + 7) / 8;
    const int64_t nt = (int64_t)B * nDq2 * d.Wo * d.nHq;
    if (nt <= 0x7fffffffLL && (int64_t)10 * d.Wo * d.Ho * Cg * 4 < 0x7fffffffLL) {
      const size_t ldsb = ((size_t)27 * 2 * 2 * 64 * 4 + (size_t)9 * 2 * 17 * (Cg + 4)) * sizeof(float);
      int resident = 256;  // one 8-wave block per CU
      resident = lr_sw_int(LR_SW_DGRAD_BLOCKS, resident);  // tuning aid
      const dim3 g2((unsigned)(nt < resident ? nt : resident)), b2(512);
#define LR_DGW32(MKV)                                                                                                      \
  do {                                                                                                                    \
    static std::atomic<uint64_t> attr_done{0}; /* one bit per device */                                                 \
    if (lr_raise_dyn_lds(reinterpret_cast<const void*>(&conv3d_dgrad_wlds32_kernel<MKV>), ldsb, attr_done) != LR_OK) return LR_ELAUNCH;\
    hipLaunchKernelGGL((conv3d_dgrad_wlds32_kernel<MKV>), g2, b2, ldsb, st, gpre, wt, gx, xs_eff, d, (int)nt);            \
  } while (0)
      if (mk == 0) LR_DGW32(0); else if (mk == 1) LR_DGW32(1); else LR_DGW32(2);
#undef LR_DGW32
      return lr_launch_status();
    }
  }
  if (Cg == 32 && Cx == 16) hipLaunchKernelGGL((conv3d_dgrad_lds_kernel<1, 2>), grid, blk, 0, st, gpre, wt, gx, x_saved, d);
  else if (Cg == 32) hipLaunchKernelGGL((conv3d_dgrad_lds_kernel<2, 2>), grid, blk, 0, st, gpre, wt, gx, x_saved, d);
  else if (Cg == 16 && Cx == 16) hipLaunchKernelGGL((conv3d_dgrad_lds_kernel<1, 1>), grid, blk, 0, st, gpre, wt, gx, x_saved, d);
  else if (Cg == 16) hipLaunchKernelGGL((conv3d_dgrad_lds_kernel<2, 1>), grid, blk, 0, st, gpre, wt, gx, x_saved, d);
  else if (Cx == 16) hipLaunchKernelGGL(conv3d_dgrad_kernel<1>, grid, blk, 0, st, gpre, wt, gx, x_saved, d);
  else hipLaunchKernelGGL(conv3d_dgrad_kernel<2>, grid, blk, 0, st, gpre, wt, gx, x_saved, d);
  return lr_launch_status();
}

extern "C" int64_t lr_conv3d_wgrad_partial_floats(int Cin, int Cout, int x_layout, int nblk) {
  if (x_layout == LR_LAYOUT_NCDHW_RBF16) x_layout = LR_LAYOUT_NCDHW;  // planar either way
  // planar: ceil((27*Cin + 1)/16) tiles (one spare column carries the bias gradient), one partial per WAVE of
  // the fast path (its waves split the voxels); channels-last: 27*ceil(Cin/16) tiles + the ones tile
  int ntiles = x_layout == LR_LAYOUT_NCDHW ? (Cin * 27 + 16) / 16 : 27 * ((Cin + 15) / 16) + 1;
  if (x_layout == LR_LAYOUT_NCDHW && Cin <= 3 && ntiles < 6) ntiles = 6;   // conv3d_wgrad_planar_bf16s_kernel always writes its six tiles (three tx groups of 32 columns)
  return (int64_t)nblk * (x_layout == LR_LAYOUT_NCDHW ? 4 : 1) * Cout * ntiles * 16;
}

static int wgrad_impl(const float* x, int x_layout, const float* gpre, int gbf, float* partial, float* gw, float* gb,
                      int B, int Cin, int Cout, int D, int W, int H, int stride, int nblk, void* stream) {
  if (!x || !gpre || !partial || !gw) return LR_ENULL;
  if (B < 1 || Cin < 1 || D < 1 || W < 1 || H < 1 || nblk < 1 || nblk > 65535) return LR_EINVAL;
  if ((stride != 1 && stride != 2) || (Cout != 16 && Cout != 32)) return LR_EUNSUPPORTED;
  if (x_layout == LR_LAYOUT_NDHWC_HPS && ((H & 1) || (Cin & 15))) return LR_EUNSUPPORTED;
  // bf16-forward training variant: x is bf16 storage (3|4), or the fp32 first-block input that the forward rounded (5)
  const bool xbf = x_layout == LR_LAYOUT_BF16_NDHWC || x_layout == LR_LAYOUT_BF16_NDHWC_HPS;
  const bool xround = x_layout == LR_LAYOUT_NCDHW_RBF16;
  if (xbf && x_layout == LR_LAYOUT_BF16_NDHWC_HPS && (H & 1)) return LR_EUNSUPPORTED;
  const int x_layout_in = x_layout;
  if (xround) x_layout = LR_LAYOUT_NCDHW;
  if (xbf) x_layout = x_layout_in == LR_LAYOUT_BF16_NDHWC_HPS ? LR_LAYOUT_NDHWC_HPS : LR_LAYOUT_NDHWC;  // same column order
  WgDims d;
  d.B = B; d.Cin = Cin; d.Cout = Cout; d.D = D; d.W = W; d.H = H; d.stride = stride; d.x_layout = x_layout;
  d.Do = (D - 1) / stride + 1; d.Wo = (W - 1) / stride + 1; d.Ho = (H - 1) / stride + 1;
  d.ntiles = x_layout == LR_LAYOUT_NCDHW ? (Cin * 27 + 15) / 16 : 27 * ((Cin + 15) / 16);
  if (d.ntiles > 4 * WG_MAXT) return LR_EUNSUPPORTED;  // Cin <= 32 channels-last, <= 33 planar
  const int64_t ngroups = (int64_t)B * d.Do * d.Wo * ((d.Ho + 3) / 4);
  hipStream_t st = lr_stream(stream);
  const bool al16 = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gpre)) & 15u) == 0;
  const int64_t V = (int64_t)D * W * H;
  int nparts = 0;  // > 0: a fast path ran and left this many partials
  bool planar_tx = false;  // the partial columns are in conv3d_wgrad_planar_bf16s_kernel's order
  if (x_layout != LR_LAYOUT_NCDHW && stride == 2 && (Cin == 16 || Cin == 32) && al16 &&
      (int64_t)4 * W * H * Cin * 4 < 0x7fffffffLL) {  // a 3-plane window within 31-bit offsets
    // blocks 1..5: LDS-staged bricks (one output row segment each)
    const int hb = 32 / (Cin / 16);
    const int64_t nbricks = (int64_t)B * d.Do * d.Wo * ((d.Ho + hb - 1) / hb);
    if (gbf && xbf) {  // both operands bf16: the bf16-MFMA kernel (32-voxel bricks)
      const int64_t nb32 = (int64_t)B * d.Do * d.Wo * ((d.Ho + 31) / 32);
      if (nb32 < 0x7fffffffLL) {
        const unsigned grid = (unsigned)(nb32 < nblk ? nb32 : nblk);
        const bool hps = x_layout == LR_LAYOUT_NDHWC_HPS;
        const u16* xh = reinterpret_cast<const u16*>(x);
        const u16* gh = reinterpret_cast<const u16*>(gpre);
#define LR_WB(CBV, NTCV)                                                                                              \
  do {                                                                                                                \
    if (hps) hipLaunchKernelGGL((conv3d_wgrad_cl_bf16_kernel<CBV, NTCV, true>), dim3(grid), dim3(256 * CBV), 0, st, xh, gh, partial, d, (int)nb32); \
    else hipLaunchKernelGGL((conv3d_wgrad_cl_bf16_kernel<CBV, NTCV, false>), dim3(grid), dim3(256 * CBV), 0, st, xh, gh, partial, d, (int)nb32);    \
  } while (0)
        if (Cin == 16 && Cout == 16) LR_WB(1, 1);
        else if (Cin == 16) LR_WB(1, 2);
        else if (Cout == 16) LR_WB(2, 1);
        else LR_WB(2, 2);
#undef LR_WB
        nparts = (int)grid;
      }
    } else if (!gbf && !xbf && Cin == 16 && lr_sw_int(LR_SW_WGRAD_SPLIT, 1) &&
               (int64_t)B * d.Do * d.Wo * ((d.Ho + 31) / 32) < 0x7fffffffLL) {
      // fp32 x and gradient, 16 input channels (block 1): exact bf16 splits on the bf16 MFMA (conv3d_wgrad_cl_split_kernel; at
      // least as close to fp64 as the fp32-MFMA kernel: tests/test_gpu_round3.py).  Alone it is 3.5 ms against the fp32-MFMA
      // kernel's 4.35 at C3; in the training step a third of that arrives (interleaved x 3, round 5: 28.13-28.62 against
      // 28.43-28.77 ms) — the default since round 5; LIFTREG_WGRAD_SPLIT=0 selects the fp32-MFMA kernels.
      const int64_t nb32 = (int64_t)B * d.Do * d.Wo * ((d.Ho + 31) / 32);
      const unsigned grid = (unsigned)(nb32 < nblk ? nb32 : nblk);
      const bool hps = x_layout == LR_LAYOUT_NDHWC_HPS;
      if (Cout == 16) {
        if (hps) hipLaunchKernelGGL((conv3d_wgrad_cl_split_kernel<1, true>), dim3(grid), dim3(256), 0, st, x, gpre, partial, d, (int)nb32);
        else hipLaunchKernelGGL((conv3d_wgrad_cl_split_kernel<1, false>), dim3(grid), dim3(256), 0, st, x, gpre, partial, d, (int)nb32);
      } else {
        if (hps) hipLaunchKernelGGL((conv3d_wgrad_cl_split_kernel<2, true>), dim3(grid), dim3(256), 0, st, x, gpre, partial, d, (int)nb32);
        else hipLaunchKernelGGL((conv3d_wgrad_cl_split_kernel<2, false>), dim3(grid), dim3(256), 0, st, x, gpre, partial, d, (int)nb32);
      }
      nparts = (int)grid;
    } else if (nbricks < 0x7fffffffLL) {
      // two output rows per brick (15 staged window rows instead of 18, half the barriers per MFMA); LIFTREG_WGRAD_ROWS=1
      // selects the one-row bricks (A/B aid; partial sums then add in another order: equal to rounding)
      const int rows = (lr_sw_int(LR_SW_WGRAD_ROWS, 0) == 1) ? 1 : 2;
      const int64_t nbr = rows == 1 ? nbricks : (int64_t)B * d.Do * ((d.Wo + 1) / 2) * ((d.Ho + hb - 1) / hb);
      const unsigned grid = (unsigned)(nbr < nblk ? nbr : nblk);
      const bool hps = x_layout == LR_LAYOUT_NDHWC_HPS;
#define LR_WCL2(CBV, NTCV, HP, XBV, RV)                                                                                  \
  do {                                                                                                                   \
    constexpr size_t ldsb = WclGeom<CBV, NTCV, RV>::LDS_BYTES;                                                          \
    static std::atomic<uint64_t> attr_done{0}; /* one bit per device */                                                 \
    if (lr_raise_dyn_lds(reinterpret_cast<const void*>(&conv3d_wgrad_cl_kernel<CBV, NTCV, HP, XBV, RV>), ldsb, attr_done) != LR_OK) return LR_ELAUNCH;\
    hipLaunchKernelGGL((conv3d_wgrad_cl_kernel<CBV, NTCV, HP, XBV, RV>), dim3(grid), dim3(256 * CBV), ldsb, st, x, gpre, \
                       partial, d, (int)nbr, gbf);                                                                       \
  } while (0)
#define LR_WCL1(CBV, NTCV, HP, XBV) \
  do { if (rows == 1) LR_WCL2(CBV, NTCV, HP, XBV, 1); else LR_WCL2(CBV, NTCV, HP, XBV, 2); } while (0)
#define LR_WCL(CBV, NTCV)                                                       \
  do {                                                                          \
    if (hps) { if (xbf) LR_WCL1(CBV, NTCV, true, true); else LR_WCL1(CBV, NTCV, true, false); }     \
    else     { if (xbf) LR_WCL1(CBV, NTCV, false, true); else LR_WCL1(CBV, NTCV, false, false); }   \
  } while (0)
      if (Cin == 16 && Cout == 16) LR_WCL(1, 1);
      else if (Cin == 16) LR_WCL(1, 2);
      else if (Cout == 16) LR_WCL(2, 1);
      else LR_WCL(2, 2);
#undef LR_WCL
#undef LR_WCL1
#undef LR_WCL2
      nparts = (int)grid;
    }
  } else if (x_layout == LR_LAYOUT_NCDHW && stride == 1 && Cout == 16 && Cin <= 12 && H % 4 == 0 && al16 &&
             (V * Cin + 8 * (int64_t)W * H) * 4 < 0x7fffffffLL) {
    // the first block: planar input, the waves split the voxels of a 4-row brick
    const int64_t nbricks = (int64_t)B * D * ((W + 3) / 4) * ((H + 63) / 64);
    if (nbricks < 0x7fffffffLL) {
      const unsigned grid = (unsigned)(nbricks < nblk ? nbricks : nblk);
      if (gbf && xround && Cin <= 3 && !lr_sw_set(LR_SW_WGRAD0_COPIES)) {   // env: the three-copies kernel (A/B aid)
        hipLaunchKernelGGL(conv3d_wgrad_planar_bf16s_kernel, dim3(grid), dim3(256), 0, st, x, reinterpret_cast<const u16*>(gpre), partial, d, (int)nbricks);
        planar_tx = true;
      } else if (gbf && xround && Cin <= 3)
        hipLaunchKernelGGL(conv3d_wgrad_planar_bf16_kernel<6>, dim3(grid), dim3(256), 0, st, x, reinterpret_cast<const u16*>(gpre), partial, d, (int)nbricks);
      else if (gbf && xround)
        hipLaunchKernelGGL(conv3d_wgrad_planar_bf16_kernel<21>, dim3(grid), dim3(256), 0, st, x, reinterpret_cast<const u16*>(gpre), partial, d, (int)nbricks);
      else if (Cin <= 3) hipLaunchKernelGGL(conv3d_wgrad_planar_kernel<6>, dim3(grid), dim3(256), 0, st, x, gpre, partial, d, (int)nbricks, (int)xround, gbf);
      else hipLaunchKernelGGL(conv3d_wgrad_planar_kernel<21>, dim3(grid), dim3(256), 0, st, x, gpre, partial, d, (int)nbricks, (int)xround, gbf);
      nparts = (int)grid * 4;
    }
  }
  if (nparts) {
    if (int e = lr_launch_status()) return e;
    // fast-path partials carry one extra column/tile: the sum of gpre (bias gradient)
    const int planar = x_layout == LR_LAYOUT_NCDHW;
    const int ncols = planar_tx ? 96 : planar ? d.ntiles * 16 : (d.ntiles + 1) * 16;
    const int n = Cout * ncols;
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3((n + 63) / 64), dim3(1024), 0, st, partial, gw, gb, nparts, Cout, Cin,
                       ncols, planar_tx ? LR_WG_PLANAR_TX : x_layout, planar_tx ? 32 + 27 : planar ? 27 * Cin : d.ntiles * 16);
    return lr_launch_status();
  }
  if (xbf) return LR_EUNSUPPORTED;  // the generic kernel reads fp32 activations (bf16 gradients and the rounded first-block input are fine)
  if (Cout == 16) hipLaunchKernelGGL(conv3d_wgrad_kernel<1>, dim3((unsigned)nblk), dim3(256), 0, st, x, gpre, partial, d, ngroups, (int)xround, gbf);
  else hipLaunchKernelGGL(conv3d_wgrad_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, st, x, gpre, partial, d, ngroups, (int)xround, gbf);
  if (int e = lr_launch_status()) return e;
  const int n = Cout * d.ntiles * 16;
  hipLaunchKernelGGL(wgrad_finish_kernel, dim3((n + 63) / 64), dim3(1024), 0, st, partial, gw, (float*)nullptr, nblk,
                     Cout, Cin, d.ntiles * 16, x_layout, -1);
  if (int e = lr_launch_status()) return e;
  if (gb) {  // generic path: the bias gradient from its own reduction (partial is free again: stream order)
    const int64_t nvox = (int64_t)B * d.Do * d.Wo * d.Ho;
    int64_t cap = (int64_t)nblk * d.ntiles * 16;  // rows of Cout floats the workspace holds
    if (cap > 1024) cap = 1024;
    const int nb = (int)(nvox < cap ? nvox : cap);
    hipLaunchKernelGGL(channel_sum_kernel, dim3(nb), dim3(256), 0, st, gpre, partial, nvox, Cout, gbf);
    if (int e = lr_launch_status()) return e;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, partial, gb, nb, Cout);
    return lr_launch_status();
  }
  return LR_OK;
}

extern "C" int lr_conv3d_wgrad_f32(const float* x, int x_layout, const float* gpre, float* partial, float* gw,
                                   float* gb, int B, int Cin, int Cout, int D, int W, int H, int stride, int nblk,
                                   void* stream) {
  return wgrad_impl(x, x_layout, gpre, 0, partial, gw, gb, B, Cin, Cout, D, W, H, stride, nblk, stream);
}

// Same with the pre-activation gradient stored as bf16 plain channels-last (the bf16-gradient training variant).
extern "C" int lr_conv3d_wgrad_bf16g_f32(const float* x, int x_layout, const void* gpre_bf16, float* partial, float* gw,
                                         float* gb, int B, int Cin, int Cout, int D, int W, int H, int stride,
                                         int nblk, void* stream) {
  return wgrad_impl(x, x_layout, reinterpret_cast<const float*>(gpre_bf16), 1, partial, gw, gb, B, Cin, Cout, D, W, H,
                    stride, nblk, stream);
}
