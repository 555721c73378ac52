// backproject.hip — K2: voxel-driven backprojection of P 2D views into a
// (B,P,D,W,H) feature volume.  HBM-write-bound: 4*P*V bytes out per sample,
// the views themselves (4*P*Pw*Ph bytes) stay L2-resident.
//
// Replaces (reference file:line)
//   src/liftreg/utils/sdct_projection_utils.py:227-250  backproj_grids_with_poses
//   src/liftreg/models/LiftRegDeformSubspaceBackproj.py:85-93  F.grid_sample 2D
//
// The reference materialises a (1,P,2,D,W,H) grid once and samples it for every
// batch; here each thread derives the shadow of its 4 consecutive-H voxels in
// registers (same fp32 op order, no contraction, IEEE divide — so floor() picks
// the same detector pixel) and re-uses it across the whole batch.
#include "lr_common.h"

namespace {

// Pixel coordinate along the Pw axis (detector rows) of voxel row x, plane y.
// grids = (x - ex) * scale + ex ; /proj_w * 2.0 ; ATen un-normalise.
__device__ __forceinline__ float shadow_grid(float x, float e, float scale, float fsize) {
  float g = (x - e) * scale;  // torch.mul(grids - poses, scale)
  g = g + e;                  // + poses[:, :, ::2]
  g = g / fsize;              // grids[:, :, c] / proj_w
  g = g * 2.0f;               // * 2.0
  return g;
}
__device__ __forceinline__ float shadow_pix(float x, float e, float scale, float fsize, int size) {
  return lr_unnormalize(shadow_grid(x, e, scale, fsize), size);
}

struct Tap {
  int i0, i1;    // clamped indices (safe to dereference)
  float w0, w1;  // weights of floor / floor+1, zeroed when out of range
};

// ATen's vectorised 2D bilinear (GridSamplerKernel.cpp): w = x - floor(x),
// e = 1 - w; out-of-range corners are dropped individually (padding 'zeros').
__device__ __forceinline__ Tap make_tap(float pix, int size) {
  Tap t;
  if (!(pix > -1.0f && pix < (float)size)) {  // also catches NaN
    t.i0 = t.i1 = 0;
    t.w0 = t.w1 = 0.0f;
    return t;
  }
  const float fl = floorf(pix);
  const float w = pix - fl;
  const float e = 1.0f - w;
  const int i0 = (int)fl, i1 = i0 + 1;
  t.w0 = (i0 >= 0) ? e : 0.0f;          // i0 < size is implied by pix < size
  t.w1 = (i1 < size) ? w : 0.0f;        // i1 >= 0 is implied by pix > -1
  t.i0 = max(i0, 0);
  t.i1 = min(i1, size - 1);
  return t;
}

template <int VEC>
__global__ __launch_bounds__(256) void backproject_kernel(
    const float* __restrict__ proj, LrPoses poses, float* __restrict__ out,
    int B, int P, int Pw, int Ph, int D, int W, int H, int d0, int Ds,
    int64_t out_batch_stride) {
  const int HV = H / VEC;
  const int64_t total = (int64_t)P * Ds * W * HV;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int kv = (int)(idx % HV);
  int64_t r = idx / HV;
  const int j = (int)(r % W);
  r /= W;
  const int i = (int)(r % Ds);
  const int p = (int)(r / Ds);

  const float ex = poses.e[p][0], ey = poses.e[p][1], ez = poses.e[p][2];
  // x = linspace(-d/2, d/2-1, d)[d0+i]; y = linspace(w-1, 0, w)[j]; z likewise.
  const float x = (float)(d0 + i) - 0.5f * (float)D;
  const float y = (float)(W - 1 - j);
  const float scale = ey / (ey - y);  // poses_y / (poses_y - grid_y)

  const Tap ty = make_tap(shadow_pix(x, ex, scale, (float)Pw, Pw), Pw);
  Tap tx[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    const float z = (float)(kv * VEC + v) - 0.5f * (float)H;
    tx[v] = make_tap(shadow_pix(z, ez, scale, (float)Ph, Ph), Ph);
  }
  // nw = s*e, ne = s*w, sw = n*e, se = n*w   (s,n = row weights)
  float nw[VEC], ne[VEC], sw[VEC], se[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    nw[v] = ty.w0 * tx[v].w0;
    ne[v] = ty.w0 * tx[v].w1;
    sw[v] = ty.w1 * tx[v].w0;
    se[v] = ty.w1 * tx[v].w1;
  }
  const int64_t view_sz = (int64_t)Pw * Ph;
  const float* r0 = proj + (int64_t)p * view_sz + (int64_t)ty.i0 * Ph;
  const float* r1 = proj + (int64_t)p * view_sz + (int64_t)ty.i1 * Ph;
  float* o = out + (((int64_t)p * Ds + i) * W + j) * H + (int64_t)kv * VEC;
  const int64_t pstride = (int64_t)P * view_sz;
  for (int b = 0; b < B; ++b) {
    float res[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const float a = r0[tx[v].i0], bq = r0[tx[v].i1];
      const float c = r1[tx[v].i0], d = r1[tx[v].i1];
      float acc = a * nw[v];
      acc = acc + bq * ne[v];
      acc = acc + c * sw[v];
      acc = acc + d * se[v];
      res[v] = acc;
    }
    if constexpr (VEC == 4) {
      *reinterpret_cast<float4*>(o) = make_float4(res[0], res[1], res[2], res[3]);
    } else {
      o[0] = res[0];
    }
    r0 += pstride;
    r1 += pstride;
    o += out_batch_stride;
  }
}

// ---------------------------------------------------------------------------
// Tiled variant (the one that runs for every shape the model uses).
// Block = one emitter p, TI=8 voxel rows (D axis) x TJ=4 coronal planes x all H.
// Per batch element the <=RCAP detector rows the tile's shadows touch are staged
// once into LDS (16-byte coalesced loads), zero-padded by PAD columns left/right
// and by zero rows above/below the view, so 'zeros' padding needs no per-corner
// masks: an out-of-view corner simply reads 0.  Lanes run along H: every tap is a
// conflict-light ds_read2_b32 of two neighbouring columns and every store is a
// fully coalesced 256-byte wavefront store.  Arithmetic (weights, 4-term sum
// order) is identical to backproject_kernel above, so results are bit-identical.
constexpr int BT_TI = 8, BT_TJ = 4, BT_RCAP = 18, BT_PAD = 4;
constexpr int BT_SENTINEL = -0x40000000;

struct TapU {
  int i0;      // floor(pix) in [-1, size-1]; BT_SENTINEL when the whole footprint is outside
  float e, w;  // weights of i0 / i0+1 (unmasked: data outside the view is 0 in the tile)
};
__device__ __forceinline__ TapU make_tap_u(float pix, int size) {
  TapU t;
  if (!(pix > -1.0f && pix < (float)size)) {
    t.i0 = BT_SENTINEL;
    t.e = t.w = 0.0f;
    return t;
  }
  const float fl = floorf(pix);
  t.w = pix - fl;
  t.e = 1.0f - t.w;
  t.i0 = (int)fl;
  return t;
}

template <int KC, bool VEC4>
__global__ __launch_bounds__(256) void backproject_tiled_kernel(
    const float* __restrict__ proj, LrPoses poses, float* __restrict__ out, int B, int P, int Pw,
    int Ph, int D, int W, int H, int d0, int Ds, int64_t out_batch_stride) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int RS = Ph + 2 * BT_PAD;                 // padded row length (floats)
  float* tile = smem;                             // [BT_RCAP][RS]
  float4* tyt = reinterpret_cast<float4*>(smem + BT_RCAP * RS);  // [TJ*TI] {row offset bits, e, w, -}
  __shared__ int s_lo, s_hi;
  const int tid = threadIdx.x;
  const int nI = (Ds + BT_TI - 1) / BT_TI, nJ = (W + BT_TJ - 1) / BT_TJ;
  const int jt = blockIdx.x % nJ, it = (blockIdx.x / nJ) % nI, p = blockIdx.x / nJ / nI;
  const int i_base = it * BT_TI, j_base = jt * BT_TJ;
  const float ex = poses.e[p][0], ey = poses.e[p][1], ez = poses.e[p][2];

  if (tid == 0) { s_lo = 0x7fffffff; s_hi = -0x7fffffff; }
  __syncthreads();
  TapU myty;
  myty.i0 = BT_SENTINEL; myty.e = myty.w = 0.0f;
  if (tid < BT_TI * BT_TJ) {
    const int ii = tid % BT_TI, jj = tid / BT_TI;
    const int i = i_base + ii, j = j_base + jj;
    if (i < Ds && j < W) {
      const float x = (float)(d0 + i) - 0.5f * (float)D;
      const float y = (float)(W - 1 - j);
      const float scale = ey / (ey - y);
      myty = make_tap_u(shadow_pix(x, ex, scale, (float)Pw, Pw), Pw);
      if (myty.i0 != BT_SENTINEL) {
        atomicMin(&s_lo, myty.i0);
        atomicMax(&s_hi, myty.i0 + 1);
      }
    }
  }
  __syncthreads();
  const int r_lo = s_lo, r_hi = s_hi;
  const bool any = r_hi >= r_lo;
  const int nrows = any ? r_hi - r_lo + 1 : 0;
  const bool direct = nrows > BT_RCAP;  // block-uniform: geometry too oblique for the tile
  if (tid < BT_TI * BT_TJ) {
    const int rel = (myty.i0 == BT_SENTINEL) ? 0 : (myty.i0 - r_lo) * RS;
    tyt[tid] = make_float4(__int_as_float(rel), myty.e, myty.w, 0.0f);
  }
  // column taps of this thread's k (per plane j): same for every row i and every batch element
  TapU txs[KC][BT_TJ];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    const int k = tid + kc * 256;
    const float z = (float)k - 0.5f * (float)H;
#pragma unroll
    for (int jj = 0; jj < BT_TJ; ++jj) {
      const float y = (float)(W - 1 - (j_base + jj));
      const float scale = ey / (ey - y);
      txs[kc][jj] = make_tap_u(shadow_pix(z, ez, scale, (float)Ph, Ph), Ph);
      if (txs[kc][jj].i0 == BT_SENTINEL) txs[kc][jj].i0 = -1;  // weights are 0; any padded column
    }
  }
  __syncthreads();

  if (!any) {  // every shadow of this tile misses the detector: exact zeros, nothing to stage or read
    for (int b = 0; b < B; ++b)
      for (int jj = 0; jj < BT_TJ && j_base + jj < W; ++jj)
        for (int ii = 0; ii < BT_TI && i_base + ii < Ds; ++ii)
          for (int k = tid; k < H; k += 256)
            out[(int64_t)b * out_batch_stride + (((int64_t)p * Ds + i_base + ii) * W + j_base + jj) * H + k] = 0.0f;
    return;
  }
  const int64_t view_sz = (int64_t)Pw * Ph;
  for (int b = 0; b < B; ++b) {
    const float* pv = proj + ((int64_t)b * P + p) * view_sz;
    if (!direct) {
      if constexpr (VEC4) {
        const int RS4 = RS >> 2;
        for (int idx = tid; idx < nrows * RS4; idx += 256) {
          const int row = idx / RS4, c4 = idx - row * RS4;
          const int grow = r_lo + row, col = c4 * 4 - BT_PAD;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (grow >= 0 && grow < Pw && col >= 0 && col < Ph)
            v = *reinterpret_cast<const float4*>(pv + (int64_t)grow * Ph + col);
          *reinterpret_cast<float4*>(tile + row * RS + c4 * 4) = v;
        }
      } else {
        for (int idx = tid; idx < nrows * RS; idx += 256) {
          const int row = idx / RS, c = idx - row * RS;
          const int grow = r_lo + row, col = c - BT_PAD;
          tile[idx] = (grow >= 0 && grow < Pw && col >= 0 && col < Ph) ? pv[(int64_t)grow * Ph + col] : 0.0f;
        }
      }
      __syncthreads();
    }
    float* ob = out + (int64_t)b * out_batch_stride;
#pragma unroll
    for (int jj = 0; jj < BT_TJ; ++jj) {
      const int j = j_base + jj;
      if (j >= W) break;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const int k = tid + kc * 256;
        if (k >= H) continue;
        const TapU tx = txs[kc][jj];
        const int cb = tx.i0 + BT_PAD;
        for (int ii = 0; ii < BT_TI; ++ii) {
          const int i = i_base + ii;
          if (i >= Ds) break;
          const float4 ty = tyt[jj * BT_TI + ii];
          const float nw = ty.y * tx.e, ne = ty.y * tx.w, sw = ty.z * tx.e, se = ty.z * tx.w;
          float a, bq, c, d;
          if (!direct) {
            const float* r0 = tile + __float_as_int(ty.x) + cb;
            a = r0[0]; bq = r0[1]; c = r0[RS]; d = r0[RS + 1];
          } else {  // same values straight from the view (zero outside it)
            const int row0 = __float_as_int(ty.x) / RS + r_lo, c0 = tx.i0;
            const bool ry0 = row0 >= 0 && row0 < Pw, ry1 = row0 + 1 >= 0 && row0 + 1 < Pw;
            const bool cx0 = c0 >= 0 && c0 < Ph, cx1 = c0 + 1 >= 0 && c0 + 1 < Ph;
            const float* g0 = pv + (int64_t)max(min(row0, Pw - 1), 0) * Ph;
            const float* g1 = pv + (int64_t)max(min(row0 + 1, Pw - 1), 0) * Ph;
            const int q0 = max(min(c0, Ph - 1), 0), q1 = max(min(c0 + 1, Ph - 1), 0);
            a = (ry0 && cx0) ? g0[q0] : 0.0f;
            bq = (ry0 && cx1) ? g0[q1] : 0.0f;
            c = (ry1 && cx0) ? g1[q0] : 0.0f;
            d = (ry1 && cx1) ? g1[q1] : 0.0f;
          }
          float acc = a * nw;
          acc = acc + bq * ne;
          acc = acc + c * sw;
          acc = acc + d * se;
          __builtin_nontemporal_store(acc, &ob[(((int64_t)p * Ds + i) * W + j) * H + k]);  // written once, read by block 0 long after it left the L2
        }
      }
    }
    if (!direct) __syncthreads();  // tile is restaged for the next batch element
  }
}

__global__ __launch_bounds__(256) void backproject_coords_kernel(
    LrPoses poses, float* __restrict__ pix, int P, int Pw, int Ph, int D, int W, int H,
    int normalized) {
  const int64_t total = (int64_t)P * D * W * H;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int k = (int)(idx % H);
  int64_t r = idx / H;
  const int j = (int)(r % W);
  r /= W;
  const int i = (int)(r % D);
  const int p = (int)(r / D);
  const float ex = poses.e[p][0], ey = poses.e[p][1], ez = poses.e[p][2];
  const float x = (float)i - 0.5f * (float)D;
  const float y = (float)(W - 1 - j);
  const float z = (float)k - 0.5f * (float)H;
  const float scale = ey / (ey - y);
  const float gu = shadow_grid(x, ex, scale, (float)Pw), gv = shadow_grid(z, ez, scale, (float)Ph);
  pix[idx * 2 + 0] = normalized ? gu : lr_unnormalize(gu, Pw);
  pix[idx * 2 + 1] = normalized ? gv : lr_unnormalize(gv, Ph);
}

// backproj_grids (sdct_projection_utils.py:179-202), the pose-less variant: the reference mixes a float64 pose array
// with float32 linspaces, so torch promotes and the whole grid is FLOAT64, built as scale·g + trans (mul, then add),
// not as (g - e)·s + e like the with-poses variant.  One thread per (p, i, j, k); both channels; channel order after
// the reference's flip(1): out[p,0] = the Ph-axis (z) coordinate, out[p,1] = the Pw-axis (x) coordinate.
struct LrPoses64 {
  double e[LR_MAX_VIEWS][3];
};
__global__ __launch_bounds__(256) void backproject_coords_poseless_f64_kernel(LrPoses64 poses, double* __restrict__ grid,
                                                                              int P, int Pw, int Ph, int D, int W, int H) {
  const int64_t plane = (int64_t)D * W * H, total = (int64_t)P * plane;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int k = (int)(idx % H);
  int64_t r = idx / H;
  const int j = (int)(r % W);
  r /= W;
  const int i = (int)(r % D);
  const int p = (int)(r / D);
  // torch.linspace(-d/2, d/2-1, d) etc. in float32: exact integers (or half-integers for odd sizes) — step is ±1
  const double x = (double)((float)i - 0.5f * (float)D), y = (double)(float)(W - 1 - j), z = (double)((float)k - 0.5f * (float)H);
  const double ex = poses.e[p][0], ey = poses.e[p][1], ez = poses.e[p][2];
  const double den = ey - y;
  const double scale = ey / den;              // :194
  const double ty = (-y) / den;               // :195  poses[:,0::2] * (-y/(e_y - y))
  const double tx = ex * ty, tz = ez * ty;
  double gu = __dadd_rn(__dmul_rn(scale, x), tx);   // :197  torch.mul(scale, grids) + trans — two roundings, no FMA
  double gv = __dadd_rn(__dmul_rn(scale, z), tz);
  gu = gu / (double)Pw * 2.0;                 // :198-199
  gv = gv / (double)Ph * 2.0;
  const int64_t o = (int64_t)p * 2 * plane + (idx - (int64_t)p * plane);
  grid[o] = gv;                               // flip(1) :201
  grid[o + plane] = gu;
}

int fill_poses(LrPoses& lp, const float* poses, int P) {
  if (!poses) return LR_ENULL;
  if (P < 1 || P > LR_MAX_VIEWS) return LR_EINVAL;
  for (int p = 0; p < P; ++p)
    for (int c = 0; c < 3; ++c) lp.e[p][c] = poses[p * 3 + c];
  return LR_OK;
}

}  // namespace

extern "C" int lr_backproject_f32(const float* proj, const float* poses, float* out,
                                  int B, int P, int Pw, int Ph, int D, int W, int H,
                                  int d0, int d1, int64_t out_batch_stride, void* stream) {
  if (!proj || !out) return LR_ENULL;
  if (B < 1 || Pw < 1 || Ph < 1 || D < 1 || W < 1 || H < 1) return LR_EINVAL;
  if (d0 < 0 || d1 > D || d1 <= d0) return LR_EINVAL;
  LrPoses lp;
  if (int e = fill_poses(lp, poses, P)) return e;
  const int Ds = d1 - d0;
  if (out_batch_stride < (int64_t)P * Ds * W * H) return LR_EINVAL;
  // Tiled kernel: H <= 1024 (<= 4 columns per thread) and the staged rows fit in LDS.
  const size_t tile_lds = ((size_t)BT_RCAP * (Ph + 2 * BT_PAD) + 4 * BT_TI * BT_TJ) * sizeof(float);
  if (H <= 1024 && tile_lds <= 64 * 1024) {
    const int64_t nb = (int64_t)P * ((Ds + BT_TI - 1) / BT_TI) * ((W + BT_TJ - 1) / BT_TJ);
    if (nb > 0x7fffffffLL) return LR_EINVAL;
    const bool v4 = (Ph % 4 == 0) && ((reinterpret_cast<uintptr_t>(proj) & 15u) == 0);
    const int KC = (H + 255) / 256;
    const dim3 grid((unsigned)nb), block(256);
    hipStream_t st = lr_stream(stream);
#define LR_BT(KCV)                                                                                   \
  do {                                                                                               \
    if (v4) hipLaunchKernelGGL((backproject_tiled_kernel<KCV, true>), grid, block, tile_lds, st, proj, lp, out, B, P, Pw, Ph, D, W, H, d0, Ds, out_batch_stride); \
    else hipLaunchKernelGGL((backproject_tiled_kernel<KCV, false>), grid, block, tile_lds, st, proj, lp, out, B, P, Pw, Ph, D, W, H, d0, Ds, out_batch_stride);    \
  } while (0)
    if (KC == 1) LR_BT(1);
    else if (KC == 2) LR_BT(2);
    else LR_BT(4);
#undef LR_BT
    return lr_launch_status();
  }
  const bool vec4 = (H % 4 == 0) && (out_batch_stride % 4 == 0) &&
                    ((reinterpret_cast<uintptr_t>(out) & 15u) == 0);
  const int64_t total = (int64_t)P * Ds * W * (vec4 ? H / 4 : H);
  const int64_t nblk = (total + 255) / 256;
  if (nblk > 0x7fffffffLL) return LR_EINVAL;
  if (vec4)
    hipLaunchKernelGGL(backproject_kernel<4>, dim3((unsigned)nblk), dim3(256), 0, lr_stream(stream),
                       proj, lp, out, B, P, Pw, Ph, D, W, H, d0, Ds, out_batch_stride);
  else
    hipLaunchKernelGGL(backproject_kernel<1>, dim3((unsigned)nblk), dim3(256), 0, lr_stream(stream),
                       proj, lp, out, B, P, Pw, Ph, D, W, H, d0, Ds, out_batch_stride);
  return lr_launch_status();
}

extern "C" int lr_backproject_coords_f32(const float* poses, float* pix, int P, int Pw, int Ph,
                                         int D, int W, int H, int normalized, void* stream) {
  if (!pix) return LR_ENULL;
  if (Pw < 1 || Ph < 1 || D < 1 || W < 1 || H < 1) return LR_EINVAL;
  LrPoses lp;
  if (int e = fill_poses(lp, poses, P)) return e;
  const int64_t total = (int64_t)P * D * W * H;
  const int64_t nblk = (total + 255) / 256;
  if (nblk > 0x7fffffffLL) return LR_EINVAL;
  hipLaunchKernelGGL(backproject_coords_kernel, dim3((unsigned)nblk), dim3(256), 0,
                     lr_stream(stream), lp, pix, P, Pw, Ph, D, W, H, normalized);
  return lr_launch_status();
}

extern "C" int lr_backproject_coords_poseless_f64(const double* poses, double* grid, int P, int Pw, int Ph, int D, int W,
                                                  int H, void* stream) {
  if (!poses || !grid) return LR_ENULL;
  if (P < 1 || P > LR_MAX_VIEWS || Pw < 1 || Ph < 1 || D < 1 || W < 1 || H < 1) return LR_EINVAL;
  LrPoses64 lp;
  for (int p = 0; p < P; ++p)
    for (int c = 0; c < 3; ++c) lp.e[p][c] = poses[p * 3 + c];
  const int64_t total = (int64_t)P * D * W * H;
  const int64_t nblk = (total + 255) / 256;
  if (nblk > 0x7fffffffLL) return LR_EINVAL;
  hipLaunchKernelGGL(backproject_coords_poseless_f64_kernel, dim3((unsigned)nblk), dim3(256), 0, lr_stream(stream), lp,
                     grid, P, Pw, Ph, D, W, H);
  return lr_launch_status();
}
