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
// Block = one emitter p, TI=8 voxel rows (D axis) x TJ=4 coronal planes x all H, for a CHUNK of the batch (blockIdx.y): the
// reference's shipped shape (160^3, 4 views, batch 30) gave 3,200 blocks of 30 elements each on 2,048 resident slots — a
// second round 56 % full; chunks of ~8 elements give 12,800 short blocks.  Rows shorter than 256 voxels: JP = 2 or 4 of the
// tile's planes side by side (thread = (plane, k); H = 160: 320 threads, all lanes busy, instead of 192 with 32 idle).
// Per batch element the <=RCAP detector rows the tile's shadows touch are staged
// once into LDS (16-byte coalesced loads), zero-padded by PAD columns left/right
// and by zero rows above/below the view, so 'zeros' padding needs no per-corner
// masks: an out-of-view corner simply reads 0.  Lanes run along H: every tap is a
// conflict-light ds_read2_b32 of two neighbouring columns and every store is a
// fully coalesced 256-byte wavefront store.  Arithmetic (weights, 4-term sum
// order) is identical to backproject_kernel above, so results are bit-identical.
#ifndef LR_BP_NT
#define LR_BP_NT 1
#endif
#ifndef LR_BP_TI
#define LR_BP_TI 8
#endif
#ifndef LR_BP_TJ
#define LR_BP_TJ 4
#endif
constexpr int BT_TI = LR_BP_TI, BT_TJ = LR_BP_TJ, BT_RCAP = 18, BT_PAD = 4;
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

// One streaming pass over the views in front of the tiled kernel (round 6).  The views are an INPUT of the step: by the time the
// kernel runs, the other kernels' streams have taken them out of the L2s and the memory-side cache, and every block would fetch its
// detector rows on demand — eight XCDs each miss every row once, underneath a saturated write stream, with the staging latency in
// the open (185 MB of fetches for 27.6 MB of views at the reference's shape).  Measured: after any other kernel the tiled kernel
// took 0.62 ms (160^3, 4 views, B = 30) / 0.254 ms (C3); with `views.sum()` in between 0.458 / 0.214 — the back-to-back loop's time.
// 27.6 MB read once at HBM speed is ~10 us.  Plain loads (they should stay in the caches); nothing is written.
__global__ __launch_bounds__(256) void touch_views_kernel(const float* __restrict__ p, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  float acc = 0.0f;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i + 3 < n; i += stride) {
    const float4 v = *reinterpret_cast<const float4*>(p + i);
    acc += (v.x + v.y) + (v.z + v.w);
  }
  asm volatile("" ::"v"(acc));   // the loads stay
}

template <int KC, bool VEC4, int JP>
__global__ __launch_bounds__(JP == 1 ? 512 : 384) void backproject_tiled_kernel(
    const float* __restrict__ proj, LrPoses poses, float* __restrict__ out, int B, int P, int Pw,
    int Ph, int D, int W, int H, int d0, int Ds, int64_t out_batch_stride, int bchunk) {
  static_assert(JP == 1 || KC == 1, "planes side by side: rows of at most one column per thread");
  constexpr int NJ = BT_TJ / JP;                  // planes per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int RS = Ph + 2 * BT_PAD;                 // padded row length (floats)
  float* tile = smem;                             // [BT_RCAP][RS]
  float4* tyt = reinterpret_cast<float4*>(smem + BT_RCAP * RS);  // [TJ*TI] {row offset bits, e, w, -}
  __shared__ int s_lo, s_hi;
  const int tid = threadIdx.x;
  const int NT = blockDim.x;   // JP = 1: the row length / KC rounded up to whole waves (<= 512); JP > 1: JP * H
  const int NK = JP == 1 ? NT : H;                // threads along a row
  const int kt = JP == 1 ? tid : tid % H, jsub = JP == 1 ? 0 : tid / H;   // this thread's column | its first plane of the tile
  const int b_lo = (int)blockIdx.y * bchunk, b_hi = min(B, b_lo + bchunk);
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
  TapU txs[KC][NJ];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    const int k = kt + kc * NK;
    const float z = (float)k - 0.5f * (float)H;
#pragma unroll
    for (int jn = 0; jn < NJ; ++jn) {
      const int jj = jsub + jn * JP;
      const float y = (float)(W - 1 - (j_base + jj));
      const float scale = ey / (ey - y);
      txs[kc][jn] = make_tap_u(shadow_pix(z, ez, scale, (float)Ph, Ph), Ph);
      if (txs[kc][jn].i0 == BT_SENTINEL) txs[kc][jn].i0 = -1;  // weights are 0; any padded column
    }
  }
  __syncthreads();

  if (!any) {  // every shadow of this tile misses the detector: exact zeros, nothing to stage or read
    for (int b = b_lo; b < b_hi; ++b)
      for (int jj = 0; jj < BT_TJ && j_base + jj < W; ++jj)
        for (int ii = 0; ii < BT_TI && i_base + ii < Ds; ++ii)
          for (int k = tid; k < H; k += NT)
            out[(int64_t)b * out_batch_stride + (((int64_t)p * Ds + i_base + ii) * W + j_base + jj) * H + k] = 0.0f;
    return;
  }
  const int64_t view_sz = (int64_t)Pw * Ph;
  for (int b = b_lo; b < b_hi; ++b) {
    const float* pv = proj + ((int64_t)b * P + p) * view_sz;
    if (!direct) {
      if constexpr (VEC4) {
        const int RS4 = RS >> 2;
        for (int idx = tid; idx < nrows * RS4; idx += NT) {
          const int row = idx / RS4, c4 = idx - row * RS4;
          const int grow = r_lo + row, col = c4 * 4 - BT_PAD;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (grow >= 0 && grow < Pw && col >= 0 && col < Ph)
            v = *reinterpret_cast<const float4*>(pv + (int64_t)grow * Ph + col);
          *reinterpret_cast<float4*>(tile + row * RS + c4 * 4) = v;
        }
      } else {
        for (int idx = tid; idx < nrows * RS; idx += NT) {
          const int row = idx / RS, c = idx - row * RS;
          const int grow = r_lo + row, col = c - BT_PAD;
          tile[idx] = (grow >= 0 && grow < Pw && col >= 0 && col < Ph) ? pv[(int64_t)grow * Ph + col] : 0.0f;
        }
      }
      __syncthreads();
    }
    float* ob = out + (int64_t)b * out_batch_stride;
#pragma unroll
    for (int jn = 0; jn < NJ; ++jn) {
      const int jj = jsub + jn * JP;
      const int j = j_base + jj;
      if (j >= W) break;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const int k = kt + kc * NK;
        if (k >= H) continue;
        const TapU tx = txs[kc][jn];
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
#if LR_BP_NT
          __builtin_nontemporal_store(acc, &ob[(((int64_t)p * Ds + i) * W + j) * H + k]);  // written once, read by block 0 long after it left the L2
#else
          ob[(((int64_t)p * Ds + i) * W + j) * H + k] = acc;
#endif
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

// ---- The encoder input of the bf16 variant with MANY views (BASELINE config C4: 11) written directly in the layout its
// first block stages: (B, Ds, W, H, 16) bf16 channels-last records [moving | view 0 .. view NV-1 | zeros] — what
// cat([moving, target_volume], dim=1) (…Backproj.py:95-98) holds, rounded to nearest-even bf16 (the rounding the first
// block applies to its input anyway: the conv's results keep their bits).  The fp32 (B,P,D,W,H) feature volume — 2.95 GB
// written and 1.6x that read back at C4 — is never materialised: 32 bytes per voxel instead of 48 + 48.
// Structure of backproject_tiled_kernel (same arithmetic, same 4-term order: bit-identical samples), turned inside out: a
// block owns a 4 x 4 (D x W) bundle of H-rows and ALL views — the <= 8 detector rows each view's shadows touch are staged
// together in zero-padded LDS (one barrier pair per batch element, not per view; 93 KB for 11 views of 256 columns), a
// thread owns the column k of half the bundle (two of its four planes), walks the NV views of a voxel with LDS taps,
// assembles the 32-byte record in registers and stores it: lanes run along H, a wave's records are one contiguous 2 KiB
// store.  (Measured on the way: global-memory taps — 176 scattered dword loads per thread — 3.5 ms for C4's 0.45 ms of
// writes; a 2 x 2 bundle — one staged float per sample — 1.7 ms: the bundle must amortise its tile like
// backproject_tiled_kernel's 8 x 4 one does.)
typedef unsigned bp_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned bp_bf16(float v) {
  const __bf16 h = (__bf16)v;
  return (unsigned)__builtin_bit_cast(unsigned short, h);
}
#ifndef LR_BE_SHAPE
#define LR_BE_SHAPE 0
#endif
#if LR_BE_SHAPE == 0   // 4 x 4 bundle, one 512-thread block per CU (105 KB of tiles), next element's tiles prefetched into registers
constexpr int BE_TI = 4, BE_TJ = 4, BE_RCAP = 9, BE_THREADS = 512, BE_NPF = 13;
constexpr bool BE_PREFETCH = true;
#else                  // 2 x 4 bundle, two 256-thread blocks per CU (70 KB each): one stages while the other computes
constexpr int BE_TI = 2, BE_TJ = 4, BE_RCAP = 6, BE_THREADS = 256, BE_NPF = 1;
constexpr bool BE_PREFETCH = false;
#endif
template <int NV>
__global__ __launch_bounds__(BE_THREADS) void backproject_encin_bf16_kernel(
    const float* __restrict__ proj, const float* __restrict__ moving, LrPoses poses, unsigned short* __restrict__ out,
    int B, int Pw, int Ph, int D, int W, int H, int d0, int Ds, int64_t out_bs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int RS = Ph + 2 * BT_PAD;                  // padded row length (floats)
  float* tile = smem;                              // [NV][BE_RCAP][RS]
  const int tile_floats = BE_PREFETCH ? ((NV * BE_RCAP * (RS >> 2) + BE_THREADS - 1) / BE_THREADS) * BE_THREADS * 4 : NV * BE_RCAP * RS;
  float4* tyt = reinterpret_cast<float4*>(smem + tile_floats);   // [NV][TJ*TI] {row offset, e, w, -}
  __shared__ int s_lo[NV], s_hi[NV];
  const int tid = threadIdx.x;
  const int nJ = (W + BE_TJ - 1) / BE_TJ;
  const int jt = blockIdx.x % nJ, it = blockIdx.x / nJ;
  const int i_base = it * BE_TI, j_base = jt * BE_TJ;
  if (tid < NV) { s_lo[tid] = 0x7fffffff; s_hi[tid] = -0x7fffffff; }
  __syncthreads();
  TapU myty;
  myty.i0 = BT_SENTINEL; myty.e = myty.w = 0.0f;
  const int tv = tid / (BE_TI * BE_TJ), tpos = tid % (BE_TI * BE_TJ);   // threads 0 .. 4 NV - 1: (view, position of the bundle)
  if (tv < NV) {
    const int ii = tpos % BE_TI, jj = tpos / BE_TI;
    const int i = i_base + ii, j = j_base + jj;
    if (i < Ds && j < W) {
      const float ex = poses.e[tv][0], ey = poses.e[tv][1];
      const float x = (float)(d0 + i) - 0.5f * (float)D;
      const float y = (float)(W - 1 - j);
      const float scale = ey / (ey - y);
      myty = make_tap_u(shadow_pix(x, ex, scale, (float)Pw, Pw), Pw);
      if (myty.i0 != BT_SENTINEL) {
        atomicMin(&s_lo[tv], myty.i0);
        atomicMax(&s_hi[tv], myty.i0 + 1);
      }
    }
  }
  __syncthreads();
  if (tv < NV) {
    const int rel = (myty.i0 == BT_SENTINEL) ? 0 : (myty.i0 - s_lo[tv]) * RS;
    tyt[tid] = make_float4(__int_as_float(rel), myty.e, myty.w, 0.0f);
  }
  // a view whose shadows span more detector rows than the tile holds (oblique geometry): the whole block takes the
  // slow path below — per-sample taps straight from the views, 2-byte stores; block-uniform, never on the model's path
  __shared__ int s_oblique;
  if (tid == 0) {
    int ob = 0;
    for (int v = 0; v < NV; ++v) ob |= (s_hi[v] >= s_lo[v] && s_hi[v] - s_lo[v] + 1 > BE_RCAP) ? 1 : 0;
    s_oblique = ob;
  }
  __syncthreads();
  if (s_oblique) {
    if (tid >= H || tid >= 256) return;   // (only the first 256 threads: one per column)
    for (int b = 0; b < B; ++b)
      for (int pos = 0; pos < BE_TI * BE_TJ; ++pos) {
        const int i = i_base + pos % BE_TI, j = j_base + pos / BE_TI;
        if (i >= Ds || j >= W) continue;
        unsigned short* rec = out + (int64_t)b * out_bs + (((int64_t)i * W + j) * H + tid) * 16;
        rec[0] = (unsigned short)bp_bf16(moving[(int64_t)b * D * W * H + ((int64_t)(d0 + i) * W + j) * H + tid]);
        for (int c = NV + 1; c < 16; ++c) rec[c] = 0;
        const float x = (float)(d0 + i) - 0.5f * (float)D, y = (float)(W - 1 - j), zz = (float)tid - 0.5f * (float)H;
        for (int v = 0; v < NV; ++v) {
          const float ex = poses.e[v][0], ey = poses.e[v][1], ez = poses.e[v][2];
          const float scale = ey / (ey - y);
          const Tap ty = make_tap(shadow_pix(x, ex, scale, (float)Pw, Pw), Pw);
          const Tap tx = make_tap(shadow_pix(zz, ez, scale, (float)Ph, Ph), Ph);
          const float nw = ty.w0 * tx.w0, ne = ty.w0 * tx.w1, sw = ty.w1 * tx.w0, se = ty.w1 * tx.w1;
          const float* r0 = proj + ((int64_t)b * NV + v) * Pw * Ph + (int64_t)ty.i0 * Ph;
          const float* r1 = proj + ((int64_t)b * NV + v) * Pw * Ph + (int64_t)ty.i1 * Ph;
          float acc = r0[tx.i0] * nw;
          acc = acc + r0[tx.i1] * ne;
          acc = acc + r1[tx.i0] * sw;
          acc = acc + r1[tx.i1] * se;
          rec[v + 1] = (unsigned short)bp_bf16(acc);
        }
      }
    return;
  }
  const int k = tid & 255, jh = tid >> 8;   // column of this thread (H <= 256: checked by the launcher), half of the planes
  constexpr int TJH = BE_TJ / (BE_THREADS / 256);
  const float z = (float)k - 0.5f * (float)H;
  // column taps of this thread per (plane jj of its half, view): the same for all four rows and every batch element
  TapU txs[TJH][NV];
#pragma unroll
  for (int jj = 0; jj < TJH; ++jj) {
    const float y = (float)(W - 1 - (j_base + jh * TJH + jj));
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const float ey = poses.e[v][1], ez = poses.e[v][2];
      const float scale = ey / (ey - y);
      txs[jj][v] = make_tap_u(shadow_pix(z, ez, scale, (float)Ph, Ph), Ph);
      if (txs[jj][v].i0 == BT_SENTINEL) txs[jj][v].i0 = -1;  // weights are 0; any padded column
    }
  }
  __syncthreads();
  const int64_t view_sz = (int64_t)Pw * Ph, V = (int64_t)D * W * H;
  const int RS4 = RS >> 2;
  // The detector rows of every view for batch element b+1 travel into registers while element b is computed (one block of
  // 8 waves per CU holds the LDS: without the look-ahead every element paid the views' L2 latency in the open).
  // (Ph % 4 == 0 and 16-byte aligned views: checked by the launcher.)
  const int total4 = NV * BE_RCAP * RS4;
  const int npf = (total4 + BE_THREADS - 1) / BE_THREADS;   // slots in use (<= BE_NPF: checked by the launcher)
  float4 pre[BE_NPF];
  // which view element every prefetch slot of this thread holds does not depend on the batch element: byte offsets (or
  // "outside": the load then returns the tile's zero padding) are computed once; the look-ahead itself is BE_NPF
  // unconditional bounds-checked buffer loads and as many unconditional LDS stores (the tile is sized in whole slots)
  unsigned poff[BE_NPF];
#pragma unroll
  for (int t = 0; t < BE_NPF; ++t) {
    const int idx = tid + t * BE_THREADS;
    unsigned off = 0x80000000u;
    if (BE_PREFETCH && idx < total4) {
      const int v = idx / (BE_RCAP * RS4), rem = idx - v * (BE_RCAP * RS4);
      const int row = rem / RS4, c4 = rem - row * RS4;
      const int lo = s_lo[v], nr = s_hi[v] >= lo ? s_hi[v] - lo + 1 : 0;
      const int grow = lo + row, col = c4 * 4 - BT_PAD;
      if (row < nr && grow >= 0 && grow < Pw && col >= 0 && col < Ph) off = (unsigned)((((int64_t)v * Pw + grow) * Ph + col) * 4);
    }
    poff[t] = off;
  }
  // The thread's TJH x BE_TI moving-image voxels of an element travel with the element's detector rows — requested one
  // element ahead, ready at the commit.  Loaded inside the sample loop (one dword per voxel, consumed at once) every sample
  // waited for vmcnt(0): a full memory latency per voxel, and with it for the look-ahead and the stores in flight.
  float mv[TJH][BE_TI], mvn[TJH][BE_TI];
  const int kc = min(k, H - 1);
  auto fetch_moving = [&](int b, float (&dst)[TJH][BE_TI]) __attribute__((always_inline)) {
#pragma unroll
    for (int jj = 0; jj < TJH; ++jj) {
      const int jc = min(j_base + jh * TJH + jj, W - 1);
#pragma unroll
      for (int ii = 0; ii < BE_TI; ++ii) {
        const int ic = min(i_base + ii, Ds - 1);   // clamped, not branched: samples outside the slab are never stored
        dst[jj][ii] = moving[(int64_t)b * V + ((int64_t)(d0 + ic) * W + jc) * H + kc];
      }
    }
  };
  auto fetch = [&](int b) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(proj + (int64_t)b * NV * view_sz), (short)0,
                                                                        (int)(NV * view_sz * 4), 0x00020000);
#pragma unroll
    for (int t = 0; t < BE_NPF; ++t) pre[t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, poff[t], 0, 0));
    fetch_moving(b, mvn);
  };
  auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < BE_NPF; ++t)
      if (t < npf) reinterpret_cast<float4*>(tile)[tid + t * BE_THREADS] = pre[t];   // block-uniform bound; tile is [v][row][RS] in float4s, sized in whole slots
  };
  auto stage_direct = [&](int b) __attribute__((always_inline)) {   // without the register look-ahead: global -> LDS in a loop
    const float* pbn = proj + (int64_t)b * NV * view_sz;
    for (int idx = tid; idx < total4; idx += BE_THREADS) {
      const int v = idx / (BE_RCAP * RS4), rem = idx - v * (BE_RCAP * RS4);
      const int row = rem / RS4, c4 = rem - row * RS4;
      const int lo = s_lo[v], nr = s_hi[v] >= lo ? s_hi[v] - lo + 1 : 0;
      const int grow = lo + row, col = c4 * 4 - BT_PAD;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < nr && grow >= 0 && grow < Pw && col >= 0 && col < Ph)
        val = *reinterpret_cast<const float4*>(pbn + (int64_t)v * view_sz + (int64_t)grow * Ph + col);
      reinterpret_cast<float4*>(tile)[idx] = val;
    }
  };
  if constexpr (BE_PREFETCH) { fetch(0); commit(); } else { stage_direct(0); fetch_moving(0, mvn); }
  __syncthreads();
  for (int b = 0; b < B; ++b) {
#pragma unroll
    for (int jj = 0; jj < TJH; ++jj)
#pragma unroll
      for (int ii = 0; ii < BE_TI; ++ii) mv[jj][ii] = mvn[jj][ii];
    if (BE_PREFETCH && b + 1 < B) fetch(b + 1);
    else if (!BE_PREFETCH && b + 1 < B) fetch_moving(b + 1, mvn);
    if (k < H) {
#pragma unroll
      for (int jj = 0; jj < TJH; ++jj) {
        const int jl = jh * TJH + jj, j = j_base + jl;
        if (j >= W) break;
#pragma unroll 1
        for (int ii = 0; ii < BE_TI; ++ii) {   // (not unrolled: all sample bodies at once cost 219 registers and 146 scalar spills)
          const int i = i_base + ii;
          if (i >= Ds) break;
          unsigned rec[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) rec[q] = 0u;
          float m = mv[jj][0];   // (the ii loop is not unrolled: select, do not index)
#pragma unroll
          for (int q = 1; q < BE_TI; ++q) m = ii == q ? mv[jj][q] : m;
          rec[0] = bp_bf16(m);
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            // a view the bundle's shadows miss entirely has zero weights and a zero-filled tile: acc = 0, no branch
            const TapU tx = txs[jj][v];
            const float4 ty = tyt[v * (BE_TI * BE_TJ) + jl * BE_TI + ii];
            const float nw = ty.y * tx.e, ne = ty.y * tx.w, sw = ty.z * tx.e, se = ty.z * tx.w;
            const float* r0 = tile + v * BE_RCAP * RS + __float_as_int(ty.x) + tx.i0 + BT_PAD;
            const float a = r0[0], bq = r0[1], c = r0[RS], dd = r0[RS + 1];
            float acc = a * nw;
            acc = acc + bq * ne;
            acc = acc + c * sw;
            acc = acc + dd * se;
            rec[(v + 1) >> 1] |= bp_bf16(acc) << (16 * ((v + 1) & 1));
          }
          bp_u32x4* dst = reinterpret_cast<bp_u32x4*>(out + (int64_t)b * out_bs + (((int64_t)i * W + j) * H + k) * 16);
          __builtin_nontemporal_store((bp_u32x4){rec[0], rec[1], rec[2], rec[3]}, dst);
          __builtin_nontemporal_store((bp_u32x4){rec[4], rec[5], rec[6], rec[7]}, dst + 1);
        }
      }
    }
    __syncthreads();  // everyone is done with element b's tile
    if (b + 1 < B) {
      if constexpr (BE_PREFETCH) commit(); else stage_direct(b + 1);
      __syncthreads();
    }
  }
}

int fill_poses(LrPoses& lp, const float* poses, int P) {
  if (!poses) return LR_ENULL;
  if (P < 1 || P > LR_MAX_VIEWS) return LR_EINVAL;
  for (int p = 0; p < P; ++p)
    for (int c = 0; c < 3; ++c) lp.e[p][c] = poses[p * 3 + c];
  return LR_OK;
}

}  // namespace

static int backproject_impl(const float* proj, const float* poses, float* out,
                            int B, int P, int Pw, int Ph, int D, int W, int H,
                            int d0, int d1, int64_t out_batch_stride, void* stream) {
  if (!proj || !out) return LR_ENULL;
  if (B < 1 || Pw < 1 || Ph < 1 || D < 1 || W < 1 || H < 1) return LR_EINVAL;
  if (d0 < 0 || d1 > D || d1 <= d0) return LR_EINVAL;
  LrPoses lp;
  if (int e = fill_poses(lp, poses, P)) return e;
  const int Ds = d1 - d0;
  if (out_batch_stride < (int64_t)P * Ds * W * H) return LR_EINVAL;
  // Tiled kernel: H <= 1024 (<= 2 columns per thread of a 512-thread block) and the staged rows fit in LDS.
  const size_t tile_lds = ((size_t)BT_RCAP * (Ph + 2 * BT_PAD) + 4 * BT_TI * BT_TJ) * sizeof(float);
  if (H <= 1024 && tile_lds <= 64 * 1024) {
    const int64_t nb = (int64_t)P * ((Ds + BT_TI - 1) / BT_TI) * ((W + BT_TJ - 1) / BT_TJ);
    if (nb > 0x7fffffffLL) return LR_EINVAL;
    const bool v4 = (Ph % 4 == 0) && ((reinterpret_cast<uintptr_t>(proj) & 15u) == 0);
    // threads along a row: whole waves, at most 512; KC columns per thread (H = 384: 384 threads x 1, not 256 x 2 with a half-idle pass)
    const int KC = (H + 511) / 512;
    // planes side by side for short rows: the largest JP in {1, 2, 4} with JP * H <= 384 threads (H = 160 -> 2 x 160 = 320)
    int JP = H >= 256 ? 1 : (4 * H <= 384 ? 4 : (2 * H <= 384 ? 2 : 1));
    if (lr_sw_set(LR_SW_BP_JP)) { const int v = lr_sw_int(LR_SW_BP_JP, JP); if ((v == 1 || v == 2 || v == 4) && (v == 1 || v * H <= 384)) JP = v; }
    if (JP * H < 64) JP = 1;   // (the tap tables are built by the first 32 threads; tiny rows keep the one-wave form)
    // batch elements per block: short blocks fill the tail of the grid — at least ~6 blocks per resident slot (8 per CU), and
    // not fewer than 4 elements per block (the prologue: tap tables, three barriers)
    int bchunk = B;
    {
      const int64_t want = (int64_t)6 * 8 * 256;
      int nch = (int)((want + nb - 1) / nb);
      if (nch > (B + 3) / 4) nch = (B + 3) / 4;
      if (nch < 1) nch = 1;
      bchunk = (B + nch - 1) / nch;
    }
    if (lr_sw_set(LR_SW_BP_CHUNK)) { const int v = lr_sw_int(LR_SW_BP_CHUNK, 0); bchunk = v >= 1 && v < B ? v : B; }
    const int nchunks = (B + bchunk - 1) / bchunk;
    if (nchunks > 65535) return LR_EINVAL;
    const unsigned nthr = JP > 1 ? (unsigned)(JP * H) : (unsigned)((((H + KC - 1) / KC) + 63) / 64 * 64);
    const dim3 grid((unsigned)nb, (unsigned)nchunks), block(nthr);
    hipStream_t st = lr_stream(stream);
    {   // the views into the caches first (touch_views_kernel); LIFTREG_BP_TOUCH=0: off (A/B aid)
      const int64_t nview = (int64_t)B * P * Pw * Ph;
      // (only where it pays: the pass costs a launch, ~5 us; small launches — a z-slab of 1/8 of C3 writes 135 MB — save less than that)
      const int64_t out_bytes = (int64_t)B * P * Ds * W * H * 4;
      if (v4 && nview * 4 <= 128LL * 1024 * 1024 && nview >= 64 * 1024 && out_bytes >= 256LL * 1024 * 1024 && lr_sw_int(LR_SW_BP_TOUCH, 1) != 0) {
        const int64_t nb4 = (nview / 4 + 255) / 256;
        hipLaunchKernelGGL(touch_views_kernel, dim3((unsigned)(nb4 < 1024 ? nb4 : 1024)), dim3(256), 0, st, proj, nview);
      }
    }
#define LR_BT(KCV, JPV)                                                                              \
  do {                                                                                               \
    if (v4) hipLaunchKernelGGL((backproject_tiled_kernel<KCV, true, JPV>), grid, block, tile_lds, st, proj, lp, out, B, P, Pw, Ph, D, W, H, d0, Ds, out_batch_stride, bchunk); \
    else hipLaunchKernelGGL((backproject_tiled_kernel<KCV, false, JPV>), grid, block, tile_lds, st, proj, lp, out, B, P, Pw, Ph, D, W, H, d0, Ds, out_batch_stride, bchunk);    \
  } while (0)
    if (JP == 4) LR_BT(1, 4);
    else if (JP == 2) LR_BT(1, 2);
    else if (KC == 1) LR_BT(1, 1);
    else LR_BT(2, 1);
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

extern "C" int lr_backproject_f32(const float* proj, const float* poses, float* out,
                                  int B, int P, int Pw, int Ph, int D, int W, int H,
                                  int d0, int d1, int64_t out_batch_stride, void* stream) {
  return backproject_impl(proj, poses, out, B, P, Pw, Ph, D, W, H, d0, d1, out_batch_stride, stream);
}

// (B,P,Pw,Ph) views + (B,1,D,W,H) moving image -> rows [d0,d1) of the bf16 channels-last encoder input (B,Ds,W,H,16):
// channel 0 = moving, 1..P = the backprojected views, P+1..15 = 0.  1 <= P <= 15; out 16-byte aligned; out_batch_stride
// in bf16 elements (>= 16*Ds*W*H).  Values = lr_backproject_f32's (and moving), rounded to nearest-even bf16.
extern "C" int lr_backproject_encin_bf16(const float* proj, const float* moving, const float* poses, void* out, int B, int P,
                                         int Pw, int Ph, int D, int W, int H, int d0, int d1, int64_t out_batch_stride,
                                         void* stream) {
  if (!proj || !moving || !out) return LR_ENULL;
  if (B < 1 || Pw < 1 || Ph < 1 || D < 1 || W < 1 || H < 1) return LR_EINVAL;
  if (d0 < 0 || d1 > D || d1 <= d0) return LR_EINVAL;
  if (P < 1 || P > 15 || H > 256 || (Ph & 3) || (reinterpret_cast<uintptr_t>(proj) & 15u)) return LR_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(out) & 15u) return LR_EALIGN;
  LrPoses lp;
  if (int e = fill_poses(lp, poses, P)) return e;
  const int Ds = d1 - d0;
  if (out_batch_stride < (int64_t)16 * Ds * W * H || (out_batch_stride & 7)) return LR_EINVAL;
  const int64_t nblk = (int64_t)((Ds + BE_TI - 1) / BE_TI) * ((W + BE_TJ - 1) / BE_TJ);
  if (nblk > 0x7fffffffLL) return LR_EINVAL;
  const size_t tile4 = (size_t)P * BE_RCAP * ((Ph + 2 * BT_PAD) / 4);
  const size_t lds = ((BE_PREFETCH ? (tile4 + BE_THREADS - 1) / BE_THREADS * BE_THREADS : tile4) * 4 + 4 * (size_t)P * BE_TI * BE_TJ) * sizeof(float);
  if ((int64_t)P * Pw * Ph * 4 >= 0x7fffffffLL) return LR_EUNSUPPORTED;
  if (lds > 150 * 1024 || (BE_PREFETCH && (int64_t)P * BE_RCAP * ((Ph + 2 * BT_PAD) / 4) > (int64_t)BE_NPF * BE_THREADS)) return LR_EUNSUPPORTED;
  unsigned short* o = reinterpret_cast<unsigned short*>(out);
  hipStream_t st = lr_stream(stream);
#define LR_BE(NVV)                                                                                                         \
  case NVV: {                                                                                                              \
    static std::atomic<uint64_t> attr_done{0};                                                                             \
    if (lr_raise_dyn_lds(reinterpret_cast<const void*>(&backproject_encin_bf16_kernel<NVV>), lds, attr_done) != LR_OK) return LR_ELAUNCH; \
    hipLaunchKernelGGL(backproject_encin_bf16_kernel<NVV>, dim3((unsigned)nblk), dim3(BE_THREADS), lds, st, proj, moving, lp, o, B, Pw, Ph, D, \
                       W, H, d0, Ds, out_batch_stride);                                                                    \
  } break
  switch (P) {
    LR_BE(1); LR_BE(2); LR_BE(3); LR_BE(4); LR_BE(5); LR_BE(6); LR_BE(7); LR_BE(8); LR_BE(9); LR_BE(10); LR_BE(11); LR_BE(12);
    LR_BE(13); LR_BE(14); LR_BE(15);
  }
#undef LR_BE
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
