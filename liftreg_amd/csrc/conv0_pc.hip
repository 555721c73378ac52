// conv0_pc.hip — the encoder's first block (Conv3d k3 p1 s1, Cin = P+1 <= 3 planar fp32 -> 16 channels-last) as a
// PRODUCER / CONSUMER kernel with a double-buffered LDS brick, and — f1 of SURVEY §8 — with the backprojection
// computed by the producers, so the (B,P,D,W,H) feature volume of …Backproj.py:89-93 is never written or read.
//
//   block = 6 wavefronts, 2 blocks per CU, persistent over 4x4x64 output bricks:
//     waves 0..3  consumers: one output plane each, the Winograd F(2,3)-along-H MFMA sweep of conv3d_planar_kernel
//                 (v_mfma_f32_16x16x4_f32, operands by immediate-offset LDS reads, transformed weights in registers,
//                 pairs of tiles stored as they finish) — and NOTHING else: no staging, one barrier per brick;
//     waves 4,5   producers: build brick u+1 in the OTHER LDS buffer while brick u is swept: channel 0 (the moving
//                 image) by 16-byte bounds-checked buffer loads (out of volume -> 0 = the conv's padding), channels
//                 1..P either loaded the same way (plain 3-channel input) or COMPUTED: each window voxel's value is
//                 its backprojection sample — backproject.hip's arithmetic op for op, taps gathered from the 2-D views
//                 (0.5 MB per registration: L2-resident) — written straight into the brick.
//   The matrix pipe never waits for a staging phase of its own wave (the persistent single-buffer kernel spent 30 % of a
//   brick period in stage / prefetch / two barriers), the vector ALU and the texture path — idle next to the MFMAs —
//   do the backprojection, and the step loses a 1.08 GB write, its 2.6 GB of (halo-amplified) re-reads and a launch.
//   Bits: the brick holds exactly what lr_backproject_f32 + the planar staging would put there, and the sweep is the
//   same instruction sequence, so outputs are bit-identical to lr_backproject_f32 + lr_conv3d_first_split_f32.
//
// Replaces (reference file:line)
//   src/liftreg/models/LiftRegDeformSubspaceBackproj.py:85-98   backprojection grid_sample + cat + encoders[0]
//   src/liftreg/utils/sdct_projection_utils.py:227-250          backproj_grids_with_poses (derived per voxel)
//   src/liftreg/layers/layers.py:365-369                        convBlock.forward (Conv3d + LeakyReLU(0.2))
#include "lr_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int PD = 4, PW = 4, PH = 64;          // output brick: 4 planes (one per consumer wave) x 4 rows x 64 voxels
constexpr int RW = PW + 2, RD = PD + 2;         // window rows / planes
constexpr int XOFF = 3;                          // the window starts 3 floats left of the halo: 16-byte aligned rows
constexpr int RSL = 72, F4 = RSL / 4;            // window row: 72 floats = 18 float4
constexpr int PS = RW * RSL, CS = RD * PS;       // plane / channel stride (floats)
constexpr int CC = 3;                            // channels of the brick (Cin <= 3; unused ones are zero)
constexpr int BRICK = CC * CS;                   // 7776 floats = 31104 bytes per buffer
constexpr int NPROD = RW * F4;                   // 108 producer lanes: one per (window row ry, float4 lf4)
constexpr unsigned OOR = 0x80000000u;
#ifndef LR_STORE_AUX
#define LR_STORE_AUX 2  /* nt: 8.6 GB per batch that nothing re-reads before it leaves the L2 */
#endif

struct C0Dims {
  int B, Cin, D, W, H;
  int nHq, nWq, nDq, nbricks;
  int64_t bs0, bsr;  // batch strides (floats) of in0 and of in_rest: V and (Cin-1)*V for split inputs, Cin*V for one tensor
};

struct BpArgs {
  const float* proj;  // (B,P,Pw,Ph)
  int P, Pw, Ph;
  LrPoses poses;
};

__device__ __forceinline__ float lrelu(float v, float slope) { return v >= 0.0f ? v : v * slope; }

// shadow of a voxel coordinate on one detector axis: backproject.hip's shadow_pix (same op order)
__device__ __forceinline__ float bp_pix(float x, float e, float scale, float fsize, int size) {
  float g = (x - e) * scale;  // torch.mul(grids - poses, scale)
  g = g + e;                  // + poses[:, :, ::2]
  g = g / fsize;              // / proj_w
  g = g * 2.0f;               // * 2.0
  return lr_unnormalize(g, size);
}
struct BpTap {
  int i0;      // floor(pix); -4 when the footprint misses the view (both weights 0 then)
  float e, w;  // weights of i0 and i0+1, unmasked: a tap outside the view reads 0
};
__device__ __forceinline__ BpTap bp_tap(float pix, int size) {
  BpTap t;
  const bool in = pix > -1.0f && pix < (float)size;  // also false for NaN
  const float fl = floorf(pix);
  const float w = pix - fl;
  t.w = in ? w : 0.0f;
  t.e = in ? 1.0f - w : 0.0f;
  t.i0 = in ? (int)fl : -4;
  return t;
}

template <int OUTL /* LR_LAYOUT_NDHWC | LR_LAYOUT_NDHWC_HPS */, bool FBP>
__global__ __launch_bounds__(384, 3) void conv0_pc_kernel(const float* __restrict__ in0, const float* __restrict__ in_rest,
                                                          const float* __restrict__ wp, const float* __restrict__ bias,
                                                          float* __restrict__ out, C0Dims d, float slope, BpArgs bp, int dbg /* timing ablations: 1 no sweep, 2 no production */) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // two bricks [CC][RD][RW][RSL]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t V = (int64_t)d.D * d.W * d.H;
  const int brick0 = (int)lr_xcd_remap(blockIdx.x, gridDim.x);
  if (brick0 >= d.nbricks) return;
  const int my_bricks = (d.nbricks - brick0 + (int)gridDim.x - 1) / (int)gridDim.x;
  auto coords = [&](int u, int& b, int& dq, int& wq, int& hq) {
    int br = brick0 + u * (int)gridDim.x;
    hq = br % d.nHq; br /= d.nHq;
    wq = br % d.nWq; br /= d.nWq;
    dq = br % d.nDq;
    b = br / d.nDq;
  };

  if (wave >= 4) {
    // =========================================================== producers ===================================
    // Issue arbitration on a SIMD is priority, then AGE: next to two older consumer waves whose MFMAs keep the matrix
    // pipe full, a producer at equal priority only gets the slots the consumers leave, and production then does not
    // overlap the sweep at all (measured: brick time = sweep + production, 4.6 ms).  The producers' instructions are few
    // and feed the next sweep, so they go first (cdna guide, "Two waves per SIMD", items 2 and 4).
    __builtin_amdgcn_s_setprio(3);
    const int pt = tid - 256;
    const bool lact = pt < NPROD;
    const int ry = lact ? pt / F4 : 0, lf4 = lact ? pt - ry * F4 : 0;
    const int ldst = ry * RSL + lf4 * 4;  // this lane's float4 inside a window plane
    auto produce = [&](int u) {
      float* buf = lds + (u & 1) * BRICK;
      int b, dq, wq, hq;
      coords(u, b, dq, wq, hq);
      const int z0 = dq * PD - 1, y0 = wq * PW - 1, x0 = hq * PH - 1 - XOFF;
      const int yi = y0 + ry, xi = x0 + lf4 * 4;  // xi is a multiple of 4 and H % 4 == 0: a float4 is all in or all out
      const bool xyok = lact & (xi >= 0) & (xi + 3 < d.H) & (yi >= 0) & (yi < d.W);
      const unsigned rowoff = (unsigned)((yi * d.H + xi) * 4);  // byte offset inside a plane (used only when xyok)
      const unsigned plane_bytes = (unsigned)(d.W * d.H * 4);
      // ---- channel 0: the moving image, six 16-byte loads per lane (one per window plane), consumed last
      const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in0 + (int64_t)b * d.bs0), (short)0,
                                                                          (int)(V * 4), 0x00020000);
      float4 mv[RD];
#pragma unroll
      for (int rz = 0; rz < RD; ++rz) {
        const int zi = z0 + rz;
        const bool ok = xyok & (zi >= 0) & (zi < d.D);
        mv[rz] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r0, ok ? (unsigned)zi * plane_bytes + rowoff : OOR, 0, 0));
      }
      if constexpr (!FBP) {
        // ---- channels 1..: loaded like channel 0 (absent channels: a zero-length resource -> zeros)
#pragma unroll
        for (int c = 1; c < CC; ++c) {
          const bool cok = c < d.Cin;
          const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
              const_cast<float*>(in_rest + (int64_t)b * d.bsr + (int64_t)(cok ? c - 1 : 0) * V), (short)0, cok ? (int)(V * 4) : 0, 0x00020000);
          float4 cv[RD];
#pragma unroll
          for (int rz = 0; rz < RD; ++rz) {
            const int zi = z0 + rz;
            const bool ok = xyok & (zi >= 0) & (zi < d.D);
            cv[rz] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rc, ok ? (unsigned)zi * plane_bytes + rowoff : OOR, 0, 0));
          }
          if (lact) {
#pragma unroll
            for (int rz = 0; rz < RD; ++rz) *reinterpret_cast<float4*>(buf + c * CS + rz * PS + ldst) = cv[rz];
          }
        }
      } else {
        // ---- channels 1..P: backprojection samples.  A lane's plane scale (its window row ry) and its four column
        //      taps are constants of a (brick, view); per window plane rz only the row tap changes.
        const float yv = (float)(d.W - 1 - yi);
        const int Pw = bp.Pw, Ph = bp.Ph;
#pragma unroll
        for (int vw = 0; vw < CC - 1; ++vw) {
          const bool view_ok = vw < bp.P;
          const int pv = view_ok ? vw : 0;
          const float ex = bp.poses.e[pv][0], ey = bp.poses.e[pv][1], ez = bp.poses.e[pv][2];
          const float scale = ey / (ey - yv);  // poses_y / (poses_y - grid_y), sdct_projection_utils.py:239
          const __amdgpu_buffer_rsrc_t rview = __builtin_amdgcn_make_buffer_rsrc(
              const_cast<float*>(bp.proj + ((int64_t)b * bp.P + pv) * Pw * Ph), (short)0, view_ok ? Pw * Ph * 4 : 0, 0x00020000);
          int cxb[4];
          float cw0[4], cw1[4];  // weights of the loaded pair's two elements
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const BpTap tc = bp_tap(bp_pix((float)(xi + v) - 0.5f * (float)d.H, ez, scale, (float)Ph, Ph), Ph);
            // the column pair is loaded at xb = clamp(i0, 0, Ph-2); i0 = -1 / Ph-1 shift it by one, and the tap that
            // fell off the view (an exact 0 in the reference's zero padding) simply has no element: weight 0
            const int xb = min(max(tc.i0, 0), Ph - 2);
            const int sh = tc.i0 - xb;
            cxb[v] = xb;
            cw0[v] = sh == 0 ? tc.e : (sh < 0 ? tc.w : 0.0f);  // multiplies pair.x
            cw1[v] = sh == 0 ? tc.w : (sh > 0 ? tc.e : 0.0f);  // multiplies pair.y
          }
          uint2 raw[RD][4][2];  // ALL six window planes of the view in flight: [plane][voxel][detector row i0 / i0+1] —
          float rwe[RD], rww[RD];  // one L2 round trip per view instead of one per plane (the producers are latency-bound)
          auto gather = [&](int rz, int set) {
            const int zi = z0 + rz;
            const bool inside = xyok & (zi >= 0) & (zi < d.D) & view_ok;
            const BpTap tr = bp_tap(bp_pix((float)zi - 0.5f * (float)d.D, ex, scale, (float)Pw, Pw), Pw);
            rwe[set] = inside ? tr.e : 0.0f;
            rww[set] = inside ? tr.w : 0.0f;
            const bool r0ok = inside & (tr.i0 >= 0), r1ok = inside & (tr.i0 + 1 >= 0) & (tr.i0 + 1 < Pw);
            const int rb0 = __mul24(tr.i0, Ph), rb1 = rb0 + Ph;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              raw[set][v][0] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rview, r0ok ? (unsigned)((rb0 + cxb[v]) << 2) : OOR, 0, 0));
              raw[set][v][1] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rview, r1ok ? (unsigned)((rb1 + cxb[v]) << 2) : OOR, 0, 0));
            }
          };
          auto finish = [&](int rz, int set) {
            // acc = a*nw + bq*ne + c*sw + d*se (backproject.hip): a,bq = row i0 at columns i0c, i0c+1; c,d = row i0+1
            float r[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const float p0x = __builtin_bit_cast(float, raw[set][v][0].x), p0y = __builtin_bit_cast(float, raw[set][v][0].y);
              const float p1x = __builtin_bit_cast(float, raw[set][v][1].x), p1y = __builtin_bit_cast(float, raw[set][v][1].y);
              float acc = p0x * (rwe[set] * cw0[v]);
              acc = acc + p0y * (rwe[set] * cw1[v]);
              acc = acc + p1x * (rww[set] * cw0[v]);
              acc = acc + p1y * (rww[set] * cw1[v]);
              r[v] = acc;
            }
            if (lact) *reinterpret_cast<float4*>(buf + (1 + vw) * CS + rz * PS + ldst) = make_float4(r[0], r[1], r[2], r[3]);
          };
#pragma unroll
          for (int rz = 0; rz < RD; ++rz) gather(rz, rz);
#pragma unroll
          for (int rz = 0; rz < RD; ++rz) finish(rz, rz);
        }
      }
      if (lact) {
#pragma unroll
        for (int rz = 0; rz < RD; ++rz) *reinterpret_cast<float4*>(buf + rz * PS + ldst) = mv[rz];
      }
    };
    produce(0);
    __syncthreads();
    for (int u = 0; u < my_bricks; ++u) {
      if (u + 1 < my_bricks && !(dbg & 2)) produce(u + 1);
      __syncthreads();
    }
    return;
  }

  // ============================================================= consumers ===================================
  // The Winograd F(2,3)-along-H sweep of conv3d_planar_kernel<…, WINO> (conv3d.hip), instruction for instruction: columns
  // of an MFMA = 16 output PAIRS of a row; a lane reads the four inputs d0..d3 under a pair's taps once, forms the four
  // differences and feeds four MFMAs (positions r = 0..3) with the transformed weights U_r that lr_conv3d_pack_weights_f32
  // leaves behind the direct ones; 28 MFMAs per 32 outputs instead of 42.
  const int col = lane & 15, kq = lane >> 4;
  int woff[7];   // LDS offset of d0 for this lane's k = (c, tz, ty) of quad q (k = 27 is padding: U = 0, address of k = 26)
  float uw[4][7];
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    const int k = min(q * 4 + kq, 26);
    const int c = k / 9, tz = (k / 3) % 3, ty = k % 3;
    woff[q] = c * CS + (wave + tz) * PS + ty * RSL + XOFF + 2 * col;
#pragma unroll
    for (int r = 0; r < 4; ++r) uw[r][q] = wp[d.Cin * 7 * 64 + (r * 7 + q) * 64 + lane];
  }
  f32x4 bvec = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bvec[r] = bias[(lane >> 4) * 4 + r];
  }
#pragma unroll
  for (int q = 0; q < 7; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(uw[r][q]));  // the loads' wait stays here, not inside the sweep
  __syncthreads();  // brick 0 is in LDS
  const int64_t plane_elems = (int64_t)d.W * d.H * 16;
  for (int u = 0; u < my_bricks; ++u) {
    const float* brick = lds + (u & 1) * BRICK;
    int b, dq, wq, hq;
    coords(u, b, dq, wq, hq);
    const int dz = dq * PD + wave;
    const bool zok = dz < d.D;
    // output plane (b, dz) as a buffer resource; zero-length (every store dropped) when the plane does not exist
    const __amdgpu_buffer_rsrc_t oplane = __builtin_amdgcn_make_buffer_rsrc(
        out + ((int64_t)b * d.D + (zok ? dz : 0)) * plane_elems, (short)0, zok ? (int)(plane_elems * 4) : 0, 0x00020000);
    constexpr int NU = PW * 2, NSW = NU * 7;   // units = (row y, half g of the 64-wide row), 7 k-quads each
    auto rd4 = [&](int sidx, float (&dst)[4]) {
      const int un = sidx / 7, q = sidx % 7;
      const float* dp = brick + woff[q] + (un >> 1) * RSL + (un & 1) * 32;
      dst[0] = dp[0]; dst[1] = dp[1]; dst[2] = dp[2]; dst[3] = dp[3];
    };
    float dv[2][4];
    rd4(0, dv[0]);
    f32x4 dacc[4];
    if (!(dbg & 1))
#pragma clang loop unroll(full)
    for (int sidx = 0; sidx < NSW; ++sidx) {
      const int un = sidx / 7, q = sidx % 7;
      if (q == 0) {
        dacc[0] = bvec;  // y0 = D0 + D1 + D2 carries the bias; y1 gets it in the epilogue
#pragma unroll
        for (int r = 1; r < 4; ++r) dacc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      if (sidx + 1 < NSW) rd4(sidx + 1, dv[(sidx + 1) & 1]);
      const float* dc = dv[sidx & 1];
      const float v0 = dc[0] - dc[2], v1 = dc[1] + dc[2], v2 = dc[2] - dc[1], v3 = dc[1] - dc[3];
      dacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(uw[0][q], v0, dacc[0], 0, 0, 0);
      dacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(uw[1][q], v1, dacc[1], 0, 0, 0);
      dacc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(uw[2][q], v2, dacc[2], 0, 0, 0);
      dacc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(uw[3][q], v3, dacc[3], 0, 0, 0);
      if (sidx + 1 < NSW) {
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
      if (q == 6) {  // unconditional, countable stores: 16 bytes per lane, 1 KiB contiguous per tile in the parity-split row
        f32x4 yo[2];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          yo[0][e] = (dacc[0][e] + dacc[1][e]) + dacc[2][e];
          yo[1][e] = ((dacc[1][e] - dacc[2][e]) - dacc[3][e]) + bvec[e];
        }
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          const int wo = wq * PW + (un >> 1), ho = hq * PH + (un & 1) * 32 + 2 * col + o;
          const int c0 = (lane >> 4) * 4;
          float4 v;
          v.x = lrelu(yo[o][0], slope); v.y = lrelu(yo[o][1], slope); v.z = lrelu(yo[o][2], slope); v.w = lrelu(yo[o][3], slope);
          unsigned off;
          if (OUTL == LR_LAYOUT_NDHWC) {
            off = (unsigned)(((wo * d.H + ho) * 16 + c0) * 4);
          } else {  // row = [parity][H/2][16 floats]
            const int hp = (ho & 1) * (d.H >> 1) + (ho >> 1);
            off = (unsigned)((wo * d.H * 16 + hp * 16 + c0) * 4);
          }
          if (wo >= d.W || ho >= d.H) off = OOR;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), oplane, off, 0, LR_STORE_AUX);
        }
      }
    }
    __syncthreads();  // every consumer is done with this buffer; the producers have filled the other one
  }
}

}  // namespace

// Launcher shared by conv3d.hip's dispatcher (plain 3-channel input: in_rest = channels 1..) and the fused entry.
// Returns LR_EUNSUPPORTED for shapes the kernel does not take; the callers fall back to the single-buffer kernel.
int lr_internal_conv0_pc(const float* in0, int64_t bs0, const float* in_rest, int64_t bsr, const float* packed_w,
                         const float* bias, float* out, int B, int Cin, int D, int W, int H, int out_layout, float slope,
                         const float* proj, const float* poses, int P, int Pw, int Ph, hipStream_t st) {
  if (Cin < 1 || Cin > 3 || (H & 3) || (out_layout != LR_LAYOUT_NDHWC && out_layout != LR_LAYOUT_NDHWC_HPS)) return LR_EUNSUPPORTED;
  if (out_layout == LR_LAYOUT_NDHWC_HPS && (H & 1)) return LR_EUNSUPPORTED;
  const int64_t V = (int64_t)D * W * H;
  if (V * 4 >= 0x7fffffffLL || (int64_t)W * H * 16 * 4 >= 0x7fffffffLL) return LR_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(in0) | reinterpret_cast<uintptr_t>(in_rest) | reinterpret_cast<uintptr_t>(out)) & 15u) return LR_EALIGN;
  C0Dims d;
  d.B = B; d.Cin = Cin; d.D = D; d.W = W; d.H = H; d.bs0 = bs0; d.bsr = bsr;
  d.nHq = (H + PH - 1) / PH; d.nWq = (W + PW - 1) / PW; d.nDq = (D + PD - 1) / PD;
  const int64_t nb = (int64_t)B * d.nDq * d.nWq * d.nHq;
  if (nb > 0x7fffffffLL) return LR_EINVAL;
  d.nbricks = (int)nb;
  int resident = 512;  // 2 persistent blocks per CU
  resident = lr_sw_int(LR_SW_CONV0_BLOCKS, resident);  // tuning aid
  const dim3 grid((unsigned)(nb < resident ? nb : resident)), block(384);
  const size_t ldsb = (size_t)2 * BRICK * sizeof(float);
#ifdef LR_DIAG_ABLATIONS   // diagnostic build only (make -B EXTRA=-DLR_DIAG_ABLATIONS): timing ablations, WRONG results
  const int dbg = getenv("LIFTREG_CONV0_DBG") ? atoi(getenv("LIFTREG_CONV0_DBG")) : 0;
#else
  const int dbg = 0;
#endif
  BpArgs a;
  a.proj = proj; a.P = P; a.Pw = Pw; a.Ph = Ph;
  for (int p = 0; p < LR_MAX_VIEWS; ++p)
    for (int c = 0; c < 3; ++c) a.poses.e[p][c] = (proj && p < P) ? poses[p * 3 + c] : 0.0f;
  if (proj) {
    if (P < 1 || P > 2 || Cin != P + 1 || Pw < 2 || Ph < 2 || (int64_t)Pw * Ph * 4 >= 0x7fffffffLL || Pw >= (1 << 23) / Ph)
      return LR_EUNSUPPORTED;
    if (out_layout == LR_LAYOUT_NDHWC_HPS)
      hipLaunchKernelGGL((conv0_pc_kernel<LR_LAYOUT_NDHWC_HPS, true>), grid, block, ldsb, st, in0, in_rest, packed_w, bias, out, d, slope, a, dbg);
    else
      hipLaunchKernelGGL((conv0_pc_kernel<LR_LAYOUT_NDHWC, true>), grid, block, ldsb, st, in0, in_rest, packed_w, bias, out, d, slope, a, dbg);
  } else {
    if (Cin > 1 && !in_rest) return LR_ENULL;
    if (out_layout == LR_LAYOUT_NDHWC_HPS)
      hipLaunchKernelGGL((conv0_pc_kernel<LR_LAYOUT_NDHWC_HPS, false>), grid, block, ldsb, st, in0, in_rest, packed_w, bias, out, d, slope, a, dbg);
    else
      hipLaunchKernelGGL((conv0_pc_kernel<LR_LAYOUT_NDHWC, false>), grid, block, ldsb, st, in0, in_rest, packed_w, bias, out, d, slope, a, dbg);
  }
  return lr_launch_status();
}
