// warp.hip — K6+K7(+K9): phi = disp + identity ; spatial-transformer trilinear
// warp of the moving CT.  Voxel-driven gather: coalesced float4 loads of the
// three displacement channels, 8 taps per voxel served by L1/L2 (displacements
// are a few voxels, so a wavefront's taps stay inside a handful of H-rows),
// float4 stores of phi and of the warped image.
//
// Replaces (reference file:line)
//   src/liftreg/utils/net_utils.py:9-56    Bilinear (channel reorder (2,1,0), grid_sample 3D,
//                                           (I+1)/2 … *2-1 intensity scaling)
//   src/liftreg/utils/net_utils.py:59-87   identity_map (passed in as three per-axis tables)
//   src/liftreg/models/LiftRegDeformSubspaceBackproj.py:54-58  moving_cp = (moving+1)*seg-1
//   src/liftreg/models/LiftRegDeformSubspaceBackproj.py:68-69  deform_field = disp + id ; warp
#include "lr_common.h"

namespace {

struct Axis3 {
  int i0, i1;
  float w0, w1;
  bool ok0, ok1;
};

template <bool BORDER>
__device__ __forceinline__ Axis3 axis_of(float g, int size) {
  float pix = lr_unnormalize(g, size);
  if constexpr (BORDER) pix = fminf((float)(size - 1), fmaxf(pix, 0.0f));  // clip_coordinates
  Axis3 a;
  if (!(pix > -1.0f && pix < (float)size)) {
    a.i0 = a.i1 = 0;
    a.w0 = a.w1 = 0.0f;
    a.ok0 = a.ok1 = false;
    return a;
  }
  const float fl = floorf(pix);
  const int i0 = (int)fl, i1 = i0 + 1;
  a.w0 = (float)i1 - pix;
  a.w1 = pix - (float)i0;
  a.ok0 = i0 >= 0;
  a.ok1 = i1 < size;
  a.i0 = max(i0, 0);
  a.i1 = min(i1, size - 1);
  return a;
}

template <bool BORDER>
__device__ __forceinline__ int nearest_of(float g, int size, bool& ok) {
  float pix = lr_unnormalize(g, size);
  if constexpr (BORDER) pix = fminf((float)(size - 1), fmaxf(pix, 0.0f));
  const float r = rintf(pix);  // std::nearbyint, ties to even
  ok = (r >= 0.0f) && (r < (float)size);
  return ok ? (int)r : 0;
}

template <bool SCALE, bool SEG>
__device__ __forceinline__ float tap(const float* img, const float* seg, int64_t off, bool ok) {
  float v = img[off];
  if constexpr (SEG) v = (v + 1.0f) * seg[off] - 1.0f;  // (moving+1)*moving_seg-1
  if constexpr (SCALE) v = (v + 1.0f) * 0.5f;           // (input1 + 1) / 2
  return ok ? v : 0.0f;
}

// The two x-taps of a corner pair are neighbours in memory: one 8-byte gather (dword-aligned
// global_load_dwordx2) fetches both, halving the gather instructions the texture addresser sees.
// xb = clamp(i0, 0, H-2) is the pair's base column; at the volume's x faces the in-range tap is
// picked out of the pair and the other one is dropped by its ok flag, exactly as before.
template <bool SCALE, bool SEG>
__device__ __forceinline__ void tap_pair(const float* img, const float* seg, int64_t rowoff, int xb,
                                         int shift, bool ok0, bool ok1, float& v0, float& v1) {
  typedef float f32x2 __attribute__((ext_vector_type(2), aligned(4)));
  const f32x2 p = *reinterpret_cast<const f32x2*>(img + rowoff + xb);
  float a = shift > 0 ? p.y : p.x;   // tap i0   (shift = i0 - xb in {-1,0,+1})
  float b = shift < 0 ? p.x : p.y;   // tap i0+1
  if constexpr (SEG) {
    const f32x2 q = *reinterpret_cast<const f32x2*>(seg + rowoff + xb);
    a = (a + 1.0f) * (shift > 0 ? q.y : q.x) - 1.0f;  // (moving+1)*moving_seg-1
    b = (b + 1.0f) * (shift < 0 ? q.x : q.y) - 1.0f;
  }
  if constexpr (SCALE) {
    a = (a + 1.0f) * 0.5f;  // (input1 + 1) / 2
    b = (b + 1.0f) * 0.5f;
  }
  v0 = ok0 ? a : 0.0f;
  v1 = ok1 ? b : 0.0f;
}

template <int VEC, bool SCALE, bool BORDER, bool NEAREST, bool SEG>
__global__ __launch_bounds__(256) void warp_kernel(
    const float* __restrict__ img, const float* __restrict__ seg, const float* __restrict__ disp,
    const float* __restrict__ id0, const float* __restrict__ id1, const float* __restrict__ id2,
    float* __restrict__ phi_out, float* __restrict__ warped, int B, int C, int D, int W, int H,
    int Dn) {
  const int HV = H / VEC;
  const int64_t per_b = (int64_t)Dn * W * HV;
  const unsigned nblk_b = (unsigned)((per_b + 255) / 256);  // blocks per batch element
  const unsigned lb = lr_xcd_remap(blockIdx.x, nblk_b);     // gridDim.x == nblk_b
  const int b = blockIdx.y;
  const int64_t idx = (int64_t)lb * 256 + threadIdx.x;
  if (idx >= per_b) return;
  const int kv = (int)(idx % HV);
  const int j = (int)((idx / HV) % W);
  const int i = (int)(idx / HV / W);
  const int64_t slabV = (int64_t)Dn * W * H;
  const int64_t V = (int64_t)D * W * H;
  const int64_t voff = ((int64_t)i * W + j) * H + (int64_t)kv * VEC;

  float d0v[VEC], d1v[VEC], d2v[VEC];
  const float* dp = disp + (int64_t)b * 3 * slabV + voff;
  if constexpr (VEC == 4) {
    const float4 a = *reinterpret_cast<const float4*>(dp);
    const float4 bq = *reinterpret_cast<const float4*>(dp + slabV);
    const float4 c = *reinterpret_cast<const float4*>(dp + 2 * slabV);
    d0v[0] = a.x; d0v[1] = a.y; d0v[2] = a.z; d0v[3] = a.w;
    d1v[0] = bq.x; d1v[1] = bq.y; d1v[2] = bq.z; d1v[3] = bq.w;
    d2v[0] = c.x; d2v[1] = c.y; d2v[2] = c.z; d2v[3] = c.w;
  } else {
    d0v[0] = dp[0]; d1v[0] = dp[slabV]; d2v[0] = dp[2 * slabV];
  }
  if (id0) {  // deform_field = disp_field + id_transform
    const float a0 = id0[i], a1 = id1[j];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      d0v[v] = d0v[v] + a0;
      d1v[v] = d1v[v] + a1;
      d2v[v] = d2v[v] + id2[kv * VEC + v];
    }
  }
  if (phi_out) {
    float* pp = phi_out + (int64_t)b * 3 * slabV + voff;
    if constexpr (VEC == 4) {
      *reinterpret_cast<float4*>(pp) = make_float4(d0v[0], d0v[1], d0v[2], d0v[3]);
      *reinterpret_cast<float4*>(pp + slabV) = make_float4(d1v[0], d1v[1], d1v[2], d1v[3]);
      *reinterpret_cast<float4*>(pp + 2 * slabV) = make_float4(d2v[0], d2v[1], d2v[2], d2v[3]);
    } else {
      pp[0] = d0v[0]; pp[slabV] = d1v[0]; pp[2 * slabV] = d2v[0];
    }
  }

  const int64_t sD = (int64_t)W * H;
  for (int c = 0; c < C; ++c) {
    const float* im = img + ((int64_t)b * C + c) * V;
    const float* sg = SEG ? seg + ((int64_t)b * C + c) * V : nullptr;
    float res[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      // grid (x,y,z) = phi channels (2,1,0): x ↔ H, y ↔ W, z ↔ D
      if constexpr (NEAREST) {
        bool okx, oky, okz;
        const int xh = nearest_of<BORDER>(d2v[v], H, okx);
        const int yw = nearest_of<BORDER>(d1v[v], W, oky);
        const int zd = nearest_of<BORDER>(d0v[v], D, okz);
        res[v] = tap<SCALE, SEG>(im, sg, (int64_t)zd * sD + (int64_t)yw * H + xh, okx && oky && okz);
      } else {
        const Axis3 ax = axis_of<BORDER>(d2v[v], H);
        const Axis3 ay = axis_of<BORDER>(d1v[v], W);
        const Axis3 az = axis_of<BORDER>(d0v[v], D);
        const int64_t o00 = (int64_t)az.i0 * sD + (int64_t)ay.i0 * H;
        const int64_t o01 = (int64_t)az.i0 * sD + (int64_t)ay.i1 * H;
        const int64_t o10 = (int64_t)az.i1 * sD + (int64_t)ay.i0 * H;
        const int64_t o11 = (int64_t)az.i1 * sD + (int64_t)ay.i1 * H;
        float v_tnw, v_tne, v_tsw, v_tse, v_bnw, v_bne, v_bsw, v_bse;
        if (H >= 2) {
          // unclamped floor: ax.i0 was clamped up from -1 exactly when tap 0 is out of range
          const int x0 = ax.ok0 ? ax.i0 : ax.i1 - 1;
          const int xb = min(max(x0, 0), H - 2), shift = x0 - xb;
          tap_pair<SCALE, SEG>(im, sg, o00, xb, shift, az.ok0 && ay.ok0 && ax.ok0, az.ok0 && ay.ok0 && ax.ok1, v_tnw, v_tne);
          tap_pair<SCALE, SEG>(im, sg, o01, xb, shift, az.ok0 && ay.ok1 && ax.ok0, az.ok0 && ay.ok1 && ax.ok1, v_tsw, v_tse);
          tap_pair<SCALE, SEG>(im, sg, o10, xb, shift, az.ok1 && ay.ok0 && ax.ok0, az.ok1 && ay.ok0 && ax.ok1, v_bnw, v_bne);
          tap_pair<SCALE, SEG>(im, sg, o11, xb, shift, az.ok1 && ay.ok1 && ax.ok0, az.ok1 && ay.ok1 && ax.ok1, v_bsw, v_bse);
        } else {
          v_tnw = tap<SCALE, SEG>(im, sg, o00 + ax.i0, az.ok0 && ay.ok0 && ax.ok0);
          v_tne = tap<SCALE, SEG>(im, sg, o00 + ax.i1, az.ok0 && ay.ok0 && ax.ok1);
          v_tsw = tap<SCALE, SEG>(im, sg, o01 + ax.i0, az.ok0 && ay.ok1 && ax.ok0);
          v_tse = tap<SCALE, SEG>(im, sg, o01 + ax.i1, az.ok0 && ay.ok1 && ax.ok1);
          v_bnw = tap<SCALE, SEG>(im, sg, o10 + ax.i0, az.ok1 && ay.ok0 && ax.ok0);
          v_bne = tap<SCALE, SEG>(im, sg, o10 + ax.i1, az.ok1 && ay.ok0 && ax.ok1);
          v_bsw = tap<SCALE, SEG>(im, sg, o11 + ax.i0, az.ok1 && ay.ok1 && ax.ok0);
          v_bse = tap<SCALE, SEG>(im, sg, o11 + ax.i1, az.ok1 && ay.ok1 && ax.ok1);
        }
        float s = v_tnw * ((ax.w0 * ay.w0) * az.w0);
        s = s + v_tne * ((ax.w1 * ay.w0) * az.w0);
        s = s + v_tsw * ((ax.w0 * ay.w1) * az.w0);
        s = s + v_tse * ((ax.w1 * ay.w1) * az.w0);
        s = s + v_bnw * ((ax.w0 * ay.w0) * az.w1);
        s = s + v_bne * ((ax.w1 * ay.w0) * az.w1);
        s = s + v_bsw * ((ax.w0 * ay.w1) * az.w1);
        s = s + v_bse * ((ax.w1 * ay.w1) * az.w1);
        res[v] = s;
      }
      if constexpr (SCALE) res[v] = res[v] * 2.0f - 1.0f;  // output * 2 - 1
    }
    float* wp = warped + ((int64_t)b * C + c) * slabV + voff;
    if constexpr (VEC == 4)
      *reinterpret_cast<float4*>(wp) = make_float4(res[0], res[1], res[2], res[3]);
    else
      wp[0] = res[0];
  }
}

// ---- the model's case (trilinear, zeros padding, float4 rows, no mask) with a third of the vector-ALU work ------
// The general kernel above spends its time in the vector ALU, not in memory (64-bit index arithmetic, integer
// divisions, per-tap range flags and selects: ≈300 instruction slots per voxel).  Same arithmetic, same results:
//  * taps are fetched with bounds-checked raw buffer loads relative to the (b,c) image: 32-bit offsets built with
//    full-rate 24-bit multiplies; a tap plane z = -1 or z = D falls outside the resource by itself and reads 0, an
//    out-of-range y row (or a sample with any axis out of range) is pushed outside with one select on the row term;
//  * an out-of-range tap is removed through its axis weight (one select per axis side) instead of per-tap flags;
//    its value never matters because it is the 0 the buffer unit returns (finite whatever the image holds);
//  * the two x taps are one 8-byte load at x0; only where a wave touches the x faces (x0 = -1 or H-1) a
//    wave-uniform branch re-bases the pair and zeroes the missing tap;
//  * (I+1)/2 … ·2-1: halving is exact, so interpolating I+1 and subtracting 1 gives the same bits
//    (fl(fl(v+1)·w)/2 = fl((fl(v+1)/2)·w), no underflow: weights are ≥ 2^-25 or 0);
//  * grid = (blocks in a plane, plane, batch element): the only per-thread division is by H/4, done in float.
// (the 8-byte buffer loads go through HIP's uint2: element access on the builtin's own ext-vector result gets
//  narrowed to a 4-byte load by this compiler, leaving .y undefined)
struct AxisF {
  float w0, w1;  // weights of taps i0 / i0+1, zero where the tap (or the whole axis) is out of range
  int i0;        // floor(pix), 0 when the axis is out of range
  bool valid;
};

__device__ __forceinline__ AxisF axis_fast(float g, int size) {
  const float pix = lr_unnormalize(g, size);
  AxisF a;
  a.valid = pix > -1.0f && pix < (float)size;
  const float fl = floorf(pix);
  a.i0 = a.valid ? (int)fl : 0;
  const float w0 = (fl + 1.0f) - pix, w1 = pix - fl;
  a.w0 = (a.valid && a.i0 >= 0) ? w0 : 0.0f;
  a.w1 = (a.valid && a.i0 + 1 < size) ? w1 : 0.0f;
  return a;
}

// One trilinear sample of the fast path: phi components (g0,g1,g2) <-> (D,W,H) axes, taps from `rsrc` (one image).
template <bool SCALE>
__device__ __forceinline__ float tri_sample_fast(const __amdgpu_buffer_rsrc_t rsrc, float g0, float g1, float g2, int D,
                                                 int W, int H, int sD) {
  constexpr int OUTSIDE = 0x20000000;  // elements; ·4 bytes = 2^31 ≥ any resource length accepted by the launchers
  // grid (x,y,z) = phi channels (2,1,0): x ↔ H, y ↔ W, z ↔ D
  const AxisF ax = axis_fast(g2, H), ay = axis_fast(g1, W), az = axis_fast(g0, D);
  const bool all = ax.valid && ay.valid && az.valid;
  const int y0 = (all && ay.i0 >= 0) ? __mul24(ay.i0, H) : OUTSIDE;
  const int y1 = (all && ay.i0 + 1 < W) ? __mul24(ay.i0 + 1, H) : OUTSIDE;
  const int z0 = __mul24(az.i0, sD), z1 = z0 + sD;
  const int xb = min(max(ax.i0, 0), H - 2);
  const unsigned xb4 = (unsigned)xb << 2;
  const unsigned o00 = ((unsigned)(z0 + y0) << 2) + xb4, o01 = ((unsigned)(z0 + y1) << 2) + xb4;
  const unsigned o10 = ((unsigned)(z1 + y0) << 2) + xb4, o11 = ((unsigned)(z1 + y1) << 2) + xb4;
  const uint2 q00 = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, o00, 0, 0));
  const uint2 q01 = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, o01, 0, 0));
  const uint2 q10 = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, o10, 0, 0));
  const uint2 q11 = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, o11, 0, 0));
  float tp[8] = {__builtin_bit_cast(float, q00.x), __builtin_bit_cast(float, q00.y),
                 __builtin_bit_cast(float, q01.x), __builtin_bit_cast(float, q01.y),
                 __builtin_bit_cast(float, q10.x), __builtin_bit_cast(float, q10.y),
                 __builtin_bit_cast(float, q11.x), __builtin_bit_cast(float, q11.y)};
  const int shift = ax.i0 - xb;  // -1: x0 = -1 (tap 0 missing, tap 1 is the pair's first); +1: x0 = H-1
  if (__builtin_amdgcn_ballot_w64(shift != 0) != 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float px = tp[2 * r], py = tp[2 * r + 1];
      tp[2 * r] = shift < 0 ? 0.0f : (shift > 0 ? py : px);
      tp[2 * r + 1] = shift > 0 ? 0.0f : (shift < 0 ? px : py);
    }
  }
  if constexpr (SCALE) {
#pragma unroll
    for (int r = 0; r < 8; ++r) tp[r] = tp[r] + 1.0f;  // (input1 + 1), the /2 is taken out exactly
  }
  float s = tp[0] * ((ax.w0 * ay.w0) * az.w0);
  s = s + tp[1] * ((ax.w1 * ay.w0) * az.w0);
  s = s + tp[2] * ((ax.w0 * ay.w1) * az.w0);
  s = s + tp[3] * ((ax.w1 * ay.w1) * az.w0);
  s = s + tp[4] * ((ax.w0 * ay.w0) * az.w1);
  s = s + tp[5] * ((ax.w1 * ay.w0) * az.w1);
  s = s + tp[6] * ((ax.w0 * ay.w1) * az.w1);
  s = s + tp[7] * ((ax.w1 * ay.w1) * az.w1);
  return SCALE ? s - 1.0f : s;  // (output/2) * 2 - 1
}

template <bool SCALE>
__global__ __launch_bounds__(256) void warp_tri_fast_kernel(
    const float* __restrict__ img, const float* __restrict__ disp, const float* __restrict__ id0,
    const float* __restrict__ id1, const float* __restrict__ id2, float* __restrict__ phi_out,
    float* __restrict__ warped, int C, int D, int W, int H, int Dn, float rcp_hv) {
  const int HV = H >> 2;
  const int t = blockIdx.x * 256 + threadIdx.x;  // float4 index inside plane i
  const int i = blockIdx.y, b = blockIdx.z;
  const int j = (int)(((float)t + 0.5f) * rcp_hv);  // exact: t < 2^20, (t+0.5)/HV is ≥ 0.5/HV away from an integer
  if (j >= W) return;
  const int kv = t - __mul24(j, HV);
  const int sD = W * H;
  const int64_t slabV = (int64_t)Dn * sD, V = (int64_t)D * sD;
  const int inplane = __mul24(j, H) + (kv << 2);
  const int64_t ubase = (int64_t)b * 3 * slabV + (int64_t)i * sD;  // wave-uniform

  const float* dp = disp + ubase + inplane;
  const float4 da = *reinterpret_cast<const float4*>(dp);
  const float4 db = *reinterpret_cast<const float4*>(dp + slabV);
  const float4 dc = *reinterpret_cast<const float4*>(dp + 2 * slabV);
  float d0v[4] = {da.x, da.y, da.z, da.w}, d1v[4] = {db.x, db.y, db.z, db.w}, d2v[4] = {dc.x, dc.y, dc.z, dc.w};
  if (id0) {  // deform_field = disp_field + id_transform
    const float a0 = id0[i], a1 = id1[j];
    const float4 a2 = *reinterpret_cast<const float4*>(id2 + (kv << 2));
    const float a2v[4] = {a2.x, a2.y, a2.z, a2.w};
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      d0v[v] = d0v[v] + a0;
      d1v[v] = d1v[v] + a1;
      d2v[v] = d2v[v] + a2v[v];
    }
  }
  if (phi_out) {
    float* pp = phi_out + ubase + inplane;
    *reinterpret_cast<float4*>(pp) = make_float4(d0v[0], d0v[1], d0v[2], d0v[3]);
    *reinterpret_cast<float4*>(pp + slabV) = make_float4(d1v[0], d1v[1], d1v[2], d1v[3]);
    *reinterpret_cast<float4*>(pp + 2 * slabV) = make_float4(d2v[0], d2v[1], d2v[2], d2v[3]);
  }

  for (int c = 0; c < C; ++c) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(img + ((int64_t)b * C + c) * V), (short)0, (int)(V * 4), 0x00020000);
    float res[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) res[v] = tri_sample_fast<SCALE>(rsrc, d0v[v], d1v[v], d2v[v], D, W, H, sD);
    float* wp = warped + ((int64_t)b * C + c) * slabV + (int64_t)i * sD + inplane;
    *reinterpret_cast<float4*>(wp) = make_float4(res[0], res[1], res[2], res[3]);
  }
}

// ---- f1: PCA reconstruction + identity + warp in one pass -----------------------------------------------------
// disp = coefs·basis + mean (the arithmetic of pca.hip's pca_kernel: the same fmaf chain per element), phi = disp + id,
// warped = trilinear(moving, phi) (tri_sample_fast) — the displacement field is written once (`params`, `phi` are
// model outputs) and never read back: the separate warp kernel re-read 4·3V bytes per sample.  A thread owns 4
// consecutive voxels of a row and ALL batch rows (the basis is streamed once per batch, as in pca_kernel): 8 x 3
// float4 accumulators, three basis streams (the rows' D-, W- and H-component thirds).
// Slab form (z-slab sharding): the launch covers output rows [d0,d1) = Dn rows; `basis`/`mean` point at the
// column of (component 0, row d0) and their three component thirds are `bcs` columns apart (V for the full (L,3V)
// basis, Dn·W·H for a rank's compact slab of it); outputs are (B,3,Dn,W,H) / (B,C,Dn,W,H) slabs; the moving image is
// whole (taps cross slabs).  id0 = the D-axis identity table from row d0 on.
// NCC (f1, "NCC moments in the warp epilogue"): with a target slab (B,1,Dn,W,H) the five fp64 raw moments of
// (warped, target) per batch row are reduced here — the similarity never re-reads `warped` (layers/losses.py:18-29
// made ~8 passes, ncc_moments_kernel one): per-wave partials [b][4*block+wave][5], fixed order, no atomics; the caller's
// ncc_reduce pass (ncc.hip) sums the blocks.  C == 1 only.
template <bool BF, bool SCALE, int BT /* batch rows per thread: 8, or 4 for small batches (half the accumulators) */, bool NCC, bool MULTI = false>
__global__ __launch_bounds__(256) void pca_warp_kernel(const float* __restrict__ coefs, const float* __restrict__ basis,
                                                       const float* __restrict__ mean, const float* __restrict__ img,
                                                       const float* __restrict__ id0, const float* __restrict__ id1,
                                                       const float* __restrict__ id2, float* __restrict__ disp_out,
                                                       float* __restrict__ phi_out, float* __restrict__ warped, int B,
                                                       int L, int C, int D, int W, int H, int64_t ldb, float rcp_hv,
                                                       int64_t bcs, const float* __restrict__ target,
                                                       double* __restrict__ ncc_partial, int gx, int Dn, int nch) {
  extern __shared__ float cs[];  // [L][BT]
  // Batches above BT rows: `nch` chunks of BT rows, each (tile, chunk) its own block.  The hardware deals consecutive block ids
  // round-robin over the 8 XCDs: ids g*8*nch + chunk*8 + x (x = 0..7) are the nch chunks of tile 8 g + x, dispatched back to
  // back on ONE XCD — the chunks stream the same basis columns at the same time and all but the first find them in that
  // XCD's L2 (the basis leaves HBM once per batch, not once per chunk).
  int bx, i;   // float4 tile inside the plane | slab row (global row d0 + i)
  if constexpr (MULTI) {
    const int bid = (int)blockIdx.x, per = 8 * nch, g = bid / per, r = bid - g * per;
    const int chunk = r >> 3, tile = g * 8 + (r & 7);
    if (tile >= gx * Dn) return;
    i = tile / gx;
    bx = tile - i * gx;
    const int64_t Vc = (int64_t)D * W * H, Vsc = (int64_t)Dn * W * H;
    coefs += (int64_t)chunk * BT * L;
    img += (int64_t)chunk * BT * C * Vc;
    disp_out += (int64_t)chunk * BT * 3 * Vsc;
    phi_out += (int64_t)chunk * BT * 3 * Vsc;
    warped += (int64_t)chunk * BT * C * Vsc;
    B = B - chunk * BT < BT ? B - chunk * BT : BT;
  } else {
    bx = (int)blockIdx.x;
    i = (int)blockIdx.y;
  }
  for (int t = threadIdx.x; t < L * BT; t += blockDim.x) {
    const int l = t / BT, b = t % BT;
    cs[t] = b < B ? coefs[(int64_t)b * L + l] : 0.0f;
  }
  __syncthreads();
  const int HV = H >> 2;
  const int t = bx * 256 + threadIdx.x;  // float4 index inside plane i
  const int jj = (int)(((float)t + 0.5f) * rcp_hv);
  const bool active = jj < W;
  if (!NCC && !active) return;
  const int j = active ? jj : 0;
  const int kv = active ? t - __mul24(j, HV) : 0;
  const int sD = W * H;
  const int64_t V = (int64_t)D * sD;        // the moving image: whole volume
  const int64_t Vs = (int64_t)Dn * sD;      // output slabs
  const int inplane = __mul24(j, H) + (kv << 2);
  const int64_t m = (int64_t)i * sD + inplane;  // voxel index inside the slab = column of the D-component third

  typedef float f32x4v __attribute__((ext_vector_type(4)));
  f32x4v acc[BT][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const f32x4v mu = *reinterpret_cast<const f32x4v*>(mean + c * bcs + m);
#pragma unroll
    for (int b = 0; b < BT; ++b) acc[b][c] = mu;
  }
#pragma unroll 4
  for (int l = 0; l < L; ++l) {
    f32x4v v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (BF) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2* bp = reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(basis) + (int64_t)l * ldb + c * bcs + m);
        const u32x2 raw = MULTI ? *bp : __builtin_nontemporal_load(bp);   // MULTI: the other chunks of the tile read the same lines from L2
        v[c][0] = __builtin_bit_cast(float, raw.x << 16);
        v[c][1] = __builtin_bit_cast(float, raw.x & 0xffff0000u);
        v[c][2] = __builtin_bit_cast(float, raw.y << 16);
        v[c][3] = __builtin_bit_cast(float, raw.y & 0xffff0000u);
      } else {
        const f32x4v* bp = reinterpret_cast<const f32x4v*>(basis + (int64_t)l * ldb + c * bcs + m);
        v[c] = MULTI ? *bp : __builtin_nontemporal_load(bp);
      }
    }
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      const float cf = cs[l * BT + b];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        acc[b][c][0] = fmaf(cf, v[c][0], acc[b][c][0]);
        acc[b][c][1] = fmaf(cf, v[c][1], acc[b][c][1]);
        acc[b][c][2] = fmaf(cf, v[c][2], acc[b][c][2]);
        acc[b][c][3] = fmaf(cf, v[c][3], acc[b][c][3]);
      }
    }
  }
  const float a0 = id0[i], a1 = id1[j];
  const f32x4v a2 = *reinterpret_cast<const f32x4v*>(id2 + (kv << 2));
  // Gathers first, for every batch row, stores afterwards: with loads AND stores in flight the wait counter is no
  // longer in order and every consumed gather would drain the stores issued before it (one store latency per row).
  if (C == 1) {
    f32x4v res[BT];
    f32x4v tg[NCC ? BT : 1];
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      if (b >= B) break;
      const __amdgpu_buffer_rsrc_t rsrc =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img + (int64_t)b * V), (short)0, (int)(V * 4), 0x00020000);
#pragma unroll
      for (int v = 0; v < 4; ++v)  // deform_field = disp_field + id_transform
        res[b][v] = tri_sample_fast<SCALE>(rsrc, acc[b][0][v] + a0, acc[b][1][v] + a1, acc[b][2][v] + a2[v], D, W, H, sD);
      if constexpr (NCC) tg[b] = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(target + (int64_t)b * Vs + m));
    }
    if (active) {
#pragma unroll
      for (int b = 0; b < BT; ++b) {
        if (b >= B) break;
        float* dp = disp_out + (int64_t)b * 3 * Vs + m;
        float* pp = phi_out + (int64_t)b * 3 * Vs + m;
        f32x4v p0, p1, p2;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          p0[v] = acc[b][0][v] + a0;
          p1[v] = acc[b][1][v] + a1;
          p2[v] = acc[b][2][v] + a2[v];
        }
        __builtin_nontemporal_store(acc[b][0], reinterpret_cast<f32x4v*>(dp));
        __builtin_nontemporal_store(acc[b][1], reinterpret_cast<f32x4v*>(dp + Vs));
        __builtin_nontemporal_store(acc[b][2], reinterpret_cast<f32x4v*>(dp + 2 * Vs));
        __builtin_nontemporal_store(p0, reinterpret_cast<f32x4v*>(pp));
        __builtin_nontemporal_store(p1, reinterpret_cast<f32x4v*>(pp + Vs));
        __builtin_nontemporal_store(p2, reinterpret_cast<f32x4v*>(pp + 2 * Vs));
        __builtin_nontemporal_store(res[b], reinterpret_cast<f32x4v*>(warped + (int64_t)b * Vs + m));
      }
    }
    if constexpr (NCC) {
      // five fp64 moments per batch row over this thread's 4 voxels (products of two fp32 are exact in fp64) …
      double s[BT][5];
#pragma unroll
      for (int b = 0; b < BT; ++b) {
        if (b < B && active) {
          const double x0 = res[b][0], x1 = res[b][1], x2 = res[b][2], x3 = res[b][3];
          const double y0 = tg[b][0], y1 = tg[b][1], y2 = tg[b][2], y3 = tg[b][3];
          s[b][0] = (x0 + x1) + (x2 + x3);
          s[b][1] = (y0 + y1) + (y2 + y3);
          s[b][2] = (x0 * y0 + x1 * y1) + (x2 * y2 + x3 * y3);
          s[b][3] = (x0 * x0 + x1 * x1) + (x2 * x2 + x3 * x3);
          s[b][4] = (y0 * y0 + y1 * y1) + (y2 * y2 + y3 * y3);
        } else {
#pragma unroll
          for (int q = 0; q < 5; ++q) s[b][q] = 0.0;
        }
      }
      // … reduced over the wave by a halving butterfly: at step k a lane keeps the half of its rows that bit k of its
      // lane id selects and receives its partner's sums for them, so BT·5 sums cost ≈ BT·5 shuffles instead of
      // BT·5·6; after log2(BT) steps a lane holds ONE row, b = bitreverse(lane & (BT-1)), summed over BT lanes
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
      for (int half = BT / 2, bit = 1; half >= 1; half >>= 1, bit <<= 1) {
        const bool up = lane & bit;
#pragma unroll
        for (int b = 0; b < half; ++b)
#pragma unroll
          for (int q = 0; q < 5; ++q) {
            const double lo = s[b][q], hi = s[b + half][q];
            const double got = __shfl_xor(up ? lo : hi, bit, 64);
            s[b][q] = (up ? hi : lo) + got;
          }
      }
#pragma unroll
      for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int o = BT; o < 64; o <<= 1) s[0][q] += __shfl_xor(s[0][q], o, 64);
      // one partial per WAVE (no LDS round, no block barrier in the epilogue): [b][4*block + wave][5]
      if (lane < BT) {
        int brow = 0;  // bit-reverse of the lane's low log2(BT) bits
#pragma unroll
        for (int bit = 1, w = BT / 2; w >= 1; bit <<= 1, w >>= 1) brow |= (lane & bit) ? w : 0;
        if (brow < B) {
          const int64_t nprt = (int64_t)gridDim.x * gridDim.y * 4;
          const int64_t prt = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave;
          double* dst = ncc_partial + ((int64_t)brow * nprt + prt) * 5;
#pragma unroll
          for (int q = 0; q < 5; ++q) dst[q] = s[0][q];
        }
      }
    }
    return;
  }
  if (!active) return;
#pragma unroll
  for (int b = 0; b < BT; ++b) {  // several image channels: row by row
    if (b >= B) break;
    float* dp = disp_out + (int64_t)b * 3 * Vs + m;
    __builtin_nontemporal_store(acc[b][0], reinterpret_cast<f32x4v*>(dp));
    __builtin_nontemporal_store(acc[b][1], reinterpret_cast<f32x4v*>(dp + Vs));
    __builtin_nontemporal_store(acc[b][2], reinterpret_cast<f32x4v*>(dp + 2 * Vs));
    f32x4v p0, p1, p2;  // deform_field = disp_field + id_transform
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      p0[v] = acc[b][0][v] + a0;
      p1[v] = acc[b][1][v] + a1;
      p2[v] = acc[b][2][v] + a2[v];
    }
    float* pp = phi_out + (int64_t)b * 3 * Vs + m;
    __builtin_nontemporal_store(p0, reinterpret_cast<f32x4v*>(pp));
    __builtin_nontemporal_store(p1, reinterpret_cast<f32x4v*>(pp + Vs));
    __builtin_nontemporal_store(p2, reinterpret_cast<f32x4v*>(pp + 2 * Vs));
    for (int c = 0; c < C; ++c) {
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(img + ((int64_t)b * C + c) * V), (short)0, (int)(V * 4), 0x00020000);
      f32x4v res;
#pragma unroll
      for (int v = 0; v < 4; ++v) res[v] = tri_sample_fast<SCALE>(rsrc, p0[v], p1[v], p2[v], D, W, H, sD);
      __builtin_nontemporal_store(res, reinterpret_cast<f32x4v*>(warped + ((int64_t)b * C + c) * Vs + m));
    }
  }
}


__global__ __launch_bounds__(256) void mask_compose_kernel(const float* __restrict__ img,
                                                           const float* __restrict__ seg,
                                                           float* __restrict__ out, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = (img[i] + 1.0f) * seg[i] - 1.0f;
}

template <int VEC, bool SCALE, bool BORDER, bool NEAREST>
int launch_warp(const float* img, const float* seg, const float* disp, const float* id0,
                const float* id1, const float* id2, float* phi_out, float* warped, int B, int C,
                int D, int W, int H, int Dn, hipStream_t st) {
  const int64_t per_b = (int64_t)Dn * W * (H / VEC);
  const int64_t nblk = (per_b + 255) / 256;
  if (nblk > 0x7fffffffLL || B > 65535) return LR_EINVAL;
  const dim3 grid((unsigned)nblk, (unsigned)B), block(256);
  if (seg)
    hipLaunchKernelGGL((warp_kernel<VEC, SCALE, BORDER, NEAREST, true>), grid, block, 0, st, img,
                       seg, disp, id0, id1, id2, phi_out, warped, B, C, D, W, H, Dn);
  else
    hipLaunchKernelGGL((warp_kernel<VEC, SCALE, BORDER, NEAREST, false>), grid, block, 0, st, img,
                       seg, disp, id0, id1, id2, phi_out, warped, B, C, D, W, H, Dn);
  return lr_launch_status();
}

template <int VEC>
int dispatch_warp(int flags, const float* img, const float* seg, const float* disp,
                  const float* id0, const float* id1, const float* id2, float* phi_out,
                  float* warped, int B, int C, int D, int W, int H, int Dn, hipStream_t st) {
#define LR_W(S, Bo, N) \
  return launch_warp<VEC, S, Bo, N>(img, seg, disp, id0, id1, id2, phi_out, warped, B, C, D, W, H, Dn, st)
  const bool s = flags & LR_WARP_USING_SCALE, bo = flags & LR_WARP_BORDER, n = flags & LR_WARP_NEAREST;
  if (s) {
    if (bo) { if (n) LR_W(true, true, true); else LR_W(true, true, false); }
    else    { if (n) LR_W(true, false, true); else LR_W(true, false, false); }
  } else {
    if (bo) { if (n) LR_W(false, true, true); else LR_W(false, true, false); }
    else    { if (n) LR_W(false, false, true); else LR_W(false, false, false); }
  }
#undef LR_W
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int lr_warp_trilinear_f32(const float* img, const float* seg, const float* disp,
                                     const float* id0, const float* id1, const float* id2,
                                     float* phi_out, float* warped, int B, int C, int D, int W,
                                     int H, int d0, int d1, int flags, void* stream) {
  if (!img || !disp || !warped) return LR_ENULL;
  if (B < 1 || C < 1 || D < 1 || W < 1 || H < 1) return LR_EINVAL;
  if (d0 < 0 || d1 > D || d1 <= d0) return LR_EINVAL;
  if (flags & ~(LR_WARP_USING_SCALE | LR_WARP_BORDER | LR_WARP_NEAREST)) return LR_EINVAL;
  const bool any_id = id0 || id1 || id2, all_id = id0 && id1 && id2;
  if (any_id && !all_id) return LR_ENULL;
  const int Dn = d1 - d0;
  const bool vec4 = (H % 4 == 0) && aligned16(disp) && aligned16(warped) &&
                    (!phi_out || aligned16(phi_out));
  if (vec4 && !seg && !(flags & (LR_WARP_BORDER | LR_WARP_NEAREST)) && (!id2 || aligned16(id2)) &&
      !lr_sw_set(LR_SW_WARP_GENERAL)) {
    const int64_t sD = (int64_t)W * H, V = sD * D;
    if (V * 4 + sD * 4 <= 0x80000000LL && sD < (1 << 23) && sD / 4 <= (1 << 20) && Dn <= 65535 && B <= 65535) {
      const dim3 grid((unsigned)((sD / 4 + 255) / 256), (unsigned)Dn, (unsigned)B), block(256);
      const float rcp_hv = 1.0f / (float)(H / 4);
      if (flags & LR_WARP_USING_SCALE)
        hipLaunchKernelGGL(warp_tri_fast_kernel<true>, grid, block, 0, lr_stream(stream), img, disp, id0, id1, id2,
                           phi_out, warped, C, D, W, H, Dn, rcp_hv);
      else
        hipLaunchKernelGGL(warp_tri_fast_kernel<false>, grid, block, 0, lr_stream(stream), img, disp, id0, id1, id2,
                           phi_out, warped, C, D, W, H, Dn, rcp_hv);
      return lr_launch_status();
    }
  }
  if (vec4)
    return dispatch_warp<4>(flags, img, seg, disp, id0, id1, id2, phi_out, warped, B, C, D, W, H,
                            Dn, lr_stream(stream));
  return dispatch_warp<1>(flags, img, seg, disp, id0, id1, id2, phi_out, warped, B, C, D, W, H, Dn,
                          lr_stream(stream));
}

static int pca_warp_impl(bool bf, const float* coefs, const float* basis, const float* mean, const float* img,
                         const float* id0, const float* id1, const float* id2, float* disp, float* phi, float* warped,
                         int B, int L, int C, int D, int W, int H, int d0, int d1, int64_t ldb, int64_t bcs, int flags,
                         const float* target, double* ncc_partial, double* ncc_moments, void* stream) {
  if (!coefs || !basis || !mean || !img || !id0 || !id1 || !id2 || !disp || !phi || !warped) return LR_ENULL;
  if (B < 1 || L < 1 || C < 1 || D < 1 || W < 1 || H < 1) return LR_EINVAL;
  if (d0 < 0 || d1 > D || d1 <= d0) return LR_EINVAL;
  if (flags & ~LR_WARP_USING_SCALE) return LR_EUNSUPPORTED;  // zeros padding, trilinear only (the model's case)
  const bool ncc = target != nullptr;
  if (ncc && (!ncc_partial || !ncc_moments)) return LR_ENULL;
  if (ncc && C != 1) return LR_EUNSUPPORTED;
  const int Dn = d1 - d0;
  const int64_t sD = (int64_t)W * H, V = sD * D, Vs = sD * Dn;
  if (B > 256 || (B > 8 && ncc) || L > 2048 || (H & 3) || bcs < Vs || ldb < 2 * bcs + Vs) return LR_EUNSUPPORTED;  // with the moments: chunks of 8 (ops.pca_warp)
  if (!(V * 4 + sD * 4 <= 0x80000000LL && sD < (1 << 23) && sD / 4 <= (1 << 20) && D <= 65535)) return LR_EUNSUPPORTED;
  if (((reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(disp) | reinterpret_cast<uintptr_t>(phi) |
        reinterpret_cast<uintptr_t>(warped) | reinterpret_cast<uintptr_t>(id2) | reinterpret_cast<uintptr_t>(target)) & 15u) ||
      (reinterpret_cast<uintptr_t>(basis) & (bf ? 7u : 15u)) || (ldb & 3) || (bcs & 3))
    return LR_EALIGN;
  const int gx = (int)((sD / 4 + 255) / 256), nch = (B + 7) / 8;
  const int64_t flat = ((int64_t)gx * Dn + 7) / 8 * 8 * nch;   // B > 8: one block per (tile, chunk of 8 rows), chunks of a tile on one XCD
  if (flat > 0x7fffffffLL) return LR_EUNSUPPORTED;
  const dim3 grid(nch > 1 ? (unsigned)flat : (unsigned)gx, nch > 1 ? 1u : (unsigned)Dn), block(256);
  const float rcp_hv = 1.0f / (float)(H / 4);
  hipStream_t st = lr_stream(stream);
  const bool sc = flags & LR_WARP_USING_SCALE;
#define LR_PW2(BFV, SCV, BTV, NCV) hipLaunchKernelGGL((pca_warp_kernel<BFV, SCV, BTV, NCV>), grid, block, (size_t)L * BTV * sizeof(float), st, coefs, basis, mean, img, id0, id1, id2, disp, phi, warped, B, L, C, D, W, H, ldb, rcp_hv, bcs, target, ncc_partial, gx, Dn, nch)
#define LR_PW1(BFV, SCV, BTV) do { if (ncc) LR_PW2(BFV, SCV, BTV, true); else LR_PW2(BFV, SCV, BTV, false); } while (0)
#define LR_PWM(BFV, SCV) hipLaunchKernelGGL((pca_warp_kernel<BFV, SCV, 8, false, true>), grid, block, (size_t)L * 8 * sizeof(float), st, coefs, basis, mean, img, id0, id1, id2, disp, phi, warped, B, L, C, D, W, H, ldb, rcp_hv, bcs, target, ncc_partial, gx, Dn, nch)
#define LR_PW(BFV, SCV) do { if (nch > 1) LR_PWM(BFV, SCV); else if (B <= 4) LR_PW1(BFV, SCV, 4); else LR_PW1(BFV, SCV, 8); } while (0)
  if (bf) { if (sc) LR_PW(true, true); else LR_PW(true, false); }
  else    { if (sc) LR_PW(false, true); else LR_PW(false, false); }
#undef LR_PW
#undef LR_PWM
#undef LR_PW1
#undef LR_PW2
  if (int e = lr_launch_status()) return e;
  if (ncc) return lr_internal_ncc_reduce(ncc_partial, ncc_moments, B, (int)(grid.x * grid.y * 4), st);
  return LR_OK;
}

extern "C" int lr_pca_warp_f32(const float* coefs, const float* basis, const float* mean, const float* img,
                               const float* id0, const float* id1, const float* id2, float* disp, float* phi,
                               float* warped, int B, int L, int C, int D, int W, int H, int64_t ldb, int flags,
                               void* stream) {
  return pca_warp_impl(false, coefs, basis, mean, img, id0, id1, id2, disp, phi, warped, B, L, C, D, W, H, 0, D, ldb,
                       (int64_t)D * W * H, flags, nullptr, nullptr, nullptr, stream);
}

extern "C" int lr_pca_warp_bf16basis_f32(const float* coefs, const void* basis_bf16, const float* mean, const float* img,
                                         const float* id0, const float* id1, const float* id2, float* disp, float* phi,
                                         float* warped, int B, int L, int C, int D, int W, int H, int64_t ldb, int flags,
                                         void* stream) {
  return pca_warp_impl(true, coefs, reinterpret_cast<const float*>(basis_bf16), mean, img, id0, id1, id2, disp, phi, warped,
                       B, L, C, D, W, H, 0, D, ldb, (int64_t)D * W * H, flags, nullptr, nullptr, nullptr, stream);
}

extern "C" int lr_pca_warp_slab_f32(const float* coefs, const void* basis, int basis_is_bf16, const float* mean,
                                    const float* img, const float* id0, const float* id1, const float* id2, float* disp,
                                    float* phi, float* warped, int B, int L, int C, int D, int W, int H, int d0, int d1,
                                    int64_t ldb, int64_t basis_comp_stride, int flags, const float* target,
                                    double* ncc_partial, double* ncc_moments, void* stream) {
  return pca_warp_impl(basis_is_bf16 != 0, coefs, reinterpret_cast<const float*>(basis), mean, img, id0, id1, id2, disp, phi,
                       warped, B, L, C, D, W, H, d0, d1, ldb, basis_comp_stride, flags, target, ncc_partial, ncc_moments,
                       stream);
}

extern "C" int lr_mask_compose_f32(const float* img, const float* seg, float* out, int64_t n,
                                   void* stream) {
  if (!img || !seg || !out) return LR_ENULL;
  if (n < 0) return LR_EINVAL;
  if (n == 0) return LR_OK;
  int64_t nblk = (n + 255) / 256;
  if (nblk > 256 * 16) nblk = 256 * 16;
  hipLaunchKernelGGL(mask_compose_kernel, dim3((unsigned)nblk), dim3(256), 0, lr_stream(stream),
                     img, seg, out, n);
  return lr_launch_status();
}
