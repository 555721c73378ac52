// drr_forward.hip — K1: cone-beam DRR forward projector (ray marching, one
// trilinear sample per coronal plane, sample grid derived in registers).
//
// Replaces (reference file:line)
//   src/liftreg/utils/sdct_projection_utils.py:15-57   project_grid_multi
//   src/liftreg/utils/sdct_projection_utils.py:59-100  calculate_projection (:81 grid_sample, :85 *0.1)
//   src/liftreg/utils/sdct_projection_utils.py:6-9     calc_relative_atten_coef (LR_DRR_HU_INPUT)
//   tools/preprocessingDRR.py:135-136                  np.flip(axis=1)          (LR_DRR_FLIP_W)
//
// The reference materialises a (P,Rd,Rh,W,3) fp32 grid (12*P*Rd*Rh*W bytes);
// this kernel never does.  Lanes of a wavefront walk 64 neighbouring detector
// columns (the Rh axis pairs with H, the volume's fastest axis), so each of the
// 8 trilinear taps of a plane is one ~256-byte contiguous run per wavefront.
// Gather-bound (L2 / Infinity-Cache resident volume), not HBM-bound.
//
// Coordinates follow the reference's fp32 op order exactly (no contraction,
// IEEE divide) because float rounding moves ~50 % of the samples off their
// coronal plane by up to 7e-5 (SURVEY.md headline fact 6) and floor() must agree.
#include "lr_common.h"

namespace {

struct RaySetup {
  float ihx, ihy, ihz;  // unit direction Î
  float rc;             // 1 / Î_y
  float dx;             // mm per unit-y step
};

// project_grid_multi (:31-41): I = pix - e ; dx = ‖(I * (1/I_y)) ⊙ spacing‖ ; Î = I/‖I‖.
// torch.norm on CPU accumulates with an FMA chain: fma(z,z,fma(y,y,x*x)).
__device__ __forceinline__ RaySetup ray_setup(int a, int b, int Rd, int Rh, float ex, float ey,
                                              float ez, float sp0, float sp1, float sp2) {
  const float px = (float)a - 0.5f * (float)Rd;  // linspace(-res_d/2, res_d/2-1, res_d)[a]
  const float pz = (float)b - 0.5f * (float)Rh;
  const float ix = px + (-ex), iy = 0.0f + (-ey), iz = pz + (-ez);
  const float rcp = 1.0f / iy;
  const float d0 = (ix * rcp) * sp0, d1 = (iy * rcp) * sp1, d2 = (iz * rcp) * sp2;
  RaySetup r;
  r.dx = sqrtf(fmaf(d2, d2, fmaf(d1, d1, d0 * d0)));
  const float nrm = sqrtf(fmaf(iz, iz, fmaf(iy, iy, ix * ix)));
  r.ihx = ix / nrm;
  r.ihy = iy / nrm;
  r.ihz = iz / nrm;
  r.rc = 1.0f / r.ihy;
  return r;
}

// Sample position on plane y=j in ATen pixel units (d,w,h): grid = Î*T + e (:50-51),
// normalise (:54-56), un-normalise align_corners=True.
__device__ __forceinline__ void sample_grid(const RaySetup& r, int j, float ex, float ey, float ez,
                                            int D, int W, int H, float& gx, float& gy, float& gz) {
  const float t = r.rc * ((float)j - ey);
  const float x = r.ihx * t + ex;
  const float y = r.ihy * t + ey;
  const float z = r.ihz * t + ez;
  // x / 2^k == x * 2^-k bit for bit (exact scaling), so power-of-two extents skip the IEEE divide sequence
  gx = ((D & (D - 1)) == 0 ? x * (1.0f / (float)D) : x / (float)D) * 2.0f;
  gy = ((y - 0.0f) / ((float)W - 1.0f)) * 2.0f + -1.0f;
  gz = ((H & (H - 1)) == 0 ? z * (1.0f / (float)H) : z / (float)H) * 2.0f;
}
__device__ __forceinline__ void sample_pix(const RaySetup& r, int j, float ex, float ey, float ez,
                                           int D, int W, int H, float& pd, float& pw, float& ph) {
  float gx, gy, gz;
  sample_grid(r, j, ex, ey, ez, D, W, H, gx, gy, gz);
  pd = lr_unnormalize(gx, D);
  pw = lr_unnormalize(gy, W);
  ph = lr_unnormalize(gz, H);
}

// x / d as q = x r, e = fma(-q, d, x), q' = fma(e, r, q) with r = RN(1 / d): the IEEE quotient for every divisor on the
// launcher's whitelist and every |x| in [2^-20, 2^12] (checked exhaustively on the CPU: tests/test_oracle_golden_r2.py through
// oracle/liftreg_oracle.c: or_fastdiv_mismatches) — sample coordinates are 0 or at least one ulp of the emitter distance
// (> 2^-15) and stay below 2^12 voxels.  Three vector-ALU operations instead of the ~11 of the IEEE divide sequence.
struct FastDiv {
  float d, r;     // divisor, RN(1 / d)
  int pow2;       // d is a power of two: x * r is the exact quotient
};
template <bool FD>
__device__ __forceinline__ float div_by(float x, const FastDiv& f) {
  if constexpr (FD) {   // (a power of two: q is already the exact quotient, e = 0, q' = q — no special case)
    const float q = x * f.r;
    const float e = fmaf(-q, f.d, x);
    return fmaf(e, f.r, q);
  } else {
    return f.pow2 ? x * f.r : x / f.d;
  }
}
template <bool FD>
__device__ __forceinline__ void sample_pix_fd(const RaySetup& r, int j, float ex, float ey, float ez, int D, int W, int H,
                                              const FastDiv& fD, const FastDiv& fW, const FastDiv& fH, float& pd, float& pw,
                                              float& ph) {
  const float t = r.rc * ((float)j - ey);
  const float x = r.ihx * t + ex;
  const float y = r.ihy * t + ey;
  const float z = r.ihz * t + ez;
  const float gx = div_by<FD>(x, fD) * 2.0f;
  const float gy = div_by<FD>(y - 0.0f, fW) * 2.0f + -1.0f;
  const float gz = div_by<FD>(z, fH) * 2.0f;
  pd = lr_unnormalize(gx, D);
  pw = lr_unnormalize(gy, W);
  ph = lr_unnormalize(gz, H);
}

struct Axis {
  int i0, i1;
  float w0, w1;  // (i1 - pix), (pix - i0) — ATen's generic 3D kernel
  bool ok0, ok1;
};

__device__ __forceinline__ Axis make_axis(float pix, int lo, int hi /*valid: lo <= i < hi*/) {
  Axis a;
  // Anything at or beyond one cell outside the volume contributes nothing.
  if (!(pix > (float)(lo - 1) && pix < (float)hi)) {
    a.i0 = a.i1 = lo;
    a.w0 = a.w1 = 0.0f;
    a.ok0 = a.ok1 = false;
    return a;
  }
  const float fl = floorf(pix);
  const int i0 = (int)fl, i1 = i0 + 1;
  a.w0 = (float)i1 - pix;
  a.w1 = pix - (float)i0;
  a.ok0 = i0 >= lo;  // i0 < hi implied
  a.ok1 = i1 < hi;   // i1 >= lo implied
  a.i0 = max(i0, lo);
  a.i1 = min(i1, hi - 1);
  return a;
}

template <bool HU>
__device__ __forceinline__ float load_mu(const float* p, bool ok) {
  float v = *p;
  if constexpr (HU) {
    v = (v < -1000.0f) ? -1000.0f : v;         // new_img[new_img < -1000] = -1000
    v = ((v + 1000.0f) / 1000.0f) * 0.2f;      // (new_img + 1000.) / 1000. * 0.2
  }
  return ok ? v : 0.0f;
}

template <bool HU>
__device__ __forceinline__ float mu_of(float v) {
  if constexpr (HU) {
    v = (v < -1000.0f) ? -1000.0f : v;
    v = ((v + 1000.0f) / 1000.0f) * 0.2f;
  }
  return v;
}
// calc_relative_atten_coef with the division by 1000 as a multiplication and ONE Markstein correction step:
//   q = x r,  e = fma(-q, 1000, x),  q' = fma(e, r, q)   with r = RN(1/1000)
// q' equals the IEEE quotient x / 1000 for EVERY finite x >= 2^-100 (checked exhaustively over all 2.1e9 non-negative floats on
// the CPU: the only mismatches are quotients in the denormal range, x < 1e-34) — and x = max(HU, -1000) + 1000 is 0 or at
// least one ulp of 1000 (6e-5).  Six vector-ALU operations per tap instead of the ~14 of the IEEE divide sequence: cheap
// enough to fold HU -> mu into the projector's tap loads, same bits as hu_to_mu_kernel.
__device__ __forceinline__ float mu_of_fast(float v) {
  v = (v < -1000.0f) ? -1000.0f : v;
  const float x = v + 1000.0f;
  const float r = 1.0f / 1000.0f;        // RN(1/1000), a compile-time constant
  const float q = x * r;
  const float e = fmaf(-q, 1000.0f, x);
  const float qc = fmaf(e, r, q);
  return qc * 0.2f;
}

// two conversions as packed fp32 operations (v_pk_add / v_pk_mul / v_pk_fma: the IEEE result of each element, as mu_of_fast)
__device__ __forceinline__ void mu_of_fast2(float& a, float& b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  // new_img[new_img < -1000] = -1000 as one v_max_f32 per value (finite input: the header's contract for LR_DRR_HU_INPUT; inline
  // asm: llvm.maxnum on a loaded value first canonicalises it — a second instruction)
  float ma, mb;
  asm("v_max_f32 %0, %1, %2" : "=v"(ma) : "v"(a), "v"(-1000.0f));
  asm("v_max_f32 %0, %1, %2" : "=v"(mb) : "v"(b), "v"(-1000.0f));
  f32x2 v = {ma, mb};
  const f32x2 k1000 = {1000.0f, 1000.0f}, r = {1.0f / 1000.0f, 1.0f / 1000.0f}, k02 = {0.2f, 0.2f};
  const f32x2 x = v + k1000;
  const f32x2 q = x * r;
  const f32x2 e = __builtin_elementwise_fma(-q, k1000, x);
  const f32x2 qc = __builtin_elementwise_fma(e, r, q);
  const f32x2 m = qc * k02;
  a = m[0];
  b = m[1];
}

template <bool HU>
__device__ __forceinline__ void load_mu_pair(const float* p, int shift, bool ok0, bool ok1, float& v0,
                                             float& v1) {
  typedef float f32x2 __attribute__((ext_vector_type(2), aligned(4)));
  const f32x2 q = *reinterpret_cast<const f32x2*>(p);
  const float a = mu_of<HU>(shift > 0 ? q.y : q.x), b = mu_of<HU>(shift < 0 ? q.x : q.y);
  v0 = ok0 ? a : 0.0f;
  v1 = ok1 ? b : 0.0f;
}

// blockDim = (64, R): lane ↔ detector column b; row r ↔ (a, segment).
template <bool HU, bool FLIP>
__global__ __launch_bounds__(1024) void drr_forward_kernel(
    const float* __restrict__ vol, LrPoses poses, float sp0, float sp1, float sp2,
    float* __restrict__ out, int D, int W, int H, int d0, int d1, int P, int Rd, int Rh,
    int nseg) {
  extern __shared__ float part[];  // [R][64]
  const int lane = threadIdx.x, row = threadIdx.y, R = blockDim.y;
  const int a_per_blk = R / nseg;
  const int nbx = (Rh + 63) >> 6;
  // blockIdx.x enumerates (p, a-group, b-chunk), remapped so one XCD walks a
  // contiguous run of detector rows (they share volume x-rows in its L2).
  const unsigned nblk = gridDim.x;
  const unsigned lb = lr_xcd_remap(blockIdx.x, nblk);
  const int bx = lb % nbx;
  const int ag = (lb / nbx) % ((Rd + a_per_blk - 1) / a_per_blk);
  const int p = lb / nbx / ((Rd + a_per_blk - 1) / a_per_blk);
  const int a = ag * a_per_blk + row / nseg;
  const int seg = row % nseg;
  const int b = bx * 64 + lane;
  const bool live = (a < Rd) && (b < Rh);

  float acc = 0.0f;
  float dxv = 0.0f;
  if (live) {
    const float ex = poses.e[p][0], ey = poses.e[p][1], ez = poses.e[p][2];
    const RaySetup rs = ray_setup(a, b, Rd, Rh, ex, ey, ez, sp0, sp1, sp2);
    dxv = rs.dx;
    const int per = (W + nseg - 1) / nseg;
    const int j0 = seg * per, j1 = min(W, j0 + per);
    const int64_t sW = H, sD = (int64_t)W * H;
    for (int j = j0; j < j1; ++j) {
      float pd, pw, ph;
      sample_pix(rs, j, ex, ey, ez, D, W, H, pd, pw, ph);
      const Axis az = make_axis(pd, d0, d1);  // grid z ↔ D (slab rows only)
      const Axis ay = make_axis(pw, 0, W);    // grid y ↔ W
      const Axis ax = make_axis(ph, 0, H);    // grid x ↔ H
      const int y0 = FLIP ? (W - 1 - ay.i0) : ay.i0;
      const int y1 = FLIP ? (W - 1 - ay.i1) : ay.i1;
      const float* p00 = vol + (int64_t)(az.i0 - d0) * sD + (int64_t)y0 * sW;
      const float* p01 = vol + (int64_t)(az.i0 - d0) * sD + (int64_t)y1 * sW;
      const float* p10 = vol + (int64_t)(az.i1 - d0) * sD + (int64_t)y0 * sW;
      const float* p11 = vol + (int64_t)(az.i1 - d0) * sD + (int64_t)y1 * sW;
      // the two H-taps of a corner pair are neighbours: one 8-byte gather fetches both (halves the
      // gather instructions); xb = clamp(floor, 0, H-2), the in-range tap is picked out of the pair
      float v_tnw, v_tne, v_tsw, v_tse, v_bnw, v_bne, v_bsw, v_bse;
      if (H >= 2) {
        const int x0 = ax.ok0 ? ax.i0 : ax.i1 - 1;  // unclamped floor (or -1 when nothing is in range)
        const int xb = min(max(x0, 0), H - 2), shift = x0 - xb;
        load_mu_pair<HU>(p00 + xb, shift, az.ok0 && ay.ok0 && ax.ok0, az.ok0 && ay.ok0 && ax.ok1, v_tnw, v_tne);
        load_mu_pair<HU>(p01 + xb, shift, az.ok0 && ay.ok1 && ax.ok0, az.ok0 && ay.ok1 && ax.ok1, v_tsw, v_tse);
        load_mu_pair<HU>(p10 + xb, shift, az.ok1 && ay.ok0 && ax.ok0, az.ok1 && ay.ok0 && ax.ok1, v_bnw, v_bne);
        load_mu_pair<HU>(p11 + xb, shift, az.ok1 && ay.ok1 && ax.ok0, az.ok1 && ay.ok1 && ax.ok1, v_bsw, v_bse);
      } else {
        v_tnw = load_mu<HU>(p00 + ax.i0, az.ok0 && ay.ok0 && ax.ok0);
        v_tne = load_mu<HU>(p00 + ax.i1, az.ok0 && ay.ok0 && ax.ok1);
        v_tsw = load_mu<HU>(p01 + ax.i0, az.ok0 && ay.ok1 && ax.ok0);
        v_tse = load_mu<HU>(p01 + ax.i1, az.ok0 && ay.ok1 && ax.ok1);
        v_bnw = load_mu<HU>(p10 + ax.i0, az.ok1 && ay.ok0 && ax.ok0);
        v_bne = load_mu<HU>(p10 + ax.i1, az.ok1 && ay.ok0 && ax.ok1);
        v_bsw = load_mu<HU>(p11 + ax.i0, az.ok1 && ay.ok1 && ax.ok0);
        v_bse = load_mu<HU>(p11 + ax.i1, az.ok1 && ay.ok1 && ax.ok1);
      }
      // weights: (x-part * y-part) * z-part, corners in ATen's order
      float s = v_tnw * ((ax.w0 * ay.w0) * az.w0);
      s = s + v_tne * ((ax.w1 * ay.w0) * az.w0);
      s = s + v_tsw * ((ax.w0 * ay.w1) * az.w0);
      s = s + v_tse * ((ax.w1 * ay.w1) * az.w0);
      s = s + v_bnw * ((ax.w0 * ay.w0) * az.w1);
      s = s + v_bne * ((ax.w1 * ay.w0) * az.w1);
      s = s + v_bsw * ((ax.w0 * ay.w1) * az.w1);
      s = s + v_bse * ((ax.w1 * ay.w1) * az.w1);
      acc = acc + s;
    }
  }
  if (nseg == 1) {
    if (live) out[((int64_t)p * Rd + a) * Rh + b] = (acc * dxv) * 0.1f;
    return;
  }
  part[row * 64 + lane] = acc;
  __syncthreads();
  if (live && seg == 0) {
    float s = part[row * 64 + lane];
    for (int q = 1; q < nseg; ++q) s = s + part[(row + q) * 64 + lane];
    out[((int64_t)p * Rd + a) * Rh + b] = (s * dxv) * 0.1f;
  }
}

// The projector's usual case (attenuation input, H >= 2, slab under 2 GB) with half the vector-ALU work per sample;
// the general kernel above is ALU-bound (64-bit addressing, per-tap range flags and selects), not gather-bound.
//  * sample coordinates: the same code (sample_pix), then clamped to [lo-1, hi] — at the clamp values every tap is
//    either outside or has weight exactly 0, which is what the general kernel's "nothing in range" branch yields;
//  * taps: bounds-checked raw buffer loads relative to the slab, 32-bit offsets from 24-bit multiplies; a z tap
//    outside the slab falls outside the resource by itself, an out-of-range y row is pushed outside with one select;
//  * the x pair is one 8-byte load; only where a wave touches the x faces a wave-uniform branch re-bases it;
//  * products and sums in the general kernel's order: same bits.
//  * batches: blockIdx.y = the volume (vol_bs / P*Rd*Rh elements apart): one launch for B volumes of one geometry;
//  * FD: the three normalising divisions by D, W - 1, H as a reciprocal multiplication + one correction step (div_by);
//  * HU: the eight conversions of a sample as four PAIRS of packed fp32 operations (same IEEE roundings per element), and
//    the y rows outside the volume need no address select (their weight is zeroed: whatever finite value is read counts 0).
template <bool FLIP, bool HU = false, bool FD = false>
__global__ __launch_bounds__(1024) void drr_forward_fast_kernel(
    const float* __restrict__ vol, LrPoses poses, float sp0, float sp1, float sp2,
    float* __restrict__ out, int D, int W, int H, int d0, int d1, int P, int Rd, int Rh, int nseg, int64_t vol_bs,
    FastDiv fD, FastDiv fW, FastDiv fH) {
  extern __shared__ float part[];  // [R][64]
  vol += (int64_t)blockIdx.y * vol_bs;
  out += (int64_t)blockIdx.y * P * Rd * Rh;
  const int lane = threadIdx.x, row = threadIdx.y, R = blockDim.y;
  const int a_per_blk = R / nseg;
  const int nbx = (Rh + 63) >> 6;
  const unsigned lb = lr_xcd_remap(blockIdx.x, gridDim.x);
  const int nag = (Rd + a_per_blk - 1) / a_per_blk;
  const int bx = lb % nbx;
  const int ag = (lb / nbx) % nag;
  const int p = lb / nbx / nag;
  const int a = ag * a_per_blk + row / nseg;
  const int seg = row % nseg;
  const int b = bx * 64 + lane;
  const bool live = (a < Rd) && (b < Rh);

  float acc = 0.0f;
  float dxv = 0.0f;
  if (live) {
    const float ex = poses.e[p][0], ey = poses.e[p][1], ez = poses.e[p][2];
    const RaySetup rs = ray_setup(a, b, Rd, Rh, ex, ey, ez, sp0, sp1, sp2);
    dxv = rs.dx;
    const int per = (W + nseg - 1) / nseg;
    const int j0 = seg * per, j1 = min(W, j0 + per);
    const int sD = W * H, Dn = d1 - d0;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vol), (short)0, Dn * sD * 4, 0x00020000);
    constexpr int OUTSIDE = 0x20000000;  // elements; ·4 bytes = 2^31 ≥ any slab accepted by the launcher
    const float zlo = (float)(d0 - 1), zhi = (float)d1, yhi = (float)W, xhi = (float)H;
    for (int j = j0; j < j1; ++j) {
      float pd, pw, ph;
      sample_pix_fd<FD>(rs, j, ex, ey, ez, D, W, H, fD, fW, fH, pd, pw, ph);
      pd = __builtin_amdgcn_fmed3f(pd, zlo, zhi);   // NaN -> a bound; both bounds contribute nothing
      pw = __builtin_amdgcn_fmed3f(pw, -1.0f, yhi);
      ph = __builtin_amdgcn_fmed3f(ph, -1.0f, xhi);
      const float fz = floorf(pd), fy = floorf(pw), fx = floorf(ph);
      const int z0 = (int)fz - d0, y0 = (int)fy, x0 = (int)fx;
      float wz0 = (fz + 1.0f) - pd, wz1 = pd - fz;   // (i1 - pix), (pix - i0) — ATen's generic 3D kernel
      float wy0 = (fy + 1.0f) - pw, wy1 = pw - fy;
      const float wx0 = (fx + 1.0f) - ph, wx1 = ph - fx;
      const int xb = min(max(x0, 0), H - 2), shift = x0 - xb;
      float tp[8];
      bool edge;   // some tap of this sample lies outside the slab / the volume
      if constexpr (HU) {
        // HU input: ONE address per sample — (z0, lower row, xb) — and three scalar strides; a tap outside the slab / volume has
        // its axis weight zeroed in the (rare, wave-uniform) edge branch below, so what its address reads does not matter: 0
        // outside the buffer resource (raw 0 would convert to 0.2), some other finite voxel inside it.
        const int rlo = FLIP ? (W - 2 - y0) : y0;
        const unsigned o_lo = ((unsigned)__mul24(z0, sD) + (unsigned)__mul24(rlo, H) + (unsigned)xb) << 2;
        const unsigned o_hi = o_lo + ((unsigned)H << 2), p_lo = o_lo + ((unsigned)sD << 2), p_hi = o_hi + ((unsigned)sD << 2);
        const uint2 q00 = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, FLIP ? o_hi : o_lo, 0, 0));
        const uint2 q01 = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, FLIP ? o_lo : o_hi, 0, 0));
        const uint2 q10 = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, FLIP ? p_hi : p_lo, 0, 0));
        const uint2 q11 = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, FLIP ? p_lo : p_hi, 0, 0));
        tp[0] = __builtin_bit_cast(float, q00.x); tp[1] = __builtin_bit_cast(float, q00.y);
        tp[2] = __builtin_bit_cast(float, q01.x); tp[3] = __builtin_bit_cast(float, q01.y);
        tp[4] = __builtin_bit_cast(float, q10.x); tp[5] = __builtin_bit_cast(float, q10.y);
        tp[6] = __builtin_bit_cast(float, q11.x); tp[7] = __builtin_bit_cast(float, q11.y);
#pragma unroll
        for (int t8 = 0; t8 < 8; t8 += 2) mu_of_fast2(tp[t8], tp[t8 + 1]);
        edge = (unsigned)z0 > (unsigned)(Dn - 2) || (unsigned)y0 > (unsigned)(W - 2) || shift != 0;   // Dn >= 2, W >= 2: the launcher's conditions
      } else {
        const int r0 = FLIP ? (W - 1 - y0) : y0, r1 = FLIP ? r0 - 1 : r0 + 1;
        const int yo0 = ((unsigned)y0 < (unsigned)W) ? __mul24(r0, H) : OUTSIDE;
        const int yo1 = ((unsigned)(y0 + 1) < (unsigned)W) ? __mul24(r1, H) : OUTSIDE;
        const int zo0 = __mul24(z0, sD), zo1 = zo0 + sD;
        const unsigned xb4 = (unsigned)xb << 2;
        const uint2 q00 = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, ((unsigned)(zo0 + yo0) << 2) + xb4, 0, 0));
        const uint2 q01 = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, ((unsigned)(zo0 + yo1) << 2) + xb4, 0, 0));
        const uint2 q10 = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, ((unsigned)(zo1 + yo0) << 2) + xb4, 0, 0));
        const uint2 q11 = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, ((unsigned)(zo1 + yo1) << 2) + xb4, 0, 0));
        tp[0] = __builtin_bit_cast(float, q00.x); tp[1] = __builtin_bit_cast(float, q00.y);
        tp[2] = __builtin_bit_cast(float, q01.x); tp[3] = __builtin_bit_cast(float, q01.y);
        tp[4] = __builtin_bit_cast(float, q10.x); tp[5] = __builtin_bit_cast(float, q10.y);
        tp[6] = __builtin_bit_cast(float, q11.x); tp[7] = __builtin_bit_cast(float, q11.y);
        edge = shift != 0;
      }
      // A wave at a face of the volume (rare): x0 in {-1, H-1, H} — the loaded pair is (xb, xb+1) = x0 shifted by `shift`; instead
      // of moving eight taps the two x weights move (a tap outside the volume gets weight 0; x + 0 = 0 + x, so the sum keeps its
      // bits); HU input: the z / y weights of taps outside are zeroed (0 * mu = +0, the bits of the mu-input kernel where the tap
      // itself reads 0).
      float wxa = wx0, wxb = wx1;
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(edge) != 0, 0)) {
        wxa = shift == 0 ? wx0 : shift < 0 ? wx1 : 0.0f;
        wxb = shift == 0 ? wx1 : shift == 1 ? wx0 : 0.0f;
        if constexpr (HU) {
          wz0 = ((unsigned)z0 < (unsigned)Dn) ? wz0 : 0.0f;
          wz1 = ((unsigned)(z0 + 1) < (unsigned)Dn) ? wz1 : 0.0f;
          wy0 = ((unsigned)y0 < (unsigned)W) ? wy0 : 0.0f;
          wy1 = ((unsigned)(y0 + 1) < (unsigned)W) ? wy1 : 0.0f;
        }
      }
      float s = tp[0] * ((wxa * wy0) * wz0);
      s = s + tp[1] * ((wxb * wy0) * wz0);
      s = s + tp[2] * ((wxa * wy1) * wz0);
      s = s + tp[3] * ((wxb * wy1) * wz0);
      s = s + tp[4] * ((wxa * wy0) * wz1);
      s = s + tp[5] * ((wxb * wy0) * wz1);
      s = s + tp[6] * ((wxa * wy1) * wz1);
      s = s + tp[7] * ((wxb * wy1) * wz1);
      acc = acc + s;
    }
  }
  if (nseg == 1) {
    if (live) out[((int64_t)p * Rd + a) * Rh + b] = (acc * dxv) * 0.1f;
    return;
  }
  part[row * 64 + lane] = acc;
  __syncthreads();
  if (live && seg == 0) {
    float s = part[row * 64 + lane];
    for (int q = 1; q < nseg; ++q) s = s + part[(row + q) * 64 + lane];
    out[((int64_t)p * Rd + a) * Rh + b] = (s * dxv) * 0.1f;
  }
}

__global__ __launch_bounds__(256) void drr_coords_kernel(LrPoses poses, float sp0, float sp1,
                                                         float sp2, float* __restrict__ pix,
                                                         float* __restrict__ dx, int D, int W,
                                                         int H, int P, int Rd, int Rh,
                                                         int normalized) {
  const int64_t total = (int64_t)P * Rd * Rh;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int b = (int)(idx % Rh);
  const int a = (int)((idx / Rh) % Rd);
  const int p = (int)(idx / Rh / Rd);
  const float ex = poses.e[p][0], ey = poses.e[p][1], ez = poses.e[p][2];
  const RaySetup rs = ray_setup(a, b, Rd, Rh, ex, ey, ez, sp0, sp1, sp2);
  if (dx) dx[idx] = rs.dx;
  if (pix) {
    for (int j = 0; j < W; ++j) {
      float pd, pw, ph;
      if (normalized) sample_grid(rs, j, ex, ey, ez, D, W, H, pd, pw, ph);
      else sample_pix(rs, j, ex, ey, ez, D, W, H, pd, pw, ph);
      float* o = pix + (idx * W + j) * 3;
      o[0] = pd;
      o[1] = pw;
      o[2] = ph;
    }
  }
}

// calc_relative_atten_coef (sdct_projection_utils.py:6-9) as its own pass: one conversion per VOXEL.  Folded into the
// projector's taps (LR_DRR_HU_INPUT) the same IEEE divide runs once per TAP — 16x as often at 256^3 / 2x256^2.
__global__ __launch_bounds__(256) void hu_to_mu_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = mu_of<true>(in[i]);
}

int fill_poses(LrPoses& lp, const float* poses, int P) {
  if (!poses) return LR_ENULL;
  if (P < 1 || P > LR_MAX_VIEWS) return LR_EINVAL;
  for (int p = 0; p < P; ++p)
    for (int c = 0; c < 3; ++c) lp.e[p][c] = poses[p * 3 + c];
  return LR_OK;
}

}  // namespace

// divisors the reciprocal division of div_by is validated for (tests/test_oracle_golden_r2.py: every |x| in [2^-20, 2^12])
static bool fastdiv_ok(int d) {
  if (d > 0 && (d & (d - 1)) == 0) return true;
  static const int ok[] = {31, 63, 95, 127, 159, 191, 255, 383, 511, 96, 160, 192, 384};
  for (int v : ok)
    if (v == d) return true;
  return false;
}
static FastDiv make_fastdiv(int d) {
  FastDiv f;
  f.d = (float)d;
  f.r = 1.0f / (float)d;
  f.pow2 = (d > 0 && (d & (d - 1)) == 0) ? 1 : 0;
  return f;
}

static int drr_forward_impl(const float* vol_slab, int64_t vol_batch_stride, const float* poses, const float* spacing,
                            float* out, int B, int D, int W, int H, int d0, int d1, int P, int Rd,
                            int Rh, int flags, int nseg, void* stream) {
  if (!vol_slab || !out || !spacing) return LR_ENULL;
  if (B < 1 || B > 65535 || D < 1 || W < 2 || H < 1 || Rd < 1 || Rh < 1) return LR_EINVAL;
  if (d0 < 0 || d1 > D || d1 <= d0) return LR_EINVAL;
  if (flags & ~(LR_DRR_HU_INPUT | LR_DRR_FLIP_W)) return LR_EINVAL;
  if (B > 1 && vol_batch_stride < (int64_t)(d1 - d0) * W * H) return LR_EINVAL;
  LrPoses lp;
  if (int e = fill_poses(lp, poses, P)) return e;
  if (nseg == 0) {
    // enough lanes to fill 256 CUs x 32 waves with ONE volume, capped at 16 runs per ray (the batch size stays out of the rule: a
    // volume's DRR has the same bits in a batch as alone)
    const int64_t rays = (int64_t)P * Rd * ((Rh + 63) / 64 * 64);
    nseg = 1;
    while (nseg < 16 && rays * nseg < 256LL * 4096 && W / (nseg * 2) >= 8) nseg *= 2;  // measured: 8 runs at 2x256^2
  }
  if (nseg != 1 && nseg != 2 && nseg != 4 && nseg != 8 && nseg != 16) return LR_EINVAL;
  const int R = nseg > 4 ? nseg : 4;
  const int a_per_blk = R / nseg;
  const int64_t nblk = (int64_t)P * ((Rd + a_per_blk - 1) / a_per_blk) * ((Rh + 63) / 64);
  if (nblk > 0x7fffffffLL) return LR_EINVAL;
  const dim3 grid((unsigned)nblk, (unsigned)B), block(64, R);
  const size_t lds = (size_t)R * 64 * sizeof(float);
  const bool hu = flags & LR_DRR_HU_INPUT, flip = flags & LR_DRR_FLIP_W;
  const int64_t sD64 = (int64_t)W * H;
  // (HU input of a ONE-plane slab: the fast kernel's z-edge test `(unsigned)z0 > (unsigned)(Dn - 2)` would wrap — the general kernel)
  if (H >= 2 && (int64_t)(d1 - d0) * sD64 * 4 + sD64 * 8 <= 0x80000000LL && sD64 < (1 << 23) && (!hu || d1 - d0 >= 2) &&
      !lr_sw_set(LR_SW_DRR_GENERAL)) {
    const FastDiv fD = make_fastdiv(D), fW = make_fastdiv(W - 1), fH = make_fastdiv(H);
    const bool fd = fastdiv_ok(D) && fastdiv_ok(W - 1) && fastdiv_ok(H);
#define LR_FAST3(FLV, HUV, FDV)                                                                                          \
  hipLaunchKernelGGL((drr_forward_fast_kernel<FLV, HUV, FDV>), grid, block, lds, lr_stream(stream), vol_slab, lp, spacing[0], \
                     spacing[1], spacing[2], out, D, W, H, d0, d1, P, Rd, Rh, nseg, vol_batch_stride, fD, fW, fH)
#define LR_FAST(FLV, HUV) do { if (fd) LR_FAST3(FLV, HUV, true); else LR_FAST3(FLV, HUV, false); } while (0)
    if (hu && flip) LR_FAST(true, true);
    else if (hu) LR_FAST(false, true);
    else if (flip) LR_FAST(true, false);
    else LR_FAST(false, false);
#undef LR_FAST
#undef LR_FAST3
    return lr_launch_status();
  }
  // the general kernel: one volume per launch
  for (int b = 0; b < B; ++b) {
    const float* vb = vol_slab + (int64_t)b * vol_batch_stride;
    float* ob = out + (int64_t)b * P * Rd * Rh;
    const dim3 grid1((unsigned)nblk);
#define LR_LAUNCH(HUV, FLV)                                                                    \
  hipLaunchKernelGGL((drr_forward_kernel<HUV, FLV>), grid1, block, lds, lr_stream(stream),      \
                     vb, lp, spacing[0], spacing[1], spacing[2], ob, D, W, H, d0, d1, P, \
                     Rd, Rh, nseg)
    if (hu && flip) LR_LAUNCH(true, true);
    else if (hu) LR_LAUNCH(true, false);
    else if (flip) LR_LAUNCH(false, true);
    else LR_LAUNCH(false, false);
#undef LR_LAUNCH
    if (int e = lr_launch_status()) return e;
  }
  return LR_OK;
}

extern "C" int lr_drr_forward_f32(const float* vol_slab, const float* poses, const float* spacing,
                                  float* out, int D, int W, int H, int d0, int d1, int P, int Rd,
                                  int Rh, int flags, int nseg, void* stream) {
  return drr_forward_impl(vol_slab, 0, poses, spacing, out, 1, D, W, H, d0, d1, P, Rd, Rh, flags, nseg, stream);
}

extern "C" int lr_drr_forward_batch_f32(const float* vol_slabs, int64_t vol_batch_stride, const float* poses,
                                        const float* spacing, float* out, int B, int D, int W, int H, int d0, int d1,
                                        int P, int Rd, int Rh, int flags, int nseg, void* stream) {
  return drr_forward_impl(vol_slabs, vol_batch_stride, poses, spacing, out, B, D, W, H, d0, d1, P, Rd, Rh, flags, nseg, stream);
}

extern "C" int lr_hu_to_mu_f32(const float* hu, float* mu, int64_t n, void* stream) {
  if (!hu || !mu) return LR_ENULL;
  if (n < 0) return LR_EINVAL;
  if (n == 0) return LR_OK;
  int64_t nblk = (n + 255) / 256;
  if (nblk > 16384) nblk = 16384;
  hipLaunchKernelGGL(hu_to_mu_kernel, dim3((unsigned)nblk), dim3(256), 0, lr_stream(stream), hu, mu, n);
  return lr_launch_status();
}

extern "C" int lr_drr_sample_coords_f32(const float* poses, const float* spacing, float* pix,
                                        float* dx, int D, int W, int H, int P, int Rd, int Rh,
                                        int normalized, void* stream) {
  if (!spacing) return LR_ENULL;
  if (!pix && !dx) return LR_ENULL;
  if (D < 1 || W < 2 || H < 1 || Rd < 1 || Rh < 1) return LR_EINVAL;
  LrPoses lp;
  if (int e = fill_poses(lp, poses, P)) return e;
  const int64_t total = (int64_t)P * Rd * Rh;
  const int64_t nblk = (total + 255) / 256;
  hipLaunchKernelGGL(drr_coords_kernel, dim3((unsigned)nblk), dim3(256), 0, lr_stream(stream), lp,
                     spacing[0], spacing[1], spacing[2], pix, dx, D, W, H, P, Rd, Rh, normalized);
  return lr_launch_status();
}
