// backward.hip — gradients of the memory-bound ops for the training step (SURVEY §8 f2,
// reference RegistrationNet.py:389-406: loss.backward()).  The reference gets these from ATen autograd;
// each kernel states the formula it implements and is tested against torch autograd of the CPU oracle.
//
//   NCC          d loss / d warped                         (layers/losses.py:14-29 backward)
//   warp         d / d disp  (the moving image has no grad) (net_utils.py:26-52 + …Backproj.py:68 backward)
//   PCA          d / d coefs = g_disp · basis^T             (…Backproj.py:102 backward, second basis read)
//   Linear       d / d x, d / d W, d / d b with the LeakyReLU mask (layers/layers.py:432-438 backward)
#include "lr_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ NCC
// loss = 1 - (1/R) sum_r ncc_r ;  configured: ncc = (cov+e2)/s, s = sqrt((vx+e2)(vy+e2))
//   d ncc / d x_i = (1/n) [ yc_i / s - ncc * xc_i / (vx+e2) ]
// squared: v = cov^2/(vx*vy+1e-12):  d v / d x_i = (2/n) [ cov*yc_i*q - cov^2*vy*xc_i ] / q^2,  q = vx*vy+1e-12
__global__ __launch_bounds__(256) void ncc_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                      const double* __restrict__ moments,
                                                      const float* __restrict__ gout, float* __restrict__ gx,
                                                      int R, int64_t N, double n_total, int variant) {
  const int r = blockIdx.y;
  const double* m = moments + (int64_t)r * 5;
  const double n = n_total;
  const double mx = m[0] / n, my = m[1] / n;
  const double cov = m[2] / n - mx * my, vx = fmax(m[3] / n - mx * mx, 0.0), vy = fmax(m[4] / n - my * my, 0.0);  // clamped as in ncc_loss_kernel
  const double g = -(double)(*gout) / (double)R;  // d loss / d ncc_r
  double cy, cx;                                  // gx_i = cy * (y_i - my) + cx * (x_i - mx)
  if (variant == LR_NCC_CONFIGURED) {
    const double e2 = 1e-20, s = sqrt((vx + e2) * (vy + e2));
    const double ncc = (cov + e2) / s;
    cy = g / (n * s);
    cx = -g * ncc / (n * (vx + e2));
  } else {
    const double q = vx * vy + 1e-12;
    cy = g * 2.0 * cov / (n * q);
    cx = -g * 2.0 * cov * cov * vy / (n * q * q);
  }
  const float fcy = (float)cy, fcx = (float)cx, fmx = (float)mx, fmy = (float)my;
  const float* xr = x + (int64_t)r * N;
  const float* yr = y + (int64_t)r * N;
  float* gr = gx + (int64_t)r * N;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride)
    gr[i] = fmaf(fcy, yr[i] - fmy, fcx * (xr[i] - fmx));
}

// d loss / d moments (R,5): the similarity as a function of the five sums (sum x, sum y, sum xy, sum x^2, sum y^2) that the
// decode node produced.  Chain rule to the voxels:  d loss / d x_i = gm0 + gm2 * y_i + 2 gm3 * x_i  — affine in (x_i, y_i),
// so the warp-gradient kernel forms it on the fly (warp_bwd_fast_kernel<…, NCC>) and ncc_bwd_kernel's pass is not needed.
__global__ __launch_bounds__(64) void ncc_bwd_moments_kernel(const double* __restrict__ moments, const float* __restrict__ gout,
                                                             double* __restrict__ gm, int R, double n_total, int variant) {
  const int r = blockIdx.x * 64 + threadIdx.x;
  if (r >= R) return;
  const double* m = moments + (int64_t)r * 5;
  const double n = n_total;
  const double mx = m[0] / n, my = m[1] / n;
  const double cov = m[2] / n - mx * my, vx = fmax(m[3] / n - mx * mx, 0.0), vy = fmax(m[4] / n - my * my, 0.0);
  const double g = -(double)(*gout) / (double)R;
  double dcov, dvx, dvy;   // d ncc_r / d cov, vx, vy
  if (variant == LR_NCC_CONFIGURED) {
    const double e2 = 1e-20, s = sqrt((vx + e2) * (vy + e2));
    const double ncc = (cov + e2) / s;
    dcov = 1.0 / s;
    dvx = -ncc / (2.0 * (vx + e2));
    dvy = -ncc / (2.0 * (vy + e2));
  } else {
    const double q = vx * vy + 1e-12;
    dcov = 2.0 * cov / q;
    dvx = -cov * cov * vy / (q * q);
    dvy = -cov * cov * vx / (q * q);
  }
  double* o = gm + (int64_t)r * 5;
  o[2] = g * dcov / n;
  o[3] = g * dvx / n;
  o[4] = g * dvy / n;
  o[0] = -o[2] * my - 2.0 * o[3] * mx;
  o[1] = -o[2] * mx - 2.0 * o[4] * my;
}

// ------------------------------------------------------------------------------------------------ warp
struct AxisB {
  int i0, i1;
  float w0, w1;
  bool ok0, ok1;
  float gmul;  // d pix / d phi (0 when 'border' clipping is active)
};

template <bool BORDER>
__device__ __forceinline__ AxisB axis_b(float g, int size) {
  float pix = ((g + 1.0f) * 0.5f) * (float)(size - 1);
  AxisB a;
  a.gmul = 0.5f * (float)(size - 1);
  if constexpr (BORDER) {  // clip_coordinates_set_grad: zero gradient outside [0, size-1]
    if (!(pix > 0.0f && pix < (float)(size - 1))) a.gmul = 0.0f;
    pix = fminf((float)(size - 1), fmaxf(pix, 0.0f));
  }
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

// g_disp[c_axis] = sum_channels g_warped * (scale ? 2 : 1) * dS/dpix_axis * (size_axis-1)/2,
// dS/dx = sum_{cy,cz} wy*wz*(val[x1]-val[x0]) (out-of-range corners are 0), likewise y, z.
template <bool SCALE, bool BORDER, bool SEG>
__global__ __launch_bounds__(256) void warp_bwd_kernel(const float* __restrict__ img, const float* __restrict__ seg,
                                                       const float* __restrict__ disp, const float* __restrict__ id0,
                                                       const float* __restrict__ id1, const float* __restrict__ id2,
                                                       const float* __restrict__ gw, float* __restrict__ gdisp,
                                                       const float* __restrict__ gadd, int B, int C, int D, int W, int H,
                                                       int Dn) {
  const int64_t per_b = (int64_t)Dn * W * H;
  const int b = blockIdx.y;
  const int64_t idx = (int64_t)lr_xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
  if (idx >= per_b) return;
  const int k = (int)(idx % H), j = (int)((idx / H) % W), i = (int)(idx / H / W);
  const int64_t V = (int64_t)D * W * H, sD = (int64_t)W * H;
  const float* dp = disp + (int64_t)b * 3 * per_b + idx;
  float p0 = dp[0], p1 = dp[per_b], p2 = dp[2 * per_b];
  if (id0) { p0 = p0 + id0[i]; p1 = p1 + id1[j]; p2 = p2 + id2[k]; }
  const AxisB ax = axis_b<BORDER>(p2, H), ay = axis_b<BORDER>(p1, W), az = axis_b<BORDER>(p0, D);
  float gx = 0.0f, gy = 0.0f, gz = 0.0f;
  for (int c = 0; c < C; ++c) {
    const float* im = img + ((int64_t)b * C + c) * V;
    const float* sg = SEG ? seg + ((int64_t)b * C + c) * V : nullptr;
    float v[2][2][2];
#pragma unroll
    for (int cz = 0; cz < 2; ++cz)
#pragma unroll
      for (int cy = 0; cy < 2; ++cy)
#pragma unroll
        for (int cx = 0; cx < 2; ++cx) {
          const bool ok = (cz ? az.ok1 : az.ok0) && (cy ? ay.ok1 : ay.ok0) && (cx ? ax.ok1 : ax.ok0);
          const int64_t off = (int64_t)(cz ? az.i1 : az.i0) * sD + (int64_t)(cy ? ay.i1 : ay.i0) * H + (cx ? ax.i1 : ax.i0);
          float t = im[off];
          if constexpr (SEG) t = (t + 1.0f) * sg[off] - 1.0f;
          if constexpr (SCALE) t = (t + 1.0f) * 0.5f;
          v[cz][cy][cx] = ok ? t : 0.0f;
        }
    const float wz[2] = {az.w0, az.w1}, wy[2] = {ay.w0, ay.w1}, wx[2] = {ax.w0, ax.w1};
    float dx = 0.0f, dy = 0.0f, dz = 0.0f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        dx += wy[a] * wz[q] * (v[q][a][1] - v[q][a][0]);
        dy += wx[a] * wz[q] * (v[q][1][a] - v[q][0][a]);
        dz += wx[a] * wy[q] * (v[1][q][a] - v[0][q][a]);
      }
    const float g = gw[((int64_t)b * C + c) * per_b + idx] * (SCALE ? 2.0f : 1.0f);
    gx += g * dx;
    gy += g * dy;
    gz += g * dz;
  }
  float* go = gdisp + (int64_t)b * 3 * per_b + idx;
  float r0 = gz * az.gmul, r1 = gy * ay.gmul, r2 = gx * ax.gmul;  // disp channels 0,1,2 <-> D,W,H
  if (gadd) {  // + the gradient that reaches the displacement field by another path (e.g. the regulariser)
    const float* ga = gadd + (int64_t)b * 3 * per_b + idx;
    r0 = r0 + ga[0];
    r1 = r1 + ga[per_b];
    r2 = r2 + ga[2 * per_b];
  }
  go[0] = r0;
  go[per_b] = r1;
  go[2 * per_b] = r2;
}

// The training step's case (zeros padding, no mask, float4 rows) with a third of the vector-ALU work — the same
// restructuring as warp.hip's warp_tri_fast_kernel: 4 voxels per thread, taps through bounds-checked raw buffer
// loads with 32-bit offsets from 24-bit multiplies (no clamping: an out-of-range tap is masked, wherever its address
// lands), one 8-byte load per x pair (re-based under a wave-uniform branch at the x faces), in-plane index from a
// float reciprocal.  (v+1)/2 and the factor 2 on the incoming gradient cancel exactly (powers of two), so both are
// dropped; every other product and sum is written in the general kernel's order — the results are the same bits.
struct AxisBF {
  float w0, w1;  // interpolation weights (0 when the axis is out of range)
  int i0;        // floor(pix), 0 when the axis is out of range
  bool ok0, ok1;
};

__device__ __forceinline__ AxisBF axis_bf(float g, int size) {
  const float pix = ((g + 1.0f) * 0.5f) * (float)(size - 1);
  const bool valid = pix > -1.0f && pix < (float)size;
  const float fl = floorf(pix);
  AxisBF a;
  a.i0 = valid ? (int)fl : 0;
  a.w0 = valid ? (fl + 1.0f) - pix : 0.0f;
  a.w1 = valid ? pix - fl : 0.0f;
  a.ok0 = valid && a.i0 >= 0;
  a.ok1 = valid && a.i0 + 1 < size;
  return a;
}

// NCC: the gradient of `warped` is not read but formed from the similarity's moment gradient (ncc_bwd_moments_kernel):
// gw_i = gm0 + gm2 * target_i + 2 gm3 * warped_i (single-channel images; gw = warped, ncc_y = target, gm = (B,5) fp64)
template <bool SCALE, bool NCC = false>
__global__ __launch_bounds__(256) void warp_bwd_fast_kernel(const float* __restrict__ img, const float* __restrict__ disp,
                                                            const float* __restrict__ id0, const float* __restrict__ id1,
                                                            const float* __restrict__ id2, const float* __restrict__ gw,
                                                            float* __restrict__ gdisp, const float* __restrict__ gadd,
                                                            int C, int D, int W, int H, int Dn, float rcp_hv,
                                                            const float* __restrict__ ncc_y = nullptr,
                                                            const double* __restrict__ gm = nullptr) {
  const int HV = H >> 2;
  const int t = blockIdx.x * 256 + threadIdx.x;  // float4 index inside plane i
  const int i = blockIdx.y, b = blockIdx.z;
  const int j = (int)(((float)t + 0.5f) * rcp_hv);
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
  float p0[4] = {da.x, da.y, da.z, da.w}, p1[4] = {db.x, db.y, db.z, db.w}, p2[4] = {dc.x, dc.y, dc.z, dc.w};
  if (id0) {
    const float a0 = id0[i], a1 = id1[j];
    const float4 a2 = *reinterpret_cast<const float4*>(id2 + (kv << 2));
    const float a2v[4] = {a2.x, a2.y, a2.z, a2.w};
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      p0[v] = p0[v] + a0;
      p1[v] = p1[v] + a1;
      p2[v] = p2[v] + a2v[v];
    }
  }
  float gx[4] = {0.f, 0.f, 0.f, 0.f}, gy[4] = {0.f, 0.f, 0.f, 0.f}, gz[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < C; ++c) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(img + ((int64_t)b * C + c) * V), (short)0, (int)(V * 4), 0x00020000);
    const float4 g4 = *reinterpret_cast<const float4*>(gw + ((int64_t)b * C + c) * slabV + (int64_t)i * sD + inplane);
    float gv[4] = {g4.x, g4.y, g4.z, g4.w};
    if constexpr (NCC) {
      const float4 y4 = *reinterpret_cast<const float4*>(ncc_y + ((int64_t)b * C + c) * slabV + (int64_t)i * sD + inplane);
      const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
      const float c0 = (float)gm[b * 5 + 0], cy = (float)gm[b * 5 + 2], cx = (float)(2.0 * gm[b * 5 + 3]);
#pragma unroll
      for (int v = 0; v < 4; ++v) gv[v] = fmaf(cy, yv[v], fmaf(cx, gv[v], c0));
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const AxisBF ax = axis_bf(p2[v], H), ay = axis_bf(p1[v], W), az = axis_bf(p0[v], D);
      const int xb = min(max(ax.i0, 0), H - 2), shift = ax.i0 - xb;
      const int y0 = __mul24(ay.i0, H), z0 = __mul24(az.i0, sD);
      const unsigned xb4 = (unsigned)xb << 2;
      unsigned off[2][2];
      off[0][0] = ((unsigned)(z0 + y0) << 2) + xb4;
      off[0][1] = ((unsigned)(z0 + y0 + H) << 2) + xb4;
      off[1][0] = ((unsigned)(z0 + sD + y0) << 2) + xb4;
      off[1][1] = ((unsigned)(z0 + sD + y0 + H) << 2) + xb4;
      float val[2][2][2];
#pragma unroll
      for (int cz = 0; cz < 2; ++cz)
#pragma unroll
        for (int cy = 0; cy < 2; ++cy) {
          const uint2 q = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off[cz][cy], 0, 0));
          val[cz][cy][0] = __builtin_bit_cast(float, q.x);
          val[cz][cy][1] = __builtin_bit_cast(float, q.y);
        }
      if (__builtin_amdgcn_ballot_w64(shift != 0) != 0) {  // x0 = -1: tap 1 is the pair's first; x0 = H-1: tap 0 its second
#pragma unroll
        for (int cz = 0; cz < 2; ++cz)
#pragma unroll
          for (int cy = 0; cy < 2; ++cy) {
            const float px = val[cz][cy][0], py = val[cz][cy][1];
            val[cz][cy][0] = shift > 0 ? py : px;
            val[cz][cy][1] = shift < 0 ? px : py;
          }
      }
#pragma unroll
      for (int cz = 0; cz < 2; ++cz)
#pragma unroll
        for (int cy = 0; cy < 2; ++cy)
#pragma unroll
          for (int cx = 0; cx < 2; ++cx) {
            const bool ok = (cz ? az.ok1 : az.ok0) && (cy ? ay.ok1 : ay.ok0) && (cx ? ax.ok1 : ax.ok0);
            const float tv = SCALE ? val[cz][cy][cx] + 1.0f : val[cz][cy][cx];
            val[cz][cy][cx] = ok ? tv : 0.0f;
          }
      const float wz[2] = {az.w0, az.w1}, wy[2] = {ay.w0, ay.w1}, wx[2] = {ax.w0, ax.w1};
      float dx = 0.0f, dy = 0.0f, dz = 0.0f;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          dx += wy[a] * wz[q] * (val[q][a][1] - val[q][a][0]);
          dy += wx[a] * wz[q] * (val[q][1][a] - val[q][0][a]);
          dz += wx[a] * wy[q] * (val[1][q][a] - val[0][q][a]);
        }
      gx[v] += gv[v] * dx;
      gy[v] += gv[v] * dy;
      gz[v] += gv[v] * dz;
    }
  }
  const float mz = 0.5f * (float)(D - 1), my = 0.5f * (float)(W - 1), mx = 0.5f * (float)(H - 1);
  float* go = gdisp + ubase + inplane;
  float4 r0 = make_float4(gz[0] * mz, gz[1] * mz, gz[2] * mz, gz[3] * mz);  // channel 0 <-> D
  float4 r1 = make_float4(gy[0] * my, gy[1] * my, gy[2] * my, gy[3] * my);  // 1 <-> W
  float4 r2 = make_float4(gx[0] * mx, gx[1] * mx, gx[2] * mx, gx[3] * mx);  // 2 <-> H
  if (gadd) {  // + the gradient that reaches the displacement field by another path (e.g. the regulariser)
    const float* ga = gadd + ubase + inplane;
    const float4 a0 = *reinterpret_cast<const float4*>(ga), a1 = *reinterpret_cast<const float4*>(ga + slabV),
                 a2 = *reinterpret_cast<const float4*>(ga + 2 * slabV);
    r0 = make_float4(r0.x + a0.x, r0.y + a0.y, r0.z + a0.z, r0.w + a0.w);
    r1 = make_float4(r1.x + a1.x, r1.y + a1.y, r1.z + a1.z, r1.w + a1.w);
    r2 = make_float4(r2.x + a2.x, r2.y + a2.y, r2.z + a2.z, r2.w + a2.w);
  }
  *reinterpret_cast<float4*>(go) = r0;
  *reinterpret_cast<float4*>(go + slabV) = r1;
  *reinterpret_cast<float4*>(go + 2 * slabV) = r2;
}

// ------------------------------------------------------------------------------------------------ PCA
// gcoefs[b][l] = sum_m g[b][m] * basis[l][m].  grid (m-blocks, l-groups of LG): a block keeps LG x BT
// accumulators per thread, streams its m-range once per l-group; per-block partials, then a fixed-order reduce.
constexpr int LG = 8;
template <int BT, bool BF /* bf16-stored basis */>
__global__ __launch_bounds__(256) void pca_bwd_kernel(const float* __restrict__ g, const float* __restrict__ basis,
                                                      float* __restrict__ partial, int B, int L, int64_t M,
                                                      int64_t ldb, int64_t gstride, int nblk) {
  // 1-D launch order: the l-groups of one m-range are issued together and land on the SAME XCD (ids 8 apart), so the
  // gradient rows they all read come from HBM once and from that XCD's L2 afterwards (they walk m in near lockstep).
  // Batches above BT rows (round 6; the reference's shipped batch is 30): ceil(B / BT) row chunks per (m-range, l-group), all of
  // them in that same run of ids — the chunks of an l-group stream the same basis rows at the same time, so the basis leaves HBM
  // once per launch, not once per chunk (four launches of 0.52 ms at 160^3 / B = 30 before).  Each (chunk, l-group) block is the
  // block of a stand-alone launch on that chunk: same sums, same bits.
  const int Btot = B;
  const unsigned nch = (unsigned)((B + BT - 1) / BT);
  const unsigned ng = (unsigned)((L + LG - 1) / LG), per8 = ng * nch * 8u;
  const unsigned q = blockIdx.x / per8, r = blockIdx.x - q * per8;
  const unsigned bx = q * 8u + (r & 7u);
  if (bx >= (unsigned)nblk) return;
  const unsigned cg = r >> 3;                       // (l-group, chunk): the chunks of an l-group next to each other
  const int l0 = (int)(cg / nch) * LG, b0 = (int)(cg % nch) * BT;
  g += (int64_t)b0 * gstride;
  B = B - b0 < BT ? B - b0 : BT;
  float acc[LG][BT];
#pragma unroll
  for (int a = 0; a < LG; ++a)
#pragma unroll
    for (int b = 0; b < BT; ++b) acc[a][b] = 0.0f;
  const int64_t step = (int64_t)nblk * 256 * 4;
  // Every load of an iteration is unconditional and issued before the first multiply: rows past B / past L read the last valid row
  // instead (their sums are dropped at the write below).  (With the `if`s around single loads hipcc waited for each basis row
  // before asking for the next — one 16-byte load in flight per lane: 2.61 ms at C3.)
  const float* grow[BT];
  const float* brow[LG];
#pragma unroll
  for (int b = 0; b < BT; ++b) grow[b] = g + (int64_t)(b < B ? b : B - 1) * gstride;
#pragma unroll
  for (int a = 0; a < LG; ++a) {
    const int l = l0 + a < L ? l0 + a : L - 1;
    brow[a] = BF ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(basis) + (int64_t)l * ldb) : basis + (int64_t)l * ldb;
  }
  for (int64_t m = ((int64_t)bx * 256 + threadIdx.x) * 4; m < M; m += step) {
    f32x4 gv[BT], bv[LG];
#pragma unroll
    for (int a = 0; a < LG; ++a) {
      if (BF) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2* bp2 = reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(brow[a]) + m);
        const u32x2 raw = nch > 1 ? *bp2 : __builtin_nontemporal_load(bp2);
        bv[a][0] = __builtin_bit_cast(float, raw.x << 16);
        bv[a][1] = __builtin_bit_cast(float, raw.x & 0xffff0000u);
        bv[a][2] = __builtin_bit_cast(float, raw.y << 16);
        bv[a][3] = __builtin_bit_cast(float, raw.y & 0xffff0000u);
      } else {
        // (several row chunks: plain loads — the other chunks of this l-group find the rows in the XCD's L2; non-temporal ones do not stay)
        bv[a] = nch > 1 ? *reinterpret_cast<const f32x4*>(brow[a] + m) : __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(brow[a] + m));
      }
    }
#pragma unroll
    for (int b = 0; b < BT; ++b) gv[b] = *reinterpret_cast<const f32x4*>(grow[b] + m);
#pragma unroll
    for (int a = 0; a < LG; ++a)
#pragma unroll
      for (int b = 0; b < BT; ++b)
        acc[a][b] = fmaf(gv[b].x, bv[a].x, fmaf(gv[b].y, bv[a].y, fmaf(gv[b].z, bv[a].z, fmaf(gv[b].w, bv[a].w, acc[a][b]))));
  }
  __shared__ float red[4][LG * BT];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int a = 0; a < LG; ++a)
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      const float s = lr_wave_sum(acc[a][b]);
      if (lane == 0) red[wave][a * BT + b] = s;
    }
  __syncthreads();
  if (threadIdx.x < LG * BT) {
    const int a = threadIdx.x / BT, b = threadIdx.x % BT;
    if (l0 + a < L && b < B)
      partial[((int64_t)bx * Btot + b0 + b) * L + l0 + a] =
          (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  }
}

// any M / leading dimension / alignment: one element per thread-iteration, same partial layout
template <bool BF>
__global__ __launch_bounds__(256) void pca_bwd_scalar_kernel(const float* __restrict__ g, const float* __restrict__ basis,
                                                             float* __restrict__ partial, int B, int L, int64_t M,
                                                             int64_t ldb, int64_t gstride) {
  const int l0 = blockIdx.y * LG;
  float acc[LG][8];
#pragma unroll
  for (int a = 0; a < LG; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = 0.0f;
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
    float gv[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) gv[b] = b < B ? g[(int64_t)b * gstride + m] : 0.0f;
#pragma unroll
    for (int a = 0; a < LG; ++a) {
      if (l0 + a < L) {
        float bv;
        if (BF) bv = __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(basis)[(int64_t)(l0 + a) * ldb + m] << 16);
        else bv = basis[(int64_t)(l0 + a) * ldb + m];
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = fmaf(gv[b], bv, acc[a][b]);
      }
    }
  }
  __shared__ float red[4][LG * 8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int a = 0; a < LG; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const float s = lr_wave_sum(acc[a][b]);
      if (lane == 0) red[wave][a * 8 + b] = s;
    }
  __syncthreads();
  if (threadIdx.x < LG * 8) {
    const int a = threadIdx.x / 8, b = threadIdx.x % 8;
    if (l0 + a < L && b < B)
      partial[((int64_t)blockIdx.x * B + b) * L + l0 + a] =
          (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  }
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                           int nblk, int n) {
  // out[i] = sum_blk partial[blk][i], fixed order, fp64 accumulate.  ONE WAVE per output: lane j adds blocks j, j+64, …, then a
  // butterfly over the lanes.  (One thread per output walking all nblk partials was a chain of nblk dependent loads:
  // 0.39 ms for 256 x 448 numbers behind every pca_bwd_kernel launch — 1.3 % of the C3 training step, by rocprofv3.)
  const int lane = threadIdx.x & 63;
  const int i = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (i >= n) return;
  double s = 0.0;
  for (int k = lane; k < nblk; k += 64) s += (double)partial[(int64_t)k * n + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) out[i] = (float)s;
}

// ------------------------------------------------------------------------------------------------ Linear
__device__ __forceinline__ float lrelu_grad(float gy, float y, float slope) { return y > 0.0f ? gy : gy * slope; }

// gw[o][k] = sum_b gpre[b][o] x[b][k] ; gb[o] = sum_b gpre[b][o]
__global__ __launch_bounds__(256) void linear_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ gy, float* __restrict__ gw,
                                                           float* __restrict__ gb, int B, int K, int O, float slope) {
  const int o = blockIdx.y;
  __shared__ float gp[32];
  if (threadIdx.x < 32)
    gp[threadIdx.x] = threadIdx.x < B ? lrelu_grad(gy[(int64_t)threadIdx.x * O + o], y[(int64_t)threadIdx.x * O + o], slope) : 0.0f;
  __syncthreads();
  if (gb && blockIdx.x == 0 && threadIdx.x == 0) {
    float s = 0.0f;
    for (int b = 0; b < B; ++b) s += gp[b];
    gb[o] = s;
  }
  for (int k = blockIdx.x * 256 + threadIdx.x; k < K; k += gridDim.x * 256) {
    float s = 0.0f;
    for (int b = 0; b < B; ++b) s = fmaf(gp[b], x[(int64_t)b * K + k], s);
    gw[(int64_t)o * K + k] = s;
  }
}

// gx[b][k] = sum_o gpre[b][o] w[o][k]
template <int BT>
__global__ __launch_bounds__(256) void linear_bwd_x_kernel(const float* __restrict__ w, const float* __restrict__ y,
                                                           const float* __restrict__ gy, float* __restrict__ gx,
                                                           int B, int b_lo, int K, int O, float slope) {
  extern __shared__ float gpre[];  // [O][BT]
  b_lo += (int)blockIdx.y * BT;   // the row chunks of a batch above BT rows: blockIdx.y of one launch
  for (int t = threadIdx.x; t < O * BT; t += 256) {
    const int o = t / BT, b = b_lo + t % BT;
    gpre[t] = b < B ? lrelu_grad(gy[(int64_t)b * O + o], y[(int64_t)b * O + o], slope) : 0.0f;
  }
  __syncthreads();
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  float acc[BT];
#pragma unroll
  for (int b = 0; b < BT; ++b) acc[b] = 0.0f;
  // sixteen rows of the weight in flight per thread (the same fmaf chain): one row at a time was O dependent-latency
  // round trips — 0.14 ms for FC1's 52 MB, 0.4 TB/s
  int o = 0;
  for (; o + 15 < O; o += 16) {
    float wv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) wv[u] = __builtin_nontemporal_load(w + (int64_t)(o + u) * K + k);
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int b = 0; b < BT; ++b) acc[b] = fmaf(gpre[(o + u) * BT + b], wv[u], acc[b]);
  }
  for (; o < O; ++o) {
    const float wv = w[(int64_t)o * K + k];
#pragma unroll
    for (int b = 0; b < BT; ++b) acc[b] = fmaf(gpre[o * BT + b], wv, acc[b]);
  }
#pragma unroll
  for (int b = 0; b < BT; ++b)
    if (b_lo + b < B) gx[(int64_t)(b_lo + b) * K + k] = acc[b];
}

}  // namespace

extern "C" int lr_ncc_bwd_f32(const float* x, const float* y, const double* moments, const float* gout, float* gx,
                              int R, int64_t N, int64_t n_total, int variant, void* stream) {
  if (!x || !y || !moments || !gout || !gx) return LR_ENULL;
  if (R < 1 || R > 65535 || N < 1 || n_total < N) return LR_EINVAL;
  if (variant != LR_NCC_CONFIGURED && variant != LR_NCC_SQUARED) return LR_EINVAL;
  int64_t nblk = (N + 255) / 256;
  if (nblk > 4096) nblk = 4096;
  hipLaunchKernelGGL(ncc_bwd_kernel, dim3((unsigned)nblk, (unsigned)R), dim3(256), 0, lr_stream(stream), x, y,
                     moments, gout, gx, R, N, (double)n_total, variant);
  return lr_launch_status();
}

static int warp_bwd_impl(const float* img, const float* seg, const float* disp, const float* id0,
                         const float* id1, const float* id2, const float* gwarped, float* gdisp, const float* gadd, int B,
                         int C, int D, int W, int H, int d0, int d1, int flags, void* stream) {
  if (!img || !disp || !gwarped || !gdisp) return LR_ENULL;
  if (B < 1 || B > 65535 || C < 1 || D < 1 || W < 1 || H < 1 || d0 < 0 || d1 > D || d1 <= d0) return LR_EINVAL;
  if (flags & ~(LR_WARP_USING_SCALE | LR_WARP_BORDER)) return LR_EUNSUPPORTED;  // nearest mode has no gradient
  const bool any_id = id0 || id1 || id2, all_id = id0 && id1 && id2;
  if (any_id && !all_id) return LR_ENULL;
  const int Dn = d1 - d0;
  hipStream_t st = lr_stream(stream);
  const bool sc = flags & LR_WARP_USING_SCALE, bo = flags & LR_WARP_BORDER;
  {
    const int64_t sD = (int64_t)W * H, V = sD * D;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    if (!seg && !bo && H % 4 == 0 && al16(disp) && al16(gwarped) && al16(gdisp) && (!gadd || al16(gadd)) && (!id2 || al16(id2)) &&
        V * 4 + sD * 8 <= 0x80000000LL && sD < (1 << 23) && sD / 4 <= (1 << 20) && Dn <= 65535 &&
        !lr_sw_set(LR_SW_WARP_GENERAL)) {
      const dim3 g3((unsigned)((sD / 4 + 255) / 256), (unsigned)Dn, (unsigned)B);
      const float rcp_hv = 1.0f / (float)(H / 4);
      if (sc) hipLaunchKernelGGL(warp_bwd_fast_kernel<true>, g3, dim3(256), 0, st, img, disp, id0, id1, id2, gwarped, gdisp, gadd, C, D, W, H, Dn, rcp_hv);
      else hipLaunchKernelGGL(warp_bwd_fast_kernel<false>, g3, dim3(256), 0, st, img, disp, id0, id1, id2, gwarped, gdisp, gadd, C, D, W, H, Dn, rcp_hv);
      return lr_launch_status();
    }
  }
  const int64_t nblk = ((int64_t)Dn * W * H + 255) / 256;
  if (nblk > 0x7fffffffLL) return LR_EINVAL;
  const dim3 grid((unsigned)nblk, (unsigned)B), block(256);
#define LR_WB(S, Bo, G) hipLaunchKernelGGL((warp_bwd_kernel<S, Bo, G>), grid, block, 0, st, img, seg, disp, id0, id1, id2, gwarped, gdisp, gadd, B, C, D, W, H, Dn)
  if (seg) {
    if (sc) { if (bo) LR_WB(true, true, true); else LR_WB(true, false, true); }
    else    { if (bo) LR_WB(false, true, true); else LR_WB(false, false, true); }
  } else {
    if (sc) { if (bo) LR_WB(true, true, false); else LR_WB(true, false, false); }
    else    { if (bo) LR_WB(false, true, false); else LR_WB(false, false, false); }
  }
#undef LR_WB
  return lr_launch_status();
}

extern "C" int lr_warp_bwd_disp_f32(const float* img, const float* seg, const float* disp, const float* id0,
                                    const float* id1, const float* id2, const float* gwarped, float* gdisp, int B,
                                    int C, int D, int W, int H, int d0, int d1, int flags, void* stream) {
  return warp_bwd_impl(img, seg, disp, id0, id1, id2, gwarped, gdisp, nullptr, B, C, D, W, H, d0, d1, flags, stream);
}

extern "C" int lr_ncc_bwd_moments(const double* moments, const float* gout, double* gmoments, int R, int64_t n_total,
                                  int variant, void* stream) {
  if (!moments || !gout || !gmoments) return LR_ENULL;
  if (R < 1 || n_total < 1) return LR_EINVAL;
  if (variant != LR_NCC_CONFIGURED && variant != LR_NCC_SQUARED) return LR_EUNSUPPORTED;
  hipLaunchKernelGGL(ncc_bwd_moments_kernel, dim3((unsigned)((R + 63) / 64)), dim3(64), 0, lr_stream(stream), moments, gout, gmoments,
                     R, (double)n_total, variant);
  return lr_launch_status();
}

extern "C" int lr_warp_bwd_disp_ncc_f32(const float* img, const float* disp, const float* id0, const float* id1,
                                        const float* id2, const float* warped, const float* target, const double* gmoments,
                                        const float* gadd, float* gdisp, int B, int D, int W, int H, int d0, int d1,
                                        int flags, void* stream) {
  if (!img || !disp || !warped || !target || !gmoments || !gdisp) return LR_ENULL;
  if (B < 1 || B > 65535 || D < 1 || W < 1 || H < 1 || d0 < 0 || d1 > D || d1 <= d0) return LR_EINVAL;
  if (flags & ~LR_WARP_USING_SCALE) return LR_EUNSUPPORTED;   // zeros padding only (the model's case)
  const bool any_id = id0 || id1 || id2, all_id = id0 && id1 && id2;
  if (any_id && !all_id) return LR_ENULL;
  const int Dn = d1 - d0;
  const int64_t sD = (int64_t)W * H, V = sD * D;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  if (!(H % 4 == 0 && al16(disp) && al16(warped) && al16(target) && al16(gdisp) && (!gadd || al16(gadd)) && (!id2 || al16(id2)) &&
        V * 4 + sD * 8 <= 0x80000000LL && sD < (1 << 23) && sD / 4 <= (1 << 20) && Dn <= 65535))
    return LR_EUNSUPPORTED;   // the caller materialises the gradient (lr_ncc_bwd_f32) and uses lr_warp_bwd_disp_f32
  const dim3 g3((unsigned)((sD / 4 + 255) / 256), (unsigned)Dn, (unsigned)B);
  const float rcp_hv = 1.0f / (float)(H / 4);
  hipStream_t st = lr_stream(stream);
  if (flags & LR_WARP_USING_SCALE)
    hipLaunchKernelGGL((warp_bwd_fast_kernel<true, true>), g3, dim3(256), 0, st, img, disp, id0, id1, id2, warped, gdisp, gadd, 1, D, W, H, Dn, rcp_hv, target, gmoments);
  else
    hipLaunchKernelGGL((warp_bwd_fast_kernel<false, true>), g3, dim3(256), 0, st, img, disp, id0, id1, id2, warped, gdisp, gadd, 1, D, W, H, Dn, rcp_hv, target, gmoments);
  return lr_launch_status();
}

// Same, plus `gadd` (B,3,Dn,W,H): gdisp = d warped / d disp · gwarped + gadd — the sum autograd would otherwise form
// in a pass of its own when the displacement field also feeds the regulariser (gdisp and gadd are distinct buffers).
extern "C" int lr_warp_bwd_disp_acc_f32(const float* img, const float* seg, const float* disp, const float* id0,
                                        const float* id1, const float* id2, const float* gwarped, const float* gadd,
                                        float* gdisp, int B, int C, int D, int W, int H, int d0, int d1, int flags,
                                        void* stream) {
  if (!gadd) return LR_ENULL;
  return warp_bwd_impl(img, seg, disp, id0, id1, id2, gwarped, gdisp, gadd, B, C, D, W, H, d0, d1, flags, stream);
}

static int pca_bwd_impl(bool bf, const float* gdisp, const float* basis, float* partial, float* gcoefs, int B,
                                   int L, int64_t M, int64_t ldb, int64_t gdisp_batch_stride, int nblk,
                                   void* stream) {
  if (!gdisp || !basis || !partial || !gcoefs) return LR_ENULL;
  if (B < 1 || B > 64 || L < 1 || M < 1 || ldb < M || gdisp_batch_stride < M || nblk < 1 || nblk > 65535)
    return B > 64 ? LR_EUNSUPPORTED : LR_EINVAL;
  hipStream_t st = lr_stream(stream);
  const dim3 grid((unsigned)nblk, (unsigned)((L + LG - 1) / LG));
  const bool vec_ok = !((M & 3) || (ldb & 3) || (gdisp_batch_stride & 3)) && !(reinterpret_cast<uintptr_t>(gdisp) & 15u) &&
                      !(reinterpret_cast<uintptr_t>(basis) & (bf ? 7u : 15u));
  if (!vec_ok) {  // odd voxel counts (3·D·W·H not a multiple of 4), sliced views
    if (B > 8) return LR_EUNSUPPORTED;   // (the scalar kernel keeps 8 rows: the caller goes in chunks of 8)
    if (bf) hipLaunchKernelGGL(pca_bwd_scalar_kernel<true>, grid, dim3(256), 0, st, gdisp, basis, partial, B, L, M, ldb, gdisp_batch_stride);
    else hipLaunchKernelGGL(pca_bwd_scalar_kernel<false>, grid, dim3(256), 0, st, gdisp, basis, partial, B, L, M, ldb, gdisp_batch_stride);
  } else {
    const unsigned nchunk = B > 8 ? (unsigned)((B + 7) / 8) : 1u;   // (B <= 4: one chunk of the 4-row kernel)
    const dim3 g1((unsigned)((nblk + 7) / 8) * 8u * (unsigned)((L + LG - 1) / LG) * nchunk);
    if (B > 4 && bf) hipLaunchKernelGGL((pca_bwd_kernel<8, true>), g1, dim3(256), 0, st, gdisp, basis, partial, B, L, M, ldb, gdisp_batch_stride, nblk);
    else if (B > 4) hipLaunchKernelGGL((pca_bwd_kernel<8, false>), g1, dim3(256), 0, st, gdisp, basis, partial, B, L, M, ldb, gdisp_batch_stride, nblk);
    else if (bf) hipLaunchKernelGGL((pca_bwd_kernel<4, true>), g1, dim3(256), 0, st, gdisp, basis, partial, B, L, M, ldb, gdisp_batch_stride, nblk);
    else hipLaunchKernelGGL((pca_bwd_kernel<4, false>), g1, dim3(256), 0, st, gdisp, basis, partial, B, L, M, ldb, gdisp_batch_stride, nblk);
  }
  if (int e = lr_launch_status()) return e;
  const int n = B * L;
  hipLaunchKernelGGL(sum_partials_kernel, dim3((n + 3) / 4), dim3(256), 0, st, partial, gcoefs, nblk, n);   // one wave per output
  return lr_launch_status();
}

extern "C" int lr_pca_bwd_coef_f32(const float* gdisp, const float* basis, float* partial, float* gcoefs, int B,
                                   int L, int64_t M, int64_t ldb, int64_t gdisp_batch_stride, int nblk,
                                   void* stream) {
  return pca_bwd_impl(false, gdisp, basis, partial, gcoefs, B, L, M, ldb, gdisp_batch_stride, nblk, stream);
}

extern "C" int lr_pca_bwd_coef_bf16basis_f32(const float* gdisp, const void* basis_bf16, float* partial,
                                             float* gcoefs, int B, int L, int64_t M, int64_t ldb,
                                             int64_t gdisp_batch_stride, int nblk, void* stream) {
  return pca_bwd_impl(true, gdisp, reinterpret_cast<const float*>(basis_bf16), partial, gcoefs, B, L, M, ldb,
                      gdisp_batch_stride, nblk, stream);
}

extern "C" int lr_linear_bwd_f32(const float* x, const float* w, const float* y, const float* gy, float* gx,
                                 float* gw, float* gb, int B, int K, int O, float negative_slope, void* stream) {
  if (!x || !w || !y || !gy) return LR_ENULL;
  if (B < 1 || B > 32 || K < 1 || O < 1 || O > 65535) return LR_EINVAL;
  hipStream_t st = lr_stream(stream);
  if (gw) {
    int kb = (K + 255) / 256;
    if (kb > 64) kb = 64;
    hipLaunchKernelGGL(linear_bwd_w_kernel, dim3((unsigned)kb, (unsigned)O), dim3(256), 0, st, x, y, gy, gw, gb, B, K, O, negative_slope);
    if (int e = lr_launch_status()) return e;
  }
  if (gx) {
    const unsigned nb = (unsigned)((K + 255) / 256);
    if ((size_t)O * 8 * 4 > 64 * 1024) return LR_EUNSUPPORTED;
    hipLaunchKernelGGL(linear_bwd_x_kernel<8>, dim3(nb, (unsigned)((B + 7) / 8)), dim3(256), (size_t)O * 8 * 4, st, w, y, gy, gx, B, 0, K, O, negative_slope);
    if (int e = lr_launch_status()) return e;
  }
  return LR_OK;
}
