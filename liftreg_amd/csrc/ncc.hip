// ncc.hip — K8: normalised cross-correlation similarity in ONE pass over the
// two volumes (the reference makes ~8 passes: means, centring, three products).
// HBM-read-bound: 8 bytes per voxel pair.  Five raw moments per row are
// accumulated in fp64 (wave shuffles → LDS → per-block partial → fixed-order
// final reduce: bitwise reproducible, no atomics) and turned into the loss by a
// one-block epilogue.  Moments of disjoint z-slabs add, so a sharded volume
// needs only a 5*R-double all-reduce.
//
// Replaces (reference file:line)
//   src/liftreg/layers/losses.py:14-29   NCCLoss (configured, cur_task_setting.json:51)
//   src/liftreg/layers/layers.py:238-255 NCCLoss (squared, per-channel variant)
#include "lr_common.h"

namespace {

__global__ __launch_bounds__(256) void ncc_moments_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ y,
                                                          double* __restrict__ partial,
                                                          int64_t N, int nblk, int vec_ok) {
  const int r = blockIdx.y;
  const float* xr = x + (int64_t)r * N;
  const float* yr = y + (int64_t)r * N;
  double sx = 0, sy = 0, sxy = 0, sxx = 0, syy = 0;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthr = (int64_t)nblk * blockDim.x;
  if (vec_ok) {
    const int64_t n4 = N >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(xr);
    const float4* y4 = reinterpret_cast<const float4*>(yr);
    for (int64_t i = tid; i < n4; i += nthr) {
      const float4 a = x4[i], b = y4[i];
      // products of two fp32 are exact in fp64
      const double ax = a.x, ay = a.y, az = a.z, aw = a.w;
      const double bx = b.x, by = b.y, bz = b.z, bw = b.w;
      sx += (ax + ay) + (az + aw);
      sy += (bx + by) + (bz + bw);
      sxy += (ax * bx + ay * by) + (az * bz + aw * bw);
      sxx += (ax * ax + ay * ay) + (az * az + aw * aw);
      syy += (bx * bx + by * by) + (bz * bz + bw * bw);
    }
    for (int64_t i = (n4 << 2) + tid; i < N; i += nthr) {
      const double a = xr[i], b = yr[i];
      sx += a; sy += b; sxy += a * b; sxx += a * a; syy += b * b;
    }
  } else {
    for (int64_t i = tid; i < N; i += nthr) {
      const double a = xr[i], b = yr[i];
      sx += a; sy += b; sxy += a * b; sxx += a * a; syy += b * b;
    }
  }
  __shared__ double red[4][5];
  sx = lr_wave_sum(sx); sy = lr_wave_sum(sy); sxy = lr_wave_sum(sxy);
  sxx = lr_wave_sum(sxx); syy = lr_wave_sum(syy);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    red[wave][0] = sx; red[wave][1] = sy; red[wave][2] = sxy; red[wave][3] = sxx; red[wave][4] = syy;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    const double s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    partial[((int64_t)r * nblk + blockIdx.x) * 5 + threadIdx.x] = s;
  }
}

// one block per row; fixed-order tree over the per-block partials
__global__ __launch_bounds__(256) void ncc_reduce_kernel(const double* __restrict__ partial,
                                                         double* __restrict__ moments, int nblk) {
  const int r = blockIdx.x;
  __shared__ double red[4][5];
  double s[5] = {0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
    const double* p = partial + ((int64_t)r * nblk + i) * 5;
#pragma unroll
    for (int q = 0; q < 5; ++q) s[q] += p[q];
  }
#pragma unroll
  for (int q = 0; q < 5; ++q) s[q] = lr_wave_sum(s[q]);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0)
    for (int q = 0; q < 5; ++q) red[wave][q] = s[q];
  __syncthreads();
  if (threadIdx.x < 5)
    moments[(int64_t)r * 5 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(64) void ncc_loss_kernel(const double* __restrict__ moments,
                                                      float* __restrict__ loss,
                                                      float* __restrict__ ncc_rows, int R,
                                                      double n, int n_batch, int variant) {
  double acc = 0.0;
  for (int r = threadIdx.x; r < R; r += 64) {
    const double* m = moments + (int64_t)r * 5;
    const double mx = m[0] / n, my = m[1] / n;
    const double cov = m[2] / n - mx * my;
    // raw-moment variances cancel to ~1e-17 on a constant volume: clamp, or (vx+e2)(vy+e2) can go negative → NaN
    // (the reference's centred form stays finite there, layers/losses.py:18-26)
    const double vx = fmax(m[3] / n - mx * mx, 0.0);
    const double vy = fmax(m[4] / n - my * my, 0.0);
    double v;
    if (variant == LR_NCC_CONFIGURED) {
      // a = x - mean(x) + 1e-10 ; mean(ab) = cov + 1e-20 (mean(x-mean) = 0)
      const double e2 = 1e-20;
      v = (cov + e2) / sqrt((vx + e2) * (vy + e2));
    } else {
      v = (cov * cov) / (vx * vy + 1e-12);
    }
    if (ncc_rows) ncc_rows[r] = (float)v;
    acc += v;
  }
  acc = lr_wave_sum(acc);
  if (threadIdx.x == 0) {
    // configured: 1 - mean over rows ; squared: 1 - (sum_b mean_c)/n_batch = 1 - sum_r/(C*n_batch) = 1 - sum_r/R
    (void)n_batch;
    *loss = (float)(1.0 - acc / (double)R);
  }
}

}  // namespace

// shared with warp.hip's one-pass decode, whose epilogue leaves per-block partials in the same [row][block][5] layout
int lr_internal_ncc_reduce(const double* partial, double* moments, int R, int nblk, hipStream_t st) {
  hipLaunchKernelGGL(ncc_reduce_kernel, dim3((unsigned)R), dim3(256), 0, st, partial, moments, nblk);
  return lr_launch_status();
}

extern "C" int lr_ncc_moments_f32(const float* x, const float* y, double* partial, double* moments,
                                  int R, int64_t N, int nblk, void* stream) {
  if (!x || !y || !partial || !moments) return LR_ENULL;
  if (R < 1 || R > 65535 || N < 1 || nblk < 1 || nblk > 65535) return LR_EINVAL;
  const int vec_ok = ((N & 3) == 0) &&
                     (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u) == 0);
  hipStream_t st = lr_stream(stream);
  hipLaunchKernelGGL(ncc_moments_kernel, dim3((unsigned)nblk, (unsigned)R), dim3(256), 0, st, x, y,
                     partial, N, nblk, vec_ok);
  if (int e = lr_launch_status()) return e;
  hipLaunchKernelGGL(ncc_reduce_kernel, dim3((unsigned)R), dim3(256), 0, st, partial, moments, nblk);
  return lr_launch_status();
}

extern "C" int lr_ncc_loss_from_moments(const double* moments, float* loss, float* ncc_rows, int R,
                                        int64_t n_total, int n_batch, int variant, void* stream) {
  if (!moments || !loss) return LR_ENULL;
  if (R < 1 || n_total < 1 || n_batch < 1 || R % n_batch) return LR_EINVAL;
  if (variant != LR_NCC_CONFIGURED && variant != LR_NCC_SQUARED) return LR_EINVAL;
  hipLaunchKernelGGL(ncc_loss_kernel, dim3(1), dim3(64), 0, lr_stream(stream), moments, loss,
                     ncc_rows, R, (double)n_total, n_batch, variant);
  return lr_launch_status();
}
