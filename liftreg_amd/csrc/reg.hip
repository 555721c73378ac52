// reg.hip — a16 (regulariser part): mean over voxels of sum_{c,axis} (d_axis disp_c)^2, one streaming pass.
//
// Replaces (reference file:line)
//   src/liftreg/losses/SubspaceLoss.py:51-67  compute_reg_loss: FD_torch(spacing*2).dXc/dYc/dZc squared, torch.mean
//
// PARITY UNPINNED: the finite-difference stencil lives in `mermaid` 0.3.2 (requirements.txt:61), which is
// neither vendored nor installed and has no fixture in the reference.  Assumed: central
// differences dXc = (I[x+1]-I[x-1]) * 0.5 / spacing with linearly extrapolated boundaries, i.e. the
// one-sided difference (I[1]-I[0])/spacing at a face; spacing passed by the reference = 2/(shape-1).
// HBM-read-bound: 12 bytes per voxel (three channels, neighbours come from L2).
#include "lr_common.h"

namespace {

__device__ __forceinline__ float diff_c(const float* p, int64_t stride, int i, int n, float inv_h) {
  // central difference with linear extrapolation at the faces
  if (n < 2) return 0.0f;
  if (i == 0) return (p[stride] - p[0]) * inv_h;
  if (i == n - 1) return (p[0] - p[-stride]) * inv_h;
  return (p[stride] - p[-stride]) * (0.5f * inv_h);
}

__global__ __launch_bounds__(256) void disp_reg_kernel(const float* __restrict__ disp,
                                                       double* __restrict__ partial, int D, int W,
                                                       int H, float ihd, float ihw, float ihh) {
  const int b = blockIdx.y;
  const int64_t V = (int64_t)D * W * H;
  const float* base = disp + (int64_t)b * 3 * V;
  double acc = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += stride) {
    const int k = (int)(v % H);
    const int j = (int)((v / H) % W);
    const int i = (int)(v / H / W);
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* p = base + c * V + v;
      const float dx = diff_c(p, (int64_t)W * H, i, D, ihd);
      const float dy = diff_c(p, H, j, W, ihw);
      const float dz = diff_c(p, 1, k, H, ihh);
      s += dx * dx + dy * dy + dz * dz;
    }
    acc += (double)s;
  }
  __shared__ double red[4];
  acc = lr_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[(int64_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void disp_reg_final_kernel(const double* __restrict__ partial,
                                                             float* __restrict__ out, int n, double denom) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = lr_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = (float)(((red[0] + red[1]) + (red[2] + red[3])) / denom);
}

// Gradient of the regulariser: reg = (1/N) sum_x sum_{c,axis} d(x)^2 with d(x') = coef(x')*(f[up(x')] - f[dn(x')]),
// interior: up/dn = x'+1/x'-1, coef = 0.5/h; faces: (1,0) resp. (n-1,n-2), coef = 1/h.
// d reg / d f[x] = (2/N) sum_axis sum_{x' in {x-1,x,x+1}} d(x') * coef(x') * ([up(x')==x] - [dn(x')==x]).
__device__ __forceinline__ float axis_grad(const float* p, int64_t stride, int i, int n, float inv_h) {
  if (n < 2) return 0.0f;
  float g = 0.0f;
#pragma unroll
  for (int o = -1; o <= 1; ++o) {
    const int q = i + o;  // the derivative sample x'
    if (q < 0 || q >= n) continue;
    const int up = q == 0 ? 1 : (q == n - 1 ? n - 1 : q + 1);
    const int dn = q == 0 ? 0 : (q == n - 1 ? n - 2 : q - 1);
    const float coef = (q == 0 || q == n - 1) ? inv_h : 0.5f * inv_h;
    const float sgn = (up == i ? 1.0f : 0.0f) - (dn == i ? 1.0f : 0.0f);
    if (sgn != 0.0f) g += coef * (p[(int64_t)(up - i) * stride] - p[(int64_t)(dn - i) * stride]) * coef * sgn;
  }
  return g;
}

__global__ __launch_bounds__(256) void disp_reg_bwd_kernel(const float* __restrict__ disp,
                                                           const float* __restrict__ gout,
                                                           float* __restrict__ gdisp, int B, int D, int W, int H,
                                                           float ihd, float ihw, float ihh) {
  const int64_t V = (int64_t)D * W * H, total = (int64_t)B * 3 * V;
  const float scale = (*gout) * 2.0f / (float)((double)B * (double)V);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const int64_t v = idx % V;
    const int k = (int)(v % H), j = (int)((v / H) % W), i = (int)(v / H / W);
    const float* p = disp + idx;
    const float g = axis_grad(p, (int64_t)W * H, i, D, ihd) + axis_grad(p, H, j, W, ihw) + axis_grad(p, 1, k, H, ihh);
    gdisp[idx] = g * scale;
  }
}

}  // namespace

extern "C" int lr_disp_reg_bwd_f32(const float* disp, const float* gout, float* gdisp, int B, int D, int W, int H,
                                   void* stream) {
  if (!disp || !gout || !gdisp) return LR_ENULL;
  if (B < 1 || D < 1 || W < 1 || H < 1) return LR_EINVAL;
  const float ihd = D > 1 ? 0.5f * (float)(D - 1) : 0.0f;
  const float ihw = W > 1 ? 0.5f * (float)(W - 1) : 0.0f;
  const float ihh = H > 1 ? 0.5f * (float)(H - 1) : 0.0f;
  int64_t nblk = ((int64_t)B * 3 * D * W * H + 255) / 256;
  if (nblk > 8192) nblk = 8192;
  hipLaunchKernelGGL(disp_reg_bwd_kernel, dim3((unsigned)nblk), dim3(256), 0, lr_stream(stream), disp, gout, gdisp, B,
                     D, W, H, ihd, ihw, ihh);
  return lr_launch_status();
}

extern "C" int lr_disp_reg_f32(const float* disp, double* partial, float* out, int B, int D, int W,
                               int H, int nblk, void* stream) {
  if (!disp || !partial || !out) return LR_ENULL;
  if (B < 1 || B > 65535 || D < 1 || W < 1 || H < 1 || nblk < 1 || nblk > 65535) return LR_EINVAL;
  // spacing = 1/(shape-1); FD_torch(spacing*2)  =>  1/h = (shape-1)/2
  const float ihd = D > 1 ? 0.5f * (float)(D - 1) : 0.0f;
  const float ihw = W > 1 ? 0.5f * (float)(W - 1) : 0.0f;
  const float ihh = H > 1 ? 0.5f * (float)(H - 1) : 0.0f;
  hipStream_t st = lr_stream(stream);
  hipLaunchKernelGGL(disp_reg_kernel, dim3((unsigned)nblk, (unsigned)B), dim3(256), 0, st, disp, partial, D,
                     W, H, ihd, ihw, ihh);
  if (int e = lr_launch_status()) return e;
  hipLaunchKernelGGL(disp_reg_final_kernel, dim3(1), dim3(256), 0, st, partial, out, B * nblk,
                     (double)B * D * W * H);
  return lr_launch_status();
}
