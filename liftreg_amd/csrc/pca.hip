// pca.hip — K5: disp = coefs · basis + mean, a streaming skinny GEMM.
// HBM-read-bound on the basis: 4*L*M bytes per BATCH (11.27 GB at 256^3, L=56),
// read exactly once with 16-byte loads, lanes along M; the B×L coefficients sit
// in LDS transposed so every l-step is one broadcast read.
//
// Replaces (reference file:line)
//   src/liftreg/models/LiftRegDeformSubspaceBackproj.py:42-43  pca_vectors (.T view), pca_mean
//   src/liftreg/models/LiftRegDeformSubspaceBackproj.py:102    F.linear(x, pca_vectors, pca_mean)
#include "lr_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// BF: the basis is bf16 storage (half the bytes of this HBM-bound kernel); 4 elements = one 8-byte load, exact
// expansion to fp32, the arithmetic is unchanged.
template <int BT, bool BF>
__global__ __launch_bounds__(256) void pca_kernel(const float* __restrict__ coefs,
                                                  const float* __restrict__ basis,
                                                  const float* __restrict__ mean,
                                                  float* __restrict__ disp, int B, int b_lo, int L,
                                                  int64_t M, int64_t ldb, int64_t dstride) {
  extern __shared__ float cs[];  // [L][BT]
  for (int t = threadIdx.x; t < L * BT; t += blockDim.x) {
    const int l = t / BT, b = t % BT;
    cs[t] = (b_lo + b < B) ? coefs[(int64_t)(b_lo + b) * L + l] : 0.0f;
  }
  __syncthreads();
  const int64_t m = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (m >= M) return;
  float4 acc[BT];
  const float4 mu = *reinterpret_cast<const float4*>(mean + m);
#pragma unroll
  for (int b = 0; b < BT; ++b) acc[b] = mu;
  const float* bp = basis + m;
#pragma unroll 8
  for (int l = 0; l < L; ++l) {
    f32x4 v;
    if (BF) {
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      const u32x2 raw = __builtin_nontemporal_load(
          reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(basis) + (int64_t)l * ldb + m));
      v[0] = __builtin_bit_cast(float, raw.x << 16);
      v[1] = __builtin_bit_cast(float, raw.x & 0xffff0000u);
      v[2] = __builtin_bit_cast(float, raw.y << 16);
      v[3] = __builtin_bit_cast(float, raw.y & 0xffff0000u);
    } else {
      v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(bp + (int64_t)l * ldb));
    }
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      const float c = cs[l * BT + b];
      acc[b].x = fmaf(c, v.x, acc[b].x);
      acc[b].y = fmaf(c, v.y, acc[b].y);
      acc[b].z = fmaf(c, v.z, acc[b].z);
      acc[b].w = fmaf(c, v.w, acc[b].w);
    }
  }
#pragma unroll
  for (int b = 0; b < BT; ++b)
    if (b_lo + b < B) *reinterpret_cast<float4*>(disp + (int64_t)(b_lo + b) * dstride + m) = acc[b];
}

// Any M / leading dimension / alignment (3·D·W·H is a multiple of 4 only when the voxel count is): one element per
// thread, 8 batch rows per pass; same fmaf chain per element as the vector kernel.
template <bool BF>
__global__ __launch_bounds__(256) void pca_scalar_kernel(const float* __restrict__ coefs, const float* __restrict__ basis,
                                                         const float* __restrict__ mean, float* __restrict__ disp, int B,
                                                         int b_lo, int L, int64_t M, int64_t ldb, int64_t dstride) {
  extern __shared__ float cs[];  // [L][8]
  for (int t = threadIdx.x; t < L * 8; t += blockDim.x) {
    const int l = t / 8, b = t % 8;
    cs[t] = (b_lo + b < B) ? coefs[(int64_t)(b_lo + b) * L + l] : 0.0f;
  }
  __syncthreads();
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  float acc[8];
  const float mu = mean[m];
#pragma unroll
  for (int b = 0; b < 8; ++b) acc[b] = mu;
  for (int l = 0; l < L; ++l) {
    float v;
    if (BF) v = __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(basis)[(int64_t)l * ldb + m] << 16);
    else v = basis[(int64_t)l * ldb + m];
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[b] = fmaf(cs[l * 8 + b], v, acc[b]);
  }
#pragma unroll
  for (int b = 0; b < 8; ++b)
    if (b_lo + b < B) disp[(int64_t)(b_lo + b) * dstride + m] = acc[b];
}

}  // namespace

static int pca_impl(const float* coefs, const float* basis, bool bf, const float* mean, float* disp, int B, int L,
                    int64_t M, int64_t ldb, int64_t disp_batch_stride, void* stream) {
  if (!coefs || !basis || !mean || !disp) return LR_ENULL;
  if (B < 1 || B > 32 || L < 1 || L > 4096 || M < 1 || ldb < M || disp_batch_stride < M)
    return LR_EINVAL;
  hipStream_t st = lr_stream(stream);
  const bool vec_ok = !((M & 3) || (ldb & 3) || (disp_batch_stride & 3)) &&
                      !((reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(disp)) & 15u) &&
                      !(reinterpret_cast<uintptr_t>(basis) & (bf ? 7u : 15u));
  if (!vec_ok) {  // odd voxel counts, sliced views: the scalar kernel (same results, lower bandwidth)
    const int64_t nb = (M + 255) / 256;
    if (nb > 0x7fffffffLL) return LR_EINVAL;
    for (int b_lo = 0; b_lo < B; b_lo += 8) {
      if (bf) hipLaunchKernelGGL(pca_scalar_kernel<true>, dim3((unsigned)nb), dim3(256), (size_t)L * 8 * 4, st, coefs, basis,
                                 mean, disp, B, b_lo, L, M, ldb, disp_batch_stride);
      else hipLaunchKernelGGL(pca_scalar_kernel<false>, dim3((unsigned)nb), dim3(256), (size_t)L * 8 * 4, st, coefs, basis,
                              mean, disp, B, b_lo, L, M, ldb, disp_batch_stride);
      if (int e = lr_launch_status()) return e;
    }
    return LR_OK;
  }
  const int64_t nblk = (M / 4 + 255) / 256;
  if (nblk > 0x7fffffffLL) return LR_EINVAL;
  // Batch tiles of 8 (4 for small batches): the basis is re-read once per tile.
  for (int b_lo = 0; b_lo < B;) {
    const int rem = B - b_lo;
    if (rem > 4) {
      if (bf) hipLaunchKernelGGL((pca_kernel<8, true>), dim3((unsigned)nblk), dim3(256), (size_t)L * 8 * 4, st,
                                 coefs, basis, mean, disp, B, b_lo, L, M, ldb, disp_batch_stride);
      else hipLaunchKernelGGL((pca_kernel<8, false>), dim3((unsigned)nblk), dim3(256), (size_t)L * 8 * 4, st,
                              coefs, basis, mean, disp, B, b_lo, L, M, ldb, disp_batch_stride);
      b_lo += 8;
    } else if (rem > 1) {
      if (bf) hipLaunchKernelGGL((pca_kernel<4, true>), dim3((unsigned)nblk), dim3(256), (size_t)L * 4 * 4, st,
                                 coefs, basis, mean, disp, B, b_lo, L, M, ldb, disp_batch_stride);
      else hipLaunchKernelGGL((pca_kernel<4, false>), dim3((unsigned)nblk), dim3(256), (size_t)L * 4 * 4, st,
                              coefs, basis, mean, disp, B, b_lo, L, M, ldb, disp_batch_stride);
      b_lo += 4;
    } else {
      if (bf) hipLaunchKernelGGL((pca_kernel<1, true>), dim3((unsigned)nblk), dim3(256), (size_t)L * 1 * 4, st,
                                 coefs, basis, mean, disp, B, b_lo, L, M, ldb, disp_batch_stride);
      else hipLaunchKernelGGL((pca_kernel<1, false>), dim3((unsigned)nblk), dim3(256), (size_t)L * 1 * 4, st,
                              coefs, basis, mean, disp, B, b_lo, L, M, ldb, disp_batch_stride);
      b_lo += 1;
    }
    if (int e = lr_launch_status()) return e;
  }
  return LR_OK;
}

extern "C" int lr_pca_reconstruct_f32(const float* coefs, const float* basis, const float* mean,
                                      float* disp, int B, int L, int64_t M, int64_t ldb,
                                      int64_t disp_batch_stride, void* stream) {
  return pca_impl(coefs, basis, false, mean, disp, B, L, M, ldb, disp_batch_stride, stream);
}

// Same with the basis stored as bf16 (L, ldb) — an opt-in storage format for the 11 GB basis (model option
// "pca_dtype": "bf16"); coefficients, mean, accumulation and the displacement field stay fp32.
extern "C" int lr_pca_reconstruct_bf16basis_f32(const float* coefs, const void* basis_bf16, const float* mean,
                                                float* disp, int B, int L, int64_t M, int64_t ldb,
                                                int64_t disp_batch_stride, void* stream) {
  return pca_impl(coefs, reinterpret_cast<const float*>(basis_bf16), true, mean, disp, B, L, M, ldb, disp_batch_stride,
                  stream);
}
