// conv3d.hip — K3: Conv3d(k=3, pad=1, stride 1|2, bias) + LeakyReLU as an
// implicit GEMM on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: exact fp32,
// bit-for-bit an fmaf chain, 64 FLOP/clk/SIMD = the fp32 roofline).
//
//   GEMM view:  M = output voxels, N = Cout (16 | 32), K = 27 * Cin
//   MFMA tile:  16 voxels (consecutive along the output H axis) x 16 couts
//   wavefront:  MT=4 voxel tiles (4 consecutive output W rows) x NT cout tiles
//   block    :  4 wavefronts = 4 consecutive output D planes -> a 4x4x16 brick,
//               so the 3x3x3 halo is re-used through the CU's L1 and, with the
//               XCD-contiguous block order, through one XCD's L2.
//
// Replaces (reference file:line)
//   src/liftreg/layers/layers.py:335-372  convBlock (Conv3d + LeakyReLU(0.2))
//   src/liftreg/models/LiftRegDeformSubspaceBackproj.py:29-33,95-100 (6 blocks, strides 1,2,2,2,2,2)
//
// Layouts.  The encoder input is the reference's NCDHW cat([moving, target_volume]);
// activations between blocks are private to the model and kept channels-last
// (NDHWC): one voxel's Cin floats are then one 64/128-byte run and a lane's
// A-operand for four k-steps is a single 16-byte load.  The last block writes
// NCDHW again so nn.Flatten sees the reference's element order.
#include "lr_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MT = 4;  // voxel tiles per wavefront (along output W)
constexpr int TD = 4;  // wavefronts per block (along output D)

struct ConvDims {
  int B, Cin, Cout, D, W, H, Do, Wo, Ho;
  int nHq, nWq, nDq;
};

__device__ __forceinline__ float lrelu(float v, float slope) { return v >= 0.0f ? v : v * slope; }

// ---- epilogue shared by both kernels --------------------------------------
template <int NT>
__device__ __forceinline__ void store_tiles(const f32x4 (&acc)[MT][NT], float* __restrict__ out,
                                            const ConvDims& d, int b, int dz, int wo0, int hq,
                                            int lane, int out_layout, float slope) {
  const int col = lane & 15, rg = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int wo = wo0 + mt;
    if (wo >= d.Wo) continue;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = nt * 16 + col;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ho = hq * 16 + rg * 4 + r;
        if (ho >= d.Ho) continue;
        const float v = lrelu(acc[mt][nt][r], slope);
        int64_t o;
        if (out_layout == LR_LAYOUT_NDHWC)
          o = ((((int64_t)b * d.Do + dz) * d.Wo + wo) * d.Ho + ho) * d.Cout + co;
        else
          o = ((((int64_t)b * d.Cout + co) * d.Do + dz) * d.Wo + wo) * d.Ho + ho;
        out[o] = v;
      }
    }
  }
}

__device__ __forceinline__ void block_coords(const ConvDims& d, int& b, int& dq, int& wq, int& hq) {
  const unsigned lb = lr_xcd_remap(blockIdx.x, gridDim.x);
  hq = lb % d.nHq;
  wq = (lb / d.nHq) % d.nWq;
  dq = (lb / d.nHq / d.nWq) % d.nDq;
  b = lb / d.nHq / d.nWq / d.nDq;
}

// ---- channels-last input (Cin % 4 == 0) ------------------------------------
// K order: tap (27) x channel block cb (16 channels) ; inside a block lane group
// kq=lane>>4 owns channels 4kq..4kq+3 and feeds them to four MFMAs.
template <int NT, int STRIDE>
__global__ __launch_bounds__(256) void conv3d_cl_kernel(const float* __restrict__ in,
                                                        const float4* __restrict__ wp,
                                                        const float* __restrict__ bias,
                                                        float* __restrict__ out, ConvDims d,
                                                        int out_layout, float slope) {
  int b, dq, wq, hq;
  block_coords(d, b, dq, wq, hq);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int dz = dq * TD + wave;
  if (dz >= d.Do) return;
  const int wo0 = wq * MT;
  const int col = lane & 15, kq = lane >> 4;
  const int ho = hq * 16 + col;  // this lane's A-operand voxel
  const int CB = (d.Cin + 15) >> 4;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = nt * 16 + col;
    const float bv = (bias && co < d.Cout) ? bias[co] : 0.0f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = (f32x4){bv, bv, bv, bv};
  }

  const int xi0 = ho * STRIDE - 1;
  bool okx[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) okx[dx] = (ho < d.Ho) && (xi0 + dx >= 0) && (xi0 + dx < d.H);
  const int64_t inb = (int64_t)b * d.D * d.W * d.H * d.Cin;

  for (int tz = 0; tz < 3; ++tz) {
    const int zi = dz * STRIDE + tz - 1;
    if (zi < 0 || zi >= d.D) continue;  // wave-uniform
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
      int64_t rowoff[MT];
      bool oky[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int yi = (wo0 + mt) * STRIDE + ty - 1;
        oky[mt] = (wo0 + mt < d.Wo) && (yi >= 0) && (yi < d.W);
        rowoff[mt] = inb + ((int64_t)zi * d.W + (oky[mt] ? yi : 0)) * d.H * d.Cin;
      }
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        const int tap = (tz * 3 + ty) * 3 + tx;
        const int xi = okx[tx] ? xi0 + tx : 0;
        for (int cb = 0; cb < CB; ++cb) {
          const int c0 = cb * 16 + kq * 4;
          const bool okc = c0 < d.Cin;
          float4 bw[NT];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bw[nt] = wp[((tap * CB + cb) * NT + nt) * 64 + lane];
          float4 a[MT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const bool ok = oky[mt] && okx[tx] && okc;
            const float4 v = *reinterpret_cast<const float4*>(
                in + rowoff[mt] + (int64_t)xi * d.Cin + (okc ? c0 : 0));
            a[mt] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, bw[nt].x, acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, bw[nt].y, acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].z, bw[nt].z, acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].w, bw[nt].w, acc[mt][nt], 0, 0, 0);
            }
          }
        }
      }
    }
  }
  store_tiles<NT>(acc, out, d, b, dz, wo0, hq, lane, out_layout, slope);
}

// ---- planar (NCDHW) input, any Cin ------------------------------------------
// K order: channel c x 7 quads of taps (27 padded to 28); lane group kq owns tap
// 4q+kq of quad q, so one scalar load per lane feeds one MFMA.
template <int NT, int STRIDE>
__global__ __launch_bounds__(256) void conv3d_planar_kernel(const float* __restrict__ in,
                                                            const float* __restrict__ wp,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ out, ConvDims d,
                                                            int out_layout, float slope) {
  int b, dq, wq, hq;
  block_coords(d, b, dq, wq, hq);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int dz = dq * TD + wave;
  if (dz >= d.Do) return;
  const int wo0 = wq * MT;
  const int col = lane & 15, kq = lane >> 4;
  const int ho = hq * 16 + col;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = nt * 16 + col;
    const float bv = (bias && co < d.Cout) ? bias[co] : 0.0f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = (f32x4){bv, bv, bv, bv};
  }

  // per-lane tap offsets / validity for the 7 quads
  int off[MT][7];
  unsigned okbits[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) okbits[mt] = 0u;
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    const int tap = q * 4 + kq;
    const int tz = tap / 9, ty = (tap / 3) % 3, tx = tap % 3;
    const int zi = dz * STRIDE + tz - 1;
    const int xi = ho * STRIDE + tx - 1;
    const bool okzx = (tap < 27) && (zi >= 0) && (zi < d.D) && (ho < d.Ho) && (xi >= 0) && (xi < d.H);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int yi = (wo0 + mt) * STRIDE + ty - 1;
      const bool ok = okzx && (wo0 + mt < d.Wo) && (yi >= 0) && (yi < d.W);
      off[mt][q] = ok ? (zi * d.W + yi) * d.H + xi : 0;
      okbits[mt] |= ok ? (1u << q) : 0u;
    }
  }
  const int64_t V = (int64_t)d.D * d.W * d.H;
  const float* inb = in + (int64_t)b * d.Cin * V;
  for (int c = 0; c < d.Cin; ++c) {
    const float* inc = inb + (int64_t)c * V;
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      float bw[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bw[nt] = wp[((c * 7 + q) * NT + nt) * 64 + lane];
      float a[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float v = inc[off[mt][q]];
        a[mt] = ((okbits[mt] >> q) & 1u) ? v : 0.0f;
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], bw[nt], acc[mt][nt], 0, 0, 0);
    }
  }
  store_tiles<NT>(acc, out, d, b, dz, wo0, hq, lane, out_layout, slope);
}

// ---- weight packing -----------------------------------------------------------
__global__ void pack_cl_kernel(const float* __restrict__ w, float4* __restrict__ packed, int Cin,
                               int Cout, int CB, int NT) {
  const int total = 27 * CB * NT * 64;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int lane = idx & 63;
  const int nt = (idx >> 6) % NT;
  const int cb = (idx >> 6) / NT % CB;
  const int tap = (idx >> 6) / NT / CB;
  const int co = nt * 16 + (lane & 15);
  float v[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int ci = cb * 16 + (lane >> 4) * 4 + m;
    v[m] = (co < Cout && ci < Cin) ? w[((int64_t)co * Cin + ci) * 27 + tap] : 0.0f;
  }
  packed[idx] = make_float4(v[0], v[1], v[2], v[3]);
}

__global__ void pack_planar_kernel(const float* __restrict__ w, float* __restrict__ packed, int Cin,
                                   int Cout, int NT) {
  const int total = Cin * 7 * NT * 64;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int lane = idx & 63;
  const int nt = (idx >> 6) % NT;
  const int q = (idx >> 6) / NT % 7;
  const int c = (idx >> 6) / NT / 7;
  const int co = nt * 16 + (lane & 15);
  const int tap = q * 4 + (lane >> 4);
  packed[idx] = (co < Cout && tap < 27) ? w[((int64_t)co * Cin + c) * 27 + tap] : 0.0f;
}

}  // namespace

extern "C" int64_t lr_conv3d_packed_floats(int Cin, int Cout, int in_layout) {
  if (Cin < 1 || (Cout != 16 && Cout != 32)) return LR_EUNSUPPORTED;
  const int NT = Cout / 16;
  if (in_layout == LR_LAYOUT_NDHWC) return (int64_t)27 * ((Cin + 15) / 16) * NT * 64 * 4;
  if (in_layout == LR_LAYOUT_NCDHW) return (int64_t)Cin * 7 * NT * 64;
  return LR_EINVAL;
}

extern "C" int lr_conv3d_pack_weights_f32(const float* weight, float* packed, int Cin, int Cout,
                                          int in_layout, void* stream) {
  if (!weight || !packed) return LR_ENULL;
  if (Cin < 1) return LR_EINVAL;
  if (Cout != 16 && Cout != 32) return LR_EUNSUPPORTED;
  const int NT = Cout / 16;
  if (in_layout == LR_LAYOUT_NDHWC) {
    if (Cin % 4) return LR_EUNSUPPORTED;
    const int CB = (Cin + 15) / 16;
    const int total = 27 * CB * NT * 64;
    hipLaunchKernelGGL(pack_cl_kernel, dim3((total + 255) / 256), dim3(256), 0, lr_stream(stream),
                       weight, reinterpret_cast<float4*>(packed), Cin, Cout, CB, NT);
  } else if (in_layout == LR_LAYOUT_NCDHW) {
    const int total = Cin * 7 * NT * 64;
    hipLaunchKernelGGL(pack_planar_kernel, dim3((total + 255) / 256), dim3(256), 0,
                       lr_stream(stream), weight, packed, Cin, Cout, NT);
  } else {
    return LR_EINVAL;
  }
  return lr_launch_status();
}

extern "C" int lr_conv3d_k3_lrelu_f32(const float* in, const float* packed_w, const float* bias,
                                      float* out, int B, int Cin, int Cout, int D, int W, int H,
                                      int stride, int in_layout, int out_layout,
                                      float negative_slope, void* stream) {
  if (!in || !packed_w || !out) return LR_ENULL;
  if (B < 1 || Cin < 1 || D < 1 || W < 1 || H < 1) return LR_EINVAL;
  if (stride != 1 && stride != 2) return LR_EUNSUPPORTED;
  if (Cout != 16 && Cout != 32) return LR_EUNSUPPORTED;
  if (out_layout != LR_LAYOUT_NCDHW && out_layout != LR_LAYOUT_NDHWC) return LR_EINVAL;
  ConvDims d;
  d.B = B; d.Cin = Cin; d.Cout = Cout; d.D = D; d.W = W; d.H = H;
  d.Do = (D - 1) / stride + 1; d.Wo = (W - 1) / stride + 1; d.Ho = (H - 1) / stride + 1;
  d.nHq = (d.Ho + 15) / 16; d.nWq = (d.Wo + MT - 1) / MT; d.nDq = (d.Do + TD - 1) / TD;
  // the planar kernel keeps per-batch spatial offsets in 32 bits
  if ((int64_t)D * W * H > 0x7fffffffLL) return LR_EINVAL;
  const int64_t nblk = (int64_t)B * d.nDq * d.nWq * d.nHq;
  if (nblk > 0x7fffffffLL) return LR_EINVAL;
  const dim3 grid((unsigned)nblk), block(256);
  hipStream_t st = lr_stream(stream);
  const int NT = Cout / 16;
#define LR_CONV(KERNEL, WT)                                                                     \
  do {                                                                                          \
    if (NT == 1 && stride == 1) hipLaunchKernelGGL((KERNEL<1, 1>), grid, block, 0, st, in, WT, bias, out, d, out_layout, negative_slope); \
    else if (NT == 1) hipLaunchKernelGGL((KERNEL<1, 2>), grid, block, 0, st, in, WT, bias, out, d, out_layout, negative_slope);           \
    else if (stride == 1) hipLaunchKernelGGL((KERNEL<2, 1>), grid, block, 0, st, in, WT, bias, out, d, out_layout, negative_slope);       \
    else hipLaunchKernelGGL((KERNEL<2, 2>), grid, block, 0, st, in, WT, bias, out, d, out_layout, negative_slope);                        \
  } while (0)
  if (in_layout == LR_LAYOUT_NDHWC) {
    if (Cin % 4) return LR_EUNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(in) & 15u) return LR_EALIGN;
    LR_CONV(conv3d_cl_kernel, reinterpret_cast<const float4*>(packed_w));
  } else if (in_layout == LR_LAYOUT_NCDHW) {
    LR_CONV(conv3d_planar_kernel, packed_w);
  } else {
    return LR_EINVAL;
  }
#undef LR_CONV
  return lr_launch_status();
}
