// conv3d.hip — K3: Conv3d(k=3, pad=1, stride 1|2, bias) + LeakyReLU as an
// implicit GEMM on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: exact fp32,
// bit-for-bit an fmaf chain, 64 FLOP/clk/SIMD = the fp32 roofline).
//
//   GEMM view :  D[cout][voxel] = sum_k W[cout][k] * X[k][voxel],  k = (tap, cin)
//   MFMA      :  A = weights (16 couts x 4 k), B = activations (4 k x 16 voxels);
//                the 16 voxels of a tile are consecutive along the output H axis.
//                With couts on the accumulator ROWS each lane ends up holding 4
//                consecutive couts of ONE voxel, so a channels-last tile leaves as
//                one 16-byte-per-lane, 1-KiB-contiguous wavefront store.
//
// Two kernels:
//  * conv3d_planar_kernel — NCDHW input, any Cin (the encoder's first block,
//    Cin = P+1).  The input brick of a 4x4x64 output brick is staged ONCE in LDS
//    (zero-filled halo = the conv's padding), the block's weights live in
//    registers, and the inner loop is ds_read_b32(immediate offset) -> MFMA with
//    no address arithmetic and no vector-memory traffic at all.
//  * conv3d_cl_kernel — channels-last (NDHWC) input, Cin % 4 == 0 (blocks 1..5):
//    one voxel's channels are contiguous, a lane's B operand for four k-steps is
//    a single 16-byte load straight from L1/L2; 4x4x16 output brick per block.
//
// Replaces (reference file:line)
//   src/liftreg/layers/layers.py:335-372  convBlock (Conv3d + LeakyReLU(0.2))
//   src/liftreg/models/LiftRegDeformSubspaceBackproj.py:29-33,95-100 (6 blocks, strides 1,2,2,2,2,2)
#include "lr_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct ConvDims {
  int B, Cin, Cout, D, W, H, Do, Wo, Ho;
  int nHq, nWq, nDq;
};

__device__ __forceinline__ float lrelu(float v, float slope) { return v >= 0.0f ? v : v * slope; }

__device__ __forceinline__ void block_coords(const ConvDims& d, int& b, int& dq, int& wq, int& hq) {
  const unsigned lb = lr_xcd_remap(blockIdx.x, gridDim.x);
  hq = lb % d.nHq;
  wq = (lb / d.nHq) % d.nWq;
  dq = (lb / d.nHq / d.nWq) % d.nDq;
  b = lb / d.nHq / d.nWq / d.nDq;
}

// One 16-voxel x 16-cout accumulator tile -> memory.  Lane l holds couts
// nt*16 + (l>>4)*4 + {0..3} of voxel (l&15).
__device__ __forceinline__ void store_tile(const f32x4& acc, float* __restrict__ out,
                                           const ConvDims& d, int b, int dz, int wo, int ho, int nt,
                                           int lane, int out_layout, float slope) {
  if (wo >= d.Wo || ho >= d.Ho) return;
  const int c0 = nt * 16 + (lane >> 4) * 4;
  f32x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = lrelu(acc[r], slope);
  if (out_layout == LR_LAYOUT_NDHWC) {
    float* o = out + ((((int64_t)b * d.Do + dz) * d.Wo + wo) * d.Ho + ho) * d.Cout + c0;
    *reinterpret_cast<f32x4*>(o) = v;
  } else {
    const int64_t vo = (int64_t)d.Do * d.Wo * d.Ho;
    float* o = out + ((int64_t)b * d.Cout + c0) * vo + ((int64_t)dz * d.Wo + wo) * d.Ho + ho;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r * vo] = v[r];
  }
}

__device__ __forceinline__ f32x4 bias_init(const float* __restrict__ bias, int nt, int lane) {
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
    const int c0 = nt * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = bias[c0 + r];
  }
  return a;
}

// ===========================================================================
// Planar (NCDHW) input through LDS.
// Output brick per block: PD=4 (one plane per wavefront) x PW=4 x PH=64 voxels.
// K order: channel c, then 7 quads of taps (27 padded to 28); lane group kq=lane>>4
// owns tap 4q+kq, so one ds_read_b32 per lane feeds one MFMA.
// ===========================================================================
constexpr int PD = 4, PW = 4, PH = 64, PCC = 3;  // PCC = channels staged per pass (P=2 views -> Cin=3: one pass)

template <int S>
struct PlanarGeom {
  static constexpr int RH = (PH - 1) * S + 3;   // input extent along H a brick needs (66 | 129)
  static constexpr int RW = (PW - 1) * S + 3;   // 6 | 9
  static constexpr int RD = (PD - 1) * S + 3;   // 6 | 9
  // Rows are staged from an ALIGNED window that starts 3 floats left of the halo
  // (x = ho0*S - 4) so every global load and LDS store is a 16-byte one.
  static constexpr int XOFF = 3;
  static constexpr int RSL = (RH + XOFF + 3) / 4 * 4;  // row stride = window length (72 | 132)
  static constexpr int F4 = RSL / 4;             // float4 per row (18 | 33)
  static constexpr int RPI = 64 / F4;            // rows per wavefront load instruction (3 | 1)
  static constexpr int PS = RW * RSL;            // plane stride
  static constexpr int CS = RD * PS;             // channel stride
  static constexpr int MAXIT = S == 1 ? (PCC * RD * RW + 4 * RPI - 1) / (4 * RPI) : 1;  // staging iterations per wave
};

template <int NT, int S>
__global__ __launch_bounds__(256) void conv3d_planar_kernel(const float* __restrict__ in,
                                                            const float* __restrict__ wp,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ out, ConvDims d,
                                                            int out_layout, float slope, int pcc, int vec4) {
  using G = PlanarGeom<S>;
  extern __shared__ float brick[];  // [PCC][RD][RW][RSL]
  int b, dq, wq, hq;
  block_coords(d, b, dq, wq, hq);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int do0 = dq * PD, wo0 = wq * PW, ho0 = hq * PH;
  const int col = lane & 15, kq = lane >> 4;

  f32x4 acc[PW][4][NT];  // [tile row along W][tile along H][cout tile]
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const f32x4 bv = bias_init(bias, nt, lane);
#pragma unroll
    for (int y = 0; y < PW; ++y)
#pragma unroll
      for (int x = 0; x < 4; ++x) acc[y][x][nt] = bv;
  }

  // per-lane LDS offsets of the 7 tap quads (tap 27 is padding: weight 0, address of tap 26)
  int qoff[7];
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    const int tap = min(q * 4 + kq, 26);
    const int tz = tap / 9, ty = (tap / 3) % 3, tx = tap % 3;
    qoff[q] = (wave * S + tz) * G::PS + ty * G::RSL + G::XOFF + tx + col * S;
  }

  const int64_t V = (int64_t)d.D * d.W * d.H;
  const int z_in0 = do0 * S - 1, y_in0 = wo0 * S - 1, x_in0 = ho0 * S - 1;
  for (int c0 = 0; c0 < d.Cin; c0 += pcc) {
    const int ncc = min(pcc, d.Cin - c0);
    if (c0) __syncthreads();  // previous pass has finished reading the brick
    // ---- stage the brick: all loads of a pass are issued before the first LDS store so their
    //      latencies overlap (one memory round trip per pass, not one per row)
    const int nrows = ncc * G::RD * G::RW;
    if (S == 1 && vec4) {  // (stride-2 planar input is off the model's path: scalar staging only)
      const int lrow = lane / G::F4, lf4 = lane - lrow * G::F4;
      const bool lact = lane < G::RPI * G::F4;
      const int xi = x_in0 - G::XOFF + lf4 * 4;  // multiple of 4: a float4 is entirely in or out
      const bool xok = xi >= 0 && xi + 3 < d.H;
      float4 st[G::MAXIT];
#pragma unroll
      for (int it = 0; it < G::MAXIT; ++it) {
        const int row = (it * 4 + wave) * G::RPI + lrow;
        const int cc = row / (G::RD * G::RW), rz = (row / G::RW) % G::RD, ry = row % G::RW;
        const int zi = z_in0 + rz, yi = y_in0 + ry;
        const bool ok = lact && row < nrows && xok && zi >= 0 && zi < d.D && yi >= 0 && yi < d.W;
        st[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok)
          st[it] = *reinterpret_cast<const float4*>(in + ((int64_t)b * d.Cin + c0 + cc) * V +
                                                    ((int64_t)zi * d.W + yi) * d.H + xi);
      }
#pragma unroll
      for (int it = 0; it < G::MAXIT; ++it) {
        const int row = (it * 4 + wave) * G::RPI + lrow;
        const int cc = row / (G::RD * G::RW), rz = (row / G::RW) % G::RD, ry = row % G::RW;
        if (lact && row < nrows)
          *reinterpret_cast<float4*>(brick + cc * G::CS + rz * G::PS + ry * G::RSL + lf4 * 4) = st[it];
      }
    } else {  // H % 4 != 0 or unaligned base: scalar staging of the same window
      for (int row = wave; row < nrows; row += 4) {
        const int cc = row / (G::RD * G::RW), rz = (row / G::RW) % G::RD, ry = row % G::RW;
        const int zi = z_in0 + rz, yi = y_in0 + ry;
        const bool rowok = zi >= 0 && zi < d.D && yi >= 0 && yi < d.W;
        const float* src = in + ((int64_t)b * d.Cin + c0 + cc) * V + ((int64_t)zi * d.W + yi) * d.H;
        float* dst = brick + cc * G::CS + rz * G::PS + ry * G::RSL;
        for (int x = lane; x < G::RSL; x += 64) {
          const int xi = x_in0 - G::XOFF + x;
          dst[x] = (rowok && xi >= 0 && xi < d.H) ? src[xi] : 0.0f;
        }
      }
    }
    // ---- this pass's weights: ncc*7*NT registers per lane
    float w[PCC][7][NT];
#pragma unroll
    for (int cc = 0; cc < PCC; ++cc)
#pragma unroll
      for (int q = 0; q < 7; ++q)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          w[cc][q][nt] = (cc < ncc) ? wp[(((c0 + cc) * 7 + q) * NT + nt) * 64 + lane] : 0.0f;
    __syncthreads();
    // ---- MFMA sweep
#pragma unroll
    for (int cc = 0; cc < PCC; ++cc) {
      if (cc < ncc) {
#pragma unroll
        for (int q = 0; q < 7; ++q) {
          const float* base = brick + cc * G::CS + qoff[q];
#pragma unroll
          for (int y = 0; y < PW; ++y)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              const float a = base[y * S * G::RSL + x * 16 * S];
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                acc[y][x][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[cc][q][nt], a, acc[y][x][nt], 0, 0, 0);
            }
        }
      }
    }
  }
  const int dz = do0 + wave;
  if (dz >= d.Do) return;
#pragma unroll
  for (int y = 0; y < PW; ++y)
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        store_tile(acc[y][x][nt], out, d, b, dz, wo0 + y, ho0 + x * 16 + col, nt, lane, out_layout, slope);
}

// ===========================================================================
// Channels-last (NDHWC) input, operands straight from L1/L2.
// Output brick per block: 4 planes (one per wavefront) x MT=4 rows x 16 voxels.
// K order: tap (27) x channel block cb (16 channels); inside a block lane group
// kq owns channels 4kq..4kq+3 and feeds them to four MFMAs.
// ===========================================================================
constexpr int MT = 4;
constexpr int TD = 4;

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
  const int ho = hq * 16 + col;  // this lane's voxel
  const int CB = (d.Cin + 15) >> 4;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const f32x4 bv = bias_init(bias, nt, lane);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = bv;
  }

  // Activations are fetched with buffer loads whose resource starts at the (-1,-1,-1) corner of
  // this wavefront's input window (it may lie before the tensor: never dereferenced there).  A tap
  // that falls into the conv's zero padding gets an out-of-range offset, for which the hardware
  // returns 0 — no branch, no select, so the loads stay unconditional and pipeline (guide T8).
  const int zi0 = dz * STRIDE - 1, yw0 = wo0 * STRIDE - 1, xh0 = hq * 16 * STRIDE - 1;
  const int64_t inb = (int64_t)b * d.D * d.W * d.H * d.Cin;
  const float* wbase = in + inb + (((int64_t)zi0 * d.W + yw0) * d.H + xh0) * d.Cin;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wbase), (short)0, 0x7fffffff, 0x00020000);
  constexpr unsigned OOR = 0x80000000u;
  const unsigned lvoff = (unsigned)(col * STRIDE * d.Cin * 4 + kq * 16);
  unsigned nvmask[MT];  // bit tap CLEAR = that tap of this lane's voxel in tile mt is inside the tensor
  {
    const int xi0 = ho * STRIDE - 1;
    unsigned zx = 0u;  // bit (tz*3+tx)
#pragma unroll
    for (int tz = 0; tz < 3; ++tz)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx)
        if (ho < d.Ho && zi0 + tz >= 0 && zi0 + tz < d.D && xi0 + tx >= 0 && xi0 + tx < d.H)
          zx |= 1u << (tz * 3 + tx);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int yi0 = (wo0 + mt) * STRIDE - 1;
      unsigned m = 0u;
#pragma unroll
      for (int tz = 0; tz < 3; ++tz)
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
          for (int tx = 0; tx < 3; ++tx)
            if ((wo0 + mt < d.Wo) && yi0 + ty >= 0 && yi0 + ty < d.W && ((zx >> (tz * 3 + tx)) & 1u))
              m |= 1u << ((tz * 3 + ty) * 3 + tx);
      nvmask[mt] = ~m;
    }
  }
  const unsigned row_bytes = (unsigned)(STRIDE * d.H * d.Cin * 4);  // next tile (output row) of the wave

  // One k-step = (tap, 16-channel block): NT weight fragments + MT activation fragments, 4 MFMAs each.
  // Steps are software-pipelined one ahead: the loads of step s+1 are in flight while step s
  // runs on the matrix pipe.
  const int NS = 27 * CB;
  auto load_step = [&](int s, float4 (&a)[MT], float4 (&bw)[NT]) {
    const int tap = s / CB, cb = s - tap * CB;  // wave-uniform
    const int tz = tap / 9, ty = (tap / 3) % 3, tx = tap % 3;
    const unsigned soff = (unsigned)(((tz * d.W + ty) * d.H + tx) * d.Cin * 4 + cb * 64);
    // branch-free: bit 31 of the offset is set (=> out of range => 0) unless tap and channel are valid
    const unsigned coor = (cb * 16 + kq * 4 < d.Cin) ? 0u : OOR;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bw[nt] = wp[((int64_t)s * NT + nt) * 64 + lane];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const unsigned voff = lvoff | coor | ((nvmask[mt] >> tap) << 31);
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff + mt * row_bytes, 0);
      a[mt] = __builtin_bit_cast(float4, v);
    }
  };
  auto mfma_step = [&](const float4 (&a)[MT], const float4 (&bw)[NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[nt].x, a[mt].x, acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[nt].y, a[mt].y, acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[nt].z, a[mt].z, acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[nt].w, a[mt].w, acc[mt][nt], 0, 0, 0);
      }
    }
  };
  float4 a0[MT], a1[MT], b0[NT], b1[NT];
  // No conditional loads inside the loop (a branch around a load makes hipcc drain vmcnt(0) at the
  // join): the tail re-loads the last step instead, and an odd step count is finished after the loop.
  load_step(0, a0, b0);
  for (int s = 0; s + 1 < NS; s += 2) {
    load_step(s + 1, a1, b1);
    mfma_step(a0, b0);
    load_step(min(s + 2, NS - 1), a0, b0);
    mfma_step(a1, b1);
  }
  if (NS & 1) mfma_step(a0, b0);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      store_tile(acc[mt][nt], out, d, b, dz, wo0 + mt, ho, nt, lane, out_layout, slope);
}

// ---- weight packing -----------------------------------------------------------
// Both layouts put W[cout = nt*16 + (lane&15)][k of lane group lane>>4] in lane order,
// i.e. the MFMA A-operand (rows = couts) of one k-step is one coalesced 256-B read.
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
  if (out_layout == LR_LAYOUT_NDHWC && (reinterpret_cast<uintptr_t>(out) & 15u)) return LR_EALIGN;
  ConvDims d;
  d.B = B; d.Cin = Cin; d.Cout = Cout; d.D = D; d.W = W; d.H = H;
  d.Do = (D - 1) / stride + 1; d.Wo = (W - 1) / stride + 1; d.Ho = (H - 1) / stride + 1;
  hipStream_t st = lr_stream(stream);
  const int NT = Cout / 16;
  const dim3 block(256);
  if (in_layout == LR_LAYOUT_NDHWC) {
    if (Cin % 4) return LR_EUNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(in) & 15u) return LR_EALIGN;
    if ((int64_t)12 * W * H * Cin + 4096 >= 0x7fffffffLL) return LR_EINVAL;  // 32-bit buffer offsets of a 3-plane window
    d.nHq = (d.Ho + 15) / 16; d.nWq = (d.Wo + MT - 1) / MT; d.nDq = (d.Do + TD - 1) / TD;
    const int64_t nblk = (int64_t)B * d.nDq * d.nWq * d.nHq;
    if (nblk > 0x7fffffffLL) return LR_EINVAL;
    const dim3 grid((unsigned)nblk);
    const float4* wt = reinterpret_cast<const float4*>(packed_w);
    if (NT == 1 && stride == 1) hipLaunchKernelGGL((conv3d_cl_kernel<1, 1>), grid, block, 0, st, in, wt, bias, out, d, out_layout, negative_slope);
    else if (NT == 1) hipLaunchKernelGGL((conv3d_cl_kernel<1, 2>), grid, block, 0, st, in, wt, bias, out, d, out_layout, negative_slope);
    else if (stride == 1) hipLaunchKernelGGL((conv3d_cl_kernel<2, 1>), grid, block, 0, st, in, wt, bias, out, d, out_layout, negative_slope);
    else hipLaunchKernelGGL((conv3d_cl_kernel<2, 2>), grid, block, 0, st, in, wt, bias, out, d, out_layout, negative_slope);
  } else if (in_layout == LR_LAYOUT_NCDHW) {
    d.nHq = (d.Ho + PH - 1) / PH; d.nWq = (d.Wo + PW - 1) / PW; d.nDq = (d.Do + PD - 1) / PD;
    const int64_t nblk = (int64_t)B * d.nDq * d.nWq * d.nHq;
    if (nblk > 0x7fffffffLL) return LR_EINVAL;
    const dim3 grid((unsigned)nblk);
    // channels staged per pass: as many as fit 60 KB of LDS (<= PCC, the register budget for weights)
    const size_t cbytes = (size_t)(stride == 1 ? PlanarGeom<1>::CS : PlanarGeom<2>::CS) * sizeof(float);
    int pcc = (int)((60 * 1024) / cbytes);
    if (pcc > PCC) pcc = PCC;
    if (pcc > Cin) pcc = Cin;
    if (pcc < 1) pcc = 1;
    const size_t lds1 = pcc * cbytes, lds2 = pcc * cbytes;
    const int vec4 = (H % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15u) == 0);
    if (NT == 1 && stride == 1) hipLaunchKernelGGL((conv3d_planar_kernel<1, 1>), grid, block, lds1, st, in, packed_w, bias, out, d, out_layout, negative_slope, pcc, vec4);
    else if (NT == 1) hipLaunchKernelGGL((conv3d_planar_kernel<1, 2>), grid, block, lds2, st, in, packed_w, bias, out, d, out_layout, negative_slope, pcc, vec4);
    else if (stride == 1) hipLaunchKernelGGL((conv3d_planar_kernel<2, 1>), grid, block, lds1, st, in, packed_w, bias, out, d, out_layout, negative_slope, pcc, vec4);
    else hipLaunchKernelGGL((conv3d_planar_kernel<2, 2>), grid, block, lds2, st, in, packed_w, bias, out, d, out_layout, negative_slope, pcc, vec4);
  } else {
    return LR_EINVAL;
  }
  return lr_launch_status();
}
