// conv3d_bf16.hip — the bf16 variant of the encoder's stride-2 blocks (BASELINE configs C4/C5: "bf16 convs").
// Activations are stored as bf16 channels-last (half the HBM bytes of the fp32 path), weights are rounded to
// bf16 once, products are exact and accumulate in fp32 on v_mfma_f32_16x16x32_bf16 (16x the fp32 MFMA rate), so
// the blocks that are matrix-pipe-bound in fp32 become HBM/L2-bound here.
//
//   conv3d_cl_bf16_kernel     blocks 1..5 (bf16 channels-last in, bf16 channels-last or fp32 NCDHW out)
//   conv0_bf16_kernel         the first block: fp32 NCDHW input rounded to bf16 while it is staged
//   conv3d_dgrad_bf16_kernel  data gradient with bf16 gradients (the bf16-gradient training variant)
//   pack / cast helpers
//
//   K order of one MFMA (32 k-values): Cin = 32: one tap, lane group kq owns channels 8kq..8kq+7;
//                                      Cin = 16: two taps, groups 0,1 the first tap, groups 2,3 the second.
//   So every lane's B operand is ONE 16-byte load of 8 consecutive bf16 channels of its voxel.
//
// Replaces (reference file:line): src/liftreg/layers/layers.py:365-369 under torch.cuda.amp-style bf16 storage
// (the reference itself ships fp32 only; the bf16 numerics contract is stated in DESIGN.md §4c and restated on the
// CPU in oracle/ref_ops.py:conv_block_bf16).
#include "lr_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
constexpr int TD = 4;

struct ConvDimsH {
  int B, Cin, Cout, D, W, H, Do, Wo, Ho;
  int nHq, nWq, nDq;
  long long out_bs;   // output elements between batch elements (dense: Cout*Do*Wo*Ho)
};

__device__ __forceinline__ float lrelu(float v, float slope) { return v >= 0.0f ? v : v * slope; }

// round-to-nearest-even fp32 -> bf16 (the hardware conversion of gfx950)
__device__ __forceinline__ u16 to_bf16(float v) {
  const __bf16 h = (__bf16)v;
  return __builtin_bit_cast(u16, h);
}

// lane l holds couts nt*16 + (l>>4)*4 + {0..3} of voxel (l&15)
__device__ __forceinline__ void store_tile_any(const f32x4& acc, void* __restrict__ out, const ConvDimsH& d, int b, int dz,
                                               int wo, int ho, int nt, int lane, int out_layout, float slope) {
  if (wo >= d.Wo || ho >= d.Ho) return;
  const int c0 = nt * 16 + (lane >> 4) * 4;
  f32x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = lrelu(acc[r], slope);
  if (out_layout == LR_LAYOUT_NCDHW) {  // fp32, the reference's layout (last block -> Flatten)
    const int64_t vo = (int64_t)d.Do * d.Wo * d.Ho;
    float* o = reinterpret_cast<float*>(out) + (int64_t)b * d.out_bs + (int64_t)c0 * vo + ((int64_t)dz * d.Wo + wo) * d.Ho + ho;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r * vo] = v[r];
    return;
  }
  const int hp = out_layout == LR_LAYOUT_BF16_NDHWC_HPS ? (ho & 1) * (d.Ho >> 1) + (ho >> 1) : ho;
  u16* o = reinterpret_cast<u16*>(out) + (int64_t)b * d.out_bs + (((int64_t)dz * d.Wo + wo) * d.Ho + hp) * d.Cout + c0;
  const unsigned lo = (unsigned)to_bf16(v[0]) | ((unsigned)to_bf16(v[1]) << 16);
  const unsigned hi = (unsigned)to_bf16(v[2]) | ((unsigned)to_bf16(v[3]) << 16);
  *reinterpret_cast<uint2*>(o) = make_uint2(lo, hi);
}

// CIN32: Cin == 32 (else 16).  PS: input rows parity-split along H ([parity][H/2][Cin]).
template <int NT, bool CIN32, bool PS, int MT>
__global__ __launch_bounds__(256) void conv3d_cl_bf16_kernel(const u16* __restrict__ in, const u32x4* __restrict__ wp,
                                                             const float* __restrict__ bias, void* __restrict__ out,
                                                             ConvDimsH d, int out_layout, float slope) {
  constexpr int CIN = CIN32 ? 32 : 16, VB = CIN * 2;  // bytes per voxel
  const unsigned lb = lr_xcd_remap(blockIdx.x, gridDim.x);
  const int hq = lb % d.nHq, wq = (lb / d.nHq) % d.nWq, dq = (lb / d.nHq / d.nWq) % d.nDq;
  const int b = lb / d.nHq / d.nWq / d.nDq;
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
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = bias[nt * 16 + kq * 4 + r];
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = bv;
  }

  // window origin = the (-1,-1,-1) corner of this wavefront's input window; padding taps get an out-of-range
  // offset and read 0 (unconditional, branch-free loads)
  const int zi0 = dz * 2 - 1, yw0 = wo0 * 2 - 1;
  const int xh0 = PS ? hq * 16 - 1 : hq * 32 - 1;
  const int64_t inb = (int64_t)b * d.D * d.W * d.H * CIN;
  const u16* wbase = in + inb + ((int64_t)zi0 * d.W + yw0) * d.H * CIN + (int64_t)xh0 * CIN;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(wbase), (short)0, 0x7fffffff, 0x00020000);
  const int half_h = (d.H + 1) >> 1;
  const unsigned lvoff = (unsigned)(col * (PS ? 1 : 2) * VB + (CIN32 ? kq : (kq & 1)) * 16);
  unsigned nvmask[MT];  // bit tap CLEAR = tap inside the tensor (bits 27.. stay set: the padding tap of Cin=16)
  {
    // valid(tz,ty,tx) = z[tz] & y[ty] & x[tx]: three 3-bit axis masks spread to the 27 tap bits with shifts
    // (the taps are bit (tz*3+ty)*3+tx); a dozen ALU ops per tile instead of 27 compares
    const int xi0 = ho * 2 - 1;
    unsigned xm = 0u, zm = 0u;
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) {
      if (ho < d.Ho && xi0 + t3 >= 0 && xi0 + t3 < d.H) xm |= 1u << t3;
      if (zi0 + t3 >= 0 && zi0 + t3 < d.D) zm |= 0x1ffu << (9 * t3);
    }
    const unsigned x27 = (xm | (xm << 3) | (xm << 6)) * 0x40201u & zm;  // x pattern in all 9 (tz,ty) triples, gated by z
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int yi0 = (wo0 + mt) * 2 - 1;
      unsigned ym = 0u;
#pragma unroll
      for (int t3 = 0; t3 < 3; ++t3)
        if ((wo0 + mt < d.Wo) && yi0 + t3 >= 0 && yi0 + t3 < d.W) ym |= 0x7u << (3 * t3);
      nvmask[mt] = ~(x27 & (ym * 0x40201u));
    }
  }
  const unsigned row_bytes = (unsigned)(2 * d.H * VB);
  auto tap_off = [&](int tap) -> unsigned {  // byte offset of a tap inside the window (wave-uniform)
    const int tz = tap / 9, ty = (tap / 3) % 3, tx = tap % 3;
    const int xs = PS ? (tx == 1 ? 1 : half_h + (tx >> 1)) : tx;
    return (unsigned)((((tz * d.W + ty) * d.H) + xs) * VB);
  };

  constexpr int NS = CIN32 ? 27 : 14;
  auto load_step = [&](int s, u32x4 (&a)[MT], u32x4 (&bw)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bw[nt] = wp[((int64_t)s * NT + nt) * 64 + lane];
    unsigned toff;
    int mytap;
    if (CIN32) {
      toff = tap_off(s);
      mytap = s;
    } else {
      const unsigned ta = tap_off(2 * s), tb = tap_off(min(2 * s + 1, 26));
      toff = (kq >> 1) ? tb : ta;
      mytap = 2 * s + (kq >> 1);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const unsigned voff = (lvoff + toff) | ((nvmask[mt] >> mytap) << 31);
      a[mt] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, mt * row_bytes, 0);
    }
  };
  auto mfma_step = [&](const u32x4 (&a)[MT], const u32x4 (&bw)[NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bw[nt]),
                                                              __builtin_bit_cast(bf16x8, a[mt]), acc[mt][nt], 0, 0, 0);
  };
  u32x4 a0[MT], a1[MT], b0[NT], b1[NT];
  if constexpr (MT <= 4) {
    // loads two steps ahead of the MFMAs (three register sets), as in the fp32 kernel; NS = 27 | 14
    u32x4 a2[MT], b2[NT];
    load_step(0, a0, b0);
    load_step(1, a1, b1);
    for (int s = 0; s < NS; s += 3) {
      load_step(min(s + 2, NS - 1), a2, b2);
      mfma_step(a0, b0);
      load_step(min(s + 3, NS - 1), a0, b0);
      if (s + 1 < NS) mfma_step(a1, b1);
      load_step(min(s + 4, NS - 1), a1, b1);
      if (s + 2 < NS) mfma_step(a2, b2);
    }
  } else {  // 8 rows per wave: a third register set would cost occupancy (measured slower) — one step ahead
    load_step(0, a0, b0);
    for (int s = 0; s + 1 < NS; s += 2) {
      load_step(s + 1, a1, b1);
      mfma_step(a0, b0);
      load_step(min(s + 2, NS - 1), a0, b0);
      mfma_step(a1, b1);
    }
    if (NS & 1) mfma_step(a0, b0);
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) store_tile_any(acc[mt][nt], out, d, b, dz, wo0 + mt, ho, nt, lane, out_layout, slope);
}

// ===========================================================================================================
// Row-major variant for Cin = 16 on parity-split rows (block 1, the largest stride-2 block).  The tap-major kernel
// above re-fetches an input row for (mt,ty=2) and later for (mt+1,ty=0); with a thousand waves per XCD that second
// fetch misses the 4 MB L2, and the kernel ran at the fabric's rate on 1.65x the input in miss traffic (PMC).  Here a
// row is loaded once (two 16-byte loads per lane: [odd half @tx=0 | even half @tx=1] and [odd half @tx=2 | nothing])
// and feeds its one or two (mt,ty) uses at once: MFMA 1 = taps (tx0,tx1), MFMA 2 = tap tx2 and 16 zero channels
// (2 instead of 1.5 MFMAs per use: the bf16 matrix pipe is 14 % busy).  The six weight fragments of a (tz,ty) are
// picked lane-wise out of the tap-pair packing (tap T, channel half h of cout c sits at fragment T/2, lane
// ((T&1)*2+h)*16+c); the current tz's stay in registers, the next tz's are fetched under the last rows.
template <int NT, int MT, bool CIN32>
__global__ __launch_bounds__(256, 3) void conv3d_cl_rows_bf16_kernel(const u16* __restrict__ in, const u32x4* __restrict__ wp,
                                                                     const float* __restrict__ bias, void* __restrict__ out,
                                                                     ConvDimsH d, int out_layout, float slope) {
  constexpr int CIN = CIN32 ? 32 : 16, VB = CIN * 2;  // bytes per voxel
  constexpr int NL = CIN32 ? 3 : 2;                    // 16-byte loads per lane and row = MFMAs per (mt, ty) use
  const unsigned lb = lr_xcd_remap(blockIdx.x, gridDim.x);
  const int hq = lb % d.nHq, wq = (lb / d.nHq) % d.nWq, dq = (lb / d.nHq / d.nWq) % d.nDq;
  const int b = lb / d.nHq / d.nWq / d.nDq;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int dz = dq * TD + wave;
  if (dz >= d.Do) return;
  const int dD = __builtin_amdgcn_readfirstlane(d.D), dW = __builtin_amdgcn_readfirstlane(d.W),
            dH = __builtin_amdgcn_readfirstlane(d.H), dHo = __builtin_amdgcn_readfirstlane(d.Ho);
  const int wo0 = wq * MT;
  const int col = lane & 15, kq = lane >> 4;
  const int ho = hq * 16 + col;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = bias[nt * 16 + kq * 4 + r];
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = bv;
  }
  const int zi0 = dz * 2 - 1, yw0 = wo0 * 2 - 1, xh0 = hq * 16 - 1;
  const int64_t inb = (int64_t)b * dD * dW * dH * CIN;
  const u16* wbase = in + inb + ((int64_t)zi0 * dW + yw0) * dH * CIN + (int64_t)xh0 * CIN;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(wbase), (short)0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_null =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(wbase), (short)0, 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(wp), (short)0, 0x7fffffff, 0x00020000);
  constexpr unsigned OOR = 0x80000000u;
  const int half_h = (dH + 1) >> 1;
  // per-lane offsets inside a row: load 1 = tx0 (lanes kq<2: odd half, index ho-1) | tx1 (kq>=2: even half, index ho),
  // load 2 = tx2 (kq<2: odd half, index ho) | nothing; xh0 carries the -1
  // (Cin = 32: one MFMA per tap, a lane holds channels 8kq..8kq+7 of its voxel: three loads tx0, tx1, tx2)
  unsigned vl[NL];
  {
    const int xi0 = ho * 2 - 1;
    const bool in_tile = ho < dHo;
    const bool ok0 = in_tile && xi0 >= 0 && xi0 < dH, ok1 = in_tile && xi0 + 1 < dH, ok2 = in_tile && xi0 + 2 < dH;
    if constexpr (CIN32) {
      const unsigned lv = (unsigned)(col * VB + kq * 16);
      vl[0] = (lv + (unsigned)half_h * VB) | (ok0 ? 0u : OOR);
      vl[1] = (lv + 1u * VB) | (ok1 ? 0u : OOR);
      vl[2] = (lv + (unsigned)(half_h + 1) * VB) | (ok2 ? 0u : OOR);
    } else {
      const unsigned lv = (unsigned)(col * VB + (kq & 1) * 16);
      vl[0] = (kq >> 1) ? (lv + 1u * VB) | (ok1 ? 0u : OOR) : (lv + (unsigned)half_h * VB) | (ok0 ? 0u : OOR);
      vl[1] = (kq >> 1) ? OOR : (lv + (unsigned)(half_h + 1) * VB) | (ok2 ? 0u : OOR);
    }
  }
  constexpr int NR = 2 * MT + 1;
  unsigned okmask = 0u;  // bit tz*NR+r SET = input row (zi0+tz, yw0+r) exists; NR*3 <= 51: two words
  unsigned okmask_hi = 0u;
#pragma unroll
  for (int q = 0; q < 3 * NR; ++q) {
    const int zi = zi0 + q / NR, yi = yw0 + q % NR;
    const unsigned bit = (unsigned)(zi >= 0) & (unsigned)(zi < dD) & (unsigned)(yi >= 0) & (unsigned)(yi < dW);
    if (q < 32) okmask |= bit << q;
    else okmask_hi |= bit << (q - 32);
  }
  // weight fragment offsets of this lane: MFMA 1 of (tz,ty) = taps T0 = (tz*3+ty)*3 (lanes kq<2) and T0+1 (kq>=2);
  // MFMA 2 = tap T0+2 for kq<2 (the other lanes multiply zeros: any fragment)
  auto woff = [&](int T, int nt) -> unsigned {
    return (unsigned)(((((T >> 1) * NT + nt) * 64) + (((T & 1) << 1) | (kq & 1)) * 16 + col) * 16);
  };
  u32x4 w[3][NL][NT];  // [ty][mfma][nt] of the current tz
  auto load_w = [&](int tz, int ty) {
    const int T0 = (tz * 3 + ty) * 3;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if constexpr (CIN32) {  // fragment of tap T: packed[(T*NT + nt)*64 + lane]
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
          w[ty][tx][nt] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (unsigned)lane * 16u, (unsigned)(((T0 + tx) * NT + nt) * 1024), 0);
      } else {
        const unsigned o1 = (kq >> 1) ? woff(T0 + 1, nt) : woff(T0, nt);
        w[ty][0][nt] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, o1, 0, 0);
        w[ty][1][nt] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, woff(T0 + 2, nt), 0, 0);
      }
    }
  };
  auto load_row = [&](int q, u32x4 (&a)[NL]) {
    const int tz = q / NR, r = q - tz * NR;
    const bool ok = ((q < 32 ? okmask >> q : okmask_hi >> (q - 32)) & 1u) != 0u;  // wave-uniform, scalar select
    const unsigned soff = (unsigned)((tz * dW + r) * dH) * VB;
#pragma unroll
    for (int j = 0; j < NL; ++j) a[j] = __builtin_amdgcn_raw_buffer_load_b128(ok ? rsrc : rsrc_null, vl[j], soff, 0);
  };
  auto use = [&](int mt, int ty, const u32x4 (&a)[NL]) {
#pragma unroll
    for (int j = 0; j < NL; ++j)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[ty][j][nt]),
                                                              __builtin_bit_cast(bf16x8, a[j]), acc[mt][nt], 0, 0, 0);
  };
#ifndef LR_BF16_ROWS_AHEAD
#define LR_BF16_ROWS_AHEAD 5
#endif
  constexpr int AH = CIN32 ? 2 : LR_BF16_ROWS_AHEAD, NSET = AH + 1;
  u32x4 rows[NSET][NL];  // loads run AH rows ahead: a row is only 2-4 MFMAs (64-128 cycles) of work
  load_w(0, 0);
  load_w(0, 1);
  load_w(0, 2);
#pragma unroll
  for (int q = 0; q < AH; ++q) load_row(q, rows[q]);
#pragma unroll
  for (int q = 0; q < 3 * NR; ++q) {
    const int tz = q / NR, r = q % NR;
    if (q + AH < 3 * NR) load_row(q + AH, rows[(q + AH) % NSET]);
    __builtin_amdgcn_sched_barrier(0);  // keep the loads ahead: the scheduler otherwise sinks them to their use
    if (r & 1) {
      use(r >> 1, 1, rows[q % NSET]);
    } else {
      if (r >= 2) use((r >> 1) - 1, 2, rows[q % NSET]);
      if (r < 2 * MT) use(r >> 1, 0, rows[q % NSET]);
    }
    if (tz < 2) {
      if (r == 2 * MT - 2) load_w(tz + 1, 0);
      if (r == 2 * MT - 1) load_w(tz + 1, 1);
      if (r == 2 * MT) load_w(tz + 1, 2);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) store_tile_any(acc[mt][nt], out, d, b, dz, wo0 + mt, ho, nt, lane, out_layout, slope);
}

// ===========================================================================================================
// The 16 -> 32 stride-2 block (block 1 of the bf16 encoder, the biggest bf16 kernel) as a z-MARCH: a block owns a
// column of MZ_TY x MZ_TX outputs and walks down the output planes; every input plane of the column's (2TY+1) x
// (2TX+1) region is read from HBM ONCE, two planes per step, by 16-byte bounds-checked buffer loads issued a whole
// step ahead, and parked in a 5-plane LDS ring in the parity-split order of the input rows ([17 odd | 16 even]
// 32-byte records per row: the lanes of one MFMA operand read consecutive records = conflict-free ds_read_b128).
// The row kernel above re-reads every plane from L2/HBM for 1.5 output planes and every row for 9/8 tiles.
//   wave w: cout tile nt = w & 1 (its 14 weight fragments stay in registers for the whole column), output rows
//   2(w>>1), 2(w>>1)+1; K order = the packing's tap pairs (2kb, 2kb+1), fp32 accumulation on top of the bias.
// One barrier per step: step oz reads planes 2oz-1..2oz+1 (slots k, k+1, k+2) and fills planes 2oz+2, 2oz+3
// (slots k+3, k+4) after its MFMAs.
constexpr int MZ_TX = 16, MZ_NO = MZ_TX + 1, MZ_NE = MZ_TX, MZ_ROWB = (MZ_NO + MZ_NE) * 32, MZ_NSLOT = 5;   // row: 1056 bytes
template <int TY>   // output rows of a column: 4 (256 threads, 3 blocks per CU) | 8 (512 threads, one block per CU: half the row halo)
struct MZ {
  static constexpr int NTHR = 64 * TY, RY = 2 * TY + 1, PLB = RY * MZ_ROWB, NCH = PLB / 16;   // TY = 4: 9504 bytes, 594 chunks
  static constexpr int NIT = (2 * NCH + NTHR - 1) / NTHR;                                      // 16-byte chunks per thread and step
  static constexpr int DUMP = MZ_NSLOT * PLB, LDSB = DUMP + (NIT * NTHR - 2 * NCH) * 16;
};

template <int TY>
__global__ __launch_bounds__(64 * TY, TY == 4 ? 3 : 1) void conv3d_march_s2_bf16_kernel(const u16* __restrict__ in, const u32x4* __restrict__ wp,
                                                                      const float* __restrict__ bias, void* __restrict__ out,
                                                                      ConvDimsH d, int zc, int out_layout, float slope) {
  using G = MZ<TY>;
  constexpr int MZ_NIT = G::NIT, MZ_NCH = G::NCH, MZ_PLB = G::PLB, MZ_DUMP = G::DUMP, NTHR = G::NTHR;
  __shared__ __attribute__((aligned(16))) unsigned char lds[G::LDSB];
  const unsigned lb = lr_xcd_remap(blockIdx.x, gridDim.x);
  const int hq = lb % d.nHq, wq = (lb / d.nHq) % d.nWq, dq = (lb / d.nHq / d.nWq) % d.nDq;
  const int b = lb / d.nHq / d.nWq / d.nDq;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kq = lane >> 4;
  const int nt = wave & 1, mp = wave >> 1;
  const int dD = __builtin_amdgcn_readfirstlane(d.D), dW = __builtin_amdgcn_readfirstlane(d.W),
            dH = __builtin_amdgcn_readfirstlane(d.H);
  const int ox0 = hq * MZ_TX, oy0 = wq * TY, oz0 = dq * zc;
  const int oz1 = min(oz0 + zc, d.Do);
  const int half_h = dH >> 1;
  const unsigned plane_b = (unsigned)dW * (unsigned)dH * 32u;

  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(wp), (short)0, 0x7fffffff, 0x00020000);
  // the batch element's bytes exactly: planes past the end read 0
  const u16* inb = in + (int64_t)b * dD * dW * dH * 16;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(inb), (short)0, (int)((unsigned)dD * plane_b), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_null = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(inb), (short)0, 0, 0x00020000);
  constexpr unsigned OOR = 0x80000000u;

  u32x4 w[14];
#pragma unroll
  for (int kb = 0; kb < 14; ++kb) w[kb] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (unsigned)lane * 16u, (unsigned)((kb * 2 + nt) * 1024), 0);
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bias[nt * 16 + kq * 4 + r];
  }

  // LDS byte offset (without the plane slot) of this lane's operand of k-block kb, output row 2mp (row 2mp+1: + 2 rows)
  unsigned toff[14];
#pragma unroll
  for (int kb = 0; kb < 14; ++kb) {
    const int Ta = 2 * kb, Tb = 2 * kb + 1 > 26 ? 26 : 2 * kb + 1;
    auto tap_off = [&](int T) -> unsigned {
      const int ty = (T / 3) % 3, tx = T % 3;
      return (unsigned)(ty * MZ_ROWB + (tx == 1 ? MZ_NO * 32 : 0) + (tx == 2 ? 32 : 0));
    };
    toff[kb] = ((kq >> 1) ? tap_off(Tb) : tap_off(Ta)) + (unsigned)(4 * mp * MZ_ROWB + col * 32 + (kq & 1) * 16);
  }

  // staging items: chunk c = tid + NTHR j of the 2 x NCH 16-byte chunks of a plane pair
  unsigned goff[MZ_NIT], loff[MZ_NIT];
  bool p2 = false;
#pragma unroll
  for (int j = 0; j < MZ_NIT; ++j) {
    const int c = tid + NTHR * j;
    const bool live = c < 2 * MZ_NCH;
    const int p = c >= MZ_NCH ? 1 : 0;
    if (NTHR * j < MZ_NCH && NTHR * (j + 1) > MZ_NCH) p2 = p != 0;   // the one item that straddles the two planes
    const int ci = c - p * MZ_NCH;
    const int row = ci / (2 * (MZ_NO + MZ_NE)), cc = ci - row * 2 * (MZ_NO + MZ_NE);
    const bool odd = cc < 2 * MZ_NO;
    const int rec = odd ? (cc >> 1) : ((cc - 2 * MZ_NO) >> 1);
    const int y = 2 * oy0 - 1 + row;
    const int xi = odd ? ox0 - 1 + rec : ox0 + rec;
    const bool ok = live && y >= 0 && y < dW && xi >= 0 && xi < half_h;
    const unsigned mrec = (unsigned)(odd ? half_h + xi : xi);
    goff[j] = ok ? ((unsigned)(p * dW + y) * (unsigned)dH + mrec) * 32u + (unsigned)(cc & 1) * 16u : OOR;
    loff[j] = live ? (unsigned)ci * 16u : (unsigned)(MZ_DUMP + (c - 2 * MZ_NCH) * 16);
  }

  u32x4 st[MZ_NIT];
  auto stage_load = [&](int zbase, bool real) {  // planes zbase, zbase+1 (zbase >= 0 when real)
    const unsigned zo = (unsigned)zbase * plane_b;
#pragma unroll
    for (int j = 0; j < MZ_NIT; ++j) st[j] = __builtin_amdgcn_raw_buffer_load_b128(real ? rsrc : rsrc_null, goff[j] + zo, 0, 0);
  };
  auto stage_write = [&](unsigned s0b, unsigned s1b) {  // LDS bases of the two planes' slots
#pragma unroll
    for (int j = 0; j < MZ_NIT; ++j) {
      const bool live = tid + NTHR * j < 2 * MZ_NCH;
      // item j covers chunks [NTHR j, NTHR (j + 1)): all of plane 0 | straddling the planes (one item) | plane 1 (the last one partly dead)
      const unsigned sb = NTHR * (j + 1) <= MZ_NCH ? s0b : NTHR * j < MZ_NCH ? (p2 ? s1b : s0b) : (live ? s1b : 0u);
      *reinterpret_cast<u32x4*>(lds + sb + loff[j]) = st[j];
    }
  };

  // prologue: planes 2oz0-2 (unused), 2oz0-1 -> slots 0, 1; planes 2oz0, 2oz0+1 -> slots 2, 3
  stage_load(2 * oz0 - 2, oz0 > 0);
  stage_write(0u, (unsigned)MZ_PLB);
  stage_load(2 * oz0, true);
  stage_write(2u * MZ_PLB, 3u * MZ_PLB);
  __syncthreads();
  int k = 1;  // slot of plane 2oz-1
  for (int oz = oz0; oz < oz1; ++oz) {
    stage_load(2 * oz + 2, oz + 1 < oz1);
    __builtin_amdgcn_sched_barrier(0);
    const int k1 = k + 1 >= MZ_NSLOT ? k + 1 - MZ_NSLOT : k + 1, k2 = k + 2 >= MZ_NSLOT ? k + 2 - MZ_NSLOT : k + 2;
    const int k3 = k + 3 >= MZ_NSLOT ? k + 3 - MZ_NSLOT : k + 3, k4 = k + 4 >= MZ_NSLOT ? k + 4 - MZ_NSLOT : k + 4;
    const unsigned sb0 = (unsigned)k * MZ_PLB, sb1 = (unsigned)k1 * MZ_PLB, sb2 = (unsigned)k2 * MZ_PLB;
    f32x4 acc0 = bv, acc1 = bv;
#pragma unroll
    for (int kb = 0; kb < 14; ++kb) {
      // taps 2kb, 2kb+1: tz = 0 for kb < 4, 1 for 5..8, 2 for 9..13; kb = 4 straddles (tap 8 | tap 9)
      const unsigned sb = kb < 4 ? sb0 : kb == 4 ? ((kq >> 1) ? sb1 : sb0) : kb < 9 ? sb1 : sb2;
      const unsigned char* a = lds + sb + toff[kb];
      const u32x4 x0 = *reinterpret_cast<const u32x4*>(a);
      const u32x4 x1 = *reinterpret_cast<const u32x4*>(a + 2 * MZ_ROWB);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[kb]), __builtin_bit_cast(bf16x8, x0), acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[kb]), __builtin_bit_cast(bf16x8, x1), acc1, 0, 0, 0);
    }
    store_tile_any(acc0, out, d, b, oz, oy0 + 2 * mp, ox0 + col, nt, lane, out_layout, slope);
    store_tile_any(acc1, out, d, b, oz, oy0 + 2 * mp + 1, ox0 + col, nt, lane, out_layout, slope);
    __builtin_amdgcn_sched_barrier(0);
    stage_write((unsigned)k3 * MZ_PLB, (unsigned)k4 * MZ_PLB);
    __syncthreads();
    k = k2;
  }
}

// The 32 -> 32 stride-2 blocks the same way (Cin = 32: one tap per MFMA, 64-byte records).  Column of 4 x 8 outputs = two
// M-tiles of 2 rows x 8 voxels; wave w: cout tile w & 1, M-tile w >> 1, all 27 taps (27 weight fragments in registers).
// Records of 64 bytes put the four 16-byte channel chunks of voxels 4 apart into the same banks: chunk k of a record in
// region row r is stored at position k ^ (r & 3) (found by enumeration: every ds_read_b128 lane group then covers all 64
// banks exactly once, for both row parities and the +1 shift of the third tap column).
constexpr int M3_TY = 4, M3_TX = 8;
constexpr int M3_RY = 2 * M3_TY + 1, M3_NO = M3_TX + 1, M3_NE = M3_TX;
constexpr int M3_ROWB = (M3_NO + M3_NE) * 64, M3_PLB = M3_RY * M3_ROWB, M3_NCH = M3_PLB / 16;  // 1088, 9792, 612
constexpr int M3_NSLOT = 5, M3_NIT = (2 * M3_NCH + 255) / 256;
constexpr int M3_DUMP = M3_NSLOT * M3_PLB, M3_LDSB = M3_DUMP + (M3_NIT * 256 - 2 * M3_NCH) * 16;
static_assert(M3_NIT == 5 && 2 * 256 < M3_NCH && 3 * 256 >= M3_NCH && 4 * 256 < 2 * M3_NCH, "item -> plane map below");

#ifndef LR_M3_BLOCKS
#define LR_M3_BLOCKS 2
#endif
__global__ __launch_bounds__(256, LR_M3_BLOCKS) void conv3d_march_s2_c32_bf16_kernel(const u16* __restrict__ in, const u32x4* __restrict__ wp,
                                                                          const float* __restrict__ bias, void* __restrict__ out,
                                                                          ConvDimsH d, int zc, int out_layout, float slope) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[M3_LDSB];
  const unsigned lb = lr_xcd_remap(blockIdx.x, gridDim.x);
  const int hq = lb % d.nHq, wq = (lb / d.nHq) % d.nWq, dq = (lb / d.nHq / d.nWq) % d.nDq;
  const int b = lb / d.nHq / d.nWq / d.nDq;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kq = lane >> 4;
  const int nt = wave & 1, mt = wave >> 1;
  const int rl = col >> 3, xl = col & 7;   // this lane's output row (of the M-tile's two) and voxel
  const int dD = __builtin_amdgcn_readfirstlane(d.D), dW = __builtin_amdgcn_readfirstlane(d.W),
            dH = __builtin_amdgcn_readfirstlane(d.H);
  const int ox0 = hq * M3_TX, oy0 = wq * M3_TY, oz0 = dq * zc;
  const int oz1 = min(oz0 + zc, d.Do);
  const int half_h = dH >> 1;
  const unsigned plane_b = (unsigned)dW * (unsigned)dH * 64u;

  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(wp), (short)0, 0x7fffffff, 0x00020000);
  const u16* inb = in + (int64_t)b * dD * dW * dH * 32;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(inb), (short)0, (int)((unsigned)dD * plane_b), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_null = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(inb), (short)0, 0, 0x00020000);
  constexpr unsigned OOR = 0x80000000u;

  u32x4 w[27];
#pragma unroll
  for (int T = 0; T < 27; ++T) w[T] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (unsigned)lane * 16u, (unsigned)((T * 2 + nt) * 1024), 0);
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bias[nt * 16 + kq * 4 + r];
  }

  // LDS byte offset (without the plane slot) of this lane's operand of tap (ty, tx)
  // (tx adds a lane-invariant constant: 0 | the even half | one record — an immediate of the ds_read)
  unsigned toff[3];
#pragma unroll
  for (int ty = 0; ty < 3; ++ty) {
    const int row = 4 * mt + 2 * rl + ty;
    toff[ty] = (unsigned)(row * M3_ROWB + xl * 64 + ((kq ^ (row & 3)) * 16));
  }

  // staging items: chunk c = tid + 256 j of the 2 x 612 16-byte chunks of a plane pair (j = 0,1: plane 0; 2: both; 3,4: plane 1)
  unsigned goff[M3_NIT], loff[M3_NIT];
  bool p2 = false;
#pragma unroll
  for (int j = 0; j < M3_NIT; ++j) {
    const int c = tid + 256 * j;
    const bool live = c < 2 * M3_NCH;
    const int p = c >= M3_NCH ? 1 : 0;
    if (j == 2) p2 = p != 0;
    const int ci = c - p * M3_NCH;
    const int row = ci / (4 * (M3_NO + M3_NE)), cc = ci - row * 4 * (M3_NO + M3_NE);
    const bool odd = cc < 4 * M3_NO;
    const int rec = odd ? (cc >> 2) : ((cc - 4 * M3_NO) >> 2);
    const int k = cc & 3;
    const int y = 2 * oy0 - 1 + row;
    const int xi = odd ? ox0 - 1 + rec : ox0 + rec;
    const bool ok = live && y >= 0 && y < dW && xi >= 0 && xi < half_h;
    const unsigned mrec = (unsigned)(odd ? half_h + xi : xi);
    goff[j] = ok ? ((unsigned)(p * dW + y) * (unsigned)dH + mrec) * 64u + (unsigned)k * 16u : OOR;
    loff[j] = live ? (unsigned)(ci - k + (k ^ (row & 3))) * 16u : (unsigned)(M3_DUMP + (c - 2 * M3_NCH) * 16);
  }

  u32x4 st[M3_NIT];
  auto stage_load = [&](int zbase, bool real) {
    const unsigned zo = (unsigned)zbase * plane_b;
#pragma unroll
    for (int j = 0; j < M3_NIT; ++j) st[j] = __builtin_amdgcn_raw_buffer_load_b128(real ? rsrc : rsrc_null, goff[j] + zo, 0, 0);
  };
  auto stage_write = [&](unsigned s0b, unsigned s1b) {
#pragma unroll
    for (int j = 0; j < M3_NIT; ++j) {
      const bool live = tid + 256 * j < 2 * M3_NCH;
      const unsigned sb = j < 2 ? s0b : j == 2 ? (p2 ? s1b : s0b) : (live ? s1b : 0u);
      *reinterpret_cast<u32x4*>(lds + sb + loff[j]) = st[j];
    }
  };

  stage_load(2 * oz0 - 2, oz0 > 0);
  stage_write(0u, (unsigned)M3_PLB);
  stage_load(2 * oz0, true);
  stage_write(2u * M3_PLB, 3u * M3_PLB);
  __syncthreads();
  int k = 1;  // slot of plane 2oz-1
  const int wo = oy0 + 2 * mt + rl, ho = ox0 + xl;
  for (int oz = oz0; oz < oz1; ++oz) {
    stage_load(2 * oz + 2, oz + 1 < oz1);
    __builtin_amdgcn_sched_barrier(0);
    const int k1 = k + 1 >= M3_NSLOT ? k + 1 - M3_NSLOT : k + 1, k2 = k + 2 >= M3_NSLOT ? k + 2 - M3_NSLOT : k + 2;
    const int k3 = k + 3 >= M3_NSLOT ? k + 3 - M3_NSLOT : k + 3, k4 = k + 4 >= M3_NSLOT ? k + 4 - M3_NSLOT : k + 4;
    const unsigned sb[3] = {(unsigned)k * M3_PLB, (unsigned)k1 * M3_PLB, (unsigned)k2 * M3_PLB};
    f32x4 acc = bv;
#pragma unroll
    for (int T = 0; T < 27; ++T) {
      constexpr int TXO[3] = {0, M3_NO * 64, 64};
      const u32x4 x = *reinterpret_cast<const u32x4*>(lds + sb[T / 9] + toff[(T / 3) % 3] + TXO[T % 3]);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[T]), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
    }
    store_tile_any(acc, out, d, b, oz, wo, ho, nt, lane, out_layout, slope);
    __builtin_amdgcn_sched_barrier(0);
    stage_write((unsigned)k3 * M3_PLB, (unsigned)k4 * M3_PLB);
    __syncthreads();
    k = k2;
  }
}

// ===========================================================================================================
// First block (planar fp32 input, stride 1) on the bf16 MFMA.  Brick = 4 planes (one per wave) x 4 rows x 64
// voxels.  K order: 27 window rows (channel, tz, ty) x 4 columns (tx = 0..2 and a zero-weight 4th) = 108 -> 128 =
// four 16x16x32 MFMAs per 16-voxel tile (the fp32 kernel needs 21 16x16x4 MFMAs = 10x the matrix-pipe time): the
// block turns from MFMA-bound into LDS/HBM-bound.  The input window of a 3-channel pass is rounded to bf16 ONCE
// while it is staged, and it is kept twice in LDS — the second copy shifted by one element — so that the 4
// consecutive bf16 a lane needs from a window row (tx = 0..3 at its voxel) always start on a 4-byte boundary in
// one of the copies: a lane's 8 k-values of an MFMA are two ds_read2_b32 at per-lane row bases + immediate tile
// offsets, with no conversion and no packing in the sweep.  More than 3 input channels (C4: 11 views + CT = 12)
// run as passes over the same accumulators.
constexpr int PH = 64, PW = 4, PD = 4, XOFF = 3;
constexpr int RB = 76;            // bf16 elements per staged window row (72 + the shift, 8-byte multiple)
constexpr int WROWS = 3 * 36;     // window rows per pass: 3 channels x 6 planes x 6 rows
constexpr int COPYB = WROWS * RB; // element offset of the shifted copy

__device__ __forceinline__ unsigned pack2_bf16(float a, float b) { return (unsigned)to_bf16(a) | ((unsigned)to_bf16(b) << 16); }

// SINGLE (Cin <= 3, the model's case): one pass, so a tile's accumulator lives only for its 4 MFMAs and is stored
// at once — a quarter of the registers, twice the resident blocks, stores spread over the sweep.
// HPSOUT: output rows parity-split (compile time: the store address is a per-lane base + a constant per tile).
template <int NT, bool SINGLE, bool HPSOUT>
__global__ __launch_bounds__(256, (SINGLE ? 3 : 2)) void conv0_bf16_kernel(const float* __restrict__ in, const u32x4* __restrict__ wp,
                                                            const float* __restrict__ bias, void* __restrict__ out,
                                                            ConvDimsH d, int out_layout, float slope, int vec4,
                                                            unsigned char* __restrict__ mask_out /* or null: LR_LAYOUT_SIGN4 */) {
  __shared__ __attribute__((aligned(16))) u16 brick[2 * WROWS * RB];
  const unsigned lb = lr_xcd_remap(blockIdx.x, gridDim.x);
  const int hq = lb % d.nHq, wq = (lb / d.nHq) % d.nWq, dq = (lb / d.nHq / d.nWq) % d.nDq;
  const int b = lb / d.nHq / d.nWq / d.nDq;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kq = lane >> 4;
  const int z0 = dq * PD, y0 = wq * PW, x0 = hq * PH;
  const int64_t V = (int64_t)d.D * d.W * d.H;

  // per-lane element offsets of the two window rows of each MFMA: k = m*32 + kq*8 + h*4 + tx -> row m*8 + kq*2 + h.
  // The lane's first window column is XOFF + 16t + col: odd -> read the shifted copy (at column + 1, even again).
  int rbase[4][2];
  {
    const int odd = (XOFF + col) & 1;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int rr = m * 8 + kq * 2 + h;
        if (rr >= 27) rr = 0;  // zero weights: any staged row
        const int c = rr / 9, tz = (rr / 3) % 3, ty = rr % 3;
        rbase[m][h] = (c * 36 + (wave + tz) * 6 + ty) * RB + XOFF + col + (odd ? COPYB + 1 : 0);
      }
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) u16*)brick;  // LDS byte offset
  f32x4 acc[SINGLE ? 1 : PW * 4][NT];
  f32x4 bvec[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = bias[nt * 16 + kq * 4 + r];
    }
    bvec[nt] = bv;
#pragma unroll
    for (int t = 0; t < (SINGLE ? 1 : PW * 4); ++t) acc[t][nt] = bv;
  }
  const int dz = z0 + wave;
  // store of tile (r, t): bf16 rows of Cout channels; this lane's voxel is x0 + 16t + col.  x0 is a multiple of 64,
  // so in a parity-split row the lane sits at (col&1)*(H/2) + x0/2 + col/2 and a tile advances 8 positions.
  const int hp_lane = HPSOUT ? (col & 1) * (d.H >> 1) + (x0 >> 1) + (col >> 1) : x0 + col;
  u16* const out_lane = reinterpret_cast<u16*>(out) + (int64_t)b * d.out_bs + (((int64_t)dz * d.W + y0) * d.H + hp_lane) * d.Cout + kq * 4;
  const int64_t row_el = (int64_t)d.H * d.Cout;
  auto store_tile = [&](const f32x4 (&a)[NT], int r, int t) {
    if (dz >= d.D || y0 + r >= d.W || x0 + t * 16 + col >= d.H) return;
    u16* o = out_lane + r * row_el + t * (HPSOUT ? 8 : 16) * d.Cout;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = lrelu(a[nt][q], slope);
      typedef unsigned u32x2nt __attribute__((ext_vector_type(2)));
      const unsigned lo = pack2_bf16(v[0], v[1]), hi = pack2_bf16(v[2], v[3]);
      __builtin_nontemporal_store((u32x2nt){lo, hi}, reinterpret_cast<u32x2nt*>(o + nt * 16));
      if (mask_out) {
        // training forward: the LeakyReLU sign mask of this lane's channel quad (LR_LAYOUT_SIGN4: one byte per (voxel,
        // quad), bit r = "stored bf16 of channel 4q+r > 0") — the next block's data gradient reads 4 bytes per voxel
        // instead of the 32-byte bf16 activation (7.2 GB -> 0.9 GB at C5)
        const unsigned m = ((short)(lo & 0xffffu) > 0 ? 1u : 0u) | ((short)(lo >> 16) > 0 ? 2u : 0u) |
                           ((short)(hi & 0xffffu) > 0 ? 4u : 0u) | ((short)(hi >> 16) > 0 ? 8u : 0u);
        // the four quads of a voxel sit in the four 16-lane rows of the wave: gfx950's row / half swaps bring them into
        // row 0, whose lanes store the voxel's four bytes as ONE dword (64 contiguous bytes per tile instead of 64 byte
        // stores: the byte-store version cost this block's forward +0.5 ms at C5)
        const auto s1 = __builtin_amdgcn_permlane32_swap(m, m, false, false);     // [1]: rows 0,1 <- rows 2,3 of m
        const unsigned xa = s1[0], xb = s1[1];
        const auto s2 = __builtin_amdgcn_permlane16_swap(xa, xa, false, false);   // [1]: row 0 <- row 1 of m
        const auto s3 = __builtin_amdgcn_permlane16_swap(xb, xb, false, false);   // [1]: row 0 <- row 3 of m
        const unsigned dw = (m & 0xffu) | (s2[1] & 0xffu) << 8 | (xb & 0xffu) << 16 | (s3[1] & 0xffu) << 24;
        if (kq == 0 && NT == 1)
          *reinterpret_cast<unsigned*>(mask_out + ((((int64_t)b * d.D + dz) * d.W + y0 + r) * d.H + x0 + t * 16 + col) * 4) = dw;
        else if (NT != 1)
          mask_out[((((int64_t)b * d.D + dz) * d.W + y0 + r) * d.H + x0 + t * 16 + col) * (d.Cout >> 2) + nt * 4 + kq] = (unsigned char)m;
      }
    }
  };
  const int npass = (d.Cin + 2) / 3;
  for (int pass = 0; pass < npass; ++pass) {
    const int c0 = pass * 3;
    if (pass) __syncthreads();  // the previous pass is done reading the brick
    if (vec4) {  // 16-byte loads from the aligned window x0-4 .. x0+67 (H % 4 == 0: a float4 is all in or all out)
      const float* xb = in + ((int64_t)b * d.Cin + c0) * V;
      const __amdgpu_buffer_rsrc_t rsrc =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), (short)0, 0x7fffffff, 0x00020000);
      f32x4 st[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int q = it * 256 + tid;
        const int row = q / 18, f4 = q - row * 18;
        const int cc = row / 36, rz = (row / 6) % 6, ry = row % 6;
        const int zi = z0 - 1 + rz, yi = y0 - 1 + ry, xi = x0 - 4 + f4 * 4;
        const bool ok = (int)(q < WROWS * 18) & (int)(c0 + cc < d.Cin) & (int)(zi >= 0) & (int)(zi < d.D) & (int)(yi >= 0) & (int)(yi < d.W) & (int)(xi >= 0) & (int)(xi < d.H);  // bitwise: no exec-mask branches around the loads' address math
        const unsigned voff = ok ? (unsigned)(((int64_t)cc * V + ((int64_t)zi * d.W + yi) * d.H + xi) * 4) : 0x80000000u;
        st[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int q = it * 256 + tid;
        if (q < WROWS * 18) {
          const int row = q / 18, f4 = q - row * 18;
          const unsigned lo = pack2_bf16(st[it][0], st[it][1]), hi = pack2_bf16(st[it][2], st[it][3]);
          u16* a = brick + row * RB + f4 * 4;
          *reinterpret_cast<uint2*>(a) = make_uint2(lo, hi);                    // copy A: elements e..e+3
          u16* sh = a + COPYB + 1;                                              // copy B: the same at e+1..e+4
          sh[0] = (u16)lo;
          *reinterpret_cast<unsigned*>(sh + 1) = (lo >> 16) | (hi << 16);
          sh[3] = (u16)(hi >> 16);
        }
      }
    } else {  // unaligned rows: scalar staging of the same window
      for (int row = wave; row < WROWS; row += 4) {
        const int cc = row / 36, rz = (row / 6) % 6, ry = row % 6;
        const int zi = z0 - 1 + rz, yi = y0 - 1 + ry;
        const bool rowok = c0 + cc < d.Cin && zi >= 0 && zi < d.D && yi >= 0 && yi < d.W;
        const float* src = in + ((int64_t)b * d.Cin + c0 + cc) * V + ((int64_t)zi * d.W + yi) * d.H;
        for (int x = lane; x < 72; x += 64) {
          const int xi = x0 - 4 + x;
          const u16 v = to_bf16((rowok && xi >= 0 && xi < d.H) ? src[xi] : 0.0f);
          brick[row * RB + x] = v;
          brick[COPYB + row * RB + x + 1] = v;
        }
      }
    }
    u32x4 w[4][NT];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) w[m][nt] = wp[(((int64_t)pass * 4 + m) * NT + nt) * 64 + lane];
    __syncthreads();
    // ds_read2_b32 by hand: the 4 bf16 of a window row start on a 4-byte (not 8-byte) boundary, and a
    // 4-byte-aligned ds_read_b64 (what the compiler picks for such a pair) runs several times slower here.
    // The wait takes the tile's 8 results as in/out operands: nothing that uses them can be scheduled above it.
    // (Issuing tile i+1's reads ahead of tile i's MFMAs measured no faster: the sweep is VALU/issue-bound.)
#pragma unroll
    for (int r = 0; r < PW; ++r)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int ti = SINGLE ? 0 : r * 4 + t;
        unsigned long long pr[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3"
                         : "=v"(pr[m][h])
                         : "v"(lds0 + (unsigned)rbase[m][h] * 2u), "n"((r * RB + t * 16) / 2), "n"((r * RB + t * 16) / 2 + 1)
                         : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(pr[0][0]), "+v"(pr[0][1]), "+v"(pr[1][0]), "+v"(pr[1][1]), "+v"(pr[2][0]), "+v"(pr[2][1]),
                       "+v"(pr[3][0]), "+v"(pr[3][1])
                     :
                     : "memory");
        if (SINGLE) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[0][nt] = bvec[nt];
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const u32x4 bv = {(unsigned)pr[m][0], (unsigned)(pr[m][0] >> 32), (unsigned)pr[m][1], (unsigned)(pr[m][1] >> 32)};
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[ti][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[m][nt]),
                                                                  __builtin_bit_cast(bf16x8, bv), acc[ti][nt], 0, 0, 0);
        }
        if (SINGLE) store_tile(acc[0], r, t);
      }
  }
  if (!SINGLE) {
#pragma unroll
    for (int r = 0; r < PW; ++r)
#pragma unroll
      for (int t = 0; t < 4; ++t) store_tile(acc[r * 4 + t], r, t);
  }
}

// packed[((pass*4 + m)*NT + nt)*64 + lane]: k = m*32 + kq*8 + h*4 + tx -> window row rr = m*8 + kq*2 + h =
// (c, tz, ty), W[co = nt*16 + (lane&15)][channel pass*3 + c][tap (tz,ty,tx)], zero for tx = 3 and rr >= 27
__global__ void pack_bf16_planar_kernel(const float* __restrict__ w, u32x4* __restrict__ packed, int Cin, int Cout,
                                        int NT) {
  const int npass = (Cin + 2) / 3;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= npass * 4 * NT * 64) return;
  const int lane = idx & 63, nt = (idx >> 6) % NT, m = (idx >> 6) / NT % 4, pass = (idx >> 6) / NT / 4;
  const int co = nt * 16 + (lane & 15), kq = lane >> 4;
  unsigned r[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    unsigned pair = 0u;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int i = 2 * p + hh, rr = m * 8 + kq * 2 + (i >> 2), tx = i & 3;
      const int ci = pass * 3 + rr / 9, tap = ((rr / 3) % 3) * 9 + (rr % 3) * 3 + tx;
      const float v = (rr < 27 && tx < 3 && ci < Cin && co < Cout) ? w[((int64_t)co * Cin + ci) * 27 + tap] : 0.0f;
      pair |= (unsigned)to_bf16(v) << (16 * hh);
    }
    r[p] = pair;
  }
  packed[idx] = (u32x4){r[0], r[1], r[2], r[3]};
}

// ===========================================================================================================
// Data gradient of a stride-2 block with bf16 gradients (the bf16-gradient training variant): gpre (B,Do,Wo,Ho,32)
// and the result are bf16 plain channels-last, the weights are the bf16-rounded ones the forward multiplied by
// (packed TRANSPOSED with pack_bf16_kernel: rows = the block's input channels, k = its 32 output channels), the
// accumulation is fp32.  Same scheme as conv3d_dgrad_lds_kernel (conv3d_bwd.hip): a block stages the 5x5x17 gpre
// voxels of its tile once in LDS and walks the 8 parity classes of gx; here one 16x16x32 MFMA covers a whole tap,
// so the kernel is bound by its HBM traffic (gpre + the mask source in, gx out), not by the matrix pipe.
// The epilogue multiplies by the producer's LeakyReLU mask (xsave = this block's saved bf16 input) and rounds to bf16.
struct DgDimsH {
  int B, Cx, D, W, H, Do, Wo, Ho, nHq, nWq, nDq, xs_layout;
  float slope;
};

// XSL = the layout of the mask source, a compile-time constant (a run-time switch around the loads makes hipcc drain the
// memory queue at every join): LR_LAYOUT_SIGN4 | LR_LAYOUT_BF16_NDHWC | LR_LAYOUT_BF16_NDHWC_HPS.
// The mask sources of a parity pass (8 * NT small loads per lane) are requested BEFORE the pass's MFMAs and consumed after
// them.  Loaded in the epilogue, next to their use, every one of them was a load -> s_waitcnt vmcnt(0) -> store sequence: a
// full memory latency per output tile with the previous tile's stores drained first — the kernel ran at 28 % of the HBM rate
// (C5: 4.4 ms for 10 GB).  Stores are unconditional bounds-checked buffer stores on the output plane.
template <int NT, int XSL>
__global__ __launch_bounds__(256, 2) void conv3d_dgrad_bf16_kernel(const u16* __restrict__ gpre, const u32x4* __restrict__ wp,
                                                                   u16* __restrict__ gx, const u16* __restrict__ xsave,
                                                                   DgDimsH d) {
  constexpr int CG = 32, VS = CG + 8, NVOX = 5 * 5 * 17, NCH = NVOX * 4;  // 16-byte chunks: 4 per voxel
  constexpr int NIT = (NCH + 255) / 256;
  __shared__ __attribute__((aligned(16))) u16 ts[NVOX * VS];
  unsigned t = lr_xcd_remap(blockIdx.x, gridDim.x);
  const int hq = t % d.nHq; t /= d.nHq;
  const int wq = t % d.nWq; t /= d.nWq;
  const int dq = t % d.nDq;
  const int b = t / d.nDq;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int zq0 = dq * 4, yq0 = wq * 4, xq0 = hq * 16;
  {
    const u16* base = gpre + ((((int64_t)b * d.Do + zq0) * d.Wo + yq0) * d.Ho + xq0) * CG;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(base), (short)0, 0x7fffffff, 0x00020000);
    u32x4 st[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = it * 256 + tid;
      const int vox = q >> 2, c8 = q & 3;
      const int xx = vox % 17, r = vox / 17, yy = r % 5, zz = r / 5;
      const bool ok = (int)(q < NCH) & (int)(zq0 + zz < d.Do) & (int)(yq0 + yy < d.Wo) & (int)(xq0 + xx < d.Ho);  // bitwise: no exec-mask branches around the loads' address math
      const unsigned voff = ok ? (unsigned)(((((zz * d.Wo) + yy) * d.Ho + xx) * CG + c8 * 8) * 2) : 0x80000000u;
      st[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = it * 256 + tid;
      if (q < NCH) *reinterpret_cast<u32x4*>(ts + (q >> 2) * VS + (q & 3) * 8) = st[it];
    }
  }
  __syncthreads();
  const int zq = zq0 + wave;
  const int col = lane & 15, kq = lane >> 4;
  const int xq = xq0 + col;
  const u16* lts = ts + col * VS + kq * 8;
  constexpr bool SIGN4 = XSL == LR_LAYOUT_SIGN4;
  constexpr unsigned OORV = 0x80000000u;
  // mask source of batch element b as one buffer resource (SIGN4: D*W*H*Cx/4 bytes; activation: D*W*H*Cx*2 bytes < 2^31: checked by the launcher)
  const int64_t VX = (int64_t)d.D * d.W * d.H;
  const __amdgpu_buffer_rsrc_t rxs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(SIGN4 ? reinterpret_cast<const u16*>(reinterpret_cast<const unsigned char*>(xsave) + (int64_t)b * VX * (d.Cx >> 2))
                             : xsave + (int64_t)b * VX * d.Cx),
      (short)0, (int)(SIGN4 ? VX * (d.Cx >> 2) : VX * d.Cx * 2), 0x00020000);
  for (int pp = 3; pp >= 0; --pp) {
    const int py = pp & 1, pz = pp >> 1;
    const int z = 2 * zq + pz;
    if (z >= d.D) continue;  // wave-uniform; no barrier below
    // ---- this pass's mask sources, in flight under its MFMAs
    unsigned mk[4][2][NT];
    u32x2 xsv[SIGN4 ? 1 : 4][SIGN4 ? 1 : 2][SIGN4 ? 1 : NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int y = 2 * (yq0 + mt) + py;
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        const int x = 2 * xq + px;
        const int ok = (int)(y < d.W) & (int)(x < d.H);
        const unsigned vox = (unsigned)((z * d.W + y) * d.H);   // (voxel index of the row's first voxel: < 2^27)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int c = nt * 16 + kq * 4;
          if constexpr (SIGN4) {
            const unsigned off = ((vox + (unsigned)x) * (unsigned)(d.Cx >> 2) + (unsigned)(c >> 2)) | (((unsigned)ok - 1u) & OORV);
            mk[mt][px][nt] = __builtin_amdgcn_raw_buffer_load_b8(rxs, off, 0, 0);
          } else {
            const unsigned xpos = XSL == LR_LAYOUT_BF16_NDHWC ? (unsigned)x : (unsigned)(px * (d.H >> 1) + xq);
            const unsigned off = (((vox + xpos) * (unsigned)d.Cx + (unsigned)c) * 2u) | (((unsigned)ok - 1u) & OORV);
            xsv[mt][px][nt] = __builtin_amdgcn_raw_buffer_load_b64(rxs, off, 0, 0);
            mk[mt][px][nt] = 0u;
          }
        }
      }
    }
    f32x4 accp[2][4][NT];
#pragma unroll
    for (int px = 1; px >= 0; --px) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) accp[px][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int ntap = 1 << (px + py + pz);
      for (int tapi = 0; tapi < ntap; ++tapi) {
        const int ix = tapi & px, r1 = tapi >> px, iy = r1 & py, iz = (r1 >> py) & pz;
        const int tx = px ? 2 * ix : 1, ox = px ? 1 - ix : 0;
        const int ty = py ? 2 * iy : 1, oy = py ? 1 - iy : 0;
        const int tz = pz ? 2 * iz : 1, oz = pz ? 1 - iz : 0;
        const int tap = (tz * 3 + ty) * 3 + tx;
        u32x4 bw[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bw[nt] = wp[((int64_t)tap * NT + nt) * 64 + lane];
        const u16* src = lts + (((wave + oz) * 5 + oy) * 17 + ox) * VS;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const u32x4 a = *reinterpret_cast<const u32x4*>(src + mt * 17 * VS);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            accp[px][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bw[nt]),
                                                                      __builtin_bit_cast(bf16x8, a), accp[px][mt][nt], 0, 0, 0);
        }
      }
    }
    // ---- epilogue: mask, round, store through the output plane's resource (W*H*Cx*2 bytes)
    const __amdgpu_buffer_rsrc_t rgx = __builtin_amdgcn_make_buffer_rsrc(
        gx + ((int64_t)b * d.D + z) * d.W * d.H * d.Cx, (short)0, d.W * d.H * d.Cx * 2, 0x00020000);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int y = 2 * (yq0 + mt) + py;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          const int x = 2 * xq + px;
          const int ok = (int)(y < d.W) & (int)(x < d.H);
          const int c = nt * 16 + kq * 4;
          f32x4 v = accp[px][mt][nt];
          if constexpr (SIGN4) {
            const unsigned m = mk[mt][px][nt];
            v[0] = (m & 1u) ? v[0] : v[0] * d.slope;
            v[1] = (m & 2u) ? v[1] : v[1] * d.slope;
            v[2] = (m & 4u) ? v[2] : v[2] * d.slope;
            v[3] = (m & 8u) ? v[3] : v[3] * d.slope;
          } else {
            const u32x2 xs = xsv[mt][px][nt];
            const short s0 = (short)(xs[0] & 0xffffu), s1 = (short)(xs[0] >> 16), s2 = (short)(xs[1] & 0xffffu), s3 = (short)(xs[1] >> 16);
            v[0] = s0 > 0 ? v[0] : v[0] * d.slope;   // bf16 > 0  <=>  its bit pattern as int16 > 0
            v[1] = s1 > 0 ? v[1] : v[1] * d.slope;
            v[2] = s2 > 0 ? v[2] : v[2] * d.slope;
            v[3] = s3 > 0 ? v[3] : v[3] * d.slope;
          }
          const unsigned lo = (unsigned)to_bf16(v[0]) | ((unsigned)to_bf16(v[1]) << 16);
          const unsigned hi = (unsigned)to_bf16(v[2]) | ((unsigned)to_bf16(v[3]) << 16);
          const unsigned off = (unsigned)(((y * d.H + x) * d.Cx + c) * 2) | (((unsigned)ok - 1u) & OORV);
          __builtin_amdgcn_raw_buffer_store_b64((u32x2){lo, hi}, rgx, off, 0, 0);
        }
    }
  }
}

// packed[(s*NT + nt)*64 + lane] = the 8 bf16 weights W[co = nt*16 + (lane&15)][k-block lane>>4] of MFMA step s
__global__ void pack_bf16_kernel(const float* __restrict__ w, u32x4* __restrict__ packed, int Cin, int Cout, int NT) {
  const int NS = Cin == 32 ? 27 : 14;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NS * NT * 64) return;
  const int lane = idx & 63, nt = (idx >> 6) % NT, s = (idx >> 6) / NT;
  const int co = nt * 16 + (lane & 15), kq = lane >> 4;
  unsigned r[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    unsigned pair = 0u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = 2 * m + h;
      const int tap = Cin == 32 ? s : 2 * s + (kq >> 1);
      const int ci = Cin == 32 ? kq * 8 + i : (kq & 1) * 8 + i;
      const float v = (tap < 27 && co < Cout) ? w[((int64_t)co * Cin + ci) * 27 + tap] : 0.0f;
      pair |= (unsigned)to_bf16(v) << (16 * h);
    }
    r[m] = pair;
  }
  packed[idx] = (u32x4){r[0], r[1], r[2], r[3]};
}

// fp32 -> bf16 storage of a channels-last tensor (tests, and a model whose first block is fed bf16 upstream)
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ in, u16* __restrict__ out, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = to_bf16(in[i]);
}

}  // namespace

extern "C" int64_t lr_conv3d_packed_bf16_bytes(int Cin, int Cout) {
  if ((Cin != 16 && Cin != 32) || (Cout != 16 && Cout != 32)) return LR_EUNSUPPORTED;
  return (int64_t)(Cin == 32 ? 27 : 14) * (Cout / 16) * 64 * 16;
}

extern "C" int lr_conv3d_pack_weights_bf16(const float* weight, void* packed, int Cin, int Cout, void* stream) {
  if (!weight || !packed) return LR_ENULL;
  if ((Cin != 16 && Cin != 32) || (Cout != 16 && Cout != 32)) return LR_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(packed) & 15u) return LR_EALIGN;
  const int total = (Cin == 32 ? 27 : 14) * (Cout / 16) * 64;
  hipLaunchKernelGGL(pack_bf16_kernel, dim3((total + 255) / 256), dim3(256), 0, lr_stream(stream), weight,
                     reinterpret_cast<u32x4*>(packed), Cin, Cout, Cout / 16);
  return lr_launch_status();
}

extern "C" int64_t lr_conv3d_packed_bf16_planar_bytes(int Cin, int Cout) {
  if (Cin < 1 || (Cout != 16 && Cout != 32)) return LR_EUNSUPPORTED;
  // [channel-pass packing | packing of the all-channels kernel (conv0_cl_bf16.hip), present for 4 < Cin <= 16]
  // ... | packing of the z-marching 3-channel kernel (conv0_split_f32.hip, NS = 1), present for Cin <= 4]
  return (int64_t)((Cin + 2) / 3) * 4 * (Cout / 16) * 64 * 16 + lr_internal_conv0_cl_bf16_packed_bytes(Cin, Cout) +
         lr_internal_conv0_split_packed_floats(Cin, Cout) * 4;
}

extern "C" int lr_conv3d_pack_weights_bf16_planar(const float* weight, void* packed, int Cin, int Cout, void* stream) {
  if (!weight || !packed) return LR_ENULL;
  if (Cin < 1 || (Cout != 16 && Cout != 32)) return LR_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(packed) & 15u) return LR_EALIGN;
  const int total = ((Cin + 2) / 3) * 4 * (Cout / 16) * 64;
  hipStream_t st = lr_stream(stream);
  hipLaunchKernelGGL(pack_bf16_planar_kernel, dim3((total + 255) / 256), dim3(256), 0, st, weight,
                     reinterpret_cast<u32x4*>(packed), Cin, Cout, Cout / 16);
  if (int e = lr_launch_status()) return e;
  if (int e = lr_internal_conv0_cl_bf16_pack(weight, reinterpret_cast<unsigned char*>(packed) + (size_t)total * 16, Cin, Cout, st)) return e;
  return lr_internal_conv0_split_pack(weight, reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(packed) + (size_t)total * 16 +
                                                                       lr_internal_conv0_cl_bf16_packed_bytes(Cin, Cout)), Cin, Cout, st);
}

// The encoder's first block in the bf16 variant: fp32 NCDHW input (rounded to bf16 on the way into the MFMA),
// stride 1, bf16 channels-last output.
static int first_bf16_impl(const float* in, const void* packed_w, const float* bias, void* out, int B, int Cin,
                           int Cout, int D, int W, int H, int out_layout, float negative_slope, long long out_bs,
                           void* stream, unsigned char* mask_out = nullptr) {
  if (!in || !packed_w || !out) return LR_ENULL;
  if (B < 1 || Cin < 1 || D < 1 || W < 1 || H < 1) return LR_EINVAL;
  if (Cout != 16 && Cout != 32) return LR_EUNSUPPORTED;
  if (out_layout != LR_LAYOUT_BF16_NDHWC && out_layout != LR_LAYOUT_BF16_NDHWC_HPS) return LR_EINVAL;
  if (out_layout == LR_LAYOUT_BF16_NDHWC_HPS && (H & 1)) return LR_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(packed_w) & 15u) || (reinterpret_cast<uintptr_t>(out) & 7u)) return LR_EALIGN;
  ConvDimsH d;
  d.B = B; d.Cin = Cin; d.Cout = Cout; d.D = D; d.W = W; d.H = H; d.Do = D; d.Wo = W; d.Ho = H;
  if (out_bs != 0 && out_bs < (long long)Cout * D * W * H) return LR_EINVAL;
  d.out_bs = out_bs ? out_bs : (long long)Cout * D * W * H;
  d.nHq = (H + PH - 1) / PH; d.nWq = (W + PW - 1) / PW; d.nDq = (D + PD - 1) / PD;
  const int64_t nblk = (int64_t)B * d.nDq * d.nWq * d.nHq;
  if (nblk > 0x7fffffffLL) return LR_EINVAL;
  const int vec4 = (H % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15u) == 0) &&
                   ((int64_t)3 * D * W * H * 4 + (int64_t)16 * W * H * 4 < 0x7fffffffLL);
  const dim3 grid((unsigned)nblk), block(256);
  hipStream_t st = lr_stream(stream);
  const u32x4* wt = reinterpret_cast<const u32x4*>(packed_w);
  // many channels (C4): all of them staged once, channels-last in LDS (conv0_cl_bf16.hip); LIFTREG_CONV0_BF16_CL=1 sends the
  // 3-channel case there too (A/B aid), LIFTREG_CONV0_BF16_PASSES=1 keeps everything on the channel-pass kernel
  if (mask_out && Cin > 3) return LR_EUNSUPPORTED;   // the mask comes out of the single-pass kernel's per-tile store
  // few channels (C3 / C5: 3): the z-marching kernel of conv0_split_f32.hip under the bf16 contract — every input
  // plane fetched once, 4 MFMAs per tile; LIFTREG_CONV0_BF16_PASSES=1 keeps the channel-pass kernel (A/B aid, tests)
  if (Cin <= (mask_out ? 3 : 4) && Cout == 16 && !lr_sw_set(LR_SW_CONV0_BF16_CL) && !lr_sw_set(LR_SW_CONV0_BF16_PASSES)) {
    const unsigned char* pm = reinterpret_cast<const unsigned char*>(packed_w) + (size_t)((Cin + 2) / 3) * 4 * (Cout / 16) * 64 * 16 +
                              lr_internal_conv0_cl_bf16_packed_bytes(Cin, Cout);
    const int e = lr_internal_conv0_march_bf16(in, pm, bias, out, B, Cin, D, W, H, out_layout, negative_slope, d.out_bs, mask_out, st);
    if (e != LR_EUNSUPPORTED) return e;
  }
  if (!mask_out && (Cin > 3 || lr_sw_set(LR_SW_CONV0_BF16_CL)) && !lr_sw_set(LR_SW_CONV0_BF16_PASSES)) {
    const int e = lr_internal_conv0_cl_bf16(in, wt + (size_t)((Cin + 2) / 3) * 4 * (Cout / 16) * 64, bias, out, B, Cin, Cout, D, W, H,
                                            out_layout, negative_slope, d.out_bs, 0, st);
    if (e != LR_EUNSUPPORTED) return e;
  }
#define LR_C0(NTV, SG)                                                                                                   \
  do {                                                                                                                    \
    if (out_layout == LR_LAYOUT_BF16_NDHWC_HPS)                                                                           \
      hipLaunchKernelGGL((conv0_bf16_kernel<NTV, SG, true>), grid, block, 0, st, in, wt, bias, out, d, out_layout, negative_slope, vec4, mask_out);  \
    else                                                                                                                  \
      hipLaunchKernelGGL((conv0_bf16_kernel<NTV, SG, false>), grid, block, 0, st, in, wt, bias, out, d, out_layout, negative_slope, vec4, mask_out); \
  } while (0)
  if (Cout == 16) { if (Cin <= 3) LR_C0(1, true); else LR_C0(1, false); }
  else            { if (Cin <= 3) LR_C0(2, true); else LR_C0(2, false); }
#undef LR_C0
  return lr_launch_status();
}

extern "C" int lr_conv3d_dgrad_bf16(const void* gpre, const void* packed_wT, void* gx, int B, int Cg, int Cx, int D,
                                    int W, int H, const void* x_saved, int x_layout, float negative_slope,
                                    void* stream) {
  if (!gpre || !packed_wT || !gx || !x_saved) return LR_ENULL;
  if (Cg != 32 || (Cx != 16 && Cx != 32)) return LR_EUNSUPPORTED;
  if (B < 1 || D < 1 || W < 1 || H < 1) return LR_EINVAL;
  if (x_layout != LR_LAYOUT_BF16_NDHWC && x_layout != LR_LAYOUT_BF16_NDHWC_HPS && x_layout != LR_LAYOUT_SIGN4) return LR_EINVAL;
  if (x_layout == LR_LAYOUT_BF16_NDHWC_HPS && (H & 1)) return LR_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(gpre) | reinterpret_cast<uintptr_t>(packed_wT)) & 15u) return LR_EALIGN;
  if (reinterpret_cast<uintptr_t>(gx) & 7u) return LR_EALIGN;
  if (x_layout != LR_LAYOUT_SIGN4 && (reinterpret_cast<uintptr_t>(x_saved) & 7u)) return LR_EALIGN;
  DgDimsH d;
  d.B = B; d.Cx = Cx; d.D = D; d.W = W; d.H = H;
  d.Do = (D - 1) / 2 + 1; d.Wo = (W - 1) / 2 + 1; d.Ho = (H - 1) / 2 + 1;
  if ((int64_t)6 * d.Wo * d.Ho * Cg * 2 >= 0x7fffffffLL) return LR_EINVAL;
  d.nHq = ((H + 1) / 2 + 15) / 16; d.nWq = ((W + 1) / 2 + 3) / 4; d.nDq = ((D + 1) / 2 + 3) / 4;
  d.xs_layout = x_layout; d.slope = negative_slope;
  const int64_t nblk = (int64_t)B * d.nDq * d.nWq * d.nHq;
  if (nblk > 0x7fffffffLL) return LR_EINVAL;
  const dim3 grid((unsigned)nblk), blk(256);
  hipStream_t st = lr_stream(stream);
  const u16* g = reinterpret_cast<const u16*>(gpre);
  const u32x4* wt = reinterpret_cast<const u32x4*>(packed_wT);
  u16* o = reinterpret_cast<u16*>(gx);
  const u16* xs = reinterpret_cast<const u16*>(x_saved);
  // 31-bit byte offsets inside one batch element of the mask source / one plane of the output
  if ((int64_t)D * W * H * Cx * 2 >= 0x7fffffffLL || (int64_t)W * H * Cx * 2 >= 0x7fffffffLL) return LR_EUNSUPPORTED;
#define LR_DGB(NTV)                                                                                                          \
  do {                                                                                                                       \
    if (x_layout == LR_LAYOUT_SIGN4) hipLaunchKernelGGL((conv3d_dgrad_bf16_kernel<NTV, LR_LAYOUT_SIGN4>), grid, blk, 0, st, g, wt, o, xs, d);        \
    else if (x_layout == LR_LAYOUT_BF16_NDHWC) hipLaunchKernelGGL((conv3d_dgrad_bf16_kernel<NTV, LR_LAYOUT_BF16_NDHWC>), grid, blk, 0, st, g, wt, o, xs, d); \
    else hipLaunchKernelGGL((conv3d_dgrad_bf16_kernel<NTV, LR_LAYOUT_BF16_NDHWC_HPS>), grid, blk, 0, st, g, wt, o, xs, d);    \
  } while (0)
  if (Cx == 16) LR_DGB(1); else LR_DGB(2);
#undef LR_DGB
  return lr_launch_status();
}

extern "C" int lr_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream) {
  if (!in || !out) return LR_ENULL;
  if (n < 0) return LR_EINVAL;
  if (n == 0) return LR_OK;
  int64_t nblk = (n + 255) / 256;
  if (nblk > 16384) nblk = 16384;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)nblk), dim3(256), 0, lr_stream(stream), in,
                     reinterpret_cast<u16*>(out), n);
  return lr_launch_status();
}

static int conv_bf16_impl(const void* in, const void* packed_w, const float* bias, void* out, int B,
                          int Cin, int Cout, int D, int W, int H, int stride, int in_layout,
                          int out_layout, float negative_slope, long long out_bs, void* stream) {
  if (!in || !packed_w || !out) return LR_ENULL;
  if (B < 1 || D < 1 || W < 1 || H < 1) return LR_EINVAL;
  if (stride != 2 || (Cin != 16 && Cin != 32) || (Cout != 16 && Cout != 32)) return LR_EUNSUPPORTED;
  if (in_layout != LR_LAYOUT_BF16_NDHWC && in_layout != LR_LAYOUT_BF16_NDHWC_HPS) return LR_EINVAL;
  if (out_layout != LR_LAYOUT_NCDHW && out_layout != LR_LAYOUT_BF16_NDHWC && out_layout != LR_LAYOUT_BF16_NDHWC_HPS)
    return LR_EINVAL;
  const bool ps = in_layout == LR_LAYOUT_BF16_NDHWC_HPS;
  if (ps && (H & 1)) return LR_EUNSUPPORTED;
  ConvDimsH d;
  d.B = B; d.Cin = Cin; d.Cout = Cout; d.D = D; d.W = W; d.H = H;
  d.Do = (D - 1) / 2 + 1; d.Wo = (W - 1) / 2 + 1; d.Ho = (H - 1) / 2 + 1;
  {
    const long long dense_bs = (long long)Cout * d.Do * d.Wo * d.Ho;
    if (out_bs != 0 && out_bs < dense_bs) return LR_EINVAL;
    d.out_bs = out_bs ? out_bs : dense_bs;
  }
  if (out_layout == LR_LAYOUT_BF16_NDHWC_HPS && (d.Ho & 1)) return LR_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(packed_w)) & 15u) return LR_EALIGN;
  if (out_layout != LR_LAYOUT_NCDHW && (reinterpret_cast<uintptr_t>(out) & 7u)) return LR_EALIGN;
  if ((int64_t)12 * W * H * Cin * 2 + 4096 >= 0x7fffffffLL) return LR_EINVAL;  // 31-bit offsets inside a window
  // rows per wave: 8 for the big first stride-2 block (fewer weight loads per MFMA), 4 otherwise; LIFTREG_BF16_MT overrides
  int mtb = (Cin == 16 && d.Wo >= 64) ? 8 : 4;
  if (lr_sw_set(LR_SW_BF16_MT)) mtb = lr_sw_int(LR_SW_BF16_MT, 4) == 8 ? 8 : 4;  // tuning aid
  const bool rows = ps && !lr_sw_set(LR_SW_CONV_TAPMAJOR);  // parity-split rows: the row-major kernel (tuning aid: tap-major)
  if (rows) mtb = 4;  // 4 rows per wave: 8 would not fit three waves per SIMD
  d.nHq = (d.Ho + 15) / 16; d.nWq = (d.Wo + mtb - 1) / mtb; d.nDq = (d.Do + TD - 1) / TD;
  const int64_t nblk = (int64_t)B * d.nDq * d.nWq * d.nHq;
  if (nblk > 0x7fffffffLL) return LR_EINVAL;
  const dim3 grid((unsigned)nblk), block(256);
  hipStream_t st = lr_stream(stream);
  const u16* x = reinterpret_cast<const u16*>(in);
  const u32x4* wt = reinterpret_cast<const u32x4*>(packed_w);
#define LR_BF(NTV, C32, PSV)                                                                                                      \
  do {                                                                                                                            \
    if (mtb == 8) hipLaunchKernelGGL((conv3d_cl_bf16_kernel<NTV, C32, PSV, 8>), grid, block, 0, st, x, wt, bias, out, d, out_layout, negative_slope); \
    else hipLaunchKernelGGL((conv3d_cl_bf16_kernel<NTV, C32, PSV, 4>), grid, block, 0, st, x, wt, bias, out, d, out_layout, negative_slope);          \
  } while (0)
  const int NT = Cout / 16;
  // 32 couts on parity-split rows: the z-marching kernels (LIFTREG_BF16_NO_MARCH: the row kernel)
  if (rows && NT == 2 && !lr_sw_on(LR_SW_BF16_NO_MARCH) && (int64_t)(D + 4) * W * H * Cin * 2 < 0x7fffffffLL) {
    const bool ty8 = Cin == 16 && lr_sw_on(LR_SW_BF16_MARCH_TY8);
    const int mty = Cin == 16 ? (ty8 ? 8 : 4) : M3_TY, mtx = Cin == 16 ? MZ_TX : M3_TX;
    const int nTy = (d.Wo + mty - 1) / mty, nTx = (d.Ho + mtx - 1) / mtx;
    // columns are serial walks: below ~one column per CU the row kernel's many small blocks win (measured: 32^3 and 16^3 inputs
    // at B = 8).  The rule looks at the plane and the batch only, never at D: a z-slab of a volume takes the same kernel as
    // the whole volume (sharded == unsharded bit for bit).  LIFTREG_BF16_MARCH_ZC set: always (tests).
    const bool march_pays = (int64_t)B * nTy * nTx >= 256 || lr_sw_set(LR_SW_BF16_MARCH_ZC);
    if (march_pays) {
      // z chunks: columns of >= 16 steps, enough units for ~8 rounds of the resident blocks
      int zc = lr_sw_int(LR_SW_BF16_MARCH_ZC, 0);
      if (zc < 1) {
        int nz = 1;
        while ((int64_t)B * nTy * nTx * nz < 6144 && d.Do / (nz * 2) >= 16) nz *= 2;
        zc = (d.Do + nz - 1) / nz;
      }
      d.nHq = nTx; d.nWq = nTy; d.nDq = (d.Do + zc - 1) / zc;
      const int64_t nb = (int64_t)B * d.nDq * d.nWq * d.nHq;
      if (nb > 0x7fffffffLL) return LR_EINVAL;
      if (Cin == 16 && ty8) hipLaunchKernelGGL(conv3d_march_s2_bf16_kernel<8>, dim3((unsigned)nb), dim3(512), 0, st, x, wt, bias, out, d, zc, out_layout, negative_slope);
      else if (Cin == 16) hipLaunchKernelGGL(conv3d_march_s2_bf16_kernel<4>, dim3((unsigned)nb), block, 0, st, x, wt, bias, out, d, zc, out_layout, negative_slope);
      else hipLaunchKernelGGL(conv3d_march_s2_c32_bf16_kernel, dim3((unsigned)nb), block, 0, st, x, wt, bias, out, d, zc, out_layout, negative_slope);
      return lr_launch_status();
    }
  }
  if (rows) {
#define LR_BR(NTV, C32) hipLaunchKernelGGL((conv3d_cl_rows_bf16_kernel<NTV, 4, C32>), grid, block, 0, st, x, wt, bias, out, d, out_layout, negative_slope)
    if (Cin == 32) { if (NT == 2) LR_BR(2, true); else LR_BR(1, true); }
    else           { if (NT == 2) LR_BR(2, false); else LR_BR(1, false); }
#undef LR_BR
    return lr_launch_status();
  }
  if (Cin == 32) {
    if (NT == 2) { if (ps) LR_BF(2, true, true); else LR_BF(2, true, false); }
    else         { if (ps) LR_BF(1, true, true); else LR_BF(1, true, false); }
  } else {
    if (NT == 2) { if (ps) LR_BF(2, false, true); else LR_BF(2, false, false); }
    else         { if (ps) LR_BF(1, false, true); else LR_BF(1, false, false); }
  }
#undef LR_BF
  return lr_launch_status();
}

extern "C" int lr_conv3d_k3_lrelu_bf16(const void* in, const void* packed_w, const float* bias, void* out, int B,
                                       int Cin, int Cout, int D, int W, int H, int stride, int in_layout,
                                       int out_layout, float negative_slope, void* stream) {
  return conv_bf16_impl(in, packed_w, bias, out, B, Cin, Cout, D, W, H, stride, in_layout, out_layout, negative_slope, 0, stream);
}

// lr_conv3d_k3_lrelu_bf16 writing into a strided batch (see lr_conv3d_k3_lrelu_obs_f32): out_batch_stride in elements of
// the output type (bf16 for the channels-last layouts, fp32 for NCDHW).
extern "C" int lr_conv3d_k3_lrelu_obs_bf16(const void* in, const void* packed_w, const float* bias, void* out, int B,
                                           int Cin, int Cout, int D, int W, int H, int stride, int in_layout,
                                           int out_layout, float negative_slope, int64_t out_batch_stride, void* stream) {
  if (out_batch_stride < 0) return LR_EINVAL;
  return conv_bf16_impl(in, packed_w, bias, out, B, Cin, Cout, D, W, H, stride, in_layout, out_layout, negative_slope,
                        (long long)out_batch_stride, stream);
}

extern "C" int lr_conv3d_first_bf16(const float* in, const void* packed_w, const float* bias, void* out, int B, int Cin,
                                    int Cout, int D, int W, int H, int out_layout, float negative_slope,
                                    void* stream) {
  return first_bf16_impl(in, packed_w, bias, out, B, Cin, Cout, D, W, H, out_layout, negative_slope, 0, stream);
}

// Training forward of the bf16 variant's first block: lr_conv3d_first_bf16 (Cin <= 3) that ALSO writes mask_out
// (B,D,W,H,Cout/4) uint8 (LR_LAYOUT_SIGN4): bit r of byte q = "stored bf16 of output channel 4q+r > 0" — what
// lr_conv3d_dgrad_bf16 / lr_conv3d_dgrad_f32 take as x_saved with x_layout = LR_LAYOUT_SIGN4 instead of the activation.
extern "C" int lr_conv3d_first_mask_bf16(const float* in, const void* packed_w, const float* bias, void* out, uint8_t* mask_out,
                                         int B, int Cin, int Cout, int D, int W, int H, int out_layout, float negative_slope,
                                         void* stream) {
  if (!mask_out) return LR_ENULL;
  return first_bf16_impl(in, packed_w, bias, out, B, Cin, Cout, D, W, H, out_layout, negative_slope, 0, stream, mask_out);
}

// lr_conv3d_first_bf16 writing into a strided batch (see lr_conv3d_k3_lrelu_obs_f32; stride in bf16 elements).
extern "C" int lr_conv3d_first_obs_bf16(const float* in, const void* packed_w, const float* bias, void* out, int B, int Cin,
                                        int Cout, int D, int W, int H, int out_layout, float negative_slope,
                                        int64_t out_batch_stride, void* stream) {
  if (out_batch_stride < 0) return LR_EINVAL;
  return first_bf16_impl(in, packed_w, bias, out, B, Cin, Cout, D, W, H, out_layout, negative_slope, (long long)out_batch_stride, stream);
}

// The first block of the bf16 variant on the CHANNELS-LAST bf16 encoder input (B,D,W,H,16) written by
// lr_backproject_encin_bf16 (channel 0 = moving, 1..Cin-1 = the backprojected views, the rest 0): same weights
// (lr_conv3d_pack_weights_bf16_planar), same products, same results as lr_conv3d_first_bf16 on the fp32 NCDHW
// cat([moving, views]) — the rounding to bf16 merely happened one kernel earlier.  Cout = 16, 1 <= Cin <= 16.
extern "C" int lr_conv3d_first_clin_bf16(const void* in, const void* packed_w, const float* bias, void* out, int B, int Cin,
                                         int Cout, int D, int W, int H, int out_layout, float negative_slope,
                                         int64_t out_batch_stride, void* stream) {
  if (!in || !packed_w || !out) return LR_ENULL;
  if (B < 1 || Cin < 1 || D < 1 || W < 1 || H < 1 || out_batch_stride < 0) return LR_EINVAL;
  if (Cout != 16 || Cin > 16) return LR_EUNSUPPORTED;
  if (out_layout != LR_LAYOUT_BF16_NDHWC && out_layout != LR_LAYOUT_BF16_NDHWC_HPS) return LR_EINVAL;
  if (out_layout == LR_LAYOUT_BF16_NDHWC_HPS && (H & 1)) return LR_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(packed_w) & 15u) || (reinterpret_cast<uintptr_t>(out) & 7u) || (reinterpret_cast<uintptr_t>(in) & 15u)) return LR_EALIGN;
  if (out_batch_stride != 0 && out_batch_stride < (int64_t)Cout * D * W * H) return LR_EINVAL;
  const u32x4* wt = reinterpret_cast<const u32x4*>(packed_w);
  return lr_internal_conv0_cl_bf16(reinterpret_cast<const float*>(in), wt + (size_t)((Cin + 2) / 3) * 4 * (Cout / 16) * 64, bias, out, B,
                                   Cin, Cout, D, W, H, out_layout, negative_slope, (long long)out_batch_stride, 1, lr_stream(stream));
}
