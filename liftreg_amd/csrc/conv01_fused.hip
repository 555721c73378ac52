// conv01_fused.hip — the first TWO encoder blocks (Cin <= 4 planar fp32 channels -> 16 channels at stride 1 -> 32 channels at
// stride 2, each Conv3d 3x3x3 + bias + LeakyReLU) as ONE z-marching kernel on the bf16 matrix pipe with EXACT three-way bf16
// splits of every fp32 operand (conv0_split_f32.hip describes the split: x = x0 + x1 + x2, six of the nine partial products,
// each exact in fp32, fp32 accumulation).  The 16-channel fp32 activation between the two blocks (8.6 GB written + 8.6 GB read
// + halo re-reads at 256^3, B = 8 — 22 of the step's 47 GB) never reaches HBM: it lives in LDS as bf16 triples.
//
// Work decomposition.  A 256-thread block (one per CU, four waves = one per SIMD, up to 512 registers per lane) owns a
// column of 8 x 8 block-1 outputs and marches DOWN the output planes oz = 0 .. Do-1.  Step oz:
//   (0) request the two input planes of step oz+1 (z = 2oz+3, 2oz+4; 19 x 24 voxels x Cin, one 16-byte load per channel per
//       thread, bounds-checked buffer loads = the conv's zero padding);
//   (A) block 0: planes z = 2oz, 2oz+1 of its output on the (17 x 17) region the 8 x 8 tile needs, from the ring of 4 input
//       planes in LDS ("ring 0": 8-byte records of 4 channels, one array per split — conv0_split_f32.hip's operand order:
//       K = 8 taps x 4 channels, 4 k-blocks x 6 products = 24 MFMAs per 16-voxel tile; 38 tiles per step);
//       bias + LeakyReLU, voxels outside the volume -> 0 (block 1's padding), split into three bf16 and stored into "ring 1"
//       (3 planes: 2oz-1 from the step before, 2oz, 2oz+1);                                          -- barrier --
//   (B) block 1: K = 27 taps x 16 channels = 13.5 k-blocks of 32; a wave owns two of the four 16-voxel tiles (rows 4mh..4mh+3)
//       and one HALF of K (taps 0..13 | 14..26: 7 k-blocks) for both 16-cout tiles, its 42 weight fragments (three splits)
//       stationary in 168 registers for the life of the block; every operand fragment is read from LDS by exactly one wave
//       (3 fragments = 6 ds_read_b64 per 12 MFMAs); 168 MFMAs per wave;
//   (C) the K halves meet through 8 KB of LDS; the ring-0 planes requested in (0) are split and written;   -- barrier --
//   (D) sum, bias, LeakyReLU, two 16-byte stores per lane.
// 408 MFMAs of 16 cycles per SIMD and step against 2 x 8 KB of HBM traffic: the pair is bound by the matrix pipe.
//
// LDS (137.6 KB): ring 0 = 4 planes x 19 rows x [3 splits][24 records of 8 bytes]; ring 1 = 3 planes x 3 splits x 17 rows
// x [4 channel quads][17 voxels] of 8 bytes, quads in the order 0,2,1,3 — a lane's two quads (8 channels) are a fixed 272 bytes
// apart, the odd quad stride (17 chunks) and the row stride (72 chunks = 8 mod 16) make both the block-0 epilogue's
// ds_write_b64 (16 lanes = 16 consecutive voxels) and block 1's ds_read_b64 (32 lanes = 8 voxels x 2 rows x 2 channel
// halves) conflict-free.
//
// Arithmetic: both stages are DIRECT convolutions whose products are exact; only the fp32 accumulation rounds (once per MFMA
// and partial sum), so against an fp64 convolution the pair is closer than the fmaf chain of the fp32 kernels
// (tests/test_gpu_conv01_fused.py).  Inf input -> NaN (Inf splits into Inf, NaN, NaN).
//
// Replaces (reference file:line): src/liftreg/layers/layers.py:365-369 (Conv3d + LeakyReLU), twice, as wired at
//   src/liftreg/models/LiftRegDeformSubspaceBackproj.py:95-100 for encoders[0] and encoders[1].
#include "lr_common.h"
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int NTHR = 256;
constexpr int TY = 8, TX = 8;            // block-1 outputs of a column (rows x voxels)
constexpr int R1 = 2 * TY + 1;           // 17: rows / voxels of block 0's output a plane of the column needs
constexpr int R0 = R1 + 2;               // 19: input rows
constexpr int NQ0 = 6;                   // aligned float4 quads of an input row: x = 2*ox0 - 4 .. 2*ox0 + 19 (record p = x - (2*ox0 - 4))
constexpr int SB0 = NQ0 * 4 * 8;         // 192 bytes: one split of a ring-0 row
constexpr int RB0 = 3 * SB0;             // 576 = 64 mod 128: the k-block-3 lane pairs (two window rows apart) use opposite bank halves
constexpr int PLB0 = R0 * RB0 + 192;     // 11136 = 128 mod 256: neighbouring planes use opposite bank halves
constexpr int NRING0 = 4;
constexpr int RING0 = NRING0 * PLB0;     // 44544
constexpr int QS1 = 17;                  // 8-byte chunks between the channel-quad runs of a ring-1 row (odd)
constexpr int RS1 = 72;                  // chunks of a ring-1 row (= 8 mod 16)
constexpr int SPB1 = R1 * RS1 * 8;       // 9792 bytes: one split of a ring-1 plane
constexpr int PLB1 = 3 * SPB1;           // 29376
constexpr int RING1_OFF = RING0;
constexpr int RING1 = 3 * PLB1;          // 88128
constexpr int SCR_OFF = RING1_OFF + RING1;   // 132672
constexpr int SCR = 4 * 2 * 64 * 16;     // 8192: [wave][cout tile][lane] partial sums of the tile the wave does not own
constexpr int LDSB = SCR_OFF + SCR;      // 140864
constexpr int NITEM = 2 * R0 * NQ0;      // 228 staging items of a step: (plane, row, x-quad), all channels
constexpr int NKB0 = 4, NKB1 = 7;
constexpr unsigned OOR = 0x80000000u;
static_assert(NITEM <= NTHR, "one staging item per thread");
static_assert(4 * QS1 <= RS1 && (QS1 & 1) == 1 && (RS1 & 15) == 8, "ring-1 bank geometry");
static_assert(LDSB <= 160 * 1024, "LDS");

struct FDims {
  int B, Cin, D, W, H, Do, Wo, Ho;
  int nTx, nTy, nunits;
  int hps;                 // output layout: 1 = LR_LAYOUT_NDHWC_HPS, 0 = LR_LAYOUT_NDHWC
  long long bs0, bsr;      // elements between batch elements of channel 0 | of channels 1..Cin-1
  long long out_bs;        // output elements between batch elements
  float slope0, slope1;
};

// (a, b) -> three packed bf16 pairs with a = a0 + a1 + a2 and b = b0 + b1 + b2 exactly
__device__ __forceinline__ void split3(float a, float b, unsigned (&p)[3]) {
  f32x2 v = {a, b};
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const bf16x2 h = __builtin_convertvector(v, bf16x2);   // round to nearest even
    const unsigned u = __builtin_bit_cast(unsigned, h);
    p[s] = u;
    if (s < 2) {
      const f32x2 f = {__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
      v = v - f;   // exact
    }
  }
}

// the 8 XCDs (block id % 8) take contiguous eighths of the unit order (x fastest, then y, batch); the blocks of an XCD
// stride through their eighth together: neighbouring columns (shared halo rows) meet in one L2
__device__ __forceinline__ void unit_range(int bid, int nblk, int nunits, int& first, int& stride, int& end) {
  if ((nblk & 7) == 0 && nunits >= nblk) {
    const int xcd = bid & 7, li = bid >> 3, per = nblk >> 3;
    const int q = nunits >> 3, r = nunits & 7;
    const int lo = xcd * q + (xcd < r ? xcd : r);
    end = lo + q + (xcd < r ? 1 : 0);
    first = lo + li;
    stride = per;
  } else {
    first = bid; stride = nblk; end = nunits;
  }
}

// ds_read_b64 takes 2 LDS cycles per wave, ds_read2_b64 8 for twice the bytes (MI355X_MICROARCH.md, LDS table): every operand
// half is read with its own ds_read_b64; both of hipcc's merging passes are off (the IR vectorizer for the file: Makefile).
#if defined(__HIP_DEVICE_COMPILE__)
#define LR_C01_NO_DS_MERGE __attribute__((target("no-load-store-opt")))
#else
#define LR_C01_NO_DS_MERGE
#endif

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), b, c, 0, 0, 0)

template <int NC>
__global__ __launch_bounds__(NTHR) LR_C01_NO_DS_MERGE void conv01_fused_kernel(
    const float* __restrict__ in0, const float* __restrict__ in_rest, const u32x4* __restrict__ wp0,
    const u32x4* __restrict__ wp1, const float* __restrict__ bias0, const float* __restrict__ bias1,
    float* __restrict__ out, FDims d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mh = wave >> 1, kh = wave & 1;   // phase B: tile pair | K half
  const int col = lane & 15, lq = lane >> 4;
  const int dD = d.D, dW = d.W, dH = d.H;
  const unsigned V4 = (unsigned)dD * dW * dH * 4u;   // bytes of one channel volume (3 of them < 2^31: launcher)

  // zero everything once: a "weight 0" operand slot multiplies whatever lies behind a row / plane and needs finite numbers
  for (int o = tid * 16; o < LDSB; o += NTHR * 16) *reinterpret_cast<u32x4*>(lds + o) = (u32x4){0u, 0u, 0u, 0u};

  // ---- stationary operands
  u32x4 w0[NKB0][3];
#pragma unroll
  for (int kb = 0; kb < NKB0; ++kb)
#pragma unroll
    for (int t = 0; t < 3; ++t) w0[kb][t] = wp0[(kb * 3 + t) * 64 + lane];
  u32x4 w1[NKB1][2][3];
#pragma unroll
  for (int kb = 0; kb < NKB1; ++kb)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int t = 0; t < 3; ++t) w1[kb][c][t] = wp1[((((kh * NKB1 + kb) * 2 + c) * 3) + t) * 64 + lane];
  f32x4 b0v = {0.f, 0.f, 0.f, 0.f}, b1v[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (bias0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) b0v[r] = bias0[lq * 4 + r];
  }
  if (bias1) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) b1v[c][r] = bias1[c * 16 + lq * 4 + r];
  }

  // ---- phase A lane geometry (ring 0).  Lane group lq supplies two taps a fixed distance apart (conv0_split_f32.hip):
  //   k-block ty = 0,1,2: lq = 0..2: (tz = lq, ty, tx = 0) | (tz = lq, ty, tx = 1);  lq = 3: (tz = 0, ty, tx = 2) | weight 0
  //   k-block 3:          lq = 0: (1,0,2) | (1,1,2);  lq = 1: (1,2,2) | weight 0;  lq = 2: (2,0,2) | (2,1,2);  lq = 3: (2,2,2) | weight 0
  // region voxel (ry, rx) of block 0's output = volume (2*oy0 - 1 + ry, 2*ox0 - 1 + rx); its tap (ty, tx) = ring-0 row ry + ty,
  // record rx + tx + 2
  const unsigned laneFx = (unsigned)((2 + (lq == 3 ? 2 : 0)) * 8);
  const unsigned laneGx = (unsigned)((lq & 1) * 2 * RB0 + 4 * 8);
  const unsigned qpos = (unsigned)(((lq & 1) << 1) | (lq >> 1));   // ring-1 position of channel quad lq: order 0,2,1,3
  // this wave's single tiles (beside its four row pairs): wave 0 row 16 of both planes, wave 1 column 16 of both planes,
  // wave 2 the corner of plane 0, wave 3 the corner of plane 1
  const int sry = wave == 1 ? col : 16, srx = wave == 0 ? col : 16;

  // ---- phase B lane geometry (ring 1).  Tile t of the wave = output rows 4*mh + 2*t + {0,1}; lane column = (row r1, voxel oxl);
  // lq = (tap of the k-block's pair, channel half)
  const int r1 = col >> 3, oxl = col & 7, hh = lq & 1, tsel = lq >> 1;
  unsigned base1[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) base1[t] = (unsigned)(RING1_OFF + ((2 * (4 * mh + 2 * t + r1)) * RS1 + 2 * oxl + hh * QS1) * 8);
  unsigned toff[NKB1];
  int tdz[NKB1];
#pragma unroll
  for (int kb = 0; kb < NKB1; ++kb) {
    int tap = 14 * kh + 2 * kb + tsel;
    tap = tap > 26 ? 26 : tap;   // the 28th slot: weight 0 on a real voxel
    const int dz = tap / 9, dy = (tap % 9) / 3, dx = tap % 3;
    toff[kb] = (unsigned)((dy * RS1 + dx) * 8);
    tdz[kb] = dz;
  }

  // ---- this thread's staging item
  const bool item_live = tid < NITEM;
  const int ipl = item_live ? tid / (R0 * NQ0) : 0, irow = item_live ? (tid % (R0 * NQ0)) / NQ0 : 0, iq = item_live ? tid % NQ0 : 0;

  auto make_rsrc = [](const void* p, unsigned bytes) __attribute__((always_inline)) -> __amdgpu_buffer_rsrc_t {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    const uint64_t s = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                       (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(s), (short)0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
  };
  auto frag = [&](unsigned pa, int off, int dlt) __attribute__((always_inline)) -> bf16x8 {
    const u32x2 a = *reinterpret_cast<const u32x2*>(lds + pa + off);
    const u32x2 b = *reinterpret_cast<const u32x2*>(lds + pa + off + dlt);
    return __builtin_bit_cast(bf16x8, (u32x4){a[0], a[1], b[0], b[1]});
  };

  int first, stride, end;
  unit_range((int)blockIdx.x, (int)gridDim.x, d.nunits, first, stride, end);

  for (int uid = first; uid < end; uid += stride) {
    const int utx = uid % d.nTx, uty = (uid / d.nTx) % d.nTy, ub = __builtin_amdgcn_readfirstlane(uid / d.nTx / d.nTy);
    const int oy0 = __builtin_amdgcn_readfirstlane(uty * TY), ox0 = __builtin_amdgcn_readfirstlane(utx * TX);
    const int Y1 = 2 * oy0 - 1, X1 = 2 * ox0 - 1;       // volume coordinates of region-1 (0, 0)
    const int Y0 = 2 * oy0 - 2, X0a = 2 * ox0 - 4;      // volume coordinates of ring-0 row 0 / record 0
    const __amdgpu_buffer_rsrc_t r0 = make_rsrc(in0 + (int64_t)ub * d.bs0, V4);
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(in_rest + (int64_t)ub * d.bsr, NC > 1 ? (unsigned)(NC - 1) * V4 : 0u);

    // input planes zb + {0,1} -> registers (plane ipl of the pair, row irow, x-quad iq; every channel)
    // (the loaded quads travel as HIP's uint4, a struct: element access on the ext-vector result of the buffer-load builtin
    // is narrowed by hipcc to a one-dword load with the other elements undefined — DESIGN.md 6a)
    auto issue_loads = [&](int zb, uint4 (&L)[NC]) __attribute__((always_inline)) {
      const int zi = zb + ipl, yi = Y0 + irow, xi = X0a + 4 * iq;
      const int ok = (int)item_live & (int)(zi >= 0) & (int)(zi < dD) & (int)(yi >= 0) & (int)(yi < dW) & (int)(xi >= 0) & (int)(xi < dH);
      const unsigned dead = ((unsigned)ok - 1u) & OOR;   // dead item: bit 31 -> outside the resource -> 0
      const unsigned voff = (unsigned)(((zi * dW + yi) * dH + xi) * 4);
      L[0] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r0, (int)(voff | dead), 0, 0));
#pragma unroll
      for (int c = 1; c < NC; ++c)
        L[c] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rr, (int)((voff + (unsigned)(c - 1) * V4) | dead), 0, 0));
    };
    // ... -> split -> ring 0 (plane z sits in slot (z + 1) & 3)
    auto write_ring0 = [&](int zb, const uint4 (&L)[NC]) __attribute__((always_inline)) {
      if (!item_live) return;
      const int slot = (zb + ipl + 1) & 3;
      unsigned char* const base = lds + (slot * PLB0 + irow * RB0 + iq * 32);
      unsigned rec[3][4][2];   // [split][voxel][channels 01 | 23]
      float v[4][4];   // [channel][voxel]
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint4 q = L[c < NC ? c : 0];
        v[c][0] = c < NC ? __builtin_bit_cast(float, q.x) : 0.0f; v[c][1] = c < NC ? __builtin_bit_cast(float, q.y) : 0.0f;
        v[c][2] = c < NC ? __builtin_bit_cast(float, q.z) : 0.0f; v[c][3] = c < NC ? __builtin_bit_cast(float, q.w) : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned p01[3], p23[3] = {0u, 0u, 0u};
        split3(v[0][j], v[1][j], p01);
        if (NC > 2) split3(v[2][j], v[3][j], p23);
#pragma unroll
        for (int s = 0; s < 3; ++s) { rec[s][j][0] = p01[s]; rec[s][j][1] = p23[s]; }
      }
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        *reinterpret_cast<u32x4*>(base + s * SB0) = (u32x4){rec[s][0][0], rec[s][0][1], rec[s][1][0], rec[s][1][1]};
        *reinterpret_cast<u32x4*>(base + s * SB0 + 16) = (u32x4){rec[s][2][0], rec[s][2][1], rec[s][3][0], rec[s][3][1]};
      }
    };

    // block-0 tile value -> LeakyReLU -> 0 outside the volume -> three bf16 -> ring 1 (slot s1, region voxel (ry, rx))
    auto emit = [&](f32x4 v, bool ok, int s1, int ry, int rx) __attribute__((always_inline)) {
      v = __builtin_elementwise_max(v, v * d.slope0);   // = LeakyReLU for 0 <= slope <= 1 (launcher)
      if (!ok) v = (f32x4){0.f, 0.f, 0.f, 0.f};
      unsigned p01[3], p23[3];
      split3(v[0], v[1], p01);
      split3(v[2], v[3], p23);
      unsigned char* const base = lds + (RING1_OFF + s1 * PLB1 + (ry * RS1 + (int)qpos * QS1 + rx) * 8);
#pragma unroll
      for (int s = 0; s < 3; ++s) *reinterpret_cast<u32x2*>(base + s * SPB1) = (u32x2){p01[s], p23[s]};
    };

    // ---- unit prologue: ring 0 <- planes -1..2, ring-1 slot of plane -1 <- 0
    __syncthreads();   // (the zero fill of the whole LDS | every read of the unit before)
    {
      uint4 la[NC], lb[NC];
      issue_loads(-1, la);
      issue_loads(1, lb);
      for (int o = tid * 16; o < PLB1; o += NTHR * 16) *reinterpret_cast<u32x4*>(lds + RING1_OFF + 2 * PLB1 + o) = (u32x4){0u, 0u, 0u, 0u};
      write_ring0(-1, la);
      write_ring0(1, lb);
    }
    __syncthreads();

    // the output voxel this lane stores: tile kh of the wave's pair
    const int oy = oy0 + 4 * mh + 2 * kh + r1, ox = ox0 + oxl;
    const bool o_in = oy < d.Wo && ox < d.Ho;
    unsigned ooff[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      unsigned o;
      if (d.hps) {   // row = [channel block of 16][parity][Ho/2][16 floats]
        const int hp = (ox & 1) * (d.Ho >> 1) + (ox >> 1);
        o = (unsigned)((oy * d.Ho * 32 + (c * d.Ho + hp) * 16 + lq * 4) * 4);
      } else {
        o = (unsigned)(((oy * d.Ho + ox) * 32 + c * 16 + lq * 4) * 4);
      }
      ooff[c] = o_in ? o : OOR;
    }
    const bool xok_row = (unsigned)(X1 + col) < (unsigned)dH;                                       // row tiles: rx = col
    const bool sok = (unsigned)(Y1 + sry) < (unsigned)dW && (unsigned)(X1 + srx) < (unsigned)dH;   // this wave's single tiles

    int m3 = 0;   // (2 oz) mod 3: ring-1 slot of plane 2oz; plane 2oz-1 sits in (m3 + 2) % 3, plane 2oz+1 in (m3 + 1) % 3
    for (int oz = 0; oz < d.Do; ++oz) {
      const int e = (2 * oz) & 3;   // ring-0 slot of plane 2oz-1
      uint4 ldn[NC];
      issue_loads(2 * oz + 3, ldn);

      // ================= phase A: block 0, planes 2oz and 2oz+1
      // four row pairs per wave: plane pl, rows 4*wave + 2*(u&1) + {0,1} ... region rows 0..15
#pragma unroll 1
      for (int u = 0; u < 4; ++u) {
        const int pl = u >> 1, ry = 4 * wave + 2 * (u & 1);
        const int sA = (e + pl) & 3, sB = (sA + 1) & 3, sC = (sA + 2) & 3;
        const int s1 = (m3 + pl) % 3;
        const bool zok = 2 * oz + pl < dD;
        const unsigned pF = (unsigned)((lq == 1 ? sB : lq == 2 ? sC : sA) * PLB0 + ry * RB0 + col * 8) + laneFx;
        const unsigned pG = (unsigned)((lq < 2 ? sB : sC) * PLB0 + ry * RB0 + col * 8) + laneGx;
        unsigned pG1 = pG + RB0;
        asm volatile("" : "+v"(pG1));   // opaque: the two loads of the shared record stay two loads (register tuples)
        bf16x8 F[4][3], G[2][3];
#pragma unroll
        for (int iy = 0; iy < 4; ++iy)
#pragma unroll
          for (int s = 0; s < 3; ++s) F[iy][s] = frag(pF, iy * RB0 + s * SB0, 8);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int s = 0; s < 3; ++s) G[r][s] = frag(r ? pG1 : pG, s * SB0, RB0);
        f32x4 hi[2] = {b0v, b0v}, lo[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        auto pair = [&](const bf16x8 (&f0)[3], int kb0, const bf16x8 (&f1)[3], int kb1) __attribute__((always_inline)) {
          // products (data split s, weight split t), small ones first: (1,1) (0,2) (2,0) (0,1) (1,0) | (0,0)
          lo[0] = MFMA(w0[kb0][1], f0[1], lo[0]); lo[1] = MFMA(w0[kb1][1], f1[1], lo[1]);
          lo[0] = MFMA(w0[kb0][2], f0[0], lo[0]); lo[1] = MFMA(w0[kb1][2], f1[0], lo[1]);
          lo[0] = MFMA(w0[kb0][0], f0[2], lo[0]); lo[1] = MFMA(w0[kb1][0], f1[2], lo[1]);
          lo[0] = MFMA(w0[kb0][1], f0[0], lo[0]); lo[1] = MFMA(w0[kb1][1], f1[0], lo[1]);
          lo[0] = MFMA(w0[kb0][0], f0[1], lo[0]); lo[1] = MFMA(w0[kb1][0], f1[1], lo[1]);
          hi[0] = MFMA(w0[kb0][0], f0[0], hi[0]); hi[1] = MFMA(w0[kb1][0], f1[0], hi[1]);
        };
        pair(F[0], 0, F[3], 2);
        pair(F[1], 1, F[1], 0);
        pair(F[2], 2, F[2], 1);
        pair(G[0], 3, G[1], 3);
#pragma unroll
        for (int r = 0; r < 2; ++r)
          emit(hi[r] + lo[r], zok && xok_row && (unsigned)(Y1 + ry + r) < (unsigned)dW, s1, ry + r, col);
      }
      // single tiles: row 16 (wave 0), column 16 (wave 1), the corner voxel (waves 2 | 3: plane 0 | 1, all lanes the same voxel)
#pragma unroll 1
      for (int pl = 0; pl < 2; ++pl) {
        if (wave >= 2 && pl != wave - 2) continue;
        const int sA = (e + pl) & 3, sB = (sA + 1) & 3, sC = (sA + 2) & 3;
        const int s1 = (m3 + pl) % 3;
        const bool zok = 2 * oz + pl < dD;
        const unsigned pF = (unsigned)((lq == 1 ? sB : lq == 2 ? sC : sA) * PLB0 + sry * RB0 + srx * 8) + laneFx;
        const unsigned pG = (unsigned)((lq < 2 ? sB : sC) * PLB0 + sry * RB0 + srx * 8) + laneGx;
        bf16x8 F[3][3], G[3];
#pragma unroll
        for (int iy = 0; iy < 3; ++iy)
#pragma unroll
          for (int s = 0; s < 3; ++s) F[iy][s] = frag(pF, iy * RB0 + s * SB0, 8);
#pragma unroll
        for (int s = 0; s < 3; ++s) G[s] = frag(pG, s * SB0, RB0);
        f32x4 hi = b0v, lo[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        auto one = [&](const bf16x8 (&f)[3], int kb) __attribute__((always_inline)) {
          lo[0] = MFMA(w0[kb][1], f[1], lo[0]); lo[1] = MFMA(w0[kb][2], f[0], lo[1]);
          lo[0] = MFMA(w0[kb][0], f[2], lo[0]); lo[1] = MFMA(w0[kb][1], f[0], lo[1]);
          lo[0] = MFMA(w0[kb][0], f[1], lo[0]); hi = MFMA(w0[kb][0], f[0], hi);
        };
        one(F[0], 0); one(F[1], 1); one(F[2], 2); one(G, 3);
        emit(hi + (lo[0] + lo[1]), zok && sok, s1, sry, srx);
      }
      __syncthreads();   // B1: ring 1 holds planes 2oz-1, 2oz, 2oz+1; ring-0 slots of planes 2oz-1, 2oz are free

      // ================= phase B: block 1, this wave's K half of its two tiles
      f32x4 hi[2][2], lo[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) { hi[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f}; lo[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      {
        const unsigned so0 = (unsigned)(((m3 + 2) % 3) * PLB1), so1 = (unsigned)(m3 * PLB1), so2 = (unsigned)(((m3 + 1) % 3) * PLB1);
#pragma unroll
        for (int kb = 0; kb < NKB1; ++kb) {
          const unsigned ko = toff[kb] + (tdz[kb] == 0 ? so0 : tdz[kb] == 1 ? so1 : so2);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const unsigned pa = base1[t] + ko;
            bf16x8 f[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) f[s] = frag(pa, s * SPB1, 2 * QS1 * 8);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              lo[t][c] = MFMA(w1[kb][c][1], f[1], lo[t][c]);
              lo[t][c] = MFMA(w1[kb][c][2], f[0], lo[t][c]);
              lo[t][c] = MFMA(w1[kb][c][0], f[2], lo[t][c]);
              lo[t][c] = MFMA(w1[kb][c][1], f[0], lo[t][c]);
              lo[t][c] = MFMA(w1[kb][c][0], f[1], lo[t][c]);
              hi[t][c] = MFMA(w1[kb][c][0], f[0], hi[t][c]);
            }
          }
        }
      }
      // ================= C: the tile this wave does not own -> LDS; next step's input planes -> ring 0
      f32x4 own[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const f32x4 a0 = hi[0][c] + lo[0][c], a1 = hi[1][c] + lo[1][c];
        own[c] = kh ? a1 : a0;
        const f32x4 give = kh ? a0 : a1;
        *reinterpret_cast<f32x4*>(lds + SCR_OFF + ((wave * 2 + c) * 64 + lane) * 16) = give;
      }
      write_ring0(2 * oz + 3, ldn);
      __syncthreads();   // B2
      // ================= D: K half 0 + K half 1 + bias, LeakyReLU, store
      {
        float* const pbase = out + ((int64_t)ub * d.out_bs + (int64_t)oz * d.Wo * d.Ho * 32);
        const __amdgpu_buffer_rsrc_t ores = make_rsrc(pbase, (unsigned)(d.Wo * d.Ho * 32 * 4));
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const f32x4 other = *reinterpret_cast<const f32x4*>(lds + SCR_OFF + (((wave ^ 1) * 2 + c) * 64 + lane) * 16);
          f32x4 v = (own[c] + other) + b1v[c];
          v = __builtin_elementwise_max(v, v * d.slope1);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ores, (int)ooff[c], 0, 0);
        }
      }
      m3 = m3 == 0 ? 2 : m3 - 1;   // (m3 + 2) % 3
    }
  }
}

// packed1[((((kh*7 + kb)*2 + c)*3) + t)*64 + lane]: lane (co = lane & 15, lq = lane >> 4) holds split t of
// W1[c*16 + co][ch = 8*(lq & 1) + e][tap = 14*kh + 2*kb + (lq >> 1)], e = 0..7 (tap 27: zeros)
__global__ void pack_c01_w1_kernel(const float* __restrict__ w, u32x4* __restrict__ packed) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 2 * NKB1 * 2 * 64) return;
  const int lane = idx & 63, c = (idx >> 6) & 1, kb = (idx >> 7) % NKB1, kh = (idx >> 7) / NKB1;
  const int co = lane & 15, lq = lane >> 4;
  const int tap = 14 * kh + 2 * kb + (lq >> 1);
  unsigned r[3][4];
#pragma unroll
  for (int pr = 0; pr < 4; ++pr) {
    float v[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int ch = 8 * (lq & 1) + 2 * pr + hh;
      v[hh] = tap < 27 ? w[((int64_t)(c * 16 + co) * 16 + ch) * 27 + tap] : 0.0f;
    }
    unsigned p[3];
    split3(v[0], v[1], p);
#pragma unroll
    for (int t = 0; t < 3; ++t) r[t][pr] = p[t];
  }
#pragma unroll
  for (int t = 0; t < 3; ++t)
    packed[((((kh * NKB1 + kb) * 2 + c) * 3) + t) * 64 + lane] = (u32x4){r[t][0], r[t][1], r[t][2], r[t][3]};
}

constexpr int64_t W1_FLOATS = (int64_t)2 * NKB1 * 2 * 3 * 64 * 4;

}  // namespace

// ---- C ABI (include/liftreg_hip.h)
extern "C" int64_t lr_conv3d_pair01_packed_floats(int Cin, int C0, int C1) {
  if (Cin < 1 || Cin > 4 || C0 != 16 || C1 != 32) return 0;
  return lr_internal_conv0_split_packed_floats(Cin, C0) + W1_FLOATS;
}

extern "C" int lr_conv3d_pair01_pack_f32(const float* w0, const float* w1, float* packed, int Cin, int C0, int C1, void* stream) {
  if (!w0 || !w1 || !packed) return LR_ENULL;
  if (lr_conv3d_pair01_packed_floats(Cin, C0, C1) == 0) return LR_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(packed) & 15u) return LR_EALIGN;
  hipStream_t st = lr_stream(stream);
  const int rc = lr_internal_conv0_split_pack(w0, packed, Cin, C0, st);
  if (rc != LR_OK) return rc;
  u32x4* p1 = reinterpret_cast<u32x4*>(packed + lr_internal_conv0_split_packed_floats(Cin, C0));
  hipLaunchKernelGGL(pack_c01_w1_kernel, dim3((2 * NKB1 * 2 * 64 + 255) / 256), dim3(256), 0, st, w1, p1);
  return lr_launch_status();
}

extern "C" int lr_conv3d_pair01_f32(const float* in0, int64_t in0_batch_stride, const float* in_rest, int64_t rest_batch_stride,
                                    const float* packed, const float* bias0, const float* bias1, float* out, int B, int Cin,
                                    int D, int W, int H, int out_layout, float slope0, float slope1, int64_t out_batch_stride,
                                    void* stream) {
  if (!in0 || !packed || !out || (Cin > 1 && !in_rest)) return LR_ENULL;
  if (B < 1 || D < 1 || W < 1 || H < 1) return LR_EINVAL;
  if (Cin < 1 || Cin > 4) return LR_EUNSUPPORTED;
  if (out_layout != LR_LAYOUT_NDHWC && out_layout != LR_LAYOUT_NDHWC_HPS) return LR_EUNSUPPORTED;
  if (!(slope0 >= 0.0f && slope0 <= 1.0f) || !(slope1 >= 0.0f && slope1 <= 1.0f)) return LR_EUNSUPPORTED;   // LeakyReLU = max(v, slope v)
  if (H & 3) return LR_EUNSUPPORTED;
  const int64_t V = (int64_t)D * W * H;
  FDims d;
  d.B = B; d.Cin = Cin; d.D = D; d.W = W; d.H = H;
  d.Do = (D - 1) / 2 + 1; d.Wo = (W - 1) / 2 + 1; d.Ho = (H - 1) / 2 + 1;
  d.hps = out_layout == LR_LAYOUT_NDHWC_HPS;
  if (d.hps && (d.Ho & 1)) return LR_EUNSUPPORTED;
  d.bs0 = in0_batch_stride ? in0_batch_stride : V;
  d.bsr = rest_batch_stride ? rest_batch_stride : (int64_t)(Cin - 1) * V;
  const int64_t dense = (int64_t)32 * d.Do * d.Wo * d.Ho;
  if (out_batch_stride != 0 && out_batch_stride < dense) return LR_EINVAL;
  d.out_bs = out_batch_stride ? out_batch_stride : dense;
  if ((reinterpret_cast<uintptr_t>(in0) & 15u) || (Cin > 1 && (reinterpret_cast<uintptr_t>(in_rest) & 15u)) ||
      (reinterpret_cast<uintptr_t>(packed) & 15u) || (reinterpret_cast<uintptr_t>(out) & 15u) || (d.bs0 & 3) || (d.bsr & 3) || (d.out_bs & 3))
    return LR_EALIGN;
  if ((int64_t)3 * V * 4 + (int64_t)8 * W * H * 4 >= 0x7fffffffLL) return LR_EUNSUPPORTED;   // 31-bit byte offsets inside a batch element
  if ((int64_t)d.Wo * d.Ho * 32 * 4 >= 0x7fffffffLL) return LR_EUNSUPPORTED;                // an output plane is one buffer resource
  d.nTx = (d.Ho + TX - 1) / TX; d.nTy = (d.Wo + TY - 1) / TY;
  const int64_t nu = (int64_t)B * d.nTy * d.nTx;
  if (nu > 0x7fffffffLL) return LR_EINVAL;
  d.nunits = (int)nu;
  d.slope0 = slope0; d.slope1 = slope1;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  int blocks = cus;   // one block per CU (LDS)
  if (blocks > d.nunits) blocks = d.nunits;
  hipStream_t st = lr_stream(stream);
  const u32x4* wp0 = reinterpret_cast<const u32x4*>(packed);
  const u32x4* wp1 = reinterpret_cast<const u32x4*>(packed + lr_internal_conv0_split_packed_floats(Cin, 16));
  if (!in_rest) in_rest = in0;   // Cin == 1: never dereferenced (zero-length resource)
#define LR_C01(NCV)                                                                                                          \
  do {                                                                                                                       \
    static std::atomic<uint64_t> attr_done{0};                                                                               \
    if (lr_raise_dyn_lds(reinterpret_cast<const void*>(&conv01_fused_kernel<NCV>), LDSB, attr_done) != LR_OK) return LR_ELAUNCH; \
    hipLaunchKernelGGL((conv01_fused_kernel<NCV>), dim3((unsigned)blocks), dim3(NTHR), LDSB, st, in0, in_rest, wp0, wp1, bias0, bias1, out, d); \
  } while (0)
  if (Cin == 1) LR_C01(1);
  else if (Cin == 2) LR_C01(2);
  else if (Cin == 3) LR_C01(3);
  else LR_C01(4);
#undef LR_C01
  return lr_launch_status();
}
