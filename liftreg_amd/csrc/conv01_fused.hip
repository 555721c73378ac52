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
constexpr int DUMP_OFF = SCR_OFF + SCR;   // 140864: where the threads without a staging item write (no branch in phase B)
constexpr int LDSB = DUMP_OFF + 512;

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

#ifdef LR_C01_STAMPS
// Diagnostic build only (make stamps; tools/c01_stamps.py): per-phase cycle sums, [wave][phase]
__device__ unsigned long long g_lr_c01_stamps[4 * 8];
#define C01_STAMP(slot)                                            \
  do {                                                             \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
    stamp_acc[slot] += now_ - stamp_t;                             \
    stamp_t = now_;                                                \
  } while (0)
#else
#define C01_STAMP(slot) do {} while (0)
#endif

// One scheduling region of the software pipeline: NM MFMAs, each followed by up to NV vector-ALU instructions (the epilogue of
// the unit before: an MFMA holds the vector issue for 8 of its 16 cycles, two 4-cycle instructions fit in its shadow), the
// first NR of them also by one LDS read (the fragments of the unit after), the last NW by one LDS store.
#ifndef LR_C01_SGB
#define LR_C01_SGB 1
#endif
#if LR_C01_SGB
#define LR_C01_SCHED(NM, NR, NV, NW)                                              \
  do {                                                                            \
    _Pragma("unroll") for (int i_ = 0; i_ < (NM); ++i_) {                         \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          \
      if (i_ < (NR)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           \
      __builtin_amdgcn_sched_group_barrier(0x002, (NV), 0);                       \
      if (i_ >= (NM) - (NW)) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   \
    }                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                            \
  } while (0)
#else
#define LR_C01_SCHED(NM, NR, NV, NW) do {} while (0)
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
  // where the block-0 epilogue stores a voxel OUTSIDE the volume: the 4 pad chunks behind row 0 of slot 0 (the same place in
  // each split's plane, SPB1 apart, like the real stores) — never read
  const int dump1 = RING1_OFF + (4 * QS1 + (lane & 3)) * 8;

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
#ifdef LR_C01_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long stamp_t = __builtin_amdgcn_s_memtime();
#endif

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
    auto issue_loads = [&](int zb, u32x4 (&L)[NC]) __attribute__((always_inline)) {
      const int zi = zb + ipl, yi = Y0 + irow, xi = X0a + 4 * iq;
      const int ok = (int)item_live & (int)(zi >= 0) & (int)(zi < dD) & (int)(yi >= 0) & (int)(yi < dW) & (int)(xi >= 0) & (int)(xi < dH);
      const unsigned dead = ((unsigned)ok - 1u) & OOR;   // dead item: bit 31 -> outside the resource -> 0
      const unsigned voff = (unsigned)(((zi * dW + yi) * dH + xi) * 4);
      L[0] = __builtin_amdgcn_raw_buffer_load_b128(r0, (int)(voff | dead), 0, 0);
#pragma unroll
      for (int c = 1; c < NC; ++c) L[c] = __builtin_amdgcn_raw_buffer_load_b128(rr, (int)((voff + (unsigned)(c - 1) * V4) | dead), 0, 0);
    };
    // ... -> split -> ring 0 (plane z sits in slot (z + 1) & 3)
    auto write_ring0 = [&](int zb, const u32x4 (&L)[NC]) __attribute__((always_inline)) {
      const int slot = (zb + ipl + 1) & 3;
      unsigned char* const base = lds + (item_live ? slot * PLB0 + irow * RB0 + iq * 32 : DUMP_OFF);
      unsigned rec[3][4][2];   // [split][voxel][channels 01 | 23]
      float v[4][4];   // [channel][voxel]
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint4 q = __builtin_bit_cast(uint4, L[c < NC ? c : 0]);
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

    // block-0 tile value -> LeakyReLU -> three bf16 -> ring 1 (slot s1, region voxel (ry, rx)).  A voxel outside the volume
    // (block 1's zero padding) keeps the zero the unit prologue wrote: its stores go to the dump area.
    auto emit = [&](f32x4 v, bool ok, int s1, int ry, int rx) __attribute__((always_inline)) {
      v = __builtin_elementwise_max(v, v * d.slope0);   // = LeakyReLU for 0 <= slope <= 1 (launcher)
      unsigned p01[3], p23[3];
      split3(v[0], v[1], p01);
      split3(v[2], v[3], p23);
      const int real = RING1_OFF + s1 * PLB1 + (ry * RS1 + (int)qpos * QS1 + rx) * 8;
      unsigned char* const base = lds + (ok ? real : dump1);
#pragma unroll
      for (int s = 0; s < 3; ++s) *reinterpret_cast<u32x2*>(base + s * SPB1) = (u32x2){p01[s], p23[s]};
    };

    // ---- unit prologue: ring 0 <- planes -1..2, ring 1 <- 0 (plane -1, and every voxel of the column outside the volume)
    __syncthreads();   // (the zero fill of the whole LDS | every read of the unit before)
    {
      u32x4 la[NC], lb[NC];
      issue_loads(-1, la);
      issue_loads(1, lb);
      for (int o = tid * 16; o < RING1; o += NTHR * 16) *reinterpret_cast<u32x4*>(lds + RING1_OFF + o) = (u32x4){0u, 0u, 0u, 0u};
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
    // ---- The step as a HAND-SCHEDULED instruction stream.  One wave per SIMD issues in order: an MFMA holds the vector issue
    // for 8 of its 16 cycles, so about two 4-cycle instructions fit into its shadow — and only if they stand right behind it.
    // Every stage below is therefore a fully unrolled run of single MFMAs, each followed by a SLICE of independent work (the
    // epilogue of the unit before, the fragment reads of the unit after, the step-before's output, the next input planes) and
    // a scheduling barrier that keeps hipcc from regrouping them (left to itself it issues the MFMAs of a unit back to back
    // and the 40 vector instructions of a tile's epilogue behind them: 7.8 k cycles for the 240 MFMAs of phase A).
    //
    // A row PAIR (region rows ry, ry+1; voxels 0..15) shares its window rows: input row iy is tap row iy of output row 0 and
    // iy - 1 of output row 1.  `pl` = plane of the step's pair (z = 2oz + pl).
    struct PairFrags { bf16x8 F[4][3], G[2][3]; };
    struct OneFrags { bf16x8 F[3][3], G[3]; };
    struct Acc2 { f32x4 v[2]; };   // block 0 (K = 81): one fp32 chain per tile, small products first inside each k-block
    struct Base { unsigned pF, pG, pG1; };
    auto pair_base = [&](int e, int pl, int ry) __attribute__((always_inline)) -> Base {
      const int sA = (e + pl) & 3, sB = (sA + 1) & 3, sC = (sA + 2) & 3;
      Base b;
      b.pF = (unsigned)((lq == 1 ? sB : lq == 2 ? sC : sA) * PLB0 + ry * RB0 + col * 8) + laneFx;
      b.pG = (unsigned)((lq < 2 ? sB : sC) * PLB0 + ry * RB0 + col * 8) + laneGx;
      b.pG1 = b.pG + RB0;
      asm volatile("" : "+v"(b.pG1));   // opaque: the two loads of the shared record stay two loads (register tuples)
      return b;
    };
    // fragment n (0..17) of a pair, in the order the MFMAs use them: F0 F3 F1 F2 G0 G1, three splits each
    auto load_pair_n = [&](int n, const Base& b, PairFrags& q) __attribute__((always_inline)) {
      const int grp = n / 3, sp = n % 3;
      if (grp < 4) {
        const int iy = grp == 0 ? 0 : grp == 1 ? 3 : grp == 2 ? 1 : 2;
        q.F[iy][sp] = frag(b.pF, iy * RB0 + sp * SB0, 8);
      } else {
        q.G[grp - 4][sp] = frag(grp == 5 ? b.pG1 : b.pG, sp * SB0, RB0);
      }
    };
    // MFMA I (0..47) of a pair: four groups of 12 = (fragment of row 0, its k-block) paired with (fragment of row 1, its
    // k-block), the two rows alternating; products (weight split, data split) in the order (1,1) (2,0) (0,2) (1,0) (0,1) (0,0)
    auto mma_pair_n = [&](int I, const PairFrags& q, Acc2& a) __attribute__((always_inline)) {
      const int g = I / 12, j = (I % 12) >> 1, r = I & 1;
      const int kb = r == 0 ? g : (g == 0 ? 2 : g == 1 ? 0 : g == 2 ? 1 : 3);
      const int wt = j == 0 ? 1 : j == 1 ? 2 : j == 2 ? 0 : j == 3 ? 1 : 0;
      const int fs = j == 0 ? 1 : j == 1 ? 0 : j == 2 ? 2 : j == 3 ? 0 : j == 4 ? 1 : 0;
      const bf16x8& f = g == 3 ? q.G[r][fs] : q.F[r == 0 ? g : (g == 0 ? 3 : g)][fs];
      a.v[r] = MFMA(w0[kb][wt], f, a.v[r]);
    };
    // single tiles beside the four row pairs: row 16 (wave 0), column 16 (wave 1), the corner voxel (waves 2 and 3, both
    // planes each: the same values to the same place — every wave runs the same 10 tiles)
    const int sry = wave == 1 ? col : 16, srx = wave == 0 ? col : 16;
    const bool sok = (unsigned)(Y1 + sry) < (unsigned)dW && (unsigned)(X1 + srx) < (unsigned)dH;
    auto one_base = [&](int e, int pl) __attribute__((always_inline)) -> Base {
      const int sA = (e + pl) & 3, sB = (sA + 1) & 3, sC = (sA + 2) & 3;
      Base b;
      b.pF = (unsigned)((lq == 1 ? sB : lq == 2 ? sC : sA) * PLB0 + sry * RB0 + srx * 8) + laneFx;
      b.pG = (unsigned)((lq < 2 ? sB : sC) * PLB0 + sry * RB0 + srx * 8) + laneGx;
      b.pG1 = 0;
      return b;
    };
    auto load_one_n = [&](int n, const Base& b, OneFrags& q) __attribute__((always_inline)) {   // n = 0..11: F0 F1 F2 G
      const int grp = n / 3, sp = n % 3;
      if (grp < 3) q.F[grp][sp] = frag(b.pF, grp * RB0 + sp * SB0, 8);
      else q.G[sp] = frag(b.pG, sp * SB0, RB0);
    };
    auto mma_one_n = [&](int I, const OneFrags& q, Acc2& a) __attribute__((always_inline)) {   // I = 0..23: two chains (even | odd k-blocks)
      const int kb = I / 6, j = I % 6;
      const int wt = j == 0 ? 1 : j == 1 ? 2 : j == 2 ? 0 : j == 3 ? 1 : 0;
      const int fs = j == 0 ? 1 : j == 1 ? 0 : j == 2 ? 2 : j == 3 ? 0 : j == 4 ? 1 : 0;
      const bf16x8& f = kb == 3 ? q.G[fs] : q.F[kb][fs];
      a.v[kb & 1] = MFMA(w0[kb][wt], f, a.v[kb & 1]);
    };
    // the epilogue of one block-0 tile in 7 slices: LeakyReLU, the two three-way splits, the three stores into ring 1
    // (a voxel outside the volume keeps the zero of the unit prologue: its stores go to the dump chunks)
    struct Epi { f32x4 x, y; f32x2 r; u32x2 s0, s1, s2; };   // s_k = the 8-byte record of split k: (channels 01 | channels 23)
    struct L0 { unsigned p; f32x2 r; };
    struct L12 { unsigned p1, p2; };
    auto lvl0 = [](float a, float b2) __attribute__((always_inline)) -> L0 {
      const f32x2 v = {a, b2};
      L0 o;
      o.p = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
      o.r = v - (f32x2){__builtin_bit_cast(float, o.p << 16), __builtin_bit_cast(float, o.p & 0xffff0000u)};
      return o;
    };
    auto lvl12 = [](f32x2 r) __attribute__((always_inline)) -> L12 {
      L12 o;
      o.p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
      r = r - (f32x2){__builtin_bit_cast(float, o.p1 << 16), __builtin_bit_cast(float, o.p1 & 0xffff0000u)};
      o.p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
      return o;
    };
    auto epi_slice = [&](int sl, Epi& E, const f32x4& acc, int addr) __attribute__((always_inline)) {
      if (sl == 0) { E.x = acc; E.y = E.x * d.slope0; }
      else if (sl == 1) E.x = __builtin_elementwise_max(E.x, E.y);   // = LeakyReLU for 0 <= slope <= 1 (launcher)
      else if (sl == 2) { const L0 t = lvl0(E.x[0], E.x[1]); E.s0 = (u32x2){t.p, 0u}; E.r = t.r; }
      else if (sl == 3) { const L12 t = lvl12(E.r); E.s1 = (u32x2){t.p1, 0u}; E.s2 = (u32x2){t.p2, 0u}; }
      else if (sl == 4) { const L0 t = lvl0(E.x[2], E.x[3]); E.s0[1] = t.p; E.r = t.r; }
      else if (sl == 5) { const L12 t = lvl12(E.r); E.s1[1] = t.p1; E.s2[1] = t.p2; }
      else {
        *reinterpret_cast<u32x2*>(lds + addr) = E.s0;
        *reinterpret_cast<u32x2*>(lds + addr + SPB1) = E.s1;
        *reinterpret_cast<u32x2*>(lds + addr + 2 * SPB1) = E.s2;
      }
    };
    const bool xok_row = (unsigned)(X1 + col) < (unsigned)dH;   // row tiles: rx = col
    auto tile_addr = [&](bool ok, int s1, int ry, int rx) __attribute__((always_inline)) -> int {
      return ok ? RING1_OFF + s1 * PLB1 + (ry * RS1 + (int)qpos * QS1 + rx) * 8 : dump1;
    };
    // D of a step in 5 slices: K half 0 + K half 1 + bias, LeakyReLU, store (oz = -1: nothing yet — zero-length resource)
    f32x4 own[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    struct Fin { f32x4 o[2]; };
    auto fin_slice = [&](int sl, Fin& F, const __amdgpu_buffer_rsrc_t& ores) __attribute__((always_inline)) {
      if (sl == 0) {
#pragma unroll
        for (int c = 0; c < 2; ++c) F.o[c] = *reinterpret_cast<const f32x4*>(lds + SCR_OFF + (((wave ^ 1) * 2 + c) * 64 + lane) * 16);
      } else if (sl == 1 || sl == 3) {
        const int c = sl >> 1;
        F.o[c] = (own[c] + F.o[c]) + b1v[c];
      } else {
        const int c = (sl >> 1) - 1;
        const f32x4 v = __builtin_elementwise_max(F.o[c], F.o[c] * d.slope1);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ores, (int)ooff[c], 0, 0);
      }
    };
    auto fin_rsrc = [&](int oz) __attribute__((always_inline)) -> __amdgpu_buffer_rsrc_t {
      float* const pbase = out + ((int64_t)ub * d.out_bs + (int64_t)(oz < 0 ? 0 : oz) * d.Wo * d.Ho * 32);
      return make_rsrc(pbase, oz < 0 ? 0u : (unsigned)(d.Wo * d.Ho * 32 * 4));
    };
#define C01_FENCE() __builtin_amdgcn_sched_barrier(0)

    int m3 = 0;   // (2 oz) mod 3: ring-1 slot of plane 2oz; plane 2oz-1 sits in (m3 + 2) % 3, plane 2oz+1 in (m3 + 1) % 3
    for (int oz = 0; oz < d.Do; ++oz) {
      const int e = (2 * oz) & 3;   // ring-0 slot of plane 2oz-1
      u32x4 ldn[NC];
      issue_loads(2 * oz + 3, ldn);
      C01_STAMP(0);
      if (2 * oz + 1 >= dD)   // odd D, last step: plane 2oz+1 lies below the volume — its slot must read 0 (its stores are dropped)
        for (int o = tid * 16; o < PLB1; o += NTHR * 16) *reinterpret_cast<u32x4*>(lds + RING1_OFF + ((m3 + 1) % 3) * PLB1 + o) = (u32x4){0u, 0u, 0u, 0u};

      // ================= phase A: block 0, planes 2oz and 2oz+1.  Units of a wave: row pairs U0..U3 = (plane 0 | 1) x (rows 4w |
      // 4w+2), then its single tile of plane 0 and of plane 1.  Stage k = MFMAs of unit k | epilogue of unit k-1 | fragments of k+1.
      {
        PairFrags fa, fb;
        OneFrags sa, sb;
        Acc2 aa, ab;
        Epi E0, E1;
        Fin FN;
        const int ry0 = 4 * wave, ry1 = 4 * wave + 2;
        const bool zok0 = 2 * oz < dD, zok1 = 2 * oz + 1 < dD;
        const int s10 = m3, s11 = (m3 + 1) % 3;
        const bool yok[4] = {(unsigned)(Y1 + ry0) < (unsigned)dW, (unsigned)(Y1 + ry0 + 1) < (unsigned)dW,
                             (unsigned)(Y1 + ry1) < (unsigned)dW, (unsigned)(Y1 + ry1 + 1) < (unsigned)dW};
        // ring-1 addresses of the ten tiles: [unit][row]
        const int adr[4][2] = {{tile_addr(zok0 && xok_row && yok[0], s10, ry0, col), tile_addr(zok0 && xok_row && yok[1], s10, ry0 + 1, col)},
                               {tile_addr(zok0 && xok_row && yok[2], s10, ry1, col), tile_addr(zok0 && xok_row && yok[3], s10, ry1 + 1, col)},
                               {tile_addr(zok1 && xok_row && yok[0], s11, ry0, col), tile_addr(zok1 && xok_row && yok[1], s11, ry0 + 1, col)},
                               {tile_addr(zok1 && xok_row && yok[2], s11, ry1, col), tile_addr(zok1 && xok_row && yok[3], s11, ry1 + 1, col)}};
        const int adr_s0 = tile_addr(zok0 && sok, s10, sry, srx), adr_s1 = tile_addr(zok1 && sok, s11, sry, srx);
        const __amdgpu_buffer_rsrc_t ores = fin_rsrc(oz - 1);
        const Base bU0 = pair_base(e, 0, ry0), bU1 = pair_base(e, 0, ry1), bU2 = pair_base(e, 1, ry0), bU3 = pair_base(e, 1, ry1);
        const Base bS0 = one_base(e, 0), bS1 = one_base(e, 1);
        // the first unit's fragments: the one exposed LDS round trip of the step
#pragma unroll
        for (int n = 0; n < 18; ++n) load_pair_n(n, bU0, fa);
        C01_FENCE();
        // stage 0: U0 | the step-before's output | fragments of U1
        aa.v[0] = b0v; aa.v[1] = b0v;
#pragma unroll
        for (int I = 0; I < 48; ++I) {
          mma_pair_n(I, fa, aa);
          if ((I & 1) == 0 && I < 36) load_pair_n(I >> 1, bU1, fb);
          if (I >= 36 && I < 46 && (I & 1) == 0) fin_slice((I - 36) >> 1, FN, ores);
          C01_FENCE();
        }
        // stages 1..3: a pair | the epilogue of the pair before (14 slices) | fragments of the unit after
#define C01_PAIR_STAGE(FC, AC, AP, ADRP, LOADN)                                         \
        AC.v[0] = b0v; AC.v[1] = b0v;                                                   \
        _Pragma("unroll") for (int I = 0; I < 48; ++I) {                                \
          mma_pair_n(I, FC, AC);                                                        \
          if ((I & 1) == 0 && I < 36) { LOADN; }                                        \
          if (I >= 3 && I < 45 && (I % 3) == 0) {                                       \
            const int k_ = I / 3 - 1;                                                   \
            if (k_ < 7) epi_slice(k_, E0, AP.v[0], ADRP[0]); else epi_slice(k_ - 7, E1, AP.v[1], ADRP[1]); \
          }                                                                             \
          C01_FENCE();                                                                  \
        }
        C01_PAIR_STAGE(fb, ab, aa, adr[0], load_pair_n(I >> 1, bU2, fa));
        C01_PAIR_STAGE(fa, aa, ab, adr[1], load_pair_n(I >> 1, bU3, fb));
        C01_PAIR_STAGE(fb, ab, aa, adr[2], if (I < 24) load_one_n(I >> 1, bS0, sa));
#undef C01_PAIR_STAGE
        // stage 4: S0 | epilogue of U3 (14 slices over 24 MFMAs) | fragments of S1
        aa.v[0] = b0v; aa.v[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int I = 0; I < 24; ++I) {
          mma_one_n(I, sa, aa);
          if ((I & 1) == 0) load_one_n(I >> 1, bS1, sb);
          if (I >= 2 && I < 23 && (I % 3) != 1) {   // I = 2,3,5,6,...,20,21: slices 0..13
            const int k_ = (I - 2) / 3 * 2 + ((I - 2) % 3 == 0 ? 0 : 1);
            if (k_ < 7) epi_slice(k_, E0, ab.v[0], adr[3][0]); else epi_slice(k_ - 7, E1, ab.v[1], adr[3][1]);
          }
          C01_FENCE();
        }
        // stage 5: S1 | epilogue of S0 (7 slices)
        const f32x4 accS0 = aa.v[0] + aa.v[1];
        ab.v[0] = b0v; ab.v[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int I = 0; I < 24; ++I) {
          mma_one_n(I, sb, ab);
          if (I >= 2 && I < 23 && (I % 3) == 2) epi_slice((I - 2) / 3, E0, accS0, adr_s0);
          C01_FENCE();
        }
        // tail: epilogue of S1
        const f32x4 accS1 = ab.v[0] + ab.v[1];
#pragma unroll
        for (int k = 0; k < 7; ++k) epi_slice(k, E1, accS1, adr_s1);
      }
      C01_STAMP(1);
      C01_STAMP(2);
      __syncthreads();   // B1: ring 1 holds planes 2oz-1, 2oz, 2oz+1; ring-0 slots of planes 2oz-1, 2oz are free

      C01_STAMP(3);
#pragma unroll
      for (int c = 0; c < NC; ++c) asm volatile("" : "+v"(ldn[c]));   // the split of the next planes belongs to phase B's MFMA shadow
      // ================= phase B: block 1, this wave's K half of its two tiles: 14 groups (k-block, tile) of 12 MFMAs; the
      // three fragments of group g+1 are read behind MFMAs 1, 5, 9 of group g; the next step's input planes are split and
      // written into ring 0 in slices behind every 12th MFMA
      f32x4 hi[2][2], lo[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) { hi[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f}; lo[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      {
        const unsigned so0 = (unsigned)(((m3 + 2) % 3) * PLB1), so1 = (unsigned)(m3 * PLB1), so2 = (unsigned)(((m3 + 1) % 3) * PLB1);
        unsigned pa[NKB1][2];
#pragma unroll
        for (int kb = 0; kb < NKB1; ++kb) {
          const unsigned ko = toff[kb] + (tdz[kb] == 0 ? so0 : tdz[kb] == 1 ? so1 : so2);
          pa[kb][0] = base1[0] + ko; pa[kb][1] = base1[1] + ko;
        }
        bf16x8 fr[2][3];
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) fr[0][sp] = frag(pa[0][0], sp * SPB1, 2 * QS1 * 8);
        C01_FENCE();
        // ring-0 staging of this thread's item, sliced: per voxel j two splits (channels 01 | 23), then the six stores
        float sv[4][4];
        unsigned rec[3][4][2];
        const int r0slot = (2 * oz + 3 + ipl + 1) & 3;
        const int r0addr = item_live ? r0slot * PLB0 + irow * RB0 + iq * 32 : DUMP_OFF;
#pragma unroll
        for (int g = 0; g < 2 * NKB1; ++g) {
          const int kb = g >> 1, t = g & 1;
#pragma unroll
          for (int j = 0; j < 12; ++j) {
            const int c = j & 1, pj = j >> 1;
            const int wt = pj == 0 ? 1 : pj == 1 ? 2 : pj == 2 ? 0 : pj == 3 ? 1 : 0;
            const int fs = pj == 0 ? 1 : pj == 1 ? 0 : pj == 2 ? 2 : pj == 3 ? 0 : pj == 4 ? 1 : 0;
            if (pj == 5) hi[t][c] = MFMA(w1[kb][c][0], fr[g & 1][0], hi[t][c]);
            else lo[t][c] = MFMA(w1[kb][c][wt], fr[g & 1][fs], lo[t][c]);
            if (g + 1 < 2 * NKB1 && (j == 1 || j == 5 || j == 9)) {
              const int sp = j >> 2;
              fr[(g + 1) & 1][sp] = frag(pa[(g + 1) >> 1][(g + 1) & 1], sp * SPB1, 2 * QS1 * 8);
            }
            if (j == 3 || j == 7 || j == 11) {   // 42 slots; 15 used
              const int slot = g * 3 + (j >> 2);
              if (slot == 0) {
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                  const uint4 q = __builtin_bit_cast(uint4, ldn[cc < NC ? cc : 0]);
                  sv[cc][0] = cc < NC ? __builtin_bit_cast(float, q.x) : 0.0f; sv[cc][1] = cc < NC ? __builtin_bit_cast(float, q.y) : 0.0f;
                  sv[cc][2] = cc < NC ? __builtin_bit_cast(float, q.z) : 0.0f; sv[cc][3] = cc < NC ? __builtin_bit_cast(float, q.w) : 0.0f;
                }
              } else if (slot >= 1 && slot <= 8) {
                const int vj = (slot - 1) >> 1;
                unsigned pp[3] = {0u, 0u, 0u};
                if ((slot - 1) & 1) { if (NC > 2) split3(sv[2][vj], sv[3][vj], pp); }
                else split3(sv[0][vj], sv[1][vj], pp);
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) rec[sp][vj][(slot - 1) & 1] = pp[sp];
              } else if (slot >= 9 && slot <= 11) {
                const int sp = slot - 9;
                *reinterpret_cast<u32x4*>(lds + r0addr + sp * SB0) = (u32x4){rec[sp][0][0], rec[sp][0][1], rec[sp][1][0], rec[sp][1][1]};
                *reinterpret_cast<u32x4*>(lds + r0addr + sp * SB0 + 16) = (u32x4){rec[sp][2][0], rec[sp][2][1], rec[sp][3][0], rec[sp][3][1]};
              }
            }
            C01_FENCE();
          }
        }
      }
      C01_STAMP(4);
      // ================= C: the tile this wave does not own -> LDS
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const f32x4 a0 = hi[0][c] + lo[0][c], a1 = hi[1][c] + lo[1][c];
        own[c] = kh ? a1 : a0;
        const f32x4 give = kh ? a0 : a1;
        *reinterpret_cast<f32x4*>(lds + SCR_OFF + ((wave * 2 + c) * 64 + lane) * 16) = give;
      }
      C01_STAMP(5);
      __syncthreads();   // B2
      C01_STAMP(6);
      m3 = m3 == 0 ? 2 : m3 - 1;   // (m3 + 2) % 3
      C01_STAMP(7);
    }
    {   // the last step's output
      Fin FN;
      const __amdgpu_buffer_rsrc_t ores = fin_rsrc(d.Do - 1);
#pragma unroll
      for (int k = 0; k < 5; ++k) fin_slice(k, FN, ores);
    }
#undef C01_FENCE
  }
#ifdef LR_C01_STAMPS
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(&g_lr_c01_stamps[wave * 8 + i], stamp_acc[i]);
  }
#endif
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

#ifdef LR_C01_STAMPS
extern "C" int lr_debug_read_c01_stamps(unsigned long long* host32, int reset) {
  if (hipMemcpyFromSymbol(host32, HIP_SYMBOL(g_lr_c01_stamps), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_lr_c01_stamps), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
#endif

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
