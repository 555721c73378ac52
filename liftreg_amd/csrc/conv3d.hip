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

#ifdef LR_CONV0_STAMPS
// Diagnostic build only (make stamps): per-phase cycle sums of the persistent planar kernel.
__device__ unsigned long long g_lr_stamps[8];
extern "C" int lr_debug_read_stamps(unsigned long long* host8, int reset) {
  if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_lr_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_lr_stamps), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
#define LR_STAMP(slot)                                                         \
  do {                                                                         \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();              \
    stamp_acc[slot] += now_ - stamp_t;                                         \
    stamp_t = now_;                                                            \
  } while (0)
#else
#define LR_STAMP(slot) do {} while (0)
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifndef LR_STORE_AUX
#define LR_STORE_AUX 2  /* nt: block 0 writes 8.6 GB per batch that nothing re-reads before it leaves the L2 */
#endif

struct ConvDims {
  int B, Cin, Cout, D, W, H, Do, Wo, Ho;
  int nHq, nWq, nDq;
  long long out_bs;   // output elements between two batch elements (dense: Cout*Do*Wo*Ho; larger: `out` is a plane range of a
                      // bigger per-sample buffer, parallel.SlabShardedRegistration)
  long long in0_bs;   // split input: elements between two batch elements of `in0` (dense: D*W*H; larger: in0 is a z-slab view
                      // of the whole moving volume)
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
// OUTL >= 0 fixes the layout at compile time (no branch per store); -1 = the runtime value.
template <int OUTL = -1>
__device__ __forceinline__ void store_tile(const f32x4& acc, float* __restrict__ out,
                                           const ConvDims& d, int b, int dz, int wo, int ho, int nt,
                                           int lane, int out_layout_rt, float slope) {
  const int out_layout = OUTL >= 0 ? OUTL : out_layout_rt;
  if (wo >= d.Wo || ho >= d.Ho) return;
  const int c0 = nt * 16 + (lane >> 4) * 4;
  f32x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = lrelu(acc[r], slope);
  if (out_layout == LR_LAYOUT_NDHWC) {
    float* o = out + (int64_t)b * d.out_bs + (((int64_t)dz * d.Wo + wo) * d.Ho + ho) * d.Cout + c0;
    *reinterpret_cast<f32x4*>(o) = v;
  } else if (out_layout == LR_LAYOUT_BF16_NDHWC || out_layout == LR_LAYOUT_BF16_NDHWC_HPS) {
    // bf16 storage (C4/C5): 4 couts of one voxel = one 8-byte store; rows plain or [parity][Ho/2][Cout]
    const int hp = out_layout == LR_LAYOUT_BF16_NDHWC_HPS ? (ho & 1) * (d.Ho >> 1) + (ho >> 1) : ho;
    unsigned short* o = reinterpret_cast<unsigned short*>(out) +
                        (int64_t)b * d.out_bs + (((int64_t)dz * d.Wo + wo) * d.Ho + hp) * d.Cout + c0;
    unsigned short h[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = __builtin_bit_cast(unsigned short, (__bf16)v[r]);  // round to nearest even
    *reinterpret_cast<uint2*>(o) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
  } else if (out_layout == LR_LAYOUT_NDHWC_HPS) {  // even voxels of the row first, then the odd ones
    // row = [channel block of 16][parity][Ho/2][16 floats]
    const int hp = (ho & 1) * (d.Ho >> 1) + (ho >> 1);
    float* o = out + (int64_t)b * d.out_bs + ((int64_t)dz * d.Wo + wo) * d.Ho * d.Cout + ((c0 >> 4) * d.Ho + hp) * 16 + (c0 & 15);
    *reinterpret_cast<f32x4*>(o) = v;
  } else {
    const int64_t vo = (int64_t)d.Do * d.Wo * d.Ho;
    float* o = out + (int64_t)b * d.out_bs + (int64_t)c0 * vo + ((int64_t)dz * d.Wo + wo) * d.Ho + ho;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r * vo] = v[r];
  }
}

// The same tile through an UNCONDITIONAL bounds-checked buffer store (fp32 channels-last layouts): `plane` is the
// resource of output plane (b, dz) — zero-length when the plane does not exist — and a lane outside the row/column
// range gets an out-of-range offset; the hardware drops it.  No branch around the store, so the compiler can COUNT the
// stores in flight (conditional ones force it to wait for all of them before the next dependent load is consumed).
template <int OUTL>
__device__ __forceinline__ void store_tile_buf(const f32x4& acc, const __amdgpu_buffer_rsrc_t plane, const ConvDims& d,
                                               int wo, int ho, int nt, int lane, float slope) {
  static_assert(OUTL == LR_LAYOUT_NDHWC || OUTL == LR_LAYOUT_NDHWC_HPS, "fp32 channels-last output");
  const int c0 = nt * 16 + (lane >> 4) * 4;
  float4 v;
  v.x = lrelu(acc[0], slope); v.y = lrelu(acc[1], slope); v.z = lrelu(acc[2], slope); v.w = lrelu(acc[3], slope);
  unsigned off;
  if (OUTL == LR_LAYOUT_NDHWC) {
    off = (unsigned)(((wo * d.Ho + ho) * d.Cout + c0) * 4);
  } else {  // row = [channel block of 16][parity][Ho/2][16 floats]
    const int hp = (ho & 1) * (d.Ho >> 1) + (ho >> 1);
    off = (unsigned)((wo * d.Ho * d.Cout + ((c0 >> 4) * d.Ho + hp) * 16 + (c0 & 15)) * 4);
  }
  if (wo >= d.Wo || ho >= d.Ho) off = 0x80000000u;
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), plane, off, 0, LR_STORE_AUX);
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
constexpr int PD = 4, PW = 4, PH = 64;
#ifndef LR_C0_WINO_BLOCKS
#define LR_C0_WINO_BLOCKS 3   // resident blocks per CU of the Winograd first-block kernel (138 registers; 4 = a 128-register build: 8 spills, 3.18 vs 3.07 ms)
#endif

// CC = channels staged per pass: 3 for stride 1 (P=2 views -> Cin=3: a single pass, 31 KB of LDS),
// 1 for stride 2 (its brick is 4.4x larger; off the model's path).
template <int S, int CC>
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
  static constexpr int NROWS = CC * RD * RW;     // rows per pass
  static constexpr int MAXIT = (NROWS + 4 * RPI - 1) / (4 * RPI);  // staging iterations per wave
  static constexpr int T = CC * 7;               // k-steps (tap quads) per pass
};

// Persistent: a block walks work items (brick, channel pass) with stride gridDim.x.  While item i
// runs on the matrix pipe the input window of item i+1 is already in flight (bounds-checked buffer
// loads: out-of-volume rows/columns and channels >= Cin get an out-of-range offset and read 0, so
// the loads are unconditional and the brick's zero halo IS the conv's padding).
// SINGLE (Cin <= CC, one channel pass per brick — the model's case): the sweep is output-stationary per
// PAIR of tiles: 2 live accumulators instead of 16, and each pair is stored the moment its CC*7 k-steps
// are done, so the 64 KB a block writes per brick drains under the MFMAs of the following pairs instead of
// stalling the wave in one 16-store burst (stamped build: that burst cost as much as the whole sweep).
// WINO (the model's first block: stride 1, Cin <= 3, 16 couts, channels-last output): the sweep runs the Winograd F(2,3)
// minimal-filtering algorithm ALONG H — two neighbouring outputs of a row share the four inputs d0..d3 under their taps,
//   m0 = (d0-d2) g0,  m1 = (d1+d2) (g0+g1+g2)/2,  m2 = (d2-d1) (g0-g1+g2)/2,  m3 = (d1-d3) g2,
//   y0 = m0+m1+m2,    y1 = m1-m2-m3                (g = the three taps along H of one (channel, tz, ty))
// i.e. 4 multiplications for 2 outputs instead of 6: the contraction index shrinks from 81 (c, 27 taps) to 4 x 27
// (position r, (c,tz,ty)) per PAIR of outputs = 28 MFMAs per 32 outputs instead of 42 — one third fewer of the
// instructions that bound this block (fp32 MFMA runs at 1/16 of the bf16 rate on gfx950; at the clock the chip holds the
// direct sweep alone is 2.74 of the block's 3.3 ms).  Columns of an MFMA = 16 output PAIRS of a row; a lane reads d0..d3
// once (two ds_read2_b32), forms the four differences on the vector ALU and feeds four MFMAs (r = 0..3) with the
// transformed weights U_r (packed by lr_conv3d_pack_weights_f32 behind the direct ones); the output transform is four
// adds per accumulator.  fp32 throughout; results differ from the direct sum by rounding (~1e-7 relative: the data-side
// transform has coefficients +-1 only).  The parity-split output row makes both stores of a pair contiguous KiBs.
template <int NT, int S, int CC, bool SINGLE, int OUTL = -1, bool MASK = false, bool WINO = false>
__global__ __launch_bounds__(256, (SINGLE ? (WINO ? LR_C0_WINO_BLOCKS : 3) : 2)) void conv3d_planar_kernel(const float* __restrict__ in,
                                                            const float* __restrict__ wp,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ out, ConvDims d,
                                                            int out_layout, float slope, int vec4_rt,
                                                            int nitems, int npass, int dbg_rt,
                                                            const float* __restrict__ in0 /* or null: see below */,
                                                            unsigned char* __restrict__ mask_out = nullptr) {
  // MASK (training forward, 16 couts): besides the activation, one byte per (output voxel, channel quad) whose bit r
  // says "channel 4q+r's output is > 0" (LR_LAYOUT_SIGN4) — the LeakyReLU mask the NEXT block's data gradient
  // multiplies by.  That gradient kernel used to re-read this block's whole fp32 output (8.6 GB per batch at C3) for
  // nothing but these signs.  A lane's four accumulators ARE one such quad: no cross-lane work, one unconditional
  // bounds-checked byte store per tile (a wave's 64 bytes are contiguous).
#ifdef LR_C0_STATIC   /* experiment: compile-time staging mode and no ablation switches (clean control flow) */
  constexpr int vec4 = 1, dbg = 0;
  (void)vec4_rt; (void)dbg_rt;
#else
  // The Winograd instances are only launched with 16-byte staging, and their timing-only ablation switches exist in a
  // diagnostic build only (-DLR_CONV0_DBG): both as compile-time constants, no branch on a kernel argument sits between
  // the prefetch of the next brick and the wait for it (at such a join hipcc waits for vmcnt(0) = for the sweep's stores).
#ifdef LR_CONV0_DBG
  const int vec4 = WINO ? 1 : vec4_rt, dbg = dbg_rt;
#else
  const int vec4 = WINO ? 1 : vec4_rt, dbg = WINO ? 0 : dbg_rt;
#endif
#endif
  // in0 != null ("split input", one pass, 16-byte staging only): channel 0 is read from in0 (B,1,D,W,H) and channels
  // 1.. from `in` (B,Cin-1,D,W,H) — the encoder's cat([moving, backprojected views]) without the copy of `moving`.
  using G = PlanarGeom<S, CC>;
  extern __shared__ __attribute__((aligned(16))) float brick[];  // [CC][RD][RW][RSL]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kq = lane >> 4;
  constexpr unsigned OOR = 0x80000000u;
  const int64_t V = (int64_t)d.D * d.W * d.H;

  // per-lane LDS offsets of the 7 tap quads (tap 27 is padding: weight 0, address of tap 26)
  int qoff[7];
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    const int tap = min(q * 4 + kq, 26);
    const int tz = tap / 9, ty = (tap / 3) % 3, tx = tap % 3;
    qoff[q] = (wave * S + tz) * G::PS + ty * G::RSL + G::XOFF + tx + col * S;
  }
  // staging slots of this lane: slot `it` covers window row (it*4+wave)*RPI + lrow, float4 lf4.
  // Row decode and the offset relative to the window origin do not depend on the brick.
  const int lrow = lane / G::F4, lf4 = lane - lrow * G::F4;
  const bool lact = lane < G::RPI * G::F4;
  // The RPI rows of one wavefront load lie in ONE window plane of ONE channel (RW % RPI == 0): which channel / plane / first
  // row a slot covers depends on (it, wave) only — scalars — and a lane adds its own row and float4.  (Per-slot VGPR
  // descriptors — five arrays of MAXIT — cost this kernel 35 registers and its fourth resident block per CU.)
  static_assert(G::RW % G::RPI == 0 && G::NROWS % G::RPI == 0, "a load instruction stays inside one window plane");
  const unsigned lane_rel = (unsigned)((lrow * d.H + lf4 * 4) * 4);   // byte offset of the lane inside its slot's first row
  const int lane_dst = lrow * G::RSL + lf4 * 4;                        // LDS float index, likewise
  auto slot_cc = [&](int it) { return ((it * 4 + wave) * G::RPI) / (G::RD * G::RW); };
  auto slot_rz = [&](int it) { return (((it * 4 + wave) * G::RPI) / G::RW) % G::RD; };
  auto slot_ry = [&](int it) { return ((it * 4 + wave) * G::RPI) % G::RW; };
  auto slot_used = [&](int it) { return (it * 4 + wave) * G::RPI < G::NROWS; };

  auto item_coords = [&](int item, int& b, int& dq, int& wq, int& hq, int& pass) {
    pass = item % npass;
    int br = item / npass;
    hq = br % d.nHq; br /= d.nHq;
    wq = br % d.nWq; br /= d.nWq;
    dq = br % d.nDq;
    b = br / d.nDq;
  };
  float4 st[G::MAXIT];
  auto prefetch = [&](int item) {
    int b, dq, wq, hq, pass;
    item_coords(item, b, dq, wq, hq, pass);
    const int c0 = pass * CC;
    const int z0 = dq * PD * S - 1, y0 = wq * PW * S - 1, x0 = hq * PH * S - 1 - G::XOFF;
    // window origin (may lie before the tensor: never dereferenced there)
    const int64_t wofs = ((int64_t)z0 * d.W + y0) * d.H + x0;
    const float* org = in + ((int64_t)b * (in0 ? d.Cin - 1 : d.Cin) + c0) * V + wofs;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(org), (short)0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_c0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(in0 ? in0 + (int64_t)b * d.in0_bs + wofs : org), (short)0, 0x7fffffff, 0x00020000);
    // ONE straight run of loads (no interior/edge branch: at a join the compiler can no longer count what is in
    // flight and drains the queue — prefetch included — at the next wait); the edge test costs ~8 ALU ops per slot
    const int xi = x0 + lf4 * 4;
    const bool xok = xi >= 0 && xi + 3 < d.H;  // multiple of 4 and H % 4 == 0: entirely in or out
    const __amdgpu_buffer_rsrc_t rsrc_null = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(org), (short)0, 0, 0x00020000);
#pragma unroll
    for (int it = 0; it < G::MAXIT; ++it) {
      const int cc = slot_cc(it), rz = slot_rz(it), ry0 = slot_ry(it);   // wave-uniform
      const int zi = z0 + rz, yi = y0 + ry0 + lrow;
      const bool sok = slot_used(it) & (zi >= 0) & (zi < d.D) & (c0 + cc < d.Cin);   // scalar: else the zero-length resource
      const bool ok = lact & xok & (yi >= 0) & (yi < d.W);
      const int ccr = (in0 && cc > 0) ? cc - 1 : cc;  // split input: channel index inside its own tensor
      const unsigned soff = (unsigned)(((int64_t)ccr * V + ((int64_t)rz * d.W + ry0) * d.H) * 4);
      // a slot's rows belong to ONE channel: the resource choice is wave-uniform
      st[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(!sok ? rsrc_null : cc == 0 ? rsrc_c0 : rsrc,
                                                                                 ok ? lane_rel : OOR, soff, 0));
    }
  };

  // Work order of a block: bricks brick0, brick0+G, ... and for each brick its npass channel passes
  // back to back (the accumulators carry across the passes of one brick).
  const int nbricks = nitems / npass;
  const int brick0 = (int)lr_xcd_remap(blockIdx.x, gridDim.x);
  if (brick0 >= nbricks) return;
  const int my_bricks = (nbricks - brick0 + (int)gridDim.x - 1) / (int)gridDim.x;
  const int my_units = my_bricks * npass;
  auto unit_item = [&](int u) { return (brick0 + (u / npass) * (int)gridDim.x) * npass + u % npass; };
  f32x4 acc[PW * 4][NT];
  float w[G::T][NT];
  float a[2][PW * 4];
  float uw[WINO ? 4 : 1][WINO ? 7 : 1];  // WINO: transformed weights U_r of k-quad q (lane: cout = lane&15, k = 4q + kq)
  // WINO: LDS offset of d0 for this lane's k of quad q (output pair `col`), for output rows un>>1 even | odd.  Window row
  // (z, y) is stored shifted right by (z + y) & 1 floats: the two lane groups a ds_read_b32 serves together (k and k + 1 =
  // neighbouring rows in y, or (z + 1, y - 2)) then sit on banks of opposite parity — with every row stride a multiple of
  // four floats and the lanes two floats apart, all 32 of them hit the same 16 banks (PMC: 3.4-4.8 conflict cycles per LDS
  // instruction, the review's item).
  int woff[WINO ? 7 : 1], woffo[WINO ? 7 : 1];
  if constexpr (WINO) {
    static_assert(NT == 1 && S == 1 && CC == 3 && SINGLE, "Winograd sweep: the model's first block");
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int k = min(q * 4 + kq, 26);   // k = (c, tz, ty); k = 27 is padding (U = 0), address of k = 26
      const int c = k / 9, tz = (k / 3) % 3, ty = k % 3;
      const int w0 = c * G::CS + (wave + tz) * G::PS + ty * G::RSL + G::XOFF + 2 * col, pl = (wave + tz + ty) & 1;
      woff[q] = w0 + pl;
      woffo[q] = w0 + (pl ^ 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) uw[r][q] = wp[d.Cin * 7 * 64 + (r * 7 + q) * 64 + lane];
    }
#pragma unroll
    for (int q = 0; q < 7; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(uw[r][q]));
  }

  // brick of unit u -> LDS (from the prefetched registers, or scalar loads when !vec4)
  auto stage = [&](int u) {
    if (vec4) {
      // no branch around the writes (a lane or slot without a row loaded zeros and writes them into the dump area behind
      // the brick): every path carries the same waits for the prefetch, so they can leave the sweep's stores in flight
#pragma unroll
      for (int it = 0; it < G::MAXIT; ++it) {
        const int dst = (slot_used(it) && lact) ? slot_cc(it) * G::CS + slot_rz(it) * G::PS + slot_ry(it) * G::RSL + lane_dst : CC * G::CS;
        if constexpr (WINO) {   // rows shifted by (z + y) & 1 floats (see woff): four dword stores, any alignment
          float* pd = brick + dst + ((slot_rz(it) + slot_ry(it) + lrow) & 1);
          pd[0] = st[it].x; pd[1] = st[it].y; pd[2] = st[it].z; pd[3] = st[it].w;
        } else {
          *reinterpret_cast<float4*>(brick + dst) = st[it];
        }
      }
    } else {  // H % 4 != 0 or unaligned base: scalar staging of the same window, no prefetch
      int b, dq, wq, hq, pass;
      item_coords(unit_item(u), b, dq, wq, hq, pass);
      const int c0 = pass * CC;
      const int z_in0 = dq * PD * S - 1, y_in0 = wq * PW * S - 1, x_in0 = hq * PH * S - 1;
      for (int row = wave; row < G::NROWS; row += 4) {
        const int cc = row / (G::RD * G::RW), rz = (row / G::RW) % G::RD, ry = row % G::RW;
        const int zi = z_in0 + rz, yi = y_in0 + ry;
        const bool rowok = c0 + cc < d.Cin && zi >= 0 && zi < d.D && yi >= 0 && yi < d.W;
        const float* src = in + ((int64_t)b * d.Cin + c0 + cc) * V + ((int64_t)zi * d.W + yi) * d.H;
        float* dst = brick + cc * G::CS + rz * G::PS + ry * G::RSL;
        for (int x = lane; x < G::RSL; x += 64) {
          const int xi = x_in0 - G::XOFF + x;
          dst[x] = (rowok && xi >= 0 && xi < d.H) ? src[xi] : 0.0f;
        }
      }
    }
  };
  // weights of unit u's pass (MFMA A operand, rows = couts) and, on a brick's first pass, the bias.
  // The empty asm pins the wait for these conditional loads HERE: otherwise hipcc, unable to count
  // across the branch, drains vmcnt(0) at their first use inside the sweep — and with it the prefetch.
  f32x4 bvec[NT];  // bias stays in registers: no vector-memory op (hence no vmcnt wait) per unit
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bvec[nt] = bias_init(bias, nt, lane);
  auto setup = [&](int u) {
    const int pass = u % npass, c0 = pass * CC;
    if (!WINO && (npass > 1 || u == 0)) {   // (the Winograd sweep has its own transformed weights, loaded once above)
#pragma unroll
      for (int t = 0; t < G::T; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          w[t][nt] = (c0 + t / 7 < d.Cin) ? wp[(((c0 + t / 7) * 7 + t % 7) * NT + nt) * 64 + lane] : 0.0f;
#pragma unroll
      for (int t = 0; t < G::T; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(w[t][nt]));
    }
    if (!SINGLE && pass == 0) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int m = 0; m < PW * 4; ++m) acc[m][nt] = bvec[nt];
    }
  };
  auto read_step = [&](int t, float (&dst)[PW * 4]) {
    const float* base = brick + (t / 7) * G::CS + qoff[t % 7];
#pragma unroll
    for (int y = 0; y < PW; ++y)
#pragma unroll
      for (int x = 0; x < 4; ++x) dst[y * 4 + x] = base[y * S * G::RSL + x * 16 * S];
  };

  // Schedule per unit:  sweep(u) | barrier | stage(u+1), store(u), setup(u+1) | barrier | prefetch(u+2)
  // so the wait for a prefetch never has a just-issued store in front of it (vmcnt is in-order and
  // counts stores too) and every global load has a whole sweep to land.
  // timing aid: delay a subset of blocks to de-phase compute and memory phases across the chip
  {
    const int mode = (dbg >> 8) & 3, amount = (dbg >> 10) & 63;
    const bool late = mode == 1 ? (blockIdx.x & 1) : mode == 2 ? (blockIdx.x >= gridDim.x / 2) : mode == 3 ? ((blockIdx.x >> 3) & 1) : false;
    if (late)
      for (int i = 0; i < amount; ++i) __builtin_amdgcn_s_sleep(127);
  }
#ifdef LR_CONV0_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long stamp_t = __builtin_amdgcn_s_memtime();
#endif
  if (vec4) prefetch(unit_item(0));
  stage(0);
  setup(0);
  __syncthreads();
  if (vec4) prefetch(unit_item(min(1, my_units - 1)));
  for (int u = 0; u < my_units; ++u) {
    // ---- MFMA sweep, software-pipelined: the 16 LDS reads of step t+1 are issued one per MFMA of
    //      step t (immediate-offset ds_read: no address arithmetic, no vector memory).
    if constexpr (SINGLE) {
      if (!(dbg & 4)) {
        int b, dq, wq, hq, pass;
        item_coords(unit_item(u), b, dq, wq, hq, pass);
        const int dz = dq * PD + wave;
        const bool zok = dz < d.Do && !(dbg & 1);
        // output plane (b, dz) of this wave as a buffer resource; zero-length (every store dropped) when it is absent
        const int64_t plane_elems = (int64_t)d.Wo * d.Ho * d.Cout;
        const __amdgpu_buffer_rsrc_t oplane = __builtin_amdgcn_make_buffer_rsrc(
            out + (int64_t)b * d.out_bs + (int64_t)(zok ? dz : 0) * plane_elems, (short)0, zok ? (int)(plane_elems * 4) : 0, 0x00020000);
        // the sign-mask plane (b, dz) of this wave (MASK): zero-length when the plane does not exist
        const __amdgpu_buffer_rsrc_t mplane = __builtin_amdgcn_make_buffer_rsrc(
            MASK ? mask_out + ((int64_t)b * d.Do + (zok ? dz : 0)) * d.Wo * d.Ho * 4 : reinterpret_cast<unsigned char*>(out),
            (short)0, (MASK && zok) ? d.Wo * d.Ho * 4 : 0, 0x00020000);
        if constexpr (WINO) {
          // units = (row y, half g of the 64-wide row): 16 output pairs each; ONE unit at a time (its four accumulators are
          // four independent MFMA chains, 128 cycles apart: no dependency stall), the (unit, k-quad) sequence fully unrolled
          // with the four LDS values of the next step loaded while the current step's MFMAs run
          constexpr int NU = PW * 2, NSW = NU * 7;
          auto rd4 = [&](int sidx, float (&dst)[4]) {
            const int un = sidx / 7, q = sidx % 7;
            const float* dp = brick + (((un >> 1) & 1) ? woffo[q] : woff[q]) + (un >> 1) * G::RSL + (un & 1) * 32;
            dst[0] = dp[0]; dst[1] = dp[1]; dst[2] = dp[2]; dst[3] = dp[3];
          };
#ifndef LR_WINO_AHEAD
#define LR_WINO_AHEAD 1
#endif
          constexpr int WA = LR_WINO_AHEAD, WNB = WA + 1;
          float dv[WNB][4];
#pragma unroll
          for (int a0 = 0; a0 < WA; ++a0) rd4(a0, dv[a0 % WNB]);
          f32x4 dacc[4];
#pragma clang loop unroll(full)
          for (int sidx = 0; sidx < NSW; ++sidx) {
            const int un = sidx / 7, q = sidx % 7;
            if (q == 0) {
              dacc[0] = bvec[0];  // y0 = D0 + D1 + D2 carries the bias; y1 gets it in the epilogue
#pragma unroll
              for (int r = 1; r < 4; ++r) dacc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            if (sidx + WA < NSW) rd4(sidx + WA, dv[(sidx + WA) % WNB]);
            const float* dc = dv[sidx % WNB];
            const float v0 = dc[0] - dc[2], v1 = dc[1] + dc[2], v2 = dc[2] - dc[1], v3 = dc[1] - dc[3];
            dacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(uw[0][q], v0, dacc[0], 0, 0, 0);
            dacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(uw[1][q], v1, dacc[1], 0, 0, 0);
            dacc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(uw[2][q], v2, dacc[2], 0, 0, 0);
            dacc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(uw[3][q], v3, dacc[3], 0, 0, 0);
            if (sidx + WA < NSW) {   // per step: the two LDS reads of a later step, then the 4 differences + 4 MFMAs
              __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            if (q == 6) {
              f32x4 y0, y1;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                y0[e] = (dacc[0][e] + dacc[1][e]) + dacc[2][e];
                y1[e] = ((dacc[1][e] - dacc[2][e]) - dacc[3][e]) + bvec[0][e];
              }
              const int wo = wq * PW + (un >> 1), ho = hq * PH + (un & 1) * 32 + 2 * col;
              if constexpr (OUTL == LR_LAYOUT_NDHWC || OUTL == LR_LAYOUT_NDHWC_HPS) {
                store_tile_buf<OUTL>(y0, oplane, d, wo, ho, 0, lane, slope);
                store_tile_buf<OUTL>(y1, oplane, d, wo, ho + 1, 0, lane, slope);
                if constexpr (MASK) {
                  // A lane holds the quad kq of voxels ho and ho+1: two mask bytes.  The four quads of a voxel sit in the four
                  // 16-lane rows of the wave; gfx950's row/half swaps (v_permlane32_swap, v_permlane16_swap: 3 VALU ops, no LDS)
                  // bring them into row 0, whose lanes then store the two voxels' dwords as ONE 8-byte store — 128
                  // contiguous bytes per wave instead of two 64-lane byte stores (sub-dword writes: the mask cost the
                  // training forward of this block 0.5 ms, 3.0 -> 3.5 ms).
                  unsigned mm[2];
#pragma unroll
                  for (int o = 0; o < 2; ++o) {
                    const f32x4 a4 = o ? y1 : y0;
                    mm[o] = (a4[0] > 0.0f ? 1u : 0u) | (a4[1] > 0.0f ? 2u : 0u) | (a4[2] > 0.0f ? 4u : 0u) | (a4[3] > 0.0f ? 8u : 0u);
                  }
                  const unsigned x = mm[0] | (mm[1] << 8);
                  const auto s1 = __builtin_amdgcn_permlane32_swap(x, x, false, false);     // [1]: rows 0,1 <- rows 2,3 of x
                  const unsigned xa = s1[0], xb = s1[1];
                  const auto s2 = __builtin_amdgcn_permlane16_swap(xa, xa, false, false);   // [1]: row 0 <- row 1 of x
                  const auto s3 = __builtin_amdgcn_permlane16_swap(xb, xb, false, false);   // [1]: row 0 <- row 3 of x
                  const unsigned q0 = x, q1 = s2[1], q2 = xb, q3 = s3[1];                   // quads 0..3 of this lane's voxel pair (row 0)
                  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                  u32x2 dw;
                  dw.x = (q0 & 0xffu) | (q1 & 0xffu) << 8 | (q2 & 0xffu) << 16 | (q3 & 0xffu) << 24;
                  dw.y = ((q0 >> 8) & 0xffu) | ((q1 >> 8) & 0xffu) << 8 | ((q2 >> 8) & 0xffu) << 16 | ((q3 >> 8) & 0xffu) << 24;
                  // Ho is even here (H % 4 == 0) and ho is even: the pair is inside or outside together
                  const unsigned moff = (kq == 0 && wo < d.Wo && ho < d.Ho) ? (unsigned)((wo * d.Ho + ho) * 4) : 0x80000000u;
                  __builtin_amdgcn_raw_buffer_store_b64(dw, mplane, moff, 0, 0);
                }
              }
            }
          }
        } else {
        constexpr int NS = (PW * 4 / 2) * G::T;  // (tile pair, k-step) sequence, fully unrolled
        float ar[2][2];
        auto rd = [&](int sidx, float (&dst)[2]) {
          const int p = sidx / G::T, t = sidx % G::T;
          const float* base = brick + (t / 7) * G::CS + qoff[t % 7];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int m = 2 * p + h;
            dst[h] = base[(m / 4) * S * G::RSL + (m % 4) * 16 * S];
          }
        };
        rd(0, ar[0]);
        f32x4 pacc[2][NT];
#pragma clang loop unroll(full)
        for (int p = 0; p < PW * 4 / 2; ++p)
#pragma clang loop unroll(full)
        for (int t = 0; t < G::T; ++t) {
          const int sidx = p * G::T + t;
          if (t == 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) pacc[h][nt] = bvec[nt];
          }
          if (sidx + 1 < NS) rd(sidx + 1, ar[(sidx + 1) & 1]);
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              pacc[h][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][nt], ar[sidx & 1][h], pacc[h][nt], 0, 0, 0);
          if (sidx + 1 < NS) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);  // NT MFMAs …
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // … then one LDS read
            }
          }
          if constexpr (OUTL == LR_LAYOUT_NDHWC || OUTL == LR_LAYOUT_NDHWC_HPS) {
            if (t == G::T - 1) {  // unconditional, countable stores (see store_tile_buf)
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const int m = 2 * p + h;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                  store_tile_buf<OUTL>(pacc[h][nt], oplane, d, wq * PW + m / 4, hq * PH + (m % 4) * 16 + col, nt, lane, slope);
                if constexpr (MASK && NT == 1) {
                  const f32x4 a4 = pacc[h][0];
                  const unsigned mm = (a4[0] > 0.0f ? 1u : 0u) | (a4[1] > 0.0f ? 2u : 0u) | (a4[2] > 0.0f ? 4u : 0u) | (a4[3] > 0.0f ? 8u : 0u);
                  const int wo = wq * PW + m / 4, ho = hq * PH + (m % 4) * 16 + col;
                  const unsigned moff = (wo < d.Wo && ho < d.Ho) ? (unsigned)((wo * d.Ho + ho) * 4 + kq) : 0x80000000u;
                  __builtin_amdgcn_raw_buffer_store_b8((unsigned char)mm, mplane, moff, 0, 0);
                }
              }
            }
          } else if (t == G::T - 1 && zok) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int m = 2 * p + h;
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                store_tile<OUTL>(pacc[h][nt], out, d, b, dz, wq * PW + m / 4, hq * PH + (m % 4) * 16 + col, nt, lane,
                           out_layout, slope);
            }
          }
        }
        }  // direct sweep
      }
    } else {
    if (!(dbg & 4)) {
    read_step(0, a[0]);
#pragma unroll
    for (int t = 0; t < G::T; ++t) {
      if (t + 1 < G::T) read_step(t + 1, a[(t + 1) & 1]);
#pragma unroll
      for (int m = 0; m < PW * 4; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][nt], a[t & 1][m], acc[m][nt], 0, 0, 0);
      if (t + 1 < G::T) {
#pragma unroll
        for (int m = 0; m < PW * 4; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);  // NT MFMAs …
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // … then one LDS read
        }
      }
    }
    }
    }
    LR_STAMP(0);  // sweep
#ifdef LR_C0_PRIO
    __builtin_amdgcn_s_setprio(3);  // experiment: hurry through the non-matrix phases
#endif
    __syncthreads();  // every wave is done reading the brick
    LR_STAMP(1);  // barrier A
    if (!(dbg & 2)) stage(min(u + 1, my_units - 1));
    LR_STAMP(2);  // stage (waits for the prefetch)
    if (!SINGLE && u % npass == npass - 1 && !(dbg & 1)) {
      int b, dq, wq, hq, pass;
      item_coords(unit_item(u), b, dq, wq, hq, pass);
      const int dz = dq * PD + wave;
      if (dz < d.Do) {
        // (an epilogue on raw_buffer_store_b128 was measured: no faster, and with two co-resident
        //  blocks per CU it stored corrupted lanes at 256^3 — plain global stores are kept)
#pragma unroll
        for (int y = 0; y < PW; ++y)
#pragma unroll
          for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              store_tile<OUTL>(acc[y * 4 + x][nt], out, d, b, dz, wq * PW + y, hq * PH + x * 16 + col, nt, lane,
                         out_layout, slope);
      }
    }
    LR_STAMP(3);  // stores (+ address math)
    if (u + 1 < my_units) setup(u + 1);
    LR_STAMP(4);  // setup
    __syncthreads();  // next brick visible in LDS
    LR_STAMP(5);  // barrier B
    if (vec4 && !(dbg & 2)) prefetch(unit_item(min(u + 2, my_units - 1)));
#ifdef LR_C0_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    LR_STAMP(6);  // prefetch issue
  }
#ifdef LR_CONV0_STAMPS
  if (lane == 0 && wave == 1)
    for (int i = 0; i < 7; ++i) atomicAdd(&g_lr_stamps[i], stamp_acc[i]);
#endif
}

// ===========================================================================
// Channels-last (NDHWC) input, operands straight from L1/L2.
// Output brick per block: 4 planes (one per wavefront) x MT=4 rows x 16 voxels.
// K order: tap (27) x channel block cb (16 channels); inside a block lane group
// kq owns channels 4kq..4kq+3 and feeds them to four MFMAs.
// ===========================================================================
constexpr int MT = 4;
constexpr int TD = 4;

// PS: the input rows are parity-split along H (LR_LAYOUT_NDHWC_HPS: even voxels, then odd voxels), which
// makes every stride-2 tap of a 16-voxel tile ONE contiguous run — 8 full cache lines per load instead of
// 16 half-used ones (the texture addresser, not the matrix pipe, bounds the plain stride-2 layout).
template <int NT, int STRIDE, bool PS>
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
  const int zi0 = dz * STRIDE - 1, yw0 = wo0 * STRIDE - 1;
  const int xh0 = PS ? hq * 16 - 1 : hq * 16 * STRIDE - 1;  // PS: position inside a parity half-row
  const int64_t inb = (int64_t)b * d.D * d.W * d.H * d.Cin;
  const float* wbase = in + inb + ((int64_t)zi0 * d.W + yw0) * d.H * d.Cin + (int64_t)xh0 * (PS ? 16 : d.Cin);
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wbase), (short)0, 0x7fffffff, 0x00020000);
  constexpr unsigned OOR = 0x80000000u;
  // PS rows are [channel block of 16][parity][H/2][16 floats]: neighbouring voxels of a half-row are 64 B apart
  const unsigned lvoff = PS ? (unsigned)(col * 64 + kq * 16) : (unsigned)(col * STRIDE * d.Cin * 4 + kq * 16);
  const int half_h = (d.H + 1) >> 1;  // PS: the odd voxels of a row start here
  // Tap validity, split so that the k-loop spends (almost) no vector ALU on it — vector ALU instructions and MFMAs
  // share the SIMD's issue port, and the former version's ~30 per k-step cost as much as 15 % of the matrix time:
  //  * x (the lane's own column): three per-lane offsets, one per tx, with bit 31 set when that tap is outside;
  //  * y, z (wave-uniform): a 9-bit scalar mask per tile; an invalid (tz,ty) switches the load to a zero-length
  //    resource, for which every lane reads 0 — scalar selects only.
  unsigned vx[3];
  {
    const int xi0 = ho * STRIDE - 1;
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) vx[t3] = lvoff | ((ho < d.Ho && xi0 + t3 >= 0 && xi0 + t3 < d.H) ? 0u : OOR);
  }
  unsigned zymask[MT];  // bit tz*3+ty SET = valid (wave-uniform)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int yi0 = (wo0 + mt) * STRIDE - 1;
    unsigned m = 0u;
#pragma unroll
    for (int tz = 0; tz < 3; ++tz)
#pragma unroll
      for (int ty = 0; ty < 3; ++ty)
        if ((wo0 + mt < d.Wo) && zi0 + tz >= 0 && zi0 + tz < d.D && yi0 + ty >= 0 && yi0 + ty < d.W) m |= 1u << (tz * 3 + ty);
    zymask[mt] = m;
  }
  const __amdgpu_buffer_rsrc_t rsrc_null =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wbase), (short)0, 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(wp), (short)0, 0x7fffffff, 0x00020000);
  const unsigned wlane = (unsigned)lane * 16u;
  const bool ragged_c = (d.Cin & 15) != 0;  // only then can a channel quad of the last 16-block be missing
  const unsigned row_bytes = (unsigned)(STRIDE * d.H * d.Cin * 4);  // next tile (output row) of the wave

  // One k-step = (tap, 16-channel block): NT weight fragments + MT activation fragments, 4 MFMAs each.
  // Steps are software-pipelined one ahead: the loads of step s+1 are in flight while step s
  // runs on the matrix pipe.
  const int NS = 27 * CB;
  auto load_step = [&](int s, float4 (&a)[MT], float4 (&bw)[NT]) {
    // step order (tz, channel block, ty, tx) — the order in which the row-major kernel below feeds an accumulator, so
    // the two kernels (and with them the plain and the parity-split layout) give the same bits; wave-uniform
    const int tz = s / (9 * CB), rem = s - tz * 9 * CB, cb = rem / 9, t9 = rem - cb * 9;
    const int ty = t9 / 3, tx = t9 - ty * 3, tap = tz * 9 + t9;
    // PS (stride 2): tap tx reads voxel 2*ho+tx-1 — tx=1: even half, index ho; tx=0: odd half, index ho-1;
    // tx=2: odd half, index ho (xh0 already carries the -1)
    const int xs = PS ? (tx == 1 ? 1 : half_h + (tx >> 1)) : tx;
    const unsigned soff = PS ? (unsigned)(((tz * d.W + ty) * d.H * d.Cin + (cb * 2 * half_h + xs) * 16) * 4)
                             : (unsigned)(((tz * d.W + ty) * d.H + xs) * d.Cin * 4 + cb * 64);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      bw[nt] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, wlane, (unsigned)(((tap * CB + cb) * NT + nt) * 1024), 0));
    unsigned voff = tx == 0 ? vx[0] : (tx == 1 ? vx[1] : vx[2]);
    if (ragged_c) voff |= (cb * 16 + kq * 4 < d.Cin) ? 0u : OOR;
    const int zy = tz * 3 + ty;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const bool ok = (zymask[mt] >> zy) & 1u;  // scalar
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ok ? rsrc : rsrc_null, voff, soff + mt * row_bytes, 0);
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
  float4 a0[MT], a1[MT], a2[MT], b0[NT], b1[NT], b2[NT];
  // Loads run TWO steps ahead of the MFMAs (three register sets; NS = 27*CB is a multiple of 3): one step (32 MFMAs
  // ≈ 0.5 µs) is shorter than the loaded L2/fabric latency, and 5 waves per SIMD do not cover the rest — two ahead
  // measured +4 % (conv1 4.28 → 4.10 ms), three ahead −5 % (a fourth register set costs a wave of occupancy).
  // No conditional loads inside the loop (a branch around a load makes hipcc drain vmcnt(0) at the join): the tail
  // re-loads the last step instead.
  load_step(0, a0, b0);
  load_step(1, a1, b1);
  for (int s = 0; s < NS; s += 3) {
    load_step(min(s + 2, NS - 1), a2, b2);
    mfma_step(a0, b0);
    load_step(min(s + 3, NS - 1), a0, b0);
    mfma_step(a1, b1);
    load_step(min(s + 4, NS - 1), a1, b1);
    mfma_step(a2, b2);
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      store_tile(acc[mt][nt], out, d, b, dz, wo0 + mt, ho, nt, lane, out_layout, slope);
}

// ===========================================================================
// Row-major variant of the parity-split stride-2 kernel for Cin = 16 (the encoder's block 1: 58 of the 118 GFLOP).
// conv3d_cl_kernel walks TAPS: every (tz,ty,tx) loads its MT rows, so an input row is fetched for (mt,ty=2) and,
// 24 loads later, again for (mt+1,ty=0) — with a thousand waves per XCD that distance is far beyond the 4 MB L2 and
// the re-read goes to the fabric: PMC showed 2.6x the input in L2-miss traffic, 6 TB/s = the fabric's rate, next to an
// 80 %-busy matrix pipe.  This kernel walks INPUT ROWS instead: a row (even half, odd half at tx=0 and tx=2) is
// loaded ONCE and feeds its one or two (mt,ty) uses immediately; the nine weight fragments of the current tz stay in
// registers (they cost what the tap-major weight loads cost) and the next tz's arrive while the last rows are on the
// matrix pipe.  27 rows per tile, loads two rows ahead, everything unrolled (straight-line vmcnt accounting).
// Every accumulator still sees its taps in (tz,ty,tx) order: the results are bit-identical to conv3d_cl_kernel's.
// ===========================================================================
// MTV: output rows of a wave's tile.  4 for the big blocks; 1 for the last, tiny blocks of the encoder (16^3 and 8^3 outputs: a
// few dozen tiles — with 4 rows a wave walks 54 input rows in sequence while most of the chip idles).
template <int NT, int CB /* 16-channel blocks of the input: Cin = 16*CB */, int MTV = MT>
__global__ __launch_bounds__(256, 3) void conv3d_cl_rows_kernel(const float* __restrict__ in,
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
  // the dimensions the loop needs, as scalars of their own (the struct itself may end up in scratch)
  const int dD = __builtin_amdgcn_readfirstlane(d.D), dW = __builtin_amdgcn_readfirstlane(d.W),
            dH = __builtin_amdgcn_readfirstlane(d.H), dHo = __builtin_amdgcn_readfirstlane(d.Ho);
  const int wo0 = wq * MTV;
  const int col = lane & 15, kq = lane >> 4;
  const int ho = hq * 16 + col;  // this lane's voxel

  f32x4 acc[MTV][NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const f32x4 bv = bias_init(bias, nt, lane);
#pragma unroll
    for (int mt = 0; mt < MTV; ++mt) acc[mt][nt] = bv;
  }
  // window origin (-1,-1,-1) of this wave's tile, as in conv3d_cl_kernel<NT, 2, true>
  constexpr int CIN = 16 * CB;
  const int zi0 = dz * 2 - 1, yw0 = wo0 * 2 - 1, xh0 = hq * 16 - 1;
  const int64_t inb = (int64_t)b * dD * dW * dH * CIN;
  const float* wbase = in + inb + ((int64_t)zi0 * dW + yw0) * dH * CIN + (int64_t)xh0 * 16;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wbase), (short)0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_null =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wbase), (short)0, 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(wp), (short)0, 0x7fffffff, 0x00020000);
  constexpr unsigned OOR = 0x80000000u;
  const unsigned lvoff = (unsigned)(col * 64 + kq * 16);
  const int half_h = (dH + 1) >> 1;
  unsigned vx[3];  // per-lane offset of the lane's own column for tx = 0,1,2; bit 31 set when that tap is outside
  {
    const int xi0 = ho * 2 - 1;
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) vx[t3] = lvoff | ((ho < dHo && xi0 + t3 >= 0 && xi0 + t3 < dH) ? 0u : OOR);
  }
  const unsigned wlane = (unsigned)lane * 16u;
  constexpr int NR = 2 * MTV + 1;  // input rows of the tile per plane

  unsigned okmask = 0u;  // bit tz*NR+r SET = input row (zi0+tz, yw0+r) exists (the rest is the conv's zero padding)
#pragma unroll
  for (int q = 0; q < 3 * NR; ++q) {
    const int zi = zi0 + q / NR, yi = yw0 + q % NR;
    okmask |= ((unsigned)(zi >= 0) & (unsigned)(zi < dD) & (unsigned)(yi >= 0) & (unsigned)(yi < dW)) << q;
  }
  float4 w[3][3][NT];  // [ty][tx][nt] of the current (tz, channel block)
  auto load_w = [&](int pl, int ty) {  // pl = tz * CB + cb
    const int tz = pl / CB, cb = pl - tz * CB;
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        w[ty][tx][nt] = __builtin_bit_cast(
            float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, wlane, (unsigned)(((((tz * 3 + ty) * 3 + tx) * CB + cb) * NT + nt) * 1024), 0));
  };
  auto load_row = [&](int q, float4 (&a)[3]) {  // q = (tz * CB + cb) * NR + r
    const int pl = q / NR, r = q - pl * NR, tz = pl / CB, cb = pl - tz * CB;
    const bool ok = (okmask >> (tz * NR + r)) & 1u;  // wave-uniform (scalar select, no branch): outside -> the zero-length resource
    const unsigned row = (unsigned)((tz * dW + r) * dH * CIN + cb * 2 * half_h * 16);
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      // tx=1: even half, index ho; tx=0: odd half, index ho-1; tx=2: odd half, index ho (xh0 carries the -1)
      const int xs = tx == 1 ? 1 : half_h + (tx >> 1);
      a[tx] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ok ? rsrc : rsrc_null, vx[tx],
                                                                                (row + (unsigned)xs * 16u) * 4u, 0));
    }
  };
  auto use = [&](int mt, int ty, const float4 (&a)[3]) {
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ty][tx][nt].x, a[tx].x, acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ty][tx][nt].y, a[tx].y, acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ty][tx][nt].z, a[tx].z, acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ty][tx][nt].w, a[tx].w, acc[mt][nt], 0, 0, 0);
      }
  };
  float4 rows[3][3];  // three register sets: rows q, q+1, q+2
  load_w(0, 0);
  load_w(0, 1);
  load_w(0, 2);
  load_row(0, rows[0]);
  load_row(1, rows[1]);
#pragma unroll
  for (int q = 0; q < 3 * CB * NR; ++q) {
    const int pl = q / NR, r = q % NR;
    if (q + 2 < 3 * CB * NR) load_row(q + 2, rows[(q + 2) % 3]);
    __builtin_amdgcn_sched_barrier(0);  // keep the loads two rows ahead: the scheduler otherwise sinks them to their use
    // row r = 2*mt + ty: the (mt, ty) order below keeps every accumulator's taps in (tz,ty,tx) order
    if (r & 1) {
      use(r >> 1, 1, rows[q % 3]);
    } else {
      if (r >= 2) use((r >> 1) - 1, 2, rows[q % 3]);
      if (r < 2 * MTV) use(r >> 1, 0, rows[q % 3]);
    }
    // a ty's fragments are dead after its last row: fetch the next tz's while the remaining rows compute
    if (pl + 1 < 3 * CB) {
      if (r == 2 * MTV - 2) load_w(pl + 1, 0);
      if (r == 2 * MTV - 1) load_w(pl + 1, 1);
      if (r == 2 * MTV) load_w(pl + 1, 2);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int mt = 0; mt < MTV; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      store_tile(acc[mt][nt], out, d, b, dz, wo0 + mt, ho, nt, lane, out_layout, slope);
}

// ---- weight packing -----------------------------------------------------------
// Both layouts put W[cout = nt*16 + (lane&15)][k of lane group lane>>4] in lane order,
// i.e. the MFMA A-operand (rows = couts) of one k-step is one coalesced 256-B read.
__global__ void pack_cl_kernel(const float* __restrict__ w, float4* __restrict__ packed, int Cin,
                               int Cout, int CB, int NT) {
  // taps 27..35 = (tz, tx): w(tz, ty=0, tx) + w(tz, ty=2, tx), the M2 fragments of the Winograd-along-W rows kernel
  // (conv3d_rows.hip); every other kernel reads the first 27 taps only
  const int total = 36 * CB * NT * 64;
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
    const float* g = w + ((int64_t)co * Cin + ci) * 27;
    if (!(co < Cout && ci < Cin)) v[m] = 0.0f;
    else if (tap < 27) v[m] = g[tap];
    else v[m] = g[((tap - 27) / 3) * 9 + (tap - 27) % 3] + g[((tap - 27) / 3) * 9 + 6 + (tap - 27) % 3];
  }
  packed[idx] = make_float4(v[0], v[1], v[2], v[3]);
}

// Winograd F(2,3)-along-H weights of the first block (Cin <= 3, 16 couts), packed BEHIND the direct ones:
// [r 0..3][q 0..6][lane] with lane = (cout = lane&15, kq = lane>>4), k = 4q + kq = (c, tz, ty), U = G g:
// U0 = g0, U1 = (g0+g1+g2)/2, U2 = (g0-g1+g2)/2, U3 = g2 over the three taps along H.
__global__ void pack_planar_wino_kernel(const float* __restrict__ w, float* __restrict__ packed, int Cin, int Cout) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 4 * 7 * 64) return;
  const int lane = idx & 63, q = (idx >> 6) % 7, r = (idx >> 6) / 7;
  const int co = lane & 15, k = q * 4 + (lane >> 4);
  float u = 0.0f;
  const int c = k / 9;
  if (k < 27 && c < Cin && co < Cout) {
    const float* g = w + ((int64_t)co * Cin + c) * 27 + (k % 9) * 3;   // taps (tz, ty, tx = 0..2): k % 9 = tz*3 + ty
    const float g0 = g[0], g1 = g[1], g2 = g[2];
    u = r == 0 ? g0 : r == 1 ? ((g0 + g1) + g2) * 0.5f : r == 2 ? ((g0 - g1) + g2) * 0.5f : g2;
  }
  packed[idx] = u;
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
  if (in_layout == LR_LAYOUT_NDHWC || in_layout == LR_LAYOUT_NDHWC_HPS) return (int64_t)36 * ((Cin + 15) / 16) * NT * 64 * 4;
  if (in_layout == LR_LAYOUT_NCDHW)
    return (int64_t)Cin * 7 * NT * 64 + ((Cin <= 3 && NT == 1) ? 4 * 7 * 64 : 0) + (NT == 1 ? lr_internal_conv0_split_packed_floats(Cin, Cout) : 0);
  return LR_EINVAL;
}

extern "C" int lr_conv3d_pack_weights_f32(const float* weight, float* packed, int Cin, int Cout,
                                          int in_layout, void* stream) {
  if (!weight || !packed) return LR_ENULL;
  if (Cin < 1) return LR_EINVAL;
  if (Cout != 16 && Cout != 32) return LR_EUNSUPPORTED;
  const int NT = Cout / 16;
  if (in_layout == LR_LAYOUT_NDHWC || in_layout == LR_LAYOUT_NDHWC_HPS) {
    if (Cin % 4) return LR_EUNSUPPORTED;
    const int CB = (Cin + 15) / 16;
    const int total = 36 * CB * NT * 64;
    hipLaunchKernelGGL(pack_cl_kernel, dim3((total + 255) / 256), dim3(256), 0, lr_stream(stream),
                       weight, reinterpret_cast<float4*>(packed), Cin, Cout, CB, NT);
  } else if (in_layout == LR_LAYOUT_NCDHW) {
    const int total = Cin * 7 * NT * 64;
    hipLaunchKernelGGL(pack_planar_kernel, dim3((total + 255) / 256), dim3(256), 0,
                       lr_stream(stream), weight, packed, Cin, Cout, NT);
    if (Cin <= 3 && NT == 1)
      hipLaunchKernelGGL(pack_planar_wino_kernel, dim3(7), dim3(256), 0, lr_stream(stream), weight, packed + total, Cin, Cout);
    if (NT == 1 && lr_internal_conv0_split_packed_floats(Cin, Cout) > 0) {   // the split operands of conv0_split_f32.hip, behind both
      const int rc = lr_internal_conv0_split_pack(weight, packed + total + (Cin <= 3 ? 4 * 7 * 64 : 0), Cin, Cout, lr_stream(stream));
      if (rc != LR_OK) return rc;
    }
  } else {
    return LR_EINVAL;
  }
  return lr_launch_status();
}

struct FusedBp {  // host-side description of the views for the fused first block (conv0_pc.hip)
  const float* proj;
  const float* poses;  // host, P x 3
  int P, Pw, Ph;
};

static int conv_impl(const float* in, const float* in0, const float* packed_w, const float* bias,
                     float* out, int B, int Cin, int Cout, int D, int W, int H,
                     int stride, int in_layout, int out_layout,
                     float negative_slope, void* stream, const FusedBp* bpa = nullptr, unsigned char* mask_out = nullptr,
                     int z_phase = 0, long long out_bs = 0, long long in0_bs = 0) {
  if (bpa) in = in0;   // no channel-1.. tensor exists: the staging never dereferences `in` for them
  if (!in || !packed_w || !out) return LR_ENULL;
  if (B < 1 || Cin < 1 || D < 1 || W < 1 || H < 1) return LR_EINVAL;
  if (stride != 1 && stride != 2) return LR_EUNSUPPORTED;
  if (Cout != 16 && Cout != 32) return LR_EUNSUPPORTED;
  if (out_layout != LR_LAYOUT_NCDHW && out_layout != LR_LAYOUT_NDHWC && out_layout != LR_LAYOUT_NDHWC_HPS &&
      out_layout != LR_LAYOUT_BF16_NDHWC && out_layout != LR_LAYOUT_BF16_NDHWC_HPS)
    return LR_EINVAL;
  if (out_layout != LR_LAYOUT_NCDHW && (reinterpret_cast<uintptr_t>(out) & 15u)) return LR_EALIGN;
  if ((out_layout == LR_LAYOUT_NDHWC_HPS || out_layout == LR_LAYOUT_BF16_NDHWC_HPS) && (((H - 1) / stride + 1) & 1))
    return LR_EUNSUPPORTED;  // needs an even output H
  ConvDims d;
  d.B = B; d.Cin = Cin; d.Cout = Cout; d.D = D; d.W = W; d.H = H;
  d.Do = (D - 1) / stride + 1; d.Wo = (W - 1) / stride + 1; d.Ho = (H - 1) / stride + 1;
  const long long dense_bs = (long long)Cout * d.Do * d.Wo * d.Ho;
  if (out_bs != 0 && out_bs < dense_bs) return LR_EINVAL;
  if (out_bs != 0 && out_bs != dense_bs && (bpa || mask_out)) return LR_EUNSUPPORTED;   // the fused / mask forms write dense outputs
  d.out_bs = out_bs ? out_bs : dense_bs;
  if (in0_bs != 0 && in0_bs < (long long)D * W * H) return LR_EINVAL;
  d.in0_bs = in0_bs ? in0_bs : (long long)D * W * H;
  hipStream_t st = lr_stream(stream);
  const int NT = Cout / 16;
  const dim3 block(256);
  if (in0 && in_layout != LR_LAYOUT_NCDHW) return LR_EUNSUPPORTED;
  if (in_layout == LR_LAYOUT_NDHWC || in_layout == LR_LAYOUT_NDHWC_HPS) {
    const bool ps = in_layout == LR_LAYOUT_NDHWC_HPS;
    if (Cin % 4) return LR_EUNSUPPORTED;
    if (ps && (stride != 2 || (H & 1) || (Cin & 15))) return LR_EUNSUPPORTED;  // parity-split rows feed stride-2 blocks only
    if (reinterpret_cast<uintptr_t>(in) & 15u) return LR_EALIGN;
    if ((int64_t)12 * W * H * Cin + 4096 >= 0x7fffffffLL) return LR_EINVAL;  // 32-bit buffer offsets of a 3-plane window
    d.nHq = (d.Ho + 15) / 16; d.nWq = (d.Wo + MT - 1) / MT; d.nDq = (d.Do + TD - 1) / TD;
    const int64_t nblk = (int64_t)B * d.nDq * d.nWq * d.nHq;
    if (nblk > 0x7fffffffLL) return LR_EINVAL;
    const dim3 grid((unsigned)nblk);
    const float4* wt = reinterpret_cast<const float4*>(packed_w);
    const size_t occ_lds = (size_t)lr_sw_int(LR_SW_CONV_LDS, 0);  // tuning aid: caps resident blocks
    const bool rows_ok = ps && (Cin == 16 || Cin == 32) && !lr_sw_set(LR_SW_CONV_TAPMAJOR);  // tuning aid: the tap-major kernel
    // default for the parity-split stride-2 blocks: conv3d_rows.hip (persistent, Winograd F(2,2) along W, fragments in
    // LDS); LIFTREG_CONV_DIRECT=1 selects the direct walk below — the oracle's fmaf chain, bit for bit (A/B aid, tests)
    if (rows_ok && !lr_sw_set(LR_SW_CONV_DIRECT)) {
      const int rc = lr_internal_conv_rows_wlds(in, packed_w, bias, out, B, Cin, Cout, D, W, H, out_layout, negative_slope,
                                                z_phase, d.out_bs, st);
      if (rc != LR_EUNSUPPORTED) return rc;
    }
    if (rows_ok && Cin == 32 && NT == 2 && nblk < lr_sw_int(LR_SW_CONV_ROWS_MT1_BELOW, 512)) {  // env: tuning aid
      // the last tiny blocks (32 -> 32): one output row per wave, four times the waves, a quarter of the serial walk; same bits
      d.nWq = d.Wo;
      const dim3 g1((unsigned)((int64_t)B * d.nDq * d.nWq * d.nHq));
      hipLaunchKernelGGL((conv3d_cl_rows_kernel<2, 2, 1>), g1, block, 0, st, in, wt, bias, out, d, out_layout, negative_slope);
      return lr_launch_status();
    }
    if (rows_ok && Cin == 16 && NT == 1) hipLaunchKernelGGL((conv3d_cl_rows_kernel<1, 1>), grid, block, occ_lds, st, in, wt, bias, out, d, out_layout, negative_slope);
    else if (rows_ok && Cin == 16) hipLaunchKernelGGL((conv3d_cl_rows_kernel<2, 1>), grid, block, occ_lds, st, in, wt, bias, out, d, out_layout, negative_slope);
    else if (rows_ok && NT == 1) hipLaunchKernelGGL((conv3d_cl_rows_kernel<1, 2>), grid, block, occ_lds, st, in, wt, bias, out, d, out_layout, negative_slope);
    else if (rows_ok) hipLaunchKernelGGL((conv3d_cl_rows_kernel<2, 2>), grid, block, occ_lds, st, in, wt, bias, out, d, out_layout, negative_slope);
    else if (ps && NT == 1) hipLaunchKernelGGL((conv3d_cl_kernel<1, 2, true>), grid, block, occ_lds, st, in, wt, bias, out, d, out_layout, negative_slope);
    else if (ps) hipLaunchKernelGGL((conv3d_cl_kernel<2, 2, true>), grid, block, occ_lds, st, in, wt, bias, out, d, out_layout, negative_slope);
    else if (NT == 1 && stride == 1) hipLaunchKernelGGL((conv3d_cl_kernel<1, 1, false>), grid, block, 0, st, in, wt, bias, out, d, out_layout, negative_slope);
    else if (NT == 1) hipLaunchKernelGGL((conv3d_cl_kernel<1, 2, false>), grid, block, 0, st, in, wt, bias, out, d, out_layout, negative_slope);
    else if (stride == 1) hipLaunchKernelGGL((conv3d_cl_kernel<2, 1, false>), grid, block, 0, st, in, wt, bias, out, d, out_layout, negative_slope);
    else hipLaunchKernelGGL((conv3d_cl_kernel<2, 2, false>), grid, block, 0, st, in, wt, bias, out, d, out_layout, negative_slope);
  } else if (in_layout == LR_LAYOUT_NCDHW) {
    d.nHq = (d.Ho + PH - 1) / PH; d.nWq = (d.Wo + PW - 1) / PW; d.nDq = (d.Do + PD - 1) / PD;
    const int cc = stride == 1 ? 3 : 1;
    const int npass = (Cin + cc - 1) / cc;
    const int64_t nitems = (int64_t)B * d.nDq * d.nWq * d.nHq * npass;
    if (nitems > 0x7fffffffLL) return LR_EINVAL;
    if ((int64_t)d.Wo * d.Ho * Cout * 4 >= 0x7fffffffLL) return LR_EINVAL;  // an output plane is one buffer resource
    // 16-byte staging needs aligned rows and a window (cc channels) within 31-bit buffer offsets
    const int vec4 = (stride == 1) && (H % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15u) == 0) &&
                     ((int64_t)cc * D * W * H * 4 + (int64_t)16 * W * H * 4 < 0x7fffffffLL);
    const bool single = npass == 1;
    if (in0 && (!vec4 || !single || stride != 1 || Cin < 2 || (reinterpret_cast<uintptr_t>(in0) & 15u)))
      return LR_EUNSUPPORTED;  // split input: one 3-channel pass with 16-byte staging (the caller concatenates otherwise)
    // LIFTREG_CONV0_SPLIT=1: conv0_split_f32.hip — the same block on the bf16 MFMA with exact three-way bf16 splits of its
    // fp32 operands (a direct conv, half the rounding error of the Winograd sweep).  Measured equal to the fp32-MFMA
    // Winograd kernel below at C3 (2.9-3.1 vs 3.0 ms: DESIGN.md §6·6), so it is NOT the default.
#ifdef LR_EXPERIMENTAL   // (make exp: superseded by the fused pair kernel, kept as an A/B aid outside the product build)
    if (!bpa && !mask_out && stride == 1 && NT == 1 && Cin <= 4 && lr_sw_on(LR_SW_CONV0_SPLIT) && !lr_sw_set(LR_SW_CONV0_DIRECT) &&
        (out_layout == LR_LAYOUT_NDHWC || out_layout == LR_LAYOUT_NDHWC_HPS)) {
      const int64_t V = (int64_t)D * W * H;
      const float* ps = packed_w + (int64_t)Cin * 7 * 64 + (Cin <= 3 ? 4 * 7 * 64 : 0);
      const int rc = in0 ? lr_internal_conv0_split_f32(in0, d.in0_bs, in, (long long)(Cin - 1) * V, ps, bias, out, B, Cin, D, W, H, out_layout,
                                                       negative_slope, d.out_bs, st)
                         : lr_internal_conv0_split_f32(in, (long long)Cin * V, in + V, (long long)Cin * V, ps, bias, out, B, Cin, D, W, H,
                                                       out_layout, negative_slope, d.out_bs, st);
      if (rc != LR_EUNSUPPORTED) return rc;
    }
#endif
    int64_t resident = 256 * (single ? 3 : 2);  // persistent blocks per CU (registers: <=168 | <=256 per lane)
    if (single && !lr_sw_set(LR_SW_CONV0_DIRECT) && Cin <= 3 && stride == 1 && NT == 1 && (out_layout == LR_LAYOUT_NDHWC || out_layout == LR_LAYOUT_NDHWC_HPS))
      resident = 256 * LR_C0_WINO_BLOCKS;  // the Winograd instance (<=128 registers)
    resident = lr_sw_int(LR_SW_CONV0_BLOCKS, resident);  // tuning aid
    const dim3 grid((unsigned)(nitems < resident ? nitems : resident));
    const size_t lds1 = (size_t)3 * PlanarGeom<1, 3>::CS * sizeof(float) + 32;   // + the staging dump area (stage(); one float of row shift)
    const size_t lds2 = (size_t)1 * PlanarGeom<2, 1>::CS * sizeof(float) + 16;
    const int ni = (int)nitems;
#ifdef LR_DIAG_ABLATIONS   // diagnostic build only (make -B EXTRA=-DLR_DIAG_ABLATIONS): timing ablations, WRONG results
    const int dbg = getenv("LIFTREG_CONV0_DBG") ? atoi(getenv("LIFTREG_CONV0_DBG")) : 0;
#else
    const int dbg = 0;
#endif
#define LR_PL(NTV, SV, CCV, SGL, LDSV, V4)                                                                   \
  hipLaunchKernelGGL((conv3d_planar_kernel<NTV, SV, CCV, SGL>), grid, block, LDSV, st, in, packed_w, bias, out, d, \
                     out_layout, negative_slope, V4, ni, npass, dbg, in0)
    // conv0_pc.hip: the same block as a producer/consumer kernel with a double-buffered brick.  It carries the fused
    // backprojection (f1); for plain inputs it measured equal to the single-buffer kernel below (3.31-3.34 vs 3.29-3.30 ms
    // at C3, same bits), which therefore stays the default — LIFTREG_CONV0_PC=1 selects it (A/B aid).
#ifdef LR_EXPERIMENTAL   // (make exp: conv0_pc.hip is not part of the product build — measured slower since round 2)
    const bool pc_on = bpa || (lr_sw_on(LR_SW_CONV0_PC));
    if (pc_on && d.out_bs == dense_bs && d.in0_bs == (long long)D * W * H && stride == 1 && NT == 1 && single && vec4 &&
        (out_layout == LR_LAYOUT_NDHWC || out_layout == LR_LAYOUT_NDHWC_HPS)) {
      const int64_t V = (int64_t)D * W * H;
      int rc;
      if (bpa)
        rc = lr_internal_conv0_pc(in0, V, nullptr, 0, packed_w, bias, out, B, Cin, D, W, H, out_layout, negative_slope, bpa->proj,
                                  bpa->poses, bpa->P, bpa->Pw, bpa->Ph, st);
      else if (in0)
        rc = lr_internal_conv0_pc(in0, V, in, (int64_t)(Cin - 1) * V, packed_w, bias, out, B, Cin, D, W, H, out_layout,
                                  negative_slope, nullptr, nullptr, 0, 0, 0, st);
      else
        rc = lr_internal_conv0_pc(in, (int64_t)Cin * V, in + V, (int64_t)Cin * V, packed_w, bias, out, B, Cin, D, W, H, out_layout,
                                  negative_slope, nullptr, nullptr, 0, 0, 0, st);
      if (rc != LR_EUNSUPPORTED || bpa) return rc;
    }
#endif
    if (bpa) return LR_EUNSUPPORTED;
    // the Winograd F(2,3)-along-H sweep (default for the model's first block); LIFTREG_CONV0_DIRECT=1 selects the direct
    // sweep (the exact fmaf chain of the oracle; A/B aid)
    const bool wino = !lr_sw_set(LR_SW_CONV0_DIRECT) && Cin <= 3 && vec4;
    if (mask_out) {  // training forward of the first block: activation + LeakyReLU sign mask (one byte per channel quad)
      if (!(stride == 1 && NT == 1 && single && vec4)) return LR_EUNSUPPORTED;
      if (out_layout == LR_LAYOUT_NDHWC_HPS && wino)
        hipLaunchKernelGGL((conv3d_planar_kernel<1, 1, 3, true, LR_LAYOUT_NDHWC_HPS, true, true>), grid, block, lds1, st, in, packed_w, bias,
                           out, d, out_layout, negative_slope, vec4, ni, npass, 0, in0, mask_out);
      else if (out_layout == LR_LAYOUT_NDHWC && wino)
        hipLaunchKernelGGL((conv3d_planar_kernel<1, 1, 3, true, LR_LAYOUT_NDHWC, true, true>), grid, block, lds1, st, in, packed_w, bias,
                           out, d, out_layout, negative_slope, vec4, ni, npass, 0, in0, mask_out);
      else if (out_layout == LR_LAYOUT_NDHWC_HPS)
        hipLaunchKernelGGL((conv3d_planar_kernel<1, 1, 3, true, LR_LAYOUT_NDHWC_HPS, true>), grid, block, lds1, st, in, packed_w, bias,
                           out, d, out_layout, negative_slope, vec4, ni, npass, 0, in0, mask_out);
      else if (out_layout == LR_LAYOUT_NDHWC)
        hipLaunchKernelGGL((conv3d_planar_kernel<1, 1, 3, true, LR_LAYOUT_NDHWC, true>), grid, block, lds1, st, in, packed_w, bias,
                           out, d, out_layout, negative_slope, vec4, ni, npass, 0, in0, mask_out);
      else
        return LR_EUNSUPPORTED;
      return lr_launch_status();
    }
    if (stride == 1 && NT == 1 && single && out_layout == LR_LAYOUT_NDHWC_HPS && wino) {        // the model's first block
      hipLaunchKernelGGL((conv3d_planar_kernel<1, 1, 3, true, LR_LAYOUT_NDHWC_HPS, false, true>), grid, block, lds1, st, in, packed_w, bias,
                         out, d, out_layout, negative_slope, vec4, ni, npass, dbg, in0, nullptr);
    } else if (stride == 1 && NT == 1 && single && out_layout == LR_LAYOUT_NDHWC && wino) {
      hipLaunchKernelGGL((conv3d_planar_kernel<1, 1, 3, true, LR_LAYOUT_NDHWC, false, true>), grid, block, lds1, st, in, packed_w, bias,
                         out, d, out_layout, negative_slope, vec4, ni, npass, dbg, in0, nullptr);
    } else if (stride == 1 && NT == 1 && single && out_layout == LR_LAYOUT_NDHWC_HPS) {
      hipLaunchKernelGGL((conv3d_planar_kernel<1, 1, 3, true, LR_LAYOUT_NDHWC_HPS>), grid, block, lds1, st, in, packed_w, bias,
                         out, d, out_layout, negative_slope, vec4, ni, npass, dbg, in0);
    } else if (stride == 1 && NT == 1 && single && out_layout == LR_LAYOUT_NDHWC) {
      hipLaunchKernelGGL((conv3d_planar_kernel<1, 1, 3, true, LR_LAYOUT_NDHWC>), grid, block, lds1, st, in, packed_w, bias,
                         out, d, out_layout, negative_slope, vec4, ni, npass, dbg, in0);
    } else if (stride == 1) {
      if (NT == 1 && single) LR_PL(1, 1, 3, true, lds1, vec4);
      else if (NT == 1) LR_PL(1, 1, 3, false, lds1, vec4);
      else if (single) LR_PL(2, 1, 3, true, lds1, vec4);
      else LR_PL(2, 1, 3, false, lds1, vec4);
    } else {
      if (NT == 1 && single) LR_PL(1, 2, 1, true, lds2, 0);
      else if (NT == 1) LR_PL(1, 2, 1, false, lds2, 0);
      else if (single) LR_PL(2, 2, 1, true, lds2, 0);
      else LR_PL(2, 2, 1, false, lds2, 0);
    }
#undef LR_PL
  } else {
    return LR_EINVAL;
  }
  return lr_launch_status();
}

// Training forward of the encoder's first block: lr_conv3d_k3_lrelu_f32 (NCDHW in, stride 1, 16 couts, channels-last
// out, H % 4 == 0) that ALSO writes mask_out (B,D,W,H,4) uint8 (LR_LAYOUT_SIGN4): bit r of byte q = "output channel
// 4q+r > 0".  The next block's data gradient takes it as x_saved with x_layout = LR_LAYOUT_SIGN4 instead of re-reading
// the fp32 activation.
extern "C" int lr_conv3d_k3_lrelu_mask_f32(const float* in, const float* packed_w, const float* bias, float* out,
                                           uint8_t* mask_out, int B, int Cin, int Cout, int D, int W, int H, int stride,
                                           int in_layout, int out_layout, float negative_slope, void* stream) {
  if (!mask_out) return LR_ENULL;
  if (in_layout != LR_LAYOUT_NCDHW || Cout != 16 || stride != 1 || Cin > 3) return LR_EUNSUPPORTED;
  return conv_impl(in, nullptr, packed_w, bias, out, B, Cin, Cout, D, W, H, stride, in_layout, out_layout, negative_slope,
                   stream, nullptr, mask_out);
}

extern "C" int lr_conv3d_k3_lrelu_f32(const float* in, const float* packed_w, const float* bias,
                                      float* out, int B, int Cin, int Cout, int D, int W, int H,
                                      int stride, int in_layout, int out_layout,
                                      float negative_slope, void* stream) {
  return conv_impl(in, nullptr, packed_w, bias, out, B, Cin, Cout, D, W, H, stride, in_layout, out_layout, negative_slope,
                   stream);
}

// lr_conv3d_k3_lrelu_f32 for a z-SLAB of a larger volume: z_phase = (global output plane of local output plane 0) & 1.
// The Winograd rows kernel (conv3d_rows.hip) alternates the order in which neighbouring output planes walk their three
// input planes by the plane's parity; a slab whose local plane 0 is an odd global plane passes z_phase = 1 and gets, plane
// for plane, the bits of the unsharded launch.  Every other kernel ignores it.
extern "C" int lr_conv3d_k3_lrelu_zphase_f32(const float* in, const float* packed_w, const float* bias, float* out, int B,
                                             int Cin, int Cout, int D, int W, int H, int stride, int in_layout,
                                             int out_layout, float negative_slope, int z_phase, void* stream) {
  if (z_phase != 0 && z_phase != 1) return LR_EINVAL;
  return conv_impl(in, nullptr, packed_w, bias, out, B, Cin, Cout, D, W, H, stride, in_layout, out_layout, negative_slope,
                   stream, nullptr, nullptr, z_phase);
}

// lr_conv3d_k3_lrelu_zphase_f32 writing into a STRIDED batch: output element (b, ...) lives at out + b*out_batch_stride + the
// dense offset inside one batch element (out_batch_stride >= Cout*Do*Wo*Ho, in elements of the output type).  The sharded
// model's activations live in per-sample halo-padded buffers; with the stride one launch covers the whole batch.
extern "C" int lr_conv3d_k3_lrelu_obs_f32(const float* in, const float* packed_w, const float* bias, float* out, int B,
                                          int Cin, int Cout, int D, int W, int H, int stride, int in_layout,
                                          int out_layout, float negative_slope, int z_phase, int64_t out_batch_stride,
                                          void* stream) {
  if (z_phase != 0 && z_phase != 1) return LR_EINVAL;
  if (out_batch_stride < 0) return LR_EINVAL;
  return conv_impl(in, nullptr, packed_w, bias, out, B, Cin, Cout, D, W, H, stride, in_layout, out_layout, negative_slope,
                   stream, nullptr, nullptr, z_phase, (long long)out_batch_stride);
}

// The encoder's first block on cat([moving, views]) WITHOUT the concatenation: channel 0 from `in0` (B,1,D,W,H),
// channels 1..Cin-1 from `in_rest` (B,Cin-1,D,W,H); same kernel, same results as the concatenated input.
// Cin in {2,3}, H % 4 == 0, 16-byte aligned inputs; otherwise LR_EUNSUPPORTED (concatenate and call the entry above).
extern "C" int lr_conv3d_first_split_f32(const float* in0, const float* in_rest, const float* packed_w, const float* bias,
                                         float* out, int B, int Cin, int Cout, int D, int W, int H, int out_layout,
                                         float negative_slope, void* stream) {
  if (!in0) return LR_ENULL;
  return conv_impl(in_rest, in0, packed_w, bias, out, B, Cin, Cout, D, W, H, 1, LR_LAYOUT_NCDHW, out_layout,
                   negative_slope, stream);
}

// lr_conv3d_first_split_f32 for the sharded model: in0 is a z-slab VIEW of the whole (replicated) moving volume — batch
// element b starts at in0 + b*in0_batch_stride — and the output goes into a strided batch (see lr_conv3d_k3_lrelu_obs_f32).
// No copy of the moving image, one launch for the whole batch.  Strides in elements, 0 = dense.
extern "C" int lr_conv3d_first_split_obs_f32(const float* in0, int64_t in0_batch_stride, const float* in_rest, const float* packed_w,
                                             const float* bias, float* out, int B, int Cin, int Cout, int D, int W, int H,
                                             int out_layout, float negative_slope, int64_t out_batch_stride, void* stream) {
  if (!in0) return LR_ENULL;
  if (in0_batch_stride < 0 || out_batch_stride < 0) return LR_EINVAL;
  return conv_impl(in_rest, in0, packed_w, bias, out, B, Cin, Cout, D, W, H, 1, LR_LAYOUT_NCDHW, out_layout,
                   negative_slope, stream, nullptr, nullptr, 0, (long long)out_batch_stride, (long long)in0_batch_stride);
}

// f1: the encoder's first block with the backprojection computed inside its staging — the
// (B,P,D,W,H) feature volume of …Backproj.py:85-93 is never written: channel 0 = in0 (the moving image), channels
// 1..P = the backprojection of `proj` (B,P,Pw,Ph) for the emitter poses (host, P x 3 floats), sample for sample the
// arithmetic of lr_backproject_f32.  Same bits as lr_backproject_f32 + lr_conv3d_first_split_f32.  P in {1,2},
// H % 4 == 0, 16-byte aligned in0; otherwise LR_EUNSUPPORTED (the caller runs the two kernels).
#ifdef LR_EXPERIMENTAL
extern "C" int lr_conv3d_first_fused_bp_f32(const float* in0, const float* proj, const float* poses, const float* packed_w,
                                            const float* bias, float* out, int B, int P, int Pw, int Ph, int Cout, int D,
                                            int W, int H, int out_layout, float negative_slope, void* stream) {
  if (!in0 || !proj || !poses) return LR_ENULL;
  if (P < 1 || P > 2 || Pw < 2 || Ph < 2 || (int64_t)Pw * Ph * 4 >= 0x7fffffffLL || Pw >= (1 << 23) / Ph) return LR_EUNSUPPORTED;
  FusedBp a;
  a.proj = proj; a.poses = poses; a.P = P; a.Pw = Pw; a.Ph = Ph;
  return conv_impl(nullptr, in0, packed_w, bias, out, B, P + 1, Cout, D, W, H, 1, LR_LAYOUT_NCDHW, out_layout, negative_slope,
                   stream, &a);
}
#endif
