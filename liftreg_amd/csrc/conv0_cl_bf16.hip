// conv0_cl_bf16.hip — the encoder's first block with MANY input channels (4 < Cin <= 16; BASELINE config C4: 11 DRR views +
// the CT = 12) on the bf16 MFMA: fp32 NCDHW input, stride 1, 16 output channels, bf16 channels-last output.
//
// The 3-channel kernel (conv3d_bf16.hip, conv0_bf16_kernel) ran C4's 12 channels as four 3-channel passes over the
// accumulators: four global-load latencies, four barriers and 8 half-rate LDS reads per 4 MFMAs of every tile — 2.5 ms =
// 27 % of the HBM rate for 5.4 GB of compulsory traffic.  This kernel stages ALL channels of a brick once:
//
//   * persistent 8-wave blocks (one per CU) walk 4 x 4 x 64 output bricks in an XCD-contiguous order;
//   * the 6 x 6 x 66 input window lives in LDS CHANNELS-LAST as bf16 (16 channels = one 32-byte record per voxel, channels
//     Cin..15 zero), in TWO buffers: while the waves sweep brick i out of one buffer, the window of brick i+1 — requested
//     before the sweep with 16-byte bounds-checked buffer loads (out of the volume -> 0 = the conv's padding), 16 loads of one
//     x-quad x 4 channels per thread — sits in registers; it is rounded to bf16 and written to the other buffer after the
//     sweep: ONE barrier per brick, load latency under a whole sweep;
//   * K order of an MFMA (v_mfma_f32_16x16x32_bf16): 2 taps x 16 channels; lane group kq = (tap half, channel half), so a
//     lane's B operand is ONE aligned ds_read_b128 (8 channels of one voxel), tap and tile offsets are immediates.  27 taps =
//     9 z-pairs (0,ty,tx)|(1,ty,tx) + 3 x-pairs (2,ty,0)|(2,ty,1) + (2,0,2)|(2,1,2) + (2,2,2)|none = 14 MFMAs per 16-voxel
//     tile, all 14 weight fragments in registers for the life of the block;
//   * LDS banks: reads of 16 consecutive voxels x 16 B are conflict-free in the natural record order; the staging writes
//     (lanes = consecutive x-quads: a 128-byte stride = ONE bank) are spread by swapping the two voxel pairs of every odd
//     quad (record p sits at p ^ ((p >> 1) & 2)) and by running the channel quad fastest over the lanes: 2.7-way instead
//     of 16-way, reads untouched (simulated per lane group).
//
// Numerics: the contract of conv0_bf16_kernel (inputs and weights rounded to nearest-even bf16, exact products, fp32
// accumulation, fp32 bias + LeakyReLU, bf16 store) with another fp32 summation order — oracle/ref_ops.py:conv_block_bf16.
// Replaces (reference file:line): src/liftreg/layers/layers.py:365-369 as wired at models/LiftRegDeformSubspaceBackproj.py:95-98.
#include "lr_common.h"
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int BX = 64;                  // output columns of a step
constexpr int NQ = 18;                  // aligned float4 quads of a window row: x0-4 .. x0+67
constexpr int SLOT = 32;                // bytes per voxel record (16 bf16 channels)
constexpr int RB = 71 * SLOT;           // bytes per window row: records p = 0..70 (p = x - x0 + 4; p = 3..68 are read)
constexpr int NSTEP = 14;
constexpr unsigned OOR = 0x80000000u;
#ifndef LR_C0CL_AHEAD
#define LR_C0CL_AHEAD 6
#endif
#ifndef LR_C0CL_DEPTH
#define LR_C0CL_DEPTH 1   // groups of input planes in flight (register sets); 2 measured no faster
#endif
#ifndef LR_C0CL_SGB
#define LR_C0CL_SGB 0   // pinning the read/MFMA order with sched_group_barrier collapsed the look-ahead (read, wait, use): off
#endif

struct C0Dims {
  int B, Cin, D, W, H;
  int nHq, nWq, nch, ZC;   // column grid (x, y), z chunks per column and planes per chunk (a multiple of SZ)
  int nunits;             // B * nch * nWq * nHq
  long long out_bs;       // output elements between batch elements (dense: 16*D*W*H)
  float slope;
  int abl;   // timing-only ablation bits (LIFTREG_C0CL_ABL, wrong results): 1 no global loads, 2 no stores, 4 no sweep, 8 no LDS writes
};

__device__ __forceinline__ u16 to_bf16(float v) {
  const __bf16 h = (__bf16)v;
  return __builtin_bit_cast(u16, h);
}
__device__ __forceinline__ unsigned pack2(float a, float b) { return (unsigned)to_bf16(a) | ((unsigned)to_bf16(b) << 16); }
__device__ __forceinline__ float lrelu(float v, float slope) { return v >= 0.0f ? v : v * slope; }
__device__ __forceinline__ int phys_rec(int p) { return p ^ ((p >> 1) & 2); }  // swap the voxel pairs of odd quads

// One LDS fragment of a tile column and the MFMAs it feeds: class = which per-lane base (A tx0|tx1|tx2, B, C, D), off = byte
// offset of its input row, s0 / s1 = the weight step it meets in output row 0 / 1 (-1: none)
struct FragUse { int cls, off, s0, s1; };
__host__ __device__ constexpr FragUse frag_use(int k) {
  if (k < 12) { const int iy = k / 3, tx = k % 3; return {tx, iy * RB, iy <= 2 ? iy * 3 + tx : -1, iy >= 1 ? (iy - 1) * 3 + tx : -1}; }
  if (k < 16) { const int iy = k - 12; return {3, iy * RB, iy <= 2 ? 9 + iy : -1, iy >= 1 ? 8 + iy : -1}; }
  if (k < 18) { const int r = k - 16; return {4, r * RB, r == 0 ? 12 : -1, r == 1 ? 12 : -1}; }
  const int r = k - 18;
  return {5, (2 + r) * RB, r == 0 ? 13 : -1, r == 1 ? 13 : -1};
}

// brick list of block `bid`: the 8 XCDs (block id % 8) take contiguous eighths of the brick order (x fastest, then y, z,
// batch), the blocks of an XCD stride through their eighth together — neighbouring bricks (shared halo rows) meet in one L2
__device__ __forceinline__ void brick_range(int bid, int nblk, int nbricks, int& first, int& stride, int& end) {
  if ((nblk & 7) == 0 && nbricks >= nblk) {
    const int xcd = bid & 7, li = bid >> 3, per = nblk >> 3;
    const int q = nbricks >> 3, r = nbricks & 7;
    const int lo = xcd * q + (xcd < r ? xcd : r);
    end = lo + q + (xcd < r ? 1 : 0);
    first = lo + li;
    stride = per;
  } else {
    first = bid; stride = nblk; end = nbricks;
  }
}

// SZ planes x BY rows x 64 columns of outputs per step; 8 waves = SZ planes x BY/2 row pairs.  (SZ, BY) = (4, 4) | (2, 8) | (1, 16).
// A block marches down z through a chunk of ONE (y, x) column: the LDS holds a ring of 2*SZ + 2 input planes
// ((BY + 2) rows x 71 records each).  "Group" f = the SZ planes a step adds; flat over the block's units (a unit = one
// column chunk, 1 + ZC/SZ groups: the first holds the two planes above the chunk), plane FP = SZ*f + j sits in ring slot
// FP mod NRING.  Iteration f: request group f+1 (registers) | sweep the step whose last planes are group f | write group
// f+1 | ONE barrier.  The slots group f+1 overwrites were last read in iteration f-1's sweep, which every wave left before
// the previous barrier; the step of iteration f reads other slots.  Every input plane is fetched once per column (+ 2 per
// chunk) instead of 1.5 times.
// CLIN: `in` is already the bf16 channels-last encoder input (B, D, W, H, 16) that lr_backproject_encin_bf16 writes — the
// staging is then a plain copy of 16-byte record halves (no rounding, no transposition, no swizzle: consecutive lanes write
// consecutive 16 bytes, the natural record order is conflict-free on both sides) and the window starts exactly at x0 - 1.
template <int CQ, bool HPSOUT, int SZ, int BY, bool CLIN = false>
__global__ __launch_bounds__(512, 2) void conv0_cl_bf16_kernel(const float* __restrict__ in, const u32x4* __restrict__ wp,
                                                               const float* __restrict__ bias, u16* __restrict__ out, C0Dims d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  static_assert(SZ * (BY / 2) == 8, "8 waves");
  constexpr int WR = BY + 2;                        // window rows per plane
  constexpr int PLB = WR * RB;                      // bytes per ring plane
  constexpr int NRING = 2 * SZ + 2;
  constexpr int NPRE = (2 + SZ - 1) / SZ;           // groups of a unit before its first step (the two planes above it)
  constexpr int LDSB = NRING * PLB + SLOT;          // (+ 64 bytes behind it: where threads without a staging item write, see write_group)
  constexpr int DUMP = LDSB;
  constexpr int NCHK = 132;                         // CLIN: 16-byte chunks of a window row (66 records)
  constexpr int NITEMS = CLIN ? SZ * WR * NCHK : CQ * SZ * WR * NQ;   // planar: one item = 4 channels x one x-quad of a row
  constexpr int NIT = (NITEMS + 511) / 512;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kq = lane >> 4, up = kq >> 1;
  const int dD = d.D, dW = d.W, dH = d.H;
  const unsigned V4 = (unsigned)dD * dW * dH * 4u;  // bytes of one channel volume (< 2^31 / Cin: checked by the launcher)

  // ---- zero the ring once: channels Cin..15 of every record stay zero for the life of the block
  for (int o = tid * 16; o < LDSB; o += 512 * 16) *reinterpret_cast<u32x4*>(lds + o) = (u32x4){0u, 0u, 0u, 0u};

  // ---- staging items of this thread (step-invariant part): three registers per item
  unsigned g_rel[NIT];    // byte offset relative to (plane 0 of the group, row y0-1, column x0-4) of channel 4*cq
  unsigned l_rec[NIT];    // LDS byte offset inside a ring plane of (row, record 4q, channel quad cq)
  unsigned it_pk[NIT];    // j (3 bits) | row << 3 (5 bits) | q << 8 (5 bits) | cq << 13 (2 bits) | live << 15
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int item = it * 512 + tid;
    const bool live = item < NITEMS;
    if constexpr (CLIN) {   // item = (plane j, row, 16-byte chunk c of the row's 66 records); q holds the chunk index
      const int c = item % NCHK, row = (item / NCHK) % WR, j = item / NCHK / WR;
      g_rel[it] = (unsigned)(((j * dW + row) * dH) * SLOT + c * 16);
      l_rec[it] = (unsigned)(row * RB + (3 + (c >> 1)) * SLOT + (c & 1) * 16);
      it_pk[it] = (unsigned)(live ? j : 0) | ((unsigned)row << 3) | ((unsigned)(c >> 1) << 8) | (live ? 1u << 15 : 0u);
      continue;
    }
    const int cq = item % CQ, rest = item / CQ;
    const int q = rest % NQ, row = (rest / NQ) % WR, j = rest / NQ / WR;
    g_rel[it] = (unsigned)cq * 4u * V4 + (unsigned)((j * dW + row) * dH + 4 * q) * 4u;
    l_rec[it] = (unsigned)(row * RB + 4 * q * SLOT + cq * 8);
    static_assert(SZ <= 8 && WR <= 32 && NQ <= 32 && CQ <= 4, "it_pk field widths");   // (CLIN: the voxel index c>>1 <= 65 takes bits 8..14)
    it_pk[it] = (unsigned)(live ? j : 0) | ((unsigned)row << 3) | ((unsigned)q << 8) | ((unsigned)cq << 13) | (live ? 1u << 15 : 0u);
  }

  // ---- B-operand offsets of this lane inside a ring plane: see the header for the four tap-pair classes
  const int zw = wave % SZ, rh = wave / SZ;
  const int hb = (kq & 1) * 16;
  auto prec = [](int p) __attribute__((always_inline)) -> int { return CLIN ? p : phys_rec(p); };
  unsigned offA[3], offB, offC, offD;
#pragma unroll
  for (int tx = 0; tx < 3; ++tx) offA[tx] = (unsigned)(2 * rh * RB + prec(col + tx + 3) * SLOT + hb);
  offB = (unsigned)(2 * rh * RB + prec(col + up + 3) * SLOT + hb);
  offC = (unsigned)((2 * rh + up) * RB + prec(col + 5) * SLOT + hb);
  offD = (unsigned)(2 * rh * RB + prec(col + 5) * SLOT + hb);

  u32x4 w[NSTEP];
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) w[s] = wp[s * 64 + lane];
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bias[kq * 4 + r];
  }

  int first, stride, end;
  brick_range((int)blockIdx.x, (int)gridDim.x, d.nunits, first, stride, end);

  f32x4 ld[LR_C0CL_DEPTH][NIT][CLIN ? 1 : 4];   // DEPTH groups in flight: register set = flat group index % DEPTH
  struct Unit { int b, zc0, zc1, y0, x0; };
  auto decode = [&](int u) __attribute__((always_inline)) -> Unit {
    const int hq = u % d.nHq, wq = (u / d.nHq) % d.nWq, ch = (u / d.nHq / d.nWq) % d.nch;
    Unit t;
    t.b = u / d.nHq / d.nWq / d.nch;
    t.zc0 = ch * d.ZC;
    t.zc1 = min(dD, t.zc0 + d.ZC);
    t.y0 = wq * BY; t.x0 = hq * BX;
    return t;
  };
  // group g of a unit = input planes z = zc0 - NPRE*SZ + 1 + SZ*g + j, j = 0..SZ-1 (planes above zc0 - 1 are not needed):
  // the window of step i ends with group i + NPRE and starts two planes before that group
  auto uniform = [](const Unit& v) __attribute__((always_inline)) -> Unit {   // block-uniform by construction: keep it in scalar registers
    Unit r;
    r.b = __builtin_amdgcn_readfirstlane(v.b); r.zc0 = __builtin_amdgcn_readfirstlane(v.zc0);
    r.zc1 = __builtin_amdgcn_readfirstlane(v.zc1); r.y0 = __builtin_amdgcn_readfirstlane(v.y0);
    r.x0 = __builtin_amdgcn_readfirstlane(v.x0);
    return r;
  };
  auto issue_loads = [&](const Unit& uv, int gv, bool validv, auto setc) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value;
    const Unit u = uniform(uv);
    const int g = __builtin_amdgcn_readfirstlane(gv);
    const bool valid = __builtin_amdgcn_readfirstlane((int)validv) != 0;
    // the resource ends with the batch element (Cin volumes): any offset outside it reads 0, never faults.  Base and size
    // go through readfirstlane: left to its own analysis hipcc kept the descriptor in vector registers and wrapped every
    // load in a waterfall loop
    const uint64_t xa = CLIN ? reinterpret_cast<uint64_t>(reinterpret_cast<const u16*>(in) + (int64_t)u.b * 16 * ((int64_t)dD * dW * dH))
                             : reinterpret_cast<uint64_t>(in + (int64_t)u.b * d.Cin * ((int64_t)dD * dW * dH));
    const uint64_t xs = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(xa >> 32)) << 32) |
                        (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xa);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(xs), (short)0,
                                                                          __builtin_amdgcn_readfirstlane((int)(CLIN ? 8u * V4 : (unsigned)d.Cin * V4)), 0x00020000);
    const int zg = u.zc0 - NPRE * SZ + 1 + SZ * g;
    const int org = CLIN ? ((zg * dW + (u.y0 - 1)) * dH + (u.x0 - 1)) * SLOT
                         : ((zg * dW + (u.y0 - 1)) * dH + (u.x0 - 4)) * 4;   // may be negative: only used where the element exists
    const int jmin = g == 0 ? NPRE * SZ - 2 : 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const unsigned pk = it_pk[it];
      const int j = (int)(pk & 7u);
      if constexpr (CLIN) {
        const int zi = zg + j, yi = u.y0 - 1 + (int)((pk >> 3) & 31u), xi = u.x0 - 1 + (int)((pk >> 8) & 127u);
        const int ok = (int)valid & (int)((pk >> 15) & 1u) & (int)(j >= jmin) & (int)(zi >= 0) & (int)(zi < dD) & (int)(yi >= 0) & (int)(yi < dW) &
                       (int)(xi >= 0) & (int)(xi < dH);
        unsigned voff = ((unsigned)org + g_rel[it]) | (((unsigned)ok - 1u) & OOR);
#ifdef LR_C0CL_ABLATIONS
        if (d.abl & 1) voff = OOR;
#endif
        ld[SET][it][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
        continue;
      }
      const int zi = zg + j, yi = u.y0 - 1 + (int)((pk >> 3) & 31u), xi = u.x0 - 4 + 4 * (int)((pk >> 8) & 31u);
      // bitwise, no short circuits: hipcc turns `a && b ? x : y` around a load into exec-mask branches with a full vmcnt drain
      const int ok = (int)valid & (int)((pk >> 15) & 1u) & (int)(j >= jmin) & (int)(zi >= 0) & (int)(zi < dD) & (int)(yi >= 0) & (int)(yi < dW) &
                     (int)(xi >= 0) & (int)(xi < dH);
      const unsigned off = (unsigned)org + g_rel[it];
      const int c0 = 4 * (int)((pk >> 13) & 3u);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned live = (unsigned)(ok & (int)(c0 + k < d.Cin));            // 1 | 0
        unsigned voff = (off + (unsigned)k * V4) | ((live - 1u) & OOR);           // dead element: bit 31 -> outside the resource -> 0
#ifdef LR_C0CL_ABLATIONS
        if (d.abl & 1) voff = OOR;
#endif
        ld[SET][it][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
      }
    }
  };
  auto write_group = [&](int wposv, auto setc) __attribute__((always_inline)) {   // wpos = ring slot of the group's plane 0
    constexpr int SET = decltype(setc)::value;
    const int wpos = __builtin_amdgcn_readfirstlane(wposv);
#ifdef LR_C0CL_ABLATIONS
    if (d.abl & 8) return;
#endif
    // No branch around the writes: a path that skips the waits for the loads makes every later wait in the loop conservative
    // (vmcnt(0) with the sweep's stores in flight = every step waits for its own stores).  A thread without an item in its
    // last slot loaded zeros and writes them into the dump area behind the ring.
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const bool has_item = (it + 1) * 512 <= NITEMS || it * 512 + tid < NITEMS;
      int sl = wpos + (int)(it_pk[it] & 7u);
      sl -= sl >= NRING ? NRING : 0;
      if constexpr (CLIN) {
        *reinterpret_cast<f32x4*>(lds + (has_item ? (unsigned)sl * PLB + l_rec[it] : (unsigned)DUMP)) = ld[SET][it][0];
        continue;
      }
      const unsigned xo = (it_pk[it] >> 2) & 64u;   // q odd (bit 8 of the pack): the quad's voxel pairs are swapped (64 bytes = two records)
      const unsigned base = (unsigned)sl * PLB + l_rec[it];
      const unsigned la = base + xo, lb2 = base + (64u - xo);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x2 rec = {pack2(ld[SET][it][0][j], ld[SET][it][1][j]), pack2(ld[SET][it][2][j], ld[SET][it][3][j])};
        *reinterpret_cast<u32x2*>(lds + (has_item ? (j < 2 ? la : lb2) + (unsigned)(j & 1) * SLOT : (unsigned)DUMP)) = rec;
      }
    }
  };
  // the step whose output planes are z0 .. z0+SZ-1; rpos = ring slot of input plane z0 - 1
  // (livev false: the window is not complete yet — the first NPRE groups of a chunk; the sweep still runs on whatever the ring
  // holds and its stores are dropped: a branch around it leaves two different store counts behind the next loads)
  auto sweep = [&](const Unit& uv, int z0v, int rposv, bool livev) __attribute__((always_inline)) {
    int live = __builtin_amdgcn_readfirstlane((int)livev);
#ifdef LR_C0CL_ABLATIONS
    if (d.abl & 4) return;
    if (d.abl & 2) live = 0;
#endif
    const Unit u = uniform(uv);
    const int z0 = __builtin_amdgcn_readfirstlane(z0v), rpos = __builtin_amdgcn_readfirstlane(rposv);
    const int dz = z0 + zw;
    int s0 = rpos + zw + up, s2 = rpos + zw + 2;   // ring slots of this lane's tap planes: tz = 0 | 1 by lane half, tz = 2
    s0 -= s0 >= NRING ? NRING : 0;
    s2 -= s2 >= NRING ? NRING : 0;
    // this lane's voxel of tile (r, t): row y0 + 2 rh + r, x = x0 + 16 t + col; channels kq*4 .. kq*4+3
    const int hp_lane = HPSOUT ? (col & 1) * (dH >> 1) + (u.x0 >> 1) + (col >> 1) : u.x0 + col;
    // one buffer resource per output plane of this wave (bounds-checked stores, no branch): W*H*32 bytes
    const bool zok = live && dz < u.zc1;
    const uint64_t oa = reinterpret_cast<uint64_t>(out + (int64_t)u.b * d.out_bs + (int64_t)(zok ? dz : 0) * dW * dH * 16);
    const uint64_t os = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(oa >> 32)) << 32) |
                        (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)oa);
    const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<u16*>(os), (short)0,
                                                                          __builtin_amdgcn_readfirstlane(zok ? dW * dH * 32 : 0), 0x00020000);
    const unsigned off_lane = (unsigned)((((u.y0 + 2 * rh) * dH + hp_lane) * 16 + kq * 4) * 2);
    const unsigned char* const pl0 = lds + (unsigned)s0 * PLB;
    const unsigned char* const pl2 = lds + (unsigned)s2 * PLB;
    const unsigned char* const cls[6] = {pl0 + offA[0], pl0 + offA[1], pl0 + offA[2], pl2 + offB, pl2 + offC, pl2 + offD};
    // A wave owns two output rows (r = 0, 1) of four 16-voxel tiles.  Per tile column t the rows share their input rows:
    // fragment (class, input row iy) feeds output row 0 with tap row ty = iy and output row 1 with ty = iy - 1, so a column
    // needs 20 fragment reads for its 28 MFMAs (FragUse), two accumulators alternating on the matrix pipe.  The LDS
    // reads run LA fragments ahead of their MFMAs in a register ring.
    constexpr int NF = 4 * 20, LA = LR_C0CL_AHEAD;
    u32x4 fr[LA + 1];
    auto rd = [&](int f) __attribute__((always_inline)) -> u32x4 {
      const FragUse fu = frag_use(f % 20);
      return *reinterpret_cast<const u32x4*>(cls[fu.cls] + fu.off + (f / 20) * 16 * SLOT);
    };
#pragma unroll
    for (int f = 0; f < LA; ++f) fr[f] = rd(f);
    f32x4 acc0 = bv, acc1 = bv;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      if (f + LA < NF) fr[(f + LA) % (LA + 1)] = rd(f + LA);
      const FragUse fu = frag_use(f % 20);
      const bf16x8 bfr = __builtin_bit_cast(bf16x8, fr[f % (LA + 1)]);
      if (fu.s0 >= 0) acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[fu.s0 < 0 ? 0 : fu.s0]), bfr, acc0, 0, 0, 0);
      if (fu.s1 >= 0) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[fu.s1 < 0 ? 0 : fu.s1]), bfr, acc1, 0, 0, 0);
#if LR_C0CL_SGB
      // pin the software pipeline: one LDS read (the fragment LA steps ahead), then this fragment's MFMAs
      if (f + LA < NF) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (fu.s0 >= 0 && fu.s1 >= 0) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      else __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#endif
      if (f % 20 == 19) {
        const int t = f / 20;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const f32x4 a = r ? acc1 : acc0;
          const int ok = (int)(u.y0 + 2 * rh + r < dW) & (int)(u.x0 + t * 16 + col < dH);
          const unsigned off = (off_lane + (unsigned)((r * dH * 16 + t * (HPSOUT ? 8 : 16) * 16) * 2)) | (((unsigned)ok - 1u) & OOR);
          const u32x2 v = {pack2(lrelu(a[0], d.slope), lrelu(a[1], d.slope)), pack2(lrelu(a[2], d.slope), lrelu(a[3], d.slope))};
          __builtin_amdgcn_raw_buffer_store_b64(v, ores, off, 0, 2 /* nt */);
        }
        acc0 = bv; acc1 = bv;
      }
    }
  };

  __syncthreads();   // the zero fill is complete
  if (first >= end) return;
  // flat walk over (unit, group).  Cur = the group written last (its step is swept now); the next DEPTH groups are in flight
  // in the register sets (flat index % DEPTH), the one requested in an iteration arrives DEPTH iterations later: with every
  // CU in the same phase, one step of look-ahead left the memory system idle while all of them swept (loads + sweep added up).
  struct Pos { Unit u; int g, ng, uid; bool ok; };
  auto advance = [&](const Pos& p) __attribute__((always_inline)) -> Pos {
    Pos n = p;
    n.g = p.g + 1;
    if (n.g >= p.ng) {
      n.uid = p.uid + stride;
      n.g = 0;
      n.ok = p.ok && n.uid < end;
      if (n.ok) { n.u = decode(n.uid); n.ng = (n.u.zc1 - n.u.zc0 + SZ - 1) / SZ + NPRE; }
    }
    return n;
  };
  constexpr int DEPTH = LR_C0CL_DEPTH;
  Pos cur;
  cur.u = decode(first); cur.g = 0; cur.uid = first; cur.ok = true;
  cur.ng = (cur.u.zc1 - cur.u.zc0 + SZ - 1) / SZ + NPRE;
  issue_loads(cur.u, 0, true, std::integral_constant<int, 0>{});
  Pos ahead = cur;                     // the newest group requested
  if constexpr (DEPTH == 2) {
    ahead = advance(ahead);
    issue_loads(ahead.u, ahead.g, ahead.ok, std::integral_constant<int, 1>{});
  }
  int wpos = 0;                        // ring slot of plane 0 of the group written last
  write_group(wpos, std::integral_constant<int, 0>{});
  __syncthreads();
  // one iteration; SETW = register set of the group written at its end (flat index f+1), which is also the set that is
  // free for the request of group f+1+DEPTH... = (f+1) % DEPTH after the write; with DEPTH 2 the request of group f+2 goes
  // into the set group f (just consumed) used
  auto iteration = [&](auto setw) __attribute__((always_inline)) -> bool {
    constexpr int SETW = decltype(setw)::value;               // set holding group f+1
    constexpr int SETR = DEPTH == 2 ? 1 - SETW : SETW;        // set the new request goes into
    const Pos nxt = advance(cur);                             // group f+1
    ahead = advance(ahead);
    if constexpr (DEPTH == 2) issue_loads(ahead.u, ahead.g, ahead.ok, std::integral_constant<int, SETR>{});
    else issue_loads(nxt.u, nxt.g, nxt.ok, std::integral_constant<int, SETR>{});
    __builtin_amdgcn_sched_barrier(0);    // keep the requests in front of the sweep
    // group g (g >= NPRE) completes the window of step g - NPRE (output planes zc0 + SZ*(g - NPRE) ..): its first input
    // plane z0 - 1 sits two ring slots before this group's plane 0
    int rpos = wpos - 2;
    rpos += rpos < 0 ? NRING : 0;
    sweep(cur.u, cur.u.zc0 + SZ * (cur.g - NPRE), rpos, cur.g >= NPRE);
    __builtin_amdgcn_sched_barrier(0);
    int wnext = wpos + SZ;
    wnext -= wnext >= NRING ? NRING : 0;
    write_group(wnext, std::integral_constant<int, SETW>{});   // past the end: zeros into a slot nobody reads
    __syncthreads();
    wpos = wnext;
    cur = nxt;
    return nxt.ok;
  };
  if constexpr (DEPTH == 2) {
    while (true) {
      if (!iteration(std::integral_constant<int, 1>{})) break;
      if (!iteration(std::integral_constant<int, 0>{})) break;
    }
  } else {
    while (iteration(std::integral_constant<int, 0>{})) {}
  }
}

// packed[s*64 + lane]: lane (co = lane & 15, kq = lane >> 4) holds W[co][ch = (kq&1)*8 + e][tap(s, kq>>1)], e = 0..7
__global__ void pack_c0cl_kernel(const float* __restrict__ w, u32x4* __restrict__ packed, int Cin, int Cout) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NSTEP * 64) return;
  const int s = idx >> 6, lane = idx & 63, co = lane & 15, kq = lane >> 4, up = kq >> 1;
  int tz, ty, tx;
  bool real = true;
  if (s < 9) { tz = up; ty = s / 3; tx = s % 3; }
  else if (s < 12) { tz = 2; ty = s - 9; tx = up; }
  else if (s == 12) { tz = 2; ty = up; tx = 2; }
  else { tz = 2; ty = 2; tx = 2; real = !up; }
  const int tap = (tz * 3 + ty) * 3 + tx;
  unsigned r[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    unsigned pair = 0u;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int ch = (kq & 1) * 8 + 2 * p + hh;
      const float v = (real && ch < Cin && co < Cout) ? w[((int64_t)co * Cin + ch) * 27 + tap] : 0.0f;
      pair |= (unsigned)to_bf16(v) << (16 * hh);
    }
    r[p] = pair;
  }
  packed[idx] = (u32x4){r[0], r[1], r[2], r[3]};
}

}  // namespace

// ---- internal entry points (conv3d_bf16.hip dispatches to them; declared in lr_common.h)
int64_t lr_internal_conv0_cl_bf16_packed_bytes(int Cin, int Cout) {
  return (Cin >= 1 && Cin <= 16 && Cout == 16) ? (int64_t)NSTEP * 64 * 16 : 0;
}

int lr_internal_conv0_cl_bf16_pack(const float* weight, void* packed, int Cin, int Cout, hipStream_t st) {
  if (lr_internal_conv0_cl_bf16_packed_bytes(Cin, Cout) == 0) return LR_OK;
  hipLaunchKernelGGL(pack_c0cl_kernel, dim3((NSTEP * 64 + 255) / 256), dim3(256), 0, st, weight, reinterpret_cast<u32x4*>(packed), Cin, Cout);
  return lr_launch_status();
}

// LR_EUNSUPPORTED -> the caller falls back to the channel-pass kernel
// clin: `in` is the (B,D,W,H,16) bf16 channels-last encoder input of lr_backproject_encin_bf16 instead of fp32 NCDHW
int lr_internal_conv0_cl_bf16(const float* in, const void* packed, const float* bias, void* out, int B, int Cin, int Cout, int D,
                              int W, int H, int out_layout, float slope, long long out_bs, int clin, hipStream_t st) {
  if (Cin < 1 || Cin > 16 || Cout != 16 || (reinterpret_cast<uintptr_t>(in) & 15u)) return LR_EUNSUPPORTED;
  if (!clin && (H & 3)) return LR_EUNSUPPORTED;
  const int64_t V = (int64_t)D * W * H;
  if ((int64_t)(clin ? 8 : Cin) * V * 4 + (int64_t)8 * W * H * 32 >= 0x7fffffffLL) return LR_EUNSUPPORTED;   // 31-bit byte offsets inside one batch element
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  int blocks = cus;
  blocks = lr_sw_int(LR_SW_CONV0_CL_BLOCKS, blocks);   // tuning aid
  int shape = 116;                                                          // planes x rows of a step: 4x4 | 2x8 | 1x16
  shape = lr_sw_int(LR_SW_C0CL_SHAPE, shape);
  if (shape != 44 && shape != 28) shape = 116;
  const int SZv = shape == 44 ? 4 : (shape == 28 ? 2 : 1), BYv = 16 / SZv;
  C0Dims d;
  d.B = B; d.Cin = Cin; d.D = D; d.W = W; d.H = H;
  d.nHq = (H + BX - 1) / BX; d.nWq = (W + BYv - 1) / BYv;
  // z chunks: enough units (column chunks) for ~8 per block, chunks of at least 16 planes (each costs 2 halo planes + one
  // exposed load latency)
  const int64_t cols = (int64_t)B * d.nWq * d.nHq;
  int nch = (int)((8 * (int64_t)blocks + cols - 1) / cols);
  nch = lr_sw_int(LR_SW_C0CL_CHUNKS, nch);          // tuning aid
  if (nch > D / 16) nch = D / 16;
  if (nch < 1) nch = 1;
  d.ZC = ((D + nch - 1) / nch + SZv - 1) / SZv * SZv;
  d.nch = (D + d.ZC - 1) / d.ZC;
  const int64_t nu = cols * d.nch;
  if (nu > 0x7fffffffLL) return LR_EINVAL;
  d.nunits = (int)nu; d.slope = slope;
  d.out_bs = out_bs ? out_bs : (long long)16 * D * W * H;
  d.abl = 0;
#ifdef LR_C0CL_ABLATIONS   // diagnostic build only (make -B EXTRA=-DLR_C0CL_ABLATIONS, tools/abl_c0cl.py): the switches give WRONG results
  if (const char* e = getenv("LIFTREG_C0CL_ABL")) d.abl = atoi(e);
#endif
  if (blocks > d.nunits) blocks = d.nunits;
  if (blocks < 1) blocks = 1;
  const int CQ = (Cin + 3) / 4;
  const bool hps = out_layout == LR_LAYOUT_BF16_NDHWC_HPS;
  const u32x4* wt = reinterpret_cast<const u32x4*>(packed);
  u16* o = reinterpret_cast<u16*>(out);
  const size_t ldsb = (size_t)(2 * SZv + 2) * (BYv + 2) * RB + SLOT + 64;   // + the dump area
#define LR_C0CL3(CQV, HP, SZV, BYV)                                                                                        \
  do {                                                                                                                     \
    static std::atomic<uint64_t> attr_done{0};                                                                             \
    if (lr_raise_dyn_lds(reinterpret_cast<const void*>(&conv0_cl_bf16_kernel<CQV, HP, SZV, BYV>), ldsb, attr_done) != LR_OK) return LR_ELAUNCH; \
    hipLaunchKernelGGL((conv0_cl_bf16_kernel<CQV, HP, SZV, BYV>), dim3((unsigned)blocks), dim3(512), ldsb, st, in, wt, bias, o, d);  \
  } while (0)
  if (clin) {   // staged from the channels-last bf16 encoder input: one instance per output layout (1 x 16 steps)
    if (shape != 116) return LR_EUNSUPPORTED;
#define LR_C0CLIN(HP)                                                                                                      \
  do {                                                                                                                     \
    static std::atomic<uint64_t> attr_done{0};                                                                             \
    if (lr_raise_dyn_lds(reinterpret_cast<const void*>(&conv0_cl_bf16_kernel<4, HP, 1, 16, true>), ldsb, attr_done) != LR_OK) return LR_ELAUNCH; \
    hipLaunchKernelGGL((conv0_cl_bf16_kernel<4, HP, 1, 16, true>), dim3((unsigned)blocks), dim3(512), ldsb, st, in, wt, bias, o, d);  \
  } while (0)
    if (hps) LR_C0CLIN(true); else LR_C0CLIN(false);
#undef LR_C0CLIN
    return lr_launch_status();
  }
#define LR_C0CL(CQV, HP) do { if (shape == 44) LR_C0CL3(CQV, HP, 4, 4); else if (shape == 28) LR_C0CL3(CQV, HP, 2, 8); else LR_C0CL3(CQV, HP, 1, 16); } while (0)
  if (CQ == 1) { if (hps) LR_C0CL(1, true); else LR_C0CL(1, false); }
  else if (CQ == 2) { if (hps) LR_C0CL(2, true); else LR_C0CL(2, false); }
  else if (CQ == 3) { if (hps) LR_C0CL(3, true); else LR_C0CL(3, false); }
  else { if (hps) LR_C0CL(4, true); else LR_C0CL(4, false); }
#undef LR_C0CL3
#undef LR_C0CL
  return lr_launch_status();
}
