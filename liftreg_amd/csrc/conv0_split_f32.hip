// conv0_split_f32.hip — the encoder's first block with few input channels (Cin <= 4 planar fp32 channels, stride 1, 16 output
// channels, channels-last output) as a z-MARCHING kernel on the bf16 matrix pipe.  Two arithmetic contracts, one kernel:
//   NS = 1, bf16 output  — the bf16 storage contract of conv3d_bf16.hip (operands rounded once to bf16, exact products, fp32
//                          accumulation, fp32 bias + LeakyReLU, bf16 store): the DEFAULT first block of the bf16 variant for
//                          Cin <= 4 on planes >= 128^2 (C3 bf16, C5), inference and training forward (+ LeakyReLU sign mask);
//                          1.75 -> 1.09 ms at C3 (5.4 TB/s);
//   NS = 3, fp32 output  — fp32 operands as EXACT three-way bf16 splits, described next; opt-in (LIFTREG_CONV0_SPLIT=1),
//                          DESIGN.md 6.6.
//
// The fp32 MFMA (v_mfma_f32_16x16x4_f32, 157 TFLOP/s) bounds this block: 2.9 ms at C3 with the Winograd sweep of conv3d.hip
// against a memory floor of 10.2 GB ~ 2.1 ms.  v_mfma_f32_16x16x32_bf16 multiplies 8x the K in half the cycles, and an fp32
// number is exactly the sum of three bf16 numbers:  x = x0 + x1 + x2  with  x0 = bf16(x), x1 = bf16(x - x0),
// x2 = bf16(x - x0 - x1)  (round to nearest even; both differences are exact in fp32 and the last one has at most 8
// significant bits, so nothing is lost: 3 x 8 = 24 bits).  With the weights split the same way,
//     x * w = sum over (s, t) of x_s * w_t        — nine products, each exact in fp32 (8 x 8 bits),
// and the kernel accumulates, in fp32 on the matrix pipe, the six with s + t <= 2: (0,0) (0,1) (1,0) (0,2) (2,0) (1,1).
// The three dropped ones are bounded by |x w| * (2 * 2^-8 * 2^-16 + 2^-32) = 2^-23 |x w| — one to two fp32 roundings of
// the product, the same class of error as the fp32 accumulation itself (a direct conv: no Winograd transform error);
// LR_C0S_PRODUCTS=9 (build flag) keeps all nine.  Six bf16 MFMAs of 16 cycles per K = 32 against eight fp32 MFMAs of 32
// cycles: 2.7x fewer matrix cycles; the block becomes bound by the 64 bytes per voxel it writes.
//
// Structure (that of conv0_cl_bf16.hip): persistent 4-wave blocks (two per CU; four with NS = 1) march DOWN z through chunks of (8 rows x 64 columns)
// columns of the volume; the LDS holds a ring of 4 input planes, each (10 rows x 72 voxels) as three arrays of 8-byte
// records (4 channels of one voxel in bf16; channels Cin..3 zero) — one array per split; iteration f requests plane f+1
// (16-byte bounds-checked buffer loads: outside the volume -> 0 = the conv's padding), sweeps output plane f out of the
// ring, splits plane f+1 and writes it over the slot whose last reader finished an iteration ago: ONE barrier per plane,
// every input plane fetched once per chunk.
//   K of an MFMA = 8 taps x 4 channels; lane group kq supplies two taps that sit a FIXED distance apart in the LDS, so
//   two ds_read_b64 deliver the operand in four consecutive registers (no assembly moves; NOT merged into a ds_read2_b64,
//   which takes four times the LDS cycles — see LR_C0S_NO_DS_MERGE below):
//     k-block ty = 0,1,2 (next voxel):  kq = 0..2: (tz=kq, ty, tx=0) | (tz=kq, ty, tx=1);   kq = 3: (tz=0, ty, tx=2) | weight 0
//     k-block 3 (next window row):      kq = 0: (1,0,2) | (1,1,2);  kq = 1: (1,2,2) | weight 0;  kq = 2: (2,0,2) | (2,1,2);
//                                       kq = 3: (2,2,2) | weight 0
//   (a "weight 0" slot multiplies a real neighbouring voxel by zero: a non-finite input reaches one voxel further than
//   its 3x3x3 neighbourhood)
//   4 k-blocks x 6 products = 24 MFMAs per 16-voxel x 16-cout tile; a wave owns two output rows of four tiles and shares the
//   row-type fragments between them (input row iy = tap row ty of output row 0 = ty + 1 of output row 1).
//   The (0,0) products and the rest go to separate accumulators (four independent chains per wave), summed at the end.
//
// Inf / NaN: x = +-Inf splits into (Inf, NaN, NaN), so an infinite input gives NaN where fp32 arithmetic may give +-Inf.
// Replaces (reference file:line): src/liftreg/layers/layers.py:365-369 (Conv3d + LeakyReLU) as wired at
//   src/liftreg/models/LiftRegDeformSubspaceBackproj.py:95-98 for the first encoder block.
#include "lr_common.h"
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#ifndef LR_C0S_PRODUCTS
#define LR_C0S_PRODUCTS 6
#endif
#ifndef LR_C0S_DEPTH
#define LR_C0S_DEPTH 4   // input planes in flight (register sets)
#endif
#ifndef LR_C0S_TFENCE
#define LR_C0S_TFENCE 1
#endif

#ifndef LR_C0S_BY
#define LR_C0S_BY 8
#endif
constexpr int BX = 64, BY = LR_C0S_BY, WR = BY + 2;
constexpr int NTHR = BY / 2 * 64;        // a wave = two output rows of four 16-voxel tiles
constexpr int NQ = 18;                   // aligned float4 quads of a window row: x0-4 .. x0+67
constexpr int NREC = 4 * NQ;             // voxel records of a row: p = x - x0 + 4
constexpr int SB = NREC * 8;             // bytes of one split of a row
constexpr int NRING = 4;
constexpr int ZPAD = 4096;               // zeros behind the ring for the lanes without a tap (largest immediate: RB + 2 SB + 3*128)
constexpr int NITEMS = WR * NQ;          // staging items of a plane: one x-quad of a row, all channels
constexpr int NKB = 4;
constexpr unsigned OOR = 0x80000000u;
static_assert(NITEMS <= NTHR, "one staging item per thread");
// NS = arrays per window row: 3 (the fp32 kernel: one per split) or 1 (the bf16-contract kernel: operands rounded once)
template <int NS>
struct SGeo {
  static constexpr int RB = NS * SB;               // a window row: [split 0][split 1][split 2]
  static constexpr int PLB = WR * RB;              // a ring plane (NS = 3 and 1: 128 mod 256 bytes — neighbouring planes
                                                   // fall into opposite bank halves for the 32-lane ds_read_b64 groups)
  static constexpr int LDSB = NRING * PLB + ZPAD;
  static_assert(RB + 2 * SB + 3 * 128 + 8 <= ZPAD, "zero area covers every immediate");
};

// products (data split s, weight split t) in issue order: small ones first, the (0,0) product last
constexpr int NPROD = LR_C0S_PRODUCTS;
static_assert(NPROD == 6 || NPROD == 9, "six or nine partial products");
__host__ __device__ constexpr int prod_s(int p) {
  constexpr int S6[6] = {1, 0, 2, 0, 1, 0}, S9[9] = {2, 1, 2, 1, 0, 2, 0, 1, 0};
  return NPROD == 6 ? S6[p] : S9[p];
}
__host__ __device__ constexpr int prod_t(int p) {
  constexpr int T6[6] = {1, 2, 0, 1, 0, 0}, T9[9] = {2, 2, 1, 1, 2, 0, 1, 0, 0};
  return NPROD == 6 ? T6[p] : T9[p];
}

struct S0Dims {
  int B, Cin, D, W, H;
  int nHq, nWq, nch, ZC;   // column grid (x, y), z chunks per column, planes per chunk
  int nunits;              // B * nch * nWq * nHq
  long long bs0, bsr;      // elements between batch elements of channel 0 | of channels 1..Cin-1
  long long out_bs;        // output elements between batch elements (dense: 16*D*W*H)
  float slope;
  int abl;   // timing-only ablation bits (diagnostic build, WRONG results): 1 no global loads, 2 no stores, 4 no sweep, 8 no split / LDS writes
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

// the 8 XCDs (block id % 8) take contiguous eighths of the unit order (x fastest, then y, z chunk, batch); the blocks of an
// XCD stride through their eighth together: neighbouring columns (shared halo rows) meet in one L2
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

// ds_read_b64 takes 2 LDS cycles per wave, ds_read2_b64 takes 8 for twice the bytes (MI355X_MICROARCH.md, LDS table): the
// kernel reads every operand half with its own ds_read_b64.  hipcc merges neighbouring DS reads twice — the IR load/store
// vectorizer (off for this file: Makefile) and the machine-level SI load/store optimizer (off for this kernel: the
// attribute below; it also paired halves of DIFFERENT operands, which cost four moves per operand to undo).
#if defined(__HIP_DEVICE_COMPILE__)
#define LR_C0S_NO_DS_MERGE __attribute__((target("no-load-store-opt")))
#else
#define LR_C0S_NO_DS_MERGE
#endif
// NS = 3, BF16OUT = false: the fp32 kernel described above.  NS = 1, BF16OUT = true: the SAME march under the bf16 storage
// contract of conv3d_bf16.hip (inputs and weights rounded once to bf16, exact products, fp32 accumulation, fp32 bias +
// LeakyReLU, bf16 channels-last store): one array per window row, one product, 4 MFMAs per tile — the 3-channel first block
// of the bf16 variant, bound by the 32 bytes per voxel it writes.
// MASK (bf16 training forward): also the LeakyReLU sign mask of the STORED output, LR_LAYOUT_SIGN4 (B,D,W,H,4) uint8 — bit r
// of byte q = "stored bf16 of channel 4q+r > 0" — which the next block's data gradient reads instead of the activation.
template <int NC, bool HPSOUT, int NS = 3, bool BF16OUT = false, bool MASK = false>
__global__ __launch_bounds__(NTHR, (NS == 1 && NC <= 3 ? 4 : 2)) LR_C0S_NO_DS_MERGE void conv0_split_f32_kernel(const float* __restrict__ in0, const float* __restrict__ in_rest,
                                                                 const u32x4* __restrict__ wp, const float* __restrict__ bias,
                                                                 void* __restrict__ out, S0Dims d, unsigned char* __restrict__ mask_out) {
  static_assert(!MASK || BF16OUT, "the sign mask belongs to the bf16 training forward");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int RB = SGeo<NS>::RB, PLB = SGeo<NS>::PLB, LDSB = SGeo<NS>::LDSB;
  constexpr int NP = NS == 1 ? 1 : NPROD;
  const int tid = threadIdx.x, lane = tid & 63;
  const int rh = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave = pair of output rows
  const int col = lane & 15, kq = lane >> 4;
  const int dD = d.D, dW = d.W, dH = d.H;
  const unsigned V4 = (unsigned)dD * dW * dH * 4u;            // bytes of one channel volume (< 2^31 / 3: checked by the launcher)

  // zero everything once: the zero area stays zero, and a "weight 0" operand slot may read a ring row that was never
  // written (the row behind a plane's last one is the next slot's first) — it must hold finite numbers
  for (int o = tid * 16; o < LDSB; o += NTHR * 16) *reinterpret_cast<u32x4*>(lds + o) = (u32x4){0u, 0u, 0u, 0u};
  __syncthreads();

  // this thread's staging item: x-quad q of window row `row`
  const bool item_live = tid < NITEMS;
  const int irow = item_live ? tid / NQ : 0, iq = item_live ? tid % NQ : 0;
  const unsigned g_rel = (unsigned)((irow * dH + 4 * iq) * 4);
  // LDS byte offset of the item = l_rec + slot * l_mul; a dead thread (it loads zeros) writes them into the zero area
  const unsigned l_rec = item_live ? (unsigned)(irow * RB + iq * 32) : (unsigned)(NRING * PLB);
  const unsigned l_mul = item_live ? (unsigned)PLB : 0u;

  // weights: [k-block][split] fragments, 48 registers for the life of the block
  u32x4 w[NKB][NS];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int t = 0; t < NS; ++t) w[kb][t] = wp[(kb * 3 + t) * 64 + lane];
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bias[kq * 4 + r];
  }

  int first, stride, end;
  unit_range((int)blockIdx.x, (int)gridDim.x, d.nunits, first, stride, end);

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
  auto uniform = [](const Unit& v) __attribute__((always_inline)) -> Unit {   // block-uniform by construction: scalar registers
    Unit r;
    r.b = __builtin_amdgcn_readfirstlane(v.b); r.zc0 = __builtin_amdgcn_readfirstlane(v.zc0);
    r.zc1 = __builtin_amdgcn_readfirstlane(v.zc1); r.y0 = __builtin_amdgcn_readfirstlane(v.y0);
    r.x0 = __builtin_amdgcn_readfirstlane(v.x0);
    return r;
  };
  auto make_rsrc = [](const void* p, unsigned bytes) __attribute__((always_inline)) -> __amdgpu_buffer_rsrc_t {
    // base and size through readfirstlane: otherwise hipcc keeps the descriptor in vector registers (waterfall loops)
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    const uint64_t s = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                       (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(s), (short)0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
  };

  // The input loads are inline asm, and so are the waits for them: hipcc's wait-count pass gives up on requests that stay in
  // flight across a loop's back edge — in the first copy of the unrolled loop body it waited for EVERY load in flight
  // (vmcnt(8) instead of vmcnt(41)), one pipeline drain per trip.  The counts are static: a plane is NC loads, a sweep is 8
  // stores, nothing else in the loop touches vector memory; wait_set() names how many of them are younger than the set it
  // needs.  (The compiler does not know these loads exist: every use of a set goes through wait_set's "+v" operands.)
  auto make_srd = [](const void* p, unsigned bytes) __attribute__((always_inline)) -> i32x4 {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(a >> 32)) & 0xffff;   // stride 0
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
  };
  f32x4 ld[LR_C0S_DEPTH][NC];   // planes in flight: register set = flat plane index % DEPTH
  // plane g of a unit = input plane z = zc0 - 1 + g; the step of output plane zc0 + i needs planes i, i+1, i+2
  auto issue_loads = [&](const Unit& uv, int gv, bool validv, auto setc) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value;
    const Unit u = uniform(uv);
    const int g = __builtin_amdgcn_readfirstlane(gv);
    const bool valid = __builtin_amdgcn_readfirstlane((int)validv) != 0;
    const i32x4 r0 = make_srd(in0 + (int64_t)u.b * d.bs0, V4);
    const i32x4 r1 = make_srd(in_rest + (int64_t)u.b * d.bsr, NC > 1 ? (unsigned)(NC - 1) * V4 : 0u);
    const int zi = u.zc0 - 1 + g;
    const int org = ((zi * dW + (u.y0 - 1)) * dH + (u.x0 - 4)) * 4;   // may be negative: only used where the element exists
    const int yi = u.y0 - 1 + irow, xi = u.x0 - 4 + 4 * iq;
    // bitwise, no short circuits (a branch around a load costs a full vmcnt drain)
    const int ok = (int)valid & (int)item_live & (int)(zi >= 0) & (int)(zi < dD) & (int)(yi >= 0) & (int)(yi < dW) & (int)(xi >= 0) & (int)(xi < dH);
    unsigned dead = ((unsigned)ok - 1u) & OOR;   // dead item: bit 31 -> outside the resource -> 0
#ifdef LR_C0S_ABLATIONS
    if (d.abl & 1) dead = OOR;
#endif
    const unsigned voff = (unsigned)org + g_rel;
    f32x4 (&L)[NC] = ld[SET];
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(L[0]) : "v"(voff | dead), "s"(r0));
#pragma unroll
    for (int c = 1; c < NC; ++c)
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(L[c]) : "v"((voff + (unsigned)(c - 1) * V4) | dead), "s"(r1));
  };
  // wait until at most `younger` vector-memory operations issued after set SET's loads are outstanding
  auto wait_set = [&](auto setc, auto youngerc) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value, NY = decltype(youngerc)::value;
    constexpr int N = NY > 63 ? 63 : NY;   // vmcnt is a 6-bit counter: waiting for fewer outstanding operations is always safe
    static_assert(N >= 0, "younger operations");
    f32x4 (&L)[NC] = ld[SET];   // (named outside the asm statements: a generic lambda does not capture through an asm operand)
    if constexpr (NC == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(L[0]) : "n"(N));
    else if constexpr (NC == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(L[0]), "+v"(L[NC > 1 ? 1 : 0]) : "n"(N));
    else if constexpr (NC == 3) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(L[0]), "+v"(L[NC > 1 ? 1 : 0]), "+v"(L[NC > 2 ? 2 : 0]) : "n"(N));
    else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(L[0]), "+v"(L[NC > 1 ? 1 : 0]), "+v"(L[NC > 2 ? 2 : 0]), "+v"(L[NC > 3 ? 3 : 0]) : "n"(N));
  };
  auto write_plane = [&](int slotv, auto setc) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value;
    const int slot = __builtin_amdgcn_readfirstlane(slotv);
#ifdef LR_C0S_ABLATIONS
    if (d.abl & 8) return;
#endif
    // no branch: a path around the waits for the loads would make every later wait in the loop conservative (vmcnt(0) with
    // the stores of the sweep in flight).  A dead thread loaded zeros and writes them into the zero area.
    unsigned rec[3][4][2];   // [split][voxel][channels 01 | 23]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned p01[3], p23[3] = {0u, 0u, 0u};
      split3(ld[SET][0][j], NC > 1 ? ld[SET][NC > 1 ? 1 : 0][j] : 0.0f, p01);
      if (NC > 2) split3(ld[SET][NC > 2 ? 2 : 0][j], NC > 3 ? ld[SET][NC > 3 ? 3 : 0][j] : 0.0f, p23);
#pragma unroll
      for (int s = 0; s < 3; ++s) { rec[s][j][0] = p01[s]; rec[s][j][1] = p23[s]; }
    }
    unsigned char* const base = lds + ((unsigned)slot * l_mul + l_rec);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      *reinterpret_cast<u32x4*>(base + s * SB) = (u32x4){rec[s][0][0], rec[s][0][1], rec[s][1][0], rec[s][1][1]};
      *reinterpret_cast<u32x4*>(base + s * SB + 16) = (u32x4){rec[s][2][0], rec[s][2][1], rec[s][3][0], rec[s][3][1]};
    }
  };
  // output plane z0; rpos = ring slot of input plane z0 - 1
  // (livev false: the window is not complete yet — the first two planes of a chunk; the sweep still runs, on whatever the
  // ring holds, and its stores are dropped: a branch around it would leave two different store counts behind the loads of
  // the next plane and force the wait for them down to vmcnt(0) = every step would wait for its own stores)
  auto sweep = [&](const Unit& uv, int z0v, int rposv, bool livev) __attribute__((always_inline)) {
    const Unit u = uniform(uv);
    const int z0 = __builtin_amdgcn_readfirstlane(z0v), rpos = __builtin_amdgcn_readfirstlane(rposv);
    int live = __builtin_amdgcn_readfirstlane((int)livev);
#ifdef LR_C0S_ABLATIONS
    if (d.abl & 2) live = 0;
    if (d.abl & 4) return;
#endif
    const int s0 = rpos, s1 = (rpos + 1) & 3, s2 = (rpos + 2) & 3;
    // per-lane fragment bases.  A lane's two taps are ALWAYS `dlt` bytes apart (row-type k-blocks: the next voxel, 8 bytes;
    // k-block 3: the next window row), so one ds_read2_b64 delivers the MFMA operand in four consecutive registers.
    const unsigned pF = (unsigned)((kq == 1 ? s1 : kq == 2 ? s2 : s0) * PLB + 2 * rh * RB + (col + 3 + (kq == 3 ? 2 : 0)) * 8);
    const unsigned pG = (unsigned)((kq < 2 ? s1 : s2) * PLB + (2 * rh + (kq & 1) * 2) * RB + (col + 5) * 8);
    // output row 1's k-block-3 pair starts where row 0's ends; behind an opaque copy of the base the compiler cannot fold
    // the two loads of that record into one (which would cost four moves per operand to rebuild the register tuples)
    unsigned pG1 = pG + RB;
    asm volatile("" : "+v"(pG1));
    auto frag = [&](unsigned pa, int off, int dlt) __attribute__((always_inline)) -> bf16x8 {
      const u32x2 a = *reinterpret_cast<const u32x2*>(lds + pa + off);
      const u32x2 b = *reinterpret_cast<const u32x2*>(lds + pa + off + dlt);
      return __builtin_bit_cast(bf16x8, (u32x4){a[0], a[1], b[0], b[1]});
    };
    // one buffer resource per output plane: 31-bit offsets inside W*H*16 output elements
    constexpr int OSZ = BF16OUT ? 2 : 4;
    unsigned char* const pbase = reinterpret_cast<unsigned char*>(out) + ((int64_t)u.b * d.out_bs + (int64_t)(live ? z0 : 0) * dW * dH * 16) * OSZ;
    const __amdgpu_buffer_rsrc_t ores = make_rsrc(pbase, live ? (unsigned)(dW * dH * 16 * OSZ) : 0u);
    const __amdgpu_buffer_rsrc_t mres = make_rsrc(MASK ? mask_out + ((int64_t)u.b * dD + (live ? z0 : 0)) * dW * dH * 4 : pbase,
                                                  (MASK && live) ? (unsigned)(dW * dH * 4) : 0u);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 hi[2] = {bv, bv}, lo[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      bf16x8 F[4][NS], G[2][NS];
#pragma unroll
      for (int iy = 0; iy < 4; ++iy)
#pragma unroll
        for (int s = 0; s < NS; ++s) F[iy][s] = frag(pF, iy * RB + s * SB + t * 128, 8);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int s = 0; s < NS; ++s) G[r][s] = frag(r ? pG1 : pG, s * SB + t * 128, RB);
      // (fragment of row 0, its k-block) paired with (fragment of row 1, its k-block): the two rows alternate on the pipe
      auto pair = [&](const bf16x8 (&f0)[NS], int kb0, const bf16x8 (&f1)[NS], int kb1) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int s = NS == 1 ? 0 : prod_s(p), tw = NS == 1 ? 0 : prod_t(p);
          if (s == 0 && tw == 0) {
            hi[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[kb0][0]), f0[0], hi[0], 0, 0, 0);
            hi[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[kb1][0]), f1[0], hi[1], 0, 0, 0);
          } else {
            lo[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[kb0][NS == 1 ? 0 : tw]), f0[NS == 1 ? 0 : s], lo[0], 0, 0, 0);
            lo[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[kb1][NS == 1 ? 0 : tw]), f1[NS == 1 ? 0 : s], lo[1], 0, 0, 0);
          }
        }
      };
      pair(F[0], 0, F[3], 2);
      pair(F[1], 1, F[1], 0);
      pair(F[2], 2, F[2], 1);
      pair(G[0], 3, G[1], 3);
      const int x = u.x0 + t * 16 + col;
      const int hp = HPSOUT ? (x & 1) * (dH >> 1) + (x >> 1) : x;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int y = u.y0 + 2 * rh + r;
        const int ok = (int)(z0 < u.zc1) & (int)(y < dW) & (int)(x < dH);
        const unsigned off = (unsigned)(((y * dH + hp) * 16 + kq * 4) * OSZ) | (((unsigned)ok - 1u) & OOR);
        f32x4 v = NS == 1 ? hi[r] : hi[r] + lo[r];
        v = __builtin_elementwise_max(v, v * d.slope);   // = LeakyReLU for 0 <= slope <= 1 (checked by the launcher)
        if constexpr (BF16OUT) {
          const bf16x2 a = __builtin_convertvector((f32x2){v[0], v[1]}, bf16x2), b2 = __builtin_convertvector((f32x2){v[2], v[3]}, bf16x2);
          const u32x2 pk = {__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b2)};
          __builtin_amdgcn_raw_buffer_store_b64(pk, ores, off, 0, 2 /* nt */);
          if constexpr (MASK) {
            // this lane's channel quad -> one byte; the four quads of a voxel sit in the four 16-lane rows of the wave: the
            // row / half swaps bring them into row 0, whose lanes store the voxel's four bytes as one dword
            const unsigned m = ((short)(pk[0] & 0xffffu) > 0 ? 1u : 0u) | ((short)(pk[0] >> 16) > 0 ? 2u : 0u) |
                               ((short)(pk[1] & 0xffffu) > 0 ? 4u : 0u) | ((short)(pk[1] >> 16) > 0 ? 8u : 0u);
            const auto s1 = __builtin_amdgcn_permlane32_swap(m, m, false, false);     // [1]: rows 0,1 <- rows 2,3 of m
            const unsigned xa = s1[0], xb = s1[1];
            const auto s2 = __builtin_amdgcn_permlane16_swap(xa, xa, false, false);   // [1]: row 0 <- row 1 of m
            const auto s3 = __builtin_amdgcn_permlane16_swap(xb, xb, false, false);   // [1]: row 0 <- row 3 of m
            const unsigned dw = (m & 0xffu) | (s2[1] & 0xffu) << 8 | (xb & 0xffu) << 16 | (s3[1] & 0xffu) << 24;
            const unsigned moff = (unsigned)((y * dH + x) * 4) | (((unsigned)(ok & (int)(kq == 0)) - 1u) & OOR);
            __builtin_amdgcn_raw_buffer_store_b32(dw, mres, moff, 0, 0);
          }
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ores, off, 0, 2 /* nt */);
        }
      }
#if LR_C0S_TFENCE
      __builtin_amdgcn_sched_barrier(0);   // keep the next tile column's 30 fragment reads behind this one's MFMAs (registers)
#endif
    }
  };

  if (first >= end) return;
  // flat walk over (unit, plane): `cur` = the plane written last, whose step is swept now while the next plane is in flight
  struct Pos { Unit u; int g, ng, uid; bool ok; };
  auto advance = [&](const Pos& p) __attribute__((always_inline)) -> Pos {
    Pos n = p;
    n.g = p.g + 1;
    if (n.g >= p.ng) {
      n.uid = p.uid + stride;
      n.g = 0;
      n.ok = p.ok && n.uid < end;
      if (n.ok) { n.u = decode(n.uid); n.ng = n.u.zc1 - n.u.zc0 + 2; }
    }
    return n;
  };
  // `cur` = the plane written last (its step is swept now); the next DEPTH planes are in flight in the register sets
  // (flat plane index % DEPTH), the one requested in an iteration arrives DEPTH iterations later.  One plane of look-ahead
  // left the blocks waiting for their loads behind 8.6 GB of stores (no loads at all: 2.28 ms, loads never waited for:
  // 2.33, one plane ahead: 2.9-3.1).
  constexpr int DEPTH = LR_C0S_DEPTH;
  static_assert(DEPTH == 1 || DEPTH == 2 || DEPTH == 4, "register sets");
  Pos cur;
  cur.u = decode(first); cur.g = 0; cur.uid = first; cur.ok = true;
  cur.ng = cur.u.zc1 - cur.u.zc0 + 2;
  issue_loads(cur.u, 0, true, std::integral_constant<int, 0>{});
  Pos ahead = cur;   // the newest plane requested
  if constexpr (DEPTH >= 2) { ahead = advance(ahead); issue_loads(ahead.u, ahead.g, ahead.ok, std::integral_constant<int, 1>{}); }
  if constexpr (DEPTH >= 4) {
    ahead = advance(ahead); issue_loads(ahead.u, ahead.g, ahead.ok, std::integral_constant<int, 2>{});
    ahead = advance(ahead); issue_loads(ahead.u, ahead.g, ahead.ok, std::integral_constant<int, 3>{});
  }
  int wpos = 0;   // ring slot of the plane written last
  wait_set(std::integral_constant<int, 0>{}, std::integral_constant<int, (DEPTH - 1) * NC>{});
  write_plane(wpos, std::integral_constant<int, 0>{});
  ahead = advance(ahead);
  issue_loads(ahead.u, ahead.g, ahead.ok, std::integral_constant<int, 0>{});
  __syncthreads();
  // one iteration: sweep the step of plane f (8 stores), write plane f+1 out of its register set SETW and request plane
  // f+1+DEPTH into the same set at once.  YOUNGER = vector-memory operations issued after the loads of set SETW:
  // (DEPTH-1) planes of loads and the stores of the sweeps since — DEPTH of them in the steady state, fewer in the first trip.
  auto iteration = [&](auto setw, auto nsweeps) __attribute__((always_inline)) -> bool {
    constexpr int SETW = decltype(setw)::value;
    constexpr int YOUNGER = (DEPTH - 1) * NC + (MASK ? 16 : 8) * decltype(nsweeps)::value;   // (MASK: 8 more stores per sweep)
    const Pos nxt = advance(cur);
    // plane g (g >= 2) completes the window of output plane zc0 + g - 2, whose first input plane sits two slots back
    sweep(cur.u, cur.u.zc0 + cur.g - 2, (wpos + 2) & 3, cur.g >= 2);
    __builtin_amdgcn_sched_barrier(0);
    const int wnext = (wpos + 1) & 3;
    wait_set(std::integral_constant<int, SETW>{}, std::integral_constant<int, YOUNGER>{});
    write_plane(wnext, std::integral_constant<int, SETW>{});   // past the end: zeros into a slot nobody reads
    ahead = advance(ahead);
    issue_loads(ahead.u, ahead.g, ahead.ok, std::integral_constant<int, SETW>{});
    __syncthreads();
    wpos = wnext;
    cur = nxt;
    return nxt.ok;
  };
  typedef std::integral_constant<int, DEPTH> Steady;
  // first trip: the sets were requested back to back in the prologue, k sweeps lie between them and iteration k
  if constexpr (DEPTH >= 2) { if (!iteration(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{})) return; }
  if constexpr (DEPTH >= 4) {
    if (!iteration(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{})) return;
    if (!iteration(std::integral_constant<int, 3>{}, std::integral_constant<int, 3>{})) return;
  }
  if constexpr (DEPTH == 1) {
    while (iteration(std::integral_constant<int, 0>{}, Steady{})) {}
  } else {
    while (true) {
      if (!iteration(std::integral_constant<int, 0>{}, Steady{})) break;
      if (!iteration(std::integral_constant<int, 1>{}, Steady{})) break;
      if constexpr (DEPTH == 4) {
        if (!iteration(std::integral_constant<int, 2>{}, Steady{})) break;
        if (!iteration(std::integral_constant<int, 3>{}, Steady{})) break;
      }
    }
  }
}

// packed[(kb*3 + t)*64 + lane]: lane (co = lane & 15, kq = lane >> 4) holds split t of W[co][ch = e & 3][tap(kb, kq, e >> 2)]
__global__ void pack_c0s_kernel(const float* __restrict__ w, u32x4* __restrict__ packed, int Cin, int Cout) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NKB * 64) return;
  const int kb = idx >> 6, lane = idx & 63, co = lane & 15, kq = lane >> 4;
  unsigned r[3][4];
#pragma unroll
  for (int pr = 0; pr < 4; ++pr) {
    float v[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int e = 2 * pr + hh, h = e >> 2, ch = e & 3;
      int tz = 0, ty = 0, tx = 0;
      bool real = false;
      if (kb < 3) {
        ty = kb;
        if (kq < 3) { tz = kq; tx = h; real = true; } else { tz = 0; tx = 2; real = h == 0; }
      } else {
        tz = kq < 2 ? 1 : 2; tx = 2;
        if (kq & 1) { ty = 2; real = h == 0; } else { ty = h; real = true; }
      }
      const int tap = (tz * 3 + ty) * 3 + tx;
      v[hh] = (real && ch < Cin && co < Cout) ? w[((int64_t)co * Cin + ch) * 27 + tap] : 0.0f;
    }
    unsigned p[3];
    split3(v[0], v[1], p);
#pragma unroll
    for (int t = 0; t < 3; ++t) r[t][pr] = p[t];
  }
#pragma unroll
  for (int t = 0; t < 3; ++t) packed[(kb * 3 + t) * 64 + lane] = (u32x4){r[t][0], r[t][1], r[t][2], r[t][3]};
}

}  // namespace

// ---- internal entry points (conv3d.hip dispatches to them; declared in lr_common.h)
int64_t lr_internal_conv0_split_packed_floats(int Cin, int Cout) {
  return (Cin >= 1 && Cin <= 4 && Cout == 16) ? (int64_t)NKB * 3 * 64 * 4 : 0;
}

int lr_internal_conv0_split_pack(const float* weight, float* packed, int Cin, int Cout, hipStream_t st) {
  if (lr_internal_conv0_split_packed_floats(Cin, Cout) == 0) return LR_OK;
  hipLaunchKernelGGL(pack_c0s_kernel, dim3(1), dim3(NKB * 64), 0, st, weight, reinterpret_cast<u32x4*>(packed), Cin, Cout);
  return lr_launch_status();
}

// Channel 0 of batch element b at in0 + b*bs0, channels 1..Cin-1 at in_rest + b*bsr + (c-1)*D*W*H (elements).
// LR_EUNSUPPORTED -> the caller runs its other kernels.  The rule looks at the plane (W, H) only, so a z-slab of a volume
// takes the same kernel as the whole volume.
template <int NS, bool BF16OUT, bool MASK = false>
static int launch_march(const float* in0, long long bs0, const float* in_rest, long long bsr, const void* packed, const float* bias,
                        void* out, int B, int Cin, int D, int W, int H, bool hps, float slope, long long out_bs, hipStream_t st,
                        unsigned char* mask_out = nullptr) {
  if (MASK && (Cin > 3 || (int64_t)W * H * 4 >= 0x7fffffffLL)) return LR_EUNSUPPORTED;
  if (Cin < 1 || Cin > 4 || (H & 3) || (reinterpret_cast<uintptr_t>(in0) & 15u) || (Cin > 1 && (reinterpret_cast<uintptr_t>(in_rest) & 15u)) ||
      (bs0 & 3) || (bsr & 3))
    return LR_EUNSUPPORTED;
  if (hps && (H & 1)) return LR_EUNSUPPORTED;
  if (!(slope >= 0.0f && slope <= 1.0f)) return LR_EUNSUPPORTED;   // the epilogue computes LeakyReLU as max(v, slope * v)
  if ((int64_t)W * H < 128 * 128) return LR_EUNSUPPORTED;   // small planes: too few columns for the persistent blocks
  const int64_t V = (int64_t)D * W * H;
  if ((int64_t)3 * V * 4 + (int64_t)8 * W * H * 4 >= 0x7fffffffLL) return LR_EUNSUPPORTED;   // 31-bit byte offsets inside a batch element's channels
  if ((int64_t)W * H * 64 >= 0x7fffffffLL) return LR_EUNSUPPORTED;
  constexpr int LDSB = SGeo<NS>::LDSB;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  int blocks = cus * (NS == 1 && Cin <= 3 ? 4 : (LDSB <= 80 * 1024 ? 2 : 1));   // blocks a CU holds (LDS, registers)
  blocks = lr_sw_int(LR_SW_CONV0_SPLIT_BLOCKS, blocks);   // tuning aid
  S0Dims d;
  d.B = B; d.Cin = Cin; d.D = D; d.W = W; d.H = H;
  d.nHq = (H + BX - 1) / BX; d.nWq = (W + BY - 1) / BY;
  // z chunks: at least two units per block, chunks of at least 16 planes (each costs 2 halo planes, two sweeps whose
  // results are dropped and one exposed latency: one chunk per column measured 2.92 ms at C3, four 3.14)
  const int64_t cols = (int64_t)B * d.nWq * d.nHq;
  int nch = (int)((2 * (int64_t)blocks + cols - 1) / cols);
  nch = lr_sw_int(LR_SW_CONV0_SPLIT_CHUNKS, nch);      // tuning aid
  if (nch > D / 16) nch = D / 16;
  if (nch < 1) nch = 1;
  d.ZC = (D + nch - 1) / nch;
  d.nch = (D + d.ZC - 1) / d.ZC;
  const int64_t nu = cols * d.nch;
  if (nu > 0x7fffffffLL) return LR_EINVAL;
  d.nunits = (int)nu; d.slope = slope;
  d.bs0 = bs0; d.bsr = bsr;
  d.out_bs = out_bs ? out_bs : (long long)16 * V;
  d.abl = 0;
#ifdef LR_C0S_ABLATIONS   // diagnostic build only: the switches give WRONG results
  if (const char* e = getenv("LIFTREG_C0S_ABL")) d.abl = atoi(e);
#endif
  if (blocks > d.nunits) blocks = d.nunits;
  if (blocks < 1) blocks = 1;
  const u32x4* wt = reinterpret_cast<const u32x4*>(packed);
  if (!in_rest) in_rest = in0;   // Cin == 1: never dereferenced (zero-length resource)
#define LR_C0S(NCV, HP)                                                                                                     \
  do {                                                                                                                      \
    static std::atomic<uint64_t> attr_done{0};                                                                              \
    if (lr_raise_dyn_lds(reinterpret_cast<const void*>(&conv0_split_f32_kernel<NCV, HP, NS, BF16OUT, MASK>), LDSB, attr_done) != LR_OK) return LR_ELAUNCH; \
    hipLaunchKernelGGL((conv0_split_f32_kernel<NCV, HP, NS, BF16OUT, MASK>), dim3((unsigned)blocks), dim3(NTHR), LDSB, st, in0, in_rest, wt, bias, out, d, mask_out); \
  } while (0)
  if (Cin == 1) { if (hps) LR_C0S(1, true); else LR_C0S(1, false); }
  else if (Cin == 2) { if (hps) LR_C0S(2, true); else LR_C0S(2, false); }
  else if (Cin == 3) { if (hps) LR_C0S(3, true); else LR_C0S(3, false); }
  else if constexpr (!MASK) { if (hps) LR_C0S(4, true); else LR_C0S(4, false); }
#undef LR_C0S
  return lr_launch_status();
}

#ifdef LR_EXPERIMENTAL   // the fp32 route of this kernel (three splits, fp32 output): superseded by conv01_fused.hip; `make exp` only
int lr_internal_conv0_split_f32(const float* in0, long long bs0, const float* in_rest, long long bsr, const float* packed,
                                const float* bias, float* out, int B, int Cin, int D, int W, int H, int out_layout, float slope,
                                long long out_bs, hipStream_t st) {
  if (out_layout != LR_LAYOUT_NDHWC && out_layout != LR_LAYOUT_NDHWC_HPS) return LR_EUNSUPPORTED;
  return launch_march<3, false>(in0, bs0, in_rest, bsr, packed, bias, out, B, Cin, D, W, H, out_layout == LR_LAYOUT_NDHWC_HPS, slope, out_bs, st);
}
#endif

// The 3-channel (Cin <= 4) first block of the bf16 variant: `in` is (B,Cin,D,W,H) fp32, `out` bf16 channels-last records;
// `packed` = the buffer of lr_internal_conv0_split_pack (its first split IS the nearest-even bf16 weight).
// mask_out != NULL (training forward, Cin <= 3, dense output): also the LR_LAYOUT_SIGN4 sign mask of the stored output.
int lr_internal_conv0_march_bf16(const float* in, const void* packed, const float* bias, void* out, int B, int Cin, int D, int W,
                                 int H, int out_layout, float slope, long long out_bs, unsigned char* mask_out, hipStream_t st) {
  if (out_layout != LR_LAYOUT_BF16_NDHWC && out_layout != LR_LAYOUT_BF16_NDHWC_HPS) return LR_EUNSUPPORTED;
  const int64_t V = (int64_t)D * W * H;
  const bool hps = out_layout == LR_LAYOUT_BF16_NDHWC_HPS;
  if (mask_out) {
    if (out_bs != 0 && out_bs != 16 * V) return LR_EUNSUPPORTED;
    return launch_march<1, true, true>(in, (long long)Cin * V, in + V, (long long)Cin * V, packed, bias, out, B, Cin, D, W, H, hps, slope,
                                       out_bs, st, mask_out);
  }
  return launch_march<1, true>(in, (long long)Cin * V, in + V, (long long)Cin * V, packed, bias, out, B, Cin, D, W, H, hps, slope, out_bs, st);
}
