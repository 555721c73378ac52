// conv01_fused.hip — the first TWO encoder blocks (Cin <= 5 planar fp32 channels -> 16 channels at stride 1 -> 32 channels at
// stride 2, each Conv3d 3x3x3 + bias + LeakyReLU) as ONE z-marching kernel on the bf16 matrix pipe with EXACT three-way bf16
// splits of every fp32 operand (conv0_split_f32.hip describes the split: x = x0 + x1 + x2, six of the nine partial products,
// each exact in fp32, fp32 accumulation).  The 16-channel fp32 activation between the two blocks (8.6 GB written + 8.6 GB read
// + halo re-reads at 256^3, B = 8 — 22 of the step's 47 GB) never reaches HBM: it lives in LDS as bf16 triples.
//
// Work decomposition: PRODUCER and CONSUMER waves.  A 512-thread block (one per CU: two waves per SIMD) owns a column of
// 4 x 8 block-1 outputs and marches DOWN the output planes.  Waves 0..3 ("A") compute block 0, waves 4..7 ("B") block 1, one
// plane of block-1 output apart, with ONE barrier per step; wave i and wave i + 4 share a SIMD.  Everything on a SIMD shares one
// matrix pipe and one vector issue port, and a wave issues in order: the step time follows the SUM of what the two waves of
// the busiest SIMD issue (ablation builds, profiles/NOTES_r05.md), so the work is dealt for equal sums.
//   step s:  A: planes z = 2s, 2s+1 of block 0's output on the (9 x 17) region the 4 x 8 tile needs, from the ring of input
//               planes in LDS ("ring 0": 8-byte records, three per voxel; three channels: the DENSE records E/F/G below, 17
//               MFMAs per 16-voxel tile; five channels: four records, 27 MFMAs; other counts: conv0_split_f32.hip's order, 24);
//               a wave: two pairs of vertically adjacent tiles sharing their fragments + one single tile = 85 MFMAs; bias +
//               LeakyReLU, split into three bf16, stored into "ring 1" (5 planes: B reads 2s-3, 2s-2, 2s-1 meanwhile); voxels
//               outside the volume keep ring 1's zero;
//            B: output plane oz = s - 1: K = 27 taps x 16 channels = 13.5 k-blocks of 32 in four K quarters (taps 0..7 | 8..15 |
//               16..21 | 22..26: 4, 4, 3, 3 k-blocks); wave kq computes its quarter of both 16-voxel tiles and both cout tiles
//               (weights stationary in up to 96 registers; 96 | 72 MFMAs, 3 fragments = 6 ds_read_b64 per 12 MFMAs), owns
//               accumulator kq and takes the three foreign quarters of it from LDS one step later (sum in quarter order, bias,
//               LeakyReLU, one 16-byte store per lane).  The two 72-MFMA waves also stage the input: each splits the 55 items
//               (11 rows x 5 quads of 4 voxels, every channel) of ONE plane it requested a step ago into ring 0 and requests
//               the next (bounds-checked 8-byte buffer loads = the conv's zero padding).
// 181 | 157 MFMAs of 16 cycles per SIMD and step against 2 x 4 KB of HBM traffic: the pair is bound by issue, not by memory.
//
// LDS (<= 160 KB, Geo<NC>): ring 0 = 8 planes (five channels: 6) x 11 rows x [3 (4) record arrays][24 records of 8 bytes]; ring 1 =
// 5 planes x 3 splits x 9 rows x [4 channel quads][17 voxels] of 8 bytes, quads in the order 0,2,1,3 — a lane's two quads (8
// channels) are a fixed 272 bytes apart, the odd quad stride (17 chunks) and the row stride (72 chunks = 8 mod 16) make both the
// block-0 epilogue's ds_write_b64 (16 lanes = 16 consecutive voxels) and block 1's ds_read_b64 (32 lanes = 8 voxels x 2 rows x
// 2 channel halves) conflict-free; the exchange area of B's foreign accumulators (two step parities).
//
// Arithmetic: both stages are DIRECT convolutions whose products are exact; only the fp32 accumulation rounds (once per MFMA
// and partial sum), so against an fp64 convolution the pair is closer than the fmaf chain of the fp32 kernels
// (tests/test_gpu_conv01_fused.py).  Inf input -> NaN (Inf splits into Inf, NaN, NaN).
//
// Replaces (reference file:line): src/liftreg/layers/layers.py:365-369 (Conv3d + LeakyReLU), twice, as wired at
//   src/liftreg/models/LiftRegDeformSubspaceBackproj.py:95-100 for encoders[0] and encoders[1].
#include "lr_common.h"
#include <type_traits>
#include <utility>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int NTHR = 512;
constexpr int TY = 4, TX = 8;            // block-1 outputs of a column (rows x voxels)
constexpr int R1Y = 2 * TY + 1;          // 9 rows of block 0's output a plane of the column needs
constexpr int R0Y = R1Y + 2;             // 11 input rows
constexpr int NQ0 = 6;                   // aligned float4 quads of an input row: x = 2*ox0 - 4 .. 2*ox0 + 19 (record p = x - (2*ox0 - 4))
constexpr int SB0 = NQ0 * 4 * 8;         // 192 bytes: one split of a ring-0 row
// Staged: records 2 .. 21 only (the 19 a column reads are 2 .. 20) as FIVE quads that start 8 bytes off the 16-byte grid, each
// fetched as two 8-byte halves (a half lies entirely inside or outside a row: H is even) — 110 items per step = two waves of
// 55 lanes instead of three of 44 (records 0, 1, 22, 23 keep the zeros of the kernel's first fill)
constexpr int NQS = 5, QS0 = 2;
constexpr int QS1 = 17;                  // 8-byte chunks between the channel-quad runs of a ring-1 row (odd)
constexpr int RS1 = 72;                  // chunks of a ring-1 row (= 8 mod 16)
constexpr int SPB1 = R1Y * RS1 * 8;      // 5184 bytes: one split of a ring-1 plane
constexpr int PLB1 = 3 * SPB1;           // 15552
constexpr int NRING1 = 5;                // B reads planes 2s-3 .. 2s-1 while A writes 2s, 2s+1
constexpr int RING1 = NRING1 * PLB1;     // 77760
constexpr int SCR = 2 * 4 * 3 * 64 * 16; // 24576: [step parity][B wave = K quarter][the three accumulators (tile, cout tile) it does not own][lane] partial sums
// Ring 0 (the split input planes) depends on the channel count: three 8-byte record arrays per voxel for <= 4 channels (one per
// split; three channels: the dense records E, F, G), four for five channels (A1..A4, comment of dense5_records)
template <int NC>
struct Geo {
  static constexpr int NARR = NC == 5 ? 4 : 3;
  static constexpr int RB0 = NARR * SB0 + (NC == 5 ? 64 : 0);   // 576 | 832 = 64 mod 128: the lane pairs two window rows apart use opposite bank halves
  static constexpr int PLB0 = R0Y * RB0 + 192;    // 6528 | 9344 = 128 mod 256: neighbouring planes use opposite bank halves
  // NC <= 4: A reads planes 2s-1 .. 2s+2 (and, ahead of the barrier, its first fragments of step s+1: .. 2s+3) while B writes 2s+5,
  // 2s+6; five channels (LDS): A reads 2s-1 .. 2s+2 while B writes 2s+3, 2s+4
  static constexpr int NRING0 = NC == 5 ? 6 : 8;
  static constexpr int PA0 = NRING0 - 6;          // planes the staging runs ahead of what the next step reads
  static constexpr int NPRO0 = 4 + PA0;           // planes the unit prologue stages
  static constexpr int RING0 = NRING0 * PLB0;     // 52224 | 56064
  static constexpr int RING1_OFF = RING0;
  static constexpr int SCR_OFF = RING1_OFF + RING1;
  static constexpr int DUMP_OFF = SCR_OFF + SCR;  // where the threads without a staging item write
  static constexpr int DUMPB = 64 * 16 + (NARR - 1) * SB0 + 32;   // a 16-byte lane stride: the lanes without an item never write one address
  static constexpr int LDSB = DUMP_OFF + ((DUMPB + 255) / 256) * 256;
  static_assert(NPRO0 * R0Y * NQS <= NTHR, "one staging item per thread");
  static_assert((PLB0 & 255) == 128 && (RB0 & 127) == 64, "ring-0 bank geometry");
  static_assert(LDSB <= 160 * 1024, "LDS");
};
constexpr int NITEM = 2 * R0Y * NQS;     // 110 staging items of a step: (plane, row, x-quad), all channels
constexpr int NKB1 = 4;   // k-blocks of block 0 | of a K quarter of block 1 (taps 0..7 | 8..15 | 16..21 | 22..26)
constexpr unsigned OOR = 0x80000000u;
static_assert(NITEM == 2 * 55, "one staging item per thread: B waves 2 and 3, one plane each");
static_assert(4 * QS1 <= RS1 && (QS1 & 1) == 1 && (RS1 & 15) == 8, "ring-1 bank geometry");

struct FDims {
  int B, Cin, D, W, H, Do, Wo, Ho;   // D = planes of the input buffers (a z-slab), Do = output planes to compute
  int Dg, zoff, zlo;                  // global depth | global z of local plane 0 (= 2 x first output plane) | global z of buffer plane 0
  int nTx, nTy, nunits;
  int hps;                 // output layout: 1 = LR_LAYOUT_NDHWC_HPS, 0 = LR_LAYOUT_NDHWC
  long long bs0, bsr;      // elements between batch elements of channel 0 | of channels 1..Cin-1
  long long out_bs;        // output elements between batch elements
  float slope0, slope1;
  // training forward (SAVE): block 0's activation (B,D,W,H,16) fp32 channels-last (save_hps: rows parity-split) and its LeakyReLU
  // sign mask (B,D,W,H,4) uint8 (LR_LAYOUT_SIGN4) are ALSO written — what block 1's weight gradient and the fused
  // dgrad1 + wgrad0 kernel read (autograd.ConvPair01Fn)
  float* act0;
  unsigned char* mask0;
  int save_hps;
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
#ifndef LR_C01_PKF32
#define LR_C01_PKF32 1
#endif
#if LR_C01_PKF32
#define LR_C01_NO_DS_MERGE __attribute__((target("no-load-store-opt")))
#else
// LR_C01_PKF32=0: no packed fp32 vector instructions (the guide prices a v_pk_add_f32 / v_pk_mul_f32 beside MFMAs above the two scalar
// instructions it replaces, and hipcc packs the splits' subtractions and the LeakyReLU products) — measured 1.5 % SLOWER here (interleaved
// A/B, 5.62-5.66 vs 5.54-5.59 ms): the packed form stays
#define LR_C01_NO_DS_MERGE __attribute__((target("no-load-store-opt,no-packed-fp32-ops")))
#endif
#else
#define LR_C01_NO_DS_MERGE
#endif

#ifdef LR_C01_STAMPS
// Diagnostic build only (make stamps; tools/c01_stamps.py): per-phase cycle sums, [wave][phase]
__device__ unsigned long long g_lr_c01_stamps[8 * 8];
#define C01_STAMP(slot)                                            \
  do {                                                             \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
    stamp_acc[slot] += now_ - stamp_t;                             \
    stamp_t = now_;                                                \
  } while (0)
#else
#define C01_STAMP(slot) do {} while (0)
#endif

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), b, c, 0, 0, 0)
// timing-only ablation bits of a diagnostic build (`make abl`, tools/c01_abl.py; WRONG results): 1 B no MFMAs, 2 A idle, 4 A no
// epilogue, 8 B no fragment reads, 16 A no MFMAs, 32 A no fragment reads, 64 fragments read ONCE per column and kept in registers
// (every MFMA still runs on defined operands: the cost of the LDS reads alone), 128 B: no split work in the staging (loads and
// barriers stay)
#ifndef LR_C01_ABL
#define LR_C01_ABL 0
#endif
#define C01_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifndef LR_C01_BPRIO
#define LR_C01_BPRIO 1
#endif

// max of two fp32 quads as four bare v_max_f32: llvm.maxnum on an MFMA result first canonicalises it (v_max x, x — four more
// vector instructions per tile in the role whose vector issue is the kernel's limit); operands here are finite
__device__ __forceinline__ f32x4 max4(const f32x4& a, const f32x4& b) {
#ifdef LR_C01_MAXNUM   // A/B build: the compiler's maxnum
  return __builtin_elementwise_max(a, b);
#endif
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float t;
    asm("v_max_f32 %0, %1, %2" : "=v"(t) : "v"(a[i]), "v"(b[i]));
    r[i] = t;
  }
  return r;
}

// ---- DENSE block 0 (three input channels): K without padding.  A voxel's nine bf16 (3 channels x 3 splits d0, d1, d2) are laid
// out as three 8-byte records (in the place of the three per-split records, same LDS footprint)
//   E = [d0c0 d0c1 d0c2 d1c0]   F = [d1c1 d1c2 d2c0 d2c1]   G = [d2c2 d1c1 d1c2 0]
// and a tap contributes FIVE record slots to K instead of six (six products x (3 channels + 1 pad)):
//   E x [w0c0 w0c1 w0c2 w0c0]   (w0 d0 x3, w0 d1 c0)        E x [w1c0 w1c1 w1c2 w1c0]   (w1 d0 x3, w1 d1 c0)
//   E x [w2c0 w2c1 w2c2 0]      (w2 d0 x3)                  F x [w0c1 w0c2 w0c0 w0c1]   (w0 d1 c1 c2, w0 d2 c0 c1)
//   G x [w0c2 w1c1 w1c2 0]      (w0 d2 c2, w1 d1 c1 c2)
// = the same 18 exact products per tap.  27 taps x 5 = 135 record slots = 17 MFMAs of 8 slots per 16-voxel tile (24 before):
// per tap row ty, the 8 taps (tz, ty, tx) without (0, ty, 2) form one data fragment per record kind, used with three (E) or
// one (F, G) weight fragments = 15 MFMAs whose fragments two vertically adjacent tiles SHARE (input row iy is tap row iy of
// the upper and iy - 1 of the lower tile); the three taps (0, ty, 2) x 5 slots fill two more MFMAs.  A tile's chain runs from
// the small products to the large: G, E x w2, F, E x w1, the two leftover MFMAs, E x w0.
// The records of one voxel; channel 2 of TWO x-neighbours is split in one go (p2[s] = split s of voxel 0 | of voxel 1 << 16: half
// the conversions of splitting it beside a zero) and HI selects the voxel.  v_perm_b32: selector bytes 0..3 pick S1's bytes, 4..7 S0's.
template <int HI>
__device__ __forceinline__ void dense_records_c2pair(const unsigned (&p01)[3], const unsigned (&p2)[3], unsigned (&rec)[3][2]) {
  rec[0][0] = p01[0];
  rec[0][1] = __builtin_amdgcn_perm(p01[1], p2[0], HI ? 0x05040302u : 0x05040100u);   // d0c2 | d1c0
  rec[1][0] = __builtin_amdgcn_perm(p2[1], p01[1], HI ? 0x07060302u : 0x05040302u);   // d1c1 | d1c2
  rec[1][1] = p01[2];
  rec[2][0] = __builtin_amdgcn_perm(p01[1], p2[2], HI ? 0x07060302u : 0x07060100u);   // d2c2 | d1c1
  rec[2][1] = HI ? p2[1] >> 16 : p2[1] & 0xffffu;
}
// FIVE input channels (the reference's shipped configuration: four views): the 15 bf16 of a voxel as FOUR records
//   A1 = [d0c0 d0c1 d0c2 d0c3]   A2 = [d0c4 d1c0 d1c1 d1c2]   A3 = [d1c3 d1c4 d2c0 d2c1]   A4 = [d2c2 d2c3 d2c4 d0c4]
// and EIGHT record slots per tap, 30 exact products in 32 element slots:
//   A1 x w0[c0..c3], A1 x w1[c0..c3], A1 x w2[c0..c3]          A2 x [w0c4 w0c0 w0c1 w0c2], A2 x [w1c4 w1c0 w1c1 w1c2]
//   A3 x [w0c3 w0c4 w0c0 w0c1], A3 x [w1c3 w1c4 0 0]           A4 x [w0c2 w0c3 w0c4 w2c4]
// 27 taps x 8 = 216 slots = 27 MFMAs per 16-voxel tile with no padding at all: per tap row ty the 8 taps without (0, ty, 2) form one
// fragment per record kind (8 MFMAs), and the leftover tap (0, ty, 2) fills exactly one more (its 8 slots = the 8 patterns above).
// Every fragment belongs to ONE tap row, so two vertically adjacent tiles share all of them.  Chain, small products first:
// A1 w2, A4, A3 w1, A3 w0, A2 w1, A1 w1, leftover, A2 w0, A1 w0 (three tap rows each).
__device__ __forceinline__ void dense5_records(const unsigned (&p01)[3], const unsigned (&p23)[3], const unsigned (&d4)[3], unsigned (&rec)[4][2]) {
  rec[0][0] = p01[0];                          // (d4[s]: split s of channel 4 in the low half, high half 0)
  rec[0][1] = p23[0];
  rec[1][0] = d4[0] | (p01[1] << 16);
  rec[1][1] = __builtin_amdgcn_alignbit(p23[1], p01[1], 16);
  rec[2][0] = (p23[1] >> 16) | (d4[1] << 16);
  rec[2][1] = p01[2];
  rec[3][0] = p23[2];
  rec[3][1] = d4[2] | (d4[0] << 16);
}

// The chain of a dense block-0 tile: which fragment each MFMA reads, the order the fragments are first used in, the loads.
//   fragment n of a PAIR of vertically adjacent tiles (upper r = 0, lower r = 1; the single tile has the first NFS): the fragments
//   of input rows 0..2 kind by kind in the order of their first use, then those of input row 3 (lower tile only).
template <int NC> struct Chain;
template <> struct Chain<3> {   // kinds: 0 G, 1 E, 2 F (arrays 2, 0, 1), 3 = the two leftover fragments of a tile (not shared)
  static constexpr int NMF = 17, NFP = 16, NFS = 11;
  static constexpr bool is_left(int n) { return (n >= 9 && n < 11) || n >= 14; }
  static constexpr int arr(int n) { const int kind = n < 9 ? n / 3 : n - 11; return kind == 0 ? 2 : kind == 1 ? 0 : 1; }
  static constexpr int iy(int n) { return n < 9 ? n % 3 : 3; }
  static constexpr int left_w(int n) { return n < 11 ? n - 9 : n - 14; }          // which of the lane's leftover bases
  static constexpr int left_row(int n) { return n < 11 ? 0 : 1; }                 // rows below the pair's first
  static constexpr int frag(int k, int r) {
    if (k >= 12 && k < 14) return (r == 0 ? 9 : 14) + (k - 12);
    const int kind = k < 3 ? 0 : k < 6 ? 1 : k < 9 ? 2 : 1;     // G E F E (L L) E
    const int i = r + (k < 12 ? k % 3 : k - 14);
    return i < 3 ? kind * 3 + i : 11 + kind;
  }
  static constexpr int need(int n) {   // the MFMA slot (relative to the pair's first) of the fragment's first use
    if (n < 9) return n;
    if (n < 11) return 12 + (n - 9);
    if (n < 14) return NMF + 2 + 3 * (n - 11);
    return NMF + 12 + (n - 14);
  }
  static constexpr int EPI_AT = NMF - 1;   // the MFMA slot of a tile behind which the epilogue of the tile before runs
};
template <> struct Chain<5> {   // load kinds: 0 A1, 1 A4, 2 A3, 3 A2 (arrays 0, 3, 2, 1), 4 = the leftover fragment of a tap row (shared like the others)
  static constexpr int NMF = 27, NFP = 20, NFS = 15;
  static constexpr int lk(int n) { return n < 15 ? n / 3 : n - 15; }
  static constexpr bool is_left(int n) { return lk(n) == 4; }
  static constexpr int arr(int n) { return lk(n) == 0 ? 0 : lk(n) == 1 ? 3 : lk(n) == 2 ? 2 : 1; }
  static constexpr int iy(int n) { return n < 15 ? n % 3 : 3; }
  static constexpr int left_w(int) { return 0; }
  static constexpr int left_row(int n) { return iy(n); }
  static constexpr int phase_kind(int p) { return p == 0 ? 0 : p == 1 ? 1 : p == 2 ? 2 : p == 3 ? 2 : p == 4 ? 3 : p == 5 ? 0 : p == 6 ? 4 : p == 7 ? 3 : 0; }
  static constexpr int frag(int k, int r) {
    const int kind = phase_kind(k / 3), i = r + k % 3;
    return i < 3 ? kind * 3 + i : 15 + kind;
  }
  static constexpr int first_phase(int kind) { return kind == 0 ? 0 : kind == 1 ? 1 : kind == 2 ? 2 : kind == 3 ? 4 : 6; }
  static constexpr int need(int n) { return n < 15 ? first_phase(lk(n)) * 3 + iy(n) : NMF + first_phase(lk(n)) * 3 + 2; }
  static constexpr int EPI_AT = NMF - 1;
};
template <int NC, bool DENSE> constexpr int block0_fragments() { if constexpr (DENSE) return Chain<NC>::NMF; else return 12; }
// Fragment loads of a step (MFMA slot g = NMF t + k; tiles t = 0..4: pair 0 upper / lower, pair 1 upper / lower, the single
// tile): behind every LR_C01_LOAD_PERIOD-th slot a burst of up to LR_C01_LOAD_BURST fragments — the next ones (sets 0, 1 = the
// pairs, 2 = the single tile, in the order above) whose first use is <= DENSE_AHEAD slots away (lgkmcnt counts 15 operations;
// more simply wait).  The first DENSE_PRE fragments of the step are requested in front (ahead of the barrier when ring 0 is
// staged a step early).  Bursts, because a gap between two MFMAs of a chain that holds anything costs a toll (NOTES_r05 §6).
#ifndef LR_C01_DENSE_AHEAD
#define LR_C01_DENSE_AHEAD 12
#endif
constexpr int DENSE_PRE = 9, DENSE_AHEAD = LR_C01_DENSE_AHEAD;   // (9: every fragment the first tile's first nine MFMAs open; 6: +0.7 %, 4: +0.5 %)
#ifndef LR_C01_LOAD_PERIOD
#define LR_C01_LOAD_PERIOD 4   // fragment loads only behind every PERIOD-th MFMA slot, up to LR_C01_LOAD_BURST fragments there (one per slot: +1 %)
#endif
#ifndef LR_C01_LOAD_BURST
#define LR_C01_LOAD_BURST 4
#endif
template <int NC>
constexpr int dense_next(int pos) {   // the fragment after pos = set * 32 + n in load order (set 3 = none left)
  using C = Chain<NC>;
  int set = pos / 32, n = pos % 32 + 1;
  if (n == (set == 2 ? C::NFS : C::NFP)) { n = 0; ++set; }
  return set * 32 + n;
}
template <int NC>
constexpr int dense_nth(int pos, int i) { for (int k = 0; k < i; ++k) pos = dense_next<NC>(pos); return pos; }
template <int NC>
constexpr int dense_load_at(int g) {   // count * 1024 + (set * 32 + n of the first): the fragment loads issued behind MFMA slot g
  using C = Chain<NC>;
  int pos = DENSE_PRE;
  for (int s = 0; s <= g; ++s) {
    int cnt = 0, first = pos;
    if (s % LR_C01_LOAD_PERIOD == 0)
      while (pos / 32 <= 2 && (pos / 32) * 2 * C::NMF + C::need(pos % 32) - s <= DENSE_AHEAD && cnt < LR_C01_LOAD_BURST) { pos = dense_next<NC>(pos); ++cnt; }
    if (s == g) return cnt * 1024 + first;
  }
  return 0;
}
template <int NC>
constexpr bool dense_schedule_ok() {   // every fragment is requested before the MFMA that first reads it, and all are requested
  using C = Chain<NC>;
  int pos = DENSE_PRE;
  for (int s = 0; s < 5 * C::NMF; ++s) {
    const int b = dense_load_at<NC>(s);
    int q = b % 1024;
    for (int i = 0; i < b / 1024; ++i) {
      if (q != pos) return false;
      if (s >= (q / 32) * 2 * C::NMF + C::need(q % 32)) return false;
      q = dense_next<NC>(q);
      pos = q;
    }
  }
  return pos / 32 == 3;
}
static_assert(dense_schedule_ok<3>() && dense_schedule_ok<5>(), "dense fragment schedule");
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int NC, bool SAVE, bool DENSE>
// LR_C01_NUM_VGPR (build flag): cap the kernel's registers per lane (the attribute counts half of the unified file's total)
#ifdef LR_C01_NUM_VGPR
#define LR_C01_VGPR_CAP __attribute__((amdgpu_num_vgpr(LR_C01_NUM_VGPR / 2)))
#else
#define LR_C01_VGPR_CAP
#endif
__global__ LR_C01_VGPR_CAP __launch_bounds__(NTHR) LR_C01_NO_DS_MERGE void conv01_fused_kernel(
    const float* __restrict__ in0, const float* __restrict__ in_rest, const u32x4* __restrict__ wp0,
    const u32x4* __restrict__ wp1, const float* __restrict__ bias0, const float* __restrict__ bias1,
    float* __restrict__ out, FDims d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  using G = Geo<NC>;
  constexpr int RB0 = G::RB0, PLB0 = G::PLB0, NRING0 = G::NRING0, PA0 = G::PA0, NPRO0 = G::NPRO0, RING1_OFF = G::RING1_OFF,
                SCR_OFF = G::SCR_OFF, DUMP_OFF = G::DUMP_OFF, LDSB = G::LDSB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_a = wave < 4;                 // block-0 producer | block-1 consumer
  const int wq = wave & 3;                    // index inside the role
  const int col = lane & 15, lq = lane >> 4;
  const int dD = d.D, dW = d.W, dH = d.H;
  const unsigned V4 = (unsigned)dD * dW * dH * 4u;   // bytes of one channel volume of the buffers (3 of them < 2^31: launcher)
  // z bookkeeping: the kernel counts planes LOCALLY (output plane 0 = the first one it computes, input / block-0 plane z_l
  // = global z - zoff); a plane exists iff 0 <= z_l + zoff < Dg and then sits at plane z_l + zoff - zlo of the buffers
  const int Dg = d.Dg, zoff = d.zoff, zrel = d.zoff - d.zlo;
  // a slab that does not start at the top of the volume: block-0 plane -1 (global zoff - 1) exists — one step s = -1 in front
  // computes it (A: planes -2, -1, the first into a slot nobody reads; B: staging only)
  const int s0 = __builtin_amdgcn_readfirstlane(zoff > 0 ? -1 : 0);
  // ring-0 slots are counted from the unit's first step (plane z sits in slot (z + 1 + zsh) mod NRING0): the ring state of step s0 is 0
  const int zsh = -2 * s0;

  // zero everything once: a "weight 0" operand slot multiplies whatever lies behind a row / plane and needs finite numbers
  for (int o = tid * 16; o < LDSB; o += NTHR * 16) *reinterpret_cast<u32x4*>(lds + o) = (u32x4){0u, 0u, 0u, 0u};

  // ---- stationary operands: ONE register array for both roles — A: the 12 fragments [k-block][split] of block 0 (DENSE: its 17),
  // B (cout tile c, K half kh): the 21 fragments [k-block][split] of its half of block 1
  const int kq = wq;                                        // B: K quarter; it also owns accumulator kq = (tile kq >> 1, cout tile kq & 1)
  const int tap0 = kq == 0 ? 0 : kq == 1 ? 8 : kq == 2 ? 16 : 22, tap1 = kq == 0 ? 8 : kq == 1 ? 16 : kq == 2 ? 22 : 27;
  const int nkb = __builtin_amdgcn_readfirstlane((tap1 - tap0 + 1) >> 1);   // 4, 4, 3, 3
  const int bc = kq & 1;
  constexpr int NA = block0_fragments<NC, DENSE>(), NWR = NA > 24 ? NA : 24;
  u32x4 wr[NWR];   // A: [k-block][split] (DENSE: the chain's fragments); B: [k-block][cout tile][split]
  {
    const u32x4* src = is_a ? wp0 : wp1 + (size_t)kq * 24 * 64;
#pragma unroll
    for (int i = 0; i < NWR; ++i) wr[i] = (i >= (is_a ? NA : 24)) ? (u32x4){0u, 0u, 0u, 0u} : src[i * 64 + lane];
  }
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};   // A: block 0's bias of this lane's channel quad; B: block 1's of cout tile c
  {
    const float* bsrc = is_a ? bias0 : bias1;
    if (bsrc) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = bsrc[(is_a ? 0 : bc * 16) + lq * 4 + r];
    }
  }

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

  if (is_a) {
  for (int uid = first; uid < end; uid += stride) {
    const int utx = uid % d.nTx, uty = (uid / d.nTx) % d.nTy, ub = __builtin_amdgcn_readfirstlane(uid / d.nTx / d.nTy);
    const int oy0 = __builtin_amdgcn_readfirstlane(uty * TY), ox0 = __builtin_amdgcn_readfirstlane(utx * TX);
    const int Y1 = 2 * oy0 - 1, X1 = 2 * ox0 - 1;       // volume coordinates of region-1 (0, 0)
    const int Y0 = 2 * oy0 - 2, X0a = 2 * ox0 - 4;      // volume coordinates of ring-0 row 0 / record 0
    const __amdgpu_buffer_rsrc_t r0 = make_rsrc(in0 + (int64_t)ub * d.bs0, V4);
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(in_rest + (int64_t)ub * d.bsr, NC > 1 ? (unsigned)(NC - 1) * V4 : 0u);

    // one staging item = input plane zi, ring-0 row irow, x-quad iq, every channel: request -> registers ...
    // (element access on the ext-vector result of the buffer-load builtin is narrowed by hipcc to a one-dword load with the
    // other elements undefined — DESIGN.md 6a: the quads are viewed through HIP's uint4, a struct, where they are used)
    auto issue_item = [&](bool live, int zi, int irow, int iq, u32x4 (&L)[NC]) __attribute__((always_inline)) {
      const int yi = Y0 + irow, xi = X0a + QS0 + 4 * iq;
      const int zg = zi + zoff;
      const int ok = (int)live & (int)(zg >= 0) & (int)(zg < Dg) & (int)(yi >= 0) & (int)(yi < dW);
      const int ok_lo = ok & (int)(xi >= 0) & (int)(xi < dH), ok_hi = ok & (int)(xi + 2 >= 0) & (int)(xi + 2 < dH);
      const unsigned dead_lo = ((unsigned)ok_lo - 1u) & OOR, dead_hi = ((unsigned)ok_hi - 1u) & OOR;   // dead half: bit 31 -> outside the resource -> 0
      const unsigned voff = (unsigned)((((zi + zrel) * dW + yi) * dH + xi) * 4);
      auto halves = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned vo) __attribute__((always_inline)) -> u32x4 {
        const uint2 a = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(vo | dead_lo), 0, 0));
        const uint2 b = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)((vo + 8u) | dead_hi), 0, 0));
        return (u32x4){a.x, a.y, b.x, b.y};
      };
      L[0] = halves(r0, voff);
#pragma unroll
      for (int c = 1; c < NC; ++c) L[c] = halves(rr, voff + (unsigned)(c - 1) * V4);
    };
    // ... -> split -> ring 0 (plane z sits in slot (z + 1) mod 6); a thread without an item writes zeros into the dump area
    auto write_item = [&](bool live, int zi, int irow, int iq, const u32x4 (&L)[NC]) __attribute__((always_inline)) {
      const int slot = (zi + 1 + zsh + NRING0) % NRING0;
      unsigned char* const base = lds + (live ? slot * PLB0 + irow * RB0 + QS0 * 8 + iq * 32 : DUMP_OFF + lane * 16);
      if constexpr (NC == 5) {   // four records per voxel (dense5_records); channel 4 is split in voxel pairs
        auto val = [&](int c, int j) __attribute__((always_inline)) -> float {
          const uint4 q = __builtin_bit_cast(uint4, L[c]);
          return __builtin_bit_cast(float, j == 0 ? q.x : j == 1 ? q.y : j == 2 ? q.z : q.w);
        };
        unsigned rec5[4][4][2];   // [record][voxel][half]
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          unsigned p4[3];
          split3(val(4, 2 * jp), val(4, 2 * jp + 1), p4);
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * jp + jj;
            unsigned p01[3], p23[3], d4[3], r4[4][2];
            split3(val(0, j), val(1, j), p01);
            split3(val(2, j), val(3, j), p23);
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) d4[sp] = jj ? (p4[sp] >> 16) : (p4[sp] & 0xffffu);
            dense5_records(p01, p23, d4, r4);
#pragma unroll
            for (int a = 0; a < 4; ++a) { rec5[a][j][0] = r4[a][0]; rec5[a][j][1] = r4[a][1]; }
          }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          *reinterpret_cast<u32x4*>(base + a * SB0) = (u32x4){rec5[a][0][0], rec5[a][0][1], rec5[a][1][0], rec5[a][1][1]};
          *reinterpret_cast<u32x4*>(base + a * SB0 + 16) = (u32x4){rec5[a][2][0], rec5[a][2][1], rec5[a][3][0], rec5[a][3][1]};
        }
        return;
      }
      float v[4][4];   // [channel][voxel]
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint4 q = __builtin_bit_cast(uint4, L[c < NC ? c : 0]);
        v[c][0] = c < NC ? __builtin_bit_cast(float, q.x) : 0.0f; v[c][1] = c < NC ? __builtin_bit_cast(float, q.y) : 0.0f;
        v[c][2] = c < NC ? __builtin_bit_cast(float, q.z) : 0.0f; v[c][3] = c < NC ? __builtin_bit_cast(float, q.w) : 0.0f;
      }
      unsigned rec[3][4][2];   // [split][voxel][channels 01 | 23]   (DENSE: [record E, F, G][voxel][half])
      if constexpr (DENSE) {   // (three channels)
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          unsigned p2[3], p01[3], r3[3][2];
          split3(v[2][2 * jp], v[2][2 * jp + 1], p2);
          split3(v[0][2 * jp], v[1][2 * jp], p01);
          dense_records_c2pair<0>(p01, p2, r3);
#pragma unroll
          for (int s = 0; s < 3; ++s) { rec[s][2 * jp][0] = r3[s][0]; rec[s][2 * jp][1] = r3[s][1]; }
          split3(v[0][2 * jp + 1], v[1][2 * jp + 1], p01);
          dense_records_c2pair<1>(p01, p2, r3);
#pragma unroll
          for (int s = 0; s < 3; ++s) { rec[s][2 * jp + 1][0] = r3[s][0]; rec[s][2 * jp + 1][1] = r3[s][1]; }
        }
      } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned p01[3], p23[3] = {0u, 0u, 0u};
        split3(v[0][j], v[1][j], p01);
        if (NC > 2) split3(v[2][j], v[3][j], p23);
#pragma unroll
        for (int s = 0; s < 3; ++s) { rec[s][j][0] = p01[s]; rec[s][j][1] = p23[s]; }
      }
      }
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        *reinterpret_cast<u32x4*>(base + s * SB0) = (u32x4){rec[s][0][0], rec[s][0][1], rec[s][1][0], rec[s][1][1]};
        *reinterpret_cast<u32x4*>(base + s * SB0 + 16) = (u32x4){rec[s][2][0], rec[s][2][1], rec[s][3][0], rec[s][3][1]};
      }
    };

    // ---- unit prologue: ring 1 <- 0 (every voxel of the column outside the volume), ring 0 <- planes 2 s0 - 1 .. 2 s0 + 4
    // (NPRO0 x 66 items); the B threads also request planes 2 s0 + 5, + 6 (their items of the first step)
    // staging items of a step: B waves 2 and 3 (the K quarters with three k-blocks: 72 MFMAs against 96) take one plane = 55 each
    const int bi = (wq - 2) * 55 + lane;
    const bool item_live = !is_a && wq >= 2 && lane < 55;
    const int ipl = item_live ? bi / (R0Y * NQS) : 0, irow = item_live ? (bi % (R0Y * NQS)) / NQS : 0, iq = item_live ? bi % NQS : 0;
    u32x4 ldn[NC];
    __syncthreads();   // (the zero fill of the whole LDS | every read of the unit before)
    {
      const bool pl_live = tid < NPRO0 * R0Y * NQS;
      const int pz = pl_live ? tid / (R0Y * NQS) - 1 + 2 * s0 : 0, prow = pl_live ? (tid % (R0Y * NQS)) / NQS : 0, pq = pl_live ? tid % NQS : 0;
      u32x4 lp[NC];
      issue_item(pl_live, pz, prow, pq, lp);
      issue_item(item_live, 2 * s0 + 3 + PA0 + ipl, irow, iq, ldn);
      for (int o = tid * 16; o < RING1; o += NTHR * 16) *reinterpret_cast<u32x4*>(lds + RING1_OFF + o) = (u32x4){0u, 0u, 0u, 0u};
      write_item(pl_live, pz, prow, pq, lp);
    }
    __syncthreads();

      // =========================================================== A: block 0 ===========================================
      // Lane group lq supplies two taps a fixed distance apart (conv0_split_f32.hip):
      //   k-block ty = 0,1,2: lq = 0..2: (tz = lq, ty, tx = 0) | (tz = lq, ty, tx = 1);  lq = 3: (tz = 0, ty, tx = 2) | weight 0
      //   k-block 3:          lq = 0: (1,0,2) | (1,1,2);  lq = 1: (1,2,2) | weight 0;  lq = 2: (2,0,2) | (2,1,2);  lq = 3: (2,2,2) | weight 0
      // region voxel (ry, rx) of block 0's output = volume (2*oy0 - 1 + ry, 2*ox0 - 1 + rx); its tap (ty, tx) = ring-0 row
      // ry + ty, record rx + tx + 2.  Units of wave w: the row pair (rows 2w, 2w+1; voxels 0..15) of plane 0 and of plane 1
      // (the pair shares its window rows: input row iy is tap row iy of output row 0 and iy - 1 of output row 1), then ONE
      // single tile: row 8 of plane 0 (w = 0) | of plane 1 (w = 1), column 16 (rows 0..8) of plane 0 (w = 2) | plane 1 (w = 3).
      const int dlF = lq == 1 ? 1 : lq == 2 ? 2 : 0, dlG = lq < 2 ? 1 : 2;       // plane of the lane's taps (tz)
      const unsigned laneF = (unsigned)((col + 2 + (lq == 3 ? 2 : 0)) * 8);
      const unsigned laneG = (unsigned)((lq & 1) * 2 * RB0 + (col + 4) * 8);
      const int spl = wq & 1, scol = wq >> 1;                                    // the single tile: plane | row 8 or column 16
#ifndef LR_C01_COLCLAMP
#define LR_C01_COLCLAMP 1
#endif
      // (column tile: lanes 9..15 have no voxel; with LR_C01_COLCLAMP they re-read row 8 — same addresses as lane 8, no extra bank conflicts)
      const int sry = scol ? (LR_C01_COLCLAMP ? (col < 8 ? col : 8) : col) : 8, srx = scol ? 16 : col;
      const unsigned laneFs = (unsigned)(sry * RB0 + (srx + 2 + (lq == 3 ? 2 : 0)) * 8);
      const unsigned laneGs = (unsigned)((sry + (lq & 1) * 2) * RB0 + (srx + 4) * 8);
      const unsigned qpos = (unsigned)(((lq & 1) << 1) | (lq >> 1));   // ring-1 position of channel quad lq: order 0,2,1,3
      // ring-1 byte offsets of this lane's stores inside a plane; a voxel OUTSIDE the volume (or a lane without a voxel: rows
      // 9..15 of the column tile) stores into the 4 pad chunks behind row 0 of slot 0 (the same place in each split's plane)
      const int dump1 = RING1_OFF + (4 * QS1 + (lane & 3)) * 8;
      const int ry0 = 2 * wq;
      const bool xok = (unsigned)(X1 + col) < (unsigned)dH;
      const bool ok_p0 = xok && (unsigned)(Y1 + ry0) < (unsigned)dW, ok_p1 = xok && (unsigned)(Y1 + ry0 + 1) < (unsigned)dW;
      const bool ok_s = (unsigned)(Y1 + sry) < (unsigned)dW && (unsigned)(X1 + srx) < (unsigned)dH && (!scol || col < R1Y);
      const int st_p0 = RING1_OFF + (ry0 * RS1 + (int)qpos * QS1 + col) * 8, st_p1 = st_p0 + RS1 * 8;
      const int st_s = RING1_OFF + (sry * RS1 + (int)qpos * QS1 + srx) * 8;
      // DENSE: ring-0 byte offsets of this lane's two fragment halves inside (plane, tile row 0) — pair tiles | the single tile — and
      // of its leftover halves [fragment w][half]: three channels: slot 8 w + 2 lq + h = tap (0, slot / 5, 2), kind slot % 5 = E x w0,
      // E x w1, E x w2, F, G (slot 15: weight 0); five channels: slot 2 lq + h = the record array 0 0 0 1 1 2 2 3 of tap (0, ty, 2)
      // The 8 taps (tz, tx) of a tap row in a main fragment — first halves: lq 0..2: (lq, 0), lq 3: (2, 1); second halves: (0, 1),
      // (1, 1), (2, 2), (1, 2); the leftover tap is (0, ty, 2).  In both ds_read_b64 of a fragment the lane groups that share an LDS
      // cycle (lq 0 with 1, lq 2 with 3) then read either complementary bank halves (planes of opposite parity, same record) or
      // the same run of records shifted by one: no bank conflict (the (tz = lq; tx 0 | 1) + (0, 2) | (1, 2) order of round 4
      // cost an extra LDS cycle on every second read).
      const unsigned dnA = (unsigned)((col + 2 + (lq == 3 ? 1 : 0)) * 8), dnB = (unsigned)((col + (lq < 2 ? 3 : 4)) * 8);
      const unsigned dnAs = (unsigned)(sry * RB0 + (srx + 2 + (lq == 3 ? 1 : 0)) * 8), dnBs = (unsigned)(sry * RB0 + (srx + (lq < 2 ? 3 : 4)) * 8);
      unsigned dnL[2][2], dnLs[2][2];
#pragma unroll
      for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          int o;
          if constexpr (NC == 5) {
            const int sidx = lq * 2 + h;
            o = (sidx < 3 ? 0 : sidx < 5 ? 1 : sidx < 7 ? 2 : 3) * SB0;
          } else {
            const int sidx = w * 8 + lq * 2 + h, sc = sidx > 14 ? 14 : sidx, lty = sc / 5, lk = sc % 5;
            o = lty * RB0 + (lk < 3 ? 0 : lk - 2) * SB0;
          }
          dnL[w][h] = (unsigned)(o + (col + 4) * 8);
          dnLs[w][h] = (unsigned)(o + sry * RB0 + (srx + 4) * 8);
        }
      // SAVE: byte offsets inside a plane of this lane's three kinds of tiles in the activation / the mask (bit 31: not this
      // column's voxel — region row 0 and column 0 are the neighbours' — or outside the volume); the plane goes in soffset
      unsigned svo_p0 = OOR, svo_p1 = OOR, svo_s = OOR, mko_p0 = OOR, mko_p1 = OOR, mko_s = OOR;
      __amdgpu_buffer_rsrc_t rs_act = make_rsrc(out, 0u), rs_msk = rs_act;
      if constexpr (SAVE) {
        auto sv = [&](bool own, int y, int x, unsigned& so, unsigned& mo) __attribute__((always_inline)) {
          const int hp = d.save_hps ? (x & 1) * (dH >> 1) + (x >> 1) : x;
          so = own ? (unsigned)((y * dH + hp) * 64 + lq * 16) : OOR;
          mo = own ? (unsigned)((y * dH + x) * 4 + lq) : OOR;     // every lane stores the byte of its own channel quad
        };
        sv(ok_p0 && ry0 >= 1 && col >= 1, Y1 + ry0, X1 + col, svo_p0, mko_p0);
        sv(ok_p1 && col >= 1, Y1 + ry0 + 1, X1 + col, svo_p1, mko_p1);
        sv(ok_s && sry >= 1 && srx >= 1, Y1 + sry, X1 + srx, svo_s, mko_s);
        rs_act = make_rsrc(d.act0 + (int64_t)ub * dD * dW * dH * 16, (unsigned)dD * dW * dH * 64u);
        rs_msk = make_rsrc(d.mask0 + (int64_t)ub * dD * dW * dH * 4, (unsigned)dD * dW * dH * 4u);
      }

      constexpr bool a_mma_g = !(LR_C01_ABL & 16), a_ld_g = !(LR_C01_ABL & (32 | 64)), a_epi_g = !(LR_C01_ABL & 4);
      struct PairFrags { bf16x8 F[4][3], G[2][3]; };
      struct OneFrags { bf16x8 F[3][3], G[3]; };
      struct Acc2 { f32x4 v[2]; };   // block 0 (K = 81): one fp32 chain per tile, small products first inside each k-block
      struct Base { unsigned pF, pG, pG1; };
      // fragment n (0..17) of a pair, in the order the MFMAs use them: F0 F3 F1 F2 G0 G1, three splits each
      auto load_pair_n = [&](int n, const Base& b, PairFrags& q) __attribute__((always_inline)) {
        if (!a_ld_g) return;
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
        if (a_mma_g) a.v[r] = MFMA(wr[kb * 3 + wt], f, a.v[r]);
      };
      auto load_one_n = [&](int n, const Base& b, OneFrags& q) __attribute__((always_inline)) {   // n = 0..11: F0 F1 F2 G
        if (!a_ld_g) return;
        const int grp = n / 3, sp = n % 3;
        if (grp < 3) q.F[grp][sp] = frag(b.pF, grp * RB0 + sp * SB0, 8);
        else q.G[sp] = frag(b.pG, sp * SB0, RB0);
      };
      auto mma_one_n = [&](int I, const OneFrags& q, Acc2& a) __attribute__((always_inline)) {   // I = 0..23: two chains (even | odd k-blocks)
        const int kb = I / 6, j = I % 6;
        const int wt = j == 0 ? 1 : j == 1 ? 2 : j == 2 ? 0 : j == 3 ? 1 : 0;
        const int fs = j == 0 ? 1 : j == 1 ? 0 : j == 2 ? 2 : j == 3 ? 0 : j == 4 ? 1 : 0;
        const bf16x8& f = kb == 3 ? q.G[fs] : q.F[kb][fs];
        if (a_mma_g) a.v[kb & 1] = MFMA(wr[kb * 3 + wt], f, a.v[kb & 1]);
      };
      // the epilogue of one block-0 tile in 7 slices (the dense path runs all seven in one MFMA gap): LeakyReLU, the two three-way
      // splits, the three stores into ring 1
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
      // (SAVE: so / mo = the tile's activation / mask offsets, zb = its plane (scalar); a plane below the volume never comes here
      // with live offsets: the caller passes OOR)
      auto epi_slice = [&](int sl, Epi& E, const f32x4& acc, int addr, unsigned so = OOR, unsigned mo = OOR, int zb = 0) __attribute__((always_inline)) {
        if (!a_epi_g) {   // (ablation: the chain stays alive through one store)
          if (sl == 6) *reinterpret_cast<f32x4*>(lds + DUMP_OFF + lane * 16) = acc;
          return;
        }
        if (sl == 0) { E.x = acc; E.y = E.x * d.slope0; }
        else if (sl == 1) {
          E.x = max4(E.x, E.y);   // = LeakyReLU for 0 <= slope <= 1 (launcher)
#ifndef LR_C01_SAVE_PARTS
#define LR_C01_SAVE_PARTS 3   // timing builds: 1 = activation only, 2 = mask only
#endif
          if constexpr (SAVE && (LR_C01_SAVE_PARTS & 1))
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, E.x), rs_act, (int)so, zb * dW * dH * 64, 0);
        }
        else if (SAVE && (LR_C01_SAVE_PARTS & 2) && sl == 3) {   // (the slice's own work follows below)
          // the four channel quads of a voxel sit in the four 16-lane rows: row/half swaps bring their sign nibbles into row 0,
          // whose lanes store one dword per voxel (conv3d.hip, MASK)
          // x > 0  <=>  its bit pattern as a signed integer is > 0 (+0, -0 and every negative are <= 0): one v_med3_i32 per value
          // (inline asm: hipcc turns the min/max form back into v_cmp + v_cndmask with s_nop hazards between them)
          auto pos = [](float v) __attribute__((always_inline)) -> unsigned {
            unsigned r;
            asm("v_med3_i32 %0, %1, 0, 1" : "=v"(r) : "v"(v));
            return r;
          };
          // the four channel quads of a voxel sit in the four 16-lane rows: each lane stores ITS byte (a wave's 64 bytes are 16
          // voxels x 4 consecutive bytes; round 4 gathered the nibbles into row 0 with three permlane swaps and stored dwords)
          const unsigned x = pos(E.x[0]) | (pos(E.x[1]) << 1) | (pos(E.x[2]) << 2) | (pos(E.x[3]) << 3);
          __builtin_amdgcn_raw_buffer_store_b8((unsigned char)x, rs_msk, (int)mo, zb * dW * dH * 4, 0);
          const L12 t = lvl12(E.r); E.s1 = (u32x2){t.p1, 0u}; E.s2 = (u32x2){t.p2, 0u};
        }
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

#if (LR_C01_ABL & 64)
      PairFrags fa, fb;
      OneFrags fs;
      {
        const unsigned pb = laneF, gb = laneG;
#pragma unroll
        for (int iy = 0; iy < 4; ++iy)
#pragma unroll
          for (int sp = 0; sp < 3; ++sp) { fa.F[iy][sp] = frag(pb, iy * RB0 + sp * SB0, 8); fb.F[iy][sp] = frag(pb + PLB0, iy * RB0 + sp * SB0, 8); }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int sp = 0; sp < 3; ++sp) { fa.G[r][sp] = frag(gb + r * RB0, sp * SB0, RB0); fb.G[r][sp] = frag(gb + PLB0 + r * RB0, sp * SB0, RB0); }
#pragma unroll
        for (int iy = 0; iy < 3; ++iy)
#pragma unroll
          for (int sp = 0; sp < 3; ++sp) fs.F[iy][sp] = frag(pb + 2 * PLB0, iy * RB0 + sp * SB0, 8);
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) fs.G[sp] = frag(gb + 2 * PLB0, sp * SB0, RB0);
      }
#endif
      int e6 = 0, m5 = __builtin_amdgcn_readfirstlane((2 * s0 + NRING1) % NRING1);   // (2s + zsh) mod 8 = ring-0 slot of plane 2s-1 (0 at s0); (2s) mod 5 = ring-1 slot of plane 2s
      // ---- DENSE: fragment addressing and the MFMA chain of a tile (Chain<NC>)
      constexpr int NCH = DENSE ? NC : 3;
      using CH = Chain<NCH>;
      struct DBase { unsigned pA, pB, pL[2][2]; int offL; };   // offL: byte offset of the leftover taps' plane (a constant per ring state)
      struct DFr { bf16x8 f[CH::NFP]; };
      auto dfrag = [&](unsigned pa, unsigned pb, int off) __attribute__((always_inline)) -> bf16x8 {
        const u32x2 a = *reinterpret_cast<const u32x2*>(lds + pa + off);
        const u32x2 b = *reinterpret_cast<const u32x2*>(lds + pb + off);
        return __builtin_bit_cast(bf16x8, (u32x4){a[0], a[1], b[0], b[1]});
      };
      auto dload = [&](auto nc, const DBase& b, DFr& q) __attribute__((always_inline)) {
        constexpr int n = decltype(nc)::value;
        if (!a_ld_g) { asm volatile("" : "+v"(q.f[n])); return; }   // (ablation: opaque, so that no MFMA is hoisted)
        if constexpr (CH::is_left(n)) q.f[n] = dfrag(b.pL[CH::left_w(n)][0], b.pL[CH::left_w(n)][1], CH::left_row(n) * RB0 + b.offL);
        else q.f[n] = dfrag(b.pA, b.pB, CH::iy(n) * RB0 + CH::arr(n) * SB0);
      };
      auto dmma = [&](auto kc, auto rc, const DFr& q, f32x4& acc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value, r = decltype(rc)::value;
        if (a_mma_g) acc = MFMA(wr[k], q.f[CH::frag(k, r)], acc);
        else asm volatile("" : "+v"(acc) : "v"(q.f[CH::frag(k, r)]));   // (ablation: the fragment reads stay alive)
      };
      DFr fa;
#if (LR_C01_ABL & (32 | 64))
      for (int i = 0; i < CH::NFP; ++i) fa.f[i] = __builtin_bit_cast(bf16x8, (u32x4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
#endif
      if constexpr (DENSE && PA0 > 0) {   // the first fragments of the column's first step (later steps: requested ahead of the barrier)
        const int t0 = (e6 % NRING0) * PLB0, t1 = ((e6 + 1) % NRING0) * PLB0, t2 = ((e6 + 2) % NRING0) * PLB0;
        DBase b0;
        b0.offL = 0;
        b0.pA = (unsigned)((lq == 0 ? t0 : lq == 1 ? t1 : t2) + ry0 * RB0) + dnA;
        b0.pB = (unsigned)((lq == 0 ? t0 : lq == 2 ? t2 : t1) + ry0 * RB0) + dnB;
        static_for<DENSE_PRE>([&](auto nc) __attribute__((always_inline)) { dload(nc, b0, fa); });
      }
      // DENSE: the ring-0 state e6 = (2 s) mod NRING0 repeats every NRING0 / 2 steps; the step body exists once per state, so
      // every plane offset is a constant of its copy and the per-lane fragment bases are LOOP INVARIANTS: plane slot v of tap
      // plane tz = 0 -> the lane's first / second half address (pair tiles PVA / PVB; the single tile PSA / PSB for the states'
      // plane spl), the leftover halves LB / LS.  (Computing them per step cost ~60 vector instructions on the producer
      // waves, the kernel's critical path.)
      unsigned PVA[NRING0], PVB[NRING0], PSA[NRING0 / 2], PSB[NRING0 / 2], LB[2][2], LS[2][2];
      if constexpr (DENSE) {
        const int dlA = lq < 3 ? lq : 2, dlB = lq < 3 ? lq : 1;
#pragma unroll
        for (int v = 0; v < NRING0; ++v) {
          PVA[v] = (unsigned)(((v + dlA) % NRING0) * PLB0 + ry0 * RB0) + dnA;
          PVB[v] = (unsigned)(((v + dlB) % NRING0) * PLB0 + ry0 * RB0) + dnB;
        }
#pragma unroll
        for (int k = 0; k < NRING0 / 2; ++k) {
          PSA[k] = (unsigned)(((2 * k + spl + dlA) % NRING0) * PLB0) + dnAs;
          PSB[k] = (unsigned)(((2 * k + spl + dlB) % NRING0) * PLB0) + dnBs;
        }
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            LB[w][h] = (unsigned)(ry0 * RB0) + dnL[w][h];
            LS[w][h] = (unsigned)(spl * PLB0) + dnLs[w][h];    // (state e6 <= NRING0 - 2: slot e6 + spl never wraps)
          }
      }
      // One step of the column.  ec: the ring-0 state as a compile-time constant, or -1 = dispatch on e6.
      auto a_iter = [&](const int s, auto ec) __attribute__((always_inline)) {
        constexpr int EC = decltype(ec)::value;
        constexpr bool a_on = !(LR_C01_ABL & 2);
        if (s < d.Do && a_on) {
          C01_STAMP(0);
          if (2 * s + 1 + zoff >= Dg)   // odd D, last step: plane 2s+1 lies below the volume — its slot must read 0 (its stores are dropped)
            for (int o = tid * 16; o < PLB1; o += 256 * 16) *reinterpret_cast<u32x4*>(lds + RING1_OFF + ((m5 + 1) % NRING1) * PLB1 + o) = (u32x4){0u, 0u, 0u, 0u};
          const bool zok0 = 2 * s + zoff < Dg, zok1 = 2 * s + 1 + zoff < Dg;
          const int so0 = m5 * PLB1, so1 = ((m5 + 1) % NRING1) * PLB1;
          // ring-1 store addresses of the five tiles
          const int a00 = zok0 && ok_p0 ? st_p0 + so0 : dump1, a01 = zok0 && ok_p1 ? st_p1 + so0 : dump1;
          const int a10 = zok1 && ok_p0 ? st_p0 + so1 : dump1, a11 = zok1 && ok_p1 ? st_p1 + so1 : dump1;
          const int as_ = (spl ? zok1 : zok0) && ok_s ? st_s + (spl ? so1 : so0) : dump1;
          if constexpr (DENSE) {
          // ---- dense K: five tiles x NMF MFMAs, one chain each; the epilogue of tile t rides beside tile t + 1.  (Ring 0 staged a
          // step ahead: the first fragments of the step were requested before the barrier.)
          auto dense_step = [&](auto ec) __attribute__((always_inline)) {
          constexpr int E6 = decltype(ec)::value;
          DBase bP0, bP1, bS, bN;
          bP0.pA = PVA[E6]; bP0.pB = PVB[E6]; bP0.offL = E6 * PLB0;
          bP1.pA = PVA[(E6 + 1) % NRING0]; bP1.pB = PVB[(E6 + 1) % NRING0]; bP1.offL = ((E6 + 1) % NRING0) * PLB0;
          bN.pA = PVA[(E6 + 2) % NRING0]; bN.pB = PVB[(E6 + 2) % NRING0]; bN.offL = 0;   // the first pair of step s + 1
          bS.pA = PSA[E6 / 2]; bS.pB = PSB[E6 / 2]; bS.offL = E6 * PLB0;
#pragma unroll
          for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              bP0.pL[w][h] = LB[w][h];
              bP1.pL[w][h] = LB[w][h];
              bS.pL[w][h] = LS[w][h];
              bN.pL[w][h] = 0u;
            }
          DFr fb, fs;
#if (LR_C01_ABL & (32 | 64))
          fb = fa; fs = fa;
#endif
          f32x4 acc0, acc1;
          Epi E0;
          C01_STAMP(1);
          auto epi_tile = [&](auto tc, int sl, const f32x4& acc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value;
            if constexpr (t == 0) epi_slice(sl, E0, acc, a00, zok0 ? svo_p0 : OOR, zok0 ? mko_p0 : OOR, 2 * s);
            else if constexpr (t == 1) epi_slice(sl, E0, acc, a01, zok0 ? svo_p1 : OOR, zok0 ? mko_p1 : OOR, 2 * s);
            else if constexpr (t == 2) epi_slice(sl, E0, acc, a10, zok1 ? svo_p0 : OOR, zok1 ? mko_p0 : OOR, 2 * s + 1);
            else if constexpr (t == 3) epi_slice(sl, E0, acc, a11, zok1 ? svo_p1 : OOR, zok1 ? mko_p1 : OOR, 2 * s + 1);
            else epi_slice(sl, E0, acc, as_, (spl ? zok1 : zok0) ? svo_s : OOR, (spl ? zok1 : zok0) ? mko_s : OOR, 2 * s + spl);
          };
          if constexpr (PA0 == 0) {   // (ring 0 is not staged ahead: the step's first fragments are requested here)
            static_for<DENSE_PRE>([&](auto nc) __attribute__((always_inline)) { dload(nc, bP0, fa); });
            C01_FENCE();
          }
          constexpr int NMF = CH::NMF;
          static_for<5 * NMF>([&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value, t = g / NMF, k = g % NMF;
            f32x4& acc = (t & 1) ? acc1 : acc0;
            const f32x4& prev = (t & 1) ? acc0 : acc1;
            if constexpr (k == 0) acc = bv;
            if constexpr (t < 2) dmma(std::integral_constant<int, k>{}, std::integral_constant<int, t>{}, fa, acc);
            else if constexpr (t < 4) dmma(std::integral_constant<int, k>{}, std::integral_constant<int, t - 2>{}, fb, acc);
            else dmma(std::integral_constant<int, k>{}, std::integral_constant<int, 0>{}, fs, acc);
            constexpr int ld = dense_load_at<NCH>(g);
            static_for<ld / 1024>([&](auto ic) __attribute__((always_inline)) {
              constexpr int q = dense_nth<NCH>(ld % 1024, decltype(ic)::value), set = q / 32, n = q % 32;
              if constexpr (set == 0) dload(std::integral_constant<int, n>{}, bP0, fa);
              else if constexpr (set == 1) dload(std::integral_constant<int, n>{}, bP1, fb);
              else dload(std::integral_constant<int, n>{}, bS, fs);
            });
            if constexpr (PA0 > 0 && g >= 5 * NMF - DENSE_PRE) dload(std::integral_constant<int, g - (5 * NMF - DENSE_PRE)>{}, bN, fa);   // (two bursts instead: the same)
            // the WHOLE epilogue of the tile before in ONE gap, the one in front of the next chain's first MFMA (seven gaps with four or
            // five vector instructions each: +3 %; two instructions in every gap: +32 %; one gap in the middle of the chain: +1.5 % —
            // an MFMA that follows its chain's predecessor directly is the cheap case)
            if constexpr (t >= 1 && k == CH::EPI_AT) {
#pragma unroll
              for (int q = 0; q < 7; ++q) epi_tile(std::integral_constant<int, t - 1>{}, q, prev);
            }
            C01_FENCE();
            if constexpr (k == NMF - 1 && t < 3) C01_STAMP(t == 0 ? 2 : t == 1 ? 3 : 7);   // (stamped build: the first three tiles)
          });
          C01_STAMP(4);
#pragma unroll
          for (int k = 0; k < 7; ++k) epi_tile(std::integral_constant<int, 4>{}, k, acc0);
          C01_STAMP(5);
          };
          // one copy of the step per ring state (e6 is wave-uniform: a scalar branch)
          if constexpr (EC >= 0) dense_step(std::integral_constant<int, EC>{});
          else if (e6 == 0) dense_step(std::integral_constant<int, 0>{});
          else if (e6 == 2) dense_step(std::integral_constant<int, 2>{});
          else if (e6 == 4) dense_step(std::integral_constant<int, 4>{});
          else dense_step(std::integral_constant<int, (NRING0 - 2)>{});      // (6 of an eight-plane ring; a six-plane ring has no fourth state)
          } else {
          // ring-0 fragment bases: plane pl of the step, the lane's taps in plane (2s - 1 + pl) + tz
          const int f0 = ((e6 + dlF) % NRING0) * PLB0, f1 = ((e6 + 1 + dlF) % NRING0) * PLB0;
          const int g0 = ((e6 + dlG) % NRING0) * PLB0, g1 = ((e6 + 1 + dlG) % NRING0) * PLB0;
          Base bP0, bP1, bS;
          bP0.pF = (unsigned)(f0 + ry0 * RB0) + laneF; bP0.pG = (unsigned)(g0 + ry0 * RB0) + laneG; bP0.pG1 = bP0.pG + RB0;
          bP1.pF = (unsigned)(f1 + ry0 * RB0) + laneF; bP1.pG = (unsigned)(g1 + ry0 * RB0) + laneG; bP1.pG1 = bP1.pG + RB0;
          asm volatile("" : "+v"(bP0.pG1), "+v"(bP1.pG1));   // opaque: the two loads of the shared record stay two loads (register tuples)
          bS.pF = (unsigned)(spl ? f1 : f0) + laneFs; bS.pG = (unsigned)(spl ? g1 : g0) + laneGs; bS.pG1 = 0;
#if !(LR_C01_ABL & 64)
          PairFrags fa, fb;
          OneFrags fs;
#endif
          Acc2 aa, ab;
          Epi E0, E1;
          // the first pair's fragments: the one LDS round trip of the step this wave waits for (its SIMD partner computes)
#pragma unroll
          for (int n = 0; n < 6; ++n) load_pair_n(n, bP0, fa);
          C01_FENCE();
          C01_STAMP(1);
          // stage 0: pair of plane 0 | its other 12 fragments (two groups ahead of their use), then those of the pair of plane 1
          aa.v[0] = bv; aa.v[1] = bv;
#pragma unroll
          for (int I = 0; I < 48; ++I) {
            mma_pair_n(I, fa, aa);
            if (I < 12) load_pair_n(6 + I, bP0, fa);
            else if ((I & 1) == 0) load_pair_n((I - 12) >> 1, bP1, fb);
            C01_FENCE();
          }
          C01_STAMP(2);
          // stage 1: pair of plane 1 | epilogue of the pair of plane 0 (14 slices) | fragments of the single tile
          ab.v[0] = bv; ab.v[1] = bv;
#pragma unroll
          for (int I = 0; I < 48; ++I) {
            mma_pair_n(I, fb, ab);
            if ((I & 1) == 0 && I < 24) load_one_n(I >> 1, bS, fs);
            if (I >= 3 && I < 45 && (I % 3) == 0) {
              const int k_ = I / 3 - 1;
              if (k_ < 7) epi_slice(k_, E0, aa.v[0], a00, zok0 ? svo_p0 : OOR, zok0 ? mko_p0 : OOR, 2 * s);
              else epi_slice(k_ - 7, E1, aa.v[1], a01, zok0 ? svo_p1 : OOR, zok0 ? mko_p1 : OOR, 2 * s);
            }
            C01_FENCE();
          }
          C01_STAMP(3);
          // stage 2: the single tile | epilogue of the pair of plane 1 (14 slices over 24 MFMAs)
          aa.v[0] = bv; aa.v[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int I = 0; I < 24; ++I) {
            mma_one_n(I, fs, aa);
            if (I >= 2 && I < 23 && (I % 3) != 1) {   // I = 2,3,5,6,...,20,21: slices 0..13
              const int k_ = (I - 2) / 3 * 2 + ((I - 2) % 3 == 0 ? 0 : 1);
              if (k_ < 7) epi_slice(k_, E0, ab.v[0], a10, zok1 ? svo_p0 : OOR, zok1 ? mko_p0 : OOR, 2 * s + 1);
              else epi_slice(k_ - 7, E1, ab.v[1], a11, zok1 ? svo_p1 : OOR, zok1 ? mko_p1 : OOR, 2 * s + 1);
            }
            C01_FENCE();
          }
          C01_STAMP(4);
          // tail: epilogue of the single tile
          const f32x4 accS = aa.v[0] + aa.v[1];
#pragma unroll
          for (int k = 0; k < 7; ++k) epi_slice(k, E0, accS, as_, (spl ? zok1 : zok0) ? svo_s : OOR, (spl ? zok1 : zok0) ? mko_s : OOR, 2 * s + spl);
          C01_STAMP(5);
          }
        }
        __syncthreads();
        C01_STAMP(6);
        e6 = (e6 + 2) % NRING0;
        m5 = __builtin_amdgcn_readfirstlane(m5 + 2 >= NRING1 ? m5 + 2 - NRING1 : m5 + 2);   // (kept in a scalar register: the store addresses and the planes' validity hang on it)
      };
      if constexpr (DENSE && PA0 > 0 && NRING0 == 8) {
        // The four ring states in program order (state 0 at s0): the fragments requested ahead of a barrier stay in the registers
        // the next state reads them from.  (With one dispatching loop the four copies merged at the loop's latch: 24 register
        // moves per step BEHIND a wait for those loads — their LDS latency, which the early request was to hide, came back.)
        int s = s0;
        for (;;) {
          a_iter(s, std::integral_constant<int, 0>{}); if (++s > d.Do) break;
          a_iter(s, std::integral_constant<int, 2>{}); if (++s > d.Do) break;
          a_iter(s, std::integral_constant<int, 4>{}); if (++s > d.Do) break;
          a_iter(s, std::integral_constant<int, 6>{}); if (++s > d.Do) break;
        }
      } else {
        for (int s = s0; s <= d.Do; ++s) a_iter(s, std::integral_constant<int, -1>{});
      }
  }
  } else {
  // The SIMD's arbiter serves the OLDER wave first (waves 0..3, the A role, were dispatched first): left alone, B only got
  // matrix and issue slots once A had finished its step (B's 84 MFMAs took 5.1 k cycles, A then waited 1.8 k at the barrier).
  // B's stream is the light one (MFMAs paced by LDS reads, little vector work): with static priority it takes its slots when
  // its operands are there and A fills the rest (MI355X_MICROARCH.md, two waves per SIMD, item 4).
  __builtin_amdgcn_s_setprio(LR_C01_BPRIO);
  for (int uid = first; uid < end; uid += stride) {
    const int utx = uid % d.nTx, uty = (uid / d.nTx) % d.nTy, ub = __builtin_amdgcn_readfirstlane(uid / d.nTx / d.nTy);
    const int oy0 = __builtin_amdgcn_readfirstlane(uty * TY), ox0 = __builtin_amdgcn_readfirstlane(utx * TX);
    const int Y0 = 2 * oy0 - 2, X0a = 2 * ox0 - 4;      // volume coordinates of ring-0 row 0 / record 0
    const __amdgpu_buffer_rsrc_t r0 = make_rsrc(in0 + (int64_t)ub * d.bs0, V4);
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(in_rest + (int64_t)ub * d.bsr, NC > 1 ? (unsigned)(NC - 1) * V4 : 0u);

    // one staging item = input plane zi, ring-0 row irow, x-quad iq, every channel: request -> registers ...
    // (element access on the ext-vector result of the buffer-load builtin is narrowed by hipcc to a one-dword load with the
    // other elements undefined — DESIGN.md 6a: the quads are viewed through HIP's uint4, a struct, where they are used)
    auto issue_item = [&](bool live, int zi, int irow, int iq, u32x4 (&L)[NC]) __attribute__((always_inline)) {
      const int yi = Y0 + irow, xi = X0a + QS0 + 4 * iq;
      const int zg = zi + zoff;
      const int ok = (int)live & (int)(zg >= 0) & (int)(zg < Dg) & (int)(yi >= 0) & (int)(yi < dW);
      const int ok_lo = ok & (int)(xi >= 0) & (int)(xi < dH), ok_hi = ok & (int)(xi + 2 >= 0) & (int)(xi + 2 < dH);
      const unsigned dead_lo = ((unsigned)ok_lo - 1u) & OOR, dead_hi = ((unsigned)ok_hi - 1u) & OOR;   // dead half: bit 31 -> outside the resource -> 0
      const unsigned voff = (unsigned)((((zi + zrel) * dW + yi) * dH + xi) * 4);
      auto halves = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned vo) __attribute__((always_inline)) -> u32x4 {
        const uint2 a = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(vo | dead_lo), 0, 0));
        const uint2 b = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)((vo + 8u) | dead_hi), 0, 0));
        return (u32x4){a.x, a.y, b.x, b.y};
      };
      L[0] = halves(r0, voff);
#pragma unroll
      for (int c = 1; c < NC; ++c) L[c] = halves(rr, voff + (unsigned)(c - 1) * V4);
    };
    // ... -> split -> ring 0 (plane z sits in slot (z + 1) mod 6); a thread without an item writes zeros into the dump area
    auto write_item = [&](bool live, int zi, int irow, int iq, const u32x4 (&L)[NC]) __attribute__((always_inline)) {
      const int slot = (zi + 1 + zsh + NRING0) % NRING0;
      unsigned char* const base = lds + (live ? slot * PLB0 + irow * RB0 + QS0 * 8 + iq * 32 : DUMP_OFF + lane * 16);
      if constexpr (NC == 5) {   // four records per voxel (dense5_records); channel 4 is split in voxel pairs
        auto val = [&](int c, int j) __attribute__((always_inline)) -> float {
          const uint4 q = __builtin_bit_cast(uint4, L[c]);
          return __builtin_bit_cast(float, j == 0 ? q.x : j == 1 ? q.y : j == 2 ? q.z : q.w);
        };
        unsigned rec5[4][4][2];   // [record][voxel][half]
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          unsigned p4[3];
          split3(val(4, 2 * jp), val(4, 2 * jp + 1), p4);
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * jp + jj;
            unsigned p01[3], p23[3], d4[3], r4[4][2];
            split3(val(0, j), val(1, j), p01);
            split3(val(2, j), val(3, j), p23);
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) d4[sp] = jj ? (p4[sp] >> 16) : (p4[sp] & 0xffffu);
            dense5_records(p01, p23, d4, r4);
#pragma unroll
            for (int a = 0; a < 4; ++a) { rec5[a][j][0] = r4[a][0]; rec5[a][j][1] = r4[a][1]; }
          }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          *reinterpret_cast<u32x4*>(base + a * SB0) = (u32x4){rec5[a][0][0], rec5[a][0][1], rec5[a][1][0], rec5[a][1][1]};
          *reinterpret_cast<u32x4*>(base + a * SB0 + 16) = (u32x4){rec5[a][2][0], rec5[a][2][1], rec5[a][3][0], rec5[a][3][1]};
        }
        return;
      }
      float v[4][4];   // [channel][voxel]
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint4 q = __builtin_bit_cast(uint4, L[c < NC ? c : 0]);
        v[c][0] = c < NC ? __builtin_bit_cast(float, q.x) : 0.0f; v[c][1] = c < NC ? __builtin_bit_cast(float, q.y) : 0.0f;
        v[c][2] = c < NC ? __builtin_bit_cast(float, q.z) : 0.0f; v[c][3] = c < NC ? __builtin_bit_cast(float, q.w) : 0.0f;
      }
      unsigned rec[3][4][2];   // [split][voxel][channels 01 | 23]   (DENSE: [record E, F, G][voxel][half])
      if constexpr (DENSE) {   // (three channels)
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          unsigned p2[3], p01[3], r3[3][2];
          split3(v[2][2 * jp], v[2][2 * jp + 1], p2);
          split3(v[0][2 * jp], v[1][2 * jp], p01);
          dense_records_c2pair<0>(p01, p2, r3);
#pragma unroll
          for (int s = 0; s < 3; ++s) { rec[s][2 * jp][0] = r3[s][0]; rec[s][2 * jp][1] = r3[s][1]; }
          split3(v[0][2 * jp + 1], v[1][2 * jp + 1], p01);
          dense_records_c2pair<1>(p01, p2, r3);
#pragma unroll
          for (int s = 0; s < 3; ++s) { rec[s][2 * jp + 1][0] = r3[s][0]; rec[s][2 * jp + 1][1] = r3[s][1]; }
        }
      } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned p01[3], p23[3] = {0u, 0u, 0u};
        split3(v[0][j], v[1][j], p01);
        if (NC > 2) split3(v[2][j], v[3][j], p23);
#pragma unroll
        for (int s = 0; s < 3; ++s) { rec[s][j][0] = p01[s]; rec[s][j][1] = p23[s]; }
      }
      }
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        *reinterpret_cast<u32x4*>(base + s * SB0) = (u32x4){rec[s][0][0], rec[s][0][1], rec[s][1][0], rec[s][1][1]};
        *reinterpret_cast<u32x4*>(base + s * SB0 + 16) = (u32x4){rec[s][2][0], rec[s][2][1], rec[s][3][0], rec[s][3][1]};
      }
    };

    // ---- unit prologue: ring 1 <- 0 (every voxel of the column outside the volume), ring 0 <- planes 2 s0 - 1 .. 2 s0 + 4
    // (NPRO0 x 66 items); the B threads also request planes 2 s0 + 5, + 6 (their items of the first step)
    // staging items of a step: B waves 2 and 3 (the K quarters with three k-blocks: 72 MFMAs against 96) take one plane = 55 each
    const int bi = (wq - 2) * 55 + lane;
    const bool item_live = !is_a && wq >= 2 && lane < 55;
    const bool wave_stages = !is_a && wq >= 2;   // wave-uniform
    const int ipl = item_live ? bi / (R0Y * NQS) : 0, irow = item_live ? (bi % (R0Y * NQS)) / NQS : 0, iq = item_live ? bi % NQS : 0;
    u32x4 ldn[NC];
    __syncthreads();   // (the zero fill of the whole LDS | every read of the unit before)
    {
      const bool pl_live = tid < NPRO0 * R0Y * NQS;
      const int pz = pl_live ? tid / (R0Y * NQS) - 1 + 2 * s0 : 0, prow = pl_live ? (tid % (R0Y * NQS)) / NQS : 0, pq = pl_live ? tid % NQS : 0;
      u32x4 lp[NC];
      issue_item(pl_live, pz, prow, pq, lp);
      issue_item(item_live, 2 * s0 + 3 + PA0 + ipl, irow, iq, ldn);
      for (int o = tid * 16; o < RING1; o += NTHR * 16) *reinterpret_cast<u32x4*>(lds + RING1_OFF + o) = (u32x4){0u, 0u, 0u, 0u};
      write_item(pl_live, pz, prow, pq, lp);
    }
    __syncthreads();

      // =========================================================== B: block 1 ===========================================
      // Tile t = output rows 2t + {0,1} of the column; lane column = (row r1, voxel oxl); lq = (tap of the k-block's pair,
      // channel half).  K quarter kq: taps tap0 + 2*kb + {0,1} < tap1, both cout tiles.
      const int r1 = col >> 3, oxl = col & 7, hh = lq & 1, tsel = lq >> 1;
      // ring-1 byte offset of this lane's operand half for tile 0, k-block kb, without the plane slot (tile 1: + T1OFF);
      // tdzp: two bits per k-block = the plane (0..2) its tap lies in
      constexpr int T1OFF = 2 * 2 * RS1 * 8;
      unsigned pq[NKB1];
      unsigned tdzp = 0;
#pragma unroll
      for (int kb = 0; kb < NKB1; ++kb) {
        int tap = tap0 + 2 * kb + tsel;
        tap = tap >= tap1 ? tap1 - 1 : tap;   // slots behind the quarter: weight 0 on a real voxel
        const int dz = tap / 9, dy = (tap % 9) / 3, dx = tap % 3;
        pq[kb] = (unsigned)(RING1_OFF + ((2 * r1 + dy) * RS1 + 2 * oxl + hh * QS1 + dx) * 8);
        tdzp |= (unsigned)dz << (2 * kb);
      }
      // the output voxel this lane stores: tile kq >> 1 of the column, cout tile bc = kq & 1
      const int oy = oy0 + 2 * (kq >> 1) + r1, ox = ox0 + oxl;
      const bool o_in = oy < d.Wo && ox < d.Ho;
      unsigned ooff;
      if (d.hps) {   // row = [channel block of 16][parity][Ho/2][16 floats]
        const int hp = (ox & 1) * (d.Ho >> 1) + (ox >> 1);
        ooff = (unsigned)((oy * d.Ho * 32 + (bc * d.Ho + hp) * 16 + lq * 4) * 4);
      } else {
        ooff = (unsigned)(((oy * d.Ho + ox) * 32 + bc * 16 + lq * 4) * 4);
      }
      if (!o_in) ooff = OOR;
      f32x4 own = {0.f, 0.f, 0.f, 0.f};
      // staging of this thread's item in 2 slices: per voxel pair (0,1 | 2,3) four splits (voxel x channels 01 | 23) and the
      // pair's three 16-byte stores; the slices read the requested quads where they landed (`ldn`), the next request goes out
      // behind the second slice (most of a step of latency).  (Finer slices kept 12 more registers alive across the MFMA groups
      // than this role has.)
      auto lval = [&](int c, int j) __attribute__((always_inline)) -> float {
        if (c >= NC) return 0.0f;
        const uint4 q = __builtin_bit_cast(uint4, ldn[c < NC ? c : 0]);
        return __builtin_bit_cast(float, j == 0 ? q.x : j == 1 ? q.y : j == 2 ? q.z : q.w);
      };
      auto stage_slice = [&](int half, int addr) __attribute__((always_inline)) {
        if constexpr (NC == 5) {
          unsigned p4[3], rec5[4][2][2];
          split3(lval(4, 2 * half), lval(4, 2 * half + 1), p4);
#pragma unroll
          for (int vv = 0; vv < 2; ++vv) {
            const int vj = 2 * half + vv;
            unsigned p01[3], p23[3], d4[3], r4[4][2];
            split3(lval(0, vj), lval(1, vj), p01);
            split3(lval(2, vj), lval(3, vj), p23);
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) d4[sp] = vv ? (p4[sp] >> 16) : (p4[sp] & 0xffffu);
            dense5_records(p01, p23, d4, r4);
#pragma unroll
            for (int a = 0; a < 4; ++a) { rec5[a][vv][0] = r4[a][0]; rec5[a][vv][1] = r4[a][1]; }
          }
#pragma unroll
          for (int a = 0; a < 4; ++a)
            *reinterpret_cast<u32x4*>(lds + addr + a * SB0 + 16 * half) = (u32x4){rec5[a][0][0], rec5[a][0][1], rec5[a][1][0], rec5[a][1][1]};
          return;
        }
        unsigned rec[3][2][2];
        if constexpr (DENSE) {
          unsigned p01[3], p2[3], r3[3][2];
          split3(lval(2, 2 * half), lval(2, 2 * half + 1), p2);
          split3(lval(0, 2 * half), lval(1, 2 * half), p01);
          dense_records_c2pair<0>(p01, p2, r3);
#pragma unroll
          for (int sp = 0; sp < 3; ++sp) { rec[sp][0][0] = r3[sp][0]; rec[sp][0][1] = r3[sp][1]; }
          split3(lval(0, 2 * half + 1), lval(1, 2 * half + 1), p01);
          dense_records_c2pair<1>(p01, p2, r3);
#pragma unroll
          for (int sp = 0; sp < 3; ++sp) { rec[sp][1][0] = r3[sp][0]; rec[sp][1][1] = r3[sp][1]; }
        } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int vj = 2 * half + (kk >> 1), hf = kk & 1;
          unsigned pp[3] = {0u, 0u, 0u};
          if (hf) { if (NC > 2) split3(lval(2, vj), lval(3, vj), pp); }
          else split3(lval(0, vj), lval(1, vj), pp);
#pragma unroll
          for (int sp = 0; sp < 3; ++sp) rec[sp][kk >> 1][hf] = pp[sp];
        }
        }
#pragma unroll
        for (int sp = 0; sp < 3; ++sp)
          *reinterpret_cast<u32x4*>(lds + addr + sp * SB0 + 16 * half) = (u32x4){rec[sp][0][0], rec[sp][0][1], rec[sp][1][0], rec[sp][1][1]};
      };
      // output plane oz in 3 slices: the K quarters of this wave's accumulator, summed in quarter order (its own from registers),
      // bias, LeakyReLU, store (oz < 0: zero-length resource, no branch)
      f32x4 fin_a, fin_b;
      auto fin_read = [&](int q, int oz) __attribute__((always_inline)) -> f32x4 {
        const int slot = q == kq ? 0 : kq - (kq > q ? 1 : 0);   // (q == kq: its own partial sum is in registers — the value read is dropped)
        return *reinterpret_cast<const f32x4*>(lds + SCR_OFF + ((((oz & 1) * 4 + q) * 3 + slot) * 64 + lane) * 16);
      };
      auto fin_slice = [&](int k, int oz) __attribute__((always_inline)) {
        if (k == 0) { fin_a = fin_read(0, oz); fin_b = fin_read(1, oz); }
        else if (k == 1) {
          fin_a = (kq == 0 ? own : fin_a) + (kq == 1 ? own : fin_b);
          fin_b = fin_read(2, oz);
        } else {
          const f32x4 p3 = fin_read(3, oz);
          float* const pbase = out + ((int64_t)ub * d.out_bs + (int64_t)(oz < 0 ? 0 : oz) * d.Wo * d.Ho * 32);
          const __amdgpu_buffer_rsrc_t ores = make_rsrc(pbase, oz < 0 ? 0u : (unsigned)(d.Wo * d.Ho * 32 * 4));
          f32x4 v = fin_a + (kq == 2 ? own : fin_b);
          v = (v + (kq == 3 ? own : p3)) + bv;
          v = max4(v, v * d.slope1);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ores, (int)ooff, 0, 0);
        }
      };
      constexpr int NST = 2, NSL = 5;   // slices of a step: 2 staging + 3 output

#if (LR_C01_ABL & 64)
      bf16x8 fr[2][3];
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) fr[g][sp] = frag(pq[g], sp * SPB1, 2 * QS1 * 8);
#endif
      for (int s = s0; s <= d.Do; ++s) {
        C01_STAMP(0);
        const int r0addr = item_live ? ((2 * s + 3 + PA0 + ipl + 1 + zsh) % NRING0) * PLB0 + irow * RB0 + QS0 * 8 + iq * 32 : DUMP_OFF + lane * 16;
        C01_STAMP(1);
        if (s >= 1) {
          const int oz = s - 1;
          // planes 2oz-1, 2oz, 2oz+1 of block 0's output: slots (2oz - 1 + 5) % 5 ...
          const int p0 = (2 * oz + 4) % NRING1;
          const unsigned so0 = (unsigned)(p0 * PLB1), so1 = (unsigned)(((p0 + 1) % NRING1) * PLB1), so2 = (unsigned)(((p0 + 2) % NRING1) * PLB1);
          unsigned pa[NKB1];
#pragma unroll
          for (int kb = 0; kb < NKB1; ++kb) {
            const unsigned dz = (tdzp >> (2 * kb)) & 3u;
            pa[kb] = pq[kb] + (dz == 0 ? so0 : dz == 1 ? so1 : so2);
          }
          constexpr bool b_ld = !(LR_C01_ABL & (8 | 64)), b_mma = !(LR_C01_ABL & 1);
          f32x4 hi[2][2], lo[2][2];
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c) { hi[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f}; lo[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
          // 2 nkb groups (k-block, tile) of 12 MFMAs (both cout tiles: every fragment feeds four MFMAs); the fragments of group
          // g + 1 are requested at the head of group g; behind each half group one slice of the staging (12) or of the output of
          // the step before (2)
#if !(LR_C01_ABL & 64)
          bf16x8 fr[2][3];
#pragma unroll
          for (int sp = 0; sp < 3; ++sp) fr[0][sp] = frag(pa[0], sp * SPB1, 2 * QS1 * 8);
#endif
          C01_FENCE();
          auto group = [&](int g) __attribute__((always_inline)) {
            const int kb = g >> 1, t = g & 1;
            if ((g + 1 < 6 || (g + 1 < 8 && nkb == 4)) && b_ld) {
#pragma unroll
              for (int sp = 0; sp < 3; ++sp) fr[(g + 1) & 1][sp] = frag(pa[(g + 1) >> 1], ((g + 1) & 1) * T1OFF + sp * SPB1, 2 * QS1 * 8);
            }
            const bf16x8 (&f)[3] = fr[g & 1];
            const u32x4 (&w)[NWR] = wr;
#define C01_W(C, T) w[(kb * 2 + (C)) * 3 + (T)]
            if (b_mma) {
            lo[t][0] = MFMA(C01_W(0, 1), f[1], lo[t][0]); lo[t][1] = MFMA(C01_W(1, 1), f[1], lo[t][1]);
            lo[t][0] = MFMA(C01_W(0, 2), f[0], lo[t][0]); lo[t][1] = MFMA(C01_W(1, 2), f[0], lo[t][1]);
            lo[t][0] = MFMA(C01_W(0, 0), f[2], lo[t][0]); lo[t][1] = MFMA(C01_W(1, 0), f[2], lo[t][1]);
            }
            C01_FENCE();
            if (2 * g < NST) { if (wave_stages && !(LR_C01_ABL & 128)) stage_slice(2 * g, r0addr); } else if (2 * g < NSL) fin_slice(2 * g - NST, oz - 1);
            C01_FENCE();
            if (b_mma) {
            lo[t][0] = MFMA(C01_W(0, 1), f[0], lo[t][0]); lo[t][1] = MFMA(C01_W(1, 1), f[0], lo[t][1]);
            lo[t][0] = MFMA(C01_W(0, 0), f[1], lo[t][0]); lo[t][1] = MFMA(C01_W(1, 0), f[1], lo[t][1]);
            hi[t][0] = MFMA(C01_W(0, 0), f[0], hi[t][0]); hi[t][1] = MFMA(C01_W(1, 0), f[0], hi[t][1]);
            }
#undef C01_W
            C01_FENCE();
            if (2 * g + 1 < NST) { if (wave_stages && !(LR_C01_ABL & 128)) stage_slice(2 * g + 1, r0addr); } else if (2 * g + 1 < NSL) fin_slice(2 * g + 1 - NST, oz - 1);
            if (2 * g + 1 == NST - 1 && wave_stages) issue_item(item_live, 2 * s + 5 + PA0 + ipl, irow, iq, ldn);
            C01_FENCE();
          };
#pragma unroll
          for (int g = 0; g < 6; ++g) group(g);
          if (nkb == 4) {   // (all 11 slices fit into the first six groups)
            group(6);
            group(7);
          }
          C01_STAMP(3);
          // the accumulators this wave does not own -> LDS (read by their owners after the barrier); its own stays in registers
          f32x4 acc[4];
#pragma unroll
          for (int a4 = 0; a4 < 4; ++a4) acc[a4] = hi[a4 >> 1][a4 & 1] + lo[a4 >> 1][a4 & 1];
          own = kq == 0 ? acc[0] : kq == 1 ? acc[1] : kq == 2 ? acc[2] : acc[3];
#pragma unroll
          for (int a3 = 0; a3 < 3; ++a3) {   // accumulator a4 = a3 + (a3 >= kq)
            const f32x4 v = kq == 0 ? acc[a3 + 1] : kq == 1 ? acc[a3 == 0 ? 0 : a3 + 1] : kq == 2 ? acc[a3 == 2 ? 3 : a3] : acc[a3];
            *reinterpret_cast<f32x4*>(lds + SCR_OFF + ((((oz & 1) * 4 + kq) * 3 + a3) * 64 + lane) * 16) = v;
          }
          C01_STAMP(4);
        } else {
          if (wave_stages) {
#pragma unroll
            for (int k = 0; k < NST; ++k) stage_slice(k, r0addr);
            issue_item(item_live, 2 * s + 5 + PA0 + ipl, irow, iq, ldn);
          }
        }
        __syncthreads();
        C01_STAMP(6);
      }
      fin_slice(0, d.Do - 1);
      fin_slice(1, d.Do - 1);
      fin_slice(2, d.Do - 1);
  }
  }
#ifdef LR_C01_STAMPS
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(&g_lr_c01_stamps[wave * 8 + i], stamp_acc[i]);
  }
#endif
}

// packed1[(((kq*4 + kb)*2 + c)*3 + t)*64 + lane]: lane (co = lane & 15, lq = lane >> 4) holds split t of
// W1[c*16 + co][ch = 8*(lq & 1) + e][tap = tap0(kq) + 2*kb + (lq >> 1)], e = 0..7 — zeros for taps behind the quarter
// (quarters: taps 0..7 | 8..15 | 16..21 | 22..26)
__global__ void pack_c01_w1_kernel(const float* __restrict__ w, u32x4* __restrict__ packed) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 4 * 4 * 2 * 64) return;
  const int lane = idx & 63, c = (idx >> 6) & 1, kb = (idx >> 7) & 3, kq = idx >> 9;
  const int co = lane & 15, lq = lane >> 4;
  const int tap0 = kq == 0 ? 0 : kq == 1 ? 8 : kq == 2 ? 16 : 22, tap1 = kq == 0 ? 8 : kq == 1 ? 16 : kq == 2 ? 22 : 27;
  const int tap = tap0 + 2 * kb + (lq >> 1);
  unsigned r[3][4];
#pragma unroll
  for (int pr = 0; pr < 4; ++pr) {
    float v[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int ch = 8 * (lq & 1) + 2 * pr + hh;
      v[hh] = tap < tap1 ? w[((int64_t)(c * 16 + co) * 16 + ch) * 27 + tap] : 0.0f;
    }
    unsigned p[3];
    split3(v[0], v[1], p);
#pragma unroll
    for (int t = 0; t < 3; ++t) r[t][pr] = p[t];
  }
#pragma unroll
  for (int t = 0; t < 3; ++t)
    packed[((((kq * 4 + kb) * 2 + c) * 3) + t) * 64 + lane] = (u32x4){r[t][0], r[t][1], r[t][2], r[t][3]};
}

constexpr int64_t W1_FLOATS = (int64_t)4 * 4 * 2 * 3 * 64 * 4;

// tap (tz, tx) of record slot (lane group lq, half h) of a dense main fragment (the kernel's dnA / dnB / plA / plB); leftover: (0, 2)
__host__ __device__ constexpr int main_tap_tz(int lq, int h) { return h == 0 ? (lq < 3 ? lq : 2) : (lq < 3 ? lq : 1); }
__host__ __device__ constexpr int main_tap_tx(int lq, int h) { return h == 0 ? (lq < 3 ? 0 : 1) : (lq < 2 ? 1 : 2); }

// The 17 weight fragments of the DENSE block 0 (three input channels), in the order of a tile's chain: WG[ty] WE2[ty] WF[ty]
// WE1[ty] WL0 WL1 WE0[ty].  Lane (co = lane & 15, lq = lane >> 4) holds the 8 bf16 of its two record slots (half h = e >> 2,
// position e & 3).  Main fragments (tap row ty): slot (lq, h) = tap (tz = lq, ty, tx = h) for lq < 3, (tz = h, ty, 2) for lq = 3;
// leftover fragments: slot s = 8 w + 2 lq + h = tap (0, s / 5, 2), kind s % 5; the patterns per kind are in
// the comment of dense_records.
__global__ void pack_c01_w0_dense_kernel(const float* __restrict__ w, u32x4* __restrict__ packed) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Chain<3>::NMF * 64) return;
  const int f = idx >> 6, lane = idx & 63, co = lane & 15, lq = lane >> 4;
  unsigned el[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int h = e >> 2, pos = e & 3;
    int kind, tap;   // kind 0..2 = E x w_kind, 3 = F, 4 = G, -1 = empty slot
    if (f < 12 || f >= 14) {
      const int grp = f < 12 ? f / 3 : 4, ty = f < 12 ? f % 3 : f - 14;
      kind = grp == 0 ? 4 : grp == 1 ? 2 : grp == 2 ? 3 : grp == 3 ? 1 : 0;
      const int tz = main_tap_tz(lq, h), tx = main_tap_tx(lq, h);
      tap = (tz * 3 + ty) * 3 + tx;
    } else {
      const int sidx = (f - 12) * 8 + lq * 2 + h;
      kind = sidx > 14 ? -1 : sidx % 5;
      tap = (0 * 3 + (sidx > 14 ? 0 : sidx / 5)) * 3 + 2;
    }
    int sp = -1, c = 0;   // weight split, channel
    if (kind >= 0 && kind <= 2) { if (pos < 3) { sp = kind; c = pos; } else if (kind < 2) { sp = kind; c = 0; } }
    else if (kind == 3) { sp = 0; c = pos == 0 ? 1 : pos == 1 ? 2 : pos == 2 ? 0 : 1; }
    else if (kind == 4) { if (pos == 0) { sp = 0; c = 2; } else if (pos < 3) { sp = 1; c = pos; } }
    unsigned p[3] = {0u, 0u, 0u};
    if (sp >= 0) split3(w[((int64_t)co * 3 + c) * 27 + tap], 0.0f, p);
    el[e] = sp >= 0 ? (p[sp == 0 ? 0 : sp == 1 ? 1 : 2] & 0xffffu) : 0u;
  }
  packed[f * 64 + lane] = (u32x4){el[0] | (el[1] << 16), el[2] | (el[3] << 16), el[4] | (el[5] << 16), el[6] | (el[7] << 16)};
}
constexpr int64_t W0D_FLOATS = (int64_t)Chain<3>::NMF * 64 * 4;

// The 27 weight fragments of block 0 with FIVE input channels, in the order of a tile's chain (Chain<5>): phase k / 3 = A1 w2, A4,
// A3 w1, A3 w0, A2 w1, A1 w1, leftover, A2 w0, A1 w0; tap row ty = k % 3.  Slots of a main fragment as in pack_c01_w0_dense_kernel;
// the leftover fragment of tap row ty: slot 2 lq + h = pattern A1 w0, A1 w1, A1 w2, A2 w0, A2 w1, A3 w0, A3 w1, A4 of tap (0, ty, 2).
// Patterns (weight split, channel) per position: comment of dense5_records.
__global__ void pack_c01_w0_dense5_kernel(const float* __restrict__ w, u32x4* __restrict__ packed) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Chain<5>::NMF * 64) return;
  const int f = idx >> 6, lane = idx & 63, co = lane & 15, lq = lane >> 4;
  const int phase = f / 3, ty = f % 3;
  unsigned el[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int h = e >> 2, pos = e & 3;
    int pat, tap;   // pattern 0..7 = A1 w0, A1 w1, A1 w2, A2 w0, A2 w1, A3 w0, A3 w1, A4
    if (phase == 6) {
      pat = lq * 2 + h;
      tap = (0 * 3 + ty) * 3 + 2;
    } else {
      pat = phase == 0 ? 2 : phase == 1 ? 7 : phase == 2 ? 6 : phase == 3 ? 5 : phase == 4 ? 4 : phase == 5 ? 1 : phase == 7 ? 3 : 0;
      const int tz = main_tap_tz(lq, h), tx = main_tap_tx(lq, h);
      tap = (tz * 3 + ty) * 3 + tx;
    }
    int sp = -1, c = 0;
    if (pat < 3) { sp = pat; c = pos; }
    else if (pat < 5) { sp = pat - 3; c = pos == 0 ? 4 : pos - 1; }
    else if (pat == 5) { sp = 0; c = pos == 0 ? 3 : pos == 1 ? 4 : pos == 2 ? 0 : 1; }
    else if (pat == 6) { if (pos < 2) { sp = 1; c = 3 + pos; } }
    else { sp = pos == 3 ? 2 : 0; c = pos == 3 ? 4 : 2 + pos; }
    unsigned p[3] = {0u, 0u, 0u};
    if (sp >= 0) split3(w[((int64_t)co * 5 + c) * 27 + tap], 0.0f, p);
    el[e] = sp >= 0 ? (p[sp == 0 ? 0 : sp == 1 ? 1 : 2] & 0xffffu) : 0u;
  }
  packed[f * 64 + lane] = (u32x4){el[0] | (el[1] << 16), el[2] | (el[3] << 16), el[4] | (el[5] << 16), el[6] | (el[7] << 16)};
}
constexpr int64_t W0D5_FLOATS = (int64_t)Chain<5>::NMF * 64 * 4;

}  // namespace

#ifdef LR_C01_STAMPS
extern "C" int lr_debug_read_c01_stamps(unsigned long long* host64, int reset) {
  if (hipMemcpyFromSymbol(host64, HIP_SYMBOL(g_lr_c01_stamps), 64 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[64] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_lr_c01_stamps), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
#endif

// ---- C ABI (include/liftreg_hip.h)
extern "C" int64_t lr_conv3d_pair01_packed_floats(int Cin, int C0, int C1) {
  if (Cin < 1 || Cin > 5 || C0 != 16 || C1 != 32) return 0;
  if (Cin == 5) return W1_FLOATS + W0D5_FLOATS;   // [block 1][block 0, dense K]
  return lr_internal_conv0_split_packed_floats(Cin, C0) + W1_FLOATS + (Cin == 3 ? W0D_FLOATS : 0);   // [block 0][block 1][block 0, dense K]
}

// floats in front of block 1's fragments
static int64_t pair01_w1_offset(int Cin) { return Cin == 5 ? 0 : lr_internal_conv0_split_packed_floats(Cin, 16); }

extern "C" int lr_conv3d_pair01_pack_f32(const float* w0, const float* w1, float* packed, int Cin, int C0, int C1, void* stream) {
  if (!w0 || !w1 || !packed) return LR_ENULL;
  if (lr_conv3d_pair01_packed_floats(Cin, C0, C1) == 0) return LR_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(packed) & 15u) return LR_EALIGN;
  hipStream_t st = lr_stream(stream);
  if (Cin <= 4) {
    const int rc = lr_internal_conv0_split_pack(w0, packed, Cin, C0, st);
    if (rc != LR_OK) return rc;
  }
  u32x4* p1 = reinterpret_cast<u32x4*>(packed + pair01_w1_offset(Cin));
  hipLaunchKernelGGL(pack_c01_w1_kernel, dim3((4 * 4 * 2 * 64 + 255) / 256), dim3(256), 0, st, w1, p1);
  if (Cin == 3)
    hipLaunchKernelGGL(pack_c01_w0_dense_kernel, dim3((Chain<3>::NMF * 64 + 255) / 256), dim3(256), 0, st, w0,
                       reinterpret_cast<u32x4*>(packed + pair01_w1_offset(Cin) + W1_FLOATS));
  if (Cin == 5)
    hipLaunchKernelGGL(pack_c01_w0_dense5_kernel, dim3((Chain<5>::NMF * 64 + 255) / 256), dim3(256), 0, st, w0,
                       reinterpret_cast<u32x4*>(packed + W1_FLOATS));
  return lr_launch_status();
}

static int pair01_impl(const float* in0, int64_t in0_batch_stride, const float* in_rest, int64_t rest_batch_stride,
                       const float* packed, const float* bias0, const float* bias1, float* out, int B, int Cin,
                       int D, int W, int H, int out_layout, float slope0, float slope1, int64_t out_batch_stride,
                       int D_global, int z_lo, int oz_lo, int n_oz, float* act0, unsigned char* mask0, int mid_layout, void* stream) {
  if (!in0 || !packed || !out || (Cin > 1 && !in_rest)) return LR_ENULL;
  const bool save = act0 != nullptr || mask0 != nullptr;
  if (save) {   // training forward: the whole volume, both side outputs, dense
    if (!act0 || !mask0) return LR_ENULL;
    if (D != D_global || z_lo != 0 || oz_lo != 0 || n_oz != (D - 1) / 2 + 1) return LR_EUNSUPPORTED;
    if (mid_layout != LR_LAYOUT_NDHWC && mid_layout != LR_LAYOUT_NDHWC_HPS) return LR_EUNSUPPORTED;
    if (mid_layout == LR_LAYOUT_NDHWC_HPS && (H & 1)) return LR_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(act0) & 15u) || (reinterpret_cast<uintptr_t>(mask0) & 3u)) return LR_EALIGN;
    if ((int64_t)D * W * H * 64 >= 0x7fffffffLL) return LR_EUNSUPPORTED;   // one batch element of the activation is one buffer resource
  }
  if (B < 1 || D < 1 || W < 1 || H < 1 || D_global < 1 || n_oz < 1 || oz_lo < 0 || z_lo < 0) return LR_EINVAL;
  // the buffers hold global planes [z_lo, z_lo + D); output planes [oz_lo, oz_lo + n_oz) read input planes 2*oz_lo - 2 ..
  // 2*(oz_lo + n_oz - 1) + 2: every one of them that exists must lie inside the buffers
  if (oz_lo + n_oz > (D_global - 1) / 2 + 1 || z_lo + D > D_global) return LR_EINVAL;
  {
    const int need_lo = 2 * oz_lo - 2 < 0 ? 0 : 2 * oz_lo - 2;
    const int need_hi = 2 * (oz_lo + n_oz - 1) + 2 >= D_global ? D_global - 1 : 2 * (oz_lo + n_oz - 1) + 2;
    if (need_lo < z_lo || need_hi >= z_lo + D) return LR_EINVAL;
  }
  if (Cin < 1 || Cin > 5) return LR_EUNSUPPORTED;
  if (out_layout != LR_LAYOUT_NDHWC && out_layout != LR_LAYOUT_NDHWC_HPS) return LR_EUNSUPPORTED;
  if (!(slope0 >= 0.0f && slope0 <= 1.0f) || !(slope1 >= 0.0f && slope1 <= 1.0f)) return LR_EUNSUPPORTED;   // LeakyReLU = max(v, slope v)
  if (H & 3) return LR_EUNSUPPORTED;
  const int64_t V = (int64_t)D * W * H;
  FDims d;
  d.B = B; d.Cin = Cin; d.D = D; d.W = W; d.H = H;
  d.Do = n_oz; d.Wo = (W - 1) / 2 + 1; d.Ho = (H - 1) / 2 + 1;
  d.Dg = D_global; d.zoff = 2 * oz_lo; d.zlo = z_lo;
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
  if ((int64_t)(Cin > 4 ? Cin - 1 : 3) * V * 4 + (int64_t)8 * W * H * 4 >= 0x7fffffffLL) return LR_EUNSUPPORTED;   // 31-bit byte offsets inside a batch element
  if ((int64_t)d.Wo * d.Ho * 32 * 4 >= 0x7fffffffLL) return LR_EUNSUPPORTED;                // an output plane is one buffer resource
  d.nTx = (d.Ho + TX - 1) / TX; d.nTy = (d.Wo + TY - 1) / TY;
  const int64_t nu = (int64_t)B * d.nTy * d.nTx;
  if (nu > 0x7fffffffLL) return LR_EINVAL;
  d.nunits = (int)nu;
  d.slope0 = slope0; d.slope1 = slope1;
  d.act0 = act0; d.mask0 = mask0; d.save_hps = mid_layout == LR_LAYOUT_NDHWC_HPS;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  int blocks = cus;   // one block per CU (LDS)
  // LIFTREG_PAIR01_BLOCKS: fewer persistent blocks leave whole CUs to a kernel of another stream (the HBM-bound decode of the
  // previous batch: liftreg_amd/pipeline.py) — this kernel fills the register file of every CU it sits on
  if (lr_sw_set(LR_SW_PAIR01_BLOCKS)) { const int v = lr_sw_int(LR_SW_PAIR01_BLOCKS, cus); if (v >= 1 && v < blocks) blocks = v; }
  if (blocks > d.nunits) blocks = d.nunits;
  hipStream_t st = lr_stream(stream);
  const u32x4* wp0 = reinterpret_cast<const u32x4*>(packed);
  const u32x4* wp1 = reinterpret_cast<const u32x4*>(packed + pair01_w1_offset(Cin));
  if (!in_rest) in_rest = in0;   // Cin == 1: never dereferenced (zero-length resource)
  // three channels: block 0 with K packed densely (17 instead of 24 MFMAs per tile); LIFTREG_PAIR01_DENSE=0: the padded form (A/B aid)
  const bool densek = Cin == 3 && lr_sw_int(LR_SW_PAIR01_DENSE, 1) != 0;
  if (densek || Cin == 5) wp0 = reinterpret_cast<const u32x4*>(packed + pair01_w1_offset(Cin) + W1_FLOATS);
#define LR_C01(NCV, SV, DN)                                                                                                    \
  do {                                                                                                                       \
    static std::atomic<uint64_t> attr_done{0};                                                                               \
    if (lr_raise_dyn_lds(reinterpret_cast<const void*>(&conv01_fused_kernel<NCV, SV, DN>), Geo<NCV>::LDSB, attr_done) != LR_OK) return LR_ELAUNCH; \
    hipLaunchKernelGGL((conv01_fused_kernel<NCV, SV, DN>), dim3((unsigned)blocks), dim3(NTHR), Geo<NCV>::LDSB, st, in0, in_rest, wp0, wp1, bias0, bias1, out, d); \
  } while (0)
  if (save) {
    if (Cin == 1) LR_C01(1, true, false);
    else if (Cin == 2) LR_C01(2, true, false);
    else if (Cin == 3 && densek) LR_C01(3, true, true);
    else if (Cin == 3) LR_C01(3, true, false);
    else if (Cin == 4) LR_C01(4, true, false);
    else LR_C01(5, true, true);
  } else {
    if (Cin == 1) LR_C01(1, false, false);
    else if (Cin == 2) LR_C01(2, false, false);
    else if (Cin == 3 && densek) LR_C01(3, false, true);
    else if (Cin == 3) LR_C01(3, false, false);
    else if (Cin == 4) LR_C01(4, false, false);
    else LR_C01(5, false, true);
  }
#undef LR_C01
  return lr_launch_status();
}

extern "C" int lr_conv3d_pair01_slab_f32(const float* in0, int64_t in0_batch_stride, const float* in_rest, int64_t rest_batch_stride,
                                         const float* packed, const float* bias0, const float* bias1, float* out, int B, int Cin,
                                         int D, int W, int H, int out_layout, float slope0, float slope1, int64_t out_batch_stride,
                                         int D_global, int z_lo, int oz_lo, int n_oz, void* stream) {
  return pair01_impl(in0, in0_batch_stride, in_rest, rest_batch_stride, packed, bias0, bias1, out, B, Cin, D, W, H, out_layout,
                     slope0, slope1, out_batch_stride, D_global, z_lo, oz_lo, n_oz, nullptr, nullptr, 0, stream);
}

// Training forward of the two blocks: lr_conv3d_pair01_f32 that ALSO writes block 0's activation act0 (B,D,W,H,16) fp32 in
// mid_layout (LR_LAYOUT_NDHWC | LR_LAYOUT_NDHWC_HPS) and its LeakyReLU sign mask mask0 (B,D,W,H,4) uint8 (LR_LAYOUT_SIGN4, as
// lr_conv3d_k3_lrelu_mask_f32) — the two tensors the backward of the pair reads (block 1's weight gradient; the fused
// data-gradient + block-0 weight-gradient kernel).  act0 holds exactly the fp32 values whose three-way splits fed block 1.
extern "C" int lr_conv3d_pair01_train_f32(const float* in0, int64_t in0_batch_stride, const float* in_rest, int64_t rest_batch_stride,
                                          const float* packed, const float* bias0, const float* bias1, float* out, float* act0,
                                          uint8_t* mask0, int B, int Cin, int D, int W, int H, int mid_layout, int out_layout,
                                          float slope0, float slope1, void* stream) {
  if (D < 1) return LR_EINVAL;
  if (!act0 || !mask0) return LR_ENULL;
  return pair01_impl(in0, in0_batch_stride, in_rest, rest_batch_stride, packed, bias0, bias1, out, B, Cin, D, W, H, out_layout,
                     slope0, slope1, 0, D, 0, 0, (D - 1) / 2 + 1, act0, mask0, mid_layout, stream);
}

extern "C" int lr_conv3d_pair01_f32(const float* in0, int64_t in0_batch_stride, const float* in_rest, int64_t rest_batch_stride,
                                    const float* packed, const float* bias0, const float* bias1, float* out, int B, int Cin,
                                    int D, int W, int H, int out_layout, float slope0, float slope1, int64_t out_batch_stride,
                                    void* stream) {
  if (D < 1) return LR_EINVAL;
  return lr_conv3d_pair01_slab_f32(in0, in0_batch_stride, in_rest, rest_batch_stride, packed, bias0, bias1, out, B, Cin, D, W, H,
                                   out_layout, slope0, slope1, out_batch_stride, D, 0, 0, (D - 1) / 2 + 1, stream);
}
