// conv3d_bwd_fused.hip — training backward of the encoder's first TWO blocks in one kernel:
//
//   gpre0 = dgrad(block 1)(gpre1) * LeakyReLU'(block 0 output)            (the data gradient of the 16 -> 32 stride-2 block)
//   gw0, gb0 = wgrad(block 0)(x0, gpre0)                                  (the weight / bias gradient of the first block)
//
// As two kernels (lr_conv3d_dgrad_f32, lr_conv3d_wgrad_f32) gpre0 — (B,D,W,H,16) fp32, 8.6 GB at C3 — is written by the
// first and read back by the second, and nothing else ever uses it (the first block's input is data: no gradient flows
// further).  Here it never leaves the registers: the data-gradient MFMAs run with the operands swapped (A = the gpre1
// window, B = the weights), so a 16-voxel x 16-channel gpre0 tile lands with the VOXELS on the accumulator rows — which
// is exactly the B-operand layout (k = voxel, n = channel) of the weight-gradient MFMA  D[(ci,tap)][co] += X^T gpre0,
// whose A operand (one input voxel of the first block per lane and k-step) comes from an LDS tile of x0.
// HBM traffic of the pair: 2.1 (gpre1) + 0.5 (sign mask) + ~2 (x0 with halo) GB instead of 11.2 + 10.2 GB.
//
// Structure = conv3d_dgrad_wlds_kernel (conv3d_bwd.hip): persistent 8-wave blocks, one per CU; all packed weights of block
// 1 in LDS for the life of the block (54 KB); a tile = 8 x 2 x 16 quotient voxels (= 16 x 4 x 32 voxels of gpre0), one
// quotient plane per wave; the gpre1 window (9 x 3 x 17 voxels x 32 channels, 57 KB, records ROTATED by the voxel index
// instead of padded so that 16-byte reads of 8 neighbouring voxels hit 8 different bank groups) and the x0 window
// (Cin0 x 18 x 6 x 36 floats, 46 KB) are staged from registers that were filled during the previous tile.  Per parity
// class (pz,py,px) of gpre0 and quotient row: the data-gradient k-loop (1..8 taps x 2 channel blocks x 4 MFMAs), the
// mask (one dword of the SIGN4 mask per voxel, bit = channel), then 4 x NTJ weight-gradient MFMAs into NTJ accumulators
// that live for the whole kernel; one partial per wave, fixed-order double reduce (wgrad0_finish_kernel).
//
// Tile shapes (template NZ = quotient planes per tile).  Cin0 <= 3: NZ = 8, a wave owns one quotient plane and both quotient
// rows (LDS 160.7 KB at Cin0 = 3).  Cin0 = 4, 5 (the reference's shipped 4-view configuration, cur_task_setting.json:56): the
// x0 window of an 8-plane tile does not fit beside the weights (191.9 KB), so NZ = 4: a tile = 4 x 2 x 16 quotient voxels, TWO
// waves per quotient plane, one quotient row each (gpre1 window 5 x 3 x 17 voxels, x0 window Cin0 x 10 x 6 x 36 floats: 131.1
// KB at Cin0 = 5); a wave's single data-gradient chain then alternates between two accumulators (even / odd k quarters) so
// that no MFMA waits for the one before it, as the two rows' chains do in the 8-plane form.
//
// Replaces: autograd of src/liftreg/layers/layers.py:365-369 for blocks 0 and 1 as wired at
//   src/liftreg/models/LiftRegDeformSubspaceBackproj.py:29-33,95-100 (RegistrationNet.py:401 total_loss.backward()).
#include "lr_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOR = 0x80000000u;

struct FzDims {
  int B, D, W, H, Do, Wo, Ho;  // gpre0 / x0 / mask0 volume (D,W,H); gpre1 is (B,Do,Wo,Ho,32)
  int nHq, nWq, nDq;
  float slope;                 // LeakyReLU slope of block 0
};

constexpr int TR = 2;                  // quotient rows per tile
constexpr int CB = 2, CG = 32, C4 = 8;  // block 1: 32 output channels = 2 channel blocks = 8 16-byte chunks per voxel
constexpr int XY = 2 * TR + 2, XX = 36; // x0 window of a tile: rows, floats per row (origin (-1,-1,-2)); planes: 2 NZ + 2
constexpr int nz_of(int cin0) { return cin0 <= 3 ? 8 : 4; }   // quotient planes per tile (LDS: see the header comment)

template <int CIN0, int NZ>
__global__ __launch_bounds__(512, 1) void dgrad_wgrad0_kernel(const float* __restrict__ gpre, const float4* __restrict__ wp,
                                                              const unsigned* __restrict__ mask0, const float* __restrict__ in0,
                                                              long long bs0, const float* __restrict__ in_rest, long long bsr,
                                                              float* __restrict__ partial, FzDims d, int ntiles) {
  constexpr int WPP = 8 / NZ;                        // waves per quotient plane
  constexpr int WMT = TR / WPP;                      // quotient rows per wave
  constexpr int XZ = 2 * NZ + 2;
  static_assert(NZ == 8 || NZ == 4, "tile shape");
  constexpr int NVOX = (NZ + 1) * (TR + 1) * 17, NCH = NVOX * C4, NIT = (NCH + 511) / 512;
  constexpr int NWF = 27 * CB * 64;                  // float4 weight fragments (the first 27 taps of the packed buffer)
  // 8-byte pairs of the x0 window, enumerated channel by channel (NPI thread-iterations each): the channel of an iteration is a
  // compile-time constant, so channel 0 (the moving image) and channels 1.. (the backprojected views) may live in two buffers
  constexpr int NPC = XZ * XY * (XX / 2), NPI = (NPC + 511) / 512, NXI = CIN0 * NPI;
  // columns (ci, tap) + the ones column (bias).  Cin0 = 3: 81 taps = 5 column tiles + ONE tap; that tap and the bias sum
  // are accumulated on the vector ALU (one broadcast LDS read + 2 flops per voxel and lane) instead of a sixth, 88 % empty
  // MFMA tile (VT); the partial buffer keeps the 6-tile layout.
  constexpr int NCOL = 27 * CIN0 + 1, NTP = (NCOL + 15) / 16;
  constexpr bool VT = (27 * CIN0) % 16 == 1;
  constexpr int NTJ = VT ? NTP - 1 : NTP;
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  float4* wl = reinterpret_cast<float4*>(dsm);  // [27][CB][64 lanes]
  float* ts = dsm + NWF * 4;                     // [NVOX] records of 32 floats, chunk c of voxel v at ((c + v) & 7)
  float* xs = ts + NVOX * 32;                    // [CIN0][XZ][XY][XX]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wz = wave / WPP, wy = (wave % WPP) * WMT;   // this wave's quotient plane and first quotient row inside the tile
  for (int i = tid; i < NWF; i += 512) wl[i] = wp[i];
  const int nWq = (d.Wo + TR - 1) / TR, nDq = (d.Do + NZ - 1) / NZ;
  auto tile_coords = [&](int t, int& b, int& zq0, int& yq0, int& xq0) {
    const int hq = t % d.nHq; t /= d.nHq;
    const int wq = t % nWq; t /= nWq;
    const int dq = t % nDq;
    b = t / nDq;
    zq0 = dq * NZ; yq0 = wq * TR; xq0 = hq * 16;
  };
  float4 st[NIT];
  float2 xst[NXI];
  // Which element of the two windows a thread fetches is the same for every tile.  Decoding it per tile cost ~600 vector
  // ALU instructions per thread (divisions by 17, 3, 18, 6) — issued by all 8 waves at once, right after the barrier,
  // with the matrix pipe idle; letting hipcc hoist the decoded offsets cost 40 registers and spilled.  So: ONE packed
  // word per element, built once — fields (x | y << 8 | z << 16) of 7 bits + a guard bit each, bits 24..26 "first in
  // x / y / z" (x0 window only), bit 27 "past the window", c4 / ci in bits 28..30 — and per tile one add of a
  // bias vector makes every field that is past the volume overflow into its guard bit: one and + compare per element.
  unsigned pkg[NIT], pkx[NXI];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int q = it * 512 + tid;
    const int vox = q / C4, c4 = q % C4;
    const int xx = vox % 17, r = vox / 17, yy = r % (TR + 1), zz = r / (TR + 1);
    pkg[it] = q < NCH ? (unsigned)(xx | yy << 8 | zz << 16 | c4 << 28) : 0x08000000u;  // bit 27: past the window
  }
#pragma unroll
  for (int it = 0; it < NXI; ++it) {
    const int q = (it % NPI) * 512 + tid;
    const int xx2 = q % (XX / 2), row = q / (XX / 2), yy = row % XY, zz = row / XY;
    pkx[it] = q < NPC ? (unsigned)(xx2 | yy << 8 | zz << 16 | (xx2 == 0) << 24 | (yy == 0) << 25 | (zz == 0) << 26)
                      : 0x08000000u;
  }
  auto prefetch = [&](int t) {
    int b, zq0, yq0, xq0;
    tile_coords(t, b, zq0, yq0, xq0);
    auto bias7 = [](int rem) { return (unsigned)(128 - (rem < 0 ? 0 : rem > 128 ? 128 : rem)); };  // field >= rem -> guard bit
    {
      // resource = the tile's own origin: offsets stay inside nine planes whatever the volume size
      const float* base = gpre + ((((int64_t)b * d.Do + zq0) * d.Wo + yq0) * d.Ho + xq0) * CG;
      const __amdgpu_buffer_rsrc_t rsrc =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, 0x7fffffff, 0x00020000);
      const unsigned bias = bias7(d.Ho - xq0) | bias7(d.Wo - yq0) << 8 | bias7(d.Do - zq0) << 16;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const unsigned pk = pkg[it];
        const unsigned xx = pk & 127u, yy = (pk >> 8) & 127u, zz = (pk >> 16) & 127u, c4 = pk >> 28;
        const bool ok = (((pk & 0x0fffffffu) + bias) & 0x08808080u) == 0u;
        const unsigned voff = ok ? (((zz * (unsigned)d.Wo + yy) * (unsigned)d.Ho + xx) * CG + c4 * 4) * 4 : OOR;
        st[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
      }
    }
    {
      // the first block's input around the tile (origin (2zq0-1, 2yq0-1, 2xq0-2)): one resource per batch element, zero
      // outside the volume (= the conv's padding); H is even, so an 8-byte pair never straddles the volume's edge
      const __amdgpu_buffer_rsrc_t rs0 =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in0 + (int64_t)b * bs0), (short)0, 0x7fffffff, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsr =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_rest + (int64_t)b * bsr), (short)0, 0x7fffffff, 0x00020000);
      const int gz0 = 2 * zq0 - 1, gy0 = 2 * yq0 - 1, gx0 = 2 * xq0 - 2;
      const unsigned bias = bias7((d.H - gx0 + 1) >> 1) | bias7(d.W - gy0) << 8 | bias7(d.D - gz0) << 16;
      const unsigned lowm = 0x08808080u | (unsigned)(xq0 == 0) << 24 | (unsigned)(yq0 == 0) << 25 | (unsigned)(zq0 == 0) << 26;
      const unsigned org = (unsigned)(((gz0 * d.W + gy0) * d.H + gx0) * 4);  // may wrap: added modulo 2^32 to a valid element's offset
#pragma unroll
      for (int it = 0; it < NXI; ++it) {
        const unsigned pk = pkx[it];
        const unsigned xx2 = pk & 127u, yy = (pk >> 8) & 127u, zz = (pk >> 16) & 127u;
        const unsigned ci = (unsigned)(it / NPI), cr = ci == 0 ? 0u : ci - 1u;      // channel | its index inside in_rest
        const bool ok = (((pk & 0x0fffffffu) + bias) & lowm) == 0u;
        const unsigned voff = ok ? ((((cr * (unsigned)d.D + zz) * (unsigned)d.W + yy) * (unsigned)d.H + 2 * xx2) * 4 + org) : OOR;
        xst[it] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(it / NPI == 0 ? rs0 : rsr, voff, 0, 0));
      }
    }
  };
  const int col = lane & 15, kq = lane >> 4;
  // gpre1 window reads: voxel vi = c + u with u = (wz*3 + wy)*17 + col per lane and c a compile-time constant of the k-step;
  // chunk (cb*4 + kq) of voxel vi sits at rotation (cb*4 + kq + vi) & 7 = (kq + u + e) & 7 with e = (c + 4*cb) & 7
  unsigned prot[8];
  {
    const int u = (wz * (TR + 1) + wy) * 17 + col;
#pragma unroll
    for (int e = 0; e < 8; ++e) prot[e] = (unsigned)(u * 32 + ((kq + u + e) & 7) * 4);
  }
  // x0 window reads of the weight-gradient A operand: column n = j*16 + col -> (ci, tap); k = kq -> voxel 4kq + r
  unsigned xbase[NTJ];
  float xconst = 0.0f;      // last column tile: value of the columns past the taps (1 = the ones column -> gb, else 0)
  bool xreal_last = true;   // ... and whether this lane's column there is a real tap
#pragma unroll
  for (int j = 0; j < NTJ; ++j) {
    int n = j * 16 + col;
    if (n >= 27 * CIN0) {
      xreal_last = false;
      xconst = n == 27 * CIN0 ? 1.0f : 0.0f;
      n = 27 * CIN0 - 1;
    }
    const int ci = n / 27, tap = n - ci * 27, tz = tap / 9, ty = (tap / 3) % 3, tx = tap % 3;
    xbase[j] = (unsigned)((((ci * XZ + tz + 2 * wz) * XY + ty + 2 * wy) * XX + tx + 8 * kq + 1));
  }
  // VT: the last tap (ci = Cin0-1, tz = ty = tx = 2) for this lane's voxels 4kq + r, and the two scalar accumulators
  const unsigned xbase_t = (unsigned)(((((CIN0 - 1) * XZ + 2 + 2 * wz) * XY + 2 + 2 * wy) * XX + 2 + 8 * kq + 1));
  float vt_tap = 0.0f, vt_bias = 0.0f;
  const unsigned msh = (unsigned)((col >> 2) * 8 + (col & 3));  // this lane's channel bit in a voxel's SIGN4 dword
  f32x4 gacc[NTJ];
#pragma unroll
  for (int j = 0; j < NTJ; ++j) gacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int t = (int)blockIdx.x;
  if (t < ntiles) prefetch(t);
  for (; t < ntiles; t += (int)gridDim.x) {
    __syncthreads();  // every wave is done with the previous tile (first pass: the weights are in LDS after the next one)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const unsigned pk = pkg[it];
      const unsigned vox = (((pk >> 16) & 127u) * (TR + 1) + ((pk >> 8) & 127u)) * 17 + (pk & 127u), c4 = pk >> 28;
      if (it * 512 + tid < NCH) *reinterpret_cast<float4*>(ts + vox * 32 + ((c4 + vox) & 7) * 4) = st[it];
    }
#pragma unroll
    for (int it = 0; it < NXI; ++it) {
      const int q = (it % NPI) * 512 + tid;
      if (q < NPC) *reinterpret_cast<float2*>(xs + ((it / NPI) * NPC + q) * 2) = xst[it];  // rows are XX floats: the pair index IS the layout
    }
    __syncthreads();
    int b, zq0, yq0, xq0;
    tile_coords(t, b, zq0, yq0, xq0);
    if (t + (int)gridDim.x < ntiles) prefetch(t + (int)gridDim.x);  // lands while this tile runs on the matrix pipe
    const int zq = zq0 + wz;
    const bool interior = 2 * (xq0 + 16) <= d.H && 2 * (yq0 + TR) <= d.W;
    const unsigned mvoff = (unsigned)(2 * (xq0 + 4 * kq) * 4);  // byte offset of this lane's first voxel in a mask row
#pragma unroll
    for (int pp = 3; pp >= 0; --pp) {
      const int py = pp & 1, pz = pp >> 1;
      const int z = 2 * zq + pz;
      if (z >= d.D) continue;  // wave-uniform; no barrier inside the class loop
      f32x4 accp[2][WMT];
      f32x4 accq[2][WMT];   // (one row per wave) the chain's second accumulator: k quarters y, w
      unsigned mw[2][WMT][4];
      // the mask dwords of the class pair first, a whole pair ahead of their use: one resource per batch element, the
      // row as the scalar offset, the voxel (2*(xq0 + 4kq + r) + px) as one per-lane offset + immediates; a row outside
      // the volume reads through the zero-length resource, a voxel past its row reads a neighbour or, past the tensor, zeros (masked below)
#pragma unroll
      for (int mt = 0; mt < WMT; ++mt) {
        const int y = 2 * (yq0 + wy + mt) + py;
        const __amdgpu_buffer_rsrc_t rsm = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned*>(mask0 + (int64_t)b * d.D * d.W * d.H), (short)0, y < d.W ? (int)((int64_t)d.D * d.W * d.H * 4) : 0, 0x00020000);
        const unsigned rowb = (unsigned)((z * d.W + y) * d.H * 4);
        // this lane's 8 voxels 2*(4kq + r) + px of the row are 8 consecutive dwords: two 16-byte loads
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(rsm, mvoff, rowb, 0);
        const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(rsm, mvoff + 16u, rowb, 0);
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
          for (int r = 0; r < 4; ++r) mw[px][mt][r] = r < 2 ? lo[2 * r + px] : hi[2 * (r - 2) + px];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int px = 1; px >= 0; --px) {
#pragma unroll
        for (int mt = 0; mt < WMT; ++mt) accp[px][mt] = accq[px][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int NS = (1 << (px + py + pz)) * CB;
        float4 a0[WMT], a1[WMT], b0, b1;
        auto load_step = [&](int s, float4 (&a)[WMT], float4& bw) __attribute__((always_inline)) {
          const int tapi = s / CB, cb = s - tapi * CB;
          const int ix = tapi & px, r1 = tapi >> px, iy = r1 & py, iz = (r1 >> py) & pz;
          const int tx = px ? 2 * ix : 1, ox = px ? 1 - ix : 0;
          const int ty = py ? 2 * iy : 1, oy = py ? 1 - iy : 0;
          const int tz = pz ? 2 * iz : 1, oz = pz ? 1 - iz : 0;
          const int sfull = ((tz * 3 + ty) * 3 + tx) * CB + cb;
          bw = wl[sfull * 64 + lane];
#pragma unroll
          for (int mt = 0; mt < WMT; ++mt) {
            const int c = ((oz * (TR + 1)) + oy + mt) * 17 + ox;  // voxel offset of this step inside the window
            a[mt] = *reinterpret_cast<const float4*>(ts + prot[(c + 4 * cb) & 7] + c * 32);
          }
        };
        auto mfma_step = [&](const float4 (&a)[WMT], const float4& bw) __attribute__((always_inline)) {
          // A = the gpre1 voxels (rows = voxels), B = the weights (cols = gpre0 channels); the two quotient rows alternate
          // on the matrix pipe so no MFMA waits for the one before it (one row per wave: two accumulators alternate)
          if constexpr (WMT == 1) {
            accp[px][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0].x, bw.x, accp[px][0], 0, 0, 0);
            accq[px][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0].y, bw.y, accq[px][0], 0, 0, 0);
            accp[px][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0].z, bw.z, accp[px][0], 0, 0, 0);
            accq[px][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0].w, bw.w, accq[px][0], 0, 0, 0);
          } else {
#pragma unroll
            for (int mt = 0; mt < WMT; ++mt) accp[px][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, bw.x, accp[px][mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < WMT; ++mt) accp[px][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, bw.y, accp[px][mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < WMT; ++mt) accp[px][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].z, bw.z, accp[px][mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < WMT; ++mt) accp[px][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].w, bw.w, accp[px][mt], 0, 0, 0);
          }
        };
        load_step(0, a0, b0);
#pragma unroll
        for (int s2 = 0; s2 + 1 < NS; s2 += 2) {
          load_step(s2 + 1, a1, b1);
          mfma_step(a0, b0);
          load_step(s2 + 2 < NS ? s2 + 2 : NS - 1, a0, b0);
          mfma_step(a1, b1);
          __builtin_amdgcn_sched_barrier(0);  // keep the reads one step ahead, not further (registers)
        }
      }
      // mask: lane = channel col of voxels 4kq + r; a voxel outside the volume contributes nothing
#pragma unroll
      for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int mt = 0; mt < WMT; ++mt) {
          f32x4 v = accp[px][mt];
          if constexpr (WMT == 1) v += accq[px][mt];
          const bool yok = 2 * (yq0 + wy + mt) + py < d.W;  // wave-uniform
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float m = ((mw[px][mt][r] >> msh) & 1u) ? 1.0f : d.slope;
            // edge tiles only (block-uniform flag folded into the compare): a voxel outside the volume contributes nothing
            const bool ok = interior | (yok && 2 * (xq0 + 4 * kq + r) + px < d.H);
            v[r] = ok ? v[r] * m : 0.0f;
          }
          accp[px][mt] = v;
        }
      __builtin_amdgcn_sched_barrier(0);
      // weight gradient: gacc[j][(ci,tap) rows][co cols] += x0(voxel + tap)^T gpre0(voxel); k-step r = voxels 4kq + r
      float xa[NTJ], xb[NTJ];
      auto ldx = [&](float (&xv)[NTJ], int g) __attribute__((always_inline)) {  // g = (px*WMT + mt)*4 + r
        const int r = g & 3, mt = (g >> 2) % WMT, px = g / (4 * WMT);
        const int off = (pz * XY + 2 * mt + py) * XX + 2 * r + px;
#pragma unroll
        for (int j = 0; j < NTJ; ++j) xv[j] = xs[xbase[j] + off];
        if (!VT && !xreal_last) xv[NTJ - 1] = xconst;
      };
      ldx(xa, 0);
#pragma unroll
      for (int g = 0; g < 2 * WMT * 4; ++g) {
        const int r = g & 3, mt = (g >> 2) % WMT, px = g / (4 * WMT);
        ldx(xb, g + 1 < 2 * WMT * 4 ? g + 1 : g);
        const float bv = accp[px][mt][r];
        if constexpr (VT) {
          const float xt = xs[xbase_t + (pz * XY + 2 * mt + py) * XX + 2 * r + px];
          vt_tap = fmaf(xt, bv, vt_tap);
          vt_bias += bv;
        }
#pragma unroll
        for (int j = 0; j < NTJ; ++j) gacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[j], bv, gacc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NTJ; ++j) xa[j] = xb[j];
        __builtin_amdgcn_sched_group_barrier(0x100, NTJ, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NTJ, 0);
      }
    }
  }
  // one partial per wave: partial[pb][co][n], n = j*16 + 4kq + r (lane: co = col)
  const int pb = (int)blockIdx.x * 8 + wave;
  constexpr int ncols = NTP * 16;
#pragma unroll
  for (int j = 0; j < NTJ; ++j)
    *reinterpret_cast<f32x4*>(partial + ((int64_t)pb * 16 + col) * ncols + j * 16 + kq * 4) = gacc[j];
  if constexpr (VT) {  // the four lane groups hold different voxels of the same channel: fold them, lane group 0 writes
    vt_tap += __shfl_xor(vt_tap, 16, 64);
    vt_tap += __shfl_xor(vt_tap, 32, 64);
    vt_bias += __shfl_xor(vt_bias, 16, 64);
    vt_bias += __shfl_xor(vt_bias, 32, 64);
    float* tail = partial + ((int64_t)pb * 16 + col) * ncols + NTJ * 16;
    if (kq == 0) { tail[0] = vt_tap; tail[1] = vt_bias; }
#pragma unroll
    for (int i = 2; i < 16; ++i)
      if (kq == 0) tail[i] = 0.0f;
  }
}

// partial[k][co][n] summed over the k partials in double, fixed order; column n = ci*27 + tap, n = 27*Cin = the bias column
__global__ __launch_bounds__(1024) void wgrad0_finish_kernel(const float* __restrict__ partial, float* __restrict__ gw,
                                                             float* __restrict__ gb, int nblk, int Cin, int ncols) {
  __shared__ double red[16][64];
  const int tx = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + tx;  // (co, n)
  const bool live = t < 16 * ncols;
  double s = 0.0;
  if (live) {
    const int64_t step = (int64_t)16 * ncols;
    const int per = (nblk + 15) / 16, k0 = sl * per, k1 = min(nblk, k0 + per);
    // four independent chains: the loads of a chain wait for nothing but each other's issue (one chain = per dependent
    // HBM/L2 round trips, 0.1 ms at 256 partials per slice)
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = k0;
    for (; k + 3 < k1; k += 4) {
      s += (double)partial[k * step + t];
      s1 += (double)partial[(k + 1) * step + t];
      s2 += (double)partial[(k + 2) * step + t];
      s3 += (double)partial[(k + 3) * step + t];
    }
    for (; k < k1; ++k) s += (double)partial[k * step + t];
    s = (s + s1) + (s2 + s3);
  }
  red[sl][tx] = s;
  __syncthreads();
  if (sl != 0 || !live) return;
#pragma unroll
  for (int i = 1; i < 16; ++i) s += red[i][tx];
  const int co = t / ncols, n = t - co * ncols;
  if (n == 27 * Cin) { if (gb) gb[co] = (float)s; }
  else if (n < 27 * Cin) gw[(int64_t)co * Cin * 27 + n] = (float)s;
}

template <int CIN0, int NZ>
int launch(const float* gpre1, const float* packed_w1T, const unsigned char* mask0, const float* in0, long long bs0,
           const float* in_rest, long long bsr, float* partial, float* gw0, float* gb0, const FzDims& d, int ntiles, int blocks,
           hipStream_t st) {
  constexpr int NTP = (27 * CIN0 + 1 + 15) / 16;
  constexpr int XZ = 2 * NZ + 2;
  constexpr size_t ldsb = ((size_t)27 * CB * 64 * 4 + (size_t)(NZ + 1) * (TR + 1) * 17 * 32 + (size_t)CIN0 * XZ * XY * XX) * sizeof(float);
  static_assert(ldsb <= 160 * 1024, "LDS");
  static std::atomic<uint64_t> attr_done{0};  // per instantiation, one bit per device
  if (lr_raise_dyn_lds(reinterpret_cast<const void*>(&dgrad_wgrad0_kernel<CIN0, NZ>), ldsb, attr_done) != LR_OK) return LR_ELAUNCH;
  hipLaunchKernelGGL((dgrad_wgrad0_kernel<CIN0, NZ>), dim3((unsigned)blocks), dim3(512), ldsb, st, gpre1,
                     reinterpret_cast<const float4*>(packed_w1T), reinterpret_cast<const unsigned*>(mask0), in0, bs0, in_rest, bsr, partial, d, ntiles);
  const int ncols = NTP * 16;
  hipLaunchKernelGGL(wgrad0_finish_kernel, dim3((16 * ncols + 63) / 64), dim3(1024), 0, st, partial, gw0, gb0, blocks * 8, CIN0, ncols);
  return lr_launch_status();
}

}  // namespace

// Floats of `partial` lr_conv3d_dgrad_wgrad0_f32 needs (one 16 x NTJ*16 partial per wave of every persistent block).
extern "C" int64_t lr_conv3d_dgrad_wgrad0_partial_floats(int Cin0) {
  if (Cin0 < 2 || Cin0 > 5) return LR_EUNSUPPORTED;
  return (int64_t)256 * 8 * 16 * ((27 * Cin0 + 1 + 15) / 16) * 16;
}

// gw0 (16,Cin0,3,3,3), gb0 (16) of the encoder's first block from the pre-activation gradient of the SECOND block:
//   gpre1 (B,Do,Wo,Ho,32) fp32 plain channels-last; packed_w1T = lr_conv3d_pack_weights_f32 of block 1's weight transposed
//   to (16,32,3,3,3), layout NDHWC (what lr_conv3d_dgrad_f32 takes); mask0 (B,D,W,H,4) uint8 = block 0's LR_LAYOUT_SIGN4
//   sign mask, 4-byte aligned; x0 (B,Cin0,D,W,H) fp32 NCDHW = block 0's input, Cin0 in 2..5; H % 4 == 0.
// Same results as lr_conv3d_dgrad_f32 (x_layout SIGN4) followed by lr_conv3d_wgrad_f32 up to fp32 summation order.
// ... with block 0's input in TWO buffers, as the model holds it: channel 0 (the moving image) at in0 + b*in0_batch_stride,
// channels 1..Cin0-1 (the backprojected views) at in_rest + b*rest_batch_stride + (c-1)*D*W*H (elements) — no concatenated copy
// of the moving image (0.17 ms per C3 step).  The concatenated form below is this one with in_rest = x0 + D*W*H.
extern "C" int lr_conv3d_dgrad_wgrad0_split_f32(const float* gpre1, const float* packed_w1T, const uint8_t* mask0, float slope0,
                                                const float* in0, int64_t in0_batch_stride, const float* in_rest,
                                                int64_t rest_batch_stride, float* partial, float* gw0, float* gb0, int B, int Cin0,
                                                int D, int W, int H, void* stream) {
  if (!gpre1 || !packed_w1T || !mask0 || !in0 || !in_rest || !partial || !gw0) return LR_ENULL;
  if (B < 1 || D < 1 || W < 1 || H < 1) return LR_EINVAL;
  if (Cin0 < 2 || Cin0 > 5 || (H & 3)) return LR_EUNSUPPORTED;
  const int64_t V = (int64_t)D * W * H;
  if (in0_batch_stride < V || rest_batch_stride < (int64_t)(Cin0 - 1) * V) return LR_EINVAL;
  if ((reinterpret_cast<uintptr_t>(gpre1) & 15u) || (reinterpret_cast<uintptr_t>(mask0) & 3u) || (reinterpret_cast<uintptr_t>(in0) & 7u) ||
      (reinterpret_cast<uintptr_t>(in_rest) & 7u) || (in0_batch_stride & 1) || (rest_batch_stride & 1))
    return LR_EALIGN;
  FzDims d;
  d.B = B; d.D = D; d.W = W; d.H = H;
  d.Do = (D - 1) / 2 + 1; d.Wo = (W - 1) / 2 + 1; d.Ho = (H - 1) / 2 + 1;
  int NZ = nz_of(Cin0);
  if (Cin0 <= 3 && lr_sw_int(LR_SW_FUSED_BWD_NZ, 8) == 4) NZ = 4;   // A/B aid: the 4-plane tile form for <= 3 channels too
  d.nHq = ((H + 1) / 2 + 15) / 16; d.nWq = (d.Wo + TR - 1) / TR; d.nDq = (d.Do + NZ - 1) / NZ;
  d.slope = slope0;
  if ((int64_t)10 * d.Wo * d.Ho * CG * 4 >= 0x7fffffffLL) return LR_EINVAL;       // 32-bit offsets of a gpre1 window
  if ((int64_t)Cin0 * V * 4 >= 0x7fffffffLL) return LR_EUNSUPPORTED;               // ... and of one batch element of the input
  const int64_t nt = (int64_t)B * d.nDq * d.nWq * d.nHq;
  if (nt > 0x7fffffffLL) return LR_EINVAL;
  int blocks = 256;  // one 8-wave block per CU; the partial buffer is sized for 256
  if (lr_sw_set(LR_SW_FUSED_BWD_BLOCKS)) { blocks = lr_sw_int(LR_SW_FUSED_BWD_BLOCKS, 256); if (blocks < 1 || blocks > 256) blocks = 256; }  // tuning aid
  if (nt < blocks) blocks = (int)nt;
  hipStream_t st = lr_stream(stream);
#define LR_FZ(C, Z) return launch<C, Z>(gpre1, packed_w1T, mask0, in0, in0_batch_stride, in_rest, rest_batch_stride, partial, gw0, gb0, d, (int)nt, blocks, st)
  if (Cin0 == 5) LR_FZ(5, 4);
  if (Cin0 == 4) LR_FZ(4, 4);
  if (Cin0 == 3 && NZ == 4) LR_FZ(3, 4);
  if (Cin0 == 2 && NZ == 4) LR_FZ(2, 4);
  if (Cin0 == 3) LR_FZ(3, 8);
  LR_FZ(2, 8);
#undef LR_FZ
}

extern "C" int lr_conv3d_dgrad_wgrad0_f32(const float* gpre1, const float* packed_w1T, const uint8_t* mask0, float slope0,
                                          const float* x0, float* partial, float* gw0, float* gb0, int B, int Cin0, int D,
                                          int W, int H, void* stream) {
  if (!x0) return LR_ENULL;
  const int64_t V = (int64_t)D * W * H;
  return lr_conv3d_dgrad_wgrad0_split_f32(gpre1, packed_w1T, mask0, slope0, x0, (int64_t)Cin0 * V, x0 + V, (int64_t)Cin0 * V, partial,
                                          gw0, gb0, B, Cin0, D, W, H, stream);
}
