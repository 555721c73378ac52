// conv3d_rows.hip — the encoder's stride-2 blocks (Cin = 16 | 32, parity-split channels-last input) as ONE persistent
// kernel: Winograd F(2,2) along W over the row walk of conv3d_cl_rows_kernel, with every weight fragment in LDS.
//
//   * the stride-2 conv along W is a 2-tap conv on the even input rows plus a 1-tap conv on the odd ones; for an output-row
//     pair (2p, 2p+1) and even input rows x0, x1, x2:  M1 = (x0 - x1) w0,  M2 = x1 (w0 + w2),  M3 = (x2 - x1) w2,
//     y(2p) = M1 + M2 + w1 o0,  y(2p+1) = M2 + M3 + w1 o1  — 10 row uses per 4-row tile instead of 12, coefficients +-1;
//   * one 768-thread block per CU (12 waves = 3 per SIMD) copies the 36 * CB * NT KiB of fragments (27 taps + the nine
//     (w0 + w2) sums, lr_conv3d_pack_weights_f32) into LDS ONCE and then walks tiles: the register-weights version of
//     the same walk (round 2, not kept) spent 24 of its 51 vector-memory instructions per plane on
//     fragments, needed 212 registers (two waves per SIMD) and kept the matrix pipe 74 % busy (PMC);
//   * a wave owns one output plane dz of a tile (4 rows x 16 voxels x 16*NT couts); the 12 waves of a block take 12
//     consecutive planes, even waves walk their three input planes top-down and odd waves bottom-up, so the input plane
//     two neighbouring waves share is read by both at the same time (L2-miss traffic 15.3 -> 12.0 GB on block 1).
// fp32 MFMA (v_mfma_f32_16x16x4_f32) throughout; the summation tree differs from the direct fmaf chain (conv3d.hip,
// LIFTREG_CONV_DIRECT=1 — the oracle's bits), the arithmetic type does not.
//
// Replaces (reference file:line)  src/liftreg/layers/layers.py:335-372 convBlock for blocks 1..5 of
//   src/liftreg/models/LiftRegDeformSubspaceBackproj.py:29-33,95-100.
#include "lr_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lrelu(float v, float slope) { return v >= 0.0f ? v : v * slope; }

struct RowsDims {
  int B, D, W, H, Do, Wo, Ho, Cout;
  int nHq, nWq, nDq;
  long long out_bs;   // output elements between batch elements (>= Cout*Do*Wo*Ho)
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// One 16-voxel x 16-cout accumulator tile -> memory (lane l: couts nt*16 + (l>>4)*4 + {0..3} of voxel l&15) through
// UNCONDITIONAL bounds-checked buffer stores: `res` covers output plane (b, dz) (channels-last layouts) or batch element
// b (NCDHW); a lane outside the row / column range gets an out-of-range offset and the hardware drops its store.  No
// branch around a store: the compiler can then count the stores in flight instead of draining them (vmcnt is shared
// with the loads) before the next tile's first rows are consumed.
template <int OUTL>
__device__ __forceinline__ void store_rows_tile(const f32x4& acc, const __amdgpu_buffer_rsrc_t res, const RowsDims& d, int dz,
                                                int wo, int ho, int nt, int lane, float slope) {
  constexpr int out_layout = OUTL;  // compile time: a run-time layout switch would put every store behind a branch
  const int c0 = nt * 16 + (lane >> 4) * 4;
  float4 v;
  v.x = lrelu(acc[0], slope); v.y = lrelu(acc[1], slope); v.z = lrelu(acc[2], slope); v.w = lrelu(acc[3], slope);
  const bool inside = wo < d.Wo && ho < d.Ho;
  if (out_layout == LR_LAYOUT_NCDHW) {
    const int vo = d.Do * d.Wo * d.Ho;
    const unsigned off = inside ? (unsigned)((c0 * vo + (dz * d.Wo + wo) * d.Ho + ho) * 4) : 0x80000000u;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v.x), res, off, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v.y), res, off, vo * 4, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v.z), res, off, vo * 8, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v.w), res, off, vo * 12, 0);
    return;
  }
  unsigned off;
  if (out_layout == LR_LAYOUT_NDHWC) {
    off = (unsigned)(((wo * d.Ho + ho) * d.Cout + c0) * 4);
  } else {  // NDHWC_HPS: row = [channel block of 16][parity][Ho/2][16 floats]
    const int hp = (ho & 1) * (d.Ho >> 1) + (ho >> 1);
    off = (unsigned)((wo * d.Ho * d.Cout + ((c0 >> 4) * d.Ho + hp) * 16 + (c0 & 15)) * 4);
  }
  if (!inside) off = 0x80000000u;
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), res, off, 0, 0);
}

constexpr int MT = 4;        // output rows of a tile = two Winograd pairs
constexpr int NWAVE = 12;    // waves of a block: three tiles of four planes
constexpr int NR = 2 * MT + 1;

struct TileDesc {  // wave-uniform
  const float* wbase;  // window origin (-1,-1,-1) of the wave's plane of the tile
  unsigned okmask;     // bit tz*NR+r: input row (zi0+tz, yw0+r) exists
  int b, dz, wo0, hq;
  bool valid;
};

template <int NT, int CB /* Cin = 16 * CB */, int OUTL>
__global__ __launch_bounds__(NWAVE * 64, 1) void conv3d_rows_wlds_kernel(const float* __restrict__ in,
                                                                          const float4* __restrict__ wp,
                                                                          const float* __restrict__ bias,
                                                                          float* __restrict__ out, RowsDims d, float slope,
                                                                          int ntiles, int xmap, int z_phase) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wl[];
  constexpr int NF4 = 36 * CB * NT * 64;  // float4s = the packed buffer, same order
  {
    // all of a thread's fragment loads in flight at once (a copy loop would pay one memory round trip per iteration:
    // 12 in a row for the 144 KB of a 32 -> 32 block, which is most of a small layer's run time)
    constexpr int K = NF4 / (NWAVE * 64);
    static_assert(K * NWAVE * 64 == NF4, "whole fragments per thread");
    float4 t[K];
#pragma unroll
    for (int k = 0; k < K; ++k) t[k] = wp[threadIdx.x + k * NWAVE * 64];
#pragma unroll
    for (int k = 0; k < K; ++k) reinterpret_cast<float4*>(wl)[threadIdx.x + k * NWAVE * 64] = t[k];
  }
  __syncthreads();  // the only barrier: from here on the waves are independent

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int slot = wave >> 2, dzl = wave & 3;
  const bool zrev = !((wave ^ z_phase) & 1);  // by the parity of the GLOBAL output plane (dz & 1 = wave & 1)
  const int dD = __builtin_amdgcn_readfirstlane(d.D), dW = __builtin_amdgcn_readfirstlane(d.W),
            dH = __builtin_amdgcn_readfirstlane(d.H), dHo = __builtin_amdgcn_readfirstlane(d.Ho);
  constexpr int CIN = 16 * CB;
  constexpr int NQ = 3 * CB * NR;
  constexpr int PF = 2, RN = PF + 2;  // rows loaded ahead; ring = rows q-2 (an open difference), q .. q+PF
  const int col = lane & 15, kq = lane >> 4;
  const unsigned lvoff = (unsigned)(col * 64 + kq * 16);
  const int half_h = (dH + 1) >> 1;
  constexpr unsigned OOR = 0x80000000u;
  // LDS addresses of this lane's fragments: plane tz0 of the wave's walk order, direct taps | (w0 + w2) sums
  unsigned wbd[3], wbs[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int tz = zrev ? 2 - t : t;
    wbd[t] = (unsigned)lane * 16u + (unsigned)(tz * 9 * CB * NT * 1024);
    wbs[t] = (unsigned)lane * 16u + (unsigned)((27 + tz * 3) * CB * NT * 1024);
  }
  f32x4 bv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    bv[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[nt][r] = bias[nt * 16 + kq * 4 + r];
    }
  }

  // Work order.  xmap (the big blocks): the chip's 8 XCDs each sweep their own share of UNITS = (batch element, group of
  // nb = gridDim/8 neighbouring (wq, hq) columns, triple of depth tiles), depth fastest — block i runs on XCD i % 8, so at
  // any time the nb blocks of an XCD hold neighbouring columns of the same 12 planes and the halo rows / voxels / planes
  // they share meet in that XCD's L2 (a plain strided order left 14.6 GB of L2 misses on block 1 against 13.4 GB).
  // Otherwise (few tiles): triple g = blockIdx + t * gridDim of the (b, wq, hq, dq) order, dq fastest.
  const int ntrip = (ntiles + 2) / 3;
  const int ncol = d.nHq * d.nWq, nb = gridDim.x >> 3, nzt = (d.nDq + 2) / 3;
  const int ncg = xmap ? (ncol + nb - 1) / nb : 1;
  int it0 = blockIdx.x, it1 = ntrip, istep = gridDim.x;
  if (xmap) {
    const int64_t U = (int64_t)d.B * ncg * nzt;
    it0 = (int)(U * (blockIdx.x & 7) / 8);
    it1 = (int)(U * ((blockIdx.x & 7) + 1) / 8);
    istep = 1;
  }
  auto make_desc = [&](int it) __attribute__((always_inline)) -> TileDesc {
    TileDesc t;
    t.valid = false; t.wbase = in; t.okmask = 0u; t.b = 0; t.dz = 0; t.wo0 = 0; t.hq = 0;
    int b, wq, hq, dq;
    if (xmap) {
      const int zt = it % nzt, cg = (it / nzt) % ncg, c = cg * nb + (blockIdx.x >> 3);
      b = it / nzt / ncg;
      dq = zt * 3 + slot;
      if (c >= ncol || dq >= d.nDq) return t;
      hq = c % d.nHq;
      wq = c / d.nHq;
    } else {
      const int tile = it * 3 + slot;
      if (tile >= ntiles) return t;
      dq = tile % d.nDq;
      int t2 = tile / d.nDq;
      hq = t2 % d.nHq;
      t2 /= d.nHq;
      wq = t2 % d.nWq;
      b = t2 / d.nWq;
    }
    const int dz = dq * 4 + dzl;
    if (dz >= d.Do) return t;
    t.valid = true; t.b = b; t.dz = dz; t.wo0 = wq * MT; t.hq = hq;
    // a row outside the volume reads through the zero-length resource, a voxel outside its row gets an out-of-range
    // offset: both return 0 = the conv's padding, no branch
    const int zi0 = dz * 2 - 1, yw0 = t.wo0 * 2 - 1, xh0 = hq * 16 - 1;
    t.wbase = in + (int64_t)b * dD * dW * dH * CIN + ((int64_t)zi0 * dW + yw0) * dH * CIN + (int64_t)xh0 * 16;
    unsigned ym = 0u;
#pragma unroll
    for (int r = 0; r < NR; ++r) ym |= ((unsigned)(yw0 + r >= 0) & (unsigned)(yw0 + r < dW)) << r;
#pragma unroll
    for (int tz = 0; tz < 3; ++tz) t.okmask |= (zi0 + tz >= 0 && zi0 + tz < dD) ? ym << (tz * NR) : 0u;
    return t;
  };
  auto next_valid = [&](int it, TileDesc& t) __attribute__((always_inline)) -> int {  // first valid item at or after `it`
    for (; it < it1; it += istep) {
      t = make_desc(it);
      if (t.valid) return it;
    }
    t.valid = false; t.okmask = 0u; t.wbase = in;
    return it1;
  };
  auto vx_of = [&](int hq, unsigned (&vx)[3]) __attribute__((always_inline)) {
    const int ho = hq * 16 + col, xi0 = ho * 2 - 1;
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) vx[t3] = lvoff | ((ho < dHo && xi0 + t3 >= 0 && xi0 + t3 < dH) ? 0u : OOR);
  };
  // row q = (tz0 * CB + cb) * NR + r of a tile's walk -> three 16-byte loads per lane (tx = 0,1,2)
  auto load_row = [&](const TileDesc& t, const unsigned (&vx)[3], int q, float4 (&a)[3]) __attribute__((always_inline)) {
    const int pl = q / NR, r = q - pl * NR, tz0 = pl / CB, cb = pl - tz0 * CB;
    const int tz = zrev ? 2 - tz0 : tz0;
    const bool ok = (t.okmask >> (tz * NR + r)) & 1u;  // wave-uniform: outside -> the zero-length resource
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(t.wbase), (short)0, ok ? 0x7fffffff : 0, 0x00020000);
    const unsigned row = (unsigned)((tz * dW + r) * dH * CIN + cb * 2 * half_h * 16);
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      // tx=1: even half, index ho; tx=0: odd half, index ho-1; tx=2: odd half, index ho (xh0 carries the -1)
      const int xs = tx == 1 ? 1 : half_h + (tx >> 1);
      a[tx] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, vx[tx], (row + (unsigned)xs * 16u) * 4u, 0));
    }
  };
  // Fragment set u of plane pl = tz0 * CB + cb: u = 0: w(ty=0), 1: w(ty=0) + w(ty=2), 2: w(ty=2), 3: w(ty=1).
  // The walk is a fixed sequence of UNITS (one row operand x one fragment set -> one accumulator, 3 tx steps of 4*NT
  // MFMAs); the fragments of a step are read from LDS one step ahead (wc = current, wn = next), the two cout tiles
  // alternate on the matrix pipe so no MFMA waits for the one before it.
  float4 wc[NT], wn[NT];
  auto ldw = [&](float4 (&w)[NT], int pl, int u, int tx) __attribute__((always_inline)) {
    const int tz0 = pl / CB, cb = pl - tz0 * CB;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const unsigned off = u == 1 ? wbs[tz0] + (unsigned)(((tx * CB + cb) * NT + nt) * 1024)
                                  : wbd[tz0] + (unsigned)(((((u == 0 ? 0 : u == 2 ? 2 : 1) * 3 + tx) * CB + cb) * NT + nt) * 1024);
      w[nt] = *reinterpret_cast<const float4*>(wl + off);
    }
  };
  auto unit = [&](f32x4 (&a3)[NT], const float4 (&a)[3], int pl, int u, int npl, int nu) __attribute__((always_inline)) {
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      if (tx < 2) ldw(wn, pl, u, tx + 1);
      else ldw(wn, npl, nu, 0);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) a3[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[nt].x, a[tx].x, a3[nt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) a3[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[nt].y, a[tx].y, a3[nt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) a3[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[nt].z, a[tx].z, a3[nt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) a3[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[nt].w, a[tx].w, a3[nt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wc[nt] = wn[nt];
      __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);     // the next step's fragment reads first ...
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0); // ... then this step's MFMAs
    }
  };

  // A tile's results leave at the top of the NEXT loop iteration (and after the loop for the last one): every path into the
  // loop header then carries the same outstanding requests — rows 0 and 1 of the coming tile, nothing after them — and the
  // compiler's wait for those rows allows the stores issued after them to stay in flight (vmcnt is shared and in order;
  // with the stores at the loop's end, the join with the store-free entry path made every tile drain its stores first).
  auto store_all = [&](const TileDesc& t, const f32x4 (&yo)[MT][NT]) __attribute__((always_inline)) {
    const int ho = t.hq * 16 + col;
    constexpr bool cl = OUTL != LR_LAYOUT_NCDHW;
    const int64_t pstride = cl ? (int64_t)d.Wo * d.Ho * d.Cout : 0;
    float* pbase = out + (int64_t)t.b * d.out_bs + (cl ? (int64_t)t.dz * pstride : 0);
    const unsigned bytes = !t.valid ? 0u : cl ? (unsigned)(pstride * 4) : (unsigned)((int64_t)d.Cout * d.Do * d.Wo * d.Ho * 4);
    const __amdgpu_buffer_rsrc_t res = __builtin_amdgcn_make_buffer_rsrc(pbase, (short)0, (int)bytes, 0x00020000);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) store_rows_tile<OUTL>(yo[mt][nt], res, d, t.dz, t.wo0 + mt, ho, nt, lane, slope);
  };

  TileDesc cur, prev;
  int it = next_valid(it0, cur);
  if (!cur.valid) return;
  prev = cur;
  prev.valid = false;  // nothing to store yet: a zero-length resource drops the first iteration's stores
  f32x4 yo[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) yo[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned vxc[3];
  vx_of(cur.hq, vxc);
  float4 ring[RN][3];
  float4 dd[3];
#pragma unroll
  for (int q = 0; q < PF; ++q) load_row(cur, vxc, q, ring[q]);
  ldw(wc, 0, 3, 0);
  while (cur.valid) {
    // the tile after this one (invalid after the last: okmask = 0, its two rows read as zeros through the null resource)
    TileDesc nxt;
    it = next_valid(it + istep, nxt);
    unsigned vxn[3];
    vx_of(nxt.hq, vxn);
    store_all(prev, yo);
    f32x4 aA[2][NT], aB[2][NT], aC[2][NT];  // per pair: M1 + w1*o0 | M2 | M3 + w1*o1
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        aA[p][nt] = bv[nt];
        aC[p][nt] = bv[nt];
        aB[p][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int pl = q / NR, r = q % NR;
      if (r >= 2 && !(r & 1)) {  // the difference this even row closes, before row q+PF takes row q-2's registers
        const bool lead = (r == 2 || r == 6);  // (x0 - x1) of the pair starting here; else (x2 - x1)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          const float4 o = ring[(q + PF) % RN][tx], c = ring[q % RN][tx];
          dd[tx] = lead ? make_float4(o.x - c.x, o.y - c.y, o.z - c.z, o.w - c.w)
                        : make_float4(c.x - o.x, c.y - o.y, c.z - o.z, c.w - o.w);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (q + PF < NQ) load_row(cur, vxc, q + PF, ring[(q + PF) % RN]);
      if (q == NQ - 1) {
        // ring slots 0 and 1 are free from here (their rows are spent, the last difference is in dd): the next tile's
        // rows 0 and 1 are requested before this tile's results are stored
        load_row(nxt, vxn, 0, ring[0]);
        load_row(nxt, vxn, 1, ring[1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      // units of a plane in walk order: (row 1, u3) (2, u0) (2, u1) (3, u3) (4, u2) (5, u3) (6, u0) (6, u1) (7, u3) (8, u2)
      const int npl = pl + 1 < 3 * CB ? pl + 1 : 0;  // after the last unit: the next tile's first fragments
      if (r == 1) unit(aA[0], ring[q % RN], pl, 3, pl, 0);
      if (r == 2) { unit(aA[0], dd, pl, 0, pl, 1); unit(aB[0], ring[q % RN], pl, 1, pl, 3); }
      if (r == 3) unit(aC[0], ring[q % RN], pl, 3, pl, 2);
      if (r == 4) unit(aC[0], dd, pl, 2, pl, 3);
      if (r == 5) unit(aA[1], ring[q % RN], pl, 3, pl, 0);
      if (r == 6) { unit(aA[1], dd, pl, 0, pl, 1); unit(aB[1], ring[q % RN], pl, 1, pl, 3); }
      if (r == 7) unit(aC[1], ring[q % RN], pl, 3, pl, 2);
      if (r == 8) unit(aC[1], dd, pl, 2, npl, 3);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          yo[2 * p][nt][e] = aA[p][nt][e] + aB[p][nt][e];
          yo[2 * p + 1][nt][e] = aB[p][nt][e] + aC[p][nt][e];
        }
    prev = cur;
    cur = nxt;
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) vxc[t3] = vxn[t3];
  }
  store_all(prev, yo);
}

int cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}

template <int NT, int CB, int OUTL>
int launch_l(const float* in, const float* packed_w, const float* bias, float* out, const RowsDims& d, float slope, int ntiles,
             int z_phase, hipStream_t st) {
  const size_t lds = (size_t)36 * CB * NT * 1024;
  static std::atomic<uint64_t> attr_done{0};  // per instantiation, one bit per device
  if (lr_raise_dyn_lds(reinterpret_cast<const void*>(conv3d_rows_wlds_kernel<NT, CB, OUTL>), lds, attr_done) != LR_OK) return LR_ELAUNCH;
  const int ntrip = (ntiles + 2) / 3;
  int blocks = cu_count();
  blocks = lr_sw_int(LR_SW_CONV_ROWS_BLOCKS, blocks);  // tuning aid
  if (blocks > ntrip) blocks = ntrip;
  if (blocks < 1) blocks = 1;
  // the XCD-aware order needs whole column groups per XCD and enough work to fill them
  int xmap = (blocks % 8 == 0) && (d.nHq * d.nWq >= blocks / 8) && ((int64_t)d.B * ((d.nDq + 2) / 3) * ((d.nHq * d.nWq + blocks / 8 - 1) / (blocks / 8)) >= 16);
  if (lr_sw_set(LR_SW_CONV_ROWS_XMAP)) xmap = xmap && lr_sw_int(LR_SW_CONV_ROWS_XMAP, 1) != 0;  // A/B aid
  hipLaunchKernelGGL((conv3d_rows_wlds_kernel<NT, CB, OUTL>), dim3((unsigned)blocks), dim3(NWAVE * 64), lds, st, in,
                     reinterpret_cast<const float4*>(packed_w), bias, out, d, slope, ntiles, xmap, z_phase);
  return lr_launch_status();
}

template <int NT, int CB>
int launch(const float* in, const float* packed_w, const float* bias, float* out, const RowsDims& d, int out_layout,
           float slope, int ntiles, int z_phase, hipStream_t st) {
  if (out_layout == LR_LAYOUT_NDHWC_HPS) return launch_l<NT, CB, LR_LAYOUT_NDHWC_HPS>(in, packed_w, bias, out, d, slope, ntiles, z_phase, st);
  if (out_layout == LR_LAYOUT_NDHWC) return launch_l<NT, CB, LR_LAYOUT_NDHWC>(in, packed_w, bias, out, d, slope, ntiles, z_phase, st);
  return launch_l<NT, CB, LR_LAYOUT_NCDHW>(in, packed_w, bias, out, d, slope, ntiles, z_phase, st);
}

}  // namespace

// Stride-2 block on a parity-split channels-last input (B,D,W,H,Cin) with Cin = 16 | 32, Cout = 16 | 32, fp32 output in
// NDHWC / NDHWC_HPS / NCDHW.  LR_EUNSUPPORTED for anything else (the caller falls back to conv3d.hip's kernels).
int lr_internal_conv_rows_wlds(const float* in, const float* packed_w, const float* bias, float* out, int B, int Cin,
                               int Cout, int D, int W, int H, int out_layout, float slope, int z_phase, long long out_bs,
                               hipStream_t st) {
  if ((Cin != 16 && Cin != 32) || (Cout != 16 && Cout != 32) || (H & 1)) return LR_EUNSUPPORTED;
  // Small planes (blocks 3..5 of the encoder: 32^2 outputs per plane and less) stay with conv3d.hip's direct kernels: a few
  // hundred tiles cannot amortise the per-block fragment staging and the serial 27/54-row walk (measured at C3: 0.14 /
  // 0.07 / 0.07 ms here against 0.12 / 0.03 / 0.03 ms).  The rule never looks at the DEPTH, so a z-slab of a volume takes
  // the same kernel as the whole volume.
  // Round 5: large batches amortise the staging on smaller planes too — the reference's shipped configuration (B = 30): 80^3 ->
  // 40^3 0.98 -> 0.84 ms, 40^3 -> 20^3 0.169 -> 0.148 ms here; 20^3 -> 10^3 0.033 -> 0.074 (stays direct).
  // CONSEQUENCE (documented in INTEGRATION.md, "batch-dependent kernel choice"): on planes of 400 .. 4095 outputs the choice
  // depends on the BATCH SIZE of the call, and the two kernels round differently (Winograd F(2,2) rows vs the direct fmaf chain,
  // both within 1e-6 of an fp64 convolution): a sample's encoder output can differ in the last bits between a full batch and a
  // smaller last batch (160^3: Winograd for B >= 7 on the 40^2 planes), and a caller that splits a batch into groups (the
  // sharded forward's exchange form) may land on the other kernel than the unsharded model.  LIFTREG_CONV_DIRECT=1 or
  // LIFTREG_CONV_ROWS_ALWAYS=1 pins one kernel for every batch size (reproducible evaluation).
  {
    const int64_t plane = (int64_t)((W - 1) / 2 + 1) * ((H - 1) / 2 + 1);
    if (plane < 4096 && !(plane >= 400 && (int64_t)B * plane >= 10000) && !lr_sw_set(LR_SW_CONV_ROWS_ALWAYS)) return LR_EUNSUPPORTED;
  }
  if (out_layout != LR_LAYOUT_NDHWC && out_layout != LR_LAYOUT_NDHWC_HPS && out_layout != LR_LAYOUT_NCDHW)
    return LR_EUNSUPPORTED;
  RowsDims d;
  d.B = B; d.D = D; d.W = W; d.H = H; d.Cout = Cout;
  d.Do = (D - 1) / 2 + 1; d.Wo = (W - 1) / 2 + 1; d.Ho = (H - 1) / 2 + 1;
  d.out_bs = out_bs ? out_bs : (long long)Cout * d.Do * d.Wo * d.Ho;
  d.nHq = (d.Ho + 15) / 16; d.nWq = (d.Wo + MT - 1) / MT; d.nDq = (d.Do + 3) / 4;
  const int64_t nt64 = (int64_t)B * d.nDq * d.nWq * d.nHq;
  if (nt64 > 0x3fffffffLL) return LR_EINVAL;
  const int ntiles = (int)nt64;
  // one buffer resource per output plane (channels-last) or batch element (NCDHW): 31-bit byte offsets
  if ((int64_t)d.Wo * d.Ho * Cout * 4 >= 0x7fffffffLL) return LR_EUNSUPPORTED;
  if (out_layout == LR_LAYOUT_NCDHW && (int64_t)Cout * d.Do * d.Wo * d.Ho * 4 >= 0x7fffffffLL) return LR_EUNSUPPORTED;
  if (Cin == 16 && Cout == 16) return launch<1, 1>(in, packed_w, bias, out, d, out_layout, slope, ntiles, z_phase, st);
  if (Cin == 16) return launch<2, 1>(in, packed_w, bias, out, d, out_layout, slope, ntiles, z_phase, st);
  if (Cout == 16) return launch<1, 2>(in, packed_w, bias, out, d, out_layout, slope, ntiles, z_phase, st);
  return launch<2, 2>(in, packed_w, bias, out, d, out_layout, slope, ntiles, z_phase, st);
}
