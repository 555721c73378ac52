/*
 * liftreg_oracle.c — TEST INFRASTRUCTURE ONLY.  Plain-C CPU restatement of the
 * LiftReg hot path (uncbiag/LiftReg), one scalar loop nest per op, every
 * function citing the reference file:line it follows (paths relative to the
 * reference checkout).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product (liftreg_amd) never does.
 *
 * Parity status: PINNED.  oracle/ is checked against golden vectors produced by
 * importing the reference itself in the build container
 * (tests/golden/make_golden.py -> the .npz files beside it; tests/test_oracle_golden.py).
 *
 * Build: gcc -O2 -fopenmp -ffp-contract=off -fno-fast-math -shared -fPIC
 * (contraction off: sample coordinates must round like the reference's separate
 * ATen ops so that floor() picks the same cell).
 *
 * The sampling arithmetic restates ATen's grid_sample (third-party dependency of
 * the reference: torch, pinned 1.9.0+cu111 in requirements.txt:133; present in
 * this image as 2.10): align_corners=True un-normalise ((g+1)/2)*(size-1),
 * corner weights and per-corner zero padding as in
 * aten/src/ATen/native/GridSampler.{h,cpp} (3-D generic kernel) and
 * aten/src/ATen/native/cpu/GridSamplerKernel.cpp (2-D vectorised kernel).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define OR_OK 0
#define OR_EINVAL (-1)

static inline float unnormalize(float g, int size) { return ((g + 1.0f) / 2.0f) * (float)(size - 1); }

/* ---------------------------------------------------------------------------
 * a1  calc_relative_atten_coef — src/liftreg/utils/sdct_projection_utils.py:6-9
 */
static inline float hu_to_mu(float hu) {
  float v = hu < -1000.0f ? -1000.0f : hu; /* new_img[new_img<-1000] = -1000 */
  return ((v + 1000.0f) / 1000.0f) * 0.2f; /* (new_img+1000.)/1000.*0.2      */
}

void or_calc_relative_atten_coef(const float* hu, float* mu, int64_t n) {
  for (int64_t i = 0; i < n; ++i) mu[i] = hu_to_mu(hu[i]);
}

/* ---------------------------------------------------------------------------
 * a3  project_grid_multi — src/liftreg/utils/sdct_projection_utils.py:15-57
 * Per-ray constants (:31-41) and per-sample position (:50-56) in pixel units.
 * torch.norm(dim=3) on CPU is an FMA chain fma(z,z,fma(y,y,x*x)) (measured in the
 * build container against the reference; see tests/golden/README.md).
 */
typedef struct {
  float ihx, ihy, ihz, rc, dx;
} ray_t;

static ray_t ray_setup(int a, int b, int Rd, int Rh, const float* e, const float* sp) {
  const float px = (float)a - 0.5f * (float)Rd; /* linspace(-res_d/2, res_d/2-1, res_d) :32 */
  const float pz = (float)b - 0.5f * (float)Rh; /* :33 */
  const float ix = px + (-e[0]), iy = 0.0f + (-e[1]), iz = pz + (-e[2]); /* torch.add(I,-I0) :38 */
  const float rcp = 1.0f / iy;                                          /* 1./I[:,:,:,1:2]  :39 */
  const float d0 = (ix * rcp) * sp[0], d1 = (iy * rcp) * sp[1], d2 = (iz * rcp) * sp[2]; /* :39,:41 */
  ray_t r;
  r.dx = sqrtf(fmaf(d2, d2, fmaf(d1, d1, d0 * d0)));                   /* :41 */
  const float nrm = sqrtf(fmaf(iz, iz, fmaf(iy, iy, ix * ix)));        /* :40 */
  r.ihx = ix / nrm;
  r.ihy = iy / nrm;
  r.ihz = iz / nrm;
  r.rc = 1.0f / r.ihy; /* 1./(matmul(I,N)) :50 ; N=(0,1,0) picks the y component exactly */
  return r;
}

static void sample_pix(const ray_t* r, int j, const float* e, int D, int W, int H, float* pd,
                       float* pw, float* ph) {
  const float t = r->rc * ((float)j - e[1]); /* T = (1/Î_y) * ((P0-I0)·N) :50 (K=1 matmul = product) */
  const float x = r->ihx * t + e[0];         /* matmul(I,T) + I0 :51 */
  const float y = r->ihy * t + e[1];
  const float z = r->ihz * t + e[2];
  const float gx = (x / (float)D) * 2.0f;                               /* :54 */
  const float gy = ((y - 0.0f) / ((float)W - 1.0f)) * 2.0f + -1.0f;     /* :55 */
  const float gz = (z / (float)H) * 2.0f;                               /* :56 */
  *pd = unnormalize(gx, D); /* flip(…,[4]) :76 then grid_sample: z<->D, y<->W, x<->H */
  *pw = unnormalize(gy, W);
  *ph = unnormalize(gz, H);
}

int or_drr_sample_coords_f32(const float* poses, const float* spacing, float* pix, float* dx, int D,
                             int W, int H, int P, int Rd, int Rh) {
  for (int p = 0; p < P; ++p)
    for (int a = 0; a < Rd; ++a)
      for (int b = 0; b < Rh; ++b) {
        const int64_t idx = ((int64_t)p * Rd + a) * Rh + b;
        ray_t r = ray_setup(a, b, Rd, Rh, poses + 3 * p, spacing);
        if (dx) dx[idx] = r.dx;
        if (pix)
          for (int j = 0; j < W; ++j) {
            float* o = pix + (idx * W + j) * 3;
            sample_pix(&r, j, poses + 3 * p, D, W, H, o, o + 1, o + 2);
          }
      }
  return OR_OK;
}

/* one axis of ATen's generic 3-D trilinear: i0=floor, i1=i0+1, weights (i1-pix),(pix-i0);
 * a corner outside [lo,hi) is dropped (padding_mode='zeros'). */
typedef struct {
  int i0, i1, ok0, ok1;
  float w0, w1;
} axis_t;

static axis_t make_axis(float pix, int lo, int hi) {
  axis_t a;
  if (!(pix > (float)(lo - 1) && pix < (float)hi)) {
    a.i0 = a.i1 = lo;
    a.ok0 = a.ok1 = 0;
    a.w0 = a.w1 = 0.0f;
    return a;
  }
  const float fl = floorf(pix);
  const int i0 = (int)fl, i1 = i0 + 1;
  a.w0 = (float)i1 - pix;
  a.w1 = pix - (float)i0;
  a.ok0 = i0 >= lo;
  a.ok1 = i1 < hi;
  a.i0 = i0 < lo ? lo : i0;
  a.i1 = i1 > hi - 1 ? hi - 1 : i1;
  return a;
}

/* ---------------------------------------------------------------------------
 * a4  calculate_projection — src/liftreg/utils/sdct_projection_utils.py:59-100
 *   out = sum_w(grid_sample(I0, grids, align_corners=True)) * dx (:81) ; *= 0.1 (:85)
 * vol_slab holds rows [d0,d1) of the (D,W,H) volume; taps outside contribute 0.
 * flags: 1 = HU input (a1 folded in), 2 = axis-1 flip (tools/preprocessingDRR.py:135-136).
 */
int or_drr_forward_f32(const float* vol_slab, const float* poses, const float* spacing, float* out,
                       int D, int W, int H, int d0, int d1, int P, int Rd, int Rh, int flags) {
  if (d0 < 0 || d1 > D || d1 <= d0 || W < 2) return OR_EINVAL;
  const int hu = flags & 1, flip = flags & 2;
  const int64_t sD = (int64_t)W * H;
#pragma omp parallel for collapse(2) schedule(static)
  for (int p = 0; p < P; ++p)
    for (int a = 0; a < Rd; ++a)
      for (int b = 0; b < Rh; ++b) {
        const float* e = poses + 3 * p;
        ray_t r = ray_setup(a, b, Rd, Rh, e, spacing);
        float acc = 0.0f;
        for (int j = 0; j < W; ++j) {
          float pd, pw, ph;
          sample_pix(&r, j, e, D, W, H, &pd, &pw, &ph);
          const axis_t az = make_axis(pd, d0, d1), ay = make_axis(pw, 0, W), ax = make_axis(ph, 0, H);
          const int zz[2] = {az.i0 - d0, az.i1 - d0};
          const int yy[2] = {flip ? W - 1 - ay.i0 : ay.i0, flip ? W - 1 - ay.i1 : ay.i1};
          const int xx[2] = {ax.i0, ax.i1};
          const int okz[2] = {az.ok0, az.ok1}, oky[2] = {ay.ok0, ay.ok1}, okx[2] = {ax.ok0, ax.ok1};
          const float wz[2] = {az.w0, az.w1}, wy[2] = {ay.w0, ay.w1}, wx[2] = {ax.w0, ax.w1};
          float s = 0.0f;
          /* corner order tnw,tne,tsw,tse,bnw,bne,bsw,bse = x fastest, then y, then z */
          for (int cz = 0; cz < 2; ++cz)
            for (int cy = 0; cy < 2; ++cy)
              for (int cx = 0; cx < 2; ++cx) {
                float v = 0.0f;
                if (okz[cz] && oky[cy] && okx[cx]) {
                  v = vol_slab[(int64_t)zz[cz] * sD + (int64_t)yy[cy] * H + xx[cx]];
                  if (hu) v = hu_to_mu(v);
                }
                s = s + v * ((wx[cx] * wy[cy]) * wz[cz]);
              }
          acc = acc + s;
        }
        out[((int64_t)p * Rd + a) * Rh + b] = (acc * r.dx) * 0.1f;
      }
  return OR_OK;
}

/* ---------------------------------------------------------------------------
 * a6  backproj_grids_with_poses — src/liftreg/utils/sdct_projection_utils.py:227-250
 */
static inline float shadow_pix(float x, float e, float scale, int size) {
  float g = (x - e) * scale; /* torch.mul(grids - poses[:, :, ::2], scale) :241-242 */
  g = g + e;                 /* + poses[:, :, ::2] :242 */
  g = g / (float)size;       /* grids[:, :, c]/proj_w :247-248 */
  g = g * 2.0f;
  return unnormalize(g, size);
}

int or_backproject_coords_f32(const float* poses, float* pix, int P, int Pw, int Ph, int D, int W,
                              int H) {
  for (int p = 0; p < P; ++p) {
    const float ex = poses[3 * p], ey = poses[3 * p + 1], ez = poses[3 * p + 2];
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < W; ++j) {
        const float x = (float)i - 0.5f * (float)D; /* linspace(-d/2, d/2-1, d) :231 */
        const float y = (float)(W - 1 - j);         /* linspace(w-1, 0, w)      :232 */
        const float scale = ey / (ey - y);          /* :239 */
        for (int k = 0; k < H; ++k) {
          const float z = (float)k - 0.5f * (float)H; /* :233 */
          float* o = pix + ((((int64_t)p * D + i) * W + j) * H + k) * 2;
          o[0] = shadow_pix(x, ex, scale, Pw);
          o[1] = shadow_pix(z, ez, scale, Ph);
        }
      }
  }
  return OR_OK;
}

/* one axis of ATen's vectorised 2-D bilinear: w = x - floor(x), e = 1 - w */
typedef struct {
  int i0, i1;
  float w0, w1;
} tap_t;

static tap_t make_tap(float pix, int size) {
  tap_t t;
  if (!(pix > -1.0f && pix < (float)size)) {
    t.i0 = t.i1 = 0;
    t.w0 = t.w1 = 0.0f;
    return t;
  }
  const float fl = floorf(pix);
  const float w = pix - fl, e = 1.0f - w;
  const int i0 = (int)fl, i1 = i0 + 1;
  t.w0 = i0 >= 0 ? e : 0.0f;
  t.w1 = i1 < size ? w : 0.0f;
  t.i0 = i0 < 0 ? 0 : i0;
  t.i1 = i1 > size - 1 ? size - 1 : i1;
  return t;
}

/* ---------------------------------------------------------------------------
 * a7  backprojection sample — src/liftreg/models/LiftRegDeformSubspaceBackproj.py:85-93
 * F.grid_sample 2-D, zeros, align_corners=True; ONE geometry for the batch (:85-87).
 */
int or_backproject_f32(const float* proj, const float* poses, float* out, int B, int P, int Pw,
                       int Ph, int D, int W, int H, int d0, int d1, int64_t out_batch_stride) {
  if (d0 < 0 || d1 > D || d1 <= d0) return OR_EINVAL;
  const int Ds = d1 - d0;
#pragma omp parallel for collapse(2) schedule(static)
  for (int p = 0; p < P; ++p)
    for (int i = 0; i < Ds; ++i) {
      const float ex = poses[3 * p], ey = poses[3 * p + 1], ez = poses[3 * p + 2];
      const float x = (float)(d0 + i) - 0.5f * (float)D;
      for (int j = 0; j < W; ++j) {
        const float y = (float)(W - 1 - j);
        const float scale = ey / (ey - y);
        const tap_t ty = make_tap(shadow_pix(x, ex, scale, Pw), Pw);
        for (int k = 0; k < H; ++k) {
          const float z = (float)k - 0.5f * (float)H;
          const tap_t tx = make_tap(shadow_pix(z, ez, scale, Ph), Ph);
          const float nw = ty.w0 * tx.w0, ne = ty.w0 * tx.w1, sw = ty.w1 * tx.w0, se = ty.w1 * tx.w1;
          for (int b = 0; b < B; ++b) {
            const float* v = proj + ((int64_t)b * P + p) * Pw * Ph;
            float acc = v[(int64_t)ty.i0 * Ph + tx.i0] * nw;
            acc = acc + v[(int64_t)ty.i0 * Ph + tx.i1] * ne;
            acc = acc + v[(int64_t)ty.i1 * Ph + tx.i0] * sw;
            acc = acc + v[(int64_t)ty.i1 * Ph + tx.i1] * se;
            out[b * out_batch_stride + (((int64_t)p * Ds + i) * W + j) * H + k] = acc;
          }
        }
      }
    }
  return OR_OK;
}

/* ---------------------------------------------------------------------------
 * a8  convBlock — src/liftreg/layers/layers.py:335-372: Conv3d(k3,p1,stride,bias)+LeakyReLU
 * NCDHW in / NCDHW out, direct sum (c, kd, kh, kw order), fp32 accumulate.
 */
int or_conv3d_k3_lrelu_f32(const float* in, const float* w, const float* bias, float* out, int B,
                           int Cin, int Cout, int D, int W, int H, int stride, float slope) {
  const int Do = (D - 1) / stride + 1, Wo = (W - 1) / stride + 1, Ho = (H - 1) / stride + 1;
  const int64_t V = (int64_t)D * W * H, Vo = (int64_t)Do * Wo * Ho;
#pragma omp parallel for collapse(3) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < Cout; ++co)
      for (int z = 0; z < Do; ++z)
        for (int y = 0; y < Wo; ++y)
          for (int x = 0; x < Ho; ++x) {
            float acc = bias ? bias[co] : 0.0f;
            for (int ci = 0; ci < Cin; ++ci)
              for (int kz = 0; kz < 3; ++kz) {
                const int zi = z * stride + kz - 1;
                if (zi < 0 || zi >= D) continue;
                for (int ky = 0; ky < 3; ++ky) {
                  const int yi = y * stride + ky - 1;
                  if (yi < 0 || yi >= W) continue;
                  for (int kx = 0; kx < 3; ++kx) {
                    const int xi = x * stride + kx - 1;
                    if (xi < 0 || xi >= H) continue;
                    acc = fmaf(in[((int64_t)b * Cin + ci) * V + ((int64_t)zi * W + yi) * H + xi],
                               w[(((int64_t)co * Cin + ci) * 3 + kz) * 9 + ky * 3 + kx], acc);
                  }
                }
              }
            out[((int64_t)b * Cout + co) * Vo + ((int64_t)z * Wo + y) * Ho + x] =
                acc >= 0.0f ? acc : acc * slope;
          }
  return OR_OK;
}

/* ---------------------------------------------------------------------------
 * a9  FullyConnectBlock — src/liftreg/layers/layers.py:413-439 (slope=1 → no nonlinearity)
 */
int or_linear_lrelu_f32(const float* x, const float* w, const float* bias, float* y, int B, int K,
                        int O, float slope) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int o = 0; o < O; ++o) {
      double acc = 0.0;
      for (int k = 0; k < K; ++k) acc += (double)x[(int64_t)b * K + k] * (double)w[(int64_t)o * K + k];
      float s = (float)acc;
      if (bias) s = s + bias[o];
      y[(int64_t)b * O + o] = s >= 0.0f ? s : s * slope;
    }
  return OR_OK;
}

/* ---------------------------------------------------------------------------
 * a10 PCA reconstruction — src/liftreg/models/LiftRegDeformSubspaceBackproj.py:42-43,102
 *   disp = F.linear(coefs, pca_vectors(.T view of the (L,3V) file), pca_mean)
 */
int or_pca_reconstruct_f32(const float* coefs, const float* basis, const float* mean, float* disp,
                           int B, int L, int64_t M, int64_t ldb, int64_t disp_batch_stride) {
#pragma omp parallel for schedule(static)
  for (int64_t m = 0; m < M; ++m)
    for (int b = 0; b < B; ++b) {
      float acc = mean[m];
      for (int l = 0; l < L; ++l) acc = fmaf(coefs[(int64_t)b * L + l], basis[(int64_t)l * ldb + m], acc);
      disp[b * disp_batch_stride + m] = acc;
    }
  return OR_OK;
}

/* ---------------------------------------------------------------------------
 * a11+a12  identity add + Bilinear — src/liftreg/utils/net_utils.py:9-56 (forward_stn :26-38,
 * scaling :48-52), identity add …Backproj.py:68, mask compose …Backproj.py:54-58.
 * flags: 1 using_scale, 2 border padding, 4 nearest.
 */
static inline float tap_value(const float* img, const float* seg, int64_t off, int scale) {
  float v = img[off];
  if (seg) v = (v + 1.0f) * seg[off] - 1.0f; /* (moving+1)*moving_seg-1 */
  if (scale) v = (v + 1.0f) / 2.0f;          /* (input1 + 1) / 2         */
  return v;
}

int or_warp_trilinear_f32(const float* img, const float* seg, const float* disp, const float* id0,
                          const float* id1, const float* id2, float* phi_out, float* warped, int B,
                          int C, int D, int W, int H, int d0, int d1, int flags) {
  if (d0 < 0 || d1 > D || d1 <= d0) return OR_EINVAL;
  const int scale = flags & 1, border = flags & 2, nearest = flags & 4;
  const int Dn = d1 - d0;
  const int64_t slabV = (int64_t)Dn * W * H, V = (int64_t)D * W * H, sD = (int64_t)W * H;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < Dn; ++i)
      for (int j = 0; j < W; ++j)
        for (int k = 0; k < H; ++k) {
          const int64_t voff = ((int64_t)i * W + j) * H + k;
          const float* dp = disp + (int64_t)b * 3 * slabV + voff;
          float p0 = dp[0], p1 = dp[slabV], p2 = dp[2 * slabV];
          if (id0) { /* deform_field = disp_field + id_transform */
            p0 = p0 + id0[i];
            p1 = p1 + id1[j];
            p2 = p2 + id2[k];
          }
          if (phi_out) {
            float* pp = phi_out + (int64_t)b * 3 * slabV + voff;
            pp[0] = p0; pp[slabV] = p1; pp[2 * slabV] = p2;
          }
          /* channel reorder (2,1,0): grid x = phi[2] <-> H, y = phi[1] <-> W, z = phi[0] <-> D */
          float ph = unnormalize(p2, H), pw = unnormalize(p1, W), pd = unnormalize(p0, D);
          if (border) { /* clip_coordinates */
            ph = fminf((float)(H - 1), fmaxf(ph, 0.0f));
            pw = fminf((float)(W - 1), fmaxf(pw, 0.0f));
            pd = fminf((float)(D - 1), fmaxf(pd, 0.0f));
          }
          for (int c = 0; c < C; ++c) {
            const float* im = img + ((int64_t)b * C + c) * V;
            const float* sg = seg ? seg + ((int64_t)b * C + c) * V : NULL;
            float res;
            if (nearest) {
              const float rx = nearbyintf(ph), ry = nearbyintf(pw), rz = nearbyintf(pd);
              const int ok = rx >= 0.0f && rx < (float)H && ry >= 0.0f && ry < (float)W && rz >= 0.0f &&
                             rz < (float)D;
              res = ok ? tap_value(im, sg, (int64_t)rz * sD + (int64_t)ry * H + (int64_t)rx, scale) : 0.0f;
            } else {
              const axis_t ax = make_axis(ph, 0, H), ay = make_axis(pw, 0, W), az = make_axis(pd, 0, D);
              const int zz[2] = {az.i0, az.i1}, yy[2] = {ay.i0, ay.i1}, xx[2] = {ax.i0, ax.i1};
              const int okz[2] = {az.ok0, az.ok1}, oky[2] = {ay.ok0, ay.ok1}, okx[2] = {ax.ok0, ax.ok1};
              const float wz[2] = {az.w0, az.w1}, wy[2] = {ay.w0, ay.w1}, wx[2] = {ax.w0, ax.w1};
              float s = 0.0f;
              for (int cz = 0; cz < 2; ++cz)
                for (int cy = 0; cy < 2; ++cy)
                  for (int cx = 0; cx < 2; ++cx) {
                    float v = 0.0f;
                    if (okz[cz] && oky[cy] && okx[cx])
                      v = tap_value(im, sg, (int64_t)zz[cz] * sD + (int64_t)yy[cy] * H + xx[cx], scale);
                    s = s + v * ((wx[cx] * wy[cy]) * wz[cz]);
                  }
              res = s;
            }
            if (scale) res = res * 2.0f - 1.0f; /* output * 2 - 1 */
            warped[((int64_t)b * C + c) * slabV + voff] = res;
          }
        }
  return OR_OK;
}

void or_mask_compose_f32(const float* img, const float* seg, float* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = (img[i] + 1.0f) * seg[i] - 1.0f; /* …Backproj.py:57-58 */
}

/* ---------------------------------------------------------------------------
 * a13 NCCLoss — src/liftreg/layers/losses.py:14-29 (variant 0, configured) and
 * src/liftreg/layers/layers.py:238-255 (variant 1, squared).  Two-pass, as the reference:
 * means first, then centred products; fp64 accumulation.
 */
int or_ncc_loss_f32(const float* x, const float* y, float* loss, float* ncc_rows, int R, int64_t N,
                    int n_batch, int variant) {
  double total = 0.0;
  for (int r = 0; r < R; ++r) {
    const float* xr = x + (int64_t)r * N;
    const float* yr = y + (int64_t)r * N;
    double sx = 0, sy = 0;
    for (int64_t i = 0; i < N; ++i) { sx += xr[i]; sy += yr[i]; }
    const double mx = sx / (double)N, my = sy / (double)N;
    const double eps = variant == 0 ? 1e-10 : 0.0;
    double sab = 0, saa = 0, sbb = 0;
    for (int64_t i = 0; i < N; ++i) {
      const double a = (double)xr[i] - mx + eps, b = (double)yr[i] - my + eps;
      sab += a * b; saa += a * a; sbb += b * b;
    }
    sab /= (double)N; saa /= (double)N; sbb /= (double)N;
    const double v = variant == 0 ? sab / sqrt(saa * sbb) : (sab * sab) / (saa * sbb + 1e-12);
    if (ncc_rows) ncc_rows[r] = (float)v;
    total += v;
  }
  (void)n_batch;
  *loss = (float)(1.0 - total / (double)R);
  return OR_OK;
}

/* raw moments, for the sharded-NCC property tests */
void or_ncc_moments_f32(const float* x, const float* y, double* moments, int R, int64_t N) {
  for (int r = 0; r < R; ++r) {
    double s[5] = {0, 0, 0, 0, 0};
    for (int64_t i = 0; i < N; ++i) {
      const double a = x[(int64_t)r * N + i], b = y[(int64_t)r * N + i];
      s[0] += a; s[1] += b; s[2] += a * b; s[3] += a * a; s[4] += b * b;
    }
    memcpy(moments + 5 * r, s, sizeof s);
  }
}


/* Test aid for the projector's reciprocal division (liftreg_amd/csrc/drr_forward.hip: div_by): how many floats x with bit patterns in
 * [lo_bits, hi_bits) — both signs — have x / d != fma(fma(-q, d, x), r, q), q = x r, r = RN(1 / d).  0 = the three-operation
 * sequence IS the IEEE division on that range. */
#include <string.h>
int64_t or_fastdiv_mismatches(float d, uint32_t lo_bits, uint32_t hi_bits) {
  const float r = 1.0f / d;
  int64_t bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
  for (int64_t u = (int64_t)lo_bits; u < (int64_t)hi_bits; ++u) {
    for (int sgn = 0; sgn < 2; ++sgn) {
      const uint32_t bits = (uint32_t)u | (sgn ? 0x80000000u : 0u);
      float x;
      memcpy(&x, &bits, 4);
      const float q = x * r;
      const float e = fmaf(-q, d, x);
      const float qc = fmaf(e, r, q);
      const float want = x / d;
      if (memcmp(&qc, &want, 4) != 0) ++bad;
    }
  }
  return bad;
}
