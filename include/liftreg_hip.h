/*
 * liftreg_hip.h — C ABI of libliftreg_hip.so (gfx950 / MI355X).
 *
 * The reference (uncbiag/LiftReg) has no FFI of its own: every op on its hot
 * path is an ATen call from Python.  Each entry point below replaces exactly
 * one of those call sites (cited as reference file:line, paths relative to
 * the reference checkout).  The library is what a maintainer binds from the
 * reference's Python with ctypes; liftreg_amd/_hip.py is
 * that binding.
 *
 * Conventions
 *  - every pointer marked "dev" is a device pointer owned by the caller; the
 *    library never allocates, frees or synchronises;
 *  - pointers marked "host" are read synchronously before the call returns;
 *  - all launches are asynchronous on `stream` (a hipStream_t passed as
 *    void*; NULL = the default stream);
 *  - volumes are (D,W,H) = (axial, coronal, sagittal) with H fastest, batches
 *    are NCDHW exactly as the reference's tensors;
 *  - return value: LR_OK (0) or a negative LR_E* code; nothing is thrown
 *    across the ABI; lr_strerror() turns a code into text;
 *  - fp32 arithmetic that decides an INDEX (sample coordinates, floor) is done
 *    with the reference's op order, no FMA contraction, IEEE divide.
 */
#ifndef LIFTREG_HIP_H
#define LIFTREG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LR_OK 0
#define LR_EINVAL (-1)   /* bad shape / size / flag                         */
#define LR_ENULL (-2)    /* required pointer is NULL                        */
#define LR_EUNSUPPORTED (-3) /* combination not built (e.g. Cout not 16/32) */
#define LR_ELAUNCH (-4)  /* hipLaunch reported an error                     */
#define LR_EALIGN (-5)   /* pointer/extent not aligned as the kernel needs  */

#define LR_MAX_VIEWS 32  /* emitter poses travel in kernel arguments        */

const char* lr_strerror(int code);
/* ABI version; bumped when a signature or a documented behaviour changes.  2 (round 5): the LIFTREG_* switches below are read
 * ONCE per process (version 1 read them at every call) — call lr_reload_switches() after changing the environment; the
 * register-light kernels (lr_backproject_light_f32, lr_pca_warp_light_f32) and the CU-masked stream calls (lr_stream_*) of
 * version 1 are gone; lr_drr_forward_batch_f32 and lr_conv3d_dgrad_wgrad0_split_f32 are new. */
int lr_abi_version(void);
/* Name of the code object's target ("gfx950"). */
const char* lr_target_arch(void);

/* ------------------------------------------------------------------------
 * Run-time switches.  The library reads the following environment variables ONCE per process (at its first launch,
 * thread-safe) — a launch never calls getenv().  They select between kernels that compute the same function (tests
 * cross-check them; the "oracle's bits" variants are the direct fmaf chains of oracle/liftreg_oracle.c) or tune launch
 * geometry; none of them is needed in production.  lr_reload_switches() re-reads them (tests / A-B tools that flip a
 * switch between two calls in one process; liftreg_amd/_hip.py: reload_switches()); it returns the number of switches.
 * lr_switch_name(i), 0 <= i < that number, is the variable's name.  Timing-only diagnostics that give WRONG results exist
 * only in builds with -DLR_DIAG_ABLATIONS / -DLR_C0CL_ABLATIONS / -DLR_C0S_ABLATIONS / -DLR_C01_ABL=... (never shipped).
 *   LIFTREG_HIP_DEBUG              print the HIP error text of a failed launch to stderr
 *   LIFTREG_CONV_DIRECT            stride-2 fp32 blocks: the direct row walk = the oracle's fmaf chain bit for bit (default: Winograd F(2,2) rows kernel)
 *   LIFTREG_CONV0_DIRECT           first fp32 block: the direct sweep = the oracle's fmaf chain (default: Winograd F(2,3) along H)
 *   LIFTREG_CONV_TAPMAJOR          stride-2 blocks: the tap-major kernel instead of the row kernels (same bits as the direct walk)
 *   LIFTREG_CONV_ROWS_ALWAYS       persistent Winograd rows kernel also on planes below 64 x 64 outputs (tests)
 *   LIFTREG_CONV0_BF16_CL          bf16 first block: the all-channels brick kernel also for <= 3 channels
 *   LIFTREG_CONV0_BF16_PASSES      bf16 first block: the channel-pass kernel (round-2 path)
 *   LIFTREG_WARP_GENERAL           trilinear warp / its gradient: the general kernels instead of the fast ones (same bits)
 *   LIFTREG_DRR_GENERAL            projector: the general kernel instead of the fast one (same bits)
 *   LIFTREG_REG_NOMARCH            displacement regulariser: the generic kernels instead of the marching ones
 *   LIFTREG_DGRAD_OLD              data gradient: the per-tile kernels instead of the persistent weights-in-LDS ones (tests cross-check both)
 *   LIFTREG_WGRAD_SPLIT=0          block 1's weight gradient on the fp32 MFMA (default since round 5: exact 3-way bf16 splits)
 *   LIFTREG_WGRAD_ROWS             weight gradient: bricks of 1 instead of 2 rows
 *   LIFTREG_WGRAD0_COPIES          bf16 training: first block's weight gradient through the three-copies kernel
 *   LIFTREG_CONV0_BLOCKS           persistent blocks of the fp32 first-block kernels
 *   LIFTREG_CONV0_SPLIT_BLOCKS     persistent blocks of conv0_split_f32.hip
 *   LIFTREG_CONV0_SPLIT_CHUNKS     z chunks per column of conv0_split_f32.hip (tests: chunk boundaries)
 *   LIFTREG_CONV0_CL_BLOCKS        persistent blocks of conv0_cl_bf16.hip
 *   LIFTREG_C0CL_SHAPE             brick shape of conv0_cl_bf16.hip
 *   LIFTREG_C0CL_CHUNKS            z chunks per column of conv0_cl_bf16.hip
 *   LIFTREG_CONV_LDS               dynamic LDS bytes that cap the resident blocks of the channels-last conv kernels
 *   LIFTREG_CONV_ROWS_MT1_BELOW    block count below which the 32->32 blocks take one output row per wave
 *   LIFTREG_CONV_ROWS_BLOCKS       persistent blocks of conv3d_rows.hip
 *   LIFTREG_CONV_ROWS_XMAP         0: plain strided tile order instead of the XCD-aware one
 *   LIFTREG_BF16_MT                output rows per tile of the bf16 row kernels (4 | 8)
 *   LIFTREG_PAIR01_BLOCKS          persistent blocks of the fused pair kernel (default: one per CU)
 *   LIFTREG_PAIR01_DENSE           0: three-channel pair kernel with block 0's padded K (24 MFMAs per tile) instead of the dense 17 (A/B aid)
 *   LIFTREG_BF16_NO_MARCH          bf16 16->32 block: the row kernel instead of the z-marching one (A/B aid)
 *   LIFTREG_BF16_MARCH_TY8         bf16 16->32 z-march: columns of 8 x 16 outputs (512 threads, one block per CU; A/B aid)
 *   LIFTREG_BF16_MARCH_ZC          output planes per z chunk of the bf16 z-marching kernel (tests: chunk boundaries)
 *   LIFTREG_DGRAD_BLOCKS           persistent blocks of the data-gradient kernels
 *   LIFTREG_FUSED_BWD_BLOCKS       persistent blocks of the fused dgrad1 + wgrad0 kernel
 *   LIFTREG_REG_BWD_BLOCKS         block cap of the regulariser's gradient kernel
 *   LIFTREG_FUSED_BWD_NZ           4: the fused dgrad1 + wgrad0 kernel's 4-plane tiles for <= 3 input channels too (default 8; A/B aid, same results up to summation order)
 *   LIFTREG_BP_TOUCH               0: no streaming pass over the views in front of the tiled backprojection (A/B aid; same bits)
 *   LIFTREG_BP_CHUNK               batch elements per block of the tiled backprojection (0 = the whole batch; same bits)
 *   LIFTREG_BP_JP                  planes a backprojection block works on side by side (1 | 2 | 4; same bits)
 */
int lr_reload_switches(void);
const char* lr_switch_name(int id);

/* ------------------------------------------------------------------------
 * K1  DRR cone-beam forward projector.
 * Replaces project_grid_multi + F.grid_sample(3D) + sum + *dx*0.1:
 *   src/liftreg/utils/sdct_projection_utils.py:15-57 (grid, dx)
 *   src/liftreg/utils/sdct_projection_utils.py:59-100 (calculate_projection, :81,:85)
 * and optionally folds calc_relative_atten_coef (:6-9) and the axis-1 flip of
 * tools/preprocessingDRR.py:135-136 into the volume load.
 *
 * out[p,a,b] = 0.1 * dx[p,a,b] * sum_{j<W} trilinear0(vol, sample(p,a,b,j))
 *
 * vol_slab : dev, (d1-d0, W, H) — rows d0..d1-1 of the full (D,W,H) volume.
 *            Taps outside [d0,d1) contribute 0, so partial DRRs of disjoint
 *            slabs SUM to the full DRR (z-slab sharding).
 *            Unsharded: d0=0, d1=D.
 * poses    : host, (P,3) fp32 emitter positions (x,y,z) in voxel units.
 * spacing  : host, 3 floats (mm).
 * flags    : LR_DRR_HU_INPUT  vol holds HU; mu=(max(HU,-1000)+1000)/1000*0.2 on load
 *            LR_DRR_FLIP_W    read plane W-1-j where the reference reads j
 * nseg     : 1,2,4 or 8 — each ray's W planes are split into nseg runs summed
 *            by separate lanes (deterministic order); 0 = library default.
 * out      : dev, (P,Rd,Rh) fp32.
 * LR_DRR_HU_INPUT expects finite values (a tap outside the volume counts 0 through its weight, not its address).
 */
#define LR_DRR_HU_INPUT 1
#define LR_DRR_FLIP_W 2
int lr_drr_forward_f32(const float* vol_slab, const float* poses, const float* spacing,
                       float* out, int D, int W, int H, int d0, int d1,
                       int P, int Rd, int Rh, int flags, int nseg, void* stream);
/* B volumes (slabs) of ONE geometry in one launch: vol_slabs + b * vol_batch_stride (elements, >= (d1-d0)*W*H) ->
 * out + b * P*Rd*Rh.  Same results as B calls of lr_drr_forward_f32 (tools/preprocessingDRR.py:123-154 projects source and
 * target of every case with one geometry; the dataset's cases are independent).  B <= 65535. */
int lr_drr_forward_batch_f32(const float* vol_slabs, int64_t vol_batch_stride, const float* poses,
                             const float* spacing, float* out, int B, int D, int W, int H, int d0, int d1,
                             int P, int Rd, int Rh, int flags, int nseg, void* stream);

/* Parity / API compatibility (forward_grids_with_poses, sdct_projection_utils.py:252-265):
 * the sample coordinates the projector uses and dx.
 * normalized=0: pixel units after ATen's align_corners=True un-normalise;
 * normalized=1: the reference's [-1,1] grid values before flip (x/D*2, y/(W-1)*2-1, z/H*2).
 * pix: dev (P,Rd,Rh,W,3) ordered (d,w,h); dx: dev (P,Rd,Rh). Either may be NULL. */
/* calc_relative_atten_coef (sdct_projection_utils.py:6-9): mu = ((max(HU,-1000)+1000)/1000)*0.2, one pass over the
 * volume.  Cheaper than LR_DRR_HU_INPUT (which converts per tap) whenever a volume is projected from more than a
 * handful of rays per voxel; bit-identical results. */
int lr_hu_to_mu_f32(const float* hu, float* mu, int64_t n, void* stream);
int lr_drr_sample_coords_f32(const float* poses, const float* spacing, float* pix, float* dx,
                             int D, int W, int H, int P, int Rd, int Rh, int normalized,
                             void* stream);

/* ------------------------------------------------------------------------
 * K2  Backprojection (voxel-driven gather of the 2D views).
 * Replaces backproj_grids_with_poses + F.grid_sample(2D):
 *   src/liftreg/utils/sdct_projection_utils.py:227-250
 *   src/liftreg/models/LiftRegDeformSubspaceBackproj.py:85-93
 *
 * out[b,p,i,j,k] = bilinear0(proj[b,p], shadow of voxel (i,j,k) for emitter p)
 *
 * proj  : dev (B,P,Pw,Ph) fp32
 * poses : host (P,3) fp32 — ONE geometry for the whole batch, as the reference
 *         caches the grid of batch element 0 (…Backproj.py:85-87).
 * out   : dev; element (b,p,i,j,k) at out[b*out_batch_stride + ((p*Ds+i)*W+j)*H+k]
 *         with Ds=d1-d0 and i in [0,Ds) standing for row d0+i. out_batch_stride
 *         lets the caller write straight into channels 1..P of the (B,P+1,D,W,H)
 *         encoder input (the torch.cat of …Backproj.py:95-98).
 */
int lr_backproject_f32(const float* proj, const float* poses, float* out,
                       int B, int P, int Pw, int Ph, int D, int W, int H,
                       int d0, int d1, int64_t out_batch_stride, void* stream);

/* Parity / API compatibility (backproj_grids_with_poses): detector coordinates of every
 * voxel shadow. normalized=0: pixel units; normalized=1: the reference's [-1,1] grid.
 * pix: dev (P,D,W,H,2) ordered (Pw axis, Ph axis). */
int lr_backproject_coords_f32(const float* poses, float* pix, int P, int Pw, int Ph,
                              int D, int W, int H, int normalized, void* stream);

/* Parity / API compatibility (backproj_grids, sdct_projection_utils.py:179-202 — the pose-less variant; only the
 * reference's dead RegNet2D3D calls it): the reference mixes a float64 pose array with float32 linspaces, so the
 * grid is FLOAT64 and built as scale*g + trans (:194-197).  poses: host (P,3) DOUBLE; grid: dev (P,2,D,W,H) double,
 * channel 0 = the Ph-axis coordinate, channel 1 = the Pw-axis coordinate (after the reference's flip(1), :201). */
int lr_backproject_coords_poseless_f64(const double* poses, double* grid, int P, int Pw, int Ph, int D, int W, int H,
                                       void* stream);

/* ------------------------------------------------------------------------
 * K3  Conv3d(k=3, pad=1, stride 1|2, bias) + LeakyReLU, implicit GEMM on
 * v_mfma_f32_16x16x4_f32.  Replaces convBlock:
 *   src/liftreg/layers/layers.py:335-372 as wired at …Backproj.py:29-33,95-100
 *
 * lr_conv3d_pack_weights_f32: weight (Cout,Cin,3,3,3) -> MFMA B-operand order.
 *   packed needs lr_conv3d_packed_floats(Cin,Cout,in_layout) floats (channels-last: the 27 taps + 9 Winograd sums
 *   w(ty=0)+w(ty=2) per (tz,tx); NCDHW with Cin<=3: the direct fragments + the F(2,3) ones).
 * lr_conv3d_k3_lrelu_f32:
 *   in  : dev; LR_LAYOUT_NCDHW (B,Cin,D,W,H) or LR_LAYOUT_NDHWC (B,D,W,H,Cin)
 *   out : dev; (B,Cout,Do,Wo,Ho) or (B,Do,Wo,Ho,Cout); Xo = (X-1)/stride+1
 *   negative_slope: LeakyReLU slope; pass 1.0f for "no nonlinearity".
 *   Supported: Cout in {16,32}; Cin any for NCDHW input, Cin%4==0 for NDHWC.
 */
#define LR_LAYOUT_NCDHW 0
#define LR_LAYOUT_NDHWC 1
/* Channels-last with every (b,d,w) row stored as [C/16][parity of h][H/2][16 floats] (even voxels of a
 * 16-channel block first, then its odd voxels): a private layout between a block and a following
 * stride-2 block, whose tap loads it makes contiguous runs.  H even, C % 16 == 0. */
#define LR_LAYOUT_NDHWC_HPS 2
/* bf16 storage (BASELINE configs C4/C5: "bf16 convs"): channels-last (B,D,W,H,C) of 2-byte bfloat16, plain or
 * with every row parity-split along H as [parity of h][H/2][C].  lr_conv3d_k3_lrelu_f32 accepts them as
 * out_layout (fp32 compute, output rounded to nearest-even bf16); lr_conv3d_k3_lrelu_bf16 takes them as input. */
#define LR_LAYOUT_BF16_NDHWC 3
#define LR_LAYOUT_BF16_NDHWC_HPS 4
/* fp32 NCDHW whose values the forward rounded to bf16 on the way into the MFMA (the saved input of
 * lr_conv3d_first_bf16): accepted as x_layout by lr_conv3d_wgrad_f32, which rounds the same way while staging. */
#define LR_LAYOUT_NCDHW_RBF16 5
/* Not an activation layout: the LeakyReLU SIGN MASK of a block output, (B,D,W,H,C/4) uint8, bit r of byte q = "channel
 * 4q+r > 0" — written by lr_conv3d_k3_lrelu_mask_f32 and accepted as x_layout by lr_conv3d_dgrad_f32 (x_saved then points
 * to the mask): the data gradient needs only the signs of the producer's output, C/4 bytes per voxel instead of 4*C. */
#define LR_LAYOUT_SIGN4 6
int lr_conv3d_k3_lrelu_mask_f32(const float* in, const float* packed_w, const float* bias, float* out, uint8_t* mask_out,
                                int B, int Cin, int Cout, int D, int W, int H, int stride, int in_layout, int out_layout,
                                float negative_slope, void* stream);
int64_t lr_conv3d_packed_floats(int Cin, int Cout, int in_layout);
int lr_conv3d_pack_weights_f32(const float* weight, float* packed, int Cin, int Cout,
                               int in_layout, void* stream);
int lr_conv3d_k3_lrelu_f32(const float* in, const float* packed_w, const float* bias, float* out,
                           int B, int Cin, int Cout, int D, int W, int H, int stride,
                           int in_layout, int out_layout, float negative_slope, void* stream);
/* Arithmetic of the stride-2 blocks on a parity-split input (in_layout LR_LAYOUT_NDHWC_HPS, Cin in {16,32}; blocks 1..5
 * of the encoder): Winograd F(2,2) along W in fp32 on the same MFMA (conv3d_rows.hip) — 10 row products per 4 output rows
 * instead of 12, coefficients +-1; results agree with the direct fmaf chain (the CPU oracle) to ~3e-6 of the activation
 * scale and are NOT bit-identical to it.  The first block (NCDHW input, Cin <= 3, H % 4 == 0) likewise uses F(2,3) along
 * H.  Environment LIFTREG_CONV_DIRECT=1 / LIFTREG_CONV0_DIRECT=1 select the direct kernels (the oracle's bits).
 *
 * lr_conv3d_k3_lrelu_zphase_f32 = lr_conv3d_k3_lrelu_f32 for a z-SLAB of a larger volume (the sharded model,
 * liftreg_amd/parallel.py): z_phase in {0,1} = parity of the GLOBAL output plane that local output plane 0 is.  The
 * Winograd rows kernel orders a plane's three input planes by that parity, so a slab gets, plane for plane, the bits of
 * the unsharded launch.  Kernels without such an order ignore it. */
int lr_conv3d_k3_lrelu_zphase_f32(const float* in, const float* packed_w, const float* bias, float* out,
                                  int B, int Cin, int Cout, int D, int W, int H, int stride,
                                  int in_layout, int out_layout, float negative_slope, int z_phase, void* stream);
/* ... writing into a STRIDED batch: output element (b, ...) lives at out + b*out_batch_stride + its dense offset inside one
 * batch element; out_batch_stride >= Cout*Do*Wo*Ho in elements of the output type (0 = dense).  The sharded model
 * (liftreg_amd/parallel.py; replaces the per-layer call of src/liftreg/layers/layers.py:365-369 under z-slab sharding,
 * SURVEY 8e) keeps every activation in per-sample halo-padded buffers [filler | halo | slab]: with the stride ONE launch
 * covers the whole batch instead of one launch per sample.  Same kernels, same bits as the dense entry points. */
int lr_conv3d_k3_lrelu_obs_f32(const float* in, const float* packed_w, const float* bias, float* out,
                               int B, int Cin, int Cout, int D, int W, int H, int stride,
                               int in_layout, int out_layout, float negative_slope, int z_phase,
                               int64_t out_batch_stride, void* stream);

/* Training backward of the encoder's first two blocks in ONE kernel (conv3d_bwd_fused.hip): the data gradient of block 1
 * (16 <- 32 channels, stride 2), the LeakyReLU mask of block 0 and the weight / bias gradient of block 0 — the (B,D,W,H,16)
 * pre-activation gradient of block 0 (8.6 GB at 256^3 x 8) is never written: nothing but block 0's weight gradient reads it.
 *   gpre1 (B,Do,Wo,Ho,32) fp32 plain channels-last = pre-activation gradient of block 1; packed_w1T =
 *   lr_conv3d_pack_weights_f32 of block 1's weight transposed to (16,32,3,3,3), layout NDHWC (as lr_conv3d_dgrad_f32 takes);
 *   mask0 (B,D,W,H,4) uint8 = block 0's LR_LAYOUT_SIGN4 mask (4-byte aligned); x0 (B,Cin0,D,W,H) fp32 = block 0's input,
 *   Cin0 in 2..5 (4, 5: tiles of 4 quotient planes, two waves per plane — the x0 window of the 8-plane tile does not fit the
 *   LDS; 5 = the reference's shipped 4-view configuration, cur_task_setting.json:56); H % 4 == 0; partial:
 *   lr_conv3d_dgrad_wgrad0_partial_floats(Cin0) floats of scratch.
 * Results: gw0 (16,Cin0,3,3,3), gb0 (16) = lr_conv3d_dgrad_f32(x_layout SIGN4) + lr_conv3d_wgrad_f32 up to fp32 summation
 * order.  Replaces autograd of layers.py:365-369 for blocks 0/1 (RegistrationNet.py:401). */
int64_t lr_conv3d_dgrad_wgrad0_partial_floats(int Cin0);
int lr_conv3d_dgrad_wgrad0_f32(const float* gpre1, const float* packed_w1T, const uint8_t* mask0, float slope0,
                               const float* x0, float* partial, float* gw0, float* gb0, int B, int Cin0, int D, int W,
                               int H, void* stream);
/* The same with block 0's input in the two buffers the model holds (…Backproj.py:95-98 concatenates them): channel 0 (the
 * moving image) at in0 + b*in0_batch_stride, channels 1..Cin0-1 (the backprojected views) at in_rest + b*rest_batch_stride +
 * (c-1)*D*W*H (elements; strides even, pointers 8-byte aligned) — no concatenated copy of the moving image. */
int lr_conv3d_dgrad_wgrad0_split_f32(const float* gpre1, const float* packed_w1T, const uint8_t* mask0, float slope0,
                                     const float* in0, int64_t in0_batch_stride, const float* in_rest,
                                     int64_t rest_batch_stride, float* partial, float* gw0, float* gb0, int B, int Cin0,
                                     int D, int W, int H, void* stream);

/* ------------------------------------------------------------------------
 * K4  Linear (+ optional LeakyReLU) for the small-batch FC head.
 * Replaces FullyConnectBlock: src/liftreg/layers/layers.py:413-439
 * (…Backproj.py:34-39).  y[b,o] = act(sum_k x[b,k]*w[o,k] + bias[o]); B <= 32.
 */
int lr_linear_lrelu_f32(const float* x, const float* w, const float* bias, float* y,
                        int B, int K, int O, float negative_slope, void* stream);

/* ------------------------------------------------------------------------
 * K5  PCA reconstruction of the displacement field (streaming skinny GEMM).
 * Replaces F.linear(coefs, pca_vectors, pca_mean): …Backproj.py:42-43,102
 *   disp[b,m] = sum_l coefs[b,l]*basis[l*ldb+m] + mean[m],  m in [0,M)
 * basis is the (L,3V) C-contiguous pca_vectors.npy (ldb=3V) or a slab of it.
 * B <= 32.  Fast path: M, ldb, stride multiples of 4 and 16-byte aligned pointers (16-byte streaming loads);
 * anything else (3V is a multiple of 4 only when the voxel count is) runs a one-element-per-thread kernel.
 */
int lr_pca_reconstruct_f32(const float* coefs, const float* basis, const float* mean, float* disp,
                           int B, int L, int64_t M, int64_t ldb, int64_t disp_batch_stride,
                           void* stream);

/* ------------------------------------------------------------------------
 * K6+K7  Spatial-transformer warp.  Replaces Bilinear.forward and the
 * identity-map add: src/liftreg/utils/net_utils.py:9-56, 59-87;
 * …Backproj.py:54-58 (mask compose), :68-69.
 *
 *  phi      = disp + id               (id from three per-axis tables)
 *  warped   = 2*trilinear((img'+1)/2, phi[(2,1,0)]) - 1   (using_scale)
 *  img'     = seg ? (img+1)*seg-1 : img
 *
 * img   : dev (B,C,Ds_src..)  full source volume (B,C,D,W,H) — always whole.
 * disp  : dev (B,3,Dn,W,H) with Dn=d1-d0 rows d0..d1-1 (slab) of the field.
 * id0/id1/id2 : dev, the identity map along D (Dn entries for rows d0..),
 *         W and H (host-built exactly as identity_map does); NULL,NULL,NULL
 *         means `disp` already IS phi.
 * seg   : dev (B,C,D,W,H) or NULL.
 * phi_out : dev (B,3,Dn,W,H) or NULL.
 * warped  : dev (B,C,Dn,W,H).
 * flags : LR_WARP_USING_SCALE, LR_WARP_BORDER (padding_mode='border',
 *         default zeros), LR_WARP_NEAREST (mode='nearest').
 */
#define LR_WARP_USING_SCALE 1
#define LR_WARP_BORDER 2
#define LR_WARP_NEAREST 4
int lr_warp_trilinear_f32(const float* img, const float* seg, const float* disp,
                          const float* id0, const float* id1, const float* id2,
                          float* phi_out, float* warped,
                          int B, int C, int D, int W, int H, int d0, int d1,
                          int flags, void* stream);

/* target_cp = (img+1)*seg-1 (…Backproj.py:57-58) */
int lr_mask_compose_f32(const float* img, const float* seg, float* out, int64_t n, void* stream);

/* ------------------------------------------------------------------------
 * K8  NCC similarity.  Replaces NCCLoss: src/liftreg/layers/losses.py:14-29
 * (configured) and the squared per-channel variant src/liftreg/layers/layers.py:238-255.
 *
 * lr_ncc_moments_f32: one pass over x,y (R rows of N elements) accumulating
 *   per row {sum x, sum y, sum xy, sum xx, sum yy} in fp64.
 *   partial : dev workspace, R*nblk*5 doubles;  moments: dev, R*5 doubles.
 *   Moments of disjoint slabs ADD (all-reduce them for z-slab sharding).
 * lr_ncc_loss_from_moments: loss (1 float) and per-row ncc (R floats).
 *   n_total = elements per row over ALL shards.
 *   variant LR_NCC_CONFIGURED: 1 - mean_r(cov/sqrt(varx*vary)) with the +1e-10 shifts;
 *   variant LR_NCC_SQUARED   : 1 - sum_r(cov^2/(varx*vary+1e-12))/n_batch/C (rows=B*C).
 */
#define LR_NCC_CONFIGURED 0
#define LR_NCC_SQUARED 1
int lr_ncc_moments_f32(const float* x, const float* y, double* partial, double* moments,
                       int R, int64_t N, int nblk, void* stream);
int lr_ncc_loss_from_moments(const double* moments, float* loss, float* ncc_rows,
                             int R, int64_t n_total, int n_batch, int variant, void* stream);

/* ------------------------------------------------------------------------
 * a16 (regulariser part)  mean over (B,D,W,H) of sum_{c,axis} (d_axis disp_c)^2.
 * Replaces compute_reg_loss: src/liftreg/losses/SubspaceLoss.py:51-67 (mermaid FD_torch(spacing*2),
 * spacing = 1/(shape-1)).  PARITY UNPINNED: the stencil is mermaid's (absent); assumed central
 * differences with linearly extrapolated faces.
 * disp: dev (B,3,D,W,H); partial: dev workspace B*nblk doubles; out: dev 1 float.
 */
int lr_disp_reg_f32(const float* disp, double* partial, float* out, int B, int D, int W, int H,
                    int nblk, void* stream);
/* The same regulariser evaluated in COEFFICIENT space (training, subspace model): the field is affine in the PCA
 * coefficients (src/liftreg/models/LiftRegDeformSubspaceBackproj.py:102: disp = coefs . basis^T + mean) and the
 * regulariser (src/liftreg/losses/SubspaceLoss.py:51-67) is a quadratic form of the field, so
 *     R = r0 + (1/B) sum_b (2 lin.c_b + c_b^T gram c_b),   dR/dc_b = (2 lin + 2 gram c_b) / B
 * with gram[k][k'] = q(basis_k, basis_k'), lin[k] = q(mean, basis_k), r0 = q(mean, mean) for q = the bilinear form
 * of lr_disp_reg_f32 (computed once per basis with lr_disp_reg_bwd_f32 + lr_pca_bwd_coef_f32: ops.subspace_reg_gram).
 * coefs: dev (B,L) fp32; gram: dev (L,L) fp64; lin: dev (L) fp64; r0: dev 1 fp64; out: dev 1 float;
 * gcoefs: dev (B,L) fp32 = dR/dcoefs, nullable. */
int lr_subspace_reg_f32(const float* coefs, const double* gram, const double* lin, const double* r0, float* out,
                        float* gcoefs, int B, int L, void* stream);

/* ========================================================================
 * Backward kernels of the training step (the reference gets these from ATen autograd at
 * RegistrationNet.py:401 `losses["total_loss"].backward()`).
 * ------------------------------------------------------------------------
 * d loss/d x of the NCC similarity (x = warped; the target has no gradient).  `moments` (R,5) as produced
 * by lr_ncc_moments_f32 (all-reduced when sharded), `gout` = dev pointer to the upstream scalar gradient. */
int lr_ncc_bwd_f32(const float* x, const float* y, const double* moments, const float* gout, float* gx,
                   int R, int64_t N, int64_t n_total, int variant, void* stream);
/* d/d disp of lr_warp_trilinear_f32 (bilinear mode; the moving image has no gradient).  Same arguments as
 * the forward; gwarped (B,C,Dn,W,H) in, gdisp (B,3,Dn,W,H) out.  flags: USING_SCALE, BORDER. */
int lr_warp_bwd_disp_f32(const float* img, const float* seg, const float* disp, const float* id0,
                         const float* id1, const float* id2, const float* gwarped, float* gdisp, int B,
                         int C, int D, int W, int H, int d0, int d1, int flags, void* stream);
/* Same, plus gadd (B,3,Dn,W,H): gdisp = (d warped / d disp)·gwarped + gadd.  The displacement field feeds the warp AND
 * the regulariser (losses/SubspaceLoss.py:63); this folds the sum autograd forms for the two paths into the warp's
 * gradient kernel (one pass over 3V floats per sample less).  gdisp and gadd are distinct buffers. */
int lr_warp_bwd_disp_acc_f32(const float* img, const float* seg, const float* disp, const float* id0,
                             const float* id1, const float* id2, const float* gwarped, const float* gadd,
                             float* gdisp, int B, int C, int D, int W, int H, int d0, int d1, int flags,
                             void* stream);
/* The similarity's gradient THROUGH ITS MOMENTS (training; replaces the pass of lr_ncc_bwd_f32 over both volumes):
 * lr_ncc_bwd_moments: gmoments (R,5) fp64 = d loss / d (sum x, sum y, sum xy, sum x^2, sum y^2) for the loss of
 * lr_ncc_loss_from_moments (layers/losses.py:14-29; layers.py:238-255), gout = dev pointer to the upstream scalar.
 * lr_warp_bwd_disp_ncc_f32: lr_warp_bwd_disp[_acc]_f32 for single-channel images with the gradient of `warped` formed on
 * the fly, gw_i = gm0 + gm2 * target_i + 2 gm3 * warped_i (the chain rule through the moments), instead of read from a
 * tensor; gadd nullable; LR_EUNSUPPORTED where the vectorised kernel does not apply (H % 4, alignment): materialise the
 * gradient with lr_ncc_bwd_f32 then. */
int lr_ncc_bwd_moments(const double* moments, const float* gout, double* gmoments, int R, int64_t n_total,
                       int variant, void* stream);
int lr_warp_bwd_disp_ncc_f32(const float* img, const float* disp, const float* id0, const float* id1,
                             const float* id2, const float* warped, const float* target, const double* gmoments,
                             const float* gadd, float* gdisp, int B, int D, int W, int H, int d0, int d1,
                             int flags, void* stream);
/* d/d coefs of lr_pca_reconstruct_f32: gcoefs (B,L) = gdisp (B,M) · basis^T.  B <= 64: above 8 rows the launch carries
 * ceil(B/8) row chunks per (m-range, l-group), dealt to one XCD so that the basis leaves HBM once (16-byte aligned inputs,
 * M, ldb, gdisp_batch_stride multiples of 4 — LR_EUNSUPPORTED otherwise: call in chunks of 8 rows); the bits of chunk-by-chunk
 * calls.  partial: dev workspace nblk*B*L floats. */
int lr_pca_bwd_coef_f32(const float* gdisp, const float* basis, float* partial, float* gcoefs, int B,
                        int L, int64_t M, int64_t ldb, int64_t gdisp_batch_stride, int nblk, void* stream);
/* Backward of lr_linear_lrelu_f32: y = its (post-activation) output, gy = upstream gradient.
 * gx (B,K) and/or (gw (O,K), gb (O)) may be NULL to skip. */
int lr_linear_bwd_f32(const float* x, const float* w, const float* y, const float* gy, float* gx,
                      float* gw, float* gb, int B, int K, int O, float negative_slope, void* stream);
/* Backward of lr_conv3d_k3_lrelu_f32 (three steps):
 *  1. lr_lrelu_bwd_f32: gpre = gy * (y > 0 ? 1 : slope) written as plain NDHWC (B,D,W,H,C), C in {16,32};
 *     gy / y (the block's saved output) may be in any LR_LAYOUT_*; gb (C) = sum over voxels (optional;
 *     gb_partial: nblk*C floats of workspace).  D,W,H = the block's OUTPUT size.
 *  2. lr_conv3d_dgrad_f32 (stride-2 blocks): gx (B,D,W,H,Cx) NDHWC from gpre (B,Do,Wo,Ho,Cg) NDHWC and
 *     packed_wT = lr_conv3d_pack_weights_f32 of the weight TRANSPOSED to (Cin,Cout,3,3,3), layout NDHWC.
 *     D,W,H = the block's INPUT size; Cx = Cin in {16,32}.  x_saved (nullable): the block's saved input,
 *     i.e. the PRODUCER block's LeakyReLU output, in x_layout (NDHWC | NDHWC_HPS, or LR_LAYOUT_SIGN4: its sign mask
 *     from lr_conv3d_k3_lrelu_mask_f32 — one byte per channel quad instead of 16); when given, the result
 *     is multiplied by (x_saved > 0 ? 1 : negative_slope) in the epilogue — it is then the producer's gpre
 *     and step 1 is skipped for the producer (write it as plain NDHWC: gx_layout = LR_LAYOUT_NDHWC).
 *  3. lr_conv3d_wgrad_f32: gw (Cout,Cin,3,3,3) from the block's saved input x (any layout) and gpre;
 *     gb (Cout, nullable) = sum of gpre over voxels (the bias gradient; free on the fast paths — a spare
 *     MFMA column multiplies by ones).  partial: lr_conv3d_wgrad_partial_floats(...) floats of workspace;
 *     nblk persistent blocks. */
int lr_lrelu_bwd_f32(const float* gy, int gy_layout, const float* y, int y_layout, float* gpre,
                     float* gb_partial, float* gb, int B, int C, int D, int W, int H,
                     float negative_slope, int nblk, void* stream);
int lr_conv3d_dgrad_f32(const float* gpre, const float* packed_wT, float* gx, int B, int Cg, int Cx,
                        int D, int W, int H, int stride, int gx_layout /* NDHWC | NDHWC_HPS */,
                        const float* x_saved, int x_layout, float negative_slope, void* stream);
/* Gradient of lr_disp_reg_f32 w.r.t. disp (gout = dev pointer to the upstream scalar gradient). */
int lr_disp_reg_bwd_f32(const float* disp, const float* gout, float* gdisp, int B, int D, int W, int H,
                        void* stream);
int64_t lr_conv3d_wgrad_partial_floats(int Cin, int Cout, int x_layout, int nblk);
int lr_conv3d_wgrad_f32(const float* x, int x_layout, const float* gpre, float* partial, float* gw,
                        float* gb, int B, int Cin, int Cout, int D, int W, int H, int stride, int nblk,
                        void* stream);

/* PCA with the basis stored as bf16 (L, ldb): an opt-in storage format that halves the bytes of the two HBM-bound
 * basis passes (model option "pca_dtype": "bf16"); everything else (coefficients, mean, accumulate, disp) fp32. */
int lr_pca_reconstruct_bf16basis_f32(const float* coefs, const void* basis_bf16, const float* mean, float* disp,
                                     int B, int L, int64_t M, int64_t ldb, int64_t disp_batch_stride, void* stream);
int lr_pca_bwd_coef_bf16basis_f32(const float* gdisp, const void* basis_bf16, float* partial, float* gcoefs, int B,
                                  int L, int64_t M, int64_t ldb, int64_t gdisp_batch_stride, int nblk, void* stream);

/* The encoder's first block on cat([moving, backprojected views], dim=1) (…Backproj.py:95-98) WITHOUT the
 * concatenation: channel 0 is read from in0 (B,1,D,W,H), channels 1..Cin-1 from in_rest (B,Cin-1,D,W,H), both NCDHW
 * fp32; packed_w as for lr_conv3d_k3_lrelu_f32 (stride 1).  Same kernel and bits as the concatenated input.
 * Cin in {2,3}, H % 4 == 0, 16-byte aligned inputs — otherwise LR_EUNSUPPORTED and the caller concatenates. */
int lr_conv3d_first_split_f32(const float* in0, const float* in_rest, const float* packed_w, const float* bias,
                              float* out, int B, int Cin, int Cout, int D, int W, int H, int out_layout,
                              float negative_slope, void* stream);
/* ... for the sharded model (liftreg_amd/parallel.py): in0 is a z-slab VIEW of the whole, replicated moving volume (batch
 * element b at in0 + b*in0_batch_stride) and the output a strided batch (lr_conv3d_k3_lrelu_obs_f32): no copy of the
 * moving image, one launch for the whole batch.  Strides in elements, 0 = dense. */
int lr_conv3d_first_split_obs_f32(const float* in0, int64_t in0_batch_stride, const float* in_rest, const float* packed_w,
                                  const float* bias, float* out, int B, int Cin, int Cout, int D, int W, int H,
                                  int out_layout, float negative_slope, int64_t out_batch_stride, void* stream);

/* Encoder blocks 0 AND 1 in one kernel (conv01_fused.hip): Conv3d(Cin -> 16, 3x3x3, stride 1, pad 1) + LeakyReLU(slope0)
 * followed by Conv3d(16 -> 32, 3x3x3, stride 2, pad 1) + LeakyReLU(slope1) — src/liftreg/layers/layers.py:365-369 twice, as
 * wired at src/liftreg/models/LiftRegDeformSubspaceBackproj.py:95-100 for encoders[0] and encoders[1].  The 16-channel
 * activation between the blocks never reaches memory.  fp32 in, fp32 out; both stages run on the bf16 MFMA with EXACT
 * three-way bf16 splits of their fp32 operands (six exact partial products, fp32 accumulation: a direct convolution with
 * fp32-class error, tests/test_gpu_conv01_fused.py).
 *   in0      : dev, channel 0 (the moving image), batch element b at in0 + b*in0_batch_stride (elements; 0 = D*W*H)
 *   in_rest  : dev, channels 1..Cin-1 planar, batch element b at in_rest + b*rest_batch_stride (0 = (Cin-1)*D*W*H);
 *              the concatenated (B,Cin,D,W,H) tensor x is (x, Cin*V, x + V, Cin*V)
 *   packed   : dev, lr_conv3d_pair01_packed_floats() floats written by lr_conv3d_pair01_pack_f32 from the two
 *              (Cout,Cin,3,3,3) weights
 *   out      : dev, (B,Do,Wo,Ho,32) in out_layout LR_LAYOUT_NDHWC or LR_LAYOUT_NDHWC_HPS, Xo = (X-1)/2+1;
 *              out_batch_stride in elements (0 = dense)
 * Cin in 1..5, H % 4 == 0, 16-byte aligned pointers, 0 <= slope <= 1 — otherwise LR_EUNSUPPORTED / LR_EALIGN and the
 * caller runs the two blocks as two kernels. */
int64_t lr_conv3d_pair01_packed_floats(int Cin, int C0, int C1);
int lr_conv3d_pair01_pack_f32(const float* w0, const float* w1, float* packed, int Cin, int C0, int C1, void* stream);
int lr_conv3d_pair01_f32(const float* in0, int64_t in0_batch_stride, const float* in_rest, int64_t rest_batch_stride,
                         const float* packed, const float* bias0, const float* bias1, float* out, int B, int Cin,
                         int D, int W, int H, int out_layout, float slope0, float slope1, int64_t out_batch_stride,
                         void* stream);
/* ... on a z-slab (liftreg_amd/parallel.py): the buffers in0 / in_rest hold the D global planes
 * [z_lo, z_lo + D) of a volume of D_global planes; output planes [oz_lo, oz_lo + n_oz) are computed into planes 0..n_oz-1 of
 * `out`.  They read input planes 2*oz_lo - 2 .. 2*(oz_lo + n_oz) — those that exist must lie inside the buffers
 * (LR_EINVAL otherwise); planes outside the GLOBAL volume are the convs' zero padding.  The two blocks need no halo
 * exchange this way: in0 is a view of the replicated moving volume, in_rest the rank's own backprojection of planes
 * [z_lo, z_lo + D).  Same bits as the whole-volume call. */
int lr_conv3d_pair01_slab_f32(const float* in0, int64_t in0_batch_stride, const float* in_rest, int64_t rest_batch_stride,
                              const float* packed, const float* bias0, const float* bias1, float* out, int B, int Cin,
                              int D, int W, int H, int out_layout, float slope0, float slope1, int64_t out_batch_stride,
                              int D_global, int z_lo, int oz_lo, int n_oz, void* stream);

/* Training forward of the two blocks: lr_conv3d_pair01_f32 (dense output) that ALSO writes
 *   act0  : dev (B,D,W,H,16) fp32, block 0's activation in mid_layout (LR_LAYOUT_NDHWC | LR_LAYOUT_NDHWC_HPS) — exactly the
 *           fp32 values whose three-way splits fed block 1; what block 1's weight gradient reads;
 *   mask0 : dev (B,D,W,H,4) uint8 (LR_LAYOUT_SIGN4), its LeakyReLU sign mask as lr_conv3d_k3_lrelu_mask_f32 writes it — what
 *           the fused data-gradient + block-0 weight-gradient kernel (lr_conv3d_dgrad_wgrad0_f32) reads.
 * Whole volumes only; Cin in 1..5; act0 16-byte, mask0 4-byte aligned; D*W*H*64 < 2^31.  Replaces layers.py:365-369 twice in the
 * training forward (RegistrationNet.py:389-406 drives it). */
int lr_conv3d_pair01_train_f32(const float* in0, int64_t in0_batch_stride, const float* in_rest, int64_t rest_batch_stride,
                               const float* packed, const float* bias0, const float* bias1, float* out, float* act0,
                               uint8_t* mask0, int B, int Cin, int D, int W, int H, int mid_layout, int out_layout,
                               float slope0, float slope1, void* stream);


/* ---- f1: PCA reconstruction + identity + trilinear warp in ONE pass (the model's decode half in inference).
 * Replaces the sequence …Backproj.py:102 (F.linear with the PCA basis) → :68 (disp + id) → :69 (Bilinear warp) and
 * writes all three model outputs: disp = `params` (B,3,D,W,H), phi (B,3,D,W,H), warped (B,C,D,W,H).  Same arithmetic
 * as lr_pca_reconstruct_f32 followed by lr_warp_trilinear_f32 (bit-identical outputs); the displacement field is
 * never read back.  basis (L, ldb >= 3·D·W·H) fp32 or bf16 storage; zeros padding, flags = 0 | LR_WARP_USING_SCALE;
 * B <= 8 per call, H % 4 == 0, 16-byte aligned buffers, volume under 2 GB: otherwise LR_EUNSUPPORTED / LR_EALIGN and
 * the caller runs the two separate entry points. */
int lr_pca_warp_f32(const float* coefs, const float* basis, const float* mean, const float* img, const float* id0,
                    const float* id1, const float* id2, float* disp, float* phi, float* warped, int B, int L, int C,
                    int D, int W, int H, int64_t ldb, int flags, void* stream);

int lr_pca_warp_bf16basis_f32(const float* coefs, const void* basis_bf16, const float* mean, const float* img,
                              const float* id0, const float* id1, const float* id2, float* disp, float* phi,
                              float* warped, int B, int L, int C, int D, int W, int H, int64_t ldb, int flags,
                              void* stream);

/* Slab form of the one-pass decode, with the similarity's moments in its epilogue:
 *  - rows [d0,d1) of D: outputs are (B,3,Dn,W,H) / (B,C,Dn,W,H) slabs (Dn = d1-d0); `img` is the WHOLE moving volume
 *    (taps cross slabs); `basis` / `mean` point at the column of (component 0, row d0) and their three component
 *    thirds are `basis_comp_stride` columns apart: D*W*H for a view into the full (L,3V) basis (ldb = its row stride),
 *    Dn*W*H for a rank's compact (L,3*Dn*W*H) slab of it (11.3 GB -> 1.4 GB per GPU at 8 ranks); id0 = the D-axis
 *    identity table from row d0 on.  basis_is_bf16: the basis is stored as bfloat16 (ldb and stride in elements).
 *  - target != NULL (C == 1): (B,1,Dn,W,H) target slab; the five fp64 raw moments of (warped, target) per batch row
 *    (layers/losses.py:18-29, the sums lr_ncc_moments_f32 produces) are accumulated while `warped` is still in
 *    registers: ncc_partial = B * 4 * gridblocks * 5 doubles of scratch (gridblocks = ceil(W*H/1024) * Dn, one partial per wave),
 *    ncc_moments = (B,5) doubles.  Slab moments add (all-reduce), lr_ncc_loss_from_moments turns them into the loss.
 * Same bits as lr_pca_warp_f32 for params / phi / warped. */
int lr_pca_warp_slab_f32(const float* coefs, const void* basis, int basis_is_bf16, const float* mean, const float* img,
                         const float* id0, const float* id1, const float* id2, float* disp, float* phi, float* warped,
                         int B, int L, int C, int D, int W, int H, int d0, int d1, int64_t ldb,
                         int64_t basis_comp_stride, int flags, const float* target, double* ncc_partial,
                         double* ncc_moments, void* stream);

/* ---- bf16 variant of the stride-2 encoder blocks (Cin, Cout in {16,32}; v_mfma_f32_16x16x32_bf16, fp32
 * accumulate, bias/LeakyReLU in fp32, output rounded to bf16 — or fp32 NCDHW for the last block).
 * in: bf16 LR_LAYOUT_BF16_NDHWC[_HPS]; packed_w: lr_conv3d_packed_bf16_bytes(...) bytes written by
 * lr_conv3d_pack_weights_bf16 from the fp32 (Cout,Cin,3,3,3) parameter; out_layout: LR_LAYOUT_NCDHW (fp32) or a
 * bf16 layout.  lr_cast_f32_to_bf16: elementwise round-to-nearest-even (layout-preserving). */
int64_t lr_conv3d_packed_bf16_bytes(int Cin, int Cout);
int lr_conv3d_pack_weights_bf16(const float* weight, void* packed, int Cin, int Cout, void* stream);
int lr_conv3d_k3_lrelu_bf16(const void* in, const void* packed_w, const float* bias, void* out, int B, int Cin,
                            int Cout, int D, int W, int H, int stride, int in_layout, int out_layout,
                            float negative_slope, void* stream);
/* lr_conv3d_k3_lrelu_bf16 into a strided batch (see lr_conv3d_k3_lrelu_obs_f32; stride in elements of the output type). */
int lr_conv3d_k3_lrelu_obs_bf16(const void* in, const void* packed_w, const float* bias, void* out, int B, int Cin,
                                int Cout, int D, int W, int H, int stride, int in_layout, int out_layout,
                                float negative_slope, int64_t out_batch_stride, void* stream);
int lr_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream);
/* The first block of the bf16 variant: fp32 NCDHW input (CT + backprojection; rounded to bf16 on the way into
 * the MFMA), stride 1, any Cin (passes of 3 channels), bf16 channels-last output (out_layout 3|4).
 * packed_w: lr_conv3d_packed_bf16_planar_bytes(...) bytes from lr_conv3d_pack_weights_bf16_planar. */
int64_t lr_conv3d_packed_bf16_planar_bytes(int Cin, int Cout);
int lr_conv3d_pack_weights_bf16_planar(const float* weight, void* packed, int Cin, int Cout, void* stream);
int lr_conv3d_first_bf16(const float* in, const void* packed_w, const float* bias, void* out, int B, int Cin,
                         int Cout, int D, int W, int H, int out_layout, float negative_slope, void* stream);
/* The bf16 variant's encoder input for MANY views (C4: 11) without the fp32 feature volume: lr_backproject_encin_bf16 writes
 * rows [d0,d1) of cat([moving, backprojected views], dim=1) (src/liftreg/models/LiftRegDeformSubspaceBackproj.py:85-98) as
 * bf16 channels-last records (B, d1-d0, W, H, 16): channel 0 = moving (B,1,D,W,H) fp32, 1..P = the samples of
 * lr_backproject_f32, the rest 0, each rounded to nearest-even bf16 — the rounding lr_conv3d_first_bf16 applies to its input
 * anyway.  lr_conv3d_first_clin_bf16 is lr_conv3d_first_bf16 on that tensor (same packed weights, same results).
 * 1 <= P <= 15; out_batch_stride in bf16 elements (0 = dense). */
int lr_backproject_encin_bf16(const float* proj, const float* moving, const float* poses, void* out, int B, int P, int Pw,
                              int Ph, int D, int W, int H, int d0, int d1, int64_t out_batch_stride, void* stream);
int lr_conv3d_first_clin_bf16(const void* in, const void* packed_w, const float* bias, void* out, int B, int Cin, int Cout,
                              int D, int W, int H, int out_layout, float negative_slope, int64_t out_batch_stride,
                              void* stream);
/* Training forward of the bf16 variant's first block (Cin <= 3): lr_conv3d_first_bf16 that ALSO writes mask_out
 * (B,D,W,H,Cout/4) uint8 = LR_LAYOUT_SIGN4 of its bf16 output (bit r of byte q: stored channel 4q+r > 0).  lr_conv3d_dgrad_bf16
 * and lr_conv3d_dgrad_f32 accept it as x_saved with x_layout = LR_LAYOUT_SIGN4: the LeakyReLU mask of
 * src/liftreg/layers/layers.py:369 for the backward of the NEXT block at 4 bytes per voxel instead of 32. */
int lr_conv3d_first_mask_bf16(const float* in, const void* packed_w, const float* bias, void* out, uint8_t* mask_out, int B,
                              int Cin, int Cout, int D, int W, int H, int out_layout, float negative_slope, void* stream);
/* ... into a strided batch (see lr_conv3d_k3_lrelu_obs_f32; stride in bf16 elements, 0 = dense). */
int lr_conv3d_first_obs_bf16(const float* in, const void* packed_w, const float* bias, void* out, int B, int Cin, int Cout,
                             int D, int W, int H, int out_layout, float negative_slope, int64_t out_batch_stride,
                             void* stream);

/* bf16-gradient training variant: the pre-activation gradients between the blocks are stored as bf16 plain
 * channels-last (what bf16 mixed precision does).  lr_conv3d_dgrad_bf16: gx (B,D,W,H,Cx) bf16 = the PRODUCER's
 * pre-activation gradient (its LeakyReLU mask, read from x_saved = this block's saved bf16 input, is applied in the
 * epilogue) from gpre (B,Do,Wo,Ho,32) bf16; packed_wT = lr_conv3d_pack_weights_bf16 of the weight transposed to
 * (Cin,Cout,3,3,3).  lr_conv3d_wgrad_bf16g_f32 = lr_conv3d_wgrad_f32 reading such a gpre (gw, gb stay fp32). */
int lr_conv3d_dgrad_bf16(const void* gpre, const void* packed_wT, void* gx, int B, int Cg, int Cx, int D, int W, int H,
                         const void* x_saved, int x_layout, float negative_slope, void* stream);
int lr_conv3d_wgrad_bf16g_f32(const float* x, int x_layout, const void* gpre_bf16, float* partial, float* gw, float* gb,
                              int B, int Cin, int Cout, int D, int W, int H, int stride, int nblk, void* stream);

/* ---- data-side prologue and evaluation reductions -------------------------------------
 * lr_normalize_clip_f32: out = ((clamp(in, lo, hi) - lo) / (hi - lo)) * 2 - 1
 *   (dataset/Registration2D3DDataset.py:196-199,207: _normalize_intensity with linear_clip and a clip_range).
 * lr_label_overlap_f32: counts[3] (int64, device) = |pred==label|, |gt==label|, |both| over n elements
 *   (utils/metrics.py:83-121 cal_metric; dice/iou/recall/precision follow on the host).  partial: nblk*3 uint64.
 * lr_jacobi_det_stats_f32: out[2] (double, device) = sum of |det J| over voxels with det J < 0, and their count,
 *   for a (B,3,D,W,H) map; sp* = the finite-difference spacing per axis (utils/utils.py:20-55
 *   compute_jacobi_map passes spacing*span).  partial: B*nblk*2 doubles.  Stencil assumed (mermaid): parity unpinned. */
/* lr_sample_points_f64: landmark sampler of tools/evaluate_dir_lab.py:46-59 (calc_warped_points): trilinear
 *   F.grid_sample(align_corners=True, zeros padding) of a DOUBLE (C,D,W,H) map at N normalised points (x,y,z) =
 *   (H,W,D axes); vol, pts (N,3), out (N,C): dev doubles.  The flip and the (dim-1)*phi_spacing scale stay on the host. */
int lr_sample_points_f64(const double* vol, const double* pts, double* out, int C, int D, int W, int H, int N, void* stream);
int lr_normalize_clip_f32(const float* in, float* out, int64_t n, float lo, float hi, void* stream);
int lr_label_overlap_f32(const float* pred, const float* gt, float label, int64_t n, void* partial, int nblk,
                         int64_t* counts, void* stream);
int lr_jacobi_det_stats_f32(const float* map, int B, int D, int W, int H, float sp0, float sp1, float sp2,
                            double* partial, int nblk, double* out, void* stream);

/* ------------------------------------------------------------------------
 * EXPERIMENTAL — not part of the product build.  `make -C liftreg_amd/csrc exp` compiles the library with -DLR_EXPERIMENTAL
 * (plus conv0_pc.hip) into libliftreg_hip_exp.so; LIFTREG_HIP_LIB=<that file> makes liftreg_amd load it (tests of these paths
 * skip on the product library).  Both paths are bit-tested against the default ones and were measured no faster:
 *   - the backprojection computed inside block 0's staging (conv0_pc.hip): slower than lr_backproject_f32 + the fused pair
 *     kernel since round 2;
 *   - the first fp32 block alone on exact bf16 splits (the fp32 route of conv0_split_f32.hip): superseded by the pair kernel.
 * Switches read only by the experimental build:
 *   LIFTREG_CONV0_SPLIT            first fp32 block alone on the bf16 MFMA with exact 3-way operand splits (conv0_split_f32.hip)
 *   LIFTREG_CONV0_PC               first fp32 block as the producer/consumer kernel (conv0_pc.hip; same bits)
 */
#ifdef LR_EXPERIMENTAL
/* f1 ("backproject -> conv0: never write the P*V volume"): the same first block with the backprojection
 * computed INSIDE its staging.  Channel 0 = in0 (B,1,D,W,H), channels 1..P = the backprojection of proj (B,P,Pw,Ph)
 * for `poses` (host, P x 3 fp32, ONE geometry for the batch, …Backproj.py:85-87) — sample for sample the arithmetic
 * of lr_backproject_f32 (sdct_projection_utils.py:227-250 + F.grid_sample 2-D, …Backproj.py:89-93), gathered from the
 * views (L2-resident).  Output bits = lr_backproject_f32 followed by lr_conv3d_first_split_f32; the (B,P,D,W,H)
 * feature volume is never written or read.  P in {1,2}, H % 4 == 0, in0 16-byte aligned, Cout = 16 — otherwise
 * LR_EUNSUPPORTED and the caller runs the two kernels. */
int lr_conv3d_first_fused_bp_f32(const float* in0, const float* proj, const float* poses, const float* packed_w,
                                 const float* bias, float* out, int B, int P, int Pw, int Ph, int Cout, int D, int W, int H,
                                 int out_layout, float negative_slope, void* stream);
#endif /* LR_EXPERIMENTAL */

#ifdef __cplusplus
}
#endif
#endif /* LIFTREG_HIP_H */
