"""The configured LiftReg model (cur_task_setting.json:60) on the HIP kernels.

Same constructor, `forward(dict) -> dict` contract, output keys and state-dict keys as
src/liftreg/models/LiftRegDeformSubspaceBackproj.py:9-113, so the reference's harness
(`RegistrationNet.py:95,396,410`) can load it through `train.model_class`:

    "model_class": "liftreg_amd.models.LiftRegDeformSubspaceBackproj.model"

Data flow (one registration batch, all on the GPU, no sampling grid is ever materialised):
    target_proj ──backproject──► channels 1..P of the encoder input (NCDHW, written in place)
    moving ─────────────────────► channel 0
    encoder input ─conv0(NCDHW→NDHWC)─conv1..4(NDHWC)─conv5(NDHWC→NCDHW)─► flatten ─FC×3─► coefs
    coefs ──pca_reconstruct──► disp (B,3,D,W,H)
    disp, moving(,seg) ──warp (adds the identity map, samples)──► phi, warped
Differences from the reference that are deliberate: the flatten width is 32·(n/32)³ instead of
the hard-coded 32·5³ (:36, 160³ only); tensors follow the input's device instead of `.cuda()`
in the constructor; `opt["pca_path"]` may be "synthetic[:seed]" for benchmarks.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import _hip, ops, ops_bwd
from ..autograd import ConvPair01Fn, DecodeFn, EncoderBf16Fn
from ..layers.layers import convBlock, FullyConnectBlock, GaussianSmoothing
from ..utils.net_utils import Bilinear, identity_axis_tables


def _opt(opt, key, default):
    """ParameterDict-style `opt[(key, default, comment)]` with a plain-dict fallback (optional, non-reference keys)."""
    try:
        return opt[(key, default, "")]
    except (KeyError, TypeError):
        return opt.get(key, default) if hasattr(opt, "get") else default


def _out_size(n, strides):
    for s in strides:
        n = (n - 1) // s + 1
    return n


class model(nn.Module):
    """Estimates the coefficients of a pre-built PCA subspace of the displacement field.

    :param img_sz: voxel shape [D, W, H]
    :param opt: mapping with "drr_feature_num" (P), "latent_dim" (L), "pca_path"
    """
    # bf16 inference: from this many views on, the encoder input is written as 32-byte channels-last bf16 records
    # (ops.backproject_encoder_input_bf16) instead of the fp32 planar feature volume (4 (P + 1) bytes per voxel)
    ENCIN_MIN_VIEWS = 8
    PAIR01_MAX_VIEWS = 4   # fp32 inference: up to this many views the first block reads `moving` and the views from their own buffers (pair kernel: Cin <= 5)

    def __init__(self, img_sz, opt=None):
        super().__init__()
        enc_filters = [16, 32, 32, 32, 32, 32]
        self.input_channel = 2
        self.output_channel = 3
        self.img_sz = [int(v) for v in img_sz]
        self.gaussian_smooth = GaussianSmoothing(4, 8, 2, dim=2)  # state-dict parity only
        self.drr_feature_num = int(opt["drr_feature_num"])
        self.latent_dim = int(opt["latent_dim"])

        self.encoders = nn.ModuleList()
        self.bilinear = Bilinear(zero_boundary=True, using_scale=True)
        self.strides = [1, 2, 2, 2, 2, 2]
        last = len(enc_filters) - 1
        for i, f in enumerate(enc_filters):
            cin = self.drr_feature_num + 1 if i == 0 else enc_filters[i - 1]
            # activations between blocks are channels-last; when a block's output H is even it is written
            # parity-split (LAYOUT_NDHWC_HPS) so the following stride-2 block's tap loads are contiguous
            h_in = _out_size(self.img_sz[2], self.strides[:i])          # H of this block's input
            h_out = _out_size(self.img_sz[2], self.strides[:i + 1])
            lay_in = ops.LAYOUT_NCDHW if i == 0 else (ops.LAYOUT_NDHWC_HPS if h_in % 2 == 0 else ops.LAYOUT_NDHWC)
            lay_out = ops.LAYOUT_NCDHW if i == last else (ops.LAYOUT_NDHWC_HPS if h_out % 2 == 0 else ops.LAYOUT_NDHWC)
            self.encoders.append(convBlock(cin, f, stride=self.strides[i], bias=True, in_layout=lay_in,
                                           out_layout=lay_out))
        # optional (non-reference) key "conv_dtype": "bf16" stores the activations between the blocks as bfloat16
        # and runs the blocks on the bf16 MFMA (BASELINE configs C4/C5); training computes fp32 gradients of that
        # bf16-forward arithmetic (autograd.EncoderBf16Fn); default "fp32"
        self.conv_dtype = str(_opt(opt, "conv_dtype", "fp32"))
        if self.conv_dtype not in ("fp32", "bf16"):
            raise ValueError('conv_dtype must be "fp32" or "bf16"')
        # with conv_dtype "bf16": "grad_dtype": "bf16" also stores the pre-activation gradients between the blocks
        # as bf16 (data gradient on the bf16 MFMA); default "fp32" = the exact gradient of the bf16 forward
        self.grad_dtype = str(_opt(opt, "grad_dtype", "fp32"))
        if self.grad_dtype not in ("fp32", "bf16") or (self.grad_dtype == "bf16" and self.conv_dtype != "bf16"):
            raise ValueError('grad_dtype must be "fp32", or "bf16" together with conv_dtype "bf16"')
        self._bf16_layouts = []
        for i in range(len(enc_filters)):
            h_in = _out_size(self.img_sz[2], self.strides[:i])
            h_out = _out_size(self.img_sz[2], self.strides[:i + 1])
            lin = ops.LAYOUT_NCDHW if i == 0 else (ops.LAYOUT_BF16_NDHWC_HPS if h_in % 2 == 0 else ops.LAYOUT_BF16_NDHWC)
            lout = ops.LAYOUT_NCDHW if i == last else (ops.LAYOUT_BF16_NDHWC_HPS if h_out % 2 == 0 else ops.LAYOUT_BF16_NDHWC)
            self._bf16_layouts.append((lin, lout))
        # optional (non-reference) key "pca_dtype": "bf16" keeps the (L,3V) basis as bfloat16 in HBM (half the bytes of
        # the two basis passes; coefficients, mean and the displacement field stay fp32); default "fp32"
        self.pca_dtype = str(_opt(opt, "pca_dtype", "fp32"))
        if self.pca_dtype not in ("fp32", "bf16"):
            raise ValueError('pca_dtype must be "fp32" or "bf16"')
        # backward chaining: block i+1's data gradient applies block i's LeakyReLU mask in its epilogue and hands
        # block i its pre-activation gradient directly (one pass over the big activations less per block)
        for i in range(last):
            self.encoders[i].premasked_grad = True
            self.encoders[i + 1].mask_input_slope = self.encoders[i]._slope
        flat = enc_filters[-1] * int(np.prod([_out_size(n, self.strides) for n in self.img_sz]))
        self.encoders.append(nn.Sequential(
            nn.Flatten(),
            FullyConnectBlock(flat, 800),
            FullyConnectBlock(800, 256),
            FullyConnectBlock(256, self.latent_dim, nonlinear=None)))

        # PCA basis: kept as the (L, 3V) C-contiguous array of pca_vectors.npy; the reference's
        # `.T` view (…Backproj.py:42) is exposed as the `pca_vectors` property.  Non-persistent
        # buffers: they follow .to()/.cuda() but stay out of state_dict() like the reference's
        # plain attributes.
        M = 3 * int(np.prod(self.img_sz))
        pca_path = opt["pca_path"]
        if isinstance(pca_path, str) and pca_path.startswith("synthetic"):
            self._synthetic_seed = int(pca_path.split(":")[1]) if ":" in pca_path else 2021
            vec = torch.empty((0, M))
            mean = torch.empty((0,))
        else:
            self._synthetic_seed = None
            vec = torch.from_numpy(np.load(f"{pca_path}/pca_vectors.npy")).float()
            mean = torch.from_numpy(np.load(f"{pca_path}/pca_mean.npy")).float()
            if tuple(vec.shape) != (self.latent_dim, M) or tuple(mean.shape) != (M,):
                raise ValueError(f"pca_vectors.npy must be ({self.latent_dim},{M}), pca_mean.npy ({M},)")
        self.register_buffer("pca_vectors_LxM", vec.contiguous(), persistent=False)
        self.register_buffer("pca_mean", mean.contiguous(), persistent=False)

        t0, t1, t2 = identity_axis_tables(self.img_sz)
        self.register_buffer("_id0", torch.from_numpy(t0), persistent=False)
        self.register_buffer("_id1", torch.from_numpy(t1), persistent=False)
        self.register_buffer("_id2", torch.from_numpy(t2), persistent=False)
        # optional (non-reference) key "fuse_ncc" (default False): in inference the one-pass decode also accumulates the
        # similarity's five fp64 moments against `target` in its epilogue (one extra read of the target there instead of
        # a pass over both volumes; moments equal to 1e-12).  Off by default: interleaved A/B at C3 measured it 0.04 ms
        # SLOWER than decode + the 0.19 ms moments kernel — the decode runs at 2 waves per SIMD and is HBM-latency-bound,
        # so the epilogue's extra stream and fp64 reduction lengthen every wave's tail by what the saved pass costs.
        self.fuse_ncc = bool(_opt(opt, "fuse_ncc", False))
        # optional (non-reference) key "fuse_backproject" (default False): in fp32 inference with P <= 2 views the
        # backprojection is computed inside the first conv block (conv0_pc.hip producers; the (B,P,D,W,H) feature volume
        # is never materialised; same bits).  Off by default because it MEASURED SLOWER at C3: the block is bound by the
        # matrix pipe, its 4x4x64 bricks recompute every sample 2.5x (window halo), and the vector-ALU work of those
        # samples takes issue slots from the MFMAs — 4.05 ms fused vs 3.31 + 0.27 ms as two kernels.
        # Worth it where HBM capacity/traffic matters more than the 4 % (the volume is 1.08 GB per batch of 8).
        # EXPERIMENTAL since round 6: the kernel is only in the `make exp` build of the library (include/liftreg_hip.h, last
        # section; LIFTREG_HIP_LIB selects it) — asking for it on the product library is an error, not a silent two-kernel run.
        self.fuse_backproject = bool(_opt(opt, "fuse_backproject", False))
        if self.fuse_backproject and not _hip.has_experimental():
            raise _hip.LiftRegHipError("fuse_backproject needs the experimental build of libliftreg_hip (make -C liftreg_amd/csrc exp; "
                                       "LIFTREG_HIP_LIB=<libliftreg_hip_exp.so>)")
        # optional (non-reference) key "fuse_first_backward" (default True): in fp32 training the first two encoder blocks
        # are one autograd node whose backward computes block 1's data gradient and block 0's weight gradient in one kernel
        # (autograd.ConvPair01Fn); False = one node per block (the gradient between them goes through memory)
        self.fuse_first_backward = bool(_opt(opt, "fuse_first_backward", True))
        # optional (non-reference) key "fuse_pair01" (default True): in fp32 inference with P <= 4 views (PAIR01_MAX_VIEWS), encoder blocks 0 and 1
        # run as ONE kernel (csrc/conv01_fused.hip): both on the bf16 matrix pipe with exact three-way bf16 splits of their fp32
        # operands (six exact partial products, fp32 accumulation — closer to an fp64 convolution than the fp32 fmaf chain,
        # tests/test_gpu_conv01_fused.py); the 16-channel activation between them (8.6 GB per batch of 8 at 256^3) never reaches
        # HBM.  False = one fp32-MFMA kernel per block (the round-3 path).
        self.fuse_pair01 = bool(_opt(opt, "fuse_pair01", True))
        # optional key "split_encoder_input" (default True): fp32 training reads `moving` and the backprojected views from their own
        # buffers in the fused forward and backward of blocks 0 + 1; False = the concatenated input of rounds 1-4 (A/B aid)
        self.split_encoder_input = bool(_opt(opt, "split_encoder_input", True))
        # optional key "fuse_pair01_train" (default True): the TRAINING forward of blocks 0 + 1 through the same fused kernel, which
        # then also writes block 0's activation and sign mask for the backward (ops.conv3d_pair01_train); False = the two
        # fp32-MFMA kernels of rounds 2-3
        self.fuse_pair01_train = bool(_opt(opt, "fuse_pair01_train", True))
        # optional (non-reference) key "reg_in_coef_space" (default True): in training the output dict also carries
        # "pca_reg_gram" = the regulariser's quadratic form on the PCA basis (ops.subspace_reg_gram, computed once per
        # basis), so that liftreg_amd.losses.SubspaceLoss evaluates R(params) and its gradient on the (B,L) coefficients
        # instead of streaming the (B,3,D,W,H) field once forward and twice backward — the same number to fp32 rounding
        # (params = coefs . basis^T + mean is affine in the coefficients and R is a quadratic form of the field)
        self.reg_in_coef_space = bool(_opt(opt, "reg_in_coef_space", True))
        # optional (non-reference) key "ncc_grad_via_moments" (default True): in training (single-channel images, no label
        # masks) the decode node also returns the similarity's five moments as a differentiable output ("ncc_moments");
        # NCCLoss is a function of them, its gradient comes back as (B,5) numbers, and the warp-gradient kernel forms
        # d loss / d warped = gm0 + gm2·target + 2·gm3·warped on the fly — the pass that writes that gradient is gone
        self.ncc_grad_via_moments = bool(_opt(opt, "ncc_grad_via_moments", True))
        self._reg_gram = None      # (key, (gram, lin, r0))
        self._poses = None         # geometry of the first batch's element 0, cached like :85-87
        self._packed = {}          # conv weights in MFMA operand order, keyed by parameter version
        self._pca_slabs = {}       # compact per-rank column slabs of the basis (pca_slab)

    # ------------------------------------------------------------------ reference-compatible surface
    @property
    def pca_vectors(self):
        return self.pca_vectors_LxM.T

    @property
    def id_transform(self):
        return torch.stack(torch.broadcast_tensors(self._id0[:, None, None], self._id1[None, :, None],
                                                   self._id2[None, None, :]))

    def get_extra_to_plot(self):
        return None, None

    def get_disp(self):
        return None, ""

    def reset_geometry(self):
        """Forget the cached emitter geometry (the reference can only do this by rebuilding the model)."""
        self._poses = None

    def set_pca(self, vectors_LxM, mean):
        if tuple(vectors_LxM.shape) != (self.latent_dim, 3 * int(np.prod(self.img_sz))):
            raise ValueError("basis must be (latent_dim, 3*D*W*H)")
        self.pca_vectors_LxM = self._basis_storage(vectors_LxM.contiguous())
        self.pca_mean = mean.contiguous()
        self._pca_slabs.clear()

    # ------------------------------------------------------------------ internals
    def _ensure_pca(self, device):
        if self.pca_vectors_LxM.numel() == 0:
            if self._synthetic_seed is None:
                raise RuntimeError("PCA basis missing")
            # randn(L,3V)*(0.02/sqrt(L)), mean 0, generated on the device in row chunks
            g = torch.Generator(device=device)
            g.manual_seed(self._synthetic_seed)
            M = 3 * int(np.prod(self.img_sz))
            vec = torch.empty((self.latent_dim, M), dtype=torch.float32, device=device)
            for l in range(self.latent_dim):
                vec[l].normal_(0.0, 0.02 / float(np.sqrt(self.latent_dim)), generator=g)
            self.pca_vectors_LxM = self._basis_storage(vec)
            self.pca_mean = torch.zeros((M,), dtype=torch.float32, device=device)
        elif self.pca_dtype == "bf16" and self.pca_vectors_LxM.dtype != torch.bfloat16:
            self.pca_vectors_LxM = self._basis_storage(self.pca_vectors_LxM)     # a basis loaded from pca_path

    def reg_gram(self):
        """(gram, lin, r0) of ops.subspace_reg_gram for the resident basis, computed on first use and kept until the
        basis tensor changes (one pass of the regulariser's gradient kernel and of the PCA-gradient kernel per 8 rows)."""
        vec, mu = self.pca_vectors_LxM, self.pca_mean
        # keyed on the tensors THEMSELVES (held here, so their addresses cannot be reused) and their versions: a reassigned or
        # reloaded basis / mean is another object, an in-place update bumps _version
        hit = self._reg_gram
        if hit is None or hit[0] is not vec or hit[1] != vec._version or hit[2] is not mu or hit[3] != mu._version:
            with torch.no_grad():
                self._reg_gram = (vec, vec._version, mu, mu._version, ops.subspace_reg_gram(vec, mu, self.img_sz))
        return self._reg_gram[4]

    def pca_slab(self, d0, d1, device):
        """(basis (L, 3·Dn·W·H), mean (3·Dn·W·H,)): the COMPACT column slab of rows [d0,d1) of D — what one rank of a
        z-slab sharded registration keeps ("shard basis rows by slab, 11.3 GB → 1.4 GB/GPU at G=8").
        Built once per (d0,d1): sliced out of the full basis when that is resident, else (synthetic basis) generated row
        by row through a (3V,) scratch row, so the full (L,3V) array never exists on a sharded rank.  Same values as
        the corresponding columns of the full basis."""
        D, W, H = self.img_sz
        key = (int(d0), int(d1), str(device))
        hit = self._pca_slabs.get(key)
        if hit is not None:
            return hit
        plane, Dn = W * H, d1 - d0
        runs = [((c * D + d0) * plane, (c * D + d1) * plane) for c in range(3)]     # three contiguous column runs
        if self.pca_vectors_LxM.numel() > 0:
            # a loaded basis: copy ONLY the three column runs to the device, from wherever the full array lives (after
            # `offload_full_basis()` that is host memory, so a sharded rank holds 1/world of the basis and nothing else);
            # no index tensor, no device copy of the full array
            vec, mu = self.pca_vectors_LxM, self.pca_mean
            basis = torch.empty((self.latent_dim, 3 * Dn * plane), dtype=vec.dtype, device=device)
            mean = torch.empty((3 * Dn * plane,), dtype=torch.float32, device=device)
            for c, (lo, hi) in enumerate(runs):
                for l in range(self.latent_dim):     # row by row: both sides contiguous -> direct copies, no staging temporaries
                    basis[l, c * Dn * plane:(c + 1) * Dn * plane].copy_(vec[l, lo:hi])
                mean[c * Dn * plane:(c + 1) * Dn * plane].copy_(mu[lo:hi])
            basis = self._basis_storage(basis)
        else:
            if self._synthetic_seed is None:
                raise RuntimeError("PCA basis missing")
            g = torch.Generator(device=device)
            g.manual_seed(self._synthetic_seed)
            row = torch.empty((3 * D * plane,), dtype=torch.float32, device=device)
            basis = torch.empty((self.latent_dim, 3 * Dn * plane), dtype=torch.float32, device=device)
            for l in range(self.latent_dim):      # the same stream of normals as _ensure_pca's full basis
                row.normal_(0.0, 0.02 / float(np.sqrt(self.latent_dim)), generator=g)
                for c, (lo, hi) in enumerate(runs):
                    basis[l, c * Dn * plane:(c + 1) * Dn * plane].copy_(row[lo:hi])
            basis = self._basis_storage(basis)
            mean = torch.zeros((3 * Dn * plane,), dtype=torch.float32, device=device)
        self._pca_slabs[key] = (basis, mean)
        return basis, mean

    def offload_full_basis(self):
        """Sharded ranks: move the full (L,3V) basis and mean (loaded from `pca_path`) to host memory, keeping only the
        per-rank column slabs `pca_slab` builds on the device (11.3 GB -> 1.4 GB per GPU at 8 ranks, C3).  The unsharded
        `forward` needs the full basis on the device again (`net.to(device)` brings it back)."""
        if self.pca_vectors_LxM.numel() > 0:
            self.pca_vectors_LxM = self.pca_vectors_LxM.cpu()
            self.pca_mean = self.pca_mean.cpu()

    def _basis_storage(self, vec):
        return vec.to(torch.bfloat16) if self.pca_dtype == "bf16" else vec.to(torch.float32)

    def invalidate_packed(self):
        """Drop the cached MFMA-ordered weight copies.  The cache is keyed on (data_ptr, tensor version, device): an
        optimizer step, `load_state_dict` and `.to()` are seen by themselves (and are hooked below anyway); an in-place
        update THROUGH `.data` (`w.data.mul_()`, EMA / clipping code, an external library) changes neither — call this
        after such an update.  In training mode the weights are re-packed on every forward (the pack kernel is
        microseconds next to a step), so only inference after an out-of-band update needs it."""
        self._packed.clear()

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.get("_packed", {}).clear()
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.__dict__.get("_packed", {}).clear()
        return super().load_state_dict(*args, **kwargs)

    def _packed_weight(self, i, bf16=False):
        blk = self.encoders[i]
        w = blk.conv.weight
        # the cache holds the weight TENSOR it was packed from and compares identity + version (an address-only key would accept a
        # replaced Parameter that landed on the freed address with the same version counter) AND storage address + device: a
        # rebind through `.data` (`p.data = ema_tensor`, the usual EMA swap-in for evaluation) keeps identity and version
        hit = self._packed.get((i, bf16))
        if self.training and w.requires_grad:
            hit = None                       # training: never trust the cache (see invalidate_packed)
        if hit is None or hit[0] is not w or hit[1] != (w._version, w.data_ptr(), w.device):
            if bf16:
                pk = ops.conv3d_pack_weights_bf16_planar(w) if i == 0 else ops.conv3d_pack_weights_bf16(w)
            else:
                pk = ops.conv3d_pack_weights(w, blk.in_layout)
            hit = (w, (w._version, w.data_ptr(), w.device), pk)
            self._packed[(i, bf16)] = hit
        return hit[2]

    def _packed_pair01(self):
        """The split-operand fragments of encoder blocks 0 and 1 (ops.conv3d_pair01_pack), cached like _packed_weight."""
        w0, w1 = self.encoders[0].conv.weight, self.encoders[1].conv.weight
        hit = self._packed.get("pair01")
        if self.training and (w0.requires_grad or w1.requires_grad):
            hit = None
        key = (w0._version, w1._version, w0.data_ptr(), w1.data_ptr(), w0.device)
        if hit is None or hit[0] is not w0 or hit[1] is not w1 or hit[2] != key:
            hit = (w0, w1, key, ops.conv3d_pair01_pack(w0, w1))
            self._packed["pair01"] = hit
        return hit[3]

    def _estimate_flow(self, moving, target_proj, poses):
        coefs = self.encode(moving, target_proj, poses)
        B, _, D, W, H = moving.shape
        disp = ops.pca_reconstruct(coefs, self.pca_vectors_LxM, self.pca_mean).view(B, 3, D, W, H)
        return coefs, disp

    def backproject_views(self, target_proj, poses, img_shape, out=None):
        """target_volume (:89-93) alone: the (B,P,D,W,H) backprojection of the views, for a caller that hands it to
        encode(target_volume=…)."""
        if self._poses is None:
            p = poses.detach().cpu().numpy() if isinstance(poses, torch.Tensor) else np.asarray(poses)
            self._poses = np.ascontiguousarray(p[0], dtype=np.float32)  # poses[0:1] (:87)
        D, W, H = img_shape
        B, P = target_proj.shape[:2]
        if out is None:
            out = torch.empty((B, P, D, W, H), dtype=torch.float32, device=target_proj.device)
        ops.backproject(target_proj, self._poses, (D, W, H), out=out, out_batch_stride=P * D * W * H)
        return out

    def encode(self, moving, target_proj, poses, target_volume=None):
        """MFMA-bound half: backprojection → 6 conv blocks → FC head → PCA coefficients (B,L).  target_volume: the
        backprojection already computed by backproject_views() (fp32 inference path; ignored elsewhere)."""
        self._ensure_pca(moving.device)
        B, _, D, W, H = moving.shape
        P = target_proj.shape[1]
        if self._poses is None:
            p = poses.detach().cpu().numpy() if isinstance(poses, torch.Tensor) else np.asarray(poses)
            self._poses = np.ascontiguousarray(p[0], dtype=np.float32)  # poses[0:1] (:87)
        V = D * W * H
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if self.conv_dtype != "bf16" and not needs_grad and P <= self.PAIR01_MAX_VIEWS:
            # inference, fp32: the first block reads `moving` and the backprojected views from their own buffers —
            # cat([moving, target_volume], dim=1) (:95-98) is never materialised (no copy of `moving`)
            mv = moving if moving.is_contiguous() else moving.contiguous()
            blk = self.encoders[0]
            if self.fuse_backproject and blk.conv.out_channels == 16 and blk.out_layout != ops.LAYOUT_NCDHW and \
                    ops.conv3d_first_fused_bp_supported(mv, target_proj):
                # f1: the backprojection is computed inside the first block's staging — target_volume (:89-93) is
                # never written or read
                x = ops.conv3d_first_fused_bp(mv, target_proj, self._poses, blk.conv.weight, blk.conv.bias,
                                              out_layout=blk.out_layout, negative_slope=blk._slope,
                                              packed=self._packed_weight(0))
                for i in range(1, 6):
                    x = self.encoders[i](x, packed=self._packed_weight(i))
                return self.encoders[6](x)
            have_tv = (target_volume is not None and tuple(target_volume.shape) == (B, P, D, W, H) and
                       target_volume.dtype == torch.float32 and target_volume.is_contiguous())
            tv = target_volume if have_tv else torch.empty((B, P, D, W, H), dtype=torch.float32, device=moving.device)
            b0, b1 = self.encoders[0], self.encoders[1]
            if (self.fuse_pair01 and not self.fuse_backproject and b1.stride == 2 and b0.out_layout == b1.in_layout and
                    ops.conv3d_pair01_supported(mv, tv, b0.conv.weight, b1.conv.weight, b1.out_layout)):
                if not have_tv:
                    ops.backproject(target_proj, self._poses, (D, W, H), out=tv, out_batch_stride=P * V)
                x = ops.conv3d_pair01(mv, tv, b0.conv.weight, b0.conv.bias, b1.conv.weight, b1.conv.bias,
                                      out_layout=b1.out_layout, slope0=b0._slope, slope1=b1._slope, packed=self._packed_pair01())
                for i in range(2, 6):
                    x = self.encoders[i](x, packed=self._packed_weight(i))
                return self.encoders[6](x)
            if ops.conv3d_first_split_supported(mv, tv):
                if not have_tv:
                    ops.backproject(target_proj, self._poses, (D, W, H), out=tv, out_batch_stride=P * V)
                blk = self.encoders[0]
                x = ops.conv3d_first_split(mv, tv, blk.conv.weight, blk.conv.bias, out_layout=blk.out_layout,
                                           negative_slope=blk._slope, packed=self._packed_weight(0))
                for i in range(1, 6):
                    x = self.encoders[i](x, packed=self._packed_weight(i))
                return self.encoders[6](x)
        if (self.conv_dtype == "bf16" and not needs_grad and P >= self.ENCIN_MIN_VIEWS and self.encoders[0].conv.out_channels == 16 and
                ops.encoder_input_bf16_supported(moving, target_proj)):
            # inference, bf16 variant, many views (C4: 11): cat([moving, target_volume]) (:89-98) is written ONCE, as the bf16
            # channels-last records the first block stages (32 bytes per voxel) — the fp32 (B,P,D,W,H) feature volume
            # (2.95 GB written, 1.6x that read back at C4) never exists; the first block's results keep their bits (its
            # inputs were rounded to bf16 anyway, now one kernel earlier)
            mv = moving if moving.is_contiguous() else moving.contiguous()
            x = ops.backproject_encoder_input_bf16(mv, target_proj, self._poses)
            blk = self.encoders[0]
            x = ops.conv3d_first_clin_bf16(x, blk.conv.weight, blk.conv.bias, out_layout=self._bf16_layouts[0][1],
                                           negative_slope=blk._slope, packed=self._packed_weight(0, bf16=True))
            for i in range(1, 6):
                blk = self.encoders[i]
                lin, lout = self._bf16_layouts[i]
                x = ops.conv3d_k3_lrelu_bf16(x, blk.conv.weight, blk.conv.bias, blk.stride, in_layout=lin, out_layout=lout,
                                             negative_slope=blk._slope, packed=self._packed_weight(i, bf16=True))
            return self.encoders[6](x)
        b0, b1 = self.encoders[0], self.encoders[1]
        if (self.conv_dtype != "bf16" and needs_grad and self.fuse_first_backward and self.fuse_pair01 and self.fuse_pair01_train and
                self.split_encoder_input and P <= self.PAIR01_MAX_VIEWS and b0.conv.weight.requires_grad and b1.conv.weight.requires_grad and b0.premasked_grad and b1.stride == 2 and
                b0.out_layout == b1.in_layout and tuple(b1.conv.weight.shape[:2]) == (32, 16)):
            # training, fp32: blocks 0 + 1 as one autograd node (fused pair forward; fused dgrad1 + wgrad0 backward) reading `moving`
            # and the backprojected views from their OWN buffers — cat([moving, target_volume]) (:95-98) is never assembled (the
            # copy of `moving` into a concatenated input cost 0.17 ms per C3 step)
            mv = moving if moving.is_contiguous() else moving.contiguous()
            tv = torch.empty((B, P, D, W, H), dtype=torch.float32, device=moving.device)
            if ops.conv3d_pair01_train_supported(mv, b0.conv.weight, b1.conv.weight, b0.out_layout, b1.out_layout, tv):
                ops.backproject(target_proj, self._poses, (D, W, H), out=tv, out_batch_stride=P * V)
                x = ConvPair01Fn.apply(mv, b0.conv.weight, b0.conv.bias, b1.conv.weight, b1.conv.bias, b0._slope, b1._slope,
                                       b0.out_layout, b1.out_layout, self._packed_weight(0), self._packed_weight(1), b1.premasked_grad,
                                       self._packed_pair01(), tv)
                for i in range(2, 6):
                    x = self.encoders[i](x, packed=self._packed_weight(i))
                return self.encoders[6](x)
            del tv
        # encoder input = cat([moving, target_volume], dim=1) (:95-98), built in place
        x = torch.empty((B, P + 1, D, W, H), dtype=torch.float32, device=moving.device)
        x[:, 0:1].copy_(moving)
        ops.backproject(target_proj, self._poses, (D, W, H), out=x[:, 1:], out_batch_stride=(P + 1) * V)
        if self.conv_dtype == "bf16" and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # training: bf16 forward, fp32 gradient math on the bf16-rounded activations — one autograd node
            blks = [self.encoders[i] for i in range(6)]
            packed = [self._packed_weight(i, bf16=True) for i in range(6)]
            wb = [t for blk in blks for t in (blk.conv.weight, blk.conv.bias)]
            feat = EncoderBf16Fn.apply(x, self._bf16_layouts, [blk._slope for blk in blks], self.strides, packed,
                                       self.grad_dtype == "bf16", *wb)
            return self.encoders[6](feat)
        if self.conv_dtype == "bf16":
            for i in range(6):
                blk = self.encoders[i]
                lin, lout = self._bf16_layouts[i]
                if i == 0:       # fp32 input, rounded to bf16 on the way into the MFMA
                    x = ops.conv3d_first_bf16(x, blk.conv.weight, blk.conv.bias, out_layout=lout,
                                              negative_slope=blk._slope, packed=self._packed_weight(0, bf16=True))
                else:
                    x = ops.conv3d_k3_lrelu_bf16(x, blk.conv.weight, blk.conv.bias, blk.stride, in_layout=lin,
                                                 out_layout=lout, negative_slope=blk._slope,
                                                 packed=self._packed_weight(i, bf16=True))
            return self.encoders[6](x)
        first = 0
        b0, b1 = self.encoders[0], self.encoders[1]
        if (self.fuse_first_backward and needs_grad and b0.conv.weight.requires_grad and b1.conv.weight.requires_grad and
                b0.premasked_grad and b1.stride == 2 and b0.out_layout == b1.in_layout and
                ops.conv3d_mask_supported(x, b0.conv.weight, b0.stride, b0.in_layout, b0.out_layout) and
                tuple(b1.conv.weight.shape[:2]) == (32, 16) and x.shape[1] in (2, 3, 4, 5) and x[0].numel() * 4 < 2 ** 31 - 1):
            # training, fp32: blocks 0 and 1 as one autograd node — the gradient between them never reaches memory
            x = ConvPair01Fn.apply(x, b0.conv.weight, b0.conv.bias, b1.conv.weight, b1.conv.bias, b0._slope, b1._slope,
                                   b0.out_layout, b1.out_layout, self._packed_weight(0), self._packed_weight(1), b1.premasked_grad,
                                   self._packed_pair01() if (self.fuse_pair01 and self.fuse_pair01_train) else None)
            first = 2
        for i in range(first, 6):
            x = self.encoders[i](x, packed=self._packed_weight(i))
        return self.encoders[6](x)

    def decode(self, moving, coefs, moving_seg=None, target=None):
        """HBM-bound half: PCA reconstruction → identity add + trilinear warp.  Returns (disp, phi, warped).
        `target` (inference, single-channel, opt key fuse_ncc): the similarity's five moments of (warped, target) are
        accumulated in the same pass and returned as a 4th value → output key "ncc_moments" (SURVEY §8 f1).
        """
        B, C, D, W, H = moving.shape
        if (moving_seg is None and not (torch.is_grad_enabled() and coefs.requires_grad) and
                ops.pca_warp_supported(coefs, self.pca_vectors_LxM, moving)):
            # inference: one pass writes params, phi and warped — the same bits as the two kernels below
            if target is not None and C == 1 and target.is_cuda and target.dtype == torch.float32 and \
                    target.is_contiguous() and target.shape == moving.shape and self.fuse_ncc:
                return ops.pca_warp(coefs, self.pca_vectors_LxM, self.pca_mean, (self._id0, self._id1, self._id2), moving,
                                    target=target)
            return ops.pca_warp(coefs, self.pca_vectors_LxM, self.pca_mean, (self._id0, self._id1, self._id2), moving)
        # training: one autograd node for PCA reconstruction → (+ identity) → warp; the mask compose of moving
        # ((moving+1)*seg-1, :57) happens on the warp's taps
        if (self.ncc_grad_via_moments and moving_seg is None and target is not None and C == 1 and target.is_cuda and
                target.dtype == torch.float32 and target.is_contiguous() and target.shape == moving.shape and
                torch.is_grad_enabled() and coefs.requires_grad and
                ops_bwd.warp_bwd_disp_ncc_supported(moving, target, (self._id0, self._id1, self._id2))):
            # training: the similarity's moments are a differentiable output of the decode node (output key "ncc_moments")
            return DecodeFn.apply(coefs, self.pca_vectors_LxM, self.pca_mean, moving, self._id0, self._id1, self._id2,
                                  None, True, target)
        return DecodeFn.apply(coefs, self.pca_vectors_LxM, self.pca_mean, moving, self._id0, self._id1, self._id2,
                              moving_seg, True)

    # ------------------------------------------------------------------ forward
    def forward(self, input):
        moving = input['source']
        target = input['target']
        target_proj = input["target_proj"]
        if tuple(moving.shape[2:]) != tuple(self.img_sz):
            raise ValueError(f"model was built for {self.img_sz}, got {tuple(moving.shape[2:])}")
        self._ensure_pca(moving.device)
        if 'source_label' in input:
            moving_seg = input['source_label']
            target_cp = ops.mask_compose(target, input['target_label'])   # (target+1)*target_seg-1
        else:
            moving_seg = None
            target_cp = target

        coefs = self.encode(moving, target_proj, input['target_poses'])
        disp_field, deform_field, warped_source, *mom = self.decode(moving, coefs, moving_seg, target=target_cp)
        extra = {}
        if self.reg_in_coef_space and torch.is_grad_enabled() and coefs.requires_grad and self.pca_vectors_LxM.is_cuda:
            extra["pca_reg_gram"] = self.reg_gram()      # training: the regulariser on the coefficients (SubspaceLoss)
            extra["pca_reg_gram_of"] = (disp_field, disp_field._version, coefs, coefs._version)
        if mom:      # what the moments describe: the loss uses them only for these very tensors, unmodified
            extra["ncc_moments_of"] = (warped_source, warped_source._version, target_cp, target_cp._version)
        return {**({"ncc_moments": mom[0]} if mom else {}),      # training (ncc_grad_via_moments) or the opt key fuse_ncc
                **extra,
                "warped": warped_source,
                "phi": deform_field,
                "params": disp_field,
                "target": target_cp,
                "pca_coefs": coefs,
                "target_proj": target_proj,
                "warped_proj": target_proj}
