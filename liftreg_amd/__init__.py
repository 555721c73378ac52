"""liftreg_amd — MI355X-native (gfx950) hot path of uncbiag/LiftReg.

Hand-written HIP kernels behind a C ABI (liftreg_amd/csrc → libliftreg_hip.so,
declared in include/liftreg_hip.h) and the Python plugin surface the reference's
harness loads through dotted class paths (`cur_task_setting.json`):

  liftreg_amd.models.LiftRegDeformSubspaceBackproj.model   (train.model_class)
  liftreg_amd.layers.losses.NCCLoss                         (train.loss.sim_class)
  liftreg_amd.losses.SubspaceLoss.loss                      (train.loss_class)
  liftreg_amd.utils.sdct_projection_utils.*                 (tools/preprocessingDRR.py)
  liftreg_amd.utils.net_utils.Bilinear / identity_map

No CPU or PyTorch fallback exists: without the built library the ops raise.
"""
__version__ = "0.1.0"
