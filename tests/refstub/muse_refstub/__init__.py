"""TEST INFRASTRUCTURE: a structure-only stand-in for the reference package (the real one does not exist on the GPU box).

Same module names (`.muse_maskgit_pytorch`, `.attend`, `.vqgan_vae`), same class names, same attribute / parameter tree as the reference
(muse_maskgit_pytorch.py:63-235, 352-489; attend.py:30-60) -- i.e. the same state-dict keys -- and NO arithmetic: every hot method raises.  The
classes are plain nn.Modules unrelated to muse_maskgit_pytorch_amd's, so `patch_reference(muse_refstub)` has to do on them exactly what it does
on the reference: swap the hot methods on foreign classes and serve them from shadows that share the foreign instances' tensors
(tests/test_gpu_patch_mode2.py)."""
from .muse_maskgit_pytorch import MaskGit, MaskGitTransformer, TokenCritic, Transformer  # noqa: F401
from .vqgan_vae import VQGanVAE  # noqa: F401
