"""MI355X-native hot path of lucidrains/muse-maskgit-pytorch (same export list as the reference's __init__.py:1-4,
minus the VAE trainer which is outside the hot path)."""
from .vqgan_vae import VQGanVAE
from .muse_maskgit import Transformer, MaskGit, Muse, MaskGitTransformer, TokenCritic
from .attend import Attend
from .patch import patch_reference, unpatch_reference

__all__ = ['VQGanVAE', 'Transformer', 'MaskGit', 'Muse', 'MaskGitTransformer', 'TokenCritic', 'Attend', 'patch_reference', 'unpatch_reference']
