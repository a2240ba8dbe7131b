"""parameter / attribute skeleton of the reference's transformer classes (no arithmetic), see the package docstring"""
import math
from functools import partial

import torch
from torch import nn

from .attend import Attend, _stub
from .vqgan_vae import VQGanVAE

_T5_DIMS = {'t5-small': 512, 't5-base': 768, 'google/t5-v1_1-base': 768}


def _encode_text(texts, name='t5-small'):
    raise RuntimeError('muse_refstub has no T5: pass text_embeds')


class LayerNorm(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))
        self.register_buffer('beta', torch.zeros(dim))


class GEGLU(nn.Module):
    pass


def FeedForward(dim, mult=4):
    inner = int(dim * mult * 2 / 3)
    return nn.Sequential(LayerNorm(dim), nn.Linear(dim, inner * 2, bias=False), GEGLU(), LayerNorm(inner), nn.Linear(inner, dim, bias=False))


class Attention(nn.Module):
    def __init__(self, dim, dim_head=64, heads=8, cross_attend=False, scale=8, flash=True, dropout=0.):
        super().__init__()
        self.scale, self.heads, self.cross_attend = scale, heads, cross_attend
        inner = dim_head * heads
        self.norm = LayerNorm(dim)
        self.attend = Attend(flash=flash, dropout=dropout, scale=scale)
        self.null_kv = nn.Parameter(torch.randn(2, heads, 1, dim_head))
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, inner * 2, bias=False)
        self.q_scale = nn.Parameter(torch.ones(dim_head))
        self.k_scale = nn.Parameter(torch.ones(dim_head))
        self.to_out = nn.Linear(inner, dim, bias=False)


class TransformerBlocks(nn.Module):
    def __init__(self, *, dim, depth, dim_head=64, heads=8, ff_mult=4, flash=True):
        super().__init__()
        self.layers = nn.ModuleList([nn.ModuleList([Attention(dim, dim_head, heads, flash=flash), Attention(dim, dim_head, heads, cross_attend=True, flash=flash),
                                                    FeedForward(dim, ff_mult)]) for _ in range(depth)])
        self.norm = LayerNorm(dim)


class Transformer(nn.Module):
    def __init__(self, *, num_tokens, dim, seq_len, dim_out=None, t5_name='t5-small', self_cond=False, add_mask_id=False, **kwargs):
        super().__init__()
        self.dim = dim
        self.mask_id = num_tokens if add_mask_id else None
        self.num_tokens = num_tokens
        self.token_emb = nn.Embedding(num_tokens + int(add_mask_id), dim)
        self.pos_emb = nn.Embedding(seq_len, dim)
        self.seq_len = seq_len
        self.transformer_blocks = TransformerBlocks(dim=dim, **kwargs)
        self.norm = LayerNorm(dim)
        self.dim_out = dim_out if dim_out is not None else num_tokens
        self.to_logits = nn.Linear(dim, self.dim_out, bias=False)
        self.encode_text = partial(_encode_text, name=t5_name)
        text_dim = _T5_DIMS[t5_name]
        self.text_embed_proj = nn.Linear(text_dim, dim, bias=False) if text_dim != dim else nn.Identity()
        self.self_cond = self_cond
        self.self_cond_to_init_embed = FeedForward(dim)

    forward = _stub
    forward_with_cond_scale = _stub


class MaskGitTransformer(Transformer):
    def __init__(self, *args, **kwargs):
        assert 'add_mask_id' not in kwargs
        super().__init__(*args, add_mask_id=True, **kwargs)


class TokenCritic(Transformer):
    def __init__(self, *args, **kwargs):
        assert 'dim_out' not in kwargs
        super().__init__(*args, dim_out=1, **kwargs)


class SelfCritic(nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net
        self.to_pred = nn.Linear(net.dim, 1)


def cosine_schedule(t):
    return torch.cos(t * math.pi * 0.5)


class MaskGit(nn.Module):
    def __init__(self, image_size, transformer, noise_schedule=cosine_schedule, token_critic=None, self_token_critic=False, vae=None, cond_vae=None,
                 cond_image_size=None, cond_drop_prob=0.5, self_cond_prob=0.9, no_mask_token_prob=0., critic_loss_weight=1.):
        super().__init__()
        assert isinstance(transformer, MaskGitTransformer) and (vae is None or isinstance(vae, VQGanVAE))      # what @beartype enforces
        self.vae = vae
        self.cond_vae = cond_vae if cond_vae is not None else vae
        self.image_size, self.cond_image_size = image_size, cond_image_size
        self.resize_image_for_cond_image = cond_image_size is not None
        self.cond_drop_prob = cond_drop_prob
        self.transformer = transformer
        self.self_cond = transformer.self_cond
        self.mask_id = transformer.mask_id
        self.noise_schedule = noise_schedule
        self.token_critic = SelfCritic(transformer) if self_token_critic else token_critic
        self.critic_loss_weight, self.self_cond_prob, self.no_mask_token_prob = critic_loss_weight, self_cond_prob, no_mask_token_prob

    generate = _stub
    forward = _stub
