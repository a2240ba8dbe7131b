"""Text-encoder boundary (reference t5.py:59-99).  Out of the hot path: stock `transformers` T5 when its weights
are available; the hot path's input contract is the OUTPUT layout -- fp32 (B, L, d_model) with padded positions
exactly zero (t5.py:93), from which Transformer.forward re-derives the key mask (muse_maskgit_pytorch.py:304)."""
from typing import List

import torch

MAX_LENGTH = 256
DEFAULT_T5_NAME = 'google/t5-v1_1-base'
T5_CONFIGS = {}

# d_model of the public T5 checkpoints, so constructing a Transformer needs no hub access
_KNOWN_DIMS = {
    't5-small': 512, 't5-base': 768, 't5-large': 1024, 't5-3b': 1024, 't5-11b': 1024,
    'google/t5-v1_1-small': 512, 'google/t5-v1_1-base': 768, 'google/t5-v1_1-large': 1024,
    'google/t5-v1_1-xl': 2048, 'google/t5-v1_1-xxl': 4096,
}


def get_encoded_dim(name):
    cfg = T5_CONFIGS.get(name, {})
    if 'config' in cfg:
        return cfg['config'].d_model
    if 'model' in cfg:
        return cfg['model'].config.d_model
    if name in _KNOWN_DIMS:
        return _KNOWN_DIMS[name]
    from transformers import T5Config
    config = T5Config.from_pretrained(name)
    T5_CONFIGS[name] = dict(config=config)
    return config.d_model


def get_model_and_tokenizer(name):
    from transformers import T5EncoderModel, T5Tokenizer
    cfg = T5_CONFIGS.setdefault(name, {})
    if 'model' not in cfg:
        cfg['model'] = T5EncoderModel.from_pretrained(name)
    if 'tokenizer' not in cfg:
        cfg['tokenizer'] = T5Tokenizer.from_pretrained(name)
    return cfg['model'], cfg['tokenizer']


@torch.no_grad()
def t5_encode_text(texts: List[str], name=DEFAULT_T5_NAME, output_device=None):
    t5, tokenizer = get_model_and_tokenizer(name)
    if torch.cuda.is_available():
        t5 = t5.cuda()
    device = next(t5.parameters()).device
    encoded = tokenizer.batch_encode_plus(texts, return_tensors='pt', padding='longest', max_length=MAX_LENGTH, truncation=True)
    input_ids = encoded.input_ids.to(device)
    attn_mask = encoded.attention_mask.to(device)
    t5.eval()
    encoded_text = t5(input_ids=input_ids, attention_mask=attn_mask).last_hidden_state.detach()
    encoded_text = encoded_text.masked_fill(~attn_mask.bool()[..., None], 0.)
    if output_device is not None:
        encoded_text = encoded_text.to(output_device)
    return encoded_text


# ------------------------------------------------------------------------------------------------ the explicit (embeds, mask) contract
# The reference hands the transformer ONE tensor and lets it re-derive the key-padding mask from exact zeros (muse_maskgit_pytorch.py:304:
# `(text_embeds != 0).any(dim = -1)`, relying on t5.py:93 having zero-filled the padding).  That is an implicit pair; these helpers make it
# explicit so a caller with its own encoder (or cached embeddings) can produce / check what the hot path consumes:
#   embeds  fp32 (B, L, d_model), L <= MAX_LENGTH, every padded position exactly 0
#   mask    bool (B, L), True = attend; a kept position must have at least one non-zero feature (otherwise the in-kernel derivation, which is
#           the reference's, would drop it) -- violated pairs are rejected instead of silently changing the attention pattern.

def derive_text_mask(text_embeds: torch.Tensor) -> torch.Tensor:
    """the mask the hot path derives (mm_transformer_context == muse_maskgit_pytorch.py:304)"""
    return (text_embeds != 0).any(dim=-1)


def pack_text_condition(text_embeds: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """(embeds, mask) -> the single tensor `text_embeds=` takes: padded positions zero-filled (t5.py:93).  Raises if the pair cannot be
    represented, i.e. if a position the mask keeps is all-zero."""
    if text_embeds.dim() != 3 or mask.shape != text_embeds.shape[:2]:
        raise ValueError(f'expected embeds (B, L, d) and mask (B, L), got {tuple(text_embeds.shape)} and {tuple(mask.shape)}')
    if text_embeds.shape[1] > MAX_LENGTH:
        raise ValueError(f'text length {text_embeds.shape[1]} exceeds MAX_LENGTH {MAX_LENGTH} (t5.py:16)')
    mask = mask.bool()
    out = text_embeds.float().masked_fill(~mask[..., None], 0.)
    if not torch.equal(derive_text_mask(out), mask):
        bad = (derive_text_mask(out) != mask).nonzero()[0].tolist()
        raise ValueError(f'position {bad} is kept by the mask but its embedding is all-zero: the reference contract (zeros == padding) cannot express it')
    return out


def unpack_text_condition(text_embeds: torch.Tensor):
    """the single-tensor form -> the explicit (embeds, mask) pair"""
    return text_embeds, derive_text_mask(text_embeds)


@torch.no_grad()
def t5_encode_text_pair(texts: List[str], name=DEFAULT_T5_NAME, output_device=None):
    """`t5_encode_text` returning the explicit pair."""
    return unpack_text_condition(t5_encode_text(texts, name=name, output_device=output_device))
