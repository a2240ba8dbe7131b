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
