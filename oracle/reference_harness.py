"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Imports the UNMODIFIED reference package from /root/reference in this container so that
(a) the CPU restatement in ``oracle/muse_oracle.py`` can be validated against the reference's
own executable code and (b) golden vectors can be generated (``oracle/make_golden.py``).

/root/reference does not exist on the GPU box: nothing under tests/ marked ``gpu``, nothing in
bench.py or __graft_entry__.smoke() may import this module.  The recipe follows SURVEY.md
Appendix B (import order matters).
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get('MUSE_REFERENCE_ROOT', '/root/reference')


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'muse_maskgit_pytorch'))


_loaded = {}


def load_reference():
    """Returns the imported reference package (module ``muse_maskgit_pytorch``)."""
    if 'pkg' in _loaded:
        return _loaded['pkg']
    assert reference_available(), f'{REFERENCE_ROOT} not present (reference runs only in the build container)'

    # (1) real transformers T5 symbols first -- before any torchvision stub is importable.
    from transformers import T5Config, T5Tokenizer, T5EncoderModel  # noqa: F401

    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import third_party_restatement as tp

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    # (2) stubs.  beartype: identity decorator.
    if 'beartype' not in sys.modules:
        mod('beartype', beartype=lambda f: f)

    class _Named:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x

    if 'torchvision' not in sys.modules:
        tv = mod('torchvision')
        tv.transforms = mod('torchvision.transforms', ToPILImage=_Named, Compose=_Named, Lambda=_Named,
                            Resize=_Named, RandomHorizontalFlip=_Named, CenterCrop=_Named, ToTensor=_Named)
        tv.datasets = mod('torchvision.datasets', ImageFolder=_Named)
        tv.utils = mod('torchvision.utils', make_grid=lambda *a, **k: None, save_image=lambda *a, **k: None)
        tv.models = mod('torchvision.models', vgg16=lambda *a, **k: None)
    if 'ema_pytorch' not in sys.modules:
        mod('ema_pytorch', EMA=_Named)
    # the two modules that carry hot-path arithmetic: CPU restatements (parity unpinned)
    mod('vector_quantize_pytorch', LFQ=tp.LFQ, VectorQuantize=tp.VectorQuantize)
    mea = mod('memory_efficient_attention_pytorch')
    mea.flash_attention = mod('memory_efficient_attention_pytorch.flash_attention',
                              FlashAttentionFunction=tp.FlashAttentionFunction)

    # (3) import the package; seed T5 configs so Transformer() never goes to the HF hub
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # our own package must not shadow the reference's
    import importlib
    pkg = importlib.import_module('muse_maskgit_pytorch')
    assert os.path.abspath(pkg.__file__).startswith(os.path.abspath(REFERENCE_ROOT)), pkg.__file__
    t5mod = importlib.import_module('muse_maskgit_pytorch.t5')
    t5mod.T5_CONFIGS['t5-small'] = dict(config=T5Config(d_model=512))
    t5mod.T5_CONFIGS['google/t5-v1_1-base'] = dict(config=T5Config(d_model=768))
    _loaded['pkg'] = pkg
    return pkg


def reference_modules():
    pkg = load_reference()
    import importlib
    mmp = importlib.import_module('muse_maskgit_pytorch.muse_maskgit_pytorch')
    vae = importlib.import_module('muse_maskgit_pytorch.vqgan_vae')
    att = importlib.import_module('muse_maskgit_pytorch.attend')
    return pkg, mmp, vae, att


class NoiseTape:
    """Record / replay for the reference's module-level noise sources
    (muse_maskgit_pytorch.py:390 ``uniform`` and :406 ``gumbel_noise``)."""

    def __init__(self, mmp):
        self.mmp = mmp
        self.uniform_draws = []      # raw U(0,1) tensors drawn inside gumbel_noise, in call order
        self._orig = None

    def __enter__(self):
        import torch
        mmp = self.mmp
        self._orig = (mmp.gumbel_noise,)
        tape = self

        def gumbel_noise(t):
            noise = torch.zeros_like(t).uniform_(0, 1)
            tape.uniform_draws.append(noise.clone())
            return -mmp.log(-mmp.log(noise))

        mmp.gumbel_noise = gumbel_noise
        return self

    def __exit__(self, *exc):
        self.mmp.gumbel_noise, = self._orig
        return False


class DecisionRecorder:
    """What tests/tie_aware.py needs from the reference's OWN run of MaskGit.generate, recorded without changing a value it computes
    (the recipe of oracle/make_golden_base.py, shared by make_golden_c4.py / make_golden_c5.py since round 6):
      scores_in      the tensor entering every step's `scores.topk(num_token_masked)` (muse_maskgit_pytorch.py:561),
      pred_ids       what `gumbel_sample` returned (:580; the reference's own function runs),
      argmax_margin  top-1 / top-2 gap of its perturbed logits in LOGIT units (an error d of a logit moves logits / T + gumbel by d / T),
      noise          `on_noise(uniforms)` is called with every uniform draw of gumbel_noise (:406-408: identical draw).
    Use:  with DecisionRecorder(mmp, (B, n), on_noise) as rec: mg.generate(...)   ->  rec.stacked()"""

    def __init__(self, mmp, scores_shape, on_noise=None):
        self.mmp, self.shape, self.on_noise = mmp, tuple(scores_shape), on_noise
        self.scores_in, self.pred_ids, self.argmax_margin = [], [], []
        self._last_gumbel = None

    def __enter__(self):
        import torch
        mmp, rec = self.mmp, self
        self._orig = (mmp.gumbel_noise, mmp.gumbel_sample, torch.Tensor.topk)
        orig_gs, orig_topk, log = mmp.gumbel_sample, torch.Tensor.topk, mmp.log

        def gumbel_noise(t):
            noise = torch.zeros_like(t).uniform_(0, 1)
            if rec.on_noise is not None:
                rec.on_noise(noise)
            rec._last_gumbel = -log(-log(noise))
            return rec._last_gumbel

        def gumbel_sample(t, temperature=1., dim=-1):
            pred = orig_gs(t, temperature=temperature, dim=dim)
            rec.pred_ids.append(pred.clone().to(torch.int32))
            g, rec._last_gumbel = rec._last_gumbel, None
            if temperature > 0:
                top2 = orig_topk(t / max(temperature, 1e-10) + g, 2, dim=-1).values
                rec.argmax_margin.append(((top2[..., 0] - top2[..., 1]) * temperature).clone())
            else:      # last step (temperature 0 -> 1e-10): a pure arg-max of the kept logits
                top2 = orig_topk(t, 2, dim=-1).values
                rec.argmax_margin.append((top2[..., 0] - top2[..., 1]).clone())
            return pred

        def topk_rec(self_, *a, **kw):      # `scores.topk(num_token_masked, dim=-1)` is the only 2-D float top-k of that shape in the loop
            if self_.dim() == 2 and tuple(self_.shape) == rec.shape and self_.is_floating_point():
                rec.scores_in.append(self_.clone())
            return orig_topk(self_, *a, **kw)

        mmp.gumbel_noise, mmp.gumbel_sample, torch.Tensor.topk = gumbel_noise, gumbel_sample, topk_rec
        return self

    def __exit__(self, *exc):
        import torch
        self.mmp.gumbel_noise, self.mmp.gumbel_sample, torch.Tensor.topk = self._orig
        return False

    def stacked(self):
        import torch
        assert len(self.scores_in) == len(self.pred_ids) == len(self.argmax_margin)
        return dict(scores_in=torch.stack(self.scores_in), pred_ids=torch.stack(self.pred_ids), argmax_margin=torch.stack(self.argmax_margin))
