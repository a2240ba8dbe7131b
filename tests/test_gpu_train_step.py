"""mm_train_step (csrc/train_step.hip, include/muse_hip.h): the training step as ONE C call -- forward with saved activations, cross-entropy on the
labelled rows and the whole hand-written backward -- against the operator-by-operator driver of training.py (whose gradients are pinned to the oracle's
autograd in tests/test_gpu_model.py / test_gpu_fuzz_train.py / test_gpu_base_size.py): the same operators in the same order, so the loss and EVERY
gradient must be bit-identical.  Reference: MaskGit.forward, muse_maskgit_pytorch.py:623-741."""
import os

import pytest
import torch

import muse_maskgit_pytorch_amd as mm
from muse_maskgit_pytorch_amd import _lib as L
from muse_maskgit_pytorch_amd import training

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _grads(tr, ids, te, labels, py):
    for p in tr.parameters():
        p.grad = None
    if py:
        os.environ['MM_TRAIN_PY'] = '1'
    try:
        loss = tr(ids, text_embeds=te, labels=labels, ignore_index=-1)
        loss.backward()
    finally:
        os.environ.pop('MM_TRAIN_PY', None)
    return loss.detach().clone(), {n: p.grad.detach().clone() for n, p in tr.named_parameters() if p.grad is not None}


@pytest.mark.parametrize('dim,heads,depth,V,n,B,L', [(128, 2, 2, 512, 64, 2, 7), (512, 8, 1, 1024, 128, 3, 5), (256, 4, 2, 4096, 256, 1, 9)])
def test_train_step_equals_the_operator_by_operator_driver(dim, heads, depth, V, n, B, L):
    torch.manual_seed(dim + n)
    tr = mm.MaskGitTransformer(num_tokens=V, seq_len=n, dim=dim, depth=depth, dim_head=64, heads=heads, t5_name='t5-small').to(DEV).train()
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, V + 1, (B, n), generator=g)
    labels = torch.randint(0, V, (B, n), generator=g)
    labels[torch.rand(B, n, generator=g) < 0.4] = -1
    te = torch.randn(B, L, 512, generator=g)
    if B > 1:
        te[1, L // 2:] = 0.
    ids, labels, te = ids.to(DEV), labels.to(DEV), te.to(DEV)
    assert training._c_step_eligible(tr, ids, te, None, False, None, None, None)
    loss_c, g_c = _grads(tr, ids, te, labels, py=False)
    loss_p, g_p = _grads(tr, ids, te, labels, py=True)
    assert torch.equal(loss_c, loss_p), (loss_c.item(), loss_p.item())
    assert set(g_c) == set(g_p) and len(g_c) > 10
    bad = [k for k in g_p if not torch.equal(g_c[k], g_p[k])]
    assert not bad, f'gradients differ from the operator-by-operator driver: {bad[:6]}'
    loss_c2, g_c2 = _grads(tr, ids, te, labels, py=False)      # and the call repeats bit for bit (no atomics, a reused workspace)
    assert torch.equal(loss_c, loss_c2) and all(torch.equal(g_c[k], g_c2[k]) for k in g_c)


def test_train_step_lowers_the_loss_through_a_torch_optimizer():
    torch.manual_seed(0)
    tr = mm.MaskGitTransformer(num_tokens=512, seq_len=64, dim=128, depth=2, dim_head=64, heads=2, t5_name='t5-small').to(DEV).train()
    opt = torch.optim.AdamW(tr.parameters(), lr=3e-3)
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, 513, (4, 64), generator=g).to(DEV)
    labels = torch.randint(0, 512, (4, 64), generator=g).to(DEV)
    te = torch.randn(4, 6, 512, generator=g).to(DEV)
    losses = []
    for _ in range(8):
        opt.zero_grad(set_to_none=True)
        loss = tr(ids, text_embeds=te, labels=labels, ignore_index=-1)
        (2.0 * loss).backward()                                   # (a scaled loss: the kept gradients are multiplied by the incoming one)
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0] - 0.5, losses


def test_train_step_side_stream_keeps_pace_with_parameter_updates():
    """The bf16 operand copies / transposes of the parameters, the dW GEMMs and the leaf reductions run on the library's second stream (csrc/train_step.hip): over several optimizer steps --
    parameters change between the calls, the workspace is reused -- the C step with the side stream, the C step on one stream (mm_debug_set2(4)) and the
    operator-by-operator driver must walk the same trajectory bit for bit."""
    import copy
    torch.manual_seed(5)
    tr0 = mm.MaskGitTransformer(num_tokens=1024, seq_len=64, dim=256, depth=3, dim_head=64, heads=4, t5_name='t5-small').to(DEV).train()
    g = torch.Generator().manual_seed(2)
    batches = [(torch.randint(0, 1025, (4, 64), generator=g).to(DEV), torch.randint(0, 1024, (4, 64), generator=g).to(DEV), torch.randn(4, 6, 512, generator=g).to(DEV))
               for _ in range(4)]

    def run(mode):
        tr = copy.deepcopy(tr0)
        opt = torch.optim.SGD(tr.parameters(), lr=0.05)
        os.environ.pop('MM_TRAIN_PY', None)
        if mode == 'py':
            os.environ['MM_TRAIN_PY'] = '1'
        if mode == 'one_stream':
            L.lib().mm_debug_set2(4)
        try:
            losses = []
            for ids, labels, te in batches:
                opt.zero_grad(set_to_none=True)
                loss = tr(ids, text_embeds=te, labels=labels, ignore_index=-1)
                loss.backward()
                opt.step()
                losses.append(loss.detach().clone())
        finally:
            os.environ.pop('MM_TRAIN_PY', None)
            L.lib().mm_debug_set2(0)
        torch.cuda.synchronize()
        return torch.stack(losses), {n: p.detach().clone() for n, p in tr.named_parameters()}

    l_side, p_side = run('side')
    l_one, p_one = run('one_stream')
    l_py, p_py = run('py')
    assert torch.equal(l_side, l_py) and torch.equal(l_one, l_py), (l_side.tolist(), l_one.tolist(), l_py.tolist())
    assert all(torch.equal(p_side[k], p_py[k]) and torch.equal(p_one[k], p_py[k]) for k in p_py)
