"""Two ranks run the REAL sharded decode (parallel.generate_sharded -> MaskGit.generate -> mm_generate) and the gathered ids must equal the
single-process ids of the whole batch: the Philox stream is keyed by the global sample index, no collective sits inside the decode loop.
With >= 2 visible devices every rank owns one and the gather is RCCL ('nccl'); on a one-GPU box both ranks share device 0 and the 32 KiB
of ids travel over gloo -- the decode path under test is the same."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(dev):
    import muse_maskgit_pytorch_amd as mm
    torch.manual_seed(21)
    t = mm.MaskGitTransformer(num_tokens=8192, seq_len=64, dim=256, depth=2, dim_head=64, heads=4, t5_name='t5-small')
    with torch.no_grad():
        t.to_logits.weight.mul_(6.)
    return mm.MaskGit(image_size=128, transformer=t, vae=None).to(dev)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from muse_maskgit_pytorch_amd.parallel import generate_sharded
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(rank % ndev)
    dist.init_process_group('nccl' if ndev >= world else 'gloo', rank=rank, world_size=world)
    dev = torch.device('cuda', rank % ndev)
    mg = _build(dev)
    g = torch.Generator().manual_seed(5)
    te = torch.randn(6, 7, 512, generator=g).to(dev)              # the GLOBAL batch, identical on every rank
    all_ids, local = generate_sharded(mg, te, dist, seed=11, timesteps=6, fmap_size=8, fused_sampling=False)
    ok = local.shape[0] == 6 // world and all_ids.shape == (6, 8, 8)
    if rank == 0:
        whole = mg.generate([''] * 6, text_embeds=te, seed=11, timesteps=6, fmap_size=8, return_ids=True, fused_sampling=False)
        ok = ok and torch.equal(all_ids, whole)
    else:                                                          # every rank holds the same gathered result
        ok = ok and int(all_ids.min()) >= 0 and int(all_ids.max()) < 8192
    q.put((rank, bool(ok), dist.get_backend()))
    dist.barrier()
    dist.destroy_process_group()


def test_real_sharded_generate_on_two_ranks():
    here = os.path.dirname(os.path.abspath(__file__))
    os.environ['PYTHONPATH'] = os.pathsep.join([here, os.path.dirname(here), os.environ.get('PYTHONPATH', '')])
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    print('[distributed] backend of the ids gather:', res[0][2], '(', torch.cuda.device_count(), 'visible devices )')
    assert [(r, ok) for r, ok, _ in res] == [(0, True), (1, True)]
