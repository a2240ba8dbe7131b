"""Two ranks run the REAL sharded decode (parallel.generate_sharded -> MaskGit.generate -> mm_generate) and the gathered ids must equal the
single-process ids of the whole batch: the Philox stream is keyed by the global sample index, no collective sits inside the decode loop.
The decode runs on the DEFAULT path (fused sampling; since round 3 it is bit-identical to the logits path, which is checked too).
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
    # codebook 65536: the 256-column persistent logits kernel is eligible at any row count, so the DEFAULT path (fused sampling) is what runs
    t = mm.MaskGitTransformer(num_tokens=65536, seq_len=64, dim=256, depth=2, dim_head=64, heads=4, t5_name='t5-small')
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
    all_ids, local = generate_sharded(mg, te, dist, seed=11, timesteps=6, fmap_size=8)          # the default path: fused sampling
    ok = local.shape[0] == 6 // world and all_ids.shape == (6, 8, 8) and mg.fused_sampling_fallbacks == 0
    if rank == 0:
        whole = mg.generate([''] * 6, text_embeds=te, seed=11, timesteps=6, fmap_size=8, return_ids=True)
        ok = ok and torch.equal(all_ids, whole)
        ok = ok and torch.equal(whole, mg.generate([''] * 6, text_embeds=te, seed=11, timesteps=6, fmap_size=8, return_ids=True, fused_sampling=False))
    else:                                                          # every rank holds the same gathered result
        ok = ok and int(all_ids.min()) >= 0 and int(all_ids.max()) < 65536
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


def _rccl_worker(q, port):
    import torch.distributed as dist
    from muse_maskgit_pytorch_amd import parallel
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    result = False
    try:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
        ids = torch.randint(0, 65536, (5, 16, 16), device='cuda')
        out = parallel.allgather_ids(ids, dist)                      # backend nccl -> the library's own communicator + mm_allgather_ids
        gather = parallel._GATHERS[parallel._gather_key(dist, None)]
        ok = torch.equal(out, ids) and out.dtype == torch.long and gather.world == 1 and isinstance(gather, parallel.IdsGather)
        out2 = parallel.allgather_ids(ids + 1 - 1, dist)             # a second call reuses the communicator
        assert len(parallel._GATHERS) == 1
        result = bool(ok and torch.equal(out2, ids))
    finally:
        q.put(result)                                                # (a worker that dies must not leave the parent waiting for its timeout)
    dist.destroy_process_group()


def test_ids_allgather_through_the_c_abi_rccl_communicator():
    """mm_comm_unique_id / mm_comm_create / mm_allgather_ids (include/muse_hip.h) on a world of one rank: RCCL resolved at run time, the
    communicator bootstrapped through the torch.distributed group, ids narrowed to int32, gathered by ncclAllGather on the current stream and
    widened again.  (More ranks than devices cannot share a GPU under RCCL; the N > 1 data path is the same call.)"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(q, _free_port()))
    p.start()
    ok = q.get(timeout=600)
    p.join(timeout=120)
    assert ok
