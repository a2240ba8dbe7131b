"""world_size-2 gloo test (CPU) of the multi-GPU path's host logic: shard bounds, the ids all-gather, and that
generate_sharded() hands every rank its contiguous shard with the GLOBAL row offset (the kernel launch itself is
replaced by a recorder: there is no GPU here)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from muse_maskgit_pytorch_amd.parallel import allgather_ids, generate_sharded, shard_bounds


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeMaskGit:
    """stands in for MaskGit.generate: ids encode (global sample index, seed) so the gather can be verified"""

    def generate(self, texts, text_embeds=None, seed=None, row_offset=0, return_ids=False, **kw):
        b = text_embeds.shape[0]
        assert len(texts) == b and return_ids
        gidx = torch.arange(row_offset, row_offset + b)
        # the shard must carry exactly the rows [row_offset, row_offset+b) of the global embeds
        assert torch.equal(text_embeds[:, 0, 0].long(), gidx)
        return (gidx[:, None, None] * 100 + seed + torch.arange(4).reshape(1, 2, 2)).long()


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    total = 6
    te = torch.zeros(total, 3, 8)
    te[:, 0, 0] = torch.arange(total).float()
    all_ids, local = generate_sharded(_FakeMaskGit(), te, dist, seed=7)
    exp = (torch.arange(total)[:, None, None] * 100 + 7 + torch.arange(4).reshape(1, 2, 2)).long()
    ok = torch.equal(all_ids, exp) and all_ids.dtype == torch.long and local.shape[0] == total // world
    big = torch.full((2, 2, 2), 65535, dtype=torch.long)            # the largest codebook id survives the int32 trip
    ok = ok and torch.equal(allgather_ids(big, dist), torch.full((2 * world, 2, 2), 65535, dtype=torch.long))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds():
    assert [shard_bounds(256, r, 8) for r in range(8)] == [(32 * r, 32 * r + 32) for r in range(8)]
    spans = [shard_bounds(10, r, 4) for r in range(4)]
    assert spans == [(0, 3), (3, 6), (6, 8), (8, 10)]


def test_sharded_generate_two_ranks_gloo():
    here = os.path.dirname(os.path.abspath(__file__))
    # spawned children re-import this module by name: make tests/ and the repo root importable for them
    os.environ['PYTHONPATH'] = os.pathsep.join([here, os.path.dirname(here), os.environ.get('PYTHONPATH', '')])
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _bucket_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from muse_maskgit_pytorch_amd.parallel import GradBucketer
    torch.manual_seed(rank)
    layers = [[torch.randn(5, 7), torch.randn(3)[:2], torch.randn(2, 1, 4)] for _ in range(3)]
    ok = True
    expect = []
    for l in layers:                       # what the average must be, via plain blocking all-reduces
        for t in l:
            e = t.clone().contiguous()
            dist.all_reduce(e)
            expect.append(e / world)
    gb = GradBucketer(dist)
    for l in layers:
        gb.push(l)
    gb.finish()
    for t, e in zip([t for l in layers for t in l], expect):
        ok = ok and torch.allclose(t, e)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_bucketer_two_ranks_gloo():
    """parallel.GradBucketer (the per-layer async all-reduce the training backward issues): averaged gradients, views included."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok in res), res
