"""Multi-GPU path: one process per GPU, the batch is sharded, the decode loop needs no communication (every reduction
in MaskGit.generate is over the vocabulary or token axis of ONE sample, muse_maskgit_pytorch.py:561,576,580,603), and the
generated token grids are exchanged with a single all-gather (RCCL over xGMI when the backend is 'nccl').  The
reference has no distributed inference code; this is the north star's requirement."""
import os

import torch


def shard_bounds(total, rank, world):
    """contiguous batch shard of `rank`: the first total % world ranks get one extra sample."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class IdsGather:
    """The path's one collective through the C ABI (include/muse_hip.h mm_comm_* / mm_allgather_ids): an RCCL communicator of the library's own,
    one rank per GPU, bootstrapped like ncclCommInitRank -- rank 0's 128-byte unique id is shared through the torch.distributed group that
    already exists for the rendezvous -- and one ncclAllGather of int32 ids on the current stream per call (no host synchronisation)."""

    def __init__(self, dist, group=None):
        import ctypes as C
        from . import _lib as L
        self.L, self.C = L, C
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.ptr = C.c_void_p()
        uid = (C.c_ubyte * 128)()
        box = [None]
        if self.rank == 0:          # a failure here must still reach the broadcast below, or the other ranks would wait in it forever
            try:
                L.check(L.lib().mm_comm_unique_id(uid), 'mm_comm_unique_id')
                box = [bytes(uid)]
            except Exception as e:
                box = [None]
                self._err0 = repr(e)
        dist.broadcast_object_list(box, src=0, group=group)
        if box[0] is None:
            raise L.MuseHipError('mm_comm_unique_id failed on rank 0' + (': ' + getattr(self, '_err0', '') if self.rank == 0 else ''))
        L.check(L.lib().mm_comm_create(box[0], self.rank, self.world, C.byref(self.ptr)), 'mm_comm_create')
        self.ws = None

    def __call__(self, ids):
        L = self.L
        send = ids.to(torch.long).contiguous()
        count = send.numel()
        out = torch.empty((self.world * send.shape[0],) + tuple(send.shape[1:]), dtype=torch.long, device=send.device)
        wsb = L.lib().mm_allgather_ids_workspace_bytes(self.ptr, count)
        if self.ws is None or self.ws.numel() < wsb or self.ws.device != send.device:
            self.ws = torch.empty(int(wsb), dtype=torch.uint8, device=send.device)
        L.check(L.lib().mm_allgather_ids(self.ptr, L.stream(), L.ptr(send), count, L.ptr(out), L.ptr(self.ws), self.ws.numel()), 'mm_allgather_ids')
        return out

    def __del__(self):
        try:
            if self.ptr:
                self.L.lib().mm_comm_destroy(self.ptr)
        except Exception:
            pass


_GATHERS = {}


def _gather_key(dist, group):
    """identity of a communicator: the group's GLOBAL rank tuple + the process-group object behind it (id() values are recycled once a group is
    destroyed; torch gives every init_process_group / new_group a fresh object, whose id is only trusted while a weak reference to it is alive)"""
    import weakref
    pg = group if group is not None else dist.distributed_c10d._get_default_group()
    ranks = tuple(dist.get_process_group_ranks(pg))
    for k in [k for k, ref in _GATHER_REFS.items() if ref() is None]:      # groups that no longer exist: drop (and destroy) their communicators
        _GATHER_REFS.pop(k, None)
        _GATHERS.pop(k, None)
    key = (id(pg), ranks)
    if key not in _GATHER_REFS:
        try:
            _GATHER_REFS[key] = weakref.ref(pg)
        except TypeError:                                                   # not weak-referenceable: never re-use across groups
            _GATHERS.pop(key, None)
            _GATHER_REFS[key] = lambda: True
    return key


_GATHER_REFS = {}


def allgather_ids(ids, dist, group=None):
    """ids int64 (b, f, f), same b on every rank -> (world*b, f, f).  Final ids are < codebook_size <= 65536 (the last
    decode step leaves no mask id, mmp.py:584-588), so they travel as int32: 4 bytes/token, 32 KiB per rank at C2.
    Backend 'nccl' (one rank per GPU): the library's own RCCL all-gather (IdsGather, C ABI); backend 'gloo' (CPU rendezvous of the tests,
    or ranks sharing one device): torch.distributed through host memory."""
    world = dist.get_world_size(group)
    if dist.get_backend(group) == 'nccl' and os.environ.get('MM_IDS_GATHER', 'rccl') != 'torch':
        key = _gather_key(dist, group)
        if key not in _GATHERS:
            try:
                _GATHERS[key] = IdsGather(dist, group)
            except Exception as e:      # a collective, not compute: torch.distributed's RCCL all-gather is an equivalent transport (recorded in .last_transport)
                _GATHERS[key] = None
                allgather_ids.last_error = repr(e)
            # every rank must take the SAME transport (a rank whose communicator failed next to ranks whose did not would wait in a different
            # collective forever): agree through the group that exists anyway
            ok = torch.tensor([1 if _GATHERS[key] is not None else 0], dtype=torch.int32, device=ids.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok.item()) == 0 and _GATHERS[key] is not None:
                _GATHERS[key] = None
                allgather_ids.last_error = 'another rank could not create the library-owned communicator'
        if _GATHERS[key] is not None:
            allgather_ids.last_transport = 'mm_allgather_ids (library-owned RCCL communicator)'
            return _GATHERS[key](ids)
    allgather_ids.last_transport = 'torch.distributed.all_gather_into_tensor'
    send = ids.to(torch.int32).contiguous()
    dev = send.device
    if dist.get_backend(group) == 'gloo':                     # CPU rendezvous (tests, or ranks sharing one device): 32 KiB through host memory
        send = send.cpu()
    out = torch.empty((world * send.shape[0],) + tuple(send.shape[1:]), dtype=torch.int32, device=send.device)
    dist.all_gather_into_tensor(out, send, group=group)      # concatenated along dim 0 in rank order
    return out.to(device=dev, dtype=torch.long)


allgather_ids.last_transport = None
allgather_ids.last_error = None


def generate_sharded(maskgit, text_embeds, dist, seed, **kw):
    """Every rank passes the GLOBAL text_embeds (B_total, L, d); it decodes its contiguous shard with the Philox stream
    keyed by the GLOBAL sample index (row_offset), so the gathered ids equal the single-GPU result for the same seed.
    Requires B_total % world == 0 (all_gather_into_tensor needs equal shards).  Returns (all ids, local ids)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    total = text_embeds.shape[0]
    assert total % world == 0, 'global batch must divide evenly across ranks'
    lo, hi = shard_bounds(total, rank, world)
    local = maskgit.generate([''] * (hi - lo), text_embeds=text_embeds[lo:hi], seed=seed, row_offset=lo, return_ids=True, **kw)
    return allgather_ids(local, dist), local


class GradBucketer:
    """Data-parallel gradient averaging for the hand-written backward (training.py): the backward pushes each layer's fp32
    gradients as soon as they exist; a push packs them into ONE contiguous bucket (a transformer layer = 4.2 M parameters =
    17 MB at C2 -- xGMI is point-to-point, ~153 GB/s per link, so a few large all-reduces beat many small ones) and launches an
    asynchronous all-reduce (RCCL when the backend is 'nccl') that overlaps with the backward of the layers below; finish()
    waits, divides by the world size and scatters the buckets back into the gradient tensors.  The reference has no
    data-parallel code for MaskGit ('plain DDP'); torch's DistributedDataParallel also works on this path, but it only
    sees the gradients when the whole backward has returned, i.e. without overlap."""

    def __init__(self, dist, group=None):
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)
        self.pending = []

    def push(self, tensors):
        tensors = [t for t in tensors if t is not None]
        if not tensors or self.world == 1:
            return
        flat = torch.cat([t.reshape(-1).float() for t in tensors])
        work = self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.pending.append((work, flat, tensors))

    def finish(self):
        for work, flat, tensors in self.pending:
            work.wait()
            flat.div_(self.world)
            off = 0
            for t in tensors:
                n = t.numel()
                t.copy_(flat[off:off + n].reshape(t.shape))
                off += n
        self.pending = []
