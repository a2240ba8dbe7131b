"""CPU models of the address arithmetic of round 5's term-sharing k-loops (csrc/gemm_terms.hip and the NP forms of gemm_big.hip / gemm.hip / gemm_wide.hip):
pure Python mirrors of the HIP code, no GPU, no library.

A k-block's products xh.wh + xl.wh (+ xh.wl) run from ONE staging of its term planes.  That rests on three index maps:
  * LDS-DMA: lane l of a DMA instruction fills 16 bytes of a 128-byte LDS row; the XOR swizzle AND the plane (h / l term = which SEGMENT of the operand row the 16
    bytes come from) are part of the lane's SOURCE address.  Every (row, plane, 8-value k-group) of the step must be fetched exactly once, to where the fragment
    read of (row, plane, k-group) looks for it;
  * the two-product form stages the weights every OTHER step as [wh(64)] (gemm_terms / gemm_wide_fused) or as unswizzled 64-byte rows [wh(32)] (gemm_big / gemm);
  * the wait counts: which requests may still be in flight at the top of a step (counted vmcnt), and which buffer a request may overwrite.
The GPU tests hold the kernels to fp64 and to the concatenated-depth form; these pin the combinatorics they rely on."""
import itertools


def sw128(row, chunk):
    return row * 128 + ((chunk ^ (row & 7)) << 4)


def _dma_sources(nfw, np_terms, KS, operand):
    """{LDS byte offset inside the staged tile: (tile row, segment of the operand row, element offset inside the segment)} for one step's requests of one operand,
    all 8 waves (gemm_terms.hip ISSUE_X / ISSUE_W at step offset 0)"""
    out = {}
    rows_per_wave, ninstr = (32, 4) if operand == 'x' else (8 * nfw, nfw)
    for wid in range(8):
        for i in range(ninstr):
            for lane in range(64):
                lc = (lane & 7) ^ (lane >> 3)
                row = rows_per_wave * wid + 8 * i + (lane >> 3)
                if operand == 'x':
                    seg, k = lc >> 2, (lc & 3) * 8                          # xoff = (lc & 3) * 16 B + (lc >> 2) * KS * 2 B
                elif np_terms == 3:
                    seg, k = 2 * (lc >> 2), (lc & 3) * 8                    # woff: the l plane of a weight row is segment 2
                else:
                    seg, k = 0, lc * 8                                      # NP 2: 64 k-values of wh
                lds = wid * (ninstr * 1024) + i * 1024 + lane * 16
                assert lds not in out, 'two lanes fill the same 16 bytes'
                out[lds] = (row, seg, k)
    return out


def test_term_plane_staging_feeds_every_fragment_read():
    for nfw, np_terms in itertools.product((3, 4), (2, 3)):
        KS = 512
        x = _dma_sources(nfw, np_terms, KS, 'x')
        w = _dma_sources(nfw, np_terms, KS, 'w')
        assert len(x) == 256 * 8 and len(w) == 64 * nfw * 8                 # every 16-byte slot of both tiles is written exactly once
        for row in range(256):                                              # token fragments: lane (fr, fg) of block b reads row 16 b + fr, chunk fg (h) / 4 + fg (l)
            for fg in range(4):
                assert x[sw128(row, fg)] == (row, 0, 8 * fg)                # xh: segment 0, k-values 8 fg .. of the 32-deep block
                assert x[sw128(row, 4 + fg)] == (row, 1, 8 * fg)            # xl: segment 1, the SAME k-values
        for row in range(64 * nfw):
            for fg in range(4):
                if np_terms == 3:
                    assert w[sw128(row, fg)] == (row, 0, 8 * fg)            # wh: segment 0
                    assert w[sw128(row, 4 + fg)] == (row, 2, 8 * fg)        # wl: segment 2 (segment 1 repeats wh and is never fetched)
                else:
                    for st in range(2):                                     # step parity inside the pair: chunks 4 st .. 4 st + 3 = k-values 32 st + 8 fg ..
                        assert w[sw128(row, 4 * st + fg)] == (row, 0, 32 * st + 8 * fg)


def test_two_product_weight_rows_of_64_bytes_are_lane_linear():
    """gemm_big.hip / gemm.hip NP = 2: one DMA instruction = 16 rows of 64 bytes [wh(32)], lane l -> row l >> 2, chunk l & 3; the fragment read of lane (fr, fg)
    addresses row * 64 + fg * 16 -- one contiguous KiB per 16-row fragment (conflict-free without a swizzle)"""
    for rows_per_wave, ninstr, waves in ((16, 1, 8), (32, 2, 4)):           # gemm_big (128 weight rows, 8 waves) / gemm (128 rows, 4 waves)
        lds = {}
        for wid in range(waves):
            for i in range(ninstr):
                for lane in range(64):
                    row = rows_per_wave * wid + 16 * i + (lane >> 2)
                    a = wid * (ninstr * 1024) + i * 1024 + lane * 16
                    assert a not in lds
                    lds[a] = (row, (lane & 3) * 8)
        assert len(lds) == 128 * 4
        for row in range(128):
            for fg in range(4):
                assert lds[row * 64 + fg * 16] == (row, 8 * fg)
        for frag in range(8):                                               # the 64 lanes of one fragment read cover one contiguous KiB
            addrs = sorted((frag * 16 + fr) * 64 + fg * 16 for fr in range(16) for fg in range(4))
            assert addrs == list(range(frag * 1024, (frag + 1) * 1024, 16))


def _simulate_schedule(np_terms, KT, tiles):
    """the request / wait schedule of gemm_terms_kernel over `tiles` tiles of KT steps: returns nothing, asserts that (a) what a step reads has landed by its wait,
    (b) no request overwrites a buffer that a step not yet finished by every wave still reads (a barrier separates step s from the requests issued in step s + 1)"""
    nfw = 4
    inflight = []                       # requests in issue order: (kind, buffer, tile, index)
    landed = set()

    def issue(kind, buf, tile, idx, n):
        inflight.append(((kind, buf, tile, idx), n))

    def wait(allow):                    # s_waitcnt vmcnt(allow): the oldest requests retire until at most `allow` INSTRUCTIONS are outstanding (in order)
        total = sum(n for _, n in inflight)
        while total > allow:
            tag, n = inflight.pop(0)
            landed.add(tag)
            total -= n

    busy = {}                           # buffer -> (tile, step) of its last reader
    issue('x', 0, 0, 0, 4)
    issue('w', 0, 0, 0, nfw)
    for tile in range(tiles):
        for kt in range(KT):
            st = kt & 1
            if kt == 0:
                wait(31 if tile else 0)                 # pending epilogue stores of the previous tile are younger than this tile's first requests
            elif np_terms == 2 and st == 1 and kt + 1 < KT:
                wait(nfw)
            else:
                wait(0)
            wbuf, widx = (st, kt) if np_terms == 3 else ((kt >> 1) & 1, kt >> 1)
            assert ('x', st, tile, kt) in landed, f'tokens of step {kt} not landed'
            assert ('w', wbuf, tile, widx) in landed, f'weights of step {kt} not landed'
            busy[('x', st)] = (tile, kt)
            busy[('w', wbuf)] = (tile, kt)

            def free(kind, buf):                        # every wave has passed this step's barrier: readers of EARLIER steps are done
                assert busy.get((kind, buf), (-1, -1)) != (tile, kt), f'request into {kind}{buf} while step {kt} reads it'
            if kt + 1 < KT:
                free('x', st ^ 1)
                issue('x', st ^ 1, tile, kt + 1, 4)
                if np_terms == 3:
                    free('w', st ^ 1)
                    issue('w', st ^ 1, tile, kt + 1, nfw)
                elif st == 0 and kt + 2 < KT:
                    nb = ((kt >> 1) + 1) & 1
                    free('w', nb)
                    issue('w', nb, tile, (kt >> 1) + 1, nfw)
            elif tile + 1 < tiles:
                free('x', 0)
                free('w', 0)
                issue('x', 0, tile + 1, 0, 4)
                issue('w', 0, tile + 1, 0, nfw)
        if tile + 1 < tiles:
            issue('store', -1, tile, 0, 32)             # the epilogue's stores go out BEHIND the next tile's first requests


def test_request_and_wait_schedule_of_the_persistent_term_sharing_loop():
    for np_terms, KT in ((3, 2), (3, 16), (3, 6), (2, 4), (2, 16), (2, 8), (2, 32)):
        _simulate_schedule(np_terms, KT, tiles=3)
    # the two-product form needs KT % 4 == 0 (the launcher checks segment length % 128): with KT = 6 the last step pair sits in weight buffer 0, which the
    # next tile's first request overwrites while the last step still reads it -- the model must catch that
    try:
        _simulate_schedule(2, 6, tiles=2)
    except AssertionError:
        pass
    else:
        raise AssertionError('the schedule model does not see the buffer conflict of a two-product loop with KT % 4 == 2')


def _three_stage(KT, nreq):
    """gemm_big.hip / gemm.hip (64-token term-sharing tiles): three stages, step kt + 2 requested while step kt computes, counted wait at the top of a step.
    Asserts that a step's operands have landed when it starts and that a request never targets the stage of a step some wave may still be reading."""
    inflight, landed = [], set()

    def wait(allow):
        total = sum(n for _, n in inflight)
        while total > allow:
            tag, n = inflight.pop(0)
            landed.add(tag)
            total -= n

    inflight.append((0, nreq))
    inflight.append(('resid', 8))           # the fp32 residual prefetch goes out between the first two requests (VMEM retires in order)
    if KT > 1:
        inflight.append((1, nreq))
    wait(nreq if KT > 1 else 0)             # prologue: step 0 (and the residual) landed, step 1 may still be in flight
    assert 0 in landed and 'resid' in landed
    for kt in range(KT):
        if kt > 0:
            wait(nreq if kt + 1 < KT else 0)
        assert kt in landed, f'step {kt} not landed at its barrier'
        # barrier passed: every wave has finished step kt - 1, whose stage (kt - 1) % 3 == (kt + 2) % 3 is now free
        if kt + 2 < KT:
            assert (kt + 2) % 3 not in {kt % 3, (kt + 1) % 3}
            inflight.append((kt + 2, nreq))


def test_three_stage_counted_waits_of_the_256x128_and_64_token_kernels():
    for KT in (1, 2, 3, 4, 16, 44):
        for nreq in (4, 5, 6):              # requests per wave and step: 64-token tiles NP 2 / 256 x 128 NP 2 / both NP 3
            _three_stage(KT, nreq)
