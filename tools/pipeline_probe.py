"""Probe (tools only): does overlapping the VAE decode of generate i with the decode loop of generate i+1 (second stream) raise the whole-job rate?
Sequential = what bench.py times.  usage: python tools/pipeline_probe.py [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device('cuda', 0)
    from muse_maskgit_pytorch_amd import _lib
    _lib.require_device()
    mg, image_size = bench.build_config('c2', dev)
    B, T = 32, 18
    te = bench.synth_text(B, 32, mg.transformer.text_embed_dim).to(dev)

    def seq(i):
        return mg.generate([''] * B, timesteps=T, cond_scale=3, text_embeds=te, seed=1000 + i, return_ids='both')

    for i in range(3):
        seq(-1 - i)
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(K):
            seq(i)
        torch.cuda.synchronize()
        ts = time.perf_counter() - t0
        print(f'sequential  {B * K / ts:7.1f} images/s  {ts / K * 1e3:6.2f} ms/step', flush=True)

        side = torch.cuda.Stream()
        main_s = torch.cuda.current_stream()
        t0 = time.perf_counter()
        pend = []
        outs = []
        for i in range(K):
            ids = mg.generate([''] * B, timesteps=T, cond_scale=3, text_embeds=te, seed=1000 + i, return_ids=True, fused_sampling='deferred')
            ev = torch.cuda.Event()
            ev.record(main_s)
            st = mg.fused_status
            side.wait_event(ev)
            with torch.cuda.stream(side):
                img = mg.vae.decode_from_ids(ids)
            outs.append((ids, img))
            pend.append((ev, st))
            if len(pend) > 1:                      # read the status of the generate before this one: its loop ended long ago
                pev, pst = pend.pop(0)
                pev.synchronize()
                assert pst.tolist()[0] == 0
            if len(outs) > 2:
                outs.pop(0)
        torch.cuda.synchronize()
        tp = time.perf_counter() - t0
        print(f'pipelined   {B * K / tp:7.1f} images/s  {tp / K * 1e3:6.2f} ms/step   ({ts / tp:.3f}x)', flush=True)


if __name__ == '__main__':
    main()
