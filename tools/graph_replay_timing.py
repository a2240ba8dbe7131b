"""tools only: one generate + VAE decode of the bench configuration captured in a hipGraph (generate(fused_sampling='deferred')) against the eager launch
sequence -- how much of the wall time is inter-launch gap.  usage: python tools/graph_replay_timing.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

DEV = 'cuda'
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
mg, _ = bench.build_models(DEV)
te = bench.synth_text(32, 32, 512).to(DEV)
kw = dict(timesteps=18, cond_scale=3., text_embeds=te, seed=7)


def timed(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eager = lambda: mg.generate([''] * 32, **kw)
for _ in range(3):
    eager()
ref = mg.generate([''] * 32, **kw)
ms_eager = timed(eager, steps)
graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
with torch.cuda.stream(side):
    with torch.cuda.graph(graph, stream=side):
        out = mg.generate([''] * 32, fused_sampling='deferred', **kw)
graph.replay(); torch.cuda.synchronize()
same = bool(torch.equal(out, ref))
ms_graph = timed(graph.replay, steps)
ms_eager2 = timed(eager, steps)
print(f'eager {ms_eager:.2f} ms ({32e3 / ms_eager:.1f} images/s)  graph replay {ms_graph:.2f} ms ({32e3 / ms_graph:.1f} images/s)  eager again {ms_eager2:.2f} ms  '
      f'replay == eager images: {same}  status {mg.fused_status.tolist()}')
