"""P prompts in flight on ONE GPU (each with its own cond/uncond HIP graph and stream): does a second independent
trajectory fill the launch gaps / tails of the first?  Same model, plan and step as bench.py.  GPU box only.
usage: python tools/bench_inflight.py [P=2] [steps=12]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import graph, synth, shard
from viditq_amd.config import loads_yaml
from viditq_amd.t2v import IDDPM

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
cfg = loads_yaml(synth.W8A8_DYNAMIC)
with torch.no_grad():
    model = synth.build_stdit(dev, depth=28)
    qnn = shard.quantize_and_distribute(model, cfg, 0, 1)
    sch = IDDPM(num_sampling_steps=100, cfg_scale=4.0)
    embeds, lens = synth.synthetic_prompts(P, dev)
    idx = list(range(sch.num_timesteps))[::-1]
    st = []
    for p in range(P):
        x = synth.synthetic_latent(p, device=dev).float()
        y = embeds["y"][p:p + 1].permute(1, 0, 2, 3, 4).reshape(2, 1, 120, 4096)
        mask = embeds["mask"][p:p + 1]
        st.append(dict(x=x, buf=torch.empty_like(x), gs=graph.GraphedSampler(qnn, y[:1], y[1:], mask),
                       stream=torch.cuda.Stream()))

    def step(j):
        i = idx[j % len(idx)]
        t_id = sch.timestep_map[i]
        for s in st:
            with torch.cuda.stream(s["stream"]):
                cond, unc = s["gs"].forward_pair(s["x"], t_id, None)
                out = sch.ddim_step(s["x"], cond, unc, i, sch.cfg_scale, 0.0, out=s["buf"])
                s["x"], s["buf"] = out, s["x"]

    for s in st:
        s["stream"].wait_stream(torch.cuda.current_stream())
    for j in range(3):
        step(j)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(3, 3 + K):
        step(j)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    assert all(torch.isfinite(s["x"]).all() for s in st)
    print("prompts in flight %d: %.2f denoising steps/s (%.2f ms per %d-prompt step)" % (P, P * K / el, el / K * 1e3, P), flush=True)
