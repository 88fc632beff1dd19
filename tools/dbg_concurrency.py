import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import viditq_amd
from viditq_amd import graph, ops, synth, shard
from viditq_amd.config import loads_yaml
from viditq_amd.t2v import IDDPM
dev = torch.device("cuda:0")
cfg = loads_yaml(synth.W8A8_DYNAMIC)
with torch.no_grad():
    model = synth.build_stdit(dev, depth=int(sys.argv[1]))
    qnn = shard.quantize_and_distribute(model, cfg, 0, 1)
    sch = IDDPM(num_sampling_steps=100)
    embeds, lens = synth.synthetic_prompts(2, dev)
    def mk(pi):
        x = synth.synthetic_latent(pi, device=dev).float()
        y = embeds["y"][pi:pi + 1].permute(1, 0, 2, 3, 4).reshape(2, 1, 120, 4096)
        return x, y[:1], y[1:], embeds["mask"][pi:pi + 1]
    res = {}
    for mode in ("seq", "conc"):
        trs = []
        for pi in range(2):
            x, yc, yu, m = mk(pi)
            st = torch.cuda.Stream() if mode == "conc" else torch.cuda.current_stream()
            trs.append(dict(x=x, gs=graph.GraphedSampler(qnn, yc, yu, m), st=st, buf=torch.empty_like(x)))
        idx = list(range(100))[::-1]
        def one(tr, j):
            i = idx[j]; t_id = sch.timestep_map[i]
            c, u = tr["gs"].forward_pair(tr["x"], t_id)
            out = sch.ddim_step(tr["x"], c, u, i, 4.0, 0.0, out=tr["buf"])
            tr["x"], tr["buf"] = out, tr["x"]
        for tr in trs:
            tr["st"].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(tr["st"]):
                one(tr, 0)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(1, 5):
            for tr in trs:
                with torch.cuda.stream(tr["st"]):
                    one(tr, j)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        res[mode] = [tr["x"].clone() for tr in trs]
        print(mode, "elapsed %.1f ms for 8 prompt-steps -> %.1f ms per prompt-step" % (el * 1e3, el * 1e3 / 8))
    for pi in range(2):
        print("prompt", pi, "seq vs conc equal:", torch.equal(res["seq"][pi], res["conc"][pi]), float(res["seq"][pi].abs().mean()))
    print("prompts differ:", not torch.equal(res["seq"][0], res["seq"][1]))
