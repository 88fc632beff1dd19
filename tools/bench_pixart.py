"""PixArt-Sigma 1024x1024 (N = 4096 tokens, Lp <= 300) W4A8 / W8A8 sampling-step rate on one MI355X: the parity
configuration 5 of BASELINE.json as a timing run (not the bench.py metric).  Synthetic weights / latents /
text embeds; weights min-max per channel (data-free), activations dynamic per token; DPM-Solver++ 2M loop,
cfg 4.5, the reference's ONE batched (uncond | cond) forward per step (B = 2: token scales shared over the
pair, as base_quantizer.py:185 does).  GPU box only.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import synth
from viditq_amd.config import to_config
from viditq_amd.qdiff.models import QuantModel
from viditq_amd.t2i import DPMS_sigma, PixArtMS_XL_2

ap = argparse.ArgumentParser()
ap.add_argument("--w-bits", type=int, default=4)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--depth", type=int, default=28)
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--graph", action="store_true", help="replay the forward as one captured HIP graph (graph.GraphedModel) instead of eager launches")
a = ap.parse_args()
dev = torch.device("cuda:0")
lat = a.size // 8
Lp = 300
with torch.no_grad():
    torch.manual_seed(0)
    m = PixArtMS_XL_2(input_size=lat, model_max_length=Lp, pe_interpolation=lat / 64, dtype=torch.float16)
    if a.depth != 28:
        m.blocks = m.blocks[:a.depth]
    synth.redraw_zero_init(m, 1)
    m = m.half().to(dev).eval()
    wq = to_config(dict(n_bits=a.w_bits, per_group="channel", channel_dim=0, scale_method="min_max", round_mode="nearest",
                        mixed_precision=[4, 6, 8]))
    aq = to_config(dict(n_bits=8, per_group="token", scale_method="min_max", round_mode="nearest_ste", running_stat=False,
                        dynamic=True, sym=False, n_spatial_token=(lat // 2) ** 2, n_temporal_token=1, n_prompt=Lp,
                        smooth_quant=dict(enable=False)))
    qnn = QuantModel(m, wq, aq, model_type="pixart")
    qnn.set_module_name_for_quantizer(qnn.model)
    qnn.fp_layer_list = ["x_embedder", "t_embedder", "t_block", "y_embedder", "csize_embedder", "ar_embedder"]
    synth.init_weight_quantizers(qnn)
    qnn.set_quant_state(True, True)
    assert all(b.fused_ok() for b in qnn.model.blocks)
    g = torch.Generator().manual_seed(1)
    y = (torch.randn(1, 1, Lp, 4096, generator=g) * 0.1).half().to(dev)
    null_y = (torch.randn(1, 1, Lp, 4096, generator=g) * 0.1).half().to(dev)
    mask = torch.zeros(1, Lp, dtype=torch.int64, device=dev)
    mask[0, :180] = 1
    z = torch.randn(1, 4, lat, lat, generator=g).to(dev)
    from viditq_amd.graph import GraphedModel
    solver = DPMS_sigma(GraphedModel(qnn.forward_with_dpmsolver, qnn=qnn) if a.graph else qnn.forward_with_dpmsolver, condition=y, uncondition=null_y, cfg_scale=4.5,
                        model_kwargs=dict(data_info=None, mask=mask))
    solver.sample(z, steps=2, order=2)                     # warm-up: packing, caches
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = solver.sample(z, steps=a.steps, order=2)
    t_host = time.perf_counter() - t0                      # host done enqueuing (the GPU may still be running)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    print(json.dumps({"workload": "PixArt-Sigma %dx%d W%dA8, %d tokens, Lp %d, DPM-Solver++ 2M, cfg 4.5, batched uncond|cond forward" % (
        a.size, a.size, a.w_bits, (lat // 2) ** 2, Lp), "steps": a.steps, "depth": a.depth, "hip_graph": a.graph,
        "steps_per_s": a.steps / el, "ms_per_step": el / a.steps * 1e3,
        "host_enqueue_ms_per_step": t_host / a.steps * 1e3, "status_word": qnn.check_status()}))
