"""fp16 yardstick of the benchmarked workload on the same box: STDiT-XL/2 16x512x512 with quantization OFF.

The reference is an fp16 model whose quantization is simulated (quant_layer.py:211: F.linear on fp16 operands); what it would
run on this part is torch's fp16 Linears plus an attention library.  Two legs, both with bench.py's synthetic model, latents
and prompt, eager launches, one denoising step = cond + uncond forward-sample + the CFG / DDIM update:

  engine-fp16   this repository's model with every QuantLayer in FP state (set_quant_state(False, False)): torch F.linear
                (hipBLASLt fp16 GEMM), torch LayerNorm / modulate / GELU / residuals, the HIP attention kernels of csrc/
  torch-fp16    the same, with attention through torch.nn.functional.scaled_dot_product_attention (ROCm flash backend) -
                the nearest thing on this image to the flash-attn / xformers calls of the reference (blocks.py:169-178,
                302-304), neither of which is installed

Prints one JSON line per leg.  NOT a parity tool and not part of the product path; the numbers go beside the W8A8 line of
bench.py in profiles/ to say what the quantized HIP path buys over fp16 on MI355X.

    python tools/fp16_yardstick.py [--steps 4] [--warmup 2] [--depth 28]
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--depth", type=int, default=28)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    import viditq_amd  # noqa: F401
    from viditq_amd import synth
    from viditq_amd.config import loads_yaml
    from viditq_amd.t2v import IDDPM
    import viditq_amd.t2v.stdit as st

    cfg = loads_yaml(synth.W8A8_DYNAMIC)
    with torch.no_grad():
        model = synth.build_stdit(dev, depth=a.depth)
        qnn = synth.wrap_model(model, cfg)
        qnn.set_quant_state(False, False)
        assert not any(b.fused_ok() for b in qnn.model.blocks), "FP state must not take the quantized fused route"
        sch = IDDPM(num_sampling_steps=100, cfg_scale=4.0)
        embeds, _ = synth.synthetic_prompts(1, dev)
        x = synth.synthetic_latent(0, device=dev).float()
        y = embeds["y"][0:1].permute(1, 0, 2, 3, 4).reshape(2, 1, 120, 4096)
        y_c, y_u, mask = y[:1], y[1:], embeds["mask"][0:1]
        idx = list(range(sch.num_timesteps))[::-1]

        def run(tag, note):
            buf = torch.empty_like(x)
            cur, nxt = x.clone(), buf

            def step(j, cur, nxt):
                i = idx[j % len(idx)]
                t_id = sch.timestep_map[i]
                t = torch.full((1,), t_id, device=dev, dtype=torch.long)
                cond = qnn(cur, t, y_c, mask=mask, timestep_id=t_id)
                unc = qnn(cur, t, y_u, mask=mask, timestep_id=t_id)
                out = sch.ddim_step(cur, cond, unc, i, sch.cfg_scale, 0.0, out=nxt)
                return out, cur
            for j in range(a.warmup):
                cur, nxt = step(j, cur, nxt)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for j in range(a.warmup, a.warmup + a.steps):
                cur, nxt = step(j, cur, nxt)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            assert bool(torch.isfinite(cur).all())
            print(json.dumps({"leg": tag, "value": a.steps / el, "unit": "denoising steps/s", "ms_per_step": el / a.steps * 1e3,
                              "steps": a.steps, "warmup": a.warmup, "depth": a.depth, "dtype": "fp16 (quantization off)",
                              "note": note}), flush=True)

        run("engine-fp16", "FP-state QuantLayers: torch F.linear (hipBLASLt) + torch elementwise + csrc/ attention kernels, eager")

        # ---- torch-fp16: attention through torch SDPA
        def self_attn(self, xx):
            Bp, Np, C = xx.shape
            H, D = self.num_heads, self.head_dim
            q = self.q(xx).view(Bp, Np, H, D).transpose(1, 2)
            k = self.k(xx).view(Bp, Np, H, D).transpose(1, 2)
            v = self.v(xx).view(Bp, Np, H, D).transpose(1, 2)
            o = F.scaled_dot_product_attention(q, k, v, scale=self.scale)
            return self.proj(o.transpose(1, 2).reshape(Bp, Np, C))
        st.Attention.forward = self_attn
        cross_cls = st.MultiHeadCrossAttention
        orig_cross = cross_cls.forward

        def cross_attn(self, xx, cond, mask=None):
            # one prompt per forward here (B = 1): a plain rectangular attention over the selected prompt tokens
            B, N, C = xx.shape
            H, D = self.num_heads, self.head_dim
            q = self.q_linear(xx).view(1, B * N, H, D).transpose(1, 2)
            kv = self.kv_linear(cond).view(1, -1, 2, H, D)
            k, v = kv[:, :, 0].transpose(1, 2), kv[:, :, 1].transpose(1, 2)
            o = F.scaled_dot_product_attention(q, k, v)
            return self.proj(o.transpose(1, 2).reshape(B, N, C))
        try:
            cross_cls.forward = cross_attn
            run("torch-fp16", "as engine-fp16 with torch scaled_dot_product_attention (ROCm flash backend) for all three attentions")
        finally:
            cross_cls.forward = orig_cross


if __name__ == "__main__":
    main()
