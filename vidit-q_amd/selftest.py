"""Tiny on-device self-check used by __graft_entry__.smoke(): one fused W8A8 STDiT block."""
from __future__ import annotations

import torch


def block_smoke(dev):
    from . import synth
    from .config import loads_yaml
    m = synth.build_stdit(dev, depth=1, hidden_size=64, num_heads=4, input_size=(4, 8, 8), model_max_length=12,
                          caption_channels=32, seed=0)
    qnn = synth.quantize_model(m, loads_yaml(synth.W8A8_DYNAMIC))
    assert all(b.fused_ok() for b in qnn.model.blocks)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 4, 4, 8, 8, generator=g).to(dev)
    y = (torch.randn(1, 1, 12, 32, generator=g) * 0.3).half().to(dev)
    mask = torch.ones(1, 12, dtype=torch.int64, device=dev)
    out = qnn(x, torch.tensor([500], device=dev), y, mask=mask)
    assert torch.isfinite(out).all()
    from oracle import stdit_ref as sr
    sd = {k: v.detach().cpu().float() for k, v in m.state_dict().items()
          if "weight_quantizer" not in k and "act_quantizer" not in k}
    cfgd = dict(T=4, S=16, H=4, depth=1, patch=(1, 2, 2), in_ch=4, out_ch=8, input_size=(4, 8, 8))
    ref = sr.stdit_forward(sd, cfgd, x.cpu().half().float(), torch.tensor([500]), y.cpu().float(), mask.cpu(),
                           sr.QSpec(w_bits=8))
    rel = float((out.cpu() - ref).norm() / ref.norm())
    assert rel < 5e-3, rel
    print("block smoke ok: fused W8A8 STDiT block rel-L2 vs oracle = %.2e" % rel)
