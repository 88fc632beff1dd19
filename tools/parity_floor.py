"""Round 5: the parity FLOOR at the full-size configurations, measured instead of argued.

CPU only (authoring container or any host): the fp32 oracle (oracle/stdit_ref.py, oracle/pixart_ref.py) runs the SAME
full-size, full-depth forwards whose outputs the imported reference left in tests/golden/*_full_ref.npz, and its rel-L2
distance from the reference's fp32 mode is recorded at every stored checkpoint next to (a) the reference's own fp16-mode
drift (in the file) and (b) the HIP path's figure of the last GPU session (profiles/r0N_parity.json, if present).  Two
fp32 implementations of the same arithmetic differ only in summation order; what they are apart after 28 blocks is the
floor no fp16-storage implementation can be expected to beat.

    python tools/parity_floor.py [stdit_full] [stdit_full_w4a8] [sigma1024_full]   ->  profiles/r05_parity_floor.json

Nothing here touches /root/reference: the golden files are data, the weights and inputs come from seeds (tests/helpers.py).
Cross attention in every fixture ran a RESTATED xformers (oracle/ref_import.py:127-143; third party, absent) - the figures
below carry that caveat."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import load_npz, rel_l2, seeded_act_scale, sigma1024_inputs, stdit_full_inputs  # noqa: E402
from oracle import pixart_ref as pr  # noqa: E402
from oracle import stdit_ref as sr  # noqa: E402
from test_oracle_golden_cpu import _seeded_sd  # noqa: E402  (plain-torch module skeletons with the reference's names)

OUT = os.path.join(ROOT, "profiles", "r05_parity_floor.json")


def _hip_figures():
    """place -> the HIP path's rel-L2 against the reference fp32 mode from the newest committed GPU parity record"""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_parity.json")):
        m = re.match(r"r(\d+)_parity\.json$", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    if best is None:
        return {}, None
    with open(best[1]) as fh:
        rec = json.load(fh)
    recs = rec.get("records", rec)
    out = {}
    for k, v in recs.items():
        if isinstance(v, dict) and "vs_ref_fp32" in v:
            out[k] = v["vs_ref_fp32"]
    return out, os.path.basename(best[1])


def _row(res, hip, place, got, gold, gold16):
    e = rel_l2(got, gold)
    r16 = rel_l2(gold16, gold)
    res[place] = {"oracle_fp32_vs_ref_fp32": e, "ref_fp16_vs_ref_fp32": r16, "hip_vs_ref_fp32": hip.get(place),
                  "hip_over_floor": (hip[place] / e) if place in hip and e > 0 else None}
    print("%-40s oracle %.3e   reference fp16 mode %.3e   HIP %s" %
          (place, e, r16, "%.3e" % hip[place] if place in hip else "-"), flush=True)


def stdit_geo(sd, depth):
    from viditq_amd.t2v import STDiT
    geo = STDiT(input_size=(16, 64, 64), depth=1, hidden_size=1152, num_heads=16, model_max_length=120, caption_channels=4096)
    sd["pos_embed"], sd["pos_embed_temporal"] = geo.pos_embed.half().float(), geo.pos_embed_temporal.half().float()
    return dict(T=16, S=1024, H=16, depth=depth, patch=(1, 2, 2), in_ch=4, out_ch=8, input_size=(16, 64, 64))


def run_stdit_full(res, hip):
    g = load_npz("stdit_full_ref.npz")
    seed = int(g["seed"])
    sd = _seeded_sd("stdit", seed, depth=28, Cc=4096, L=120)
    cfg = stdit_geo(sd, 28)
    x, y, mask, t = stdit_full_inputs(seed)
    out, blocks = sr.stdit_forward(sd, cfg, x, t, y, mask, sr.QSpec(w_bits=8), return_blocks=True)
    for i in (0, 13, 27):
        _row(res, hip, "stdit_full/block%d" % i, blocks[i][:, ::256], g["block%d" % i], g["block%d_ref_fp16" % i])
    _row(res, hip, "stdit_full/out", out[:, :, :, ::2, ::2], g["out"], g["out_ref_fp16"])


def run_stdit_full_w4a8(res, hip):
    g = load_npz("stdit_full_w4a8_ref.npz")
    seed = int(g["seed"])
    sd = _seeded_sd("stdit", seed, depth=28, Cc=4096, L=120)
    cfg = stdit_geo(sd, 28)
    x, y, mask, _ = stdit_full_inputs(seed)
    names = [k[:-len(".weight")] for k in sd if k.startswith("blocks.") and k.endswith(".weight") and sd[k].dim() == 2]
    act = {n: seeded_act_scale(n, sd[n + ".weight"].shape[1], seed) for n in names}
    assert len(act) == 13 * 28, len(act)
    for case, tv, mp in (("w4a8_t721", 721, {}), ("w4a8_mp_t300", 300, {n: (8 if ".mlp." in n else 4) for n in names})):
        spec = sr.QSpec(w_bits=4, act_scale=act, alpha=[0.11, 0.11], timerange=[[0, 500], [501, 1000]], layer_w_bits=dict(mp))
        out, blocks = sr.stdit_forward(sd, cfg, x, torch.tensor([tv]), y, mask, spec, return_blocks=True)
        _row(res, hip, "stdit_full_w4a8/%s_block27" % case, blocks[27][:, ::512], g[case + "_block27"], g[case + "_block27_ref_fp16"])
        _row(res, hip, "stdit_full_w4a8/%s_out" % case, out[:, :, ::2, ::2, ::2], g[case + "_out"], g[case + "_out_ref_fp16"])


def run_sigma1024_full(res, hip):
    from viditq_amd import t2i
    g = load_npz("sigma1024_full_ref.npz")
    seed = int(g["seed"])
    sd = _seeded_sd("pixart", seed, depth=28, Cc=4096, L=300)
    geo = t2i.PixArtMS(input_size=128, depth=1, hidden_size=1152, num_heads=16, model_max_length=300, caption_channels=4096,
                       pe_interpolation=2.0, dtype=torch.float16)
    x, y, mask, t = sigma1024_inputs(seed)
    geo.h = geo.w = x.shape[-1] // 2
    pe = geo._pos_embed(torch.device("cpu"), torch.float32)     # the sin-cos table as the model forms it (PixArtMS.py:179-183)
    out, blocks = pr.pixart_forward(sd, dict(H=16, depth=28, patch=2, out_ch=8), x, t, y, mask,
                                    sr.QSpec(w_bits=4, fp_layers=pr.T2I_FP_LAYERS), pe, return_blocks=True)
    for i in (0, 27):
        _row(res, hip, "sigma1024_full/block%d" % i, blocks[i][:, ::128], g["block%d" % i], g["block%d_ref_fp16" % i])
    _row(res, hip, "sigma1024_full/out", out[:, :, ::2, ::2], g["out"], g["out_ref_fp16"])


def main():
    which = [a for a in sys.argv[1:] if not a.startswith("-")] or ["stdit_full", "stdit_full_w4a8", "sigma1024_full"]
    hip, src = _hip_figures()
    res = {}
    if os.path.exists(OUT):
        with open(OUT) as f:
            res = json.load(f).get("records", {})
    torch.manual_seed(0)
    with torch.no_grad():
        for w in which:
            t0 = time.time()
            {"stdit_full": run_stdit_full, "stdit_full_w4a8": run_stdit_full_w4a8, "sigma1024_full": run_sigma1024_full}[w](res, hip)
            print("[%s] %.0f s on %d torch threads" % (w, time.time() - t0, torch.get_num_threads()), flush=True)
            with open(OUT, "w") as f:
                json.dump({"what": "fp32 oracle vs the reference's fp32 mode at the full-size golden checkpoints (the parity "
                                   "floor), beside the reference's own fp16-mode drift and the HIP path's last recorded figure",
                           "hip_figures_from": src, "torch": torch.__version__,
                           "cross_attention": "xformers restated in the fixtures (oracle/ref_import.py:127-143)",
                           "records": res}, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
