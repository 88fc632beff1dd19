"""Rounds 5-6: the parity FLOOR at the full-size configurations, measured instead of argued.

CPU only (authoring container or any host): the fp32 oracle (oracle/stdit_ref.py, oracle/pixart_ref.py) runs the SAME
full-size, full-depth forwards whose outputs the imported reference left in tests/golden/*_full_ref.npz, and its rel-L2
distance from the reference's fp32 mode is recorded at every stored checkpoint next to (a) the reference's own fp16-mode
drift (in the file) and (b) the HIP path's figure of the last GPU session (profiles/r0N_parity.json, if present).  Two
fp32 implementations of the same arithmetic differ only in summation order; what they are apart after 28 blocks is the
floor no fp16-storage implementation can be expected to beat.

    python tools/parity_floor.py [stdit_full] [stdit_full_w4a8] [sigma1024_full] [stdit_full_ddim2] [stdit_full_static]
                                 [alpha256_full]   ->  profiles/r06_parity_floor.json (+ r06_parity_floor_log.txt: every line
                                 printed, the sha256 of every golden file read and of the records written)

Round 6 adds the places that had only the reference's fp16-mode drift as a bound: the two-step DDIM loop, the static
tensor-wise plan, and the alpha-256 trajectory after 10 and 20 solver steps.

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

OUT = os.path.join(ROOT, "profiles", "r06_parity_floor.json")
LOG = os.path.join(ROOT, "profiles", "r06_parity_floor_log.txt")
_LOG_LINES, _GOLDEN_USED = [], set()


def _say(msg):
    print(msg, flush=True)
    _LOG_LINES.append(msg)


def _npz(name):
    _GOLDEN_USED.add(name)
    return load_npz(name)


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
    _say("%-40s oracle %.3e   reference fp16 mode %.3e   HIP %s" %
         (place, e, r16, "%.3e" % hip[place] if place in hip else "-"))


def stdit_geo(sd, depth):
    from viditq_amd.t2v import STDiT
    geo = STDiT(input_size=(16, 64, 64), depth=1, hidden_size=1152, num_heads=16, model_max_length=120, caption_channels=4096)
    sd["pos_embed"], sd["pos_embed_temporal"] = geo.pos_embed.half().float(), geo.pos_embed_temporal.half().float()
    return dict(T=16, S=1024, H=16, depth=depth, patch=(1, 2, 2), in_ch=4, out_ch=8, input_size=(16, 64, 64))


def run_stdit_full(res, hip):
    g = _npz("stdit_full_ref.npz")
    seed = int(g["seed"])
    sd = _seeded_sd("stdit", seed, depth=28, Cc=4096, L=120)
    cfg = stdit_geo(sd, 28)
    x, y, mask, t = stdit_full_inputs(seed)
    out, blocks = sr.stdit_forward(sd, cfg, x, t, y, mask, sr.QSpec(w_bits=8), return_blocks=True)
    for i in (0, 13, 27):
        _row(res, hip, "stdit_full/block%d" % i, blocks[i][:, ::256], g["block%d" % i], g["block%d_ref_fp16" % i])
    _row(res, hip, "stdit_full/out", out[:, :, :, ::2, ::2], g["out"], g["out_ref_fp16"])


def run_stdit_full_w4a8(res, hip):
    g = _npz("stdit_full_w4a8_ref.npz")
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
    g = _npz("sigma1024_full_ref.npz")
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


def run_stdit_full_ddim2(res, hip):
    """the reference's IDDPM(2 steps, cfg 4.0, DDIM eta 0) + forward_with_cfg around the W8A8 model (make_golden.py::
    stdit_full_ddim2): four full-size oracle forwards"""
    from helpers import stdit_full_null_y
    g = _npz("stdit_full_ddim2_ref.npz")
    seed = int(g["seed"])
    sd = _seeded_sd("stdit", seed, depth=28, Cc=4096, L=120)
    cfg = stdit_geo(sd, 28)
    x, y, mask, _ = stdit_full_inputs(seed)
    ynull = stdit_full_null_y(seed)
    tmap, acp = sr.spaced_schedule(2)
    assert tmap == [int(v) for v in g["timestep_map"]]
    spec = sr.QSpec(w_bits=8)
    for i in (1, 0):
        t = torch.tensor([tmap[i]])
        cond = sr.stdit_forward(sd, cfg, x, t, y, mask, spec)
        unc = sr.stdit_forward(sd, cfg, x, t, ynull, mask, spec)
        x = sr.cfg_ddim_step(x, cond, unc, acp, i, 4.0)
    _row(res, hip, "stdit_full/ddim2_final", x[:, :, :, ::2, ::2], g["final"], g["final_ref_fp16"])


def run_stdit_full_static(res, hip):
    """the static tensor-wise plan (w8a8_naive: the reference's 364 calibrated activation grids, cfg_split False): one joint
    B = 2 forward (make_golden.py::stdit_full_static)"""
    from helpers import stdit_full_null_y
    g = _npz("stdit_full_static_ref.npz")
    seed = int(g["seed"])
    sd = _seeded_sd("stdit", seed, depth=28, Cc=4096, L=120)
    cfg = stdit_geo(sd, 28)
    x, y, mask, _ = stdit_full_inputs(seed)
    names = sorted({k.split("/")[1] for k in g if k.startswith("act/")})
    assert len(names) == 13 * 28, len(names)
    ag = {n: (g["act/%s/delta" % n].float().reshape(1, 1, 1), g["act/%s/zero_point" % n].float().reshape(1, 1, 1)) for n in names}
    spec = sr.QSpec(w_bits=8, act_mode="static", a_per_group=False, a_grid=ag, n_prompt=120)
    out, blocks = sr.stdit_forward(sd, cfg, torch.cat([x, x]), torch.tensor([721, 721]), torch.cat([y, stdit_full_null_y(seed)]),
                                   mask, spec, return_blocks=True)
    _row(res, hip, "stdit_full_static/joint_t721_block27", blocks[27].reshape(2, 16384, 1152)[:, ::512], g["joint_t721_block27"],
         g["joint_t721_block27_ref_fp16"])
    _row(res, hip, "stdit_full_static/joint_t721_out", out[:, :, ::2, ::2, ::2], g["joint_t721_out"], g["joint_t721_out_ref_fp16"])


def run_alpha256_full(res, hip):
    """PixArt-alpha 256 x 256 W8A8, DPM-Solver++ 2M, 20 steps, cfg 4.5 (make_golden.py::alpha256_full): the whole trajectory
    under this repository's solver with the oracle forward - first model output, latent after 1 / 5 / 10 / 20 steps"""
    from helpers import alpha256_inputs
    from viditq_amd.t2i.dpm_solver import DPMS_alpha
    g = _npz("alpha256_full_ref.npz")
    seed = int(g["seed"])
    sd = _seeded_sd("pixart", seed, depth=28, Cc=4096, L=120)
    pe = g["pos_embed"].float()
    z, y, null_y, mask = alpha256_inputs(seed)
    cfg = dict(H=16, depth=28, patch=2, out_ch=8)
    spec = sr.QSpec(w_bits=8, fp_layers=pr.T2I_FP_LAYERS)
    seen, first = [], {}

    def model(x, t, y_, mask=None, **kw):
        seen.append(x[:1].clone())
        m_ = mask if mask.shape[0] == y_.shape[0] else mask.repeat(y_.shape[0] // mask.shape[0], 1)
        out, blocks = pr.pixart_forward(sd, cfg, x, t, y_, m_, spec, pe, return_blocks=True)
        if len(seen) == 1:
            first["block0"], first["eps"] = blocks[0][:, ::8], out.chunk(2, dim=1)[0]
        return out.chunk(2, dim=1)[0]
    solver = DPMS_alpha(model, condition=y, uncondition=null_y, cfg_scale=4.5, model_kwargs=dict(data_info=None, mask=mask))
    final = solver.sample(z, steps=20, order=2, skip_type="time_uniform", method="multistep")
    assert len(seen) == 20
    _row(res, hip, "alpha256_full/call0_block0", first["block0"], g["call0_block0"], g["call0_block0_ref_fp16"])
    _row(res, hip, "alpha256_full/call0_eps", first["eps"], g["call0_eps"], g["call0_eps_ref_fp16"])
    for k in (1, 5, 10):
        _row(res, hip, "alpha256_full/x%d" % k, seen[k], g["x%d" % k], g["x%d_ref_fp16" % k])
    _row(res, hip, "alpha256_full/final", final, g["final"], g["final_ref_fp16"])


def main():
    which = [a for a in sys.argv[1:] if not a.startswith("-")] or ["stdit_full", "stdit_full_w4a8", "sigma1024_full"]
    hip, src = _hip_figures()
    res = {}
    prev = OUT if os.path.exists(OUT) else os.path.join(ROOT, "profiles", "r06_parity_floor.json")   # (records carry over between rounds)
    if os.path.exists(prev):
        with open(prev) as f:
            res = json.load(f).get("records", {})
    torch.manual_seed(0)
    with torch.no_grad():
        for w in which:
            t0 = time.time()
            {"stdit_full": run_stdit_full, "stdit_full_w4a8": run_stdit_full_w4a8, "sigma1024_full": run_sigma1024_full,
             "stdit_full_ddim2": run_stdit_full_ddim2, "stdit_full_static": run_stdit_full_static,
             "alpha256_full": run_alpha256_full}[w](res, hip)
            _say("[%s] %.0f s on %d torch threads" % (w, time.time() - t0, torch.get_num_threads()))
            with open(OUT, "w") as f:
                json.dump({"what": "fp32 oracle vs the reference's fp32 mode at the full-size golden checkpoints (the parity "
                                   "floor), beside the reference's own fp16-mode drift and the HIP path's last recorded figure",
                           "hip_figures_from": src, "torch": torch.__version__,
                           "cross_attention": "xformers restated in the fixtures (oracle/ref_import.py:127-143)",
                           "records": res}, f, indent=1, sort_keys=True)
    # the checksummed log: what ran, on which golden files, and the digest of the records it left
    import hashlib
    sha = lambda b: hashlib.sha256(b).hexdigest()      # noqa: E731
    with open(LOG, "a") as f:
        f.write("== python tools/parity_floor.py %s  (torch %s, %d threads)\n" % (" ".join(which), torch.__version__, torch.get_num_threads()))
        for ln in _LOG_LINES:
            f.write(ln + "\n")
        for name in sorted(_GOLDEN_USED):
            with open(os.path.join(ROOT, "tests", "golden", name), "rb") as gf:
                f.write("sha256 tests/golden/%s %s\n" % (name, sha(gf.read())))
        f.write("sha256 records(%s) %s\n" % (",".join(sorted(k for k in res if k.split("/")[0] in which or
                                                              any(k.startswith(w + "/") or k.startswith(w.replace("_ddim2", "") + "/ddim2") for w in which))),
                                              sha(json.dumps({k: res[k]["oracle_fp32_vs_ref_fp32"] for k in sorted(res)}, sort_keys=True).encode())))


if __name__ == "__main__":
    main()
