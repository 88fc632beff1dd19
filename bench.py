#!/usr/bin/env python
"""bench.py - denoising steps/s of the quantized STDiT 16x512x512 hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched with
torch.distributed.run, one rank per GPU (RCCL).  Prints ONE JSON line on rank 0.

A "step" = one DDIM iteration for one prompt = 2 STDiT-XL/2 forward-samples (cond + uncond,
cfg_split as in w8a8_dynamic.yaml) + the fused CFG/DDIM update; inputs (latent, text embeds, packed
int8 weights) are resident in HBM when the timed region starts.  Multi-GPU: prompts are sharded
over ranks (weak scaling, one prompt in flight per GPU as the reference's batch_size=1), packed
weights are broadcast once from rank 0, no collective inside a step.

Besides the headline (`value`: BASELINE.json's metric, W8A8) the line carries, at N = 1, `extras`: short legs of the
other single-GPU configurations of BASELINE.json on the same build - W4A8 timestep-aware, W4A8 mixed precision
(20-step schedule) and PixArt-Sigma 1024^2 W4A8 - each with its own steps/s and GEMM fraction (`--no-extras` skips them).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_INT8 = 5.03e15      # dense int8 MFMA ops/s, 256 CU x 2.4 GHz x 8192 op/clk/CU (MI355X_MICROARCH.md / datasheet)
PEAK_HBM = 8.0e12
MP_SAMPLING = 20          # the mixed-precision plan's schedule: ONE constant for the widths that travel in the broadcast
                          # (stdit_legs) and the config applied per step range (_measure_plan) - they must name the same layers
EXTRA_STEPS, EXTRA_WARMUP = 10, 3     # timed steps / warm-ups of the `extras` legs (4 / 2 until round 4: too short to be stable)
GEMM_KERNEL = ("gemm_i8_wide_kernel<256,288,4,2,EPI,stagger,W4,0,INT=1> (W8A8 Linear: int8 MFMA 16x16x64, full-line LDS-DMA double "
               "buffer, interior form: scalar-addressed stage pieces issued by one wave per SIMD, fused dequant epilogue)")



# --------------------------------------------------------------------------- telemetry (round 5)
# The driver's number is taken on a box of its own; kernels that did not change have come out 6-14 % apart between boxes.
# Every leg therefore records what the part was doing: board power, shader clock and temperature before / during / after
# (amdgpu sysfs hwmon where the container exposes it, `rocm-smi --json` otherwise), and the clock the GEMM itself ran at
# (one stamped launch behind a back-to-back burst: shader cycle counter against the chip's 100 MHz wall clock).
def _pci_bus_id(index):
    """'0000:05:00.0' of HIP device `index` (torch's device properties; None when the build does not expose them)"""
    try:
        p = torch.cuda.get_device_properties(index)
        return "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:  # noqa: BLE001
        return None


def _sysfs_gpu(index):
    """hwmon files of the amdgpu card that IS HIP device `index`: matched by PCI bus id (a GPU box shows every card of the
    node in /sys/class/drm, the process owns one of them); else the only card whose render node exists in /dev/dri; else None
    (rocm-smi is used instead) - never a guess."""
    import glob
    cards = []
    for dev_dir in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        try:
            if open(os.path.join(dev_dir, "vendor")).read().strip() != "0x1002":
                continue
        except OSError:
            continue
        hw = sorted(glob.glob(os.path.join(dev_dir, "hwmon", "hwmon*")))
        if hw:
            cards.append((os.path.basename(os.path.realpath(dev_dir)), dev_dir, hw[0]))
    bus = _pci_bus_id(index)
    pick = [c for c in cards if bus and c[0].lower() == bus.lower()]
    if not pick:
        mine = [c for c in cards if any(os.path.exists(os.path.join("/dev/dri", os.path.basename(r)))
                                        for r in glob.glob(os.path.join(c[1], "drm", "renderD*")))]
        pick = mine if len(mine) == 1 else []
    if not pick:
        return None
    _, dev_dir, hw = pick[0]
    f = {}
    for key, names in (("power_w", ("power1_average", "power1_input")), ("sclk_mhz", ("freq1_input",)),
                       ("mclk_mhz", ("freq2_input",)), ("temp_c", ("temp2_input", "temp1_input"))):
        for n in names:
            if os.path.exists(os.path.join(hw, n)):
                f[key] = os.path.join(hw, n)
                break
    # the DPM tables of the fabric / SoC / memory clocks (current level marked '*'): HBM-bound kernels have come out 15-30 %
    # apart between boxes at EQUAL shader clock, power and temperature - if a box runs its fabric slower it shows here
    for key, name in (("fclk_mhz", "pp_dpm_fclk"), ("socclk_mhz", "pp_dpm_socclk"), ("mclk_dpm_mhz", "pp_dpm_mclk")):
        if os.path.exists(os.path.join(dev_dir, name)):
            f["_dpm_" + key] = os.path.join(dev_dir, name)
    if f:
        f["_card"] = os.path.basename(os.path.dirname(dev_dir)) + " @ " + pick[0][0]
    return f or None


def _read_sysfs(files):
    out = {}
    for k, path in files.items():
        if k.startswith("_dpm_"):
            try:
                import re
                cur = [ln for ln in open(path).read().splitlines() if "*" in ln]
                m = re.search(r"(\d+)\s*mhz", cur[0].lower()) if cur else None
                if m:
                    out[k[5:]] = float(m.group(1))
            except OSError:
                pass
            continue
        if k.startswith("_"):
            continue
        try:
            v = float(open(path).read().strip())
        except (OSError, ValueError):
            continue
        out[k] = v / 1e6 if k in ("power_w", "sclk_mhz", "mclk_mhz") else v / 1e3
    return out


def _rocm_smi(index):
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return {"error": "no rocm-smi"}
    try:
        r = subprocess.run([exe, "-d", str(index), "--showpower", "--showclocks", "--showtemp", "--json"],
                           capture_output=True, text=True, timeout=20)
        card = next(iter(json.loads(r.stdout).values()))
    except Exception as e:  # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, e)}
    out = {}
    import re
    for k, v in card.items():
        kl = k.lower()
        m = re.search(r"-?\d+(\.\d+)?", str(v))
        if not m:
            continue
        x = float(m.group(0))
        if "power" in kl and "power_w" not in out:
            out["power_w"] = x
        elif kl.startswith("sclk") and "sclk_mhz" not in out:
            out["sclk_mhz"] = x
        elif kl.startswith("mclk") and "mclk_mhz" not in out:
            out["mclk_mhz"] = x
        elif "temperature" in kl and ("junction" in kl or "temp_c" not in out):
            out["temp_c"] = x
    return out or {"error": "unparsed", "keys": sorted(card)[:12]}


class Telemetry:
    """snapshot(): one full reading (power, clocks, temperature, the fabric / SoC / memory DPM tables); start() / stop(): a
    sampler thread around a timed region that reads ONLY board power and shader clock (two hwmon files) five times a second -
    round-5 advisor: the full set at 20 Hz cost ~250 ms per sample on the driver's box (SMU metric queries behind the pp_dpm
    tables), i.e. the headline was timed with SMU traffic and a Python thread beside it.  Temperature and the DPM tables come
    from the snapshots before and after the region.  (sysfs only: a rocm-smi process per sample would perturb what it
    measures.)  A/B of this sampler against --no-telemetry on one box: profiles/r06_experiments.md."""
    LIGHT = ("power_w", "sclk_mhz")

    def __init__(self, index):
        self.index = index
        self.files = _sysfs_gpu(index)
        self.source = ("sysfs hwmon of %s" % self.files["_card"]) if self.files else "rocm-smi"
        self._stop = None
        self._rows = []

    def snapshot(self):
        return _read_sysfs(self.files) if self.files else _rocm_smi(self.index)

    def start(self, resume=False):
        """resume: keep the rows of earlier start / stop pairs (several prompts: one sampled region per prompt's TIMED loop,
        the untimed graph captures between them are not sampled)"""
        import threading
        if not self.files:
            # no sysfs sensor for this device: ONE rocm-smi reading taken while the timed region runs (the process start
            # of rocm-smi, ~0.3-0.6 s, puts the reading inside a region of >= 10 steps; the loop is GPU-bound)
            self._one = {}
            self._th = threading.Thread(target=lambda: self._one.update(_rocm_smi(self.index)), daemon=True)
            self._th.start()
            return
        if not resume:
            self._rows = []
        self._stop = threading.Event()

        light = {k: v for k, v in self.files.items() if k in self.LIGHT} or self.files

        def run():
            while not self._stop.is_set():
                self._rows.append(_read_sysfs(light))
                self._stop.wait(0.2)
        self._th = threading.Thread(target=run, daemon=True)
        self._th.start()

    def stop(self):
        if not self.files:
            if getattr(self, "_th", None) is None:
                return None
            self._th.join(timeout=30)
            self._th = None
            return dict(self._one, samples=1, how="one rocm-smi reading started with the timed region")
        if self._stop is None:
            return None
        self._stop.set()
        self._th.join()
        self._stop = None
        rows = [r for r in self._rows if r]
        if not rows:
            return None
        out = {"samples": len(rows)}
        for k in rows[0]:
            v = [r[k] for r in rows if k in r]
            out[k] = {"min": min(v), "mean": sum(v) / len(v), "max": max(v)}
        return out


_CLOCK_PROBE = {}
TEL = None          # Telemetry of this rank's device, set in main()


def gemm_clock_probe(dev, burst=24, reps=7):
    """The shader clock UNDER the GEMM on this box, now: `burst` back-to-back launches of the qkv shape (16384 x 3456 x
    1152, ~75 us each), then one stamped launch of the same problem (ops.gemm_i8_stamped)."""
    from viditq_amd import ops
    if "qa" not in _CLOCK_PROBE:
        g = torch.Generator().manual_seed(0)
        x = torch.randn(1, 16384, 1152, generator=g).half().to(dev)
        W = (torch.randn(3456, 1152, generator=g) * 0.03).half().to(dev)
        d, z = ops.weight_minmax(W, 8)
        _CLOCK_PROBE.update(qa=ops.rowquant(x), pw=ops.pack_weight(W, d, z, 8),
                            out=torch.empty((16384, 3456), dtype=torch.float16, device=dev))
    qa, pw, out = _CLOCK_PROBE["qa"], _CLOCK_PROBE["pw"], _CLOCK_PROBE["out"]
    try:
        ts = []
        for _ in range(reps):
            for _ in range(burst):
                ops.gemm_i8(qa, pw, out=out, variant=11)
            ts.append(ops.gemm_i8_stamped(qa, pw)[1])
        torch.cuda.synchronize()
        t = sorted((ops.shader_clock_ghz(st) for st in ts), key=lambda d: d["ghz"])
        med = t[len(t) // 2]
        return {"ghz": round(med["ghz"], 3), "ghz_min": round(t[0]["ghz"], 3), "ghz_max": round(t[-1]["ghz"], 3),
                "tile_cycles": round(med["tile_cycles"]), "launch_span_us": round(med["launch_span_us"], 1),
                "phase_cycles": {k: round(v) for k, v in med["phase_cycles"].items()},
                "how": "%d x (%d back-to-back qkv-shape launches, then one stamped launch of the same problem): shader cycle counter "
                       "/ 100 MHz wall clock, median over the 6144 waves of a launch; median / min / max over the %d stamped launches"
                       % (reps, burst, reps)}
    except Exception as e:  # noqa: BLE001  (telemetry must never lose the line)
        return {"error": "%s: %s" % (type(e).__name__, e)}


def memory_probe(dev):
    """What the box's memory system delivers right now, beside the rates of the HBM-bound kernels: a 1 GiB device copy
    (HBM: 2 GiB of traffic per copy) and a 48 MiB one (resident in the 256 MiB Infinity Cache after the first pass), best of
    the timed repetitions.  Boxes of one pool have differed by 15-30 % here at equal shader clock and power."""
    try:
        out = {}
        for name, nbytes, reps in (("hbm_copy_1GiB_TBps", 1 << 30, 6), ("mall_copy_48MiB_TBps", 48 << 20, 40)):
            a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            b = torch.empty_like(a)
            b.copy_(a)
            best = None
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    b.copy_(a)
                e1.record()
                torch.cuda.synchronize()
                t = e0.elapsed_time(e1) * 1e-3 / reps
                best = t if best is None or t < best else best
            out[name] = round(2.0 * nbytes / best / 1e12, 3)
            del a, b
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, e)}


def leg_telemetry(tel, before, during, dev):
    """what a leg records: readings before / during / after its timed region, the GEMM's own clock and the memory system's
    copy rates right after it"""
    clock = gemm_clock_probe(dev)
    return {"source": tel.source, "before": before, "during_timed_region": during, "after": tel.snapshot(),
            "gemm_shader_clock": clock, "memory": memory_probe(dev)}



def collective_info(dist, rehearsal):
    if dist is None:
        return {"backend": None, "note": "one rank: no process group"}
    out = {"backend": "gloo (rehearsal)" if rehearsal else "nccl = RCCL", "torch": torch.__version__,
           "hip": getattr(torch.version, "hip", None)}
    try:
        out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:  # noqa: BLE001
        out["rccl_version"] = "unavailable: %s" % type(e).__name__
    for k in ("NCCL_DEBUG", "HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_P2P_DISABLE", "RCCL_MSCCL_ENABLE"):
        if k in os.environ:
            out.setdefault("env", {})[k] = os.environ[k]
    return out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--depth", type=int, default=28, help="STDiT depth (28 = STDiT-XL/2; anything else is a debug run)")
    ap.add_argument("--plan", default="w8a8", choices=["w8a8", "w4a8", "w4a8_mp"],
                    help="w8a8 = BASELINE metric (w8a8_dynamic.yaml); w4a8 = ViDiT-Q W4A8 timestep-aware channel "
                         "balancing, synthetic calibration; w4a8_mp = the same with the per-layer mixed-precision "
                         "config on the 20-step schedule")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline-events", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the W4A8 / W4A8-MP / PixArt-Sigma legs")
    ap.add_argument("--no-telemetry", action="store_true", help="no power / clock / temperature sampling, no stamped GEMM launch")
    ap.add_argument("--gemm-variant", type=int, default=None)
    ap.add_argument("--no-graph", action="store_true", help="eager launches in the timed region (no HIP graph)")
    ap.add_argument("--one-stream", action="store_true", help="cond and uncond serialised on one stream")
    ap.add_argument("--prompts", type=int, default=None,
                    help="BASELINE config 4: P prompts sharded round-robin over the ranks (prompt i -> rank i mod N, 64 over "
                         "8 GPUs = 8 per rank), each prompt with its own captured graph, W warm-up + K timed steps per "
                         "prompt; default: one prompt per GPU")
    return ap.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start the N ranks ourselves,
    through the same `torch.distributed.run` line the contract names, and exit with its status.  A box with fewer devices
    than ranks is an ERROR (never a silent one-rank run that prints n_gpus 1) unless VQ_BENCH_REHEARSAL=1."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < a.gpus and os.environ.get("VQ_BENCH_REHEARSAL") != "1":
        sys.exit("bench.py: --gpus %d asked for, %d device(s) visible - refusing to run (VQ_BENCH_REHEARSAL=1 shares "
                 "devices between ranks over gloo as a control-flow dry run)" % (a.gpus, have))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def cpu_block_timer():
    """-> a function that runs ONE full-size STDiTBlock forward-sample of the oracle (x [1,16384,1152], 80 prompt tokens) and
    returns its wall time (the sample of `cpu_baseline`; tools/oracle_threads.py sweeps thread counts with it)."""
    from oracle import stdit_ref as sr
    torch.manual_seed(0)
    C, T, S, H = 1152, 16, 1024, 16
    g = torch.Generator().manual_seed(0)
    sd = {}
    p = "blocks.0"

    def lin(name, n, k):
        sd["%s.%s.weight" % (p, name)] = (torch.randn(n, k, generator=g) * (2.0 / (n + k)) ** 0.5).half().float()
        sd["%s.%s.bias" % (p, name)] = torch.zeros(n)
    for nm in ("attn.q", "attn.k", "attn.v", "attn.proj", "attn_temp.q", "attn_temp.k", "attn_temp.v", "attn_temp.proj",
               "cross_attn.q_linear", "cross_attn.proj"):
        lin(nm, C, C)
    lin("cross_attn.kv_linear", 2 * C, C)
    lin("mlp.fc1", 4 * C, C)
    lin("mlp.fc2", C, 4 * C)
    sd[p + ".scale_shift_table"] = torch.randn(6, C, generator=g) / C ** 0.5
    x = torch.randn(1, T * S, C, generator=g).half().float()
    y = (torch.randn(1, 80, C, generator=g) * 0.3).half().float()
    t0 = torch.randn(1, 6 * C, generator=g) * 0.1
    tpe = torch.randn(1, T, C, generator=g) * 0.1
    spec = sr.QSpec(w_bits=8)

    def once():
        t_ = time.perf_counter()
        with torch.no_grad():
            sr.stdit_block(sd, 0, x, y, t0, [80], tpe, T, S, H, spec)
        return time.perf_counter() - t_
    return once


def cpu_baseline(depth_total=28):
    """The oracle (CPU restatement of the reference fake-quant path, fp32) timed on this box's host
    cores on a bounded sample: ONE full-size STDiTBlock forward-sample (x [1,16384,1152], 80 prompt
    tokens); steps/s extrapolated as 1 / (t_block * 28 blocks * 2 forward-samples)."""
    once = cpu_block_timer()
    # the baseline gets the thread count that serves IT best: on the GPU box's host (torch default: 128 threads) this
    # memory-bound path is ~3 x faster on 24 threads (tools/oracle_threads.py) - one run per setting, a second at the best
    n0 = torch.get_num_threads()
    sweep = {}
    with torch.no_grad():
        try:
            for n in [n0] + [k for k in (64, 32, 24, 16) if k < n0]:
                torch.set_num_threads(n)
                sweep[n] = once()
            best = min(sweep, key=sweep.get)
            torch.set_num_threads(best)
            tb = min(sweep[best], once())
        finally:
            torch.set_num_threads(n0)
    return {"value": 1.0 / (tb * depth_total * 2), "unit": "denoising steps/s", "cores": best,
            "kind": "port",
            "sample": "1 full-size STDiTBlock forward-sample (16384 tokens x 1152, 80 prompt tokens), fp32 oracle, "
                      "best of 2 = %.2f s on %d threads (one run each on %s threads: %s s); extrapolated x28 blocks x2 "
                      "forward-samples per step" % (tb, best, "/".join(str(k) for k in sweep),
                                                    "/".join("%.2f" % v for v in sweep.values()))}


def gemm_traffic():
    """HBM / fabric bytes per GEMM launch: PMC passes cannot run inside this process (rocprofv3 wraps the process);
    the committed measurement of the W8A8 command on the shipping kernels at depth 28 (tools/measure_round.sh ->
    profiles/r0N_gemm_traffic.json) is reported - and REFUSED (traffic null + a reason) when any GEMM source is newer
    than the measurement, so a kernel change can never ship stale bytes."""
    import glob
    import re
    cands = [(int(m.group(1)), f) for f in glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_traffic.json"))
             for m in [re.match(r"r(\d+)_gemm_traffic\.json$", os.path.basename(f))] if m]
    if not cands:
        return None, "no profiles/rNN_gemm_traffic.json"
    tj = max(cands)[1]                                  # the latest ROUND (numeric: r10 follows r09)
    with open(tj) as f:
        t_ = json.load(f)
    import hashlib
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(ROOT, "vidit-q_amd", "csrc", "gemm_*"))):
        if fn.endswith((".h", ".hip")):
            with open(fn, "rb") as f:
                h.update(f.read())
    if h.hexdigest() != t_.get("gemm_sources_sha256"):
        return None, ("STALE: %s was measured on other GEMM sources (source hash differs or is absent) - re-run "
                      "tools/measure_round.sh" % os.path.basename(tj))
    return t_["hbm_bytes_per_launch"], t_["source"]


def event_pair_overhead_us(n=200):
    """What an EMPTY event pair reads on this box (record, record, nothing between): the part of every bracketed launch that
    is the bracket itself.  Reported beside the raw figure, never subtracted from `achieved` / `frac`."""
    try:
        torch.cuda.synchronize()
        torch.cuda._sleep(int(0.01 * 2.1e9))               # the host runs ahead, as in the bracketed pass
        pairs = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        v = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in pairs)
        return v[len(v) // 2]
    except Exception:  # noqa: BLE001
        return None


def gemm_roofline(timing, el, with_traffic):
    tot_ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in timing)
    tot_ops = sum(o for _, _, o, _ in timing)
    ach = tot_ops / (tot_ms * 1e-3)
    ov = event_pair_overhead_us()
    traffic, traffic_src = gemm_traffic() if with_traffic else (None, "not measured for this plan")
    return {"bound": "mfma", "kernel": GEMM_KERNEL,
            "achieved": ach / 1e12, "peak": PEAK_INT8 / 1e12, "unit": "TFLOP/s", "frac": ach / PEAK_INT8,
            "traffic": traffic, "traffic_source": traffic_src, "launches": len(timing), "avg_launch_us": tot_ms * 1e3 / len(timing),
            # the bracket's own cost, measured live (an empty event pair), and what the kernel time would be without it - for
            # comparison with the rocprofv3 average of profiles/; `achieved` / `frac` above stay the RAW bracketed figures
            "event_pair_overhead_us": ov,
            "avg_launch_us_net_of_event_pair": (tot_ms * 1e3 / len(timing) - ov) if ov is not None else None,
            "frac_net_of_event_pair": (tot_ops / ((tot_ms * 1e-3) - len(timing) * ov * 1e-6) / PEAK_INT8) if ov is not None else None,
            "gemm_time_share_of_step": (tot_ms * 1e-3 / el) if el else None,
            "measured": "HIP events around every GEMM launch, eager re-run of the same K steps after the timed region",
            "algorithmic_bytes_per_launch_avg": sum(b for _, _, _, b in timing) / len(timing)}


def stdit_legs(a, dev, rank, world, plans, steps, warmup, dist=None, events=True, hoisted=True):
    """STDiT-XL/2 16x512x512 measurements of ``plans`` on ONE model build (plans of one weight format share it: the
    mixed-precision plan is the W4A8 model with per-layer bit widths switched per step range): build + quantize
    (+ broadcast) once, then per plan: capture, W warm-up steps, K timed steps, and (events) the same K steps eagerly
    with an event pair around every GEMM launch."""
    from viditq_amd import shard, synth
    from viditq_amd.config import loads_yaml
    cfg = loads_yaml(synth.W8A8_DYNAMIC if plans[0] == "w8a8" else synth.W4A8_TIMESTEP_AWARE)
    out = []
    with torch.no_grad():
        model = synth.build_stdit(dev, depth=a.depth)
        # a plan that switches bit widths per step range names them up front: they travel in the one broadcast, and
        # ranks > 0 (which drop their fp16 master weights) never have to re-pack
        mp_w = synth.synthetic_mp_config(model, MP_SAMPLING)[0] if "w4a8_mp" in plans else None
        qnn = shard.quantize_and_distribute(model, cfg, rank, world, mp_weight_cfg=mp_w)   # rank 0 calibrates + packs, RCCL broadcast
        assert all(b.fused_ok() for b in qnn.model.blocks), "hot path must be the fused HIP route"
        for plan in plans:
            res = _measure_plan(a, dev, rank, world, qnn, cfg, plan, steps, warmup, dist, events, hoisted)
            res["broadcast"] = getattr(qnn, "_broadcast_stats", None)
            out.append(res)
        del qnn, model
    torch.cuda.empty_cache()
    return out


def _measure_plan(a, dev, rank, world, qnn, cfg, plan, steps, warmup, dist, events, hoisted):
    from viditq_amd import graph, ops, synth
    from viditq_amd.t2v import IDDPM
    res = {}
    n_sampling = MP_SAMPLING if plan == "w4a8_mp" else 100
    sch = IDDPM(num_sampling_steps=n_sampling, cfg_scale=4.0)
    mp = None
    if plan == "w4a8_mp":
        from viditq_amd import ptq
        from viditq_amd.t2v.iddpm import TimestepMP
        ptq.enable_timestep_wise_mp(qnn, *synth.synthetic_mp_config(qnn, n_sampling))
        mp = TimestepMP(qnn)
    # prompt i -> rank i mod R; default one prompt per GPU (prompt index = rank), --prompts P: P prompts round-robin
    from viditq_amd import shard
    n_prompts = a.prompts if a.prompts else world
    mine = shard.prompts_of_rank(n_prompts, rank, world)
    assert mine, "rank %d has no prompt (%d prompts over %d ranks)" % (rank, n_prompts, world)
    embeds, lens = synth.synthetic_prompts(n_prompts, dev)
    idx = list(range(sch.num_timesteps))[::-1]
    gs = None

    def inputs(pi):
        x = synth.synthetic_latent(pi, device=dev).float()
        y = embeds["y"][pi:pi + 1]                                   # [1, 2, 1, 120, 4096]
        y = y.permute(1, 0, 2, 3, 4).reshape(2, 1, 120, 4096)
        return x, torch.empty_like(x), y[:1], y[1:], embeds["mask"][pi:pi + 1]

    def step(j, x, buf, eager=False):
        i = idx[j % len(idx)]
        t_id = sch.timestep_map[i]
        key = mp.apply(i) if mp is not None else None   # per-layer bit widths of this step's range
        if gs is not None and not eager:            # both forward-samples replayed from one HIP graph
            cond, unc = gs.forward_pair(x, t_id, key)
        else:
            t = torch.full((1,), t_id, device=dev, dtype=torch.long)
            cond = qnn(x, t, y_c, mask=mask, timestep_id=t_id)
            unc = qnn(x, t, y_u, mask=mask, timestep_id=t_id)
        out = sch.ddim_step(x, cond, unc, i, sch.cfg_scale, 0.0, out=buf)
        return out, x

    el = 0.0
    host_enqueue = 0.0
    tel_before = None
    for n_done, pi in enumerate(mine):
        # a prompt's token selection is baked into its captured graphs: each prompt captures its own (untimed, as the
        # reference's per-prompt set-up is), then runs W warm-up and K timed steps
        x, buf, y_c, y_u, mask = inputs(pi)
        gs = None if a.no_graph else graph.GraphedSampler(qnn, y_c, y_u, mask, two_streams=not a.one_stream)
        if mp is not None:                              # pack + capture every mixed-precision key up front
            for i in idx[::max(1, len(idx) // 4)]:
                key = mp.apply(i)
                if gs is not None:
                    gs.forward_pair(x, sch.timestep_map[i], key)
        elif gs is not None and synth.uses_smooth_quant(cfg):   # one graph per smooth-quant time-range
            for t_probe in (999, 0):
                gs.forward_pair(x, t_probe, None)
        for j in range(warmup):
            x, buf = step(j, x, buf)
        torch.cuda.synchronize()
        if dist is not None and n_done == 0:
            dist.barrier()
        torch.cuda.synchronize()
        if TEL is not None:
            if n_done == 0:
                tel_before = TEL.snapshot()
            TEL.start(resume=n_done > 0)
        t0 = time.perf_counter()
        for j in range(warmup, warmup + steps):
            x, buf = step(j, x, buf)
        t_enq = time.perf_counter() - t0                # host done enqueueing (graph replays: a few ms; eager: the launch rate)
        torch.cuda.synchronize()
        if dist is not None and n_done == len(mine) - 1:
            dist.barrier()
            torch.cuda.synchronize()
        el += time.perf_counter() - t0
        host_enqueue += t_enq
        if gs is not None and n_done == len(mine) - 1:
            # What a replay costs the HOST, measured with an EMPTY queue (a synchronize before every launch).  The figure above
            # (host_enqueue_ms_per_step) is the time until the loop's last replay call returned: with more than ~7 steps in
            # flight hipGraphLaunch blocks on the full hardware queue, so over 20 steps it reads ~GPU time per step
            # (25-27 ms of a 38 ms step) while over 6 steps it reads 3.5 ms - back-pressure, not launch cost.
            idle = []
            for j in range(warmup + steps, warmup + steps + 4):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                x, buf = step(j, x, buf)
                idle.append(time.perf_counter() - t1)
            torch.cuda.synchronize()
            res["host_graph_launch_ms_idle_queue"] = sorted(idle)[len(idle) // 2] * 1e3
            res["graph_nodes"] = graph_census(gs)
        if TEL is not None:
            during = TEL.stop()                         # aggregated over the timed loops of every prompt so far
            if n_done == len(mine) - 1:
                res["telemetry"] = leg_telemetry(TEL, tel_before, during, dev)
        if n_done < len(mine) - 1:
            gs = None                                   # frees this prompt's graphs before the next capture
    assert torch.isfinite(x).all()
    # live roofline: the SAME K steps once more, launched eagerly with a HIP-event pair around
    # every GEMM launch on the launch stream (events cannot be recorded inside a captured graph)
    timing = [] if events else None
    if timing is not None:
        ops.GEMM_TIMING = timing
        for j in range(warmup, warmup + steps):
            # park the GPU for ~60 ms first so the host runs AHEAD of it: every event pair then brackets
            # kernel execution only, not the idle gap of an eager, host-bound launch
            torch.cuda._sleep(int(0.06 * 2.1e9))
            x, buf = step(j, x, buf, eager=True)
        torch.cuda.synchronize()
        ops.GEMM_TIMING = None
    res["status"] = qnn.check_status()
    # extra (NOT the reported value): the same K steps with the step-invariant prompt work hoisted out of the loop
    # (y_embedder + every block's cross-attention K/V computed once per prompt; bit-identical outputs)
    cached = None
    if hoisted and gs is not None and hasattr(qnn.model, "set_prompt_cache"):
        qnn.model.set_prompt_cache(True)
        gs2 = graph.GraphedSampler(qnn, y_c, y_u, mask, two_streams=not a.one_stream)
        gs_keep, gs = gs, gs2
        for j in range(warmup):
            x, buf = step(j, x, buf)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for j in range(warmup, warmup + steps):
            x, buf = step(j, x, buf)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t1
        cached = {"value_this_rank": steps / el2, "ms_per_step": el2 / steps * 1e3,
                  "note": "prompt K/V + embedding computed once per prompt instead of once per forward; exact; not the headline"}
        gs = gs_keep
        qnn.model.set_prompt_cache(False)
    gs = None
    res.update(el=el, steps=steps * len(mine), n_sampling=n_sampling, cached=cached, n_prompts=n_prompts, prompts_here=len(mine),
           host_enqueue_ms_per_step=host_enqueue / (steps * len(mine)) * 1e3,
           roofline=gemm_roofline(timing, el / len(mine), plan == "w8a8") if timing else None)
    return res


def graph_census(gs):
    """Node counts of the captured step graphs by kind (VQ_GRAPH_DUMP=<prefix>: graph.StepGraph writes the runtime's .dot
    dump of each capture); None without the dump."""
    out = None
    for g in getattr(gs, "graphs", {}).values():
        if out is None and getattr(g, "c_abi_calls", None):
            out = {"c_abi_calls_per_step": g.c_abi_calls,
                   "note": "entry points of libviditq_hip.so recorded into one step graph (cond + uncond), one kernel node each; "
                           "torch's own elementwise kernels of the FP edges (~60 per step) come on top"}
        path = getattr(g, "dot_path", None)
        if not path or not os.path.exists(path):
            continue
        txt = open(path, errors="replace").read()
        nodes = [ln for ln in txt.splitlines() if "label=" in ln and "->" not in ln]
        c = {"nodes": len(nodes), "edges": txt.count("->")}
        for kind in ("KERNEL", "MEMCPY", "MEMSET", "EMPTY", "EVENT", "HOST"):
            c[kind.lower()] = sum(1 for ln in nodes if kind in ln.upper())
        if out is None or "nodes" not in out:
            out = dict(out or {}, **c)
        try:
            os.remove(path)
        except OSError:
            pass
    return out


def pixart_leg(dev, steps=10, w_bits=4, size=1024, Lp=300, warmup=3):
    """BASELINE config 5 as a timing leg: PixArt-Sigma 1024^2 (4096 tokens, prompts of up to 300 tokens), 4-bit weights,
    dynamic per-token 8-bit activations, DPM-Solver++ 2M, cfg 4.5, the t2i loop's ONE batched (uncond | cond) forward
    per step (quant_txt2img.py:130-153; B = 2: token scales shared over the pair as base_quantizer.py:185 does)."""
    from viditq_amd import ops, synth
    from viditq_amd.config import to_config
    from viditq_amd.qdiff.models import QuantModel
    from viditq_amd.t2i import DPMS_sigma, PixArtMS_XL_2
    lat = size // 8
    with torch.no_grad():
        torch.manual_seed(0)
        m = PixArtMS_XL_2(input_size=lat, model_max_length=Lp, pe_interpolation=lat / 64, dtype=torch.float16)
        synth.redraw_zero_init(m, 1)
        m = m.half().to(dev).eval()
        wq = to_config(dict(n_bits=w_bits, per_group="channel", channel_dim=0, scale_method="min_max", round_mode="nearest",
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
        # eager launches: a forward is 19.9 ms of GPU work and Python (10.9 ms of launches per step) stays ahead of it;
        # replayed as a HIP graph (graph.GraphedModel) the step takes the same time (49.7 vs 50.3 steps/s)
        solver = DPMS_sigma(qnn.forward_with_dpmsolver, condition=y, uncondition=null_y, cfg_scale=4.5,
                            model_kwargs=dict(data_info=None, mask=mask))
        solver.sample(z, steps=warmup, order=2)                # warm-up: packing, caches
        torch.cuda.synchronize()
        tel_before = TEL.snapshot() if TEL is not None else None
        if TEL is not None:
            TEL.start()
        t0 = time.perf_counter()
        out = solver.sample(z, steps=steps, order=2)
        t_enq = time.perf_counter() - t0                       # eager launches: the host's enqueue time (GPU-bound iff < el)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        telemetry = leg_telemetry(TEL, tel_before, TEL.stop(), dev) if TEL is not None else None
        assert torch.isfinite(out).all()
        timing = []
        ops.GEMM_TIMING = timing
        torch.cuda._sleep(int(0.06 * 2.1e9))
        solver.sample(z, steps=2, order=2)
        torch.cuda.synchronize()
        ops.GEMM_TIMING = None
        status = qnn.check_status()
        del solver, qnn, m
    torch.cuda.empty_cache()
    roof = gemm_roofline(timing, None, False)
    return {"workload": "PixArt-Sigma %dx%d W%dA8: %d tokens, Lp %d (180 live), DPM-Solver++ 2M, cfg 4.5, batched uncond|cond "
                        "forward, depth 28, eager launches; %d-bit weights as BASELINE names the config (the released yaml "
                        "says n_bits 6), dynamic per-token A8, smooth-quant OFF everywhere - the released script keeps "
                        "blocks.27.mlp.fc2 smoothed with a running statistic (quant_txt2img.py:297-300): that layer is "
                        "parity-tested (tiny_pixart_w4a8) but not part of this timing"
                        % (size, size, w_bits, (lat // 2) ** 2, Lp, w_bits),
            "value": steps / el, "unit": "sampling steps/s", "steps": steps, "warmup": warmup, "ms_per_step": el / steps * 1e3,
            "host_enqueue_ms_per_step": t_enq / steps * 1e3, "telemetry": telemetry,
            "status_word": status, "gemm_frac_of_int8_peak": roof["frac"], "gemm_avg_launch_us": roof["avg_launch_us"]}


def launch_only(a, rank, world):
    """VQ_BENCH_LAUNCH_ONLY=1: the launch / rendezvous / gather / print skeleton of main() with no device work (gloo) -
    what the CPU test of the plain-`python bench.py --gpus N` form runs."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    el_t = torch.tensor([1.0 + rank], dtype=torch.float64)
    per_rank = [float(el_t)]
    if world > 1:
        every = [torch.zeros_like(el_t) for _ in range(world)]
        dist.all_gather(every, el_t)
        per_rank = [float(t.item()) for t in every]
    if rank == 0:
        line = {"launch_only": True, "n_gpus": world, "per_rank_steps_per_s": [a.steps / t for t in per_rank]}
        check_line(line, a)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def check_line(line, a):
    """The line must describe the run that was asked for: one rate per rank, as many ranks as --gpus, and - for N > 1 -
    a weight broadcast that moved bytes."""
    assert line["n_gpus"] == a.gpus == len(line["per_rank_steps_per_s"]), \
        "n_gpus %r, --gpus %r, %d per-rank rates" % (line["n_gpus"], a.gpus, len(line["per_rank_steps_per_s"]))
    if a.gpus > 1 and not line.get("launch_only"):
        wb = line.get("weights_broadcast")
        assert wb and wb["bytes"] > 0, "N > 1 run without a weight broadcast: %r" % (wb,)


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)                                   # never returns
    if world != a.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d (or plain "
                 "`python bench.py --gpus %d`, which starts its own ranks)" % (a.gpus, world, a.gpus, a.gpus))
    # VQ_BENCH_REHEARSAL=1: the N > 1 control flow on a box with fewer devices than ranks (ranks share devices, gloo
    # instead of RCCL, which refuses two ranks on one device) - a dry run of the launch line, never a measurement
    rehearsal = os.environ.get("VQ_BENCH_REHEARSAL") == "1"
    if os.environ.get("VQ_BENCH_LAUNCH_ONLY") == "1":
        return launch_only(a, rank, world)
    if torch.cuda.device_count() < (world if not rehearsal else 1):
        sys.exit("bench.py: %d rank(s) on this node, %d device(s) visible" % (world, torch.cuda.device_count()))
    if rehearsal:
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import viditq_amd  # noqa: F401
    from viditq_amd import ops

    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if a.gemm_variant is not None:
        ops.DEFAULT_GEMM_VARIANT = a.gemm_variant
    global TEL
    if rank == 0 and not a.no_telemetry:
        TEL = Telemetry(local)

    head = stdit_legs(a, dev, rank, world, [a.plan], a.steps, a.warmup, dist=dist, events=not a.no_roofline_events)[0]
    el = head["el"]
    el_t = torch.tensor([el], device=dev, dtype=torch.float64)
    per_rank = [el]
    if dist is not None:
        every = [torch.zeros_like(el_t) for _ in range(world)]
        dist.all_gather(every, el_t)
        per_rank = [float(t.item()) for t in every]
        dist.all_reduce(el_t, op=dist.ReduceOp.MAX)
    el_max = float(el_t.item())
    # N > 1 self-diagnosis: every rank's own time inside the weight broadcast (the receivers block until rank 0 has
    # calibrated and packed, so a slow link and a slow rank 0 look different here) and the collective library's version
    bc_rank = None
    if dist is not None:
        bc = head.get("broadcast") or {}
        bc_t = torch.tensor([float(bc.get("seconds", -1.0))], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(bc_t) for _ in range(world)]
        dist.all_gather(every, bc_t)
        bc_rank = [float(t.item()) for t in every]

    extras = None
    if rank == 0 and world == 1 and a.plan == "w8a8" and not a.no_extras and a.depth == 28 and not a.prompts:
        extras = {}
        # a leg that fails reports its error; the headline line above is already measured and is printed regardless
        try:
            legs = stdit_legs(a, dev, 0, 1, ["w4a8", "w4a8_mp"], EXTRA_STEPS, EXTRA_WARMUP, events=not a.no_roofline_events,
                              hoisted=False)
            for plan, r in zip(("w4a8", "w4a8_mp"), legs):
                extras[plan] = {"value": r["steps"] / r["el"], "unit": "denoising steps/s", "steps": r["steps"], "warmup": EXTRA_WARMUP,
                                "host_enqueue_ms_per_step": r["host_enqueue_ms_per_step"], "telemetry": r.get("telemetry"),
                                "host_graph_launch_ms_idle_queue": r.get("host_graph_launch_ms_idle_queue"),
                                "graph_nodes": (r.get("graph_nodes") or {}).get("c_abi_calls_per_step"),
                                "ms_per_step": r["el"] / r["steps"] * 1e3, "schedule": "DDIM-%d" % r["n_sampling"],
                                "status_word": r["status"],
                                "gemm_frac_of_int8_peak": r["roofline"]["frac"] if r["roofline"] else None,
                                "gemm_avg_launch_us": r["roofline"]["avg_launch_us"] if r["roofline"] else None,
                                "gemm_time_share_of_step": r["roofline"]["gemm_time_share_of_step"] if r["roofline"] else None}
        except Exception as e:  # noqa: BLE001
            extras["w4a8"] = extras["w4a8_mp"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            extras["pixart_sigma_1024_w4a8"] = pixart_leg(dev)
        except Exception as e:  # noqa: BLE001
            extras["pixart_sigma_1024_w4a8"] = {"error": "%s: %s" % (type(e).__name__, e)}
        extras["note"] = ("other single-GPU configurations of BASELINE.json on the same build; synthetic calibration "
                          "(viditq_amd.synth); NOT the headline value")

    if rank == 0:
        from viditq_amd import shard
        n_prompts = head["n_prompts"]
        steps_of = [a.steps * len(shard.prompts_of_rank(n_prompts, r, world)) for r in range(world)]
        value = sum(steps_of) / el_max
        plan_name = {"w8a8": "W8A8", "w4a8": "W4A8 (timestep-aware channel balancing)",
                     "w4a8_mp": "W4A8 mixed precision (per-layer bit widths)"}[a.plan]
        plan_yaml = {"w8a8": "w8a8_dynamic.yaml", "w4a8": "w4a8_timestep_aware_cb.yaml, synthetic calibration",
                     "w4a8_mp": "w4a8_timestep_aware_cb.yaml + t20 mixed-precision config, synthetic calibration"}[a.plan]
        line = {"metric": "denoising steps/sec (whole node), OpenSORA STDiT 16x512x512 " + plan_name, "value": value,
                "unit": "denoising steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": el_max / max(steps_of) * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None,
                "dtype": ("int8 (W8A8 Linear, int32 acc)" if a.plan == "w8a8" else "int8 x int4/int8 weights (W4A8 Linear, int32 acc)")
                         + " + fp16 attention/residual",
                "data": "synthetic (random-init STDiT-XL/2 weights, N(0,1) latents, random text embeds)",
                "config": {"workload": "OpenSORA STDiT-XL/2 16x512x512 %s (%s), 1 prompt per GPU, "
                                       "DDIM-%d schedule, cfg 4.0, cfg_split, depth %d" % (plan_name, plan_yaml, head["n_sampling"], a.depth)
                                       + ("" if not a.prompts else "; %d prompts round-robin over the ranks, per prompt: own graph "
                                          "capture (untimed), %d warm-up + %d timed steps; time = sum of the timed regions"
                                          % (n_prompts, a.warmup, a.steps)),
                           "tokens": 16384, "prompts_in_flight": world, "prompts": n_prompts, "sharding": "prompt -> rank (no in-step collective)",
                           "status_word": head["status"], "hip_graph": not a.no_graph,
                           "cond_uncond_streams": 1 if (a.one_stream or a.no_graph) else 2},
                # self-diagnosis of a multi-GPU run: every rank's own rate, and what the one set-up collective moved
                "per_rank_steps_per_s": [k / t for k, t in zip(steps_of, per_rank)],
                "weights_broadcast": (dict(head["broadcast"], per_rank_seconds=bc_rank) if head["broadcast"] and bc_rank
                                      else head["broadcast"]),
                "collectives": collective_info(dist, rehearsal),
                "prompt_invariants_hoisted": head["cached"],
                "whole_step_int8_frac": 43.87e12 * (a.depth / 28.0) * value / world / PEAK_INT8,
                "roofline": head["roofline"],
                # what the part was doing during THIS leg (power / shader clock / temperature before, during and after the
                # timed region; the clock the GEMM itself ran at right after it) and how far ahead of the GPU the host was
                "telemetry": head.get("telemetry"),
                "host_enqueue_ms_per_step": head["host_enqueue_ms_per_step"],
                # (the figure above includes the time hipGraphLaunch waits on a FULL hardware queue once the host is ~7 steps
                #  ahead; the one below is a replay launched into an empty queue - the host's own cost per step)
                "host_graph_launch_ms_idle_queue": head.get("host_graph_launch_ms_idle_queue"),
                "graph_nodes": head.get("graph_nodes"),
                "extras": extras}
        if rehearsal:
            line["rehearsal"] = "ranks share devices, gloo backend: control-flow dry run, NOT a measurement"
        if world == 1 and not a.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # noqa: BLE001  (a reported side measurement: its failure must not lose the GPU line)
                line["cpu_baseline"] = {"value": None, "unit": "denoising steps/s", "cores": 0, "kind": "port",
                                        "sample": "failed: %s: %s" % (type(e).__name__, e)}
        check_line(line, a)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
