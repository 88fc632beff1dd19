"""HIP-graph capture of the per-step forwards.

One denoising step launches ~2300 kernels from Python (28 blocks x ~20 kernels x 2 forward-samples);
at ~64 ms of GPU work per step the eager loop is partly host-bound.  Shapes, buffers and the kernel
sequence are static within a smooth-quant time-range, so both forward-samples of a step (cond +
uncond) are captured ONCE into a HIP graph (``torch.cuda.CUDAGraph`` records the raw
``hipLaunchKernelGGL`` calls our C ABI makes on the capturing stream) and replayed with new latent /
timestep / text-embedding contents copied into the static input buffers.  One graph per time-range
(the packed weights and smoothing vectors differ between ranges) and per mixed-precision key (the
per-layer bit widths and FP layer set differ between keys, iddpm.TimestepMP).
"""
from __future__ import annotations

import os
from typing import Dict, Hashable, Optional, Tuple

import torch


class StepGraph:
    """cond/uncond forwards of a QuantModel for one prompt as a replayable HIP graph."""

    def __init__(self, qnn, x: torch.Tensor, y_cond: torch.Tensor, y_uncond: torch.Tensor,
                 mask: Optional[torch.Tensor], timestep_id: int, warmup: int = 2, two_streams: bool = True):
        self.qnn = qnn
        # cond and uncond are independent chains: captured as two parallel branches of the graph (fork /
        # join through a second HIP stream) so that the HBM-bound kernels of one chain (quantizers,
        # temporal attention: few VGPRs, no LDS) can run on CUs whose MFMA pipes the other chain's GEMM
        # workgroups keep busy
        self.two_streams = two_streams
        self.side = torch.cuda.Stream() if two_streams else None
        self.x = x.clone()
        self.t = torch.full((x.shape[0],), int(timestep_id), device=x.device, dtype=torch.long)
        self.yc, self.yu = y_cond.clone(), y_uncond.clone()
        self.mask = mask
        self.cfg_split = bool(getattr(qnn, "cfg_split", False))
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            # first pass on ONE stream: it packs weights and fills the module-level caches (fused qkv weights, prompt
            # K/V, mask selection).  Run on two streams that pass would let the cond chain read a cached tensor that the
            # uncond chain's stream is still producing - and such a tensor would persist into the captured graph
            self._forward(timestep_id, serial=True)
            side.synchronize()
            for _ in range(warmup):                      # then the capture's own launch pattern
                self._forward(timestep_id)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        dump = os.environ.get("VQ_GRAPH_DUMP")          # measurement: a .dot file of the captured graph (node census, bench.py)
        if dump:
            self.graph.enable_debug_mode()
        from . import _lib
        calls0 = _lib.CALLS[0]
        # thread_local: with a process group alive (N > 1) the RCCL watchdog thread polls events while we capture; in
        # the default 'global' mode such a call from ANOTHER thread would invalidate the capture
        with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.cond, self.uncond = self._forward(timestep_id)
        # C-ABI calls recorded into the graph (= its HIP-kernel nodes from this library: one launch per entry point, the
        # fast quantizer entry points included; the ~60 torch elementwise kernels of the FP edges come on top)
        self.c_abi_calls = _lib.CALLS[0] - calls0
        self.dot_path = None
        if dump:
            try:
                self.dot_path = "%s.%d.dot" % (dump, id(self))
                self.graph.debug_dump(self.dot_path)
            except Exception:                           # the census is optional; the capture itself stands
                self.dot_path = None

    def _forward(self, t_id, serial=False):
        q = self.qnn
        if self.cfg_split and self.two_streams and not serial:
            cur = torch.cuda.current_stream()
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                unc = q(self.x, self.t, self.yu, mask=self.mask, timestep_id=t_id)
            cond = q(self.x, self.t, self.yc, mask=self.mask, timestep_id=t_id)
            cur.wait_stream(self.side)
            return cond, unc
        if self.cfg_split:
            return (q(self.x, self.t, self.yc, mask=self.mask, timestep_id=t_id),
                    q(self.x, self.t, self.yu, mask=self.mask, timestep_id=t_id))
        n = self.x.shape[0]
        out = q(torch.cat([self.x, self.x]), torch.cat([self.t, self.t]), torch.cat([self.yc, self.yu]),
                mask=self.mask, timestep_id=t_id)
        return out[:n], out[n:]

    def run(self, x: torch.Tensor, timestep_id: int):
        self.x.copy_(x)
        self.t.fill_(int(timestep_id))
        self.graph.replay()
        return self.cond, self.uncond


class GraphedSampler:
    """Lazily captures one StepGraph per (smooth-quant time-range, mixed-precision key) of ``qnn``."""

    def __init__(self, qnn, y_cond, y_uncond, mask, two_streams: bool = True):
        self.qnn, self.yc, self.yu, self.mask = qnn, y_cond, y_uncond, mask
        self.two_streams = two_streams
        self.graphs: Dict[Tuple[int, Hashable], StepGraph] = {}
        self._epoch = None

    def _range_of(self, t_id: int) -> int:
        from .qdiff.models.quant_layer import find_interval
        for _, layer in self.qnn.quant_layers():
            if getattr(layer, "smooth_quant", False) and hasattr(layer, "timerange"):
                return find_interval(layer.timerange, t_id)
        return 0

    def forward_pair(self, x, t_id: int, mp_key: Hashable = None):
        """``mp_key``: whatever identifies the current per-layer bit-width / FP-layer state (the caller has
        already applied it to ``qnn``); a new key captures a new graph."""
        from .qdiff.models.quant_layer import PACK_EPOCH
        if self._epoch != PACK_EPOCH[0]:
            # some layer dropped its packed weights since the last capture (invalidate_packed / set_quant_params_dict):
            # the captured graphs reference the freed buffers - drop them, the next call captures afresh
            self.graphs.clear()
            self._epoch = PACK_EPOCH[0]
        r = (self._range_of(t_id), mp_key)
        g = self.graphs.get(r)
        if g is None:
            g = self.graphs[r] = StepGraph(self.qnn, x, self.yc, self.yu, self.mask, t_id,
                                           two_streams=self.two_streams)
        return g.run(x, t_id)


def _ident(v):
    """Identity of a keyword argument for the graph key: tensors by storage and version (their CONTENT is baked into the
    capture through host-side decisions such as the prompt-token selection), containers recursively, scalars by value."""
    if torch.is_tensor(v):
        return ("t", v.data_ptr(), v._version, tuple(v.shape), str(v.dtype))
    if isinstance(v, dict):
        return tuple((k, _ident(x)) for k, x in sorted(v.items()))
    if isinstance(v, (list, tuple)):
        return tuple(_ident(x) for x in v)
    return v


class ForwardGraph:
    """One forward ``fn(x, t, y, **kwargs)`` as a replayable HIP graph with static x / t / y buffers."""

    def __init__(self, fn, x, t, y, kwargs, warmup: int = 2):
        self.x, self.t, self.y = x.clone(), t.clone(), y.clone()
        self.kwargs = kwargs
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup + 1):                  # packs weights, fills the host-side caches (mask selection, ...)
                fn(self.x, self.t, self.y, **kwargs)
            side.synchronize()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.out = fn(self.x, self.t, self.y, **kwargs)

    def run(self, x, t, y):
        self.x.copy_(x)
        self.t.copy_(t)
        self.y.copy_(y)
        self.graph.replay()
        return self.out


class GraphedModel:
    """``model(x, t, y, **kwargs)`` for a sampling loop that calls one forward per step with the same shapes and the same
    keyword arguments (the t2i DPM-Solver loop, quant_txt2img.py:130-153: ``DPMS_sigma(GraphedModel(qnn.forward_with_dpmsolver),
    ...)``).  The ~540 launches of a PixArt forward are captured once per (shapes, keyword identities, smooth-quant
    time-range, per-layer quant state and bit widths, pack epoch) and replayed with the new latent / timestep / text embedding copied into the static inputs;
    the result is a fresh tensor each call (multistep solvers keep earlier outputs).  Bit-identical to eager launches
    (tested).  At PixArt-Sigma 1024 x 1024 a forward is 19.9 ms of GPU work either way (50.3 steps/s eager, 49.7
    replayed): eager Python launches need 10.9 ms of host time per step and stay ahead of the GPU, a replay needs 3.0 ms;
    the replay pays when the forward is shorter than its ~25 us per launch from Python (small latents) or the host
    is needed for something else."""

    def __init__(self, fn, qnn=None, max_graphs: int = 8):
        self.fn, self.qnn = fn, qnn if qnn is not None else getattr(fn, "__self__", None)
        self.graphs: Dict[Hashable, ForwardGraph] = {}
        self.max_graphs = max_graphs
        self._epoch = None
        self._qlayers = None

    def _layers(self):
        """The QuantLayers under the wrapped model, collected once (wrapping happens before the loop; the module tree
        does not change afterwards)."""
        if self._qlayers is None:
            root = self.qnn
            from .qdiff.models.quant_layer import QuantLayer
            self._qlayers = [m for m in root.modules() if isinstance(m, QuantLayer)] \
                if root is not None and hasattr(root, "modules") else []
        return self._qlayers

    def _state(self, t_id):
        """ONE pass over the cached layer list per call: (fingerprint of everything that changes which kernels a forward
        launches - per-layer quant state, weight / activation bit widths, smooth-quant switch -, the smooth-quant
        time-range of the forward, whether some layer keeps host-visible running state).  The attributes are plain
        Python fields that scripts also set directly (quant_txt2img.py:297-300), so they are read, not tracked."""
        from .qdiff.models.quant_layer import find_interval
        fp, rng, running = [], None, False
        for m in self._layers():
            sq = bool(getattr(m, "smooth_quant", False))
            fp.append((m.weight_quant, m.act_quant, m.weight_quantizer.n_bits, m.act_quantizer.n_bits, sq))
            if sq and rng is None and hasattr(m, "timerange"):
                # from ``timestep_id`` when the call passes one (QuantModel.forward), else from the state the script set
                # on the layers (set_timestep_id_for_quantlayer)
                rng = find_interval(m.timerange, int(t_id)) if t_id is not None else int(m._range_and_alpha()[0])
            if getattr(m, "smooth_quant_running_stat", False) and "momentum" in str(getattr(m, "channel_wise_scale_type", "")):
                # a RUNNING smooth-quant statistic at inference (the released t2i W4A8 plan leaves it on for
                # blocks.27.mlp.fc2, quant_txt2img.py:297-300) updates host-visible state in every forward: not capturable
                running = True
        return hash(tuple(fp)), rng or 0, running

    def __call__(self, x, t, y, **kwargs):
        from .qdiff.models.quant_layer import PACK_EPOCH
        if not x.is_cuda:
            return self.fn(x, t, y, **kwargs)
        fingerprint, rng, running = self._state(kwargs.get("timestep_id"))
        if running:
            return self.fn(x, t, y, **kwargs)
        if self._epoch != PACK_EPOCH[0]:
            self.graphs.clear()                          # captured graphs reference re-packed weight buffers
            self._epoch = PACK_EPOCH[0]
        kw = {k: v for k, v in kwargs.items() if k != "timestep_id"}
        key = (tuple(x.shape), str(x.dtype), tuple(t.shape), str(t.dtype), tuple(y.shape), str(y.dtype), _ident(kw),
               rng, fingerprint)
        g = self.graphs.get(key)
        if g is None:
            if len(self.graphs) >= self.max_graphs:
                self.graphs.clear()
            g = self.graphs[key] = ForwardGraph(self.fn, x, t, y, kwargs)
            if self._epoch != PACK_EPOCH[0]:             # the first pass packed weights: the capture came after it
                self._epoch = PACK_EPOCH[0]
        return g.run(x, t, y).clone()
