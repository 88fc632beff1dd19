"""Round 6 (lab): attn_fwd64s_kernel (tools/lab/attn_stream.h, a library built with -DVQ_ATTN_STREAM_LAB, VIDITQ_LIB pointing
at it) against the fp32 softmax reference and BIT-IDENTICAL to the product's 32-query kernel (VQ_ATTN_STREAM=0 is read per call) on
five shapes: the spatial shape, ragged tiles, odd tile counts, a one-row second tile, D = 64.  GPU box only: pytest tools/attn_stream_check.py"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import h16, rel_l2  # noqa: E402
from test_kernels_gpu import _attn_ref  # noqa: E402
from conftest import *  # noqa: E402,F401,F403  (the ops / dev fixtures)


@pytest.mark.gpu
@pytest.mark.parametrize("n_seq,Lq,Lk,H,D", [(16, 1024, 1024, 16, 72),     # STDiT spatial attention: 256 pairs x 2 query tiles
                                             (16, 1000, 1000, 16, 72),     # ragged second query tile, ragged last key tile
                                             (11, 1536, 1111, 16, 72),     # three query tiles per pair: a two-tile and a one-tile workgroup
                                             (9, 1025, 640, 32, 64),       # second tile holds ONE row; D = 64 (8 park pieces)
                                             (8, 2000, 590, 32, 72)])      # four tiles; the fewest key tiles that carry the parked rows out (10)
def test_attn_fwd_stream_kernel(ops, dev, n_seq, Lq, Lk, H, D, monkeypatch):
    """attn_fwd64s_kernel (round 6): the two query tiles of a (sequence, head) pair as one key-tile stream, Q of the second
    tile prefetched into an LDS park, O of the first parked there and stored under the second tile's loop.  Against the fp32
    softmax reference, and BIT-IDENTICAL to the 32-query kernel it replaces for these launches (VQ_ATTN_STREAM=0 is read per
    call); rows the launch does not own keep their NaN-free sentinel."""
    Cc = H * D
    q = h16(n_seq * Lq, Cc, scale=1.0, seed=Lq + D).to(dev)
    kv = h16(n_seq * Lk, 2 * Cc, scale=1.0, seed=Lk + D).to(dev)
    o = torch.full((n_seq * Lq + 3, Cc), 7.0, dtype=torch.float16, device=dev)     # three guard rows behind the output
    monkeypatch.delenv("VQ_ATTN_STREAM", raising=False)
    ops.attn_fwd(q, kv, kv[:, Cc:], o, n_seq, Lq, Lk, H, D, Lq * Cc, Cc, Lk * 2 * Cc, 2 * Cc, Lq * Cc, Cc)
    assert torch.equal(o[n_seq * Lq:], torch.full((3, Cc), 7.0, dtype=torch.float16, device=dev))
    o = o[:n_seq * Lq]
    monkeypatch.setenv("VQ_ATTN_STREAM", "0")
    o32 = torch.full((n_seq * Lq, Cc), float("nan"), dtype=torch.float16, device=dev)
    ops.attn_fwd(q, kv, kv[:, Cc:], o32, n_seq, Lq, Lk, H, D, Lq * Cc, Cc, Lk * 2 * Cc, 2 * Cc, Lq * Cc, Cc)
    monkeypatch.delenv("VQ_ATTN_STREAM", raising=False)
    assert torch.isfinite(o).all()
    assert torch.equal(o, o32)
    # the fp32 reference on a slice of the sequences (the full einsum of the first shape is 17 GB of scores)
    ns = min(n_seq, 2)
    ref = _attn_ref(q[:ns * Lq].cpu().reshape(ns, Lq, H, D), kv[:ns * Lk, :Cc].cpu().reshape(ns, Lk, H, D),
                    kv[:ns * Lk, Cc:].cpu().reshape(ns, Lk, H, D), D ** -0.5).reshape(ns * Lq, Cc)
    assert rel_l2(o[:ns * Lq].cpu().float(), ref) < 1e-3
    assert (o[:ns * Lq].cpu().float() - ref).abs().max() < 4e-3


