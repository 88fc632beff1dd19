"""Multi-process CPU tests (gloo, world_size 2) of the prompt-sharding path: the partition, the
single flat-buffer broadcast of quant grids + packed weights, and the gather of final latents.
On GPUs the same code runs over RCCL (backend 'nccl'); the logic is backend independent."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tiny_qnn():
    import viditq_amd  # noqa
    from viditq_amd.config import to_config
    from viditq_amd.qdiff.models import QuantModel
    from viditq_amd.t2v import STDiT
    torch.manual_seed(0)
    m = STDiT(input_size=(4, 8, 8), depth=1, hidden_size=64, num_heads=4, model_max_length=12, caption_channels=32)
    wq = to_config(dict(n_bits=8, per_group="channel", channel_dim=0, scale_method="min_max", round_mode="nearest"))
    aq = to_config(dict(n_bits=8, per_group="token", scale_method="min_max", round_mode="nearest_ste",
                        running_stat=False, dynamic=True, sym=False, n_spatial_token=16, n_temporal_token=4,
                        n_prompt=12, smooth_quant=dict(enable=False)))
    qnn = QuantModel(m, wq, aq)
    qnn.set_module_name_for_quantizer(qnn.model)
    return qnn


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from viditq_amd import ops, shard
        qnn = _tiny_qnn()
        layers = dict(qnn.quant_layers())
        g = torch.Generator().manual_seed(123)
        if rank == 0:   # fabricate the state rank 0 would hold after weight PTQ + packing
            for name, layer in layers.items():
                N, K = layer.weight.shape
                wq = layer.weight_quantizer
                wq.delta_list = torch.rand(1, 1, N, 1, generator=g)
                wq.zero_point_list = torch.randint(0, 255, (1, 1, N, 1), generator=g).float()
                wq.delta, wq.zero_point = wq.delta_list[0, 0], wq.zero_point_list[0, 0]
                Kp = ops.pad128(K)
                pw = ops.PackedWeight(torch.randint(-128, 127, (N, Kp), generator=g, dtype=torch.int8),
                                      torch.rand(N, generator=g), torch.randint(-128, 127, (N,), generator=g, dtype=torch.int32),
                                      torch.randint(-9999, 9999, (N,), generator=g, dtype=torch.int32), N, K, Kp, 8)
                layer.install_packed(0, pw)
        nbytes = shard.broadcast_quant_state(qnn, rank, src=0)
        # every rank now holds identical grids and packed weights: checksum of checksums
        acc = torch.zeros(1, dtype=torch.float64)
        for name, layer in sorted(layers.items()):
            pw = layer._packed[(0, 8)][0]
            assert layer._packed[(0, 8)][1] is layer.weight_quantizer.delta
            assert pw.wq.dtype == torch.int8 and pw.N == layer.weight.shape[0]
            acc += pw.wq.double().sum() + pw.sw.double().sum() + pw.zw.double().sum() + pw.cs.double().sum()
            acc += layer.weight_quantizer.delta_list.double().sum()
        both = [torch.zeros_like(acc) for _ in range(world)]
        dist.all_gather(both, acc)
        assert torch.equal(both[0], both[1]) and nbytes > 0
        # partition + gather: 5 prompts over 2 ranks, round robin
        n_prompts = 5
        mine = shard.prompts_of_rank(n_prompts, rank, world)
        x_local = torch.stack([torch.full((2, 3), float(i)) for i in mine])
        full = shard.gather_latents(x_local, mine, n_prompts, rank, world)
        assert torch.equal(full[:, 0, 0], torch.arange(n_prompts, dtype=torch.float32))
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_partition_is_a_bijection():
    import viditq_amd  # noqa
    from viditq_amd import shard
    for n, w in ((64, 8), (5, 2), (3, 4), (1, 1)):
        parts = [shard.prompts_of_rank(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_blob_roundtrip_is_zero_copy_and_aligned():
    import viditq_amd  # noqa
    from viditq_amd import shard
    ts = [("a", torch.arange(7, dtype=torch.int8)), ("b", torch.rand(3, 5)), ("c", torch.arange(4, dtype=torch.int32))]
    meta, blob = shard.pack_blob(ts, "cpu")
    out = shard.unpack_blob(meta, blob)
    for name, t in ts:
        assert torch.equal(out[name], t) and out[name].dtype == t.dtype
        assert out[name].data_ptr() >= blob.data_ptr() and (out[name].data_ptr() - blob.data_ptr()) % 256 == 0


@pytest.mark.timeout(300)
def test_broadcast_and_gather_world2_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}
