"""Multi-process (world_size 2, gloo, CPU) test of the N>1 path: clip sharding covers every clip exactly once,
the weight broadcast delivers identical tensors, and the timing reduction takes the max over ranks."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from neurons_amd.distributed import broadcast_state_dict, clip_indices_for_rank, global_clip_index, max_over_ranks
    from neurons_amd.unet3d import UNet3DConfig, random_state_dict, state_dict_schema
    cfg = UNet3DConfig(block_out_channels=(64, 64, 128, 128), cross_attention_dim=64, use_motion_module=False,
                       motion_module_kwargs={})
    schema = state_dict_schema(cfg)
    sd = random_state_dict(cfg, seed=7) if rank == 0 else None
    got = broadcast_state_dict(schema, sd, src=0)
    checksum = float(sum(v.double().sum() for v in got.values()))
    mine = clip_indices_for_rank(11, rank, world)
    assert all(global_clip_index(rank, i, world) == g for i, g in enumerate(mine))
    slowest = max_over_ranks(1.0 + rank)
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, checksum, slowest))
    if rank == 0:
        q.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_sharding_and_broadcast():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (c0, s0, m0), (c1, s1, m1) = res
    assert sorted(c0 + c1) == list(range(11)) and not set(c0) & set(c1)
    assert c0 == [0, 2, 4, 6, 8, 10] and c1 == [1, 3, 5, 7, 9]
    assert s0 == s1                      # identical weights on both ranks
    assert m0 == m1 == 2.0               # max over ranks


class _FakeNet:
    """Stands in for a native network on CPU: export_weights / import_weights with the manifest format of the C ABI
    (include/neurons_amd.h: "D <name> <offset> <bytes>", 256-byte aligned, name order), so the world-size-2 plumbing of
    broadcast_native_weights (object broadcast of the manifest + ONE byte-arena broadcast) is exercised without a GPU."""

    def __init__(self, tensors=None):
        self.device = torch.device("cpu")
        self.tensors = tensors or {}

    def export_weights(self):
        lines, off, parts = ["NRW1 0"], 0, []
        for name in sorted(self.tensors):
            raw = self.tensors[name].contiguous().view(torch.uint8).reshape(-1)
            lines.append(f"D {name} {off} {raw.numel()}")
            pad = (-raw.numel()) % 256
            parts += [raw, torch.zeros(pad, dtype=torch.uint8)]
            off += raw.numel() + pad
        return ("\n".join(lines) + "\n").encode(), torch.cat(parts)

    def import_weights(self, manifest, arena):
        out = {}
        for line in manifest.decode().splitlines()[1:]:
            _, name, off, nb = line.split(" ")
            out[name] = arena[int(off):int(off) + int(nb)].clone()
        self.tensors = out


def _worker_native(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from neurons_amd.distributed import broadcast_native_weights
    g = torch.Generator().manual_seed(5)
    src = {"lin:b.weight": torch.randn(7, 5, generator=g).to(torch.bfloat16), "f32:a.bias": torch.randn(33, generator=g),
           "cat:q|k|v|": torch.randn(3, 300, generator=g).to(torch.bfloat16)}
    net = _FakeNet(src if rank == 0 else None)
    broadcast_native_weights(net, src=0)
    got = {k: v.view(torch.uint8).reshape(-1) if rank == 0 else v for k, v in net.tensors.items()}
    digest = {k: int(v.to(torch.int64).sum()) for k, v in got.items()}
    gathered = [None] * world
    dist.all_gather_object(gathered, digest)
    if rank == 0:
        q.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_native_weight_arena_broadcast():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_native, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1] and set(res[0]) == {"lin:b.weight", "f32:a.bias", "cat:q|k|v|"}
