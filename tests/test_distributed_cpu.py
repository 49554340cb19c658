"""world_size-2 gloo tests (CPU) of the data-parallel layer: process-group helpers and the
flat-bucket gradient reducer that replaces DistributedDataParallel (dino.py:190-193)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
    import dino_amd.distributed as D
    from dino_amd.engine.flat import FlatLayout
    try:
        D.enable(backend="gloo")
        assert D.is_enabled() and D.is_enabled_and_multiple_gpus()
        assert D.get_global_size() == world and D.get_global_rank() == rank
        assert D.get_local_rank() == rank and D.get_local_size() == world
        assert D.is_main_process() == (rank == 0)
        shapes = {"a.weight": (300, 70), "a.bias": (70,), "b.weight": (50, 50), "c.g": (33, 1),
                  "d.weight": (20000,)}
        lay = FlatLayout(shapes)
        grad = torch.zeros(lay.total)
        for i, n in enumerate(lay.names):
            lay.view(grad, n).fill_(float((rank + 1) * (i + 1)))
        red = D.FlatGradReducer(lay, grad, bucket_bytes=64 * 1024)
        assert len(red.buckets) >= 2
        covered = sorted((b["start"], b["end"]) for b in red.buckets)
        assert covered[0][0] == 0 and covered[-1][1] == lay.total
        assert all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))
        # backward order: last tensors first; "c.g" never reports (frozen) -> flushed by finish()
        red.mark_ready(["d.weight"])
        red.mark_ready(["b.weight", "a.bias"])
        red.mark_ready(["a.weight"])
        red.finish()
        mean_scale = sum(r + 1 for r in range(world)) / world
        for i, n in enumerate(lay.names):
            exp = mean_scale * (i + 1)
            assert torch.allclose(lay.view(grad, n), torch.full(shapes[n], exp)), n
        # second step reuses the reducer
        grad.fill_(float(rank))
        red.mark_ready(list(lay.names))
        red.finish()
        assert torch.allclose(grad, torch.full_like(grad, (world - 1) / 2))
        # strict mode: a trainable tensor that is never announced is an error, a frozen one is not
        red = D.FlatGradReducer(lay, grad, bucket_bytes=64 * 1024, strict=True, frozen=["c.g"])
        red.mark_ready(["d.weight", "b.weight", "a.bias", "a.weight"])
        red.finish()
        red.mark_ready(["d.weight", "b.weight", "a.weight"])
        try:
            red.finish()
            raise AssertionError("strict finish() accepted an unannounced tensor")
        except RuntimeError as e:
            assert "a.bias" in str(e)
        # the engine's announcement order over the real ViT-S/16, ViT-B/16 and ViT-L/14 layouts: every
        # bucket must be launched by mark_ready during backward, none left to the final flush
        from collections import OrderedDict
        from dino_amd.engine.vit import backward_ready_sequence
        from dino_amd.models import vision_transformer as vits
        for ctor, P in ((vits.vit_small, 16), (vits.vit_base, 16), (vits.vit_large, 14)):
            bb = ctor(patch_size=P)
            hd = vits.DINOHead(bb.embed_dim, 65536, norm_last_layer=False)
            shapes = OrderedDict([("backbone." + k, tuple(p.shape)) for k, p in bb.named_parameters()]
                                 + [("head." + k, tuple(p.shape)) for k, p in hd.named_parameters()])
            depth = len(bb.blocks)
            del bb, hd
            lay2 = FlatLayout(shapes)
            g2 = torch.zeros(lay2.total)
            g2.fill_(float(rank + 1))
            red2 = D.FlatGradReducer(lay2, g2, strict=True)
            assert len(red2.buckets) >= 2
            launched_early = []
            for names in backward_ready_sequence(list(shapes), depth):
                red2.mark_ready(names)
                launched_early.append(len(red2._works))
            assert launched_early[-1] == len(red2.buckets), (ctor.__name__, launched_early)
            assert launched_early[0] >= 1            # the head's bucket(s) start before the backbone
            red2.finish()                            # strict: raises if anything was left for the flush
            assert red2.late_buckets == []
            assert torch.allclose(g2, torch.full_like(g2, (1 + world) / 2))
        # eval-side feature gather (reference dino/eval/features.py:55-72,136-163): each rank embeds its
        # shard of a 10-sample dataset; the main process ends with every row at its dataset index
        from dino_amd.eval import extract_features, extract_multiple_features

        class _DS(torch.utils.data.Dataset):
            def __len__(self):
                return 10

            def __getitem__(self, i):
                g = torch.Generator().manual_seed(i)
                return i, torch.randn(6, generator=g), i % 3
        ds = _DS()
        shard = torch.utils.data.Subset(ds, list(range(rank, 10, world)))
        shard.dataset_len = 10

        class _Loader:
            dataset = ds

            def __iter__(self):
                return iter(torch.utils.data.DataLoader(shard, batch_size=5))
        torch.manual_seed(0)
        ms, mt = torch.nn.Linear(6, 4), torch.nn.Linear(6, 3)
        feats, labels = extract_multiple_features(ms, mt, _Loader(), torch.device("cpu"))
        f1, l1 = extract_features(ms, _Loader(), torch.device("cpu"))
        if rank == 0:
            allx = torch.stack([ds[i][1] for i in range(10)])
            with torch.no_grad():
                assert torch.allclose(feats["student"], ms(allx), atol=1e-6)
                assert torch.allclose(feats["teacher"], mt(allx), atol=1e-6)
                assert torch.allclose(f1, ms(allx), atol=1e-6)
            assert labels.tolist() == [i % 3 for i in range(10)] == l1.tolist()
        else:
            assert feats["student"] is None and f1 is None
        # centre all-reduce semantics (dino/components/loss.py:69-74): SUM then / (rows * world)
        acc = torch.full((8,), float(rank + 1))
        torch.distributed.all_reduce(acc)
        assert torch.allclose(acc, torch.full((8,), float(sum(r + 1 for r in range(world)))))
        D.disable()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.timeout(180)
def test_flat_grad_reducer_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(r[1] == "ok" for r in res), res
