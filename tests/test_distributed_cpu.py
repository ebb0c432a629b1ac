"""world_size-2 gloo tests (CPU) of the data-parallel path: sharding, loss-scaling contract,
bucketed flat all-reduce, C3/C4 scalar collectives."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from a3t_amd import trainer as T


def test_shard_batches_and_buckets():
    batches = [["a", "b", "c", "d", "e"], ["f", "g"]]
    assert T.shard_batches(batches, 0, 2) == [["a", "c", "e"], ["f"]]
    assert T.shard_batches(batches, 1, 2) == [["b", "d"], ["g"]]
    with pytest.raises(RuntimeError):
        T.shard_batches([["a"]], 0, 2)
    r = T.bucket_ranges(1000, [100, 300, 350, 900], 200)
    assert r[0][1] == 1000 and r[-1][0] == 0
    assert all(a[0] == b[1] for a, b in zip(r[:-1], r[1:]))            # contiguous, descending
    assert all(hi - lo >= 200 for lo, hi in r[:-1])
    assert abs(T.noam_lr(1, 1.0, 384, 4000) - 384 ** -0.5 * 4000 ** -1.5) < 1e-15
    assert abs(T.noam_lr(4000, 1.0, 384, 4000) - 384 ** -0.5 * 4000 ** -0.5) < 1e-15


def _worker(rank, world, init_file, out_file):
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    torch.manual_seed(0)
    # global batch of 5 utterances, strided over ranks; per-utterance "gradient" = f(utt)
    keys = list(range(5))
    mine = T.shard_batches([keys], rank, world)[0]
    n = 1000
    per_utt = {k: torch.from_numpy(np.random.RandomState(k).standard_normal(n).astype(np.float32)) for k in keys}
    # rank-local mean loss gradient and weight (= local batch size)
    g_local = torch.stack([per_utt[k] for k in mine]).mean(0)
    w_local = torch.tensor([len(mine)])
    stats, w_tot = T.average_stats({"loss": torch.tensor([float(sum(mine)) / len(mine)])}, w_local)
    scale = T.grad_scale(float(w_local), float(w_tot), world)
    flat = g_local * scale
    ranges = T.bucket_ranges(n, [100, 400, 700], 250)
    red = T.FlatAllReduce(flat, ranges)
    for i in range(len(ranges)):
        red.reduce_range(i)
    red.wait()
    flat /= world                                  # the DDP division (folded into clip_adam's gscale on the GPU)
    stop = T.iterator_stop(rank == 1)
    if rank == 0:
        torch.save(dict(flat=flat, loss=stats["loss"], w=w_tot, stop=stop), out_file)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_equals_global_batch_mean():
    with tempfile.TemporaryDirectory() as d:
        init, out = os.path.join(d, "init"), os.path.join(d, "out.pt")
        mp.spawn(_worker, args=(2, init, out), nprocs=2, join=True)
        r = torch.load(out)
    keys = list(range(5))
    ref = torch.stack([torch.from_numpy(np.random.RandomState(k).standard_normal(1000).astype(np.float32))
                       for k in keys]).mean(0)
    np.testing.assert_allclose(r["flat"].numpy(), ref.numpy(), atol=1e-6)
    assert abs(float(r["loss"]) - sum(keys) / 5) < 1e-6 and float(r["w"]) == 5.0
    assert r["stop"] is True


def _overlap_worker(rank, world, init_file, out_file):
    """A3TTrainer's overlapped reduction driven by a fake engine on CPU tensors: every flat-buffer
    element must be all-reduced exactly once, whatever order the backward schedule reports groups in."""
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    from a3t_amd.config import A3TConfig
    from a3t_amd.params import ParamStore
    c = A3TConfig(adim=32, heads=2, ff=64, enc_blocks=2, dec_blocks=2, postnet_layers=2, postnet_chans=16, vocab=11)
    store = ParamStore(c, "cpu")

    class FakeEngine:
        def backward(self, on_group_done=None):
            names = ["sfc.w"] + [f"dec.{i}.ffm.ln.g" for i in (1, 0)] + [f"enc.{i}.ffm.ln.g" for i in (1, 0)] + ["seg"]
            hi = store.total
            for n in names:                       # gradients become final from the top of the buffer down
                lo = store.offsets[n][0]
                store.grad[lo:hi] = float(rank + 1)
                on_group_done(n)
                hi = lo

    tr = T.A3TTrainer.__new__(T.A3TTrainer)
    tr.store, tr.engine, tr.cfg = store, FakeEngine(), c
    bounds = [store.offsets[f"enc.{i}.ffm.ln.g"][0] for i in range(2)] + [store.offsets[f"dec.{i}.ffm.ln.g"][0] for i in range(2)]
    bounds += [store.offsets["sfc.w"][0]]
    tr.ranges = T.bucket_ranges(store.total, bounds, 10000)
    tr.reducer = T.FlatAllReduce(store.grad, tr.ranges)
    store.grad.zero_()
    tr._backward_overlapped()
    ok = bool((store.grad == float(sum(range(1, world + 1)))).all())
    if rank == 0:
        torch.save(dict(ok=ok, nranges=len(tr.ranges)), out_file)
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_bucket_reduction_covers_flat_buffer_once():
    with tempfile.TemporaryDirectory() as d:
        init, out = os.path.join(d, "init"), os.path.join(d, "out.pt")
        mp.spawn(_overlap_worker, args=(2, init, out), nprocs=2, join=True)
        r = torch.load(out)
    assert r["ok"] and r["nranges"] >= 2


def test_bench_launches_n_ranks_itself():
    """`python bench.py --gpus 2` with no launcher in the environment must start 2 ranks (mp.spawn, one per GPU like
    abs_task.py:1026-1045) and report n_gpus = the RCCL/gloo world size; --fake-cpu swaps the kernels for a gloo
    all-reduce so the launch path is testable without a GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--fake-cpu", "--steps", "3"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                       # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dist_world_size"] == 2 and out["rank_sum"] == 3.0 and out["steps"] == 3
    # --gpus must agree with a launcher's WORLD_SIZE
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--fake-cpu"],
                       capture_output=True, text=True, timeout=120, env=dict(env, WORLD_SIZE="4", RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout)


def _bf16_bucket_worker(rank, world, init_file, out_file):
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    n = 4096
    g = torch.from_numpy(np.random.RandomState(rank).standard_normal(n).astype(np.float32))
    ref = g.clone()
    dist.all_reduce(ref)
    red = T.FlatAllReduce(g, T.bucket_ranges(n, [1000, 3000], 1000), comm_dtype=torch.bfloat16)
    for i in range(len(red.ranges)):
        red.reduce_range(i)
    red.wait()
    if rank == 0:
        torch.save(dict(err=float((g - ref).abs().max()), scale=float(ref.abs().max()), nbytes=red.bytes_per_step), out_file)
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_bucket_allreduce_matches_fp32_within_bf16_rounding():
    with tempfile.TemporaryDirectory() as d:
        init, out = os.path.join(d, "init"), os.path.join(d, "out.pt")
        mp.spawn(_bf16_bucket_worker, args=(2, init, out), nprocs=2, join=True)
        r = torch.load(out)
    assert r["err"] <= 2.0 ** -7 * r["scale"] and r["nbytes"] == 4096 * 2


def test_loss_weight_and_accumulation_contract():
    """trainer.py:583-597: loss_r * w_r / sum(w) * world / accum_grad; summed over ranks and divided by world this is
    the global-batch mean -- with unequal shards (3 + 2 utterances) and two accumulation micro-steps."""
    world, accum = 2, 2
    per_utt = [np.random.RandomState(k).standard_normal(8) for k in range(10)]
    micro = [list(range(0, 5)), list(range(5, 10))]
    total = np.zeros(8)
    for mb in micro:
        for rank in range(world):
            mine = T.shard_batches([mb], rank, world)[0]
            g_local = np.mean([per_utt[k] for k in mine], axis=0)          # rank-local mean loss gradient
            total += g_local * T.grad_scale(len(mine), len(mb), world) / accum
    total /= world
    np.testing.assert_allclose(total, np.mean([np.mean([per_utt[k] for k in mb], axis=0) for mb in micro], axis=0), atol=1e-12)
