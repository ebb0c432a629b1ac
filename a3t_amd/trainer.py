"""Training-step runtime for the MI355X path: the slice of ``Trainer.train_one_epoch``
(espnet2/train/trainer.py:528-703) that sits on the hot path, re-designed around one flat
gradient buffer per rank.

Data parallelism = the reference's: one process per GPU, every global batch strided over ranks
(``batch[rank::world]``, espnet2/tasks/abs_task.py:1504-1513), one gradient all-reduce per step
(C1), scalar all-reduces for the iterator-stop flag (C3) and the reported statistics (C4).  On
MI355X ``backend="nccl"`` is RCCL over xGMI.  Instead of DDP's 25 MB buckets fired from autograd
hooks, the flat fp32 gradient buffer is reduced in a few large contiguous buckets on a side HIP
stream as soon as the backward schedule has finished the parameter range each one covers
(decoder blocks first), so the xGMI transfer overlaps the rest of the backward pass.
"""
import os
from typing import Dict, List, Sequence

import torch
import torch.distributed as dist

from .config import A3TConfig


# ----------------------------------------------------------------------------------------------
# pure helpers (CPU-testable with gloo)
# ----------------------------------------------------------------------------------------------
def shard_batches(batches: Sequence[Sequence[str]], rank: int, world: int) -> List[List[str]]:
    """abs_task.py:1504-1513: every mini-batch (list of utterance keys) is strided over ranks;
    the sampler guarantees len(batch) >= world (abs_task.py:1485-1487)."""
    for b in batches:
        if len(b) < world:
            raise RuntimeError(f"The batch-size must be equal or more than world_size: {len(b)} < {world}")
    return [list(b[rank::world]) for b in batches]


def noam_lr(step: int, base_lr: float, model_size: int, warmup: int) -> float:
    """NoamLR.get_lr (espnet2/schedulers/noam_lr.py:58-65); step = number of optimizer steps taken + 1."""
    return base_lr * model_size ** -0.5 * min(step ** -0.5, step * warmup ** -1.5)


def iterator_stop(local_done: bool, device="cpu") -> bool:
    """C3: stop the epoch on every rank as soon as any rank ran out of batches (trainer.py:533-536)."""
    if not (dist.is_available() and dist.is_initialized()):
        return local_done
    t = torch.tensor([1 if local_done else 0], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return bool(t.item() > 0)


def average_stats(stats: Dict[str, torch.Tensor], weight: torch.Tensor):
    """C4: recursive_average (espnet2/torch_utils/recursive_op.py:17-44): weighted mean of every
    statistic over ranks; returns (stats, total_weight)."""
    w = weight.to(torch.float32).sum()
    out = {k: (v.to(torch.float32) * w) for k, v in stats.items() if v is not None}
    if dist.is_available() and dist.is_initialized():
        for v in out.values():
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
        dist.all_reduce(w, op=dist.ReduceOp.SUM)
    return {k: v / w for k, v in out.items()}, w


def grad_scale(local_weight: float, total_weight: float, world: int) -> float:
    """Loss scaling contract of trainer.py:583-595: loss_r * w_r / sum(w) * world, and the SUM
    all-reduce is later divided by world (DDP semantics) -> net factor w_r / sum(w) per rank."""
    return local_weight / total_weight * world


def bucket_ranges(total: int, boundaries: Sequence[int], min_elems: int) -> List[tuple]:
    """Contiguous [lo, hi) ranges of the flat buffer, cut at parameter-group boundaries so that a
    bucket becomes reducible as soon as backward has passed its lowest offset; merged until each
    holds >= min_elems (few, large collectives: xGMI rings are per-link bound)."""
    cuts = sorted(set([0, total] + [b for b in boundaries if 0 < b < total]))
    ranges, hi = [], total
    for lo in reversed(cuts[:-1]):
        if hi - lo >= min_elems or lo == 0:
            ranges.append((lo, hi))
            hi = lo
    return ranges            # ordered as backward completes them: highest offsets first


class FlatAllReduce:
    """Bucketed all-reduce of one flat gradient buffer on a side stream.

    comm_dtype=torch.bfloat16 halves the bytes on the xGMI links: each bucket is cast into a bf16 staging buffer,
    reduced, and cast back into the fp32 gradient (the optimizer state stays fp32).  The sum over ranks is then
    carried out in bf16 by RCCL, so this is an opt-in throughput knob, not the default."""

    def __init__(self, flat_grad: torch.Tensor, ranges: List[tuple], comm_dtype=torch.float32):
        self.g = flat_grad
        self.ranges = ranges
        self.cuda = flat_grad.is_cuda
        if self.cuda:
            from .engine import shared_stream
            self.stream = shared_stream(flat_grad.device, "allreduce")
        else:
            self.stream = None
        self.works = []
        self.comm_dtype = comm_dtype
        self.stage = None
        if comm_dtype != torch.float32:
            self.stage = torch.empty(flat_grad.numel(), dtype=comm_dtype, device=flat_grad.device)
        self.bytes_per_step = sum(hi - lo for lo, hi in ranges) * (2 if comm_dtype == torch.bfloat16 else 4)

    def _reduce(self, lo, hi):
        if self.stage is None:
            self.works.append((dist.all_reduce(self.g[lo:hi], op=dist.ReduceOp.SUM, async_op=True), None))
        else:
            st = self.stage[lo:hi]
            st.copy_(self.g[lo:hi])
            self.works.append((dist.all_reduce(st, op=dist.ReduceOp.SUM, async_op=True), (lo, hi)))

    def reduce_range(self, i: int):
        """Call when backward has finished writing range i (on the current stream)."""
        lo, hi = self.ranges[i]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                self._reduce(lo, hi)
        else:
            self._reduce(lo, hi)

    def wait(self):
        for w, rng in self.works:
            if rng is None:
                w.wait()
                continue
            # bf16 bucket: cast the reduced bucket back into the fp32 gradient.  Work.wait() blocks the CURRENT stream on the
            # collective's end event, so it has to be called on the stream that issues the copy-back (round 3, ADVICE: it used
            # to be called on the caller's stream, leaving the copy on self.stream unordered against the all-reduce)
            if self.cuda:
                with torch.cuda.stream(self.stream):
                    w.wait()
                    self.g[rng[0]:rng[1]].copy_(self.stage[rng[0]:rng[1]])
            else:
                w.wait()
                self.g[rng[0]:rng[1]].copy_(self.stage[rng[0]:rng[1]])
        self.works = []
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.stream)


# ----------------------------------------------------------------------------------------------
# the training step on the HIP engine
# ----------------------------------------------------------------------------------------------
class A3TTrainer:
    """forward + loss + backward + gradient all-reduce + clip + Adam + NoamLR on flat buffers
    (trainer.py:545,610,631-679; optimizer/scheduler of egs2/vctk/sedit/conf/fsp2_conformer.yaml:77-83)."""

    def __init__(self, cfg: A3TConfig, store, compute="bf16", lr=1.0, warmup_steps=4000, grad_clip=1.0,
                 betas=(0.9, 0.999), eps=1e-8, overlap=True, dropout=True, force_reducer=False,
                 bucket_min_elems=16 * 1024 * 1024, comm_dtype=torch.float32):
        from .engine import MLMEngine
        self.cfg, self.store = cfg, store
        self.engine = MLMEngine(cfg, store, compute=compute, training=True, dropout=dropout)
        dev = store.device
        self.m = torch.zeros_like(store.flat)
        self.v = torch.zeros_like(store.flat)
        self.partial = torch.zeros(1024, dtype=torch.float64, device=dev)
        self.norm = torch.zeros(1, device=dev)
        self.step_no = 0          # host mirror = number of step() calls that reached the optimizer
        # device-resident optimizer clock: [updates applied, steps skipped for a non-finite gradient norm].  Adam's bias
        # correction and the Noam schedule advance only with applied updates (trainer.py:640-679), without a host sync.
        self.opt_state = torch.zeros(2, dtype=torch.int32, device=dev)
        self.comm_dtype = comm_dtype
        self.lr, self.warmup, self.clip, self.betas, self.eps = lr, warmup_steps, grad_clip, betas, eps
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self._main = None
        if dev.type == "cuda":
            from .engine import shared_stream
            self._main = shared_stream(dev, "main", high=True)
        self.reducer = None
        if self.world > 1 or (force_reducer and dist.is_available() and dist.is_initialized()):
            # (force_reducer: run the bucketed, overlapped reduction on a 1-rank group -- test hook for the stream /
            #  event plumbing on a single GPU)
            dist.broadcast(store.flat, 0)                      # C5
            for b in store.buf.values():                       # C2 (once: BN statistics then stay rank-local)
                dist.broadcast(b, 0)
            bounds = [store.offsets[f"enc.{i}.ffm.ln.g"][0] for i in range(cfg.enc_blocks)]
            bounds += [store.offsets[f"dec.{i}.ffm.ln.g"][0] for i in range(cfg.dec_blocks)]
            bounds += [store.offsets["sfc.w"][0]]
            self.ranges = bucket_ranges(store.total, bounds, bucket_min_elems)   # default: >= 64 MB fp32 per collective
            if overlap:
                self.reducer = FlatAllReduce(store.grad, self.ranges, comm_dtype)

    def step(self, batch: Dict[str, torch.Tensor], total_weight: float = None, accum_grad: int = 1,
             accum_index: int = 0) -> torch.Tensor:
        """One iteration of the trainer loop body (see _step).  On a GPU the whole schedule runs on a HIGH-priority stream
        owned by the trainer: the data-gradient chain and its row kernels are the critical path, the weight gradients on
        the engine's (default-priority) side stream are not, and the dispatcher should hand free CUs to the former first
        (round 2, A/B: 51.0 -> 50.6 ms per step; on a CPU device the schedule runs on the caller's stream)."""
        if self._main is None:
            return self._step(batch, total_weight, accum_grad, accum_index)
        cur = torch.cuda.current_stream(self.store.device)
        self._main.wait_stream(cur)
        with torch.cuda.stream(self._main):
            loss = self._step(batch, total_weight, accum_grad, accum_index)
        cur.wait_stream(self._main)
        loss.record_stream(cur)
        return loss

    def _step(self, batch: Dict[str, torch.Tensor], total_weight: float = None, accum_grad: int = 1,
              accum_index: int = 0) -> torch.Tensor:
        """One iteration of the trainer loop body.

        total_weight: sum over ALL ranks of the batch weights (= utterance counts, AbsESPnetModel contract) of this
        global mini-batch.  Every rank iterates the same global batch list and takes ``batch[rank::world]``
        (abs_task.py:1504-1513), so the total is known on the host without a collective; the loss of this rank is
        scaled by w_r / total * world before backward and the summed gradient divided by world afterwards
        (trainer.py:583-595 + DDP averaging).  None = equal weights on every rank (scale 1).
        accum_grad / accum_index: gradient accumulation (trainer.py:597, 610-679): micro-step 0 clears the gradient
        buffer, every micro-step adds loss/accum_grad gradients, only the last one all-reduces and updates."""
        from . import ops
        st = self.store
        if accum_index == 0:
            st.zero_grad()
        if self.engine.bf16:           # 16-byte DMA granules: extend the batch padding to T_mel, T % 8 == 0 (no-op if aligned)
            from .espnet_model import ESPnetMLMEncAsDecoderModel
            batch = ESPnetMLMEncAsDecoderModel._pad_to_dma_granule(batch)
        w = 1.0
        if total_weight is not None:
            w = grad_scale(float(batch["speech"].shape[0]), float(total_weight), self.world)
        out = self.engine.forward(batch, gscale=w / accum_grad)
        last = (accum_index == accum_grad - 1)
        if self.reducer is not None and last:
            self._backward_overlapped()
        else:
            self.engine.backward()
            if self.world > 1 and last:
                dist.all_reduce(st.grad)
        if not last:
            return out["loss"]
        self.step_no += 1
        ops.sumsq(st.grad, self.partial)
        ops.clip_adam_noam(st.flat, st.grad, self.m, self.v, self.partial, self.norm, self.opt_state, self.lr,
                           self.cfg.adim, self.warmup, clip=self.clip, gscale=1.0 / self.world, betas=self.betas,
                           eps=self.eps)
        return out["loss"]

    def optimizer_steps(self):
        """(updates applied, steps skipped because the gradient norm was not finite) -- reads the device clock (syncs)."""
        a, sk = self.opt_state.tolist()
        if sk:
            import logging
            logging.warning(f"{sk} step(s) had a non-finite gradient norm and were skipped (trainer.py:640-645)")
        return a, sk

    def _backward_overlapped(self):
        """Backward with per-range all-reduce hooks: a range is reduced once the schedule has moved
        below its lowest parameter offset (ranges are ordered decoder -> encoder -> embedding)."""
        eng, red = self.engine, self.reducer
        state = {"next": 0}
        lows = [lo for lo, _ in red.ranges]
        offs = self.store.offsets

        def after(name):          # called after the backward of the parameter group starting at `name`
            lo = offs[name][0]
            if state["next"] < len(lows) and lows[state["next"]] >= lo and hasattr(eng, "join_side"):
                eng.join_side()       # the weight gradients of the range come from the engine's side streams
            while state["next"] < len(lows) and lows[state["next"]] >= lo:
                red.reduce_range(state["next"])
                state["next"] += 1

        eng.backward(on_group_done=after)
        while state["next"] < len(lows):
            red.reduce_range(state["next"])
            state["next"] += 1
        red.wait()

    # ---- checkpoint (trainer.py:366-388: checkpoint.pth = {model, optimizers, schedulers, ...}) ----
    def state(self):
        applied = int(self.opt_state[0])
        return dict(model=self.store.state_dict(), optimizers=[dict(m=self.m.cpu(), v=self.v.cpu(), step=applied)],
                    schedulers=[dict(step=applied, warmup_steps=self.warmup)], reporter=None, scaler=None)

    def load_state(self, st):
        self.store.load_state_dict(st["model"])
        o = st["optimizers"][0]
        self.m.copy_(o["m"])
        self.v.copy_(o["v"])
        self.step_no = int(o["step"])
        self.opt_state[0] = self.step_no
        self.opt_state[1] = 0
