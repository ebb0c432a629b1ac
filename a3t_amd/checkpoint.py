"""Checkpoint files of the recipe, as the reference's trainer writes and its inference driver reads them.

  save_checkpoint        espnet2/train/trainer.py:366-400  (checkpoint.pth, {N}epoch.pth, latest.pth, *.best.pth links)
  resume                 espnet2/train/trainer.py:162-190  (Trainer.resume)
  average_nbest_models   espnet2/main_funcs/average_nbest_models.py:14-112
  EpochReport            the slice of espnet2/train/reporter.py the above need (state_dict layout :575-580,
                         has :430-437, sort_epochs_and_values :372-394, get_best_epoch)

`{N}epoch.pth` holds `model.state_dict()` under the REFERENCE's parameter names and layouts (ESPnetMLMEncAsDecoderModel
translates), so files are interchangeable with the reference in both directions: `MLMTask.build_model_from_file` loads
either side's file.  The optimizer entry of checkpoint.pth is this build's own (flat fp32 Adam moments + step); it
resumes this trainer, not torch.optim.Adam.
"""
import logging
import warnings
from pathlib import Path
from typing import Collection, Dict, List, Optional, Sequence, Tuple, Union

import torch


class EpochReport:
    """Per-epoch scalar statistics: stats[epoch][phase][key] = float."""

    def __init__(self, epoch: int = 0):
        self.epoch = epoch
        self.stats: Dict[int, Dict[str, Dict[str, float]]] = {}

    def set_epoch(self, epoch: int):
        if epoch < 0:
            raise ValueError(f"epoch must be 0 or more: {epoch}")
        self.epoch = epoch

    def get_epoch(self) -> int:
        return self.epoch

    def register(self, phase: str, values: Dict[str, float], epoch: Optional[int] = None):
        e = self.epoch if epoch is None else epoch
        self.stats.setdefault(e, {}).setdefault(phase, {}).update({k: float(v) for k, v in values.items()})

    def has(self, key: str, key2: str, epoch: int = None) -> bool:
        if epoch is None:
            epoch = self.get_epoch()
        return epoch in self.stats and key in self.stats[epoch] and key2 in self.stats[epoch][key]

    def sort_epochs_and_values(self, key: str, key2: str, mode: str) -> List[Tuple[int, float]]:
        if mode not in ("min", "max"):
            raise ValueError(f"mode must min or max: {mode}")
        if not self.has(key, key2):
            raise KeyError(f"{key}.{key2} is not found")
        values = [(e, self.stats[e][key][key2]) for e in self.stats]
        return sorted(values, key=(lambda x: x[1]) if mode == "min" else (lambda x: -x[1]))   # stable, like the reference

    def get_best_epoch(self, key: str, key2: str, mode: str, nbest: int = 0) -> int:
        return self.sort_epochs_and_values(key, key2, mode)[nbest][0]

    def state_dict(self):
        return {"stats": self.stats, "epoch": self.epoch}

    def load_state_dict(self, state_dict: dict):
        self.epoch = state_dict["epoch"]
        self.stats = state_dict["stats"]


def _relink(link: Path, target_name: str):
    if link.is_symlink() or link.exists():
        link.unlink()
    link.symlink_to(target_name)


def save_checkpoint(output_dir: Union[str, Path], model, reporter: EpochReport, iepoch: int,
                    best_model_criterion: Sequence[Sequence[str]] = (("valid", "loss", "min"),), trainer=None) -> List[str]:
    """End-of-epoch files; returns the criteria this epoch improved (the "_improved" log line of the reference)."""
    output_dir = Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    torch.save({"model": model.state_dict(), "reporter": reporter.state_dict(),
                "optimizers": [trainer.state() if trainer is not None else None], "schedulers": [None], "scaler": None},
               output_dir / "checkpoint.pth")
    torch.save(model.state_dict(), output_dir / f"{iepoch}epoch.pth")
    _relink(output_dir / "latest.pth", f"{iepoch}epoch.pth")
    improved = []
    for phase, k, mode in best_model_criterion:
        if reporter.has(phase, k) and reporter.get_best_epoch(phase, k, mode) == iepoch:
            _relink(output_dir / f"{phase}.{k}.best.pth", f"{iepoch}epoch.pth")
            improved.append(f"{phase}.{k}")
    return improved


def resume(checkpoint: Union[str, Path], model, reporter: EpochReport, trainer=None, map_location="cpu"):
    states = torch.load(checkpoint, map_location=map_location)
    model.load_state_dict(states["model"])
    reporter.load_state_dict(states["reporter"])
    opt = states.get("optimizers", [None])[0]
    if trainer is not None and opt is not None:
        trainer.load_state(opt)
    logging.info(f"The training was resumed using {checkpoint}")


@torch.no_grad()
def average_nbest_models(output_dir: Union[str, Path], reporter: EpochReport,
                         best_model_criterion: Sequence[Sequence[str]], nbest: Union[Collection[int], int]) -> None:
    """`{phase}.{criterion}.ave_{n}best.pth` = element-wise mean of the n best epochs' state dicts (integer entries,
    i.e. BatchNorm.num_batches_tracked, are summed, not averaged); n = 1 is a link to the best epoch's file;
    `{phase}.{criterion}.ave.pth` links to the largest n.  Every epoch file is read once and never modified in memory
    (the reference accumulates INTO its cached copy of the best epoch, so a second criterion that reuses that epoch
    averages a polluted tensor; with one criterion -- the recipe's case -- the results are identical)."""
    output_dir = Path(output_dir)
    nbests = [nbest] if isinstance(nbest, int) else list(nbest)
    if len(nbests) == 0:
        warnings.warn("At least 1 nbest values are required")
        nbests = [1]
    ranked = [(ph, k, reporter.sort_epochs_and_values(ph, k, m)[:max(nbests)])
              for ph, k, m in best_model_criterion if reporter.has(ph, k)]
    loaded: Dict[int, Dict[str, torch.Tensor]] = {}
    for ph, cr, epoch_and_values in ranked:
        ns = [i for i in nbests if i <= len(epoch_and_values)] or [1]
        for n in ns:
            if n == 0:
                continue
            if n == 1:
                e, _ = epoch_and_values[0]
                _relink(output_dir / f"{ph}.{cr}.ave_1best.pth", f"{e}epoch.pth")
                continue
            op = output_dir / f"{ph}.{cr}.ave_{n}best.pth"
            logging.info(f'Averaging {n}best models: criterion="{ph}.{cr}": {op}')
            avg = None
            for e, _ in epoch_and_values[:n]:
                if e not in loaded:
                    loaded[e] = torch.load(output_dir / f"{e}epoch.pth", map_location="cpu")
                st = loaded[e]
                if avg is None:
                    avg = {k: v.clone() for k, v in st.items()}
                else:
                    for k in avg:
                        avg[k] = avg[k] + st[k]       # same left-to-right order as the reference
            for k in avg:
                if not str(avg[k].dtype).startswith("torch.int"):
                    avg[k] = avg[k] / n
            torch.save(avg, op)
        _relink(output_dir / f"{ph}.{cr}.ave.pth", f"{ph}.{cr}.ave_{max(ns)}best.pth")
