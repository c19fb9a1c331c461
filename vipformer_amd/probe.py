"""Per-epoch linear-SVM probe and best-checkpoint files of the pre-training loop (pretrain.py:226-290).

  * ``extract_features`` -- the eval-mode feature pass of pretrain.py:228-249,256-268: ``pc_model(data)[1]`` (the [B, 2D] backbone
    features) batch by batch; the kernels are the hot path's own (eval mode: BatchNorm running statistics, dropout off), the result
    is a host array for sklearn exactly like the reference's ``feats.tolist()``.
  * ``svm_probe`` -- sklearn ``SVC(C, kernel='linear')`` fit / score (pretrain.py:251-276): CPU work in the reference, CPU work here.
  * ``save_best`` / ``load_pretrained`` -- the checkpoint FILES: ``torch.save(module.state_dict())`` with the reference's key set
    (pretrain.py:283-287), and the fine-tuning scripts' loading convention (``"module." + key``, strict=False; ft_cls.py:92-98,
    ft_partseg.py:80-83).  Files written here load into the reference's modules and vice versa (tests/test_host_cpu.py checks a
    reference-written file).
"""
from __future__ import annotations

import os
from typing import Iterable, Tuple

import numpy as np
import torch


@torch.no_grad()
def extract_features(pc_model: torch.nn.Module, batches: Iterable[Tuple[torch.Tensor, torch.Tensor]], device=None, trainer=None):
    """pretrain.py:235-249.  batches yield (data [B,N,3], label [B] or [B,1]); returns (feats float64 [n, 2D], labels int64 [n]).
    trainer: the train.Pretrainer that owns `pc_model`, if any -- data-parallel, rank 0's BatchNorm running statistics are broadcast
    first (the reference's eval forwards go through the DDP wrapper, pretrain.py:243,266, which does that in front of every forward
    pass); EVERY rank must then make this call, as every rank runs the probe loop in the reference."""
    if trainer is not None:
        trainer.sync_buffers(0)
    was_training = pc_model.training
    pc_model.eval()
    dev = device if device is not None else next(pc_model.parameters()).device
    feats, labels = [], []
    try:
        for data, label in batches:
            f = pc_model(data.to(dev))[1]
            feats.append(np.asarray(f.float().cpu().numpy(), dtype=np.float64))
            labels.append(np.asarray(torch.as_tensor(label).reshape(len(label), -1)[:, 0].cpu().numpy(), dtype=np.int64))
    finally:
        pc_model.train(was_training)
    return np.concatenate(feats, 0), np.concatenate(labels, 0)


def svm_probe(train_feats, train_labels, test_feats, test_labels, C: float = 0.01) -> float:
    """pretrain.py:251-276: linear SVC on the training features, accuracy on the test features."""
    from sklearn.svm import SVC
    svm = SVC(C=C, kernel="linear")
    svm.fit(train_feats, train_labels)
    return float(svm.score(test_feats, test_labels))


def save_best(pc_model, img_model, directory: str) -> Tuple[str, str]:
    """pretrain.py:283-287: pc_model_best.pth / img_model_best.pth = torch.save(module.state_dict())."""
    os.makedirs(directory, exist_ok=True)
    unwrap = lambda m: m.module if hasattr(m, "module") and isinstance(m.module, torch.nn.Module) else m
    p1, p2 = os.path.join(directory, "pc_model_best.pth"), os.path.join(directory, "img_model_best.pth")
    torch.save({k: v.detach().cpu() for k, v in unwrap(pc_model).state_dict().items()}, p1)
    if img_model is not None:
        torch.save({k: v.detach().cpu() for k, v in unwrap(img_model).state_dict().items()}, p2)
    return p1, p2


def load_pretrained(model: torch.nn.Module, path: str, strict: bool = False, map_location="cpu"):
    """ft_cls.py:92-98 / ft_partseg.py:80-83: the file holds un-prefixed keys; a DistributedDataParallel wrapper gets them with
    ``"module."`` in front; strict=False drops what the fine-tuning model does not have (``latent_head.*``) and leaves its new head
    at initialisation.  Returns torch's (missing_keys, unexpected_keys)."""
    sd = torch.load(path, map_location=map_location)
    if hasattr(model, "module") and isinstance(model.module, torch.nn.Module):
        sd = {"module." + k: v for k, v in sd.items()}
    return model.load_state_dict(sd, strict=strict)
