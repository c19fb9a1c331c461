"""MI355X mirror of ``vipformer.model.pointcloud.classifier.PointCloudInputAdapter``
(reference classifier.py:25-50).  Only the adapter is on the scripted path; the
Perceiver-IO ``PointCloudClassifier`` (dead code in the reference, classifier.py:55-57
passes a kwarg the adapter does not accept) is out of scope."""
from __future__ import annotations

from typing import Tuple

import torch.nn as nn

from ... import ops


class InputAdapter(nn.Module):
    """partseg.py:216-230 / core/modules.py: carries ``num_input_channels``."""

    def __init__(self, num_input_channels: int):
        super().__init__()
        self._num_input_channels = num_input_channels

    @property
    def num_input_channels(self):
        return self._num_input_channels

    def forward(self, x):
        raise NotImplementedError()


class PointCloudInputAdapter(InputAdapter):
    """Per-point MLP Linear(C,64) -> LayerNorm(64) -> ReLU -> Linear(64,D)  (classifier.py:31-36).

    forward([B,N,C]) -> [B,N,D].  The result is the key/value source of the encoder's
    cross-attention and nothing else, so it is produced in h16 (MFMA operand precision)."""

    def __init__(self, pointcloud_shape: Tuple[int, ...], num_input_channels: int):
        super().__init__(num_input_channels=num_input_channels)
        _, self.point_channels = pointcloud_shape
        self.point_mlp = nn.Sequential(nn.Linear(self.point_channels, 64), nn.LayerNorm(64), nn.ReLU(),
                                       nn.Linear(64, num_input_channels))

    def forward(self, x):
        return ops.AdapterFn.apply(x, self, *self.parameters())
