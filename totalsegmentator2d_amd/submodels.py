"""The full ts2d-v2 model set on one GPU (BASELINE config 3: "5 sub-models, 117 labels"): one engine per sub-model, all driven
on the SAME batch of slices, the per-model masks joined into one packed 117-channel mask.

The reference drives its sub-models one after the other (``ts2d/tool.py:110-112``) and merges the label channels in the order
"sorted sub-model id, then label" (``ts2d/core/util/image.py:490-510``, ``tool.py:114-122``); the same order is used here.
Head counts are those of TotalSegmentator v2's five tasks (cardiac 18, muscles 23, organs 24, ribs 26,
vertebrae 26 = 117 labels, reference ``README.md:16``).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

from .arch import UNetArch
from .engine import Engine

TS2D_V2_HEADS: Dict[str, int] = {'cardiac': 18, 'muscles': 23, 'organs': 24, 'ribs': 26, 'vertebrae': 26}


class SubModelSet:
    def __init__(self, models: Sequence[Tuple[str, UNetArch, object]], device: int = 0, precision: str = 'f16'):
        """models: (id, arch, weight blob) triples; engines are created in sorted-id order."""
        self.ids: List[str] = []
        self.engines: List[Engine] = []
        for mid, arch, blob in sorted(models, key=lambda m: m[0]):
            e = Engine(arch, blob, device=device)
            e.set_precision(precision)
            self.ids.append(mid)
            self.engines.append(e)
        self.channels = [e.arch.num_classes for e in self.engines]
        self._ws = None              # ONE activation workspace for all engines (they run one after the other on one stream)

    @property
    def num_labels(self) -> int:
        return sum(self.channels)

    def channel_range(self, mid: str) -> Tuple[int, int]:
        i = self.ids.index(mid)
        lo = sum(self.channels[:i])
        return lo, lo + self.channels[i]

    def set_precision(self, mode: str):
        """Arithmetic mode of every sub-model; the shared workspace is re-sized at the next forward (a mode's activation plan differs)."""
        for e in self.engines:
            e.set_precision(mode)
        self._reserved = None

    def reserve(self, B: int, H: int, W: int):
        """One workspace of the largest engine's size, shared by all sub-models (the reference drives them sequentially,
        ``ts2d/tool.py:110-112``; here they follow each other on one stream): 1x instead of 5x the activation memory.
        Sized under the engines' CURRENT precision mode / options / keep flag; grown when any of them needs more."""
        import torch
        need = max(e.workspace_bytes(B, H, W) for e in self.engines)
        if self._ws is None or self._ws.numel() < need + 256:
            for e in self.engines:
                e.set_workspace(None)
            self._ws = None
            self._ws_set = None
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=torch.device('cuda', self.engines[0].device))
        base = (self._ws.data_ptr() + 255) // 256 * 256
        size = self._ws.numel() - (base - self._ws.data_ptr())
        if getattr(self, '_ws_set', None) != (base, size):          # (set_workspace synchronises twice per engine and invalidates its plan: only on change)
            for e in self.engines:
                e.set_workspace(base, size)
            self._ws_set = (base, size)
        for e in self.engines:
            e.reserve(B, H, W)
        self._reserved = (B, H, W)

    def forward_masks(self, x, out_masks: Optional[list] = None, stream: int = 0) -> list:
        """x: torch CUDA [B,C,H,W].  Runs every sub-model on the batch; returns the per-model packed masks
        [B, K_i, H, W/32] (int32) in sorted-id order (pass `out_masks` to reuse buffers)."""
        import torch
        B, _, H, W = x.shape
        r = getattr(self, '_reserved', None)
        # (a member engine's mode / options / keep flag may have changed since the last reserve: its need is re-read before every run)
        if r is None or r[0] < B or r[1:] != (H, W) or max(e.workspace_bytes(B, H, W) for e in self.engines) + 256 > self._ws.numel():
            self.reserve(B, H, W)
        if out_masks is None:
            out_masks = [torch.empty((B, k, H, W // 32), dtype=torch.int32, device=x.device) for k in self.channels]
        for e, m in zip(self.engines, out_masks):
            e.forward(x, logits=False, mask=True, out_mask=m, stream=stream)
        return out_masks

    @staticmethod
    def merge(masks: list):
        """Per-model masks -> one [B, sum K_i, H, W/32] packed mask, channels in sorted-id then label order."""
        import torch
        return torch.cat(masks, dim=1)

    def close(self):
        for e in self.engines:
            e.close()
        self.engines = []
        self._ws = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
