"""Local model zoo: key resolution and loading (reference ``ts2d/core/inference/zoo.py`` + ``database.py``, local part only).

Model keys are ``<model>[_<group>][:r###]``; aliases come from the packaged ``default-resolve`` map (reference
``ts2d/data/config.json:6-10``); models live in ``~/.ts2d/models/<model>_<group>/r###/`` with a ``model.json``
(reference ``ts2d/core/util/path.py:12-16``, ``database.py:127-140``).  The remote (Zenodo/gdown) database is out of
scope - there is no network on either box; ``use_remote=True`` is accepted and ignored with a log line.
"""
from __future__ import annotations

import json
import os
import re
from typing import Dict, List, Optional

DEFAULT_MODEL = 'ts2d-v2-ep4000b2'
DEFAULT_RESOLVE = {'ts2d': 'ts2d-v2', 'ts2d-v2': 'ts2d-v2-ep4000b2', 'ts2d-v1': 'ts2d-v1-ep4000b2'}


_LABEL_COLORS = None


def get_label_colors() -> Dict[str, str]:
    """label name (lower case) -> '#RRGGBB' from the packaged ``data/label-colors.csv`` (the reference's own data file, read
    by ``ts2d/core/util/config.py:13-20`` and passed to every model as ``nnu.result.colors``, ``ts2d/tool.py:29-33``)."""
    global _LABEL_COLORS
    if _LABEL_COLORS is None:
        import csv
        fp = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'label-colors.csv')
        with open(fp, newline='') as f:
            _LABEL_COLORS = {row['Label'].strip().lower(): row['Color'].strip() for row in csv.DictReader(f)
                             if row.get('Label') and row.get('Color')}
    return _LABEL_COLORS


def models_root() -> str:
    return os.environ.get('TS2D_MODELS', os.path.join(os.path.expanduser('~'), '.ts2d', 'models'))


def decompose_model_key(key: str):
    """'ts2d-v2-ep4000b2_cardiac' -> ('ts2d-v2-ep4000b2', 'cardiac') (reference database.py:17-22)."""
    key = key.split(':')[0]
    model, _, group = key.partition('_')
    return model, (group or None)


class LocalZoo:
    def __init__(self, root: Optional[str] = None):
        self.root = root or models_root()

    def list(self) -> List[str]:
        if not os.path.isdir(self.root):
            return []
        return sorted(d for d in os.listdir(self.root) if os.path.isdir(os.path.join(self.root, d)))

    def resolve(self, key: str, unique_model: bool = True) -> List[str]:
        """alias map loop, then prefix match on '-'-separated model strings (reference zoo.py:54-77, database.py:87-98)."""
        seen = set()
        while key in DEFAULT_RESOLVE and key not in seen:
            seen.add(key)
            key = DEFAULT_RESOLVE[key]
        model, group = decompose_model_key(key)
        ids = []
        for d in self.list():
            m, g = decompose_model_key(d)
            if (m == model or m.startswith(model + '-')) and (group is None or g == group):
                ids.append(d)
        if unique_model and ids:
            models = sorted({decompose_model_key(i)[0] for i in ids})
            ids = [i for i in ids if decompose_model_key(i)[0] == models[-1]]
        return sorted(ids)

    def latest_revision(self, mid: str) -> Optional[str]:
        p = os.path.join(self.root, mid)
        revs = sorted(d for d in os.listdir(p) if re.match(r'r\d+$', d)) if os.path.isdir(p) else []
        return revs[-1] if revs else None

    def load_config(self, mid: str, param: Optional[dict] = None) -> dict:
        rev = self.latest_revision(mid)
        if rev is None:
            raise RuntimeError(f"model {mid} has no revision directory under {self.root}")
        root = os.path.join(self.root, mid, rev)
        with open(os.path.join(root, 'model.json')) as f:
            cfg = json.load(f)
        if 'param' not in cfg:
            raise RuntimeError(f"model.json of {mid} has no 'param' entry")          # reference zoo.py:161
        cfg['root'] = root
        cfg.setdefault('model', mid)
        cfg.setdefault('revision', int(rev[1:]))
        cfg['param'] = {**cfg['param'], **(param or {})}
        return cfg
