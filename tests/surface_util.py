"""Synthetic-weight sub-models for the TS2D surface tests (the Zenodo models are a run-time download, not available)."""
import numpy as np

from tests import cases
from totalsegmentator2d_amd import weights
from totalsegmentator2d_amd.model import HIPModel
from tests.host_predictor import HostLogicPredictor


class HostModel(HIPModel):
    """The model surface on a machine without a GPU: its predictor is the host restatement fed by the torch oracle."""
    def _make_predictor(self, kw):
        return HostLogicPredictor(network=self._config['oracle_network'], **kw)


def synthetic_model(mid, K, seed, channels=('mean', 'max'), patch=(64, 64), network=None, mirror=True, feats=(32, 32, 64)):
    arch = cases.unet(len(feats), feats, K, cin=len(channels))
    sd = weights.synthetic_state_dict(arch, seed)
    blob = weights.pack_blob(arch, sd)
    ds = {'channel_names': {str(i): c for i, c in enumerate(channels)},
          'labels': {'background': 0, **{f'{mid.split("_")[-1]}_{i + 1}': i + 1 for i in range(K)}},
          'file_ending': '.nrrd', 'multilabel': True}
    cfg = {'model': mid, 'revision': 1, 'param': {'nnu.predict.augment': mirror, 'nnu.result.colors': {n: (i * 20 % 255, 80, 200) for i, n in enumerate(ds['labels'])}},
           'synthetic': {'arch': arch, 'blobs': [blob], 'patch_size': patch, 'dataset_json': ds}}
    if network is not None:
        from oracle import torch_oracle as O
        cfg['oracle_network'] = lambda batch, fold=0: np.concatenate([O.unet_forward(arch, sd, batch[i:i + 1]).numpy() for i in range(batch.shape[0])])
        return HostModel(cfg), arch, sd
    return HIPModel(cfg), arch, sd
