"""TS2D.predict() on the MI355X engine vs the same surface driven by the torch oracle (network hook)."""
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN
from tests.surface_util import synthetic_model
from totalsegmentator2d_amd import nrrd
from totalsegmentator2d_amd.tool import TS2D

pytestmark = pytest.mark.gpu
A = os.path.join(GOLDEN, 'assets')


@pytest.mark.parametrize('asset,collapse', [('sample_s0521.nrrd', False), ('sample_s0616.nrrd', False), ('sample_s0332.nrrd', True)])
def test_predict_matches_oracle_pipeline(asset, collapse, tmp_path):
    ids = ('ts2d-v2-ep4000b2_cardiac', 'ts2d-v2-ep4000b2_ribs')
    gpu = {m: synthetic_model(m, 3 + i, 31 + i)[0] for i, m in enumerate(ids)}                       # HIP engines, mirroring on
    ref = {m: synthetic_model(m, 3 + i, 31 + i, network=True)[0] for i, m in enumerate(ids)}        # torch oracle underneath
    with TS2D(models=gpu) as ts, TS2D(models=ref) as tr:
        a = ts.predict(os.path.join(A, asset), collapse=collapse)
        b = tr.predict(os.path.join(A, asset), collapse=collapse)
        sa, sb = a.get_segmentation(), b.get_segmentation()
        assert sa.components == sb.components == 7 and sa.size == sb.size and sa.array.dtype == np.uint8
        assert (sa.array != sb.array).mean() < 2e-3              # only pixels whose fp16 end-of-pipeline logit sits at ~0 may differ
        assert sa.meta == sb.meta
        for m in ids:
            t = a.data['models'][m]['timestamps']
            assert t['start'] <= t['preprocessed'] <= t['predicted'] <= t['exported']
        files = a.save(dest=str(tmp_path), name='case', models='all', targets=['segmentation', 'projection'], content='file')
        assert os.path.exists(os.path.join(str(tmp_path), 'case.seg.nrrd')) and len(files) >= 3
        assert nrrd.read(os.path.join(str(tmp_path), 'case.seg.nrrd')).components == 7
