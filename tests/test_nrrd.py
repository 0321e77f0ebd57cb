import os

import numpy as np

from tests.conftest import GOLDEN
from totalsegmentator2d_amd import nrrd


def test_read_reference_samples():
    a = os.path.join(GOLDEN, 'assets')
    s = nrrd.read(os.path.join(a, 'sample_s0616.nrrd'))
    assert s.size == (337, 644) and s.components == 2 and s.array.dtype == np.float64 and s.array.shape == (644, 337, 2)
    assert abs(s.spacing[0] - 1.499) < 1e-3 and -1100 < s.array[..., 0].min() < -1000 and s.array[..., 1].max() > 3000
    v = nrrd.read(os.path.join(a, 'sample_s0521.nrrd'))
    assert v.size == (53, 120, 133) and v.array.dtype == np.int16 and v.space == 'left-posterior-superior'
    assert v.direction == (-1.0, 0.0, 0.0, 0.0, -1.0, 0.0, 0.0, 0.0, 1.0)
    x = nrrd.read(os.path.join(a, 'sample_chexpert.nrrd'))
    assert x.size == (320, 320) and x.array.dtype == np.uint8 and x.spacing == (1.25, 1.25)
    p = nrrd.read(os.path.join(a, 'sample_s0332.nrrd'))
    assert p.size == (269, 1, 505) and p.components == 2 and p.meta.get('ITK_InputFilterName') == 'NrrdImageIO'


def test_write_read_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    seg = (rng.random((40, 64, 5)) > 0.5).astype(np.uint8)
    img = nrrd.Image(seg, (1.5, 1.5), (10.0, -3.25), (1.0, 0.0, 0.0, 1.0), components=5,
                     meta={'Segment0_Name': 'heart', 'Segment0_Layer': '0', 'Segment0_LabelValue': '1'})
    for compress in (True, False):
        fp = str(tmp_path / f'x{int(compress)}.seg.nrrd')
        nrrd.write(img, fp, compress)
        back = nrrd.read(fp)
        assert np.array_equal(back.array, seg) and back.components == 5 and back.spacing == (1.5, 1.5)
        assert back.origin == (10.0, -3.25) and back.meta['Segment0_Name'] == 'heart'
        head = open(fp, 'rb').read(400).decode('latin1')
        assert 'kinds: vector domain domain' in head and 'sizes: 5 64 40' in head and 'type: unsigned char' in head
    vol = rng.normal(size=(3, 4, 5)).astype(np.float32)
    v = nrrd.Image(vol, (1.0, 2.0, 3.0), (0.0, 0.0, 0.0), (1, 0, 0, 0, 1, 0, 0, 0, 1), space='left-posterior-superior')
    nrrd.write(v, str(tmp_path / 'v.nrrd'))
    b = nrrd.read(str(tmp_path / 'v.nrrd'))
    assert np.array_equal(b.array, vol) and b.spacing == (1.0, 2.0, 3.0) and b.space == 'left-posterior-superior'
