"""TS2D.predict() / Result.save() / ts2d CLI surface and the image utilities either side of the hot path, on CPU
(the network is injected through tests/surface_util.py's subclass of the model; tests/test_gpu_surface.py runs the same surface on the engine).
Counterparts of the reference's test_020/021/022/030 (which only assert types and file names)."""
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN
from tests.surface_util import synthetic_model
from totalsegmentator2d_amd import image, nrrd
from totalsegmentator2d_amd.main import ts2d_run, _enumerate_cases
from totalsegmentator2d_amd.tool import TS2D
from totalsegmentator2d_amd.zoo import LocalZoo, decompose_model_key

A = os.path.join(GOLDEN, 'assets')


def test_reorient_keeps_physical_positions():
    v = nrrd.read(os.path.join(A, 'sample_s0521.nrrd'))                 # direction diag(-1,-1,1) (LPS space)
    r = image.reorient_image(v, 'RAI')
    assert r.direction == (1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0) and r.size == v.size

    def phys(img, idx):
        D = np.asarray(img.direction).reshape(3, 3)
        return np.asarray(img.origin) + D @ (np.asarray(idx) * np.asarray(img.spacing))
    i, j, k = 5, 17, 100
    ri, rj = v.size[0] - 1 - i, v.size[1] - 1 - j                        # x and y are flipped
    assert v.array[k, j, i] == r.array[k, rj, ri]
    assert np.allclose(phys(v, (i, j, k)), phys(r, (ri, rj, k)))


def test_projection_and_dimension_round_trip():
    v = image.reorient_image(nrrd.read(os.path.join(A, 'sample_s0521.nrrd')))
    mx = image.project(v, 'max', 'coronal')
    mn = image.cast(image.project(v, 'mean', 'coronal'), np.float32)
    assert mx.size == (53, 1, 133) and mn.array.dtype == np.float32
    assert np.array_equal(mx.array[:, 0, :], v.array.max(axis=1))
    from oracle import input_oracle as IO                              # the checker: real-valued mean (double sum / n, then Float32)
    assert np.array_equal(mn.array[:, 0, :], IO.project_f32(v.array, 'mean', 1))
    assert np.mean(mn.array != np.round(mn.array)) > 0.9              # NOT truncated back to int16
    two = image.reduce_dimensions(image.compose([mn, image.cast(mx, np.float32)]))
    assert two.dimension == 2 and two.size == (53, 133) and two.components == 2 and two.spacing == (1.5, 1.5)
    back = image.restore_dimension(two, mx)
    assert back.size == (53, 1, 133) and back.components == 2
    pre = image.reduce_dimensions(nrrd.read(os.path.join(A, 'sample_s0332.nrrd')))     # pre-projected sample of test_020
    assert pre.size == (269, 505) and pre.components == 2


def test_zoo_key_resolution(tmp_path):
    for d in ('ts2d-v2-ep4000b2_cardiac', 'ts2d-v2-ep4000b2_ribs', 'ts2d-v1-ep4000b2_bones', 'tsxr-v2-ep1000b2_ribs'):
        os.makedirs(tmp_path / d / 'r001')
    z = LocalZoo(str(tmp_path))
    assert z.resolve('ts2d') == ['ts2d-v2-ep4000b2_cardiac', 'ts2d-v2-ep4000b2_ribs']          # ts2d -> ts2d-v2 -> ts2d-v2-ep4000b2
    assert z.resolve('ts2d-v2-ep4000b2_cardiac') == ['ts2d-v2-ep4000b2_cardiac']
    assert z.resolve('tsxr-v2-ep1000b2_ribs') == ['tsxr-v2-ep1000b2_ribs'] and z.resolve('nope') == []
    assert decompose_model_key('ts2d-v2-ep4000b2_cardiac:r001') == ('ts2d-v2-ep4000b2', 'cardiac')
    with pytest.raises(RuntimeError, match='No models were resolved'):
        TS2D(key='nope', zoo_root=str(tmp_path))


@pytest.fixture(scope='module')
def two_models():
    m1, _, _ = synthetic_model('ts2d-v2-ep4000b2_cardiac', 3, 31, network=True, mirror=False, feats=(32, 32))
    m2, _, _ = synthetic_model('ts2d-v2-ep4000b2_ribs', 4, 32, network=True, mirror=False, feats=(32, 32))
    return {'ts2d-v2-ep4000b2_cardiac': m1, 'ts2d-v2-ep4000b2_ribs': m2}


def test_predict_3d_volume_merge_and_save(tmp_path, two_models):
    """Counterpart of reference test_021 / test_022: 3-D CT -> projections -> two sub-models -> merged multilabel image."""
    with TS2D(models=dict(two_models)) as ts:
        res = ts.predict(os.path.join(A, 'sample_s0521.nrrd'))
        assert res.models == ['ts2d-v2-ep4000b2_cardiac', 'ts2d-v2-ep4000b2_ribs']
        seg = res.get_segmentation()
        assert isinstance(seg, nrrd.Image) and seg.components == 7 and seg.size == (53, 1, 133) and seg.array.dtype == np.uint8
        assert set(np.unique(seg.array)) <= {0, 1}
        assert seg.meta['Segment0_Name'] == 'cardiac_1' and seg.meta['Segment3_Name'] == 'ribs_1' and seg.meta['Segment6_Layer'] == '6'
        part = res.get_segmentation('ts2d-v2-ep4000b2_ribs')
        assert part.components == 4 and np.array_equal(part.array[..., 0], seg.array[..., 3])       # merge order: sub-model, then label
        assert sorted(res.get_projection()) == ['max', 'mean']
        files = res.save(dest=str(tmp_path), name='test', models='all', targets='all', content='file')
        names = sorted(os.path.basename(f) for f in files)
        assert names == ['test-cardiac.nrrd', 'test-cardiac.seg.nrrd', 'test-ribs.nrrd', 'test-ribs.seg.nrrd', 'test.nrrd',
                         'test.seg.nrrd', 'test_max.nrrd', 'test_mean.nrrd']
        back = nrrd.read(str(tmp_path / 'test.seg.nrrd'))
        assert back.components == 7 and np.array_equal(back.array, seg.array) and back.meta['Segment1_LabelValue'] == '1'
        collapsed = ts.predict(os.path.join(A, 'sample_s0521.nrrd'), collapse=True).get_segmentation()
        assert collapsed.dimension == 2 and collapsed.size == (53, 133)


def test_predict_preprojected_2d_and_channel_mismatch(two_models):
    """Counterpart of reference test_020 (2-D pre-projected input); a 1-channel image must be rejected (tool.py:161-167)."""
    with TS2D(models={'ts2d-v2-ep4000b2_cardiac': two_models['ts2d-v2-ep4000b2_cardiac']}) as ts:
        res = ts.predict(os.path.join(A, 'sample_s0616.nrrd'))
        seg = res.get_segmentation()
        assert seg.dimension == 2 and seg.size == (337, 644) and seg.components == 3
        with pytest.raises(RuntimeError, match='number of channels'):
            ts.predict(os.path.join(A, 'sample_chexpert.nrrd'))


def test_cli_file_names(tmp_path, two_models):
    """Counterpart of reference test_030: expected output names for single-file and folder input."""
    src = tmp_path / 'in'
    os.makedirs(src)
    import shutil
    shutil.copy(os.path.join(A, 'sample_s0521.nrrd'), src / 'sample_s0521.nrrd')
    (src / 'notes.txt').write_text('skip me')
    assert [n for n, _ in _enumerate_cases(str(src))] == ['sample_s0521']
    with pytest.raises(ValueError):
        list(_enumerate_cases(str(src / 'notes.txt')))
    ts2d_run(str(src), str(tmp_path / 'out'), models=dict(two_models), visualize=False, save_all=True, silent=True)
    out = sorted(os.listdir(tmp_path / 'out'))
    assert out == ['sample_s0521-cardiac.seg.nrrd', 'sample_s0521-ribs.seg.nrrd', 'sample_s0521.seg.nrrd',
                   'sample_s0521_max.nrrd', 'sample_s0521_mean.nrrd']


def test_label_colours_are_stamped_like_the_reference(tmp_path):
    """reference ts2d/tool.py:29-33 passes the packaged label-colors.csv as 'nnu.result.colors'; meta.py:219-231 stamps
    Segment*_Color as three 0..1 floats with three decimals."""
    from totalsegmentator2d_amd.zoo import get_label_colors
    colors = get_label_colors()
    assert len(colors) >= 117 and colors['autochthon-left'] == '#9370DB'
    assert image.to_color_str_rgb_floats('#9370DB') == '0.576 0.439 0.859'
    assert image.to_color_str_rgb_floats((255, 0, 128)) == '1.000 0.000 0.502'
    seg = nrrd.Image(np.zeros((4, 4, 2), np.uint8), (1.0, 1.0), (0.0, 0.0), (1.0, 0.0, 0.0, 1.0), components=2)
    image.set_annotation_meta(seg, {1: 'autochthon-left', 2: 'no-such-label'}, colors)
    assert seg.meta['Segment0_Color'] == '0.576 0.439 0.859' and seg.meta['Segment0_ColorAutoGenerated'] == '0'
    assert 'Segment1_Color' not in seg.meta and seg.meta['Segment1_Name'] == 'no-such-label'


def test_stage_named_errors(two_models):
    """reference prediction_worker.py:183-242: '<Stage> failed for <name>: <cause>', wrapped by nnu.py:217-219."""
    m = two_models['ts2d-v2-ep4000b2_cardiac']
    m.start()
    try:
        bad = nrrd.Image(np.zeros((8, 8, 3), np.float32), (1.5, 1.5), (0.0, 0.0), (1.0, 0.0, 0.0, 1.0), components=3)
        with pytest.raises(RuntimeError, match=r'Prediction failed for: image1: (Preprocessing|Prediction) failed for image1'):
            m.apply(bad)
    finally:
        m.stop()


def test_case_enumeration(tmp_path):
    for fn in ('a.nrrd', 'b.seg.nrrd', 'c.nii.gz', 'd.txt', 'noext'):
        (tmp_path / fn).write_bytes(b'x')
    assert [n for n, _ in _enumerate_cases(str(tmp_path))] == ['a']            # 'b.seg.nrrd' has extension 'seg.nrrd'
    with pytest.raises(ValueError, match='Unsupported file extension'):
        _enumerate_cases(str(tmp_path / 'd.txt'))
    with pytest.raises(ValueError, match='only NRRD'):
        _enumerate_cases(str(tmp_path / 'c.nii.gz'))
    with pytest.raises(FileNotFoundError):
        _enumerate_cases(str(tmp_path / 'missing.nrrd'))
    with pytest.raises(ValueError, match='does not have an extension'):
        _enumerate_cases(str(tmp_path / 'noext'))


def test_combined_segmentation_equals_the_composed_label_masks(tmp_path):
    """``combine_segmentations`` (reference ts2d/core/util/image.py:490-510: per label of every sub-model the ``> 0`` mask, composed
    into one vector image) writes its masks plane by plane and returns the interleaved VIEW; the values, the label order, the
    metadata and what ``nrrd.write`` serialises must be those of the composed image - also for a label-map member and for a
    member whose labels are listed out of channel order."""
    from totalsegmentator2d_amd.export import segmentation_to_image
    rng = np.random.default_rng(3)
    ref = nrrd.Image(np.zeros((20, 12)), (1.5, 2.0), (3.0, 4.0), (1.0, 0.0, 0.0, 1.0), 1, {}, None)
    a = segmentation_to_image((rng.random((3, 1, 20, 12)) > 0.5).astype(np.uint8) * 7, ref, True, {1: 'a1', 2: 'a2', 3: 'a3'}, None)
    b = segmentation_to_image((rng.random((4, 1, 20, 12)) > 0.5).astype(np.uint8), ref, True, {1: 'b1', 2: 'b2', 3: 'b3', 4: 'b4'}, None)
    assert not a.array.flags['C_CONTIGUOUS'] and a.array.shape == (20, 12, 3)
    # labels of b out of channel order: Segment0 -> layer 2, Segment1 -> layer 0, ...
    for i, layer in enumerate((2, 0, 3, 1)):
        b.meta[f'Segment{i}_Layer'] = str(layer)
    c = nrrd.Image(rng.integers(0, 3, (20, 12)).astype(np.uint8), ref.spacing, ref.origin, ref.direction, 1, {}, None)
    image.set_annotation_meta(c, {1: 'c1', 2: 'c2'}, None)
    got = image.combine_segmentations([a, b, c])
    want, names = [], []
    for seg in (a, b, c):
        for name, info in image.get_annotation_labels(seg).items():
            want.append(image.get_label_mask(seg, info['value']).array)
            names.append(name)
    want = np.stack(want, axis=-1)
    assert got.components == 9 and got.array.shape == want.shape and got.array.dtype == np.uint8
    assert np.array_equal(got.array, want)
    assert list(image.get_annotation_labels(got)) == names
    path = str(tmp_path / 'combined.nrrd')
    nrrd.write(got, path, True)
    back = nrrd.read(path)
    assert back.components == 9 and np.array_equal(back.array, want) and back.spacing == got.spacing
    one = image.combine_segmentations([nrrd.Image(c.array, c.spacing, c.origin, c.direction, 1,
                                                  {k: v for k, v in c.meta.items() if not k.startswith('Segment1_')}, None)])
    assert one.components == 1 and np.array_equal(one.array, (c.array == 1).astype(np.uint8))
    # the plane-major view vs what sitk.GetArrayFromImage hands out: contiguous() is the documented way to the interleaved bytes
    cg = got.contiguous()
    assert cg.array.flags['C_CONTIGUOUS'] and np.array_equal(cg.array, want) and cg.array.tobytes() == np.ascontiguousarray(want).tobytes()
    assert cg.contiguous() is cg and cg.meta == got.meta
    # no labelled member at all: an error, as sitk.Compose([]) is in the reference (ts2d/core/util/image.py:508)
    with pytest.raises(ValueError, match='carries a label'):
        image.combine_segmentations([nrrd.Image(c.array, c.spacing, c.origin, c.direction, 1, {}, None)])
