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
    mirror = asset == 'sample_s0521.nrrd'                    # mirroring TTA on the small volume only (keeps the oracle side short)
    patch = (64, 64) if mirror else (128, 128)
    gpu = {m: synthetic_model(m, 3 + i, 31 + i, patch=patch, mirror=mirror)[0] for i, m in enumerate(ids)}                # HIP engines
    ref = {m: synthetic_model(m, 3 + i, 31 + i, patch=patch, mirror=mirror, network=True)[0] for i, m in enumerate(ids)}  # torch oracle
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


def test_concurrent_sub_models_equal_the_serial_order():
    """TS2D.predict drives the sub-models of a case concurrently (one host thread and one HIP stream per engine; the reference drives
    them one after the other, ts2d/tool.py:110-112): the merged segmentation, every per-model segmentation and the metadata must be
    identical to the serial order, on the 3-D sample (GPU projection + device z-score shared by the sub-models) and on a 2-D one."""
    ids = ('ts2d-v2-ep4000b2_cardiac', 'ts2d-v2-ep4000b2_muscles', 'ts2d-v2-ep4000b2_ribs')
    models = {m: synthetic_model(m, 3 + 2 * i, 41 + i, patch=(64, 64), mirror=True)[0] for i, m in enumerate(ids)}
    with TS2D(models=models) as ts:
        for asset in ('sample_s0521.nrrd', 'sample_s0616.nrrd'):
            ts.concurrent_models = True
            a = ts.predict(os.path.join(A, asset))
            a2 = ts.predict(os.path.join(A, asset))
            ts.concurrent_models = False
            b = ts.predict(os.path.join(A, asset))
            assert a.models == b.models == sorted(ids)
            assert np.array_equal(a.get_segmentation().array, b.get_segmentation().array) and a.get_segmentation().meta == b.get_segmentation().meta
            assert np.array_equal(a.get_segmentation().array, a2.get_segmentation().array)
            for m in ids:
                assert np.array_equal(a.get_segmentation(m).array, b.get_segmentation(m).array)
                t = a.data['models'][m]['timestamps']
                assert t['start'] <= t['preprocessed'] <= t['predicted'] <= t['exported']
            # concurrent: the prediction stages of the sub-models overlap in time
            ta = [a.data['models'][m]['timestamps'] for m in ids]
            assert max(t['preprocessed'] for t in ta) < min(t['predicted'] for t in ta) or len(ids) == 1


def test_device_thresholded_segmentation_equals_the_logits_path():
    """Round 6: where the export step needs no logits (multilabel, one fold, no resampling) ``HIPModel`` takes the segmentation thresholded on
    the device (K uint8 planes to the host instead of K float16 ones; kernels_sw.h) - the reference's seam, ``predict_logits_from_preprocessed_data``
    + ``export_prediction_from_logits`` (ts2d/core/inference/prediction_worker.py:209-221), must give the same bytes and metadata."""
    ids = ('ts2d-v2-ep4000b2_cardiac', 'ts2d-v2-ep4000b2_ribs')
    models = {m: synthetic_model(m, 3 + 2 * i, 51 + i, patch=(64, 64), mirror=True)[0] for i, m in enumerate(ids)}
    with TS2D(models=models) as ts:
        for asset in ('sample_s0521.nrrd', 'sample_s0616.nrrd'):
            for m in ts.models.values():
                m.device_threshold = True
            a = ts.predict(os.path.join(A, asset))
            for m in ts.models.values():
                m.device_threshold = False
            b = ts.predict(os.path.join(A, asset))
            assert np.array_equal(a.get_segmentation().array, b.get_segmentation().array) and a.get_segmentation().meta == b.get_segmentation().meta
            for m in ids:
                assert np.array_equal(a.get_segmentation(m).array, b.get_segmentation(m).array)
            assert a.get_segmentation().array.any()


def test_gpu_projection_equals_oracle_projection():
    """ts2d_project_coronal (strided view, no reorientation copy) against oracle/input_oracle.py (DICOMOrient 'RAI' + ITK max /
    mean projection + Float32 cast; the mean is real-valued, pinned by the reference's assets in tests/test_oracle.py): bit for
    bit - integer sums are exact, the float volume is summed in the same index order in double."""
    from oracle import input_oracle as IO
    from totalsegmentator2d_amd import image
    v = nrrd.read(os.path.join(A, 'sample_s0521.nrrd'))                                   # int16, direction diag(-1,-1,1)
    rng = np.random.default_rng(3)
    vols = [v,
            nrrd.Image(rng.normal(0, 300, (40, 33, 50)).astype(np.float32), (1.0, 2.0, 3.0), (5.0, -7.0, 11.0),
                       (0.0, -1.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, -1.0), 1, {}, 'left-posterior-superior'),   # permuted + flipped axes
            nrrd.Image(rng.integers(0, 255, (20, 16, 24)).astype(np.uint8), (1.0, 1.0, 1.0), (0.0, 0.0, 0.0),
                       (1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0), 1, {}, None),
            nrrd.Image(rng.integers(-1024, 3000, (24, 261, 40)).astype(np.int16), (1.5, 1.5, 1.5), (0.0, 0.0, 0.0),
                       (1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0), 1, {}, None)]          # 261 slices, as sample_s0616
    for vol in vols:
        got = image.project_coronal_gpu(vol)
        ref = IO.coronal_projections_f32(vol.array, vol.direction)
        r = image.reorient_image(vol)                         # geometry only (host logic, tested on CPU in tests/test_surface_cpu.py)
        for mode in ('max', 'mean'):
            g = got[mode]
            assert g.size == (r.size[0], 1, r.size[2]) and g.spacing == r.spacing
            assert np.allclose(g.origin, r.origin) and np.allclose(g.direction, r.direction)
            assert g.array.dtype == np.float32 and np.array_equal(g.array[:, 0, :], ref[mode]), mode
        if np.issubdtype(vol.array.dtype, np.integer):
            assert np.mean(got['mean'].array != np.round(got['mean'].array)) > 0.5    # real-valued, not truncated


def test_device_zscore_behind_the_projection():
    """ts2d_project_coronal_zscore: the per-channel z-score of the (max, mean) projections on the device (float64 two-pass
    statistics) against numpy - float32 values to a few ulps, statistics to float64 accuracy - and its use by the preprocessor:
    taken when nnU-Net's crop-to-nonzero is the identity, ignored (host pass, bit-identical to before) when it is not."""
    from types import SimpleNamespace
    from oracle import input_oracle as IO
    from totalsegmentator2d_amd import image, preprocess
    rng = np.random.default_rng(5)
    eye = (1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0)
    full = nrrd.Image((rng.normal(0, 300, (40, 33, 50)) - 200).astype(np.int16), (1.5, 1.5, 1.5), (0.0, 0.0, 0.0), eye, 1, {}, None)
    framed = np.zeros((30, 20, 40), np.int16)
    framed[5:25, :, 8:30] = rng.integers(-900, 900, (20, 20, 22))
    framed = nrrd.Image(framed, (1.5, 1.5, 1.5), (0.0, 0.0, 0.0), eye, 1, {}, None)
    v = nrrd.read(os.path.join(A, 'sample_s0521.nrrd'))
    for vol, expect_full in ((full, True), (v, None), (framed, False)):
        got = image.project_coronal_gpu(vol, zscore=True)
        zs = got['zscore']
        nz, nx = zs['shape']
        oplanes = IO.coronal_projections_f32(vol.array, vol.direction)           # the oracle's planes, not the kernel's own
        planes = [got[m].array.reshape(nz, nx) for m in ('max', 'mean')]
        for k, m in enumerate(('max', 'mean')):
            assert np.array_equal(planes[k], oplanes[m])
            ref = IO.zscore(oplanes[m])
            assert np.abs(zs['norm'][k] - ref).max() <= 4e-6 * max(1.0, float(np.abs(ref).max())), k
            mean64, std64 = IO.zscore_stats64(oplanes[m])
            assert abs(zs['stats'][2 * k] - mean64) <= 1e-9 * max(1.0, abs(mean64))
            assert abs(zs['stats'][2 * k + 1] - std64) <= 1e-9 * max(1.0, std64)
        nzmask = (planes[0] != 0) | (planes[1] != 0)
        rows, cols = np.where(nzmask.any(1))[0], np.where(nzmask.any(0))[0]
        assert zs['box'] == (rows[0], rows[-1], cols[0], cols[-1])
        is_full = zs['box'] == (0, nz - 1, 0, nx - 1)
        if expect_full is not None:
            assert is_full == expect_full
        # the preprocessor: same call with and without the device result
        data = np.stack(planes)[:, None].astype(np.float32)
        pm, cm = SimpleNamespace(transpose_forward=[0, 1, 2]), SimpleNamespace(spacing=(1.5, 1.5))
        pre = preprocess.DefaultPreprocessor(verbose=False)
        host, _, ph = pre.run_case_npy(data.copy(), None, {'spacing': (999.0, 1.5, 1.5)}, pm, cm, {})
        dev, _, pd = pre.run_case_npy(data.copy(), None, {'spacing': (999.0, 1.5, 1.5), 'device_zscore': dict(zs, order=(0, 1))}, pm, cm, {})
        assert ph['bbox_used_for_cropping'] == pd['bbox_used_for_cropping'] and host.shape == dev.shape
        if is_full:
            assert np.array_equal(dev[:, 0], zs['norm'])
            assert np.abs(dev - host).max() <= 4e-6 * max(1.0, float(np.abs(host).max()))
        else:
            assert np.array_equal(dev, host)                      # cropped first on the host: the device planes do not apply
