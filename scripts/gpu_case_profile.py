"""cProfile of TS2D.predict() on sample_s0616 (five canonical sub-models, serial order): where the host time of a case goes."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from totalsegmentator2d_amd import nrrd
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd.model import HIPModel
from totalsegmentator2d_amd.tool import TS2D

groups = [('cardiac', 18), ('muscles', 23), ('organs', 24), ('ribs', 26), ('vertebrae', 26)]
models = {}
for i, (g, K) in enumerate(groups):
    arch = UNetArch.canonical(num_classes=K)
    blob = (np.random.default_rng(i).standard_normal(arch.n_params()) * 0.02).astype(np.float32)
    ds = {'channel_names': {'0': 'mean', '1': 'max'}, 'labels': {'background': 0, **{f'{g}_{j+1}': j + 1 for j in range(K)}},
          'file_ending': '.nrrd', 'multilabel': True}
    models[f'ts2d-v2-ep4000b2_{g}'] = HIPModel({'model': f'ts2d-v2-ep4000b2_{g}', 'revision': 1, 'param': {},
                                               'synthetic': {'arch': arch, 'blobs': [blob], 'patch_size': (512, 512), 'dataset_json': ds}})
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'assets', 'sample_s0616.nrrd')
with TS2D(models=models) as ts:
    img = nrrd.read(path)
    ts.concurrent_models = len(sys.argv) > 1 and sys.argv[1] == 'concurrent'
    for _ in range(3): ts.predict(img)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): ts.predict(img)
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue())
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(25); print(s.getvalue())
