"""``ts2d`` command line (reference ``ts2d/main.py:10-115``): same flags, same output file names."""
from __future__ import annotations

import os
from glob import glob

from .tool import TS2D
from .zoo import DEFAULT_MODEL


_KNOWN_EXT = ('nrrd', 'nii', 'nii.gz', 'mha', 'mhd')     # what the reference CLI accepts (ts2d/main.py:25)
_READABLE_EXT = ('nrrd',)                                 # what this build can read (no SimpleITK / nibabel)


def _check_case(path: str):
    """(case name, path) for one input file, or a ValueError / FileNotFoundError saying why it is not a case."""
    if not os.path.isfile(path):
        if not os.path.exists(path):
            raise FileNotFoundError(f"Source file does not exist: {path}")
        raise ValueError(f"Source is not a regular file: {path}")
    name, dot, ext = os.path.basename(path).partition('.')           # case name = up to the FIRST dot ('a.nii.gz' -> 'a')
    if not dot:
        raise ValueError(f"Source file does not have an extension: {os.path.basename(path)}")
    if ext not in _KNOWN_EXT:
        raise ValueError(f"Unsupported file extension: {ext} in {os.path.basename(path)}")
    if ext not in _READABLE_EXT:
        raise ValueError(f"only NRRD input is implemented in the MI355X build (no SimpleITK): {os.path.basename(path)}")
    return name, path


def _enumerate_cases(src: str):
    """A single file must be a valid case (its error propagates); a directory contributes every file that is one and silently
    skips the rest - the behaviour of the reference CLI (ts2d/main.py:10-32), in sorted order."""
    if not os.path.isdir(src):
        return [_check_case(src)]
    cases = []
    for path in sorted(glob(os.path.join(src, "*.*"))):
        try:
            cases.append(_check_case(path))
        except (ValueError, FileNotFoundError):
            pass
    return cases


def ts2d_run(src: str, dest: str, model: str = None, use_remote: bool = True, fetch_remote: bool = True, collapse: bool = False,
             visualize: bool = True, save_all: bool = False, silent: bool = False, models=None):
    model = DEFAULT_MODEL if model is None else model
    content = 'all' if visualize else 'file'
    which = 'all' if save_all else 'final'
    log = (lambda *a: None) if silent else print
    log("TS2D is a research tool. It is NOT validated for clinical use and should NOT be used for medical diagnosis or treatment.")
    with TS2D(key=model, use_remote=use_remote, fetch_remote=fetch_remote, models=models) as ts:
        cases = list(_enumerate_cases(src))
        log(f"Predicting {len(cases)} case{'s' if len(cases) != 1 else ''}")
        for i, (name, path) in enumerate(cases):
            log(f"[{i + 1}/{len(cases)}] Processing: {name}")
            res = ts.predict(path, collapse=collapse)
            res.save(dest=dest, name=name, models=which, content=content, targets=['segmentation', 'projection'])


def ts2d_entry_point(argv=None):
    import argparse
    p = argparse.ArgumentParser(description="Runs TotalSegmentator2D (TS2D) on images or directories of images (MI355X engine).")
    p.add_argument("--src", "-i", "--input", type=str, required=True, help="Input image file or directory (nrrd)")
    p.add_argument("--dest", "-o", "--output", type=str, required=True, help="Output directory for results.")
    p.add_argument("--model", type=str, default=None, help=f"Model key for prediction, defaults to '{DEFAULT_MODEL}'.")
    p.add_argument("--no-remote", action="store_true", help="Disable remote model download (always the case here).")
    p.add_argument("--no-fetch", action="store_true", help="Do not fetch model URLs (always the case here).")
    p.add_argument("--collapse", action="store_true", help="Collapse projected images to 2D.")
    p.add_argument("--visualize", action="store_true", help="Visualize the results as PNG images (not implemented).")
    p.add_argument("--save-all", action="store_true", help="Also save results for each individual model.")
    p.add_argument("--silent", action="store_true", help="Hides any unnecessary output.")
    a = p.parse_args(argv)
    ts2d_run(src=a.src, dest=a.dest, model=a.model, use_remote=not a.no_remote, fetch_remote=not a.no_fetch, collapse=a.collapse,
             visualize=a.visualize, save_all=a.save_all, silent=a.silent)


if __name__ == '__main__':
    ts2d_entry_point()
