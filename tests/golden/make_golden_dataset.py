#!/usr/bin/env python3
"""tests/golden/dataset.npz from the REAL reference Dataset classes (/root/reference/dataset.py).

The reference's classes list their files with ``librosa.util.find_files`` (absent here): the shim installs that
function's documented behaviour (recursive, sorted).  tests/dataset_fixture.py writes a small corpus of ``.pt`` feature
files in the reference's naming scheme; every class is constructed on it and iterated twice in a fixed order under
``np.random.seed`` (so the crop draws are pinned); every item - feature tensor (as fx.digest: shape, position-weighted float64 sums, first / last frame), filename, tag, label, channel / device -
and the DataLoader batches are stored.  The silence frame the reference computes at import (dataset.py:13-16) is
stored too (the build computes it with its HIP kernel; CPU tests take it from here).

Usage:  python tests/golden/make_golden_dataset.py
"""
import os
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import numpy as np
import torch

import dataset_fixture as fx
import make_golden as mg


def main():
    mg.install_shims()
    import librosa

    def find_files(directory, ext=None, recurse=True, case_sensitive=False, limit=None, offset=0):
        exts = [ext] if isinstance(ext, str) else list(ext)
        out = []
        for dp, _, fs in os.walk(directory):
            out += [os.path.join(dp, f) for f in fs if any(f.lower().endswith("." + e.lower()) for e in exts)]
        return sorted(out)

    librosa.util.find_files = find_files
    import dataset as ref
    from torch.utils.data import DataLoader

    arrays = {"silence_pad_value": ref.silence_pad_value.numpy()}
    with tempfile.TemporaryDirectory() as root:
        fx.build(root)
        for name, make in fx.cases(ref, root).items():
            ds = make()
            n = len(ds)
            order = list(range(n)) + list(reversed(range(n)))  # two passes: the crop draws move on
            np.random.seed(1234)
            metas = []
            for k, i in enumerate(order):
                item = ds[i]
                for key, v in fx.digest(item[0].numpy()).items():
                    arrays["%s/feat%d/%s" % (name, k, key)] = v
                metas.append([str(v) if isinstance(v, str) else np.asarray(v).tolist() for v in item[1:]])
            arrays["%s/meta" % name] = np.array(repr(metas))
            arrays["%s/order" % name] = np.array(order)
            # a DataLoader batch through the class's own collate_fn (main_train.py:246)
            np.random.seed(99)
            dl = DataLoader(ds, batch_size=3, shuffle=False, num_workers=0, collate_fn=ds.collate_fn)
            for bi, batch in enumerate(dl):
                for key, v in fx.digest(batch[0].numpy()).items():
                    arrays["%s/batch%d_feat/%s" % (name, bi, key)] = v
                arrays["%s/batch%d_rest" % (name, bi)] = np.array(repr([
                    (list(b) if isinstance(b, (list, tuple)) else b.tolist()) for b in batch[1:]]))
            print("  %-12s %d items, %d batches" % (name, n, bi + 1))
    mg.save("dataset.npz", **arrays)


if __name__ == "__main__":
    main()
