"""tests/golden/check_goldens.py -- re-run the round-3 fixture generators against /root/reference into a scratch directory and
compare every array with the committed fixture (build container only).

    python tests/golden/check_goldens.py

Prints, per file, the number of arrays and the largest absolute difference (strings / hashes: equal or not); exit code 1 on
any difference.  The generators take their output directory from LFD_GOLDEN_OUT."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GENERATORS = ['make_golden_train_step.py', 'make_golden_fullsize_results.py', 'make_golden_fullsize_model.py',
              'make_golden_nms_large.py', 'make_golden_config1.py']


def main():
    bad = 0
    with tempfile.TemporaryDirectory() as d:
        for gen in GENERATORS:
            r = subprocess.run([sys.executable, os.path.join(HERE, gen)], env=dict(os.environ, LFD_GOLDEN_OUT=d), capture_output=True,
                               text=True)
            if r.returncode != 0:
                print(gen, 'FAILED', r.stderr[-500:])
                bad += 1
        for f in sorted(os.listdir(d)):
            new, old = np.load(os.path.join(d, f)), np.load(os.path.join(HERE, f))
            worst, keys = 0.0, sorted(new.files)
            if keys != sorted(old.files):
                print(f, 'KEYS DIFFER', sorted(set(keys) ^ set(old.files))[:6])
                bad += 1
                continue
            for k in keys:
                a, b = new[k], old[k]
                if a.dtype.kind in 'US' or b.dtype.kind in 'US':
                    same = a.shape == b.shape and bool(np.all(a == b))
                    worst = max(worst, 0.0 if same else float('inf'))
                elif a.shape != b.shape:
                    worst = float('inf')
                elif a.size:
                    worst = max(worst, float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))))
            print('%-40s %4d arrays, max abs diff %g' % (f, len(keys), worst))
            bad += worst != 0.0
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
