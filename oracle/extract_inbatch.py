"""ORACLE tooling -- build-container only. The reference's result for ONE slice when it is evaluated inside a larger batch: takes the
output of `make_golden_cfg2.py --stage ref --geometry 2x256x256` (same inputs and noise for slice 0 as the B = 1 golden: everything is
keyed by the global slice index) and stores slice 0 of its final image as loop_full_B1_256x256_T1000_start0_inbatch2.npz -- a second
reference execution of the B = 1 chain for tests/test_gpu_headline.py::reference_self_consistency. On this geometry the thread count
changes nothing (8 vs 4 threads: bit-identical), so the batch composition is what varies torch's summation order.
    python oracle/extract_inbatch.py"""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
src, dst = "loop_full_B2_256x256_T1000_start0", "loop_full_B1_256x256_T1000_start0_inbatch2"
g = np.load(os.path.join(GOLD, src + ".npz"))
one = np.load(os.path.join(GOLD, "loop_full_B1_256x256_T1000_start0.npz"))
np.savez_compressed(os.path.join(GOLD, dst + ".npz"), out=g["out"][:1])
d = np.abs(g["out"][:1].astype(np.float64) - one["out"])
states = {k: float(np.abs(g[k][:1].astype(np.float64) - one[k]).max()) for k in one.files if k.startswith("x_t")}
mp = os.path.join(GOLD, "MANIFEST.json")
m = json.load(open(mp))
m["cases"].pop(src, None)
m["cases"][dst] = dict(note="slice 0 of the reference's B = 2 run at 256 x 256 (final image only): the B = 1 golden's slice evaluated inside a batch of 2",
                       vs_B1_golden=dict(max=float(d.max()), rms=float(np.sqrt((d ** 2).mean())), n_over_1e4=int((d > 1e-4).sum()), states_max=states))
json.dump(m, open(mp, "w"), indent=1, sort_keys=True)
os.remove(os.path.join(GOLD, src + ".npz"))
print(dst, m["cases"][dst]["vs_B1_golden"])
