#!/usr/bin/env python3
"""tools/nn_probe.py -- a minimal workload for rocprofv3 counter passes: `reps`
reciprocal correspondence searches (forward + reverse brute-force NN, filter,
moments) on one synthetic scan pair.  Prints one JSON line with HIP-event
timings of the NN kernel family.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU ... -- python3 tools/nn_probe.py 200000 3
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    fma = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    knobs = dict(kv.split("=") for kv in sys.argv[4:])          # e.g. nn_q=8 nn_sub=32 nn_blocks_per_cu=3
    mvr = importlib.import_module("multi-view-registration_amd")
    sp = mvr.synth_params(12, 3)
    a, b = mvr.synth_view(sp, 0, n), mvr.synth_view(sp, 1, n)
    piv, ax = mvr.synth_prior(sp)
    with mvr.Context(0) as ctx:
        ctx.tune(**{k: int(v) for k, v in knobs.items()})
        ctx.upload(0, a); ctx.upload(1, b)
        ctx.transform(1, 1, mvr.axis_rotation(piv, ax, mvr.turntable_angle(1, 12)))
        ctx.pair_moments2(1, 0, 4.0, np.array(sp.pivot), fma=bool(fma))          # warm-up
        ctx.prof_reset(); ctx.prof_enable(True)
        for _ in range(reps):
            m2 = ctx.pair_moments2(1, 0, 4.0, np.array(sp.pivot), fma=bool(fma))
        ctx.prof_enable(False)
        launches, ms, evals = ctx.prof_get(mvr.K_NN)
        dbg = ctx.debug_counters()
        print(json.dumps(dict(n=n, reps=reps, fma=fma, knobs=knobs, nn_launches=launches, nn_ms=ms, nn_evals=evals,
                              evals_per_s=evals / (ms * 1e-3), tflops_canonical=8 * evals / (ms * 1e-3) / 1e12,
                              n_corr=m2.n, max_tiles_per_wave=dbg[2], max_tested_per_wave=dbg[3])))


if __name__ == "__main__":
    main()
