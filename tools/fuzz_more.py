#!/usr/bin/env python3
"""tests/test_fuzz_parity.py's two randomised parity tests over seeds the suite does not hold: python tools/fuzz_more.py [first] [count] [variant]
(variant: one of test_fuzz_parity.VARIANTS -- the few-worker test on a fall-back path; default: both tests on the default path)
(a one-off sweep for latent bugs: every case is the HIP path, both arithmetic modes, submitted and streamed, against the oracle)."""
import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_fuzz_parity as F  # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    variant = sys.argv[3] if len(sys.argv) > 3 else None
    bad = []
    t0 = time.time()
    fns = (F.test_random_configuration_matches_oracle, F.test_random_few_worker_configuration_matches_oracle)
    if variant:
        def on_variant(seed, mp):
            F._few_worker_case(seed, mp, variant)
        on_variant.__name__ = "few_worker[" + variant + "]"
        fns = (on_variant,)
    for seed in range(first, first + count):
        for fn in fns:
            with pytest.MonkeyPatch.context() as mp:
                try:
                    fn.__wrapped__(seed, mp) if hasattr(fn, "__wrapped__") else fn(seed, mp)
                except Exception as e:  # noqa: BLE001
                    bad.append((fn.__name__, seed, str(e)[:600]))
                    print("FAIL", fn.__name__, seed, str(e)[:600], flush=True)
    vtag = ", " + variant if variant else ""
    print(f"{len(fns) * count} cases (seeds {first}..{first + count - 1}{vtag}), {len(bad)} failures, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
