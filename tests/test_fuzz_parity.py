"""Randomised parity: the HIP path (both arithmetic modes) against the oracle on random profiles, option flags, k-mer
sizes, worker counts, seeds and ragged batches -- several batches per context, so that the carried stream states,
the lean/generic split, every dwell regime and the segment tails of k_events are all exercised.  Bit-exact."""
import numpy as np
import pytest

import orc
from squigulator_amd import api, model, profiles

FLAG_SETS = [0, profiles.SQ_PREFIX, profiles.SQ_RNA, profiles.SQ_RNA | profiles.SQ_PREFIX, profiles.SQ_IDEAL_TIME,
             profiles.SQ_IDEAL_AMP, profiles.SQ_IDEAL, profiles.SQ_RNA | profiles.SQ_PREFIX | profiles.SQ_IDEAL_TIME]


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    base, _ = profiles.get_profile(["dna-r9-prom", "dna-r9-min", "rna-r9-prom", "dna-r10-prom", "rna004-prom"][seed % 5])
    dwell_mean = float(rng.choice([2.0, 5.0, 9.0, 13.0, 31.0, 43.0, 120.0, 600.0]))
    dwell_std = float(rng.choice([0.0, 0.5, 4.0, dwell_mean * 0.8])) if seed % 7 else 0.0
    prof = base.replace(dwell_mean=dwell_mean, dwell_std=dwell_std,
                        offset_mean=base.offset_mean + float(rng.normal(0, 30)),
                        offset_std=float(rng.choice([0.0, 5.0, 20.0])),
                        range=base.range * float(rng.uniform(0.5, 2.0)))
    flags = FLAG_SETS[int(rng.integers(0, len(FLAG_SETS)))]
    k = int(rng.choice([5, 6, 6, 9]))
    T = int(rng.integers(1, 9))
    amp = float(rng.choice([1.0, 1.0, 0.3, 2.5]))
    nb = int(rng.integers(1, 4))
    batches = []
    for _ in range(nb):
        n = int(rng.integers(1, 2 * T + 2))
        lens = rng.choice([1, 3, k - 1, k, k + 1, 63, 64, 65, 200, 511, 512, 513, 1025, 3000], n)
        batches.append([bytes(rng.choice(list(b"ACGTacgtNRY"), int(m), p=[.22, .22, .22, .22, .02, .02, .02, .02, .02, .01, .01]).astype(np.uint8))
                        for m in lens])
    return prof, flags, k, T, amp, int(rng.integers(1, 1 << 30)), batches


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(40))
def test_random_configuration_matches_oracle(seed, monkeypatch):
    if seed % 2:                                         # odd seeds: the 256-thread k_events even for these small launches
        monkeypatch.setenv("SQG_EVENTS_WIDE_MAX", "0")
    prof, flags, k, T, amp, s, batches = _case(seed)
    mean, stdv = model.synthetic_model(k, salt=seed)
    orac = orc.Oracle(prof, flags, k, mean, stdv, s, num_workers=T, amp_noise=amp)
    want = [orac.run_batch_seqs(bt) for bt in batches]
    orac.close()
    for mode in (api.MODE_CERTIFIED, api.MODE_EXACT):
        gen = api.SignalGenerator(prof, flags, k, mean, stdv, s, num_workers=T, amp_noise=amp, mode=mode)
        for bi, bt in enumerate(batches):
            b = gen.submit(bt)
            sig, dw = b.signal(), b.dwell()
            for i, w in enumerate(want[bi]):
                np.testing.assert_array_equal(sig[b.sig_off[i]:b.sig_off[i + 1]], w.sig,
                                              err_msg=f"seed {seed} mode {mode} batch {bi} read {i} (k={k} T={T} flags={flags:#x} dwell={prof.dwell_mean}/{prof.dwell_std})")
                np.testing.assert_array_equal(dw[b.ev_off[i]:b.ev_off[i + 1]], w.ss)
                assert b.offset[i] == w.offset and b.median_before[i] == w.median_before
            b.free()
        gen.close()


def _few_case(seed):
    """few workers, chains cut into links (forced: the batches are small): every k of the bucketed hand-out (7, 8, 9: 4, 16, 64
    partitions) and of its one-partition case (5, 6), dwells from 2 to 600 samples (constant ones included), every flag set"""
    rng = np.random.default_rng(5000 + seed)
    base, _ = profiles.get_profile(["dna-r10-prom", "rna004-prom", "dna-r9-prom", "rna-r9-prom"][seed % 4])
    dwell_mean = float(rng.choice([2.0, 9.0, 13.0, 31.0, 120.0, 600.0]))
    dwell_std = float(rng.choice([0.0, 0.5, 4.0, dwell_mean * 0.8])) if seed % 5 else 0.0
    prof = base.replace(dwell_mean=dwell_mean, dwell_std=dwell_std, offset_std=float(rng.choice([0.0, 5.0])),
                        range=base.range * float(rng.uniform(0.6, 1.8)))
    flags = FLAG_SETS[int(rng.integers(0, len(FLAG_SETS)))]
    k = [9, 8, 7, 6, 5, 9][seed % 6]
    T = int(rng.integers(1, 5))
    batches = []
    for _ in range(int(rng.integers(2, 4))):
        n = int(rng.integers(T + 1, 3 * T + 8))
        lens = rng.choice([1, k - 1, k, k + 1, 64, 65, 300, 511, 512, 513, 1023, 1024, 1025, 1600, 2100], n)
        batches.append([bytes(rng.choice(list(b"ACGTacgtNRY"), int(m), p=[.22, .22, .22, .22, .02, .02, .02, .02, .02, .01, .01]).astype(np.uint8))
                        for m in lens])
    links = str(rng.choice([2, 7, 40, 100000]))
    return prof, flags, k, T, int(rng.integers(1, 1 << 30)), batches, links


# The fall-back paths of the few-worker regime -- what a device that fails the LDS-order check, a 5^8 / 5^9 table, a batch that is not
# staged ahead ... run -- take the same randomised cases as the default path (fewer of them): "order-free" through cfg.flags (the
# release library's own switch, include/sqg.h SQG_ORDER_FREE), the others through the development library's knobs.
VARIANTS = {
    "default": {},
    "order-free": {"cfg": profiles.SQ_ORDER_FREE},                 # k_part_hand (claims), k_events<..., PART>; k <= 6: per-link rows
    "per-link-rows": {"env": {"SQG_NO_PART": "1"}},                # k_events<HIST> + k_link_prefix (round 1's path)
    "no-precount": {"env": {"SQG_NO_PRECOUNT": "1"}},              # the first event pass always as a launch of its own
    "wg-per-link": {"env": {"SQG_PART_WG_EVENTS": "1"}},           # k_events<..., PART> with the ordered hand-out
}


def _few_worker_case(seed, monkeypatch, variant="default"):
    prof, flags, k, T, s, batches, links = _few_case(seed)
    monkeypatch.setenv("SQG_SPLIT_CHAINS", links)
    if variant == "default" and seed % 9 == 8:
        monkeypatch.setenv("SQG_PART_CLAIMS", "1")       # the order-free kernels
    for name, val in VARIANTS[variant].get("env", {}).items():
        monkeypatch.setenv(name, val)
    flags |= VARIANTS[variant].get("cfg", 0)
    mean, stdv = model.synthetic_model(k, salt=seed)
    orac = orc.Oracle(prof, flags & ~profiles.SQ_ORDER_FREE, k, mean, stdv, s, num_workers=T)
    want = [orac.run_batch_seqs(bt) for bt in batches]
    orac.close()
    def check(b, bi, mode, how):
        sig, dw = b.signal(), b.dwell()
        for i, w in enumerate(want[bi]):
            np.testing.assert_array_equal(sig[b.sig_off[i]:b.sig_off[i + 1]], w.sig,
                                          err_msg=f"seed {seed} {variant} mode {mode} {how} batch {bi} read {i} (k={k} T={T} flags={flags:#x} links={links} dwell={prof.dwell_mean}/{prof.dwell_std})")
            np.testing.assert_array_equal(dw[b.ev_off[i]:b.ev_off[i + 1]], w.ss)
            assert b.offset[i] == w.offset and b.median_before[i] == w.median_before
    for mode in (api.MODE_CERTIFIED, api.MODE_EXACT):
        gen = api.SignalGenerator(prof, flags, k, mean, stdv, s, num_workers=T, mode=mode)
        for bi, bt in enumerate(batches):
            b = gen.submit(bt)
            check(b, bi, mode, "submit")
            b.free()
        gen.close()
        # the same job streamed: batch i+2 staged, batch i+1 queued, batch i consumed -- every run finds its successor staged, whose first
        # event pass then rides along with this batch's hand-out (k_part_hand_count) wherever the kind of batch allows it
        gen = api.SignalGenerator(prof, flags, k, mean, stdv, s, num_workers=T, mode=mode)
        cur = gen.stage(batches[0]).run()
        nxt = gen.stage(batches[1]) if len(batches) > 1 else None
        for bi in range(len(batches)):
            nn = gen.stage(batches[bi + 2]) if bi + 2 < len(batches) else None
            if nxt is not None:
                nxt.run()
            cur.wait()
            check(cur, bi, mode, "streamed")
            cur.free()
            cur, nxt = nxt, nn
        gen.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(36))
def test_random_few_worker_configuration_matches_oracle(seed, monkeypatch):
    _few_worker_case(seed, monkeypatch)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(36, 52))
@pytest.mark.parametrize("variant", [v for v in VARIANTS if v != "default"])
def test_few_worker_fallback_paths_match_oracle(variant, seed, monkeypatch):
    """(seeds of their own: sixteen more random cases per fall-back, every k of the bucketed hand-out and of its one-partition case)"""
    _few_worker_case(seed, monkeypatch, variant)
