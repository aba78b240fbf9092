"""bench.py --gpus N launches the N ranks itself and runs the HIP path in every rank.  On a one-GPU
box the ranks share the GPU (--backend gloo); what must hold is what holds on N GPUs: the ranks' signals are, read for
read, those of the single-rank run of the same job -- in both sharding modes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-store-probe", "--steps", "2",
                        "--warmup", "1"] + [str(a) for a in args], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("name,single,dual", [
    # 9-mers, one worker per GPU (the headline regime): -t 2 -K 1024
    ("r10_by_worker", ["--genome-mb", 24, "--workers-per-gpu", 2, "--batch-reads", 1024, "--digest", 4],
                      ["--genome-mb", 24, "--workers-per-gpu", 1, "--batch-reads", 512, "--digest", 2]),
    # 9-mers, strict -t 1: every rank owns the worker and generates half of each batch; counts all-gathered per batch
    ("r10_by_range", ["--genome-mb", 24, "--job-workers", 1, "--batch-reads", 1024, "--digest", 4],
                     ["--genome-mb", 24, "--job-workers", 1, "--batch-reads", 512, "--digest", 2]),
    # 6-mers, one worker per read: -t 512 -K 512
    ("r9_t_equals_k", ["--workload", "ncov-r9", "--batch-reads", 512, "--digest", 4],
                      ["--workload", "ncov-r9", "--batch-reads", 256, "--digest", 2]),
    # 6-mers by range with three workers
    ("r9_by_range_t3", ["--workload", "ncov-r9", "--job-workers", 3, "--batch-reads", 600, "--digest", 4],
                       ["--workload", "ncov-r9", "--job-workers", 3, "--batch-reads", 300, "--digest", 2]),
], ids=lambda v: v if isinstance(v, str) else None)
def test_two_ranks_equal_one(name, single, dual):
    one = _bench("--gpus", 1, *single)
    two = _bench("--gpus", 2, "--backend", "gloo", *dual)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["steps"] == 2 and len(two["digest"]) == 2
    assert one["digest"] == two["digest"], (one["digest"], two["digest"])
    # the same reads and samples in total (value = whole-job samples / max-over-ranks time)
    n1 = one["samples_per_step_per_gpu"] * one["steps"]
    n2 = two["value"] * two["ms_per_step"] * 1e-3 * two["steps"]
    assert abs(n1 - n2) <= 1e-6 * n1
    assert two["reads_per_s"] * two["ms_per_step"] * 1e-3 * two["steps"] == pytest.approx(2 * one["config"]["reads_per_step_per_gpu"], rel=1e-6)
    # every rank checked its own shard against the oracle's run of the whole job (worker sharding; bench.parity_check_rank)
    if "by_range" not in name:
        pr = two["parity_check_ranks"]
        assert pr["equal"] is True and pr["reads_differing"] == [0, 0] and all(n > 0 for n in pr["reads"]) and len(set(pr["digest"])) == 2
    else:
        assert "parity_check_ranks" not in two


@pytest.mark.gpu
@pytest.mark.parametrize("name,single,eight", [
    # the node's size, by worker: -t 8 -K 2048, worker g = GPU g = the reads [256 g, 256 (g + 1)) of every batch
    ("r10_by_worker", ["--genome-mb", 24, "--workers-per-gpu", 8, "--batch-reads", 2048, "--digest", 8],
                      ["--genome-mb", 24, "--workers-per-gpu", 1, "--batch-reads", 256, "--digest", 1]),
    # strict -t 1 over eight ranks: the 8-way all-gather of the stream counts, once per batch
    ("r10_by_range", ["--genome-mb", 24, "--job-workers", 1, "--batch-reads", 2048, "--digest", 8],
                     ["--genome-mb", 24, "--job-workers", 1, "--batch-reads", 256, "--digest", 1]),
], ids=lambda v: v if isinstance(v, str) else None)
def test_eight_ranks_equal_one(name, single, eight):
    """world size 8 -- eight contexts, eight staging pools, shard.worker_range(r, 8, 8), the 8-way exchange -- sharing the box's GPU
    (--backend gloo): read for read the single-rank run of the same job"""
    one = _bench("--gpus", 1, *single)
    many = _bench("--gpus", 8, "--backend", "gloo", *eight)
    assert many["n_gpus"] == 8 and many["ranks"]["world_size"] == 8 and len(many["digest"]) == 2 and len(many["digest"][0]) == 8
    assert one["digest"] == many["digest"], (one["digest"], many["digest"])
    assert many["reads_per_s"] * many["ms_per_step"] * 1e-3 * many["steps"] == pytest.approx(2 * 2048, rel=1e-6)
    if name == "r10_by_worker":
        pr = many["parity_check_ranks"]                              # eight ranks, each its own worker of -t 8 against the oracle's -t 8 run
        assert pr["equal"] is True and pr["reads_differing"] == [0] * 8 and pr["reads"] == [8] * 8 and len(set(pr["digest"])) == 8


@pytest.mark.gpu
def test_eight_ranks_stream_on_the_box_cpu_quota():
    """the streaming leg with eight ranks on the box's CPU quota (16 CPUs on the pool's boxes): every rank sizes its staging helpers
    from its share of the quota (bench.py: sqg_set_stage_threads), reports its host time per batch, and the leg completes on every rank"""
    d = _bench("--gpus", 8, "--backend", "gloo", "--genome-mb", 24, "--batch-reads", 512, "--workers-per-gpu", 1, "--pipeline-seconds", 0.5)
    pl = d["pipeline"]
    assert d["n_gpus"] == 8 and pl["value"] > 0 and pl["batches_per_gpu"] >= 9
    assert 0 < pl["host_stage_ms_per_batch_min"] <= pl["host_stage_ms_per_batch"] <= pl["host_stage_ms_per_batch_max"]
    assert 1 <= pl["stage_threads"] <= 4 and pl["cpus_per_rank"] >= 1


@pytest.mark.gpu
def test_genome_broadcast_from_rank0_and_numa_pin():
    """SURVEY.md section 2, C1 at N > 1: rank 0 makes the genome, the other ranks receive its bytes (one broadcast); --numa-pin on keeps
    a rank's host threads on its GPU's NUMA node.  Same reads, same signals as the plain run."""
    args = ["--genome-mb", 24, "--workers-per-gpu", 1, "--batch-reads", 512, "--digest", 2]
    plain = _bench("--gpus", 2, "--backend", "gloo", *args)
    bc = _bench("--gpus", 2, "--backend", "gloo", "--genome-from-rank0", "--numa-pin", "on", *args)
    assert plain["digest"] == bc["digest"]
    assert bc["ranks"]["genome"] == "broadcast from rank 0" and plain["ranks"]["genome"] == "made on every rank"
    pin = bc["ranks"]["numa_pin"]
    assert isinstance(pin, dict) and (pin["pinned"] is True and pin["cpus"] >= 1 or pin["why"])
    assert plain["ranks"]["numa_pin"] is None
    rccl = _bench("--gpus", 1, "--force-dist", "--backend", "nccl", "--genome-from-rank0", "--genome-mb", 24, "--workers-per-gpu", 1,
                  "--batch-reads", 1024, "--digest", 4)          # the broadcast through RCCL (one rank)
    one = _bench("--gpus", 1, "--genome-mb", 24, "--workers-per-gpu", 1, "--batch-reads", 1024, "--digest", 4)
    assert rccl["digest"] == one["digest"]


@pytest.mark.gpu
def test_order_free_flag_gives_the_same_bytes():
    """bench.py --order-free (SQG_ORDER_FREE in cfg.flags: the kernels that do not rely on lane-ordered LDS atomics) -- the fallback whose cost
    profiles/r04_summary.md quotes: same reads, same signals"""
    args = ["--genome-mb", 24, "--workers-per-gpu", 1, "--batch-reads", 1024, "--digest", 4]
    a = _bench("--gpus", 1, *args)
    b = _bench("--gpus", 1, "--order-free", *args)
    assert a["digest"] == b["digest"] and b["config"]["order_free"] is True and a["config"]["order_free"] is False


@pytest.mark.gpu
def test_gpus_flag_without_enough_devices_fails_loudly():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    import torch
    n = torch.cuda.device_count() + 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-store-probe", "--genome-mb", "8"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0
    assert "ranks but" in (p.stderr + p.stdout)


@pytest.mark.gpu
def test_bench_line_carries_the_contract():
    """the one JSON line of the driver's command (small genome and batches, short CPU legs): every field the contract names, the
    roofline and cpu_baseline objects, the parity check of the timed regime"""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--genome-mb", "24",
                        "--batch-reads", "1024", "--cpu-seconds", "1", "--pipeline-seconds", "0.5", "--e2e-seconds", "0.3", "--small-batch-seconds", "0.3"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["metric"] == "simulated raw samples/sec" and d["unit"] == "samples/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None and d["value"] > 0 and d["ms_per_step"] > 0 and isinstance(d["dtype"], str)
    assert "workload" in d["config"] and "dna-r10-prom" in d["config"]["workload"] and "model" not in d["config"]
    assert d["config"]["lds_ordered_hand_out"] == {"in_use": True, "probe_mismatches": 0}      # (the fast hand-out ran; a device that fails the probe says so here)
    # No clocks and no rate ratios below: presence, type, units and the line's internal consistency only.  How long a leg
    # took, or which leg is faster than which, is what the line REPORTS, not what a test asserts.
    r = d["roofline"]
    assert r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["bound"] in ("hbm", "valu", "l2_requests", "stores")
    assert "traffic" in r and r["kernel"] == "k_samples_lean" and r["kernel_ms"] > 0 and r["step_frac"] > 0
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9, rel=1e-6)
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["unit"] == "samples/s" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert d["parity_check"]["equal"] is True and d["parity_check"]["reads_differing"] == 0
    # the streaming leg (nothing staged ahead), the library that was timed, the ranks' own clocks
    pl = d["pipeline"]
    assert pl["unit"] == "samples/s" and pl["value"] > 0 and pl["seconds"] > 0 and pl["batches_per_gpu"] >= 9      # (>= 8 + 1 batches by construction)
    assert pl["vs_value"] == pytest.approx(pl["value"] / d["value"], rel=1e-9)
    lib = d["library"]
    assert lib["in_tree"] is True and lib["stale"] is False and len(lib["sha256_16"]) == 16 and len(lib["source_hash"]) == 16
    assert lib["built_from"] == lib["source_hash"] and lib["dev"] is False          # the release library, stamped with the tree's hash
    assert d["ranks"]["world_size"] == 1 and d["ranks"]["ms_per_step_min"] == pytest.approx(d["ms_per_step"], rel=1e-9)
    # the end-to-end legs (src/sim.c:602-611,630-641): raw int16 / svb-zd into pinned host memory, BLOW5 into /dev/shm
    e = d["e2e"]
    for leg in ("pinned_int16", "pinned_svb", "blow5", "blow5_fast", "blow5_fast_4files"):
        assert e[leg]["unit"] == "samples/s" and e[leg]["value"] > 0 and e[leg]["batches"] >= 2 and e[leg]["seconds"] > 0, (leg, e[leg])
    assert e["pinned_int16"]["bytes_per_sample"] == 2.0 and 0.5 < e["pinned_svb"]["bytes_per_sample"] < 2.0
    assert 0.4 < e["blow5"]["bytes_per_sample"] < e["pinned_svb"]["bytes_per_sample"]      # (zlib over the svb-zd bytes: a property of the bytes, not of time)
    # the stored-block writer (SQG_BLOW5_STORED): the svb-zd bytes + ~130 B of framing per record (every leg draws batches of its own:
    # the bytes per sample agree to a few percent, not to the byte)
    assert e["pinned_svb"]["bytes_per_sample"] * 0.95 < e["blow5_fast"]["bytes_per_sample"] < e["pinned_svb"]["bytes_per_sample"] * 1.05 + 0.01
    assert e["blow5_fast_4files"]["bytes_per_sample"] == pytest.approx(e["blow5_fast"]["bytes_per_sample"], rel=0.05)
    if c["kind"] == "reference":
        assert c["to_blow5"] > 0
    # kernel_ms over >= 20 launches (a leg in which every batch carries the phase events), the reference's default batch size streaming
    # with one and with eight virtual workers (src/sim.c:208-209)
    ev = d["kernel_ms_every_batch"]
    assert ev["launches"] >= 20 and ev["k_samples_lean"] > 0 and ev["k_samples_lean_min"] <= ev["k_samples_lean"] <= ev["k_samples_lean_max"]
    sb = d["small_batch"]
    for leg in ("-t 1 -K 1000", "-t 8 -K 1000"):
        assert sb[leg]["unit"] == "samples/s" and sb[leg]["value"] > 0 and sb[leg]["seconds"] > 0 and sb[leg]["batches"] >= 65, (leg, sb[leg])
        assert sb[leg]["ms_per_batch"] == pytest.approx(sb[leg]["seconds"] / sb[leg]["batches"] * 1e3, rel=1e-9)      # (>= 64 + 1 batches by construction)
    assert "resources" in r and (r["resources"] is None or r["resources"]["bound"] in ("valu", "l2_requests", "stores", "hbm"))
    if r["resources"] is not None:
        assert r["bound"] == r["resources"]["bound"]                 # the line says ONE thing about its bound
    # roofline.traffic measured by the run itself (two rocprofv3 --pmc child passes on this box) where rocprofv3 exists and its passes
    # succeed; a failed pass leaves its counters out of the line, never the line out -- and never this test red
    assert r.get("traffic_source") in ("live", "profile", None)
    if r.get("traffic_source") == "live":
        lv = r["traffic_live"]
        assert lv["launches"] >= 1
        assert r["traffic"] == pytest.approx((2 * lv["FETCH_SIZE_KiB"] + lv["WRITE_SIZE_KiB"]) * 1024, rel=1e-9)
        assert isinstance(lv["failed_passes"], list)
        rs = r["resources"]
        if rs is not None:
            assert rs["source"].startswith("live") and rs["bound"] in ("valu", "l2_requests", "stores", "hbm")


def test_live_traffic_parsing(tmp_path):
    """the per-kernel HBM bytes bench.py makes of rocprofv3's counter CSVs: one row per launch and counter, KiB, FETCH_SIZE doubled, the
    step's sum over its kernels (the fused hand-out replaces the first batch's two launches)"""
    sys.path.insert(0, ROOT)
    import bench
    hdr = "Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp\n"

    def row(kernel, counter, value):
        return f'1,1,1,1,1,1,64,1,"{kernel}",256,0,0,60,0,88,{counter},{value},0,1\n'
    names = {"lean": "void k_samples_lean<false, 4>(SigParams, int)", "ev10": "void k_part_events<1, 0>(SigParams, int, unsigned int)",
             "ev01": "void k_part_events<0, 1>(SigParams, int, unsigned int)", "hc": "void k_part_hand_count<1, 0>(unsigned int const*, unsigned int*)",
             "hist": "k_part_hist(unsigned int const*, unsigned int const*)", "probe": "k_store_probe(HIP_vector_type<unsigned int, 4u>*, unsigned long, unsigned int)",
             "other": "k_sample_try(GenomeParams, unsigned int const*)"}
    f = tmp_path / "fetch_size_counter_collection.csv"
    f.write_text(hdr + row(names["lean"], "FETCH_SIZE", 1000) + row(names["lean"], "FETCH_SIZE", 3000) + row(names["ev10"], "FETCH_SIZE", 10)
                 + row(names["ev01"], "FETCH_SIZE", 100) + row(names["hc"], "FETCH_SIZE", 200) + row(names["hist"], "FETCH_SIZE", 300)
                 + row(names["other"], "FETCH_SIZE", 7) + row(names["lean"], "SQ_WAVES", 5))
    w = tmp_path / "write_size_counter_collection.csv"
    w.write_text(hdr + row(names["lean"], "WRITE_SIZE", 4000) + row(names["lean"], "WRITE_SIZE", 4000) + row(names["ev10"], "WRITE_SIZE", 20)
                 + row(names["ev01"], "WRITE_SIZE", 1500) + row(names["hc"], "WRITE_SIZE", 1000) + row(names["hist"], "WRITE_SIZE", 100)
                 + row(names["probe"], "WRITE_SIZE", 1048576) + row(names["other"], "WRITE_SIZE", 9))
    fe, wr = bench.parse_pmc_csv(str(f), "FETCH_SIZE"), bench.parse_pmc_csv(str(w), "WRITE_SIZE")
    assert fe["k_samples_lean"] == [1000.0, 3000.0] and "k_part_events<1, 0>" in fe and wr["k_store_probe"] == [1048576.0]
    ks = bench.traffic_from_pmc(fe, wr)
    assert ks["k_samples_lean"]["hbm_bytes_per_launch"] == (2 * 2000 + 4000) * 1024 and ks["k_samples_lean"]["launches"] == 2
    assert "k_store_probe" not in ks                                       # (no FETCH_SIZE row: the fetch pass runs without the probe)
    # a step: lean + scatter pass + fused hand-out + hist; not the first batch's counting pass, not the sampler
    want = sum(ks[k]["hbm_bytes_per_launch"] for k in ("k_samples_lean", "k_part_events<0, 1>", "k_part_hand_count", "k_part_hist"))
    assert bench.step_traffic_of(ks) == want
    assert bench.under_profiler() is False


def test_resources_are_priced_from_the_profile_of_the_same_sources(tmp_path, monkeypatch):
    """roofline.resources: the PMC counters of profiles/traffic_latest.json (hash-matched) priced with the run's own kernel time"""
    sys.path.insert(0, ROOT)
    import bench
    from squigulator_amd import build
    kk = {"hbm_bytes_per_launch": 6.9e9, "WRITE_SIZE_KiB": 4.2e6, "FETCH_SIZE_KiB": 1.2e6, "SQ_ACTIVE_INST_VALU": 1.2e9, "SQ_INSTS_VALU": 9.6e8,
          "GRBM_GUI_ACTIVE": 4.0e7, "TCP_TCC_READ_REQ_sum": 1.9e8, "TCP_TCC_WRITE_REQ_sum": 1.0e8, "kernel_us": 2400.0}
    doc = {"workload_key": "w", "source_hash": build.source_hash(), "kernels": {"k_samples_lean": kk}, "calib": {"rgather8_l2_req_per_cycle": 100.0}}
    (tmp_path / "profiles").mkdir()
    (tmp_path / "profiles" / "traffic_latest.json").write_text(json.dumps(doc))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    res = bench.pmc_resources("w", 2.4, 2.13e9, 5000.0)
    assert res["valu_busy"] == pytest.approx(1.2e9 * 4 / 1024 / 5.0e6) and res["valu_inst_per_sample"] == pytest.approx(9.6e8 * 64 / 2.13e9)
    assert res["l2_req_frac"] == pytest.approx(2.9e8 / 5.0e6 / 100.0) and res["store_frac"] == pytest.approx(4.2e6 * 1024 / 2.4e-3 / 1e9 / 5000.0)
    assert res["hbm_frac"] == pytest.approx(6.9e9 / 2.4e-3 / 8e12) and res["bound"] == "valu"
    # the kernel at single issue: 4 cycles per VALU wave-instruction and SIMD + a second pass per transcendental (3 per 64 samples), at the counters' clock
    clk = 5.0e6 / 2400.0e3
    want_ms = (9.6e8 * 4 + 3 * (2.13e9 / 64) * 4) / 1024 / (clk * 1e9) * 1e3
    assert res["valu_single_issue_ms"] == pytest.approx(want_ms) and res["kernel_ms_over_valu_single_issue"] == pytest.approx(2.4 / want_ms)
    assert bench.pmc_resources("other workload", 2.4, 2.13e9, 5000.0) is None
    doc["source_hash"] = "0" * 16
    (tmp_path / "profiles" / "traffic_latest.json").write_text(json.dumps(doc))
    assert bench.pmc_resources("w", 2.4, 2.13e9, 5000.0) is None


@pytest.mark.gpu
def test_bench_refuses_an_unnamed_library(tmp_path):
    """SQG_LIB in the environment redirects the Python binding (A/B builds); bench.py must not time a library it does not name"""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["SQG_LIB"] = os.path.join(ROOT, "oracle", "libsqg_cpu.so")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--genome-mb", "8",
                        "--no-cpu-baseline", "--no-store-probe"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0 and "--lib" in (p.stderr + p.stdout)


def test_traffic_file_is_only_quoted_for_the_sources_it_was_measured_on(tmp_path, monkeypatch):
    """profiles/traffic_latest.json carries a hash of csrc/ + include/sqg.h; bench.py nulls `traffic` when it does not match"""
    sys.path.insert(0, ROOT)
    import bench
    from squigulator_amd import build
    doc = {"workload_key": "w", "source_hash": build.source_hash(), "kernels": {"k_samples_lean": {"hbm_bytes_per_launch": 5.0},
                                                                                 "k_part_hist": {"hbm_bytes_per_launch": 2.0}}}
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "profiles" / "traffic_latest.json").write_text(json.dumps(doc))
    assert bench.pmc_traffic("w") == 5.0 and bench.pmc_step_traffic("w") == 7.0
    assert bench.pmc_traffic("other") is None
    doc["source_hash"] = "0" * 16
    (tmp_path / "profiles" / "traffic_latest.json").write_text(json.dumps(doc))
    assert bench.pmc_traffic("w") is None and bench.pmc_step_traffic("w") is None


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["--genome-mb", 24, "--job-workers", 1, "--batch-reads", 1024],
                                  ["--genome-mb", 24, "--workers-per-gpu", 1, "--batch-reads", 1024],
                                  ["--workload", "ncov-r9", "--job-workers", 3, "--batch-reads", 600]],
                         ids=["r10_range", "r10_worker", "r9_range_t3"])
def test_rccl_runs_with_one_rank(args):
    """RCCL (torch.distributed backend nccl) on the one GPU of the box: a single rank forced through init_process_group runs the
    pore-model broadcast, the barriers and reductions and -- range sharding -- the per-batch all-gather of the device-resident
    stream counts (the __cuda_array_interface__ tensor over sqg_batch_run_begin's pointer) through RCCL.  Results: the plain run's."""
    plain = _bench("--gpus", 1, "--digest", 4, *args)
    rccl = _bench("--gpus", 1, "--digest", 4, "--force-dist", "--backend", "nccl", *args)
    assert plain["ranks"]["backend"] is None and rccl["ranks"]["backend"] == "nccl" and rccl["ranks"]["world_size"] == 1
    assert plain["digest"] == rccl["digest"]


@pytest.mark.gpu
@pytest.mark.parametrize("workload,profile,regime", [("sequin-rna004", "rna004-prom", "-t 1"), ("ncov-r9", "dna-r9-prom", "T = K")])
def test_bench_other_workloads(workload, profile, regime):
    """BASELINE.json configs[4] (rnasequin -x rna004-prom --prefix=yes, whole transcripts, one worker per GPU) and configs[1] (nCoV
    -x dna-r9-prom, T = K) through bench.py itself: the line, the roofline of the dominant kernel, the streaming leg, and the
    parity check of the timed regime against the oracle"""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "3", "--warmup", "1",
                        "--batch-reads", "2048", "--cpu-seconds", "1", "--pipeline-seconds", "0.2", "--e2e-seconds", "0", "--small-batch-seconds", "0"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert profile in d["config"]["workload"] and d["value"] > 0 and d["pipeline"]["value"] > 0
    assert d["roofline"]["kernel"] == "k_samples_lean" and 0 < d["roofline"]["frac"] < 1
    assert d["parity_check"]["equal"] is True and d["parity_check"]["regime"] == regime
    assert d["cpu_baseline"]["value"] > 0


@pytest.mark.gpu
def test_two_ranks_run_the_streaming_leg_and_report_their_clocks():
    """N > 1 without --digest: the timed steps, then the streaming leg on every rank (the same number of batches everywhere: range
    sharding has a collective per batch), reductions over the ranks, `ranks` and `pipeline` in rank 0's line"""
    for extra in (["--workers-per-gpu", 1], ["--job-workers", 1]):
        d = _bench("--gpus", 2, "--backend", "gloo", "--genome-mb", 24, "--batch-reads", 512, "--pipeline-seconds", 0.3, *extra)
        assert d["n_gpus"] == 2 and d["ranks"]["world_size"] == 2 and d["ranks"]["backend"] == "gloo"
        assert d["ranks"]["ms_per_step_min"] <= d["ranks"]["ms_per_step_max"] == pytest.approx(d["ms_per_step"], rel=1e-9)
        pl = d["pipeline"]
        assert pl["value"] > 0 and pl["batches_per_gpu"] >= 9 and pl["reads_per_s"] > 0
        # both ranks' samples are in the whole-job numbers
        assert d["value"] * d["ms_per_step"] * 1e-3 > d["samples_per_step_per_gpu"]
