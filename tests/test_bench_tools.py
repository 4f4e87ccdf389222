"""bench.py's own data tools (CPU): the synthetic tokenised cache `bench.py --full` writes is the reference's cache format
(utils/util.py:257-307) as `ance_amd.cache.TokenCache` reads it, block boundaries included, and deterministic."""
import hashlib
import os

import numpy as np


def test_synthetic_cache_round_trip(tmp_path):
    import bench
    from ance_amd.cache import TokenCache
    p = str(tmp_path / "passages")
    secs, mean_len = bench.write_synthetic_cache(p, 5000, 128, 70.0, 0.45, 8, seed=1, block=2048)  # 3 blocks, last one short
    c = TokenCache(p)
    assert len(c) == 5000 and c.embedding_size == 128 and c.record_size == 4 + 4 * 128
    assert os.path.getsize(p) == 5000 * c.record_size
    L, ids = c.lengths(), c.ids()
    assert L.min() >= 8 and L.max() <= 128 and abs(float(L.mean()) - mean_len) < 1e-9
    rows = np.arange(5000)
    assert np.all(ids[:, 0] == 0) and np.all(ids[rows, L - 1] == 2)          # <s> ... </s>
    assert np.all((np.arange(128)[None, :] < L[:, None]) | (ids == 1))       # pad = 1 after the length
    inner = (np.arange(128)[None, :] > 0) & (np.arange(128)[None, :] < (L - 1)[:, None])
    assert np.all(ids[inner] >= 3) and np.all(ids[inner] < 50265)
    plen, tok = c[4999]                                                       # the reference's __getitem__ contract
    assert plen == int(L[4999]) and np.array_equal(tok, ids[4999])
    # deterministic in (seed, block)
    p2 = str(tmp_path / "again")
    bench.write_synthetic_cache(p2, 5000, 128, 70.0, 0.45, 8, seed=1, block=2048)
    assert hashlib.sha256(open(p, "rb").read()).digest() == hashlib.sha256(open(p2, "rb").read()).digest()


def test_bench_records_match_cache_layout():
    import bench
    rng = np.random.default_rng(0)
    rec, lens = bench.synthetic_records(rng, 64, 32)
    assert rec.shape == (64, 33) and rec.dtype == np.int32
    assert np.array_equal(rec[:, 0].view(">u4").astype(np.int64).reshape(-1), lens.astype(np.int64))  # big-endian header


def test_gpus_flag_refuses_to_measure_fewer_devices(tmp_path):
    """`bench.py --gpus N` starts N ranks itself; with fewer visible GPUs than ranks it must fail loudly instead of
    printing a one-rank line (the flag used to be parsed and never read)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2 and "visible GPU" in r.stderr and r.stdout.strip() == ""
    # a launcher that started a different number of ranks than --gpus says is refused as well
    env["WORLD_SIZE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


def test_roofline_evidence_helpers_quote_only_this_round(tmp_path, monkeypatch):
    """bench.py reads counters (profiles/pmc_traffic.json) and kernel-trace averages (profiles/rNN_*.csv) into the JSON line:
    an entry of another round must not be quoted (stale round-1 counters were printed under round-2 kernels),
    and frac_from_profiles comes from the CSV row of the named kernel."""
    import json
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "pmc_traffic.json").write_text(json.dumps({"encode": {
        "gemm_ffn1": {"round": bench.CURRENT_ROUND, "hbm_bytes_per_launch": 1.0},
        "gemm_qk": {"round": "r01", "hbm_bytes_per_launch": 2.0},
        "gemm_res": {"round": bench.CURRENT_ROUND + " (final)", "hbm_bytes_per_launch": 3.0}}}))
    (prof / ("%s_rocprofv3_encode_single_stream_kernel_stats.csv" % bench.CURRENT_ROUND)).write_text(
        '"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'
        '"void ance::(anonymous namespace)::gemm256_f16_desc_kernel<6>(ance::GemmArgs)",10,3000000,300000.0,50,1,2,3\n'
        '"void ance::(anonymous namespace)::gemm256_f16_desc_kernel<4>(ance::GemmArgs)",20,4000000,200000.0,50,1,2,3\n')
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.pmc_traffic("encode", "gemm_ffn1")["hbm_bytes_per_launch"] == 1.0
    assert bench.pmc_traffic("encode", "gemm_qk") is None                       # another round's counters
    assert bench.pmc_traffic("encode", "gemm_ffn2")["hbm_bytes_per_launch"] == 3.0  # both RES GEMMs are one kernel
    assert bench.pmc_traffic("search", "ip_topk_fast") is None
    name = "%s_rocprofv3_encode_single_stream_kernel_stats.csv" % bench.CURRENT_ROUND
    assert bench.trace_avg_ns(name, bench.KERNEL_OF["gemm_ffn1"]) == 300000.0
    assert bench.trace_avg_ns(name, bench.KERNEL_OF["gemm_attn_out"]) == 200000.0
    assert bench.trace_avg_ns("missing.csv", "x") is None


def test_power_sampler_reads_hwmon_files(tmp_path):
    """bench.PowerSampler: socket power (uW) and shader clock (Hz) from an amdgpu hwmon directory, sampled by a thread while a timed
    region runs; a platform without the files (and without rocm-smi devices) reports None instead of failing the bench."""
    import time
    import bench
    d = tmp_path / "hwmon7"
    d.mkdir()
    (d / "power1_input").write_text("1396000000\n")
    (d / "freq1_input").write_text("1572000000\n")
    (d / "power1_cap").write_text("1400000000\n")
    with bench.PowerSampler(None, 0, hwmon_dirs=[str(d)]) as ps:
        time.sleep(0.7)
    r = ps.report()
    assert r["samples"] >= 2 and r["power_cap_w"] == 1400.0
    assert r["socket_power_w"] == {"min": 1396.0, "mean": 1396.0, "max": 1396.0}
    assert r["sclk_mhz"]["mean"] == 1572.0 and r["measured_in_this_run"] is True
    empty = tmp_path / "hwmon8"
    empty.mkdir()
    ps2 = bench.PowerSampler(None, 0, hwmon_dirs=[str(empty)])
    assert ps2.files is None or ps2.report() is None
