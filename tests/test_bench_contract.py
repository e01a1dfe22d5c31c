"""The bench line contract (metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better /
scaling / vs_baseline / dtype / data / config + roofline + cpu_baseline), checked on the committed line of the
last GPU run and on bench.py's helpers (no GPU needed)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _latest(pattern):
    import glob
    return sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))[-1]


def test_committed_bench_line_follows_the_contract():
    line = json.load(open(_latest("bench_r*_config1.json")))
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(line[k], t), k
    assert line["vs_baseline"] is None and line["scaling"] == "weak" and line["dtype"] == "f32"
    assert "workload" in line["config"] and "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0
    c = line["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    if "loops" in c:        # round 2 on: the reference-faithful Python-loop LocalPadder variant beside the vectorised port
        assert c["loops"]["kind"] == "port-loops" and 0 < c["loops"]["value"] <= c["value"]
    if "membound" in r:     # achieved GB/s of the memory-bound operators against 8 TB/s
        timed = [m for m in r["membound"] if "gbps" in m]           # HIP-event rows (graph-replayed launches)
        assert timed and all(0 < m["frac_of_8tbps"] < 1 and abs(m["gbps"] - m["mbytes"] / m["us"] * 1e3) / m["gbps"] < 2e-2 for m in timed)
        # round 6: the same launches over a > 600 MB working set (past the Infinity Cache), and rocprof-derived in-step rows
        cold = [m for m in r["membound"] if "gbps_hbm" in m]
        assert all(0 < m["frac_of_8tbps_hbm"] < 1 for m in cold)
        assert all(m["gbps_hbm"] <= m["gbps"] * 1.05 for m in cold if "gbps" in m)      # HBM cannot beat the cache-assisted replay
    if "step_hbm" in r:     # round 6: the whole iteration against the HBM roofline, and the bound the two fractions name
        h = r["step_hbm"]
        assert h["unit"] == "GB/s" and h["peak"] == 8000.0 and abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-3
        assert h["bound"] in ("mfma", "hbm", "launch") and h["algorithmic_bytes_per_step"] > 0 and h["abi_calls_per_step"] > 0
        assert abs(h["achieved"] - h["algorithmic_bytes_per_step"] / (line["ms_per_step"] * 1e-3) / 1e9) / h["achieved"] < 1e-2
    # value is consistent with the step time: batch 8 per GPU
    assert abs(line["value"] - 8 * line["n_gpus"] / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-3


def test_bench_helpers():
    b = _bench()
    assert b.NECESSARY_GF_PER_STEP == 855.5 and b.FP32_MFMA_PEAK_TF == 157.3
    assert 1 <= b.host_cores() <= 16
    t = json.load(open(_latest("r*_hbm_traffic.json")))
    k = json.load(open(_latest("bench_r*_config1.json")))["roofline"]["kernel"]
    # the traffic summary is only used when it was measured on the kernel sources that are checked out (its _meta hash):
    # a summary of another build yields None plus a reason, never a stale number
    val, note = b.hbm_traffic(k)
    if t.get("_meta", {}).get("kernel_source_sha16") == b.kernel_source_hash():
        assert k in t and val == int(t[k]["fetch_bytes"] + t[k]["write_bytes"]) and b.kernel_source_hash() in note
    else:
        assert val is None and "withheld" in note
    assert b.hbm_traffic("no_such_kernel")[0] is None
    assert len(b.kernel_source_hash()) == 16
    a = b.U_FLAGS if hasattr(b, "U_FLAGS") else b.FLAGS
    assert "--n_layers_G" in a and a[a.index("--n_layers_G") + 1] == "6" and a[a.index("--random_crop") + 1] == "192"
