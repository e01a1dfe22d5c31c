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
        assert all(0 < m["frac_of_8tbps"] < 1 and abs(m["gbps"] - m["mbytes"] / m["us"] * 1e3) / m["gbps"] < 2e-2 for m in r["membound"])
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
