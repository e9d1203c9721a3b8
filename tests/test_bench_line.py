"""bench.py's printed line: compact enough for the driver (an 8 KB stdout tail, last JSON line parsed), complete enough
for the measurement row (`roofline` and `cpu_baseline` present), built from a canned full record (round 5's own)."""
import copy
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _canned():
    with open(os.path.join(ROOT, "profiles", "r05_bench_final.json")) as fh:
        return json.load(fh)


def test_compact_line_fits_and_carries_the_contract():
    bench = _bench()
    full = _canned()
    assert len(json.dumps(full)) > 20000                       # the record that the driver could not parse in round 5
    text = bench.compact_line(full, "gpurun_out/bench_full.json")
    assert "\n" not in text and len(text) <= 6000
    line = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert key in line, key
    assert line["metric"] == full["metric"] and line["config"]["name"] == "c234" and "workload" in line["config"]
    assert abs(line["value"] / full["value"] - 1) < 1e-4 and abs(line["ms_per_step"] / full["ms_per_step"] - 1) < 1e-4
    r = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_kernel_ms", "algorithmic_bytes_per_launch", "kernels"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert {k["codec"] for k in r["kernels"]} == {"mp3", "vorbis", "flac"} and all("avg_kernel_ms" in k and "frac" in k for k in r["kernels"])
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 16 and c["value"] > 0 and c["unit"] == "samples/s" and c["sample"]
    assert all(p["mismatches"] == 0 for p in line["parity"].values()) and set(line["parity"]) == {"mp3", "vorbis", "flac"}
    ow = line["other_workloads"]
    assert ow["c5"]["value"] > 0 and ow["flac_e2e"]["value"] > 0 and ow["flac_e2e"]["cpu"] > 0
    assert len(ow["vorbis_shapes"]["shapes"]) == len(full["other_workloads"]["vorbis_shapes"]["shapes"])


def test_compact_line_of_the_round_6_record():
    bench = _bench()
    with open(os.path.join(ROOT, "profiles", "r06_bench_final.json")) as fh:
        full = json.load(fh)
    text = bench.compact_line(full, "gpurun_out/bench_full.json")
    assert len(text) <= 6000
    line = json.loads(text)
    assert line["roofline"]["kernel"] == "flac_restore1_kernel" and line["roofline"]["traffic"] and (line["roofline"]["traffic_from"].startswith("profiles/r06_pmc_") or line["roofline"]["traffic_from"].startswith("measured before this run"))
    assert line["cpu_baseline"]["cores"] >= 1 and line["other_workloads"]["flac_e2e"]["at_cpu_quota"] > 0
    assert len(line["other_workloads"]["flac_shapes"]["shapes"]) == 10


def test_compact_line_sheds_detail_rather_than_overflow():
    bench = _bench()
    full = _canned()
    shapes = full["other_workloads"]["vorbis_shapes"]["shapes"]
    full["other_workloads"]["vorbis_shapes"]["shapes"] = shapes * 40          # a leg that grew
    for i in range(60):
        full["other_workloads"][f"extra_{i}_e2e"] = copy.deepcopy(full["other_workloads"]["flac_e2e"])
    text = bench.compact_line(full, None)
    assert len(text) <= 6000
    line = json.loads(text)
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0 and line["value"] > 0


def test_compact_line_without_optional_blocks():
    bench = _bench()
    full = _canned()
    full.pop("other_workloads")
    full["cpu_baseline"] = None
    full["roofline"]["traffic"] = None
    line = json.loads(bench.compact_line(full))
    assert line["cpu_baseline"] is None and line["roofline"]["traffic"] is None and "other_workloads" not in line


def test_notes_land_on_their_own_codec():
    """Round 5 stamped FLAC's int16-row note on the Vorbis entry; the source must gate it by codec."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    i = src.index('kernels[-1]["input_rows"] = "int16 residual rows')
    assert 'if name == "flac":' in src[i - 700:i]
