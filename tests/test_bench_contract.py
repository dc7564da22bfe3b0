"""The bench line's contract, checked offline on the committed line of the round (profiles/r05_final_bench.json): every key the driver
reads, the roofline and cpu_baseline objects, and the rule that `value` is the HBM-resident one-stream rate (the PCIe-inclusive and
two-replica figures sit beside it). Also: the profile / trace condensers run on synthetic rocprofv3 CSVs."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    with open(os.path.join(ROOT, "profiles", "r05_final_bench.json")) as f:
        rows = [ln for ln in f if ln.startswith("{")]
    assert len(rows) == 1, "bench.py prints ONE JSON line"
    return json.loads(rows[0])


def test_committed_line_has_every_contract_key():
    d = _line()
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["unit"] == "audio-s/s" and ("audio" in str(base.get("metric", "")).lower() or "rtf" in str(base.get("metric", "")).lower())
    # value = clips x seconds x steps / time
    secs = d["config"]["clips_per_gpu"] * d["config"]["clip_seconds"]
    assert abs(d["value"] - secs / (d["ms_per_step"] * 1e-3)) / d["value"] < 2e-3
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.0 < r["frac"] < 1.0
    assert r["frac_of_power_limited_peak"] > r["frac"]          # the measured ceiling is below the data-sheet one
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["unit"] == d["unit"] and c["cores"] >= 1 and c["token_ids_match_gpu"] is True
    assert any(s.get("batch_size") == 8 for s in c["thread_settings"])


def test_value_is_the_resident_one_stream_rate_and_the_variants_sit_beside_it():
    d = _line()
    assert d["value_pcie_inclusive"] == d["pcie_inclusive"]["value"] and "excluded" in d["config"]["h2d"]
    assert d["two_replicas"]["replicas"] == 2 and d["two_replicas"]["ids_equal_main"] is True
    assert d["two_replicas"]["value"] != d["value"] and d["pcie_inclusive"]["value"] != d["value"]
    sv = d["sensevoice"]
    assert sv["cpu_oracle_clips_checked"] == 128 and sv["clips_differing_from_cpu_oracle"] == [] and sv["ids_equal_cpu_oracle"] is True
    p = d["hbm_copy_probe"]
    assert p["GBps_read_plus_write"] > 1000 and p["own_read_GBps"] > 1000 and p["own_L2_reread_GBps"] > p["own_read_GBps"]
    st = d["cpu_baseline"]["full_config_parity"]["cif_margin_statistic"]
    assert st["clips"] >= 512 and st["clips_with_different_token_count"] == 0


def _trace_csv(path, rows):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kind", "Agent_Id", "Queue_Id", "Kernel_Id", "Kernel_Name", "Correlation_Id", "Start_Timestamp", "End_Timestamp"])
        for name, s, e in rows:
            w.writerow(["KERNEL_DISPATCH", 1, 1, 1, name, 1, s, e])


def test_stream_trace_analysis_on_a_synthetic_trace(tmp_path):
    rows, t = [], 1_000_000
    for _step in range(4):
        for k in range(6):                                     # a step: the marker kernel, then five 7-us kernels, all abutting
            name = "void pf::(anonymous namespace)::stream_embed_kernel(pf::StreamEmbedArgs)" if k == 0 else "void pf::(anonymous namespace)::gemm_skinny_kernel<1, 1, 0>(pf::GemmArgs)"
            d = 5_000 if k == 0 else 7_000
            rows.append((name, t, t + d))
            t += d
        t += 500_000                                           # host gap between steps
    f = str(tmp_path / "t_kernel_trace.csv")
    _trace_csv(f, rows)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "analyze_stream_trace.py"), f, "--idle-us", "100"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    assert out["kernels_per_step"] == 6 and out["steps_analysed"] == 4
    assert out["wall_us_median"] == 40.0 and out["kernel_time_us_median"] == 40.0 and out["gap_time_us_median"] == 0.0
    assert out["per_kernel"]["pf::gemm_skinny_kernel<1, 1, 0>"] == {"per_step": 5.0, "avg_us": 7.0, "us_per_step": 35.0}
    assert out["per_kernel"]["pf::stream_embed_kernel"]["avg_us"] == 5.0
