"""The bench line's contract: (i) the functions bench.py assembles the line with, on synthetic measurements (schema, value and roofline
arithmetic, the flagged fallback of the power-limited peak); (ii) as a separate sanity check, a committed line of an earlier round
(profiles/r06_final_bench.json): every key the driver reads, the roofline and cpu_baseline objects, `value` as the HBM-resident one-stream
rate with the PCIe-inclusive and two-replica figures beside it. Also: the profile / trace condensers run on synthetic rocprofv3 CSVs."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    with open(os.path.join(ROOT, "profiles", "r06_final_bench.json")) as f:
        rows = [ln for ln in f if ln.startswith("{")]
    assert len(rows) == 1, "bench.py prints ONE JSON line"
    return json.loads(rows[0])


def _import_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_line_assembly_schema_on_synthetic_inputs():
    """what bench.py EMITS (bench.assemble_roofline / assemble_line / apply_power_limited_peak), on made-up measurements: every key
    the driver reads, `value` = clips x seconds x steps / time, frac = achieved / peak, and a power-limited peak that was not measured
    in the run never becomes a fraction of the run"""
    b = _import_bench()
    gemm = {"work_per_step": 11.67e12, "ms_per_step": 41.2, "launches_per_step": 283}
    roof = b.assemble_roofline(gemm, "gemm_f16x2_", 2500.0 / 3, 52.4, (4.0e8, "profiles/x_traffic.json"), 3)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and abs(roof["achieved"] - 11.67e12 / 41.2e-3 / 1e12) < 0.01
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and roof["traffic"] == 4.0e8 and roof["launches_per_step"] == 283
    steps, dt = 20, 1.048
    line = b.assemble_line(value=64 * 30.0 * steps / dt, dt=dt, steps=steps, warmup=5, world=1, dtype="f32 (...)", B=64, seconds=30.0,
                           token_num=[170, 168, 171], arena_bytes=0, n_pad=512, per_rank_ms=None, bcast_all=None, weights_route="arena",
                           host_pin=None, roofline=roof, kernels={}, telemetry={"sclk_mhz_mean": 2000.0, "power_w_mean": 1200.0},
                           output_layer="random-init")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 5 and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["vs_baseline"] is None and line["data"] == "synthetic" and "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - 64 * 30.0 / (line["ms_per_step"] * 1e-3)) / line["value"] < 2e-3
    assert line["config"]["hypothesis_gather_bytes_per_rank_per_step"] == 0 and line["config"]["weights_route"] is None
    json.dumps(line)                                            # serialisable as it stands
    multi = b.assemble_line(value=1.0, dt=1.0, steps=2, warmup=1, world=8, dtype="x", B=16, seconds=10.0, token_num=[5], arena_bytes=880_000_000,
                            n_pad=512, per_rank_ms=[1.0] * 8, bcast_all=[0.5] * 8, weights_route="arena", host_pin={"cores": [0, 1], "threads": 2},
                            roofline=roof, kernels={}, telemetry=None, output_layer="random-init")
    assert multi["config"]["rccl_ranks"] == 8 and multi["config"]["hypothesis_gather_bytes_per_rank_per_step"] == 16 * 513 * 4
    assert multi["config"]["weights_route"] == "arena" and multi["sclk_mhz_mean"] is None
    live = b.apply_power_limited_peak(dict(roof), {"TFLOPs": 1650.0, "source": "live"}, 3)
    assert abs(live["frac_of_power_limited_peak"] - roof["achieved"] / 550.0) < 1e-3 and live["frac_of_power_limited_peak"] > roof["frac"]
    rec = b.apply_power_limited_peak(dict(roof), {"TFLOPs": None, "recorded_TFLOPs": 1647.0, "source": "NOT measured in this run; recorded: ..."}, 3)
    assert rec["frac_of_power_limited_peak"] is None and rec["power_limited_peak"] is None and rec["power_limited_peak_recorded"] == 549.0


def test_committed_line_sanity_of_the_recorded_artefact():
    """a sanity check of a RECORDED artefact (not of bench.py: the schema test above is that)"""
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
    # round 6: the statistic goes down to token ids, and the line says how its loop ran
    ids = st["token_ids"]
    assert ids["confident_output_layer"]["clips_with_different_ids"] == 0 and ids["random_output_layer"]["different_ids_among_fire_mismatch_clips"] == 0
    il = d["interleave"]
    assert "second HIP stream" in il["loop"] and il["sequential_ms_per_step"] > d["ms_per_step"] and il["one_stream_ms_per_step"] > d["ms_per_step"]
    assert "ONE stream" in d["roofline"]["measured_in"]


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
