"""The streaming step in its f16x2 form (StreamBatch(precision="f16x2"), pf_stream_set_option("gemm_mode", 3): every GEMM of
the step on the fp16 matrix cores with two-plane operands, fp32 results; the throughput path of many lock-step streams)
against the REFERENCE's ParaformerStreaming sessions, the reference-pinned streaming oracle and the default fp32 step.
The cases live in tests/_stream_f16x2_cases.py (also runnable one process each: `python tests/_stream_f16x2_cases.py <case>`,
how they first met the hardware, tools/gpu_shot_r03v.sh)."""
import pytest

from tests._stream_f16x2_cases import CASES, build, load

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(CASES))
def test_streaming_f16x2_case(cuda, name):
    CASES[name](cuda)


def test_default_precision_follows_the_stream_count(cuda):
    """precision=None: the fp32 step for a few streams (latency), the f16x2 step from AUTO_F16X2_MIN_STREAMS on (throughput)"""
    from funasr_amd.paraformer_streaming import StreamBatch
    g, cfg, sd = load()
    model = build(cfg, sd, cuda)
    few = StreamBatch(model, 2, [0, 10, 5], 4, 1)
    many = StreamBatch(model, StreamBatch.AUTO_F16X2_MIN_STREAMS, [0, 10, 5], 4, 1)
    assert few.precision == "fp32" and many.precision == "f16x2"
    with pytest.raises(ValueError):
        StreamBatch(model, 1, [0, 10, 5], 4, 1, precision="bf16")
    few.close()
    many.close()
