"""Two options of the mirrored classes raise at construction instead of running: CifPredictorV3(upsample_type="cnn_attn") and
ContextualParaformer(bias_encoder_type="mean"). This file pins WHY that is parity and not a gap: the reference's own classes
cannot run these options at inference either -- `cnn_attn` hands the T-frame mask to an attention over the 3T upsampled
frames (bicif_paraformer/cif_predictor.py:251,327: size mismatch in masked_fill), and `mean` builds no `bias_encoder` although
`cal_decoder_with_predictor` calls it unconditionally (contextual_paraformer/model.py:81-82,357,371). Build container only
(imports the reference from /root/reference through oracle/ref_import.py)."""
import pytest
import torch

from oracle import ref_import

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference checkout not present (GPU box)")


def test_reference_cnn_attn_upsampler_cannot_run_with_a_mask():
    ref_import.install()
    from funasr.models.bicif_paraformer.cif_predictor import CifPredictorV3
    p = CifPredictorV3(idim=64, l_order=1, r_order=1, threshold=1.0, upsample_times=3, use_cif1_cnn=False,
                       upsample_type="cnn_attn", tail_threshold=0.45).eval()
    hidden = torch.randn(2, 20, 64)
    mask = torch.ones(2, 1, 20)
    mask[1, 0, 15:] = 0
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="must match the size"):
            p(hidden, mask=mask)                                            # forward, :251
        with pytest.raises(RuntimeError, match="must match the size"):
            p.get_upsample_timestamp(hidden, mask, torch.tensor([5, 4]))      # the inference call of BiCifParaformer, :327
    # the working variant of the same class runs on the same inputs
    ok = CifPredictorV3(idim=64, l_order=1, r_order=1, threshold=1.0, upsample_times=3, use_cif1_cnn=False,
                        upsample_type="cnn_blstm", tail_threshold=0.45).eval()
    with torch.no_grad():
        assert ok.get_upsample_timestamp(hidden, mask, torch.tensor([5, 4]))[2].shape == (2, 60)


def test_reference_mean_bias_encoder_has_no_inference_path():
    """read off the reference source (building the whole ContextualParaformer needs its full config): the `mean` branch of
    __init__ creates only `bias_embed`, and both hotword branches of cal_decoder_with_predictor call `self.bias_encoder`"""
    import inspect
    ref_import.install()
    from funasr.models.contextual_paraformer.model import ContextualParaformer
    init_src = inspect.getsource(ContextualParaformer.__init__)
    mean_branch = init_src.split('bias_encoder_type == "mean"')[1].split("else:")[0]
    assert "self.bias_embed" in mean_branch and "self.bias_encoder" not in mean_branch
    dec_src = inspect.getsource(ContextualParaformer.cal_decoder_with_predictor)
    assert dec_src.count("self.bias_encoder(") == 2 and "bias_encoder_type" not in dec_src


def test_our_classes_refuse_both_options_at_construction():
    from funasr_amd.cif_predictor import CifPredictorV3
    with pytest.raises(NotImplementedError, match="cnn_attn"):
        CifPredictorV3(idim=64, l_order=1, r_order=1, upsample_times=3, upsample_type="cnn_attn")
