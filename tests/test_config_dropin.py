"""CPU: the reference's own example configs load UNCHANGED through the same runpy path
(utils/utils.py:521) and resolve to the MI355X-native plugin classes; CLI overrides of
nested leaves work as in the reference (utils.py:535-543)."""
import os

import pytest

REF = "/root/reference/example_configs"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")


def test_jasper_config_loads_and_maps():
  from openseq2seq_amd.utils.utils import get_base_config
  cfg = os.path.join(REF, "speech2text", "jasper10x5_LibriSpeech_nvgrad_masks.py")
  args, base, model_cls, mod = get_base_config(
      ["--config_file=" + cfg, "--mode=train", "--benchmark", "--bench_steps=7",
       "--batch_size_per_gpu=4", "--lr_policy_params/learning_rate=0.01"])
  import openseq2seq_amd as impl
  from openseq2seq_amd.models.speech2text import Speech2Text
  from openseq2seq_amd.encoders.tdnn_encoder import TDNNEncoder
  from openseq2seq_amd.optimizers.novograd import NovoGrad
  from openseq2seq_amd.configs.jasper import jasper_convnet_layers
  assert model_cls is Speech2Text and base["encoder"] is TDNNEncoder and base["optimizer"] is NovoGrad
  assert base["batch_size_per_gpu"] == 4 and base["lr_policy_params"]["learning_rate"] == 0.01
  assert args.bench_steps == 7
  # the programmatic config used by bench.py is the same network
  assert base["encoder_params"]["convnet_layers"] == jasper_convnet_layers()
  enc = TDNNEncoder(base["encoder_params"], None, mode="train")      # check_params passes
  assert enc.params["activation_fn"].__name__ == "relu"
  assert "tensorflow" not in __import__("sys").modules or \
      getattr(__import__("sys").modules["tensorflow"], "__version__", "").endswith("shim") is False


def test_transformer_big_config_loads_and_maps():
  from openseq2seq_amd.utils.utils import get_base_config
  cfg = os.path.join(REF, "text2text", "en-de", "transformer-big.py")
  _, base, model_cls, mod = get_base_config(["--config_file=" + cfg, "--mode=train"])
  from openseq2seq_amd.models.text2text import Text2Text
  from openseq2seq_amd.encoders.transformer_encoder import TransformerEncoder
  from openseq2seq_amd.decoders.transformer_decoder import TransformerDecoder
  from openseq2seq_amd.losses.sequence_loss import PaddedCrossEntropyLossWithSmoothing
  from openseq2seq_amd.optimizers.optimizers import _optimizer_id
  assert model_cls is Text2Text and base["encoder"] is TransformerEncoder
  assert base["decoder"] is TransformerDecoder and base["loss"] is PaddedCrossEntropyLossWithSmoothing
  assert _optimizer_id(base["optimizer"]) == 3                      # LazyAdam -> Adam kernel
  assert base["lr_policy"].__name__ == "transformer_policy"
  assert mod["train_params"]["data_layer_params"]["max_length"] == 56
  p = dict(base["encoder_params"], src_vocab_size=32768)
  TransformerEncoder(p, None, mode="train")                          # schema accepted


def test_unknown_param_is_rejected():
  from openseq2seq_amd.encoders.tdnn_encoder import TDNNEncoder
  with pytest.raises(ValueError):
    TDNNEncoder({"dropout_keep_prob": 0.5, "convnet_layers": [], "activation_fn": None,
                 "bogus": 1}, None)


def test_baseline_configs_load_unchanged():
  """The example configs BASELINE.json names (plus their infer / eval variants) load through the
  reference's own runpy path and resolve to classes of this package."""
  from openseq2seq_amd.utils.utils import get_base_config
  import openseq2seq_amd.decoders as D
  import openseq2seq_amd.encoders as E
  import openseq2seq_amd.models as M
  cases = {
      "text2text/en-de/en-de-nmt-small.py": (M.Text2Text, E.BidirectionalRNNEncoderWithEmbedding,
                                             D.RNNDecoderWithAttention),
      "speech2text/ds2_large_8gpus.py": (M.Speech2Text, E.DeepSpeech2Encoder, D.FullyConnectedCTCDecoder),
      "speech2text/quartznet15x5_LibriSpeech.py": (M.Speech2Text, E.TDNNEncoder, D.FullyConnectedCTCDecoder),
      "text2speech/tacotron_gst.py": (M.Text2SpeechTacotron, E.Tacotron2Encoder, D.Tacotron2Decoder),
      "text2text/en-de/transformer-base.py": (M.Text2Text, E.TransformerEncoder, D.TransformerDecoder),
      "text2text/en-de/en-de-gnmt-like-4GPUs.py": (M.Text2Text, E.GNMTLikeEncoderWithEmbedding,
                                                   D.RNNDecoderWithAttention),
  }
  for rel, (model, enc, dec) in cases.items():
    for mode in ("train", "infer"):
      _, base, model_cls, mod = get_base_config(["--config_file=" + os.path.join(REF, rel), "--mode=" + mode])
      assert model_cls is model and base["encoder"] is enc and base["decoder"] is dec, (rel, mode)
  _, _, _, mod = get_base_config(["--config_file=" + os.path.join(REF, "text2text/en-de/en-de-nmt-small.py"),
                                  "--mode=infer"])
  assert mod["infer_params"]["decoder"] is D.BeamSearchRNNDecoderWithAttention
  assert mod["infer_params"]["decoder_params"]["beam_width"] == 10


def test_toy_speech_test_configs_equal_the_reference():
  """open_seq2seq/test_utils/test_speech_configs/*.py of THIS repository restate the reference's acceptance
  configurations (DS2 / W2L convergence tests, block-dropout runs): same dictionaries, value for value (the
  clipped-ReLU lambda is compared by what it resolves to)."""
  from openseq2seq_amd.utils.utils import get_base_config
  from openseq2seq_amd.parts.cnns.conv_blocks import act_id
  repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

  def norm(x):
    if isinstance(x, dict):
      return {k: norm(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
      return [norm(v) for v in x]
    if callable(x) and getattr(x, "__name__", "") == "<lambda>":
      return "activation id %d" % act_id(x)
    if callable(x) or hasattr(x, "__name__"):
      return getattr(x, "__name__", str(x))
    return x

  cwd = os.getcwd()
  os.chdir(repo)
  try:
    for n in ("ds2", "w2l", "jasper_res_blockout"):
      rel = "open_seq2seq/test_utils/test_speech_configs/%s_test_config.py" % n
      ours = get_base_config(["--config_file=" + rel, "--mode=train"])
      ref = get_base_config(["--config_file=" + os.path.join("/root/reference", rel), "--mode=train"])
      assert norm(ours[1]) == norm(ref[1]), n
      for key in ("train_params", "eval_params"):
        assert norm(ours[3][key]) == norm(ref[3][key]), (n, key)
      assert norm(ours[1])["encoder_params"]["activation_fn"] == "activation id 3"      # min(relu(x), 20)
  finally:
    os.chdir(cwd)


def test_check_params_reports_problems_in_the_reference_order():
  """open_seq2seq/utils/utils.py:403-429 walks the required table (absent -> 'has to be specified', wrong kind ->
  'has to be of type'), then the kinds of the optional entries, unknown keys LAST: with two mistakes in one
  config both code bases must name the same one (ours named the unknown key first)."""
  import pytest
  from openseq2seq_amd.utils.utils import check_params
  req, opt = {"a": int, "b": str}, {"c": float, "d": ["x", "y"]}
  with pytest.raises(ValueError, match="a has to be of type"):
    check_params({"a": "1", "b": "s", "zzz": 1}, req, opt)           # wrong kind beats unknown key
  with pytest.raises(ValueError, match="d has to be one of"):
    check_params({"a": 1, "b": "s", "d": "q", "zzz": 1}, req, opt)
  with pytest.raises(ValueError, match="b parameter has to be specified"):
    check_params({"a": 1, "zzz": 1}, req, opt)
  with pytest.raises(ValueError, match="a has to be of type"):
    check_params({"a": "1"}, req, opt)                               # kind of `a` before the absence of `b`
  with pytest.raises(ValueError, match="Unknown parameter: zzz"):
    check_params({"a": 1, "b": "s", "c": 2, "zzz": 1}, req, opt)     # an int where a float is asked for passes
  check_params({"a": 1, "b": u"s", "d": "y"}, req, opt)
