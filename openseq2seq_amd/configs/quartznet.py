"""Programmatic equivalent of example_configs/speech2text/quartznet15x5_LibriSpeech.py
(QuartzNet 15x5: time-channel separable convolutions — 1 + 15x5 + 1 "sep_conv1d" layers and a
1x1 conv, 19 M parameters; NovoGrad beta2 0.5 + cosine decay with warm-up). The reference
file itself also loads unchanged through load_config_module."""
from ..data.speech2text.speech2text import Speech2TextDataLayer
from ..decoders.fc_decoders import FullyConnectedCTCDecoder
from ..encoders.tdnn_encoder import TDNNEncoder
from ..losses.ctc_loss import CTCLoss
from ..models.speech2text import Speech2Text
from ..optimizers.lr_policies import cosine_decay
from ..optimizers.novograd import NovoGrad

_BLOCKS = [(33, 256)] * 3 + [(39, 256)] * 3 + [(51, 512)] * 3 + [(63, 512)] * 3 + [(75, 512)] * 3


def quartznet_convnet_layers(blocks=_BLOCKS, repeat=5):
  def layer(t, k, c, rep=1, stride=1, dil=1, **extra):
    d = {"type": t, "repeat": rep, "kernel_size": [k], "stride": [stride], "num_channels": c,
         "padding": "SAME", "dilation": [dil]}
    d.update(extra)
    return d
  layers = [layer("sep_conv1d", 33, 256, stride=2)]
  for k, c in blocks:
    layers.append(layer("sep_conv1d", k, c, rep=repeat, residual=True, residual_dense=False))
  layers.append(layer("sep_conv1d", 87, 512, dil=2, residual=True, residual_dense=False))
  layers.append(layer("conv1d", 1, 1024))
  return layers


def quartznet15x5_config(batch_size_per_gpu=32, use_horovod=True, max_steps=100000, vocab_file=None):
  base_params = {
      "random_seed": 0, "use_horovod": use_horovod, "batch_size_per_gpu": batch_size_per_gpu,
      "iter_size": 1, "max_steps": max_steps,
      "optimizer": NovoGrad,
      "optimizer_params": {"beta1": 0.95, "beta2": 0.5, "epsilon": 1e-08, "weight_decay": 0.001,
                           "grad_averaging": False},
      "lr_policy": cosine_decay,
      "lr_policy_params": {"learning_rate": 0.01, "min_lr": 0.0, "warmup_steps": 1000},
      "dtype": "mixed", "loss_scaling": "Backoff",
      "encoder": TDNNEncoder,
      "encoder_params": {
          "convnet_layers": quartznet_convnet_layers(), "dropout_keep_prob": 1.0,
          "initializer": "xavier_initializer", "initializer_params": {"uniform": False},
          "normalization": "batch_norm", "activation_fn": "relu", "data_format": "channels_last",
          "use_conv_mask": True,
      },
      "decoder": FullyConnectedCTCDecoder,
      "decoder_params": {"initializer": "xavier_initializer", "use_language_model": False,
                         "infer_logits_to_pickle": False},
      "loss": CTCLoss, "loss_params": {},
      "data_layer": Speech2TextDataLayer,
      "data_layer_params": {
          "num_audio_features": 64, "input_type": "logfbank", "vocab_file": vocab_file,
          "norm_per_feature": True, "window": "hanning", "precompute_mel_basis": True,
          "sample_freq": 16000, "pad_to": 16, "dither": 1e-5, "backend": "librosa",
          "dataset_files": [], "max_duration": 16.7, "shuffle": True,
      },
  }
  return Speech2Text, base_params
