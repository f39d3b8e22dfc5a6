"""Programmatic equivalent of the reference config
example_configs/speech2text/jasper10x5_LibriSpeech_nvgrad_masks.py (Jasper 10x5
Dense Residual: 1 + 10x5 + 2 conv layers, 332.6 M parameters). The reference file
itself also loads unchanged through openseq2seq_amd.utils.config.load_config; this
builder exists so bench.py does not need /root/reference on the GPU box."""
from ..data.speech2text.speech2text import Speech2TextDataLayer
from ..decoders.fc_decoders import FullyConnectedCTCDecoder
from ..encoders.tdnn_encoder import TDNNEncoder
from ..losses.ctc_loss import CTCLoss
from ..models.speech2text import Speech2Text
from ..optimizers.lr_policies import poly_decay
from ..optimizers.novograd import NovoGrad

# (kernel, channels, keep_prob) of the ten residual blocks (each repeated 5x)
_BLOCKS = [(11, 256, 0.8), (11, 256, 0.8), (13, 384, 0.8), (13, 384, 0.8), (17, 512, 0.8),
           (17, 512, 0.8), (21, 640, 0.7), (21, 640, 0.7), (25, 768, 0.7), (25, 768, 0.7)]


def jasper_convnet_layers(blocks=_BLOCKS, repeat=5, residual_dense=True, first_channels=256):
  def layer(k, c, keep, rep=1, stride=1, dil=1, **extra):
    d = {"type": "conv1d", "repeat": rep, "kernel_size": [k], "stride": [stride],
         "num_channels": c, "padding": "SAME", "dilation": [dil], "dropout_keep_prob": keep}
    d.update(extra)
    return d
  layers = [layer(11, first_channels, 0.8, stride=2)]
  for k, c, keep in blocks:
    layers.append(layer(k, c, keep, rep=repeat, residual=True, residual_dense=residual_dense))
  layers.append(layer(29, 896, 0.6, dil=2))
  layers.append(layer(1, 1024, 0.6))
  return layers


def jasper10x5_config(batch_size_per_gpu=32, use_horovod=True, max_steps=None, vocab_file=None):
  base_params = {
      "random_seed": 0,
      "use_horovod": use_horovod,
      "batch_size_per_gpu": batch_size_per_gpu,
      "iter_size": 1,
      "optimizer": NovoGrad,
      "optimizer_params": {"beta1": 0.95, "beta2": 0.98, "epsilon": 1e-08,
                           "weight_decay": 0.001, "grad_averaging": False},
      "lr_policy": poly_decay,
      "lr_policy_params": {"learning_rate": 0.02, "min_lr": 1e-5, "power": 2.0},
      "larc_params": {"larc_eta": 0.001},
      "dtype": "mixed",
      "loss_scaling": "Backoff",
      "encoder": TDNNEncoder,
      "encoder_params": {
          "convnet_layers": jasper_convnet_layers(),
          "dropout_keep_prob": 0.7,
          "initializer": "xavier_initializer",
          "initializer_params": {"uniform": False},
          "normalization": "batch_norm",
          "activation_fn": "relu",
          "data_format": "channels_last",
          "use_conv_mask": True,
      },
      "decoder": FullyConnectedCTCDecoder,
      "decoder_params": {"initializer": "xavier_initializer", "use_language_model": False,
                         "infer_logits_to_pickle": False},
      "loss": CTCLoss,
      "loss_params": {},
      "data_layer": Speech2TextDataLayer,
      "data_layer_params": {
          "num_audio_features": 64, "input_type": "logfbank", "vocab_file": vocab_file,
          "norm_per_feature": True, "window": "hanning", "precompute_mel_basis": True,
          "sample_freq": 16000, "pad_to": 16, "dither": 1e-5, "backend": "librosa",
          "dataset_files": [], "max_duration": 16.7, "shuffle": True,
      },
  }
  if max_steps is not None:
    base_params["max_steps"] = max_steps
  else:
    base_params["num_epochs"] = 400
  return Speech2Text, base_params
