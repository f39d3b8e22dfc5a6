"""Programmatic equivalent of example_configs/speech2text/ds2_large_8gpus.py (DeepSpeech2
large: 2 x conv2d+BN+ReLU, 5 bidirectional cuDNN GRU layers of 800 units, dense 1600,
FC-CTC; Momentum + exp_decay + LARC; batch 16 per GPU)."""
from ..data.speech2text.speech2text import Speech2TextDataLayer
from ..decoders.fc_decoders import FullyConnectedCTCDecoder
from ..encoders.ds2_encoder import DeepSpeech2Encoder
from ..losses.ctc_loss import CTCLoss
from ..models.speech2text import Speech2Text
from ..optimizers.lr_policies import poly_decay


def ds2_large_config(batch_size_per_gpu=16, max_steps=100000):
  base_params = {
      "random_seed": 0, "use_horovod": True, "batch_size_per_gpu": batch_size_per_gpu,
      "max_steps": max_steps,
      "optimizer": "Momentum", "optimizer_params": {"momentum": 0.90},
      "lr_policy": poly_decay,
      "lr_policy_params": {"learning_rate": 0.001, "power": 0.5},
      "larc_params": {"larc_eta": 0.001},
      "dtype": "mixed", "loss_scaling": "Backoff",
      "regularizer": "l2_regularizer", "regularizer_params": {"scale": 0.0005},
      "encoder": DeepSpeech2Encoder,
      "encoder_params": {
          "conv_layers": [
              {"kernel_size": [11, 41], "stride": [2, 2], "num_channels": 32, "padding": "SAME"},
              {"kernel_size": [11, 21], "stride": [1, 2], "num_channels": 32, "padding": "SAME"},
          ],
          "num_rnn_layers": 5, "rnn_cell_dim": 800, "use_cudnn_rnn": True,
          "rnn_type": "cudnn_gru", "rnn_unidirectional": False, "row_conv": False,
          "n_hidden": 1600, "dropout_keep_prob": 0.5, "activation_fn": "relu",
          "data_format": "channels_first",
      },
      "decoder": FullyConnectedCTCDecoder,
      "decoder_params": {"use_language_model": False},
      "loss": CTCLoss, "loss_params": {},
      "data_layer": Speech2TextDataLayer,
      "data_layer_params": {
          "num_audio_features": 160, "input_type": "spectrogram", "vocab_file": None,
          "dataset_files": [], "max_duration": 16.7, "shuffle": True, "pad_to": 8,
      },
  }
  return Speech2Text, base_params
