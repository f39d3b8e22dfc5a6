"""Programmatic equivalent of example_configs/text2text/en-de/en-de-nmt-small.py
(2-layer bidirectional LSTM-512 encoder with embedding, 2-layer GNMT-v2 attention decoder,
BasicSequenceLoss, Adam 1e-3 + LARC, batch 128) on synthetic token batches of the config's
shape (32 k BPE vocabulary, max_length 50)."""
from ..data.text2text.text2text import ParallelTextDataLayer
from ..decoders.rnn_decoders import RNNDecoderWithAttention
from ..encoders.rnn_encoders import BidirectionalRNNEncoderWithEmbedding
from ..losses.sequence_loss import BasicSequenceLoss
from ..models.text2text import Text2Text
from ..optimizers.lr_policies import fixed_lr


def nmt_small_config(batch_size_per_gpu=128, max_steps=100000, vocab=32768):
  cell = {"num_units": 512, "forget_bias": 1.0}
  base_params = {
      "random_seed": 0, "use_horovod": True, "batch_size_per_gpu": batch_size_per_gpu,
      "max_steps": max_steps,
      "optimizer": "Adam", "optimizer_params": {},
      "lr_policy": fixed_lr, "lr_policy_params": {"learning_rate": 0.001},
      "larc_params": {"larc_eta": 0.001},
      "dtype": "mixed", "loss_scaling": "Backoff",
      "encoder": BidirectionalRNNEncoderWithEmbedding,
      "encoder_params": {
          "core_cell": "LSTMCell", "core_cell_params": cell, "encoder_layers": 2,
          "encoder_dp_input_keep_prob": 0.8, "encoder_dp_output_keep_prob": 1.0,
          "encoder_use_skip_connections": False, "src_emb_size": 512, "use_swap_memory": True,
      },
      "decoder": RNNDecoderWithAttention,
      "decoder_params": {
          "core_cell": "LSTMCell", "core_cell_params": cell, "decoder_layers": 2,
          "decoder_dp_input_keep_prob": 0.8, "decoder_dp_output_keep_prob": 1.0,
          "decoder_use_skip_connections": False, "GO_SYMBOL": 2, "END_SYMBOL": 1,
          "tgt_emb_size": 512, "attention_type": "gnmt_v2", "attention_layer_size": 512,
          "use_swap_memory": True,
      },
      "loss": BasicSequenceLoss,
      "loss_params": {"offset_target_by_one": True, "average_across_timestep": False, "do_mask": True},
      "data_layer": ParallelTextDataLayer,
      "data_layer_params": {
          "src_vocab_file": None, "tgt_vocab_file": None, "source_file": "", "target_file": "",
          "delimiter": " ", "shuffle": True, "repeat": True, "max_length": 50,
          "synthetic_vocab_size": vocab,
      },
  }
  return Text2Text, base_params
