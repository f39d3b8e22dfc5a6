"""Programmatic equivalent of example_configs/text2text/en-de/transformer-big.py
(Transformer-big: d_model 1024, 16 heads, filter 4096, 6+6 layers, shared 32k vocabulary,
LazyAdam + transformer_policy, mixed precision with Backoff loss scaling)."""
from ..data.text2text.text2text import ParallelTextDataLayer, SpecialTextTokens
from ..decoders.transformer_decoder import TransformerDecoder
from ..encoders.transformer_encoder import TransformerEncoder
from ..losses.sequence_loss import PaddedCrossEntropyLossWithSmoothing
from ..models.text2text import Text2Text
from ..optimizers.lr_policies import transformer_policy


def transformer_config(d_model=1024, num_layers=6, num_heads=16, batch_size_per_gpu=256,
                       vocab_size=32768, max_length=56, max_steps=300000):
  base_params = {
      "use_horovod": True,
      "batch_size_per_gpu": batch_size_per_gpu,
      "max_steps": max_steps,
      "dtype": "mixed",
      "loss_scaling": "Backoff",
      "optimizer": "LazyAdam",
      "optimizer_params": {"beta1": 0.9, "beta2": 0.997, "epsilon": 1e-09},
      "lr_policy": transformer_policy,
      "lr_policy_params": {"learning_rate": 2.0, "warmup_steps": 8000, "d_model": d_model},
      "encoder": TransformerEncoder,
      "encoder_params": {
          "encoder_layers": num_layers, "hidden_size": d_model, "num_heads": num_heads,
          "attention_dropout": 0.1, "filter_size": 4 * d_model, "relu_dropout": 0.3,
          "layer_postprocess_dropout": 0.3, "pad_embeddings_2_eight": True,
          "remove_padding": True,
      },
      "decoder": TransformerDecoder,
      "decoder_params": {
          "layer_postprocess_dropout": 0.3, "num_hidden_layers": num_layers,
          "hidden_size": d_model, "num_heads": num_heads, "attention_dropout": 0.1,
          "relu_dropout": 0.3, "filter_size": 4 * d_model, "beam_size": 4, "alpha": 0.6,
          "extra_decode_length": 50, "EOS_ID": SpecialTextTokens.EOS_ID.value,
          "GO_SYMBOL": SpecialTextTokens.S_ID.value, "END_SYMBOL": SpecialTextTokens.EOS_ID.value,
          "PAD_SYMBOL": SpecialTextTokens.PAD_ID.value,
      },
      "loss": PaddedCrossEntropyLossWithSmoothing,
      "loss_params": {"label_smoothing": 0.1},
      "data_layer": ParallelTextDataLayer,
      "data_layer_params": {
          "pad_vocab_to_eight": True, "src_vocab_file": None, "tgt_vocab_file": None,
          "source_file": "", "target_file": "", "delimiter": " ", "shuffle": True,
          "repeat": True, "max_length": max_length, "synthetic_vocab_size": vocab_size,
      },
  }
  return Text2Text, base_params
