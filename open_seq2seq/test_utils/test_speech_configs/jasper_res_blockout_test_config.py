"""Toy residual TDNN with stochastic block dropping (drop_block_prob 0.2) on the ten-utterance toy corpus.

Same parameter values as the reference's open_seq2seq/test_utils/test_speech_configs/jasper_res_blockout_test_config.py (the
configuration of its block-dropout runs, scripts/run_all_tests.sh:71-83), restated; tests/test_config_dropin.py compares
the loaded dictionaries with the reference's file when the reference checkout is present. Runs from the
repository root, as in the reference: the toy corpus lives under open_seq2seq/test_utils/toy_speech_data/.
"""
import tensorflow as tf
from open_seq2seq.models import Speech2Text
from open_seq2seq.encoders import TDNNEncoder
from open_seq2seq.decoders import FullyConnectedCTCDecoder
from open_seq2seq.data import Speech2TextDataLayer
from open_seq2seq.losses import CTCLoss
from open_seq2seq.optimizers.lr_policies import poly_decay

TOY = "open_seq2seq/test_utils/toy_speech_data/"


def _toy_data(shuffle):
  return {"data_layer": Speech2TextDataLayer,
          "data_layer_params": {"num_audio_features": 40, "input_type": "logfbank",
                                "vocab_file": TOY + "vocab.txt", "dataset_files": [TOY + "toy_data.csv"],
                                "shuffle": shuffle}}


def _conv(**kw):
  return dict(type="conv1d", stride=[1], padding="SAME", dilation=[1], **kw)


base_model = Speech2Text
train_params = _toy_data(True)
eval_params = _toy_data(False)

base_params = dict(
    use_horovod=False, num_gpus=1, batch_size_per_gpu=10, num_epochs=500,
    save_summaries_steps=10, print_loss_steps=10, print_samples_steps=20, eval_steps=50,
    save_checkpoint_steps=50, logdir="tmp_log_folder",
    optimizer="Momentum", optimizer_params={"momentum": 0.90},
    lr_policy=poly_decay, lr_policy_params={"learning_rate": 0.01, "power": 2},
    larc_params={"larc_eta": 0.001}, dtype=tf.float32,
    summaries=["learning_rate", "variables", "gradients", "larc_summaries", "variable_norm",
               "gradient_norm", "global_gradient_norm"],
    encoder=TDNNEncoder,
    encoder_params={
        "convnet_layers": [_conv(repeat=1, kernel_size=[7], num_channels=128),
                           _conv(repeat=2, kernel_size=[7], num_channels=256, residual=True),
                           _conv(repeat=2, kernel_size=[1], num_channels=256, residual=True)],
        "drop_block_prob": 0.2, "drop_block_index": -1,
        "dropout_keep_prob": 0.9,
        "initializer": tf.contrib.layers.xavier_initializer, "initializer_params": {"uniform": False},
        "activation_fn": lambda x: tf.minimum(tf.nn.relu(x), 20.0),      # ReLU clipped at 20
        "normalization": "batch_norm", "data_format": "channels_last",
    },
    decoder=FullyConnectedCTCDecoder,
    decoder_params={"initializer": tf.contrib.layers.xavier_initializer, "use_language_model": False},
    loss=CTCLoss, loss_params={},
)
