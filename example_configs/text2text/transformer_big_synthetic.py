# Transformer-big (reference: example_configs/text2text/en-de/transformer-big.py) on
# synthetic token batches.
from open_seq2seq.configs.transformer import transformer_config

base_model, base_params = transformer_config()
base_params["print_loss_steps"] = 10
