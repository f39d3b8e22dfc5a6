# Jasper 10x5 Dense Residual on synthetic batches (same network and optimizer as the
# reference's jasper10x5_LibriSpeech_nvgrad_masks.py; no dataset files needed).
#   python run.py --config_file=example_configs/speech2text/jasper10x5_synthetic.py \
#       --mode=train --benchmark --bench_steps=30
from open_seq2seq.configs.jasper import jasper10x5_config

base_model, base_params = jasper10x5_config(batch_size_per_gpu=32, use_horovod=True)
base_params["print_loss_steps"] = 10
