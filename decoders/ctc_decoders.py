"""`from ctc_decoders import Scorer, ctc_beam_search_decoder_batch, ...` of the reference's scripts
(scripts/decode.py:11-13, scripts/ctc_decoders_test.py:5) resolves here when `decoders/` is on
the path, as it is for the reference's swig build (decoders/ctc_decoders.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openseq2seq_amd.ctc_decoders import (Scorer, ctc_beam_search_decoder,  # noqa: E402,F401
                                          ctc_beam_search_decoder_batch, ctc_greedy_decoder)
