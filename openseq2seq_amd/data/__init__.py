from .data_layer import DataLayer
from .speech2text.speech2text import Speech2TextDataLayer
from .text2text.text2text import ParallelTextDataLayer
from .text2speech.text2speech import Text2SpeechDataLayer
