from .loss import Loss
from .ctc_loss import CTCLoss
from .sequence_loss import PaddedCrossEntropyLossWithSmoothing
from .sequence_loss import BasicSequenceLoss
