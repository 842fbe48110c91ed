"""Host-side mirror of torchmultimodal/modules/layers/transformer.py:21-27 (the output record every encoder returns)."""
from typing import List, NamedTuple, Optional, Tuple

from torch import Tensor


class TransformerOutput(NamedTuple):
    last_hidden_state: Optional[Tensor] = None
    pooler_output: Optional[Tensor] = None
    hidden_states: Optional[List[Tensor]] = None
    attentions: Optional[List[Tensor]] = None
    image_labels: Optional[Tensor] = None
    current_key_values: Optional[List[Tuple[Tensor, Tensor]]] = None
