"""Batch types of the hot path — same keys and meaning as reference manner/data/components/mind_batch.py:6-17."""
from typing import Any, Dict, Optional, TypedDict

import torch


class MINDRecBatch(TypedDict):
    batch_hist: torch.Tensor        # int64 [sum h_i]  sorted segment ids
    batch_cand: torch.Tensor        # int64 [sum c_i]
    x_hist: Dict[str, Any]          # {"text": {"input_ids", "attention_mask"}, "entities", "category", "sentiment", "sentiment_score"}
    x_cand: Dict[str, Any]
    labels: Optional[torch.Tensor]  # float32 [sum c_i]
    users: torch.Tensor             # int64 [B]


class MINDNewsBatch(TypedDict):
    news: Dict[str, Any]
    labels: torch.Tensor
