"""DotProduct — mirror of reference manner/models/components/click_predictors.py:5-12."""
import torch
import torch.nn as nn

from manner_amd import hip, train


class DotProduct(nn.Module):
    def __init__(self) -> None:
        super().__init__()

    def forward(self, clicked_news_vector: torch.Tensor, candidate_news_vector: torch.Tensor) -> torch.Tensor:
        # [B,1,D] x [B,D,C] -> [B,C]; the permuted view the reference passes is read in place
        if torch.is_grad_enabled() and (clicked_news_vector.requires_grad or candidate_news_vector.requires_grad):
            return train.dot(clicked_news_vector, candidate_news_vector)         # training: with the bmm's backward
        return hip.dot(clicked_news_vector, candidate_news_vector)
