"""NAMLUserEncoder — mirror of reference manner/models/components/user_encoder.py:9-21
(imported by the reference as ``UserEncoder``, cr_module.py:16)."""
import torch
import torch.nn as nn

from manner_amd.models.components.attention import AdditiveAttention


class NAMLUserEncoder(nn.Module):
    def __init__(self, news_embedding_dim: int, query_vector_dim: int) -> None:
        super().__init__()
        self.additive_attention = AdditiveAttention(input_dim=news_embedding_dim, query_dim=query_vector_dim)

    def forward(self, clicked_news_vector: torch.Tensor) -> torch.Tensor:
        # batch_size, num_clicked_news_per_user, news_embedding_dim -> batch_size, news_embedding_dim
        return self.additive_attention(clicked_news_vector)
