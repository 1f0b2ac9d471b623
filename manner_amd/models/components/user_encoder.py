"""NAMLUserEncoder / NRMSUserEncoder — mirror of reference manner/models/components/user_encoder.py:9-42
(NAML: imported by the reference as ``UserEncoder``, cr_module.py:16; NRMS: the user encoder of the PLM baselines)."""
import torch
import torch.nn as nn

from manner_amd import hip, train
from manner_amd.models.components.attention import AdditiveAttention


class NAMLUserEncoder(nn.Module):
    def __init__(self, news_embedding_dim: int, query_vector_dim: int) -> None:
        super().__init__()
        self.additive_attention = AdditiveAttention(input_dim=news_embedding_dim, query_dim=query_vector_dim)

    def forward(self, clicked_news_vector: torch.Tensor) -> torch.Tensor:
        # batch_size, num_clicked_news_per_user, news_embedding_dim -> batch_size, news_embedding_dim
        return self.additive_attention(clicked_news_vector)


class NRMSUserEncoder(nn.Module):
    """reference user_encoder.py:24-42.  Batch-faithful: the un-masked batch_first=False MultiheadAttention sees the dense
    [B, Hmax, D] history and so attends ACROSS THE USERS OF THE BATCH at each history slot (zero-padded slots included);
    the additive pooler then runs over all Hmax slots.  With grad mode on and anything that requires grad (train() or eval():
    the module has no dropout) the differentiable operators of manner_amd.train run — in-projection, axis-0 attention,
    out-projection and pooler, each with its hand-written backward (csrc/train_small.hip) — as
    baselines/nrms_plm_module.py:119-135 trains it; otherwise the inference kernels."""

    def __init__(self, news_embedding_dim: int, num_attention_heads: int, query_vector_dim: int) -> None:
        super().__init__()
        self.multihead_attention = nn.MultiheadAttention(news_embedding_dim, num_attention_heads)
        self.additive_attention = AdditiveAttention(news_embedding_dim, query_vector_dim)

    def forward(self, clicked_news_vector: torch.Tensor) -> torch.Tensor:
        mha = self.multihead_attention
        if torch.is_grad_enabled() and (clicked_news_vector.requires_grad or any(p.requires_grad for p in self.parameters())):
            if mha.dropout != 0.0:
                raise RuntimeError("attention-probability dropout inside nn.MultiheadAttention is not built (the reference uses 0)")
            user_vector = train.mha_axis0(clicked_news_vector, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias,
                                          mha.num_heads)
            return self.additive_attention(user_vector)
        user_vector = hip.mha_axis0(clicked_news_vector, mha.in_proj_weight.detach(), mha.in_proj_bias.detach(),
                                    mha.out_proj.weight.detach(), mha.out_proj.bias.detach(), mha.num_heads)
        return self.additive_attention(user_vector)
