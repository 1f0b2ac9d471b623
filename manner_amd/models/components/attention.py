"""AdditiveAttention — mirror of reference manner/models/components/attention.py:6-29."""
import torch
import torch.nn as nn

from manner_amd import hip


class AdditiveAttention(nn.Module):
    def __init__(self, input_dim: int, query_dim: int) -> None:
        super().__init__()
        self.linear = nn.Linear(input_dim, query_dim)
        self.query = nn.Parameter(torch.empty(query_dim).uniform_(-0.1, 0.1))

    def forward(self, input_vector: torch.Tensor) -> torch.Tensor:
        """(batch, seq, dim) -> (batch, dim); unmasked softmax over ``seq`` as in the reference."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and self.training:
            raise RuntimeError("manner_amd AdditiveAttention is inference-only (call .eval() / torch.no_grad())")
        return hip.additive_pool(input_vector, self.linear.weight.detach(), self.linear.bias.detach(),
                                 self.query.detach())
