"""AdditiveAttention — mirror of reference manner/models/components/attention.py:6-29."""
import torch
import torch.nn as nn

from manner_amd import hip, train


class AdditiveAttention(nn.Module):
    def __init__(self, input_dim: int, query_dim: int) -> None:
        super().__init__()
        self.linear = nn.Linear(input_dim, query_dim)
        self.query = nn.Parameter(torch.empty(query_dim).uniform_(-0.1, 0.1))

    def forward(self, input_vector: torch.Tensor) -> torch.Tensor:
        """(batch, seq, dim) -> (batch, dim); unmasked softmax over ``seq`` as in the reference."""
        # autograd records this forward whenever the reference's torch ops would (train() or eval(): there is no dropout here)
        if torch.is_grad_enabled() and (input_vector.requires_grad or any(p.requires_grad for p in self.parameters())):
            return train.additive_pool(input_vector, self.linear.weight, self.linear.bias, self.query)      # with its backward
        return hip.additive_pool(input_vector, self.linear.weight.detach(), self.linear.bias.detach(),
                                 self.query.detach())
