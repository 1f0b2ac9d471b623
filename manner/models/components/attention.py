"""reference manner/models/components/attention.py:6-29 -> the HIP mirror class."""
from manner_amd.models.components.attention import AdditiveAttention  # noqa: F401
