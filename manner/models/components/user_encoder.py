"""reference manner/models/components/user_encoder.py:9-21 -> the HIP mirror class."""
from manner_amd.models.components.user_encoder import NAMLUserEncoder  # noqa: F401
