"""reference manner/models/components/user_encoder.py:9-42 -> the HIP mirror classes."""
from manner_amd.models.components.user_encoder import NAMLUserEncoder, NRMSUserEncoder  # noqa: F401
