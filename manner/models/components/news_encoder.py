"""reference manner/models/components/news_encoder.py:11-171 -> the HIP mirror classes."""
from manner_amd.models.components.news_encoder import (MannerEntityEncoder, MannerNewsEncoder,  # noqa: F401
                                                        MannerTextEncoder, PLMTextEncoder)
