"""reference manner/models/components/click_predictors.py:5-12 -> the HIP mirror class."""
from manner_amd.models.components.click_predictors import DotProduct  # noqa: F401
