"""Weight-side conversion helpers: what runs once, before the hot path."""
from .smooth import smooth_ln_fcs  # noqa: F401
