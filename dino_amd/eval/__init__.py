"""Evaluation-side helpers that sit on the accelerated backbone: eval-mode
feature extraction with the cross-rank gather of dino/eval/features.py.  The Tuner, its plugins and the
PathoROB benchmark themselves are out of scope."""
from .features import extract_features, extract_multiple_features, gather_rows  # noqa: F401
