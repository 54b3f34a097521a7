"""sylber_amd — MI355X-native implementation of the SYLBER ``Segmenter`` forward path
(reference API: sylber/__init__.py:1 exports ``Segmenter``)."""
from .segmenter import Segmenter, HubertEncoderHIP  # noqa: F401

__all__ = ["Segmenter", "HubertEncoderHIP"]
