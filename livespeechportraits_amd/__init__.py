"""MI355X-native feature2face renderer for LiveSpeechPortraits (hot path only).

Public surface mirrors the reference: ``create_model(opt)`` ->
``Feature2FaceModel`` with ``setup`` / ``eval`` / ``inference``.
"""
from .models import create_model  # noqa: F401
from .feature2face_model import Feature2FaceModel  # noqa: F401
from .engine import Engine  # noqa: F401

__all__ = ["create_model", "Feature2FaceModel", "Engine"]
