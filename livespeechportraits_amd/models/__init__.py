"""``from livespeechportraits_amd.models import create_model`` -- the reference's factory
(models/__init__.py:29-71) for the models this package replaces: feature2face (the renderer) and
audio2headpose and audio2feature (rank 4)."""
from __future__ import annotations

import importlib

from ..base_model import BaseModel


def find_model_using_name(model_name: str):
    if model_name not in ("feature2face", "audio2headpose", "audio2feature"):
        raise NotImplementedError("livespeechportraits_amd replaces --model feature2face, audio2headpose and audio2feature; "
                                  "%r stays with the reference implementation" % (model_name,))
    lib = importlib.import_module("livespeechportraits_amd.%s_model" % model_name)
    target = model_name.replace("_", "") + "model"
    for name, cls in vars(lib).items():
        if name.lower() == target and isinstance(cls, type) and issubclass(cls, BaseModel):
            return cls
    raise ImportError("no BaseModel subclass named like %s" % target)


def create_model(opt):
    instance = find_model_using_name(opt.model)(opt)
    print("model [%s] was created" % type(instance).__name__)
    return instance
