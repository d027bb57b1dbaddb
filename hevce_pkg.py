"""Import helper: the package directory is named hevc-image-encoder-lite_amd (dashes), so load it by path."""
import importlib.util
import os
import sys

_NAME = "hevc_image_encoder_lite_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    root = os.path.dirname(os.path.abspath(__file__))
    pkg = os.path.join(root, "hevc-image-encoder-lite_amd")
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(pkg, "__init__.py"), submodule_search_locations=[pkg])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
