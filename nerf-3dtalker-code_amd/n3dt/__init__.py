"""n3dt: MI355X-native volumetric head rendering behind NeRF-3DTalker's HeadNeRFNet interface."""
from .options import BaseOptions  # noqa: F401
from .headnerf import HeadNeRFNet, NeuralRenderer, MLPforNeRF  # noqa: F401
from . import checkpoint, render_utils, parallel, train, fitting  # noqa: F401,E402
