from .clip import *  # noqa: F401,F403
from . import clip, model  # noqa: F401
