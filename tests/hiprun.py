"""The scene -> drop-in packages -> C ABI runner lives in the product tree (gsrast.runner); tests import it under the old name."""
from gsrast.runner import *  # noqa: F401,F403
from gsrast.runner import VID  # noqa: F401
