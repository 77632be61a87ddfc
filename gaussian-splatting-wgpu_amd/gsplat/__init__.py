"""gsplat: Python host mirror of the reference's Renderer / Camera / PackedGaussians surface over the
MI355X-native C ABI (include/gsplat/gs_abi.h).  The JavaScript host that drops in behind the
reference's TypeScript is in ../js; this package exists for tests, bench.py and torch.distributed."""
from . import _abi, camera, synth  # noqa: F401
from .camera import Camera  # noqa: F401
from .renderer import Canvas, InteractiveCamera, PackedGaussians, PipelinedRenderer, Renderer  # noqa: F401
